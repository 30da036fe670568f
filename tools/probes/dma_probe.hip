// dma_probe.hip -- how fast can a CU take data in?  Streams bytes into a workgroup (512 threads, 8 waves) by
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, lane-linear)
//   mode 1: global_load_dwordx4 into registers (xor-reduced)
//   mode 2: LDS-DMA with the "two 512-byte runs, lanes alternating" pattern of the S_b / weight-gradient staging
// with `depth` wave-instructions in flight per wave, from
//   src 0: a private stream per workgroup (HBM / Infinity Cache),
//   src 1: one region per XCD (blockIdx % 8) shared by its 32 workgroups (L2 hits),
//   src 2: one region shared by 4 consecutive workgroups of an XCD (the 4 column blocks of a question)
// hipcc --offload-arch=gfx950 -O3 tools/probes/dma_probe.hip -o tools/probes/bin/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16b(const char* g, char* l) {
  const uint32_t a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)l);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(a) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void probe(const char* base, size_t region, size_t wg_stride, int src, int iters, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  size_t r0;
  if (src == 0) r0 = (size_t)blockIdx.x * wg_stride;
  else if (src == 1) r0 = (size_t)(blockIdx.x & 7) * wg_stride;
  else r0 = (size_t)((blockIdx.x & 7) * 64 + (blockIdx.x >> 5)) * wg_stride;       // 4 workgroups (bid>>3 = 4j..4j+3) of one XCD
  const char* p = base + r0;
  u32x4 acc = {0, 0, 0, 0};
  size_t off = (size_t)wave * 1024;
  int loff = lane * 16;
  if (MODE == 2) loff = (lane & 1) * (int)(region / 2) + (lane >> 1) * 16;           // two runs half a region apart
  const size_t wrap = MODE == 2 ? region / 2 : region;
  char* ring = lds + wave * (DEPTH * 1024);
  for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const char* g = p + off + loff;
      if (MODE == 1) {
        u32x4 v = *reinterpret_cast<const u32x4*>(g);
        acc ^= v;
      } else {
        dma16b(g, ring + d * 1024);
      }
      off += (MODE == 2 ? 4096 : 8192);
      if (off >= wrap) off -= wrap;
    }
    if (MODE != 1) wait_vm<DEPTH / 2>();
  }
  if (MODE != 1) { wait_vm<0>(); __syncthreads(); acc = *reinterpret_cast<u32x4*>(lds + tid * 16); }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = 1;
}

template <int MODE, int DEPTH>
double run(const char* base, size_t region, size_t wg_stride, int src, int iters, uint32_t* out) {
  auto k = probe<MODE, DEPTH>;
  const size_t l = 8 * DEPTH * 1024 > 8192 ? 8 * DEPTH * 1024 : 8192;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), l, 0, base, region, wg_stride, src, iters, out);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), l, 0, base, region, wg_stride, src, iters, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * 8 * 1024 * iters;
  return bytes / (ms / reps * 1e-3) / 1e12;   // TB/s aggregate
}

int main() {
  const size_t total = (size_t)3 << 30;
  char* base; uint32_t* out;
  hipMalloc(&base, total); hipMalloc(&out, 64);
  hipMemset(base, 1, total);
  const char* srcn[3] = {"private stream 8 MiB/WG (2 GiB)", "1 MiB per XCD shared by 32 WGs", "2 MiB shared by 4 WGs of an XCD"};
  for (int src = 0; src < 3; ++src) {
    const size_t region = src == 0 ? (8u << 20) : (src == 1 ? (1u << 20) : (2u << 20));
    const size_t stride = region;
    const int iters = 1024;      // x 8 KiB per WG-iteration = 8 MiB per WG
    printf("src %d (%s)\n", src, srcn[src]);
    printf("  dma  lane-linear  depth 2/4/8/16: %.2f %.2f %.2f %.2f TB/s\n", run<0, 2>(base, region, stride, src, iters, out),
           run<0, 4>(base, region, stride, src, iters, out), run<0, 8>(base, region, stride, src, iters, out), run<0, 16>(base, region, stride, src, iters, out));
    printf("  regs lane-linear  depth 2/4/8/16: %.2f %.2f %.2f %.2f TB/s\n", run<1, 2>(base, region, stride, src, iters, out),
           run<1, 4>(base, region, stride, src, iters, out), run<1, 8>(base, region, stride, src, iters, out), run<1, 16>(base, region, stride, src, iters, out));
    printf("  dma  2 x 512 B    depth 2/4/8/16: %.2f %.2f %.2f %.2f TB/s\n", run<2, 2>(base, region, stride, src, iters, out),
           run<2, 4>(base, region, stride, src, iters, out), run<2, 8>(base, region, stride, src, iters, out), run<2, 16>(base, region, stride, src, iters, out));
  }
  return 0;
}
