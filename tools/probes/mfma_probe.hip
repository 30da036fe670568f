// mfma_probe.hip -- what does the fp16 matrix pipe of a SIMD sustain for the instruction streams the chain kernels issue?
//   instr  0: v_mfma_f32_16x16x32_f16   1: v_mfma_f32_32x32x16_f16
//   NACC   independent accumulators a wave cycles through (the distance between two instructions on the same accumulator)
//   waves  per SIMD: 1 (256-thread workgroups) or 2 (512-thread workgroups), one workgroup per CU
//   data   0: zero operands  1: random fp16 operands (the chip clocks to its power budget: MI355X_MICROARCH.md "DVFS give-back")
//   ops    1: every instruction reads fresh A/B registers (8 operand pairs rotated)  0: the same pair
// Prints cycles per instruction per SIMD at the clock s_memtime runs at (100 MHz constant) converted with the measured
// wall time, and the chip-wide TFLOP/s.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_probe.hip -o tools/probes/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int INSTR, int NACC, int NT>
__global__ __launch_bounds__(NT) void probe(const f16x8* src, int iters, float* out) {
  f16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = src[(threadIdx.x + 64 * i) & 1023];
    b[i] = src[(threadIdx.x + 64 * i + 512) & 1023];
  }
  float sum = 0.f;
  if constexpr (INSTR == 0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 48 / NACC; ++r)
#pragma unroll
        for (int i = 0; i < NACC; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(r * NACC + i) & 7], b[(r + i) & 7], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) sum += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
        for (int i = 0; i < NACC; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(r * NACC + i) & 7], b[(r + i) & 7], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) sum += acc[i][0] + acc[i][15];
  }
  if (sum == 12345.678f) out[0] = sum;
}

template <int INSTR, int NACC, int NT>
void run(const f16x8* src, float* out, const char* data) {
  const int iters = 2000;
  auto k = probe<INSTR, NACC, NT>;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(NT), 0, 0, src, 10, out);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(NT), 0, 0, src, iters, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double per_wave = (double)iters * (INSTR == 0 ? 48 : 24);
  const double per_simd = per_wave * (NT / 256);
  const double flop = per_wave * (NT / 64) * 256.0 * 32768.0 / (INSTR == 0 ? 2 : 1);
  printf("%s  %-9s nacc %2d  waves/SIMD %d  %8.3f ms  %6.2f ns per instr per SIMD (= %5.1f cycles at 2.4 GHz)  %7.1f TF\n", data,
         INSTR == 0 ? "16x16x32" : "32x32x16", NACC, NT / 256, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, flop / ms / 1e9);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  f16x8* src; float* out;
  hipMalloc(&src, 1024 * sizeof(f16x8));
  hipMalloc(&out, 64);
  for (int data = 0; data < 2; ++data) {
    std::vector<uint16_t> h(8192);
    for (auto& v : h) v = data ? (uint16_t)((rand() & 0x3FFF) | ((rand() & 1) << 15) | 0x2000) : 0;      // random finite fp16 / zeros
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const char* d = data ? "random" : "zeros ";
    run<0, 1, 256>(src, out, d); run<0, 2, 256>(src, out, d); run<0, 4, 256>(src, out, d); run<0, 8, 256>(src, out, d); run<0, 16, 256>(src, out, d);
    run<0, 2, 512>(src, out, d); run<0, 4, 512>(src, out, d); run<0, 8, 512>(src, out, d); run<0, 16, 512>(src, out, d);
    run<1, 1, 256>(src, out, d); run<1, 2, 256>(src, out, d); run<1, 4, 256>(src, out, d); run<1, 8, 256>(src, out, d);
    run<1, 2, 512>(src, out, d); run<1, 4, 512>(src, out, d);
  }
  return 0;
}
