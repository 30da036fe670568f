// Probe: what does ds_read_b64_tr_b16 return?  LDS holds u16 value = its own element index; every lane passes a byte
// address; we print, for each lane, the four 16-bit values it received.  hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const uint32_t* addr_in, uint32_t* out) {
  __shared__ uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t a = addr_in[threadIdx.x] + (uint32_t)(uintptr_t)lds;
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[2 * threadIdx.x] = (uint32_t)v;
  out[2 * threadIdx.x + 1] = (uint32_t)(v >> 32);
}
int main() {
  uint32_t h_addr[64], h_out[128];
  uint32_t *d_addr, *d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 3; ++pat) {
    const int RS = pat == 2 ? 80 : 64;          // row stride in bytes
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, t = l & 15;
      if (pat == 0) h_addr[l] = (uint32_t)(l * 8);                              // lane-linear 8-byte chunks
      else h_addr[l] = (uint32_t)((8 * g + t / 4) * RS + (t % 4) * 8);          // group g: rows 8g..8g+3, chunk (row t/4, cols 4(t%4)..)
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d (row stride %d B; element index = byte/2)\n", pat, RS);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d addr %4u(elem %4u): %5u %5u %5u %5u\n", l, h_addr[l], h_addr[l] / 2, h_out[2 * l] & 0xFFFF, h_out[2 * l] >> 16,
             h_out[2 * l + 1] & 0xFFFF, h_out[2 * l + 1] >> 16);
    }
  }
  return 0;
}
