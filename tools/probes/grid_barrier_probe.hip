// grid_barrier_probe.hip -- what an in-kernel barrier among G workgroups costs on MI355X, against the launch boundary it would replace.
//
// The [B,d] recurrence of the MAC cell is ~100 dependent launches of 5-11 us per training step; fusing
// two dependent [64,512] x [512,512] products into one launch needs every output of the first before any of the second: a barrier
// among the workgroups of the launch.  This probe measures that barrier for G = 8 .. 256 workgroups, (a) spread over the chip as the
// dispatcher places them (workgroup i -> XCD i % 8) and (b) confined to ONE XCD (8 G workgroups launched, those with blockIdx % 8 != 0
// exit at once; HW_REG_XCC_ID is read back to check the placement), and checks that data written before the barrier by every
// workgroup is visible behind it to every other one (agent-scope release / acquire: L2 write-back + invalidate across XCDs).
//   barrier = monotonic counter: atomicAdd (agent scope) + spin on an agent-scope load, one thread per workgroup, __syncthreads around
// Also timed in the same process: an empty-kernel launch boundary (K dependent launches on one stream).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier_probe.hip -o tools/probes/bin/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

// one barrier round among `g` participants; bounded spin (a lost participant must not hang the box)
__device__ __forceinline__ bool grid_sync(uint32_t* counter, uint32_t target, uint32_t* fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();                                                    // release: this workgroup's stores reach the device scope
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { ok = false; atomicAdd(fail, 1u); break; }
    }
    __threadfence();                                                    // acquire: later loads see the other workgroups' stores
  }
  __syncthreads();
  return ok;
}

// rounds of: write a word per workgroup, barrier, read every workgroup's word (checks visibility), barrier
__global__ __launch_bounds__(256) void barrier_kernel(int G, int stride, int rounds, uint32_t* counter, uint32_t* slots, uint32_t* fail,
                                                      uint32_t* xcc_seen, uint32_t* bad) {
  if (blockIdx.x % stride != 0) return;
  const int me = blockIdx.x / stride;
  if (me >= G) return;
  if (threadIdx.x == 0) xcc_seen[me] = xcc_id();
  uint32_t target = 0;
  for (int r = 1; r <= rounds; ++r) {
    if (threadIdx.x == 0) slots[me * 32] = (uint32_t)r * 1000u + me;     // one cache line per workgroup
    target += G;
    if (!grid_sync(counter, target, fail)) return;
    uint32_t wrong = 0;
    for (int j = threadIdx.x; j < G; j += blockDim.x) wrong += slots[j * 32] != (uint32_t)r * 1000u + j;
    if (wrong) atomicAdd(bad, wrong);
    target += G;
    if (!grid_sync(counter, target, fail)) return;
  }
}

__global__ void empty_kernel(uint32_t* p) { if (p && threadIdx.x == 12345) p[0] = 1; }

int main() {
  uint32_t *counter, *slots, *fail, *xcc, *bad;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&slots, 256 * 32 * 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&xcc, 256 * 4)); CK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 200;
  printf("# grid barrier among G workgroups of 256 threads, %d rounds x 2 barriers per launch; us per barrier = (launch time - launch time at 1 round) / (2 (rounds - 1))\n", rounds);
  printf("# %-8s %-10s %-12s %-10s %-10s %s\n", "G", "placement", "us/barrier", "fail", "stale", "XCDs seen");
  for (int stride : {1, 8}) {
    for (int G : {8, 16, 32, 64, 128, 256}) {
      if (stride == 8 && G > 32) continue;            // one XCD holds 32 CUs
      float ms[2] = {0.f, 0.f};
      uint32_t hf = 0, hb = 0;
      std::vector<uint32_t> hx(256, 99);
      for (int which = 0; which < 2; ++which) {
        const int rr = which ? rounds : 1;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipMemset(counter, 0, 4)); CK(hipMemset(fail, 0, 4)); CK(hipMemset(bad, 0, 4));
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(barrier_kernel, dim3(G * stride), dim3(256), 0, 0, G, stride, rr, counter, slots, fail, xcc, bad);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float t;
          CK(hipEventElapsedTime(&t, e0, e1));
          if (t < best) best = t;
          uint32_t f, b;
          CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
          hf += f; hb += b;
        }
        ms[which] = best;
      }
      CK(hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost));
      uint32_t mask = 0;
      for (int i = 0; i < G; ++i) mask |= 1u << (hx[i] & 15);
      printf("  %-8d %-10s %-12.3f %-10u %-10u 0x%02x\n", G, stride == 1 ? "spread" : "one XCD", (ms[1] - ms[0]) * 1e3f / (2.f * (rounds - 1)), hf, hb, mask);
    }
  }
  // launch boundary: K dependent empty launches on one stream
  {
    const int K = 2000;
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_kernel, dim3(128), dim3(256), 0, 0, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(empty_kernel, dim3(128), dim3(256), 0, 0, nullptr);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    printf("# empty kernel, 128 workgroups, %d dependent launches on one stream (eager): %.3f us per launch\n", K, t * 1e3f / K);
    // the same as a captured graph (what the training step is replayed from)
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(128), dim3(256), 0, st, nullptr);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&t, e0, e1));
    printf("# the same from a captured graph of 200 kernel nodes, 10 replays: %.3f us per node\n", t * 1e3f / 2000.f);
  }
  return 0;
}
