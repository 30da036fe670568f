// macx_common.hip.h -- shared device helpers for the MI355X (gfx950) MAC-cell kernels.
//
// Everything here is wave64 / CDNA4 specific on purpose: no CUDA shims, no dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <map>
#include <set>
#include <utility>
#include "../../include/macx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace macx {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): the attribute is per-device state, so the
// cache is keyed by both (a process that drives several GPUs from one thread each must not skip the call on the second
// device), and it is guarded for concurrent callers.
inline hipError_t lds_attr_once(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({fn, dev})) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done.insert({fn, dev});
  return e;
}

// compute units of the current device (0 if it cannot be asked): a device fact, cached per device like the attribute above
inline int device_cu_count() {
  static std::mutex mu;
  static std::map<int, int> known;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lock(mu);
  auto it = known.find(dev);
  if (it != known.end()) return it->second;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
  known[dev] = n;
  return n;
}

// ---------------------------------------------------------------------------------------------
// The per-call tuning table (macx_opts.tune, include/macx.h: MACX_TUNE_*).  An entry point that takes macx_opts installs the
// caller's table for the duration of the call ON THE CALLING THREAD (TuneScope, via ModeScope in macx_api.hip); launchers read it
// through tune_get().  Entry points without macx_opts run the defaults.  Nothing outlives the call: no process-wide state.
// ---------------------------------------------------------------------------------------------
inline const int32_t*& tune_current() { static thread_local const int32_t* t = nullptr; return t; }
inline int tune_get(int key, int dflt) {
  const int32_t* t = tune_current();
  return (t && t[key]) ? (int)t[key] - 1 : dflt;
}
struct TuneScope {
  const int32_t* saved;
  explicit TuneScope(const int32_t* t) : saved(tune_current()) { tune_current() = t; }
  ~TuneScope() { tune_current() = saved; }
};

// ---------------------------------------------------------------------------------------------
// Counter-based dropout stream.
//
// The reference draws its masks from TF's stateful Philox stream (ops.py:312, :674-679,
// :1054-1059, mac_cell.py:217,463), which is not reproducible across runs even in TF itself. The
// product replaces it with a stateless hash of (site key, flat element index): the mask of a
// given (seed, site, step, element) is a pure function, so forward and backward regenerate it
// instead of storing [B,N,d] masks, and a data-parallel shard sees exactly the masks the
// full-batch run would have used (the element index is built from the GLOBAL question index).
//
//   h = fmix((idx >> 1) ^ key);  bits = idx odd ? h >> 16 : h & 0xFFFF;   keep  <=>  (bits << 8) < thr24,   thr24 = floor(keep * 2^24)
//
// oracle/dropout_hash.py restates the same function in numpy; tests compare the two bit-exactly.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t hash_mix(uint32_t h) {
  h *= 0x9E3779B1u;
  h ^= h >> 15;
  h *= 0x85EBCA77u;
  h ^= h >> 13;
  h *= 0xC2B2AE3Du;
  h ^= h >> 16;
  return h;
}

__host__ __device__ __forceinline__ uint32_t site_key(uint32_t seed, uint32_t site, uint32_t step) {
  return hash_mix(hash_mix(seed ^ 0xA511E9B3u) ^ hash_mix(site * 0x632BE5ABu + step * 0x2545F491u + 0x1B873593u));
}

// One 32-bit hash decides TWO neighbouring elements (16 bits each): the hash is the price of a dropout site inside the fused
// kernels (2 x 64 per thread in front of the chain kernel's first product), and 2^-16 is resolution enough for a keep
// probability.
__host__ __device__ __forceinline__ bool keep_bit(uint32_t idx, uint32_t key, uint32_t thr24) {
  const uint32_t h = hash_mix((idx >> 1) ^ key);
  return (((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) << 8) < thr24;
}
// elements idx (even) and idx + 1 from one hash: bit 0 / bit 1 of the result
__host__ __device__ __forceinline__ uint32_t keep_pair(uint32_t idx_even, uint32_t key, uint32_t thr24) {
  const uint32_t h = hash_mix((idx_even >> 1) ^ key);
  return (((h & 0xFFFFu) << 8) < thr24 ? 1u : 0u) | (((h >> 16) << 8) < thr24 ? 2u : 0u);
}

// dropout sites inside the cell (the numbers are part of the mask definition)
enum : uint32_t {
  SITE_MEM_VAR = 1,   // ops.py:1054  variational memory mask, drawn once per batch in zero_state
  SITE_MEM     = 2,   // mac_cell.py:217  plain memory dropout (memoryVariationalDropout off)
  SITE_READ_KB = 3,   // ops.py:678   dropout(x) on the knowledge base inside ops.mul(proj=...)
  SITE_READ_MEM = 4,  // ops.py:679   dropout(y) on the memory inside ops.mul(proj=...)
  SITE_READ_ATT = 5,  // ops.py:312 via :142  dropout on the interactions before the d->1 logits
  SITE_WRITE_INFO = 6, // mac_cell.py:463  dropout on the retrieved information
  SITE_OUT_FC0 = 7,   // ops.py:312 via FCLayer (ops.py:349-359): dropout on the classifier's first layer input
  SITE_OUT_FC1 = 8,   // ... and on its second layer input
  SITE_STEM0 = 9,     // ops.py:400  dropout on the input of stem conv layer 0 (the image features)
  SITE_STEM1 = 10,    // ... and of stem conv layer 1
  SITE_ENC_INPUT = 11, // ops.py:880  dropout on the embedded question words (encInputDropout)
  SITE_QUESTION = 12  // model.py:292 dropout on the question vector (qDropout)
};

struct DropSpec {
  uint32_t key;      // site_key(seed, site, step)
  uint32_t thr24;    // floor(keep * 2^24); 1<<24 means keep everything
  float inv_keep;    // 1/keep
  const uint32_t* word;   // macx_dropout.mask_word: one device word XORed into the key when the kernel RUNS (null = 0)
};

// The key a kernel hashes with: the host's site key, XOR the run's mask word read from device memory.  A captured graph bakes
// `key` (a kernel argument) but reads the word on every replay, so one capture draws fresh masks per replay (macx.h, macx_dropout).
__device__ __forceinline__ uint32_t run_key(uint32_t key, const uint32_t* word) { return word ? key ^ *word : key; }
__device__ __forceinline__ uint32_t run_key(const DropSpec& d) { return run_key(d.key, d.word); }

// (callers that apply a spec to many elements resolve the key once: drop_resolve)
__device__ __forceinline__ DropSpec drop_resolve(DropSpec d) { d.key = run_key(d); d.word = nullptr; return d; }
__device__ __forceinline__ float drop_apply(float v, uint32_t idx, const DropSpec& d) {
  return keep_bit(idx, d.key, d.thr24) ? v * d.inv_keep : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// activations (ops.py:161-187).  config.relu selects what "RELU" means; args.txt uses ELU.
// ---------------------------------------------------------------------------------------------
enum : int { ACT_NON = 0, ACT_TANH = 1, ACT_SIGMOID = 2, ACT_ELU = 3, ACT_RELU = 4 };

// ELU (ops.py:170): x > 0 ? x : expm1(x).  expm1 on the negative side as a degree-7 Taylor polynomial for x > -0.35 (no
// cancellation, truncation error 3e-8 relative) and exp(x) - 1 below (cancellation error <= 1 ulp of 1 on a result >= 0.3):
// 2.7e-7 relative at worst against fp64 -- the libm expm1f this replaces measures 1.9e-7 -- at about a third of its
// instructions; the epilogues that apply it hold 52 values per lane.
__device__ __forceinline__ float elu_f(float x) {
  const float e = __expf(x) - 1.0f;
  float r = 1.0f / 5040.0f;
  r = fmaf(r, x, 1.0f / 720.0f);
  r = fmaf(r, x, 1.0f / 120.0f);
  r = fmaf(r, x, 1.0f / 24.0f);
  r = fmaf(r, x, 1.0f / 6.0f);
  r = fmaf(r, x, 0.5f);
  r = fmaf(r, x, 1.0f);
  r *= x;
  return x > 0.0f ? x : (x > -0.35f ? r : e);
}
// derivative of ELU expressed through its OUTPUT h = elu(x): x>0 -> 1, else exp(x) = h + 1
__device__ __forceinline__ float elu_grad_from_out(float h) { return h > 0.0f ? 1.0f : h + 1.0f; }
// derivative of ELU from its INPUT
__device__ __forceinline__ float elu_grad_from_in(float x) { return x > 0.0f ? 1.0f : expf(x); }

__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case ACT_TANH: return tanhf(x);
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
    case ACT_ELU: return elu_f(x);
    case ACT_RELU: return fmaxf(x, 0.0f);
    default: return x;
  }
}
// d act / d x given the activation OUTPUT o
__device__ __forceinline__ float act_grad_from_out(int act, float o) {
  switch (act) {
    case ACT_TANH: return 1.0f - o * o;
    case ACT_SIGMOID: return o * (1.0f - o);
    case ACT_ELU: return elu_grad_from_out(o);
    case ACT_RELU: return o > 0.0f ? 1.0f : 0.0f;
    default: return 1.0f;
  }
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the 32 lanes of one half-wave (lanes 0-31 and 32-63 reduce independently)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace macx
