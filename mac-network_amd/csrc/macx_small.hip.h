// macx_small.hip.h -- the [B,d]-sized and HBM-bound kernels of the cell: weight packing, the small
// linears (ops.py:298-333 on [B,d] inputs), the control unit's word attention
// (mac_cell.py:153-181), the read unit's softmax + summary over the knowledge base
// (mac_cell.py:264-275; ops.py:140-150) and their backward twins.
#pragma once
#include "macx_common.hip.h"
#include "macx_h2.hip.h"
#include "macx_lin_tile.hip.h"

namespace macx {

// ---------------------------------------------------------------------------------------------
// weight packing for the MFMA kernels:  dst[Q][g][j][e] = src[(16Q+4g+e)*ld_k + j*ld_j]
//   (ld_k, ld_j) = (Nout, 1) packs W[K][Nout];  (1, K) packs W^T from W[Nout][K]
// ---------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ src, int ld_k, int ld_j, int K, int Nout, float* dst) {
  const size_t total = (size_t)K * Nout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = i & 3;
    const size_t t = i >> 2;
    const int j = t % Nout;
    const size_t qg = t / Nout;
    const int g = qg & 3;
    const int Q = qg >> 2;
    const int k = 16 * Q + 4 * g + e;
    dst[i] = src[(size_t)k * ld_k + (size_t)j * ld_j];
  }
}

// several weights in one launch (blockIdx.y selects the descriptor)
//   fmt 0: dst[Q][g][j][e] fp32 (16x16x4 f32 MFMA kernels, small linears)
//   fmt 1: dst[kt][plane][j][32] bf16 -- the exact 3-way bf16 split of W (macx_gemm6.hip.h, B_PLAIN); K*Nout*3/2 floats
//   fmt 2: dst[kt][j][32] fp32 k-major tiles (macx_gemm6.hip.h, B_YMIX_*: mixed in fp32, split while staging)
//   fmt 3: H2 weight planes dst[kt][plane][g][Nout] x 16 B fp16 + the matrix exponent (macx_h2.hip.h, macx_gemm_h2.hip.h)
struct PackDesc { const float* src; float* dst; int ld_k, ld_j, K, Nout; int k_src, n_src; int fmt; int maxabs_n; const float* maxabs; float* exp_dst; float* maxabs_out; };      // 72 bytes   // zero fill for k >= k_src or j >= n_src; exp_dst: where format 3 leaves its exponent (null: behind the planes)
constexpr int PACK_MAX = 56;       // 56 x 72 B of kernel arguments (the limit is 4 KB)
static_assert(sizeof(PackDesc) * PACK_MAX <= 4096, "PackList travels by value as a kernel argument");
struct PackList { PackDesc d[PACK_MAX]; };
__global__ void pack_weights_kernel(PackList L) {
  const PackDesc q = L.d[blockIdx.y];
  const size_t total = (size_t)q.K * q.Nout;
  if (q.fmt == 3) {
    pack_h2_weight(q.src, q.ld_k, q.ld_j, q.K, q.Nout, q.k_src, q.n_src, q.maxabs, q.dst,
                   (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x, q.exp_dst, q.maxabs_n, q.maxabs_out);
    return;
  }
  if (q.fmt == 0) {
    // one thread per group of four consecutive k of a column: one 16-byte store (and, for a transposed source, one 16-byte load)
    const bool vec4 = q.ld_k == 1 && (q.ld_j & 3) == 0 && (reinterpret_cast<uintptr_t>(q.src) & 15) == 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total / 4; t += (size_t)gridDim.x * blockDim.x) {
      const int j = (int)(t % q.Nout);
      const size_t qg = t / q.Nout;
      const int k0 = 16 * (int)(qg >> 2) + 4 * (int)(qg & 3);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (j < q.n_src) {
        if (vec4 && k0 + 4 <= q.k_src) {
          v = *reinterpret_cast<const f32x4*>(q.src + (size_t)k0 + (size_t)j * q.ld_j);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + e < q.k_src) v[e] = q.src[(size_t)(k0 + e) * q.ld_k + (size_t)j * q.ld_j];
        }
      }
      reinterpret_cast<f32x4*>(q.dst)[t] = v;
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    {
      const int kk = i & 31;
      const size_t t = i >> 5;
      const int j = t % q.Nout;
      const int kt = (int)(t / q.Nout);
      const int k = kt * 32 + kk;
      const float x = (k < q.k_src && j < q.n_src) ? q.src[(size_t)k * q.ld_k + (size_t)j * q.ld_j] : 0.f;
      if (q.fmt == 2) {
        q.dst[i] = x;
      } else {
        // x = h1 + h2 + h3 exactly (round-to-nearest residual chain, the same split the kernel applies to A)
        const __bf16 h1 = (__bf16)x;
        const float r1 = x - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const __bf16 h3 = (__bf16)(r1 - (float)h2);
        __bf16* d = reinterpret_cast<__bf16*>(q.dst);
        const size_t plane = (size_t)q.Nout * 32;
        const size_t o = ((size_t)kt * 3 * q.Nout + j) * 32 + kk;
        d[o] = h1; d[o + plane] = h2; d[o + 2 * plane] = h3;
      }
    }
  }
}

// dropout of the knowledge base (ops.py:678) as its own pass: out = kb * mask / keep, plus the keep
// bits (1 per element) the backward pass needs.  16 B per lane; 8 lanes assemble one 32-bit word.
__global__ __launch_bounds__(256) void kb_dropout_kernel(const float* __restrict__ kb, size_t n4, uint32_t key, uint32_t thr24,
                                                        float inv_keep, uint32_t first, float* out, uint32_t* bits, uint32_t key2,
                                                        uint32_t* bits2, const uint32_t* word = nullptr) {
  key = run_key(key, word);
  key2 = run_key(key2, word);
  // bits2 (optional): keep bits of a second dropout site over the same index range (the read unit's attention dropout,
  // ops.py:312 via :142) -- the pass is HBM-bound, the second hash is free and saves a launch per step
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4r = (n4 + 7) & ~(size_t)7;   // whole words: every lane of an 8-lane group takes part in the shuffle
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4r; i += stride) {
    const bool in = i < n4;
    f32x4 v = in ? reinterpret_cast<const f32x4*>(kb)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t nib = 0, nib2 = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t idx = first + (uint32_t)(i * 4) + e;
      const bool keep = keep_bit(idx, key, thr24);
      nib |= (keep ? 1u : 0u) << e;
      v[e] = keep ? v[e] * inv_keep : 0.f;
      if (bits2) nib2 |= (keep_bit(idx, key2, thr24) ? 1u : 0u) << e;
    }
    if (in) reinterpret_cast<f32x4*>(out)[i] = v;
    uint32_t w = nib << (4 * (threadIdx.x & 7));
    w |= __shfl_xor(w, 1, 64);
    w |= __shfl_xor(w, 2, 64);
    w |= __shfl_xor(w, 4, 64);
    if ((threadIdx.x & 7) == 0 && in) bits[i >> 3] = w;
    if (bits2) {
      uint32_t w2 = nib2 << (4 * (threadIdx.x & 7));
      w2 |= __shfl_xor(w2, 1, 64);
      w2 |= __shfl_xor(w2, 2, 64);
      w2 |= __shfl_xor(w2, 4, 64);
      if ((threadIdx.x & 7) == 0 && in) bits2[i >> 3] = w2;
    }
  }
}

// keep bits of a dropout site: word w holds elements first + 32w .. first + 32w + 31 (bit i = element i)
__global__ void mask_bits_kernel(uint32_t key, uint32_t thr24, uint32_t first, size_t nwords, uint32_t* out, const uint32_t* word = nullptr) {
  key = run_key(key, word);
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (size_t)gridDim.x * blockDim.x) {
    const uint32_t base = first + (uint32_t)(w << 5);
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) bits |= (keep_bit(base + i, key, thr24) ? 1u : 0u) << i;
    out[w] = bits;
  }
}

// dst[z][c][r] = src[z][r][c]   (z = blockIdx.z; matrices stored back to back)
__global__ void transpose_kernel(const float* __restrict__ src, int R, int C, float* dst) {
  __shared__ float tile[32][33];
  src += (size_t)blockIdx.z * R * C;
  dst += (size_t)blockIdx.z * R * C;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
  for (int i = ty; i < 32; i += 8) {
    const int r = by + i, c = bx + tx;
    tile[i][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = bx + i, r = by + tx;
    if (r < R && c < C) dst[(size_t)c * R + r] = tile[tx][i];
  }
}

// ---------------------------------------------------------------------------------------------
// small linear:  out[z][r][j] = epi( sum_k x(z,r,k) W[z][k][j] + bias[z][j] + bias_const )
// for the [B,d]-sized linears of the cell (ops.linear, ops.py:298-333).  x is the concatenation of
// up to 3 segments along k (ops.concat, ops.py:65-78, never materialised).
//
// fp32 MFMA 16x16x4.  A workgroup owns 64 rows x 16 columns; its 4 waves take interleaved 16-wide
// k groups of the whole reduction (operands are L2-resident, fragments are loaded straight into
// registers: A as float4 along k, W from the packed [K/16][4][n_out][4] layout), then a fixed-order
// LDS combine and a float4 epilogue.  Grid = n_out/16 x ceil(rows/64) x batch.
// ---------------------------------------------------------------------------------------------
// (LinSeg, LinP and the tile function small_linear_tile: macx_lin_tile.hip.h)
template <int RTL, bool PART = false>
__global__ __launch_bounds__(256) void small_linear_kernel(LinP p) {
  __shared__ float red[4][16 * RTL][20];
  small_linear_tile<RTL, PART>(p, blockIdx.x, blockIdx.y, blockIdx.z, red, (int)threadIdx.x);
}

// the PART form (LinP::part): the dy-linear of the backward recurrence.  (Measured and removed in rounds 5-6: two dependent linears
// in one launch with a device-scope barrier between them -- 2-13 % slower per step, profiles/r05_pair_launch_ab.txt,
// r05_grid_barrier_probe.txt -- and 8-wave workgroups for the long reductions -- no gain, profiles/r05_wide_linear_ab.txt.)
inline hipError_t small_linear_part_launch(const LinP& p, hipStream_t st) {
  if (!p.part || !p.part_sum || p.rows > 128 || p.part_shift < 4 || p.part_shift > 6 ||
      ((p.part_N - 2) >> p.part_shift) + 2 > LIN_PART_TILES) return hipErrorInvalidValue;
  hipLaunchKernelGGL((small_linear_kernel<1, true>), dim3(p.n_out / 16, (p.rows + 15) / 16, 1), dim3(256), 0, st, p);
  return hipGetLastError();
}
inline hipError_t small_linear_launch(const LinP& p, int nz, hipStream_t st) {
  if (p.rows <= 128) {
    hipLaunchKernelGGL(small_linear_kernel<1>, dim3(p.n_out / 16, (p.rows + 15) / 16, nz), dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL(small_linear_kernel<4>, dim3(p.n_out / 16, (p.rows + 63) / 64, nz), dim3(256), 0, st, p);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// elementwise helpers on [B,d]
// ---------------------------------------------------------------------------------------------
// out = x * f1 * f2 with two dropout streams indexed (row0+r)*dl + j   (mac_cell.py:214-217 then ops.py:679); dl: the logical
// width of a zero-padded cell (macx_shapes.d_logical), 0 = d
__global__ void drop2_kernel(const float* __restrict__ x, int rows, int d, uint32_t row0, DropSpec d1, DropSpec d2, float* out, int dl = 0) {
  d1 = drop_resolve(d1);
  d2 = drop_resolve(d2);
  const int n = rows * d;
  if (dl <= 0) dl = d;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int r = i / d, j = i - r * d;
    const uint32_t idx = (row0 + (uint32_t)r) * (uint32_t)dl + (uint32_t)j;
    float v = x[i];
    v = drop_apply(v, idx, d1);
    v = drop_apply(v, idx, d2);
    out[i] = v;
  }
}

// initial state (mac_cell.py:496-505): PRM -> tile the [d] variable, ZERO, Q -> copy vecQuestions
// ... both states of a run in one launch
// ... and, with md != null, step 0's dropped memory md = m_0 * f1 * f2 (mac_cell.py:214-217 then ops.py:679: what drop2_kernel would
// write in a launch of its own in front of the first projY linear); dl: the logical width of a zero-padded cell, 0 = d
constexpr int SYNC_WORDS = 64 + 32 * 16;      // SavedLayout::sync: [i] y counter of step i, [63] fail word, [64 + 16 i ..] step i's block counters
__global__ void init_states_kernel(int mode_c, const float* __restrict__ prm_c, float* out_c, int mode_m, const float* __restrict__ prm_m,
                                   float* out_m, const float* __restrict__ vecQ, int rows, int d, float* md = nullptr, uint32_t row0 = 0,
                                   DropSpec d1 = DropSpec{}, DropSpec d2 = DropSpec{}, int dl = 0, uint32_t* sync = nullptr) {
  const int n = rows * d;
  // the counters workgroups of ONE launch signal each other through (SavedLayout::sync): zero at the start of every forward pass
  if (sync && blockIdx.x == 0)
    for (int i = threadIdx.x; i < SYNC_WORDS; i += blockDim.x) sync[i] = 0u;
  if (md) { d1 = drop_resolve(d1); d2 = drop_resolve(d2); }
  if (dl <= 0) dl = d;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float q = (mode_c == 2 || mode_m == 2) ? vecQ[i] : 0.f;
    out_c[i] = mode_c == 0 ? prm_c[i % d] : (mode_c == 2 ? q : 0.f);
    const float m = mode_m == 0 ? prm_m[i % d] : (mode_m == 2 ? q : 0.f);
    out_m[i] = m;
    if (md) {
      const int r = i / d, j = i - r * d;
      const uint32_t idx = (row0 + (uint32_t)r) * (uint32_t)dl + (uint32_t)j;
      md[i] = drop_apply(drop_apply(m, idx, d1), idx, d2);
    }
  }
}

// start of a backward pass: the running gradients DM / DC [p + 1][rows * d] are zero except their last slab, which takes the
// incoming dL/dm_p / dL/dc_p (null: zero).  One launch instead of a fill and two copies.  DM and DC adjacent
// (DC == DM + (p + 1) n); `tail` more words after DC are zeroed too.
__global__ void bwd_init_kernel(float* DM, size_t slab, int p, const float* __restrict__ d_memory, const float* __restrict__ d_control, int tail) {
  const size_t per = (size_t)(p + 1) * slab, total = 2 * per + (size_t)tail;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < 2 * per) {
      const size_t w = i >= per ? i - per : i;
      if (w >= (size_t)p * slab) {
        const float* src = i >= per ? d_control : d_memory;
        if (src) v = src[w - (size_t)p * slab];
      }
    }
    DM[i] = v;
  }
}

__global__ void dropout_mask_kernel(uint32_t key, uint32_t thr24, uint32_t first, size_t n, float* out, const uint32_t* word = nullptr) {
  key = run_key(key, word);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = keep_bit(first + (uint32_t)i, key, thr24) ? 1.f : 0.f;
}

// dst[i] = sum over `rows` rows of src[r*ld + i]   (bias gradients from per-question partials).
// One workgroup (1024 threads) per 16 columns: 64 row groups x 16 columns, each thread sums its interleaved rows, then a
// fixed-order LDS combine.  (16 rather than 64 columns per workgroup: at n = 512 that is 32 workgroups instead of 8 on a
// 256-CU chip; the launch is latency-bound.)
constexpr int ROWSUM_COLS = 16;
__global__ __launch_bounds__(1024) void rowsum_kernel(const float* __restrict__ src, int rows, int n, size_t ld, float* dst, size_t zsrc,
                                                      size_t zdst) {
  __shared__ float red[64][ROWSUM_COLS + 1];
  src += blockIdx.y * zsrc;      // blockIdx.y: independent sums of one launch
  dst += blockIdx.y * zdst;
  const int c = threadIdx.x & (ROWSUM_COLS - 1), rg = threadIdx.x / ROWSUM_COLS;
  const int i = blockIdx.x * ROWSUM_COLS + c;
  float s0 = 0.f, s1 = 0.f;
  if (i < n) {
    int r = rg;
    for (; r + 64 < rows; r += 128) {
      s0 += src[(size_t)r * ld + i];
      s1 += src[(size_t)(r + 64) * ld + i];
    }
    if (r < rows) s0 += src[(size_t)r * ld + i];
  }
  red[rg][c] = s0 + s1;
  __syncthreads();
  if (rg == 0 && i < n) {
    float t = red[0][c];
#pragma unroll
    for (int w = 1; w < 64; ++w) t += red[w][c];
    dst[i] = t;
  }
}

// several row sums in one launch (blockIdx.y selects the descriptor): the backward pass ends in a dozen bias gradients, each
// a launch of a few microseconds with nothing depending on it but the caller
struct RowsumDesc { const float* src; float* dst; int rows, n; size_t ld; };
constexpr int ROWSUM_MAX = 40;
struct RowsumList { RowsumDesc d[ROWSUM_MAX]; };
__global__ __launch_bounds__(1024) void rowsum_list_kernel(RowsumList L) {
  __shared__ float red[64][ROWSUM_COLS + 1];
  const RowsumDesc q = L.d[blockIdx.y];
  if ((int)blockIdx.x * ROWSUM_COLS >= q.n) return;
  const int c = threadIdx.x & (ROWSUM_COLS - 1), rg = threadIdx.x / ROWSUM_COLS;
  const int i = blockIdx.x * ROWSUM_COLS + c;
  float s0 = 0.f, s1 = 0.f;
  if (i < q.n) {           // the same order of additions as rowsum_kernel
    int r = rg;
    for (; r + 64 < q.rows; r += 128) {
      s0 += q.src[(size_t)r * q.ld + i];
      s1 += q.src[(size_t)(r + 64) * q.ld + i];
    }
    if (r < q.rows) s0 += q.src[(size_t)r * q.ld + i];
  }
  red[rg][c] = s0 + s1;
  __syncthreads();
  if (rg == 0 && i < q.n) {
    float t = red[0][c];
#pragma unroll
    for (int w = 1; w < 64; ++w) t += red[w][c];
    q.dst[i] = t;
  }
}

// dst[r][j] = src[r*ld_src + col0 + j] * dropfactor((row0+r)*d + j)
__global__ void copy_cols_drop_kernel(const float* __restrict__ src, int ld_src, int col0, int rows, int d, uint32_t row0,
                                      DropSpec ds, float* dst, int dl = 0) {
  ds = drop_resolve(ds);
  const int n = rows * d;
  if (dl <= 0) dl = d;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int r = i / d, j = i - r * d;
    dst[i] = drop_apply(src[(size_t)r * ld_src + col0 + j], (row0 + (uint32_t)r) * (uint32_t)dl + (uint32_t)j, ds);
  }
}

// dst = g * act'(o) where o is the activation OUTPUT
__global__ void mul_actgrad_kernel(const float* __restrict__ g, const float* __restrict__ o, int act, size_t n, float* dst) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = g[i] * act_grad_from_out(act, o[i]);
}

__global__ void axpy_kernel(const float* __restrict__ x, size_t n, float* y) {   // y += x
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] += x[i];
}

// ---------------------------------------------------------------------------------------------
// control unit, step 2 (mac_cell.py:153-181): one workgroup per (question, step)
//   logits[s] = sum_d cc[d] words[s][d] w[d] + b ;  att = softmax(logits + (1-mask)*(-1e30))
//   control[d] = sum_s att[s] words[s][d]
// Requires d <= 1024 (2 x float2 per thread would be needed beyond 512; handled by a d-loop).
// ---------------------------------------------------------------------------------------------
struct CtrlP {
  int B, S, d;
  const float* cc; size_t z_cc;        // [z][B][d]
  const float* words;                  // [B][S][d]
  const int32_t* lengths;              // [B]
  const float* w; const float* bias;   // [d], [1]
  float* att; size_t z_att;            // [z][B][S]
  float* control; size_t z_ctl;        // [z][B][d]
};

constexpr int C_MAXS = 256;   // max padded question length held in LDS
constexpr int CA_U = 8;       // words a wave requests together in the logit / da passes of the word-attention kernels

__global__ __launch_bounds__(256) void control_attend_kernel(CtrlP p) {
  __shared__ float s_logit[C_MAXS];
  __shared__ float s_red[8];
  const int b = blockIdx.x, z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* cc = p.cc + (size_t)z * p.z_cc + (size_t)b * p.d;
  const float* words = p.words + (size_t)b * p.S * p.d;
  const int L = p.lengths[b];

  // a wave takes the words s = wave, wave + 4, ..; CA_U of them are requested together (rows past the end re-read the last one):
  // the launch is a chain of dependent L2 round trips, one per batch instead of one per word (round 6: 23.8 -> ~12 us at S = 50)
  const float bias0 = p.bias[0];
  for (int s0 = wave; s0 < p.S; s0 += 4 * CA_U) {
    float part[CA_U];
#pragma unroll
    for (int u = 0; u < CA_U; ++u) part[u] = 0.f;
    for (int k = lane * 4; k < p.d; k += 256) {
      f32x4 wd[CA_U];
#pragma unroll
      for (int u = 0; u < CA_U; ++u) wd[u] = *reinterpret_cast<const f32x4*>(words + (size_t)min(s0 + 4 * u, p.S - 1) * p.d + k);
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(cc + k);
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.w + k);
      // (cc * words) * w, the order of mac_cell.py:155 then ops.py:317
#pragma unroll
      for (int u = 0; u < CA_U; ++u)
        part[u] += (c4[0] * wd[u][0]) * w4[0] + (c4[1] * wd[u][1]) * w4[1] + (c4[2] * wd[u][2]) * w4[2] + (c4[3] * wd[u][3]) * w4[3];
    }
#pragma unroll
    for (int u = 0; u < CA_U; ++u) {
      const int s = s0 + 4 * u;
      const float t = wave_sum(part[u]);
      if (lane == 0 && s < p.S) {
        const float logit = t + bias0;
        // ops.expMask (ops.py:243-247): seq + (1 - mask) * (-1e30)
        s_logit[s] = logit + (s < L ? 0.f : 1.0f) * (-1e30f);
      }
    }
  }
  __syncthreads();
  float m = -INFINITY;
  for (int s = tid; s < p.S; s += 256) m = fmaxf(m, s_logit[s]);
  m = wave_max(m);
  if (lane == 0) s_red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  float sum = 0.f;
  for (int s = tid; s < p.S; s += 256) sum += expf(s_logit[s] - m);
  sum = wave_sum(sum);
  if (lane == 0) s_red[4 + wave] = sum;
  __syncthreads();
  sum = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
  const float inv = 1.0f / sum;
  __syncthreads();
  for (int s = tid; s < p.S; s += 256) {
    const float a = expf(s_logit[s] - m) * inv;
    s_logit[s] = a;
    p.att[(size_t)z * p.z_att + (size_t)b * p.S + s] = a;
  }
  __syncthreads();
  for (int k = tid * 2; k < p.d; k += 512) {
    float o0 = 0.f, o1 = 0.f;
    for (int s = 0; s < p.S; ++s) {
      const float a = s_logit[s];
      const float2 wd = *reinterpret_cast<const float2*>(words + (size_t)s * p.d + k);
      o0 = fmaf(a, wd.x, o0);
      o1 = fmaf(a, wd.y, o1);
    }
    float* dst = p.control + (size_t)z * p.z_ctl + (size_t)b * p.d + k;
    dst[0] = o0;
    dst[1] = o1;
  }
}

// backward of the word attention, in two launches so that every question x step (then every
// question x column slab) gets its own workgroup and d(words) is written once, without atomics:
//   (1) per (question, step): da[s] = dc . words[s];  dl = softmax backward;  db partial
//   (2) per (question, 64-column slab): R[z] = sum_s dl[z][s] words[s];  dcc = w * R;
//       dw partial = sum_z cc[z] * R[z];  dwords[s] = sum_z att[z][s] dc[z] + dl[z][s] cc[z] w
struct CtrlBwdP {
  int B, S, d, nz;
  const float* dcontrol; size_t z_dc;   // [z][B][d]  gradient wrt the control of step z
  const float* cc; size_t z_cc;         // [z][B][d]
  const float* att; size_t z_att;       // [z][B][S]
  const float* words;                   // [B][S][d]
  const float* w;                       // [d]
  float* dl;                            // [z][B][S]   scratch (written by (1), read by (2))
  float* dcc; size_t z_dcc;             // [z][B][d]   out
  float* dwords;                        // [B][S][d]   out (written; accumulated when acc_words)
  int acc_words;                        // recurrent control: steps are differentiated one launch at a time
  float* dw_part;                       // [B][d]      out (sum over steps)
  float* db_part;                       // [nz*B]      out
};

__global__ __launch_bounds__(256) void control_bwd_dl_kernel(CtrlBwdP p) {
  __shared__ float s_da[C_MAXS];
  __shared__ float s_red[4];
  const int b = blockIdx.x, z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* words = p.words + (size_t)b * p.S * p.d;
  const float* dc = p.dcontrol + (size_t)z * p.z_dc + (size_t)b * p.d;
  const float* att = p.att + (size_t)z * p.z_att + (size_t)b * p.S;
  for (int s0 = wave; s0 < p.S; s0 += 4 * CA_U) {       // (CA_U words per batch of loads, as in control_attend_kernel)
    float part[CA_U];
#pragma unroll
    for (int u = 0; u < CA_U; ++u) part[u] = 0.f;
    for (int k = lane * 4; k < p.d; k += 256) {
      f32x4 wd[CA_U];
#pragma unroll
      for (int u = 0; u < CA_U; ++u) wd[u] = *reinterpret_cast<const f32x4*>(words + (size_t)min(s0 + 4 * u, p.S - 1) * p.d + k);
      const f32x4 g = *reinterpret_cast<const f32x4*>(dc + k);
#pragma unroll
      for (int u = 0; u < CA_U; ++u) part[u] += g[0] * wd[u][0] + g[1] * wd[u][1] + g[2] * wd[u][2] + g[3] * wd[u][3];
    }
#pragma unroll
    for (int u = 0; u < CA_U; ++u) {
      const int s = s0 + 4 * u;
      const float t = wave_sum(part[u]);
      if (lane == 0 && s < p.S) s_da[s] = t;
    }
  }
  __syncthreads();
  float dot = 0.f;
  for (int s = tid; s < p.S; s += 256) dot += att[s] * s_da[s];
  dot = wave_sum(dot);
  if (lane == 0) s_red[wave] = dot;
  __syncthreads();
  dot = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  __syncthreads();
  float dls = 0.f;
  for (int s = tid; s < p.S; s += 256) {
    const float dl = att[s] * (s_da[s] - dot);   // masked words have att == 0 -> dl == 0
    p.dl[((size_t)z * p.B + b) * p.S + s] = dl;
    dls += dl;
  }
  dls = wave_sum(dls);
  if (lane == 0) s_red[wave] = dls;
  __syncthreads();
  if (tid == 0) p.db_part[(size_t)z * p.B + b] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

constexpr int CB_MAXZ = 32;   // max netLength handled by the slab kernel's LDS staging
__global__ __launch_bounds__(256) void control_bwd_apply_kernel(CtrlBwdP p) {
  __shared__ float s_words[64][64];     // [s][col] for the 64 words of the current pass
  __shared__ float s_dl[CB_MAXZ][64];   // [z][s]
  __shared__ float s_att[CB_MAXZ][64];
  __shared__ float s_dc[CB_MAXZ][64];
  __shared__ float s_ccw[CB_MAXZ][64];
  __shared__ float s_R[4][64];
  const int b = blockIdx.x, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x, col = tid & 63, grp = tid >> 6;
  const float wc = p.w[c0 + col];
  for (int z = grp; z < p.nz; z += 4) {
    s_dc[z][col] = p.dcontrol[(size_t)z * p.z_dc + (size_t)b * p.d + c0 + col];
    s_ccw[z][col] = p.cc[(size_t)z * p.z_cc + (size_t)b * p.d + c0 + col] * wc;
  }
  float Racc[CB_MAXZ / 4];
#pragma unroll
  for (int i = 0; i < CB_MAXZ / 4; ++i) Racc[i] = 0.f;
  for (int sb = 0; sb < p.S; sb += 64) {
    const int sn = min(64, p.S - sb);
    __syncthreads();
    for (int s = grp; s < sn; s += 4) s_words[s][col] = p.words[((size_t)b * p.S + sb + s) * p.d + c0 + col];
    for (int i = tid; i < p.nz * 64; i += 256) {
      const int z = i >> 6, s = i & 63;
      const bool ok = s < sn;
      s_dl[z][s] = ok ? p.dl[((size_t)z * p.B + b) * p.S + sb + s] : 0.f;
      s_att[z][s] = ok ? p.att[(size_t)z * p.z_att + (size_t)b * p.S + sb + s] : 0.f;
    }
    __syncthreads();
    // R[z][col] += sum_s dl[z][s] words[s][col]   (group g takes steps z = g, g+4, ...)
#pragma unroll
    for (int i = 0; i < CB_MAXZ / 4; ++i) {
      const int z = grp + 4 * i;
      if (z < p.nz) {
        float r = Racc[i];
        for (int s = 0; s < sn; ++s) r = fmaf(s_dl[z][s], s_words[s][col], r);
        Racc[i] = r;
      }
    }
    // dwords[s][col] = sum_z att[z][s] dc[z][col] + dl[z][s] cc[z][col] w[col]   (group g takes s = g, g+4, ...)
    for (int s = grp; s < sn; s += 4) {
      float v = 0.f;
      for (int z = 0; z < p.nz; ++z) v += s_att[z][s] * s_dc[z][col] + s_dl[z][s] * s_ccw[z][col];
      float* dst = p.dwords + ((size_t)b * p.S + sb + s) * p.d + c0 + col;
      *dst = p.acc_words ? *dst + v : v;
    }
  }
  float dwp = 0.f;
#pragma unroll
  for (int i = 0; i < CB_MAXZ / 4; ++i) {
    const int z = grp + 4 * i;
    if (z < p.nz) {
      p.dcc[(size_t)z * p.z_dcc + (size_t)b * p.d + c0 + col] = Racc[i] * wc;
      // dw[col] += cc[z][col] * R[z][col]  (s_ccw holds cc * w; divide out w exactly by recomputing cc)
      dwp += p.cc[(size_t)z * p.z_cc + (size_t)b * p.d + c0 + col] * Racc[i];
    }
  }
  s_R[grp][col] = dwp;
  __syncthreads();
  if (grp == 0) {
    float* dst = p.dw_part + (size_t)b * p.d + c0 + col;
    const float t = (s_R[0][col] + s_R[1][col]) + (s_R[2][col] + s_R[3][col]);
    *dst = p.acc_words ? *dst + t : t;
  }
}

// ---------------------------------------------------------------------------------------------
// read unit, step 3 (mac_cell.py:264-275): softmax over the N knowledge-base cells of the logits
// the memKbProj_2 epilogue left behind, then the attention-weighted KB summary.
// One workgroup per (question, 128-column slab); 16-byte coalesced loads down the N axis.
// ---------------------------------------------------------------------------------------------
struct KbAttP {
  int B, N, d, nparts;
  const float* logit_part;   // [nparts][B*N]
  const float* bias;         // [1]
  const float* kb;           // [B][N][d]
  float* att;                // [B][N]
  float* info;               // [B][d]
};
constexpr int K_MAXN = 1024;

constexpr int KA_THREADS = 1024;     // 16 waves: the summary pass is a latency-bound stream of 16-byte loads, so many in flight
__global__ __launch_bounds__(KA_THREADS) void kb_attend_kernel(KbAttP p) {
  constexpr int NWV = KA_THREADS / 64, NRG = KA_THREADS / 32;
  __shared__ float s_att[K_MAXN];
  __shared__ float s_red[2 * NWV];
  __shared__ f32x4 s_acc[NRG][32];
  const int b = blockIdx.x, slab = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the knowledge-base rows of the summary pass do not depend on the softmax: the first KA_PRE of this thread's rows are
  // requested now and arrive while the three reductions below run (N <= 256: all of them)
  constexpr int KA_PRE = 8;
  const int rg = tid >> 5, c4 = tid & 31;
  const float* kb = p.kb + (size_t)b * p.N * p.d + slab * 128 + c4 * 4;
  f32x4 pv[KA_PRE];
#pragma unroll
  for (int i = 0; i < KA_PRE; ++i) pv[i] = *reinterpret_cast<const f32x4*>(kb + (size_t)min(rg + i * NRG, p.N - 1) * p.d);
  float m = -INFINITY;
  for (int n = tid; n < p.N; n += KA_THREADS) {
    float l = p.bias[0];
    for (int q = 0; q < p.nparts; ++q) l += p.logit_part[(size_t)q * p.B * p.N + (size_t)b * p.N + n];
    s_att[n] = l;
    m = fmaxf(m, l);
  }
  m = wave_max(m);
  if (lane == 0) s_red[wave] = m;
  __syncthreads();
  m = s_red[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) m = fmaxf(m, s_red[w]);
  float sum = 0.f;
  for (int n = tid; n < p.N; n += KA_THREADS) {
    const float e = expf(s_att[n] - m);
    s_att[n] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) s_red[NWV + wave] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) tot += s_red[NWV + w];      // fixed order
  const float inv = 1.0f / tot;
  for (int n = tid; n < p.N; n += KA_THREADS) {
    const float a = s_att[n] * inv;
    s_att[n] = a;
    if (slab == 0) p.att[(size_t)b * p.N + n] = a;
  }
  __syncthreads();
  // summary: NRG row groups x 32 float4 columns (ascending rows per thread)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KA_PRE; ++i) {
    const int n = rg + i * NRG;
    if (n < p.N) acc += pv[i] * s_att[n];
  }
  for (int n = rg + KA_PRE * NRG; n < p.N; n += NRG) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(kb + (size_t)n * p.d);
    const float a = s_att[n];
    acc += v * a;
  }
  s_acc[rg][c4] = acc;
  __syncthreads();
  if (tid < 32) {
    f32x4 t = s_acc[0][tid];
#pragma unroll
    for (int g = 1; g < NRG; ++g) t += s_acc[g][tid];
    *reinterpret_cast<f32x4*>(p.info + (size_t)b * p.d + slab * 128 + tid * 4) = t;
  }
}

// backward, part 1: da[b][n] = dr[b] . KB[b][n]   (one wave per knowledge-base cell)
__global__ __launch_bounds__(256) void kb_att_da_kernel(const float* __restrict__ dr, int ld_dr, const float* __restrict__ kb,
                                                        int B, int N, int d, float* da) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * 4 + wave;
  if (row >= (size_t)B * N) return;
  const int b = row / N;
  float part = 0.f;
  for (int k = lane * 4; k < d; k += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(kb + row * d + k);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dr + (size_t)b * ld_dr + k);
    part += v[0] * g[0] + v[1] * g[1] + v[2] * g[2] + v[3] * g[3];
  }
  part = wave_sum(part);
  if (lane == 0) da[row] = part;
}

// backward, part 2: softmax backward + everything elementwise between the logits and I2
// (SURVEY appendix A rows "softmax", "logit", "ctrl-mul").  One workgroup per (question, 128-column slab).
struct ReadAttBwdP {
  int B, N, d, b0;
  const float* att;      // [B][N]
  const float* da;       // [B][N]
  const float* I2;       // [B][N][d]
  const float* c;        // [B][d]   control of this step
  const float* wk;       // [d]
  int act;               // readCtrlAct
  const uint32_t* bits;  // keep bits of the SITE_READ_ATT mask of this step, [B*N][d/32]; null = keep all
  float inv_keep;
  float* dI2;            // [B][N][d]
  float* dc;             // [B][d]        dL/dc_i, accumulated in place (+=)
  float* dwk_part;       // [B][d]        (written)
  float* db2_part;       // [B][d]        column sums of dI2 (written)
  float* dbk_part;       // [B]           (written, by slab 0)
};

constexpr int RAB_THREADS = 1024;
constexpr int RAB_RG = RAB_THREADS / 32;
__global__ __launch_bounds__(RAB_THREADS) void read_att_bwd_kernel(ReadAttBwdP p) {
  __shared__ float s_dl[K_MAXN];
  __shared__ float s_red[RAB_THREADS / 64];
  __shared__ f32x4 s_acc[3][RAB_RG][32];
  const int b = blockIdx.x, slab = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NWV = RAB_THREADS / 64;
  float dot = 0.f;
  for (int n = tid; n < p.N; n += RAB_THREADS) dot += p.att[(size_t)b * p.N + n] * p.da[(size_t)b * p.N + n];
  dot = wave_sum(dot);
  if (lane == 0) s_red[wave] = dot;
  __syncthreads();
  dot = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) dot += s_red[w];
  __syncthreads();
  float dls = 0.f;
  for (int n = tid; n < p.N; n += RAB_THREADS) {
    const float dl = p.att[(size_t)b * p.N + n] * (p.da[(size_t)b * p.N + n] - dot);
    s_dl[n] = dl;
    dls += dl;
  }
  dls = wave_sum(dls);
  if (lane == 0) s_red[wave] = dls;
  __syncthreads();
  if (tid == 0 && slab == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) t += s_red[w];
    p.dbk_part[b] = t;
  }

  const int rg = tid >> 5, c4 = tid & 31;
  const int col = slab * 128 + c4 * 4;
  const f32x4 cv = *reinterpret_cast<const f32x4*>(p.c + (size_t)b * p.d + col);
  const f32x4 wv = *reinterpret_cast<const f32x4*>(p.wk + col);
  f32x4 a_dc = {0.f, 0.f, 0.f, 0.f}, a_dw = {0.f, 0.f, 0.f, 0.f}, a_db = {0.f, 0.f, 0.f, 0.f};
  for (int n = rg; n < p.N; n += RAB_RG) {
    const size_t off = ((size_t)b * p.N + n) * p.d + col;
    const f32x4 i2 = *reinterpret_cast<const f32x4*>(p.I2 + off);
    const float dl = s_dl[n];
    uint32_t bits = 0xFu;
    if (p.bits) bits = p.bits[((size_t)b * p.N + n) * (p.d >> 5) + (col >> 5)] >> (col & 31);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float zv = i2[e] * cv[e];
      const float g = act_apply(p.act, zv);
      const float f = ((bits >> e) & 1u) ? p.inv_keep : 0.f;
      a_dw[e] = fmaf(dl, g * f, a_dw[e]);                 // dw_k += dl * dropped(G)
      const float dz = (dl * wv[e]) * f * act_grad_from_out(p.act, g);
      a_dc[e] = fmaf(dz, i2[e], a_dc[e]);                 // dc += dZ * I2
      o[e] = dz * cv[e];                                  // dI2 = dZ * c
      a_db[e] += o[e];
    }
    *reinterpret_cast<f32x4*>(p.dI2 + off) = o;
  }
  s_acc[0][rg][c4] = a_dc;
  s_acc[1][rg][c4] = a_dw;
  s_acc[2][rg][c4] = a_db;
  __syncthreads();
  if (tid < 96) {
    const int which = tid >> 5, cc4 = tid & 31;
    f32x4 t = s_acc[which][0][cc4];
#pragma unroll
    for (int g = 1; g < RAB_RG; ++g) t += s_acc[which][g][cc4];
    float* dst = (which == 0 ? p.dc : which == 1 ? p.dwk_part : p.db2_part) + (size_t)b * p.d + slab * 128 + cc4 * 4;
    if (which == 0) t += *reinterpret_cast<const f32x4*>(dst);
    *reinterpret_cast<f32x4*>(dst) = t;
  }
}

// ---------------------------------------------------------------------------------------------
// write unit extras: memory gate (mac_cell.py:358-367) and self-attention over the previous
// control / memory states (mac_cell.py:316-330)
// ---------------------------------------------------------------------------------------------
__global__ void gate_mix_kernel(const float* __restrict__ mnew, const float* __restrict__ z, const float* __restrict__ mprev,
                                size_t n, float* out) {   // newMemory * z + memory * (1 - z)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = mnew[i] * z[i] + mprev[i] * (1.0f - z[i]);
}
__global__ void gate_bwd_kernel(const float* __restrict__ dm, const float* __restrict__ z, const float* __restrict__ mnew,
                                const float* __restrict__ mprev, size_t n, float* dmnew, float* dprev, float* dzpre) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float g = dm[i], zz = z[i];
    dmnew[i] = g * zz;
    dprev[i] = g * (1.0f - zz);
    dzpre[i] = g * (mnew[i] - mprev[i]) * zz * (1.0f - zz);   // through the sigmoid
  }
}

struct SelfAttP {
  int B, d, nh;               // nh = history entries visible at this step (initial state + earlier steps)
  const float* sc;            // [B][d]  projected (continuous) control
  const float* C;             // [nh][B][d] controls history
  const float* M;             // [nh][B][d] memories history
  const float* w; const float* bias;
  float* att; int ld_att;     // [B][ld_att]
  float* smry;                // [B][d]
};
constexpr int SA_MAXH = 64;

__global__ __launch_bounds__(256) void self_attend_kernel(SelfAttP p) {
  __shared__ float s_l[SA_MAXH];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t Bd = (size_t)p.B * p.d;
  const float* sc = p.sc + (size_t)b * p.d;
  for (int j = wave; j < p.nh; j += 4) {
    const float* c = p.C + (size_t)j * Bd + (size_t)b * p.d;
    float part = 0.f;
    for (int k = lane * 4; k < p.d; k += 256) {
      const f32x4 cv = *reinterpret_cast<const f32x4*>(c + k);
      const f32x4 sv = *reinterpret_cast<const f32x4*>(sc + k);
      const f32x4 wv = *reinterpret_cast<const f32x4*>(p.w + k);
      part += (cv[0] * sv[0]) * wv[0] + (cv[1] * sv[1]) * wv[1] + (cv[2] * sv[2]) * wv[2] + (cv[3] * sv[3]) * wv[3];
    }
    part = wave_sum(part);
    if (lane == 0) s_l[j] = part + p.bias[0];
  }
  __syncthreads();
  if (tid == 0) {   // nh <= netLength + 1: a serial softmax is the cheapest correct thing
    float m = -INFINITY, sum = 0.f;
    for (int j = 0; j < p.nh; ++j) m = fmaxf(m, s_l[j]);
    for (int j = 0; j < p.nh; ++j) { s_l[j] = expf(s_l[j] - m); sum += s_l[j]; }
    const float inv = 1.0f / sum;
    for (int j = 0; j < p.nh; ++j) { s_l[j] *= inv; p.att[(size_t)b * p.ld_att + j] = s_l[j]; }
  }
  __syncthreads();
  for (int k = tid * 2; k < p.d; k += 512) {
    float o0 = 0.f, o1 = 0.f;
    for (int j = 0; j < p.nh; ++j) {
      const float2 mv = *reinterpret_cast<const float2*>(p.M + (size_t)j * Bd + (size_t)b * p.d + k);
      o0 = fmaf(s_l[j], mv.x, o0);
      o1 = fmaf(s_l[j], mv.y, o1);
    }
    p.smry[(size_t)b * p.d + k] = o0;
    p.smry[(size_t)b * p.d + k + 1] = o1;
  }
}

struct SelfAttBwdP {
  int B, d, nh;
  const float* dsmry; int ld_ds;   // [B][ld_ds] gradient wrt the self summary
  const float* sc; const float* C; const float* M; const float* w;
  const float* att; int ld_att;
  float* DMh;                      // [nh][B][d]  += att[j] * dsmry
  float* DCh;                      // [nh][B][d]  += dl[j] * w * sc
  float* dsc;                      // [B][d]      written
  float* dw_part;                  // [B][d]      written
  float* db_part;                  // [B]         written
};

__global__ __launch_bounds__(256) void self_attend_bwd_kernel(SelfAttBwdP p) {
  __shared__ float s_dl[SA_MAXH];
  __shared__ float s_att[SA_MAXH];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t Bd = (size_t)p.B * p.d;
  const float* ds = p.dsmry + (size_t)b * p.ld_ds;
  for (int j = wave; j < p.nh; j += 4) {
    const float* m = p.M + (size_t)j * Bd + (size_t)b * p.d;
    float part = 0.f;
    for (int k = lane * 4; k < p.d; k += 256) {
      const f32x4 mv = *reinterpret_cast<const f32x4*>(m + k);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(ds + k);
      part += mv[0] * gv[0] + mv[1] * gv[1] + mv[2] * gv[2] + mv[3] * gv[3];
    }
    part = wave_sum(part);
    if (lane == 0) { s_dl[j] = part; s_att[j] = p.att[(size_t)b * p.ld_att + j]; }
  }
  __syncthreads();
  if (tid == 0) {
    float dot = 0.f, dls = 0.f;
    for (int j = 0; j < p.nh; ++j) dot += s_att[j] * s_dl[j];
    for (int j = 0; j < p.nh; ++j) { s_dl[j] = s_att[j] * (s_dl[j] - dot); dls += s_dl[j]; }
    p.db_part[b] = dls;
  }
  __syncthreads();
  for (int k = tid * 2; k < p.d; k += 512) {
    const float2 sv = *reinterpret_cast<const float2*>(p.sc + (size_t)b * p.d + k);
    const float2 wv = *reinterpret_cast<const float2*>(p.w + k);
    const float2 gv = *reinterpret_cast<const float2*>(ds + k);
    float q0 = 0.f, q1 = 0.f;   // sum_j dl[j] C[j]
    for (int j = 0; j < p.nh; ++j) {
      const size_t off = (size_t)j * Bd + (size_t)b * p.d + k;
      const float2 cv = *reinterpret_cast<const float2*>(p.C + off);
      const float dl = s_dl[j], a = s_att[j];
      q0 = fmaf(dl, cv.x, q0);
      q1 = fmaf(dl, cv.y, q1);
      float2 dm = *reinterpret_cast<float2*>(p.DMh + off);
      dm.x += a * gv.x; dm.y += a * gv.y;
      *reinterpret_cast<float2*>(p.DMh + off) = dm;
      float2 dc = *reinterpret_cast<float2*>(p.DCh + off);
      dc.x += dl * (wv.x * sv.x); dc.y += dl * (wv.y * sv.y);
      *reinterpret_cast<float2*>(p.DCh + off) = dc;
    }
    p.dsc[(size_t)b * p.d + k] = q0 * wv.x;
    p.dsc[(size_t)b * p.d + k + 1] = q1 * wv.y;
    p.dw_part[(size_t)b * p.d + k] = q0 * sv.x;
    p.dw_part[(size_t)b * p.d + k + 1] = q1 * sv.y;
  }
}

// out[k][j] = sum_r x[r][k] g[r][j]  for a few dozen rows and any n: the classifier's weight gradients
__global__ void outer_sum_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg, int rows, int K, int J,
                                 float* out) {
  const size_t total = (size_t)K * J;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = i / J, j = i - (size_t)k * J;
    float acc = 0.f;
    for (int r = 0; r < rows; ++r) acc = fmaf(x[(size_t)r * ldx + k], g[(size_t)r * ldg + j], acc);
    out[i] = acc;
  }
}

// dst[r][j] = src[r][j] (j < n_src) else 0, row stride ld_dst   (pads [B, answers] to a multiple of 16 columns)
__global__ void pad_cols_kernel(const float* __restrict__ src, int n_src, int rows, int ld_dst, float* dst) {
  const int n = rows * ld_dst;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int r = i / ld_dst, j = i - r * ld_dst;
    dst[i] = j < n_src ? src[(size_t)r * n_src + j] : 0.f;
  }
}
// dst[r][j] = src[r*ld_src + j] for j < n   (the inverse crop)
__global__ void crop_cols_kernel(const float* __restrict__ src, int ld_src, int rows, int n, float* dst) {
  const int t = rows * n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < t; i += gridDim.x * blockDim.x) {
    const int r = i / n, j = i - r * n;
    dst[i] = src[(size_t)r * ld_src + j];
  }
}

// ---------------------------------------------------------------------------------------------
// stem CNN helpers (model.py:165-204, ops.py:380-438): halo-padded NHWC staging.
// A [B][H*W][C] tensor is copied into [B][(H+2)*(W+2)][C] with a zero border, so that every 3x3 tap of
// the implicit-GEMM convolution is a plain row offset and never needs a bounds check.
// ---------------------------------------------------------------------------------------------
struct PadP { int B, N, w, wp, np, C; };
// dst_pad[b][pp(n)] = dropout(src[b][n]);  optional keep bits indexed like src ([B*N][C/32])
__global__ __launch_bounds__(256) void pad_drop_kernel(const float* __restrict__ src, PadP q, uint32_t key, uint32_t thr24, float inv_keep,
                                                      uint32_t first, float* dst, uint32_t* bits) {
  const int c4n = q.C >> 2;
  const size_t total = (size_t)q.B * q.np * c4n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int h = q.N / q.w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ((total + 7) & ~(size_t)7); i += stride) {
    const bool in = i < total;
    const size_t pos = in ? i / c4n : 0;
    const int c4 = in ? (int)(i - pos * c4n) : 0;
    const int b = (int)(pos / q.np), pq = (int)(pos - (size_t)b * q.np);
    const int y = pq / q.wp, x = pq - y * q.wp;
    const bool interior = in && y >= 1 && y <= h && x >= 1 && x <= q.w;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    uint32_t nib = 0;
    size_t sidx = 0;
    if (interior) {
      sidx = ((size_t)b * q.N + (size_t)(y - 1) * q.w + (x - 1)) * c4n + c4;
      v = reinterpret_cast<const f32x4*>(src)[sidx];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool keep = keep_bit(first + (uint32_t)(sidx * 4) + e, key, thr24);
        nib |= (keep ? 1u : 0u) << e;
        v[e] = keep ? v[e] * inv_keep : 0.f;
      }
    }
    if (in) reinterpret_cast<f32x4*>(dst)[i] = v;
    uint32_t w = nib << (4 * (threadIdx.x & 7));
    w |= __shfl_xor(w, 1, 64);
    w |= __shfl_xor(w, 2, 64);
    w |= __shfl_xor(w, 4, 64);
    if (bits && interior && (threadIdx.x & 7) == 0) bits[sidx >> 3] = w;
  }
}
// gradient through the activation of a conv layer: dy = g * act'(o), written both plain ([B][N][C]: the
// weight-gradient operand) and halo-padded (the backward-data operand)
__global__ __launch_bounds__(256) void pad_mul_actgrad_kernel(const float* __restrict__ g, const float* __restrict__ o, int act, PadP q,
                                                             float* dst_plain, float* dst_pad) {
  const int c4n = q.C >> 2;
  const size_t total = (size_t)q.B * q.np * c4n;
  const int h = q.N / q.w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pos = i / c4n;
    const int c4 = (int)(i - pos * c4n);
    const int b = (int)(pos / q.np), pq = (int)(pos - (size_t)b * q.np);
    const int y = pq / q.wp, x = pq - y * q.wp;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y >= 1 && y <= h && x >= 1 && x <= q.w) {
      const size_t sidx = ((size_t)b * q.N + (size_t)(y - 1) * q.w + (x - 1)) * c4n + c4;
      const f32x4 gv = reinterpret_cast<const f32x4*>(g)[sidx];
      const f32x4 ov = reinterpret_cast<const f32x4*>(o)[sidx];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gv[e] * act_grad_from_out(act, ov[e]);
      reinterpret_cast<f32x4*>(dst_plain)[sidx] = v;
    }
    reinterpret_cast<f32x4*>(dst_pad)[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// question encoder (model.py:208-307, ops.py:749-911): embedding lookup + bidirectional
// BasicLSTMCell(h) under tf.nn.bidirectional_dynamic_rnn semantics (per-question lengths: the state
// is carried through and the output is zero past the end; the backward direction runs over the
// length-reversed question).
// ---------------------------------------------------------------------------------------------
// x[b][s][0:E] = dropout(embeddings[idx]) with embeddings = concat([zeros(1,E), emb]) (model.py:217); columns E..Ep-1 = 0
__global__ void embed_gather_kernel(const int32_t* __restrict__ idx, const float* __restrict__ emb, int rows, int E, int Ep, uint32_t row0,
                                    DropSpec ds, float* x) {
  const size_t total = (size_t)rows * Ep;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / Ep;
    const int c = (int)(i - r * Ep);
    float v = 0.f;
    if (c < E) {
      const int t = idx[r];
      if (t > 0) v = emb[(size_t)(t - 1) * E + c];
      v = drop_apply(v, (uint32_t)((row0 + r) * E + c), ds);
    }
    x[i] = v;
  }
}
// BasicLSTMCell (TF1): gates i, j, f, o ; c' = c * sigmoid(f + 1) + sigmoid(i) * tanh(j) ; h' = tanh(c') * sigmoid(o)
// ---------------------------------------------------------------------------------------------
// One LSTM time step of both directions in ONE launch: R = h_prev Wh on fp32 MFMA, then the cell.  The two launches this
// replaces (small_linear + lstm_cell_kernel) were ~6 us each of pure latency, 100 dependent pairs per forward pass.
// A workgroup owns 16 questions x 16 hidden units of one direction and therefore all four gates of its units: four waves
// split the reduction over h, an LDS combine in fixed order, then one (question, unit) per thread.
// Grid (h / 16, ceil(B / 16), 2).
// ---------------------------------------------------------------------------------------------
struct LstmStepP {
  int B, S, h, tau;
  const int32_t* len;      // [B]
  const float* Wh;         // [2] packed [h/16][4][4h][4] (pack format 0)
  const float* Zx;         // [2][B*S][4h]  x Wx + b  (all positions)
  float* hs; float* cs;    // [2][S+1][B][h]
  float* gates;            // [2][S][B][4h]
  float* out;              // [B][S][2h]
};
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStepP p) {
  __shared__ float red[4][4][16][17];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int u0 = blockIdx.x * 16, r0 = blockIdx.y * 16, dir = blockIdx.z;
  const int h = p.h, G = 4 * h;
  const size_t Bh = (size_t)p.B * h;
  const float* hp = p.hs + (size_t)(dir * (p.S + 1) + p.tau) * Bh;
  const float* Wd = p.Wh + (size_t)dir * h * G;
  const int rowc = min(r0 + li, p.B - 1);
  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nQ = h >> 4;
  constexpr int PF = 4;                     // k groups in flight per wave (h = 256: the wave's whole share)
  // what the cell needs behind the product -- this thread's (question, unit): length, previous state, the input projection's
  // four gate columns -- is requested BEFORE the product, so that its round trip does not start behind the barrier
  const int crow = tid >> 4, cuu = tid & 15;
  const int cb = min(r0 + crow, p.B - 1), cu = u0 + cuu;
  const int cL = min(max(p.len[cb], 0), p.S);      // never index past the padded question (host validates too)
  const int cpos = dir == 0 ? p.tau : max(cL - 1 - p.tau, 0);
  const size_t cs0 = (size_t)(dir * (p.S + 1) + p.tau) * Bh + (size_t)cb * h + cu;
  const float hpv_pre = p.hs[cs0], cp_pre = p.cs[cs0];
  const float* Zpre = p.Zx + ((size_t)dir * p.B * p.S + (size_t)cb * p.S + cpos) * G;
  const float z0 = Zpre[cu], z1 = Zpre[h + cu], z2 = Zpre[2 * h + cu], z3 = Zpre[3 * h + cu];
  for (int Q0 = wave; Q0 < nQ; Q0 += 4 * PF) {
    f32x4 af[PF], bf[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int Q = min(Q0 + 4 * u, nQ - 1);
      af[u] = *reinterpret_cast<const f32x4*>(hp + (size_t)rowc * h + Q * 16 + lg * 4);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bf[u][g] = *reinterpret_cast<const f32x4*>(Wd + ((size_t)(Q * 4 + lg) * G + g * h + u0 + li) * 4);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (Q0 + 4 * u < nQ) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][e], bf[u][g][e], acc[g], 0, 0, 0);
      }
  }
  // accumulator map: unit = lane & 15, question = (lane >> 4) * 4 + e
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][g][lg * 4 + e][li] = acc[g][e];
  __syncthreads();
  const int row = tid >> 4, uu = tid & 15;
  const int b = r0 + row, u = u0 + uu;
  if (b >= p.B) return;
  float R[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) R[g] = ((red[0][g][row][uu] + red[1][g][row][uu]) + red[2][g][row][uu]) + red[3][g][row][uu];
  const int L = cL;
  const bool active = p.tau < L;
  const int pos = cpos;
  const size_t s0 = cs0, s1 = s0 + Bh;
  const float hpv = hpv_pre, cp = cp_pre;
  float hn = hpv, cn = cp;
  float gi = 0.f, gj = 0.f, gf = 0.f, go = 0.f;
  if (active) {
    gi = 1.0f / (1.0f + expf(-(R[0] + z0)));
    gj = tanhf(R[1] + z1);
    gf = 1.0f / (1.0f + expf(-(R[2] + z2 + 1.0f)));
    go = 1.0f / (1.0f + expf(-(R[3] + z3)));
    cn = cp * gf + gi * gj;
    hn = tanhf(cn) * go;
    p.out[((size_t)b * p.S + pos) * 2 * h + dir * h + u] = hn;
  }
  p.hs[s1] = hn;
  p.cs[s1] = cn;
  float* gp = p.gates + ((size_t)(dir * p.S + p.tau) * p.B + b) * G;
  gp[u] = gi; gp[h + u] = gj; gp[2 * h + u] = gf; gp[3 * h + u] = go;
}

// The backward step likewise: gate gradients of the workgroup's 16 questions (all 4h columns, recomputed by each of the h / 16
// workgroups that share the questions -- elementwise work, cheaper than a launch), then dh_prev = dG Wh^T for the workgroup's
// 16 units from the LDS tile.  The running dh / dc are double-buffered by the parity of tau (a workgroup reads what its
// neighbours would otherwise be overwriting); the workgroup with blockIdx.x == 0 writes dc, dG and dZ.
struct LstmStepBwdP {
  int B, S, h, tau;
  const int32_t* len;
  const float* WhT;        // [2] packed [4h/16][4][h][4] (pack format 0 of Wh^T)
  const float* cs;         // [2][S+1][B][h]
  const float* gates;      // [2][S][B][4h]
  const float* dout;       // [B][S][2h]
  const float* dh_in; const float* dc_in;     // [2][B][h] running state gradients after step tau + 1
  float* dh_out; float* dc_out;               // ... after this step
  float* dG;               // [2][S][B][4h]
  float* dZ;               // [2][B*S][4h] (zero-initialised)
};
constexpr int LSB_THREADS = 1024;      // 16 waves: the elementwise part is h / 64 (question, unit) pairs per thread, the product 4 k groups per wave at h = 256
__global__ __launch_bounds__(LSB_THREADS) void lstm_step_bwd_kernel(LstmStepBwdP p) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  const int h = p.h, G = 4 * h;
  float* sG = lsm;                          // [16][G + 4] gate gradients of the workgroup's questions
  float* sPass = sG + 16 * (G + 4);         // [16][16] dh that bypasses the recurrent product (questions that have ended)
  float* red = sPass + 256;                 // [16 waves][16][17]
  const int ldg = G + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int u0 = blockIdx.x * 16, r0 = blockIdx.y * 16, dir = blockIdx.z;
  const size_t Bh = (size_t)p.B * h;
  // the recurrent weights of this wave's k groups first: their L2 round trip runs under the cell's backward instead of
  // behind the barrier (the launch is latency, 100 of them in a row)
  const float* Wd = p.WhT + (size_t)dir * G * h;
  const int nQ = G >> 4;
  constexpr int PF = 4, NW = LSB_THREADS / 64;
  f32x4 bf0[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) bf0[u] = *reinterpret_cast<const f32x4*>(Wd + ((size_t)(min(wave + NW * u, nQ - 1) * 4 + lg) * h + u0 + li) * 4);
  // ---- the cell's backward for 16 questions x h units.  Every workgroup of a question block computes all of it (its
  //      product needs all 4h gate gradients) and WRITES the columns of its own 16 units: dc, dG, dZ
#pragma unroll 4
  for (int it = tid; it < 16 * h; it += LSB_THREADS) {
    const int row = it / h, u = it - row * h;
    const int b = r0 + row;
    float di = 0.f, dj = 0.f, df = 0.f, dO = 0.f, pass = 0.f, dc = 0.f;
    if (b < p.B) {
      const size_t i = (size_t)dir * Bh + (size_t)b * h + u;
      const int L = min(max(p.len[b], 0), p.S);
      const bool active = p.tau < L;
      const int pos = dir == 0 ? p.tau : L - 1 - p.tau;
      float dh = p.dh_in[i];
      dc = p.dc_in[i];
      pass = dh;
      if (active) {
        const float* ga = p.gates + ((size_t)(dir * p.S + p.tau) * p.B + b) * G;
        const float gi = ga[u], gj = ga[h + u], gf = ga[2 * h + u], go = ga[3 * h + u];
        const size_t r = (size_t)b * h + u;
        const float cp = p.cs[(size_t)(dir * (p.S + 1) + p.tau) * Bh + r];
        const float cn = p.cs[(size_t)(dir * (p.S + 1) + p.tau + 1) * Bh + r];
        dh += p.dout[((size_t)b * p.S + pos) * 2 * h + dir * h + u];
        const float tc = tanhf(cn);
        dO = dh * tc * go * (1.0f - go);
        const float dct = dc + dh * go * (1.0f - tc * tc);
        di = dct * gj * gi * (1.0f - gi);
        dj = dct * gi * (1.0f - gj * gj);
        df = dct * cp * gf * (1.0f - gf);
        dc = dct * gf;
        pass = 0.f;
        if (u >= u0 && u < u0 + 16) {
          float* z = p.dZ + ((size_t)dir * p.B * p.S + (size_t)b * p.S + pos) * G;
          z[u] = di; z[h + u] = dj; z[2 * h + u] = df; z[3 * h + u] = dO;
        }
      }
      if (u >= u0 && u < u0 + 16) {
        float* g = p.dG + ((size_t)(dir * p.S + p.tau) * p.B + b) * G;
        g[u] = di; g[h + u] = dj; g[2 * h + u] = df; g[3 * h + u] = dO;
        p.dc_out[i] = dc;
      }
    }
    float* sg = sG + row * ldg;
    sg[u] = di; sg[h + u] = dj; sg[2 * h + u] = df; sg[3 * h + u] = dO;
    if (u >= u0 && u < u0 + 16) sPass[row * 16 + (u - u0)] = pass;
  }
  __syncthreads();
  // ---- dh_prev[16 questions][16 units] = dG[16][4h] WhT[4h][units]
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int Q0 = wave; Q0 < nQ; Q0 += NW * PF) {
    f32x4 af[PF], bf[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int Q = min(Q0 + NW * u, nQ - 1);
      af[u] = *reinterpret_cast<const f32x4*>(sG + li * ldg + Q * 16 + lg * 4);
      bf[u] = Q0 == wave ? bf0[u] : *reinterpret_cast<const f32x4*>(Wd + ((size_t)(Q * 4 + lg) * h + u0 + li) * 4);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (Q0 + NW * u < nQ) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][e], bf[u][e], acc, 0, 0, 0);
      }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[(wave * 16 + lg * 4 + e) * 17 + li] = acc[e];
  __syncthreads();
  if (tid >= 256) return;
  const int row = tid >> 4, uu = tid & 15;
  const int b = r0 + row;
  if (b >= p.B) return;
  float v = red[row * 17 + uu];
#pragma unroll
  for (int w = 1; w < NW; ++w) v += red[(w * 16 + row) * 17 + uu];        // fixed order
  p.dh_out[(size_t)dir * Bh + (size_t)b * h + u0 + uu] = v + sPass[row * 16 + uu];
}
inline size_t lstm_step_bwd_lds(int h) { return ((size_t)16 * (4 * h + 4) + 256 + (LSB_THREADS / 64) * 16 * 17) * sizeof(float); }

// d emb[v-1][c] = sum over tokens == v of dropmask * dx[r][c]   (one workgroup per vocabulary row: no atomics).
// The rows that hold token v are found 256 at a time by all threads (ballot + ordered compaction into LDS), then summed in
// ascending row order: the scan is rows / 256 short passes instead of `rows` sequential compares per thread.
__global__ __launch_bounds__(256) void embed_grad_kernel(const int32_t* __restrict__ idx, const float* __restrict__ dx, int rows, int E, int Ep,
                                                        uint32_t row0, DropSpec ds, float* demb) {
  __shared__ int list[256];
  __shared__ int wcount[4];
  const int v = blockIdx.x + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int CMAX = 4;                           // columns per thread held in registers: blockIdx.y selects a chunk of 1024 columns
  const int c0 = blockIdx.y * 1024;
  float acc[CMAX];
#pragma unroll
  for (int k = 0; k < CMAX; ++k) acc[k] = 0.f;
  for (int r0 = 0; r0 < rows; r0 += 256) {
    const int r = r0 + tid;
    const bool m = r < rows && idx[r] == v;
    const unsigned long long mask = __ballot(m);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wcount[wave] = __popcll(mask);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { base += w < wave ? wcount[w] : 0; total += wcount[w]; }
    if (m) list[base + before] = r;
    __syncthreads();
    for (int k = 0; k < total; ++k) {                // ascending rows
      const int rr = list[k];
#pragma unroll
      for (int j = 0; j < CMAX; ++j) {
        const int c = c0 + tid + 256 * j;
        if (c < E) acc[j] += drop_apply(dx[(size_t)rr * Ep + c], (uint32_t)((row0 + rr) * E + c), ds);
      }
    }
    __syncthreads();                                 // list / wcount are rewritten by the next pass
  }
#pragma unroll
  for (int j = 0; j < CMAX; ++j) {
    const int c = c0 + tid + 256 * j;
    if (c < E) demb[(size_t)(v - 1) * E + c] = acc[j];
  }
}

__global__ void add_bias_kernel(const float* __restrict__ bias, int rows, int n, float* x) {
  const int t = rows * n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < t; i += gridDim.x * blockDim.x) x[i] += bias[i % n];
}

// ---------------------------------------------------------------------------------------------
// optimizer step over one flat fp32 buffer (model.py:615-669): global-norm clip -> Adam -> EMA.
// Two launches: per-workgroup partial sums of g^2, then every workgroup folds the (<= 1024) partials in
// a fixed order and updates its slice.  HBM-bound: reads g, p, m, v, ema and writes p, m, v, ema once.
// ---------------------------------------------------------------------------------------------
constexpr int OPT_BLOCKS = 1024;
__global__ __launch_bounds__(256) void opt_sumsq_kernel(const float* __restrict__ g, size_t n, float* part) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s = fmaf(g[i], g[i], s);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
struct OptP {
  size_t n;
  float* p; const float* g; float* m; float* v; float* ema;
  float lr_t, beta1, beta2, eps, clip, ema_decay;   // lr_t already holds sqrt(1-b2^t)/(1-b1^t); clip <= 0: off; ema_decay < 0: off
  const float* part; int nparts;
  float* norm_out;                                   // [1] global gradient norm (before clipping)
};
__global__ __launch_bounds__(256) void opt_apply_kernel(OptP q) {
  __shared__ float s_norm;
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < q.nparts; ++i) t += q.part[i];
    s_norm = sqrtf(t);
    if (blockIdx.x == 0 && q.norm_out) q.norm_out[0] = s_norm;
  }
  __syncthreads();
  // tf.clip_by_global_norm: g * clip / max(norm, clip)
  const float scale = q.clip > 0.f ? q.clip / fmaxf(s_norm, q.clip) : 1.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < q.n; i += (size_t)gridDim.x * 256) {
    const float g = q.g[i] * scale;
    const float m = q.beta1 * q.m[i] + (1.0f - q.beta1) * g;
    const float v = q.beta2 * q.v[i] + (1.0f - q.beta2) * g * g;
    const float p = q.p[i] - q.lr_t * m / (sqrtf(v) + q.eps);
    q.m[i] = m; q.v[i] = v; q.p[i] = p;
    if (q.ema_decay >= 0.f) q.ema[i] = q.ema[i] - (1.0f - q.ema_decay) * (q.ema[i] - p);   // tf.train.ExponentialMovingAverage
  }
}

// dy[b][k] = sum over parts   (S_b kernel leaves 2*d/128 partials)
// ... optionally also dst2 = sum * act'(o) (o: the activation's OUTPUT): the mul_actgrad launch that would follow
__global__ void sum_parts_kernel(const float* __restrict__ part, int nparts, size_t n, float* dst, const float* __restrict__ o = nullptr,
                                 int act = 0, float* dst2 = nullptr) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = part[i];
    for (int q = 1; q < nparts; ++q) s += part[(size_t)q * n + i];
    dst[i] = s;
    if (dst2) dst2[i] = s * act_grad_from_out(act, o[i]);
  }
}

// the per-question ends of stage B0's column sums: dc[q] += sum over the segments of the question's tiles, db_k partial
struct DcReduceP {
  int B, N, d;
  const float* dc_part;     // [tiles][3][d]
  const float* dls_part;    // [tiles][3]
  float* dc;                // [B][d] accumulated in place
  float* dbk_part;          // [B]
  size_t part_step, dls_step, dc_step, dbk_step;      // blockIdx.y = step: floats between the steps' buffers
  const float* dy_part;     // [tiles][3][d] (chain_bwd_kernel stage B2) or null
  float* dy;                // [B][d] written
  int tile_shift;           // log2 of the rows per tile (chain_tile_shift)
};
__global__ __launch_bounds__(128) void dc_reduce_kernel(DcReduceP p) {
  const int q = blockIdx.x;
  p.dc_part += blockIdx.y * p.part_step; p.dls_part += blockIdx.y * p.dls_step;
  p.dc += blockIdx.y * p.dc_step; p.dbk_part += blockIdx.y * p.dbk_step;
  const uint32_t r0 = (uint32_t)q * (uint32_t)p.N, r1 = r0 + (uint32_t)p.N - 1;      // (rows < 2^31: 32-bit divisions, no 64-bit software routine)
  const int ts = p.tile_shift;
  const int t0 = (int)(r0 >> ts), t1 = (int)(r1 >> ts);
  auto seg_of = [&](int t) { return q - (int)(((uint32_t)t << ts) / (uint32_t)p.N); };   // 0: the question that owns the tile's first row
  for (int c4 = threadIdx.x * 4; c4 < p.d; c4 += 512) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int t = t0; t <= t1; ++t)                          // fixed order
      s += *reinterpret_cast<const f32x4*>(p.dc_part + ((size_t)t * 3 + seg_of(t)) * p.d + c4);
    f32x4* dst = reinterpret_cast<f32x4*>(p.dc + (size_t)q * p.d + c4);
    *dst = *dst + s;
    if (p.dy_part) {
      f32x4 sy = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int t = t0; t <= t1; ++t)                        // fixed order
        sy += *reinterpret_cast<const f32x4*>(p.dy_part + ((size_t)t * 3 + seg_of(t)) * p.d + c4);
      *reinterpret_cast<f32x4*>(p.dy + (size_t)q * p.d + c4) = sy;
    }
  }
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = t0; k <= t1; ++k) t += p.dls_part[(size_t)k * 3 + seg_of(k)];
    p.dbk_part[q] = t;
  }
}

}  // namespace macx
