// macx_wgrad_h2.hip.h -- the backward kernels that CONSUME H2 tensors along their rows:
//   read_att_bwd_h2_kernel   softmax backward + the elementwise chain down to dI2, I2 in / dI2 out as H2
//   wgrad_h2_kernel          weight-gradient contractions  C[k][j] = sum_m A[m][k] G[m][j]  over H2 operands
//   sb_h2_kernel             per-question interaction gradient S_b = X_b^T dI1_b (dW1a, dW1b, dy) over H2 operands
// The contractions reduce over ROWS, so a lane's MFMA fragment is 8 consecutive rows of one column while an H2 slot is 8
// consecutive columns of one row: the producer waves transpose in registers while staging (one v_perm_b32 per two
// elements) and bring every row to the tensor's common exponent with one v_pk_mul_f16 by an exact power of two (rows far
// below the largest lose low bits exactly as their share of the sum warrants).  Two fp16 planes, three MFMA terms.
// Tiling, slabs and the fixed-order slab reduction are those of macx_wgrad6.cuh.
#pragma once
#include "macx_gemm_tn.cuh"
#include "macx_h2.hip.h"

namespace macx {

__device__ __forceinline__ float sum16_h2(float v) {     // sum over the 16 lanes that share lane >> 4
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ uint32_t pk_mul_f16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2_t, a) * __builtin_bit_cast(f16x2_t, b));
}
// {f, f} as packed fp16 for f = 2^k, k <= 0 (k < -24 -> 0)
__device__ __forceinline__ uint32_t pk_pow2_f16(int k) {
  const float f = k < -30 ? 0.f : h2_pow2(k);
  return pk_f16(f, f);
}

// ---------------------------------------------------------------------------------------------------------------
// read-unit attention backward (the same arithmetic as read_att_bwd_kernel in macx_small.cuh); one workgroup per
// (question, 128-column block), a row's 128 columns live in the 32 lanes of a half-wave.
// ---------------------------------------------------------------------------------------------------------------
struct ReadAttBwdH2P {
  int B, N, d;
  const float* att;      // [B][N]
  const float* da;       // [B][N]
  H2View I2;             // [B*N][d]
  const float* c;        // [B][d]
  const float* wk;       // [d]
  int act;
  const uint8_t* bytes;  // keep bytes of the SITE_READ_ATT mask, slot order [d/8][Rp]; null = keep all
  float inv_keep;
  H2View dI2;            // [B*N][d]
  float* dc;             // [B][d]  accumulated in place
  float* dwk_part;       // [B][d]
  float* db2_part;       // [B][d]
  float* dbk_part;       // [B]
  int* qmin;             // [B][d/128] common (minimum) exponent of the question's dI2 rows, written
};

constexpr int RABH_THREADS = 1024;
constexpr int RABH_RG = RABH_THREADS / 32;
__global__ __launch_bounds__(RABH_THREADS) void read_att_bwd_h2_kernel(ReadAttBwdH2P p) {
  __shared__ float s_dl[1024];
  __shared__ float s_red[RABH_THREADS / 64];
  __shared__ f32x4 s_acc[3][RABH_RG][32];
  const int b = blockIdx.x, slab = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NWV = RABH_THREADS / 64;
  float dot = 0.f;
  for (int n = tid; n < p.N; n += RABH_THREADS) dot += p.att[(size_t)b * p.N + n] * p.da[(size_t)b * p.N + n];
  dot = wave_sum(dot);
  if (lane == 0) s_red[wave] = dot;
  __syncthreads();
  dot = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) dot += s_red[w];
  __syncthreads();
  float dls = 0.f;
  for (int n = tid; n < p.N; n += RABH_THREADS) {
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
  const int kg = slab * 16 + (c4 >> 1);                 // slot column of this lane's 4 values; (c4 & 1) selects the half
  const size_t hoff = (size_t)(c4 & 1) * 8;
  const f32x4 cv = *reinterpret_cast<const f32x4*>(p.c + (size_t)b * p.d + col);
  const f32x4 wv = *reinterpret_cast<const f32x4*>(p.wk + col);
  const size_t iRp = p.I2.Rp(), oRp = p.dI2.Rp();
  const char* i0 = p.I2.plane(0);
  const size_t ipb = p.I2.plane_bytes(), opb = p.dI2.plane_bytes();
  char* o0 = p.dI2.plane(0);
  const int icb = p.I2.cb(), ocb = p.dI2.cb();
  f32x4 a_dc = {0.f, 0.f, 0.f, 0.f}, a_dw = {0.f, 0.f, 0.f, 0.f}, a_db = {0.f, 0.f, 0.f, 0.f};
  int emin = 127;
  for (int n0 = 0; n0 < p.N; n0 += RABH_RG) {
    const int n = n0 + rg;
    const bool ok = n < p.N;
    const size_t row = (size_t)b * p.N + min(n, p.N - 1);
    const char* src = i0 + ((size_t)kg * iRp + row) * 16 + hoff;
    float i2[4];
    h2_join4(*reinterpret_cast<const u32x2*>(src), *reinterpret_cast<const u32x2*>(src + ipb),
             h2_pow2(-(int)p.I2.exps()[row * icb + slab]), i2);
    const float dl = s_dl[min(n, p.N - 1)];
    uint32_t bits = 0xFu;
    if (p.bytes) bits = (uint32_t)p.bytes[(size_t)kg * iRp + row] >> (4 * (c4 & 1));
    float o[4];
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float zv = i2[e] * cv[e];
      const float g = act_apply(p.act, zv);
      const float f = ((bits >> e) & 1u) ? p.inv_keep : 0.f;
      const float dz = (dl * wv[e]) * f * act_grad_from_out(p.act, g);
      o[e] = dz * cv[e];                                  // dI2 = dZ * c
      if (ok) {
        a_dw[e] = fmaf(dl, g * f, a_dw[e]);               // dw_k += dl * dropped(G)
        a_dc[e] = fmaf(dz, i2[e], a_dc[e]);               // dc += dZ * I2
        a_db[e] += o[e];
      }
      m = fmaxf(m, fabsf(o[e]));
    }
    // the row's 128 columns sit in this half-wave: its exponent needs no shared memory
#pragma unroll
    for (int s = 16; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    const int ex = h2_exponent(m);
    const float sc = h2_pow2(ex);
    const uint32_t h0 = pk_f16(o[0] * sc, o[1] * sc), h1 = pk_f16(o[2] * sc, o[3] * sc);
    const f32x2_t b0 = unpk_f16(h0), b1 = unpk_f16(h1);
    const uint32_t l0 = pk_f16(o[0] * sc - b0[0], o[1] * sc - b0[1]), l1 = pk_f16(o[2] * sc - b1[0], o[3] * sc - b1[1]);
    if (ok) {
      char* dst = o0 + ((size_t)kg * oRp + row) * 16 + hoff;
      *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(dst + opb) = u32x2{l0, l1};
      if (c4 == 0) p.dI2.exps()[row * ocb + slab] = (int8_t)ex;
      emin = min(emin, ex);
    }
  }
  __syncthreads();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) emin = min(emin, __shfl_xor(emin, o, 64));
  if (lane == 0) s_red[wave] = __int_as_float(emin);
  __syncthreads();
  if (tid == 0) {
    int t = __float_as_int(s_red[0]);
#pragma unroll
    for (int w = 1; w < NWV; ++w) t = min(t, __float_as_int(s_red[w]));
    p.qmin[(size_t)b * ocb + slab] = t;
  }
  s_acc[0][rg][c4] = a_dc;
  s_acc[1][rg][c4] = a_dw;
  s_acc[2][rg][c4] = a_db;
  __syncthreads();
  if (tid < 96) {
    const int which = tid >> 5, cc4 = tid & 31;
    f32x4 t = s_acc[which][0][cc4];
#pragma unroll
    for (int g = 1; g < RABH_RG; ++g) t += s_acc[which][g][cc4];
    float* dst = (which == 0 ? p.dc : which == 1 ? p.dwk_part : p.db2_part) + (size_t)b * p.d + slab * 128 + cc4 * 4;
    if (which == 0) t += *reinterpret_cast<const f32x4*>(dst);
    *reinterpret_cast<f32x4*>(dst) = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weight-gradient contraction over H2 operands.  Reduction row m lives in tensor m / rows_per_tensor of a sequence of
// H2 tensors `*_stride` bytes apart (the per-step activations kept by the forward / backward pass); A may instead be ONE
// tensor reused by every step (a_mod = its rows: the undropped knowledge base).
// ---------------------------------------------------------------------------------------------------------------
struct TnH2P {
  int M;                 // reduction rows over all tensors
  int Kd, Jd;            // output dims (multiples of 128)
  int nsplit, rows_per_split;
  int R;                 // rows per H2 tensor
  const char* A; size_t a_stride; int a_mod;     // a_mod > 0: A row of reduction row m is m % a_mod of tensor 0
  const char* G; size_t g_stride;
  const int* ecomA;      // [Kd/128] common exponents (h2_min_exp_kernel)
  const int* ecomG;      // [Jd/128]
  float* part;           // [nsplit][Kd][Jd]
};

constexpr int WH_GS = 128 * 16 + 32;        // bytes between m-groups of a plane (A operand, 128 columns)
constexpr int WH_APL = 4 * WH_GS;
constexpr int WH_AOP = 2 * WH_APL;

template <int JW>
constexpr int wh_stage_bytes() { return WH_AOP + 2 * 4 * (JW * 128 * 16 + 32); }

template <int JW>
__global__ __launch_bounds__(512) void wgrad_h2_kernel(TnH2P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  constexpr int JT = JW * T_TILE;
  constexpr int GG = JW * 128 * 16 + 32;
  constexpr int GPL = 4 * GG;
  constexpr int STAGE = WH_AOP + 2 * GPL;

  const int ntj = p.Jd / JT;
  const int ntk = p.Kd / T_TILE;
  const int ntile = ntj * ntk;
  const int nblk = gridDim.x;
  int v = blockIdx.x;
  if ((nblk & 7) == 0) v = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int split = v / ntile;
  const int tile = v % ntile;
  const int tk = tile / ntj, tj = tile % ntj;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int nchunk = (m_end - m_begin + 31) >> 5;
  const int nloop = (nchunk + 2) / 3 * 3;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  typedef uint32_t gw_t __attribute__((ext_vector_type(JW)));      // a lane's 2 JW columns of one G row and plane

  const H2View a0{const_cast<char*>(p.A), p.R, p.Kd}, g0{const_cast<char*>(p.G), p.R, p.Jd};
  const size_t Rp = a0.Rp();
  const size_t apb = a0.plane_bytes(), gpb = g0.plane_bytes();
  const int acb = a0.cb(), gcb = g0.cb();

  if (wave < 4) {
    // ================= producer waves: H2 slots -> common exponent -> in-register transpose -> LDS planes =================
    const int mg = wave;
    // this lane's 2 (A) / 2 JW (G) columns: slot column and byte offset inside the slot
    const size_t a_lane = ((size_t)(tk * 16 + (lane >> 2)) * Rp) * 16 + (lane & 3) * 4;
    const int gcol = tj * JT + 2 * JW * lane;
    const size_t g_lane = ((size_t)(gcol >> 3) * Rp) * 16 + (gcol & 7) * 2;
    const int gblk = gcol >> 7;                                     // 128-column block of this lane's G columns
    const int eA = p.ecomA[tk], eG = p.ecomG[gblk];
    uint32_t ra[3][2][8];
    gw_t rg[3][2][8];
    uint32_t fa[3][8], fg[3][8];                                    // per-row factors {f, f} (0 for rows past the end)
    auto load = [&](auto slot_c, int ch) __attribute__((always_inline)) {
      constexpr int SL = decltype(slot_c)::value;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int m = m_begin + ch * 32 + mg * 8 + r;
        const bool ok = m < m_end;
        const int mc = min(m, p.M - 1);
        const int ti = mc / p.R, rr = mc - ti * p.R;               // tensor of the sequence, row inside it (wave-uniform)
        const int ar = p.a_mod ? (mc % p.a_mod) : rr;
        const char* ab = p.A + (p.a_mod ? 0 : (size_t)ti * p.a_stride);
        const char* gb = p.G + (size_t)ti * p.g_stride;
        const char* as = ab + a_lane + (size_t)ar * 16;
        const char* gs = gb + g_lane + (size_t)rr * 16;
        ra[SL][0][r] = *reinterpret_cast<const uint32_t*>(as);
        ra[SL][1][r] = *reinterpret_cast<const uint32_t*>(as + apb);
        rg[SL][0][r] = *reinterpret_cast<const gw_t*>(gs);
        rg[SL][1][r] = *reinterpret_cast<const gw_t*>(gs + gpb);
        const int ea = (int)reinterpret_cast<const int8_t*>(ab + 2 * apb)[(size_t)ar * acb + tk];
        const int eg = (int)reinterpret_cast<const int8_t*>(gb + 2 * gpb)[(size_t)rr * gcb + gblk];
        fa[SL][r] = ok ? pk_pow2_f16(eA - ea) : 0u;
        fg[SL][r] = ok ? pk_pow2_f16(eG - eg) : 0u;
      }
    };
    auto store = [&](auto slot_c, int ch) __attribute__((always_inline)) {
      constexpr int SL = decltype(slot_c)::value;
      char* dA = lds + (ch & 1) * STAGE + mg * WH_GS + (2 * lane) * 16;
      char* dG = lds + (ch & 1) * STAGE + WH_AOP + mg * GG + (2 * JW * lane) * 16;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        uint32_t w[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) w[r] = fa[SL][r] ? pk_mul_f16(ra[SL][pl][r], fa[SL][r]) : 0u;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const uint32_t sel = c ? 0x07060302u : 0x05040100u;
          u32x4 s;
#pragma unroll
          for (int h = 0; h < 4; ++h) s[h] = __builtin_amdgcn_perm(w[2 * h + 1], w[2 * h], sel);
          *reinterpret_cast<u32x4*>(dA + pl * WH_APL + c * 16) = s;
        }
#pragma unroll
        for (int q = 0; q < JW; ++q) {
#pragma unroll
          for (int r = 0; r < 8; ++r) w[r] = fg[SL][r] ? pk_mul_f16(rg[SL][pl][r][q], fg[SL][r]) : 0u;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const uint32_t sel = c ? 0x07060302u : 0x05040100u;
            u32x4 s;
#pragma unroll
            for (int h = 0; h < 4; ++h) s[h] = __builtin_amdgcn_perm(w[2 * h + 1], w[2 * h], sel);
            *reinterpret_cast<u32x4*>(dG + pl * GPL + (2 * q + c) * 16) = s;
          }
        }
      }
    };
    load(S0{}, 0);
    load(S1{}, 1);
    load(S2{}, 2);
    store(S0{}, 0);
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < nloop; ch += 3) {
      load(S0{}, ch + 3); store(S1{}, ch + 1); __syncthreads();
      load(S1{}, ch + 4); store(S2{}, ch + 2); __syncthreads();
      load(S2{}, ch + 5); store(S0{}, ch + 3); __syncthreads();
    }
  } else {
    // ================= consumer waves: LDS fragments -> MFMA; wave tile 64 x (64 JW) =================
    const int cw = wave - 4;
    const int wr = cw >> 1, wc = cw & 1;
    constexpr int NC = 4 * JW;
    f32x4 acc[4][NC];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) __attribute__((always_inline)) {
      const char* sa = lds + buf * STAGE + (lane >> 4) * WH_GS + (wr * 64 + (lane & 15)) * 16;
      const char* sg = lds + buf * STAGE + WH_AOP + (lane >> 4) * GG + (wc * 64 * JW + (lane & 15)) * 16;
      u32x4 af[2][4];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int t = 0; t < 4; ++t) af[pl][t] = *reinterpret_cast<const u32x4*>(sa + pl * WH_APL + t * 256);
      // smallest terms first: G_lo x A_hi ; G_hi x {A_lo, A_hi}
#pragma unroll
      for (int bp = 1; bp >= 0; --bp) {
        u32x4 gf[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) gf[c] = *reinterpret_cast<const u32x4*>(sg + bp * GPL + c * 256);
#pragma unroll
        for (int ap = 1 - bp; ap >= 0; --ap)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[t][c] = mfma_f16(af[ap][t], gf[c], acc[t][c]);
      }
    };
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < nloop; ++ch) {
      compute(ch & 1);
      __syncthreads();
    }
    // a consumer wave's 64 JW columns lie inside one 128-column block of G
    const float sc = h2_unscale(p.ecomA[tk], p.ecomG[(tj * JT + wc * 64 * JW) >> 7]);
    float* out = p.part + (size_t)split * p.Kd * p.Jd;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = tk * T_TILE + wr * 64 + t * 16 + (lane >> 4) * 4 + e;
          const int j = tj * JT + wc * 64 * JW + c * 16 + (lane & 15);
          out[(size_t)k * p.Jd + j] = acc[t][c][e] * sc;
        }
  }
}

inline int wgrad_h2_jw(int Jd) { return (Jd % 256 == 0) ? 2 : 1; }

template <int JW>
inline hipError_t wgrad_h2_launch_t(const TnH2P& p, hipStream_t st) {
  auto kern = wgrad_h2_kernel<JW>;
  constexpr size_t lds = 2 * wh_stage_bytes<JW>();
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  const int grid = (p.Kd / T_TILE) * (p.Jd / (JW * T_TILE)) * p.nsplit;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p);
  return hipGetLastError();
}
inline hipError_t wgrad_h2_launch(const TnH2P& p, hipStream_t st) {
  return wgrad_h2_jw(p.Jd) == 2 ? wgrad_h2_launch_t<2>(p, st) : wgrad_h2_launch_t<1>(p, st);
}

// ---------------------------------------------------------------------------------------------------------------
// Per-question interaction gradient  S_b = X_b^T dI1_b  over H2 operands (see sb_wgrad_kernel in macx_gemm_tn.cuh for what
// S_b is for):  dW1a += diag(y_b) S_b,  dW1b += S_b,  dy[b][k] = sum_j W1a[k][j] S_b[k][j].  4 waves, one per SIMD, each
// stages and multiplies; the rows of a question are brought to the question's own common exponents (qmin arrays, written by
// the kernels that produced X and dI1).
// ---------------------------------------------------------------------------------------------------------------
struct SbH2P {
  int B, N, d;
  int qpg;
  H2View X;                // [B*N][d]
  H2View dI1;              // [B*N][d]
  const int* qminX;        // [B][d/128] common exponent of each question's rows (written by the producer of X)
  const int* qminG;        // [B][d/128] ... of dI1
  const float* y;          // [B][d]
  const float* W1a;        // [d][d] row-major (k, j)
  float* dW1a_part;        // [ngroup][d][d]
  float* dW1b_part;
  float* dy_part;          // [2*d/128][B][d]
};

constexpr int SBH_STAGE = 2 * WH_AOP;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void sb_h2_kernel(SbH2P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);

  const int nt = p.d / T_TILE;
  const int ntile = nt * nt;
  const int nblk = gridDim.x;
  int v = blockIdx.x;
  if ((nblk & 7) == 0) v = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int group = v / ntile;
  const int tile = v % ntile;
  const int tk = tile / nt, tj = tile % nt;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int mg = wave;

  f32x4 accS[4][4], accA[4][4], accB[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) accS[t][c] = accA[t][c] = accB[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunk = (p.N + 31) >> 5;
  const int b_begin = group * p.qpg;
  const int b_end = min(p.B, b_begin + p.qpg);
  const int nq = b_end - b_begin;
  const int total = nq * nchunk;

  const size_t Rp = p.X.Rp();
  const size_t xpb = p.X.plane_bytes(), gpb = p.dI1.plane_bytes();
  const int xcb = p.X.cb(), gcb = p.dI1.cb();
  const int8_t* xe = p.X.exps();
  const int8_t* ge = p.dI1.exps();
  const size_t x_lane = ((size_t)(tk * 16 + (lane >> 2)) * Rp) * 16 + (lane & 3) * 4;
  const size_t g_lane = ((size_t)(tj * 16 + (lane >> 2)) * Rp) * 16 + (lane & 3) * 4;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  uint32_t ra[2][2][8], rg[2][2][8];
  int e_row[2][8];                                      // (exponent of the X row) | (exponent of the dI1 row) << 8, both int8
  auto load = [&](auto slot_c, int s_raw) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
    const int s = min(s_raw, total - 1);
    const int qi = s / nchunk, ch = s - qi * nchunk;
    const size_t r0 = (size_t)(b_begin + qi) * p.N;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int n = min(ch * 32 + mg * 8 + r, p.N - 1);
      const char* xs = p.X.plane(0) + x_lane + (r0 + n) * 16;
      const char* gs = p.dI1.plane(0) + g_lane + (r0 + n) * 16;
      ra[SL][0][r] = *reinterpret_cast<const uint32_t*>(xs);
      ra[SL][1][r] = *reinterpret_cast<const uint32_t*>(xs + xpb);
      rg[SL][0][r] = *reinterpret_cast<const uint32_t*>(gs);
      rg[SL][1][r] = *reinterpret_cast<const uint32_t*>(gs + gpb);
      e_row[SL][r] = ((int)xe[(r0 + n) * xcb + tk] & 0xFF) | (((int)ge[(r0 + n) * gcb + tj] & 0xFF) << 8);
    }
  };
  auto store = [&](auto slot_c, int s_raw) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
    const int s = min(s_raw, total - 1);
    const int qi = s / nchunk, ch = s - qi * nchunk;
    const int nrow = ch * 32 + mg * 8;
    const int eX = p.qminX[(size_t)(b_begin + qi) * xcb + tk], eG = p.qminG[(size_t)(b_begin + qi) * gcb + tj];
    char* dst = lds + (s_raw & 1) * SBH_STAGE + mg * WH_GS + (2 * lane) * 16;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        uint32_t w[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int er = o ? (int)(int8_t)(e_row[SL][r] >> 8) : (int)(int8_t)e_row[SL][r];
          const uint32_t f = pk_pow2_f16((o ? eG : eX) - er);
          w[r] = (nrow + r < p.N) ? pk_mul_f16(o ? rg[SL][pl][r] : ra[SL][pl][r], f) : 0u;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const uint32_t sel = c ? 0x07060302u : 0x05040100u;
          u32x4 sl;
#pragma unroll
          for (int h = 0; h < 4; ++h) sl[h] = __builtin_amdgcn_perm(w[2 * h + 1], w[2 * h], sel);
          *reinterpret_cast<u32x4*>(dst + o * WH_AOP + pl * WH_APL + c * 16) = sl;
        }
      }
  };
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* sa = lds + buf * SBH_STAGE + (lane >> 4) * WH_GS + (wr * 64 + (lane & 15)) * 16;
    const char* sg = lds + buf * SBH_STAGE + WH_AOP + (lane >> 4) * WH_GS + (wc * 64 + (lane & 15)) * 16;
    u32x4 gf[2][4];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int c = 0; c < 4; ++c) gf[pl][c] = *reinterpret_cast<const u32x4*>(sg + pl * WH_APL + c * 256);
#pragma unroll
    for (int ap = 1; ap >= 0; --ap) {
      u32x4 af[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) af[t] = *reinterpret_cast<const u32x4*>(sa + ap * WH_APL + t * 256);
#pragma unroll
      for (int bp = 1 - ap; bp >= 0; --bp)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) accS[t][c] = mfma_f16(af[t], gf[bp][c], accS[t][c]);
    }
  };
  // question finished: fold S_b (in units of its two common exponents) into the three outputs and clear it
  auto consume = [&](int b) __attribute__((always_inline)) {
    const float sc = h2_unscale(p.qminX[(size_t)b * xcb + tk], p.qminG[(size_t)b * gcb + tj]);
    const float* yb = p.y + (size_t)b * p.d;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = tk * T_TILE + wr * 64 + t * 16 + (lane >> 4) * 4 + e;
        const float yk = yb[k];
        float dyp = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float sv = accS[t][c][e] * sc;
          accB[t][c][e] += sv;
          accA[t][c][e] = fmaf(yk, sv, accA[t][c][e]);
          dyp = fmaf(p.W1a[(size_t)k * p.d + tj * T_TILE + wc * 64 + c * 16 + (lane & 15)], sv, dyp);   // L2-resident, once per question
          accS[t][c][e] = 0.f;
        }
        dyp = sum16_h2(dyp);
        if ((lane & 15) == 0) p.dy_part[((size_t)(tj * 2 + wc) * p.B + b) * p.d + k] = dyp;
      }
  };

  if (total > 0) {
    load(S0{}, 0);
    load(S1{}, 1);
    store(S0{}, 0);
    __syncthreads();
    int qch = 0, b = b_begin;
    auto step = [&](auto mine, auto next, int s) __attribute__((always_inline)) {
      load(mine, s + 2);
      compute(s & 1);
      store(next, s + 1);
      if (++qch == nchunk) { consume(b); qch = 0; ++b; }
      __syncthreads();
    };
    int s = 0;
#pragma unroll 1
    for (; s + 2 <= total; s += 2) {
      step(S0{}, S1{}, s);
      step(S1{}, S0{}, s + 1);
    }
    if (s < total) step(S0{}, S1{}, s);
  }

  float* oa = p.dW1a_part + (size_t)group * p.d * p.d;
  float* ob = p.dW1b_part + (size_t)group * p.d * p.d;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = tk * T_TILE + wr * 64 + t * 16 + (lane >> 4) * 4 + e;
        const int j = tj * T_TILE + wc * 64 + c * 16 + (lane & 15);
        oa[(size_t)k * p.d + j] = accA[t][c][e];
        ob[(size_t)k * p.d + j] = accB[t][c][e];
      }
}

inline hipError_t sb_h2_launch(const SbH2P& p, hipStream_t st) {
  constexpr size_t lds = 2 * SBH_STAGE;
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(sb_h2_kernel), lds);
  if (e != hipSuccess) return e;
  const int nt = p.d / T_TILE;
  const int ngroup = (p.B + p.qpg - 1) / p.qpg;
  hipLaunchKernelGGL(sb_h2_kernel, dim3(nt * nt * ngroup), dim3(256), lds, st, p);
  return hipGetLastError();
}

}  // namespace macx
