// macx_wgrad6.hip.h -- the weight-gradient contractions  C[k][j] = sum_m A[m][k] * G[m][j]  (macx_gemm_tn.hip.h) on the
// bf16 matrix pipe with fp32-class numerics: the same exact 3-way bf16 operand split and six MFMA terms as
// macx_gemm6.hip.h.  Both operands are row-major over the reduction index m, and v_mfma_f32_16x16x32_bf16 wants 8
// consecutive m per lane, so the transpose happens in registers while staging: a thread loads an 8 (m) x 2 (columns)
// block (eight 8-byte loads, 512 B contiguous per row across a wave), splits the 16 values and writes, per column and
// plane, one 16-byte slot [m-group][column] -- conflict-free b128 stores, and fragments are conflict-free b128 reads.
//   workgroup: 128 x 128 output tile, 8 waves SPECIALISED: waves 0-3 produce (load, split, LDS store of stage c+1) while
//   waves 4-7 consume (fragments + MFMAs of stage c, wave tile 64 x 64 = 4 x 4 MFMA tiles) -- one of each per SIMD, so the
//   vector ALU work of the split runs beside the matrix pipe instead of in a separate phase of the same waves;
//   32 reduction rows per stage, two LDS stages, one barrier per stage.
// Determinism as in macx_gemm_tn.hip.h: a workgroup owns one (split, tile) slab, slabs are summed in a fixed order.
#pragma once
#include "macx_gemm6.hip.h"
#include "macx_gemm_tn.hip.h"

namespace macx {

constexpr int W6_GS = 128 * 16 + 32;       // bytes between m-groups of a plane
constexpr int W6_PLANE = 4 * W6_GS;
constexpr int W6_OPER = 3 * W6_PLANE;
constexpr int W6_STAGE = 2 * W6_OPER;

// JW = j-width of the output tile in units of 128 columns.  JW = 2 (128 x 256 tiles, consumer wave tile 64 x 128) doubles
// the MFMA work per barrier and reads the A operand half as often; used whenever Jd is a multiple of 256.
template <int JW>
constexpr int w6_stage_bytes() { return W6_OPER + 3 * 4 * (JW * 128 * 16 + 32); }

template <bool CONV, int JW>
__device__ __forceinline__ void wgrad6_body(const TnP& p, const int zidx, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  constexpr int JT = JW * T_TILE;                   // output tile width
  constexpr int GG = JW * 128 * 16 + 32;            // bytes between m-groups of a G plane
  constexpr int GPL = 4 * GG;                       // G plane
  constexpr int STAGE = W6_OPER + 3 * GPL;

  const int ntj = p.Jd / JT;
  const int ntk = p.Kd / T_TILE;
  const int ntile = ntj * ntk;
  int v = bid;
  if ((nblk & 7) == 0) v = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int split = v / ntile;
  const int tile = v % ntile;
  const int tk = tile / ntj, tj = tile % ntj;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int nchunk = (m_end - m_begin + 31) >> 5;
  const int nloop = (nchunk + 2) / 3 * 3;          // whole groups of three stages; the extra ones multiply zeros

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  typedef float gvec __attribute__((ext_vector_type(2 * JW)));     // a lane's columns of one G row

  if (wave < 4) {
    // ================= producer waves: global -> registers -> exact bf16 split -> LDS planes =================
    // wave w stages m-group w (8 rows) of BOTH operands; every row address is wave-uniform (scalar ALU), the lane
    // contributes its columns (2 of A, 2 JW of G).  Three register sets keep the loads of stages c+1..c+3 in flight.
    int conv_shift = 0, a_col0 = tk * T_TILE;
    if (CONV) {
      const int per = p.conv_cin / T_TILE;
      const int tap = tk / per;
      conv_shift = (tap / 3 - 1) * p.conv_wp + (tap - (tap / 3) * 3 - 1);
      a_col0 = (tk - tap * per) * T_TILE;
    }
    const int mg = wave;
    const float* baseA = p.A + a_col0 + 2 * lane;
    const float* baseG = p.G + (size_t)zidx * p.zG + tj * JT + 2 * JW * lane;
    int amod_row = (m_begin + mg * 8) % p.a_mod;   // A row of reduction row m is m % a_mod, kept incrementally
    f32x2_t ra[3][8];
    gvec rg[3][8];
    auto load = [&](auto slot_c, int ch) __attribute__((always_inline)) {
      constexpr int SL = decltype(slot_c)::value;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int m = m_begin + ch * 32 + mg * 8 + r;
        const int mc = min(m, p.M - 1);        // rows past the end are zeroed when the stage is split, not here
        int arow;
        if (CONV) {
          const int img = (int)__umulhi((uint32_t)mc, p.magic_n);
          const int n = mc - img * p.conv_n;
          const int yy = (int)__umulhi((uint32_t)n, p.magic_w);
          arow = img * p.conv_np + (yy + 1) * p.conv_wp + (n - yy * p.conv_w) + 1 + conv_shift;
        } else {
          arow = amod_row + r;
          while (arow >= p.a_mod) arow -= p.a_mod;     // a_mod may be smaller than a stage (tiny [B,d]-sized contractions)
        }
        ra[SL][r] = *reinterpret_cast<const f32x2_t*>(baseA + (size_t)arow * p.lda);
        rg[SL][r] = *reinterpret_cast<const gvec*>(baseG + (size_t)mc * p.ldg);
      }
      amod_row += 32;
      while (amod_row >= p.a_mod) amod_row -= p.a_mod;
    };
    auto store = [&](auto slot_c, int ch) __attribute__((always_inline)) {
      constexpr int SL = decltype(slot_c)::value;
      const int mrow = m_begin + ch * 32 + mg * 8;
      char* dA = lds + (ch & 1) * STAGE + mg * W6_GS + (2 * lane) * 16;
      char* dG = lds + (ch & 1) * STAGE + W6_OPER + mg * GG + (2 * JW * lane) * 16;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = (mrow + r < m_end) ? ra[SL][r][c] : 0.f;
        u32x4 s0, s1, s2;
        split8(x, s0, s1, s2);
        *reinterpret_cast<u32x4*>(dA + c * 16) = s0;
        *reinterpret_cast<u32x4*>(dA + W6_PLANE + c * 16) = s1;
        *reinterpret_cast<u32x4*>(dA + 2 * W6_PLANE + c * 16) = s2;
      }
#pragma unroll
      for (int c = 0; c < 2 * JW; ++c) {
        float x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = (mrow + r < m_end) ? rg[SL][r][c] : 0.f;
        u32x4 s0, s1, s2;
        split8(x, s0, s1, s2);
        *reinterpret_cast<u32x4*>(dG + c * 16) = s0;
        *reinterpret_cast<u32x4*>(dG + GPL + c * 16) = s1;
        *reinterpret_cast<u32x4*>(dG + 2 * GPL + c * 16) = s2;
      }
    };
    // straight-line pipeline: no conditional loads (a branch around a load drains vmcnt at the join)
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
    // ================= consumer waves: LDS fragments -> MFMA; wave tile 64 x (64 JW) = 4 x (4 JW) MFMA tiles =================
    const int cw = wave - 4;
    const int wr = cw >> 1, wc = cw & 1;
    constexpr int NC = 4 * JW;
    f32x4 acc[4][NC];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) __attribute__((always_inline)) {
      const char* sa = lds + buf * STAGE + (lane >> 4) * W6_GS + (wr * 64 + (lane & 15)) * 16;
      const char* sg = lds + buf * STAGE + W6_OPER + (lane >> 4) * GG + (wc * 64 * JW + (lane & 15)) * 16;
      u32x4 af[3][4];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int t = 0; t < 4; ++t) af[pl][t] = *reinterpret_cast<const u32x4*>(sa + pl * W6_PLANE + t * 256);
      // G plane by plane (smallest terms first): G2 x A0 ; G1 x {A1, A0} ; G0 x {A2, A1, A0}
#pragma unroll
      for (int bp = 2; bp >= 0; --bp) {
        u32x4 gf[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) gf[c] = *reinterpret_cast<const u32x4*>(sg + bp * GPL + c * 256);
#pragma unroll
        for (int ap = 2 - bp; ap >= 0; --ap)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[t][c] = mfma_bf16(af[ap][t], gf[c], acc[t][c]);
      }
    };
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < nloop; ++ch) {
      compute(ch & 1);
      __syncthreads();
    }
    // 16x16 accumulator map: col = lane & 15 (j), row = (lane >> 4) * 4 + reg (k)
    float* out = p.part + (size_t)zidx * p.zpart + (size_t)split * p.Kd * p.Jd;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = tk * T_TILE + wr * 64 + t * 16 + (lane >> 4) * 4 + e;
          const int j = tj * JT + wc * 64 * JW + c * 16 + (lane & 15);
          out[(size_t)k * p.Jd + j] = acc[t][c][e];
        }
  }
}

template <bool CONV, int JW>
__global__ __launch_bounds__(512) void wgrad6_kernel(TnP p) { wgrad6_body<CONV, JW>(p, blockIdx.y, blockIdx.x, gridDim.x); }

// several independent contractions of the same tile geometry in ONE launch (the [B,d]-sized linears' weight gradients at the end
// of a backward pass: 768 reduction rows each, a launch of their own was 17 us of latency + a slab reduction): blockIdx.y
// selects the contraction, workgroups past a contraction's own count leave at once
struct TnList { TnP d[8]; int nblk[8]; };
template <int JW>
__global__ __launch_bounds__(512) void wgrad6_list_kernel(TnList L) {
  const int n = L.nblk[blockIdx.y];
  if ((int)blockIdx.x >= n) return;
  wgrad6_body<false, JW>(L.d[blockIdx.y], 0, blockIdx.x, n);
}

inline int wgrad6_jw(int Jd) { return (Jd % 256 == 0) ? 2 : 1; }

template <bool CONV, int JW>
inline hipError_t wgrad6_launch_t(const TnP& p, hipStream_t st) {
  auto kern = wgrad6_kernel<CONV, JW>;
  constexpr size_t lds = 2 * w6_stage_bytes<JW>();
  {
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);   // per (kernel, device)
    if (e != hipSuccess) return e;
  }
  const int grid = (p.Kd / T_TILE) * (p.Jd / (JW * T_TILE)) * p.nsplit;
  hipLaunchKernelGGL(kern, dim3(grid, p.nz > 1 ? p.nz : 1), dim3(512), lds, st, p);
  return hipGetLastError();
}

template <int JW>
inline hipError_t wgrad6_list_launch_t(const TnList& L, int count, hipStream_t st) {
  auto kern = wgrad6_list_kernel<JW>;
  constexpr size_t lds = 2 * w6_stage_bytes<JW>();
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  int gx = 1;
  for (int i = 0; i < count; ++i) gx = L.nblk[i] > gx ? L.nblk[i] : gx;
  hipLaunchKernelGGL(kern, dim3(gx, count), dim3(512), lds, st, L);
  return hipGetLastError();
}
inline hipError_t wgrad6_launch(const TnP& p, hipStream_t st) {
  if (wgrad6_jw(p.Jd) == 2) return p.conv_taps ? wgrad6_launch_t<true, 2>(p, st) : wgrad6_launch_t<false, 2>(p, st);
  return p.conv_taps ? wgrad6_launch_t<true, 1>(p, st) : wgrad6_launch_t<false, 1>(p, st);
}

// ---------------------------------------------------------------------------------------------
// Per-question interaction gradient  S_b = X_b^T dI1_b  on the split-bf16 path (see sb_wgrad_kernel in
// macx_gemm_tn.hip.h for what S_b is for):  dW1a += diag(y_b) S_b,  dW1b += S_b,  dy[b][k] = sum_j W1a[k][j] S_b[k][j].
// 4 waves, one per SIMD (the three 64-register accumulator sets + the W1a tile need the whole 512-entry register file),
// each wave both stages (8 rows x 2 columns of X and of dI1 per lane, exact bf16 split, in-register transpose) and
// multiplies its 64 x 64 share of the 128 x 128 tile; the pipeline runs straight through question boundaries.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sum16(float v) {     // sum over the 16 lanes that share lane >> 4
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void sb6_wgrad_kernel(SbP p) {
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
  const int mg = wave;                                  // this wave stages rows [8 mg, 8 mg + 8) of every 32-row stage

  f32x4 accS[4][4], accA[4][4], accB[4][4], w1a[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      accS[t][c] = accA[t][c] = accB[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = tk * T_TILE + wr * 64 + t * 16 + (lane >> 4) * 4 + e;
        const int j = tj * T_TILE + wc * 64 + c * 16 + (lane & 15);
        w1a[t][c][e] = p.W1a[(size_t)k * p.d + j];
      }
    }

  const int nchunk = (p.N + 31) >> 5;
  const int b_begin = group * p.qpg;
  const int b_end = min(p.B, b_begin + p.qpg);
  const int total = (b_end - b_begin) * nchunk;         // stages of this workgroup, over all its questions

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  f32x2_t ra[2][8], rg[2][8];
  // stage s = (question qi = s / nchunk, rows [32 ch, 32 ch + 32) of it); every row address is wave-uniform
  auto load = [&](auto slot_c, int s_raw) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
    const int s = min(s_raw, total - 1);
    const int qi = s / nchunk, ch = s - qi * nchunk;
    const size_t qoff = (size_t)(b_begin + qi) * p.N * p.d;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int n = min(ch * 32 + mg * 8 + r, p.N - 1);       // rows past N are zeroed when the stage is split
      ra[SL][r] = *reinterpret_cast<const f32x2_t*>(p.X + qoff + (size_t)n * p.d + tk * T_TILE + 2 * lane);
      rg[SL][r] = *reinterpret_cast<const f32x2_t*>(p.dI1 + qoff + (size_t)n * p.d + tj * T_TILE + 2 * lane);
    }
  };
  auto store = [&](auto slot_c, int s_raw) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
    const int s = min(s_raw, total - 1);
    const int ch = s % nchunk;
    const int nrow = ch * 32 + mg * 8;
    char* dst = lds + (s_raw & 1) * W6_STAGE + mg * W6_GS + (2 * lane) * 16;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = (nrow + r < p.N) ? (o ? rg[SL][r][c] : ra[SL][r][c]) : 0.f;
        u32x4 s0, s1, s2;
        split8(x, s0, s1, s2);
        char* d = dst + o * W6_OPER + c * 16;
        *reinterpret_cast<u32x4*>(d) = s0;
        *reinterpret_cast<u32x4*>(d + W6_PLANE) = s1;
        *reinterpret_cast<u32x4*>(d + 2 * W6_PLANE) = s2;
      }
  };
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* sa = lds + buf * W6_STAGE + (lane >> 4) * W6_GS + (wr * 64 + (lane & 15)) * 16;
    const char* sg = lds + buf * W6_STAGE + W6_OPER + (lane >> 4) * W6_GS + (wc * 64 + (lane & 15)) * 16;
    u32x4 gf[3][4];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int c = 0; c < 4; ++c) gf[pl][c] = *reinterpret_cast<const u32x4*>(sg + pl * W6_PLANE + c * 256);
#pragma unroll
    for (int ap = 2; ap >= 0; --ap) {
      u32x4 af[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) af[t] = *reinterpret_cast<const u32x4*>(sa + ap * W6_PLANE + t * 256);
#pragma unroll
      for (int bp = 2 - ap; bp >= 0; --bp)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) accS[t][c] = mfma_bf16(af[t], gf[bp][c], accS[t][c]);
    }
  };
  // question finished: fold S_b into the three outputs and clear it
  auto consume = [&](int b) __attribute__((always_inline)) {
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
          const float sv = accS[t][c][e];
          accB[t][c][e] += sv;
          accA[t][c][e] = fmaf(yk, sv, accA[t][c][e]);
          dyp = fmaf(w1a[t][c][e], sv, dyp);
          accS[t][c][e] = 0.f;
        }
        dyp = sum16(dyp);
        if ((lane & 15) == 0) p.dy_part[((size_t)(tj * 2 + wc) * p.B + b) * p.d + k] = dyp;
      }
  };

  if (total > 0) {
    load(S0{}, 0);
    load(S1{}, 1);
    store(S0{}, 0);
    __syncthreads();
    int qch = 0, b = b_begin;                           // stage index inside the current question, current question
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

inline hipError_t sb6_wgrad_launch(const SbP& p, hipStream_t st) {
  constexpr size_t lds = 2 * W6_STAGE;
  {
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(sb6_wgrad_kernel), lds);   // per (kernel, device)
    if (e != hipSuccess) return e;
  }
  const int nt = p.d / T_TILE;
  const int ngroup = (p.B + p.qpg - 1) / p.qpg;
  hipLaunchKernelGGL(sb6_wgrad_kernel, dim3(nt * nt * ngroup), dim3(256), lds, st, p);
  return hipGetLastError();
}

}  // namespace macx
