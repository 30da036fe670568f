// macx_wgrad_h2.hip.h -- the backward kernels that CONSUME H2 tensors along their rows:
//   read_att_bwd_h2_kernel   softmax backward + the elementwise chain down to dI2, I2 in / dI2 out as H2
//   wgrad_h2_kernel          weight-gradient contractions  C[k][j] = sum_m A[m][k] G[m][j]  over H2 operands
//   sb_h2_kernel             per-question interaction gradient S_b = X_b^T dI1_b (dW1a, dW1b, dy) over H2 operands
// The contractions reduce over ROWS, so a lane's MFMA fragment is 8 consecutive rows of one column while an H2 slot is 8
// consecutive columns of one row: the producer waves transpose in registers while staging (one v_perm_b32 per two
// elements) and bring every row to the tensor's common exponent with one v_pk_mul_f16 by an exact power of two (rows far
// below the largest lose low bits exactly as their share of the sum warrants).  Two fp16 planes, three MFMA terms.
// Tiling, slabs and the fixed-order slab reduction are those of macx_wgrad6.hip.h.
#pragma once
#include "macx_gemm_tn.hip.h"
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
// read-unit attention backward (the same arithmetic as read_att_bwd_kernel in macx_small.hip.h).
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
  const float* dl;       // [B][N] softmax backward already done (kb_att_dl_kernel): att / da / dbk_part are not touched then
  int no_out;            // 1: only the column sums dc / dwk_part / db2_part (dI2 comes from chain_bwd_kernel)
};

// One workgroup per (question, 128-column block), 16 waves: wave w owns slot column w (8 columns) and its lanes run along
// ROWS -- a wave instruction reads / writes 64 consecutive slots of one slot column = one contiguous KiB per plane (the
// first version ran lanes along columns on this slot-major data: 8-byte pieces of 32 different cache lines per
// instruction, 2.0 TB/s).  A row's exponent needs the maximum over the 16 slot columns: partial maxima go through a small
// LDS table, the values wait in registers (128 rows per pass).  The three column sums stay inside a wave (it owns its 8
// columns): lanes accumulate their rows, one wave reduction per value at the end, no cross-wave combine.
constexpr int RABH_THREADS = 1024;
constexpr int RABH_CHUNK = 128;           // rows per pass (values wait in registers: 16 per lane at the 128-register cap of 1024 threads)
__global__ __launch_bounds__(RABH_THREADS) void read_att_bwd_h2_kernel(ReadAttBwdH2P p) {
  __shared__ float s_dl[1024];
  __shared__ float s_red[RABH_THREADS / 64];
  __shared__ float s_mx[16][RABH_CHUNK];
  const int b = blockIdx.x, slab = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NWV = RABH_THREADS / 64;
  if (p.dl) {
    for (int n = tid; n < p.N; n += RABH_THREADS) s_dl[n] = p.dl[(size_t)b * p.N + n];
    __syncthreads();
  } else {
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
  }

  const int kg = slab * 16 + wave;                       // this wave's slot column
  const int col0 = slab * 128 + wave * 8;
  float cv[8], wv[8];
  {
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(p.c + (size_t)b * p.d + col0), c1 = *reinterpret_cast<const f32x4*>(p.c + (size_t)b * p.d + col0 + 4);
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.wk + col0), w1 = *reinterpret_cast<const f32x4*>(p.wk + col0 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { cv[e] = c0[e]; cv[4 + e] = c1[e]; wv[e] = w0[e]; wv[4 + e] = w1[e]; }
  }
  const size_t iRp = p.I2.Rp(), oRp = p.dI2.Rp();
  const char* i0 = p.I2.plane(0);
  const size_t ipb = p.I2.plane_bytes(), opb = p.dI2.plane_bytes();
  char* o0 = p.dI2.plane(0);
  const int icb = p.I2.cb(), ocb = p.dI2.cb();
  float a_dc[8], a_dw[8], a_db[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a_dc[e] = a_dw[e] = a_db[e] = 0.f;

  // (the activation is selected by ONE switch around the row loop, not per value: see the row pass of kb_gemm_h2_kernel)
  auto rows = [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    for (int n0 = 0; n0 < p.N; n0 += RABH_CHUNK) {
      float o[RABH_CHUNK / 64][8];
#pragma unroll
      for (int j = 0; j < RABH_CHUNK / 64; ++j) {
        const int lrow = lane + 64 * j, n = n0 + lrow;
        float m = 0.f;
        if (n < p.N) {
          const size_t row = (size_t)b * p.N + n;
          const char* src = i0 + ((size_t)kg * iRp + row) * 16;
          float i2[8];
          h2_join8(*reinterpret_cast<const u32x4*>(src), *reinterpret_cast<const u32x4*>(src + ipb),
                   h2_pow2(-(int)p.I2.exps()[row * icb + slab]), i2);
          const float dl = s_dl[n];
          const uint32_t bits = p.bytes ? (uint32_t)p.bytes[(size_t)kg * iRp + row] : 0xFFu;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float zv = i2[e] * cv[e];
            const float g = act_apply(ACT, zv);
            const float f = ((bits >> e) & 1u) ? p.inv_keep : 0.f;
            const float dz = (dl * wv[e]) * f * act_grad_from_out(ACT, g);
            const float ov = dz * cv[e];                      // dI2 = dZ * c
            o[j][e] = ov;
            a_dw[e] = fmaf(dl, g * f, a_dw[e]);               // dw_k += dl * dropped(G)
            a_dc[e] = fmaf(dz, i2[e], a_dc[e]);               // dc += dZ * I2
            a_db[e] += ov;
            m = fmaxf(m, fabsf(ov));
          }
        }
        s_mx[wave][lrow] = m;
      }
      if (p.no_out) continue;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RABH_CHUNK / 64; ++j) {
        const int lrow = lane + 64 * j, n = n0 + lrow;
        if (n < p.N) {
          float m = s_mx[0][lrow];
#pragma unroll
          for (int k = 1; k < 16; ++k) m = fmaxf(m, s_mx[k][lrow]);
          const int ex = h2_exponent(m);
          const float sc = h2_pow2(ex);
          float xs[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) xs[e] = o[j][e] * sc;
          u32x4 hi, lo;
          h2_split8(xs, hi, lo);
          const size_t row = (size_t)b * p.N + n;
          char* dst = o0 + ((size_t)kg * oRp + row) * 16;
          *reinterpret_cast<u32x4*>(dst) = hi;
          *reinterpret_cast<u32x4*>(dst + opb) = lo;
          if (wave == 0) p.dI2.exps()[row * ocb + slab] = (int8_t)ex;
        }
      }
      __syncthreads();                    // s_mx is reused by the next pass
    }
  };
  switch (p.act) {
    case ACT_TANH: rows(std::integral_constant<int, ACT_TANH>{}); break;
    case ACT_SIGMOID: rows(std::integral_constant<int, ACT_SIGMOID>{}); break;
    case ACT_ELU: rows(std::integral_constant<int, ACT_ELU>{}); break;
    case ACT_RELU: rows(std::integral_constant<int, ACT_RELU>{}); break;
    default: rows(std::integral_constant<int, ACT_NON>{}); break;
  }
  // column sums: this wave's 8 columns, summed over its lanes' rows
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a_dc[e] = wave_sum(a_dc[e]);
    a_dw[e] = wave_sum(a_dw[e]);
    a_db[e] = wave_sum(a_db[e]);
  }
  if (lane == 0) {
    float* dc = p.dc + (size_t)b * p.d + col0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dc[e] += a_dc[e];
      p.dwk_part[(size_t)b * p.d + col0 + e] = a_dw[e];
      p.db2_part[(size_t)b * p.d + col0 + e] = a_db[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weight-gradient contraction over H2 operands.  Reduction row m lives in tensor m / R of a sequence of H2 tensors
// `*_stride` bytes apart (the per-step activations kept by the forward / backward pass); A may instead be ONE tensor reused
// by every step (a_mod = its rows: the undropped knowledge base).
//
// The contraction reduces over ROWS while an H2 slot holds 8 consecutive COLUMNS of one row, so the operands need a
// transpose on the way to the MFMA fragments.  It is done by the LDS: slots travel exactly as they lie in memory, by
// LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land lane-linear = one [2 row halves][16 rows][16 columns] column tile
// of the image below), and the waves read their fragments with ds_read_b64_tr_b16, the gfx950 transpose read: within 16
// lanes, lane t passes the address of chunk (row t/4, columns 4(t%4)..+3) and receives column t of the four rows -- four
// consecutive reduction rows of one output row/column, i.e. half an MFMA fragment.  No registers, no vector ALU and no
// strided global access in the staging.
//   Rows are brought to the tensor family's common exponents on the way from LDS to the MFMA: ONE combined factor per
//   (row, 128-column block of A, 128-column block of G), 2^(EA - ea_row + EG - eg_row) as fp16 (0 past the last row), written
//   by wgrad_h2_factors_kernel before the contraction and fetched per stage by a 4-byte LDS-DMA; it multiplies the G fragments
//   only (two v_pk_mul_f16 per transpose read).  Rows far below the largest lose low bits exactly as their share of the sum
//   warrants.
//   workgroup: (128 KW) x (128 JW) output tile, all 8 waves multiply (2 x 4, wave tile 64 KW x 32 JW); 32 reduction rows per
//   stage.  The kernel is bound by the L2 -> LDS traffic of its operands (every A column is read by Jd / (128 JW) workgroups,
//   every G column by Kd / (128 KW)), so the tile is as large as the accumulators allow: 256 x 256 (128 registers per lane,
//   64 KB stages, ring of 2) when both dimensions allow, else ring of 3 (48 KB stages) or 4; the wait for a stage is an
//   explicit s_waitcnt vmcnt(n) that leaves the younger stages' DMA in flight (see dma16b).
// Determinism as in macx_gemm_tn.hip.h: a workgroup owns one (split, tile) slab, slabs are summed in a fixed order.
// ---------------------------------------------------------------------------------------------------------------
struct TnH2P {
  int M;                 // reduction rows over all tensors
  int Kd, Jd;            // output dims (multiples of 128)
  int nsplit, rows_per_split;                    // rows_per_split % 32 == 0: a stage never straddles two splits
  int R;                 // rows per H2 tensor
  const char* A; size_t a_stride; int a_mod;     // a_mod > 0: A row of reduction row m is m % a_mod of tensor 0
  const char* G; size_t g_stride;
  const int* ecomA;      // [ecom_nb][8] partial minima of A's row exponents per 128-column block (h2_emin_list_kernel)
  const int* ecomG;      // [ecom_nb][8] ... of G's
  int ecom_nb;
  uint16_t* ftab;        // [Kd/128][Jd/128][Mpad] combined row factors, Mpad = wgrad_h2_mpad(M)
  float* part;           // [nsplit][Kd][Jd]
  int dbg;               // measurement knobs (macx_opts.tune[MACX_TUNE_PHASE_MASK]): 1024 skip fragments + MFMAs, 2048 skip the in-loop DMA
  // (Round 5 also had a DUAL form here -- two A families against one G, for dW1a / dW1b as one contraction over X * y and X kept by the
  // forward chain kernel: twice the per-question kernel's matrix work, 392 against 345 us, profiles/r05_dual_contraction_ab.txt.  Removed.)
};
__host__ __device__ inline size_t wgrad_h2_mpad(size_t M) { return (M + 63) & ~(size_t)31; }       // a stage starting below M stays inside

typedef short s16x4 __attribute__((ext_vector_type(4)));
// half an MFMA fragment: 4 consecutive reduction rows of column (lane & 15), see the header comment
__device__ __forceinline__ u32x2 tr_read(const char* lds_addr) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_addr);
  return __builtin_bit_cast(u32x2, v);
}

// LDS image of one operand plane and stage: sub-tiles of [16 reduction rows][16 columns] fp16 = 512 contiguous bytes, laid
// out [column tile][row half].  One transpose read covers exactly one sub-tile with lane-linear 8-byte chunks -- the
// conflict-free pattern of the instruction -- and gives lane (i, g) column i of rows 4g..4g+3; the two row halves make the
// lane's 8-element fragment (reduction rows {4g+e} and {16+4g+e}: any assignment works as long as both operands share it).
constexpr int WH_APL = 8 * 2 * 512;         // one plane of a 128-column image
template <int KW, int JW> constexpr int wh_stage_bytes() { return 2 * (KW + JW) * WH_APL + 256; }   // + the stage's factors
template <int KW, int JW> constexpr int wh_ring() { return KW + JW == 4 ? 2 : (KW + JW == 3 ? 3 : 4); }

// one thread per reduction row: the fp16 factor of every (A block, G block) pair
__global__ __launch_bounds__(256) void wgrad_h2_factors_kernel(TnH2P p0, TnH2P p1) {
  const TnH2P& p = blockIdx.y ? p1 : p0;             // grid.y = 2: the tables of two contractions in one launch
  __shared__ int sE[2][8];
  if (threadIdx.x < 16) {
    const int w = threadIdx.x >> 3, k = threadIdx.x & 7;
    sE[w][k] = h2_emin_final(w ? p.ecomG : p.ecomA, p.ecom_nb, k);
  }
  __syncthreads();
  const size_t mpad = wgrad_h2_mpad((size_t)p.M);
  const size_t m = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= mpad) return;
  const int acb = p.Kd >> 7, gcb = p.Jd >> 7;
  const H2View a0{const_cast<char*>(p.A), p.R, p.Kd}, g0{const_cast<char*>(p.G), p.R, p.Jd};
  if (m >= (size_t)p.M) {
    for (int i = 0; i < acb * gcb; ++i) p.ftab[(size_t)i * mpad + m] = 0;
    return;
  }
  const int ti = (int)(m / p.R), rr = (int)(m - (size_t)ti * p.R);
  const int ar = p.a_mod ? (int)(m % p.a_mod) : rr;
  const int8_t* ea = reinterpret_cast<const int8_t*>(p.A + (p.a_mod ? 0 : (size_t)ti * p.a_stride) + 2 * a0.plane_bytes()) + (size_t)ar * acb;
  const int8_t* eg = reinterpret_cast<const int8_t*>(p.G + (size_t)ti * p.g_stride + 2 * g0.plane_bytes()) + (size_t)rr * gcb;
  for (int k = 0; k < acb; ++k) {
    const int da = sE[0][k] - (int)ea[k];
    for (int j = 0; j < gcb; ++j)
      p.ftab[((size_t)k * gcb + j) * mpad + m] = (uint16_t)(pk_pow2_f16(da + sE[1][j] - (int)eg[j]) & 0xFFFFu);
  }
}

__device__ __forceinline__ void dma4b(const char* g, char* lds_wave_base) {     // 64 lanes x 4 B, lane-linear (see dma16b)
  const uint32_t l = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(l) : "memory", "m0");
}

// PIPE = 1 (KW = JW = 2 only, round 5): the same two-buffer ring, but a buffer's halves are re-requested as soon as the waves are
// done with them instead of one iteration later.  compute() reads ALL of a stage's G fragments into registers first and the A planes
// one after the other, so inside iteration s the G half of buffer s % 2 is free after the first few LDS reads and its A half after
// a third of the products: the DMA of stage s + 2's G half is issued behind a barrier right there, the A half behind a second one.
// Two stages (up to 128 KB per CU) are in flight instead of one, and a request has 1.7 - 2.0 iterations to land instead of one --
// with RING = 2 the old loop had `vmcnt(0)` at the end of every iteration, i.e. every stage's full latency + transfer on the
// critical path whenever it exceeded one iteration of products (round 4: 248 us against 134 compute-only / 146 DMA-only).
// Same products in the same order: results are bit-identical to PIPE = 0.
// 3 (default): as 2 with HALF the reduction splits per contraction (the caller's choice, macx_api.hip), so that both contractions'
// workgroups are resident at once; 2: PIPE = 1 and the two all-steps contractions of the backward pass as ONE launch (grid.y = 2);
// 1: PIPE = 1, one launch each; 0: round 4's loop.  macx_opts.tune[MACX_TUNE_WGRAD_PIPE]: the A/B hook
inline int wgrad_pipe_mode() { const int v = tune_get(MACX_TUNE_WGRAD_PIPE, 3); return (v >= 0 && v <= 3) ? v : 3; }

template <int KW, int JW, int PIPE = 0>
__global__ __launch_bounds__(512) void wgrad_h2_kernel(TnH2P p0, TnH2P p1) {
  // grid.y = 2: two contractions of the same shape in ONE launch (dW2 and dWx of the cell's backward pass): the second one's
  // workgroups take the CUs the first one's free -- one ramp and one tail instead of two, one launch boundary less
  const TnH2P& p = blockIdx.y ? p1 : p0;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  constexpr int KT = KW * T_TILE;
  constexpr int JT = JW * T_TILE;
  constexpr int APL = KW * WH_APL;
  constexpr int GPL = JW * WH_APL;
  constexpr int DATA = 2 * APL + 2 * GPL;
  constexpr int STAGE = wh_stage_bytes<KW, JW>();
  constexpr int RING = wh_ring<KW, JW>();
  constexpr int NI = 2 * (KW + JW);           // data DMA instructions per wave and stage (16 (KW + JW) in all)
  constexpr int NR = 4 * KW;                  // 16-row tiles of a wave
  constexpr int NC = 2 * JW;                  // 16-column tiles of a wave

  const int ntj = p.Jd / JT;
  const int ntk = p.Kd / KT;
  const int ntile = ntj * ntk;
  const int nblk = gridDim.x;
  int v = blockIdx.x;
  if ((nblk & 7) == 0) v = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int split = v / ntile;
  const int tile = v % ntile;
  const int tk = tile / ntj, tj = tile % ntj;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;    // 64 KW rows (k) x 32 JW columns (j) of the tile
  const int aq = KW == 2 ? wr : 0;            // the 128-column block of A this wave's rows lie in
  const int gq = JW == 2 ? (wc >> 1) : 0;     // the 128-column block of G this wave's columns lie in
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int nchunk = (m_end - m_begin + 31) >> 5;

  const H2View a0{const_cast<char*>(p.A), p.R, p.Kd}, g0{const_cast<char*>(p.G), p.R, p.Jd};
  const size_t Rp = a0.Rp();
  const size_t apb = a0.plane_bytes(), gpb = g0.plane_bytes();
  const int gcb = g0.cb();
  const size_t mpad = wgrad_h2_mpad((size_t)p.M);

  f32x4 acc[NR][NC];
#pragma unroll
  for (int t = 0; t < NR; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging: data instruction u of a stage (u = NI wave .. NI wave + NI - 1) fills one column tile: u < 16 KW -> plane
  //      u / (8 KW), tile u % (8 KW) of A; else plane, tile of G likewise.  Lane q of it copies slot (row half q / 32,
  //      row (q % 32) / 2, 8-column group 2 tile + q % 2), the lane-linear order of the [half][row][16 columns] tile.
  //      Wave 0 also fetches the stage's factors: lane l < 16 KW JW -> fp16 pair l % 16 of table (l / 16) = (A block, G block).
  const int sl_m = (lane >> 5) * 16 + ((lane & 31) >> 1), sl_kg = lane & 1;
  const int ft_t = lane < 16 * KW * JW ? (lane >> 4) : 0;
  const uint16_t* ft_src = p.ftab + ((size_t)(tk * KW + ft_t / JW) * gcb + tj * JW + ft_t % JW) * mpad + (lane < 16 * KW * JW ? (lane & 15) * 2 : 0);
  auto issue = [&](int s_raw) __attribute__((always_inline)) {
    const int ch = min(s_raw, nchunk - 1);
    const int m0 = m_begin + ch * 32;
    const int mc = min(m0 + sl_m, p.M - 1);               // rows past the end re-read the last row (finite data, factor 0)
    const int ti = mc / p.R, rr = mc - ti * p.R;
    const int ar = p.a_mod ? (mc % p.a_mod) : rr;
    const char* ab = p.A + (p.a_mod ? 0 : (size_t)ti * p.a_stride);
    const char* gb = p.G + (size_t)ti * p.g_stride;
    char* st = lds + (s_raw % RING) * STAGE;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int u = wave * NI + j;
      if (u < 16 * KW) {
        const int pl = u / (8 * KW), ct = u - pl * 8 * KW;
        dma16b(ab + pl * apb + ((size_t)(tk * 16 * KW + 2 * ct + sl_kg) * Rp + ar) * 16, st + pl * APL + ct * 1024);
      } else {
        const int ug = u - 16 * KW;
        const int pl = ug / (8 * JW), ct = ug - pl * 8 * JW;
        dma16b(gb + pl * gpb + ((size_t)(tj * 16 * JW + 2 * ct + sl_kg) * Rp + rr) * 16, st + 2 * APL + pl * GPL + ct * 1024);
      }
    }
    if (wave == 0) dma4b(reinterpret_cast<const char*>(ft_src + m0), st + DATA);
  };
  // PIPE: one half of a stage -- part 0 = the G planes + the factors (waves 4 - 7 and wave 0), part 1 = the A planes (waves 0 - 3)
  auto issue_part = [&](int s_raw, int part) __attribute__((always_inline)) {
    static_assert(!PIPE || (KW == 2 && JW == 2), "the split issue assumes waves 0-3 stage A and waves 4-7 stage G");
    const int ch = min(s_raw, nchunk - 1);
    const int m0 = m_begin + ch * 32;
    const int mc = min(m0 + sl_m, p.M - 1);
    const int ti = mc / p.R, rr = mc - ti * p.R;
    char* st = lds + (s_raw % RING) * STAGE;
    if (part == 1) {
      if (wave < 4) {
        const int ar = p.a_mod ? (mc % p.a_mod) : rr;
        const char* ab = p.A + (p.a_mod ? 0 : (size_t)ti * p.a_stride);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int u = wave * NI + j;
          const int pl = u / (8 * KW), ct = u - pl * 8 * KW;
          dma16b(ab + pl * apb + ((size_t)(tk * 16 * KW + 2 * ct + sl_kg) * Rp + ar) * 16, st + pl * APL + ct * 1024);
        }
      }
    } else {
      if (wave >= 4) {
        const char* gb = p.G + (size_t)ti * p.g_stride;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int ug = wave * NI + j - 16 * KW;
          const int pl = ug / (8 * JW), ct = ug - pl * 8 * JW;
          dma16b(gb + pl * gpb + ((size_t)(tj * 16 * JW + 2 * ct + sl_kg) * Rp + rr) * 16, st + 2 * APL + pl * GPL + ct * 1024);
        }
      }
      if (wave == 0) dma4b(reinterpret_cast<const char*>(ft_src + m0), st + DATA);
    }
  };
  auto frag = [&](const char* tile) __attribute__((always_inline)) {       // the two row halves of one column tile
    const u32x2 lo = tr_read(tile + lane * 8), hi = tr_read(tile + 512 + lane * 8);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  // smallest terms first: A_lo x G_hi ; A_hi x {G_lo, G_hi}
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* sa = lds + buf * STAGE + (wr * NR) * 1024;
    const char* sg = lds + buf * STAGE + 2 * APL + (wc * NC) * 1024;
    // this lane's reduction rows: 4 g + {0..3} of each row half -> two packed factor pairs per half
    const uint16_t* ft = reinterpret_cast<const uint16_t*>(lds + buf * STAGE + DATA) + (aq * JW + gq) * 32 + (lane >> 4) * 4;
    const u32x2 f0 = *reinterpret_cast<const u32x2*>(ft), f1 = *reinterpret_cast<const u32x2*>(ft + 16);
    u32x4 gf[2][NC];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        u32x4 w = frag(sg + pl * GPL + c * 1024);
        w[0] = pk_mul_f16(w[0], f0[0]); w[1] = pk_mul_f16(w[1], f0[1]);
        w[2] = pk_mul_f16(w[2], f1[0]); w[3] = pk_mul_f16(w[3], f1[1]);
        gf[pl][c] = w;
      }
#pragma unroll
    for (int ap = 1; ap >= 0; --ap) {
      u32x4 af[NR];
#pragma unroll
      for (int t = 0; t < NR; ++t) af[t] = frag(sa + ap * APL + t * 1024);
#pragma unroll
      for (int bp = 1 - ap; bp >= 0; --bp)
#pragma unroll
        for (int t = 0; t < NR; ++t)
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[t][c] = mfma_f16(af[t], gf[bp][c], acc[t][c]);
    }
  };

  if constexpr (PIPE) {
    if (nchunk > 0) {
      const int my_n = NI + (wave == 0 ? 1 : 0);            // this wave's DMA instructions of ONE stage (its half + wave 0's factors)
      issue(0);
      issue(1);
      wait_vmcnt_n(my_n);                                   // stage 0 has landed
      __syncthreads();
#pragma unroll 1
      for (int s = 0; s < nchunk; ++s) {
        const int buf = s & 1;
        const char* sa = lds + buf * STAGE + (wr * NR) * 1024;
        const char* sg = lds + buf * STAGE + 2 * APL + (wc * NC) * 1024;
        const uint16_t* ft = reinterpret_cast<const uint16_t*>(lds + buf * STAGE + DATA) + (aq * JW + gq) * 32 + (lane >> 4) * 4;
        const u32x2 f0 = *reinterpret_cast<const u32x2*>(ft), f1 = *reinterpret_cast<const u32x2*>(ft + 16);
        u32x4 gf[2][NC];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            u32x4 w = frag(sg + pl * GPL + c * 1024);
            w[0] = pk_mul_f16(w[0], f0[0]); w[1] = pk_mul_f16(w[1], f0[1]);
            w[2] = pk_mul_f16(w[2], f1[0]); w[3] = pk_mul_f16(w[3], f1[1]);
            gf[pl][c] = w;
          }
        __syncthreads();                                    // every wave holds its G fragments and factors: the G half is free
        if (!(p.dbg & 2048)) issue_part(s + 2, 0);
        u32x4 af[NR];
#pragma unroll
        for (int t = 0; t < NR; ++t) af[t] = frag(sa + APL + t * 1024);          // A lo
        if (!(p.dbg & 1024)) {
#pragma unroll
          for (int t = 0; t < NR; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[t][c] = mfma_f16(af[t], gf[0][c], acc[t][c]);   // A_lo x G_hi
        }
#pragma unroll
        for (int t = 0; t < NR; ++t) af[t] = frag(sa + t * 1024);                // A hi
        __syncthreads();                                    // every wave holds its last A fragments: the A half is free
        if (!(p.dbg & 2048)) issue_part(s + 2, 1);
        if (!(p.dbg & 1024)) {
#pragma unroll
          for (int bp = 1; bp >= 0; --bp)
#pragma unroll
            for (int t = 0; t < NR; ++t)
#pragma unroll
              for (int c = 0; c < NC; ++c) acc[t][c] = mfma_f16(af[t], gf[bp][c], acc[t][c]);   // A_hi x G_lo, A_hi x G_hi
        }
        if (p.dbg & 2048) wait_vmcnt<0>(); else wait_vmcnt_n(my_n);              // stage s + 1 has landed (s + 2 may fly)
        __syncthreads();
      }
      wait_vmcnt<0>();                                      // the speculative stages past the end
    }
  } else
  if (nchunk > 0) {
    const int my_n = (NI + (wave == 0 ? 1 : 0)) * (RING - 2);     // this wave's DMA instructions of the stages that may stay in flight
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) issue(s);
    wait_vmcnt_n(my_n);                                   // stage 0 has landed
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < nchunk; ++s) {
      if (!(p.dbg & 2048)) issue(s + RING - 1);           // ring slot (s - 1) % RING was last read in iteration s - 1
      if (!(p.dbg & 1024)) compute(s % RING);
      wait_vmcnt_n(my_n);                                 // stage s + 1 has landed
      __syncthreads();
    }
    wait_vmcnt<0>();                                      // the speculative stages past the end
  }

  __syncthreads();                                        // every wave's DMA has landed: the ring is about to be reused

  // ---- the slab leaves through LDS, 32 rows of the wave's share at a time (row-major in the wave's own corner of the idle
  //      ring, read back as float4): a store instruction writes whole 128 / 256-byte row segments instead of 64-byte pieces
  const float sc = h2_unscale(h2_emin_final(p.ecomA, p.ecom_nb, tk * KW + aq), h2_emin_final(p.ecomG, p.ecom_nb, tj * JW + gq));
  float* out = p.part + (size_t)split * p.Kd * p.Jd + (size_t)(tk * KT + wr * 16 * NR) * p.Jd + tj * JT + wc * 32 * JW;
  constexpr int CW = 32 * JW, LDW = CW + 4;
  constexpr int LPR = CW / 4, RPI = 64 / LPR;             // lanes per row, rows per store instruction
  float* tw = reinterpret_cast<float*>(lds) + wave * (32 * LDW);
#pragma unroll
  for (int tp = 0; tp < NR; tp += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) tw[(u * 16 + (lane >> 4) * 4 + e) * LDW + c * 16 + (lane & 15)] = acc[tp + u][c][e] * sc;
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
      const int r = i * RPI + lane / LPR, c4 = (lane % LPR) * 4;
      f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)(tp * 16 + r) * p.Jd + c4);
      f32x4 val = *reinterpret_cast<const f32x4*>(tw + r * LDW + c4);
      *dst = val;
    }
  }
}

inline int wgrad_h2_jw(int Jd) { return (Jd % 256 == 0) ? 2 : 1; }
inline int wgrad_h2_kw(int Kd) { return (Kd % 256 == 0) ? 2 : 1; }
inline int wgrad_h2_tiles(int Kd, int Jd) { return (Kd / (wgrad_h2_kw(Kd) * T_TILE)) * (Jd / (wgrad_h2_jw(Jd) * T_TILE)); }

template <int KW, int JW, int PIPE = 0>
inline hipError_t wgrad_h2_launch_t(const TnH2P& p, hipStream_t st) {
  auto kern = wgrad_h2_kernel<KW, JW, PIPE>;
  constexpr size_t lds = (size_t)wh_ring<KW, JW>() * wh_stage_bytes<KW, JW>();
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  const int grid = wgrad_h2_tiles(p.Kd, p.Jd) * p.nsplit;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, p);
  return hipGetLastError();
}
// two contractions with the same M, Kd, Jd and split (own operands, tables and slabs) as one launch each of the factor and the
// contraction kernel; wgrad_pipe_mode() < 2 or another tile shape: two launches each
inline hipError_t wgrad_h2_launch(const TnH2P& p, hipStream_t st);
inline hipError_t wgrad_h2_launch_pair(const TnH2P& a, const TnH2P& b, hipStream_t st) {
  const bool same = a.M == b.M && a.Kd == b.Kd && a.Jd == b.Jd && a.nsplit == b.nsplit && a.rows_per_split == b.rows_per_split && a.R == b.R;
  if (!(same && wgrad_pipe_mode() >= 2 && wgrad_h2_kw(a.Kd) == 2 && wgrad_h2_jw(a.Jd) == 2 && a.ftab != b.ftab && a.part != b.part)) {
    hipError_t e = wgrad_h2_launch(a, st);
    return e != hipSuccess ? e : wgrad_h2_launch(b, st);
  }
  if (a.rows_per_split % 32 != 0 || !a.ftab || !b.ftab) return hipErrorInvalidValue;
  const size_t mpad = wgrad_h2_mpad((size_t)a.M);
  hipLaunchKernelGGL(wgrad_h2_factors_kernel, dim3((unsigned)((mpad + 255) / 256), 2), dim3(256), 0, st, a, b);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  auto kern = wgrad_h2_kernel<2, 2, 1>;
  constexpr size_t lds = (size_t)wh_ring<2, 2>() * wh_stage_bytes<2, 2>();
  e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(wgrad_h2_tiles(a.Kd, a.Jd) * a.nsplit, 2), dim3(512), lds, st, a, b);
  return hipGetLastError();
}
inline hipError_t wgrad_h2_launch(const TnH2P& p, hipStream_t st) {
  if (p.rows_per_split % 32 != 0 || !p.ftab) return hipErrorInvalidValue;
  const size_t mpad = wgrad_h2_mpad((size_t)p.M);
  hipLaunchKernelGGL(wgrad_h2_factors_kernel, dim3((unsigned)((mpad + 255) / 256)), dim3(256), 0, st, p, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int kw = wgrad_h2_kw(p.Kd), jw = wgrad_h2_jw(p.Jd);
  if (kw == 2 && jw == 2) return wgrad_pipe_mode() ? wgrad_h2_launch_t<2, 2, 1>(p, st) : wgrad_h2_launch_t<2, 2, 0>(p, st);
  if (kw == 2) return wgrad_h2_launch_t<2, 1>(p, st);
  return jw == 2 ? wgrad_h2_launch_t<1, 2>(p, st) : wgrad_h2_launch_t<1, 1>(p, st);
}

// ---------------------------------------------------------------------------------------------------------------
// Per-question interaction gradient  S_b = X_b^T dI1_b  over H2 operands (see sb_wgrad_kernel in macx_gemm_tn.hip.h for what
// S_b is for):  dW1a += diag(y_b) S_b,  dW1b += S_b,  dy[b][k] = sum_j W1a[k][j] S_b[k][j].
//
// All 8 waves multiply (one 64 x 32 eighth of the 128 x 128 tile each, three accumulator sets = 96 registers, two waves per
// SIMD).  Staging costs no registers and no vector ALU: the H2 slots go from HBM/L2 straight into the LDS images by LDS-DMA
// (global_load_lds_dwordx4: 64 lanes x 16 B land lane-linear, which is exactly one [2 row halves][16 rows][16 columns]
// column tile of the transpose-read image) through a four-stage ring, and the wait for a stage is an explicit s_waitcnt vmcnt(8)
// -- this wave's eight DMA instructions of the next two stages stay in flight.  The rows of a question are brought to the
// question's common exponents on the way from LDS to the MFMA: one combined factor per row, 2^(EX - ex_row + EG - eg_row)
// (0 for rows past the question's end), precomputed for all of the workgroup's questions into an LDS table when the kernel
// starts and applied to the dI1 fragments only (two v_pk_mul_f16 per transpose read).  A stage never crosses a question
// boundary, so the fold of S_b happens between stages.
// ---------------------------------------------------------------------------------------------------------------
struct SbH2P {
  int B, N, d;
  int qpg;
  H2View X;                // [B*N][d]
  H2View dI1;              // [B*N][d]
  // (the common exponents of a question's rows -- the minimum of their exponent bytes, per operand and 128-column block -- are
  // found by the kernel itself while it builds its factor table)
  const float* y;          // [B][d]
  const float* W1a;        // [d][d] row-major (k, j)
  float* dW1a_part;        // [ngroup][d][d]
  float* dW1b_part;
  float* dy_part;          // [4*d/128][B][d]; null: dy is not computed here (the chain kernel's stage B2 has it)
  // all steps in one launch (the caller gets dy elsewhere, so nothing in the recurrence waits for S_b): the operands of step i
  // lie x_step / g_step BYTES and y_step FLOATS behind step 0's; the two accumulators run through every step and the slabs
  // are written once
  int nsteps;
  size_t x_step, g_step, y_step;
  int dbg;                 // measurement knobs (macx_opts.tune[MACX_TUNE_PHASE_MASK]): 512 skip the per-question fold, 1024 skip fragments + MFMAs,
                           // 2048 skip the DMA issue
};

constexpr int SBH_STAGE = 4 * WH_APL;      // X hi, X lo, dI1 hi, dI1 lo images of 32 rows x 128 columns
constexpr int SBH_CW = 8;                  // multiplying waves; dy_part holds SBH_CW / 2 partials per 128 columns
constexpr int SBH_MAXROWS = 4096;          // rows of one workgroup's questions covered by the factor table (8 KB)
constexpr int SBH_RING = 4;                // LDS stages: three stages of DMA in flight behind the one being multiplied
constexpr int SBH_MAXQ = 32;               // questions per workgroup (their y_b[128] shares sit in LDS: 16 KB)

template <bool DY>
__global__ __launch_bounds__(512) void sb_h2_kernel(SbH2P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  uint16_t* ftab = reinterpret_cast<uint16_t*>(lds + SBH_RING * SBH_STAGE);      // [questions][nchunk * 32] combined row factors (fp16)
  float* ytab = reinterpret_cast<float*>(lds + SBH_RING * SBH_STAGE + SBH_MAXROWS * 2);   // [questions][128] y_b of this tile's k rows
  float* sctab = ytab + SBH_MAXQ * T_TILE;                // [questions] 2^-(EX + EG): the unit of a question's S_b
  int* qmn = reinterpret_cast<int*>(sctab + SBH_MAXQ);    // [2][questions] minimum exponent of the question's X / dI1 rows (this tile's blocks)

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
  const int wr = wave >> 2, wc = wave & 3;                // 64 rows (k) x 32 columns (j) of the tile

  const int nchunk = (p.N + 31) >> 5;
  const int b_begin = group * p.qpg;
  const int b_end = min(p.B, b_begin + p.qpg);
  const int nq = b_end - b_begin;
  const int total = nq * nchunk;                          // stages of this workgroup, over all its questions
  const size_t Rp = p.X.Rp();
  const size_t xpb = p.X.plane_bytes(), gpb = p.dI1.plane_bytes();
  const int xcb = p.X.cb(), gcb = p.dI1.cb();
  // the current step's operands
  H2View vX = p.X, vG = p.dI1;
  const float* yS = p.y;
  const int rows_q = nchunk * 32;                         // table rows per question (rows past N hold factor 0)

  // ---- staging: instruction u of a stage (u = 4 wave .. 4 wave + 3) fills column tile u & 7 of image u >> 3
  //      (0: X hi, 1: X lo, 2: dI1 hi, 3: dI1 lo); lane q of it copies slot (row half q / 32, row (q % 32) / 2, 8-column group
  //      2 (u & 7) + q % 2), the lane-linear order of the [half][row][16 columns] tile
  const int sl_half = lane >> 5, sl_row = (lane & 31) >> 1, sl_kg = lane & 1;
  const int sl_m = sl_half * 16 + sl_row;                 // stage row of this lane's slot
  auto issue = [&](int s_raw) __attribute__((always_inline)) {
    const int s = min(s_raw, total - 1);
    const int qi = s / nchunk, ch = s - qi * nchunk;
    const int n = min(ch * 32 + sl_m, p.N - 1);           // rows past the end re-read the last row (finite data, factor 0)
    const size_t row = (size_t)(b_begin + qi) * p.N + n;
    char* st = lds + (s_raw % SBH_RING) * SBH_STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int u = wave * 4 + j;
      const int im = u >> 3, ct = u & 7;
      const char* src = (im < 2 ? vX.base + (im & 1) * xpb + ((size_t)(tk * 16 + 2 * ct + sl_kg) * Rp + row) * 16
                                : vG.base + (im & 1) * gpb + ((size_t)(tj * 16 + 2 * ct + sl_kg) * Rp + row) * 16);
      dma16b(src, st + im * WH_APL + ct * 1024);
    }
  };
  auto tables = [&]() __attribute__((always_inline)) {
  // ---- the questions' common exponents: integer minima through LDS (order-independent), one LDS atomic per wave, question and
  //      operand -- then the factor table
  if (tid < 2 * SBH_MAXQ) qmn[tid] = 127;
  __syncthreads();
  for (int i0 = 0; i0 < nq * rows_q; i0 += 512) {
    const int i = i0 + tid;
    const int qi = i / rows_q, n = i - qi * rows_q;
    int ex = 127, eg = 127;
    if (i < nq * rows_q && n < p.N) {
      const size_t row = (size_t)(b_begin + qi) * p.N + n;
      ex = (int)vX.exps()[row * xcb + tk];
      eg = (int)vG.exps()[row * gcb + tj];
    }
    const int q_lo = min(i0 + (tid & ~63), nq * rows_q - 1) / rows_q, q_hi = min(i0 + (tid | 63), nq * rows_q - 1) / rows_q;   // questions of this wave's rows
    for (int q = q_lo; q <= q_hi; ++q) {
      int mx = qi == q ? ex : 127, mg = qi == q ? eg : 127;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { mx = min(mx, __shfl_xor(mx, o, 64)); mg = min(mg, __shfl_xor(mg, o, 64)); }
      if (lane == 0) { atomicMin(qmn + q, mx); atomicMin(qmn + SBH_MAXQ + q, mg); }
    }
  }
  __syncthreads();
  for (int i = tid; i < nq * rows_q; i += 512) {
    const int qi = i / rows_q, n = i - qi * rows_q;
    const int b = b_begin + qi;
    uint16_t f = 0;
    if (n < p.N) {
      const size_t row = (size_t)b * p.N + n;
      const int k = (min(qmn[qi], 126) - (int)vX.exps()[row * xcb + tk]) + (min(qmn[SBH_MAXQ + qi], 126) - (int)vG.exps()[row * gcb + tj]);
      f = (uint16_t)(pk_pow2_f16(k) & 0xFFFFu);
    }
    ftab[i] = f;
  }
  for (int i = tid; i < nq * T_TILE; i += 512)
    ytab[i] = yS[(size_t)(b_begin + (i >> 7)) * p.d + tk * T_TILE + (i & 127)];
  if (tid < nq) sctab[tid] = h2_unscale(min(qmn[tid], 126), min(qmn[SBH_MAXQ + tid], 126));
  };

  f32x4 accS[4][2], accA[4][2], accB[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) accS[t][c] = accA[t][c] = accB[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto frag = [&](const char* tile) __attribute__((always_inline)) {
    const u32x2 lo = tr_read(tile + lane * 8), hi = tr_read(tile + 512 + lane * 8);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  // smallest terms first: X_lo x dI1_hi ; X_hi x {dI1_lo, dI1_hi}
  auto compute = [&](int buf, int qi, int ch) __attribute__((always_inline)) {
    const char* sa = lds + buf * SBH_STAGE + (wr * 4) * 1024;
    const char* sg = lds + buf * SBH_STAGE + 2 * WH_APL + (wc * 2) * 1024;
    // this lane's reduction rows: 4 g + {0..3} of each row half -> two packed factor pairs per half
    const uint16_t* ft = ftab + qi * rows_q + ch * 32 + (lane >> 4) * 4;
    const u32x2 f0 = *reinterpret_cast<const u32x2*>(ft), f1 = *reinterpret_cast<const u32x2*>(ft + 16);
    u32x4 gf[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        u32x4 w = frag(sg + pl * WH_APL + c * 1024);
        w[0] = pk_mul_f16(w[0], f0[0]); w[1] = pk_mul_f16(w[1], f0[1]);
        w[2] = pk_mul_f16(w[2], f1[0]); w[3] = pk_mul_f16(w[3], f1[1]);
        gf[pl][c] = w;
      }
#pragma unroll
    for (int ap = 1; ap >= 0; --ap) {
      u32x4 af[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) af[t] = frag(sa + ap * WH_APL + t * 1024);
#pragma unroll
      for (int bp = 1 - ap; bp >= 0; --bp)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int c = 0; c < 2; ++c) accS[t][c] = mfma_f16(af[t], gf[bp][c], accS[t][c]);
    }
  };
  // this wave's 64 x 32 share of W1a stays in registers for the whole kernel (32 per lane; the staging needs none) and y_b
  // comes from the LDS table: the per-question fold has no global load to wait for, so it does not drain the DMA queue
  float w1[4][4][2];
  if (DY) {
    const float* wb = p.W1a + (size_t)(tk * T_TILE + wr * 64 + (lane >> 4) * 4) * p.d + tj * T_TILE + wc * 32 + (lane & 15);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c) w1[t][e][c] = wb[(size_t)(t * 16 + e) * p.d + c * 16];
    // a use here makes the compiler wait for these loads NOW; left to itself it waits at the first fold inside the loop,
    // with vmcnt(0), on every trip -- draining the DMA queue it knows nothing about
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(w1[t][e][c]));
  }
  // question finished: fold S_b (in units of its two common exponents) into the three outputs and clear it
  auto consume = [&](int b) __attribute__((always_inline)) {
    const float sc = sctab[b - b_begin];
    f32x4 y4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) y4[t] = *reinterpret_cast<const f32x4*>(ytab + (b - b_begin) * T_TILE + wr * 64 + t * 16 + (lane >> 4) * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = tk * T_TILE + wr * 64 + t * 16 + (lane >> 4) * 4 + e;
        const float yk = y4[t][e];
        float dyp = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float sv = accS[t][c][e] * sc;
          accB[t][c][e] += sv;
          accA[t][c][e] = fmaf(yk, sv, accA[t][c][e]);
          if (DY) dyp = fmaf(w1[t][e][c], sv, dyp);
          accS[t][c][e] = 0.f;
        }
        if (DY) {
          dyp = sum16_h2(dyp);
          if ((lane & 15) == 0) p.dy_part[((size_t)(tj * 4 + wc) * p.B + b) * p.d + k] = dyp;
        }
      }
  };

#pragma unroll 1
  for (int step = 0; step < p.nsteps; ++step) {
  vX.base = p.X.base + (size_t)step * p.x_step; vG.base = p.dI1.base + (size_t)step * p.g_step;
  yS = p.y + (size_t)step * p.y_step;
  if (total > 0) {                                        // in flight while the tables are built
    issue(0);
    issue(1);
    issue(2);
  }
  tables();
  if (total > 0) {
    wait_vmcnt<8>();                                      // stage 0 has landed (stages 1, 2 may be in flight)
    __syncthreads();                                      // ... for every wave's share of it, and the factor table is complete
    int qi = 0, qch = 0;
#pragma unroll 1
    for (int s = 0; s < total; ++s) {
      if (!(p.dbg & 2048)) issue(s + 3);                  // ring slot (s + 3) % 4 was last read in iteration s - 1
      if (!(p.dbg & 1024)) compute(s % SBH_RING, qi, qch);
      // stage s + 1 must have landed; this wave's 8 DMA instructions of stages s + 2, s + 3 stay in flight.  The wait sits in
      // front of the fold so that the fold's dy stores (younger than every DMA; vmcnt retires in order) are not waited for
      wait_vmcnt<8>();
      if (++qch == nchunk) {
        if (!(p.dbg & 512)) consume(b_begin + qi);
        qch = 0; ++qi;
      }
      __syncthreads();
    }
    wait_vmcnt<0>();                                      // the speculative stages past the end
  }
  __syncthreads();                                        // ... of every wave: the ring and the tables are about to be reused
  }   // step

  // ---- the two slabs leave through LDS: a wave's 64 x 32 share row-major in its own 9 KB of the (now idle) ring, read back
  //      as float4 so that a store instruction writes 8 rows x 128 contiguous bytes instead of 64-byte pieces
  float* oa = p.dW1a_part + (size_t)group * p.d * p.d;
  float* ob = p.dW1b_part + (size_t)group * p.d * p.d;
  constexpr int LDW = 36;
  float* tw = reinterpret_cast<float*>(lds) + wave * (64 * LDW);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          tw[(t * 16 + (lane >> 4) * 4 + e) * LDW + c * 16 + (lane & 15)] = which ? accB[t][c][e] : accA[t][c][e];
    float* o = (which ? ob : oa) + (size_t)(tk * T_TILE + wr * 64) * p.d + tj * T_TILE + wc * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = i * 8 + (lane >> 3), c4 = (lane & 7) * 4;
      *reinterpret_cast<f32x4*>(o + (size_t)r * p.d + c4) = *reinterpret_cast<const f32x4*>(tw + r * LDW + c4);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same contraction on a 128 x 256 output tile (round 4).  The 128 x 128 kernel above issues one transpose read per MFMA
// (each wave re-reads its 4 X fragments for only 2 dI1 column tiles): 17.9 M LDS instructions for 16.5 M MFMAs at the
// metric's shape -- it runs at the LDS's pace, 0.69 PF against wgrad_h2_kernel's 1.04.  A wave tile of 64 x 64 halves the
// X re-reads (0.67 reads per MFMA), but three accumulator sets of that size (S_b, dW1a, dW1b: 192 registers) do not fit.
// They are not needed: with T_k = S_1 + .. + S_k (the questions and steps of this workgroup in processing order)
//     dW1b = T_K,      dW1a = sum_k diag(y_k) S_k = sum_k diag(y_k - y_{k+1}) T_k      (y_{K+1} = 0; summation by parts)
// so ONE running accumulator that is never cleared serves both, and the fold at a question's end is
//     dW1a += (y_k - y_{k+1}) * T_k                       (two sets: 128 registers).
// T_k is kept in the current question's unit 2^-(EX_k + EG_k) and moved to the next question's with an exact power of two
// when the next question starts -- which is also when y_{k+1} is at hand (the LDS table of the step's questions), so the
// fold of question k runs at the start of question k + 1.  Rounding: a fold rounds at 2^-24 |T_k| instead of 2^-24 |S_k|;
// |T_k| grows like sqrt(k) |S| over the <= 2 questions x p steps of a workgroup.
// No dy (that is per question and needs S_b itself): the deferred all-steps launch only.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SBW_STAGE = 2 * WH_APL + 4 * WH_APL;     // X hi, X lo (128 columns) | dI1 hi, dI1 lo (256 columns): 48 KB
constexpr int SBW_RING = 3;
constexpr int SBW_MAXQ = 4;                            // questions per workgroup
constexpr int SBW_MAXROWS = 1024;                      // factor-table rows (SBW_MAXQ questions of <= 256 rows)

// CONT (round 5): ONE stage stream over all steps instead of a pipeline that is drained, re-primed and given fresh tables at every
// step.  The phase knobs priced the per-step restart at ~100 of the kernel's 350 us (profiles/r05_phase_knobs.txt: 99 us with both
// the products and the in-loop DMA skipped): per step two clamped stage re-reads past the end, a full drain, and tables() -- two
// rounds of global loads and three barriers -- in front of the first product.  Here stage g + 2 is requested while stage g is
// multiplied across step boundaries (nothing is requested twice, nothing drains), and the tables of step i + 1 (a second table
// set in LDS) are built INSIDE step i's loop, spread over its first three iterations behind their barriers: iteration 0 presets the
// minima, iteration 1 requests every row's exponent bytes and the y values (inline assembly, so that the explicit vmcnt(6) that
// already ends the iteration covers them -- a load the compiler knows about would be waited for with vmcnt(0) and drain the two
// stages in flight) and folds them into the questions' minima, iteration 2 writes the row factors.  Same products, folds and
// order as the per-step form: bit-identical results.  (Tried on top, not kept: a stage's G fragments read one iteration ahead,
// behind the previous iteration's barrier and under its last products -- one loop-carried fragment set more, 5 spilled registers,
// and a scratch reload inside this loop is a vmcnt(0): it drains both stages in flight every iteration.  The LDS-read exposure it
// would hide is ~0.1 us of a 1.9 us stage.)
constexpr int SBW_TABLES = SBW_MAXROWS * 2 + SBW_MAXQ * T_TILE * 4 + 4 * SBW_MAXQ * 4;      // one table set: ftab | ytab | qmn (16-byte multiple)
template <bool CONT>
__global__ __launch_bounds__(512) void sb_h2w_kernel(SbH2P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  char* tab0 = lds + SBW_RING * SBW_STAGE;
  uint16_t* ftab = reinterpret_cast<uint16_t*>(tab0);                                        // [questions][nchunk * 32]
  float* ytab = reinterpret_cast<float*>(tab0 + SBW_MAXROWS * 2);                            // [questions][128]
  int* qmn = reinterpret_cast<int*>(ytab + SBW_MAXQ * T_TILE);                               // [2][questions] EX | EG of the step's questions
  float* yold = reinterpret_cast<float*>(tab0 + (CONT ? 2 : 1) * SBW_TABLES);                // [2][128] y of the question before (ytab is rebuilt per step)

  const int ntk = p.d / T_TILE, ntj = p.d / (2 * T_TILE);
  const int ntile = ntk * ntj;
  const int nblk = gridDim.x;
  int v = blockIdx.x;
  if ((nblk & 7) == 0) v = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int group = v / ntile;
  const int tile = v % ntile;
  const int tk = tile / ntj, tj = tile % ntj;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;                // 64 rows (k) x 64 columns (j) of the 128 x 256 tile
  const int gq = wc >> 1;                                 // the 128-column block of dI1 this wave's columns lie in

  const int nchunk = (p.N + 31) >> 5;
  const int b_begin = group * p.qpg;
  const int b_end = min(p.B, b_begin + p.qpg);
  const int nq = b_end - b_begin;
  const int total = nq * nchunk;
  const size_t Rp = p.X.Rp();
  const size_t xpb = p.X.plane_bytes(), gpb = p.dI1.plane_bytes();
  const int xcb = p.X.cb(), gcb = p.dI1.cb();
  H2View vX = p.X, vG = p.dI1;
  const float* yS = p.y;
  const int rows_q = nchunk * 32;

  // ---- staging: 48 column tiles of 16 per stage (X: 2 planes x 8, dI1: 2 planes x 16).  Wave w copies X tiles 2 w, 2 w + 1
  //      and dI1 tiles 4 w .. 4 w + 3 (tile = plane * tiles-per-plane + column tile); every one of its six instructions is a
  //      wave-uniform base + the SAME per-lane offset (slot of row sl_m, column group sl_kg of the tile): one address register
  const int sl_half = lane >> 5, sl_row = (lane & 31) >> 1, sl_kg = lane & 1;
  const int sl_m = sl_half * 16 + sl_row;
  const uint32_t lds0 = lds_addr_of(lds);
  auto issue = [&](int s_raw) __attribute__((always_inline)) {
    const int s = min(s_raw, total - 1);
    const int qi = s / nchunk, ch = s - qi * nchunk;
    const int n = min(ch * 32 + sl_m, p.N - 1);
    const uint32_t voff = (uint32_t)(((size_t)sl_kg * Rp + (size_t)(b_begin + qi) * p.N + n) * 16);
    const uint32_t st = lds0 + (uint32_t)((s_raw % SBW_RING) * SBW_STAGE);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int u = wave * 2 + j, pl = u >> 3, ct = u & 7;
      dma16b_s(vX.base + pl * xpb + (size_t)(tk * 16 + 2 * ct) * Rp * 16, voff, st + pl * WH_APL + ct * 1024);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int u = wave * 4 + j, pl = u >> 4, ct = u & 15;
      dma16b_s(vG.base + pl * gpb + (size_t)(tj * 32 + 2 * ct) * Rp * 16, voff, st + 2 * WH_APL + pl * 2 * WH_APL + ct * 1024);
    }
  };
  // per step: the questions' common exponents (integer minima through LDS), the combined row factors (one table per dI1
  // column block of the tile: the chain kernels give a row ONE exponent, other producers one per 128 columns), y, units
  // (the table set in use: CONT alternates between two sets by step parity, the per-step form only ever uses set 0)
  uint16_t* ftab_c = ftab;
  float* ytab_c = ytab;
  int* qmn_c = qmn;
  auto tables = [&]() __attribute__((always_inline)) {
    if (tid < 3 * SBW_MAXQ) qmn[tid] = 127;
    __syncthreads();
    for (int i0 = 0; i0 < nq * rows_q; i0 += 512) {
      const int i = i0 + tid;
      const int qi = i / rows_q, n = i - qi * rows_q;
      int ex = 127, eg0 = 127, eg1 = 127;
      if (i < nq * rows_q && n < p.N) {
        const size_t row = (size_t)(b_begin + qi) * p.N + n;
        ex = (int)vX.exps()[row * xcb + tk];
        eg0 = (int)vG.exps()[row * gcb + 2 * tj];
        eg1 = (int)vG.exps()[row * gcb + 2 * tj + 1];
      }
      const int q_lo = min(i0 + (tid & ~63), nq * rows_q - 1) / rows_q, q_hi = min(i0 + (tid | 63), nq * rows_q - 1) / rows_q;
      for (int q = q_lo; q <= q_hi; ++q) {
        int mx = qi == q ? ex : 127, mg = qi == q ? min(eg0, eg1) : 127;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mx = min(mx, __shfl_xor(mx, o, 64)); mg = min(mg, __shfl_xor(mg, o, 64)); }
        if (lane == 0) { atomicMin(qmn + q, mx); atomicMin(qmn + SBW_MAXQ + q, mg); }
      }
    }
    __syncthreads();
    // the two column blocks of dI1 share ONE common exponent (their minimum): one accumulator unit per question
    for (int i = tid; i < 2 * nq * rows_q; i += 512) {
      const int blk = i / (nq * rows_q), r = i - blk * nq * rows_q;
      const int qi = r / rows_q, n = r - qi * rows_q;
      uint16_t f = 0;
      if (n < p.N) {
        const size_t row = (size_t)(b_begin + qi) * p.N + n;
        const int k = (min(qmn[qi], 126) - (int)vX.exps()[row * xcb + tk]) + (min(qmn[SBW_MAXQ + qi], 126) - (int)vG.exps()[row * gcb + 2 * tj + blk]);
        f = (uint16_t)(pk_pow2_f16(k) & 0xFFFFu);
      }
      ftab[blk * (SBW_MAXROWS / 2) + r] = f;
    }
    for (int i = tid; i < nq * T_TILE; i += 512)
      ytab[i] = yS[(size_t)(b_begin + (i >> 7)) * p.d + tk * T_TILE + (i & 127)];
  };

  f32x4 accT[4][4], accA[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) accT[t][c] = accA[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto frag = [&](const char* tile) __attribute__((always_inline)) {
    const u32x2 lo = tr_read(tile + lane * 8), hi = tr_read(tile + 512 + lane * 8);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  auto compute = [&](int buf, int qi, int ch) __attribute__((always_inline)) {
    const char* sa = lds + buf * SBW_STAGE + (wr * 4) * 1024;
    const char* sg = lds + buf * SBW_STAGE + 2 * WH_APL + (wc * 4) * 1024;
    const uint16_t* ft = ftab_c + gq * (SBW_MAXROWS / 2) + qi * rows_q + ch * 32 + (lane >> 4) * 4;
    const u32x2 f0 = *reinterpret_cast<const u32x2*>(ft), f1 = *reinterpret_cast<const u32x2*>(ft + 16);
    u32x4 gf[2][4];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        u32x4 w = frag(sg + pl * 2 * WH_APL + c * 1024);
        w[0] = pk_mul_f16(w[0], f0[0]); w[1] = pk_mul_f16(w[1], f0[1]);
        w[2] = pk_mul_f16(w[2], f1[0]); w[3] = pk_mul_f16(w[3], f1[1]);
        gf[pl][c] = w;
      }
#pragma unroll
    for (int ap = 1; ap >= 0; --ap) {                     // smallest terms first: X_lo x dI1_hi ; X_hi x {dI1_lo, dI1_hi}
      u32x4 af[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) af[t] = frag(sa + ap * WH_APL + t * 1024);
#pragma unroll
      for (int bp = 1 - ap; bp >= 0; --bp)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) accT[t][c] = mfma_f16(af[t], gf[bp][c], accT[t][c]);
    }
  };
  // the fold of the question that just ended, run when the next one (its y in `ynext`, its unit exponent `e_next`) starts:
  //   dW1a += (y_k - y_next) * T_k * 2^-e_k ;  T_k (unit 2^-e_k) -> unit 2^-e_next
  // (y of the question before lives in LDS, two buffers taken in turn: a question is at least one stage -- one barrier -- long,
  // so the buffer a fold reads was written a barrier ago and is overwritten two questions later)
  int ex_prev = 0, eg_prev = 0, ybuf = 0;
  bool pending = false;
  auto keep_y = [&](const float* ynext) __attribute__((always_inline)) {
    if (tid < T_TILE) yold[(ybuf ^ 1) * T_TILE + tid] = ynext ? ynext[tid] : 0.f;
    ybuf ^= 1;
  };
  auto fold = [&](const float* ynext, int ex_next, int eg_next) __attribute__((always_inline)) {
    const float sc = h2_unscale(ex_prev, eg_prev);
    const float mv = h2_pow2(max(min((ex_next + eg_next) - (ex_prev + eg_prev), 126), -126));
    const float* yp = yold + ybuf * T_TILE;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k0 = wr * 64 + t * 16 + (lane >> 4) * 4;
      f32x4 yn = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ynext) yn = *reinterpret_cast<const f32x4*>(ynext + k0);
      const f32x4 yo = *reinterpret_cast<const f32x4*>(yp + k0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dy = (yo[e] - yn[e]) * sc;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          accA[t][c][e] = fmaf(dy, accT[t][c][e], accA[t][c][e]);
          accT[t][c][e] *= mv;
        }
      }
    }
    keep_y(ynext);
    ex_prev = ex_next; eg_prev = eg_next;
  };

  if constexpr (CONT) {
    const int G = p.nsteps * total;
    auto issue_g = [&](int g_raw) __attribute__((always_inline)) {
      const int g = min(g_raw, G - 1);
      const int ist = g / total, sg = g - ist * total;
      const int qi = sg / nchunk, ch = sg - qi * nchunk;
      const int n = min(ch * 32 + sl_m, p.N - 1);
      const uint32_t voff = (uint32_t)(((size_t)sl_kg * Rp + (size_t)(b_begin + qi) * p.N + n) * 16);
      const uint32_t st = lds0 + (uint32_t)((g_raw % SBW_RING) * SBW_STAGE);
      const char* xb = p.X.base + (size_t)ist * p.x_step;
      const char* gb = p.dI1.base + (size_t)ist * p.g_step;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int u = wave * 2 + j, pl = u >> 3, ct = u & 7;
        dma16b_s(xb + pl * xpb + (size_t)(tk * 16 + 2 * ct) * Rp * 16, voff, st + pl * WH_APL + ct * 1024);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int u = wave * 4 + j, pl = u >> 4, ct = u & 15;
        dma16b_s(gb + pl * gpb + (size_t)(tj * 32 + 2 * ct) * Rp * 16, voff, st + 2 * WH_APL + pl * 2 * WH_APL + ct * 1024);
      }
    };
    issue_g(0);
    issue_g(1);
    tables();                                               // step 0's, into set 0 (vX, vG, yS are step 0's)
    wait_vmcnt<6>();                                        // stage 0 has landed (stage 1 may be in flight)
    __syncthreads();
    // this thread's share of the NEXT step's tables: one row of the workgroup's questions (nq * rows_q <= 512), one y value
    const int nrows = nq * rows_q;
    const int t_q = min(tid, nrows - 1) / rows_q, t_n = min(tid, nrows - 1) - t_q * rows_q;
    const bool t_row = tid < nrows && t_n < p.N;
    const size_t t_grow = (size_t)(b_begin + t_q) * p.N + min(t_n, p.N - 1);
    const bool t_yv = tid < nq * T_TILE;
    const size_t t_yoff = (size_t)(b_begin + (t_yv ? (tid >> 7) : 0)) * p.d + tk * T_TILE + (tid & 127);
    int t_ex = 127, t_g0 = 127, t_g1 = 127;
    float t_y = 0.f;
    int step = 0, sg = 0, qi = 0, qch = 0;
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
      const bool more = step + 1 < p.nsteps;
      char* nset = tab0 + ((step + 1) & 1) * SBW_TABLES;
      uint16_t* ftab_n = reinterpret_cast<uint16_t*>(nset);
      float* ytab_n = reinterpret_cast<float*>(nset + SBW_MAXROWS * 2);
      int* qmn_n = reinterpret_cast<int*>(ytab_n + SBW_MAXQ * T_TILE);
      if (more && sg == 1) {
        // the next step's exponent bytes of this thread's row and its y value: requested in front of this iteration's DMA, waited
        // for by the vmcnt(6) that ends it
        const H2View nX{const_cast<char*>(p.X.base) + (size_t)(step + 1) * p.x_step, p.X.R, p.X.C};
        const H2View nG{const_cast<char*>(p.dI1.base) + (size_t)(step + 1) * p.g_step, p.dI1.R, p.dI1.C};
        const int8_t* ax = nX.exps() + t_grow * xcb + tk;
        const int8_t* ag = nG.exps() + t_grow * gcb + 2 * tj;
        const float* ay = p.y + (size_t)(step + 1) * p.y_step + t_yoff;
        asm volatile("global_load_sbyte %0, %1, off" : "=v"(t_ex) : "v"(ax) : "memory");
        asm volatile("global_load_sbyte %0, %1, off" : "=v"(t_g0) : "v"(ag) : "memory");
        asm volatile("global_load_sbyte %0, %1, off offset:1" : "=v"(t_g1) : "v"(ag) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(t_y) : "v"(ay) : "memory");
      }
      if (!(p.dbg & 2048)) issue_g(g + 2);                  // ring slot (g + 2) % 3 was last read in iteration g - 1
      if (qch == 0) {
        const float* yn = ytab_c + qi * T_TILE;
        const int exn = min(qmn_c[qi], 126), egn = min(qmn_c[SBW_MAXQ + qi], 126);
        if (pending) {
          if (!(p.dbg & 512)) fold(yn, exn, egn);
        } else {
          keep_y(yn);
          ex_prev = exn; eg_prev = egn;
          pending = true;
        }
      }
      if (!(p.dbg & 1024)) compute(g % SBW_RING, qi, qch);
      if (p.dbg & 2048) wait_vmcnt<0>(); else wait_vmcnt<6>();      // stage g + 1 has landed (and this iteration's table loads); g + 2 stays in flight
      if (more) {
        if (sg == 0) {
          if (tid < 3 * SBW_MAXQ) qmn_n[tid] = 127;
        } else if (sg == 1) {
          asm volatile("" : "+v"(t_ex), "+v"(t_g0), "+v"(t_g1), "+v"(t_y));
          if (!t_row) { t_ex = 127; t_g0 = 127; t_g1 = 127; }
          // the questions' minimum exponents (integer minima through LDS, as tables() does for step 0)
          const int q_lo = min(tid & ~63, nrows - 1) / rows_q, q_hi = min(tid | 63, nrows - 1) / rows_q;
          for (int q = q_lo; q <= q_hi; ++q) {
            int mx = t_q == q ? t_ex : 127, mg = t_q == q ? min(t_g0, t_g1) : 127;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { mx = min(mx, __shfl_xor(mx, o, 64)); mg = min(mg, __shfl_xor(mg, o, 64)); }
            if (lane == 0) { atomicMin(qmn_n + q, mx); atomicMin(qmn_n + SBW_MAXQ + q, mg); }
          }
        } else if (sg == 2) {
          if (tid < nrows) {
            const int bx = min(qmn_n[t_q], 126), bg = min(qmn_n[SBW_MAXQ + t_q], 126);
            ftab_n[tid] = t_row ? (uint16_t)(pk_pow2_f16((bx - t_ex) + (bg - t_g0)) & 0xFFFFu) : (uint16_t)0;
            ftab_n[(SBW_MAXROWS / 2) + tid] = t_row ? (uint16_t)(pk_pow2_f16((bx - t_ex) + (bg - t_g1)) & 0xFFFFu) : (uint16_t)0;
          }
          if (t_yv) ytab_n[tid] = t_y;
        }
      }
      if (++qch == nchunk) { qch = 0; ++qi; }
      if (++sg == total) {
        sg = 0; qi = 0; ++step;
        char* cset = tab0 + (step & 1) * SBW_TABLES;
        ftab_c = reinterpret_cast<uint16_t*>(cset);
        ytab_c = reinterpret_cast<float*>(cset + SBW_MAXROWS * 2);
        qmn_c = reinterpret_cast<int*>(ytab_c + SBW_MAXQ * T_TILE);
      }
      __syncthreads();
    }
    wait_vmcnt<0>();
    __syncthreads();
  } else {
#pragma unroll 1
  for (int step = 0; step < p.nsteps; ++step) {
    vX.base = p.X.base + (size_t)step * p.x_step; vG.base = p.dI1.base + (size_t)step * p.g_step;
    yS = p.y + (size_t)step * p.y_step;
    if (total > 0) { issue(0); issue(1); }
    tables();
    if (total > 0) {
      wait_vmcnt<6>();                                      // stage 0 has landed (stage 1 may be in flight)
      __syncthreads();                                      // ... every wave's share of it; the tables are complete
      int qi = 0, qch = 0;
#pragma unroll 1
      for (int s = 0; s < total; ++s) {
        if (!(p.dbg & 2048)) issue(s + 2);                  // ring slot (s + 2) % 3 was last read in iteration s - 1
        if (qch == 0) {
          // a question starts: the fold of the one before it (none in front of the very first), or just its y and unit
          const float* yn = ytab + qi * T_TILE;
          const int exn = min(qmn[qi], 126), egn = min(qmn[SBW_MAXQ + qi], 126);
          if (pending) {
            if (!(p.dbg & 512)) fold(yn, exn, egn);
          } else {
            keep_y(yn);
            ex_prev = exn; eg_prev = egn;
            pending = true;
          }
        }
        if (!(p.dbg & 1024)) compute(s % SBW_RING, qi, qch);
        wait_vmcnt<6>();                                    // stage s + 1 has landed; stage s + 2 stays in flight
        if (++qch == nchunk) { qch = 0; ++qi; }
        __syncthreads();
      }
      wait_vmcnt<0>();
    }
    __syncthreads();                                        // the ring and the tables are about to be reused
  }
  }
  if (pending && !(p.dbg & 512)) fold(nullptr, ex_prev, eg_prev);     // y_{K+1} = 0; T stays in its unit
  const float scT = h2_unscale(ex_prev, eg_prev);

  // ---- the two slabs leave through LDS (a wave's 64 x 64 share row-major in its own 17 KB of the idle ring)
  float* oa = p.dW1a_part + (size_t)group * p.d * p.d;
  float* ob = p.dW1b_part + (size_t)group * p.d * p.d;
  constexpr int LDW = 68;
  float* tw = reinterpret_cast<float*>(lds) + wave * (64 * LDW);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          tw[(t * 16 + (lane >> 4) * 4 + e) * LDW + c * 16 + (lane & 15)] = which ? accT[t][c][e] * scT : accA[t][c][e];
    float* o = (which ? ob : oa) + (size_t)(tk * T_TILE + wr * 64) * p.d + tj * 2 * T_TILE + wc * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = i * 4 + (lane >> 4), c4 = (lane & 15) * 4;
      *reinterpret_cast<f32x4*>(o + (size_t)r * p.d + c4) = *reinterpret_cast<const f32x4*>(tw + r * LDW + c4);
    }
  }
}

inline bool sb_h2_wide_ok(int B, int N, int d) { return d % 256 == 0 && B >= 1 && N >= 1 && ((N + 31) / 32) * 32 <= SBW_MAXROWS / 2; }
// questions per workgroup of the wide kernel: tiles x groups ~ 256 workgroups
inline int sb_h2_wide_qpg(int B, int N, int d) {
  const int tiles = (d / 128) * (d / 256);
  int groups = 256 / tiles;
  if (groups < 1) groups = 1;
  int qpg = (B + groups - 1) / groups;
  const int rows_q = ((N + 31) / 32) * 32;
  const int cap = min(SBW_MAXQ, (SBW_MAXROWS / 2) / rows_q);
  if (qpg > cap) qpg = cap;
  return qpg < 1 ? 1 : qpg;
}
inline int sb_cont_mode() { return tune_get(MACX_TUNE_SB_CONT, 1); }       // MACX_TUNE_SB_CONT = 0 | 1: sb_h2w_kernel per step / as one stage stream
inline hipError_t sb_h2w_launch(const SbH2P& p, hipStream_t st) {
  const int nchunk = (p.N + 31) >> 5;
  if (p.qpg < 1 || p.qpg > SBW_MAXQ || p.qpg * nchunk * 32 > SBW_MAXROWS / 2 || p.d % 256 || p.dy_part || p.nsteps < 1) return hipErrorInvalidValue;
  const int ngroup = (p.B + p.qpg - 1) / p.qpg;
  // the continuous stage stream builds the next step's tables in iterations 0 - 2 of a step: at least three stages per step, and
  // one table row per thread
  const bool cont = sb_cont_mode() && p.nsteps > 1 && nchunk >= 3 && p.qpg * nchunk * 32 <= 512;
  if (cont) {
    constexpr size_t lds = (size_t)SBW_RING * SBW_STAGE + 2 * SBW_TABLES + 2 * T_TILE * 4;
    static_assert(lds <= 160 * 1024, "two table sets fit beside the ring");
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(sb_h2w_kernel<true>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sb_h2w_kernel<true>, dim3((p.d / 128) * (p.d / 256) * ngroup), dim3(512), lds, st, p);
    return hipGetLastError();
  }
  constexpr size_t lds = (size_t)SBW_RING * SBW_STAGE + SBW_TABLES + 2 * T_TILE * 4;
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(sb_h2w_kernel<false>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(sb_h2w_kernel<false>, dim3((p.d / 128) * (p.d / 256) * ngroup), dim3(512), lds, st, p);
  return hipGetLastError();
}

inline hipError_t sb_h2_launch(const SbH2P& p, hipStream_t st) {
  const int nchunk = (p.N + 31) >> 5;
  if (p.qpg * nchunk * 32 > SBH_MAXROWS || p.qpg > SBH_MAXQ) return hipErrorInvalidValue;
  constexpr size_t lds = SBH_RING * SBH_STAGE + SBH_MAXROWS * 2 + SBH_MAXQ * T_TILE * 4 + SBH_MAXQ * 4 + 2 * SBH_MAXQ * 4;
  if (p.nsteps < 1 || (p.nsteps > 1 && p.dy_part)) return hipErrorInvalidValue;    // dy is per step: its buffer is not
  auto kern = p.dy_part ? sb_h2_kernel<true> : sb_h2_kernel<false>;
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  const int nt = p.d / T_TILE;
  const int ngroup = (p.B + p.qpg - 1) / p.qpg;
  hipLaunchKernelGGL(kern, dim3(nt * nt * ngroup), dim3(512), lds, st, p);
  return hipGetLastError();
}

}  // namespace macx
