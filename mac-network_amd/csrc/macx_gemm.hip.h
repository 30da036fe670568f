// macx_gemm.hip.h -- the knowledge-base GEMM family of the read unit (forward and backward-data).
//
// One kernel template covers every  [B*N, K] x [K, d]  contraction of the read unit
// (mac_cell.py:209-277 / ops.py:668-725, 298-333): projX, memKbProj, memKbProj_2 forward and the
// three dY @ W^T products of the backward pass.  fp32 in / fp32 accumulate on
// v_mfma_f32_16x16x4_f32 (exact fp32, bit-equal to an fmaf chain, 157 TF peak on MI355X).
//
// Tiling is per QUESTION, not over the flat row index: a workgroup owns RT*16 consecutive
// knowledge-base cells of ONE question x 128 output columns.  With N = 196 (CLEVR) RT = 13 covers a
// question in one tile (208 rows, 6 % padding) and the grid is B x 4 = 256 workgroups for B = 64 --
// one per CU -- and every per-question vector the read unit broadcasts over the KB (projected memory
// y, control c) is workgroup-uniform, which is what lets the row-broadcast products move out of the
// [B,N,d] tensors and into the weight tile (B_YMIX_*) or the epilogue (E_I2_LOGIT).
//
// Workgroup = 8 waves (2 per SIMD, so one wave's LDS/global/barrier waits hide under the other's
// MFMAs); wave w owns output columns [16w, 16w+16) of the tile and all RT row tiles.
//
// Layouts
//   A  (activations)  row-major [B][N][lda], read as float4 along k, staged in LDS with a
//                     128-byte rows whose 16-byte chunks are XOR-swizzled by (row & 7) (conflict-free
//                     ds_read_b128 for the 16x16x4 A fragment without padding).
//   W  (weights)      PRE-PACKED [K/16][4][Nout][4]  with  Wp[Q][g][j][e] = W[16Q + 4g + e][j]
//                     so a lane's four consecutive MFMA B operands are one ds_read_b128 and the
//                     global->LDS copy is linear.  MFMA e of group Q contracts k = {16Q+e, 16Q+4+e,
//                     16Q+8+e, 16Q+12+e}; the A side uses the same pairing (lane group g reads
//                     A[i][16Q+4g .. 16Q+4g+3]).
//   dropout masks     1 bit per element, 32 consecutive k (or columns) per uint32 word, produced by
//                     mask_bits_kernel from the stateless stream of macx_common.hip.h.
//   epilogue          accumulators are transposed through LDS into a row-major [RT*16][132] tile
//                     and then processed 16 bytes per lane, 512 B per row: coalesced stores, row
//                     reductions (attention logits) inside one half-wave, column sums (bias
//                     gradients) by a fixed-order LDS combine.
#pragma once
#include <type_traits>
#include "macx_common.hip.h"

namespace macx {

constexpr int G_BK = 32;           // reduction slice per stage
constexpr int G_LDA = G_BK;        // LDS row stride of the A stage (floats); 16-byte chunks XOR-swizzled by row

enum : int { A_PLAIN = 0, A_DROP = 1 };
enum : int { B_PLAIN = 0, B_YMIX_ROW = 1, B_YMIX_COL = 2 };
enum : int {
  E_BIAS_ACT = 0,   // out = act(acc + bias)                              (X: act NON; H1: readMemAct)
  E_I2_LOGIT = 2,   // I2 = acc + bias; logit partial of act(I2*c) . w_k  (read-unit attention logits)
  E_MUL_DACT = 3,   // out = acc * act'(aux)                              (dI1 = (dI2 W2^T) * act'(I1))
  E_PLAIN = 4,      // out = acc                                          (dX)
  E_DKB = 5         // out (+)= acc * dropmask + att[b,n] * dr[b][j]      (dKB accumulation)
};

struct GemmP {
  // problem
  int B, N, K, Nout;
  // A operand
  const float* A;
  int lda;
  const uint32_t* a_bits; // A_DROP: keep bits of A, [B*N][lda/32]
  float a_inv_keep;
  // implicit-GEMM 3x3 convolution over a halo-padded NHWC image (stem CNN, ops.py:380-438): output
  // pixel n = (y, x) of a conv_w-wide image reads padded row (y+1)*conv_wp + (x+1) + tap offset, the
  // reduction index is (tap, channel): K = conv_taps * conv_cin.  conv_taps == 0: plain GEMM.
  int conv_taps, conv_w, conv_wp, conv_cin, conv_sign;
  size_t a_qstride;       // floats between consecutive questions/images in A (0: N * lda)
  // weights
  const float* Wp;        // packed
  const float* Wp2;       // packed second weight (B_YMIX_*)
  const float* y;         // [B][ldy] per-question vector mixed into the weight tile
  int ldy;
  // epilogue
  float* out;
  int ldo;
  const float* bias;      // [Nout]
  int act;                // activation code (E_BIAS_ACT, E_I2_LOGIT, E_MUL_DACT)
  const float* aux;       // E_MUL_DACT: H1 [B*N][ldo];  E_DKB: dr [B][ld_aux]
  int ld_aux;
  const float* cvec;      // E_I2_LOGIT: control [B][Nout]
  const float* wvec;      // E_I2_LOGIT: logits weight [Nout]
  const float* att;       // E_DKB: kb attention [B][N]
  float* logit_part;      // E_I2_LOGIT: [Nout/128][B*N]
  float* colsum_part;     // optional: column sums of `out` per workgroup-row  [B*nrb][Nout]
  const uint32_t* e_bits; // E_I2_LOGIT: keep bits of act(I2*c) [B*N][Nout/32]; E_DKB: keep bits of KB; null = keep all
  float e_inv_keep;
  int accumulate;         // E_DKB: 1 -> out += , 0 -> out =
  int dbg;                // measurement knobs (macx_opts.tune[MACX_TUNE_PHASE_MASK]): 1 skip epilogue, 2 skip in-loop staging, 8 no epilogue stores
  const float* a_maxabs;  // kb_gemm3h_kernel: largest magnitude of A (device float)
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// NW = waves per workgroup = 16-column slabs per tile.  NW = 4 (64-column tiles, 256 threads, ~69 KB
// LDS) lets two independent workgroups share a CU, so one's load/store/barrier phase overlaps the
// other's MFMA phase; NW = 8 (128-column tiles, 512 threads) reads the A panel half as often.
// 16-byte global -> LDS DMA (global_load_lds_dwordx4): the 64 lanes of a wave fetch 64 x 16 B from
// per-lane global addresses into LDS at wave_base + lane*16.  No VGPR round trip, no ds_write.
// 16-byte global store, plain or write-through (sc0 sc1: the bytes leave the XCD's L2 as they are issued instead of in
// the end-of-kernel write-back burst; MI355X_MICROARCH "publish-large").  Timing experiment only.
__device__ __forceinline__ void store16(float* g, f32x4 v) { *reinterpret_cast<f32x4*>(g) = v; }

__device__ __forceinline__ void dma16(const float* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// ... for a wave-uniform run-time n (the immediate has to be a constant); n > 16 waits for everything
__device__ __forceinline__ void wait_vmcnt_n(int n) {
  switch (n) {
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 9: wait_vmcnt<9>(); break;
    case 10: wait_vmcnt<10>(); break;
    case 11: wait_vmcnt<11>(); break;
    case 12: wait_vmcnt<12>(); break;
    case 13: wait_vmcnt<13>(); break;
    case 14: wait_vmcnt<14>(); break;
    case 15: wait_vmcnt<15>(); break;
    case 16: wait_vmcnt<16>(); break;
    default: wait_vmcnt<0>(); break;
  }
}

template <int RT, int NW>
constexpr int kb_gemm_lds_floats() {
  constexpr int stage = 2 * RT * 16 * G_LDA + 2 * G_BK * 16 * NW;
  constexpr int epi = RT * 16 * (16 * NW + 4) + 256 * NW;   // tile + [16 row groups][4 NW float4] column partials
  return stage > epi ? stage : epi;
}

// ---- epilogue, step 2 (shared by the fp32 and the split-bf16 kernels): the workgroup's accumulators sit row-major in
// LDS (T[ROWS][G_BN + 4]); one float4 per lane, one row per CG lanes.
template <int RT, int NW, int EP, bool COLSUM>
__device__ __forceinline__ void kb_epilogue_rows(const GemmP& p, float* smem, int b, int cb, int rbi, int nrb, int row0, int row_end) {
  constexpr int G_THREADS = 64 * NW;
  constexpr int G_BN = 16 * NW;
  constexpr int G_LDT = G_BN + 4;
  constexpr int CG = G_BN / 4;
  constexpr int RG = G_THREADS / CG;
  constexpr int ROWS = RT * 16;
  const int tid = threadIdx.x;
  float* T = smem;
  f32x4* red = reinterpret_cast<f32x4*>(smem + ROWS * G_LDT);   // [RG][CG] column partials
  // ---- step 2: row-major pass, one float4 per lane, one row per half-wave
  const int c4 = tid % CG;
  const int rg = tid / CG;
  const int col = cb * G_BN + c4 * 4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, cj = bias4, wj = bias4, drj = bias4;
  if (EP == E_BIAS_ACT || EP == E_I2_LOGIT) bias4 = *reinterpret_cast<const f32x4*>(p.bias + col);
  if (EP == E_I2_LOGIT) {
    cj = *reinterpret_cast<const f32x4*>(p.cvec + (size_t)b * p.Nout + col);
    wj = *reinterpret_cast<const f32x4*>(p.wvec + col);
  }
  if (EP == E_DKB) drj = *reinterpret_cast<const f32x4*>(p.aux + (size_t)b * p.ld_aux + col);
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  const int wpr = p.Nout >> 5;   // mask words per output row
  // every global operand the epilogue needs (activation outputs for act', the running dKB, mask words,
  // attention weights) is requested for ALL of this thread's rows before the first one is used: RT
  // independent loads in flight instead of RT serial round trips
  f32x4 auxv[RT];
  uint32_t bitv[RT];
  float attv[RT];
  if (EP == E_MUL_DACT || EP == E_DKB || EP == E_I2_LOGIT) {
#pragma unroll
    for (int it = 0; it < RT; ++it) {
      const int n = row0 + rg + it * RG;
      const size_t orow = (size_t)b * p.N + min(n, row_end - 1);
      if (EP == E_MUL_DACT) {
        auxv[it] = *reinterpret_cast<const f32x4*>(p.aux + orow * p.ldo + col);
        bitv[it] = p.e_bits ? p.e_bits[orow * wpr + (col >> 5)] >> (col & 31) : 0xFu;
      }
      if (EP == E_DKB) {
        auxv[it] = p.accumulate ? *reinterpret_cast<const f32x4*>(p.out + orow * p.ldo + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        attv[it] = p.att[orow];
        bitv[it] = p.e_bits ? p.e_bits[orow * (p.ldo >> 5) + (col >> 5)] >> (col & 31) : 0xFu;
      }
      if (EP == E_I2_LOGIT) bitv[it] = p.e_bits ? p.e_bits[orow * wpr + (col >> 5)] >> (col & 31) : 0xFu;
    }
  }
#pragma unroll
  for (int it = 0; it < RT; ++it) {
    const int lrow = rg + it * RG;
    const int n = row0 + lrow;
    const bool ok = n < row_end && !(p.dbg & 8);     // dbg 8: timing experiment, epilogue without its global stores
    const size_t orow = (size_t)b * p.N + (n < row_end ? n : row_end - 1);
    f32x4 val = *reinterpret_cast<const f32x4*>(T + lrow * G_LDT + c4 * 4);
    float* optr = p.out + orow * p.ldo + col;
    if (EP == E_BIAS_ACT) {
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] = act_apply(p.act, val[e] + bias4[e]);
      if (ok) store16(optr, val);
    } else if (EP == E_I2_LOGIT) {
      val += bias4;
      if (ok) store16(optr, val);
      // mac_cell.py:248,262,266: act(I2 * c) -> dropout -> . w   (bias b_k added in kb_attend)
      const uint32_t bits = bitv[it];
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float g = act_apply(p.act, val[e] * cj[e]);
        g = ((bits >> e) & 1u) ? g * p.e_inv_keep : 0.f;
        part = fmaf(g, wj[e], part);
      }
      // the CG lanes of one row are contiguous and CG-aligned inside a wave
#pragma unroll
      for (int o = CG / 2; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      if (c4 == 0 && ok) p.logit_part[(size_t)cb * p.B * p.N + orow] = part;
    } else if (EP == E_MUL_DACT) {
      const f32x4 h = auxv[it];
      const uint32_t bits = bitv[it];      // optional dropout mask of the tensor this gradient flows into
      const float ik = p.e_bits ? p.e_inv_keep : 1.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] *= ((bits >> e) & 1u) ? act_grad_from_out(p.act, h[e]) * ik : 0.f;
      if (ok) store16(optr, val);
    } else if (EP == E_PLAIN) {
      if (ok) store16(optr, val);
    } else if (EP == E_DKB) {
      const uint32_t bits = bitv[it];
      const float a = attv[it];
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] = (((bits >> e) & 1u) ? val[e] * p.e_inv_keep : 0.f) + a * drj[e];
      val += auxv[it];
      if (ok) store16(optr, val);
    }
    if (COLSUM && ok) csum += val;
  }
  if (COLSUM) {
    red[rg * CG + c4] = csum;
    __syncthreads();
    if (tid < CG) {
      f32x4 t = red[tid];
#pragma unroll
      for (int g = 1; g < RG; ++g) t += red[g * CG + tid];
      *reinterpret_cast<f32x4*>(p.colsum_part + (size_t)(b * nrb + rbi) * p.Nout + cb * G_BN + tid * 4) = t;
    }
  }
}

template <int RT, int NW, int AP, int BP, int EP, bool COLSUM>
__global__ __launch_bounds__(64 * NW) void kb_gemm_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int G_THREADS = 64 * NW;
  constexpr int G_BN = 16 * NW;             // output columns per workgroup
  constexpr int G_BTILE = G_BK * G_BN;
  constexpr int G_LDT = G_BN + 4;           // epilogue tile row stride (floats)
  constexpr int B_F4 = G_BTILE / 4;         // float4 per B stage
  constexpr int B_IT = B_F4 / G_THREADS;    // = 2
  constexpr int CG = G_BN / 4;              // float4 column groups per row in the epilogue
  constexpr int RG = G_THREADS / CG;        // row groups in the epilogue (= 16)
  constexpr int ROWS = RT * 16;
  constexpr int A_TILE = ROWS * G_LDA;
  constexpr int A_F4 = ROWS * 8;                       // float4 per A stage
  constexpr int A_IT = (A_F4 + G_THREADS - 1) / G_THREADS;
  float* sA = smem;                  // [2][A_TILE]
  float* sB = smem + 2 * A_TILE;     // [2][G_BTILE]

  // ---- workgroup -> (question, row block, column block); XCD-aware so that the column blocks
  // of one question (which share the A rows) sit on one XCD's L2.
  const int nblk = gridDim.x;
  int v = blockIdx.x;
  if ((nblk & 7) == 0) v = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int ncb = p.Nout / G_BN;
  const int nrb = (p.N + ROWS - 1) / ROWS;
  const int cb = v % ncb;
  const int rbi = (v / ncb) % nrb;
  const int b = v / (ncb * nrb);
  // the ceil(N/16) row tiles of a question are dealt evenly over its nrb row blocks (196 rows, RT = 7: 7 + 6 tiles), so
  // only the question's last tile carries padding; a block with RT - 1 tiles skips the MFMAs of its last slot
  const int ntiles = (p.N + 15) >> 4;
  const int tbase = ntiles / nrb, textra = ntiles - tbase * nrb;
  const int nt = tbase + (rbi < textra ? 1 : 0);
  const int row0 = (rbi * tbase + min(rbi, textra)) << 4;
  const int row_end = min(p.N, row0 + (nt << 4));       // rows >= row_end belong to the next block (or to nobody)
  const bool full = nt >= RT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nk = p.K / G_BK;

  f32x4 acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 ra[A_IT];
  uint32_t rbits[A_IT];
  f32x4 rw[B_IT];
  f32x4 rw2[B_IT];
  f32x4 ry[B_IT];        // B_YMIX_ROW: the slice of y_b the weight rows are scaled by (fetched with the tile, not at use)

  const bool conv = p.conv_taps > 0;
  const float* Abase = p.A + (size_t)b * (p.a_qstride ? p.a_qstride : (size_t)p.N * p.lda);
  const uint32_t* Bitbase = (AP == A_DROP) ? p.a_bits + (size_t)b * p.N * (p.lda >> 5) : nullptr;
  // per-thread staging coordinates (k-invariant)
  int a_off[A_IT];        // float offset of this thread's float4 in A (row clamped into the question)
  int a_row[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int f = tid + G_THREADS * i;
    const int n = row0 + (f >> 3);
    a_ok[i] = (f < A_F4) && (n < row_end);
    const int nc = min(n, p.N - 1);
    a_row[i] = nc;
    const int srow = conv ? (nc / p.conv_w + 1) * p.conv_wp + (nc % p.conv_w) + 1 : nc;
    a_off[i] = srow * p.lda + (f & 7) * 4;
  }
  float ycol = 0.f;
  if (BP == B_YMIX_COL) ycol = p.y[(size_t)b * p.ldy + cb * G_BN + (tid % G_BN)];

  auto load_tiles = [&](int kt) {
    int koff = kt * G_BK;
    if (conv) {   // slice kt = 32 channels of one tap: a row shift inside the padded image
      const int per = p.conv_cin >> 5;
      const int tap = kt / per;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      koff = p.conv_sign * (dy * p.conv_wp + dx) * p.lda + ((kt - tap * per) << 5);
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      ra[i] = *reinterpret_cast<const f32x4*>(Abase + a_off[i] + koff);
      if (AP == A_DROP) rbits[i] = Bitbase[a_row[i] * (p.lda >> 5) + kt];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int f = tid + G_THREADS * i;
      const int chunk = f / G_BN;   // Q*4 + g
      const int j = f % G_BN;
      const size_t off = ((size_t)(kt * 8 + chunk) * p.Nout + cb * G_BN + j) * 4;
      rw[i] = *reinterpret_cast<const f32x4*>(p.Wp + off);
      if (BP != B_PLAIN) rw2[i] = *reinterpret_cast<const f32x4*>(p.Wp2 + off);
      if (BP == B_YMIX_ROW) ry[i] = *reinterpret_cast<const f32x4*>(p.y + (size_t)b * p.ldy + kt * G_BK + (chunk >> 2) * 16 + (chunk & 3) * 4);
    }
  };

  auto store_tiles = [&](int buf, int kt) {
    float* dA = sA + buf * A_TILE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int f = tid + G_THREADS * i;
      f32x4 val = ra[i];
      if (AP == A_DROP) {
        const uint32_t bits = rbits[i] >> ((f & 7) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = ((bits >> e) & 1u) ? val[e] * p.a_inv_keep : 0.f;
      }
      if (!a_ok[i]) val = f32x4{0.f, 0.f, 0.f, 0.f};
      // chunk c of row r lives at chunk slot c ^ (r & 7): conflict-free b128 fragment reads, no padding
      if (f < A_F4) *reinterpret_cast<f32x4*>(dA + (f >> 3) * G_LDA + (((f & 7) ^ ((f >> 3) & 7)) << 2)) = val;
    }
    float* dB = sB + buf * G_BTILE;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int f = tid + G_THREADS * i;
      f32x4 val = rw[i];
      if (BP == B_YMIX_ROW) {
        // B_eff[k][j] = y[b][k] * W1a[k][j] + W1b[k][j]   (ops.py:703,718 folded into the weights)
        val = val * ry[i] + rw2[i];
      } else if (BP == B_YMIX_COL) {
        // B_eff[k][j] = y[b][j] * W1a^T[k][j] + W1b^T[k][j]   (backward-data of the same product)
        val = val * ycol + rw2[i];
      }
      *reinterpret_cast<f32x4*>(dB + f * 4) = val;
    }
  };

  auto compute = [&](int buf, int Q) {
    // lane (i = lane & 15, g = lane >> 4) reads A[i][16Q + 4g ..] = chunk (4Q + g) ^ (i & 7) of row i
    const float* a = sA + buf * A_TILE + (lane & 15) * G_LDA + ((((Q << 2) + (lane >> 4)) ^ (lane & 7)) << 2);
    const float* bq = sB + buf * G_BTILE + ((lane >> 4) * G_BN + wave * 16 + (lane & 15)) * 4;
    const f32x4 bf = *reinterpret_cast<const f32x4*>(bq + Q * 4 * G_BN * 4);
    f32x4 af[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) af[r] = *reinterpret_cast<const f32x4*>(a + r * 16 * G_LDA);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < RT - 1; ++r) acc[r] = mfma16(af[r][e], bf[e], acc[r]);
    if (full) {      // workgroup-uniform: the last tile slot of a block that was dealt RT - 1 tiles is empty
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[RT - 1] = mfma16(af[RT - 1][e], bf[e], acc[RT - 1]);
    }
  };

  // ---- main loop: register-staged double buffer, one barrier per k-slice.  The next slice's LDS
  // stores sit between the two halves of the current slice's MFMAs so that they drain under them.
  load_tiles(0);
  store_tiles(0, 0);
  __syncthreads();
  const bool stage = !(p.dbg & 2);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = stage ? (kt & 1) : 0;
    if (stage && kt + 1 < nk) load_tiles(kt + 1);
    compute(cur, 0);
    compute(cur, 1);
    if (stage && kt + 1 < nk) store_tiles(cur ^ 1, kt + 1);
    __syncthreads();
  }
  if (p.dbg & 1) {
    if (acc[0][0] == 123.456f) p.out[0] = acc[RT - 1][3];
    return;
  }

  // ---- epilogue, step 1: accumulators -> row-major LDS tile.
  // 16x16 accumulator map: col = lane & 15, row = (lane >> 4) * 4 + reg
  float* T = smem;                              // [ROWS][G_LDT]
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) T[(r * 16 + (lane >> 4) * 4 + e) * G_LDT + wave * 16 + (lane & 15)] = acc[r][e];
  __syncthreads();

  kb_epilogue_rows<RT, NW, EP, COLSUM>(p, smem, b, cb, rbi, nrb, row0, row_end);
}

// ---- host-side launcher ---------------------------------------------------------------------
// per-call knob (MACX_TUNE_NATIVE_WAVES): waves per workgroup of the kb GEMM, 4 or 8
inline int kb_gemm_nw() { const int v = tune_get(MACX_TUNE_NATIVE_WAVES, 8); return v == 4 ? 4 : 8; }
inline int kb_gemm_dbg() { return tune_get(MACX_TUNE_PHASE_MASK, 0); }      // timing experiments of the profiling tools
inline int& gemm_default_mode() { static int m = 2; return m; }     // process default (macx_gemm_mode)
inline int& gemm_call_override() { static thread_local int m = -1; return m; }   // set for the duration of one ABI call (macx_opts.gemm_family)
inline int gemm_split_mode() { return gemm_call_override() >= 0 ? gemm_call_override() : gemm_default_mode(); }   // kernel family of the read unit: 0 native f32 MFMA, 1 split-bf16 (macx_gemm6.hip.h), 2 H2 fp16 planes (macx_gemm_h2.hip.h)
inline int kb_gemm_force_rt() { const int v = tune_get(MACX_TUNE_ROW_TILES, 0); return (v == 1 || v == 2 || v == 4 || v == 7 || v == 13) ? v : 0; }   // tuning override (macx_debug_set key 2)

template <int RT, int NW, int AP, int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm_launch_rt(const GemmP& p, hipStream_t st) {
  auto kern = kb_gemm_kernel<RT, NW, AP, BP, EP, COLSUM>;
  constexpr size_t lds = (size_t)kb_gemm_lds_floats<RT, NW>() * sizeof(float);
  {
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);   // per (kernel, device)
    if (e != hipSuccess) return e;
  }
  const int ncb = p.Nout / (16 * NW);
  const int nrb = (p.N + RT * 16 - 1) / (RT * 16);
  const int grid = p.B * nrb * ncb;
  GemmP q = p;
  q.dbg = kb_gemm_dbg();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, st, q);
  return hipGetLastError();
}

// row tiles per workgroup: the smallest of {1,2,4,7,13} x 16 rows that covers a question's N cells
// (196 -> 13, 49 -> 4, 14 -> 1; N > 208 -> 13 with several row blocks per question).  With few
// questions per GPU (small per-rank batches, strong scaling) one tile per question would leave most of
// the 256 CUs idle, so the tile height drops until the grid reaches ~one workgroup per CU.
inline int kb_gemm_pick_rt(int N, int B, int ncb = 4) {
  static const int cand[5] = {13, 7, 4, 2, 1};
  if (kb_gemm_force_rt()) return kb_gemm_force_rt();
  int k = 0;
  if (N <= 16) k = 4; else if (N <= 32) k = 3; else if (N <= 64) k = 2; else if (N <= 112) k = 1;
  auto blocks = [&](int rt) { return B * ncb * ((N + rt * 16 - 1) / (rt * 16)); };
  while (k < 3 && blocks(cand[k]) < 200) ++k;      // not below 2 x 16 rows: the weight panel traffic grows as tiles shrink
  return cand[k];
}
inline int kb_gemm_rows(int N, int B) { return kb_gemm_pick_rt(N, B) * 16; }

template <int NW, int AP, int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm_launch_nw(const GemmP& p, hipStream_t st) {
  switch (kb_gemm_pick_rt(p.N, p.B, p.Nout / (16 * NW))) {
    case 1: return kb_gemm_launch_rt<1, NW, AP, BP, EP, COLSUM>(p, st);
    case 2: return kb_gemm_launch_rt<2, NW, AP, BP, EP, COLSUM>(p, st);
    case 4: return kb_gemm_launch_rt<4, NW, AP, BP, EP, COLSUM>(p, st);
    case 7: return kb_gemm_launch_rt<7, NW, AP, BP, EP, COLSUM>(p, st);
    default: return kb_gemm_launch_rt<13, NW, AP, BP, EP, COLSUM>(p, st);
  }
}

template <int AP, int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm_launch(const GemmP& p, hipStream_t st) {
  if (kb_gemm_nw() == 8) return kb_gemm_launch_nw<8, AP, BP, EP, COLSUM>(p, st);
  return kb_gemm_launch_nw<4, AP, BP, EP, COLSUM>(p, st);
}

}  // namespace macx
