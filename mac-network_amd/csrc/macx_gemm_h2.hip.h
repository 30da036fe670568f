// macx_gemm_h2.hip.h -- the knowledge-base GEMM family on H2 operands (macx_h2.hip.h): [B*N, K] x [K, Nout] with both
// operands as two fp16 planes, three MFMA terms per product on v_mfma_f32_16x16x32_f16, fp32 accumulation.
//
// Same per-question tiling as the other two families (RT*16 rows of ONE question x 128 columns, 8 waves arranged
// 2 row halves x 4 column groups, grid = B x Nout/128 = one workgroup per CU at the CLEVR shape), but
//   * the A operand arrives pre-split from the kernel that produced it and goes from HBM/L2 straight into LDS by LDS-DMA
//     (global_load_lds_dwordx4, 64 lanes x 16 B landing lane-linear): no registers, no ds_write, no vector-ALU work;
//     pre-split weights (B_PLAIN) travel the same way;
//   * 32-wide K slices through a THREE-stage LDS ring (43 KB per stage at RT = 13): the DMA of slice s+2 is issued before
//     the MFMAs of slice s, and the wait at the end of slice s (an explicit s_waitcnt vmcnt(n) -- see dma16b in
//     macx_h2.hip.h for why it has to be explicit) only asks for slice s+1, so a slice has two multiply phases to land;
//   * per-(row, 128-column block) exponents: the MFMAs of one 128-wide K block accumulate in a block accumulator that is
//     folded into the running sum with the row's exact power-of-two factor (4 folds per launch at K = 512);
//   * weights: B_PLAIN pre-split once per call (pack format 3, per-matrix exponent); B_YMIX_* mixed with the question's
//     vector in fp32 while staging as before, scaled by a per-question bound, then split.
// Epilogue: accumulators -> row-major LDS tile -> ONE pass with lanes running along rows (bias / activation / act' from an
// H2 tensor / attention-logit partials), per-row maxima and logit partials combined through small LDS tables, then the
// tile is written as H2 (a wave stores 64 rows of one slot column = one contiguous KiB per plane).  E_DKB keeps the
// row-major fp32 epilogue of macx_gemm.hip.h (its output is the caller's fp32 gradient).
#pragma once
#include "macx_gemm.hip.h"
#include "macx_h2.hip.h"

namespace macx {

struct GemmH2P {
  int B, N, K, Nout;
  H2View A;                 // [B*N][K]
  const char* Wh;           // B_PLAIN: fp16 planes in slot order [K/32][2][4][Nout] x 16 B (pack format 3)
  const int* w_exp;         // B_PLAIN: device int, exponent of the packed matrix
  const float* Wt;          // B_YMIX_*: fp32 k-major tiles [K/32][Nout][32] (pack format 2) of W1a / W1a^T
  const float* Wt2;         // ... of W1b / W1b^T
  const float* w_max;       // B_YMIX_*: device floats {max |W1a|, max |W1b|}
  const float* y; int ldy;  // per-question vector mixed into the weight tile
  H2View out;               // every epilogue but E_DKB
  float* out_f32; int ldo;  // E_DKB: caller's fp32 [B*N][ldo]
  const float* bias; int act;
  H2View aux;               // E_MUL_DACT: the activation OUTPUT whose derivative multiplies the result (H1)
  const float* dr; int ld_dr;   // E_DKB: dinfo [B][ld_dr]
  const float* cvec;        // E_I2_LOGIT: control [B][Nout]
  const float* wvec;        // E_I2_LOGIT: logits weight [Nout]
  const float* att;         // E_DKB: kb attention [B][N]
  float* logit_part;        // E_I2_LOGIT: [Nout/128][B*N]
  float* colsum_part;       // COLSUM: column sums of the result per workgroup row block [B*nrb][Nout]
  const uint8_t* e_bytes;   // E_I2_LOGIT: keep bits of act(I2*c), one byte per slot, [Nout/8][Rp]; null = keep all
  const uint32_t* e_bits;   // E_DKB: keep bits of the knowledge base, row-major [B*N][ldo/32]; null = keep all
  float e_inv_keep;
  int accumulate;           // E_DKB: add to the output instead of writing it (one launch per step, steps after the first)
  // E_DKB runs ALL steps of the backward pass in one launch: the K loop walks `nsteps` A tensors (the kept dX_i, `a_step_bytes`
  // apart) against the same weight, the step's keep bits multiply each block sum as it is folded, and the epilogue adds
  // sum_i att_i (x) dinfo_i -- the caller's fp32 gradient is written once instead of read-modify-written every step
  int nsteps; size_t a_step_bytes;
  size_t bits_step_words;   // uint32 words between the keep bits of consecutive steps
  size_t att_step, dr_step; // floats between att_i / dinfo_i of consecutive steps
  int a_row_exp;            // E_DKB: every A tensor has ONE exponent per row (chain_bwd_kernel wrote it): the UNI kernel may fold per step
  int dbg;                  // measurement knobs (macx_opts.tune[MACX_TUNE_PHASE_MASK]): 1 skip the epilogue, 32 skip the in-loop staging,
                            // 64 skip the fragment reads + MFMAs, 256 return at once, 512 return in front of the K loop
};

template <int RT>
constexpr int kb_gemm_h2_lds_bytes() {
  constexpr int ROWS = RT * 16;
  constexpr int stage = 2 * 4 * ROWS * 16 + 2 * 4 * 128 * 16;
  constexpr int ring = 3 * stage + 2 * ROWS * 32 + 4096 + 64 + 256 * 12 + 2 * 256 * 16;   // ... + E_DKB: raw exponents and keep bits of the next step   // + the fold factors of the tile's rows + the question's mixing vector + a reduction scratch
  constexpr int epi = ROWS * 132 * 4 + 2 * 16 * ROWS * 4 + ROWS * 4 + 16 * 32 * 16 + 3 * 128 * 4;   // (E_DKB: tile + 16 x (ROWS + 128) floats, smaller)
  return ring > epi ? ring : epi;
}

// UNI (E_DKB, K = 512 only; round 5): the A tensors carry ONE exponent per row -- what the chain kernels write (macx_chain_h2.hip.h)
// -- so the four 128-wide K blocks of a step share their fold factor: the block accumulator runs through the whole step and is
// folded (keep bits, row factor) ONCE per step instead of four times.  The phase knobs priced the loop's skeleton -- folds,
// barriers, epilogue -- at 135 of the merged dKB launch's 335 us (profiles/r05_phase_knobs.txt); 36 of its 48 folds go away.
template <int RT, int BP, int EP, bool COLSUM, bool UNI = false>
__global__ __launch_bounds__(512) void kb_gemm_h2_kernel(GemmH2P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  if (p.dbg & 256) return;
  constexpr int G_THREADS = 512;
  constexpr int G_BN = 128;
  constexpr int G_LDT = G_BN + 4;
  constexpr int ROWS = RT * 16;
  constexpr int A_GS = ROWS * 16;                    // bytes between k-groups of an A plane
  constexpr int A_PLANE = 4 * A_GS;
  constexpr int B_GS = G_BN * 16;
  constexpr int B_PLANE = 4 * B_GS;
  constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
  constexpr int A_SLOTS = 2 * 4 * ROWS;
  constexpr int NA = A_SLOTS / 64;                   // DMA instructions per slice for A (one KiB each); wave w issues w, w + 8, ...
  constexpr int A_IT = (NA + 7) / 8;
  constexpr int HT = (RT + 1) / 2;

  const int nblk = gridDim.x;
  int v = blockIdx.x;
  if ((nblk & 7) == 0) v = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int ncb = p.Nout / G_BN;
  const int nrb = (p.N + ROWS - 1) / ROWS;
  const int cb = v % ncb;
  const int rbi = (v / ncb) % nrb;
  const int b = v / (ncb * nrb);
  const int ntiles = (p.N + 15) >> 4;
  const int tbase = ntiles / nrb, textra = ntiles - tbase * nrb;
  const int nt = tbase + (rbi < textra ? 1 : 0);
  const int row0 = (rbi * tbase + min(rbi, textra)) << 4;
  const int row_end = min(p.N, row0 + (nt << 4));
  const int nvalid = row_end - row0;
  const size_t grow0 = (size_t)b * p.N + row0;       // first global row of the tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2;
  const int cgp = wave & 3;
  const int t0 = half * HT;
  const int my_nt = max(0, min(nt - t0, half ? RT - HT : HT));
  const int nk = p.K >> 5;
  const int nkb = p.K >> 7;                          // 128-wide K blocks = exponent blocks of A

  // ---- LDS behind the ring: fold factors, the question's mixing vector, a reduction scratch.  They are filled AFTER the first
  //      two K slices have been requested (below): their global loads then overlap the DMA instead of preceding it
  float* sF = reinterpret_cast<float*>(lds + 3 * STAGE);           // [2 (E_DKB: step parity)][K / 128][ROWS]: 2^-(eA[row][kb] + eB)
  float* sY = reinterpret_cast<float*>(lds + 3 * STAGE + 2 * ROWS * 32);  // B_YMIX_ROW: y_b[K]
  int eB = 0;
  float sB = 1.f;
  // E_DKB: raw exponents / keep bits of the step after the current one (filled by DMA, see issue_tables below)
  constexpr int X_OFF = 3 * STAGE + 2 * ROWS * 32 + 4096 + 64;
  uint32_t* sRaw = reinterpret_cast<uint32_t*>(lds + X_OFF);              // [3][256] dwords
  uint32_t* sM = reinterpret_cast<uint32_t*>(lds + X_OFF + 256 * 12);     // [2 (step parity)][256 rows][4 words]
  const bool fast_tables = EP == E_DKB && nk >= 12;

  f32x4 acc[HT][2], tot[HT][2];
#pragma unroll
  for (int t = 0; t < HT; ++t) acc[t][0] = acc[t][1] = tot[t][0] = tot[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging: slot f of a stage sits at byte 16 f; DMA instruction u covers slots 64 u .. 64 u + 63
  const size_t Rp = p.A.Rp();
  const size_t a_kstep = 4 * Rp * 16;                // bytes per K slice in a plane
  const char* a_base = p.A.base;
  uint32_t a_off[A_IT];                              // byte offset of this lane's slot in slice 0 (both planes < 4 GB)
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int f = min(tid + G_THREADS * i, A_SLOTS - 1);
    const int q = f / ROWS, lrow = f - q * ROWS;     // q = 4 plane + k-group
    a_off[i] = (uint32_t)((q >> 2) * p.A.plane_bytes() + ((size_t)(q & 3) * Rp + grow0 + lrow) * 16);
  }
  const int my_na = wave < NA ? (NA - wave + 7) >> 3 : 0;          // this wave's A instructions per slice
  const int my_n = my_na + (BP == B_PLAIN ? 2 : 4);                // ... plus its share of B (DMA) / its 4 weight loads
  f32x4 rwa[2][2], rwb[2][2];                                      // B_YMIX: raw W1a / W1b of a slice, two register sets
  const int yg = lane >> 4;                                        // B_YMIX: this thread's k-group and column
  const int ycolm = (lane & 15) + 16 * (tid >> 6);
  float ycol = 0.f;
  if (BP == B_YMIX_COL) ycol = p.y[(size_t)b * p.ldy + cb * G_BN + ycolm];

  // B_YMIX: the raw fp32 weights of a slice -> register set S.  Inline assembly like the DMA and for the same reason: a
  // load the compiler knows about is waited for with a vmcnt that ignores the DMA instructions queued behind it, i.e. with
  // far too small a count -- it would drain the slice that was just requested.  The registers are handed back to the
  // compiler by settle_w(), behind an explicit wait that counts everything.
  auto load_w = [&](auto set_c, int kt) {
    constexpr int S = decltype(set_c)::value;
    const size_t off = ((size_t)kt * p.Nout + cb * G_BN + ycolm) * 32 + yg * 8;
    const float* pa = p.Wt + off;
    const float* pb = p.Wt2 + off;
    f32x4 a0, a1, b0, b1;                                          // (plain names: an asm operand cannot name a captured array)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a0) : "v"(pa) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(a1) : "v"(pa) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b0) : "v"(pb) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(b1) : "v"(pb) : "memory");
    rwa[S][0] = a0; rwa[S][1] = a1; rwb[S][0] = b0; rwb[S][1] = b1;
  };
  auto settle_w = [&](auto set_c, int n) {                         // at most n younger VMEM instructions may still fly
    constexpr int S = decltype(set_c)::value;
    wait_vmcnt_n(n);
    f32x4 a0 = rwa[S][0], a1 = rwa[S][1], b0 = rwb[S][0], b1 = rwb[S][1];
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
    rwa[S][0] = a0; rwa[S][1] = a1; rwb[S][0] = b0; rwb[S][1] = b1;
  };
  const uint32_t lds0 = lds_addr_of(lds);
  auto issue = [&](int g) {                                        // DMA of (global) slice g into ring stage g % 3
    const uint32_t dA = lds0 + (uint32_t)(g % 3) * STAGE;
    int kt = g;
    const char* ab = a_base;
    if (EP == E_DKB) {                                             // slice g = step g / nk, K slice g % nk
      const int stp = g / nk;
      kt = g - stp * nk;
      ab = a_base + (size_t)stp * p.a_step_bytes;
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (wave + 8 * i < NA) dma16b(ab + (size_t)kt * a_kstep + a_off[i], dA + (wave + 8 * i) * 1024);
    if (BP == B_PLAIN) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int f = tid + G_THREADS * i;                         // plane i, k-group (f >> 7) & 3, column f & 127
        dma16b(p.Wh + ((((size_t)kt * 2 + i) * 4 + ((f >> 7) & 3)) * p.Nout + cb * G_BN + (f & 127)) * 16,
               dA + 2 * A_PLANE + (wave + 8 * i) * 1024);
      }
    }
  };
  auto store_w = [&](auto set_c, int kt) {                         // B_YMIX: mix, scale, split -> ring stage kt % 3
    constexpr int S = decltype(set_c)::value;
    char* dB = lds + (kt % 3) * STAGE + 2 * A_PLANE;
    float x[8];
    f32x4 ry[2];
    if (BP == B_YMIX_ROW) {
      ry[0] = *reinterpret_cast<const f32x4*>(sY + (kt << 5) + yg * 8);
      ry[1] = *reinterpret_cast<const f32x4*>(sY + (kt << 5) + yg * 8 + 4);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // ROW: B_eff[k][j] = y[b][k] W1a[k][j] + W1b[k][j]     (ops.py:703,718 folded into the weights)
      // COL: B_eff[k][j] = y[b][j] W1a^T[k][j] + W1b^T[k][j] (backward-data of the same product)
      const float yy = (BP == B_YMIX_ROW) ? ry[e >> 2][e & 3] : ycol;
      x[e] = fmaf(rwa[S][e >> 2][e & 3], yy, rwb[S][e >> 2][e & 3]) * sB;
    }
    u32x4 hi, lo;
    h2_split8(x, hi, lo);
    char* d = dB + yg * B_GS + ycolm * 16;
    *reinterpret_cast<u32x4*>(d) = hi;
    *reinterpret_cast<u32x4*>(d + B_PLANE) = lo;
  };

  // lane (i = lane & 15, g = lane >> 4) holds k = 8g .. 8g+7 of row / column i for both operands
  // first: the slice opens a 128-wide K block -- its first product per tile starts from zero (C = 0 costs nothing on the
  // MFMA) instead of from an accumulator cleared by the fold
  auto compute = [&](int buf, auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const char* sA = lds + buf * STAGE + (lane >> 4) * A_GS + (t0 * 16 + (lane & 15)) * 16;
    const char* sB_ = lds + buf * STAGE + 2 * A_PLANE + (lane >> 4) * B_GS + (cgp * 32 + (lane & 15)) * 16;
    u32x4 bf[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int c = 0; c < 2; ++c) bf[pl][c] = *reinterpret_cast<const u32x4*>(sB_ + pl * B_PLANE + c * 256);
    // smallest terms first: A_lo x B_hi ; A_hi x {B_lo, B_hi}
    // row tiles in groups of TG so that only TG A fragments are live at a time (the kernel runs at the 256-register cap)
    constexpr int TG = BP != B_PLAIN ? 1 : 4;        // (the y-mixing kernels also hold two sets of raw weights)
#pragma unroll
    for (int ap = 1; ap >= 0; --ap) {
#pragma unroll
      for (int tb = 0; tb < HT; tb += TG) {
        u32x4 af[TG];
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          const int t = tb + u;
          if (t < HT && (t < HT - 1 || t < my_nt)) af[u] = *reinterpret_cast<const u32x4*>(sA + ap * A_PLANE + t * 256);
        }
#pragma unroll
        for (int bp = 1 - ap; bp >= 0; --bp) {
#pragma unroll
          for (int u = 0; u < TG; ++u) {
            const int t = tb + u;
            if (t < HT && (t < HT - 1 || t < my_nt)) {
              acc[t][0] = mfma_f16(af[u], bf[bp][0], (FIRST && ap == 1) ? zero : acc[t][0]);
              acc[t][1] = mfma_f16(af[u], bf[bp][1], (FIRST && ap == 1) ? zero : acc[t][1]);
            }
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 3");       // 20 wait states: more than any MFMA-result -> VALU-read distance
    __builtin_amdgcn_sched_barrier(0);
  };
  // end of a 128-wide K block: running sum += block sum * 2^-(eA[row][kb] + eB)
  auto fold = [&](int blk) {                                       // blk = global 128-wide block = step * nkb + kb
    int kb = blk, stp = 0;
    if (EP == E_DKB) { stp = blk / nkb; kb = blk - stp * nkb; }
    const int toff = (stp & 1) * (8 * ROWS) + kb * ROWS;
    const bool masked = EP == E_DKB && p.e_bits != nullptr;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      f32x4 f = *reinterpret_cast<const f32x4*>(sF + toff + min((t0 + t) * 16, ROWS - 16) + (lane >> 4) * 4);
      if (masked) {
        // the step's keep bits of the knowledge base (ops.py:678): row-major words, this lane's two columns are bits
        // (lane & 15) and 16 + (lane & 15) of word `cgp` of the row's 128-column block
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int lrow = min((t0 + t) * 16 + (lane >> 4) * 4 + e, nvalid - 1);
          const uint32_t w = sM[((stp & 1) * 256 + lrow) * 4 + cgp];
          const float fe = f[e] * p.e_inv_keep;
          tot[t][0][e] = fmaf(acc[t][0][e], ((w >> (lane & 15)) & 1u) ? fe : 0.f, tot[t][0][e]);
          tot[t][1][e] = fmaf(acc[t][1][e], ((w >> (16 + (lane & 15))) & 1u) ? fe : 0.f, tot[t][1][e]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) tot[t][c] += acc[t][c] * f;
      }
    }
  };
  using FirstT = std::integral_constant<bool, true>;
  using FirstF = std::integral_constant<bool, false>;
  // The K loop runs in groups of the four slices of one 128-wide block, written out: the zero-C variant of the first slice
  // and the accumulating variant of the other three must NOT meet in a control-flow join.  (At a join the accumulators pass
  // through copies -- v_mov reads of MFMA results, which need software wait states on this hardware -- and the compiler's
  // hazard pass was seen to leave them out there: rows 4g, 4g+1 of every tile came out stale.)  compute() also ends in
  // enough wait states for any vector-ALU read of its results, whatever follows it.

  const bool do_stage = !(p.dbg & 32), do_compute = !(p.dbg & 64);
  // the first two K slices are requested before anything else is loaded
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (BP == B_PLAIN) {
    issue(0);
    if (nk > 1) issue(1);
  } else {
    load_w(S0{}, 0);
    issue(0);
    if (nk > 1) { load_w(S1{}, 1); issue(1); }
  }
  // ---- weight exponent; fold factors of this tile's rows
  if (BP == B_PLAIN) {
    eB = *p.w_exp;
  } else {
    if (BP == B_YMIX_ROW)
      for (int k = tid; k < p.K; k += G_THREADS) sY[k] = p.y[(size_t)b * p.ldy + k];
    // |y W1a + W1b| <= max|y_b| max|W1a| + max|W1b|: a power of two above the bound costs at most the low end of the range
    float* red = reinterpret_cast<float*>(lds + 3 * STAGE + 2 * ROWS * 32 + 4096);
    float m = 0.f;
    for (int k = tid; k < p.K; k += G_THREADS) m = fmaxf(m, fabsf(p.y[(size_t)b * p.ldy + k]));
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
    eB = h2_weight_exponent(fmaf(m, p.w_max[0], p.w_max[1]));
  }
  sB = h2_pow2(eB);
  const int nsteps = (EP == E_DKB) ? p.nsteps : 1;
  auto fill_sF = [&](int stp) {                                    // fold factors of step `stp` -> table stp & 1
    const int8_t* eA = reinterpret_cast<const int8_t*>(p.A.base + (size_t)stp * p.a_step_bytes + 2 * p.A.plane_bytes());
    const int acb = p.A.cb();
    const int toff = (stp & 1) * (8 * ROWS);
    for (int i = tid; i < ROWS * acb; i += G_THREADS) {
      const int r = i / acb, k = i - r * acb;                      // rows past the tensor end lie in the pad rows
      sF[toff + k * ROWS + r] = h2_unscale((int)eA[(grow0 + r) * acb + k], eB);
    }
  };
  fill_sF(0);
  // E_DKB, steps after the first: the next step's row exponents and keep bits arrive by DMA while the current step is being
  // multiplied (global loads the compiler knows about would drain the DMA queue at every fold): three dwords per row
  // covering the row's K / 128 exponent bytes wherever they start, and the 16 bytes of keep bits of the row's 128 columns
  const int xrow = min(wave * 64 + lane, nvalid - 1);                     // waves 0..3: the tile row this lane fetches for
  auto issue_tables = [&](int stp) {
    if (wave < 4) {
      const int acb = p.A.cb();
      const char* eA = p.A.base + (size_t)stp * p.a_step_bytes + 2 * p.A.plane_bytes();
      const size_t start = (grow0 + xrow) * (size_t)acb;
#pragma unroll
      for (int j = 0; j < 3; ++j) dma4b(eA + (start & ~(size_t)3) + 4 * j, lds0 + X_OFF + (j * 256 + wave * 64) * 4);
      if (p.e_bits)
        dma16b(reinterpret_cast<const char*>(p.e_bits + (size_t)stp * p.bits_step_words + (grow0 + xrow) * (size_t)(p.ldo >> 5) + cb * 4),
               lds0 + X_OFF + 256 * 12 + ((stp & 1) * 256 + wave * 64) * 16);
    }
  };
  auto convert_tables = [&](int stp) {                                    // raw exponent bytes -> fold factors (LDS to LDS)
    const int acb = p.A.cb();
    const int toff = (stp & 1) * (8 * ROWS);
    for (int i = tid; i < ROWS * acb; i += G_THREADS) {
      const int r = i / acb, k = i - r * acb;
      const int rr = min(r, nvalid - 1);
      const int sh = (int)(((grow0 + rr) * (size_t)acb) & 3) + k;         // byte offset inside the row's three dwords
      const uint32_t w = sRaw[(sh >> 2) * 256 + rr];
      const int e = (int)(int8_t)((w >> (8 * (sh & 3))) & 0xFFu);
      sF[toff + k * ROWS + r] = h2_unscale(e, eB);
    }
  };
  auto fill_sM = [&](int stp) {                                           // keep bits by ordinary loads (step 0; short K)
    if (p.e_bits)
      for (int i = tid; i < 256 * 4; i += G_THREADS) {
        const int r = min(i >> 2, nvalid - 1);
        sM[(stp & 1) * 1024 + i] = p.e_bits[(size_t)stp * p.bits_step_words + (grow0 + r) * (size_t)(p.ldo >> 5) + cb * 4 + (i & 3)];
      }
  };
  if (EP == E_DKB) fill_sM(0);
  if (BP == B_PLAIN) {
    if (nk > 1) wait_vmcnt_n(my_n); else wait_vmcnt<0>();
    __syncthreads();                   // slice 0 has landed, for every wave's share of it; the tables are complete
    if (p.dbg & 512) { wait_vmcnt<0>(); return; }
    const int nkt = nk * nsteps;                        // E_DKB: the K loop runs through all steps
    auto step = [&](auto first_c, int kt) {
      if (kt + 2 < nkt && do_stage) issue(kt + 2);      // ring stage (kt + 2) % 3 was last read in iteration kt - 1
      if (do_compute) compute(kt % 3, first_c);
      if ((kt & 3) == 3) fold(kt >> 2);
      if (kt + 2 < nkt) wait_vmcnt_n(my_n); else wait_vmcnt<0>();    // slice kt + 1 has landed; slice kt + 2 may fly
      __syncthreads();
    };
    if constexpr (UNI) {
      // nk == 16: one fold per step, after its last slice; the 16 slices are written out (the zero-C variant of slice 0 and the
      // accumulating variant of the other fifteen must not meet in a control-flow join, see below)
      auto step_u = [&](auto first_c, int kt, bool last) __attribute__((always_inline)) {
        if (kt + 2 < nkt && do_stage) issue(kt + 2);
        if (do_compute) compute(kt % 3, first_c);
        if (last) fold(kt >> 2);                          // (fold() takes a 128-wide block index: step = blk / nkb, table kb = 3 -- same factor as 0..2)
        if (kt + 2 < nkt) wait_vmcnt_n(my_n); else wait_vmcnt<0>();
        __syncthreads();
      };
      for (int kt = 0; kt < nkt; kt += 16) {
        const int stp = kt >> 4;
        step_u(FirstT{}, kt, false);
        step_u(FirstF{}, kt + 1, false);
        step_u(FirstF{}, kt + 2, false);
        step_u(FirstF{}, kt + 3, false);
        if (nsteps > 1 && stp + 1 < nsteps) issue_tables(stp + 1);       // older than the next slice DMA: landed by its wait
        step_u(FirstF{}, kt + 4, false);
        step_u(FirstF{}, kt + 5, false);
        step_u(FirstF{}, kt + 6, false);
        step_u(FirstF{}, kt + 7, false);
        if (nsteps > 1 && stp + 1 < nsteps) convert_tables(stp + 1);
        step_u(FirstF{}, kt + 8, false);
        step_u(FirstF{}, kt + 9, false);
        step_u(FirstF{}, kt + 10, false);
        step_u(FirstF{}, kt + 11, false);
        step_u(FirstF{}, kt + 12, false);
        step_u(FirstF{}, kt + 13, false);
        step_u(FirstF{}, kt + 14, false);
        step_u(FirstF{}, kt + 15, true);
      }
    } else
    for (int kt = 0; kt < nkt; kt += 4) {
      if (EP == E_DKB && nsteps > 1) {
        // half-way through a step the fold factors of the next one are built (the other table; its global loads make
        // the compiler drain the DMA queue -- once per step)
        const int stp = kt / nk, loc = kt - stp * nk;
        if (fast_tables) {
          if (loc == 4 && stp + 1 < nsteps) issue_tables(stp + 1);     // older than this iteration's slice DMA: landed by its wait
          if (loc == 8 && stp + 1 < nsteps) convert_tables(stp + 1);
        } else if (loc == (nk >> 1 & ~3) && stp + 1 < nsteps) {
          fill_sF(stp + 1);
          fill_sM(stp + 1);
        }
      }
      step(FirstT{}, kt);
      step(FirstF{}, kt + 1);
      step(FirstF{}, kt + 2);
      step(FirstF{}, kt + 3);
    }
  } else {
    // the mixed weights pass through registers, two slices ahead like the DMA (two register sets, the loop runs in pairs;
    // K / 32 is a multiple of 4).  Iteration kt: request slice kt + 2 (weights, then A rows); wait until the weights of slice
    // kt + 1 are in (everything younger -- A rows of kt + 1, all of kt + 2 -- may still fly); mix / split / store them as
    // ordinary code next to the MFMAs of slice kt, so the scheduler can interleave the two; wait for the A rows of kt + 1
    if (nk > 1) settle_w(S0{}, my_n); else settle_w(S0{}, 0);
    store_w(S0{}, 0);
    __syncthreads();
    auto iter = [&](auto cur_c, auto nxt_c, auto first_c, int kt) {   // cur: set of slice kt (and kt + 2), nxt: of slice kt + 1
      const bool more2 = kt + 2 < nk, more1 = kt + 1 < nk;
      if (more2 && do_stage) { load_w(cur_c, kt + 2); issue(kt + 2); }
      if (more1) {
        settle_w(nxt_c, more2 ? my_n + my_na : my_na);
        if (do_stage) store_w(nxt_c, kt + 1);                       // ring stage (kt + 1) % 3 was last read in iteration kt - 2
      }
      if (do_compute) compute(kt % 3, first_c);
      if ((kt & 3) == 3) fold(kt >> 2);
      if (more2) wait_vmcnt_n(my_n); else wait_vmcnt<0>();
      __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 4) {
      iter(S0{}, S1{}, FirstT{}, kt);
      iter(S1{}, S0{}, FirstF{}, kt + 1);
      iter(S0{}, S1{}, FirstF{}, kt + 2);
      iter(S1{}, S0{}, FirstF{}, kt + 3);
    }
  }
  if (p.dbg & 1) {
    if (tot[0][0][0] == 123.456f) p.out.exps()[0] = 1;
    return;
  }
  (void)nkb;

  // ---- epilogue, step 1: accumulators -> row-major LDS tile (16x16 map: col = lane & 15, row = (lane >> 4) * 4 + reg)
  float* T = smem;
  // the per-column constants of the row pass (bias | control vector | logits weight of this 128-column block) go to LDS
  // once: in the row pass all 64 lanes of a wave work on the same slot column, so these become broadcast LDS reads instead
  // of six dependent global loads per item
  float* sC = smem + ROWS * G_LDT + 2 * 16 * ROWS + ROWS + 16 * 32 * 4;      // [3][128], behind T | Mx | Px | rexp | red
  if (EP != E_DKB && tid < 96) {
    const int which = tid >> 5, c4 = (tid & 31) * 4;
    const float* src = which == 0 ? p.bias : (which == 1 ? p.cvec + (size_t)b * p.Nout : p.wvec);
    const bool have = which == 0 ? (EP == E_BIAS_ACT || EP == E_I2_LOGIT) : (EP == E_I2_LOGIT);
    if (have) *reinterpret_cast<f32x4*>(sC + which * 128 + c4) = *reinterpret_cast<const f32x4*>(src + cb * G_BN + c4);
  }
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    if (t0 + t < RT) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          T[((t0 + t) * 16 + (lane >> 4) * 4 + e) * G_LDT + cgp * 32 + c * 16 + (lane & 15)] = tot[t][c][e];
    }
  }
  __syncthreads();
  if (p.dbg & 128) return;                         // timing experiment: stop after the accumulator tile is in LDS

  if (EP == E_DKB) {
    // dKB = sum_i (dX_i Wx^T) * kbmask_i  [in T: the masks were applied block by block]  +  sum_i att_i (x) dinfo_i.
    // att_i of this tile's rows and dinfo_i of its 128 columns go to LDS 16 steps at a time; one float4 per lane,
    // a row's 128 columns in 32 consecutive lanes (512-byte row segments)
    constexpr int SCH = 16;
    float* sAtt = smem + ROWS * G_LDT;                 // [SCH][ROWS]
    float* sDr = sAtt + SCH * ROWS;                    // [SCH][128]
    constexpr int NIT = (ROWS * 32 + G_THREADS - 1) / G_THREADS;
    f32x4 o[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = tid + G_THREADS * i, lrow = it >> 5, c4 = (it & 31) * 4;
      o[i] = lrow < nvalid ? *reinterpret_cast<const f32x4*>(T + lrow * G_LDT + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int s0 = 0; s0 < nsteps; s0 += SCH) {
      const int ns = min(SCH, nsteps - s0);
      __syncthreads();
      for (int i = tid; i < ns * ROWS; i += G_THREADS) {
        const int si = i / ROWS, r = i - si * ROWS;
        sAtt[i] = r < nvalid ? p.att[(size_t)(s0 + si) * p.att_step + grow0 + r] : 0.f;
      }
      for (int i = tid; i < ns * 128; i += G_THREADS) {
        const int si = i >> 7, c = i & 127;
        sDr[i] = p.dr[(size_t)(s0 + si) * p.dr_step + (size_t)b * p.ld_dr + cb * G_BN + c];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int it = tid + G_THREADS * i, lrow = min(it >> 5, ROWS - 1), c4 = (it & 31) * 4;
        for (int si = 0; si < ns; ++si) o[i] += sAtt[si * ROWS + lrow] * *reinterpret_cast<const f32x4*>(sDr + si * 128 + c4);
      }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int it = tid + G_THREADS * i, lrow = it >> 5, c4 = (it & 31) * 4;
      if (lrow < nvalid) {
        f32x4* dst = reinterpret_cast<f32x4*>(p.out_f32 + (grow0 + lrow) * (size_t)p.ldo + cb * G_BN + c4);
        *dst = p.accumulate ? *dst + o[i] : o[i];
      }
    }
    return;
  }

  // ---- step 2: lanes along rows.  item it = tid + 512 i  ->  slot column kgl = it / RP, local row it % RP
  constexpr int RP = ROWS;                        // (not rounded to whole waves: at ROWS = 208 that would idle a fifth of the lanes)
  constexpr int ITEMS = (16 * RP + G_THREADS - 1) / G_THREADS;
  float* Mx = smem + ROWS * G_LDT;                 // [16][ROWS] partial row maxima
  float* Px = Mx + 16 * ROWS;                      // [16][ROWS] partial attention logits
  int* rexp = reinterpret_cast<int*>(Px + 16 * ROWS);       // [ROWS]
  f32x4* red = reinterpret_cast<f32x4*>(rexp + ROWS);       // [16][32] column partials
  float val[ITEMS][8];
  const size_t oRp = p.out.Rp();
  // The activation is a run-time option but must not be a per-value branch: with `switch (p.act)` inside, the 64 values of a
  // thread unrolled into ~9400 instructions of branch chains (tanh / sigmoid / elu / relu bodies per value) -- more code than
  // the instruction cache holds, a taken branch every few instructions -- and this pass took 13 of the kernel's 41 us.
  // The pass is instantiated once per activation and selected by ONE switch.
  auto row_pass = [&](auto act_c) {
  constexpr int ACT = decltype(act_c)::value;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int it = tid + G_THREADS * i;
    const int kgl = it / RP, lrow = it - kgl * RP;
    if (kgl < 16 && lrow < nvalid) {
      const int col = cb * G_BN + kgl * 8;
      float* tp = T + lrow * G_LDT + kgl * 8;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(tp), a1 = *reinterpret_cast<const f32x4*>(tp + 4);
      float x[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[e] = a0[e]; x[4 + e] = a1[e]; }
      if (EP == E_BIAS_ACT || EP == E_I2_LOGIT) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sC + kgl * 8), b1 = *reinterpret_cast<const f32x4*>(sC + kgl * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] += b0[e]; x[4 + e] += b1[e]; }
      }
      if (EP == E_BIAS_ACT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = act_apply(ACT, x[e]);
      }
      if (EP == E_I2_LOGIT) {
        // mac_cell.py:248,262,266: act(I2 * c) -> dropout -> . w   (the bias b_k is added in kb_attend)
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(sC + 128 + kgl * 8), c1 = *reinterpret_cast<const f32x4*>(sC + 128 + kgl * 8 + 4);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sC + 256 + kgl * 8), w1 = *reinterpret_cast<const f32x4*>(sC + 256 + kgl * 8 + 4);
        const uint32_t bits = p.e_bytes ? p.e_bytes[(size_t)(cb * 16 + kgl) * oRp + grow0 + lrow] : 0xFFu;
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float g = act_apply(ACT, x[e] * (e < 4 ? c0[e & 3] : c1[e & 3]));
          g = ((bits >> e) & 1u) ? g * p.e_inv_keep : 0.f;
          part = fmaf(g, e < 4 ? w0[e & 3] : w1[e & 3], part);
        }
        Px[kgl * ROWS + lrow] = part;
      }
      if (EP == E_MUL_DACT) {
        const size_t arp = p.aux.Rp();
        const char* ap = p.aux.plane(0) + ((size_t)(cb * 16 + kgl) * arp + grow0 + lrow) * 16;
        const u32x4 hh = *reinterpret_cast<const u32x4*>(ap), hl = *reinterpret_cast<const u32x4*>(ap + p.aux.plane_bytes());
        float h[8];
        h2_join8(hh, hl, h2_pow2(-(int)p.aux.exps()[(grow0 + lrow) * p.aux.cb() + cb]), h);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] *= act_grad_from_out(ACT, h[e]);
      }
      float m = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { val[i][e] = x[e]; m = fmaxf(m, fabsf(x[e])); }
      Mx[kgl * ROWS + lrow] = m;
      if (COLSUM) {
        *reinterpret_cast<f32x4*>(tp) = f32x4{x[0], x[1], x[2], x[3]};
        *reinterpret_cast<f32x4*>(tp + 4) = f32x4{x[4], x[5], x[6], x[7]};
      }
    }
  }
  };
  switch (p.act) {
    case ACT_TANH: row_pass(std::integral_constant<int, ACT_TANH>{}); break;
    case ACT_SIGMOID: row_pass(std::integral_constant<int, ACT_SIGMOID>{}); break;
    case ACT_ELU: row_pass(std::integral_constant<int, ACT_ELU>{}); break;
    case ACT_RELU: row_pass(std::integral_constant<int, ACT_RELU>{}); break;
    default: row_pass(std::integral_constant<int, ACT_NON>{}); break;
  }
  __syncthreads();
  if (p.dbg & 2) return;                           // timing experiment: stop after the row pass
  if (tid < nvalid) {
    float m = Mx[tid];
#pragma unroll
    for (int k = 1; k < 16; ++k) m = fmaxf(m, Mx[k * ROWS + tid]);
    const int e = h2_exponent(m);
    rexp[tid] = e;
    p.out.exps()[(grow0 + tid) * p.out.cb() + cb] = (int8_t)e;
    if (EP == E_I2_LOGIT) {
      float s = Px[tid];
#pragma unroll
      for (int k = 1; k < 16; ++k) s += Px[k * ROWS + tid];          // fixed order
      p.logit_part[(size_t)cb * p.B * p.N + grow0 + tid] = s;
    }
  }
  if (COLSUM) {
    const int c4 = tid & 31, rg = tid >> 5;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    for (int lrow = rg; lrow < nvalid; lrow += 16) cs += *reinterpret_cast<const f32x4*>(T + lrow * G_LDT + c4 * 4);
    red[rg * 32 + c4] = cs;
  }
  __syncthreads();
  if (COLSUM && tid < 32) {
    f32x4 t = red[tid];
#pragma unroll
    for (int g = 1; g < 16; ++g) t += red[g * 32 + tid];
    *reinterpret_cast<f32x4*>(p.colsum_part + (size_t)(b * nrb + rbi) * p.Nout + cb * G_BN + tid * 4) = t;
  }
  char* o0 = p.out.plane(0);
  const size_t opb = p.out.plane_bytes();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int it = tid + G_THREADS * i;
    const int kgl = it / RP, lrow = it - kgl * RP;
    if (kgl < 16 && lrow < nvalid) {
      const float s = h2_pow2(rexp[lrow]);
      float xs[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) xs[e] = val[i][e] * s;
      u32x4 hi, lo;
      h2_split8(xs, hi, lo);
      char* d = o0 + ((size_t)(cb * 16 + kgl) * oRp + grow0 + lrow) * 16;
      if (p.dbg & 8) continue;                       // timing experiment: no output stores
      if (p.dbg & 16) {                       // timing experiment: non-temporal stores
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(d), "v"(hi) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(d + opb), "v"(lo) : "memory");
      } else {
        *reinterpret_cast<u32x4*>(d) = hi;
        *reinterpret_cast<u32x4*>(d + opb) = lo;
      }
    }
  }
}

inline int dkb_uni_mode() { return tune_get(MACX_TUNE_DKB_UNI, 1); }        // MACX_TUNE_DKB_UNI = 0 | 1: one fold per step in the merged dKB launch
template <int RT, int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm_h2_launch_rt(const GemmH2P& p, hipStream_t st) {
  if constexpr (EP == E_DKB && RT == 13 && BP == B_PLAIN && !COLSUM) {
    if (p.a_row_exp && p.K == 512 && p.nsteps >= 1 && dkb_uni_mode()) {
      auto kern = kb_gemm_h2_kernel<RT, BP, EP, COLSUM, true>;
      constexpr size_t lds = (size_t)kb_gemm_h2_lds_bytes<RT>();
      hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);
      if (e != hipSuccess) return e;
      const int ncb = p.Nout / 128;
      const int nrb = (p.N + RT * 16 - 1) / (RT * 16);
      hipLaunchKernelGGL(kern, dim3(p.B * nrb * ncb), dim3(512), lds, st, p);
      return hipGetLastError();
    }
  }
  auto kern = kb_gemm_h2_kernel<RT, BP, EP, COLSUM>;
  constexpr size_t lds = (size_t)kb_gemm_h2_lds_bytes<RT>();
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  const int ncb = p.Nout / 128;
  const int nrb = (p.N + RT * 16 - 1) / (RT * 16);
  hipLaunchKernelGGL(kern, dim3(p.B * nrb * ncb), dim3(512), lds, st, p);
  return hipGetLastError();
}

template <int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm_h2_launch(const GemmH2P& p, hipStream_t st) {
  switch (kb_gemm_pick_rt(p.N, p.B, p.Nout / 128)) {
    case 1: return kb_gemm_h2_launch_rt<1, BP, EP, COLSUM>(p, st);
    case 2: return kb_gemm_h2_launch_rt<2, BP, EP, COLSUM>(p, st);
    case 4: return kb_gemm_h2_launch_rt<4, BP, EP, COLSUM>(p, st);
    case 7: return kb_gemm_h2_launch_rt<7, BP, EP, COLSUM>(p, st);
    default: return kb_gemm_h2_launch_rt<13, BP, EP, COLSUM>(p, st);
  }
}

}  // namespace macx
