// macx_chain_h2.hip.h -- the read unit's knowledge-base products chained inside ONE kernel per direction.
//
// forward (mac_cell.py:230-266, ops.py:668-725), chain_fwd_kernel:
//     KBd = dropout(KB)                         ops.py:678
//     X   = KBd Wx + bx                         ops.py:688            (stage 1)
//     H1  = act([X*y, X] W1 + b1)               ops.py:703,718        (stage 2: X W1b, then (X*y) W1a on the same accumulators)
//     I2  = H1 W2 + b2                          ops.py:326            (stage 3)
//     l   = dropout(act(I2 * c)) . w_k          mac_cell.py:248-266
// backward (SURVEY appendix A rows "logit" .. "mem-mul"), chain_bwd_kernel:
//     dI2 = (dl w_k * mask * act'(I2 * c)) * c                        (stage B0, elementwise from the kept I2)
//     dI1 = (dI2 W2^T) * act'(H1)                                     (stage B1)
//     dX  = (dI1 W1a^T) * y + dI1 W1b^T                               (stage B2: two products on the same accumulators)
//     dy  = sum_rows (dI1 W1a^T) * X                                  (stage B2, between its two products: per-tile partials)
//
// A workgroup owns 64 (for small row counts 32 or 16: ChainGeo) consecutive rows of the [B*N, d] activation and ALL d
// columns of every stage, so a stage's output
// never leaves the CU before it is the next stage's operand: it is written in place, as H2 planes, over the LDS tile the
// stage just multiplied.  Every tensor the other direction or a weight-gradient kernel needs still goes to HBM once, as
// stores that drain behind the next stage's matrix work; nothing is read back.  Against the launches these replace
// (h2_from_f32 + three kb_gemm_h2 forward, read_att_bwd_h2 + two kb_gemm_h2 backward) a step saves the HBM round trips of
// the intermediates, the launch ramps and the store tails.
//
// How the pieces map onto a CU (8 waves, one workgroup per CU at d = 512: 128 KB of LDS):
//   * rows are taken from the flat [B*N] axis, not per question: the question enters a row only through y, c and the dropout
//     index, all looked up per row -- tiles never pad a question (B*N = 12544 = 196 x 64).
//   * a wave owns a block of 16-column tiles for its rows (d = 512: all 64 rows x 64 columns per wave).  Weight fragments go
//     from L2 straight to registers in MFMA operand order (pack format 3 stores a lane's 8 k-values of one column as one
//     16-byte slot, 16 columns = 256 contiguous bytes), one K slice ahead of use, and never touch LDS.  The activation tile
//     is shared by all waves and LDS-resident for the whole stage: the K loop has NO barrier.
//   * v_mfma_f32_16x16x32_f16 with the operands swapped (D^T = W^T A^T): the weight slot is the MFMA's A operand and the
//     activation slot its B operand, so lane (i, g) ends up with FOUR CONSECUTIVE COLUMNS 4g .. 4g + 3 of row i of the tile
//     instead of four rows of one column.  Bias, activation, row maxima, the logit dot product and the hi/lo split run on
//     the accumulators where they lie, and half a slot (4 columns x fp16) leaves as one 8-byte store: no transpose through
//     LDS, no row pass.  (Measured: the same loop on v_mfma_f32_32x32x16_f16 -- half the instructions, 4 accumulator tiles
//     per wave instead of 16 -- ran 18.5 us against 16.5 us per product with no loads in the loop, 25 against 17 with them.)
//   * one exponent per ROW (all d columns) instead of per (row, 128 columns): the workgroup sees the whole row, so a stage needs
//     no per-K-block fold and no second accumulator set; the same exponent is written to each of the row's d/128 exponent
//     bytes, which keeps the H2 tensors readable by every other kernel (macx_h2.hip.h).
//   * y enters on the activation side.  Forward: stage 2 first accumulates X W1b, then the tile is rewritten in place as
//     split(X * y_q) (fp32 product, rounded once like the reference's X*y, own row exponents), the accumulators are brought
//     to the new unit by an exact power of two, and (X*y) W1a is added -- K = 2d like the reference's concat, no
//     per-question weight mixing.  Backward: dI1 W1a^T is multiplied by y per output column on the accumulators, then
//     dI1 W1b^T is added.
#pragma once
#include <hip/hip_ext.h>
#include "macx_chain_api.hip.h"

namespace macx {


// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15); every lane of the row receives the same value (rotations
// by 8, 4, 2, 1: the pairing is the same tree in every lane, and addition commutes)
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov_f<0x128>(v);      // row_ror:8
  v += dpp_mov_f<0x124>(v);      // row_ror:4
  v += dpp_mov_f<0x122>(v);      // row_ror:2
  v += dpp_mov_f<0x121>(v);      // row_ror:1
  return v;
}

// sum over the 8 lanes 8 k .. 8 k + 7; every lane of the group receives the same value
__device__ __forceinline__ float row8_sum(float v) {
  v += dpp_mov_f<0x141>(v);      // row_half_mirror: i <-> 7 - i
  v += dpp_mov_f<0x1B>(v);       // quad_perm [3,2,1,0]
  v += dpp_mov_f<0xB1>(v);       // quad_perm [1,0,3,2]
  return v;
}

// ELU inside the chain kernels: exp(x) - 1 on the negative side, without the cancellation-free polynomial of elu_f.  The
// absolute error near zero (1 ulp of 1.0) is below what the value loses when it is stored as an H2 slot (2^-22 of the row
// maximum) or summed into a logit, and the epilogues apply it to 64 values per lane.
// tanh inside the chain kernels: libm's tanhf brought 100 spilled registers into stage B0 (64 values per lane at the 256-register
// cap).  |x| < 0.3: the odd Taylor polynomial through x^9 (truncation 5e-8 relative); else 1 - 2 / (exp(2|x|) + 1) with the
// hardware exponential and reciprocal (result >= 0.29: absolute error ~1 ulp of 1.0 -> <= 3e-7 relative).
__device__ __forceinline__ float chain_tanh(float x) {
  const float a = fabsf(x), x2 = x * x;
  float r = 62.0f / 2835.0f;
  r = fmaf(r, x2, -17.0f / 315.0f);
  r = fmaf(r, x2, 2.0f / 15.0f);
  r = fmaf(r, x2, -1.0f / 3.0f);
  r = fmaf(r, x2, 1.0f);
  r *= x;
  const float e = __expf(2.0f * fminf(a, 20.0f));
  const float t = 1.0f - 2.0f * __frcp_rn(e + 1.0f);
  return a < 0.3f ? r : copysignf(t, x);
}
__device__ __forceinline__ float chain_act(int act, float x) {
  if (act == ACT_ELU) return x > 0.0f ? x : __expf(x) - 1.0f;
  if (act == ACT_TANH) return chain_tanh(x);
  return act_apply(act, x);
}

// R_: rows of a tile.  64 whenever that gives the chip enough tiles; 32 / 16 for small batches (a data-parallel shard of the
// metric's 64 questions): a launch lasts as long as ONE tile takes, so with 25 tiles of 64 rows on 256 CUs the kernel ran as
// long as with 196 -- shorter tiles put the rows on more CUs (the weights then stream from L2 once per 16 or 32 rows instead
// of once per 64: the K loop turns L2-bound, at about half its time).
template <int D_, int R_ = 64>
struct ChainGeo {
  static constexpr int D = D_, R = R_, KG = D / 8, CB = D / 128, KS = D / 32;
  // the matrix instruction is v_mfma_f32_16x16x32_f16: 16 accumulator tiles of 16 x 16 per wave at d = 512.  (v_mfma_f32_32x32x16_f16
  // -- 4 tiles of 32 x 32 -- was built, parity-green and measured 13-24 us per launch SLOWER in round 4: on random fp16 operands the
  // 32 x 32 form issues at 48.5 cycles against 2 x 21.4, profiles/r04_mfma_probe.txt, r04_kloop_variants.txt.  Removed in round 6.)
  static constexpr int TM = 16;                      // edge of the instruction's output tile
  static constexpr int NWC = D >= 512 ? 8 : 4;       // waves along the columns
  static constexpr int NWR = 8 / NWC;                // waves along the rows
  static constexpr int CW = D / NWC;                 // columns per wave
  static constexpr int RPW = R / NWR;                // rows per wave
  static constexpr int RT = RPW / TM;                // row tiles per wave
  // accumulators are addressed as f32x4 acc[RT][CT]: four consecutive columns of one row per entry, one entry per 16-column tile
  static constexpr int CT = CW / 16;
  static constexpr int NRG = R / 16;                 // 16-row groups of the tile
  static constexpr int WPG = 8 / NRG;                // conversion passes: waves that share a row group
  static constexpr int IT = KG / (4 * WPG);          // ... slot columns per lane (16 rows x 4 slot columns per wave step)
  static constexpr int SB = R / 8;                   // backward stage B0: 8-row blocks per lane
  static_assert(R == 64 || ((R == 32 || R == 16) && NWR == 1), "short tiles: every wave holds all rows (d = 512)");
  static constexpr size_t P_BYTES = (size_t)4 * R * D;
  static constexpr int QS = 5;                       // backward: questions a tile can touch (N >= 16) -- their control vectors are staged in LDS
  static constexpr int BITS_LD = KG + 4;             // bytes per row of sBits (+4: rows 4 apart would share a bank)
  // P | sMax [8][R] | sPart [8][R] | sE [R] | sE2 [R] | sCol [2][D] | sW [D] | sC [QS][D] | sBits [R][BITS_LD]
  static constexpr size_t LDS = P_BYTES + (size_t)R * (8 + 8 + 1 + 1) * 4 + (size_t)(2 + 1 + QS) * D * 4 + (size_t)R * BITS_LD;
  static_assert((R * BITS_LD) % 16 == 0, "LDS carve-outs stay 16-byte aligned");
  static_assert(D % 128 == 0 && D >= 128 && D <= 512, "one workgroup holds 64 rows x D as H2 planes in LDS");
};

// what both kernels share: the tile in LDS, who owns what, the K loop, the row-exponent bookkeeping, the H2 emitters
template <int D_, int R_ = 64>
struct ChainCtx {
  using G = ChainGeo<D_, R_>;
  static constexpr int D = G::D, R = G::R, KG = G::KG, CB = G::CB, KS = G::KS, NWC = G::NWC, NWR = G::NWR, CT = G::CT, RT = G::RT, IT = G::IT,
                       WPG = G::WPG, SB = G::SB, TM = G::TM, CW = G::CW, RPW = G::RPW;
  char* P;          // [2 planes][KG][R] x 16 B: the stage's activation operand
  float* sMax;      // [8][R] partial row maxima
  float* sPart;     // [8][R] partial row sums (attention logits)
  int* sE;          // [R] exponents of the rows in P
  int* sE2;         // [R] second exponent table
  float* sCol;      // [2][D] column sums of the two row halves
  float* sW;        // [D] backward: the logits weight
  float* sC;        // [QS][D] backward: control vectors of the tile's questions
  uint8_t* sBits;   // [R][BITS_LD] forward: keep bits of the attention dropout (ops.py:312), one byte per 8 columns
  int tid, lane, wave, li, lg;
  int ar, ah;       // accumulator geometry: this lane's row within an output tile, and which group of four columns it holds
  int wr, wc, colbase, rowbase;
  int M, N, nvalid;
  size_t grow0;
  int crow; size_t cgrow; bool cvalid;      // conversion passes: this lane's row

  __device__ __forceinline__ void init(char* lds, int M_, int N_) {
    P = lds;
    sMax = reinterpret_cast<float*>(lds + G::P_BYTES);
    sPart = sMax + 8 * R;
    sE = reinterpret_cast<int*>(sPart + 8 * R);
    sE2 = sE + R;
    sCol = reinterpret_cast<float*>(sE2 + R);
    sW = sCol + 2 * D;
    sC = sW + D;
    sBits = reinterpret_cast<uint8_t*>(sC + G::QS * D);
    tid = threadIdx.x; lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    li = lane & 15; lg = lane >> 4;
    ar = li; ah = lg;
    wr = wave / NWC; wc = wave % NWC;
    colbase = wc * CW; rowbase = wr * RPW;
    M = M_; N = N_;
    crow = (wave / WPG) * 16 + li;
    set_tile(blockIdx.x);
  }
  __device__ __forceinline__ void set_tile(size_t t) {
    grow0 = t * R;
    nvalid = (int)min((size_t)R, (size_t)M - grow0);
    cgrow = grow0 + crow;
    cvalid = crow < nvalid;
  }
  __device__ __forceinline__ int ckg(int j) const { return ((wave % WPG) * IT + j) * 4 + lg; }   // conversion passes: slot column j of this lane
  __device__ __forceinline__ size_t qrow(size_t grow) const { return min((uint32_t)grow, (uint32_t)M - 1) / (uint32_t)N; }   // question of a row (rows < 2^31)

  __device__ __forceinline__ int row_exponent(int row, int w0, int nw) const {
    float m = sMax[w0 * R + row];
    for (int w = 1; w < nw; ++w) m = fmaxf(m, sMax[(w0 + w) * R + row]);
    return h2_exponent(m);
  }
  // exponent of a row after an accumulator epilogue (partial maxima from the NWC waves of its row half) / a conversion pass
  __device__ __forceinline__ int row_exponent_epi(int row) const { return row_exponent(row, (row / RPW) * NWC, NWC); }
  __device__ __forceinline__ int row_exponent_conv(int row) const { return row_exponent(row, (row >> 4) * WPG, WPG); }
  __device__ __forceinline__ int row_exponent_blk(int row) const { return row_exponent(row, 0, KG / 8); }
  enum { PASS_CONV = 0, PASS_EPI = 1, PASS_BLK = 2 };
  __device__ __forceinline__ int row_exponent_of(int row, int pass) const {
    return pass == PASS_EPI ? row_exponent_epi(row) : (pass == PASS_BLK ? row_exponent_blk(row) : row_exponent_conv(row));
  }

  // per-row bookkeeping once sMax is complete: exponent table and exponent bytes (wave 0 works).  (What a contraction over rows
  // needs besides -- the minimum exponent of a question's / of all rows -- its consumers take from these bytes themselves:
  // h2_emin_list_kernel, sb_h2_kernel.)
  __device__ __forceinline__ void publish_rows(int* eTab, int pass, const H2View& out) const {
    if (tid < 64) {
      const int r = min(tid, R - 1);                // (lanes past a short tile: no row)
      const bool v = tid < nvalid;
      const int e = row_exponent_of(r, pass);
      if (tid < R) eTab[r] = e;
      if (out.base && v) {
        int8_t* ex = out.exps() + (grow0 + r) * CB;
#pragma unroll
        for (int k = 0; k < CB; ++k) ex[k] = (int8_t)e;
      }
    }
  }
  // conversion pass, second half: v[j][0..7] (fp32, this lane's slots) -> row exponents -> H2 slots in P (+ HBM)
  __device__ __forceinline__ void convert_finish(float (&v)[IT][8], float m, int* eTab, const H2View& out) const {
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (lane < 16) sMax[wave * R + crow] = m;
    __syncthreads();
    const float s = h2_pow2(row_exponent_conv(crow));
    const size_t Rp = out.Rp();
    const size_t opb = out.plane_bytes();
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int kg = ckg(j);
      float xs[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) xs[q] = v[j][q] * s;
      u32x4 hi, lo;
      h2_split8(xs, hi, lo);
      char* d = P + ((size_t)kg * R + crow) * 16;
      *reinterpret_cast<u32x4*>(d) = hi;
      *reinterpret_cast<u32x4*>(d + (size_t)KG * R * 16) = lo;
      if (out.base && cvalid) {
        char* g = out.base + ((size_t)kg * Rp + cgrow) * 16;
        *reinterpret_cast<u32x4*>(g) = hi;
        *reinterpret_cast<u32x4*>(g + opb) = lo;
      }
    }
    publish_rows(eTab, PASS_CONV, out);
    __syncthreads();
  }
  // ---- row-block conversion pass (backward stage B0): lane (r8 = lane & 7, kq = lane >> 3) of wave w < KG / 8 holds slot column
  // 8 w + kq of the rows r8 + 8 s, s = 0..7 -- eight rows per lane, so sums over rows accumulate in registers, a wave's 8-lane
  // groups read and write whole 128-byte lines, and a wave owns its columns for the whole tile.
  // The values are parked in P as fp32 while the pass runs (blk_park: slot (kg, row) = 32 bytes, the tile's fp32 image is
  // exactly the size of its two fp16 planes) and only come back to registers here, once the pass's own accumulators are
  // gone: holding all 64 through the pass cost 86 spilled registers.
  __device__ __forceinline__ char* blk_slot(int kg, int row) const { return P + ((size_t)kg * R + row) * 32; }
  __device__ __forceinline__ void blk_park(int kg, int row, const float (&o)[8]) const {
    char* d = blk_slot(kg, row);
    const int h = ((lane >> 3) & 1) * 16;            // lanes 8 apart would write the same banks: swap the slot's halves
    *reinterpret_cast<f32x4*>(d + h) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(d + 16 - h) = f32x4{o[4], o[5], o[6], o[7]};
  }
  // mx[s]: max |value| of this lane's slot of row block s
  __device__ __forceinline__ void convert_finish_blk(const float (&mx)[8], const H2View& out, int* eTab) const {
    const int r8 = lane & 7, kq = lane >> 3;
    const bool active = wave < KG / 8;
    const int kg = 8 * wave + kq;
    if (active) {
#pragma unroll
      for (int sb = 0; sb < SB; ++sb) {
        float m = mx[sb];
        m = fmaxf(m, dpp_mov_f<0x128>(m));                 // lane ^ 8 (row_ror:8 within 16 lanes)
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (kq == 0) sMax[wave * R + 8 * sb + r8] = m;
      }
    }
    float v[8][8];
    if (active) {
      const int h = (kq & 1) * 16;
#pragma unroll
      for (int sb = 0; sb < SB; ++sb) {
        const char* d = blk_slot(kg, 8 * sb + r8);
        const f32x4 a = *reinterpret_cast<const f32x4*>(d + h), b = *reinterpret_cast<const f32x4*>(d + 16 - h);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[sb][q] = a[q]; v[sb][4 + q] = b[q]; }
      }
    }
    __syncthreads();                                       // sMax complete; every fp32 slot is in its owner's registers
    if (active) {
      const size_t Rp = out.Rp();
      const size_t opb = out.plane_bytes();
#pragma unroll
      for (int sb = 0; sb < SB; ++sb) {
        const int row = 8 * sb + r8;
        const float s = h2_pow2(row_exponent_blk(row));
        float xs[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) xs[q] = v[sb][q] * s;
        u32x4 hi, lo;
        h2_split8(xs, hi, lo);
        char* d = P + ((size_t)kg * R + row) * 16;
        *reinterpret_cast<u32x4*>(d) = hi;
        *reinterpret_cast<u32x4*>(d + (size_t)KG * R * 16) = lo;
        if (out.base && row < nvalid) {
          char* g = out.base + ((size_t)kg * Rp + grow0 + row) * 16;
          *reinterpret_cast<u32x4*>(g) = hi;
          *reinterpret_cast<u32x4*>(g + opb) = lo;
        }
      }
    }
    publish_rows(eTab, PASS_BLK, out);
    __syncthreads();
  }

  // an H2 tensor's rows of this tile -> P (a pure copy, lanes along rows) and its exponents -> eTab
  __device__ __forceinline__ void load_tile(const H2View& src, int* eTab) const {
    const size_t Rp = src.Rp();
    for (int f = tid; f < 2 * KG * R; f += 512) {
      const int r = f % R, s = f / R;                       // s = plane * KG + kg
      const size_t gr = min(grow0 + r, (size_t)M - 1);
      *reinterpret_cast<u32x4*>(P + (size_t)f * 16) = *reinterpret_cast<const u32x4*>(src.base + ((size_t)s * Rp + gr) * 16);
    }
    if (tid < R) eTab[tid] = (int)src.exps()[min(grow0 + tid, (size_t)M - 1) * CB];
    __syncthreads();
  }

  // ---- the K loop of one product: acc[T][c] += W^T-slot x A-slot over all d (three fp16 terms, smallest first).
  // KV is a bit mask.  KV & 4 (the default, CHAIN_KV_DEFAULT): the activation fragments of the next slice are requested in the MIDDLE
  // of the current slice's products instead of in front of them (their s_waitcnt lgkmcnt(0) then finds them long complete;
  // requested in front, the wait sits between the request and the first product) -- macx_opts.tune[MACX_TUNE_CHAIN_KV] = MACX_TUNE(0) is the A/B hook.
  // KV & 3 (only in a build with -DMACX_PROFILE_VARIANTS): measurement variants -- 1 no MFMA (the loads stay), 2 no weight loads
  // inside the loop, 3 no loads at all inside the loop; timing only, results are wrong.
  // (Measured and removed: static s_setprio for the second-dispatched waves, a term-major product order, the 32 x 32 x 16
  // instruction -- profiles/r04_kloop_variants.txt.)
  template <int KV>
  __device__ __forceinline__ void kloop(f32x4 (&acc)[RT][CT], const char* W) const {
    constexpr int KM = KV & 3;
    constexpr bool MID = (KV & 4) != 0 && RT >= 2;
    const char* wb = W + ((size_t)lg * D + colbase + li) * 16;
    const char* pa = P + ((size_t)lg * R + rowbase + li) * 16;
    // both operands of slice kt + 1 are requested before slice kt is multiplied: the weight slots from L2 (about one slice of
    // matrix work away), the activation slots from LDS.  (Round 5 tried requesting a product's FIRST weight slice ahead of the
    // elementwise work in front of the product -- 2 x CT registers carried through the epilogues -- instead of here, where its L2
    // round trip sits in front of the first MFMA: chain_fwd 93.6 -> 93.2 us, chain_bwd 88.1 -> 88.7, the step unchanged
    // (gpurun call 9); two waves per SIMD already cover it.  Not kept.  Neither was sending the LAST stage's stores (I2 forward, dX
    // backward) out ahead of its final vector pass -- the logits pass, the column sums -- so that the pass runs while they drain:
    // chain_fwd 96.1 -> 97.3 us, chain_bwd 91.7 -> 92.1 (call 11): the extra barrier costs what the overlap returns.)
    u32x4 bq[2][2][CT], aq[2][2][RT];        // [set][plane][tile]
    auto load_b = [&](auto set_c, int kt, bool in_loop) __attribute__((always_inline)) {
      constexpr int S = decltype(set_c)::value;
      if (!(KM >= 2 && in_loop)) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int c = 0; c < CT; ++c)
            bq[S][pl][c] = *reinterpret_cast<const u32x4*>(wb + ((size_t)(kt * 2 + pl) * 4 * D) * 16 + c * 256);
      }
    };
    auto load_a = [&](auto set_c, int kt, bool in_loop) __attribute__((always_inline)) {
      constexpr int S = decltype(set_c)::value;
      if (!(KM >= 3 && in_loop)) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int t = 0; t < RT; ++t)
            aq[S][pl][t] = *reinterpret_cast<const u32x4*>(pa + ((size_t)(pl * KG + 4 * kt) * R + 16 * t) * 16);
      }
    };
    auto mm = [&](auto set_c, int t0, int t1) __attribute__((always_inline)) {
      constexpr int S = decltype(set_c)::value;
      if (KM == 1) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int c = 0; c < CT; ++c) asm volatile("" ::"v"(bq[S][pl][c]));
#pragma unroll
          for (int t = 0; t < RT; ++t) asm volatile("" ::"v"(aq[S][pl][t]));
        }
        return;
      }
#pragma unroll
      for (int t = t0; t < t1; ++t) {
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[t][c] = mfma_f16(bq[S][0][c], aq[S][1][t], acc[t][c]);     // w_hi a_lo
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[t][c] = mfma_f16(bq[S][1][c], aq[S][0][t], acc[t][c]);     // w_lo a_hi
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[t][c] = mfma_f16(bq[S][0][c], aq[S][0][t], acc[t][c]);     // w_hi a_hi
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // the scheduler would sink a slice's loads to the end of the previous slice's products (shortest live range), i.e. to where
    // they are needed: the order "request slice kt + 1, multiply slice kt" is pinned
    load_b(S0{}, 0, false); load_a(S0{}, 0, false);
    if (KM >= 2) { load_b(S1{}, 1, false); load_a(S1{}, 1, false); }
    auto half_step = [&](auto cur, auto nxt, int kn) __attribute__((always_inline)) {
      load_b(nxt, kn, true);
      if (!MID) load_a(nxt, kn, true);
      __builtin_amdgcn_sched_barrier(0);
      if (MID) {
        mm(cur, 0, RT / 2);
        __builtin_amdgcn_sched_barrier(0);
        load_a(nxt, kn, true);
        __builtin_amdgcn_sched_barrier(0);
        mm(cur, RT / 2, RT);
      } else {
        mm(cur, 0, RT);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int kt = 0; kt < KS; kt += 2) {
      half_step(S0{}, S1{}, kt + 1);
      half_step(S1{}, S0{}, min(kt + 2, KS - 2));   // (unconditional: behind a branch the wait-count pass drains every load at the join)
    }
  }
  __device__ __forceinline__ void zero_acc(f32x4 (&acc)[RT][CT]) const {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- accumulator geometry: lane (ar = lane & 15, ah = lane >> 4) holds, for row tile t and 16-column tile c, row rowbase + 16 t + ar
  //      and the four columns acol(c) .. + 3 in acc[t][c][0..3].  A group is half of the slot akg(c).
  __device__ __forceinline__ int arow(int t) const { return rowbase + TM * t + ar; }
  __device__ __forceinline__ int acol(int c) const { return colbase + 16 * c + 4 * ah; }
  __device__ __forceinline__ int akg(int c) const { return (colbase >> 3) + 2 * c + (ah >> 1); }
  __device__ __forceinline__ int ahalf() const { return (ah & 1) * 8; }
  // reductions over the lanes that hold the same row (different column groups) / the same columns (different rows)
  __device__ __forceinline__ float rowred_max(float m) const {
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    return fmaxf(m, __shfl_xor(m, 32, 64));
  }
  __device__ __forceinline__ float rowred_sum(float v) const {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
  }
  __device__ __forceinline__ float colred_sum(float v) const {
    return row16_sum(v);
  }
  __device__ __forceinline__ bool row_writer() const { return ah == 0; }     // one lane per row
  __device__ __forceinline__ bool col_writer() const { return ar == 0; }     // one lane per column group

  // partial row maxima of the (finished) accumulator values -> sMax[wave]
  __device__ __forceinline__ void rowmax(const f32x4 (&acc)[RT][CT]) const {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      float m = 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) m = fmaxf(m, fabsf(acc[t][c][q]));
      m = rowred_max(m);
      if (row_writer()) sMax[wave * R + arow(t)] = m;
    }
  }
  // after the barrier that completed sMax: split the accumulators, write them over P (to_p) and to `out`
  __device__ __forceinline__ void emit(const f32x4 (&acc)[RT][CT], bool to_p, const H2View& out) const {
    const size_t Rp = out.Rp();
    const size_t opb = out.plane_bytes();
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int row = arow(t);
      const float s = h2_pow2(row_exponent_epi(row));
      const bool st = out.base && row < nvalid;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const f32x4 xv = acc[t][c] * s;
        u32x2 hi, lo;
        hi[0] = pk_f16(xv[0], xv[1]);
        hi[1] = pk_f16(xv[2], xv[3]);
        const f32x2_t b0 = unpk_f16(hi[0]), b1 = unpk_f16(hi[1]);
        lo[0] = pk_f16(xv[0] - b0[0], xv[1] - b0[1]);
        lo[1] = pk_f16(xv[2] - b1[0], xv[3] - b1[1]);
        const int kg = akg(c);
        const int half = ahalf();
        if (to_p) {
          char* d = P + ((size_t)kg * R + row) * 16 + half;
          *reinterpret_cast<u32x2*>(d) = hi;
          *reinterpret_cast<u32x2*>(d + (size_t)KG * R * 16) = lo;
        }
        if (st) {
          char* g = out.base + ((size_t)kg * Rp + grow0 + row) * 16 + half;
          *reinterpret_cast<u32x2*>(g) = hi;
          *reinterpret_cast<u32x2*>(g + opb) = lo;
        }
      }
    }
  }
  // column sums of the accumulator values over the tile's rows -> part[D] (rows past the end hold zeros)
  __device__ __forceinline__ void colsum(const f32x4 (&acc)[RT][CT], float* part) const {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float s = acc[0][c][q];
#pragma unroll
        for (int t = 1; t < RT; ++t) s += acc[t][c][q];
        s = colred_sum(s);
        if (col_writer()) {
          const int col = acol(c) + q;
          if (NWR == 1) part[col] = s;
          else sCol[wr * D + col] = s;
        }
      }
    }
  }
  __device__ __forceinline__ void colsum_finish(float* part) const {     // after a barrier; NWR == 2 only
    if (NWR == 2)
      for (int c = tid; c < D; c += 512) part[c] = sCol[c] + sCol[D + c];
  }
};

// =========================================================================================================================

template <int D_, int KV = 0, int R_ = 64>
__global__ __launch_bounds__(512) void chain_fwd_kernel(const ChainFwdP p) {
  using C = ChainCtx<D_, R_>;
  constexpr int D = C::D, R = C::R, KG = C::KG, CT = C::CT, RT = C::RT, IT = C::IT;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  C x;
  x.init(lds, p.M, p.N);
  const int M = p.M;

  // =====================================================================================================================
  // stage 0: the operand of the first product.  dropout(KB) of the tile x stands on -> H2 planes in P (+ HBM), the keep bits of both
  // read-dropout sites (the second site's also into sBits: stage 3 reads them there)
  auto stage0 = [&](uint32_t k1, uint32_t k2, uint8_t* bits1, uint8_t* bytes2, const H2View& KBd) __attribute__((always_inline)) {
    float v[IT][8];
    float m = 0.f;
    const bool drop1 = p.thr1 < (1u << 24), drop2 = p.thr2 < (1u << 24);
    const uint32_t key1 = run_key(k1, p.word), key2 = run_key(k2, p.word);
    const size_t Rp2 = (size_t)M + H2_PAD_ROWS;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int kg = x.ckg(j);
      // (rows past the end read the last row and are zeroed: a branch around a load costs a full wait per load)
      const float* src = p.kb + min(x.cgrow, (size_t)M - 1) * D + kg * 8;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[j][q] = x.cvalid ? a0[q] : 0.f; v[j][4 + q] = x.cvalid ? a1[q] : 0.f; }
    }
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int kg = x.ckg(j);
      const uint32_t e0 = p.first + (uint32_t)(x.cgrow * (size_t)p.dlog + kg * 8);
      if (drop1) {
        uint32_t byte = 0;             // (e0 is even: the width is a multiple of 128)
#pragma unroll
        for (int q = 0; q < 8; q += 2) byte |= keep_pair(e0 + q, key1, p.thr1) << q;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[j][q] = ((byte >> q) & 1u) ? v[j][q] * p.inv1 : 0.f;
        if (bits1 && x.cvalid) bits1[x.cgrow * KG + kg] = (uint8_t)byte;
      }
      if (drop2) {
        // hashed once per element, here: the logits epilogue of stage 3 reads the bits back from LDS
        uint32_t byte = 0;
#pragma unroll
        for (int q = 0; q < 8; q += 2) byte |= keep_pair(e0 + q, key2, p.thr2) << q;
        x.sBits[x.crow * C::G::BITS_LD + kg] = (uint8_t)byte;
        if (bytes2 && x.cvalid) bytes2[(size_t)kg * Rp2 + x.cgrow] = (uint8_t)byte;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) m = fmaxf(m, fabsf(v[j][q]));
    }
    x.convert_finish(v, m, x.sE, KBd);
  };
  if constexpr (D == 512) {
    // the first p.pre.nfill workgroups are fillers on the CUs the tile grid leaves idle (ChainPreP): the previous step's write-unit
    // linear, this step's y (both handed to the tiles of this launch through counters), then stage 0 of the NEXT step
#ifdef MACX_FILL_PROF
#define MACX_STAMP(cond, k) do { if ((cond) && p.pre.step == 5 && p.pre.fail && x.tid == 0) p.pre.fail[-31 + (k)] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
#define MACX_STAMP(cond, k) do { } while (0)
#endif
    if ((int)blockIdx.x < p.pre.nfill) {
      const int f = blockIdx.x, nf = p.pre.nfill;
      MACX_STAMP(f == 0, 0);
      float (*red)[16][20] = reinterpret_cast<float (*)[16][20]>(lds);       // two linear tiles side by side: [2 halves][2 x 4][16][20]
      const int half = x.tid >> 8, tl = x.tid & 255;
      // every wave's stores have left, then one increment
      auto signal = [&](uint32_t* c0, uint32_t* c1) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (x.tid == 0 && c0) __hip_atomic_fetch_add(c0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (x.tid == 64 && c1) __hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      auto wait_ge = [&](uint32_t* c, uint32_t v) __attribute__((always_inline)) {
        if (x.tid == 0) {
          uint32_t spins = 0;
          while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22)) { __hip_atomic_store(p.pre.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          }
        }
        __syncthreads();
      };
      const bool tail = p.pre.wlin.n_out != 0;
      if (tail) {
        // the previous step's write tiles (16 rows x 32 columns: 64 tiles at B = 64, one round on the fillers' 2 x nf halves)
        const LinP& wl = p.pre.wlin;
        const int ntx = wl.n_out / 32, nun = ntx * ((wl.rows + 15) / 16);
#pragma unroll 1
        for (int u0 = 2 * f; u0 < nun; u0 += 2 * nf) {
          const int ua = u0, ub = min(u0 + 1, nun - 1), bya = ua / ntx, byb = ub / ntx;
          const int u = half ? ub : ua;
          small_linear_tile<1, false, true, false, 2>(wl, u % ntx, u / ntx, 0, red + 8 * half, tl, u0 + half < nun);
          signal(p.pre.gflag + bya, u0 + 1 < nun ? p.pre.gflag + byb : nullptr);
        }
      }
      MACX_STAMP(f == 0, 1);
      if (p.pre.ylin.n_out) {
        const LinP& yl = p.pre.ylin;
        const int ntx = yl.n_out / 32, nunits = ntx * ((yl.rows + 15) / 16);
        const uint32_t wtiles = (uint32_t)(p.pre.wlin.n_out / 32);
#pragma unroll 1
        for (int u0 = 2 * f; u0 < nunits; u0 += 2 * nf) {
          const int ua = u0, ub = min(u0 + 1, nunits - 1), bya = ua / ntx, byb = ub / ntx;
          if (tail) {
            wait_ge(p.pre.gflag + bya, wtiles);
            if (byb != bya) wait_ge(p.pre.gflag + byb, wtiles);
          }
          const int u = half ? ub : ua;
          small_linear_tile<1, false, true, true, 2>(yl, u % ntx, u / ntx, 0, red + 8 * half, tl, u0 + half < nunits);
          __syncthreads();
        }
        signal(p.pre.yflag, nullptr);
      }
      MACX_STAMP(f == 0, 2);
      if (p.pre.KBd.base) {
        const int ntile = (M + R - 1) / R;
#pragma unroll 1
        for (int t = f; t < ntile; t += p.pre.nfill) {
          x.set_tile((size_t)t);
          stage0(p.pre.key1, p.pre.key2, p.pre.bits1, p.pre.bytes2, p.pre.KBd);
        }
      }
      MACX_STAMP(f == 0, 3);
      return;
    }
    x.set_tile((size_t)blockIdx.x - p.pre.nfill);
    MACX_STAMP((int)blockIdx.x == p.pre.nfill, 8);
  }
  if (p.mode == 0) {
    stage0(p.key1, p.key2, p.bits1, p.bytes2, p.KBd);
  } else if (p.mode == 2) {
    // the previous launch's fillers left the planes and the keep bytes behind
    if (p.thr2 < (1u << 24)) {
      const size_t Rp2 = (size_t)M + H2_PAD_ROWS;
      const int r = x.tid & 63, k0 = x.tid >> 6;
      const size_t gr = min(x.grow0 + r, (size_t)M - 1);
      if (r < R) {
#pragma unroll
        for (int j = 0; j < KG / 8; ++j) x.sBits[r * C::G::BITS_LD + k0 + 8 * j] = p.bytes2[(size_t)(k0 + 8 * j) * Rp2 + gr];
      }
    }
    x.load_tile(p.KBd, x.sE);
  } else {
    x.load_tile(p.X, x.sE);
  }
  if (p.dbg & 1) return;

  f32x4 acc[RT][CT];
  // x = act(acc * 2^-(eA[row] + eW) + bias), left in acc
  auto bias_act = [&](auto act_c, const int* eTab, int eW, const float* bias) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_c)::value;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + x.acol(c));
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const float s = h2_unscale(eTab[x.arow(t)], eW);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][c][q] = chain_act(ACT, fmaf(acc[t][c][q], s, bv[q]));
      }
    }
  };
  auto bias_act_any = [&](int act, const int* eTab, int eW, const float* bias) __attribute__((always_inline)) {
    switch (act) {     // ONE switch per stage, not one per value (macx_gemm_h2.hip.h has the story)
      case ACT_TANH: bias_act(std::integral_constant<int, ACT_TANH>{}, eTab, eW, bias); break;
      case ACT_SIGMOID: bias_act(std::integral_constant<int, ACT_SIGMOID>{}, eTab, eW, bias); break;
      case ACT_ELU: bias_act(std::integral_constant<int, ACT_ELU>{}, eTab, eW, bias); break;
      case ACT_RELU: bias_act(std::integral_constant<int, ACT_RELU>{}, eTab, eW, bias); break;
      default: bias_act(std::integral_constant<int, ACT_NON>{}, eTab, eW, bias); break;
    }
  };

  // =====================================================================================================================
  // stage 1: X = KBd Wx + bx
  if (p.mode != 1) {
    x.zero_acc(acc);
    x.template kloop<KV>(acc, p.Wx.planes);
    bias_act(std::integral_constant<int, ACT_NON>{}, x.sE, *p.Wx.exp, p.bx);
    x.rowmax(acc);
    __syncthreads();                       // every wave is done reading P; sMax is complete
    x.emit(acc, true, p.X);
    x.publish_rows(x.sE, C::PASS_EPI, p.X);
    __syncthreads();
  }
  if (p.dbg & 2) return;

  // =====================================================================================================================
  // stage 2: H1 = act(X W1b + (X * y) W1a + b1)
  x.zero_acc(acc);
  x.template kloop<KV>(acc, p.W1b.planes);
  constexpr bool YCOH = D == 512;                 // y may come from this launch's fillers (ChainPreP): agent-scope loads
  if constexpr (YCOH) {
    MACX_STAMP((int)blockIdx.x == p.pre.nfill, 9);
    if (p.pre.ylin.n_out && x.tid == 0) {
      uint32_t spins = 0;
      while (__hip_atomic_load(p.pre.yflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)p.pre.nfill) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) { __hip_atomic_store(p.pre.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
  }
  if constexpr (YCOH) MACX_STAMP((int)blockIdx.x == p.pre.nfill, 10);
  __syncthreads();                         // every wave is done reading X (and y is complete)
  {
    float v[IT][8];
    float m = 0.f;
    const float inv = h2_pow2(-x.sE[x.crow]);
    const float* yq = p.y + x.qrow(x.cgrow) * D;
    auto ldy = [&](const float* q) __attribute__((always_inline)) {
      if constexpr (YCOH) {
        const uint64_t* q8 = reinterpret_cast<const uint64_t*>(q);
        const uint64_t lo = __hip_atomic_load(q8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t hi = __hip_atomic_load(q8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return f32x4{__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float((uint32_t)hi),
                     __uint_as_float((uint32_t)(hi >> 32))};
      } else {
        return *reinterpret_cast<const f32x4*>(q);
      }
    };
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int kg = x.ckg(j);
      const char* s = x.P + ((size_t)kg * R + x.crow) * 16;
      const u32x4 hi = *reinterpret_cast<const u32x4*>(s), lo = *reinterpret_cast<const u32x4*>(s + (size_t)KG * R * 16);
      float xv[8];
      h2_join8(hi, lo, inv, xv);
      const f32x4 y0 = ldy(yq + kg * 8), y1 = ldy(yq + kg * 8 + 4);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        v[j][q] = xv[q] * (q < 4 ? y0[q & 3] : y1[q & 3]);      // ops.py:703: the product the reference rounds to fp32
        m = fmaxf(m, fabsf(v[j][q]));
      }
    }
    x.convert_finish(v, m, x.sE2, H2View{nullptr, M, D});
  }
  {
    // accumulators: units 2^-(eX + e1b)  ->  2^-(eXy + e1a), exactly
    const int de = *p.W1a.exp - *p.W1b.exp;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int k = x.sE2[x.arow(t)] - x.sE[x.arow(t)] + de;
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][c][q] = ldexpf(acc[t][c][q], k);
    }
  }
  x.template kloop<KV>(acc, p.W1a.planes);
  bias_act_any(p.act1, x.sE2, *p.W1a.exp, p.b1);
  x.rowmax(acc);
  __syncthreads();
  x.emit(acc, true, p.H1);
  x.publish_rows(x.sE, C::PASS_EPI, p.H1);
  __syncthreads();
  if (p.dbg & 4) return;

  // =====================================================================================================================
  // stage 3: I2 = H1 W2 + b2 ; logits = dropout(act(I2 * c)) . w_k
  x.zero_acc(acc);
  x.template kloop<KV>(acc, p.W2.planes);
  bias_act(std::integral_constant<int, ACT_NON>{}, x.sE, *p.W2.exp, p.b2);
  x.rowmax(acc);
  {
    const bool drop2 = p.thr2 < (1u << 24);
    auto logit_pass = [&](auto act_c) __attribute__((always_inline)) {
      constexpr int ACT = decltype(act_c)::value;
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const size_t gr = min(x.grow0 + x.arow(t), (size_t)M - 1);
        const float* cq = p.c + (size_t)((uint32_t)gr / (uint32_t)p.N) * D;
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const int col = x.acol(c);
          const f32x4 cv = *reinterpret_cast<const f32x4*>(cq + col);
          const f32x4 wv = *reinterpret_cast<const f32x4*>(p.wk + col);
          const uint32_t kb = drop2 ? (uint32_t)x.sBits[x.arow(t) * C::G::BITS_LD + (col >> 3)] >> (col & 4) : 0xFu;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float g = chain_act(ACT, acc[t][c][q] * cv[q]);
            if (drop2) g = ((kb >> q) & 1u) ? g * p.inv2 : 0.f;
            part = fmaf(g, wv[q], part);
          }
        }
        part = x.rowred_sum(part);
        if (x.row_writer()) x.sPart[x.wave * R + x.arow(t)] = part;
      }
    };
    switch (p.act2) {
      case ACT_TANH: logit_pass(std::integral_constant<int, ACT_TANH>{}); break;
      case ACT_SIGMOID: logit_pass(std::integral_constant<int, ACT_SIGMOID>{}); break;
      case ACT_ELU: logit_pass(std::integral_constant<int, ACT_ELU>{}); break;
      case ACT_RELU: logit_pass(std::integral_constant<int, ACT_RELU>{}); break;
      default: logit_pass(std::integral_constant<int, ACT_NON>{}); break;
    }
  }
  __syncthreads();
  x.emit(acc, false, p.I2);
  x.publish_rows(x.sE2, C::PASS_EPI, p.I2);
  if constexpr (D == 512) MACX_STAMP((int)blockIdx.x == p.pre.nfill, 11);
  if (x.tid < x.nvalid) {
    const int w0 = (x.tid / C::RPW) * C::NWC;
    float s = x.sPart[w0 * R + x.tid];
#pragma unroll
    for (int w = 1; w < C::NWC; ++w) s += x.sPart[(w0 + w) * R + x.tid];          // fixed order
    p.logits[x.grow0 + x.tid] = s;
  }
}

// e0 / e1 (both or neither): HIP events that receive the KERNEL's own start and stop timestamps (hipExtLaunchKernelGGL: the dispatch
// packet's signal times, what a kernel trace reports) -- macx_cell_forward_chain_time
template <int D_, int KV = 0, int R_ = 64>
inline hipError_t chain_fwd_launch_t(const ChainFwdP& p, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  auto kern = chain_fwd_kernel<D_, KV, R_>;
  constexpr size_t lds = ChainGeo<D_, R_>::LDS;
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  const int grid = (p.M + R_ - 1) / R_ + (D_ == 512 ? p.pre.nfill : 0);
  if (e0 && e1) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, e0, e1, 0, p);
  else hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p);
  return hipGetLastError();
}


#ifdef MACX_CHAIN_FWD_TU      // macx_chain_fwd.hip: the one translation unit that instantiates the forward kernels
hipError_t chain_fwd_launch(const ChainFwdP& p, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  switch (p.d / 128) {
    case 1: return chain_fwd_launch_t<128>(p, st, e0, e1);
    case 2: return chain_fwd_launch_t<256>(p, st, e0, e1);
    case 3: return chain_fwd_launch_t<384>(p, st, e0, e1);
    case 4:
      switch (chain_tile_rows(p.d, (size_t)p.M)) {
        case 16: return chain_fwd_launch_t<512, 0, 16>(p, st, e0, e1);
        case 32: return chain_fwd_launch_t<512, 0, 32>(p, st, e0, e1);
        default: break;
      }
#ifdef MACX_PROFILE_VARIANTS
      switch (p.dbg >> 3) {
        case 1: return chain_fwd_launch_t<512, 1>(p, st, e0, e1);
        case 2: return chain_fwd_launch_t<512, 2>(p, st, e0, e1);
        case 3: return chain_fwd_launch_t<512, 3>(p, st, e0, e1);
        default: break;
      }
#endif
      return chain_kv() == 0 ? chain_fwd_launch_t<512, 0>(p, st, e0, e1) : chain_fwd_launch_t<512, CHAIN_KV_DEFAULT>(p, st, e0, e1);
    default: return hipErrorInvalidValue;
  }
}
#endif


// =========================================================================================================================


// =========================================================================================================================
// dKB jobs (ChainDkbP, macx_chain_api.hip.h): dKB[tile] (+)= (dX_s[tile] Wx^T) * kbmask_s / keep + att_s (x) dinfo_s for one 64-row
// tile of one step; a workgroup runs a SEQUENCE of jobs (the fillers of a chain_bwd launch, the closing launch).
//   * the tile comes from HBM as it lies -- 16 slots per thread, a pure copy -- and the NEXT job's slots are requested right behind
//     the current job's K loop and written to LDS behind its epilogue: only a sequence's first tile load is exposed;
//   * the K loop is the chain kernels' (ChainCtx::kloop);
//   * the epilogue stays on the accumulators where they lie (lane (ar, ah): four consecutive columns of row 16 t + ar per 16-column
//     tile): keep bits, + att (x) dinfo, then the read-modify-write of the caller's gradient as 16-byte pieces, four lanes per
//     64-byte run.  (The first version staged the accumulators through LDS for a row-major pass over whole 2 KB rows: 5 us of
//     staging + two barriers per job, profiles/r06_dkb_jobs.txt -- a job was 27 us of which the K loop 14.5.)
// No load sits behind a branch (the wait-count pass drains every load at a join): a job that stores reads the old values all the same,
// missing keep bits are stood in for by any readable bytes, and behind the last job the next tile's request degenerates to cache hits.
struct DkbJob { int s, t; bool write, valid; };

template <int KV, class C, class NEXT>
__device__ __forceinline__ void dkb_run(C& x, const ChainDkbP& q, DkbJob job, NEXT next) {
  constexpr int D = C::D, R = C::R, KG = C::KG, CB = C::CB, CT = C::CT, RT = C::RT;
  constexpr int NS = 2 * KG * R / 512;           // tile slots per thread
  static_assert(R == 64 && D == 512 && CT == 4 && RT == 4, "dKB jobs: d = 512, 64-row tiles");
  const size_t Rp = (size_t)x.M + H2_PAD_ROWS;
  u32x4 pf[NS];
  f32x4 dq = {0.f, 0.f, 0.f, 0.f};
  int pe = 0;
  // tile copy: thread (lr = tid & 63, ls = tid >> 6) moves row lr of the slot columns (plane * KG + kg) ls + 8 k.
  // (the opaque zeros: addresses are rebuilt where they are used -- hoisted out of the job loop they would live through the K loop,
  // whose registers are full: 175 spilled registers without them)
  // real = false (behind the last job): the same sixteen loads, all of one slot column -- cache hits nobody waits long for
  auto request = [&](const DkbJob& j, bool real) __attribute__((always_inline)) {
    int z = 0;
    asm volatile("" : "+v"(z));
    const int lr = (x.tid & 63) + z, ls = (x.tid >> 6) + z;
    const char* base = q.dX + (size_t)j.s * q.dx_step;
    const size_t gr = min((size_t)j.t * R + lr, (size_t)x.M - 1);
    const size_t cs = real ? Rp : 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) pf[k] = *reinterpret_cast<const u32x4*>(base + ((size_t)(ls + 8 * k) * cs + gr) * 16);
    pe = (int)reinterpret_cast<const int8_t*>(base + 2 * (size_t)KG * Rp * 16)[gr * CB];       // (threads 0 .. R - 1 use theirs)
    // dinfo of the (at most two, N >= 64) questions the tile touches: 2 x D floats for sC, one float4 per thread of the first 256
    const uint32_t un = (uint32_t)x.N, jq0 = (uint32_t)((size_t)j.t * R) / un, qlast = ((uint32_t)x.M - 1) / un;
    const int dq_q = ((x.tid >> 7) & 1) + z, dq_c = (x.tid & 127) + z;
    dq = *reinterpret_cast<const f32x4*>(q.dinfo + (size_t)j.s * q.dinfo_step + (size_t)min(jq0 + (uint32_t)dq_q, qlast) * q.ld_dinfo + dq_c * 4);
  };
#ifdef MACX_FILL_PROF
  int nstamp = 0;
#define MACX_DSTAMP() do { if (q.prof && q.step == 5 && blockIdx.x == gridDim.x - q.nfill && x.tid == 0 && nstamp < 15) q.prof[nstamp++] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
#define MACX_DSTAMP() do { } while (0)
#endif
  if (!job.valid) return;
  MACX_DSTAMP();
  request(job, true);
#pragma unroll 1
  while (true) {
    x.set_tile((size_t)job.t);
    // (opaque per job: the tile's addresses are rebuilt where they are used instead of living through the K loop)
    asm volatile("" : "+s"(job.t), "+s"(job.s));
    {
      int z = 0;
      asm volatile("" : "+v"(z));
      const int lr = (x.tid & 63) + z, ls = (x.tid >> 6) + z;
#pragma unroll
      for (int k = 0; k < NS; ++k) *reinterpret_cast<u32x4*>(x.P + ((size_t)(ls + 8 * k) * R + lr) * 16) = pf[k];
    }
    if (x.tid < R) x.sE[x.tid] = pe;
    if (x.tid < 256) *reinterpret_cast<f32x4*>(x.sC + (x.tid >> 7) * D + (x.tid & 127) * 4) = dq;
    __syncthreads();
    MACX_DSTAMP();                               // tile in LDS
    f32x4 acc[RT][CT];
    x.zero_acc(acc);
    if (!(q.dbg & 1)) x.template kloop<KV>(acc, q.WxT.planes);
    MACX_DSTAMP();                               // K loop done
    const DkbJob nj = next(job);
    request(nj.valid ? nj : job, nj.valid);      // lands while the epilogue runs
    if (!(q.dbg & 8)) {
      const bool write = job.write || (q.dbg & 2);
      int z = 0;
      asm volatile("" : "+v"(z));
      const int ar = x.ar + z, ah = x.ah + z;
      const uint32_t un = (uint32_t)x.N, q0 = (uint32_t)x.grow0 / un;
      const uint32_t qb1 = (q0 + 1) * un;          // first row of the next question
      // (N >= 64, dkb_fill_plan: a tile touches at most two questions)
      const uint8_t* bits = q.bits ? q.bits + (size_t)job.s * q.bits_step : reinterpret_cast<const uint8_t*>(q.out);
      const uint32_t bits_or = q.bits ? 0u : 0xFFu;
      const float* att = q.att + (size_t)job.s * q.att_step;
      const float* dinfo = x.sC + x.colbase + 4 * ah;      // [2][D] in LDS: the tile's first question and the one behind it
      const int eW = *q.WxT.exp;
      // the old values first (the longest round trip), then what the new contribution needs
      f32x4 old[RT][CT];
      uint32_t gr[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        gr[t] = min((uint32_t)x.grow0 + x.rowbase + 16 * t + ar, (uint32_t)x.M - 1);
        const float* o = q.out + (size_t)gr[t] * D + x.colbase + 4 * ah;
#pragma unroll
        for (int c = 0; c < CT; ++c) old[t][c] = *reinterpret_cast<const f32x4*>(o + 16 * c);
      }
      uint64_t kb8[RT];                            // the row's keep bytes of this wave's 64 columns
      float a[RT], sc[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        a[t] = att[gr[t]];
        kb8[t] = *reinterpret_cast<const uint64_t*>(bits + (size_t)gr[t] * KG + (x.colbase >> 3));
        sc[t] = h2_unscale(x.sE[x.rowbase + 16 * t + ar], eW);
      }
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int row = x.rowbase + 16 * t + ar;
        const bool second = gr[t] >= qb1;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const f32x4 dv = *reinterpret_cast<const f32x4*>(dinfo + (second ? D : 0) + 16 * c);
          const uint32_t nib = ((uint32_t)(kb8[t] >> (16 * c + 4 * ah)) | bits_or) & 0xFu;      // byte 2 c + (ah >> 1), its half ah & 1
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float m = ((nib >> e) & 1u) ? acc[t][c][e] * sc[t] * q.inv_keep : 0.f;
            o[e] = fmaf(a[t], dv[e], m) + (write ? 0.f : old[t][c][e]);
          }
          if (row < x.nvalid) *reinterpret_cast<f32x4*>(q.out + ((size_t)x.grow0 + row) * D + x.colbase + 16 * c + 4 * ah) = o;
        }
      }
    }
    MACX_DSTAMP();                               // epilogue issued
    if (!nj.valid) break;
    job = nj;
    __syncthreads();                             // every wave is done with the tile and its exponents: the next one overwrites them
  }
}

// filler workgroup f of the chain_bwd launch that carries step q.step's jobs: tiles u = f, f + nfill, .. of the covered list
template <int KV, class C>
__device__ __forceinline__ void dkb_fill(C& x, const ChainDkbP& q, int f) {
  const int ntile = (x.M + C::R - 1) / C::R;
  const int s = q.step;
  const int w0 = q.nskip ? dkb_win0(s, q.nskip, ntile) : 0;
  auto job_of = [&](int j) {
    DkbJob r;
    const int u = f + q.nfill * j;
    r.valid = j < q.njobs && u < ntile - q.nskip;
    r.s = s;
    r.t = u < w0 ? u : u + q.nskip;
    r.write = true;                              // the first contribution a tile receives is stored, not added
    for (int s2 = s + 1; s2 < q.p; ++s2) r.write = r.write && q.nskip && dkb_skipped(s2, r.t, q.nskip, ntile);
    return r;
  };
  int j = 0;
  dkb_run<KV>(x, q, job_of(0), [&](const DkbJob&) { return job_of(++j); });
}

// the closing launch: per tile, the (step, tile) pairs the filler launches left out (steps p - 1 .. 1), then step 0
template <int D_, int KV>
__global__ __launch_bounds__(512) void chain_dkb_rest_kernel(const ChainDkbP q, int M, int N) {
  using C = ChainCtx<D_, 64>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  C x;
  x.init(lds, M, N);
  const int ntile = gridDim.x, t = blockIdx.x;
  bool touched = false;
  for (int s = 1; s < q.p; ++s) touched = touched || !(q.nskip && dkb_skipped(s, t, q.nskip, ntile));
  auto below = [&](int s0, bool wr) {            // the first job among steps s0, s0 - 1, .. 0
    DkbJob r;
    int s = s0;
    while (s > 0 && !(q.nskip && dkb_skipped(s, t, q.nskip, ntile))) --s;
    r.s = s; r.t = t; r.write = wr; r.valid = s >= 0;
    return r;
  };
  dkb_run<KV>(x, q, below(q.p - 1, !touched), [&](const DkbJob& cur) { return cur.s > 0 ? below(cur.s - 1, false) : DkbJob{0, 0, false, false}; });
}

// A2: readCtrlAct as a compile-time constant -- one kernel per activation (chain_bwd_launch_t); several arms in one kernel met in one
// register allocation (two arms: 86 spilled registers; a per-value run-time switch: 102-104).
template <int D_, int KV = 0, int A2 = -1, int R_ = 64>
__global__ __launch_bounds__(512) void chain_bwd_kernel(const ChainBwdP p) {
  using C = ChainCtx<D_, R_>;
  constexpr int D = C::D, R = C::R, KG = C::KG, CB = C::CB, CT = C::CT, RT = C::RT, IT = C::IT;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  C x;
  x.init(lds, p.M, p.N);
  const int M = p.M;
  if constexpr (D == 512 && R == 64) {
    // workgroups past the tile grid: dKB jobs of the previously differentiated step on the CUs this launch leaves idle
    const int ntile = (M + R - 1) / R;
    if ((int)blockIdx.x >= ntile) {
      dkb_fill<KV>(x, p.dkb, (int)blockIdx.x - ntile);
      return;
    }
  }
#ifdef MACX_FILL_PROF
#define MACX_BSTAMP(k) do { if (p.dkb.prof && p.dkb.step == 5 && blockIdx.x == 0 && x.tid == 0) p.dkb.prof[336 + (k)] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
#define MACX_BSTAMP(k) do { } while (0)
#endif
  MACX_BSTAMP(0);

  // =====================================================================================================================
  // stage B0 (SURVEY appendix A rows "softmax", "logit", "ctrl-mul"):
  //   dl = a (da - sum_j a_j da_j) ; dGd = dl w_k ; dG = dGd * mask / keep ; dZ = dG * act'(I2 * c) ; dI2 = dZ * c
  // elementwise on this lane's slots of the kept I2; the sums over rows (dc, dw_k, db2, db_k) leave as per-row-group partials
  {
    const int q0 = (int)((uint32_t)x.grow0 / (uint32_t)p.N), q1 = (int)(((uint32_t)x.grow0 + x.nvalid - 1) / (uint32_t)p.N);
    const int nq = q1 - q0 + 1;
    float* sRed = x.sPart;              // [QS][8]
    float* sDot = x.sPart + 64;         // [nq <= QS]
    for (int i = x.tid; i < D; i += 512) x.sW[i] = p.wk[i];
    for (int i = x.tid; i < nq * D; i += 512) x.sC[i] = p.c[(size_t)q0 * D + i];        // nq <= QS: the caller guarantees N >= 16
    // row-block mapping (ChainCtx::convert_finish_blk): this lane's slot column and its eight rows
    const int r8 = x.lane & 7, kq = x.lane >> 3;
    const bool active = x.wave < KG / 8;
    const int kg = active ? 8 * x.wave + kq : 0;
    const size_t Rp = p.I2.Rp();
    const size_t ipb = p.I2.plane_bytes();
    const size_t tile = blockIdx.x;
    const bool sums = p.dc_part != nullptr;
    // (no keep bytes: any readable bytes stand in and are overridden -- a branch around a load costs a full wait per load)
    const uint8_t* bytes_src = p.bytes2 ? p.bytes2 : reinterpret_cast<const uint8_t*>(p.I2.base);
    const uint32_t bits_or = p.bytes2 ? 0u : 0xFFu;
    float mx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the kept I2 slots, the keep bytes and the rows' scalars are requested three row blocks ahead of their use (all eight at
    // once would hold 100 registers more than the kernel has).  Rows past the end lie in the tensors' pad rows
    // (H2_PAD_ROWS = 64 = one tile): readable, never used
    u32x4 raw[8][2];
    uint32_t bits[8];
    float dlr[8], inv[8];
    int seg[8];
    const char* src0 = p.I2.base + ((size_t)kg * Rp + x.grow0 + r8) * 16;
    const uint8_t* b0 = bytes_src + (size_t)kg * Rp + x.grow0 + r8;
    const int8_t* e0 = p.I2.exps() + (x.grow0 + r8) * CB + (kg >> 4);
    const uint32_t g0 = (uint32_t)x.grow0 + r8, last = (uint32_t)M - 1, un = (uint32_t)p.N;
    const uint32_t qb1 = (uint32_t)(q0 + 1) * un, qb2 = qb1 + un;       // first rows of the next two questions
    // (round 5) the first three row blocks' loads -- I2 slots, keep bytes, exponent bytes, the rows' att / da -- are requested HERE, in
    // front of the softmax-backward scalars below (two barriers and a round of global loads of their own), so that the tile's HBM
    // burst runs under them; what needs the scalars (dl) is finished in fetch_b.  Only where all eight waves take part (d = 512): a
    // load under `if (active)` would be drained at the join.
    constexpr bool EARLY = (KG / 8) >= 8 && C::SB >= 3;
    int eraw[3];
    float a_r[3], d_r[3];
    auto fetch_a = [&](auto sb_c) __attribute__((always_inline)) {
      constexpr int sb = decltype(sb_c)::value;
      raw[sb][0] = *reinterpret_cast<const u32x4*>(src0 + sb * 128);
      raw[sb][1] = *reinterpret_cast<const u32x4*>(src0 + ipb + sb * 128);
      bits[sb] = (uint32_t)b0[sb * 8];
      eraw[sb] = (int)e0[sb * 8 * CB];
      const uint32_t gr = min(g0 + 8 * sb, last);
      a_r[sb] = p.att[gr];
      d_r[sb] = p.da[gr];
    };
    if constexpr (EARLY) {
      fetch_a(std::integral_constant<int, 0>{});
      fetch_a(std::integral_constant<int, 1>{});
      fetch_a(std::integral_constant<int, 2>{});
    }
    {
      // sum_j a_j da_j of each question the tile touches (every tile of a question: same order, same value).  All questions in
      // one pass: their loads are in flight together and the workgroup meets twice, not twice per question
      float part[C::G::QS];
#pragma unroll
      for (int k = 0; k < C::G::QS; ++k) {
        part[k] = 0.f;
        if (k < nq) {
          const float* aq = p.att + (size_t)(q0 + k) * p.N;
          const float* dq = p.da + (size_t)(q0 + k) * p.N;
          for (int n = x.tid; n < p.N; n += 512) part[k] += aq[n] * dq[n];
        }
      }
#pragma unroll
      for (int k = 0; k < C::G::QS; ++k)
        if (k < nq) {
          const float t = wave_sum(part[k]);
          if (x.lane == 0) sRed[k * 8 + x.wave] = t;
        }
      __syncthreads();
      if (x.tid < nq) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += sRed[x.tid * 8 + w];
        sDot[x.tid] = t;
      }
      __syncthreads();
    }
    auto fetch = [&](auto sb_c) __attribute__((always_inline)) {
      constexpr int sb = decltype(sb_c)::value;
      raw[sb][0] = *reinterpret_cast<const u32x4*>(src0 + sb * 128);
      raw[sb][1] = *reinterpret_cast<const u32x4*>(src0 + ipb + sb * 128);
      bits[sb] = (uint32_t)b0[sb * 8];
      inv[sb] = h2_pow2(-(int)e0[sb * 8 * CB]);
      const uint32_t gr = min(g0 + 8 * sb, last);
      seg[sb] = nq <= 3 ? (int)(gr >= qb1) + (int)(gr >= qb2) : (int)(gr / un) - q0;
      const float dl_row = p.att[gr] * (p.da[gr] - sDot[seg[sb]]);
      dlr[sb] = 8 * sb + r8 < x.nvalid ? dl_row : 0.f;
    };
    auto fetch_b = [&](auto sb_c) __attribute__((always_inline)) {       // the rest of an early block: what needed the scalars
      constexpr int sb = decltype(sb_c)::value;
      if constexpr (sb < 3) {
        inv[sb] = h2_pow2(-eraw[sb]);
        const uint32_t gr = min(g0 + 8 * sb, last);
        seg[sb] = nq <= 3 ? (int)(gr >= qb1) + (int)(gr >= qb2) : (int)(gr / un) - q0;
        dlr[sb] = 8 * sb + r8 < x.nvalid ? a_r[sb] * (d_r[sb] - sDot[seg[sb]]) : 0.f;
      }
    };
    float a_dw[8], a_db[8], a_dc[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a_dw[e] = a_db[e] = a_dc[0][e] = a_dc[1][e] = a_dc[2][e] = 0.f;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(x.sW + kg * 8), w1 = *reinterpret_cast<const f32x4*>(x.sW + kg * 8 + 4);
    auto block = [&](auto act_c, auto sb_c) __attribute__((always_inline)) {
      constexpr int ACTC = decltype(act_c)::value;
      const int ACT = ACTC >= 0 ? ACTC : p.act2;
      constexpr int sb = decltype(sb_c)::value;
      const float* cq = x.sC + (size_t)seg[sb] * D + kg * 8;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(cq), c1 = *reinterpret_cast<const f32x4*>(cq + 4);
      float i2[8];
      h2_join8(raw[sb][0], raw[sb][1], inv[sb], i2);
      const bool rv = 8 * sb + r8 < x.nvalid;
      float o[8];
      float m = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        i2[e] = rv ? i2[e] : 0.f;            // (pad rows hold anything, NaN included: select, do not multiply by zero)
        const float cv = e < 4 ? c0[e & 3] : c1[e & 3], wv = e < 4 ? w0[e & 3] : w1[e & 3];
        const float g = chain_act(ACT, i2[e] * cv);
        const float f = (((bits[sb] | bits_or) >> e) & 1u) ? p.inv2 : 0.f;
        const float dz = (dlr[sb] * wv) * f * act_grad_from_out(ACT, g);
        const float ov = dz * cv;
        o[e] = ov;
        m = fmaxf(m, fabsf(ov));
        a_dw[e] = fmaf(dlr[sb], g * f, a_dw[e]);              // dw_k += dl * dropped(G)
        a_db[e] += ov;
        const float t = dz * i2[e];                            // dc += dZ * I2, per question of the row
        a_dc[0][e] += seg[sb] == 0 ? t : 0.f;
        a_dc[1][e] += seg[sb] == 1 ? t : 0.f;
        a_dc[2][e] += seg[sb] == 2 ? t : 0.f;
      }
      mx[sb] = m;
      x.blk_park(kg, 8 * sb + r8, o);
    };
    auto pass = [&](auto act_c) __attribute__((always_inline)) {
      constexpr int SBc = C::SB;                       // 8-row blocks of the tile: 8, 4 or 2
      // row blocks requested ahead of their use: three.  (Two for the identity activation, whose allocation is 8 - 13 registers over
      // the cap, was tried: 45 spilled registers -- the shorter body lets the scheduler hoist more of the next blocks' arithmetic.)
      constexpr int PFD = 3;
      if constexpr (EARLY) {
        fetch_b(std::integral_constant<int, 0>{});
        fetch_b(std::integral_constant<int, 1>{});
        fetch_b(std::integral_constant<int, 2>{});
      } else {
        fetch(std::integral_constant<int, 0>{});
        if constexpr (SBc > 1) fetch(std::integral_constant<int, 1>{});
        if constexpr (SBc > 2 && PFD > 2) fetch(std::integral_constant<int, 2>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      auto step = [&](auto sb_c) __attribute__((always_inline)) {
        constexpr int sb = decltype(sb_c)::value;
        if constexpr (sb < SBc) {
          if constexpr (sb + PFD < SBc) fetch(std::integral_constant<int, sb + PFD>{});
          block(act_c, sb_c);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    };
    if (active) pass(std::integral_constant<int, A2>{});
    if (sums && active) {
      // the wave owns these columns for the whole tile: sum the eight lanes that share a slot column, one lane stores
      float* dw = p.dwk_part + tile * D + kg * 8;
      float* db = p.db2_part + tile * D + kg * 8;
      float* dc = p.dc_part + tile * 3 * D + kg * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a_dw[e] = row8_sum(a_dw[e]);
        a_db[e] = row8_sum(a_db[e]);
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < nq) a_dc[k][e] = row8_sum(a_dc[k][e]);
      }
      if (r8 == 0) {
        *reinterpret_cast<f32x4*>(dw) = f32x4{a_dw[0], a_dw[1], a_dw[2], a_dw[3]};
        *reinterpret_cast<f32x4*>(dw + 4) = f32x4{a_dw[4], a_dw[5], a_dw[6], a_dw[7]};
        *reinterpret_cast<f32x4*>(db) = f32x4{a_db[0], a_db[1], a_db[2], a_db[3]};
        *reinterpret_cast<f32x4*>(db + 4) = f32x4{a_db[4], a_db[5], a_db[6], a_db[7]};
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < nq) {
            *reinterpret_cast<f32x4*>(dc + k * D) = f32x4{a_dc[k][0], a_dc[k][1], a_dc[k][2], a_dc[k][3]};
            *reinterpret_cast<f32x4*>(dc + k * D + 4) = f32x4{a_dc[k][4], a_dc[k][5], a_dc[k][6], a_dc[k][7]};
          }
      }
    }
    if (sums && x.tid < 64) {          // db_k partials: sum of dl over the tile's rows, per question segment (fixed order)
      const uint32_t gr = min((uint32_t)x.grow0 + x.tid, (uint32_t)M - 1);
      const int sg = (int)(gr / (uint32_t)p.N) - q0;
      const float dl = x.tid < x.nvalid ? p.att[gr] * (p.da[gr] - sDot[sg]) : 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float t = wave_sum(sg == k ? dl : 0.f);
        if (x.tid == 0) p.dls_part[tile * 3 + k] = t;
      }
    }
    x.convert_finish_blk(mx, p.dI2, x.sE);
  }
  MACX_BSTAMP(1);                             // B0 done
  if (p.dbg & 1) return;

  f32x4 acc[RT][CT];
  const size_t tile = blockIdx.x;
  // =====================================================================================================================
  // stage B1: dI1 = (dI2 W2^T) * act'(H1), act' from the kept activation OUTPUT
  x.zero_acc(acc);
  x.template kloop<KV>(acc, p.W2T.planes);
  MACX_BSTAMP(2);                             // B1 product done
  {
    const int eW = *p.W2T.exp;
    const size_t Rp = p.H1.Rp();
    const size_t hpb = p.H1.plane_bytes();
    auto pass = [&](auto act_c) __attribute__((always_inline)) {
      constexpr int ACT = decltype(act_c)::value;
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const size_t gr = min(x.grow0 + x.arow(t), (size_t)M - 1);
        const float s = h2_unscale(x.sE[x.arow(t)], eW);
        const int8_t* ex = p.H1.exps() + gr * CB;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const int kg = x.akg(c);
          const char* src = p.H1.base + ((size_t)kg * Rp + gr) * 16 + x.ahalf();
          const u32x2 hh = *reinterpret_cast<const u32x2*>(src), hl = *reinterpret_cast<const u32x2*>(src + hpb);
          float h[4];
          h2_join4(hh, hl, h2_pow2(-(int)ex[kg >> 4]), h);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[t][c][q] = acc[t][c][q] * s * act_grad_from_out(ACT, h[q]);
        }
      }
    };
    switch (p.act1) {
      case ACT_TANH: pass(std::integral_constant<int, ACT_TANH>{}); break;
      case ACT_SIGMOID: pass(std::integral_constant<int, ACT_SIGMOID>{}); break;
      case ACT_ELU: pass(std::integral_constant<int, ACT_ELU>{}); break;
      case ACT_RELU: pass(std::integral_constant<int, ACT_RELU>{}); break;
      default: pass(std::integral_constant<int, ACT_NON>{}); break;
    }
  }
  x.colsum(acc, p.db1_part + tile * D);
  x.rowmax(acc);
  __syncthreads();                         // every wave is done reading dI2; sMax (and sCol) complete
  x.emit(acc, true, p.dI1);
  x.publish_rows(x.sE, C::PASS_EPI, p.dI1);
  x.colsum_finish(p.db1_part + tile * D);
  __syncthreads();
  MACX_BSTAMP(3);                             // B1 epilogue done
  if (p.dbg & 2) return;

  // =====================================================================================================================
  // stage B2: dX = (dI1 W1a^T) * y + dI1 W1b^T   (ops.py:703: d(x*y)/dx = y per column, per question)
  x.zero_acc(acc);
  x.template kloop<KV>(acc, p.W1aT.planes);
  MACX_BSTAMP(4);                             // B2 first product done
  if (p.dy_part) {
    const int q0 = (int)((uint32_t)x.grow0 / (uint32_t)p.N), q1 = (int)(((uint32_t)x.grow0 + x.nvalid - 1) / (uint32_t)p.N);
    const int nq = q1 - q0 + 1;                          // <= 3: the caller asks for this only when N >= 32
    const uint32_t qb1 = (uint32_t)(q0 + 1) * (uint32_t)p.N, qb2 = qb1 + (uint32_t)p.N;
    const int eW = *p.W1aT.exp;
    const size_t Rp = p.X.Rp();
    const size_t xpb = p.X.plane_bytes();
    float sr[RT];
    int sg[RT];
    size_t gr[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      gr[t] = min(x.grow0 + x.arow(t), (size_t)M - 1);
      sg[t] = (int)((uint32_t)gr[t] >= qb1) + (int)((uint32_t)gr[t] >= qb2);
      sr[t] = h2_unscale(x.sE[x.arow(t)], eW);
    }
    float* stage = x.sW;                                 // [NWR][3][D] when the rows are split over two wave groups (sW and sC are idle)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int kg = x.akg(c);
      float xv[RT][4];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const char* src = p.X.base + ((size_t)kg * Rp + gr[t]) * 16 + x.ahalf();
        const u32x2 hh = *reinterpret_cast<const u32x2*>(src), hl = *reinterpret_cast<const u32x2*>(src + xpb);
        h2_join4(hh, hl, h2_pow2(-(int)p.X.exps()[gr[t] * CB + (kg >> 4)]), xv[t]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float s3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          // (rows past the end read the last row again: select, their accumulators are zero but X need not be finite there)
          const float v = x.arow(t) < x.nvalid ? acc[t][c][q] * sr[t] * xv[t][q] : 0.f;
          s3[0] += sg[t] == 0 ? v : 0.f;
          s3[1] += sg[t] == 1 ? v : 0.f;
          s3[2] += sg[t] == 2 ? v : 0.f;
        }
        const int col = x.acol(c) + q;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < nq) {
            const float sum = x.colred_sum(s3[k]);
            if (x.col_writer()) {
              if (C::NWR == 1) p.dy_part[(tile * 3 + k) * D + col] = sum;
              else stage[(x.wr * 3 + k) * D + col] = sum;
            }
          }
      }
    }
  }
  {
    const int de = *p.W1bT.exp - *p.W1aT.exp;          // units 2^-(e + e1a) -> 2^-(e + e1b), exactly
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const float* yq = p.y + x.qrow(x.grow0 + x.arow(t)) * D;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(yq + x.acol(c));
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][c][q] = ldexpf(acc[t][c][q] * yv[q], de);
      }
    }
  }
  MACX_BSTAMP(5);                             // dy sums and the y scaling done
  x.template kloop<KV>(acc, p.W1bT.planes);
  MACX_BSTAMP(6);                             // B2 second product done
  {
    const int eW = *p.W1bT.exp;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const float s = h2_unscale(x.sE[x.arow(t)], eW);
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[t][c] *= s;
    }
  }
  x.colsum(acc, p.dbx_part + tile * D);
  x.rowmax(acc);
  __syncthreads();
  x.emit(acc, false, p.dX);
  x.publish_rows(x.sE2, C::PASS_EPI, p.dX);
  x.colsum_finish(p.dbx_part + tile * D);
  MACX_BSTAMP(7);                             // dX emitted
  if (C::NWR == 2 && p.dy_part) {                       // the two row halves of the dy partials (staged before the last product)
    const int nq = (int)(((uint32_t)x.grow0 + x.nvalid - 1) / (uint32_t)p.N) - (int)((uint32_t)x.grow0 / (uint32_t)p.N) + 1;
    for (int i = x.tid; i < nq * D; i += 512) p.dy_part[tile * 3 * D + i] = x.sW[i] + x.sW[3 * D + i];
  }
}

template <int D_, int KV, int A2, int R_ = 64>
inline hipError_t chain_bwd_launch_a(const ChainBwdP& p, hipStream_t st) {
  auto kern = chain_bwd_kernel<D_, KV, A2, R_>;
  constexpr size_t lds = ChainGeo<D_, R_>::LDS;
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  const int nfill = (D_ == 512 && R_ == 64 && p.dkb.njobs > 0) ? p.dkb.nfill : 0;
  hipLaunchKernelGGL(kern, dim3((p.M + R_ - 1) / R_ + nfill), dim3(512), lds, st, p);
  return hipGetLastError();
}

// One kernel per readCtrlAct (round 5).  The run-time variant (A2 = -1: a per-value switch inside stage B0's 64-value body) met
// every activation's code in one register allocation and spilled 102-104 VGPRs at every width -- `--relu STD`, TANH and NON
// option sets ran a quarter of their step through scratch.  It is no longer instantiated.
template <int D_, int KV = 0, int R_ = 64>
inline hipError_t chain_bwd_launch_t(const ChainBwdP& p, hipStream_t st) {
  switch (p.act2) {
    case ACT_ELU: return chain_bwd_launch_a<D_, KV, ACT_ELU, R_>(p, st);
    case ACT_RELU: return chain_bwd_launch_a<D_, KV, ACT_RELU, R_>(p, st);
    case ACT_TANH: return chain_bwd_launch_a<D_, KV, ACT_TANH, R_>(p, st);
    case ACT_SIGMOID: return chain_bwd_launch_a<D_, KV, ACT_SIGMOID, R_>(p, st);
    default: return chain_bwd_launch_a<D_, KV, ACT_NON, R_>(p, st);
  }
}

#ifdef MACX_CHAIN_BWD_TU      // macx_chain_bwd.hip: the one translation unit that instantiates the backward kernels
hipError_t chain_dkb_rest_launch(const ChainDkbP& q, int M, int N, int d, hipStream_t st) {
  if (d != 512 || chain_tile_rows(d, (size_t)M) != 64) return hipErrorInvalidValue;
  auto kern = chain_dkb_rest_kernel<512, CHAIN_KV_DEFAULT>;
  constexpr size_t lds = ChainGeo<512, 64>::LDS;
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((M + 63) / 64), dim3(512), lds, st, q, M, N);
  return hipGetLastError();
}

hipError_t chain_bwd_launch(const ChainBwdP& p, hipStream_t st) {
  switch (p.d / 128) {
    case 1: return chain_bwd_launch_t<128>(p, st);
    case 2: return chain_bwd_launch_t<256>(p, st);
    case 3: return chain_bwd_launch_t<384>(p, st);
    case 4:
      switch (chain_tile_rows(p.d, (size_t)p.M)) {
        case 16: return chain_bwd_launch_t<512, 0, 16>(p, st);
        case 32: return chain_bwd_launch_t<512, 0, 32>(p, st);
        default: break;
      }
      // the A/B and measurement variants of the K loop exist for the published configurations' activation (ELU) only
      if (p.act2 != ACT_ELU) return chain_bwd_launch_t<512, CHAIN_KV_DEFAULT>(p, st);
#ifdef MACX_PROFILE_VARIANTS
      switch (p.dbg >> 3) {
        case 1: return chain_bwd_launch_a<512, 1, ACT_ELU>(p, st);
        case 3: return chain_bwd_launch_a<512, 3, ACT_ELU>(p, st);
        default: break;
      }
#endif
      return chain_kv() == 0 ? chain_bwd_launch_a<512, 0, ACT_ELU>(p, st) : chain_bwd_launch_a<512, CHAIN_KV_DEFAULT, ACT_ELU>(p, st);
    default: return hipErrorInvalidValue;
  }
}
#endif


}  // namespace macx
