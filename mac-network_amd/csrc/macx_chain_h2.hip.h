// macx_chain_h2.hip.h -- the read unit's three knowledge-base products as ONE kernel (mac_cell.py:230-266, ops.py:668-725):
//
//     KBd = dropout(KB)                         ops.py:678
//     X   = KBd Wx + bx                         ops.py:688            (stage 1)
//     H1  = act([X*y, X] W1 + b1)               ops.py:703,718        (stage 2: X W1b, then (X*y) W1a on the same accumulators)
//     I2  = H1 W2 + b2                          ops.py:326            (stage 3)
//     l   = dropout(act(I2 * c)) . w_k          mac_cell.py:248-266
//
// A workgroup owns R = 16 NT consecutive rows of the [B*N, d] activation and ALL d columns of every stage, so a stage's output
// never leaves the CU before it is the next stage's operand: it is written in place, as H2 planes, over the LDS tile the
// stage just multiplied.  X, H1 and I2 still go to HBM once each (the backward pass reads them), as stores that drain behind
// the next stage's matrix work; nothing is read back.  Against the four launches this replaces (h2_from_f32 + three
// kb_gemm_h2) a step saves the HBM round trips of KBd, X and H1, three launch ramps and three store tails.
//
// How the pieces map onto a CU (8 waves, one workgroup per CU at d = 512: 128 KB of LDS):
//   * rows are taken from the flat [B*N] axis, not per question: the question enters a row only through y (the A side, below),
//     c and the dropout index, all of which are looked up per row -- so tiles never pad a question (B*N = 12544 = 196 x 64).
//   * wave w owns columns [w d/8, (w+1) d/8) of the output for all R rows.  Weight fragments are therefore private to a wave:
//     they go from L2 straight to registers in MFMA operand order (pack format 3 stores a lane's 8 k-values of one column as
//     one 16-byte slot, 16 columns = 256 contiguous bytes), one K slice ahead of use, and never touch LDS.  The activation
//     tile is shared by all waves and is LDS-resident for the whole stage: the K loop has NO barrier.
//   * operands are swapped on the matrix pipe (D^T = W^T A^T): the weight slot is the MFMA's A operand and the activation
//     slot its B operand -- the same two fragments -- so lane (i, g) ends up with FOUR CONSECUTIVE COLUMNS 4g..4g+3 of row
//     i instead of four rows of one column.  Bias, activation, row maxima, the logit dot product and the hi/lo split all run
//     on the accumulators where they lie, and half a slot (4 columns x fp16) leaves as one 8-byte store: no transpose through
//     LDS, no row pass.
//   * one exponent per ROW (all d columns) instead of per (row, 128 columns): the workgroup sees the whole row, so a stage needs
//     no per-K-block fold and no second accumulator set; the same exponent is written to each of the row's d/128 exponent
//     bytes, which keeps the H2 tensors readable by every other kernel (macx_h2.hip.h).
//   * y enters on the A side: stage 2 first accumulates X W1b, then the tile is rewritten in place as split(X * y_q) (fp32
//     product, rounded once like the reference's X*y, own row exponents), the accumulators are brought to the new unit by an
//     exact power of two, and (X*y) W1a is added -- K = 2d like the reference's concat, no per-question weight mixing.
#pragma once
#include "macx_h2.hip.h"

namespace macx {

struct ChainW {          // a weight matrix in pack format 3 (macx_h2.hip.h: pack_h2_weight)
  const char* planes;    // [K/32][2][4][Nout] x 16 B
  const int* exp;        // device int: the stored fp16 are W * 2^exp
};

struct ChainFwdP {
  int M, N, d;              // rows (B*N), rows per question, width
  int mode;                 // 0: KB -> X -> H1 -> I2 ; 1: X is read back (no read dropout: the projected KB is step-invariant)
  int dbg;                  // timing knobs: 1 stop after stage 0, 2 after stage 1, 4 after stage 2 (results incomplete)
  // stage 0
  const float* kb;          // [M][d] fp32, row-major
  uint32_t first;           // flat dropout index of element (0, 0): b0 * N * d
  uint32_t key1, thr1; float inv1;      // ops.py:678 site (thr = 1 << 24: keep everything)
  uint8_t* bits1;           // its keep bits, row-major, one byte per 8 columns [M][d/8] (= uint32 words [M][d/32]); may be null
  uint32_t key2, thr2; float inv2;      // ops.py:312 site on act(I2 * c)
  uint8_t* bytes2;          // its keep bits in slot order [d/8][M + pad]; may be null
  H2View KBd; int* qmin_KBd;            // base null: not written
  ChainW Wx, W1a, W1b, W2;
  const float *bx, *b1, *b2;
  int act1, act2;           // readMemAct (on H1), readCtrlAct (on I2 * c)
  const float* y;           // [B][d]
  const float* c;           // [B][d]
  const float* wk;          // [d]
  H2View X; int* qmin_X;    // written in mode 0, read in mode 1
  H2View H1; int* qmin_H1;  // base null: not written (inference)
  H2View I2;                // base null: not written
  float* logits;            // [M] (without the bias b_k, which kb_attend adds)
};

template <int NT, int CT>
constexpr size_t chain_fwd_lds_bytes() { return (size_t)4 * (16 * NT) * (128 * CT) + (size_t)(16 * NT) * (8 + 8 + 1 + 1) * 4; }

// KV: measurement variants of the K loop (0 = the product): 1 no MFMA (the loads stay), 2 no weight loads inside the loop,
// 3 no loads at all inside the loop -- timing only, results are wrong
template <int NT, int CT, int KV = 0>
__global__ __launch_bounds__(512) void chain_fwd_kernel(const ChainFwdP p) {
  constexpr int R = 16 * NT, D = 128 * CT, KG = D / 8, KT = D / 32, CB = D / 128;
  constexpr int UPR = KG / 4;            // conversion units (16 rows x 4 slot columns) per row group
  constexpr int IT = NT * UPR / 8;       // ... per wave
  constexpr int WPR = 8 / NT;            // waves that share a row group in the conversion passes
  static_assert(NT == 1 || NT == 2 || NT == 4, "row tiles per workgroup");
  static_assert((NT * UPR) % 8 == 0 && UPR % IT == 0 && KT % 2 == 0, "conversion units divide among the waves");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const P = lds;                                             // [2 planes][KG][R] x 16 B: the stage's A operand
  float* const sMax = reinterpret_cast<float*>(lds + (size_t)4 * R * D);   // [8][R] partial row maxima
  float* const sPart = sMax + 8 * R;                                       // [8][R] partial attention logits
  int* const sE = reinterpret_cast<int*>(sPart + 8 * R);                   // [R] exponents of the rows in P
  int* const sE2 = sE + R;                                                 // [R] ... of the X*y tile

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int M = p.M;
  const size_t grow0 = (size_t)blockIdx.x * R;
  const int nvalid = (int)min((size_t)R, (size_t)M - grow0);

  // ---- conversion passes: lane (li, lg) of wave w handles row crow and slot columns ckg(j), j < IT
  const int crow = (wave / WPR) * 16 + li;
  const size_t cgrow = grow0 + crow;
  const bool cvalid = crow < nvalid;
  auto ckg = [&](int j) { return ((wave % WPR) * IT + j) * 4 + lg; };

  // per-row bookkeeping after a pass produced partial maxima in sMax[w0 .. w0+nw)[R]: exponent table, exponent bytes, qmin
  auto row_exponent = [&](int row, int w0, int nw) {
    float m = sMax[w0 * R + row];
    for (int w = 1; w < nw; ++w) m = fmaxf(m, sMax[(w0 + w) * R + row]);
    return h2_exponent(m);
  };
  auto publish_rows = [&](int* eTab, int wpr, const H2View& out, int* qmin) {     // called by every thread; wave 0 works
    if (tid < 64) {
      const int r = tid;
      const bool v = r < nvalid;
      int e = 127;
      if (r < R) {
        e = row_exponent(r, wpr == 8 ? 0 : (r >> 4) * wpr, wpr);
        eTab[r] = e;
        if (out.base && v) {
          int8_t* ex = out.exps() + (grow0 + r) * CB;
#pragma unroll
          for (int k = 0; k < CB; ++k) ex[k] = (int8_t)e;
        }
      }
      if (out.base && qmin) {
        const int q0 = (int)(grow0 / p.N), q1 = (int)((grow0 + nvalid - 1) / p.N);
        const int qr = v ? (int)((grow0 + r) / p.N) : -1;
        for (int qq = q0; qq <= q1; ++qq) {
          int mn = (qr == qq) ? e : 127;
#pragma unroll
          for (int s = 32; s > 0; s >>= 1) mn = min(mn, __shfl_xor(mn, s, 64));
          if (tid < CB) atomicMin(qmin + (size_t)qq * CB + tid, mn);
        }
      }
    }
  };
  // v[j][0..7] (fp32, this lane's slots) -> row exponents -> H2 slots in P (+ HBM)
  auto convert_finish = [&](float (&v)[IT][8], float m, int* eTab, const H2View& out, int* qmin) {
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (lane < 16) sMax[wave * R + crow] = m;
    __syncthreads();
    const int e = row_exponent(crow, (wave / WPR) * WPR, WPR);
    const float s = h2_pow2(e);
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
    publish_rows(eTab, WPR, out, qmin);
    __syncthreads();
  };

  // =====================================================================================================================
  // stage 0: the operand of the first product
  if (p.mode == 0) {
    float v[IT][8];
    float m = 0.f;
    const bool drop1 = p.thr1 < (1u << 24), drop2 = p.thr2 < (1u << 24);
    const size_t Rp2 = (size_t)M + H2_PAD_ROWS;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int kg = ckg(j);
      // (rows past the end read the last row and are zeroed: a branch around a load costs a full wait per load)
      const float* src = p.kb + min(cgrow, (size_t)M - 1) * D + kg * 8;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[j][q] = cvalid ? a0[q] : 0.f; v[j][4 + q] = cvalid ? a1[q] : 0.f; }
    }
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int kg = ckg(j);
      const uint32_t e0 = p.first + (uint32_t)(cgrow * D + kg * 8);
      if (drop1) {
        uint32_t byte = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const bool keep = keep_bit(e0 + q, p.key1, p.thr1);
          byte |= (keep ? 1u : 0u) << q;
          v[j][q] = keep ? v[j][q] * p.inv1 : 0.f;
        }
        if (p.bits1 && cvalid) p.bits1[cgrow * KG + kg] = (uint8_t)byte;
      }
      if (drop2 && p.bytes2 && cvalid) {
        uint32_t byte = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) byte |= (keep_bit(e0 + q, p.key2, p.thr2) ? 1u : 0u) << q;
        p.bytes2[(size_t)kg * Rp2 + cgrow] = (uint8_t)byte;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) m = fmaxf(m, fabsf(v[j][q]));
    }
    convert_finish(v, m, sE, p.KBd, p.qmin_KBd);
  } else {
    // X as an earlier call left it: a pure copy, lanes along rows
    const size_t Rp = p.X.Rp();
    for (int f = tid; f < 2 * KG * R; f += 512) {
      const int r = f % R, s = f / R;                       // s = plane * KG + kg
      const size_t gr = min(grow0 + r, (size_t)M - 1);
      *reinterpret_cast<u32x4*>(P + (size_t)f * 16) = *reinterpret_cast<const u32x4*>(p.X.base + ((size_t)s * Rp + gr) * 16);
    }
    if (tid < R) sE[tid] = (int)p.X.exps()[min(grow0 + tid, (size_t)M - 1) * CB];
    __syncthreads();
  }
  if (p.dbg & 1) return;

  // =====================================================================================================================
  // the K loop of one product: acc[t][c] += W^T-slot x A-slot over all d (three fp16 terms, smallest first)
  f32x4 acc[NT][CT];
  const int colbase = wave * (16 * CT);
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  auto kloop = [&](const char* W) {
    const char* wb = W + ((size_t)lg * D + colbase + li) * 16;
    const char* pa = P + ((size_t)lg * R + li) * 16;
    // both operands of slice kt + 1 are requested before slice kt is multiplied: the weight slots from L2 (about one slice of
    // matrix work away), the activation slots from LDS (whose latency would otherwise be paid once per row tile)
    u32x4 bq[2][2][CT], aq[2][2][NT];
    auto load_ab = [&](auto set_c, int kt, bool in_loop) {
      constexpr int S = decltype(set_c)::value;
      if (!(KV >= 2 && in_loop)) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int c = 0; c < CT; ++c)
            bq[S][pl][c] = *reinterpret_cast<const u32x4*>(wb + ((size_t)(kt * 2 + pl) * 4 * D) * 16 + c * 256);
      }
      if (!(KV >= 3 && in_loop)) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            aq[S][pl][t] = *reinterpret_cast<const u32x4*>(pa + ((size_t)(pl * KG + kt * 4) * R + t * 16) * 16);
      }
    };
    auto mm = [&](auto set_c) {
      constexpr int S = decltype(set_c)::value;
      if (KV == 1) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int c = 0; c < CT; ++c) asm volatile("" ::"v"(bq[S][pl][c]));
#pragma unroll
          for (int t = 0; t < NT; ++t) asm volatile("" ::"v"(aq[S][pl][t]));
        }
        return;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[t][c] = mfma_f16(bq[S][0][c], aq[S][1][t], acc[t][c]);     // w_hi a_lo
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[t][c] = mfma_f16(bq[S][1][c], aq[S][0][t], acc[t][c]);     // w_lo a_hi
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[t][c] = mfma_f16(bq[S][0][c], aq[S][0][t], acc[t][c]);     // w_hi a_hi
      }
    };
    // the scheduler would sink a slice's loads to the end of the previous slice's products (shortest live range), i.e. to where
    // they are needed: the order of "request slice kt + 1, multiply slice kt" is pinned
    load_ab(S0{}, 0, false);
    if (KV >= 2) load_ab(S1{}, 1, false);
#pragma unroll 1
    for (int kt = 0; kt < KT; kt += 2) {
      load_ab(S1{}, kt + 1, true);
      __builtin_amdgcn_sched_barrier(0);
      mm(S0{});
      __builtin_amdgcn_sched_barrier(0);
      load_ab(S0{}, min(kt + 2, KT - 2), true);    // (unconditional: behind a branch the wait-count pass drains every load at the join)
      __builtin_amdgcn_sched_barrier(0);
      mm(S1{});
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue on the accumulators: lane (li, lg) holds, for row tile t and column tile c, row 16 t + li and the four
  //      columns colbase + 16 c + 4 lg .. + 3
  // x = act(acc * 2^-(eA[row] + eW) + bias); returns with x in acc and the partial row maxima in sMax[wave]
  auto bias_act = [&](auto act_c, const int* eTab, int eW, const float* bias) {
    constexpr int ACT = decltype(act_c)::value;
    f32x4 bv[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bv[c] = *reinterpret_cast<const f32x4*>(bias + colbase + 16 * c + 4 * lg);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float s = h2_unscale(eTab[16 * t + li], eW);
      float m = 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x = act_apply(ACT, fmaf(acc[t][c][q], s, bv[c][q]));
          acc[t][c][q] = x;
          m = fmaxf(m, fabsf(x));
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      if (lg == 0) sMax[wave * R + 16 * t + li] = m;
    }
  };
  auto bias_act_any = [&](int act, const int* eTab, int eW, const float* bias) {
    switch (act) {     // ONE switch per stage, not one per value (macx_gemm_h2.hip.h has the story)
      case ACT_TANH: bias_act(std::integral_constant<int, ACT_TANH>{}, eTab, eW, bias); break;
      case ACT_SIGMOID: bias_act(std::integral_constant<int, ACT_SIGMOID>{}, eTab, eW, bias); break;
      case ACT_ELU: bias_act(std::integral_constant<int, ACT_ELU>{}, eTab, eW, bias); break;
      case ACT_RELU: bias_act(std::integral_constant<int, ACT_RELU>{}, eTab, eW, bias); break;
      default: bias_act(std::integral_constant<int, ACT_NON>{}, eTab, eW, bias); break;
    }
  };
  // after the barrier that completed sMax: split the accumulators, write them over P (to_p) and to `out`
  auto emit = [&](bool to_p, const H2View& out) {
    const size_t Rp = out.Rp();
    const size_t opb = out.plane_bytes();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int row = 16 * t + li;
      const float s = h2_pow2(row_exponent(row, 0, 8));
      const bool st = out.base && row < nvalid;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const f32x4 x = acc[t][c] * s;
        u32x2 hi, lo;
        hi[0] = pk_f16(x[0], x[1]);
        hi[1] = pk_f16(x[2], x[3]);
        const f32x2_t b0 = unpk_f16(hi[0]), b1 = unpk_f16(hi[1]);
        lo[0] = pk_f16(x[0] - b0[0], x[1] - b0[1]);
        lo[1] = pk_f16(x[2] - b1[0], x[3] - b1[1]);
        const int kg = (colbase >> 3) + 2 * c + (lg >> 1);
        const int half = (lg & 1) * 8;
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
  };

  // =====================================================================================================================
  // stage 1: X = KBd Wx + bx
  if (p.mode == 0) {
    zero_acc();
    kloop(p.Wx.planes);
    bias_act(std::integral_constant<int, ACT_NON>{}, sE, *p.Wx.exp, p.bx);
    __syncthreads();                       // every wave is done reading P; sMax is complete
    emit(true, p.X);
    publish_rows(sE, 8, p.X, p.qmin_X);
    __syncthreads();
  }
  if (p.dbg & 2) return;

  // =====================================================================================================================
  // stage 2: H1 = act(X W1b + (X * y) W1a + b1)
  zero_acc();
  kloop(p.W1b.planes);
  __syncthreads();                         // every wave is done reading X
  {
    float v[IT][8];
    float m = 0.f;
    const float inv = h2_pow2(-sE[crow]);
    const float* yq = p.y + (min(cgrow, (size_t)M - 1) / p.N) * D;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int kg = ckg(j);
      const char* s = P + ((size_t)kg * R + crow) * 16;
      const u32x4 hi = *reinterpret_cast<const u32x4*>(s), lo = *reinterpret_cast<const u32x4*>(s + (size_t)KG * R * 16);
      float x[8];
      h2_join8(hi, lo, inv, x);
      const f32x4 y0 = *reinterpret_cast<const f32x4*>(yq + kg * 8), y1 = *reinterpret_cast<const f32x4*>(yq + kg * 8 + 4);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        v[j][q] = x[q] * (q < 4 ? y0[q & 3] : y1[q & 3]);      // ops.py:703: the product the reference rounds to fp32
        m = fmaxf(m, fabsf(v[j][q]));
      }
    }
    convert_finish(v, m, sE2, H2View{nullptr, M, D}, nullptr);
  }
  {
    // accumulators: units 2^-(eX + e1b)  ->  2^-(eXy + e1a), exactly
    const int de = *p.W1a.exp - *p.W1b.exp;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int k = sE2[16 * t + li] - sE[16 * t + li] + de;
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][c][q] = ldexpf(acc[t][c][q], k);
    }
  }
  kloop(p.W1a.planes);
  bias_act_any(p.act1, sE2, *p.W1a.exp, p.b1);
  __syncthreads();
  emit(true, p.H1);
  publish_rows(sE, 8, p.H1, p.qmin_H1);
  __syncthreads();
  if (p.dbg & 4) return;

  // =====================================================================================================================
  // stage 3: I2 = H1 W2 + b2 ; logits = dropout(act(I2 * c)) . w_k
  zero_acc();
  kloop(p.W2.planes);
  bias_act(std::integral_constant<int, ACT_NON>{}, sE, *p.W2.exp, p.b2);
  {
    const bool drop2 = p.thr2 < (1u << 24);
    auto logit_pass = [&](auto act_c) {
      constexpr int ACT = decltype(act_c)::value;
      f32x4 wv[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) wv[c] = *reinterpret_cast<const f32x4*>(p.wk + colbase + 16 * c + 4 * lg);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const size_t gr = min(grow0 + 16 * t + li, (size_t)M - 1);
        const float* cq = p.c + (gr / p.N) * D + colbase + 4 * lg;
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const f32x4 cv = *reinterpret_cast<const f32x4*>(cq + 16 * c);
          const uint32_t e0 = p.first + (uint32_t)(gr * D + colbase + 16 * c + 4 * lg);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float g = act_apply(ACT, acc[t][c][q] * cv[q]);
            if (drop2) g = keep_bit(e0 + q, p.key2, p.thr2) ? g * p.inv2 : 0.f;
            part = fmaf(g, wv[c][q], part);
          }
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (lg == 0) sPart[wave * R + 16 * t + li] = part;
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
  emit(false, p.I2);
  publish_rows(sE2, 8, p.I2, nullptr);
  if (tid < nvalid) {
    float s = sPart[tid];
#pragma unroll
    for (int w = 1; w < 8; ++w) s += sPart[w * R + tid];          // fixed order
    p.logits[grow0 + tid] = s;
  }
}

template <int NT, int CT, int KV = 0>
inline hipError_t chain_fwd_launch_t(const ChainFwdP& p, hipStream_t st) {
  auto kern = chain_fwd_kernel<NT, CT, KV>;
  constexpr size_t lds = chain_fwd_lds_bytes<NT, CT>();
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);
  if (e != hipSuccess) return e;
  const int R = 16 * NT;
  hipLaunchKernelGGL(kern, dim3((p.M + R - 1) / R), dim3(512), lds, st, p);
  return hipGetLastError();
}

inline bool chain_fwd_supported(int d) { return d % 128 == 0 && d >= 128 && d <= 512; }

inline hipError_t chain_fwd_launch(const ChainFwdP& p, hipStream_t st) {
  switch (p.d / 128) {
    case 1: return chain_fwd_launch_t<4, 1>(p, st);
    case 2: return chain_fwd_launch_t<4, 2>(p, st);
    case 3: return chain_fwd_launch_t<4, 3>(p, st);
    case 4:
      switch (p.dbg >> 3) {
        case 1: return chain_fwd_launch_t<4, 4, 1>(p, st);
        case 2: return chain_fwd_launch_t<4, 4, 2>(p, st);
        case 3: return chain_fwd_launch_t<4, 4, 3>(p, st);
        default: return chain_fwd_launch_t<4, 4>(p, st);
      }
    default: return hipErrorInvalidValue;
  }
}

}  // namespace macx
