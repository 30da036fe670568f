// macx_chain_api.hip.h -- what the host side needs of the read unit's chain kernels (macx_chain_h2.hip.h): the argument structs, the
// tile geometry and the two launchers.  The kernels themselves are compiled in translation units of their own
// (macx_chain_fwd.hip, macx_chain_bwd.hip: two thirds of the library's compile time, built in parallel with the rest).
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
  int dbg;                  // timing knobs: 1 stop after stage 0, 2 after stage 1, 4 after stage 2; dbg >> 3 = K-loop variant
  // stage 0
  const float* kb;          // [M][d] fp32, row-major
  uint32_t first;           // flat dropout index of element (0, 0): b0 * N * dlog
  int dlog;                 // row stride of the dropout index: the logical width of a zero-padded cell (macx_shapes.d_logical), else d
  uint32_t key1, thr1; float inv1;      // ops.py:678 site (thr = 1 << 24: keep everything)
  uint8_t* bits1;           // its keep bits, row-major, one byte per 8 columns [M][d/8] (= uint32 words [M][d/32]); may be null
  uint32_t key2, thr2; float inv2;      // ops.py:312 site on act(I2 * c)
  const uint32_t* word;                 // macx_dropout.mask_word (device, may be null): XORed into both keys when the kernel runs
  uint8_t* bytes2;          // its keep bits in slot order [d/8][M + pad]; may be null
  H2View KBd;               // base null: not written
  ChainW Wx, W1a, W1b, W2;
  const float *bx, *b1, *b2;
  int act1, act2;           // readMemAct (on H1), readCtrlAct (on I2 * c)
  const float* y;           // [B][d]
  const float* c;           // [B][d]
  const float* wk;          // [d]
  H2View X;                 // written in mode 0, read in mode 1
  H2View H1;                // base null: not written (inference)
  H2View I2;                // base null: not written
  float* logits;            // [M] (without the bias b_k, which kb_attend adds)
};

struct ChainBwdP {
  int M, N, d;
  int dbg;                  // timing knobs: 1 stop after stage B0, 2 after B1; dbg >> 3 = K-loop variant
                            // (tried: pulling the kept H1 rows toward L2 by DMA from inside the K loop of stage B1, whose
                            // epilogue reads them -- no change, 4.399 vs 4.400 ms per step)
  // stage B0: dI2 from the kept I2
  const float* att;         // [B][N] knowledge-base attention of the step
  const float* da;          // [B][N] dinfo . KB[n] (kb_att_da_kernel)
  H2View I2;
  const float* c;           // [B][d]
  const float* wk;          // [d]
  int act2;                 // readCtrlAct
  const uint8_t* bytes2;    // keep bits of the attention dropout in slot order [d/8][M + pad]; null = keep all
  float inv2;
  H2View dI2;
  // column sums of stage B0 per 64-row tile, summed over tiles by the caller in a fixed order; a tile may touch up to three
  // questions (N >= 32), so what is per question has three segments.  All four null: not computed here
  // (read_att_bwd_h2_kernel in its sums-only mode does it).
  float* dwk_part;          // [tiles][d]     sum_rows dl * dropped(act(I2 * c))
  float* db2_part;          // [tiles][d]     sum_rows dI2
  float* dc_part;           // [tiles][3][d]  sum_rows dZ * I2, per question segment
  float* dls_part;          // [tiles][3]     sum_rows dl, per question segment
  // stage B1: dI1 = (dI2 W2^T) * act'(H1)
  ChainW W2T;
  H2View H1; int act1;      // readMemAct
  H2View dI1;
  float* db1_part;          // [tiles][d] column sums of dI1
  // stage B2: dX = (dI1 W1a^T) * y + dI1 W1b^T
  ChainW W1aT, W1bT;
  const float* y;           // [B][d]
  H2View dX;
  float* dbx_part;          // [tiles][d] column sums of dX
  // dy[q][k] = sum over the question's rows of (dI1 W1a^T)[r][k] * X[r][k]  (ops.py:703: d(x * y)/dy = x): the first product of
  // stage B2 is exactly the left factor, so dy costs one read of the kept X tile -- and the per-question contraction
  // S_b = X_b^T dI1_b that used to deliver it (sb_h2_kernel) leaves the recurrence and runs once, over all steps, at the end
  H2View X;                 // the step's kept X
  float* dy_part;           // [tiles][3][d] per question segment, like dc_part; null: not computed
};

// K-loop variant of the d = 512, 64-row chain kernels (ChainCtx::kloop: 4 = activation reads in mid-slice); -1 = the default,
// 0 = reads in front of the slice.  macx_opts.tune[MACX_TUNE_CHAIN_KV]: the A/B hook of that decision
constexpr int CHAIN_KV_DEFAULT = 4;
inline int chain_kv() { return tune_get(MACX_TUNE_CHAIN_KV, -1); }
inline bool chain_supported(int d, int N) { return d % 128 == 0 && d >= 128 && d <= 512 && N >= 16; }
// rows per tile: the tallest tile that still leaves the chip one tile per CU (short tiles exist for d = 512 only)
inline int chain_tile_rows(int d, size_t M) {
  if (d != 512) return 64;
  if ((M + 31) / 32 > 256) return 64;
  return (M + 15) / 16 > 256 ? 32 : 16;
}
inline int chain_tile_shift(int d, size_t M) { const int r = chain_tile_rows(d, M); return r == 64 ? 6 : (r == 32 ? 5 : 4); }
inline size_t chain_tiles(int d, size_t M) { const size_t r = (size_t)chain_tile_rows(d, M); return (M + r - 1) / r; }

// defined in macx_chain_fwd.hip / macx_chain_bwd.hip
hipError_t chain_fwd_launch(const ChainFwdP& p, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);   // e0 / e1: the kernel's own start / stop timestamps
hipError_t chain_bwd_launch(const ChainBwdP& p, hipStream_t st);

}  // namespace macx
