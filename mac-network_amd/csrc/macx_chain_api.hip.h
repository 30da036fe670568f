// macx_chain_api.hip.h -- what the host side needs of the read unit's chain kernels (macx_chain_h2.hip.h): the argument structs, the
// tile geometry and the two launchers.  The kernels themselves are compiled in translation units of their own
// (macx_chain_fwd.hip, macx_chain_bwd.hip: two thirds of the library's compile time, built in parallel with the rest).
#pragma once
#include "macx_h2.hip.h"
#include "macx_lin_tile.hip.h"

namespace macx {

struct ChainW {          // a weight matrix in pack format 3 (macx_h2.hip.h: pack_h2_weight)
  const char* planes;    // [K/32][2][4][Nout] x 16 B
  const int* exp;        // device int: the stored fp16 are W * 2^exp
};

// Stage 0 (dropout(KB) -> H2, both keep-bit sites) of the NEXT step on the CUs a d = 512 chain_fwd launch leaves idle: it depends on
// no state, only on the step's site keys.  The launch of step i carries `nfill` extra workgroups that walk the tiles of step i + 1
// (tile f, f + nfill, ..); the launch of step i + 1 then starts from the finished planes (ChainFwdP::mode 2).
// The same workgroups first compute THIS step's y = md Wy + by (the [B,d] linear that would otherwise be a launch of its own in front
// of the chain launch): tiles of small_linear_kernel's arithmetic (macx_lin_tile.hip.h), outputs written through the L2 (agent-scope
// stores), then one increment of `yflag` per filler workgroup.  The chain tiles need y only after their second product (~45 us
// into the launch); they wait for yflag == nfill (bounded: `fail` is set if a tile gives up) and read y with agent-scope loads --
// the per-XCD L2s are not coherent inside a launch.  Fillers carry the LOWEST workgroup ids: they are dispatched before any tile,
// so a tile never waits for a workgroup that is not running.
struct ChainPreP {
  int nfill;                // filler workgroups; 0: none
  uint32_t key1, key2;      // the next step's site keys (thresholds and scales are the step's own: one keep probability per run)
  uint8_t* bits1;           // its outputs, as ChainFwdP's
  uint8_t* bytes2;
  H2View KBd;               // base null: no stage 0 to do (the last step)
  LinP ylin;                // n_out = 0: y was computed by a launch
  uint32_t* yflag;          // device word, zero when the launch starts
  uint32_t* fail;           // device word, sticky
  // ... and, in front of y, the PREVIOUS step's write unit (wlin.n_out = 0: a launch did it): the [B,d] linears between two chain
  // launches then run beside the tiles' first two stages instead of between the launches.  Filler-to-filler order per block of 16
  // questions: its write tiles -> its y tiles (which read the dropped new memory the write tiles leave); a filler finishes ALL its
  // write tiles before it waits for anything, so the counters always complete.
  // (Measured and not kept: the attention over the knowledge base -- kb_attend_kernel's units -- on the fillers too: 60 CUs take
  // 30 us over the 25.7 MB the launch reads in 6.8, the tiles wait 39 us for y; profiles/r06_fillers.txt.)
  LinP wlin;                // [m_prev, info] Wm + bm -> m_new, and the dropped copy this step's y reads
  uint32_t* gflag;          // [8] device words, zero when the launch starts: write tiles finished per block of 16 questions
  int step;                 // (-DMACX_FILL_PROF: step 5's filler 0 / tile 0 leave 100 MHz timestamps in sync[32..47], tools/fill_prof.py)
};

struct ChainFwdP {
  int M, N, d;              // rows (B*N), rows per question, width
  int mode;                 // 0: KB -> X -> H1 -> I2 ; 1: X is read back (no read dropout: the projected KB is step-invariant);
                            // 2: as 0 with stage 0 done by the previous launch's fillers (KBd and bytes2 are read)
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
  ChainPreP pre;
};

// dKB = sum_i (dX_i Wx^T) * kbmask_i + att_i (x) dinfo_i on the CUs a chain_bwd launch leaves idle (d = 512, 64-row tiles: 196 tiles
// on 256 CUs at B = 64).  The launch of step i carries `nfill` extra workgroups -- one per idle CU: a chain workgroup fills a CU's
// LDS -- and each runs `njobs` JOBS: a job is one 64-row tile of step `step` = i + 1 (whose dX is complete): tile -> LDS, one K loop
// against Wx^T, keep bits, + att (x) dinfo, read-modify-write of the caller's fp32 gradient.  nfill * njobs tiles are covered per
// launch; the `nskip` tiles left out are a window that moves with the step (dkb_win0), so that what is left for the closing launch
// (chain_dkb_rest_kernel: step 0 everywhere + the skipped (step, tile) pairs) is ~2 jobs per tile, spread over the whole chip.
// Every (step, tile) pair is added exactly once, in a fixed order per tile (steps p - 1 .. 1 as launched, then the closing launch).
struct ChainDkbP {
  int njobs;                 // jobs per filler workgroup; 0: no fillers in this launch
  int nfill;                 // filler workgroups
  int step;                  // fillers: the step whose dX they multiply
  int p;                     // steps of the run
  int nskip;                 // tiles a filler launch leaves out
  const char* dX; size_t dx_step;             // H2 tensor dX_0, bytes between steps
  ChainW WxT;
  const uint8_t* bits; size_t bits_step;      // keep bits of the knowledge-base dropout of step 0, row-major [M][d/8]; bytes between steps; null = keep all
  float inv_keep;
  const float* att; size_t att_step;          // [M] attention of step 0, floats between steps
  const float* dinfo; int ld_dinfo; size_t dinfo_step;   // [B][ld_dinfo] of step 0
  float* out;                // dKB [M][d] fp32
  uint32_t* prof;            // -DMACX_FILL_PROF: filler 0 of step 5's launch leaves shader-clock stamps here (sync words 48..62, tools/fill_prof.py)
  int dbg;                   // timing knobs (macx_opts.tune[MACX_TUNE_PHASE_MASK] bits 22-26; results are wrong except under 16): 1 no K loop,
                             // 2 no read of the gradient (store only), 4 no tile load, 8 no row pass, 16 atomic adds instead of read-modify-write
};
// first tile of the window filler launches of step s leave out (s >= 1)
__host__ __device__ inline int dkb_win0(int s, int nskip, int ntile) { return (int)(((uint32_t)(s - 1) * (uint32_t)nskip) % (uint32_t)(ntile - nskip + 1)); }
__host__ __device__ inline bool dkb_skipped(int s, int t, int nskip, int ntile) {
  const int w = dkb_win0(s, nskip, ntile);
  return t >= w && t < w + nskip;
}

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
  ChainDkbP dkb;            // dKB jobs of the PREVIOUSLY differentiated step on the idle CUs (njobs = 0: none)
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
hipError_t chain_dkb_rest_launch(const ChainDkbP& q, int M, int N, int d, hipStream_t st);    // d = 512, 64-row tiles only
// jobs per filler workgroup of a chain_bwd launch (0: the merged dKB launch instead) and what follows from it
// MACX_TUNE_PRE_FILL: 1 (default) stage 0 of step i + 1 on the idle CUs of chain_fwd's launch of step i; 0 every launch its own
inline int pre_fill_mode() { return tune_get(MACX_TUNE_PRE_FILL, 1); }
inline int pre_fill_count(int d, size_t M, int ncu) {
  if (!pre_fill_mode() || d != 512 || ncu <= 0) return 0;
  const int ntile = (int)chain_tiles(d, M);
  if (chain_tile_rows(d, M) == 16) {
    // 16-row tiles (small batches, data-parallel shards of 8 - 16 questions): at most one tile per CU and a third of a CU's LDS each,
    // so fillers fit anywhere; a quarter of the chip takes the handful of linear tiles and 2 - 4 short stage-0 jobs each
    return ncu >= 64 ? ncu / 4 : 0;
  }
  // 32- and 64-row tiles fill a CU's LDS: fillers are the CUs without a tile in the launch's last round
  const int idle = (ncu - ntile % ncu) % ncu;
  return idle * 6 >= ntile ? idle : 0;           // (a filler walks ntile / idle tiles at about a sixth of a chain tile's time each)
}
inline int dkb_fill_mode() { return tune_get(MACX_TUNE_DKB_FILL, 3); }
struct DkbFillPlan { int njobs, nfill, nskip; };
inline DkbFillPlan dkb_fill_plan(int d, size_t M, int N, int p, int ncu) {
  DkbFillPlan f{0, 0, 0};
  const int nj = dkb_fill_mode();
  if (nj <= 0 || d != 512 || N < 64 || p < 2 || ncu <= 0 || chain_tile_rows(d, M) != 64) return f;      // (N >= 64: a tile touches <= 2 questions)
  const int ntile = (int)chain_tiles(d, M);
  const int idle = (ncu - ntile % ncu) % ncu;       // CUs without a tile in the launch's last round
  if (idle * nj * 4 < ntile) return f;              // too few to matter: the closing launch would carry most of the work
  f.njobs = nj; f.nfill = idle;
  f.nskip = ntile > idle * nj ? ntile - idle * nj : 0;
  return f;
}

}  // namespace macx
