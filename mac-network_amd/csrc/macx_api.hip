// macx_api.hip -- extern "C" entry points of libmacx.so (see include/macx.h) and the host-side
// sequencing of one MAC-cell run: what MACnet.MACnetwork's loop (model.py:453-458) makes TF execute,
// expressed as a fixed schedule of gfx950 kernels on one HIP stream.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/macx.h"
#include "macx_common.hip.h"
#include "macx_gemm.hip.h"
#include "macx_gemm6.hip.h"
#include "macx_gemm_tn.hip.h"
#include "macx_wgrad6.hip.h"
#include "macx_gemm3h.hip.h"
#include "macx_h2.hip.h"
#include "macx_gemm_h2.hip.h"
#include "macx_chain_api.hip.h"
#include "macx_wgrad_h2.hip.h"
#include "macx_small.hip.h"
#include "macx_ops.hip.h"

using namespace macx;

#define CK(expr)                         \
  do {                                   \
    hipError_t _e = (expr);              \
    if (_e != hipSuccess) return (int)_e;\
  } while (0)
#define CKI(expr)             \
  do {                        \
    int _r = (expr);          \
    if (_r != 0) return _r;   \
  } while (0)

namespace {

inline size_t al4(size_t n) { return (n + 3) & ~(size_t)3; }

// The knowledge-base GEMM family runs on one of two kernels (macx_gemm_mode / macx_opts.gemm_family):
//   split (1): macx_gemm6.hip.h, fp32 operands as three exact bf16 pieces on the bf16 matrix pipe, fp32 accumulate;
//   native (0): macx_gemm.hip.h, v_mfma_f32_16x16x4_f32.
// They take different weight packings: plain weights -> format 1 (bf16 planes, 1.5x the floats) / 0, weights mixed
// with a per-question vector (B_YMIX_*) -> format 2 (fp32 k-major tiles) / 0.  Buffers are sized for the larger one.
//   h2 (2, the default): macx_gemm_h2.hip.h -- every [B,N,d] activation lives in HBM as two fp16 planes + per-row-block
//     exponents (macx_h2.hip.h), three fp16 MFMA terms per product; plain weights -> format 3 (H2 planes + exponent).
// the kernel family an ABI call runs on (macx_opts.gemm_family, when set) and the call's tuning table (macx_opts.tune): installed for
// the duration of the call, on this thread
struct ModeScope {
  int saved;
  TuneScope ts;
  explicit ModeScope(const macx_opts* o) : saved(gemm_call_override()), ts(o ? o->tune : nullptr) {
    if (o && o->gemm_family >= 1 && o->gemm_family <= 3) gemm_call_override() = o->gemm_family - 1;
  }
  ~ModeScope() { gemm_call_override() = saved; }
};
inline bool h2_mode() { return gemm_split_mode() == 2; }
// the read unit's products as one kernel per direction (macx_chain_h2.hip.h); MACX_TUNE_CHAIN = 0 falls back to the per-product
// launches that also serve d > 512 and N < 16
inline int chain_mode() { return tune_get(MACX_TUNE_CHAIN, 1); }
inline bool use_chain(int d, int N) { return h2_mode() && chain_mode() && chain_supported(d, N); }
// dy from the chain kernel and S_b of all steps as one deferred launch; MACX_TUNE_SB_DEFER = 0: sb_h2 once per step (what N < 32 runs)
inline int sb_defer_mode() { return tune_get(MACX_TUNE_SB_DEFER, 1); }
// the deferred S_b contraction on 128 x 256 tiles with summation by parts (sb_h2w_kernel); MACX_TUNE_SB_WIDE = 0: the 128 x 128 kernel.
// (Measured and removed: dW1a / dW1b as ONE dual-A contraction over X * y and X -- twice the matrix work, 392 against 345 us,
// profiles/r05_dual_contraction_ab.txt; a side queue for the per-step dKB / dW2 contractions -- +8 % / +130 us per step,
// profiles/r04_side_queue_ab.txt.)
inline int sb_wide_mode() { return tune_get(MACX_TUNE_SB_WIDE, 1) ? 1 : 0; }
inline int wfmt_plain() { return h2_mode() ? 3 : (gemm_split_mode() ? 1 : 0); }
inline int wfmt_ymix() { return gemm_split_mode() ? 2 : 0; }
inline size_t wsize(size_t K, size_t n) { return K * n * 3 / 2; }     // covers format 1 (3/2) and format 3 (1 + the exponent)
inline hipError_t wgrad_any(const TnP& t, hipStream_t st) {
  return (gemm_split_mode() && !(kb_gemm_dbg() & 128)) ? wgrad6_launch(t, st) : wgrad_tn_launch<A_PLAIN>(t, st);   // dbg 128: f32 TN kernel
}
// fp32-operand GEMM (stem convolutions, unit entry points, and the cell in modes 0 / 1); in h2 mode the fp32-operand
// callers that remain (the stem's implicit GEMM) run on the split-bf16 kernel, whose weight format they pack (1)
template <int AP, int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm(const GemmP& g, hipStream_t st) {
  return gemm_split_mode() ? kb_gemm6_launch<AP, BP, EP, COLSUM>(g, st) : kb_gemm_launch<AP, BP, EP, COLSUM>(g, st);
}
inline int wfmt_plain_f32ops() { return gemm_split_mode() ? 1 : 0; }

inline DropSpec make_drop(float keep, uint32_t seed, uint32_t site, uint32_t step) {
  DropSpec s;
  s.word = nullptr;
  s.key = site_key(seed, site, step);
  if (keep >= 1.0f) {
    s.thr24 = 1u << 24;
    s.inv_keep = 1.0f;
  } else {
    s.thr24 = (uint32_t)floor((double)keep * 16777216.0);
    s.inv_keep = 1.0f / keep;
  }
  return s;
}
// a site of the cell: the run's mask word (macx_dropout.mask_word, device memory, may be null) rides along
inline DropSpec make_drop(float keep, const macx_dropout* dp, uint32_t site, uint32_t step) {
  DropSpec s = make_drop(keep, dp->seed, site, step);
  s.word = dp->mask_word;
  return s;
}
inline DropSpec no_drop() {
  DropSpec s;
  s.word = nullptr;
  s.key = 0;
  s.thr24 = 1u << 24;
  s.inv_keep = 1.0f;
  return s;
}

inline int write_in_dim(const macx_opts* o, int d) {
  int dim = d;
  if (o->write_inputs == MACX_WRITE_BOTH) dim = 2 * d;
  if (o->write_self_att) dim += d;
  return dim;
}

// ---- layout of `saved` ------------------------------------------------------------------------
// The backward pass reads its weights in layouts of its own (transposes; H2 planes for the chain kernels): [wxT | w1aT | w1bT | w2T |
// wyT | wmT | wqT | wqUT | wccT | wcc2T | wscT | wgT], the first region of the backward workspace layout (make_bwd).  A run that keeps
// its activations packs them in the FORWARD pass's pack launch, into `saved` (round 5: one launch and its boundary less per step).
inline size_t bwd_packs_floats(const macx_opts* o, const macx_shapes* s) {
  const size_t d = s->d, p = s->p, win = write_in_dim(o, s->d);
  return 4 * al4(wsize(d, d)) + al4(d * d) + al4(win * d) + al4(d * d) + al4((o->control_input_unshared ? p : 1) * d * d) +
         al4(2 * d * d) + 3 * al4(d * d);
}

struct SavedLayout {
  size_t seg[MACX_SEG_COUNT];
  size_t seg_count[MACX_SEG_COUNT];
  size_t wx_p, w1a_p, w1b_p, w2_p;  // packed forward weights (read unit)
  size_t wy_p, wm_p, wq_p, wqU_p;   // packed forward weights of the [B,d] linears
  size_t wg_p, ws_p;                // packed gate / self-attention control projection
  size_t wc_p, wc2_p;               // packed contControl (+ _2) weights (recurrent control)
  size_t kb_bits, att_bits;         // [pk][B*N*d/32] keep bits of the two [B,N,d] dropout sites
  size_t bits_stride;               // words per step (0 when activations are not kept)
  size_t ctrl_t;                    // [B,d]   act(qInput(vecQ))
  size_t cI;                        // [p,B,d] controlInput per step (mac_cell.py:447)
  size_t cc;                        // [p,B,d] continuous control (alias of cI unless controlFeedPrev)
  size_t cc_h;                      // [p,B,d] hidden act(contControl) when controlContAct != NON
  size_t md;                        // [p,B,d] dropped memory fed to projY
  size_t y;                         // [p,B,d] projected memory
  size_t info_raw;                  // [p,B,d] information before write dropout
  size_t wlin;                      // [p,B,d] pre-activation output of linearLayernewMemory
  size_t mnew;                      // [p,B,d] post-activation, pre-gate new memory (gate only)
  size_t sc;                        // [p,B,d] projected control for self attention
  size_t self_smry;                 // [p,B,d]
  size_t logit_part;                // [d/128][B*N]
  size_t X, H1, I2;                 // [pk][B,N,d], pk = p (keep) or 1
  size_t KBd;                       // [pk][B,N,d] dropped knowledge base (ops.py:678)
  size_t act_stride;                // floats per kept step of X / H1 / I2 / KBd if keep else 0
  size_t act_floats;                // floats of one such tensor (B*N*d, or the H2 size in h2 mode)
  size_t wmax;                      // h2: max |W| of projX, memKbProj2, W1a, W1b (4 floats)
  size_t bwd_packs;                 // keep: the backward pass's weight packs (bwd_packs_floats), written by the forward pack launch; else 0
  size_t sync;                      // SYNC_WORDS uint32: [i] the y counter of step i's chain launch (ChainPreP::yflag, p <= 32), [63] the fail
                                    // word, [64 + 16 i ..] step i's per-block counters (gflag); zeroed by macx_cell_begin's init_states launch
  size_t total;
};

SavedLayout make_saved(const macx_opts* o, const macx_shapes* s, int keep) {
  SavedLayout L;
  memset(&L, 0, sizeof(L));
  const size_t B = s->B, S = s->S, N = s->N, d = s->d, p = s->p;
  size_t off = 0;
  auto take = [&](size_t n) { size_t r = off; off += al4(n); return r; };
  const size_t gate_w = o->write_gate ? (o->write_gate_shared ? 1 : d) : 0;
  const size_t counts[MACX_SEG_COUNT] = {(p + 1) * B * d, (p + 1) * B * d, p * B * d, p * B * S, p * B * N,
                                         o->write_self_att ? p * B * p : 0, p * B * gate_w};
  for (int i = 0; i < MACX_SEG_COUNT; ++i) {
    L.seg_count[i] = counts[i];
    L.seg[i] = take(counts[i]);
  }
  L.wx_p = take(wsize(d, d));
  L.w1a_p = take(wsize(d, d));
  L.w1b_p = take(wsize(d, d));
  L.w2_p = take(wsize(d, d));
  L.wy_p = take(d * d);
  L.wm_p = take((size_t)write_in_dim(o, s->d) * d);
  L.wq_p = take(d * d);
  L.wqU_p = take((o->control_input_unshared ? p : 1) * d * d);
  L.wg_p = o->write_gate ? take(d * d) : 0;
  L.ws_p = o->write_self_att ? take(d * d) : 0;
  L.wc_p = o->control_feed_prev ? take(2 * d * d) : 0;
  L.wc2_p = o->control_feed_prev ? take(d * d) : 0;
  L.ctrl_t = take(B * d);
  L.cI = take(p * B * d);
  if (o->control_feed_prev) {
    L.cc = take(p * B * d);
    L.cc_h = take(p * B * d);
  } else {
    L.cc = L.cI;
    L.cc_h = L.cI;
  }
  L.md = take(p * B * d);
  L.y = take(p * B * d);
  L.info_raw = take(p * B * d);
  L.wlin = take(p * B * d);
  L.mnew = o->write_gate ? take(p * B * d) : L.wlin;
  if (o->write_self_att) {
    L.sc = take(p * B * d);
    L.self_smry = take(p * B * d);
  }
  L.logit_part = take((d / 64) * B * N);   // one partial per 64- or 128-column tile
  const size_t pk = keep ? p : 1;
  // h2 mode: activations are H2 tensors (4 bytes per element + exponents + pad rows); the attention keep bits are kept as
  // one byte per 8-column slot in slot order [d/8][Rp]
  L.act_floats = h2_mode() ? al4(h2_floats(B * N, d)) : B * N * d;
  L.act_stride = keep ? L.act_floats : 0;
  L.X = take(pk * L.act_floats);
  L.H1 = take(pk * L.act_floats);
  L.I2 = take(pk * L.act_floats);
  L.KBd = take(pk * L.act_floats);
  const size_t bits_floats = h2_mode() ? al4(((B * N + H2_PAD_ROWS) * (d / 8) + 3) / 4) : B * N * d / 32;
  L.bits_stride = keep ? bits_floats : 0;
  L.kb_bits = take(pk * bits_floats);
  L.att_bits = take(pk * bits_floats);
  L.wmax = take(8 + 4 * 64);          // the four maxima (+ 4 spare), then absmax4's per-workgroup partials
  L.bwd_packs = keep ? take(bwd_packs_floats(o, s)) : 0;
  L.sync = take(SYNC_WORDS);
  L.total = off;
  return L;
}

inline int nrb_of(int N, int B, int d) {
  const int rows = kb_gemm_pick_rt(N, B, d / (16 * kb_gemm_nw())) * 16;
  return (N + rows - 1) / rows;
}

inline int wgrad_splits(int M, int Kd, int Jd) {
  // workgroups = splits x output tiles ~ one per CU; the split-bf16 kernel uses 128 x 256 tiles when Jd allows
  const int jw = (gemm_split_mode() && Jd % 256 == 0) ? 2 : 1;
  const int tiles = (Kd / T_TILE) * (Jd / (jw * T_TILE));
  int ns = 256 / tiles;
  if (ns < 1) ns = 1;
  const int max_by_rows = (M + 63) / 64;   // at least 64 rows per split
  if (ns > max_by_rows) ns = max_by_rows;
  if (ns < 1) ns = 1;
  return ns;
}
// ... of the deferred H2 contractions (macx_wgrad_h2.hip.h: up to 256 x 256 output tiles)
inline int wgrad_big_splits(int M, int Kd, int Jd) {
  if (!h2_mode()) return wgrad_splits(M, Kd, Jd);
  int ns = 256 / wgrad_h2_tiles(Kd, Jd);
  const int max_by_rows = (M + 255) / 256;
  if (ns > max_by_rows) ns = max_by_rows;
  return ns < 1 ? 1 : ns;
}
inline int rows_per_split(int M, int ns) {
  int r = (M + ns - 1) / ns;
  return h2_mode() ? (r + 31) & ~31 : (r + 1) & ~1;       // H2: a 32-row stage never straddles two splits
}
inline int sb_qpg(int B, int N = 0) {   // questions per workgroup group in the S_b kernel: 16 tiles x groups ~ 256
  int qpg = (B + 15) / 16;
  if (qpg < 1) qpg = 1;
  if (h2_mode() && N > 0) {                 // the H2 kernel keeps one fp16 factor per (question, row) of a group in LDS
    const int rows_q = ((N + 31) / 32) * 32;
    const int cap = SBH_MAXROWS / rows_q;
    if (qpg > cap) qpg = cap < 1 ? 1 : cap;
    if (qpg > SBH_MAXQ) qpg = SBH_MAXQ;
  }
  return qpg;
}

// ---- layout of the backward workspace ---------------------------------------------------------
struct BwdLayout {
  size_t wxT_p, w1aT_p, w1bT_p, w2T_p;
  size_t wyT, wmT, wqT, wqUT, wccT, wcc2T, wscT, wgT;
  size_t packs_end;
  size_t dI2, dI1, dX, da;
  size_t DM;        // [p+1,B,d]  dL/d memories
  size_t DC;        // [p+1,B,d]  dL/d controls
  size_t dcI;       // [p,B,d]    dL/d controlInput_i
  size_t dcc;       // [p,B,d]    dL/d continuous control (alias dcI unless controlFeedPrev)
  size_t dwlin;     // [p,B,d]    dL/d (newMemory linear output)
  size_t dwin;      // [p][B, win]   dL/d write inputs of every step
  size_t dinfo;     // [B,d]
  size_t dy_part;   // [2*d/128][B][d]
  size_t DY;        // [p,B,d]
  size_t dmd;       // [B,d]
  size_t dt, du;    // [B,d]
  size_t dt_part;   // [p,B,d] per-step products of the unshared control inputs' backward linear
  size_t slab_w2, slab_wx, slab_w1a, slab_w1b;
  size_t ns_big, ngroup;
  bool sb_wide;               // the deferred S_b contraction runs on sb_h2w_kernel (128 x 256 tiles, summation by parts)
  int sb_qpg;                 // questions per workgroup group of the S_b kernel in use
  size_t db_rows;   // rows per step of db1_part / dbx_part
  size_t dwk_rows;  // rows per step of dwk_part / db2_part
  size_t dc_part, dls_part;   // chain kernel: per-row-group partials of dc / db_k, [p][dwk_rows][3][d] and [p][dwk_rows][3]
  size_t dyc_part;            // chain kernel: per-row-group partials of dy, [dwk_rows][3][d]
  bool chain_sums;
  bool dy_in_linear;          // ... and its per-tile partials are summed by the dy linear itself (small_linear PART form)
  bool sb_deferred;           // dy comes from the chain kernel: dI1 is kept per step and S_b of all steps is ONE launch at the end
  size_t dI1_stride;          // floats between the steps' dI1 (0: one buffer)
  size_t db2_part, db1_part, dbx_part, dwk_part, dbk_part, dwc_part, dbc_part, ctrl_dl;
  size_t tmpBd[4];  // [B,d] scratch
  size_t dccx;      // [p+1,B,d] gradient reaching cc_i from the NEXT step's contControl input (feedPrevAtt off)
  size_t dlin1;     // [p,B,d] gradient wrt the first contControl layer's pre-activation
  size_t dxc;       // [B,2d]  gradient wrt the contControl input of the current step
  size_t dzpre;     // [p,B,d] gate pre-activation gradient
  size_t dsc;       // [p,B,d] gradient of the projected control of the self attention
  size_t dws_part, dbs_part;   // [p,B,d], [p,B]
  size_t small_slab, small_slab_stride;
  size_t tmp_dd;    // [d,d] scratch
  size_t act_floats;                 // floats of one [B,N,d] activation (H2 size in h2 mode)
  size_t ecom;                       // h2: [4][EMIN_NB][8] ints, partial minima of the row exponents of H1 / dI2 / KBd / dX over all steps
  size_t wg_ftab, wg_ftab2;          // h2: [d/128][d/128][Mpad] fp16 row factors of the two deferred weight-gradient contractions
  size_t total;
};

BwdLayout make_bwd(const macx_opts* o, const macx_shapes* s) {
  BwdLayout L;
  memset(&L, 0, sizeof(L));
  const size_t B = s->B, N = s->N, d = s->d, p = s->p;
  const size_t win = write_in_dim(o, s->d);
  size_t off = 0;
  auto take = [&](size_t n) { size_t r = off; off += al4(n); return r; };
  L.wxT_p = take(wsize(d, d)); L.w1aT_p = take(wsize(d, d)); L.w1bT_p = take(wsize(d, d)); L.w2T_p = take(wsize(d, d));
  L.wyT = take(d * d);
  L.wmT = take(win * d);
  L.wqT = take(d * d);
  L.wqUT = take((o->control_input_unshared ? p : 1) * d * d);
  L.wccT = take(2 * d * d); L.wcc2T = take(d * d); L.wscT = take(d * d); L.wgT = take(d * d);
  L.packs_end = off;                  // == bwd_packs_floats(o, s).  These are offsets into SavedLayout::bwd_packs: the forward pass's pack
  off = 0;                            // launch writes the backward pass's packs into `saved`; the workspace proper starts here
  L.act_floats = h2_mode() ? al4(h2_floats(B * N, d)) : B * N * d;
  L.chain_sums = use_chain((int)d, (int)N) && N >= 32;
  L.sb_deferred = L.chain_sums && sb_defer_mode();
  L.dI1_stride = L.sb_deferred ? L.act_floats : 0;
  L.dy_in_linear = L.sb_deferred && B <= 128 && ((N - 2) >> chain_tile_shift((int)d, B * N)) + 2 <= (size_t)LIN_PART_TILES;
  L.dI2 = take(p * L.act_floats); L.dI1 = take((L.sb_deferred ? p : 1) * L.act_floats); L.dX = take(p * L.act_floats); L.da = take(B * N);   // dI2, dX (dI1) kept per step
  L.DM = take((p + 1) * B * d);
  L.DC = take((p + 1) * B * d);
  L.dcI = take(p * B * d);
  L.dcc = o->control_feed_prev ? take(p * B * d) : L.dcI;
  L.dwlin = take(p * B * d);
  L.dwin = take(p * B * win);     // per step: the deferred dKB launch reads every step's dinfo (a column view of it)
  L.dinfo = take(p * B * d);
  L.dy_part = take((4 * d / 128) * B * d);      // 2 column partials per 128-column tile (4 on the H2 kernel)
  L.DY = take(p * B * d);
  L.dmd = take(B * d);
  L.dt = take(B * d); L.du = take(B * d);
  L.dt_part = o->control_input_unshared ? take(p * B * d) : 0;
  L.ns_big = wgrad_big_splits((int)(p * B * N), (int)d, (int)d);
  L.sb_wide = L.sb_deferred && h2_mode() && sb_wide_mode() && sb_h2_wide_ok((int)B, (int)N, (int)d);
  L.sb_qpg = L.sb_wide ? sb_h2_wide_qpg((int)B, (int)N, (int)d) : sb_qpg((int)B, (int)N);
  L.ngroup = (B + L.sb_qpg - 1) / L.sb_qpg;
  L.slab_w2 = take(L.ns_big * d * d);
  L.slab_wx = take(L.ns_big * d * d);
  L.slab_w1a = take((L.sb_deferred ? 1 : p) * L.ngroup * d * d);
  L.slab_w1b = take((L.sb_deferred ? 1 : p) * L.ngroup * d * d);
  // column-sum partials of dI1 / dX: one row per GEMM workgroup row block, or per 64-row tile of the chain kernel
  const size_t nrb = use_chain((int)d, (int)N) ? chain_tiles((int)d, B * N) : B * nrb_of((int)N, (int)B, (int)d);
  L.db_rows = nrb;
  // dw_k / db2 partials: one row per question, or per 64-row tile when the chain kernel sums them (N >= 32)
  L.dwk_rows = L.chain_sums ? chain_tiles((int)d, B * N) : B;
  L.db2_part = take(p * L.dwk_rows * d);
  L.db1_part = take(p * nrb * d);
  L.dbx_part = take(p * nrb * d);
  L.dwk_part = take(p * L.dwk_rows * d);
  if (L.chain_sums) { L.dc_part = take(p * L.dwk_rows * 3 * d); L.dls_part = take(p * L.dwk_rows * 3); L.dyc_part = take(L.dwk_rows * 3 * d); }
  L.dbk_part = take(p * B);
  L.dwc_part = take(B * d);
  L.dbc_part = take(p * B);
  L.ctrl_dl = take(p * B * (size_t)s->S);
  for (int i = 0; i < 4; ++i) L.tmpBd[i] = take(B * d);
  if (o->control_feed_prev) { L.dccx = take((p + 1) * B * d); L.dlin1 = take(p * B * d); L.dxc = take(B * 2 * d); }
  if (o->write_gate) L.dzpre = take(p * B * d);
  if (o->write_self_att) { L.dsc = take(p * B * d); L.dws_part = take(p * B * d); L.dbs_part = take(p * B); }
  // scratch slabs for the small weight gradients (largest: write unit, rows p*B, [win x d])
  size_t small = 0;
  {
    const int ns = wgrad_splits((int)(p * B), (int)d, (int)d);
    small = (size_t)ns * d * d;
  }
  L.small_slab_stride = al4(small);
  L.small_slab = take(8 * L.small_slab_stride);        // SmallWgradBatch: one slab set per batched contraction
  L.tmp_dd = take(d * d);
  L.ecom = take(4 * EMIN_NB * 8);
  L.wg_ftab = take(h2_mode() ? (d / 128) * (d / 128) * wgrad_h2_mpad(p * B * N) / 2 + 4 : 4);
  L.wg_ftab2 = take(h2_mode() ? (d / 128) * (d / 128) * wgrad_h2_mpad(p * B * N) / 2 + 4 : 4);
  L.total = off;
  return L;
}

// row stride of the dropout element index: the logical width of a zero-padded cell, else the width
inline int dlog_of(const macx_shapes* s) { return (s->d_logical > 0 && s->d_logical < s->d) ? s->d_logical : s->d; }

int check_impl(const macx_opts* o, const macx_shapes* s) {
  if (!o || !s) return MACX_EINVAL;
  if (o->abi_version != MACX_ABI_VERSION) return MACX_EINVAL;
  if (s->B < 1 || s->S < 1 || s->N < 1 || s->p < 1 || s->d < 128) return MACX_EINVAL;
  if (s->d % 128 != 0 || s->d > 1024) return MACX_EINVAL;
  if (s->d_logical != 0 && s->d_logical != s->d) {
    // a zero-padded cell: the dropout index is taken at the logical width (sites of the H2 kernel family only)
    if (s->d_logical % 8 != 0 || s->d_logical <= s->d - 128 || s->d_logical > s->d) return MACX_EINVAL;
    if (!h2_mode()) return MACX_EUNSUPPORTED;
  }
  if (s->S > C_MAXS || s->N > K_MAXN || s->p > CB_MAXZ) return MACX_EINVAL;
  if ((size_t)(s->b0 + s->B) * s->N * s->d >= (1ull << 32)) return MACX_EINVAL;  // 32-bit dropout index
  if (o->write_inputs != MACX_WRITE_BOTH) return MACX_EUNSUPPORTED;
  if (o->read_mem_act == MACX_ACT_NON) return MACX_EUNSUPPORTED;   // no memKbProj_2 layer then (ops.py:325)
  if (o->write_gate && o->write_gate_shared) return MACX_EREJECTED;   // [B,d] * [B] does not broadcast in the reference
  if (o->write_self_att && s->p + 1 > SA_MAXH) return MACX_EINVAL;
  return MACX_OK;
}

// the parameter block of the read unit's forward chain kernel for step `i`; `ob`: the step whose X / H1 / I2 / KBd / keep-bit
// buffers receive the outputs (= i in a run; the timing hook rotates it)
LinP lin_basic(const float* x, int ldx, int K, int rows, const float* Wp, const float* bias, int n_out, int act, float* out, int ldo);

// the write unit's linear of step i without self attention and gate (mac_cell.py:305-375, writeInputs = BOTH): [m_i, info_i] Wm + bm
// -> m_{i+1}; with md_next its epilogue also leaves the next step's dropped memory (mac_cell.py:214-217, ops.py:679)
LinP make_write_lin(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P, float* saved,
                    const SavedLayout& L, int i, const float* info, float* wout, bool md_next) {
  const int B = s->B, d = s->d;
  const size_t Bd = (size_t)B * d;
  LinP l = lin_basic(saved + L.seg[MACX_SEG_MEMORIES] + (size_t)i * Bd, d, d, B, saved + L.wm_p, P->newMemory_b, d, o->write_mem_act, wout, d);
  l.seg[1] = LinSeg{info, d, d, 0};
  l.Ktot = 2 * d;
  if (md_next) {
    l.use_drop = 2; l.drop_ld = dlog_of(s);
    l.d1 = o->memory_variational_dropout ? make_drop(dp->keep_memory, dp, SITE_MEM_VAR, 0) : make_drop(dp->keep_memory, dp, SITE_MEM, i + 1);
    l.d2 = make_drop(dp->keep_read, dp, SITE_READ_MEM, i + 1);
    l.drop_row0 = (uint32_t)s->b0;
    l.out_drop = saved + L.md + (size_t)(i + 1) * Bd; l.ld_od = d;
  }
  return l;
}

// pre: bit 2 -- this launch's filler workgroups compute the step's y (ChainPreP::ylin); bit 0 -- stage 0 of this step was done by the previous step's launch (ChainFwdP::mode 2); bit 1 -- this launch's fillers do
// stage 0 of step i + 1 (ChainPreP).  Both only where macx_cell_forward sequences the steps itself.
ChainFwdP make_chain_fwd(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                         const macx_inputs* in, float* saved, const SavedLayout& L, int keep, int i, int ob, int pre = 0) {
  const int B = s->B, N = s->N, d = s->d;
  const int R = B * N;
  const size_t Bd = (size_t)B * d, dd_ = (size_t)d * d;
  const bool rdrop = dp->keep_read < 1.0f;
  auto wref = [&](size_t off) { return ChainW{reinterpret_cast<const char*>(saved + off), reinterpret_cast<const int*>(saved + off) + dd_}; };
  ChainFwdP c;
  memset(&c, 0, sizeof(c));
  c.M = R; c.N = N; c.d = d;
  // without read dropout the projected knowledge base is step-invariant: inference reads step 0's X back; a run that
  // keeps its activations recomputes it into the step's own buffers, as the reference's graph does (ops.py:688)
  c.mode = (rdrop || L.act_stride != 0 || i == 0) ? 0 : 1;
  c.dbg = (kb_gemm_dbg() >> 12) & 31;
  c.kb = in->knowledgeBase;
  c.dlog = dlog_of(s);
  c.first = (uint32_t)((size_t)s->b0 * N * c.dlog);
  c.thr1 = 1u << 24; c.inv1 = 1.0f; c.thr2 = 1u << 24; c.inv2 = 1.0f;
  if (rdrop) {
    const DropSpec dk = make_drop(dp->keep_read, dp, SITE_READ_KB, i);
    const DropSpec da = make_drop(dp->keep_read, dp, SITE_READ_ATT, i);
    c.key1 = dk.key; c.thr1 = dk.thr24; c.inv1 = dk.inv_keep;
    c.bits1 = reinterpret_cast<uint8_t*>(saved + L.kb_bits + (size_t)ob * L.bits_stride);
    c.key2 = da.key; c.thr2 = da.thr24; c.inv2 = da.inv_keep;
    c.word = dp->mask_word;
    c.bytes2 = reinterpret_cast<uint8_t*>(saved + L.att_bits + (size_t)ob * L.bits_stride);
  }
  if (rdrop || i == 0) {
    c.KBd = h2_view(rdrop ? saved + L.KBd + (size_t)ob * L.act_stride : saved + L.KBd, R, d);
  }
  c.Wx = wref(L.wx_p); c.W1a = wref(L.w1a_p); c.W1b = wref(L.w1b_p); c.W2 = wref(L.w2_p);
  c.bx = P->projX_b; c.b1 = P->memKbProj_b; c.b2 = P->memKbProj2_b;
  c.act1 = o->read_mem_act; c.act2 = o->read_ctrl_act;
  c.y = saved + L.y + (size_t)i * Bd;
  c.c = saved + L.seg[MACX_SEG_CONTROLS] + (size_t)(i + 1) * Bd;
  c.wk = P->kbLogits_w;
  c.X = h2_view(saved + L.X + (size_t)ob * L.act_stride, R, d);
  if (keep) {
    c.H1 = h2_view(saved + L.H1 + (size_t)ob * L.act_stride, R, d);
    c.I2 = h2_view(saved + L.I2 + (size_t)ob * L.act_stride, R, d);
  }
  c.logits = saved + L.logit_part;
  if (rdrop && c.mode == 0 && (pre & 1)) c.mode = 2;
  if (pre & 6) c.pre.nfill = pre_fill_count(d, (size_t)R, device_cu_count());
  if ((pre & 4) && c.pre.nfill) {
    // this step's y = md Wy + by on the filler workgroups (cell_step_impl then launches no linear for it)
    c.pre.ylin = lin_basic(saved + L.md + (size_t)i * Bd, d, d, B, saved + L.wy_p, P->projY_b, d, MACX_ACT_NON, saved + L.y + (size_t)i * Bd, d);
    c.pre.step = i;
    c.pre.yflag = reinterpret_cast<uint32_t*>(saved + L.sync) + i;
    c.pre.fail = reinterpret_cast<uint32_t*>(saved + L.sync) + 63;
    if ((pre & 16) && i > 0) {
      // ... after the previous step's write unit (cell_step_impl did not launch it)
      c.pre.wlin = make_write_lin(o, s, dp, P, saved, L, i - 1, saved + L.seg[MACX_SEG_INFOS] + (size_t)(i - 1) * Bd,
                                  saved + L.seg[MACX_SEG_MEMORIES] + (size_t)i * Bd, true);
      c.pre.gflag = reinterpret_cast<uint32_t*>(saved + L.sync) + 64 + 16 * i;
    }
  }
  if (rdrop && (pre & 2) && c.pre.nfill) {
    c.pre.key1 = make_drop(dp->keep_read, dp, SITE_READ_KB, i + 1).key;
    c.pre.key2 = make_drop(dp->keep_read, dp, SITE_READ_ATT, i + 1).key;
    c.pre.bits1 = reinterpret_cast<uint8_t*>(saved + L.kb_bits + (size_t)(ob + 1) * L.bits_stride);
    c.pre.bytes2 = reinterpret_cast<uint8_t*>(saved + L.att_bits + (size_t)(ob + 1) * L.bits_stride);
    c.pre.KBd = h2_view(saved + L.KBd + (size_t)(ob + 1) * L.act_stride, R, d);
  }
  return c;
}

inline bool misaligned(const void* p) { return ((uintptr_t)p & 15) != 0; }

hipError_t pack(const float* src, int ld_k, int ld_j, int K, int Nout, float* dst, hipStream_t st) {
  hipLaunchKernelGGL(pack_weight_kernel, dim3(256), dim3(256), 0, st, src, ld_k, ld_j, K, Nout, dst);
  return hipGetLastError();
}
// ---- device-side fills and copies as KERNELS.  No hipMemsetAsync / hipMemcpyAsync anywhere in this library: under HIP-graph
// replay (ROCm 7.2) a memset node was seen to run out of order with the kernel nodes around it -- in round 3 behind the packed
// weights' maxima, in round 4 behind dL/dc of a captured training step (the control unit's gradients differed from the eager
// step's in three of four fresh processes, every replay alike; tools/graph_train_probe_verify.py) -- while kernel nodes keep
// their order.  Sizes in bytes, multiples of 4.
__global__ void fill_u32_kernel(uint32_t* dst, size_t n, uint32_t v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ void copy2d_u32_kernel(uint32_t* dst, size_t dpitch, const uint32_t* __restrict__ src, size_t spitch, size_t width, size_t rows) {
  const size_t n = width * rows;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / width, c = i - r * width;
    dst[r * dpitch + c] = src[r * spitch + c];
  }
}
inline unsigned fill_grid(size_t n) { return (unsigned)std::min<size_t>(1024, std::max<size_t>(1, (n + 1023) / 1024)); }
inline hipError_t dev_zero(void* dst, size_t bytes, hipStream_t st) {
  if (!bytes) return hipSuccess;
  hipLaunchKernelGGL(fill_u32_kernel, dim3(fill_grid(bytes / 4)), dim3(256), 0, st, reinterpret_cast<uint32_t*>(dst), bytes / 4, 0u);
  return hipGetLastError();
}
inline hipError_t dev_copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, hipStream_t st) {
  if (!width || !rows) return hipSuccess;
  hipLaunchKernelGGL(copy2d_u32_kernel, dim3(fill_grid(width / 4 * rows)), dim3(256), 0, st, reinterpret_cast<uint32_t*>(dst), dpitch / 4,
                     reinterpret_cast<const uint32_t*>(src), spitch / 4, width / 4, rows);
  return hipGetLastError();
}
inline hipError_t dev_copy(void* dst, const void* src, size_t bytes, hipStream_t st) { return dev_copy2d(dst, bytes, src, bytes, bytes, 1, st); }

// the maximum |W| behind a format-3 pack's exponent: matrix k of absmax4's four (projX, memKbProj_2, W1a, W1b).  Where the chain
// kernels run every consumer of the maxima is a pack of this run's ONE pack launch, which combines absmax_kernel's per-workgroup
// partials itself (fold: no absmax_finish launch); elsewhere the y-mixing GEMM kernels read the finished values.
struct WMaxRef { const float* src; int n; float* out; };
inline WMaxRef wmax_ref(const float* saved, size_t wmax, int k, bool fold) {
  float* base = const_cast<float*>(saved) + wmax;
  return fold ? WMaxRef{base + 8 + (size_t)k * 64, 64, base + k} : WMaxRef{base + k, 0, nullptr};
}

struct Packer {
  PackList L;
  int n = 0;
  void add(const float* src, int ld_k, int ld_j, int K, int Nout, float* dst, int k_src = -1, int n_src = -1, int fmt = 0,
           const float* maxabs = nullptr, float* exp_dst = nullptr, int maxabs_n = 0, float* maxabs_out = nullptr) {
    L.d[n++] = PackDesc{src, dst, ld_k, ld_j, K, Nout, k_src < 0 ? K : k_src, n_src < 0 ? Nout : n_src, fmt, maxabs_n, maxabs, exp_dst, maxabs_out};
  }
  hipError_t run(hipStream_t st) {
    if (n == 0) return hipSuccess;
    size_t big = 0;
    for (int i = 0; i < n; ++i) big = std::max(big, (size_t)L.d[i].K * L.d[i].Nout);
    // (grid-stride kernels: 64 workgroups per matrix cover the cell's d x d weights in two passes; the stem's 9 Cin x Cout
    // kernels are 18 - 36 times larger and took 60 - 90 us on that grid)
    hipLaunchKernelGGL(pack_weights_kernel, dim3(big > ((size_t)1 << 20) ? 512 : 64, n), dim3(256), 0, st, L);
    n = 0;
    return hipGetLastError();
  }
};


// ---- H2 helpers (macx_h2.hip.h) -------------------------------------------------------------------------------
// min over n entries of [n][cb] minimum-exponent arrays -> out[cb]
// minimum row exponents of up to four families of H2 tensors -> per-workgroup partials (macx_h2.hip.h: h2_emin_list_kernel)
hipError_t emin_list(const EminList& L, int count, hipStream_t st) {
  hipLaunchKernelGGL(h2_emin_list_kernel, dim3(EMIN_NB, count), dim3(256), 0, st, L);
  return hipGetLastError();
}
hipError_t h2_from_f32(const H2FromP& f, hipStream_t st) {
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(h2_from_f32_kernel), H2C_LDS);
  if (e != hipSuccess) return e;
  const int nrb = (f.N + H2C_ROWS - 1) / H2C_ROWS;
  hipLaunchKernelGGL(h2_from_f32_kernel, dim3(f.B * nrb, f.C / 128), dim3(H2C_THREADS), H2C_LDS, st, f);
  return hipGetLastError();
}
// out[0..3] = max |.| of four tensors; part: 4 * ABSMAX_BLOCKS floats of scratch, or null (one workgroup per tensor: slower)
constexpr int ABSMAX_BLOCKS = 64;
hipError_t absmax4(const float* a, size_t na, const float* b, size_t nb, const float* c, size_t nc, const float* d_, size_t nd,
                   float* out, float* part, hipStream_t st, bool finish = true) {
  AbsMaxList L;
  memset(&L, 0, sizeof(L));
  L.src[0] = a; L.n[0] = na; L.src[1] = b; L.n[1] = nb; L.src[2] = c; L.n[2] = nc; L.src[3] = d_; L.n[3] = nd;
  L.out = out; L.part = part;
  hipLaunchKernelGGL(absmax_kernel, dim3(part ? ABSMAX_BLOCKS : 1, 4), dim3(256), 0, st, L);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !part || !finish) return e;        // !finish: the consumer (pack_h2_weight) combines the partials itself
  hipLaunchKernelGGL(absmax_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)part, ABSMAX_BLOCKS, out);
  return hipGetLastError();
}

// out[0] = max |.| of one large tensor (n % 4 == 0); part: AMAX_BIG_BLOCKS floats of scratch
constexpr int AMAX_BIG_BLOCKS = 512;
hipError_t absmax_big(const float* a, size_t n, float* out, float* part, hipStream_t st) {
  hipLaunchKernelGGL(absmax_vec_kernel, dim3(AMAX_BIG_BLOCKS), dim3(256), 0, st, a, n / 4, part);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(absmax_finish_kernel, dim3(1), dim3(64), 0, st, (const float*)part, AMAX_BIG_BLOCKS, out);
  return hipGetLastError();
}

hipError_t transpose(const float* src, int R, int C, float* dst, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, st, src, R, C, dst);
  return hipGetLastError();
}

LinP lin_basic(const float* x, int ldx, int K, int rows, const float* Wp, const float* bias, int n_out,
               int act, float* out, int ldo) {
  LinP p;
  memset(&p, 0, sizeof(p));
  p.seg[0] = LinSeg{x, ldx, K, 0};
  p.seg[1] = LinSeg{nullptr, 0, 1 << 30, 0};
  p.seg[2] = LinSeg{nullptr, 0, 1 << 30, 0};
  p.Ktot = K;
  p.rows = rows;
  p.n_out = n_out;
  p.W = Wp;
  p.bias = bias;
  p.act = act;
  p.out = out; p.ldo = ldo;
  p.d1 = no_drop(); p.d2 = no_drop();
  return p;
}

hipError_t mask_bits(float keep, uint32_t seed, uint32_t site, uint32_t step, uint32_t first, size_t nwords, uint32_t* out,
                     hipStream_t st) {
  const DropSpec ds = make_drop(keep, seed, site, step);
  int grid = (int)((nwords + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(mask_bits_kernel, dim3(grid), dim3(256), 0, st, ds.key, ds.thr24, first, nwords, out);
  return hipGetLastError();
}

hipError_t rowsum(const float* src, int rows, int n, size_t ld, float* dst, hipStream_t st, int nz = 1, size_t zsrc = 0, size_t zdst = 0) {
  if (!dst) return hipSuccess;
  hipLaunchKernelGGL(rowsum_kernel, dim3((n + ROWSUM_COLS - 1) / ROWSUM_COLS, nz), dim3(1024), 0, st, src, rows, n, ld, dst, zsrc, zdst);
  return hipGetLastError();
}
// row sums collected while a pass is enqueued and launched together at its end (sources must stay untouched until then)
struct RowsumBatch {
  RowsumList L;
  int n = 0, max_n = 0;
  hipError_t add(const float* src, int rows, int cols, size_t ld, float* dst, hipStream_t st, int nz = 1, size_t zsrc = 0, size_t zdst = 0) {
    if (!dst) return hipSuccess;
    for (int z = 0; z < nz; ++z) {
      if (n == ROWSUM_MAX) { hipError_t e = run(st); if (e != hipSuccess) return e; }
      L.d[n++] = RowsumDesc{src + (size_t)z * zsrc, dst + (size_t)z * zdst, rows, cols, ld};
      max_n = cols > max_n ? cols : max_n;
    }
    return hipSuccess;
  }
  hipError_t run(hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rowsum_list_kernel, dim3((max_n + ROWSUM_COLS - 1) / ROWSUM_COLS, n), dim3(1024), 0, st, L);
    n = 0; max_n = 0;
    return hipGetLastError();
  }
};
hipError_t axpy(const float* x, size_t n, float* y, hipStream_t st) {
  hipLaunchKernelGGL(axpy_kernel, dim3(64), dim3(256), 0, st, x, n, y);
  return hipGetLastError();
}

int wgrad_impl(const float* A, int lda, const float* G, int ldg, int M, int Kd, int Jd, float* out, float* ws,
               hipStream_t st) {
  TnP t;
  memset(&t, 0, sizeof(t));
  t.M = M; t.Kd = Kd; t.Jd = Jd;
  t.nsplit = wgrad_splits(M, Kd, Jd);
  t.rows_per_split = rows_per_split(M, t.nsplit);
  t.A = A; t.lda = lda; t.a_mod = M; t.G = G; t.ldg = ldg;
  t.part = (t.nsplit == 1) ? out : ws;
  CK(wgrad_any(t, st));
  if (t.nsplit > 1) CK(slab_reduce_launch(ws, t.nsplit, (size_t)Kd * Jd, out, 0, st));
  return 0;
}

// The weight gradients of the [B,d]-sized linears (768 reduction rows each at p = 12: 17 us of latency per launch + a slab
// reduction) collected and run as ONE contraction launch + ONE slab reduction (wgrad6_list_kernel).  An entry that does not fit
// the batch (other tile geometry, the f32 kernel family, more than 8) runs on its own at once.
struct SmallWgradBatch {
  TnList L; SlabList S;
  int n = 0, nred = 0, jw = 0;
  size_t max_n = 0;
  float* slab; size_t slab_stride;
  SmallWgradBatch(float* slab_, size_t stride) : slab(slab_), slab_stride(stride) { memset(&L, 0, sizeof(L)); memset(&S, 0, sizeof(S)); }
  int add(const float* A, int lda, const float* G, int ldg, int M, int Kd, int Jd, float* out, hipStream_t st) {
    const bool ok = gemm_split_mode() && !(kb_gemm_dbg() & 128) && n < 8 && (jw == 0 || jw == wgrad6_jw(Jd));
    if (!ok) return wgrad_impl(A, lda, G, ldg, M, Kd, Jd, out, slab + (size_t)7 * slab_stride, st);
    jw = wgrad6_jw(Jd);
    TnP& t = L.d[n];
    t.M = M; t.Kd = Kd; t.Jd = Jd;
    t.nsplit = wgrad_splits(M, Kd, Jd);
    t.rows_per_split = rows_per_split(M, t.nsplit);
    t.A = A; t.lda = lda; t.a_mod = M; t.G = G; t.ldg = ldg;
    t.part = (t.nsplit == 1) ? out : slab + (size_t)n * slab_stride;
    L.nblk[n] = (Kd / T_TILE) * (Jd / (jw * T_TILE)) * t.nsplit;
    if (t.nsplit > 1) {
      S.d[nred++] = SlabDesc{t.part, t.nsplit, (size_t)Kd * Jd / 4, out, 0};
      max_n = std::max(max_n, (size_t)Kd * Jd);
    }
    ++n;
    return 0;
  }
  int run(hipStream_t st) {
    if (n == 0) return 0;
    CK(jw == 2 ? wgrad6_list_launch_t<2>(L, n, st) : wgrad6_list_launch_t<1>(L, n, st));
    if (nred) CK(slab_reduce_list_launch(S, nred, max_n, st));
    n = nred = 0;
    return 0;
  }
};

}  // namespace

// =================================================================================================
extern "C" {

int macx_abi_version(void) { return MACX_ABI_VERSION; }

const char* macx_strerror(int code) {
  switch (code) {
    case MACX_OK: return "ok";
    case MACX_EINVAL: return "invalid shapes, null or misaligned pointer";
    case MACX_EUNSUPPORTED: return "option combination has no HIP path yet";
    case MACX_EREJECTED: return "option value that raises in the reference";
    case MACX_ESMALL: return "saved/ws buffer too small";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown macx error";
  }
}

int macx_check(const macx_opts* o, const macx_shapes* s) { ModeScope ms(o); return check_impl(o, s); }

size_t macx_saved_floats(const macx_opts* o, const macx_shapes* s, int keep) {
  ModeScope ms(o);
  if (check_impl(o, s) != MACX_OK) return 0;
  return make_saved(o, s, keep).total;
}

size_t macx_ws_floats(const macx_opts* o, const macx_shapes* s, int for_backward) {
  ModeScope ms(o);
  if (check_impl(o, s) != MACX_OK) return 0;
  if (!for_backward) return 4;   // forward keeps everything in `saved`
  return make_bwd(o, s).total;
}

int macx_saved_segment(const macx_opts* o, const macx_shapes* s, int keep, int segment, size_t* offset, size_t* count) {
  ModeScope ms(o);
  CKI(check_impl(o, s));
  if (segment < 0 || segment >= MACX_SEG_COUNT || !offset || !count) return MACX_EINVAL;
  SavedLayout L = make_saved(o, s, keep);
  *offset = L.seg[segment];
  *count = L.seg_count[segment];
  return MACX_OK;
}

// -------------------------------------------------------------------------------------------------
namespace {
enum { U_CONTROL = 1, U_READ = 2, U_WRITE = 4, U_ALL = 7 };   // which units of a step an entry point runs

// weights -> MFMA operand layout  (memKbProj rows [0,d) multiply x*y, rows [d,2d) multiply x: ops.py:718)
struct Packer;
int add_bwd_packs(Packer& pk, const macx_opts* o, const macx_shapes* s, const macx_params* P, float* wT, const BwdLayout& W,
                  const float* saved, const SavedLayout& L, int units, hipStream_t st);
int pack_forward_weights(const macx_opts* o, const macx_shapes* s, const macx_params* P, float* saved, const SavedLayout& L,
                         int keep, int units, hipStream_t st) {
  const int B = s->B, d = s->d, p = s->p;
  {
    Packer pk;
    if (h2_mode() && (units & U_READ)) {
      // per-matrix maxima -> weight exponents (plain weights) and the bound of the question-mixed tile (W1a, W1b)
      const size_t dd_ = (size_t)d * d;
      static_assert(ABSMAX_BLOCKS == 64, "wmax_ref's partial layout");
      CK(absmax4(P->projX_W, dd_, P->memKbProj2_W, dd_, P->memKbProj_W, dd_, P->memKbProj_W + dd_, dd_, saved + L.wmax,
                 saved + L.wmax + 8, st, !use_chain(d, s->N)));
    }
    const bool fold = h2_mode() && use_chain(d, s->N);
    auto mx = [&](int k) { return wmax_ref(saved, L.wmax, k, fold); };
    if (units & U_READ) {
      pk.add(P->projX_W, d, 1, d, d, saved + L.wx_p, -1, -1, wfmt_plain(), mx(0).src, nullptr, mx(0).n, mx(0).out);
      if (use_chain(d, s->N)) {      // the chain kernel scales the A side by y: W1a and W1b are plain H2 weights there
        pk.add(P->memKbProj_W, d, 1, d, d, saved + L.w1a_p, -1, -1, 3, mx(2).src, nullptr, mx(2).n, mx(2).out);
        pk.add(P->memKbProj_W + (size_t)d * d, d, 1, d, d, saved + L.w1b_p, -1, -1, 3, mx(3).src, nullptr, mx(3).n, mx(3).out);
      } else {
        pk.add(P->memKbProj_W, d, 1, d, d, saved + L.w1a_p, -1, -1, wfmt_ymix());
        pk.add(P->memKbProj_W + (size_t)d * d, d, 1, d, d, saved + L.w1b_p, -1, -1, wfmt_ymix());
      }
      pk.add(P->memKbProj2_W, d, 1, d, d, saved + L.w2_p, -1, -1, wfmt_plain(), mx(1).src, nullptr, mx(1).n, mx(1).out);
      pk.add(P->projY_W, d, 1, d, d, saved + L.wy_p);
    }
    if (units & U_WRITE) {
      pk.add(P->newMemory_W, d, 1, write_in_dim(o, d), d, saved + L.wm_p);
      if (o->write_gate) pk.add(P->gate_W, d, 1, d, d, saved + L.wg_p);
      if (o->write_self_att) pk.add(P->selfCtrl_W, d, 1, d, d, saved + L.ws_p);
    }
    // a run that keeps its activations will be differentiated: the backward pass's transposed packs ride this launch
    if (keep && L.bwd_packs) CKI(add_bwd_packs(pk, o, s, P, saved + L.bwd_packs, make_bwd(o, s), saved, L, units, st));
    if (!(units & U_CONTROL)) return pk.n ? pk.run(st) : hipSuccess;
    pk.add(P->qInput_W, d, 1, d, d, saved + L.wq_p);
    if (o->control_feed_prev) {
      pk.add(P->contControl_W, d, 1, o->control_feed_inputs ? 2 * d : d, d, saved + L.wc_p);
      if (o->control_cont_act != MACX_ACT_NON) pk.add(P->contControl2_W, d, 1, d, d, saved + L.wc2_p);
    }
    for (int i = 0; i < (o->control_input_unshared ? p : 1); ++i) {
      if (pk.n == PACK_MAX) CK(pk.run(st));
      pk.add(P->qInputU_W + (size_t)i * d * d, d, 1, d, d, saved + L.wqU_p + (size_t)i * d * d);
    }
    CK(pk.run(st));
  }
  return MACX_OK;
}
}  // namespace

int macx_cell_begin(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                    const macx_inputs* in, float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                    int keep, void* stream) {
  ModeScope ms(o);
  CKI(check_impl(o, s));
  if (!dp || !P || !in || !saved) return MACX_EINVAL;
  if (misaligned(saved) || misaligned(in->knowledgeBase) || misaligned(in->words) || misaligned(in->vecQuestions))
    return MACX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const SavedLayout L = make_saved(o, s, keep);
  if (saved_floats < L.total) return MACX_ESMALL;
  (void)ws; (void)ws_floats;
  const int B = s->B, d = s->d, p = s->p;

  CKI(pack_forward_weights(o, s, P, saved, L, keep, U_ALL, st));

  // initial state (mac_cell.py:546-553), both tensors in one launch
  float* controls = saved + L.seg[MACX_SEG_CONTROLS];
  float* memories = saved + L.seg[MACX_SEG_MEMORIES];
  // ... and step 0's dropped memory (mac_cell.py:214-217, ops.py:679) where macx_cell_step would otherwise launch drop2 for it
  // (md_fused there: the whole cell without a gate; later steps get theirs from the write unit's linear)
  {
    const bool md0 = !o->write_gate;
    const DropSpec dm = o->memory_variational_dropout ? make_drop(dp->keep_memory, dp, SITE_MEM_VAR, 0) : make_drop(dp->keep_memory, dp, SITE_MEM, 0);
    const DropSpec dry = make_drop(dp->keep_read, dp, SITE_READ_MEM, 0);
    hipLaunchKernelGGL(init_states_kernel, dim3(64), dim3(256), 0, st, o->init_ctrl, P->initCtrl, controls, o->init_mem, P->initMem, memories,
                       in->vecQuestions, B, d, md0 ? saved + L.md : nullptr, (uint32_t)s->b0, dm, dry, dlog_of(s),
                       reinterpret_cast<uint32_t*>(saved + L.sync));
    CK(hipGetLastError());
  }

  // control inputs (mac_cell.py:442-448).  qInput is step-invariant; qInput{i} is batched over steps.
  {
    LinP l = lin_basic(in->vecQuestions, d, d, B, saved + L.wq_p, P->qInput_b, d, o->control_input_act, saved + L.ctrl_t, d);
    CK(small_linear_launch(l, 1, st));
    LinP u = lin_basic(saved + L.ctrl_t, d, d, B, saved + L.wqU_p, P->qInputU_b, d, MACX_ACT_NON, saved + L.cI, d);
    if (o->control_input_unshared) { u.zW = (size_t)d * d; u.zb = d; }
    u.zout = (size_t)B * d;
    CK(small_linear_launch(u, p, st));
  }
  if (!o->control_feed_prev) {
    // control is not recurrent (mac_cell.py:141-151): all p word attentions in one launch
    CtrlP c;
    c.B = B; c.S = s->S; c.d = d;
    c.cc = saved + L.cc; c.z_cc = (size_t)B * d;
    c.words = in->words; c.lengths = in->questionLengths;
    c.w = P->ctrlLogits_w; c.bias = P->ctrlLogits_b;
    c.att = saved + L.seg[MACX_SEG_ATT_QUESTION]; c.z_att = (size_t)B * s->S;
    c.control = controls + (size_t)B * d; c.z_ctl = (size_t)B * d;
    hipLaunchKernelGGL(control_attend_kernel, dim3(B, p), dim3(256), 0, st, c);
    CK(hipGetLastError());
  }
  return MACX_OK;
}

namespace {
// macx_cell_forward_chain_time: an event pair that receives the start / stop timestamps of every forward chain launch of the passes
// this thread runs while `on`
struct ChainProbe { hipEvent_t ev[2 * 64]; int n; bool on; };
inline ChainProbe* chain_probe() { static thread_local ChainProbe p = {}; return &p; }

int cell_step_impl(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                   const macx_inputs* in, float* saved, size_t saved_floats, int keep, int step, int units, void* stream, int pre = 0) {
  CKI(check_impl(o, s));
  if (!dp || !P || !in || !saved) return MACX_EINVAL;
  if (step < 0 || step >= s->p) return MACX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const SavedLayout L = make_saved(o, s, keep);
  if (saved_floats < L.total) return MACX_ESMALL;
  const int B = s->B, N = s->N, d = s->d;
  const size_t Bd = (size_t)B * d;
  const int i = step;

  float* controls = saved + L.seg[MACX_SEG_CONTROLS];
  float* memories = saved + L.seg[MACX_SEG_MEMORIES];
  const float* c_i = controls + (size_t)(i + 1) * Bd;
  const float* m_prev = memories + (size_t)i * Bd;
  float* m_new = memories + (size_t)(i + 1) * Bd;
  float* md = saved + L.md + (size_t)i * Bd;
  const bool md_fused = units == U_ALL && !o->write_gate;   // the write unit's linear also writes the next step's dropped memory
  float* y = saved + L.y + (size_t)i * Bd;
  float* X = saved + L.X + (size_t)i * L.act_stride;
  float* H1 = saved + L.H1 + (size_t)i * L.act_stride;
  float* I2 = saved + L.I2 + (size_t)i * L.act_stride;
  float* info = saved + L.seg[MACX_SEG_INFOS] + (size_t)i * Bd;
  const bool wdrop = dp->keep_write < 1.0f;
  float* info_raw = wdrop ? saved + L.info_raw + (size_t)i * Bd : info;

  // ---- control unit when it is recurrent (mac_cell.py:141-151, configs/args1.txt)
  if ((units & U_CONTROL) && o->control_feed_prev) {
    const float* prev = o->control_feed_prev_att ? controls + (size_t)i * Bd
                                                 : (i == 0 ? controls : saved + L.cc + (size_t)(i - 1) * Bd);
    float* cc_i = saved + L.cc + (size_t)i * Bd;
    const bool two = o->control_cont_act != MACX_ACT_NON;       // ops.linear stacks a second layer (ops.py:325-328)
    float* lin1 = two ? saved + L.cc_h + (size_t)i * Bd : cc_i;
    LinP l = lin_basic(prev, d, d, B, saved + L.wc_p, P->contControl_b, d, o->control_cont_act, lin1, d);
    if (o->control_feed_inputs) { l.seg[1] = LinSeg{saved + L.cI + (size_t)i * Bd, d, d, 0}; l.Ktot = 2 * d; }
    CK(small_linear_launch(l, 1, st));
    if (two) {
      LinP l2 = lin_basic(lin1, d, d, B, saved + L.wc2_p, P->contControl2_b, d, MACX_ACT_NON, cc_i, d);
      CK(small_linear_launch(l2, 1, st));
    }
    CtrlP c;
    c.B = B; c.S = s->S; c.d = d;
    c.cc = cc_i; c.z_cc = 0;
    c.words = in->words; c.lengths = in->questionLengths;
    c.w = P->ctrlLogits_w; c.bias = P->ctrlLogits_b;
    c.att = saved + L.seg[MACX_SEG_ATT_QUESTION] + (size_t)i * B * s->S; c.z_att = 0;
    c.control = controls + (size_t)(i + 1) * Bd; c.z_ctl = 0;
    hipLaunchKernelGGL(control_attend_kernel, dim3(B, 1), dim3(256), 0, st, c);
    CK(hipGetLastError());
  }
  // ---- read unit (mac_cell.py:209-277)
  if (units & U_READ) {
  // memory dropout (mac_cell.py:214-217) then the read-dropout of ops.mul's y input (ops.py:679)
  const DropSpec dm = o->memory_variational_dropout ? make_drop(dp->keep_memory, dp, SITE_MEM_VAR, 0)
                                                    : make_drop(dp->keep_memory, dp, SITE_MEM, i);
  const DropSpec dry = make_drop(dp->keep_read, dp, SITE_READ_MEM, i);
  // (md_fused: macx_cell_begin left step 0's dropped memory behind, the previous step's write unit every later one's)
  if (!md_fused) {
    hipLaunchKernelGGL(drop2_kernel, dim3(64), dim3(256), 0, st, m_prev, B, d, (uint32_t)s->b0, dm, dry, md, dlog_of(s));
    CK(hipGetLastError());
  }
  if (!((pre & 4) && h2_mode() && use_chain(d, s->N))) {     // (pre & 4: the chain launch's filler workgroups compute y, ChainPreP)
    LinP l = lin_basic(md, d, d, B, saved + L.wy_p, P->projY_b, d, MACX_ACT_NON, y, d);
    CK(small_linear_launch(l, 1, st));
  }
  // keep bits of the two [B,N,d] read-dropout sites of this step (ops.py:678 and ops.py:312 via :142)
  const bool rdrop = dp->keep_read < 1.0f;
  const size_t nwords = (size_t)B * N * d / 32;
  uint32_t* kb_bits = reinterpret_cast<uint32_t*>(saved + L.kb_bits + (size_t)i * L.bits_stride);
  uint32_t* att_bits = reinterpret_cast<uint32_t*>(saved + L.att_bits + (size_t)i * L.bits_stride);
  float* KBd = saved + L.KBd + (size_t)i * L.act_stride;
  if (h2_mode()) {
    // the same three products on H2 operands (macx_gemm_h2.hip.h): the knowledge base enters the format (through its
    // dropout site) once per step -- once per run without read dropout -- and X, H1, I2 never exist as fp32 tensors
    const int R = B * N, CB = d / 128;
    const H2View hX = h2_view(X, R, d), hH1 = h2_view(H1, R, d), hI2 = h2_view(I2, R, d);
    const H2View hKB = h2_view(rdrop ? KBd : saved + L.KBd, R, d);
    uint8_t* att_bytes = rdrop ? reinterpret_cast<uint8_t*>(att_bits) : nullptr;
    if (use_chain(d, s->N)) {
      // KB -> X -> H1 -> I2 -> logits in one launch (macx_chain_h2.hip.h)
      const ChainFwdP c = make_chain_fwd(o, s, dp, P, in, saved, L, keep, i, i, pre);
      ChainProbe& cp = *chain_probe();
      if (cp.on && cp.n < 64) {      // the kernel's own start / stop timestamps into the probe's event pair
        CK(chain_fwd_launch(c, st, cp.ev[2 * cp.n], cp.ev[2 * cp.n + 1]));
        ++cp.n;
      } else {
        CK(chain_fwd_launch(c, st));
      }
    } else {
    if (rdrop || i == 0) {
      H2FromP f;
      memset(&f, 0, sizeof(f));
      f.src = in->knowledgeBase; f.B = B; f.N = N; f.C = d; f.out = hKB;
      f.ldrop = dlog_of(s);
      f.first = (uint32_t)((size_t)s->b0 * N * f.ldrop);
      f.thr24 = 1u << 24; f.inv_keep = 1.0f; f.thr24_2 = 1u << 24;
      if (rdrop) {
        const DropSpec dk = make_drop(dp->keep_read, dp, SITE_READ_KB, i);
        const DropSpec da = make_drop(dp->keep_read, dp, SITE_READ_ATT, i);
        f.key = dk.key; f.thr24 = dk.thr24; f.inv_keep = dk.inv_keep; f.bits = kb_bits;
        f.key2 = da.key; f.thr24_2 = da.thr24; f.bytes2 = att_bytes;
        f.word = dp->mask_word;
      }
      CK(h2_from_f32(f, st));
    }
    GemmH2P g;
    memset(&g, 0, sizeof(g));
    g.B = B; g.N = N; g.K = d; g.Nout = d;
    g.dbg = kb_gemm_dbg() & (1 | 2 | 4 | 8 | 16 | 32 | 64 | 128);      // phase-timing knobs (results are wrong under a non-zero mask)
    g.e_inv_keep = rdrop ? 1.0f / dp->keep_read : 1.0f;
    // X = dropout(KB) Wx + bx  (ops.py:678,688)
    g.A = hKB;
    g.Wh = reinterpret_cast<const char*>(saved + L.wx_p); g.w_exp = reinterpret_cast<const int*>(saved + L.wx_p) + (size_t)d * d;
    g.out = hX; g.bias = P->projX_b; g.act = MACX_ACT_NON;
    if (rdrop || L.act_stride != 0 || i == 0) CK((kb_gemm_h2_launch<B_PLAIN, E_BIAS_ACT, false>(g, st)));
    // H1 = act( X (diag(y) W1a + W1b) + b1 )   (ops.py:703,718; mac_cell.py:237)
    g.A = hX; g.Wh = nullptr; g.w_exp = nullptr;
    g.Wt = saved + L.w1a_p; g.Wt2 = saved + L.w1b_p; g.w_max = saved + L.wmax + 2; g.y = y; g.ldy = d;
    g.out = hH1; g.bias = P->memKbProj_b; g.act = o->read_mem_act;
    CK((kb_gemm_h2_launch<B_YMIX_ROW, E_BIAS_ACT, false>(g, st)));
    // I2 = H1 W2 + b2 ; logits = dropout(act(I2 * c)) . w_k   (ops.py:326; mac_cell.py:248,262,266)
    g.A = hH1; g.Wt = nullptr; g.Wt2 = nullptr; g.y = nullptr;
    g.Wh = reinterpret_cast<const char*>(saved + L.w2_p); g.w_exp = reinterpret_cast<const int*>(saved + L.w2_p) + (size_t)d * d;
    g.out = hI2; g.bias = P->memKbProj2_b; g.act = o->read_ctrl_act;
    g.cvec = c_i; g.wvec = P->kbLogits_w; g.logit_part = saved + L.logit_part;
    g.e_bytes = att_bytes;
    CK((kb_gemm_h2_launch<B_PLAIN, E_I2_LOGIT, false>(g, st)));
    }
    (void)CB;
  } else {
  if (rdrop) {
    const uint32_t first = (uint32_t)((size_t)s->b0 * N * d);
    const DropSpec dk = make_drop(dp->keep_read, dp, SITE_READ_KB, i);
    const DropSpec da = make_drop(dp->keep_read, dp, SITE_READ_ATT, i);     // same keep probability, own stream
    hipLaunchKernelGGL(kb_dropout_kernel, dim3(2048), dim3(256), 0, st, in->knowledgeBase, (size_t)B * N * d / 4, dk.key, dk.thr24,
                       dk.inv_keep, first, KBd, kb_bits, da.key, att_bits, dp->mask_word);
    CK(hipGetLastError());
  }
  GemmP g;
  memset(&g, 0, sizeof(g));
  g.B = B; g.N = N; g.K = d; g.Nout = d;
  g.e_inv_keep = rdrop ? 1.0f / dp->keep_read : 1.0f;
  // X = dropout(KB) Wx + bx  (ops.py:678,688)
  g.A = rdrop ? KBd : in->knowledgeBase; g.lda = d;
  g.Wp = saved + L.wx_p;
  g.out = X; g.ldo = d; g.bias = P->projX_b; g.act = MACX_ACT_NON;
  // Without read dropout the projected knowledge base is the same at every step (the reference recomputes it,
  // ops.py:688); when the activations are not kept (inference) all steps share one X buffer, so step 0's is reused.
  if (rdrop || L.act_stride != 0 || i == 0) CK((kb_gemm<A_PLAIN, B_PLAIN, E_BIAS_ACT, false>(g, st)));
  // H1 = act( concat([X*y, X]) W1 + b1 ) = act( X (diag(y) W1a + W1b) + b1 )   (ops.py:703,718; mac_cell.py:237)
  g.A = X;
  g.Wp = saved + L.w1a_p; g.Wp2 = saved + L.w1b_p; g.y = y; g.ldy = d;
  g.out = H1; g.bias = P->memKbProj_b; g.act = o->read_mem_act;
  CK((kb_gemm<A_PLAIN, B_YMIX_ROW, E_BIAS_ACT, false>(g, st)));
  // I2 = H1 W2 + b2 ; logits = dropout(act(I2 * c)) . w_k   (ops.py:326; mac_cell.py:248,262,266)
  g.A = H1; g.Wp = saved + L.w2_p; g.Wp2 = nullptr; g.y = nullptr;
  g.out = I2; g.bias = P->memKbProj2_b; g.act = o->read_ctrl_act;
  g.cvec = c_i; g.wvec = P->kbLogits_w;
  g.logit_part = saved + L.logit_part;
  g.e_bits = rdrop ? att_bits : nullptr;
  CK((kb_gemm<A_PLAIN, B_PLAIN, E_I2_LOGIT, false>(g, st)));
  }
  // attention over the knowledge base + summary (mac_cell.py:266-275)
  {
    KbAttP a;
    a.B = B; a.N = N; a.d = d; a.nparts = use_chain(d, s->N) ? 1 : d / (16 * kb_gemm_nw());
    a.logit_part = saved + L.logit_part; a.bias = P->kbLogits_b;
    a.kb = in->knowledgeBase;
    a.att = saved + L.seg[MACX_SEG_ATT_KB] + (size_t)i * B * N;
    a.info = info_raw;
    hipLaunchKernelGGL(kb_attend_kernel, dim3(B, d / 128), dim3(KA_THREADS), 0, st, a);
    CK(hipGetLastError());
  }
  if (pre & 8) return MACX_OK;      // the write unit: on the next chain launch's filler workgroups (ChainPreP::wlin)
  }   // U_READ
  if (!(units & U_WRITE)) return MACX_OK;
  // write dropout (mac_cell.py:461-463); self.infos keeps the dropped value (mac_cell.py:474)
  if (wdrop) {
    const DropSpec dw = make_drop(dp->keep_write, dp, SITE_WRITE_INFO, i);
    hipLaunchKernelGGL(drop2_kernel, dim3(64), dim3(256), 0, st, (const float*)info_raw, B, d, (uint32_t)s->b0, dw, no_drop(), info, dlog_of(s));
    CK(hipGetLastError());
  }
  // ---- write unit (mac_cell.py:305-375), writeInputs = BOTH: act(concat([memory, info (, selfSmry)]) W + b)
  float* self_smry = nullptr;
  if (o->write_self_att) {
    // mac_cell.py:316-330.  selfControl = contControl (CONT) or the new control; histories hold the
    // initial state and steps 0..i-1 because the appends happen after write (mac_cell.py:472-474)
    const float* sctl = o->write_self_att_cont ? saved + L.cc + (size_t)i * Bd : c_i;
    float* sc = saved + L.sc + (size_t)i * Bd;
    LinP l = lin_basic(sctl, d, d, B, saved + L.ws_p, P->selfCtrl_b, d, MACX_ACT_NON, sc, d);
    CK(small_linear_launch(l, 1, st));
    self_smry = saved + L.self_smry + (size_t)i * Bd;
    SelfAttP q;
    q.B = B; q.d = d; q.nh = i + 1;
    q.sc = sc; q.C = controls; q.M = memories; q.w = P->selfLogits_w; q.bias = P->selfLogits_b;
    q.att = saved + L.seg[MACX_SEG_ATT_SELF] + (size_t)i * B * s->p; q.ld_att = s->p;
    q.smry = self_smry;
    hipLaunchKernelGGL(self_attend_kernel, dim3(B), dim3(256), 0, st, q);
    CK(hipGetLastError());
  }
  {
    float* wout = o->write_gate ? saved + L.mnew + (size_t)i * Bd : m_new;
    // (md_fused: the new memory is the next step's read-unit input: its two dropouts, mac_cell.py:214-217 and ops.py:679, ride the epilogue)
    LinP l = make_write_lin(o, s, dp, P, saved, L, i, info, wout, md_fused && i + 1 < s->p);
    if (self_smry) { l.seg[2] = LinSeg{self_smry, d, d, 0}; l.Ktot = 3 * d; }
    CK(small_linear_launch(l, 1, st));
    if (o->write_gate) {
      // z = sigmoid(control Wg + bg + gateBias); m = newMemory * z + memory * (1 - z)   (mac_cell.py:358-367)
      float* z = saved + L.seg[MACX_SEG_ATT_GATE] + (size_t)i * Bd;
      LinP gl = lin_basic(c_i, d, d, B, saved + L.wg_p, P->gate_b, d, MACX_ACT_SIGMOID, z, d);
      gl.bias_const = o->write_gate_bias;
      CK(small_linear_launch(gl, 1, st));
      hipLaunchKernelGGL(gate_mix_kernel, dim3(64), dim3(256), 0, st, (const float*)wout, (const float*)z, m_prev, Bd, m_new);
      CK(hipGetLastError());
    }
  }
  return MACX_OK;
}
}  // namespace

int macx_cell_step(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                   const macx_inputs* in, float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                   int keep, int step, void* stream) {
  ModeScope ms(o);
  (void)ws; (void)ws_floats;
  return cell_step_impl(o, s, dp, P, in, saved, saved_floats, keep, step, U_ALL, stream);
}

int macx_cell_forward(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                      const macx_inputs* in, float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                      int keep, void* stream) {
  CKI(macx_cell_begin(o, s, dp, P, in, saved, saved_floats, ws, ws_floats, keep, stream));
  ModeScope ms(o);
  // a training run's steps are sequenced here: stage 0 of the read unit's chain for step i + 1 (state-independent) rides the
  // launch of step i on its idle CUs (ChainPreP).  The step-wise entry point makes no assumption about what ran before it.
  const bool fill = h2_mode() && use_chain(s->d, s->N) && s->p <= 32 && s->B <= 128 && pre_fill_count(s->d, (size_t)s->B * s->N, device_cu_count()) > 0;
  const bool pre = fill && keep && dp && dp->keep_read < 1.0f;
  // the write unit's linear too, where it is one launch: the plain write unit of the published flag files
  const bool tail = fill && s->B <= 128 && !o->write_gate && !o->write_self_att && !o->control_feed_prev && dp && !(dp->keep_write < 1.0f) &&
                    tune_get(MACX_TUNE_PRE_FILL, 1) >= 1 && tune_get(MACX_TUNE_PRE_FILL, 1) != 2;
  for (int i = 0; i < s->p; ++i) {
    // bit 0: stage 0 was done by the previous launch; 1: this launch does the next step's; 2: this launch's fillers compute y;
    // 3: this step's write unit is left to the next launch; 4: this launch's fillers run the previous step's
    const int flags = (pre && i > 0 ? 1 : 0) | (pre && i + 1 < s->p ? 2 : 0) | (fill ? 4 : 0) | (tail && i + 1 < s->p ? 8 : 0) |
                      (tail && i > 0 ? 16 : 0);
    CKI(cell_step_impl(o, s, dp, P, in, saved, saved_floats, keep, i, U_ALL, stream, flags));
  }
  return MACX_OK;
}

// -------------------------------------------------------------------------------------------------
// phase 0: everything; 1: all of it except the deferred read-unit weight contractions; 2: only those (after a phase-1 call
// on the same buffers).  A data-parallel host launches the all-reduce of every gradient phase 1 completes on a side stream
// while phase 2 -- the last ~10 % of the backward pass -- still runs (macx.dp.OverlappedBuckets).
namespace {
// gradients that cross a unit's boundary when one unit is differentiated on its own (macx_read_bwd / macx_write_bwd)
struct UnitGrads {
  const float* d_info_in = nullptr;     // read: dL/d(info), [B,d]
  float* d_memory = nullptr;            // out: dL/d(memory input of the unit)
  float* d_control = nullptr;           // out: dL/d(control input of the unit)
  float* d_info_out = nullptr;          // write: dL/d(info) (before the write dropout)
};

// the backward pass's weight packs into `wT` (offsets: BwdLayout's first region), appended to the caller's pack list
int add_bwd_packs(Packer& pk, const macx_opts* o, const macx_shapes* s, const macx_params* P, float* wT, const BwdLayout& W,
                  const float* saved, const SavedLayout& L, int units, hipStream_t st) {
  const int d = s->d, p = s->p;
  const size_t dd = (size_t)d * d;
  const int win = write_in_dim(o, d);
  const int nU = o->control_input_unshared ? p : 1;

    const bool fold = h2_mode() && use_chain(d, s->N);        // (these packs ride the forward pass's pack launch: same rule as there)
    auto mx = [&](int k) { return wmax_ref(saved, L.wmax, k, fold); };
    if (units & U_READ) {
      pk.add(P->projX_W, 1, d, d, d, wT + W.wxT_p, -1, -1, wfmt_plain(), mx(0).src, nullptr, mx(0).n, mx(0).out);            // Wx^T
      if (use_chain(d, s->N)) {      // the chain kernel applies y to the accumulators: plain H2 weights
        pk.add(P->memKbProj_W, 1, d, d, d, wT + W.w1aT_p, -1, -1, 3, mx(2).src, nullptr, mx(2).n, mx(2).out);       // W1a^T
        pk.add(P->memKbProj_W + dd, 1, d, d, d, wT + W.w1bT_p, -1, -1, 3, mx(3).src, nullptr, mx(3).n, mx(3).out);  // W1b^T
      } else {
        pk.add(P->memKbProj_W, 1, d, d, d, wT + W.w1aT_p, -1, -1, wfmt_ymix());       // W1a^T
        pk.add(P->memKbProj_W + dd, 1, d, d, d, wT + W.w1bT_p, -1, -1, wfmt_ymix());  // W1b^T
      }
      pk.add(P->memKbProj2_W, 1, d, d, d, wT + W.w2T_p, -1, -1, wfmt_plain(), mx(1).src, nullptr, mx(1).n, mx(1).out);       // W2^T
      pk.add(P->projY_W, 1, d, d, d, wT + W.wyT);              // Wy^T
    }
    if (units & U_WRITE) {
      pk.add(P->newMemory_W, 1, d, d, win, wT + W.wmT);        // Wm^T: [d] -> [win]
      if (o->write_gate) pk.add(P->gate_W, 1, d, d, d, wT + W.wgT);
      if (o->write_self_att) pk.add(P->selfCtrl_W, 1, d, d, d, wT + W.wscT);
    }
    if (units & U_CONTROL) pk.add(P->qInput_W, 1, d, d, d, wT + W.wqT);
    if ((units & U_CONTROL) && o->control_feed_prev) {
      pk.add(P->contControl_W, 1, d, d, o->control_feed_inputs ? 2 * d : d, wT + W.wccT);   // Wc^T: [d] -> [d or 2d]
      if (o->control_cont_act != MACX_ACT_NON) pk.add(P->contControl2_W, 1, d, d, d, wT + W.wcc2T);
    }
    for (int i = 0; i < ((units & U_CONTROL) ? nU : 0); ++i) {
      if (pk.n == PACK_MAX) CK(pk.run(st));
      pk.add(P->qInputU_W + (size_t)i * dd, 1, d, d, d, wT + W.wqUT + (size_t)i * dd);
    }
  return MACX_OK;
}

int cell_backward_impl(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                       const macx_inputs* in, const float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                       const float* d_memory, const float* d_control, const macx_param_grads* GP,
                       const macx_input_grads* GI, int phase, int units, const UnitGrads* ug, void* stream) {
  if (phase < 0 || phase > 2) return MACX_EINVAL;
  CKI(check_impl(o, s));
  if (!dp || !P || !in || !saved || !ws || !GP || !GI) return MACX_EINVAL;
  if ((units & U_READ) && !GI->knowledgeBase) return MACX_EINVAL;
  if ((units & U_CONTROL) && (!GI->words || !GI->vecQuestions)) return MACX_EINVAL;
  if (units != U_ALL && (!ug || s->p != 1)) return MACX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  RowsumBatch rs;                     // bias-gradient row sums of this call: one launch at the end of each phase
  const SavedLayout L = make_saved(o, s, 1);
  const BwdLayout W = make_bwd(o, s);
  if (saved_floats < L.total || ws_floats < W.total) return MACX_ESMALL;
  SmallWgradBatch wb(ws + W.small_slab, W.small_slab_stride);       // weight gradients of the [B,d] linears: one launch at the end of phase 1
  const int B = s->B, N = s->N, d = s->d, p = s->p, S = s->S;
  const size_t Bd = (size_t)B * d;
  const size_t BNd = (size_t)B * N * d;
  const size_t dd = (size_t)d * d;
  const int win = write_in_dim(o, d);
  const int nrb = nrb_of(N, B, d);
  const bool rdrop = dp->keep_read < 1.0f;

  if (phase != 2) {
  // ---- weights in the layouts the backward kernels read: packed by the forward pass's pack launch into `saved`
  //      (SavedLayout::bwd_packs; a run is differentiated only if it kept its activations, and then it packed these too)
  if (!L.bwd_packs) return MACX_EINVAL;
  const float* wT = saved + L.bwd_packs;

  float* DM = ws + W.DM;
  float* DC = ws + W.DC;
  // dL/d(newMemory linear output) for all steps; with writeMemAct = NON it IS dL/dm_{1..p}
  float* dwlin_all = (o->write_mem_act == MACX_ACT_NON && !o->write_gate) ? DM + Bd : ws + W.dwlin;
  if (DC == DM + (size_t)(p + 1) * Bd && !misaligned(d_memory) && !misaligned(d_control)) {
    // (adjacent in the workspace: zeros and the incoming gradients in the last slabs in ONE launch)
    hipLaunchKernelGGL(bwd_init_kernel, dim3(fill_grid(2 * (size_t)(p + 1) * Bd)), dim3(256), 0, st, DM, Bd, p, d_memory, d_control, 0);
    CK(hipGetLastError());
  } else {
    CK(dev_zero(DM, (size_t)(p + 1) * Bd * sizeof(float), st));
    CK(dev_zero(DC, (size_t)(p + 1) * Bd * sizeof(float), st));
    if (d_memory) CK(dev_copy(DM + (size_t)p * Bd, d_memory, Bd * sizeof(float), st));
    if (d_control) CK(dev_copy(DC + (size_t)p * Bd, d_control, Bd * sizeof(float), st));
  }
  if ((units & U_CONTROL) && o->control_feed_prev) {
    CK(dev_zero(GI->words, (size_t)B * S * d * sizeof(float), st));
    CK(dev_zero(ws + W.dwc_part, Bd * sizeof(float), st));
    CK(dev_zero(ws + W.dccx, (size_t)(p + 1) * Bd * sizeof(float), st));
  }

  const float* controls = saved + L.seg[MACX_SEG_CONTROLS];
  const float* memories = saved + L.seg[MACX_SEG_MEMORIES];
  const float* infos = saved + L.seg[MACX_SEG_INFOS];
  const float* att_kb = saved + L.seg[MACX_SEG_ATT_KB];

  // the recurrent control unit differentiates through dL/dc_i inside iteration i; otherwise every step's dc / db_k partials are
  // reduced in one launch after the loop
  const bool dc_in_loop = (units & U_CONTROL) && o->control_feed_prev;
  // dKB of step i + 1 rides chain_bwd's launch of step i on the CUs that launch leaves idle (ChainDkbP, macx_chain_api.hip.h); what
  // the fillers leave out and step 0 run in chain_dkb_rest_launch after the last step.  Off (njobs = 0): the merged dKB launch.
  const DkbFillPlan dkb_plan = (units == U_ALL && h2_mode() && use_chain(d, s->N))
                                   ? dkb_fill_plan(d, (size_t)B * N, N, p, device_cu_count()) : DkbFillPlan{0, 0, 0};
  ChainDkbP dkb_q;
  memset(&dkb_q, 0, sizeof(dkb_q));
  if (dkb_plan.njobs) {
    const bool wd = dp->keep_write < 1.0f;
    dkb_q.njobs = dkb_plan.njobs; dkb_q.nfill = dkb_plan.nfill; dkb_q.nskip = dkb_plan.nskip; dkb_q.p = p;
    dkb_q.dX = reinterpret_cast<const char*>(ws + W.dX); dkb_q.dx_step = W.act_floats * sizeof(float);
    dkb_q.WxT = ChainW{reinterpret_cast<const char*>(wT + W.wxT_p), reinterpret_cast<const int*>(wT + W.wxT_p) + dd};
    dkb_q.bits = rdrop ? reinterpret_cast<const uint8_t*>(saved + L.kb_bits) : nullptr;
    dkb_q.bits_step = L.bits_stride * sizeof(uint32_t);
    dkb_q.inv_keep = rdrop ? 1.0f / dp->keep_read : 1.0f;
    dkb_q.att = att_kb; dkb_q.att_step = (size_t)B * N;
    dkb_q.dinfo = wd ? ws + W.dinfo : ws + W.dwin + d; dkb_q.ld_dinfo = wd ? d : win; dkb_q.dinfo_step = wd ? Bd : (size_t)B * win;
    dkb_q.out = GI->knowledgeBase;
    dkb_q.dbg = (kb_gemm_dbg() >> 22) & 31;
#ifdef MACX_FILL_PROF
    dkb_q.prof = reinterpret_cast<uint32_t*>(const_cast<float*>(saved) + L.sync) + 48;
#endif
  }
  for (int i = p - 1; i >= 0; --i) {
    const float* c_i = controls + (size_t)(i + 1) * Bd;
    const float* X = saved + L.X + (size_t)i * L.act_stride;
    const float* H1 = saved + L.H1 + (size_t)i * L.act_stride;
    const float* I2 = saved + L.I2 + (size_t)i * L.act_stride;
    const float* y = saved + L.y + (size_t)i * Bd;
    const float* dm_i = DM + (size_t)(i + 1) * Bd;   // dL/d m_i, complete at this point
    float* dm_prev = DM + (size_t)i * Bd;
    float* dwlin = dwlin_all + (size_t)i * Bd;
    float* dI2_i = ws + W.dI2 + (size_t)i * W.act_floats;
    float* dX_i = ws + W.dX + (size_t)i * W.act_floats;
    float* dwin = ws + W.dwin + (size_t)i * B * win;

    const float* dinfo = ug ? ug->d_info_in : nullptr;
    int ld_dinfo = d;
    if (units & U_WRITE) {
    // ---- write unit backward
    const float* dmnew = dm_i;              // gradient wrt the (post-activation) new memory
    const float* mnew_out = memories + (size_t)(i + 1) * Bd;
    if (o->write_gate) {
      // m_i = mnew * z + m_{i-1} * (1 - z)
      const float* z = saved + L.seg[MACX_SEG_ATT_GATE] + (size_t)i * Bd;
      mnew_out = saved + L.mnew + (size_t)i * Bd;
      float* dzpre = ws + W.dzpre + (size_t)i * Bd;
      hipLaunchKernelGGL(gate_bwd_kernel, dim3(64), dim3(256), 0, st, dm_i, z, mnew_out, memories + (size_t)i * Bd, Bd,
                         ws + W.tmpBd[2], ws + W.tmpBd[3], dzpre);
      CK(hipGetLastError());
      dmnew = ws + W.tmpBd[2];
      // dL/dc_i += dzpre Wg^T
      LinP gl = lin_basic(dzpre, d, d, B, wT + W.wgT, nullptr, d, MACX_ACT_NON, DC + (size_t)(i + 1) * Bd, d);
      gl.addend = DC + (size_t)(i + 1) * Bd; gl.ld_add = d;
      CK(small_linear_launch(gl, 1, st));
    }
    // dwlin = dmnew * act'(mnew) ; [dm_prev part | dinfo (| dselfSmry)] = dwlin Wm^T
    if (o->write_mem_act != MACX_ACT_NON || o->write_gate) {
      hipLaunchKernelGGL(mul_actgrad_kernel, dim3(64), dim3(256), 0, st, dmnew, mnew_out, o->write_mem_act, Bd, dwlin);
      CK(hipGetLastError());
    }
    {
      LinP l = lin_basic(dwlin, d, d, B, wT + W.wmT, nullptr, win, MACX_ACT_NON, dwin, win);
      CK(small_linear_launch(l, 1, st));
    }
    // d(info) through the write dropout (mac_cell.py:463); without write dropout it is a column view of dwin
    dinfo = dwin + d;
    ld_dinfo = win;
    if (dp->keep_write < 1.0f) {
      hipLaunchKernelGGL(copy_cols_drop_kernel, dim3(64), dim3(256), 0, st, (const float*)dwin, win, d, B, d, (uint32_t)s->b0,
                         make_drop(dp->keep_write, dp, SITE_WRITE_INFO, i), ws + W.dinfo + (size_t)i * Bd, dlog_of(s));
      CK(hipGetLastError());
      dinfo = ws + W.dinfo + (size_t)i * Bd;
      ld_dinfo = d;
    }

    if (o->write_self_att) {
      SelfAttBwdP q;
      q.B = B; q.d = d; q.nh = i + 1;
      q.dsmry = dwin + 2 * d; q.ld_ds = win;
      q.sc = saved + L.sc + (size_t)i * Bd; q.C = controls; q.M = memories; q.w = P->selfLogits_w;
      q.att = saved + L.seg[MACX_SEG_ATT_SELF] + (size_t)i * B * p; q.ld_att = p;
      q.DMh = DM; q.DCh = DC;
      q.dsc = ws + W.dsc + (size_t)i * Bd;
      q.dw_part = ws + W.dws_part + (size_t)i * Bd;
      q.db_part = ws + W.dbs_part + (size_t)i * B;
      hipLaunchKernelGGL(self_attend_bwd_kernel, dim3(B), dim3(256), 0, st, q);
      CK(hipGetLastError());
    }
    }   // U_WRITE
    if (!(units & U_READ)) {
      // the write unit alone: dL/d(memory) = dwin[:, :d] (+ dm (1 - z) under the gate), dL/d(info), dL/d(control) (the gate's)
      const size_t pitch = (size_t)win * sizeof(float), wbytes = (size_t)d * sizeof(float);
      CK(dev_copy2d(ug->d_memory, wbytes, dwin, pitch, wbytes, B, st));
      if (o->write_gate) CK(axpy(ws + W.tmpBd[3], Bd, ug->d_memory, st));
      CK(dev_copy2d(ug->d_info_out, wbytes, dinfo, (size_t)ld_dinfo * sizeof(float), wbytes, B, st));
      CK(dev_copy(ug->d_control, DC + (size_t)(i + 1) * Bd, Bd * sizeof(float), st));
      continue;
    }

    // ---- read unit backward
    hipLaunchKernelGGL(kb_att_da_kernel, dim3((B * N + 3) / 4), dim3(256), 0, st, dinfo, ld_dinfo, in->knowledgeBase, B, N, d,
                       ws + W.da);
    CK(hipGetLastError());
    if (h2_mode()) {
      const int R = B * N, CB = d / 128;
      const H2View hI2 = h2_view(I2, R, d), hH1 = h2_view(H1, R, d), hX = h2_view(X, R, d);
      const H2View hdI2 = h2_view(dI2_i, R, d), hdI1 = h2_view(ws + W.dI1 + (size_t)i * W.dI1_stride, R, d), hdX = h2_view(dX_i, R, d);
      const bool chain = use_chain(d, s->N);
      if (!(chain && W.chain_sums)) {
        ReadAttBwdH2P r;
        r.dl = nullptr; r.no_out = chain ? 1 : 0;    // chain with tiny N: only dc / dw_k / db2 / db_k (dI2 comes from the chain kernel)
        r.B = B; r.N = N; r.d = d;
        r.att = att_kb + (size_t)i * B * N; r.da = ws + W.da; r.I2 = hI2; r.c = c_i; r.wk = P->kbLogits_w;
        r.act = o->read_ctrl_act;
        r.bytes = rdrop ? reinterpret_cast<const uint8_t*>(saved + L.att_bits + (size_t)i * L.bits_stride) : nullptr;
        r.inv_keep = rdrop ? 1.0f / dp->keep_read : 1.0f;
        r.dI2 = hdI2;
        r.dc = DC + (size_t)(i + 1) * Bd;
        r.dwk_part = ws + W.dwk_part + (size_t)i * W.dwk_rows * d;
        r.db2_part = ws + W.db2_part + (size_t)i * W.dwk_rows * d;
        r.dbk_part = ws + W.dbk_part + (size_t)i * B;
        hipLaunchKernelGGL(read_att_bwd_h2_kernel, dim3(B, d / 128), dim3(RABH_THREADS), 0, st, r);
        CK(hipGetLastError());
      }
      GemmH2P g;
      memset(&g, 0, sizeof(g));
      g.B = B; g.N = N; g.K = d; g.Nout = d;
      g.dbg = kb_gemm_dbg() & (1 | 2 | 4 | 8 | 16 | 32 | 64 | 128);      // phase-timing knobs (results are wrong under a non-zero mask)
      g.e_inv_keep = rdrop ? 1.0f / dp->keep_read : 1.0f;
      if (chain) {
        // dl -> dI2 -> dI1 -> dX in one launch (macx_chain_h2.hip.h)
        auto wref = [&](size_t off) { return ChainW{reinterpret_cast<const char*>(wT + off), reinterpret_cast<const int*>(wT + off) + dd}; };
        ChainBwdP c;
        memset(&c, 0, sizeof(c));
        c.M = R; c.N = N; c.d = d;
        c.dbg = (kb_gemm_dbg() >> 17) & 31;
        c.att = att_kb + (size_t)i * B * N; c.da = ws + W.da;
        c.I2 = hI2; c.c = c_i; c.wk = P->kbLogits_w; c.act2 = o->read_ctrl_act;
        if (W.chain_sums) {
          c.dwk_part = ws + W.dwk_part + (size_t)i * W.dwk_rows * d;
          c.db2_part = ws + W.db2_part + (size_t)i * W.dwk_rows * d;
          c.dc_part = ws + W.dc_part + (size_t)i * W.dwk_rows * 3 * d; c.dls_part = ws + W.dls_part + (size_t)i * W.dwk_rows * 3;
        }
        c.bytes2 = rdrop ? reinterpret_cast<const uint8_t*>(saved + L.att_bits + (size_t)i * L.bits_stride) : nullptr;
        c.inv2 = rdrop ? 1.0f / dp->keep_read : 1.0f;
        c.dI2 = hdI2;
        c.W2T = wref(W.w2T_p); c.H1 = hH1; c.act1 = o->read_mem_act;
        c.dI1 = hdI1; c.db1_part = ws + W.db1_part + (size_t)i * W.db_rows * d;
        c.W1aT = wref(W.w1aT_p); c.W1bT = wref(W.w1bT_p); c.y = y;
        c.dX = hdX; c.dbx_part = ws + W.dbx_part + (size_t)i * W.db_rows * d;
        if (W.sb_deferred) { c.X = hX; c.dy_part = ws + W.dyc_part; }
        if (dkb_plan.njobs && i + 1 < p) { c.dkb = dkb_q; c.dkb.step = i + 1; }
        CK(chain_bwd_launch(c, st));
        if (W.chain_sums && (dc_in_loop || (W.sb_deferred && !W.dy_in_linear))) {
          // the per-tile partials of this step: dL/dc_i += read-unit part, db_k partials, dy_i (the next launch needs dy_i)
          DcReduceP q;
          memset(&q, 0, sizeof(q));
          q.B = B; q.N = N; q.d = d;
          q.dc_part = ws + W.dc_part + (size_t)i * W.dwk_rows * 3 * d; q.dls_part = ws + W.dls_part + (size_t)i * W.dwk_rows * 3;
          q.dc = DC + (size_t)(i + 1) * Bd; q.dbk_part = ws + W.dbk_part + (size_t)i * B;
          q.tile_shift = chain_tile_shift(d, (size_t)B * N);
          if (W.sb_deferred && !W.dy_in_linear) { q.dy_part = ws + W.dyc_part; q.dy = ws + W.DY + (size_t)i * Bd; }
          hipLaunchKernelGGL(dc_reduce_kernel, dim3(B, 1), dim3(128), 0, st, q);
          CK(hipGetLastError());
        }
      } else {
      // dI1 = (dI2 W2^T) * act'(H1) ; db1 partials
      g.A = hdI2;
      g.Wh = reinterpret_cast<const char*>(wT + W.w2T_p); g.w_exp = reinterpret_cast<const int*>(wT + W.w2T_p) + dd;
      g.out = hdI1; g.aux = hH1; g.act = o->read_mem_act;
      g.colsum_part = ws + W.db1_part + (size_t)i * B * nrb * d;
      CK((kb_gemm_h2_launch<B_PLAIN, E_MUL_DACT, true>(g, st)));
      // dX = dI1 (diag(y) W1a + W1b)^T ; dbx partials
      g.A = hdI1; g.Wh = nullptr; g.w_exp = nullptr;
      g.Wt = wT + W.w1aT_p; g.Wt2 = wT + W.w1bT_p; g.w_max = saved + L.wmax + 2; g.y = y; g.ldy = d;
      g.out = hdX;
      g.colsum_part = ws + W.dbx_part + (size_t)i * B * nrb * d;
      CK((kb_gemm_h2_launch<B_YMIX_COL, E_PLAIN, true>(g, st)));
      }
      // S_b = X_b^T dI1_b -> dW1a / dW1b slabs and dy partials (deferred: one launch over all steps in phase 2, dy from the chain kernel)
      if (!W.sb_deferred) {
        SbH2P q;
        memset(&q, 0, sizeof(q));
        q.nsteps = 1;
        q.B = B; q.N = N; q.d = d; q.qpg = sb_qpg(B, N);
        q.X = hX; q.dI1 = hdI1;
        q.y = y; q.W1a = P->memKbProj_W;
        q.dW1a_part = ws + W.slab_w1a + (size_t)i * W.ngroup * dd;
        q.dW1b_part = ws + W.slab_w1b + (size_t)i * W.ngroup * dd;
        q.dy_part = ws + W.dy_part;
        q.dbg = kb_gemm_dbg();
        CK(sb_h2_launch(q, st));
      }
      // dKB = sum_i (dX_i Wx^T) * kbmask_i + att_i (x) dinfo_i: ONE launch over all steps after step 0 -- or, where chain_bwd's
      // launches carried the products of steps p - 1 .. 1 on their idle CUs, the closing launch of that route
      if (i == 0 && dkb_plan.njobs) {
        CK(chain_dkb_rest_launch(dkb_q, B * N, N, d, st));
      } else if (i == 0) {
        const int i0 = 0;
        g.A = h2_view(ws + W.dX + (size_t)i0 * W.act_floats, B * N, d); g.Wt = nullptr; g.Wt2 = nullptr; g.y = nullptr;
        g.Wh = reinterpret_cast<const char*>(wT + W.wxT_p); g.w_exp = reinterpret_cast<const int*>(wT + W.wxT_p) + dd;
        g.nsteps = p; g.a_step_bytes = W.act_floats * sizeof(float);
        g.a_row_exp = chain ? 1 : 0;          // chain_bwd_kernel gives a row of dX ONE exponent: the merged launch may fold once per step
        g.out_f32 = GI->knowledgeBase; g.ldo = d;
        const bool wd = dp->keep_write < 1.0f;
        g.dr = wd ? ws + W.dinfo : ws + W.dwin + d; g.ld_dr = wd ? d : win; g.dr_step = wd ? Bd : (size_t)B * win;
        if (!(units & U_WRITE)) { g.dr = dinfo; g.ld_dr = d; g.dr_step = Bd; }
        else g.dr += (size_t)i0 * g.dr_step;
        g.att = att_kb + (size_t)i0 * B * N; g.att_step = (size_t)B * N;
        g.e_bits = rdrop ? reinterpret_cast<const uint32_t*>(saved + L.kb_bits + (size_t)i0 * L.bits_stride) : nullptr;
        g.bits_step_words = L.bits_stride;
        g.accumulate = 0;
        g.colsum_part = nullptr; g.aux = H2View{nullptr, 0, 0};
        CK((kb_gemm_h2_launch<B_PLAIN, E_DKB, false>(g, st)));
      }
    } else {
    {
      ReadAttBwdP r;
      r.B = B; r.N = N; r.d = d; r.b0 = s->b0;
      r.att = att_kb + (size_t)i * B * N; r.da = ws + W.da; r.I2 = I2; r.c = c_i; r.wk = P->kbLogits_w;
      r.act = o->read_ctrl_act;
      r.bits = rdrop ? reinterpret_cast<const uint32_t*>(saved + L.att_bits + (size_t)i * L.bits_stride) : nullptr;
      r.inv_keep = rdrop ? 1.0f / dp->keep_read : 1.0f;
      r.dI2 = dI2_i;
      r.dc = DC + (size_t)(i + 1) * Bd;   // dL/dc_i += read-unit part
      r.dwk_part = ws + W.dwk_part + (size_t)i * Bd;
      r.db2_part = ws + W.db2_part + (size_t)i * Bd;
      r.dbk_part = ws + W.dbk_part + (size_t)i * B;
      hipLaunchKernelGGL(read_att_bwd_kernel, dim3(B, d / 128), dim3(RAB_THREADS), 0, st, r);
      CK(hipGetLastError());
    }
    GemmP g;
    memset(&g, 0, sizeof(g));
    g.B = B; g.N = N; g.K = d; g.Nout = d;
    g.e_inv_keep = rdrop ? 1.0f / dp->keep_read : 1.0f;
    const uint32_t* kb_bits = rdrop ? reinterpret_cast<const uint32_t*>(saved + L.kb_bits + (size_t)i * L.bits_stride) : nullptr;
    // dI1 = (dI2 W2^T) * act'(H1) ; db1 partials
    g.A = dI2_i; g.lda = d; g.Wp = wT + W.w2T_p;
    g.out = ws + W.dI1; g.ldo = d; g.aux = H1; g.act = o->read_mem_act;
    g.colsum_part = ws + W.db1_part + (size_t)i * B * nrb * d;
    CK((kb_gemm<A_PLAIN, B_PLAIN, E_MUL_DACT, true>(g, st)));
    // dX = dI1 (diag(y) W1a + W1b)^T ; dbx partials
    g.A = ws + W.dI1; g.Wp = wT + W.w1aT_p; g.Wp2 = wT + W.w1bT_p; g.y = y; g.ldy = d;
    g.out = dX_i; g.aux = nullptr;
    g.colsum_part = ws + W.dbx_part + (size_t)i * B * nrb * d;
    CK((kb_gemm<A_PLAIN, B_YMIX_COL, E_PLAIN, true>(g, st)));
    // S_b = X_b^T dI1_b -> dW1a / dW1b slabs and dy partials
    {
      SbP q;
      q.B = B; q.N = N; q.d = d; q.qpg = sb_qpg(B, N);
      q.X = X; q.dI1 = ws + W.dI1; q.y = y; q.W1a = P->memKbProj_W;
      q.dW1a_part = ws + W.slab_w1a + (size_t)i * W.ngroup * dd;
      q.dW1b_part = ws + W.slab_w1b + (size_t)i * W.ngroup * dd;
      q.dy_part = ws + W.dy_part;
      CK((gemm_split_mode() && !(kb_gemm_dbg() & 256)) ? sb6_wgrad_launch(q, st) : sb_wgrad_launch(q, st));   // dbg 256: f32 kernel
    }
    // dKB (+)= (dX Wx^T) * kbmask + att * dinfo
    g.A = dX_i; g.Wp = wT + W.wxT_p; g.Wp2 = nullptr; g.y = nullptr;
    g.out = GI->knowledgeBase; g.aux = dinfo; g.ld_aux = ld_dinfo; g.att = att_kb + (size_t)i * B * N;
    g.e_bits = kb_bits;
    g.accumulate = (i != p - 1);
    g.colsum_part = nullptr;
    CK((kb_gemm<A_PLAIN, B_PLAIN, E_DKB, false>(g, st)));
    }
    // dy -> d(md) -> dL/d m_{i-1} = dwin[:, :d] + (dy Wy^T) * memmask * readmask
    float* DYi = ws + W.DY + (size_t)i * Bd;
    if (!(h2_mode() && W.sb_deferred)) {
      hipLaunchKernelGGL(sum_parts_kernel, dim3(256), dim3(256), 0, st, (const float*)(ws + W.dy_part), (h2_mode() ? SBH_CW / 2 : 2) * d / 128, Bd, DYi);
      CK(hipGetLastError());
    }
    {
      // with self attention DM[i] already holds the parts later steps sent to this memory: accumulate
      const bool acc_prev = (units & U_WRITE) && (o->write_self_att || o->write_gate);
      LinP l = lin_basic(DYi, d, d, B, wT + W.wyT, nullptr, d, MACX_ACT_NON, acc_prev ? ws + W.tmpBd[0] : dm_prev, d);
      l.use_drop = 1; l.drop_ld = dlog_of(s);
      l.d1 = o->memory_variational_dropout ? make_drop(dp->keep_memory, dp, SITE_MEM_VAR, 0)
                                           : make_drop(dp->keep_memory, dp, SITE_MEM, i);
      l.d2 = make_drop(dp->keep_read, dp, SITE_READ_MEM, i);
      l.drop_row0 = (uint32_t)s->b0;
      if (units & U_WRITE) { l.addend = dwin; l.ld_add = win; }
      const bool part_form = h2_mode() && W.dy_in_linear;
      if (part_form) { l.part = ws + W.dyc_part; l.part_N = N; l.part_sum = DYi; l.part_shift = chain_tile_shift(d, (size_t)B * N); }
      if (part_form) {
        CK(small_linear_part_launch(l, st));
      } else {
        CK(small_linear_launch(l, 1, st));
      }
      if (acc_prev) {
        CK(axpy(ws + W.tmpBd[0], Bd, dm_prev, st));
        if (o->write_gate) CK(axpy(ws + W.tmpBd[3], Bd, dm_prev, st));   // dm * (1 - z)
      }
    }
    // ---- recurrent control backward for this step (dL/dc_i is complete now)
    if ((units & U_CONTROL) && o->control_feed_prev) {
      const int cin = o->control_feed_inputs ? 2 * d : d;
      const bool two = o->control_cont_act != MACX_ACT_NON;
      if (o->write_self_att && !o->write_self_att_cont) {
        LinP l = lin_basic(ws + W.dsc + (size_t)i * Bd, d, d, B, wT + W.wscT, nullptr, d, MACX_ACT_NON, DC + (size_t)(i + 1) * Bd, d);
        l.addend = DC + (size_t)(i + 1) * Bd; l.ld_add = d;
        CK(small_linear_launch(l, 1, st));
      }
      CtrlBwdP c;
      c.B = B; c.S = S; c.d = d; c.nz = 1;
      c.dcontrol = DC + (size_t)(i + 1) * Bd; c.z_dc = 0;
      c.cc = saved + L.cc + (size_t)i * Bd; c.z_cc = 0;
      c.att = saved + L.seg[MACX_SEG_ATT_QUESTION] + (size_t)i * B * S; c.z_att = 0;
      c.words = in->words; c.w = P->ctrlLogits_w;
      c.dl = ws + W.ctrl_dl;
      c.dcc = ws + W.dcc + (size_t)i * Bd; c.z_dcc = 0;
      c.dwords = GI->words; c.acc_words = 1;
      c.dw_part = ws + W.dwc_part; c.db_part = ws + W.dbc_part + (size_t)i * B;
      hipLaunchKernelGGL(control_bwd_dl_kernel, dim3(B, 1), dim3(256), 0, st, c);
      hipLaunchKernelGGL(control_bwd_apply_kernel, dim3(B, d / 64), dim3(256), 0, st, c);
      CK(hipGetLastError());
      float* dcc_i = ws + W.dcc + (size_t)i * Bd;
      // parts of dL/dcc_i that did not come through the word attention
      if (!o->control_feed_prev_att) CK(axpy(ws + W.dccx + (size_t)(i + 1) * Bd, Bd, dcc_i, st));
      if (o->write_self_att && o->write_self_att_cont) {
        LinP l = lin_basic(ws + W.dsc + (size_t)i * Bd, d, d, B, wT + W.wscT, nullptr, d, MACX_ACT_NON, dcc_i, d);
        l.addend = dcc_i; l.ld_add = d;
        CK(small_linear_launch(l, 1, st));
      }
      // through contControl(_2): dlin1 = (dcc Wc2^T) * act'(h)  or  dcc
      float* dlin1 = ws + W.dlin1 + (size_t)i * Bd;
      if (two) {
        LinP l2 = lin_basic(dcc_i, d, d, B, wT + W.wcc2T, nullptr, d, MACX_ACT_NON, dlin1, d);
        l2.actgrad_src = saved + L.cc_h + (size_t)i * Bd; l2.actgrad_act = o->control_cont_act; l2.ld_ag = d;
        CK(small_linear_launch(l2, 1, st));
      } else {
        CK(dev_copy(dlin1, dcc_i, Bd * sizeof(float), st));
      }
      // dx = dlin1 Wc^T = [d prev | d cI_i]
      LinP lx = lin_basic(dlin1, d, d, B, wT + W.wccT, nullptr, cin, MACX_ACT_NON, ws + W.dxc, cin);
      CK(small_linear_launch(lx, 1, st));
      float* dprev_dst = o->control_feed_prev_att ? DC + (size_t)i * Bd : (i == 0 ? DC : ws + W.dccx + (size_t)i * Bd);
      hipLaunchKernelGGL(copy_cols_drop_kernel, dim3(64), dim3(256), 0, st, (const float*)(ws + W.dxc), cin, 0, B, d, 0u, no_drop(),
                         ws + W.tmpBd[0]);
      CK(hipGetLastError());
      CK(axpy(ws + W.tmpBd[0], Bd, dprev_dst, st));
      if (o->control_feed_inputs) {
        hipLaunchKernelGGL(copy_cols_drop_kernel, dim3(64), dim3(256), 0, st, (const float*)(ws + W.dxc), cin, d, B, d, 0u, no_drop(),
                           ws + W.dcI + (size_t)i * Bd);
        CK(hipGetLastError());
      } else {
        CK(dev_zero(ws + W.dcI + (size_t)i * Bd, Bd * sizeof(float), st));
      }
    }
  }
  if ((units & U_READ) && h2_mode() && W.chain_sums && !(dc_in_loop || (W.sb_deferred && !W.dy_in_linear))) {
    DcReduceP q;
    memset(&q, 0, sizeof(q));
    q.B = B; q.N = N; q.d = d;
    q.dc_part = ws + W.dc_part; q.dls_part = ws + W.dls_part; q.dc = DC + Bd; q.dbk_part = ws + W.dbk_part;
    q.part_step = W.dwk_rows * 3 * d; q.dls_step = W.dwk_rows * 3; q.dc_step = Bd; q.dbk_step = B;
    q.tile_shift = chain_tile_shift(d, (size_t)B * N);
    hipLaunchKernelGGL(dc_reduce_kernel, dim3(B, p), dim3(128), 0, st, q);
    CK(hipGetLastError());
  }
  if (units == U_ALL) {
  if (o->write_self_att && !o->write_self_att_cont && !o->control_feed_prev) {
    // selfControl = the NEW control: dL/dc_i += dsc_i Ws^T before the word attention is differentiated
    LinP l = lin_basic(ws + W.dsc, d, d, B, wT + W.wscT, nullptr, d, MACX_ACT_NON, DC + Bd, d);
    l.seg[0].zstride = Bd; l.zout = Bd; l.addend = DC + Bd; l.ld_add = d; l.zadd = Bd;
    CK(small_linear_launch(l, p, st));
  }
  if (o->control_feed_prev) {
    // word-attention parameter partials were accumulated step by step
    CK(rs.add(ws + W.dwc_part, B, d, d, GP->ctrlLogits_w, st));
    CK(rs.add(ws + W.dbc_part, p * B, 1, 1, GP->ctrlLogits_b, st));
    // contControl weights: one contraction over all p*B rows per input segment
    const bool two = o->control_cont_act != MACX_ACT_NON;
    const float* prev_all = o->control_feed_prev_att ? controls : nullptr;   // rows of step i = c_{i-1} = controls[i]
    if (o->control_feed_prev_att) {
      CKI(wgrad_impl(prev_all, d, ws + W.dlin1, d, p * B, d, d, GP->contControl_W, ws + W.small_slab, st));
    } else {
      // prev of step 0 is the initial control, prev of step i > 0 is cc_{i-1}
      CKI(wgrad_impl(controls, d, ws + W.dlin1, d, B, d, d, GP->contControl_W, ws + W.small_slab, st));
      if (p > 1) {
        CKI(wgrad_impl(saved + L.cc, d, ws + W.dlin1 + Bd, d, (p - 1) * B, d, d, ws + W.tmp_dd, ws + W.small_slab, st));
        CK(axpy(ws + W.tmp_dd, dd, GP->contControl_W, st));
      }
    }
    if (o->control_feed_inputs)
      CKI(wgrad_impl(saved + L.cI, d, ws + W.dlin1, d, p * B, d, d, GP->contControl_W + dd, ws + W.small_slab, st));
    CK(rs.add(ws + W.dlin1, p * B, d, d, GP->contControl_b, st));
    if (two) {
      CKI(wgrad_impl(saved + L.cc_h, d, ws + W.dcc, d, p * B, d, d, GP->contControl2_W, ws + W.small_slab, st));
      CK(rs.add(ws + W.dcc, p * B, d, d, GP->contControl2_b, st));
    }
  }
  // ---- control unit backward.  Not recurrent: every control depends on the question only, so all
  // p steps are handled together.
  if (!o->control_feed_prev) {
    CtrlBwdP c;
    c.B = B; c.S = S; c.d = d; c.nz = p;
    c.dcontrol = DC + Bd; c.z_dc = Bd;
    c.cc = saved + L.cc; c.z_cc = Bd;
    c.att = saved + L.seg[MACX_SEG_ATT_QUESTION]; c.z_att = (size_t)B * S;
    c.words = in->words; c.w = P->ctrlLogits_w;
    c.dl = ws + W.ctrl_dl;
    c.dcc = ws + W.dcc; c.z_dcc = Bd;
    c.dwords = GI->words; c.acc_words = 0;
    c.dw_part = ws + W.dwc_part; c.db_part = ws + W.dbc_part;
    hipLaunchKernelGGL(control_bwd_dl_kernel, dim3(B, p), dim3(256), 0, st, c);
    hipLaunchKernelGGL(control_bwd_apply_kernel, dim3(B, d / 64), dim3(256), 0, st, c);
    CK(hipGetLastError());
    CK(rs.add(ws + W.dwc_part, B, d, d, GP->ctrlLogits_w, st));
    CK(rs.add(ws + W.dbc_part, p * B, 1, 1, GP->ctrlLogits_b, st));
  }
  if (o->write_self_att) {
    // the self-attention control projection reads contControl (== controlInput here) or the control:
    // d(source)_i += dsc_i Ws^T, in place, all steps in one launch
    float* dst = o->write_self_att_cont ? ws + W.dcc : DC + Bd;
    LinP l = lin_basic(ws + W.dsc, d, d, B, wT + W.wscT, nullptr, d, MACX_ACT_NON, dst, d);
    l.seg[0].zstride = Bd; l.zout = Bd; l.addend = dst; l.ld_add = d; l.zadd = Bd;
    if (o->write_self_att_cont && !o->control_feed_prev) CK(small_linear_launch(l, p, st));
    const float* src = o->write_self_att_cont ? saved + L.cc : controls + Bd;
    CKI(wb.add(src, d, ws + W.dsc, d, p * B, d, d, GP->selfCtrl_W, st));
    CK(rs.add(ws + W.dsc, p * B, d, d, GP->selfCtrl_b, st));
    CK(rs.add(ws + W.dws_part, p * B, d, d, GP->selfLogits_w, st));
    CK(rs.add(ws + W.dbs_part, p * B, 1, 1, GP->selfLogits_b, st));
  }
  }   // U_ALL
  if (o->write_gate && (units & U_WRITE)) {
    CKI(wb.add(controls + Bd, d, ws + W.dzpre, d, p * B, d, d, GP->gate_W, st));
    CK(rs.add(ws + W.dzpre, p * B, d, d, GP->gate_b, st));
  }
  if (units == U_ALL) {
  // ---- control inputs backward (mac_cell.py:442-448): dt = sum_i dcI_i WqU_i^T ; du = dt * act'(t)
  const float* ctrl_t = saved + L.ctrl_t;
  float* dcI_sum = ws + W.tmpBd[1];
  if (o->control_input_unshared) {
    // dt = sum_i dcI_i WqU_i^T: one linear over K = p d (the per-step inputs read as one [B, p d] operand, the per-step
    // packed transposes are contiguous = one packed [p d, d] matrix)
    // dt = sum_i dcI_i WqU_i^T: one batched launch of the p products (1536 workgroups at p = 12, one batch of operand loads each)
    // and a fixed-order sum over the steps.  (Rounds 2-5 ran it as ONE linear over K = p d: 128 workgroups whose waves walked 96
    // k-groups in twelve dependent load batches -- 27.6 us of latency for 0.4 GFLOP.)
    {
      LinP li = lin_basic(ws + W.dcI, d, d, B, wT + W.wqUT, nullptr, d, MACX_ACT_NON, ws + W.dt_part, d);
      li.seg[0].zstride = Bd; li.zW = dd; li.zout = Bd;
      CK(small_linear_launch(li, p, st));
      // ... which also leaves du = dt * act'(t) (the mul_actgrad launch of the shared-weights path below)
      hipLaunchKernelGGL(sum_parts_kernel, dim3(256), dim3(256), 0, st, (const float*)(ws + W.dt_part), p, Bd, ws + W.dt, ctrl_t,
                         (int)o->control_input_act, ws + W.du);
      CK(hipGetLastError());
    }
    // the p weight gradients ctrl_t^T dcI_i share A: one batched launch (B rows -> a single split, no slabs)
    if (gemm_split_mode() && !(kb_gemm_dbg() & 128) && wgrad_splits(B, d, d) == 1) {
      TnP t;
      memset(&t, 0, sizeof(t));
      t.M = B; t.Kd = d; t.Jd = d; t.nsplit = 1; t.rows_per_split = rows_per_split(B, 1);
      t.A = ctrl_t; t.lda = d; t.a_mod = B; t.G = ws + W.dcI; t.ldg = d;
      t.part = GP->qInputU_W; t.nz = p; t.zG = Bd; t.zpart = dd;
      CK(wgrad6_launch(t, st));
    } else {
      for (int i = 0; i < p; ++i)
        CKI(wgrad_impl(ctrl_t, d, ws + W.dcI + (size_t)i * Bd, d, B, d, d, GP->qInputU_W + (size_t)i * dd, ws + W.small_slab, st));
    }
    CK(rs.add(ws + W.dcI, B, d, d, GP->qInputU_b, st, p, Bd, d));
  } else {
    hipLaunchKernelGGL(sum_parts_kernel, dim3(256), dim3(256), 0, st, (const float*)(ws + W.dcI), p, Bd, dcI_sum);
    CK(hipGetLastError());
    LinP ls = lin_basic(dcI_sum, d, d, B, wT + W.wqUT, nullptr, d, MACX_ACT_NON, ws + W.dt, d);
    CK(small_linear_launch(ls, 1, st));
    CKI(wgrad_impl(ctrl_t, d, dcI_sum, d, B, d, d, GP->qInputU_W, ws + W.small_slab, st));
    CK(rs.add(dcI_sum, B, d, d, GP->qInputU_b, st));
  }
  if (!o->control_input_unshared) {
    hipLaunchKernelGGL(mul_actgrad_kernel, dim3(64), dim3(256), 0, st, (const float*)(ws + W.dt), ctrl_t, o->control_input_act, Bd,
                       ws + W.du);
    CK(hipGetLastError());
  }
  // dvecQ = du Wq^T (+ dL/d(initial state) where a state is initialised from the question vector, mac_cell.py:496-505: the first of
  // them rides this launch's epilogue instead of an axpy of its own)
  const float* q_add = o->init_ctrl == MACX_INIT_Q ? DC : (o->init_mem == MACX_INIT_Q ? DM : nullptr);
  {
    LinP l = lin_basic(ws + W.du, d, d, B, wT + W.wqT, nullptr, d, MACX_ACT_NON, GI->vecQuestions, d);
    if (q_add) { l.addend = q_add; l.ld_add = d; }
    CK(small_linear_launch(l, 1, st));
  }
  CKI(wb.add(in->vecQuestions, d, ws + W.du, d, B, d, d, GP->qInput_W, st));
  CK(rs.add(ws + W.du, B, d, d, GP->qInput_b, st));

  // ---- initial state (mac_cell.py:496-505)
  if (o->init_mem == MACX_INIT_PRM) CK(rs.add(DM, B, d, d, GP->initMem, st));
  else if (o->init_mem == MACX_INIT_Q && q_add != DM) CK(axpy(DM, Bd, GI->vecQuestions, st));
  if (o->init_ctrl == MACX_INIT_PRM) CK(rs.add(DC, B, d, d, GP->initCtrl, st));

  }   // U_ALL
  // ---- weight gradients of the [B,d] linears, one contraction over all p*B rows each
  if (units & U_READ) {
    CKI(wb.add(saved + L.md, d, ws + W.DY, d, p * B, d, d, GP->projY_W, st));
    CK(rs.add(ws + W.DY, p * B, d, d, GP->projY_b, st));
  }
  if (units & U_WRITE) {
    CKI(wb.add(memories, d, dwlin_all, d, p * B, d, d, GP->newMemory_W, st));
    CKI(wb.add(infos, d, dwlin_all, d, p * B, d, d, GP->newMemory_W + dd, st));
    if (o->write_self_att)
      CKI(wb.add(saved + L.self_smry, d, dwlin_all, d, p * B, d, d, GP->newMemory_W + 2 * dd, st));
    CK(rs.add(dwlin_all, p * B, d, d, GP->newMemory_b, st));
  }
  if (units == U_READ) {
    // the read unit alone: dL/d(memory) = (dy Wy^T) through the two masks, dL/d(control) from the attention logits
    CK(dev_copy(ug->d_memory, DM, Bd * sizeof(float), st));
    CK(dev_copy(ug->d_control, DC + Bd, Bd * sizeof(float), st));
  }

  CKI(wb.run(st));
  }   // phase != 2
  CK(rs.run(st));
  if (phase == 1 || !(units & U_READ)) return MACX_OK;

  int ns_w = (int)W.ns_big;          // reduction splits of the dW2 / dWx contractions (H2 family: decided below)
  // ---- read-unit weights: fixed-order reduction of the per-step slabs
  // dW2 = sum_i H1_i^T dI2_i and dWx = sum_i dropout_i(KB)^T dX_i: ONE contraction each over all
  // p*B*N rows (the per-step operands are kept; 288 GB of HBM makes that the cheap choice)
  if (h2_mode()) {
    const int CB = d / 128;
    int* ecom = reinterpret_cast<int*>(ws + W.ecom);       // [H1 | dI2 | KBd | dX][EMIN_NB][8]
    constexpr int ES = EMIN_NB * 8;
    {
      EminList el;
      memset(&el, 0, sizeof(el));
      el.R = B * N; el.C = d; el.part = ecom;
      el.base[0] = reinterpret_cast<const char*>(saved + L.H1); el.stride[0] = L.act_stride * sizeof(float); el.nt[0] = p;
      el.base[1] = reinterpret_cast<const char*>(ws + W.dI2); el.stride[1] = W.act_floats * sizeof(float); el.nt[1] = p;
      el.base[2] = reinterpret_cast<const char*>(saved + L.KBd); el.stride[2] = L.act_stride * sizeof(float); el.nt[2] = rdrop ? p : 1;
      el.base[3] = reinterpret_cast<const char*>(ws + W.dX); el.stride[3] = W.act_floats * sizeof(float); el.nt[3] = p;
      CK(emin_list(el, 4, st));
    }
    TnH2P t;
    memset(&t, 0, sizeof(t));
    // mode 3: the two contractions share ONE launch AND the chip -- half the reduction splits each, so that the 2 x 128 workgroups
    // are all resident at once (one per CU) instead of 2 x 256 in two rounds: a workgroup's prologue, its 256 KB slab and the slab
    // reduction's input are paid for once per CU instead of twice
    ns_w = (wgrad_pipe_mode() >= 3 && wgrad_h2_kw(d) == 2 && wgrad_h2_jw(d) == 2 && W.ns_big >= 2) ? (int)W.ns_big / 2 : (int)W.ns_big;
    t.M = p * B * N; t.Kd = d; t.Jd = d; t.nsplit = ns_w; t.rows_per_split = rows_per_split(t.M, t.nsplit);
    t.R = B * N;
    t.A = reinterpret_cast<const char*>(saved + L.H1); t.a_stride = L.act_stride * sizeof(float); t.a_mod = 0;
    t.G = reinterpret_cast<const char*>(ws + W.dI2); t.g_stride = W.act_floats * sizeof(float);
    t.ecomA = ecom; t.ecomG = ecom + ES; t.ecom_nb = EMIN_NB;
    t.ftab = reinterpret_cast<uint16_t*>(ws + W.wg_ftab);
    t.dbg = kb_gemm_dbg();
    t.part = ws + W.slab_w2;
    const TnH2P t_w2 = t;
    t.A = reinterpret_cast<const char*>(saved + L.KBd);
    t.a_mod = rdrop ? 0 : B * N;                              // no dropout: the same (converted) KB every step
    t.G = reinterpret_cast<const char*>(ws + W.dX);
    t.ecomA = ecom + 2 * ES; t.ecomG = ecom + 3 * ES;
    t.part = ws + W.slab_wx;
    t.ftab = reinterpret_cast<uint16_t*>(ws + W.wg_ftab2);    // (a table of its own: the pair form builds both before either is read)
    CK(wgrad_h2_launch_pair(t_w2, t, st));
  } else {
    TnP t;
    memset(&t, 0, sizeof(t));
    t.M = p * B * N; t.Kd = d; t.Jd = d; t.nsplit = (int)W.ns_big; t.rows_per_split = rows_per_split(t.M, t.nsplit);
    t.A = saved + L.H1; t.lda = d; t.a_mod = t.M; t.G = ws + W.dI2; t.ldg = d;
    t.part = ws + W.slab_w2;
    CK(wgrad_any(t, st));
    if (rdrop) { t.A = saved + L.KBd; t.a_mod = t.M; }          // the dropped KB of every step was kept
    else { t.A = in->knowledgeBase; t.a_mod = B * N; }          // no dropout: the same KB each step
    t.G = ws + W.dX;
    t.part = ws + W.slab_wx;
    CK(wgrad_any(t, st));
  }

  if (h2_mode() && W.sb_deferred) {
    // dW1a = sum_i sum_b diag(y_ib) S_ib, dW1b = sum_i sum_b S_ib, S_ib = X_ib^T dI1_ib: every step in one launch, the two
    // accumulators of a workgroup run through all of them (macx_wgrad_h2.hip.h)
    const int CB = d / 128;
    SbH2P q;
    memset(&q, 0, sizeof(q));
    q.B = B; q.N = N; q.d = d; q.qpg = sb_qpg(B, N);
    q.X = h2_view(saved + L.X, B * N, d); q.dI1 = h2_view(ws + W.dI1, B * N, d);
    q.y = saved + L.y; q.W1a = P->memKbProj_W;
    q.dW1a_part = ws + W.slab_w1a; q.dW1b_part = ws + W.slab_w1b; q.dy_part = nullptr;
    q.nsteps = p;
    q.x_step = L.act_stride * sizeof(float); q.g_step = W.dI1_stride * sizeof(float); q.y_step = Bd;
    q.dbg = kb_gemm_dbg();
    q.qpg = W.sb_qpg;
    CK(W.sb_wide ? sb_h2w_launch(q, st) : sb_h2_launch(q, st));
  }
  const int nslab1 = (int)((h2_mode() && W.sb_deferred ? 1 : p) * W.ngroup);
  {
    SlabList sl;
    sl.d[0] = SlabDesc{ws + W.slab_w2, ns_w, dd / 4, GP->memKbProj2_W, 0};
    sl.d[1] = SlabDesc{ws + W.slab_wx, ns_w, dd / 4, GP->projX_W, 0};
    sl.d[2] = SlabDesc{ws + W.slab_w1a, nslab1, dd / 4, GP->memKbProj_W, 0};
    sl.d[3] = SlabDesc{ws + W.slab_w1b, nslab1, dd / 4, GP->memKbProj_W + dd, 0};
    CK(slab_reduce_list_launch(sl, 4, dd, st));
  }
  CK(rs.add(ws + W.db2_part, p * (int)W.dwk_rows, d, d, GP->memKbProj2_b, st));
  CK(rs.add(ws + W.db1_part, p * (int)W.db_rows, d, d, GP->memKbProj_b, st));
  CK(rs.add(ws + W.dbx_part, p * (int)W.db_rows, d, d, GP->projX_b, st));
  CK(rs.add(ws + W.dwk_part, p * (int)W.dwk_rows, d, d, GP->kbLogits_w, st));
  CK(rs.add(ws + W.dbk_part, p * B, 1, 1, GP->kbLogits_b, st));
  CK(rs.run(st));
  return MACX_OK;
}
}  // namespace

int macx_cell_backward_phase(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                             const macx_inputs* in, const float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                             const float* d_memory, const float* d_control, const macx_param_grads* GP,
                             const macx_input_grads* GI, int phase, void* stream) {
  ModeScope ms(o);
  return cell_backward_impl(o, s, dp, P, in, saved, saved_floats, ws, ws_floats, d_memory, d_control, GP, GI, phase, U_ALL, nullptr, stream);
}

int macx_cell_backward(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                       const macx_inputs* in, const float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                       const float* d_memory, const float* d_control, const macx_param_grads* GP,
                       const macx_input_grads* GI, void* stream) {
  return macx_cell_backward_phase(o, s, dp, P, in, saved, saved_floats, ws, ws_floats, d_memory, d_control, GP, GI, 0, stream);
}

// =================================================================================================
// unit-level entry points
// =================================================================================================
// ---- one read / write unit on caller-owned buffers: the cell's own step code restricted to a unit.
// shapes->p == 1: the unit of step 0 (the dropout streams are keyed by (seed, site, step 0)).
namespace {
int unit_check(const macx_opts* o, const macx_shapes* s, int units) {
  CKI(check_impl(o, s));
  if (s->p != 1) return MACX_EINVAL;
  if ((units & U_WRITE) && o->write_self_att) return MACX_EUNSUPPORTED;   // needs the histories of a running cell
  return MACX_OK;
}
}  // namespace

size_t macx_workspace_bytes(const macx_opts* o, const macx_shapes* s, int for_backward) {
  return macx_ws_floats(o, s, for_backward) * sizeof(float);
}

int macx_read_fwd(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                  const float* knowledgeBase, const float* memory, const float* control, float* saved, size_t saved_floats,
                  float* info, float* att, void* stream) {
  ModeScope ms(o);
  CKI(unit_check(o, s, U_READ));
  if (!dp || !P || !knowledgeBase || !memory || !control || !saved || !info || !att) return MACX_EINVAL;
  if (misaligned(saved) || misaligned(knowledgeBase)) return MACX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const SavedLayout L = make_saved(o, s, 1);
  if (saved_floats < L.total) return MACX_ESMALL;
  const size_t Bd = (size_t)s->B * s->d;
  CKI(pack_forward_weights(o, s, P, saved, L, 1, U_READ, st));
  CK(dev_copy(saved + L.seg[MACX_SEG_MEMORIES], memory, Bd * sizeof(float), st));
  CK(dev_copy(saved + L.seg[MACX_SEG_CONTROLS] + Bd, control, Bd * sizeof(float), st));
  macx_inputs in;
  memset(&in, 0, sizeof(in));
  in.knowledgeBase = knowledgeBase;
  CKI(cell_step_impl(o, s, dp, P, &in, saved, saved_floats, 1, 0, U_READ, stream));
  const float* info_raw = dp->keep_write < 1.0f ? saved + L.info_raw : saved + L.seg[MACX_SEG_INFOS];
  CK(dev_copy(info, info_raw, Bd * sizeof(float), st));
  CK(dev_copy(att, saved + L.seg[MACX_SEG_ATT_KB], (size_t)s->B * s->N * sizeof(float), st));
  return MACX_OK;
}

int macx_read_bwd(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                  const float* knowledgeBase, const float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                  const float* d_info, const macx_param_grads* GP, float* d_knowledgeBase, float* d_memory, float* d_control,
                  void* stream) {
  ModeScope ms(o);
  CKI(unit_check(o, s, U_READ));
  if (!knowledgeBase || !d_info || !d_knowledgeBase || !d_memory || !d_control) return MACX_EINVAL;
  macx_inputs in;
  memset(&in, 0, sizeof(in));
  in.knowledgeBase = knowledgeBase;
  macx_input_grads GI;
  memset(&GI, 0, sizeof(GI));
  GI.knowledgeBase = d_knowledgeBase;
  UnitGrads ug;
  ug.d_info_in = d_info; ug.d_memory = d_memory; ug.d_control = d_control;
  return cell_backward_impl(o, s, dp, P, &in, saved, saved_floats, ws, ws_floats, nullptr, nullptr, GP, &GI, 0, U_READ, &ug, stream);
}

int macx_write_fwd(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                   const float* memory, const float* info, const float* control, float* saved, size_t saved_floats,
                   float* new_memory, void* stream) {
  ModeScope ms(o);
  CKI(unit_check(o, s, U_WRITE));
  if (!dp || !P || !memory || !info || !control || !saved || !new_memory) return MACX_EINVAL;
  if (misaligned(saved)) return MACX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const SavedLayout L = make_saved(o, s, 1);
  if (saved_floats < L.total) return MACX_ESMALL;
  const size_t Bd = (size_t)s->B * s->d;
  CKI(pack_forward_weights(o, s, P, saved, L, 1, U_WRITE, st));
  float* info_raw = dp->keep_write < 1.0f ? saved + L.info_raw : saved + L.seg[MACX_SEG_INFOS];
  CK(dev_copy(saved + L.seg[MACX_SEG_MEMORIES], memory, Bd * sizeof(float), st));
  CK(dev_copy(info_raw, info, Bd * sizeof(float), st));
  CK(dev_copy(saved + L.seg[MACX_SEG_CONTROLS] + Bd, control, Bd * sizeof(float), st));
  macx_inputs in;
  memset(&in, 0, sizeof(in));
  CKI(cell_step_impl(o, s, dp, P, &in, saved, saved_floats, 1, 0, U_WRITE, stream));
  CK(dev_copy(new_memory, saved + L.seg[MACX_SEG_MEMORIES] + Bd, Bd * sizeof(float), st));
  return MACX_OK;
}

int macx_write_bwd(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                   const float* saved, size_t saved_floats, float* ws, size_t ws_floats, const float* d_new_memory,
                   const macx_param_grads* GP, float* d_memory, float* d_info, float* d_control, void* stream) {
  ModeScope ms(o);
  CKI(unit_check(o, s, U_WRITE));
  if (!d_new_memory || !d_memory || !d_info || !d_control) return MACX_EINVAL;
  macx_inputs in;
  memset(&in, 0, sizeof(in));
  macx_input_grads GI;
  memset(&GI, 0, sizeof(GI));
  UnitGrads ug;
  ug.d_memory = d_memory; ug.d_info_out = d_info; ug.d_control = d_control;
  return cell_backward_impl(o, s, dp, P, &in, saved, saved_floats, ws, ws_floats, d_new_memory, nullptr, GP, &GI, 0, U_WRITE, &ug, stream);
}

int macx_linear(const float* x1, int k1, const float* x2, int k2, int rows, const float* Wp, const float* b, float bias_const,
                int n_out, int act, float* out, void* stream) {
  if (!x1 || !Wp || !out || rows < 1 || n_out < 16 || n_out % 16) return MACX_EINVAL;
  if (k1 % 16 != 0 || k2 % 16 != 0 || k1 < 16 || (x2 == nullptr) != (k2 == 0)) return MACX_EINVAL;
  if (misaligned(x1) || misaligned(x2) || misaligned(Wp) || misaligned(out)) return MACX_EINVAL;
  LinP l = lin_basic(x1, k1, k1, rows, Wp, b, n_out, act, out, n_out);
  if (x2) { l.seg[1] = LinSeg{x2, k2, k2, 0}; l.Ktot = k1 + k2; }
  l.bias_const = bias_const;
  CK(small_linear_launch(l, 1, (hipStream_t)stream));
  return MACX_OK;
}

int macx_pack_weight(const float* Wt, int K, int n_out, int flags, float* out, void* stream) {
  const int transpose = flags;
  if (!Wt || !out || K < 16 || K % 16 || n_out < 16 || n_out % 16) return MACX_EINVAL;
  const int fmt = transpose >> 1;      // bits 1..: pack format (0 fp32 MFMA layout, 1 split-bf16 planes, 2 fp32 k-major tiles)
  if (fmt < 0 || fmt > 2 || (fmt && K % 32)) return MACX_EINVAL;
  Packer pk;
  if (transpose & 1) pk.add(Wt, 1, K, K, n_out, out, -1, -1, fmt);
  else pk.add(Wt, n_out, 1, K, n_out, out, -1, -1, fmt);
  CK(pk.run((hipStream_t)stream));
  return MACX_OK;
}

int macx_kb_project(const macx_shapes* s, const macx_dropout* dp, int step, const float* kb, const float* Wp, const float* b,
                    float* out, float* bits_ws, void* stream) {
  if (!s || !dp || !kb || !Wp || !b || !out) return MACX_EINVAL;
  if (s->d % 128 != 0 || s->N > K_MAXN || ((size_t)s->B * s->N * s->d) % 32) return MACX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int d = s->d;
  GemmP g;
  memset(&g, 0, sizeof(g));
  g.B = s->B; g.N = s->N; g.K = d; g.Nout = d;
  g.A = kb; g.lda = d;
  g.Wp = Wp; g.out = out; g.ldo = d; g.bias = b; g.act = MACX_ACT_NON;
  if (dp->keep_read < 1.0f) {
    if (!bits_ws) return MACX_EINVAL;
    // scratch: [B*N*d] dropped KB, then [B*N*d/32] keep bits
    const size_t n = (size_t)s->B * s->N * d;
    const DropSpec dk = make_drop(dp->keep_read, dp, SITE_READ_KB, step);
    hipLaunchKernelGGL(kb_dropout_kernel, dim3(2048), dim3(256), 0, st, kb, n / 4, dk.key, dk.thr24, dk.inv_keep,
                       (uint32_t)((size_t)s->b0 * s->N * d), bits_ws, reinterpret_cast<uint32_t*>(bits_ws + n), 0u, (uint32_t*)nullptr,
                       dk.word);
    CK(hipGetLastError());
    g.A = bits_ws;
  }
  CK((kb_gemm<A_PLAIN, B_PLAIN, E_BIAS_ACT, false>(g, st)));     // Wp: pack format 1 (split mode) or 0
  return MACX_OK;
}

int macx_control_attend(const macx_shapes* s, const float* cc, const float* words, const int32_t* lengths, const float* w,
                        const float* b, float* att, float* control, void* stream) {
  if (!s || !cc || !words || !lengths || !w || !b || !att || !control) return MACX_EINVAL;
  if (s->S > C_MAXS || s->d % 4 != 0) return MACX_EINVAL;
  CtrlP c;
  c.B = s->B; c.S = s->S; c.d = s->d;
  c.cc = cc; c.z_cc = 0; c.words = words; c.lengths = lengths; c.w = w; c.bias = b;
  c.att = att; c.z_att = 0; c.control = control; c.z_ctl = 0;
  hipLaunchKernelGGL(control_attend_kernel, dim3(s->B, 1), dim3(256), 0, (hipStream_t)stream, c);
  CK(hipGetLastError());
  return MACX_OK;
}

size_t macx_control_attend_bwd_ws_floats(const macx_shapes* s) {
  if (!s) return 0;
  return al4((size_t)s->B * s->S) + al4((size_t)s->B * s->d) + al4((size_t)s->B);
}

int macx_control_attend_bwd(const macx_shapes* s, const float* d_control, const float* cc, const float* att, const float* words,
                            const float* w, float* ws, size_t ws_floats, float* d_cc, float* d_words, float* d_w, float* d_b,
                            void* stream) {
  if (!s || !d_control || !cc || !att || !words || !w || !ws || !d_cc || !d_words || !d_w || !d_b) return MACX_EINVAL;
  if (s->S > C_MAXS || s->d % 64 != 0) return MACX_EINVAL;
  if (ws_floats < macx_control_attend_bwd_ws_floats(s)) return MACX_ESMALL;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, S = s->S, d = s->d;
  CtrlBwdP c;
  c.B = B; c.S = S; c.d = d; c.nz = 1;
  c.dcontrol = d_control; c.z_dc = 0; c.cc = cc; c.z_cc = 0; c.att = att; c.z_att = 0;
  c.words = words; c.w = w;
  c.dl = ws;
  c.dw_part = ws + al4((size_t)B * S);
  c.db_part = c.dw_part + al4((size_t)B * d);
  c.dcc = d_cc; c.z_dcc = 0; c.dwords = d_words; c.acc_words = 0;
  hipLaunchKernelGGL(control_bwd_dl_kernel, dim3(B, 1), dim3(256), 0, st, c);
  hipLaunchKernelGGL(control_bwd_apply_kernel, dim3(B, d / 64), dim3(256), 0, st, c);
  CK(hipGetLastError());
  CK(rowsum(c.dw_part, B, d, d, d_w, st));
  CK(rowsum(c.db_part, B, 1, 1, d_b, st));
  return MACX_OK;
}

int macx_embed_lookup(const int32_t* ids, const float* emb, int rows, int E, int ld, float keep, uint32_t seed, uint32_t first_row,
                      float* x, void* stream) {
  if (!ids || !emb || !x || rows < 1 || E < 1 || ld < E) return MACX_EINVAL;
  hipLaunchKernelGGL(embed_gather_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, ids, emb, rows, E, ld, first_row,
                     make_drop(keep, seed, SITE_ENC_INPUT, 0), x);
  CK(hipGetLastError());
  return MACX_OK;
}

int macx_embed_lookup_bwd(const int32_t* ids, const float* dx, int rows, int E, int ld, int V, float keep, uint32_t seed,
                          uint32_t first_row, float* d_emb, void* stream) {
  if (!ids || !dx || !d_emb || rows < 1 || E < 1 || ld < E || V < 1) return MACX_EINVAL;
  hipLaunchKernelGGL(embed_grad_kernel, dim3(V, (E + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, ids, dx, rows, E, ld, first_row,
                     make_drop(keep, seed, SITE_ENC_INPUT, 0), d_emb);
  CK(hipGetLastError());
  return MACX_OK;
}

int macx_dropout_mask_w(uint32_t seed, uint32_t site, uint32_t step, float keep, uint32_t first, size_t n, const uint32_t* mask_word,
                        float* out, void* stream) {
  if (!out) return MACX_EINVAL;
  const DropSpec ds = make_drop(keep, seed, site, step);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, ds.key, ds.thr24, first, n, out, mask_word);
  CK(hipGetLastError());
  return MACX_OK;
}
int macx_dropout_mask(uint32_t seed, uint32_t site, uint32_t step, float keep, uint32_t first, size_t n, float* out, void* stream) {
  return macx_dropout_mask_w(seed, site, step, keep, first, n, nullptr, out, stream);
}

// ---- answer loss + prediction (model.py:593-612): SURVEY 8f row 2 -------------------------------------------------
// loss_rows[b] = sparse softmax cross-entropy of question b (the caller's loss is their mean: tf.reduce_mean, model.py:596);
// pred[b] = argmax; dlogits (may be NULL) = (softmax - onehot) * grad_scale, i.e. pass 1/B for the gradient of the mean loss
int macx_answer_loss(const float* logits, const int32_t* answers, int B, int A, float* loss_rows, int32_t* pred, float* dlogits,
                     float grad_scale, void* stream) {
  if (!logits || !answers || !loss_rows || !pred || B < 1 || A < 1) return MACX_EINVAL;
  hipLaunchKernelGGL(answer_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, answers, B, A, loss_rows, pred,
                     dlogits, grad_scale);
  CK(hipGetLastError());
  return MACX_OK;
}

// ---- the knowledge-base attention unit on its own (the fused cell's kernels behind a per-unit contract) -----------------
// forward: att = softmax_n(logits + bias), info = sum_n att KB       (ops.inter2att :140-146 + ops.att2Smry :149-150)
int macx_kb_attend_fwd(int B, int N, int d, const float* logits, const float* bias, const float* kb, float* att, float* info,
                       void* stream) {
  if (!logits || !bias || !kb || !att || !info || B < 1 || N < 1 || N > K_MAXN || d < 128 || d % 128) return MACX_EINVAL;
  KbAttP a;
  a.B = B; a.N = N; a.d = d; a.nparts = 1;
  a.logit_part = logits; a.bias = bias; a.kb = kb; a.att = att; a.info = info;
  hipLaunchKernelGGL(kb_attend_kernel, dim3(B, d / 128), dim3(KA_THREADS), 0, (hipStream_t)stream, a);
  CK(hipGetLastError());
  return MACX_OK;
}
size_t macx_kb_attend_bwd_ws_floats(int B, int N, int d) { return (size_t)B * N + 8; (void)d; }
// backward: da = dinfo . KB ; dlogits = att (da - sum_n att da) ; dKB (=, or += with accumulate) att (x) dinfo
int macx_kb_attend_bwd(int B, int N, int d, const float* att, const float* kb, const float* dinfo, float* dlogits, float* dkb,
                       int accumulate, float* ws, size_t ws_floats, void* stream) {
  if (!att || !kb || !dinfo || !dlogits || !ws || B < 1 || N < 1 || d < 128 || d % 128) return MACX_EINVAL;
  if (ws_floats < macx_kb_attend_bwd_ws_floats(B, N, d)) return MACX_ESMALL;
  hipStream_t st = (hipStream_t)stream;
  float* da = ws;
  hipLaunchKernelGGL(kb_att_da_kernel, dim3((B * N + 3) / 4), dim3(256), 0, st, dinfo, d, kb, B, N, d, da);
  hipLaunchKernelGGL(op_softmax_bwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, att, (const float*)da, (size_t)B, N, dlogits);
  if (dkb) {
    const size_t n = (size_t)B * N * d;
    hipLaunchKernelGGL(kb_attend_dkb_kernel, dim3(op_grid(n)), dim3(256), 0, st, att, dinfo, n, N, d, accumulate, dkb);
  }
  CK(hipGetLastError());
  return MACX_OK;
}

// ---- the ops.py primitives as single kernels (macx_ops.hip.h): the generic option path ---------------------------
int macx_op_act(int act, const float* x, const float* alpha, size_t n, int inner, float* out, void* stream) {
  const bool ext = act == OP_ACT_PRELU || act == OP_ACT_RSQRT_EPS;
  if (!x || !out || inner < 1 || (ext && !alpha) || (!ext && (act < 0 || act > ACT_RELU))) return MACX_EINVAL;
  if (n == 0) return MACX_OK;
  hipLaunchKernelGGL(op_act_kernel, dim3(op_grid(n)), dim3(256), 0, (hipStream_t)stream, act, x, alpha, n, inner, out);
  CK(hipGetLastError());
  return MACX_OK;
}
int macx_op_act_bwd(int act, const float* x, const float* alpha, const float* dy, size_t n, int inner, float* dx, float* dalpha_elem,
                    void* stream) {
  const bool ext = act == OP_ACT_PRELU || act == OP_ACT_RSQRT_EPS;
  if (!x || !dy || !dx || inner < 1 || (ext && !alpha) || (act == OP_ACT_PRELU && !dalpha_elem) || (!ext && (act < 0 || act > ACT_RELU)))
    return MACX_EINVAL;
  if (n == 0) return MACX_OK;
  hipLaunchKernelGGL(op_act_bwd_kernel, dim3(op_grid(n)), dim3(256), 0, (hipStream_t)stream, act, x, alpha, dy, n, inner, dx, dalpha_elem);
  CK(hipGetLastError());
  return MACX_OK;
}
int macx_op_binary(int op, int bmode, const float* a, const float* b, size_t n, int mid, int inner, float scale, float* out, void* stream) {
  if (!a || !b || !out || op < OP_ADD || op > OP_MUL || bmode < OP_B_SAME || bmode > OP_B_ROW || mid < 1 || inner < 1) return MACX_EINVAL;
  if (n % ((size_t)inner * (bmode == OP_B_MID ? mid : 1))) return MACX_EINVAL;
  if (n == 0) return MACX_OK;
  hipLaunchKernelGGL(op_binary_kernel, dim3(op_grid(n)), dim3(256), 0, (hipStream_t)stream, op, bmode, a, b, n, mid, inner, scale, out);
  CK(hipGetLastError());
  return MACX_OK;
}
int macx_op_reduce(int mode, const float* x, size_t outer, int mid, int inner, float* out, float* ws, void* stream) {
  if (!x || !out || outer < 1 || mid < 1 || inner < 1) return MACX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (mode == OP_R_MID) {
    if (outer > 0x7FFFFFFF) return MACX_EINVAL;
    hipLaunchKernelGGL(op_reduce_mid_kernel, dim3((unsigned)((outer * inner + 255) / 256)), dim3(256), 0, st, x, (int)outer, mid, inner, out);
  } else if (mode == OP_R_LAST) {
    hipLaunchKernelGGL(op_reduce_last_kernel, dim3((unsigned)((outer + 3) / 4)), dim3(256), 0, st, x, outer, inner, out);
  } else if (mode == OP_R_ROWS) {
    if (!ws) return MACX_EINVAL;
    hipLaunchKernelGGL(op_reduce_rows_kernel, dim3((inner + 255) / 256, OP_ROWS_SPLIT), dim3(256), 0, st, x, outer, inner, ws);
    hipLaunchKernelGGL(op_reduce_rows_final_kernel, dim3((inner + 255) / 256), dim3(256), 0, st, ws, inner, out);
  } else {
    return MACX_EINVAL;
  }
  CK(hipGetLastError());
  return MACX_OK;
}
int macx_op_softmax(const float* x, const int32_t* lengths, int rows_per_len, size_t rows, int n, float* out, void* stream) {
  if (!x || !out || rows < 1 || n < 1 || (lengths && rows_per_len < 1)) return MACX_EINVAL;
  hipLaunchKernelGGL(op_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, lengths, rows_per_len, rows, n,
                     out);
  CK(hipGetLastError());
  return MACX_OK;
}
int macx_op_softmax_bwd(const float* a, const float* da, size_t rows, int n, float* dx, void* stream) {
  if (!a || !da || !dx || rows < 1 || n < 1) return MACX_EINVAL;
  hipLaunchKernelGGL(op_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, da, rows, n, dx);
  CK(hipGetLastError());
  return MACX_OK;
}
int macx_op_dropout_w(const float* x, size_t n, uint32_t seed, uint32_t site, uint32_t step, float keep, uint32_t first,
                      const uint32_t* mask_word, float* out, void* stream) {
  if (!x || !out || !(keep > 0.f) || keep > 1.f) return MACX_EINVAL;
  if (n == 0) return MACX_OK;
  const DropSpec ds = make_drop(keep, seed, site, step);
  hipLaunchKernelGGL(op_dropout_kernel, dim3(op_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, first, ds.key, ds.thr24, ds.inv_keep, out,
                     mask_word);
  CK(hipGetLastError());
  return MACX_OK;
}
int macx_op_dropout(const float* x, size_t n, uint32_t seed, uint32_t site, uint32_t step, float keep, uint32_t first, float* out,
                    void* stream) {
  return macx_op_dropout_w(x, n, seed, site, step, keep, first, nullptr, out, stream);
}

// =================================================================================================
// output unit + classifier (model.py:512-576, ops.py:349-359): SURVEY 8f row 2
// =================================================================================================
namespace {
struct OutLayout {
  size_t woq_p, w0_p, w1_p;   // packed forward weights
  size_t eq;                  // [B,d]   outQuestion(vecQ)
  size_t x0;                  // [B,in]  dropped concat([memory, eq])
  size_t h;                   // [B,H]   act(fc_0)
  size_t x1;                  // [B,H]   dropped h
  size_t logits_pad;          // [B,Ap]
  size_t total;
};
inline int pad16(int n) { return (n + 15) & ~15; }
OutLayout make_out(const macx_out_shapes* s) {
  OutLayout L;
  memset(&L, 0, sizeof(L));
  size_t off = 0;
  auto take = [&](size_t n) { size_t r = off; off += al4(n); return r; };
  const size_t B = s->B, d = s->d, in = 2 * (size_t)s->d, H = s->hidden, Ap = pad16(s->answers);
  L.woq_p = take(d * d); L.w0_p = take(in * H); L.w1_p = take(H * Ap);
  L.eq = take(B * d); L.x0 = take(B * in); L.h = take(B * H); L.x1 = take(B * H); L.logits_pad = take(B * Ap);
  L.total = off;
  return L;
}
struct OutBwdLayout {
  size_t woqT, w0T, w1T;
  size_t dlog_pad, dx1, dh, dx0, deq, total;
};
OutBwdLayout make_out_bwd(const macx_out_shapes* s) {
  OutBwdLayout L;
  memset(&L, 0, sizeof(L));
  size_t off = 0;
  auto take = [&](size_t n) { size_t r = off; off += al4(n); return r; };
  const size_t B = s->B, d = s->d, in = 2 * (size_t)s->d, H = s->hidden, Ap = pad16(s->answers);
  L.woqT = take(d * d); L.w0T = take(H * in); L.w1T = take(Ap * H);
  L.dlog_pad = take(B * Ap); L.dx1 = take(B * H); L.dh = take(B * H); L.dx0 = take(B * in); L.deq = take(B * d);
  L.total = off;
  return L;
}
int out_check(const macx_out_shapes* s) {
  if (!s || s->B < 1 || s->d < 16 || s->d % 16 || s->hidden < 16 || s->hidden % 16 || s->answers < 1) return MACX_EINVAL;
  return MACX_OK;
}
}  // namespace

size_t macx_output_saved_floats(const macx_out_shapes* s) { return out_check(s) ? 0 : make_out(s).total; }
size_t macx_output_ws_floats(const macx_out_shapes* s) { return out_check(s) ? 0 : make_out_bwd(s).total; }

int macx_output_forward(const macx_out_shapes* s, int act, float keep, uint32_t seed, const macx_out_params* P, const float* memory,
                        const float* vecQ, float* logits, float* saved, size_t saved_floats, void* stream) {
  CKI(out_check(s));
  if (!P || !memory || !vecQ || !logits || !saved) return MACX_EINVAL;
  const OutLayout L = make_out(s);
  if (saved_floats < L.total) return MACX_ESMALL;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, d = s->d, in = 2 * d, H = s->hidden, A = s->answers, Ap = pad16(A);
  Packer pk;
  pk.add(P->outQuestion_W, d, 1, d, d, saved + L.woq_p);
  pk.add(P->fc0_W, H, 1, in, H, saved + L.w0_p);
  pk.add(P->fc1_W, A, 1, H, Ap, saved + L.w1_p, H, A);
  CK(pk.run(st));
  // outputOp (model.py:512-528): features = concat([memory, outQuestion(vecQ)])
  LinP q = lin_basic(vecQ, d, d, B, saved + L.woq_p, P->outQuestion_b, d, MACX_ACT_NON, saved + L.eq, d);
  CK(small_linear_launch(q, 1, st));
  // classifier (model.py:547-576 -> ops.FCLayer ops.py:349-359): dropout on every layer input, act between layers
  const DropSpec d0 = make_drop(keep, seed, SITE_OUT_FC0, 0), d1 = make_drop(keep, seed, SITE_OUT_FC1, 0);
  // the concat is built in place: columns [0,d) memory, [d,2d) eq, with the layer-input mask indexed over [B, 2d]
  CK(dev_copy2d(saved + L.x0, (size_t)in * sizeof(float), memory, (size_t)d * sizeof(float), (size_t)d * sizeof(float), B, st));
  CK(dev_copy2d(saved + L.x0 + d, (size_t)in * sizeof(float), saved + L.eq, (size_t)d * sizeof(float), (size_t)d * sizeof(float), B, st));
  hipLaunchKernelGGL(drop2_kernel, dim3(64), dim3(256), 0, st, (const float*)(saved + L.x0), B, in, (uint32_t)s->b0, d0, no_drop(),
                     saved + L.x0);
  LinP f0 = lin_basic(saved + L.x0, in, in, B, saved + L.w0_p, P->fc0_b, H, act, saved + L.h, H);
  CK(small_linear_launch(f0, 1, st));
  hipLaunchKernelGGL(drop2_kernel, dim3(64), dim3(256), 0, st, (const float*)(saved + L.h), B, H, (uint32_t)s->b0, d1, no_drop(),
                     saved + L.x1);
  CK(hipGetLastError());
  // fc_1: bias padded with zeros by reading through a guarded copy
  CK(dev_zero(saved + L.logits_pad, (size_t)B * Ap * sizeof(float), st));
  LinP f1 = lin_basic(saved + L.x1, H, H, B, saved + L.w1_p, nullptr, Ap, MACX_ACT_NON, saved + L.logits_pad, Ap);
  CK(small_linear_launch(f1, 1, st));
  hipLaunchKernelGGL(crop_cols_kernel, dim3(16), dim3(256), 0, st, (const float*)(saved + L.logits_pad), Ap, B, A, logits);
  // + bias (the packed layer ran without it because the bias vector is not padded)
  hipLaunchKernelGGL(add_bias_kernel, dim3(16), dim3(256), 0, st, P->fc1_b, B, A, logits);
  CK(hipGetLastError());
  return MACX_OK;
}

int macx_output_backward(const macx_out_shapes* s, int act, float keep, uint32_t seed, const macx_out_params* P, const float* memory,
                         const float* vecQ, const float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                         const float* d_logits, const macx_out_grads* G, float* d_memory, float* d_vecQ, void* stream) {
  CKI(out_check(s));
  if (!P || !memory || !vecQ || !saved || !ws || !d_logits || !G || !d_memory || !d_vecQ) return MACX_EINVAL;
  const OutLayout L = make_out(s);
  const OutBwdLayout W = make_out_bwd(s);
  if (saved_floats < L.total || ws_floats < W.total) return MACX_ESMALL;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, d = s->d, in = 2 * d, H = s->hidden, A = s->answers, Ap = pad16(A);
  Packer pk;
  pk.add(P->outQuestion_W, 1, d, d, d, ws + W.woqT);
  pk.add(P->fc0_W, 1, H, H, in, ws + W.w0T);          // W0^T: [H] -> [2d]
  pk.add(P->fc1_W, 1, A, Ap, H, ws + W.w1T, A, H);    // W1^T: [Ap] -> [H], rows past A are zero
  CK(pk.run(st));
  const DropSpec d0 = make_drop(keep, seed, SITE_OUT_FC0, 0), d1 = make_drop(keep, seed, SITE_OUT_FC1, 0);
  hipLaunchKernelGGL(pad_cols_kernel, dim3(16), dim3(256), 0, st, d_logits, A, B, Ap, ws + W.dlog_pad);
  // fc_1
  hipLaunchKernelGGL(outer_sum_kernel, dim3(64), dim3(256), 0, st, saved + L.x1, H, d_logits, A, B, H, A, G->fc1_W);
  CK(rowsum(d_logits, B, A, A, G->fc1_b, st));
  LinP b1 = lin_basic(ws + W.dlog_pad, Ap, Ap, B, ws + W.w1T, nullptr, H, MACX_ACT_NON, ws + W.dx1, H);
  CK(small_linear_launch(b1, 1, st));
  // through dropout(h) and the activation: dh = dx1 * mask1 * act'(h)
  hipLaunchKernelGGL(drop2_kernel, dim3(64), dim3(256), 0, st, (const float*)(ws + W.dx1), B, H, (uint32_t)s->b0, d1, no_drop(), ws + W.dx1);
  hipLaunchKernelGGL(mul_actgrad_kernel, dim3(64), dim3(256), 0, st, (const float*)(ws + W.dx1), saved + L.h, act, (size_t)B * H, ws + W.dh);
  // fc_0
  hipLaunchKernelGGL(outer_sum_kernel, dim3(512), dim3(256), 0, st, saved + L.x0, in, (const float*)(ws + W.dh), H, B, in, H, G->fc0_W);
  CK(rowsum(ws + W.dh, B, H, H, G->fc0_b, st));
  LinP b0 = lin_basic(ws + W.dh, H, H, B, ws + W.w0T, nullptr, in, MACX_ACT_NON, ws + W.dx0, in);
  CK(small_linear_launch(b0, 1, st));
  hipLaunchKernelGGL(drop2_kernel, dim3(64), dim3(256), 0, st, (const float*)(ws + W.dx0), B, in, (uint32_t)s->b0, d0, no_drop(), ws + W.dx0);
  // split the concat gradient
  hipLaunchKernelGGL(copy_cols_drop_kernel, dim3(64), dim3(256), 0, st, (const float*)(ws + W.dx0), in, 0, B, d, 0u, no_drop(), d_memory);
  hipLaunchKernelGGL(copy_cols_drop_kernel, dim3(64), dim3(256), 0, st, (const float*)(ws + W.dx0), in, d, B, d, 0u, no_drop(), ws + W.deq);
  // outQuestion
  hipLaunchKernelGGL(outer_sum_kernel, dim3(256), dim3(256), 0, st, vecQ, d, (const float*)(ws + W.deq), d, B, d, d, G->outQuestion_W);
  CK(rowsum(ws + W.deq, B, d, d, G->outQuestion_b, st));
  LinP bq = lin_basic(ws + W.deq, d, d, B, ws + W.woqT, nullptr, d, MACX_ACT_NON, d_vecQ, d);
  CK(small_linear_launch(bq, 1, st));
  CK(hipGetLastError());
  return MACX_OK;
}

// =================================================================================================
// stem CNN (model.py:165-204, ops.CNNLayer ops.py:380-438): SURVEY 8f row 1.  Two 3x3 SAME
// convolutions as implicit GEMMs on the knowledge-base GEMM kernel (K = 9 * C_in, per-image tiles).
// =================================================================================================
namespace {
struct StemLayout {
  size_t k0_p, k1_p;     // packed HWIO kernels [9*Cin][Cmid], [9*Cmid][Cout]
  size_t in0p;           // [B][Np][Cin]   dropped, halo-padded image features
  size_t X1;             // [B][N][Cmid]   act(conv0)
  size_t in1p;           // [B][Np][Cmid]  dropped, halo-padded X1
  size_t bits1;          // [B*N*Cmid/32]  keep bits of the layer-1 input dropout
  size_t amax;           // 3h family: largest magnitudes [kernel0, kernel1, in0p, in1p] (+ 4 spare), then scratch for the absmax passes
  size_t total;
};
struct StemGeo { int N, wp, np; };
inline StemGeo stem_geo(const macx_stem_shapes* s) { return StemGeo{s->H * s->W, s->W + 2, (s->H + 2) * (s->W + 2)}; }
StemLayout make_stem(const macx_stem_shapes* s) {
  StemLayout L;
  memset(&L, 0, sizeof(L));
  const StemGeo g = stem_geo(s);
  size_t off = 0;
  auto take = [&](size_t n) { size_t r = off; off += al4(n); return r; };
  const size_t B = s->B, Ci = s->Cin, Cm = s->Cmid, Co = s->Cout;
  L.k0_p = take(wsize(9 * Ci, Cm)); L.k1_p = take(wsize(9 * Cm, Co));
  L.in0p = take(B * g.np * Ci); L.X1 = take(B * g.N * Cm); L.in1p = take(B * g.np * Cm);
  L.bits1 = take(B * g.N * Cm / 32 + 8);
  L.amax = take(8 + 4 * AMAX_BIG_BLOCKS);
  L.total = off;
  return L;
}
inline int conv_splits(int tiles, int M) {
  // fill whole rounds of 256 workgroups; fewest splits among the best fillings
  int best = 1; double beff = 0.0;
  for (int ns = 1; ns <= 16; ++ns) {
    if ((M + ns - 1) / ns < 256) break;
    const int blocks = tiles * ns;
    const double eff = (double)blocks / (double)(((blocks + 255) / 256) * 256);
    if (eff > beff + 0.02) { beff = eff; best = ns; }
  }
  return best;
}
struct StemBwdLayout {
  size_t k1T_p;          // packed backward-data weights [9*Cout][Cmid]
  size_t dY2, dY2p, dY1; // [B][N][Cout], [B][Np][Cout], [B][N][Cmid]
  size_t slab0, slab1;   // weight-gradient partial slabs
  int ns0, ns1;
  size_t amax;           // 3h family: largest magnitudes [dY2, dY1] (+ 6 spare), then scratch
  size_t total;
};
StemBwdLayout make_stem_bwd(const macx_stem_shapes* s) {
  StemBwdLayout L;
  memset(&L, 0, sizeof(L));
  const StemGeo g = stem_geo(s);
  size_t off = 0;
  auto take = [&](size_t n) { size_t r = off; off += al4(n); return r; };
  const size_t B = s->B, Ci = s->Cin, Cm = s->Cmid, Co = s->Cout;
  L.k1T_p = take(wsize(9 * Co, Cm));
  L.dY2 = take(B * g.N * Co); L.dY2p = take(B * g.np * Co); L.dY1 = take(B * g.N * Cm);
  L.ns0 = conv_splits((int)(9 * Ci / T_TILE * (Cm / T_TILE)), (int)(B * g.N));
  L.ns1 = conv_splits((int)(9 * Cm / T_TILE * (Co / T_TILE)), (int)(B * g.N));
  L.slab0 = take((size_t)L.ns0 * 9 * Ci * Cm);
  L.slab1 = take((size_t)L.ns1 * 9 * Cm * Co);
  L.amax = take(8 + 2 * AMAX_BIG_BLOCKS);
  L.total = off;
  return L;
}
int stem_check(const macx_stem_shapes* s) {
  if (!s || s->B < 1 || s->H < 1 || s->W < 2) return MACX_EINVAL;
  if (s->Cin % 128 || s->Cmid % 128 || s->Cout % 128 || s->Cin < 128 || s->Cmid < 128 || s->Cout < 128) return MACX_EINVAL;
  if (s->H * s->W > K_MAXN) return MACX_EINVAL;
  if ((size_t)(s->b0 + s->B) * s->H * s->W * (size_t)(s->Cin > s->Cmid ? s->Cin : s->Cmid) >= (1ull << 32)) return MACX_EINVAL;
  return MACX_OK;
}
hipError_t launch_pad_drop(const float* src, const PadP& q, float keep, uint32_t seed, uint32_t site, uint32_t first, float* dst,
                           uint32_t* bits, hipStream_t st) {
  const DropSpec ds = make_drop(keep, seed, site, 0);
  hipLaunchKernelGGL(pad_drop_kernel, dim3(2048), dim3(256), 0, st, src, q, ds.key, ds.thr24, ds.inv_keep, first, dst, bits);
  return hipGetLastError();
}
void conv_gemm_params(GemmP& g, const macx_stem_shapes* s, const StemGeo& geo, const float* Apad, int cin, int nout, int sign) {
  memset(&g, 0, sizeof(g));
  g.B = s->B; g.N = geo.N; g.K = 9 * cin; g.Nout = nout;
  g.A = Apad; g.lda = cin; g.a_qstride = (size_t)geo.np * cin;
  g.conv_taps = 9; g.conv_w = s->W; g.conv_wp = geo.wp; g.conv_cin = cin; g.conv_sign = sign;
  g.ldo = nout; g.e_inv_keep = 1.0f;
}
int conv_wgrad(const macx_stem_shapes* s, const StemGeo& geo, const float* Apad, int cin, const float* G, int cout, int ns, float* slab,
               float* out, hipStream_t st, const float* a_max = nullptr, const float* g_max = nullptr) {
  TnP t;
  memset(&t, 0, sizeof(t));
  t.M = s->B * geo.N; t.Kd = 9 * cin; t.Jd = cout; t.nsplit = ns; t.rows_per_split = rows_per_split(t.M, ns);
  t.A = Apad; t.lda = cin; t.a_mod = t.M; t.G = G; t.ldg = cout;
  t.conv_taps = 9; t.conv_w = s->W; t.conv_wp = geo.wp; t.conv_cin = cin; t.conv_n = geo.N; t.conv_np = geo.np;
  t.magic_n = (uint32_t)(((1ull << 32) + geo.N - 1) / geo.N);
  t.magic_w = (uint32_t)(((1ull << 32) + s->W - 1) / s->W);
  t.part = (ns == 1) ? out : slab;
  t.a_maxabs = a_max; t.g_maxabs = g_max;
  if (a_max && g_max) CK(wgrad3h_launch(t, st));      // three fp16 terms (macx_gemm3h.hip.h)
  else CK(wgrad_any(t, st));
  if (ns > 1) CK(slab_reduce_launch(slab, ns, (size_t)t.Kd * t.Jd, out, 0, st));
  return 0;
}
}  // namespace

size_t macx_stem_saved_floats(const macx_stem_shapes* s) { return stem_check(s) ? 0 : make_stem(s).total; }
size_t macx_stem_ws_floats(const macx_stem_shapes* s) { return stem_check(s) ? 0 : make_stem_bwd(s).total; }

int macx_stem_forward(const macx_stem_shapes* s, int act, float keep, uint32_t seed, const macx_stem_params* P, const float* images,
                      float* kb, float* saved, size_t saved_floats, void* stream) {
  CKI(stem_check(s));
  if (!P || !images || !kb || !saved || misaligned(images) || misaligned(kb) || misaligned(saved)) return MACX_EINVAL;
  const StemLayout L = make_stem(s);
  if (saved_floats < L.total) return MACX_ESMALL;
  hipStream_t st = (hipStream_t)stream;
  const StemGeo geo = stem_geo(s);
  const int Ci = s->Cin, Cm = s->Cmid, Co = s->Cout;
  // the default (H2) family runs the stem on three fp16 MFMA terms with one exponent per operand tensor (macx_gemm3h.hip.h);
  // the other two families keep the six-term bf16 split / the f32 MFMA
  const bool h3 = h2_mode();
  float* amax = saved + L.amax;
  Packer pk;
  if (h3) {
    CK(absmax_big(P->kernel0, (size_t)9 * Ci * Cm, amax + 0, amax + 8, st));
    CK(absmax_big(P->kernel1, (size_t)9 * Cm * Co, amax + 1, amax + 8 + AMAX_BIG_BLOCKS, st));
    pk.add(P->kernel0, Cm, 1, 9 * Ci, Cm, saved + L.k0_p, -1, -1, 3, amax + 0);
    pk.add(P->kernel1, Co, 1, 9 * Cm, Co, saved + L.k1_p, -1, -1, 3, amax + 1);
  } else {
  pk.add(P->kernel0, Cm, 1, 9 * Ci, Cm, saved + L.k0_p, -1, -1, wfmt_plain_f32ops());     // HWIO flattened = [9*Cin][Cmid] row-major
  pk.add(P->kernel1, Co, 1, 9 * Cm, Co, saved + L.k1_p, -1, -1, wfmt_plain_f32ops());
  }
  CK(pk.run(st));
  float* ascr = amax + 8 + 2 * AMAX_BIG_BLOCKS;
  PadP q0{s->B, geo.N, s->W, geo.wp, geo.np, Ci};
  PadP q1{s->B, geo.N, s->W, geo.wp, geo.np, Cm};
  // cnn_0: dropout -> conv3x3 SAME -> + b -> act   (ops.py:400-411)
  CK(launch_pad_drop(images, q0, keep, seed, SITE_STEM0, (uint32_t)((size_t)s->b0 * geo.N * Ci), saved + L.in0p, nullptr, st));
  GemmP g;
  conv_gemm_params(g, s, geo, saved + L.in0p, Ci, Cm, +1);
  g.Wp = saved + L.k0_p; g.out = saved + L.X1; g.bias = P->bias0; g.act = act;
  if (h3) {
    CK(absmax_big(saved + L.in0p, (size_t)s->B * geo.np * Ci, amax + 2, ascr, st));
    g.a_maxabs = amax + 2;
    CK((kb_gemm3h_launch<A_PLAIN, B_PLAIN, E_BIAS_ACT, false>(g, st)));
  } else
  CK((kb_gemm<A_PLAIN, B_PLAIN, E_BIAS_ACT, false>(g, st)));
  // cnn_1
  CK(launch_pad_drop(saved + L.X1, q1, keep, seed, SITE_STEM1, (uint32_t)((size_t)s->b0 * geo.N * Cm), saved + L.in1p,
                     reinterpret_cast<uint32_t*>(saved + L.bits1), st));
  conv_gemm_params(g, s, geo, saved + L.in1p, Cm, Co, +1);
  g.Wp = saved + L.k1_p; g.out = kb; g.bias = P->bias1; g.act = act;
  if (h3) {
    CK(absmax_big(saved + L.in1p, (size_t)s->B * geo.np * Cm, amax + 3, ascr + AMAX_BIG_BLOCKS, st));
    g.a_maxabs = amax + 3;
    CK((kb_gemm3h_launch<A_PLAIN, B_PLAIN, E_BIAS_ACT, false>(g, st)));
  } else
  CK((kb_gemm<A_PLAIN, B_PLAIN, E_BIAS_ACT, false>(g, st)));
  return MACX_OK;
}

int macx_stem_backward(const macx_stem_shapes* s, int act, float keep, uint32_t seed, const macx_stem_params* P, const float* kb,
                       const float* saved, size_t saved_floats, float* ws, size_t ws_floats, const float* d_kb,
                       const macx_stem_grads* G, void* stream) {
  CKI(stem_check(s));
  if (!P || !kb || !saved || !ws || !d_kb || !G) return MACX_EINVAL;
  const StemLayout L = make_stem(s);
  const StemBwdLayout W = make_stem_bwd(s);
  if (saved_floats < L.total || ws_floats < W.total) return MACX_ESMALL;
  hipStream_t st = (hipStream_t)stream;
  const StemGeo geo = stem_geo(s);
  const int Ci = s->Cin, Cm = s->Cmid, Co = s->Cout, M = s->B * geo.N;
  (void)seed;
  const bool h3 = h2_mode();
  const float* fmax = saved + L.amax;                    // [kernel0, kernel1, in0p, in1p] from the forward pass
  float* bmax = ws + W.amax;
  // backward-data weights of cnn_1: B[(tap, co)][ci] = K1[tap][ci][co]
  {
    Packer pk;
    for (int tap = 0; tap < 9; ++tap) {
      if (h3)      // nine per-tap transposes back to back = ONE [9 Cout][Cmid] matrix of format 3, one exponent behind the last
        pk.add(P->kernel1 + (size_t)tap * Cm * Co, 1, Co, Co, Cm, ws + W.k1T_p + (size_t)tap * Co * Cm, -1, -1, 3, fmax + 1,
               ws + W.k1T_p + (size_t)9 * Co * Cm);
      else
      pk.add(P->kernel1 + (size_t)tap * Cm * Co, 1, Co, Co, Cm, ws + W.k1T_p + (size_t)tap * (gemm_split_mode() ? wsize(Co, Cm) : (size_t)Co * Cm), -1, -1,
             wfmt_plain_f32ops());
    }
    CK(pk.run(st));
  }
  // dY2 = d_kb * act'(kb)
  PadP q2{s->B, geo.N, s->W, geo.wp, geo.np, Co};
  hipLaunchKernelGGL(pad_mul_actgrad_kernel, dim3(2048), dim3(256), 0, st, d_kb, kb, act, q2, ws + W.dY2, ws + W.dY2p);
  CK(hipGetLastError());
  CK(rowsum(ws + W.dY2, M, Co, Co, G->bias1, st));
  if (h3) CK(absmax_big(ws + W.dY2, (size_t)M * Co, bmax + 0, bmax + 8, st));
  CKI(conv_wgrad(s, geo, saved + L.in1p, Cm, ws + W.dY2, Co, W.ns1, ws + W.slab1, G->kernel1, st, h3 ? fmax + 3 : nullptr, h3 ? bmax + 0 : nullptr));
  // dY1 = convT(dY2) * dropmask1 * act'(X1): gather with the opposite tap offsets
  GemmP g;
  conv_gemm_params(g, s, geo, ws + W.dY2p, Co, Cm, -1);
  g.Wp = ws + W.k1T_p; g.out = ws + W.dY1; g.aux = saved + L.X1; g.act = act;
  if (keep < 1.0f) { g.e_bits = reinterpret_cast<const uint32_t*>(saved + L.bits1); g.e_inv_keep = 1.0f / keep; }
  if (h3) {
    g.a_maxabs = bmax + 0;                              // the padded copy holds the same values
    CK((kb_gemm3h_launch<A_PLAIN, B_PLAIN, E_MUL_DACT, false>(g, st)));
  } else
  CK((kb_gemm<A_PLAIN, B_PLAIN, E_MUL_DACT, false>(g, st)));
  CK(rowsum(ws + W.dY1, M, Cm, Cm, G->bias0, st));
  if (h3) CK(absmax_big(ws + W.dY1, (size_t)M * Cm, bmax + 1, bmax + 8 + AMAX_BIG_BLOCKS, st));
  CKI(conv_wgrad(s, geo, saved + L.in0p, Ci, ws + W.dY1, Cm, W.ns0, ws + W.slab0, G->kernel0, st, h3 ? fmax + 2 : nullptr, h3 ? bmax + 1 : nullptr));
  return MACX_OK;
}

int macx_images_to_nhwc(const float* nchw, int B, int C, int HW, float* nhwc, void* stream) {
  if (!nchw || !nhwc || B < 1 || C < 1 || HW < 1 || B > 65535) return MACX_EINVAL;
  hipLaunchKernelGGL(transpose_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, nchw, C, HW, nhwc);
  return (int)hipGetLastError();
}

// =================================================================================================
// question encoder (model.py:208-307; ops.biRNNLayer ops.py:859-911): SURVEY 8f row 4
// =================================================================================================
namespace {
inline int pad128(int n) { return (n + 127) & ~127; }
struct EncLayout {
  size_t wx_p, wh_p;     // packed [2][Ep x 4h], [2][h x 4h]
  size_t Xp;             // [B*S][Ep]  dropped embedded words (columns E..Ep-1 zero)
  size_t Zx;             // [2][B*S][4h]
  size_t hs, cs;         // [2][S+1][B][h]
  size_t gates;          // [2][S][B][4h]
  size_t total;
};
EncLayout make_enc(const macx_enc_shapes* s) {
  EncLayout L;
  memset(&L, 0, sizeof(L));
  size_t off = 0;
  auto take = [&](size_t n) { size_t r = off; off += al4(n); return r; };
  const size_t B = s->B, S = s->S, h = s->h, Ep = pad128(s->E), G = 4 * h;
  L.wx_p = take(2 * Ep * G); L.wh_p = take(2 * h * G);
  L.Xp = take(B * S * Ep); L.Zx = take(2 * B * S * G);
  L.hs = take(2 * (S + 1) * B * h); L.cs = take(2 * (S + 1) * B * h);
  L.gates = take(2 * S * B * G);
  L.total = off;
  return L;
}
struct EncBwdLayout {
  size_t wxT_p, whT_p, dG, dZ, dh, dc, dh_pass, dc2, dXp, tmpW, slab, dq, total;
};
EncBwdLayout make_enc_bwd(const macx_enc_shapes* s) {
  EncBwdLayout L;
  memset(&L, 0, sizeof(L));
  size_t off = 0;
  auto take = [&](size_t n) { size_t r = off; off += al4(n); return r; };
  const size_t B = s->B, S = s->S, h = s->h, Ep = pad128(s->E), G = 4 * h;
  L.wxT_p = take(2 * G * Ep); L.whT_p = take(2 * G * h);
  L.dG = take(2 * S * B * G); L.dZ = take(2 * B * S * G);
  L.dh = take(2 * B * h); L.dc = take(2 * B * h); L.dh_pass = take(2 * B * h); L.dc2 = take(2 * B * h);   // dh_pass: the second dh buffer
  L.dXp = take(B * S * Ep); L.tmpW = take(Ep * G);
  // split-reduction slabs of the two kernel-gradient contractions (input block [Ep, 4h], recurrent block [h, 4h]): the larger
  {
    const size_t in_blk = (size_t)wgrad_splits((int)(B * S), (int)Ep, (int)G) * Ep * G;
    const size_t rec_blk = (size_t)wgrad_splits((int)(B * S), (int)h, (int)G) * h * G;
    L.slab = take(in_blk > rec_blk ? in_blk : rec_blk);
  }
  L.dq = take(B * 2 * h);
  L.total = off;
  return L;
}
int enc_check(const macx_enc_shapes* s) {
  if (!s || s->B < 1 || s->S < 1 || s->V < 1 || s->E < 1 || s->h < 128 || s->h % 128) return MACX_EINVAL;
  return MACX_OK;
}
}  // namespace

size_t macx_encoder_saved_floats(const macx_enc_shapes* s) { return enc_check(s) ? 0 : make_enc(s).total; }
size_t macx_encoder_ws_floats(const macx_enc_shapes* s) { return enc_check(s) ? 0 : make_enc_bwd(s).total; }

int macx_encoder_forward(const macx_enc_shapes* s, float keep_input, float keep_question, uint32_t seed, const macx_enc_params* P,
                         const int32_t* questions, const int32_t* lengths, float* words, float* vecQ, float* saved,
                         size_t saved_floats, void* stream) {
  CKI(enc_check(s));
  if (!P || !questions || !lengths || !words || !vecQ || !saved) return MACX_EINVAL;
  const EncLayout L = make_enc(s);
  if (saved_floats < L.total) return MACX_ESMALL;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, S = s->S, E = s->E, h = s->h, Ep = pad128(E), G = 4 * h;
  const size_t Bh = (size_t)B * h;
  Packer pk;
  for (int dir = 0; dir < 2; ++dir) {
    const float* K = dir ? P->bw_kernel : P->fw_kernel;      // [E + h, 4h]: rows [0,E) input, [E,E+h) recurrent
    pk.add(K, G, 1, Ep, G, saved + L.wx_p + (size_t)dir * Ep * G, E, G);
    pk.add(K + (size_t)E * G, G, 1, h, G, saved + L.wh_p + (size_t)dir * h * G);
  }
  CK(pk.run(st));
  hipLaunchKernelGGL(embed_gather_kernel, dim3(512), dim3(256), 0, st, questions, P->emb, B * S, E, Ep, (uint32_t)s->b0 * (uint32_t)S,
                     make_drop(keep_input, seed, SITE_ENC_INPUT, 0), saved + L.Xp);
  CK(hipGetLastError());
  // input projections of every position, Zx = X Wx + b: the cell bias is added here once per position, so the per-step
  // recurrent linear needs none and both directions go in one launch (the two bias vectors are separate tensors)
  for (int dir = 0; dir < 2; ++dir) {
    LinP l = lin_basic(saved + L.Xp, Ep, Ep, B * S, saved + L.wx_p + (size_t)dir * Ep * G, dir ? P->bw_bias : P->fw_bias, G, MACX_ACT_NON,
                       saved + L.Zx + (size_t)dir * B * S * G, G);
    CK(small_linear_launch(l, 1, st));
  }
  CK(dev_zero(saved + L.hs, 2 * (size_t)(S + 1) * Bh * sizeof(float), st));
  CK(dev_zero(saved + L.cs, 2 * (size_t)(S + 1) * Bh * sizeof(float), st));
  CK(dev_zero(words, (size_t)B * S * 2 * h * sizeof(float), st));
  for (int tau = 0; tau < S; ++tau) {
    // R = h_prev Wh and the cell, both directions, one launch per time step
    LstmStepP c;
    c.B = B; c.S = S; c.h = h; c.tau = tau; c.len = lengths;
    c.Wh = saved + L.wh_p; c.Zx = saved + L.Zx; c.hs = saved + L.hs; c.cs = saved + L.cs; c.gates = saved + L.gates; c.out = words;
    hipLaunchKernelGGL(lstm_step_kernel, dim3(h / 16, (B + 15) / 16, 2), dim3(256), 0, st, c);
    CK(hipGetLastError());
  }
  // vecQuestions = dropout(concat([h_fw_final, h_bw_final]))   (ops.py:905-906, model.py:292)
  for (int dir = 0; dir < 2; ++dir)
    CK(dev_copy2d(vecQ + dir * h, (size_t)2 * h * sizeof(float), saved + L.hs + ((size_t)dir * (S + 1) + S) * Bh,
                        (size_t)h * sizeof(float), (size_t)h * sizeof(float), B, st));
  hipLaunchKernelGGL(drop2_kernel, dim3(64), dim3(256), 0, st, (const float*)vecQ, B, 2 * h, (uint32_t)s->b0,
                     make_drop(keep_question, seed, SITE_QUESTION, 0), no_drop(), vecQ);
  CK(hipGetLastError());
  return MACX_OK;
}

int macx_encoder_backward(const macx_enc_shapes* s, float keep_input, float keep_question, uint32_t seed, const macx_enc_params* P,
                          const int32_t* questions, const int32_t* lengths, const float* saved, size_t saved_floats, float* ws,
                          size_t ws_floats, const float* d_words, const float* d_vecQ, const macx_enc_grads* Gr, void* stream) {
  CKI(enc_check(s));
  if (!P || !questions || !lengths || !saved || !ws || !d_words || !d_vecQ || !Gr) return MACX_EINVAL;
  const EncLayout L = make_enc(s);
  const EncBwdLayout W = make_enc_bwd(s);
  if (saved_floats < L.total || ws_floats < W.total) return MACX_ESMALL;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, S = s->S, E = s->E, h = s->h, Ep = pad128(E), G = 4 * h;
  const size_t Bh = (size_t)B * h;
  Packer pk;
  for (int dir = 0; dir < 2; ++dir) {
    const float* K = dir ? P->bw_kernel : P->fw_kernel;
    pk.add(K, 1, G, G, Ep, ws + W.wxT_p + (size_t)dir * G * Ep, G, E);           // Wx^T: [4h] -> [Ep], columns >= E zero
    pk.add(K + (size_t)E * G, 1, G, G, h, ws + W.whT_p + (size_t)dir * G * h);   // Wh^T: [4h] -> [h]
  }
  CK(pk.run(st));
  CK(dev_zero(ws + W.dZ, 2 * (size_t)B * S * G * sizeof(float), st));
  CK(dev_zero(ws + W.dc, 2 * Bh * sizeof(float), st));
  // d(final states) = d_vecQ through the question dropout, split per direction
  hipLaunchKernelGGL(drop2_kernel, dim3(64), dim3(256), 0, st, d_vecQ, B, 2 * h, (uint32_t)s->b0,
                     make_drop(keep_question, seed, SITE_QUESTION, 0), no_drop(), ws + W.dq);
  for (int dir = 0; dir < 2; ++dir)
    CK(dev_copy2d(ws + W.dh + (size_t)dir * Bh, (size_t)h * sizeof(float), ws + W.dq + dir * h, (size_t)2 * h * sizeof(float),
                        (size_t)h * sizeof(float), B, st));
  {
    const size_t lds = lstm_step_bwd_lds(h);
    CK(lds_attr_once(reinterpret_cast<const void*>(lstm_step_bwd_kernel), lds));
    // the running dh / dc alternate between two buffers (a workgroup reads what its neighbours would otherwise overwrite)
    float* dhb[2] = {ws + W.dh, ws + W.dh_pass};
    float* dcb[2] = {ws + W.dc, ws + W.dc2};
    int cur = 0;
    for (int tau = S - 1; tau >= 0; --tau, cur ^= 1) {
      LstmStepBwdP c;
      c.B = B; c.S = S; c.h = h; c.tau = tau; c.len = lengths;
      c.WhT = ws + W.whT_p; c.cs = saved + L.cs; c.gates = saved + L.gates; c.dout = d_words;
      c.dh_in = dhb[cur]; c.dc_in = dcb[cur]; c.dh_out = dhb[cur ^ 1]; c.dc_out = dcb[cur ^ 1];
      c.dG = ws + W.dG; c.dZ = ws + W.dZ;
      hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3(h / 16, (B + 15) / 16, 2), dim3(LSB_THREADS), lds, st, c);
      CK(hipGetLastError());
    }
  }
  for (int dir = 0; dir < 2; ++dir) {
    float* dK = dir ? Gr->bw_kernel : Gr->fw_kernel;
    // input block of the kernel: X^T dZ (rows [0,E) of the padded product)
    CKI(wgrad_impl(saved + L.Xp, Ep, ws + W.dZ + (size_t)dir * B * S * G, G, B * S, Ep, G, ws + W.tmpW, ws + W.slab, st));
    CK(dev_copy(dK, ws + W.tmpW, (size_t)E * G * sizeof(float), st));
    // recurrent block: sum_tau h_prev(tau)^T dG_tau  -- one contraction over S*B rows
    CKI(wgrad_impl(saved + L.hs + (size_t)dir * (S + 1) * Bh, h, ws + W.dG + (size_t)dir * S * B * G, G, S * B, h, G,
                   dK + (size_t)E * G, ws + W.slab, st));
    CK(rowsum(ws + W.dG + (size_t)dir * S * B * G, S * B, G, G, dir ? Gr->bw_bias : Gr->fw_bias, st));
  }
  // d(embedded words) = [dZ_fw | dZ_bw] [Wx_fw^T ; Wx_bw^T], then through the input dropout into the embedding rows
  {
    LinP l = lin_basic(ws + W.dZ, G, G, B * S, ws + W.wxT_p, nullptr, Ep, MACX_ACT_NON, ws + W.dXp, Ep);
    l.seg[1] = LinSeg{ws + W.dZ + (size_t)B * S * G, G, G, 0};
    l.Ktot = 2 * G;
    CK(small_linear_launch(l, 1, st));
  }
  hipLaunchKernelGGL(embed_grad_kernel, dim3(s->V, (E + 1023) / 1024), dim3(256), 0, st, questions, (const float*)(ws + W.dXp), B * S, E, Ep,
                     (uint32_t)s->b0 * (uint32_t)S, make_drop(keep_input, seed, SITE_ENC_INPUT, 0), Gr->emb);
  CK(hipGetLastError());
  return MACX_OK;
}

// =================================================================================================
// optimizer step (model.py:615-669): SURVEY 8f row 3
// =================================================================================================
int macx_adam_ema_step(size_t n, float* params, const float* grads, float* m, float* v, float* ema, float lr, float beta1,
                       float beta2, float eps, int step, float clip_norm, float ema_decay, float* ws, float* norm_out,
                       void* stream) {
  if (!params || !grads || !m || !v || !ws || n == 0 || step < 1) return MACX_EINVAL;
  if (ema_decay >= 0.f && !ema) return MACX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)((n + 255) / 256);
  if (blocks > OPT_BLOCKS) blocks = OPT_BLOCKS;
  hipLaunchKernelGGL(opt_sumsq_kernel, dim3(blocks), dim3(256), 0, st, grads, n, ws);
  OptP q;
  q.n = n; q.p = params; q.g = grads; q.m = m; q.v = v; q.ema = ema;
  // tf.train.AdamOptimizer: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
  q.lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step)));
  q.beta1 = beta1; q.beta2 = beta2; q.eps = eps; q.clip = clip_norm; q.ema_decay = ema_decay;
  q.part = ws; q.nparts = blocks; q.norm_out = norm_out;
  hipLaunchKernelGGL(opt_apply_kernel, dim3(blocks), dim3(256), 0, st, q);
  CK(hipGetLastError());
  return MACX_OK;
}

/* test/tuning hook: key 0 = waves per workgroup of the kb GEMM (4 or 8) */
int macx_gemm_mode(int mode) {
  if (mode == MACX_GEMM_NATIVE || mode == MACX_GEMM_SPLIT || mode == MACX_GEMM_H2) gemm_default_mode() = mode;
  return gemm_split_mode();
}

size_t macx_h2_floats(size_t rows, size_t cols) { return (cols % 128) ? 0 : al4(h2_floats(rows, cols)); }

int macx_h2_from_f32(const float* src, int B, int N, int C, float* h2, void* stream) {
  if (!src || !h2 || B < 1 || N < 1 || C < 128 || C % 128 || misaligned(src) || misaligned(h2)) return MACX_EINVAL;
  H2FromP f;
  memset(&f, 0, sizeof(f));
  f.src = src; f.B = B; f.N = N; f.C = C; f.out = h2_view(h2, B * N, C);
  f.thr24 = 1u << 24; f.inv_keep = 1.0f; f.thr24_2 = 1u << 24;
  CK(h2_from_f32(f, (hipStream_t)stream));
  return MACX_OK;
}

int macx_h2_to_f32(const float* h2, int rows, int C, float* out, void* stream) {
  if (!h2 || !out || rows < 1 || C < 128 || C % 128) return MACX_EINVAL;
  hipLaunchKernelGGL(h2_to_f32_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, h2_view(h2, rows, C), out);
  CK(hipGetLastError());
  return MACX_OK;
}

int macx_h2_pack_weight(const float* Wm, int K, int n_out, int transpose, float* out, void* stream) {
  if (!Wm || !out || K < 128 || K % 128 || n_out < 128 || n_out % 128 || misaligned(Wm) || misaligned(out)) return MACX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* wmax = out + al4((size_t)K * n_out + 4);
  CK(absmax4(Wm, (size_t)K * n_out, Wm, 1, Wm, 1, Wm, 1, wmax, nullptr, st));
  Packer pk;
  if (transpose) pk.add(Wm, 1, K, K, n_out, out, -1, -1, 3, wmax);
  else pk.add(Wm, n_out, 1, K, n_out, out, -1, -1, 3, wmax);
  CK(pk.run(st));
  return MACX_OK;
}

int macx_h2_gemm_planes(const float* hA, int B, int N, int K, const float* Wh, int n_out, const float* bias, int act, float* hO,
                        void* stream) {
  if (!hA || !Wh || !bias || !hO || B < 1 || N < 1 || N > K_MAXN || K < 128 || K % 128 || n_out < 128 || n_out % 128)
    return MACX_EINVAL;
  GemmH2P g;
  memset(&g, 0, sizeof(g));
  g.B = B; g.N = N; g.K = K; g.Nout = n_out;
  g.A = h2_view(hA, B * N, K);
  g.Wh = reinterpret_cast<const char*>(Wh); g.w_exp = reinterpret_cast<const int*>(Wh) + (size_t)K * n_out;
  g.out = h2_view(hO, B * N, n_out); g.bias = bias; g.act = act; g.e_inv_keep = 1.0f;
  g.dbg = kb_gemm_dbg();
  static const int reps = getenv("MACX_H2_DEBUG_REPS") ? atoi(getenv("MACX_H2_DEBUG_REPS")) : 1;   // timing aid (tools/h2_gemm_time.py)
  for (int r = 0; r < reps; ++r) CK((kb_gemm_h2_launch<B_PLAIN, E_BIAS_ACT, false>(g, (hipStream_t)stream)));
  return MACX_OK;
}

int macx_h2_gemm(const float* A, int B, int N, int K, const float* Wm, int n_out, const float* bias, int act, float* out,
                 float* ws, size_t ws_floats, void* stream) {
  if (!A || !Wm || !bias || !out || !ws || B < 1 || N < 1 || N > K_MAXN || K < 128 || K % 128 || n_out < 128 || n_out % 128)
    return MACX_EINVAL;
  const size_t fa = al4(h2_floats((size_t)B * N, K)), fo = al4(h2_floats((size_t)B * N, n_out));
  const size_t fw = al4((size_t)K * n_out + 4);
  if (ws_floats < fa + fo + fw + 8) return MACX_ESMALL;
  float* hA = ws; float* hO = ws + fa; float* wp = hO + fo;
  CKI(macx_h2_from_f32(A, B, N, K, hA, stream));
  CKI(macx_h2_pack_weight(Wm, K, n_out, 0, wp, stream));
  CKI(macx_h2_gemm_planes(hA, B, N, K, wp, n_out, bias, act, hO, stream));
  return macx_h2_to_f32(hO, B * N, n_out, out, stream);
}

/* bench / profiling hook: the read unit's forward chain kernel of `step`, re-launched `reps` times on `stream` between two HIP
   events; *ms_out = average milliseconds per launch.  `saved` must come from macx_cell_begin + macx_cell_step(.., step) with
   keep = 1 (the step's y and control exist then); the outputs of launch r go to the buffers of step (step + r) % p, so the run
   must not be differentiated afterwards. */
int macx_read_chain_time(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                         const macx_inputs* in, float* saved, size_t saved_floats, int step, int reps, float* ms_out,
                         void* stream) {
  ModeScope ms(o);
  CKI(check_impl(o, s));
  if (!dp || !P || !in || !saved || !ms_out || reps < 1 || step < 0 || step >= s->p) return MACX_EINVAL;
  if (!use_chain(s->d, s->N)) return MACX_EUNSUPPORTED;
  const SavedLayout L = make_saved(o, s, 1);
  if (saved_floats < L.total) return MACX_ESMALL;
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(chain_fwd_launch(make_chain_fwd(o, s, dp, P, in, saved, L, 1, step, step), st));      // untimed: code object, LDS attribute
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) CK(chain_fwd_launch(make_chain_fwd(o, s, dp, P, in, saved, L, 1, step, (step + r) % s->p), st));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float t = 0.f;
  CK(hipEventElapsedTime(&t, e0, e1));
  *ms_out = t / reps;
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return MACX_OK;
}

/* The same kernel timed IN A RUNNING FORWARD PASS: one macx_cell_forward (keep = 1) on `stream`, each of its p chain launches issued
   through hipExtLaunchKernelGGL with a start and a stop event -- the kernel's own dispatch timestamps, what a kernel trace reports;
   in front of a launch sits the step's [B,d] linear, behind it the attention kernel, as in a training step.  *ms_out = average
   milliseconds per launch (synchronises `stream`).  What bench.py's roofline.kernel_ms reports. */
int macx_cell_forward_chain_time(const macx_opts* o, const macx_shapes* s, const macx_dropout* dp, const macx_params* P,
                                 const macx_inputs* in, float* saved, size_t saved_floats, float* ws, size_t ws_floats, float* ms_out,
                                 void* stream) {
  ModeScope ms(o);
  CKI(check_impl(o, s));
  if (!ms_out) return MACX_EINVAL;
  if (!use_chain(s->d, s->N)) return MACX_EUNSUPPORTED;
  ChainProbe& cp = *chain_probe();
  if (cp.on) return MACX_EINVAL;
  const int ne = 2 * (s->p < 64 ? s->p : 64);
  for (int i = 0; i < ne; ++i) CK(hipEventCreate(&cp.ev[i]));
  // three untimed passes in front of the timed one, nothing but launches in between: the timed pass runs on a chip that has been
  // busy for milliseconds, like a pass inside a training loop (a lone pass after a host synchronisation measured 10 % slow: clocks)
  int rc = MACX_OK;
  for (int w = 0; w < 3 && rc == MACX_OK; ++w) rc = macx_cell_forward(o, s, dp, P, in, saved, saved_floats, ws, ws_floats, 1, stream);
  cp.n = 0; cp.on = true;
  if (rc == MACX_OK) rc = macx_cell_forward(o, s, dp, P, in, saved, saved_floats, ws, ws_floats, 1, stream);
  cp.on = false;
  int err = rc;
  if (rc == MACX_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) err = MACX_EINVAL;
  double total = 0.0;
  for (int k = 0; err == MACX_OK && k < cp.n; ++k) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, cp.ev[2 * k], cp.ev[2 * k + 1]) != hipSuccess) err = MACX_EINVAL;
    total += t;
  }
  for (int i = 0; i < ne; ++i) (void)hipEventDestroy(cp.ev[i]);
  if (err != MACX_OK) return err;
  if (cp.n < 1) return MACX_EUNSUPPORTED;
  *ms_out = (float)(total / cp.n);
  return MACX_OK;
}

// ---- the control unit's question projections (mac_cell.py:442-448) behind a contract of their own -----------------
//   t = act(vecQ Wq + bq)  [B,d]  (controlInputAct);   cI_i = t WqU_i + bqU_i  [p,B,d]  (one matrix per step with
//   controlInputUnshared, else the same one p times) -- the two launches macx_cell_begin issues for them.
size_t macx_ctrl_inputs_ws_floats(const macx_opts* o, const macx_shapes* s) {
  if (!o || !s || s->d < 1 || s->B < 1 || s->p < 1) return 0;
  const size_t dd = (size_t)s->d * s->d, Bd = (size_t)s->B * s->d, nU = o->control_input_unshared ? s->p : 1;
  return (1 + nU) * dd + 3 * Bd + (size_t)wgrad_splits(s->B, s->d, s->d) * dd + 64;
}

int macx_ctrl_inputs_fwd(const macx_opts* o, const macx_shapes* s, const macx_params* P, const float* vecQuestions, float* ctrl_t,
                         float* ctrl_inputs, float* ws, size_t ws_floats, void* stream) {
  ModeScope ms(o);
  CKI(check_impl(o, s));
  if (!P || !vecQuestions || !ctrl_t || !ctrl_inputs || !ws || !P->qInput_W || !P->qInputU_W) return MACX_EINVAL;
  if (misaligned(vecQuestions) || misaligned(ctrl_t) || misaligned(ctrl_inputs) || misaligned(ws)) return MACX_EINVAL;
  if (ws_floats < macx_ctrl_inputs_ws_floats(o, s)) return MACX_ESMALL;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, d = s->d, p = s->p;
  const size_t dd = (size_t)d * d;
  const int nU = o->control_input_unshared ? p : 1;
  float* wq_p = ws;
  float* wqU_p = ws + dd;
  Packer pk;
  pk.add(P->qInput_W, d, 1, d, d, wq_p);
  for (int i = 0; i < nU; ++i) {
    if (pk.n == PACK_MAX) CK(pk.run(st));
    pk.add(P->qInputU_W + (size_t)i * dd, d, 1, d, d, wqU_p + (size_t)i * dd);
  }
  CK(pk.run(st));
  LinP l = lin_basic(vecQuestions, d, d, B, wq_p, P->qInput_b, d, o->control_input_act, ctrl_t, d);
  CK(small_linear_launch(l, 1, st));
  LinP u = lin_basic(ctrl_t, d, d, B, wqU_p, P->qInputU_b, d, MACX_ACT_NON, ctrl_inputs, d);
  if (o->control_input_unshared) { u.zW = dd; u.zb = d; }
  u.zout = (size_t)B * d;
  CK(small_linear_launch(u, p, st));
  return MACX_OK;
}

// backward of the above: d_ctrl_inputs [p,B,d] -> d_vecQuestions [B,d] and the gradients of qInput / qInputU (the non-NULL fields of
// `GP`: qInput_W [d,d], qInput_b [d], qInputU_W [nU,d,d], qInputU_b [nU,d]) -- the arithmetic of the cell's backward pass for this
// unit (SURVEY appendix A rows "qInput", "qInput{i}"): dt = sum_i dcI_i WqU_i^T, du = dt * act'(t), dvecQ = du Wq^T.
int macx_ctrl_inputs_bwd(const macx_opts* o, const macx_shapes* s, const macx_params* P, const float* vecQuestions, const float* ctrl_t,
                         const float* d_ctrl_inputs, const macx_param_grads* GP, float* d_vecQuestions, float* ws, size_t ws_floats,
                         void* stream) {
  ModeScope ms(o);
  CKI(check_impl(o, s));
  if (!P || !vecQuestions || !ctrl_t || !d_ctrl_inputs || !GP || !d_vecQuestions || !ws || !P->qInput_W || !P->qInputU_W) return MACX_EINVAL;
  if (misaligned(vecQuestions) || misaligned(ctrl_t) || misaligned(d_ctrl_inputs) || misaligned(d_vecQuestions) || misaligned(ws)) return MACX_EINVAL;
  if (ws_floats < macx_ctrl_inputs_ws_floats(o, s)) return MACX_ESMALL;
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, d = s->d, p = s->p;
  const size_t dd = (size_t)d * d, Bd = (size_t)B * d;
  const int nU = o->control_input_unshared ? p : 1;
  float* wqT = ws;
  float* wqUT = wqT + dd;
  float* dt = wqUT + (size_t)nU * dd;
  float* du = dt + Bd;
  float* dsum = du + Bd;
  float* slab = dsum + Bd;
  Packer pk;
  pk.add(P->qInput_W, 1, d, d, d, wqT);
  for (int i = 0; i < nU; ++i) {
    if (pk.n == PACK_MAX) CK(pk.run(st));
    pk.add(P->qInputU_W + (size_t)i * dd, 1, d, d, d, wqUT + (size_t)i * dd);
  }
  CK(pk.run(st));
  if (o->control_input_unshared) {
    LinP li = lin_basic(d_ctrl_inputs, d, d, B, wqUT, nullptr, d, MACX_ACT_NON, dt, d);
    li.Ktot = p * d;
    li.rep_stride = Bd;
    CK(small_linear_launch(li, 1, st));
    if (GP->qInputU_W)
      for (int i = 0; i < p; ++i)
        CKI(wgrad_impl(ctrl_t, d, d_ctrl_inputs + (size_t)i * Bd, d, B, d, d, GP->qInputU_W + (size_t)i * dd, slab, st));
    CK(rowsum(d_ctrl_inputs, B, d, d, GP->qInputU_b, st, p, Bd, d));
  } else {
    hipLaunchKernelGGL(sum_parts_kernel, dim3(256), dim3(256), 0, st, d_ctrl_inputs, p, Bd, dsum);
    CK(hipGetLastError());
    LinP ls = lin_basic(dsum, d, d, B, wqUT, nullptr, d, MACX_ACT_NON, dt, d);
    CK(small_linear_launch(ls, 1, st));
    if (GP->qInputU_W) CKI(wgrad_impl(ctrl_t, d, dsum, d, B, d, d, GP->qInputU_W, slab, st));
    CK(rowsum(dsum, B, d, d, GP->qInputU_b, st));
  }
  hipLaunchKernelGGL(mul_actgrad_kernel, dim3(64), dim3(256), 0, st, (const float*)dt, ctrl_t, o->control_input_act, Bd, du);
  CK(hipGetLastError());
  LinP l = lin_basic(du, d, d, B, wqT, nullptr, d, MACX_ACT_NON, d_vecQuestions, d);
  CK(small_linear_launch(l, 1, st));
  if (GP->qInput_W) CKI(wgrad_impl(vecQuestions, d, du, d, B, d, d, GP->qInput_W, slab, st));
  CK(rowsum(du, B, d, d, GP->qInput_b, st));
  return MACX_OK;
}

// One of the read unit's kept [B*N, d] activations of step `step` as fp32 row-major: which = 0 dropout(KB) (ops.py:678), 1 X
// (ops.py:688), 2 H1 (ops.py:718), 3 I2 (ops.py:326) -- what the forward pass left in `saved` (keep = 1) for the backward pass,
// whatever format the kernel family keeps it in (H2 planes in the default family).  Inspection / tests: the chain kernel's
// intermediate products are checked against fp64 through this (tests/test_gpu_h2.py).
int macx_saved_activation(const macx_opts* o, const macx_shapes* s, int which, int step, const float* saved, size_t saved_floats,
                          float* out, void* stream) {
  ModeScope ms(o);
  CKI(check_impl(o, s));
  if (!saved || !out || which < 0 || which > 3 || step < 0 || step >= s->p || misaligned(out)) return MACX_EINVAL;
  const SavedLayout L = make_saved(o, s, 1);
  if (saved_floats < L.total) return MACX_ESMALL;
  const size_t base = which == 0 ? L.KBd : (which == 1 ? L.X : (which == 2 ? L.H1 : L.I2));
  const float* src = saved + base + (size_t)step * L.act_stride;
  hipStream_t st = (hipStream_t)stream;
  const size_t R = (size_t)s->B * s->N;
  if (h2_mode()) {
    hipLaunchKernelGGL(h2_to_f32_kernel, dim3(1024), dim3(256), 0, st, h2_view(src, (int)R, s->d), out);
    CK(hipGetLastError());
  } else {
    CK(dev_copy(out, src, R * s->d * sizeof(float), st));
  }
  return MACX_OK;
}

int macx_wgrad_splits(int M, int Kd, int Jd) {
  if (M < 1 || Kd < 128 || Jd < 128 || Kd % 128 || Jd % 128) return MACX_EINVAL;
  return wgrad_splits(M, Kd, Jd);
}

int macx_wgrad(const float* A, int lda, const float* G, int ldg, int M, int Kd, int Jd, float* out, float* ws, void* stream) {
  if (!A || !G || !out || !ws) return MACX_EINVAL;
  if (M < 1 || Kd < 128 || Jd < 128 || Kd % 128 || Jd % 128 || lda % 4 || ldg % 4) return MACX_EINVAL;
  return wgrad_impl(A, lda, G, ldg, M, Kd, Jd, out, ws, (hipStream_t)stream);
}

}  // extern "C"
