// macx_lin_tile.hip.h -- one output tile of a [B,d]-sized linear (ops.linear, ops.py:298-333): the arithmetic of small_linear_kernel
// (macx_small.hip.h) as a device function, so that the forward chain kernel's filler workgroups (macx_chain_h2.hip.h) can compute
// the read unit's y = md Wy + by with exactly the launch's products, summation order and epilogue.
#pragma once
#include "macx_common.hip.h"

namespace macx {

struct LinSeg { const float* x; int ld; int K; size_t zstride; };
// rep_stride != 0: the input is seg[0] repeated along k -- block r (k in [r K0, (r+1) K0)) lives at seg[0].x + r * rep_stride
// (per-step [B,d] tensors stored [p][B][d] read as one [B, p d] operand)
struct LinP {
  LinSeg seg[3];
  int Ktot, rows, n_out;
  const float* W; size_t zW;        // packed [Ktot/16][4][n_out][4]
  const float* bias; size_t zb; float bias_const;
  int act;
  float* out; int ldo; size_t zout;
  // optional epilogue pieces (applied in this order after bias/act)
  const float* actgrad_src; int actgrad_act; int ld_ag; size_t zag;  // val *= act'(src)
  int use_drop; DropSpec d1, d2; uint32_t drop_row0;                // val *= f1*f2, idx=(drop_row0+r)*drop_ld+j
  int drop_ld;                      // row stride of the dropout index: the logical width of a zero-padded cell; 0 = n_out
  float* out_drop; int ld_od;       // use_drop == 2: `out` keeps val, out_drop receives val*f1*f2 (the next consumer's dropped copy)
  const float* addend; int ld_add; size_t zadd;                      // val += addend
  size_t rep_stride;
  // PART form (small_linear_part_launch): the input row of question b is the sum of the chain kernel's per-tile partials
  // part[tile][3][Ktot] over the 64-row tiles the question's `part_N` rows touch (macx_chain_h2.hip.h, dy_part) -- the reduction
  // that would otherwise be a launch of its own in front of this one.  Column block 0 also writes the summed rows to part_sum.
  const float* part; int part_N; float* part_sum; int part_shift;      // part_shift: log2 of the rows per tile
};
constexpr int LIN_PART_TILES = 6;     // tiles a question may touch in the PART form: N <= 320

// RTL row tiles of 16 per workgroup: 4 for tall inputs; 1 for the [B,d] chain (B <= 128 rows), where 64-row workgroups
// would leave a 32-workgroup grid on a 256-CU chip and the launch is pure latency.
// one (16 RTL rows) x (16 columns) output tile of a LinP: tile (bx, by) of matrix z.  `red`: [4][16 RTL][20] floats of LDS.
// Every thread of the workgroup must call it (barrier inside); callable more than once per kernel (a barrier guards `red`).
// tid: 0 .. 255 within the four waves that work on this tile (a 512-thread workgroup runs two tiles side by side: both halves call
// with their own tile, tid and `red`, and meet in the same barrier).  live = false: a half without a tile goes through the motions
// and stores nothing.  COH: the outputs leave as agent-scope stores (written through the XCD's L2: another workgroup of the SAME
// launch reads them with agent-scope loads, macx_chain_h2.hip.h).
// COHIN: the input rows are read with agent-scope loads too (another workgroup of the same launch wrote them).
// CTL: 16-column tiles side by side (1 for the launches; 2 -- `red` then holds [2 x 4] slabs -- where fewer, wider tiles save a
// round on a handful of workgroups): every output keeps its products and its summation order.
template <int RTL, bool PART = false, bool COH = false, bool COHIN = false, int CTL = 1>
__device__ __forceinline__ void small_linear_tile(const LinP& p, int bx, int by, int z, float (*red)[16 * RTL][20], int tid, bool live = true) {
  constexpr int L_ROWS = 16 * RTL;
  constexpr int NWV = 4;               // waves that share the tile's reduction dimension
  static_assert(CTL == 1 || (CTL == 2 && !PART), "one or two column tiles");
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = bx * 16 * CTL;
  const int r0 = by * L_ROWS;
  const int li = lane & 15, lg = lane >> 4;

  f32x4 acc[RTL][CTL];
#pragma unroll
  for (int t = 0; t < RTL; ++t)
#pragma unroll
    for (int c = 0; c < CTL; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  int rowc[RTL];
#pragma unroll
  for (int t = 0; t < RTL; ++t) rowc[t] = min(r0 + 16 * t + li, p.rows - 1);

  const float* Wz = p.W + (size_t)z * p.zW + ((size_t)lg * p.n_out + c0 + li) * 4;
  const int nQ = p.Ktot >> 4;
  // the epilogue's global operands (bias, addend, the activation output act' is taken from) do not depend on the product: they are
  // requested HERE, in front of the K loop, so that their L2 / HBM round trip runs under the loop's instead of behind the
  // cross-wave combine (round 6: one dependent round trip less in a launch that is a chain of three)
  const int er = tid / (4 * CTL), ecq = (tid % (4 * CTL)) * 4;
  const int erow = min(r0 + er, p.rows - 1), ecol = c0 + ecq;
  f32x4 e_bias = {0.f, 0.f, 0.f, 0.f}, e_add = {0.f, 0.f, 0.f, 0.f}, e_ag = {0.f, 0.f, 0.f, 0.f};
  auto ld4 = [](const float* q) __attribute__((always_inline)) {       // 16-byte load where the caller's pointer allows it
    if ((reinterpret_cast<uintptr_t>(q) & 15) == 0) return *reinterpret_cast<const f32x4*>(q);
    return f32x4{q[0], q[1], q[2], q[3]};
  };
  if (er < L_ROWS) {
    if (p.bias) e_bias = ld4(p.bias + (size_t)z * p.zb + ecol);
    if (p.addend) e_add = ld4(p.addend + (size_t)z * p.zadd + (size_t)erow * p.ld_add + ecol);
    if (p.actgrad_src) e_ag = ld4(p.actgrad_src + (size_t)z * p.zag + (size_t)erow * p.ld_ag + ecol);
  }
  // Operands are L2-resident and the chain is latency-bound: fetch the fragments of 8 k-groups
  // (40 x 16 B per lane in flight) before touching the matrix pipe.
  constexpr int PF = PART ? 4 : 8;
  // PART: the partial rows of this lane's question (fixed order: ascending tiles, as dc_reduce_kernel sums them)
  const float* pbase[LIN_PART_TILES];
  bool pvalid[LIN_PART_TILES];
  if (PART) {
    const uint32_t first = (uint32_t)rowc[0] * (uint32_t)p.part_N;
    const int t0 = (int)(first >> p.part_shift), t1 = (int)((first + p.part_N - 1) >> p.part_shift);
#pragma unroll
    for (int j = 0; j < LIN_PART_TILES; ++j) {
      const int tt = min(t0 + j, t1);
      const int seg = rowc[0] - (int)(((uint32_t)tt << p.part_shift) / (uint32_t)p.part_N);
      pbase[j] = p.part + ((size_t)tt * 3 + seg) * p.Ktot + lg * 4;
      pvalid[j] = t0 + j <= t1;
    }
  }
  for (int Q0 = wave; Q0 < nQ; Q0 += NWV * PF) {
    f32x4 bf[PF][CTL], af[PF][RTL];
    if (PART) {
      f32x4 pv[PF][LIN_PART_TILES];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int Q = min(Q0 + NWV * u, nQ - 1);
        bf[u][0] = *reinterpret_cast<const f32x4*>(Wz + (size_t)Q * 4 * p.n_out * 4);
#pragma unroll
        for (int j = 0; j < LIN_PART_TILES; ++j) pv[u][j] = *reinterpret_cast<const f32x4*>(pbase[j] + Q * 16);
      }
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        f32x4 sum = pv[u][0];
#pragma unroll
        for (int j = 1; j < LIN_PART_TILES; ++j) sum += pvalid[j] ? pv[u][j] : f32x4{0.f, 0.f, 0.f, 0.f};
        af[u][0] = sum;
        const int Q = Q0 + NWV * u;
        if (live && bx == 0 && Q < nQ && r0 + li < p.rows)
          *reinterpret_cast<f32x4*>(p.part_sum + (size_t)(r0 + li) * p.Ktot + Q * 16 + lg * 4) = sum;
      }
    } else {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int Q = min(Q0 + NWV * u, nQ - 1);
      int s = 0, koff = Q * 16;
      size_t rep_off = 0;
      if (p.rep_stride) {
        const int r = koff / p.seg[0].K;
        koff -= r * p.seg[0].K;
        rep_off = (size_t)r * p.rep_stride;
      } else {
        while (koff >= p.seg[s].K) { koff -= p.seg[s].K; ++s; }
      }
      const float* xs = p.seg[s].x + (size_t)z * p.seg[s].zstride + rep_off + koff + lg * 4;
      const int ld = p.seg[s].ld;
#pragma unroll
      for (int c = 0; c < CTL; ++c) bf[u][c] = *reinterpret_cast<const f32x4*>(Wz + (size_t)Q * 4 * p.n_out * 4 + c * 64);
#pragma unroll
      for (int t = 0; t < RTL; ++t) {
        if constexpr (COHIN) {
          const uint64_t* q8 = reinterpret_cast<const uint64_t*>(xs + (size_t)rowc[t] * ld);
          const uint64_t lo = __hip_atomic_load(q8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint64_t hi = __hip_atomic_load(q8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          af[u][t] = f32x4{__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float((uint32_t)hi),
                           __uint_as_float((uint32_t)(hi >> 32))};
        } else {
          af[u][t] = *reinterpret_cast<const f32x4*>(xs + (size_t)rowc[t] * ld);
        }
      }
    }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (Q0 + NWV * u < nQ) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < RTL; ++t)
#pragma unroll
            for (int c = 0; c < CTL; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][t][e], bf[u][c][e], acc[t][c], 0, 0, 0);
      }
    }
  }
  // accumulator map: col = lane & 15, row = 16 t + (lane >> 4) * 4 + e
#pragma unroll
  for (int t = 0; t < RTL; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < CTL; ++c) red[c * NWV + wave][16 * t + lg * 4 + e][li] = acc[t][c][e];
  __syncthreads();
  const int r = tid / (4 * CTL), cqt = (tid % (4 * CTL)) * 4, cs = (cqt >> 4) * NWV, cq = cqt & 15;
  const int row = r0 + r;
  if (live && r < L_ROWS && row < p.rows) {
  f32x4 val, vald = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t = ((red[cs][r][cq + e] + red[cs + 1][r][cq + e]) + red[cs + 2][r][cq + e]) + red[cs + 3][r][cq + e];      // fixed order
    val[e] = t;
  }
  const int col = c0 + cqt;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = val[e];
    v += e_bias[e];
    v += p.bias_const;
    v = act_apply(p.act, v);
    if (p.actgrad_src) v *= act_grad_from_out(p.actgrad_act, e_ag[e]);
    if (p.use_drop) {
      const uint32_t idx = (p.drop_row0 + row) * (uint32_t)(p.drop_ld > 0 ? p.drop_ld : p.n_out) + col + e;
      float f = 1.f;
      if (!keep_bit(idx, run_key(p.d1), p.d1.thr24)) f = 0.f; else f *= p.d1.inv_keep;
      if (!keep_bit(idx, run_key(p.d2), p.d2.thr24)) f = 0.f; else f *= p.d2.inv_keep;
      if (p.use_drop == 2) vald[e] = v * f;
      else v *= f;
    }
    v += e_add[e];
    val[e] = v;
  }
  if (COH) {
    float* o = p.out + (size_t)z * p.zout + (size_t)row * p.ldo + col;
#pragma unroll
    for (int e = 0; e < 4; ++e) __hip_atomic_store(o + e, val[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *reinterpret_cast<f32x4*>(p.out + (size_t)z * p.zout + (size_t)row * p.ldo + col) = val;
  }
  if (p.use_drop == 2) {
    if (COH) {
      float* o = p.out_drop + (size_t)row * p.ld_od + col;
#pragma unroll
      for (int e = 0; e < 4; ++e) __hip_atomic_store(o + e, vald[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      *reinterpret_cast<f32x4*>(p.out_drop + (size_t)row * p.ld_od + col) = vald;
    }
  }
  }
}

}  // namespace macx
