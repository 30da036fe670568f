// macx_gemm3h.hip.h -- the fp32-operand GEMM / implicit 3 x 3 convolution of macx_gemm6.hip.h on THREE fp16 MFMA terms instead
// of six bf16 ones (round 4; the stem CNN, model.py:165-204, was the last large contraction on the 6-term split).
//
// Same tiling, staging and epilogues as kb_gemm6_kernel (per-image tiles of RT*16 rows x 128 columns, 8 waves as 2 row halves
// x 4 column groups, fp32 operands loaded to registers one K slice ahead and split while they are stored to LDS).  What
// changes is the split: x * 2^e = hi + lo as two fp16 (the H2 element format of macx_h2.hip.h, |error| <= 2^-24 |x 2^e|), with
// ONE exponent per operand TENSOR instead of H2's one per (row, 128 columns): the A operand's from its largest magnitude
// (GemmP::a_maxabs, a device float an absmax pass left), the weights' from their pack (format 3: planes + exponent).  A
// product is a_lo b_hi + a_hi b_lo + a_hi b_hi on v_mfma_f32_16x16x32_f16, fp32 accumulate, smallest terms first.
// What one exponent per tensor costs: an element below 2^-17 of its tensor's largest keeps fewer than 22 significant bits
// (its lo part leaves fp16's normal range) -- an absolute error below 2^-38 of the tensor's maximum, invisible in a sum over
// K >= 1152 terms of feature-map data; the stem's tests hold their round-3 tolerances (tests/test_gpu_stem.py).
// B_PLAIN weights only (the per-question weight mixing of the read unit has its own kernels).
#pragma once
#include "macx_gemm6.hip.h"
#include "macx_gemm_tn.hip.h"
#include "macx_h2.hip.h"

namespace macx {

constexpr int GS_PAD3 = 0;

template <int RT>
constexpr int kb_gemm3h_lds_bytes() {
  constexpr int ROWS = RT * 16;
  constexpr int stage = 2 * 4 * (ROWS * 16 + GS_PAD3);
  constexpr int epi = (ROWS * (128 + 4) + 256 * 8) * 4;
  constexpr int loop = 2 * stage + 2 * 512 * 16;   // two stages + the scratch slots (both planes) of threads without a slot
  return loop > epi ? loop : epi;
}

template <int RT, int AP, int BP, int EP, bool COLSUM>
__global__ __launch_bounds__(512) void kb_gemm3h_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW = 8;
  constexpr int G_THREADS = 512;
  constexpr int G_BN = 128;
  constexpr int G_LDT = G_BN + 4;
  constexpr int ROWS = RT * 16;
  constexpr int A_GS = ROWS * 16 + GS_PAD3;           // bytes between k-groups of an A plane
  constexpr int A_PLANE = 4 * A_GS;                  // bytes
  constexpr int STAGE = 2 * A_PLANE;                 // the weights go from L2 to registers (load_b)
  constexpr int A_SLOTS = ROWS * 4;                  // 16-byte slots (8 k of one row) per A plane and stage
  constexpr int A_IT = (A_SLOTS + G_THREADS - 1) / G_THREADS;
  constexpr int HT = (RT + 1) / 2;                   // row tiles of the upper wave half (the lower one has RT - HT)
  constexpr int SCRATCH = 2 * STAGE;                 // 512 x 16 B behind the two stages: where threads without a slot store
  constexpr int VPM = (A_IT * 44 + 6 * HT - 1) / (6 * HT) + 1;      // vector instructions placed behind each product
  char* lds = reinterpret_cast<char*>(smem);

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

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2;                        // 0: tiles [0, HT), 1: tiles [HT, RT)
  const int cgp = wave & 3;                          // 32-column group
  const int t0 = half * HT;
  const int my_nt = max(0, min(nt - t0, half ? RT - HT : HT));      // valid tiles of this wave (wave-uniform)
  const int nk = p.K >> 5;

  // one power-of-two unit per operand TENSOR: A from its largest magnitude (a_maxabs, device), the weights' from their pack
  const int eA = h2_exponent(*p.a_maxabs);
  const int eW = *(reinterpret_cast<const int*>(p.Wp) + (size_t)p.K * p.Nout);
  const float a_scale = h2_pow2(eA);
  const float unscale = h2_unscale(eA, eW);
  f32x4 acc[HT][2];
#pragma unroll
  for (int t = 0; t < HT; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging: one thread <-> one 16-byte LDS slot (8 consecutive k of one row of A / one column of the weight tile).
  // Slots are dealt so that the 64 lanes of a wave cover 16 consecutive rows x 4 k-groups with lane = 16 g + row: the
  // ds_write_b128 / ds_read_b128 lane groups of gfx950 then hit 16 distinct 4-bank slots, i.e. NO bank conflicts on either
  // side with an unpadded k-group stride (the PMC pass of the first layout showed half of the LDS-active cycles were
  // conflict cycles), and a lane's two float4 loads are 32 contiguous bytes of a row (a wave touches 16 full 128-B lines).
  // TWO register sets: the slice that is split and stored while the matrix pipe works on the current one was loaded a whole
  // half-iteration earlier, the loads issued now are for the slice after it (round 4: with one set the split ran after the
  // products of every slice, all waves at once -- the pipe idled 40 % of the loop, tools/stem_knobs.py)
  f32x4 ra[2][A_IT][2];
  uint32_t rbits[2][A_IT];
  // The weights never touch LDS: pack format 3 stores them as the matrix pipe reads them (16 B = 8 k of one column per plane
  // and k group), so a lane loads ITS operand fragments of the next slice straight from L2 into registers while the current
  // slice multiplies -- no staging stores, no LDS reads for B, and the LDS pipe (85 % busy with both operands) is left to A.
  u32x4 wb[2][2][2];         // [set][plane][column tile]
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  const bool conv = p.conv_taps > 0;
  const float* Abase = p.A + (size_t)b * (p.a_qstride ? p.a_qstride : (size_t)p.N * p.lda);
  const uint32_t* Bitbase = (AP == A_DROP) ? p.a_bits + (size_t)b * p.N * (p.lda >> 5) : nullptr;
  const int sg = lane >> 4;                 // k-group of this thread's slots (A and B)
  int a_off[A_IT];
  int a_row[A_IT];
  int a_dst[A_IT], a_buf[A_IT], a_pl[A_IT]; // LDS byte offset of the slot inside a stage, the stage stride, the plane stride -- a thread
                                            // past the last slot stores into scratch bytes behind the stages instead (stage stride 0)
  bool a_ok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int f = tid + G_THREADS * i;
    const int lrow = (f >> 6) * 16 + (lane & 15);
    const int n = row0 + lrow;
    const bool in = f < A_SLOTS;
    a_ok[i] = in && (n < row_end);
    const int nc = min(n, p.N - 1);
    a_row[i] = nc;
    const int srow = conv ? (nc / p.conv_w + 1) * p.conv_wp + (nc % p.conv_w) + 1 : nc;
    a_off[i] = srow * p.lda + sg * 8;
    a_dst[i] = in ? sg * A_GS + lrow * 16 : SCRATCH + tid * 16;      // no predicated store: the loop body stays ONE basic block
    a_buf[i] = in ? STAGE : 0;
    a_pl[i] = in ? A_PLANE : 512 * 16;
  }

  auto load_tiles = [&](auto set_c, int kt_raw) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    int kt = min(kt_raw, nk - 1);          // past the end: re-read the last slice (stored to a buffer nobody multiplies)
    int koff = kt << 5;
    if (conv) {
      const int per = p.conv_cin >> 5;
      const int tap = kt / per;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      koff = p.conv_sign * (dy * p.conv_wp + dx) * p.lda + ((kt - tap * per) << 5);
    }
    if (p.dbg & 32) koff = 0;       // timing experiments: 32 = every slice re-reads slice 0 of A (L1/L2-hot), 64 = of B
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      ra[SET][i][0] = *reinterpret_cast<const f32x4*>(Abase + a_off[i] + koff);
      ra[SET][i][1] = *reinterpret_cast<const f32x4*>(Abase + a_off[i] + koff + 4);
      if (AP == A_DROP) rbits[SET][i] = Bitbase[a_row[i] * (p.lda >> 5) + kt];
    }
  };
  auto load_b = [&](auto set_c, int kt_raw) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    int kt = min(kt_raw, nk - 1);
    if (p.dbg & 64) kt = 0;
    // pack format 3: [kt][plane][k group][Nout] x 16 B fp16; lane (i = lane & 15, g = lane >> 4) holds k = 8 g .. 8 g + 7 of column i
    const char* src = reinterpret_cast<const char*>(p.Wp) + ((((size_t)kt * 2) * 4 + sg) * p.Nout + (size_t)cb * G_BN + cgp * 32 + (lane & 15)) * 16;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int c = 0; c < 2; ++c) wb[SET][pl][c] = *reinterpret_cast<const u32x4*>(src + ((size_t)pl * 4 * p.Nout + c * 16) * 16);
  };

  auto store_tiles = [&](auto set_c, int buf) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    char* dA = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float vv = ra[SET][i][e >> 2][e & 3];
        if (AP == A_DROP) vv = ((rbits[SET][i] >> (sg * 8 + e)) & 1u) ? vv * p.a_inv_keep : 0.f;
        x[e] = a_ok[i] ? vv * a_scale : 0.f;
      }
      u32x4 s0, s1;
      h2_split8(x, s0, s1);
      char* d = lds + buf * a_buf[i] + a_dst[i];
      *reinterpret_cast<u32x4*>(d) = s0;
      *reinterpret_cast<u32x4*>(d + a_pl[i]) = s1;
    }
    (void)dA;
  };

  // lane (i = lane & 15, g = lane >> 4) holds k = 8g .. 8g+7 of row / column i for BOTH operands.  Every wave multiplies HT
  // row tiles: the lower half's last one lies past its rows when RT is odd (its accumulators are never stored) -- a branch
  // around it would cut the loop body into blocks the scheduler cannot interleave.
  auto compute = [&](int buf, auto set_c) __attribute__((always_inline)) {
    constexpr int BSET = decltype(set_c)::value;
    const char* sA = lds + buf * STAGE + (lane >> 4) * A_GS + (t0 * 16 + (lane & 15)) * 16;
    u32x4 bf[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int c = 0; c < 2; ++c) bf[pl][c] = wb[BSET][pl][c];
    // term order: smallest products first into the accumulator:  A_lo x B_hi ; A_hi x {B_lo, B_hi}
#pragma unroll
    for (int ap = 1; ap >= 0; --ap) {
      u32x4 af[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) af[t] = *reinterpret_cast<const u32x4*>(sA + ap * A_PLANE + t * 16 * 16);
#pragma unroll
      for (int bp = 1 - ap; bp >= 0; --bp) {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          acc[t][0] = mfma_f16(af[t], bf[bp][0], acc[t][0]);
          acc[t][1] = mfma_f16(af[t], bf[bp][1], acc[t][1]);
        }
      }
    }
  };
  // the matrix pipe's instructions with the split of the next slice between them (one of 6 HT products, then a share of the
  // vector work): both come from the same wave, the pipe runs the product while the wave issues the split
  auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 6 * HT; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, VPM, 0);
    }
  };

  const bool stage = !(p.dbg & 2);
  load_tiles(S0{}, 0);
  load_b(S0{}, 0);
  store_tiles(S0{}, 0);
  load_tiles(S1{}, 1);
  __syncthreads();
  if (stage) {
    for (int kt = 0; kt < nk; kt += 2) {
      // buffer 0 holds slice kt of A, weight set 0 slice kt; A set 1 holds slice kt + 1 (in flight since the previous half)
      load_tiles(S0{}, kt + 2);
      load_b(S1{}, kt + 1);
      compute(0, S0{});
      store_tiles(S1{}, 1);
      interleave();
      __syncthreads();
      if (kt + 1 >= nk) break;
      load_tiles(S1{}, kt + 3);
      load_b(S0{}, kt + 2);
      compute(1, S1{});
      store_tiles(S0{}, 0);
      interleave();
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) compute(0, S0{});          // timing experiment: the products alone
  }
  if (p.dbg & 1) {
    if (acc[0][0][0] == 123.456f) p.out[0] = acc[HT - 1][1][3];
    return;
  }

  // ---- epilogue, step 1: accumulators -> row-major LDS tile (16x16 map: col = lane & 15, row = (lane >> 4) * 4 + reg)
  float* T = smem;
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    if (t0 + t < RT) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          T[((t0 + t) * 16 + (lane >> 4) * 4 + e) * G_LDT + cgp * 32 + c * 16 + (lane & 15)] = acc[t][c][e] * unscale;
    }
  }
  __syncthreads();
  kb_epilogue_rows<RT, NW, EP, COLSUM>(p, smem, b, cb, rbi, nrb, row0, row_end);
}

template <int RT, int AP, int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm3h_launch_rt(const GemmP& p, hipStream_t st) {
  auto kern = kb_gemm3h_kernel<RT, AP, BP, EP, COLSUM>;
  constexpr size_t lds = (size_t)kb_gemm3h_lds_bytes<RT>();
  {
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);   // per (kernel, device)
    if (e != hipSuccess) return e;
  }
  const int ncb = p.Nout / 128;
  const int nrb = (p.N + RT * 16 - 1) / (RT * 16);
  GemmP q = p;
  q.dbg = kb_gemm_dbg();
  hipLaunchKernelGGL(kern, dim3(p.B * nrb * ncb), dim3(512), lds, st, q);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// The 3 x 3 convolution on the chain kernels' structure (round 4, measured first as tools/probes/conv_chain_probe.hip: 366 / 183 us
// for the stem's two convolutions at B = 64 against kb_gemm3h_kernel's 440 / 214).  Same operands, same results' meaning and
// the same epilogue code as kb_gemm3h_kernel<RT, A_PLAIN, B_PLAIN, EP, false> -- fp32 halo-padded A, one exponent per operand
// tensor, format-3 weights, kb_epilogue_rows -- but
//   * a workgroup owns 64 rows of the FLATTENED [B N] output x ALL 512 columns (wave w: columns 64 w .. 64 w + 63), so the fp16
//     split of A is done once per element instead of once per 128-column block;
//   * A goes through LDS in chunks of 256 channels of one tap (64 KB, two buffers): ONE barrier per 384 MFMAs of a wave;
//   * the weights go from L2 straight into MFMA operand registers, one 32-wide slice ahead, and the order "request slice q + 1,
//     multiply slice q" is pinned with sched_barrier (left alone the scheduler sinks the loads to their first use: 0.61 PF
//     instead of 0.97 in the probe);
//   * operands swapped (D^T = W^T A^T): a lane ends up with four consecutive columns of a row.
// Shapes: conv_taps == 9, Nout == 512, conv_cin a multiple of 256.
// ---------------------------------------------------------------------------------------------------------------
constexpr int CC_ROWS = 64, CC_NOUT = 512, CC_KC = 256, CC_KG = CC_KC / 8;
constexpr int CC_BUF = 2 * CC_KG * CC_ROWS * 16;                 // [plane][k group][row] x 16 B
constexpr int CC_LDT = 128 + 4;
constexpr int CC_EPI = 4 * CC_ROWS * CC_LDT * 4;                // four 64 x 128 epilogue tiles
constexpr int CC_LDS = 2 * CC_BUF > CC_EPI ? 2 * CC_BUF : CC_EPI;
constexpr bool conv_chain_mode() { return true; }      // (A/B against kb_gemm3h_kernel closed in round 5: profiles/r05_stem_conv_chain_ab.txt; that kernel remains the fallback by shape)

template <int EP>
__global__ __launch_bounds__(512) void kb_conv_chain_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int M = p.B * p.N;
  const int row0 = blockIdx.x * CC_ROWS;
  const int cpt = p.conv_cin / CC_KC;               // chunks per tap
  const int nchunk = 9 * cpt;
  const int nks = p.K >> 5;

  const int eA = h2_exponent(*p.a_maxabs);
  const int eW = *(reinterpret_cast<const int*>(p.Wp) + (size_t)p.K * p.Nout);
  const float a_scale = h2_pow2(eA);
  const float unscale = h2_unscale(eA, eW);

  // ---- loader: slot s = tid + 512 i (i < 4) of a chunk = (16-row tile s >> 6 & 3, k quad s >> 8, lane = 16 g + row): 8 channels of
  //      one row, fp32 in, split to both planes while it is stored
  constexpr int A_IT = CC_ROWS * CC_KG / 512;       // 4
  const float* a_src[A_IT];
  int a_dst[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int s = tid + 512 * i;
    const int blk = s >> 6;
    const int rt = blk & 3, kq = blk >> 2;
    const int row = rt * 16 + li, kg = kq * 4 + lg;
    const int m = min(row0 + row, M - 1);           // rows past the end re-read the last row (the epilogue does not store them)
    const int b = m / p.N, n = m - b * p.N;
    const int srow = (n / p.conv_w + 1) * p.conv_wp + (n % p.conv_w) + 1;
    a_src[i] = p.A + (size_t)b * p.a_qstride + (size_t)srow * p.lda + kg * 8;
    a_dst[i] = (kg * CC_ROWS + row) * 16;
  }
  f32x4 ra[A_IT][2];
  auto load_chunk = [&](int ch_raw) __attribute__((always_inline)) {
    const int ch = min(ch_raw, nchunk - 1);
    const int tap = ch / cpt, cc = ch - tap * cpt;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const int shift = p.conv_sign * (dy * p.conv_wp + dx) * p.lda + cc * CC_KC;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      ra[i][0] = *reinterpret_cast<const f32x4*>(a_src[i] + shift);
      ra[i][1] = *reinterpret_cast<const f32x4*>(a_src[i] + shift + 4);
    }
  };
  auto store_chunk = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = ra[i][e >> 2][e & 3] * a_scale;
      u32x4 s0, s1;
      h2_split8(x, s0, s1);
      char* d = lds + buf * CC_BUF + a_dst[i];
      *reinterpret_cast<u32x4*>(d) = s0;
      *reinterpret_cast<u32x4*>(d + CC_KG * CC_ROWS * 16) = s1;
    }
  };

  // ---- weights (pack format 3: [32-wide slice][plane][k group][Nout] x 16 B): lane (i = li, g = lg) holds k = 8 g .. 8 g + 7 of
  //      column i of each of the wave's four 16-column tiles, both planes
  u32x4 wb[2][2][4], aq[2][2][4];          // [set][plane][tile]
  auto load_w = [&](auto set_c, int ks_raw) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const int ks = min(ks_raw, nks - 1);
    const char* src = reinterpret_cast<const char*>(p.Wp) + ((((size_t)ks * 2) * 4 + lg) * CC_NOUT + wave * 64 + li) * 16;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int c = 0; c < 4; ++c) wb[SET][pl][c] = *reinterpret_cast<const u32x4*>(src + ((size_t)pl * 4 * CC_NOUT + c * 16) * 16);
  };
  auto load_a = [&](auto set_c, int buf, int q) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const char* sA = lds + buf * CC_BUF + ((q * 4 + lg) * CC_ROWS + li) * 16;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int t = 0; t < 4; ++t) aq[SET][pl][t] = *reinterpret_cast<const u32x4*>(sA + pl * (CC_KG * CC_ROWS * 16) + t * 256);
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mm = [&](auto set_c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[t][c] = mfma_f16(wb[SET][0][c], aq[SET][1][t], acc[t][c]);     // w_hi a_lo
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[t][c] = mfma_f16(wb[SET][1][c], aq[SET][0][t], acc[t][c]);     // w_lo a_hi
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[t][c] = mfma_f16(wb[SET][0][c], aq[SET][0][t], acc[t][c]);     // w_hi a_hi
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  load_chunk(0);
  load_w(S0{}, 0);
  store_chunk(0);
  __syncthreads();
#pragma unroll 1
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    load_chunk(ch + 1);                         // in flight under this chunk's 8 slices
    load_a(S0{}, buf, 0);
    const int ks0 = ch * (CC_KC / 32);
#pragma unroll
    for (int q = 0; q < CC_KC / 32; q += 2) {
      load_w(S1{}, ks0 + q + 1);
      load_a(S1{}, buf, q + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(S0{});
      __builtin_amdgcn_sched_barrier(0);
      load_w(S0{}, ks0 + q + 2);                // (q + 2 == 8: slice 0 of the next chunk)
      if (q + 2 < CC_KC / 32) load_a(S0{}, buf, q + 2);
      __builtin_amdgcn_sched_barrier(0);
      mm(S1{});
      __builtin_amdgcn_sched_barrier(0);
    }
    store_chunk(buf ^ 1);                       // the other buffer was last read in the previous iteration, before its barrier
    __syncthreads();
  }

  // ---- epilogue: the four 64 x 128 column blocks as row-major LDS tiles (the loop's buffers are idle), then the epilogue code of the
  //      other GEMM kernels on each, with the batch seen as ONE image of B N rows (orow = n)
  //      accumulator map (swapped operands): row = 16 t + (lane & 15), columns 16 c + 4 (lane >> 4) .. + 3 of the wave's 64
  float* T = smem + (wave >> 1) * (CC_ROWS * CC_LDT);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
      *reinterpret_cast<f32x4*>(T + (t * 16 + li) * CC_LDT + (wave & 1) * 64 + c * 16 + lg * 4) = acc[t][c] * unscale;
  __syncthreads();
  GemmP q = p;
  q.N = M; q.B = 1;
  const int row_end = min(M, row0 + CC_ROWS);
#pragma unroll 1
  for (int cb = 0; cb < 4; ++cb)
    kb_epilogue_rows<4, 8, EP, false>(q, smem + cb * (CC_ROWS * CC_LDT), 0, cb, 0, 1, row0, row_end);
}

template <int EP>
inline hipError_t kb_conv_chain_launch(const GemmP& p, hipStream_t st) {
  auto kern = kb_conv_chain_kernel<EP>;
  {
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), (size_t)CC_LDS);
    if (e != hipSuccess) return e;
  }
  GemmP q = p;
  q.dbg = kb_gemm_dbg();
  hipLaunchKernelGGL(kern, dim3((p.B * p.N + CC_ROWS - 1) / CC_ROWS), dim3(512), CC_LDS, st, q);
  return hipGetLastError();
}

template <int AP, int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm3h_launch(const GemmP& p, hipStream_t st) {
  if constexpr (AP == A_PLAIN && BP == B_PLAIN && !COLSUM && (EP == E_BIAS_ACT || EP == E_MUL_DACT)) {
    if (conv_chain_mode() && p.conv_taps == 9 && p.Nout == CC_NOUT && p.conv_cin % CC_KC == 0 && p.K == 9 * p.conv_cin)
      return kb_conv_chain_launch<EP>(p, st);
  }
  switch (kb_gemm_pick_rt(p.N, p.B, p.Nout / 128)) {
    case 1: return kb_gemm3h_launch_rt<1, AP, BP, EP, COLSUM>(p, st);
    case 2: return kb_gemm3h_launch_rt<2, AP, BP, EP, COLSUM>(p, st);
    case 4: return kb_gemm3h_launch_rt<4, AP, BP, EP, COLSUM>(p, st);
    case 7: return kb_gemm3h_launch_rt<7, AP, BP, EP, COLSUM>(p, st);
    default: return kb_gemm3h_launch_rt<13, AP, BP, EP, COLSUM>(p, st);
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The weight-gradient contraction of macx_wgrad6.hip.h (C[k][j] = sum_m A[m][k] G[m][j], both operands fp32 row-major over
// m, producer waves transpose and split in registers, consumer waves multiply) on the same three fp16 terms: two planes per
// operand, one exponent per operand tensor (TnP::a_maxabs / g_maxabs).
// ---------------------------------------------------------------------------------------------------------------
constexpr int W3_GS = 128 * 16 + 32;       // bytes between m-groups of a plane
constexpr int W3_PLANE = 4 * W3_GS;
constexpr int W3_OPER = 2 * W3_PLANE;
constexpr int W3_STAGE = 2 * W3_OPER;

// JW = j-width of the output tile in units of 128 columns.  JW = 2 (128 x 256 tiles, consumer wave tile 64 x 128) doubles
// the MFMA work per barrier and reads the A operand half as often; used whenever Jd is a multiple of 256.
template <int JW>
constexpr int w3_stage_bytes() { return W3_OPER + 2 * 4 * (JW * 128 * 16 + 32); }

template <bool CONV, int JW>
__global__ __launch_bounds__(512) void wgrad3h_kernel(TnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  constexpr int JT = JW * T_TILE;                   // output tile width
  constexpr int GG = JW * 128 * 16 + 32;            // bytes between m-groups of a G plane
  constexpr int GPL = 4 * GG;                       // G plane
  constexpr int STAGE = W3_OPER + 2 * GPL;

  const int ntj = p.Jd / JT;
  const int ntk = p.Kd / T_TILE;
  const int ntile = ntj * ntk;
  const int nblk = gridDim.x;
  int v = blockIdx.x;
  if ((nblk & 7) == 0) v = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int split = v / ntile;
  const int tile = v % ntile;
  const int tk = tile / ntj, tj = tile % ntj;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int nchunk = (m_end - m_begin + 31) >> 5;
  const int nloop = (nchunk + 2) / 3 * 3;          // whole groups of three stages; the extra ones multiply zeros
  // one power-of-two unit per operand tensor (macx_gemm3h.hip.h): from the largest magnitudes an absmax pass left on the device
  const int eA = h2_exponent(*p.a_maxabs), eG = h2_exponent(*p.g_maxabs);
  const float a_scale = h2_pow2(eA), g_scale = h2_pow2(eG);
  const float unscale = h2_unscale(eA, eG);

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  typedef float gvec __attribute__((ext_vector_type(2 * JW)));     // a lane's columns of one G row

  if (wave < 4) {
    // ================= producer waves: global -> registers -> fp16 hi/lo split -> LDS planes =================
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
    const float* baseG = p.G + (size_t)blockIdx.y * p.zG + tj * JT + 2 * JW * lane;
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
      char* dA = lds + (ch & 1) * STAGE + mg * W3_GS + (2 * lane) * 16;
      char* dG = lds + (ch & 1) * STAGE + W3_OPER + mg * GG + (2 * JW * lane) * 16;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = (mrow + r < m_end) ? ra[SL][r][c] * a_scale : 0.f;
        u32x4 s0, s1;
        h2_split8(x, s0, s1);
        *reinterpret_cast<u32x4*>(dA + c * 16) = s0;
        *reinterpret_cast<u32x4*>(dA + W3_PLANE + c * 16) = s1;
      }
#pragma unroll
      for (int c = 0; c < 2 * JW; ++c) {
        float x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = (mrow + r < m_end) ? rg[SL][r][c] * g_scale : 0.f;
        u32x4 s0, s1;
        h2_split8(x, s0, s1);
        *reinterpret_cast<u32x4*>(dG + c * 16) = s0;
        *reinterpret_cast<u32x4*>(dG + GPL + c * 16) = s1;
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
      const char* sa = lds + buf * STAGE + (lane >> 4) * W3_GS + (wr * 64 + (lane & 15)) * 16;
      const char* sg = lds + buf * STAGE + W3_OPER + (lane >> 4) * GG + (wc * 64 * JW + (lane & 15)) * 16;
      u32x4 af[2][4];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int t = 0; t < 4; ++t) af[pl][t] = *reinterpret_cast<const u32x4*>(sa + pl * W3_PLANE + t * 256);
      // G plane by plane (smallest terms first): G_lo x A_hi ; G_hi x {A_lo, A_hi}
#pragma unroll
      for (int bp = 1; bp >= 0; --bp) {
        u32x4 gf[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) gf[c] = *reinterpret_cast<const u32x4*>(sg + bp * GPL + c * 256);
#pragma unroll
        for (int ap = 1 - bp; ap >= 0; --ap)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[t][c] = mfma_f16(af[ap][t], gf[c], acc[t][c]);
      }
    };
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < nloop; ++ch) {
      compute(ch & 1);
      __syncthreads();
    }
    // 16x16 accumulator map: col = lane & 15 (j), row = (lane >> 4) * 4 + reg (k)
    float* out = p.part + (size_t)blockIdx.y * p.zpart + (size_t)split * p.Kd * p.Jd;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = tk * T_TILE + wr * 64 + t * 16 + (lane >> 4) * 4 + e;
          const int j = tj * JT + wc * 64 * JW + c * 16 + (lane & 15);
          out[(size_t)k * p.Jd + j] = acc[t][c][e] * unscale;
        }
  }
}

inline int wgrad3h_jw(int Jd) { return (Jd % 256 == 0) ? 2 : 1; }

template <bool CONV, int JW>
inline hipError_t wgrad3h_launch_t(const TnP& p, hipStream_t st) {
  auto kern = wgrad3h_kernel<CONV, JW>;
  constexpr size_t lds = 2 * w3_stage_bytes<JW>();
  {
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);   // per (kernel, device)
    if (e != hipSuccess) return e;
  }
  const int grid = (p.Kd / T_TILE) * (p.Jd / (JW * T_TILE)) * p.nsplit;
  hipLaunchKernelGGL(kern, dim3(grid, p.nz > 1 ? p.nz : 1), dim3(512), lds, st, p);
  return hipGetLastError();
}

inline hipError_t wgrad3h_launch(const TnP& p, hipStream_t st) {
  if (wgrad3h_jw(p.Jd) == 2) return p.conv_taps ? wgrad3h_launch_t<true, 2>(p, st) : wgrad3h_launch_t<false, 2>(p, st);
  return p.conv_taps ? wgrad3h_launch_t<true, 1>(p, st) : wgrad3h_launch_t<false, 1>(p, st);
}

}  // namespace macx
