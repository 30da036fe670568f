// macx_gemm6.hip.h -- the knowledge-base GEMM family on the bf16 matrix pipe with fp32-equivalent numerics.
//
// gfx950 runs f32-input MFMA at 1/16 of the bf16 rate (157 TF vs 2.5 PF) and has no TF32/xf32 form.  Every fp32
// operand x is therefore split EXACTLY into three bf16 pieces  x = x1 + x2 + x3  (round-to-nearest residual
// chain: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2); 3 x 8 significant bits cover fp32's 24), and a
// product is evaluated as the six leading cross terms
//     a*b ~= a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1          (dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-23 |ab|)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: each bf16 x bf16 product is exact in fp32, the accumulator is
// the same fp32 accumulator the f32 MFMA uses, so the result carries fp32-class error (measured against fp64 in
// tests/test_gpu_units.py next to the native f32 kernel) at 6/16 of the f32 MFMA issue time.
//
// Tiling is the same per-question tiling as macx_gemm.hip.h (RT*16 rows of one question x 128 columns, 8 waves), with
// the waves arranged 2 (row halves) x 4 (32-column groups) so that an A fragment read from LDS feeds two MFMAs
// (three bf16 planes per operand make LDS fragment traffic, not the matrix pipe, the next limit otherwise).
//   LDS stage:  A planes [3][4 k-groups][ROWS] x 16 B (8 bf16 of one row); B planes [3][4 k-groups][128 cols] x 16 B.
//               A fragment = lane (i, g) reads slot [g][row0 + i]: the 16 lanes of a k-group read 256 contiguous bytes
//               (conflict-free b128); the k-group stride is padded by 32 B so the staging writes spread over all banks.
//   weights:    B_PLAIN   -> pre-split planes  Wb[K/32][3][Nout][32] bf16      (pack format 1)
//               B_YMIX_*  -> fp32 k-major tiles Wt[K/32][Nout][32]             (pack format 2), mixed with the
//                            per-question vector in fp32, then split while staging
// The epilogues are the shared kb_epilogue_rows of macx_gemm.hip.h.
#pragma once
#include "macx_gemm.hip.h"

namespace macx {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32 (RNE): low half = a
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// exact 3-way split of four consecutive-k fp32 values into three packed-bf16 pairs-of-pairs (8 B per plane)
__device__ __forceinline__ void split4(const f32x4 x, u32x2& p0, u32x2& p1, u32x2& p2) {
  float r[4], q[4];
  p0[0] = pk_bf16(x[0], x[1]);
  p0[1] = pk_bf16(x[2], x[3]);
  r[0] = x[0] - __uint_as_float(p0[0] << 16);
  r[1] = x[1] - __uint_as_float(p0[0] & 0xFFFF0000u);
  r[2] = x[2] - __uint_as_float(p0[1] << 16);
  r[3] = x[3] - __uint_as_float(p0[1] & 0xFFFF0000u);
  p1[0] = pk_bf16(r[0], r[1]);
  p1[1] = pk_bf16(r[2], r[3]);
  q[0] = r[0] - __uint_as_float(p1[0] << 16);
  q[1] = r[1] - __uint_as_float(p1[0] & 0xFFFF0000u);
  q[2] = r[2] - __uint_as_float(p1[1] << 16);
  q[3] = r[3] - __uint_as_float(p1[1] & 0xFFFF0000u);
  p2[0] = pk_bf16(q[0], q[1]);
  p2[1] = pk_bf16(q[2], q[3]);
}
// exact 3-way split of 8 fp32 values (8 consecutive k of one row / column) into three 16-byte bf16x8 slots
__device__ __forceinline__ void split8(const float* x, u32x4& p0, u32x4& p1, u32x4& p2) {
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const float a = x[2 * h], b = x[2 * h + 1];
    p0[h] = pk_bf16(a, b);
    const float ra = a - __uint_as_float(p0[h] << 16), rb = b - __uint_as_float(p0[h] & 0xFFFF0000u);
    p1[h] = pk_bf16(ra, rb);
    p2[h] = pk_bf16(ra - __uint_as_float(p1[h] << 16), rb - __uint_as_float(p1[h] & 0xFFFF0000u));
  }
}

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4 a, const u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// weight formats for this kernel (macx_small.hip.h pack_weights_kernel, PackDesc::fmt)
//   1: Wb[kt][plane][n][32] bf16 = split planes of W[k][n];  2: Wt[kt][n][32] fp32 k-major tiles
constexpr size_t gemm6_plane_floats(size_t K, size_t Nout) { return K * Nout * 3 / 2; }   // format 1 size in floats

constexpr int GS_PAD = 0;       // padding of the k-group stride (bytes): 0 keeps the b128 fragment reads conflict-free

template <int RT>
constexpr int kb_gemm6_lds_bytes() {
  constexpr int ROWS = RT * 16;
  constexpr int stage = 3 * 4 * (ROWS * 16 + GS_PAD) + 3 * 4 * (128 * 16 + GS_PAD);
  constexpr int epi = (ROWS * (128 + 4) + 256 * 8) * 4;
  return 2 * stage > epi ? 2 * stage : epi;
}

template <int RT, int AP, int BP, int EP, bool COLSUM>
__global__ __launch_bounds__(512) void kb_gemm6_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW = 8;
  constexpr int G_THREADS = 512;
  constexpr int G_BN = 128;
  constexpr int G_LDT = G_BN + 4;
  constexpr int ROWS = RT * 16;
  constexpr int A_GS = ROWS * 16 + GS_PAD;           // bytes between k-groups of an A plane
  constexpr int B_GS = G_BN * 16 + GS_PAD;
  constexpr int A_PLANE = 4 * A_GS;                  // bytes
  constexpr int B_PLANE = 4 * B_GS;
  constexpr int STAGE = 3 * A_PLANE + 3 * B_PLANE;
  constexpr int A_SLOTS = ROWS * 4;                  // 16-byte slots (8 k of one row) per A plane and stage
  constexpr int A_IT = (A_SLOTS + G_THREADS - 1) / G_THREADS;
  constexpr int HT = (RT + 1) / 2;                   // row tiles of the upper wave half (the lower one has RT - HT)
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

  f32x4 acc[HT][2];
#pragma unroll
  for (int t = 0; t < HT; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging: one thread <-> one 16-byte LDS slot (8 consecutive k of one row of A / one column of the weight tile).
  // Slots are dealt so that the 64 lanes of a wave cover 16 consecutive rows x 4 k-groups with lane = 16 g + row: the
  // ds_write_b128 / ds_read_b128 lane groups of gfx950 then hit 16 distinct 4-bank slots, i.e. NO bank conflicts on either
  // side with an unpadded k-group stride (the PMC pass of the first layout showed half of the LDS-active cycles were
  // conflict cycles), and a lane's two float4 loads are 32 contiguous bytes of a row (a wave touches 16 full 128-B lines).
  f32x4 ra[A_IT][2];
  uint32_t rbits[A_IT];
  u32x4 rb[3];               // B_PLAIN: raw plane bytes
  f32x4 rw[2], rw2[2];       // B_YMIX: 8 consecutive k of W1a / W1b for this thread's column
  f32x4 ry[2];               // B_YMIX_ROW: the matching 8 entries of y_b

  const bool conv = p.conv_taps > 0;
  const float* Abase = p.A + (size_t)b * (p.a_qstride ? p.a_qstride : (size_t)p.N * p.lda);
  const uint32_t* Bitbase = (AP == A_DROP) ? p.a_bits + (size_t)b * p.N * (p.lda >> 5) : nullptr;
  const int sg = lane >> 4;                 // k-group of this thread's slots (A and B)
  int a_off[A_IT];
  int a_row[A_IT];
  int a_lrow[A_IT];
  bool a_ok[A_IT], a_in[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int f = tid + G_THREADS * i;
    const int lrow = (f >> 6) * 16 + (lane & 15);
    const int n = row0 + lrow;
    a_lrow[i] = lrow;
    a_in[i] = f < A_SLOTS;
    a_ok[i] = a_in[i] && (n < row_end);
    const int nc = min(n, p.N - 1);
    a_row[i] = nc;
    const int srow = conv ? (nc / p.conv_w + 1) * p.conv_wp + (nc % p.conv_w) + 1 : nc;
    a_off[i] = srow * p.lda + sg * 8;
  }
  const int bcol = (lane & 15) + 16 * (tid >> 6);      // this thread's column of the 128-column weight tile
  float ycol = 0.f;
  if (BP == B_YMIX_COL) ycol = p.y[(size_t)b * p.ldy + cb * G_BN + bcol];

  auto load_tiles = [&](int kt) {
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
      ra[i][0] = *reinterpret_cast<const f32x4*>(Abase + a_off[i] + koff);
      ra[i][1] = *reinterpret_cast<const f32x4*>(Abase + a_off[i] + koff + 4);
      if (AP == A_DROP) rbits[i] = Bitbase[a_row[i] * (p.lda >> 5) + kt];
    }
    if (p.dbg & 64) kt = 0;
    if (BP == B_PLAIN) {
      // plane pl of this slice and column block: [128 cols][32 k] bf16; the thread copies its slot (column bcol, k-group sg)
      const char* src = reinterpret_cast<const char*>(p.Wp) + ((size_t)kt * 3 * p.Nout + (size_t)cb * G_BN + bcol) * 64 + sg * 16;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) rb[pl] = *reinterpret_cast<const u32x4*>(src + (size_t)pl * p.Nout * 64);
    } else {
      const size_t off = ((size_t)kt * p.Nout + cb * G_BN + bcol) * 32 + sg * 8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        rw[h] = *reinterpret_cast<const f32x4*>(p.Wp + off + 4 * h);
        rw2[h] = *reinterpret_cast<const f32x4*>(p.Wp2 + off + 4 * h);
        if (BP == B_YMIX_ROW) ry[h] = *reinterpret_cast<const f32x4*>(p.y + (size_t)b * p.ldy + (kt << 5) + sg * 8 + 4 * h);
      }
    }
  };

  auto store_tiles = [&](int buf, int kt) {
    char* dA = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float vv = ra[i][e >> 2][e & 3];
        if (AP == A_DROP) vv = ((rbits[i] >> (sg * 8 + e)) & 1u) ? vv * p.a_inv_keep : 0.f;
        x[e] = a_ok[i] ? vv : 0.f;
      }
      u32x4 s0, s1, s2;
      split8(x, s0, s1, s2);
      if (a_in[i]) {
        char* d = dA + sg * A_GS + a_lrow[i] * 16;
        *reinterpret_cast<u32x4*>(d) = s0;
        *reinterpret_cast<u32x4*>(d + A_PLANE) = s1;
        *reinterpret_cast<u32x4*>(d + 2 * A_PLANE) = s2;
      }
    }
    char* dB = dA + 3 * A_PLANE + sg * B_GS + bcol * 16;
    if (BP == B_PLAIN) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dB + pl * B_PLANE) = rb[pl];
    } else {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // ROW: B_eff[k][j] = y[b][k] * W1a[k][j] + W1b[k][j]   (ops.py:703,718 folded into the weights)
        // COL: B_eff[k][j] = y[b][j] * W1a^T[k][j] + W1b^T[k][j]   (backward-data of the same product)
        const float yy = (BP == B_YMIX_ROW) ? ry[e >> 2][e & 3] : ycol;
        x[e] = fmaf(rw[e >> 2][e & 3], yy, rw2[e >> 2][e & 3]);
      }
      u32x4 s0, s1, s2;
      split8(x, s0, s1, s2);
      *reinterpret_cast<u32x4*>(dB) = s0;
      *reinterpret_cast<u32x4*>(dB + B_PLANE) = s1;
      *reinterpret_cast<u32x4*>(dB + 2 * B_PLANE) = s2;
    }
  };

  // lane (i = lane & 15, g = lane >> 4) holds k = 8g .. 8g+7 of row / column i for BOTH operands
  auto compute = [&](int buf) {
    const char* sA = lds + buf * STAGE + (lane >> 4) * A_GS + (t0 * 16 + (lane & 15)) * 16;
    const char* sB = lds + buf * STAGE + 3 * A_PLANE + (lane >> 4) * B_GS + (cgp * 32 + (lane & 15)) * 16;
    u32x4 bf[3][2];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int c = 0; c < 2; ++c) bf[pl][c] = *reinterpret_cast<const u32x4*>(sB + pl * B_PLANE + c * 16 * 16);
    // term order: smallest products first into the accumulator
    //   A plane 2 x B0 ; A plane 1 x {B1, B0} ; A plane 0 x {B2, B1, B0}
#pragma unroll
    for (int ap = 2; ap >= 0; --ap) {
      u32x4 af[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t)
        if (t < HT - 1 || t < my_nt) af[t] = *reinterpret_cast<const u32x4*>(sA + ap * A_PLANE + t * 16 * 16);
#pragma unroll
      for (int bp = 2 - ap; bp >= 0; --bp) {
#pragma unroll
        for (int t = 0; t < HT - 1; ++t) {
          acc[t][0] = mfma_bf16(af[t], bf[bp][0], acc[t][0]);
          acc[t][1] = mfma_bf16(af[t], bf[bp][1], acc[t][1]);
        }
        if (HT - 1 < my_nt) {
          acc[HT - 1][0] = mfma_bf16(af[HT - 1], bf[bp][0], acc[HT - 1][0]);
          acc[HT - 1][1] = mfma_bf16(af[HT - 1], bf[bp][1], acc[HT - 1][1]);
        }
      }
    }
  };

  load_tiles(0);
  store_tiles(0, 0);
  __syncthreads();
  const bool stage = !(p.dbg & 2);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = stage ? (kt & 1) : 0;
    if (stage && kt + 1 < nk) load_tiles(kt + 1);
    compute(cur);
    if (stage && kt + 1 < nk) store_tiles(cur ^ 1, kt + 1);
    __syncthreads();
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
          T[((t0 + t) * 16 + (lane >> 4) * 4 + e) * G_LDT + cgp * 32 + c * 16 + (lane & 15)] = acc[t][c][e];
    }
  }
  __syncthreads();
  kb_epilogue_rows<RT, NW, EP, COLSUM>(p, smem, b, cb, rbi, nrb, row0, row_end);
}

template <int RT, int AP, int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm6_launch_rt(const GemmP& p, hipStream_t st) {
  auto kern = kb_gemm6_kernel<RT, AP, BP, EP, COLSUM>;
  constexpr size_t lds = (size_t)kb_gemm6_lds_bytes<RT>();
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

template <int AP, int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm6_launch(const GemmP& p, hipStream_t st) {
  switch (kb_gemm_pick_rt(p.N, p.B, p.Nout / 128)) {
    case 1: return kb_gemm6_launch_rt<1, AP, BP, EP, COLSUM>(p, st);
    case 2: return kb_gemm6_launch_rt<2, AP, BP, EP, COLSUM>(p, st);
    case 4: return kb_gemm6_launch_rt<4, AP, BP, EP, COLSUM>(p, st);
    case 7: return kb_gemm6_launch_rt<7, AP, BP, EP, COLSUM>(p, st);
    default: return kb_gemm6_launch_rt<13, AP, BP, EP, COLSUM>(p, st);
  }
}

}  // namespace macx
