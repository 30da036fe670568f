// macx_gemm_tn.hip.h -- weight-gradient contractions of the read unit:  C[k][j] = sum_m A[m][k] * G[m][j]
// reduced over the B*N knowledge-base rows (dW2 = H1^T dI2, dWx = KBd^T dX,
// dW1 = [X*y, X]^T dI1).  Both operands are row-major over m, which is already the MFMA-friendly
// layout for a TN product (fragments read along the contiguous dimension, conflict-free b32).
//
// Determinism: no float atomics.  Every workgroup owns one (split, 128x128 tile) slab of a partial
// buffer; a fixed-order reduction kernel sums the slabs.
#pragma once
#include "macx_common.hip.h"
#include "macx_gemm.hip.h"

namespace macx {

constexpr int T_BM = 32;            // reduction rows per LDS stage
constexpr int T_TILE = 128;         // output tile edge
constexpr int T_STAGE = T_BM * T_TILE;
constexpr int T_RING = 4;           // LDS stages of the DMA ring (A and G each): 4 x 32 KB = 128 KB

struct TnP {
  int M;                 // reduction rows (B*N)
  int Kd, Jd;            // output dims (multiples of 128)
  int nsplit;
  int rows_per_split;    // multiple of 2
  const float* A; int lda;
  int a_mod;             // A row of reduction row m is (m % a_mod): the same KB under p different masks
  const float* G; int ldg;
  float* part;           // [nsplit][Kd][Jd]
  // kernel gradient of a 3x3 convolution over a halo-padded NHWC input (stem CNN): reduction row
  // m = (image, pixel n) reads padded input row  image*conv_np + (y+1)*conv_wp + (x+1) + tap offset;
  // output row k = tap * conv_cin + channel  (HWIO order).  conv_taps == 0: plain contraction.
  int conv_taps, conv_w, conv_wp, conv_cin, conv_n, conv_np;
  uint32_t magic_n, magic_w;   // ceil(2^32 / conv_n), ceil(2^32 / conv_w)
  int nz; size_t zG, zpart;   // nz > 1: blockIdx.y selects one of nz independent contractions sharing A (G += z*zG, part += z*zpart); split kernel only
  int dbg;               // timing experiments (macx_opts.tune[MACX_TUNE_PHASE_MASK]): 16 no MFMAs, 32 no in-loop loads, 64 no in-loop split/store
  const float* a_maxabs; const float* g_maxabs;   // wgrad3h_kernel: largest magnitude of A / of G (device floats)
};

// 8 waves: waves 0-3 and 4-7 each cover the 128x128 tile as 2x2 sub-tiles of 64x64 and take
// alternate k-steps (row pairs) of every stage, so each SIMD holds two waves; the two half-sums are
// combined through LDS in a fixed order.  Both operands stream HBM/L2 -> LDS by DMA through a
// 4-stage ring (two stages of latency budget, no staging registers, no ds_write).
template <int AP>
__global__ __launch_bounds__(512) void wgrad_tn_kernel(TnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                       // [T_RING][T_STAGE]
  float* sG = smem + T_RING * T_STAGE;    // [T_RING][T_STAGE]

  const int ntj = p.Jd / T_TILE;
  const int ntk = p.Kd / T_TILE;
  const int ntile = ntj * ntk;
  // tiles of one split are neighbours in the virtual order -> same XCD after the remap, so the
  // 16 tiles that re-read the same A/G rows share one L2.
  const int nblk = gridDim.x;
  int v = blockIdx.x;
  if ((nblk & 7) == 0) v = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int split = v / ntile;
  const int tile = v % ntile;
  const int tk = tile / ntj, tj = tile % ntj;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int nchunk = (m_end - m_begin + T_BM - 1) / T_BM;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;

  int conv_shift = 0, a_col0 = tk * T_TILE;
  if (p.conv_taps) {
    const int per = p.conv_cin / T_TILE;
    const int tap = tk / per;
    conv_shift = (tap / 3 - 1) * p.conv_wp + (tap - (tap / 3) * 3 - 1);
    a_col0 = (tk - tap * per) * T_TILE;
  }
  // DMA slots: a stage holds 32 rows x 32 float4 per operand = 1024 slots; 512 threads -> 2 + 2 per stage
  auto dma_stage = [&](int buf, int ch) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f0 = wave * 64 + 512 * i;
      const int f = f0 + lane;
      const int m = min(m_begin + ch * T_BM + (f >> 5), p.M - 1);   // rows past the end are clamped, never multiplied
      const int c4 = (f & 31) * 4;
      size_t arow;
      if (p.conv_taps) {
        const int img = (int)__umulhi((uint32_t)m, p.magic_n);
        const int n = m - img * p.conv_n;
        const int yy = (int)__umulhi((uint32_t)n, p.magic_w);
        arow = (size_t)img * p.conv_np + (yy + 1) * p.conv_wp + (n - yy * p.conv_w) + 1 + conv_shift;
      } else {
        arow = (size_t)(m % p.a_mod);
      }
      dma16(p.A + arow * p.lda + a_col0 + c4, sA + buf * T_STAGE + f0 * 4);
      dma16(p.G + (size_t)m * p.ldg + tj * T_TILE + c4, sG + buf * T_STAGE + f0 * 4);
    }
  };
  auto compute = [&](int buf, int steps) {
    const float* a = sA + buf * T_STAGE + (lane >> 5) * T_TILE + wr * 64 + (lane & 31);
    const float* g = sG + buf * T_STAGE + (lane >> 5) * T_TILE + wc * 64 + (lane & 31);
#pragma unroll 4
    for (int s = grp; s < steps; s += 2) {
      const float a0 = a[s * 2 * T_TILE], a1 = a[s * 2 * T_TILE + 32];
      const float g0 = g[s * 2 * T_TILE], g1 = g[s * 2 * T_TILE + 32];
      acc[0][0] = mfma32(a0, g0, acc[0][0]);
      acc[0][1] = mfma32(a0, g1, acc[0][1]);
      acc[1][0] = mfma32(a1, g0, acc[1][0]);
      acc[1][1] = mfma32(a1, g1, acc[1][1]);
    }
  };

  // prologue: stages 0..2 in flight, stage 0 awaited
  for (int s = 0; s < T_RING - 1 && s < nchunk; ++s) dma_stage(s, s);
  if (nchunk >= 3) wait_vmcnt<8>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
#pragma unroll 1
  for (int ch = 0; ch < nchunk; ++ch) {
    const int cur = ch % T_RING;
    // stage (ch-1) % RING was fully read before the barrier that ended iteration ch-1: refill it
    if (ch + T_RING - 1 < nchunk) dma_stage((ch + T_RING - 1) % T_RING, ch + T_RING - 1);
    const int rows = min(T_BM, m_end - (m_begin + ch * T_BM));
    if (rows & 1) {
      // odd tail: the MFMA consumes rows in pairs -> clear the row after the last one
      for (int c = tid; c < T_TILE; c += 512) { sA[cur * T_STAGE + rows * T_TILE + c] = 0.f; sG[cur * T_STAGE + rows * T_TILE + c] = 0.f; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    compute(cur, (rows + 1) >> 1);
    // stage ch+1 must have landed before anybody reads it; up to two younger stages may still fly
    const int ahead = min(nchunk - 1, ch + T_RING - 1) - (ch + 1);   // stages issued after stage ch+1
    if (ahead >= 2) wait_vmcnt<8>(); else if (ahead == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }

  // combine the two k-step groups (fixed order: group 0 + group 1), then store
  float* red = smem;   // [4 waves][4 tiles][16 regs][64 lanes]
  if (grp == 1) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[(((wave & 3) * 4 + a * 2 + c) * 16 + e) * 64 + lane] = acc[a][c][e];
  }
  __syncthreads();
  if (grp == 0) {
    float* out = p.part + (size_t)split * p.Kd * p.Jd;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int k = tk * T_TILE + wr * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const int j = tj * T_TILE + wc * 64 + c * 32 + (lane & 31);
          out[(size_t)k * p.Jd + j] = acc[a][c][e] + red[(((wave & 3) * 4 + a * 2 + c) * 16 + e) * 64 + lane];
        }
  }
}

template <int AP>
inline hipError_t wgrad_tn_launch(const TnP& p, hipStream_t st) {
  auto kern = wgrad_tn_kernel<AP>;
  constexpr size_t lds = 2 * T_RING * T_STAGE * sizeof(float);
  {
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(kern), lds);   // per (kernel, device)
    if (e != hipSuccess) return e;
  }
  const int grid = (p.Kd / T_TILE) * (p.Jd / T_TILE) * p.nsplit;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Per-question interaction gradient  S_b = X_b^T dI1_b  (one [d x d] matrix per question), consumed
// in registers for the three quantities the folded weight  W_eff(b) = diag(y_b) W1a + W1b  needs:
//     dW1a += diag(y_b) S_b        dW1b += S_b        dy[b][k] = sum_j W1a[k][j] S_b[k][j]
// (SURVEY Appendix A "W1" + "mem-mul" rows, rewritten through S_b so that neither the [B,N,2d]
// concat nor dP is ever formed.)  A workgroup owns one 128x128 tile and a group of questions.
// ---------------------------------------------------------------------------------------------
struct SbP {
  int B, N, d;
  int qpg;                 // questions per group
  const float* X;          // [B][N][d]
  const float* dI1;        // [B][N][d]
  const float* y;          // [B][d]
  const float* W1a;        // [d][d] row-major (k, j)
  float* dW1a_part;        // [ngroup][d][d]
  float* dW1b_part;        // [ngroup][d][d]
  float* dy_part;          // [2*d/128][B][d]
};

__global__ __launch_bounds__(256) void sb_wgrad_kernel(SbP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;
  float* sG = smem + 2 * T_STAGE;

  const int nt = p.d / T_TILE;
  const int ntile = nt * nt;
  const int nblk = gridDim.x;
  int v = blockIdx.x;
  if ((nblk & 7) == 0) v = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int group = v / ntile;
  const int tile = v % ntile;
  const int tk = tile / nt, tj = tile % nt;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int hh = lane >> 5;

  f32x16 accS[2][2], accA[2][2], accB[2][2], w1a[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        accA[a][c][e] = 0.f;
        accB[a][c][e] = 0.f;
        const int k = tk * T_TILE + wr * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        const int j = tj * T_TILE + wc * 64 + c * 32 + (lane & 31);
        w1a[a][c][e] = p.W1a[(size_t)k * p.d + j];
      }

  const int nchunk = (p.N + T_BM - 1) / T_BM;
  const int b_begin = group * p.qpg;
  const int b_end = min(p.B, b_begin + p.qpg);

  f32x4 ra[4], rg[4];
  for (int b = b_begin; b < b_end; ++b) {
    const float* Xb = p.X + (size_t)b * p.N * p.d;
    const float* Gb = p.dI1 + (size_t)b * p.N * p.d;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) accS[a][c][e] = 0.f;

    auto load_stage = [&](int ch) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        const int n = ch * T_BM + (f >> 5);
        const int c4 = (f & 31) * 4;
        if (n < p.N) {
          ra[i] = *reinterpret_cast<const f32x4*>(Xb + (size_t)n * p.d + tk * T_TILE + c4);
          rg[i] = *reinterpret_cast<const f32x4*>(Gb + (size_t)n * p.d + tj * T_TILE + c4);
        } else {
          ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          rg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        *reinterpret_cast<f32x4*>(sA + buf * T_STAGE + f * 4) = ra[i];
        *reinterpret_cast<f32x4*>(sG + buf * T_STAGE + f * 4) = rg[i];
      }
    };

    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
      const int cur = ch & 1;
      if (ch + 1 < nchunk) load_stage(ch + 1);
      const int rows = min(T_BM, p.N - ch * T_BM);
      const int steps = (rows + 1) >> 1;
      const float* a = sA + cur * T_STAGE + hh * T_TILE + wr * 64 + (lane & 31);
      const float* g = sG + cur * T_STAGE + hh * T_TILE + wc * 64 + (lane & 31);
#pragma unroll 4
      for (int s = 0; s < steps; ++s) {
        const float a0 = a[s * 2 * T_TILE], a1 = a[s * 2 * T_TILE + 32];
        const float g0 = g[s * 2 * T_TILE], g1 = g[s * 2 * T_TILE + 32];
        accS[0][0] = mfma32(a0, g0, accS[0][0]);
        accS[0][1] = mfma32(a0, g1, accS[0][1]);
        accS[1][0] = mfma32(a1, g0, accS[1][0]);
        accS[1][1] = mfma32(a1, g1, accS[1][1]);
      }
      if (ch + 1 < nchunk) store_stage(cur ^ 1);
      __syncthreads();
    }

    // consume S_b
    const float* yb = p.y + (size_t)b * p.d;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = tk * T_TILE + wr * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        const float yk = yb[k];
        const float s0 = accS[a][0][e], s1 = accS[a][1][e];
        accB[a][0][e] += s0;
        accB[a][1][e] += s1;
        accA[a][0][e] = fmaf(yk, s0, accA[a][0][e]);
        accA[a][1][e] = fmaf(yk, s1, accA[a][1][e]);
        float dyp = half_sum(w1a[a][0][e] * s0 + w1a[a][1][e] * s1);
        if ((lane & 31) == 0) p.dy_part[((size_t)(tj * 2 + wc) * p.B + b) * p.d + k] = dyp;
      }
    }
  }

  float* oa = p.dW1a_part + (size_t)group * p.d * p.d;
  float* ob = p.dW1b_part + (size_t)group * p.d * p.d;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = tk * T_TILE + wr * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        const int j = tj * T_TILE + wc * 64 + c * 32 + (lane & 31);
        oa[(size_t)k * p.d + j] = accA[a][c][e];
        ob[(size_t)k * p.d + j] = accB[a][c][e];
      }
}

inline hipError_t sb_wgrad_launch(const SbP& p, hipStream_t st) {
  constexpr size_t lds = 4 * T_STAGE * sizeof(float);
  {
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(sb_wgrad_kernel), lds);   // per (kernel, device)
    if (e != hipSuccess) return e;
  }
  const int nt = p.d / T_TILE;
  const int ngroup = (p.B + p.qpg - 1) / p.qpg;
  hipLaunchKernelGGL(sb_wgrad_kernel, dim3(nt * nt * ngroup), dim3(256), lds, st, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// fixed-order reduction of partial slabs:  dst[i] (+)= sum_s part[s][i]
// ---------------------------------------------------------------------------------------------
__global__ void slab_reduce_kernel(const float* __restrict__ part, int nslab, size_t n4, float* dst, int accumulate) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    // eight slabs in flight per thread; the summation order (four interleaved partial sums, then a fixed tree) does not
    // depend on anything but nslab, so runs stay bit-identical
    const f32x4* src = reinterpret_cast<const f32x4*>(part) + i;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int k = 0;
    for (; k + 8 <= nslab; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * n4];
      a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
      a0 += v[4]; a1 += v[5]; a2 += v[6]; a3 += v[7];
    }
    for (; k < nslab; ++k) a0 += src[(size_t)k * n4];
    f32x4 s = (a0 + a1) + (a2 + a3);
    if (accumulate) s += reinterpret_cast<const f32x4*>(dst)[i];
    reinterpret_cast<f32x4*>(dst)[i] = s;
  }
}

// several reductions in one launch (blockIdx.y selects one): the read unit's four weight gradients end this way
struct SlabDesc { const float* part; int nslab; size_t n4; float* dst; int accumulate; };
struct SlabList { SlabDesc d[8]; };
__global__ void slab_reduce_list_kernel(SlabList L) {
  const SlabDesc q = L.d[blockIdx.y];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q.n4; i += stride) {
    // (the order of slab_reduce_kernel)
    const f32x4* src = reinterpret_cast<const f32x4*>(q.part) + i;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int k = 0;
    for (; k + 8 <= q.nslab; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * q.n4];
      a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
      a0 += v[4]; a1 += v[5]; a2 += v[6]; a3 += v[7];
    }
    for (; k < q.nslab; ++k) a0 += src[(size_t)k * q.n4];
    f32x4 s = (a0 + a1) + (a2 + a3);
    if (q.accumulate) s += reinterpret_cast<const f32x4*>(q.dst)[i];
    reinterpret_cast<f32x4*>(q.dst)[i] = s;
  }
}
inline hipError_t slab_reduce_list_launch(const SlabList& L, int n, size_t max_n, hipStream_t st) {
  int grid = (int)((max_n / 4 + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(slab_reduce_list_kernel, dim3(grid, n), dim3(256), 0, st, L);
  return hipGetLastError();
}

inline hipError_t slab_reduce_launch(const float* part, int nslab, size_t n, float* dst, int accumulate, hipStream_t st) {
  const size_t n4 = n / 4;
  int grid = (int)((n4 + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid), dim3(256), 0, st, part, nslab, n4, dst, accumulate);
  return hipGetLastError();
}

}  // namespace macx
