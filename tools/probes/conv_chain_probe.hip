// conv_chain_probe.hip -- PROTOTYPE (not part of the library): the stem's 3 x 3 convolution as an implicit GEMM on the chain
// kernels' structure instead of kb_gemm3h_kernel's: what rate does the loop reach, before anything is
// integrated?
//   * a workgroup owns 64 output rows (positions of the flattened [B][H][W] grid) x ALL 512 output channels; wave w its 64 columns
//   * the A operand (halo-padded NHWC input, already split x 2^e = hi + lo into two fp16 planes with ONE exponent for the tensor,
//     the 3h convention of macx_gemm3h.hip.h) goes through LDS in chunks of 256 channels of one tap: 64 KB per chunk, two buffers,
//     ONE barrier per chunk = per 384 MFMAs of a wave (kb_gemm3h: one per 42)
//   * the weights never touch LDS: pack format 3 ([32-wide slice][plane][k group][column] x 16 B) is the MFMA operand layout,
//     a lane loads its fragments of the next slice from L2 while the current slice multiplies
//   * a product is a_lo b_hi + a_hi b_lo + a_hi b_hi on v_mfma_f32_16x16x32_f16, fp32 accumulate, smallest terms first
// The program checks itself (B = 3 images against an fp64 direct convolution of the same split operands) and then times the two
// convolutions of the stem at B = 64 (14 x 14 x 1024 -> 512 and 14 x 14 x 512 -> 512), with and without the output stores.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/conv_chain_probe.hip -o tools/probes/bin/conv_chain_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));        \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

constexpr int ROWS = 64;            // output rows of a workgroup
constexpr int NOUT = 512;           // output channels: all of them in one workgroup, 64 per wave
constexpr int KC = 256;             // channels of one tap that are in LDS at a time
constexpr int KG = KC / 8;          // 16-byte k groups of a chunk
constexpr int BUF = 2 * KG * ROWS * 16;   // one LDS buffer: [plane][k group][row] x 16 B = 64 KB
constexpr int SLOTS = 2 * KG * ROWS;      // 16-byte slots of a chunk (both planes)
constexpr int A_IT = SLOTS / 512;         // ... per thread: 8

struct ConvP {
  const _Float16* a_hi;   // [B][(H + 2)(W + 2)][C]: halo-padded input, plane hi
  const _Float16* a_lo;   // ... plane lo
  const char* wp;         // pack format 3 of W [9 C][NOUT]: [K / 32][plane][4][NOUT] x 16 B
  float* out;             // [B H W][NOUT]
  int B, H, W, C;
  float unscale;          // 2^-(eA + eW)
  int store;              // 0: timing without the output stores
};

__device__ __forceinline__ f32x4 mfma_f16(const u32x4 a, const u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(512) void conv_chain_kernel(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int M = p.B * p.H * p.W, HW = p.H * p.W, Wp = p.W + 2, Rp = (p.H + 2) * Wp;
  const int row0 = blockIdx.x * ROWS;
  const int cpt = p.C / KC;                 // chunks per tap
  const int nchunk = 9 * cpt;

  // ---- loader: slot s = tid + 512 i of a chunk: plane s / 2048; within the plane a wave covers 16 rows x 4 k groups (lane = 16 g + row)
  size_t a_src[A_IT];        // element offset of the slot's 8 channels at tap (0, 0), chunk 0 -- in units of fp16
  int a_dst[A_IT];
  int a_plane[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int s = tid + 512 * i;
    const int pl = s / (KG * ROWS), s2 = s - pl * (KG * ROWS);
    const int blk = s2 >> 6;                      // (k quad, 16-row tile)
    const int rt = blk & 3, kq = blk >> 2;
    const int row = rt * 16 + li, kg = kq * 4 + lg;
    const int m = min(row0 + row, M - 1);         // rows past the end re-read the last row (their outputs are not stored)
    const int b = m / HW, pos = m - b * HW, y = pos / p.W, x = pos - y * p.W;
    a_src[i] = ((size_t)b * Rp + (size_t)(y + 1) * Wp + (x + 1)) * p.C + kg * 8;
    a_dst[i] = ((pl * KG + kg) * ROWS + row) * 16;
    a_plane[i] = pl;
  }
  u32x4 ra[A_IT];
  auto load_chunk = [&](int ch_raw) __attribute__((always_inline)) {
    const int ch = min(ch_raw, nchunk - 1);
    const int tap = ch / cpt, cc = ch - tap * cpt;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const long shift = ((long)dy * Wp + dx) * p.C + cc * KC;
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      ra[i] = *reinterpret_cast<const u32x4*>((a_plane[i] ? p.a_lo : p.a_hi) + (long)a_src[i] + shift);
  };
  auto store_chunk = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) *reinterpret_cast<u32x4*>(lds + buf * BUF + a_dst[i]) = ra[i];
  };

  // ---- weights: lane (i = li, g = lg) holds k = 8 g .. 8 g + 7 of column i of each of the wave's four 16-column tiles, both planes
  u32x4 wb[2][2][4];         // [set][plane][column tile]
  auto load_w = [&](auto set_c, int ks_raw) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const int ks = min(ks_raw, 9 * p.C / 32 - 1);
    const char* src = p.wp + ((((size_t)ks * 2) * 4 + lg) * NOUT + wave * 64 + li) * 16;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int c = 0; c < 4; ++c) wb[SET][pl][c] = *reinterpret_cast<const u32x4*>(src + ((size_t)pl * 4 * NOUT + c * 16) * 16);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the activation fragments of a slice, from LDS: [set][plane][16-row tile]
  u32x4 aq[2][2][4];
  auto load_a = [&](auto set_c, int buf, int q) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const char* sA = lds + buf * BUF + ((q * 4 + lg) * ROWS + li) * 16;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int t = 0; t < 4; ++t) aq[SET][pl][t] = *reinterpret_cast<const u32x4*>(sA + pl * (KG * ROWS * 16) + t * 256);
  };
  // the products of one slice, operands swapped as in the chain kernels (D^T = W^T A^T: a lane ends up with FOUR CONSECUTIVE
  // COLUMNS of one row, a 16-byte store), smallest terms first
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

  load_chunk(0);
  load_w(S0{}, 0);
  store_chunk(0);
  __syncthreads();
  // The order "request slice q + 1, multiply slice q" is PINNED (sched_barrier): left alone, the scheduler sinks a slice's loads to
  // the end of the previous slice's products -- shortest live range -- i.e. to where they are needed, and the L2 round trip of the
  // weights is exposed in every slice (the first version of this probe: 0.61 PF).
#pragma unroll 1
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    load_chunk(ch + 1);                         // in flight under this chunk's 8 slices
    load_a(S0{}, buf, 0);
    const int ks0 = ch * (KC / 32);
#pragma unroll
    for (int q = 0; q < KC / 32; q += 2) {
      load_w(S1{}, ks0 + q + 1);
      load_a(S1{}, buf, q + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(S0{});
      __builtin_amdgcn_sched_barrier(0);
      load_w(S0{}, ks0 + q + 2);                // (q + 2 == 8: slice 0 of the next chunk)
      if (q + 2 < KC / 32) load_a(S0{}, buf, q + 2);
      __builtin_amdgcn_sched_barrier(0);
      mm(S1{});
      __builtin_amdgcn_sched_barrier(0);
    }
    store_chunk(buf ^ 1);                       // the other buffer was last read in the previous iteration, before its barrier
    __syncthreads();
  }

  if (!p.store) {
    if (acc[0][0][0] == 123.456f) p.out[0] = acc[3][3][3];
    return;
  }
  // accumulator map (swapped operands): row = lane & 15 of the 16-row tile, columns 4 (lane >> 4) .. + 3 of the 16-column tile
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int m = row0 + t * 16 + li;
    if (m < M) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<f32x4*>(p.out + (size_t)m * NOUT + wave * 64 + c * 16 + lg * 4) = acc[t][c] * p.unscale;
    }
  }
}

// fp64 direct convolution of the same split operands: out[m][n] = sum_tap sum_c (hi + lo)(src) (whi + wlo)(k, n)
__global__ void conv_ref_kernel(ConvP p, const _Float16* w_hi, const _Float16* w_lo, double* out) {
  const int M = p.B * p.H * p.W, HW = p.H * p.W, Wp = p.W + 2, Rp = (p.H + 2) * Wp;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)M * NOUT) return;
  const int m = (int)(idx / NOUT), n = (int)(idx - (size_t)m * NOUT);
  const int b = m / HW, pos = m - b * HW, y = pos / p.W, x = pos - y * p.W;
  double s = 0.0;
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const size_t src = ((size_t)b * Rp + (size_t)(y + 1 + dy) * Wp + (x + 1 + dx)) * p.C;
    for (int c = 0; c < p.C; ++c) {
      const double a = (double)p.a_hi[src + c] + (double)p.a_lo[src + c];
      const size_t k = (size_t)tap * p.C + c;
      const double w = (double)w_hi[k * NOUT + n] + (double)w_lo[k * NOUT + n];
      s += a * w;
    }
  }
  out[idx] = s * (double)p.unscale;
}

static float frand(uint32_t& st) {
  st = st * 1664525u + 1013904223u;
  return ((st >> 8) & 0xFFFFFF) / 16777216.0f * 2.0f - 1.0f;
}

struct Case {
  int B, H, W, C;
  std::vector<_Float16> a_hi, a_lo, w_hi, w_lo;
  std::vector<char> wp;
  float unscale;
};

// random input (ELU-like: mostly positive, some small negatives) and Xavier-like weights, split on the host with one exponent each
static void make_case(Case& cs, int B, int H, int W, int C, uint32_t seed) {
  cs.B = B; cs.H = H; cs.W = W; cs.C = C;
  const int Wp = W + 2, Rp = (H + 2) * Wp;
  const size_t na = (size_t)B * Rp * C, K = (size_t)9 * C;
  std::vector<float> a(na, 0.f), w(K * NOUT);
  uint32_t st = seed;
  float amax = 0.f, wmax = 0.f;
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x)
        for (int c = 0; c < C; ++c) {
          float v = frand(st) * 2.0f;
          v = v > 0.f ? v : 0.3f * v;
          a[((size_t)b * Rp + (size_t)(y + 1) * Wp + (x + 1)) * C + c] = v;
          amax = fmaxf(amax, fabsf(v));
        }
  const float lim = sqrtf(6.0f / (float)(K + NOUT));
  for (auto& v : w) { v = frand(st) * lim; wmax = fmaxf(wmax, fabsf(v)); }
  // x 2^e with the largest magnitude just below 2^14 (fp16 range with headroom for nothing: the products accumulate in fp32)
  auto expo = [](float m) { int e; frexpf(m, &e); return 14 - e; };
  const int eA = expo(amax), eW = expo(wmax);
  cs.unscale = ldexpf(1.0f, -(eA + eW));
  cs.a_hi.resize(na); cs.a_lo.resize(na);
  for (size_t i = 0; i < na; ++i) {
    const float s = ldexpf(a[i], eA);
    const _Float16 h = (_Float16)s;
    cs.a_hi[i] = h; cs.a_lo[i] = (_Float16)(s - (float)h);
  }
  cs.w_hi.resize(K * NOUT); cs.w_lo.resize(K * NOUT);
  cs.wp.resize(K * NOUT * 4);
  _Float16* wp = reinterpret_cast<_Float16*>(cs.wp.data());
  for (size_t k = 0; k < K; ++k)
    for (int n = 0; n < NOUT; ++n) {
      const float s = ldexpf(w[k * NOUT + n], eW);
      const _Float16 h = (_Float16)s, l = (_Float16)(s - (float)h);
      cs.w_hi[k * NOUT + n] = h; cs.w_lo[k * NOUT + n] = l;
      const size_t ks = k / 32, g = (k % 32) / 8, e = k % 8;
      wp[((((ks * 2 + 0) * 4 + g) * NOUT) + n) * 8 + e] = h;
      wp[((((ks * 2 + 1) * 4 + g) * NOUT) + n) * 8 + e] = l;
    }
}

template <class T>
static T* to_dev(const std::vector<T>& v) {
  T* d;
  CK(hipMalloc(&d, v.size() * sizeof(T)));
  CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
  // ---- correctness: 3 images (588 rows: 9 full tiles + one of 12 rows), 256 channels
  {
    Case cs;
    make_case(cs, 3, 14, 14, 256, 7u);
    ConvP p{to_dev(cs.a_hi), to_dev(cs.a_lo), to_dev(cs.wp), nullptr, cs.B, cs.H, cs.W, cs.C, cs.unscale, 1};
    const int M = cs.B * cs.H * cs.W;
    CK(hipMalloc(&p.out, (size_t)M * NOUT * 4));
    CK(hipMemset(p.out, 0xFF, (size_t)M * NOUT * 4));
    _Float16 *whi = to_dev(cs.w_hi), *wlo = to_dev(cs.w_lo);
    double* ref;
    CK(hipMalloc(&ref, (size_t)M * NOUT * 8));
    hipLaunchKernelGGL(conv_chain_kernel, dim3((M + ROWS - 1) / ROWS), dim3(512), 2 * BUF, 0, p);
    hipLaunchKernelGGL(conv_ref_kernel, dim3(((size_t)M * NOUT + 255) / 256), dim3(256), 0, 0, p, whi, wlo, ref);
    CK(hipDeviceSynchronize());
    std::vector<float> got((size_t)M * NOUT);
    std::vector<double> want((size_t)M * NOUT);
    CK(hipMemcpy(got.data(), p.out, got.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(want.data(), ref, want.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0, big = 0.0;
    for (size_t i = 0; i < got.size(); ++i) {
      worst = fmax(worst, fabs((double)got[i] - want[i]));
      big = fmax(big, fabs(want[i]));
    }
    printf("check (B=3, 14x14x256 -> 512): max |err| = %.3e of max |out| = %.3e  -> %.2e relative  %s\n", worst, big, worst / big,
           worst / big < 2e-6 ? "OK" : "MISMATCH");
    if (!(worst / big < 2e-6)) return 1;
  }
  // ---- timing: the two convolutions of the stem at B = 64
  for (int C : {1024, 512}) {
    Case cs;
    make_case(cs, 64, 14, 14, C, 11u);
    ConvP p{to_dev(cs.a_hi), to_dev(cs.a_lo), to_dev(cs.wp), nullptr, cs.B, cs.H, cs.W, cs.C, cs.unscale, 1};
    const int M = cs.B * cs.H * cs.W;
    CK(hipMalloc(&p.out, (size_t)M * NOUT * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int store = 1; store >= 0; --store) {
      p.store = store;
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(conv_chain_kernel, dim3((M + ROWS - 1) / ROWS), dim3(512), 2 * BUF, 0, p);
      CK(hipDeviceSynchronize());
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(conv_chain_kernel, dim3((M + ROWS - 1) / ROWS), dim3(512), 2 * BUF, 0, p);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / 4);
      }
      const double flop = 2.0 * M * 9.0 * C * NOUT;
      printf("conv 14x14x%-4d -> 512, B=64 (%d workgroups), %s: %8.1f us  %6.2f PF executed on 3 fp16 terms (%.0f TF algorithmic)\n", C,
             (M + ROWS - 1) / ROWS, store ? "with the output stores   " : "without the output stores", best * 1e3, 3 * flop / (best * 1e-3) / 1e15,
             flop / (best * 1e-3) / 1e12);
    }
  }
  printf("kb_gemm3h_kernel today (profiles/r04_model_level_kernel_stats_final.txt): conv 1024 -> 512: 440 us, conv 512 -> 512: 214 us\n");
  return 0;
}
