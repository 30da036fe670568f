// macx_h2.hip.h -- the "H2" tensor format: fp32 values stored ONCE, by the kernel that produces them, as the two fp16
// planes the matrix pipe consumes.
//
// gfx950 multiplies fp16 at the bf16 rate (2.5 PF dense) and f32 at 1/16 of it.  An fp32 value x scaled by a power of two
// into fp16's range splits as  x 2^e = hi + lo,  hi = fp16(x 2^e),  lo = fp16(x 2^e - hi)  (round-to-nearest both times):
// 11 + 11 significant bits plus the two rounding halvings leave |x 2^e - hi - lo| <= 2^-24 |x 2^e| -- fp32's own rounding
// unit -- as long as lo stays a normal fp16, i.e. for elements within 2^-16 of the block maximum; below that the error is
// absolute, <= 2^-40 of the block maximum.  A product is the three leading terms
//     a b  ~=  a_lo b_hi + a_hi b_lo + a_hi b_hi                (dropped: a_lo b_lo <= 2^-24 |a b|)
// on v_mfma_f32_16x16x32_f16 with fp32 accumulation (every fp16 x fp16 product is exact in fp32): fp32-class error
// (tests/test_gpu_units.py measures it against fp64 next to the native f32 MFMA kernel) at 3/16 of the f32 MFMA issue
// time, with 4 bytes per element -- the same HBM and LDS bytes as the fp32 tensor it replaces.
//
// Layout of an H2 tensor of R rows x C columns (C % 128 == 0), "slot-major":
//     plane p (0 = hi, 1 = lo):  slot[p][kg][row] = 8 consecutive columns 8 kg .. 8 kg + 7 of one row, 16 bytes,
//                                at byte ((p * C/8 + kg) * Rp + row) * 16,   Rp = R + H2_PAD_ROWS
//     exponents:                 e[row][cb] (int8) for the 128-column block cb: the stored fp16 are x * 2^e
// so the 16 lanes of an MFMA k-group read 16 consecutive rows = 256 contiguous bytes, a wave copies 64 rows of one slot
// column as one contiguous KiB, and every consumer's staging is a pure copy.  Exponents are per (row, 128 columns): a
// producer workgroup owns exactly such blocks, so no cross-workgroup reduction is needed and every row keeps its own
// relative precision.
#pragma once
#include "macx_common.hip.h"

namespace macx {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int H2_PAD_ROWS = 64;      // tiles may read (never use) up to this many rows past the end
constexpr int H2_E_MIN = -113, H2_E_MAX = 90;   // 2^e stays a normal fp32 for every finite fp32 block maximum

struct H2View {          // device view of one H2 tensor
  char* base;            // plane 0 | plane 1 | exponents
  int R, C;
  __host__ __device__ __forceinline__ int Rp() const { return R + H2_PAD_ROWS; }
  __host__ __device__ __forceinline__ size_t plane_bytes() const { return (size_t)(C >> 3) * Rp() * 16; }
  __host__ __device__ __forceinline__ char* plane(int p) const { return base + p * plane_bytes(); }
  __host__ __device__ __forceinline__ int8_t* exps() const { return reinterpret_cast<int8_t*>(base + 2 * plane_bytes()); }
  __host__ __device__ __forceinline__ int cb() const { return C >> 7; }
};
// floats a caller-owned buffer must hold for an H2 tensor (planes + exponents, 16-byte multiple)
__host__ __device__ inline size_t h2_floats(size_t R, size_t C) {
  const size_t Rp = R + H2_PAD_ROWS;
  const size_t bytes = 2 * (C >> 3) * Rp * 16 + Rp * (C >> 7);
  return ((bytes + 15) & ~(size_t)15) / 4;
}
inline H2View h2_view(float* buf, int R, int C) { return H2View{reinterpret_cast<char*>(buf), R, C}; }
inline H2View h2_view(const float* buf, int R, int C) { return H2View{reinterpret_cast<char*>(const_cast<float*>(buf)), R, C}; }

// exponent that lifts a block whose largest magnitude is `maxabs` into [2^14, 2^15)
__device__ __forceinline__ int h2_exponent(float maxabs) {
  const int be = (int)((__float_as_uint(maxabs) >> 23) & 0xFFu);
  int e = 141 - be;
  e = maxabs == 0.0f ? 0 : e;
  return min(max(e, H2_E_MIN), H2_E_MAX);
}
__device__ __forceinline__ float h2_pow2(int e) { return __uint_as_float((uint32_t)(127 + e) << 23); }   // 2^e, -126 <= e <= 127
// 2^-(ea + eb) for two stored exponents: as a product, so that a sum outside fp32's exponent range overflows / underflows
// the way the true scale would instead of wrapping the bit pattern
__device__ __forceinline__ float h2_unscale(int ea, int eb) { return h2_pow2(-ea) * h2_pow2(-eb); }

__device__ __forceinline__ uint32_t pk_f16(float a, float b) {        // v_cvt_pk_f16_f32 (RNE): low half = a
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ f32x2_t unpk_f16(uint32_t w) {
  return __builtin_convertvector(__builtin_bit_cast(f16x2_t, w), f32x2_t);
}
// 8 consecutive columns of one row, already multiplied by 2^e -> the hi and lo slots
__device__ __forceinline__ void h2_split8(const float* xs, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const float a = xs[2 * h], b = xs[2 * h + 1];
    hi[h] = pk_f16(a, b);
    const f32x2_t back = unpk_f16(hi[h]);
    lo[h] = pk_f16(a - back[0], b - back[1]);
  }
}
// the inverse: (hi + lo) * 2^-e; hi + lo is exact in fp32 (at most 24 significant bits)
__device__ __forceinline__ void h2_join8(const u32x4 hi, const u32x4 lo, float inv_scale, float* x) {
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const f32x2_t a = unpk_f16(hi[h]), b = unpk_f16(lo[h]);
    x[2 * h] = (a[0] + b[0]) * inv_scale;
    x[2 * h + 1] = (a[1] + b[1]) * inv_scale;
  }
}
__device__ __forceinline__ void h2_join4(const u32x2 hi, const u32x2 lo, float inv_scale, float* x) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x2_t a = unpk_f16(hi[h]), b = unpk_f16(lo[h]);
    x[2 * h] = (a[0] + b[0]) * inv_scale;
    x[2 * h + 1] = (a[1] + b[1]) * inv_scale;
  }
}

// 16-byte global -> LDS DMA (global_load_lds_dwordx4): the 64 lanes of a wave fetch 64 x 16 B from per-lane global
// addresses into LDS at lds_wave_base + lane * 16.  No VGPR round trip, no ds_write; counted by vmcnt.
// Issued as inline assembly ON PURPOSE: the compiler's wait-count pass answers an LDS-DMA it knows about with
// s_waitcnt vmcnt(0) in front of the next LDS read (it cannot prove the two do not alias), which drains the whole
// prefetch queue every stage.  Written this way the queue depth is ours: every consumer of a stage must sit behind
// wait_vmcnt<N>() + a workgroup barrier placed by hand.  (Loads the compiler does know about can only over-wait:
// vmcnt retires in order.)
__device__ __forceinline__ void dma16b(const char* g, char* lds_wave_base) {
  const uint32_t l = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory", "m0");
}

// ... with the LDS address as an integer (lds_addr_of() once per kernel + offsets): keeps pointer casts out of loops
__device__ __forceinline__ uint32_t lds_addr_of(const void* lds_ptr) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds_ptr;
}
__device__ __forceinline__ void dma16b(const char* g, uint32_t lds_wave_addr) {
  const uint32_t l = __builtin_amdgcn_readfirstlane(lds_wave_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory", "m0");
}

// ... with the global address as a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset: several DMA instructions
// of a stage that differ only in their base (plane, column tile) share ONE offset register instead of a 64-bit address each
__device__ __forceinline__ void dma16b_s(const char* sbase, uint32_t voff, uint32_t lds_wave_addr) {
  const uint32_t l = __builtin_amdgcn_readfirstlane(lds_wave_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(l) : "memory", "m0");
}

__device__ __forceinline__ void dma4b(const char* g, uint32_t lds_wave_addr) {      // 64 lanes x 4 B, lane-linear
  const uint32_t l = __builtin_amdgcn_readfirstlane(lds_wave_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(l) : "memory", "m0");
}

__device__ __forceinline__ f32x4 mfma_f16(const u32x4 a, const u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// v_mfma_f32_32x32x16_f16: lane l supplies row / column l & 31 and the k values 8 (l >> 5) .. + 7 of a 16-wide slice; the
// result D[i][j]: lane l holds j = l & 31 and i = 8 (reg >> 2) + 4 (l >> 5) + (reg & 3)
__device__ __forceinline__ f32x16 mfma32_f16(const u32x4 a, const u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Workgroup tile -> H2.  A workgroup that holds `rows` x 128 finished values (one question's rows, one 128-column
// block) row-major in LDS (T[lrow * ldt + col]) writes them out: lanes run along ROWS (a wave stores 64 consecutive rows
// of one slot column = one contiguous KiB per plane), the per-row maximum over the 128 columns is combined across the
// 16 slot columns through a small LDS table.  NT threads, NT % 64 == 0.  `scratch`: rows * 17 floats of LDS.
//   item = (slot column kgl in [0,16), local row): item index it = tid + NT * i,  kgl = it / rows_pad, lrow = it % rows_pad
// with rows_pad = rows rounded up to 64 so that a wave never straddles two slot columns.
// ---------------------------------------------------------------------------------------------------------------
template <int NT, int MAXROWS>
__device__ __forceinline__ void h2_store_tile(const float* T, int ldt, int rows, const H2View& o, size_t row0, int cb,
                                              float* scratch) {
  constexpr int RP = (MAXROWS + 63) & ~63;
  constexpr int ITEMS = (16 * RP + NT - 1) / NT;
  const int tid = threadIdx.x;
  float* rmax = scratch;                       // [16][rows] partial maxima, then [rows] exponents in row 0
  float v[ITEMS][8];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int it = tid + NT * i;
    const int kgl = it / RP, lrow = it - kgl * RP;
    if (kgl < 16 && lrow < rows) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(T + lrow * ldt + kgl * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(T + lrow * ldt + kgl * 8 + 4);
      float m = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[i][e] = a[e]; v[i][4 + e] = b[e]; m = fmaxf(m, fmaxf(fabsf(a[e]), fabsf(b[e]))); }
      rmax[kgl * rows + lrow] = m;
    }
  }
  __syncthreads();
  if (tid < rows) {
    float m = rmax[tid];
#pragma unroll
    for (int k = 1; k < 16; ++k) m = fmaxf(m, rmax[k * rows + tid]);
    const int e = h2_exponent(m);
    o.exps()[(row0 + tid) * o.cb() + cb] = (int8_t)e;
    reinterpret_cast<int*>(rmax)[16 * rows + tid] = e;
  }
  __syncthreads();
  const int* rexp = reinterpret_cast<const int*>(rmax) + 16 * rows;
  char* p0 = o.plane(0);
  const size_t pb = o.plane_bytes();
  const size_t Rp = o.Rp();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int it = tid + NT * i;
    const int kgl = it / RP, lrow = it - kgl * RP;
    if (kgl < 16 && lrow < rows) {
      const float s = h2_pow2(rexp[lrow]);
      float xs[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) xs[e] = v[i][e] * s;
      u32x4 hi, lo;
      h2_split8(xs, hi, lo);
      char* d = p0 + ((size_t)(cb * 16 + kgl) * Rp + row0 + lrow) * 16;
      *reinterpret_cast<u32x4*>(d) = hi;
      *reinterpret_cast<u32x4*>(d + pb) = lo;
    }
  }
}


#ifndef MACX_H2_NO_KERNELS     // (the chain kernels' translation units take the format above, not the kernels below)
// ---------------------------------------------------------------------------------------------------------------
// fp32 [B][N][C] row-major -> H2 (the caller's knowledge base enters the format here), optionally through a dropout
// site (ops.py:678: out = kb * mask / keep) whose keep bits are kept row-major for the dKB epilogue, and optionally
// emitting the keep BYTES of a second site over the same index range (the attention dropout, ops.py:312 via :142) in
// slot order -- the pass is HBM-bound, the second hash is free.  One workgroup per (question row block, 128 columns).
// ---------------------------------------------------------------------------------------------------------------
constexpr int H2C_ROWS = 64;           // rows per workgroup: 38 KB of LDS, four workgroups per CU -- one's load, hash and
                                       // store phases overlap the others' (208-row tiles: one workgroup per CU, 17 us)
constexpr int H2C_THREADS = 512;
constexpr int H2C_LDT = 132;
constexpr size_t H2C_LDS = (size_t)H2C_ROWS * H2C_LDT * 4 + (size_t)17 * H2C_ROWS * 4;

struct H2FromP {
  const float* src; int B, N, C;
  H2View out;
  uint32_t first;                        // flat dropout index of element (0,0,0): b0 * N * ldrop
  int ldrop;                             // row stride of the dropout index (the logical width of a zero-padded tensor); 0 = C
  uint32_t key, thr24; float inv_keep;   // site 1 (thr24 = 1 << 24: keep everything)
  uint32_t* bits;                        // site 1 keep bits, row-major [B*N][C/32]; may be null
  uint32_t key2, thr24_2;                // site 2
  const uint32_t* word;                  // macx_dropout.mask_word (device, may be null): XORed into both keys at run time
  uint8_t* bytes2;                       // site 2 keep bytes, slot order [C/8][Rp]; may be null
};

__global__ __launch_bounds__(H2C_THREADS) void h2_from_f32_kernel(H2FromP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* T = smem;
  float* scratch = smem + H2C_ROWS * H2C_LDT;
  const int nrb = (p.N + H2C_ROWS - 1) / H2C_ROWS;
  const int b = blockIdx.x / nrb, rbi = blockIdx.x - b * nrb, cb = blockIdx.y;
  const int row0 = rbi * H2C_ROWS;
  const int rows = min(H2C_ROWS, p.N - row0);
  const size_t grow0 = (size_t)b * p.N + row0;
  const int tid = threadIdx.x, c4 = tid & 31, rg = tid >> 5;
  const bool drop = p.thr24 < (1u << 24);
  const int ldrop = p.ldrop > 0 ? p.ldrop : p.C;
  const uint32_t key = run_key(p.key, p.word), key2 = run_key(p.key2, p.word);
  for (int lrow = rg; lrow < rows; lrow += H2C_THREADS / 32) {
    const size_t e0 = (grow0 + lrow) * p.C + cb * 128 + c4 * 4;
    const uint32_t d0 = (uint32_t)((grow0 + lrow) * (size_t)ldrop + cb * 128 + c4 * 4);
    f32x4 v = *reinterpret_cast<const f32x4*>(p.src + e0);
    if (drop) {
      uint32_t nib = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool keep = keep_bit(p.first + d0 + e, key, p.thr24);
        nib |= (keep ? 1u : 0u) << e;
        v[e] = keep ? v[e] * p.inv_keep : 0.f;
      }
      if (p.bits) {
        uint32_t w = nib << (4 * (c4 & 7));
        w |= __shfl_xor(w, 1, 64);
        w |= __shfl_xor(w, 2, 64);
        w |= __shfl_xor(w, 4, 64);
        if ((c4 & 7) == 0) p.bits[e0 >> 5] = w;
      }
    }
    *reinterpret_cast<f32x4*>(T + lrow * H2C_LDT + c4 * 4) = v;
  }
  __syncthreads();
  h2_store_tile<H2C_THREADS, H2C_ROWS>(T, H2C_LDT, rows, p.out, grow0, cb, scratch);
  if (p.bytes2) {
    const size_t Rp = p.out.Rp();
    static_assert((H2C_ROWS & (H2C_ROWS - 1)) == 0, "row index by mask");
    for (int it = tid; it < 16 * H2C_ROWS; it += H2C_THREADS) {
      const int kgl = it / H2C_ROWS, lrow = it & (H2C_ROWS - 1);
      if (lrow < rows) {
        const uint32_t e0 = p.first + (uint32_t)((grow0 + lrow) * (size_t)ldrop + cb * 128 + kgl * 8);
        uint32_t byte = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) byte |= (keep_bit(e0 + e, key2, p.thr24_2) ? 1u : 0u) << e;
        p.bytes2[(size_t)(cb * 16 + kgl) * Rp + grow0 + lrow] = (uint8_t)byte;
      }
    }
  }
}

// H2 -> fp32 row-major (tests and tools; not on the timed path)
__global__ void h2_to_f32_kernel(H2View in, float* out) {
  const size_t nslot = (size_t)(in.C >> 3) * in.R;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslot; i += (size_t)gridDim.x * blockDim.x) {
    const size_t kg = i / in.R, row = i - kg * in.R;
    const char* s = in.plane(0) + (kg * in.Rp() + row) * 16;
    const u32x4 hi = *reinterpret_cast<const u32x4*>(s), lo = *reinterpret_cast<const u32x4*>(s + in.plane_bytes());
    float x[8];
    h2_join8(hi, lo, h2_pow2(-(int)in.exps()[row * in.cb() + (kg >> 4)]), x);
    float* d = out + row * in.C + kg * 8;
    *reinterpret_cast<f32x4*>(d) = f32x4{x[0], x[1], x[2], x[3]};
    *reinterpret_cast<f32x4*>(d + 4) = f32x4{x[4], x[5], x[6], x[7]};
  }
}

// Minimum over the rows of the exponents of up to four FAMILIES of H2 tensors (a family: `nt` tensors `stride` bytes apart), per
// 128-column block: the common exponent a contraction over rows brings every row to.  Plain stores only: workgroup x of family
// y writes its partial minima to part[(y * gridDim.x + x) * 8 + k]; a consumer takes the minimum over the gridDim.x partials
// (h2_emin_final).  The rounds before ran these minima as an atomicMin into an array preset by a memset; under HIP-graph replay
// that pair was seen to run out of order, so no kernel of this library orders itself against a memset any more.
constexpr int EMIN_NB = 64;
struct EminList { const char* base[8]; size_t stride[8]; int nt[8]; int R, C; int* part; };
__global__ __launch_bounds__(256) void h2_emin_list_kernel(EminList L) {
  __shared__ int red[8][4];
  const int f = blockIdx.y;
  const int ncb = L.C >> 7;
  int m[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = 127;
  for (int t = 0; t < L.nt[f]; ++t) {
    const H2View v{const_cast<char*>(L.base[f]) + (size_t)t * L.stride[f], L.R, L.C};
    const int8_t* e = v.exps();
    if (ncb == 4 && (reinterpret_cast<uintptr_t>(e) & 3) == 0) {
      // (a row's four exponent bytes as one word: the byte loop made this kernel 22 us for 2.4 MB)
      const uint32_t* e4 = reinterpret_cast<const uint32_t*>(e);
      for (int r = blockIdx.x * 256 + threadIdx.x; r < L.R; r += gridDim.x * 256) {
        const uint32_t w = e4[r];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = min(m[k], (int)(int8_t)(w >> (8 * k)));
      }
    } else
    for (int r = blockIdx.x * 256 + threadIdx.x; r < L.R; r += gridDim.x * 256)
      for (int k = 0; k < ncb; ++k) m[k] = min(m[k], (int)e[(size_t)r * ncb + k]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < ncb; ++k) {
    int x = m[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = min(x, __shfl_xor(x, o, 64));
    if (lane == 0) red[k][wave] = x;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    int x = 127;
    if ((int)threadIdx.x < ncb) x = min(min(red[threadIdx.x][0], red[threadIdx.x][1]), min(red[threadIdx.x][2], red[threadIdx.x][3]));
    L.part[((size_t)f * gridDim.x + blockIdx.x) * 8 + threadIdx.x] = x;
  }
}
// the family's common exponent of column block k from its nb partials (an all-zero family keeps exponent-0 rows: 127 never
// reaches a kernel)
__device__ __forceinline__ int h2_emin_final(const int* part, int nb, int k) {
  int m = 127;
  for (int b = 0; b < nb; ++b) m = min(m, part[(size_t)b * 8 + k]);
  return min(m, 126);
}

// ---- weights -------------------------------------------------------------------------------------------------
// max |W| of up to 8 matrices: blockIdx.y selects the matrix, gridDim.x workgroups share it.  With more than one workgroup per
// matrix each writes its partial maximum and absmax_finish_kernel combines them -- plain stores all the way: the earlier form
// (a 16-byte memset + an integer atomicMax per workgroup) made the exponent of every packed weight depend on the memset having
// landed before the first atomic, and replayed HIP graphs were seen to give finite but 1e-2-wrong runs in a fraction of the
// processes (mac-network_amd/graph.py).  NaN entries are skipped by fmaxf.
struct AbsMaxList { const float* src[8]; size_t n[8]; float* out; float* part; };   // part: [8][gridDim.x], or null when gridDim.x == 1
__global__ __launch_bounds__(256) void absmax_kernel(AbsMaxList L) {
  __shared__ float red[4];
  const float* s = L.src[blockIdx.y];
  const size_t n = L.n[blockIdx.y];
  float m = 0.f;
  if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(s) & 15) == 0) {
    // 16-byte loads, four per thread in flight: the launch is a chain of load round trips (round 6: 10.3 -> ~5 us for four d x d weights)
    const f32x4* s4 = reinterpret_cast<const f32x4*>(s);
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = s4[min(i + u * stride, n4 - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u][0]), fabsf(v[u][1]))), fmaxf(fabsf(v[u][2]), fabsf(v[u][3])));
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(s[i]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (gridDim.x == 1) L.out[blockIdx.y] = t;
    else L.part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
  }
}
// ... of ONE large tensor (n % 4 == 0, 16-byte aligned): 16-byte loads, gridDim.x partials for absmax_finish_kernel
__global__ __launch_bounds__(256) void absmax_vec_kernel(const float* __restrict__ s, size_t n4, float* part) {
  __shared__ float red[4];
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(s)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// out[mat] = max over part[mat][0..nblk): one wave per matrix (blockDim = 64 * nmat <= 512)
__global__ void absmax_finish_kernel(const float* __restrict__ part, int nblk, float* out) {
  const int mat = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float m = 0.f;
  for (int i = lane; i < nblk; i += 64) m = fmaxf(m, part[(size_t)mat * nblk + i]);
  m = wave_max(m);
  if (lane == 0) out[mat] = m;
}
// pack format 3: W (or W^T) as H2 weight planes  dst[kt][plane][g][Nout] x 16 B  (8 consecutive k of one column),
// scaled by the per-matrix exponent derived from *maxabs; the exponent is stored as an int after the planes.
constexpr int H2_WE_LO = -20, H2_WE_HI = 36;
__device__ __forceinline__ int h2_weight_exponent(float maxabs) { return min(max(h2_exponent(maxabs), H2_WE_LO), H2_WE_HI); }
__device__ __forceinline__ void pack_h2_weight(const float* src, int ld_k, int ld_j, int K, int Nout, int k_src, int n_src,
                                               const float* maxabs, float* dst, size_t tid_global, size_t nthreads,
                                               float* exp_dst = nullptr, int maxabs_n = 0, float* maxabs_out = nullptr) {
  // maxabs_n > 0: `maxabs` holds that many partial maxima (absmax_kernel's per-workgroup results): every thread combines them itself
  // (max is order-free) and the finishing launch between absmax and pack is not needed; the combined value is left in *maxabs_out
  float mabs = maxabs[0];
  for (int i = 1; i < maxabs_n; ++i) mabs = fmaxf(mabs, maxabs[i]);
  if (maxabs_out && tid_global == 0) *maxabs_out = mabs;
  const int e = h2_weight_exponent(mabs);
  const float s = h2_pow2(e);
  const size_t nslot = (size_t)(K >> 3) * Nout;
  const bool vec8 = ld_k == 1 && (ld_j & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
  char* d0 = reinterpret_cast<char*>(dst);
  for (size_t i = tid_global; i < nslot; i += nthreads) {
    const int j = (int)(i % Nout);
    const int kg = (int)(i / Nout);
    float x[8];
    if (vec8 && kg * 8 + 8 <= k_src && j < n_src) {
      // transposed source: the slot's eight k values are contiguous -- two 16-byte loads instead of eight scalar ones whose
      // lanes sit a whole row apart
      const float* sp = src + (size_t)kg * 8 + (size_t)j * ld_j;
      const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { x[q] = a[q] * s; x[4 + q] = b[q] * s; }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = kg * 8 + q;
        x[q] = (k < k_src && j < n_src) ? src[(size_t)k * ld_k + (size_t)j * ld_j] * s : 0.f;
      }
    }
    u32x4 hi, lo;
    h2_split8(x, hi, lo);
    const int kt = kg >> 2, g = kg & 3;
    char* d = d0 + ((((size_t)kt * 2) * 4 + g) * Nout + j) * 16;
    *reinterpret_cast<u32x4*>(d) = hi;
    *reinterpret_cast<u32x4*>(d + (size_t)4 * Nout * 16) = lo;
  }
  // (exp_dst: several row blocks packed back to back as ONE matrix -- the per-tap transposes of a convolution's backward-data
  // weights -- share one exponent behind the last of them)
  if (tid_global == 0) *reinterpret_cast<int*>(exp_dst ? exp_dst : dst + (size_t)K * Nout) = e;
}

#endif  // MACX_H2_NO_KERNELS

}  // namespace macx
