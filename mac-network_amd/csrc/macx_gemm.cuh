// macx_gemm.cuh -- the knowledge-base GEMM family of the read unit (forward and backward-data).
//
// One kernel template covers every  [B*N, K] x [K, 512]  contraction of the read unit
// (mac_cell.py:209-277 / ops.py:668-725, 298-333): projX, memKbProj, memKbProj_2 forward and the
// three dY @ W^T products of the backward pass.  fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32 (exact fp32, bit-equal to an fmaf chain, 157 TF peak on MI355X).
//
// Tiling is per QUESTION, not over the flat row index:  a workgroup owns RT*32 consecutive
// knowledge-base cells of ONE question x 128 output columns.  With N = 196 (CLEVR) and RT = 7 the
// grid is B x 4 = 256 workgroups for B = 64 -- exactly one per CU -- and every per-question vector
// the read unit broadcasts over the KB (projected memory y, control c) is workgroup-uniform, which
// is what allows the row-broadcast products to move out of the [B,N,d] tensors and into the
// weight tile (B_YMIX_*) or the epilogue (E_I2_LOGIT).
//
// Layouts
//   A  (activations)  row-major [B][N][lda], read as float4 along k, staged in LDS with a
//                     36-float row stride (conflict-free ds_read_b128 by 16-lane groups).
//   W  (weights)      PRE-PACKED [K/8][2][Nout][4]  with  Wp[q][h][j][e] = W[8q + 4h + e][j]
//                     so a lane's four consecutive MFMA B operands are one ds_read_b128 and the
//                     global->LDS copy is linear.  The MFMA k-pairing {8q+e, 8q+4+e} is the same
//                     on the A side (lane half h reads A[i][8q+4h .. 8q+4h+3]).
#pragma once
#include "macx_common.cuh"

namespace macx {

constexpr int G_BK = 32;           // reduction slice per stage
constexpr int G_BN = 128;          // output columns per workgroup (4 waves x 32)
constexpr int G_LDA = G_BK + 4;    // padded LDS row stride (floats)
constexpr int G_BTILE = G_BK * G_BN;

enum : int { A_PLAIN = 0, A_DROP = 1 };
enum : int { B_PLAIN = 0, B_YMIX_ROW = 1, B_YMIX_COL = 2 };
enum : int {
  E_BIAS_ACT = 0,   // out = act(acc + bias)                              (X: act NON; H1: readMemAct)
  E_I2_LOGIT = 2,   // I2 = acc + bias; logit partial of act(I2*c) . w_k  (read-unit attention logits)
  E_MUL_DACT = 3,   // out = acc * act'(aux)                              (dI1 = (dI2 W2^T) * act'(I1))
  E_PLAIN = 4,      // out = acc                                          (dX)
  E_DKB = 5         // out (+)= acc * dropmask + att[b,n] * dr[b][j]      (dKB accumulation)
};

struct GemmP {
  // problem
  int B, N, K, Nout;
  int b0;                 // global index of question 0 (dropout stream / data-parallel shard offset)
  // A operand
  const float* A;
  int lda;
  DropSpec a_drop;        // A_DROP: mask indexed ((b0+b)*N + n)*lda + k
  // weights
  const float* Wp;        // packed
  const float* Wp2;       // packed second weight (B_YMIX_*)
  const float* y;         // [B][ldy] per-question vector mixed into the weight tile
  int ldy;
  // epilogue
  float* out;
  int ldo;
  const float* bias;      // [Nout]
  int act;                // activation code (E_BIAS_ACT, E_I2_LOGIT, E_MUL_DACT)
  const float* aux;       // E_MUL_DACT: H1 [B*N][ldo];  E_DKB: dr [B][Nout]
  const float* cvec;      // E_I2_LOGIT: control [B][Nout]
  const float* wvec;      // E_I2_LOGIT: logits weight [Nout]
  const float* att;       // E_DKB: kb attention [B][N]
  float* logit_part;      // E_I2_LOGIT: [Nout/128][B*N]
  float* colsum_part;     // optional: column sums of `out` per workgroup-row  [B*nrb][Nout]
  DropSpec e_drop;        // E_I2_LOGIT: mask on elu(I2*c) indexed ((b0+b)*N+n)*Nout + j ; E_DKB: KB mask
  int accumulate;         // E_DKB: 1 -> out += , 0 -> out =
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int RT, int AP, int BP, int EP, bool COLSUM>
__global__ __launch_bounds__(256) void kb_gemm_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int A_TILE = RT * 32 * G_LDA;
  float* sA = smem;                  // [2][A_TILE]
  float* sB = smem + 2 * A_TILE;     // [2][G_BTILE]

  // ---- workgroup -> (question, row block, column block); XCD-aware so that the column blocks
  // of one question (which share the A rows) sit on one XCD's L2.
  const int nblk = gridDim.x;
  int v = blockIdx.x;
  if ((nblk & 7) == 0) v = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int ncb = p.Nout / G_BN;
  const int nrb = (p.N + RT * 32 - 1) / (RT * 32);
  const int cb = v % ncb;
  const int rbi = (v / ncb) % nrb;
  const int b = v / (ncb * nrb);
  const int row0 = rbi * RT * 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nk = p.K / G_BK;

  f32x16 acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.0f;

  f32x4 ra[RT];
  f32x4 rw[4];
  f32x4 rw2[4];

  const float* Abase = p.A + (size_t)b * p.N * p.lda;

  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int f = tid + 256 * i;
      const int n = row0 + (f >> 3);
      const int kq = f & 7;
      if (n < p.N) {
        ra[i] = *reinterpret_cast<const f32x4*>(Abase + (size_t)n * p.lda + kt * G_BK + kq * 4);
      } else {
        ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int chunk = f >> 7;   // q*2 + h
      const int j = f & 127;
      const size_t off = ((size_t)(kt * 8 + chunk) * p.Nout + cb * G_BN + j) * 4;
      rw[i] = *reinterpret_cast<const f32x4*>(p.Wp + off);
      if (BP != B_PLAIN) rw2[i] = *reinterpret_cast<const f32x4*>(p.Wp2 + off);
    }
  };

  auto store_tiles = [&](int buf, int kt) {
    float* dA = sA + buf * A_TILE;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int f = tid + 256 * i;
      const int row = f >> 3;
      const int kq = f & 7;
      f32x4 val = ra[i];
      if (AP == A_DROP) {
        const int n = row0 + row;
        const uint32_t idx = (uint32_t)(((size_t)(p.b0 + b) * p.N + n) * p.lda + kt * G_BK + kq * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = drop_apply(val[e], idx + e, p.a_drop);
      }
      *reinterpret_cast<f32x4*>(dA + row * G_LDA + kq * 4) = val;
    }
    float* dB = sB + buf * G_BTILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      f32x4 val = rw[i];
      if (BP == B_YMIX_ROW) {
        // B_eff[k][j] = y[b][k] * W1a[k][j] + W1b[k][j]   (ops.py:703,718 folded into the weights)
        const int chunk = f >> 7;
        const int k = kt * G_BK + (chunk >> 1) * 8 + (chunk & 1) * 4;
        const f32x4 y4 = *reinterpret_cast<const f32x4*>(p.y + (size_t)b * p.ldy + k);
        val = val * y4 + rw2[i];
      } else if (BP == B_YMIX_COL) {
        // B_eff[k][j] = y[b][j] * W1a^T[k][j] + W1b^T[k][j]   (backward-data of the same product)
        const float ys = p.y[(size_t)b * p.ldy + cb * G_BN + (f & 127)];
        val = val * ys + rw2[i];
      }
      *reinterpret_cast<f32x4*>(dB + f * 4) = val;
    }
  };

  auto compute = [&](int buf) {
    const float* a = sA + buf * A_TILE + (lane & 31) * G_LDA + (lane >> 5) * 4;
    const float* bq = sB + buf * G_BTILE + ((lane >> 5) * G_BN + wave * 32 + (lane & 31)) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bf = *reinterpret_cast<const f32x4*>(bq + q * 2 * G_BN * 4);
      f32x4 af[RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) af[r] = *reinterpret_cast<const f32x4*>(a + r * 32 * G_LDA + q * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = mfma32(af[r][e], bf[e], acc[r]);
    }
  };

  // ---- main loop: register-staged double buffer, one barrier per k-slice
  load_tiles(0);
  store_tiles(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
    compute(cur);
    if (kt + 1 < nk) store_tiles(cur ^ 1, kt + 1);
    __syncthreads();
  }

  // ---- epilogue.  32x32 accumulator map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col = cb * G_BN + wave * 32 + (lane & 31);
  const int rhalf = 4 * (lane >> 5);
  float bias = 0.f;
  if (EP == E_BIAS_ACT || EP == E_I2_LOGIT) bias = p.bias[col];
  float cj = 0.f, wj = 0.f, drj = 0.f;
  if (EP == E_I2_LOGIT) {
    cj = p.cvec[(size_t)b * p.Nout + col];
    wj = p.wvec[col];
  }
  if (EP == E_DKB) drj = p.aux[(size_t)b * p.Nout + col];
  float csum = 0.f;

  float* red = smem;   // [4 waves][RT*32] row partials (E_I2_LOGIT); safe: all LDS reads are done
#pragma unroll
  for (int r = 0; r < RT; ++r) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int lrow = r * 32 + (e & 3) + 8 * (e >> 2) + rhalf;
      const int n = row0 + lrow;
      const bool ok = n < p.N;
      const size_t orow = (size_t)b * p.N + n;
      float val = acc[r][e];
      if (EP == E_BIAS_ACT) {
        val = act_apply(p.act, val + bias);
        if (ok) p.out[orow * p.ldo + col] = val;
      } else if (EP == E_I2_LOGIT) {
        val += bias;
        if (ok) p.out[orow * p.ldo + col] = val;
        // mac_cell.py:248,262,266: act(I2 * c) -> dropout -> . w  (bias b_k added in kb_attend)
        float g = act_apply(p.act, val * cj);
        const uint32_t idx = (uint32_t)(((size_t)(p.b0 + b) * p.N + n) * p.Nout + col);
        g = drop_apply(g, idx, p.e_drop);
        float part = half_sum(g * wj);
        if ((lane & 31) == 0) red[wave * (RT * 32) + lrow] = part;
      } else if (EP == E_MUL_DACT) {
        if (ok) {
          const float h = p.aux[orow * p.ldo + col];
          val *= act_grad_from_out(p.act, h);
          p.out[orow * p.ldo + col] = val;
        } else {
          val = 0.f;
        }
      } else if (EP == E_PLAIN) {
        if (ok) p.out[orow * p.ldo + col] = val; else val = 0.f;
      } else if (EP == E_DKB) {
        if (ok) {
          const uint32_t idx = (uint32_t)(((size_t)(p.b0 + b) * p.N + n) * p.ldo + col);
          val = drop_apply(val, idx, p.e_drop) + p.att[orow] * drj;
          if (p.accumulate) val += p.out[orow * p.ldo + col];
          p.out[orow * p.ldo + col] = val;
        }
      }
      if (COLSUM) csum += ok ? val : 0.f;
    }
  }
  if (EP == E_I2_LOGIT) {
    __syncthreads();
    for (int lrow = tid; lrow < RT * 32; lrow += 256) {
      const int n = row0 + lrow;
      if (n < p.N) {
        const float s = red[lrow] + red[RT * 32 + lrow] + red[2 * RT * 32 + lrow] + red[3 * RT * 32 + lrow];
        p.logit_part[(size_t)cb * p.B * p.N + (size_t)b * p.N + n] = s;
      }
    }
  }
  if (COLSUM) {
    // column sum over this workgroup's rows: the two half-waves hold different rows of one column
    csum += __shfl_xor(csum, 32, 64);
    if (lane < 32) p.colsum_part[(size_t)(b * nrb + rbi) * p.Nout + col] = csum;
  }
}

template <int RT>
constexpr size_t kb_gemm_lds_bytes() {
  return (size_t)(2 * RT * 32 * G_LDA + 2 * G_BTILE) * sizeof(float);
}

// ---- host-side launcher ---------------------------------------------------------------------
template <int RT, int AP, int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm_launch_rt(const GemmP& p, hipStream_t st) {
  auto kern = kb_gemm_kernel<RT, AP, BP, EP, COLSUM>;
  constexpr size_t lds = kb_gemm_lds_bytes<RT>();
  static bool attr_set = false;   // one attribute call per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int ncb = p.Nout / G_BN;
  const int nrb = (p.N + RT * 32 - 1) / (RT * 32);
  const int grid = p.B * nrb * ncb;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);
  return hipGetLastError();
}

// rows-per-workgroup choice: the smallest RT in {1,2,4,7} whose tile covers a question's N rows
// in the fewest equal row blocks (196 -> 7, 49 -> 2, 14 -> 1; N > 224 -> 7 with several row blocks).
inline int kb_gemm_pick_rt(int N) {
  if (N <= 32) return 1;
  if (N <= 64) return 2;
  if (N <= 128) return 4;
  return 7;
}

template <int AP, int BP, int EP, bool COLSUM>
inline hipError_t kb_gemm_launch(const GemmP& p, hipStream_t st) {
  switch (kb_gemm_pick_rt(p.N)) {
    case 1: return kb_gemm_launch_rt<1, AP, BP, EP, COLSUM>(p, st);
    case 2: return kb_gemm_launch_rt<2, AP, BP, EP, COLSUM>(p, st);
    case 4: return kb_gemm_launch_rt<4, AP, BP, EP, COLSUM>(p, st);
    default: return kb_gemm_launch_rt<7, AP, BP, EP, COLSUM>(p, st);
  }
}

}  // namespace macx
