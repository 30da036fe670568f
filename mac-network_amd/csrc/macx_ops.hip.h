// macx_ops.hip.h -- the ops.py primitives as single kernels: the building blocks of the GENERIC option path
// (mac-network_amd/generic.py), which runs every legal option combination the fused cell kernels do not cover as one
// kernel per reference op.  All of it is fp32, HBM-bound streaming work: coalesced float4 / row-contiguous access, one pass.
//
//   act / act_bwd        ops.activations (ops.py:161-187): NON TANH SIGMOID ELU RELU, and PRELU with a per-channel alpha
//   binary               out = scale * (a (+|*) b) with b the same shape, [B,d] over [B,N,d], [d] over rows, or [rows] over columns
//   reduce               sum over the middle axis of [B,N,d], over the last axis of [rows,d], or over the rows of [rows,d]
//   softmax / _bwd       softmax over the last axis with the -inf length mask of ops.expMask (ops.py:243-247)
//   dropout              x / keep * mask from the stateless stream (macx_common.hip.h), any site / step / first element
// Deterministic: every output element is produced by one thread or one fixed-order tree.
#pragma once
#include "macx_common.hip.h"

namespace macx {

constexpr int OP_ACT_PRELU = 16;        // beyond MACX_ACT_*: relu(x) - alpha[c] * relu(-x)  (ops.py:171-173)
constexpr int OP_ACT_RSQRT_EPS = 17;    // 1 / sqrt(x + alpha[0]): the normaliser of tf.contrib.layers.batch_norm (mac_cell.py:370-373)

enum { OP_ADD = 0, OP_MUL = 1 };
enum { OP_B_SAME = 0, OP_B_MID = 1, OP_B_CHANNEL = 2, OP_B_ROW = 3 };
enum { OP_R_MID = 0, OP_R_LAST = 1, OP_R_ROWS = 2 };

// ---- activations -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void op_act_kernel(int act, const float* x, const float* alpha, size_t n, int inner, float* out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = x[i];
    out[i] = act == OP_ACT_PRELU ? (v > 0.f ? v : alpha[i % inner] * v)
             : act == OP_ACT_RSQRT_EPS ? 1.0f / sqrtf(v + alpha[0]) : act_apply(act, v);
  }
}
// dx = dy * act'(x); PRELU also needs d alpha[c] = sum_rows dy * min(x, 0): the caller reduces `dalpha_elem` over rows
__global__ __launch_bounds__(256) void op_act_bwd_kernel(int act, const float* x, const float* alpha, const float* dy, size_t n, int inner,
                                                         float* dx, float* dalpha_elem) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = x[i], g = dy[i];
    float d;
    if (act == OP_ACT_PRELU) {
      d = v > 0.f ? 1.f : alpha[i % inner];
      dalpha_elem[i] = v > 0.f ? 0.f : g * v;
    } else if (act == OP_ACT_RSQRT_EPS) {
      const float r = 1.0f / sqrtf(v + alpha[0]);
      d = -0.5f * r * r * r;
    } else if (act == ACT_ELU) {
      d = elu_grad_from_in(v);
    } else {
      d = act_grad_from_out(act, act_apply(act, v));
    }
    dx[i] = g * d;
  }
}

// ---- broadcasting binary ---------------------------------------------------------------------------------------
// a, out: [n] viewed as [.., mid, inner]; b by mode: SAME [n] | MID [n / (mid * inner)][inner] | CHANNEL [inner] | ROW [n / inner]
__global__ __launch_bounds__(256) void op_binary_kernel(int op, int bmode, const float* a, const float* b, size_t n, int mid, int inner,
                                                        float scale, float* out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    size_t j;
    switch (bmode) {
      case OP_B_MID: j = (i / ((size_t)mid * inner)) * inner + i % inner; break;
      case OP_B_CHANNEL: j = i % inner; break;
      case OP_B_ROW: j = i / inner; break;
      default: j = i; break;
    }
    const float x = a[i], y = b[j];
    out[i] = scale * (op == OP_MUL ? x * y : x + y);
  }
}

// ---- reductions -------------------------------------------------------------------------------------------------
// MID: out[b][c] = sum_n x[b][n][c]; one thread per (b, c), rows read coalesced along c
__global__ __launch_bounds__(256) void op_reduce_mid_kernel(const float* x, int B, int N, int d, float* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * d) return;
  const size_t b = i / d, c = i % d;
  const float* p = x + b * N * d + c;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += p[(size_t)n * d];
  out[i] = s;
}
// LAST: out[r] = sum_c x[r][c]; one wave per row
__global__ __launch_bounds__(256) void op_reduce_last_kernel(const float* x, size_t rows, int d, float* out) {
  const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += x[r * d + c];
  s = wave_sum(s);
  if (lane == 0) out[r] = s;
}
// ROWS: out[c] = sum_r x[r][c]; two passes with a fixed split so that the order does not depend on the launch
constexpr int OP_ROWS_SPLIT = 64;
__global__ __launch_bounds__(256) void op_reduce_rows_kernel(const float* x, size_t rows, int d, float* part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const size_t per = (rows + OP_ROWS_SPLIT - 1) / OP_ROWS_SPLIT;
  const size_t r0 = blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
  float s = 0.f;
  for (size_t r = r0; r < r1; ++r) s += x[r * d + c];
  part[(size_t)blockIdx.y * d + c] = s;
}
__global__ __launch_bounds__(256) void op_reduce_rows_final_kernel(const float* part, int d, float* out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  float s = 0.f;
  for (int k = 0; k < OP_ROWS_SPLIT; ++k) s += part[(size_t)k * d + c];
  out[c] = s;
}

// ---- softmax over the last axis; columns >= lengths[row / rows_per_len] count as -inf --------------------------------
__global__ __launch_bounds__(256) void op_softmax_kernel(const float* x, const int32_t* lengths, int rows_per_len, size_t rows, int n, float* out) {
  const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const int len = lengths ? min(max(lengths[r / rows_per_len], 0), n) : n;
  const float* p = x + r * n;
  float m = -INFINITY;
  for (int c = lane; c < len; c += 64) m = fmaxf(m, p[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < len; c += 64) s += expf(p[c] - m);
  s = wave_sum(s);
  const float inv = 1.0f / s;             // len == 0: 1/0 = inf, exp(..) never evaluated, every output 0 * .. -> the reference's NaN row
  for (int c = lane; c < n; c += 64) out[r * n + c] = c < len ? expf(p[c] - m) * inv : (len == 0 ? NAN : 0.f);
}
// dx = a * (da - sum_c a da)
__global__ __launch_bounds__(256) void op_softmax_bwd_kernel(const float* a, const float* da, size_t rows, int n, float* dx) {
  const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s += a[r * n + c] * da[r * n + c];
  s = wave_sum(s);
  for (int c = lane; c < n; c += 64) dx[r * n + c] = a[r * n + c] * (da[r * n + c] - s);
}

// ---- dropout from the stateless stream -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void op_dropout_kernel(const float* x, size_t n, uint32_t first, uint32_t key, uint32_t thr24, float inv_keep,
                                                         float* out, const uint32_t* word = nullptr) {
  key = run_key(key, word);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = keep_bit((uint32_t)(first + i), key, thr24) ? x[i] * inv_keep : 0.f;
}

// addAnswerLossOp + addPredOp (model.py:593-612): per question the sparse softmax cross-entropy of its logits row and the
// argmax (first maximum, as tf.argmax); one wave per question.  dlogits = (softmax - onehot) * scale for the backward pass.
__global__ __launch_bounds__(256) void answer_loss_kernel(const float* logits, const int32_t* answers, int B, int A, float* loss_rows,
                                                          int32_t* pred, float* dlogits, float scale) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const float* p = logits + (size_t)b * A;
  float m = -INFINITY;
  int am = 0x7FFFFFFF;
  for (int c = lane; c < A; c += 64) {
    const float v = p[c];
    if (v > m) { m = v; am = c; }                       // strict: keeps the first maximum of this lane's columns
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    const float om = __shfl_xor(m, s, 64);
    const int oa = __shfl_xor(am, s, 64);
    if (om > m || (om == m && oa < am)) { m = om; am = oa; }
  }
  float z = 0.f;
  for (int c = lane; c < A; c += 64) z += expf(p[c] - m);
  z = wave_sum(z);
  const int ans = answers[b];
  const bool valid = ans >= 0 && ans < A;               // an out-of-range label gives a NaN loss (TF's GPU kernel) and no gradient
  if (lane == 0) {
    loss_rows[b] = valid ? (m + logf(z)) - p[ans] : NAN;   // -log softmax[ans]
    pred[b] = am;
  }
  if (dlogits) {
    const float inv = 1.0f / z;
    for (int c = lane; c < A; c += 64)
      dlogits[(size_t)b * A + c] = valid ? (expf(p[c] - m) * inv - (c == ans ? 1.f : 0.f)) * scale : 0.f;
  }
}

// dKB[b][n][c] (+)= att[b][n] * dinfo[b][c]: the knowledge-base gradient of ops.att2Smry
__global__ __launch_bounds__(256) void kb_attend_dkb_kernel(const float* att, const float* dinfo, size_t n, int N, int d, int accumulate,
                                                            float* dkb) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t row = i / d, b = row / N;
    const float v = att[row] * dinfo[b * d + i % d];
    dkb[i] = accumulate ? dkb[i] + v : v;
  }
}

inline unsigned op_grid(size_t n) {
  const size_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace macx
