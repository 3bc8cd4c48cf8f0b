// HBM-bound helper kernels (embedding gather, RMSNorm, head rows, argmax, cross-entropy) for gfx950.
// One wave (64 lanes) per row, 16-byte vector accesses, shuffle reductions.
#pragma once
#include "common.h"

// hf: modeling_t5.py:644,678 (embed_tokens): hidden[t][:] = (fp32) E[ids[t]][:]; no scaling, dropout = identity.
// Folded-norm form of the encoder (xraw != nullptr): the row also goes out as fp16 x xs (A operand of the first QKV GEMM)
// with its RMSNorm row factor rsqrt(mean(x^2) + eps) / xs (GemmArgs::rowscale).
__global__ __launch_bounds__(256) void embed_gather_kernel(const int* __restrict__ ids, const half_t* __restrict__ table,
                                                           float* __restrict__ out, int n_rows, int d, int vocab,
                                                           half_t* __restrict__ xraw, float* __restrict__ rowscale,
                                                           float xs, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  int id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // host validates; clamp keeps a bad id from faulting
  const half_t* src = table + (size_t)id * d;
  float* dst = out + (size_t)row * d;
  float ss = 0.f;
  for (int c = lane * 8; c < d; c += 64 * 8) {
    const half8 v = *(const half8*)(src + c);
    f32x4 a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    f32x4 b = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    *(f32x4*)(dst + c) = a;
    *(f32x4*)(dst + c + 4) = b;
    if (xraw) {
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float x = (float)v[j]; ss += x * x; o[j] = f2h_sat(x * xs); }
      *(half8*)(xraw + (size_t)row * d + c) = o;
    }
  }
  if (xraw) {
    ss = wave_sum(ss);
    if (lane == 0) rowscale[row] = rsqrtf(ss / (float)d + eps) / xs;
  }
}

// Folded RMSNorm, statistics step: the fp32-residual GEMM epilogue left the sums of squares of every new row per
// 64-column block in ssq [n_rows, nb] (gemm.h); rowscale[m] = rsqrt(sum_j ssq[m][j] / d + eps) / xs, blocks added in
// increasing order.  hf: modeling_t5.py:59-72 computes the same fp32 mean of squares.
__global__ __launch_bounds__(256) void rowscale_kernel(const float* __restrict__ ssq, float* __restrict__ rowscale,
                                                       int n_rows, int nb, int d, float eps, float xs) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= n_rows) return;
  rowscale[row] = rk_row_factor(ssq + (size_t)row * nb, nb, d, eps, xs);
}

// hf: modeling_t5.py:59-72 (T5LayerNorm): y = w * x * rsqrt(mean(x^2) + eps); fp32 statistics, fp16 result
// (the GEMM input).  row_map (optional) gathers source rows: out row r reads x row row_map[r].
// One wave per row; the row (d <= 64*4*NV floats) stays in registers so HBM is read exactly once.
template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      half_t* __restrict__ out, const int* __restrict__ row_map,
                                                      int n_rows, int d, float eps, float out_scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const float* src = x + (size_t)(row_map ? row_map[row] : row) * d;
  f32x4 v[NV];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < d) {
      v[i] = *(const f32x4*)(src + c);
      ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)d + eps) * out_scale;
  half_t* dst = out + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < d) {
      const f32x4 g = *(const f32x4*)(w + c);
      half4 o = {f2h_sat(v[i][0] * rs * g[0]), f2h_sat(v[i][1] * rs * g[1]), f2h_sat(v[i][2] * rs * g[2]),
                 f2h_sat(v[i][3] * rs * g[3])};
      *(half4*)(dst + c) = o;
    }
  }
}

// Final-token logit extraction for the label / yes-no rows only (hf: modeling_t5.py:1044-1047 lm_head, but
// just the n_out vocabulary rows the rankers read: ref pointwise.py:120-121, setwise.py:186).
// out[b][j] = dot(x[b], head[out_ids[j]]); one wave per (b, j).
__global__ __launch_bounds__(256) void head_rows_kernel(const half_t* __restrict__ x, const half_t* __restrict__ head,
                                                        const int* __restrict__ out_ids, float* __restrict__ out,
                                                        int n_seq, int n_out, int d) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (idx >= n_seq * n_out) return;
  const int b = idx / n_out, j = idx % n_out;
  const half_t* xr = x + (size_t)b * d;
  const half_t* hr = head + (size_t)out_ids[j] * d;
  float s = 0.f;
  for (int c = lane * 8; c < d; c += 512) {
    const half8 a = *(const half8*)(xr + c);
    const half8 w = *(const half8*)(hr + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)w[e];
  }
  s = wave_sum(s);
  if (lane == 0) out[idx] = s;
}

// Greedy decoding step, second half of the fused head: the weight-streaming head GEMM (EPI_ARGMAX_F32) left, per 32-column
// block, the row maximum and its first column; one workgroup per row picks the first index of the overall maximum
// (torch.argmax tie rule; hf: generation/utils.py greedy).
__global__ __launch_bounds__(256) void argmax_blocks_kernel(const float* __restrict__ bval, const int* __restrict__ bidx,
                                                            int n_blocks, int* __restrict__ out) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int row = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = tid; c < n_blocks; c += 256) {
    const float v = bval[(size_t)row * n_blocks + c];
    const int i = bidx[(size_t)row * n_blocks + c];
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { sv[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    out[row] = bi;
  }
}

// QLM score (ref: llmrankers/pointwise.py:77-79) from the fused head (gemm.h: EPI_LSE_F32): stats [rows, nblk] = (block max, sum exp(x - block max)), xlab [rows] =
// the label's logit.  out[b] = -sum_t ( logsumexp_t - xlab[b, t] ), logsumexp_t = M + log(sum_blocks s * exp(m - M)).
// One block per sequence; the blocks of a position are merged in a fixed order, the positions summed in order (deterministic).
__global__ __launch_bounds__(256) void qlm_lse_kernel(const float2* __restrict__ stats, int nblk, const float* __restrict__ xlab,
                                                      int n_pos, float* __restrict__ out) {
  __shared__ float sred[4];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float total = 0.f;
  for (int t = 0; t < n_pos; ++t) {
    const size_t row = (size_t)b * n_pos + t;
    const float2* src = stats + row * nblk;
    float mx = -INFINITY;
    for (int c = tid; c < nblk; c += 256) mx = fmaxf(mx, src[c].x);
    mx = wave_max(mx);
    if (lane == 0) sred[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    __syncthreads();
    float se = 0.f;
    for (int c = tid; c < nblk; c += 256) { const float2 v = src[c]; se += v.y * expf(v.x - mx); }
    se = wave_sum(se);
    if (lane == 0) sred[wave] = se;
    __syncthreads();
    se = sred[0] + sred[1] + sred[2] + sred[3];
    __syncthreads();
    total += (mx + logf(se)) - xlab[row];
  }
  if (tid == 0) out[b] = -total;
}
