// Decoder-only (Llama family) kernels for gfx950: rotary position embedding and causal grouped-query attention with
// head_dim = 128 (every Llama-2/3 size).  Everything else of the Llama forward - RMSNorm (folded), the QKV / O / gate-up /
// down projections (SwiGLU epilogue), the final-token head - runs on the kernels the T5 path already uses.
//
// Semantics restated from hf: models/llama/modeling_llama.py: apply_rotary_pos_emb :137-160 (rotate_half pairs element i
// with i + head_dim/2), eager_attention_forward :192-214 (scaling = head_dim**-0.5, causal mask, fp32 softmax),
// repeat_kv :180-189 (query head h reads kv head h / (n_heads / n_kv_heads)).
#pragma once
#include "common.h"

// In place on the fused QKV buffer [T, ld]: the first n_rot heads of a row (all query heads, then all key heads) are
// rotated by the row's position.  cos / sin: [max_pos, 64] fp32 (the two halves of HF's table are equal).
// One workgroup per token; a thread takes 8 consecutive pairs of one head: 16-byte accesses.
__global__ __launch_bounds__(256) void rope128_kernel(half_t* __restrict__ qkv, const int* __restrict__ pos,
                                                      const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                      int ld, int n_rot) {
  const int t = blockIdx.x;
  const int p = pos[t];
  half_t* row = qkv + (size_t)t * ld;
  const float* cr = cos_t + (size_t)p * 64;
  const float* sr = sin_t + (size_t)p * 64;
  for (int c = threadIdx.x; c < n_rot * 8; c += 256) {
    const int head = c >> 3, i0 = (c & 7) * 8;
    half_t* x = row + head * 128 + i0;
    const half8 a = *(const half8*)x, b = *(const half8*)(x + 64);
    half8 oa, ob;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float co = cr[i0 + j], si = sr[i0 + j];
      const float x1 = (float)a[j], x2 = (float)b[j];
      oa[j] = f2h_sat(x1 * co - x2 * si);          // q * cos + rotate_half(q) * sin, first half: -x2
      ob[j] = f2h_sat(x2 * co + x1 * si);          // second half: +x1
    }
    *(half8*)x = oa;
    *(half8*)(x + 64) = ob;
  }
}

struct AttnCausalArgs {
  const half_t* qkv;     // [T, ld]: q heads at column 0, k heads at n_heads*128, v heads at (n_heads + n_kv)*128
  half_t* ctx;           // [T, ldctx] (n_heads * 128 columns)
  const int* seq_off;    // [B+1]
  int ld, ldctx, n_heads, n_kv;
  float scale_log2e;     // head_dim**-0.5 * log2(e): the softmax runs in the log2 domain
};

// Flash-style causal attention, d = 128.  grid = (ceil(maxL / 128), n_heads, B); 256 threads = 4 waves x 32 queries.
// Per 64-key tile: K rows and V TRANSPOSED are staged in LDS; S^T = K Q^T by MFMA 32x32x16 (A = K rows, B = Q^T) so a
// lane owns ONE query column and the online-softmax state is per-lane scalars; the fp16 probabilities are already in
// B-operand position for O^T = V^T P^T (four 32-row d fragments).  Tiles above the diagonal are skipped.
#define ATC_KSTR 136   // sK row stride in halfs (272 B: 16-B aligned)
#define ATC_VSTR 68    // sVt row stride in halfs (136 B: 8-B aligned)
__global__ __launch_bounds__(256) void attn_causal128_kernel(AttnCausalArgs p) {
  __shared__ __attribute__((aligned(16))) half_t sK[64 * ATC_KSTR];
  __shared__ __attribute__((aligned(16))) half_t sVt[128 * ATC_VSTR];
  const int b = blockIdx.z, h = blockIdx.y, qt = blockIdx.x;
  const int tok0 = p.seq_off[b];
  const int L = p.seq_off[b + 1] - tok0;
  if (qt * 128 >= L) return;   // uniform for the whole block
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int hh = lane >> 5, l31 = lane & 31;
  const int kvh = h / (p.n_heads / p.n_kv);
  const half_t* kbase = p.qkv + (size_t)(p.n_heads + kvh) * 128;
  const half_t* vbase = p.qkv + (size_t)(p.n_heads + p.n_kv + kvh) * 128;
  const int q0 = qt * 128 + wave * 32;
  const bool wave_active = q0 < L;
  const int qpos = q0 + l31;
  const int qrow = qpos < L ? qpos : L - 1;
  half8 qf[8];
  {
    const half_t* qptr = p.qkv + (size_t)(tok0 + qrow) * p.ld + h * 128 + 8 * hh;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *(const half8*)(qptr + 16 * s);
  }
  f32x16 o[4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[f][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const int last_q = min(qt * 128 + 127, L - 1);
  const int nkt = (last_q >> 6) + 1;                 // key tiles this block of queries can see (causal)
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();                                   // the previous tile's fragments are read
    // ---- stage K (row-major) and V^T (key pairs) of keys kt*64 .. kt*64+63; rows beyond L are clamped copies (masked) ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i, row = c >> 4, cc = c & 15;
      const int key = min(kt * 64 + row, L - 1);
      *(half8*)(sK + row * ATC_KSTR + cc * 8) = *(const half8*)(kbase + (size_t)(tok0 + key) * p.ld + cc * 8);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i, kp = c >> 4, cc = c & 15;
      const int k0 = min(kt * 64 + 2 * kp, L - 1), k1 = min(kt * 64 + 2 * kp + 1, L - 1);
      const half8 v0 = *(const half8*)(vbase + (size_t)(tok0 + k0) * p.ld + cc * 8);
      const half8 v1 = *(const half8*)(vbase + (size_t)(tok0 + k1) * p.ld + cc * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const half2v pr = {v0[j], v1[j]};
        *(half2v*)(sVt + (cc * 8 + j) * ATC_VSTR + 2 * kp) = pr;
      }
    }
    __syncthreads();
    if (!wave_active || kt * 64 > q0 + 31) continue;    // this wave's queries see none of these keys
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const half8 k0 = *(const half8*)(sK + l31 * ATC_KSTR + 16 * s + 8 * hh);
      const half8 k1 = *(const half8*)(sK + (32 + l31) * ATC_KSTR + 16 * s + 8 * hh);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[s], s1, 0, 0, 0);
    }
    // ---- online softmax (log2 domain); lane = query qpos, register r <-> key key_base + (r&3) + 8(r>>2) (+32 for s1) ----
    const int key_base = kt * 64 + 4 * hh;
    const bool need_mask = kt * 64 + 63 > q0 || kt * 64 + 63 >= L;   // the tile touches the diagonal or the sequence end
    float tmax = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] *= p.scale_log2e;
      s1[r] *= p.scale_log2e;
      if (need_mask) {
        const int key0 = key_base + (r & 3) + 8 * (r >> 2);
        s0[r] = (key0 <= qpos && key0 < L) ? s0[r] : -1e30f;
        s1[r] = (key0 + 32 <= qpos && key0 + 32 < L) ? s1[r] : -1e30f;
      }
      tmax = fmaxf(tmax, fmaxf(s0[r], s1[r]));
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
      s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_new);
      psum += s0[r] + s1[r];
    }
    psum += __shfl_xor(psum, 32);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[f][r] *= alpha;
    m_run = m_new;
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        half8 pf;
#pragma unroll
        for (int i = 0; i < 8; ++i) pf[i] = (half_t)(sub == 0 ? s0[8 * sp + i] : s1[8 * sp + i]);
        const int kb = sub * 32 + 16 * sp + 4 * hh;   // keys kb..kb+3 and kb+8..kb+11 <-> regs 8sp..8sp+7
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const half_t* vr = sVt + (f * 32 + l31) * ATC_VSTR + kb;
          const half4 v0 = *(const half4*)vr, v1 = *(const half4*)(vr + 8);
          const half8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          o[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[f], 0, 0, 0);
        }
      }
    }
  }
  if (wave_active && qpos < L) {
    // a query always sees its own key, so l_run > 0
    const float inv = 1.0f / l_run;
    half_t* dst = p.ctx + (size_t)(tok0 + qpos) * p.ldctx + h * 128;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        half4 a;
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = f2h_sat(o[f][4 * q + j] * inv);
        *(half4*)(dst + f * 32 + 8 * q + 4 * hh) = a;
      }
  }
}
