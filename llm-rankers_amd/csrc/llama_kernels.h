// Decoder-only (Llama family) kernels for gfx950: rotary position embedding and causal grouped-query attention with
// head_dim = 128 (every Llama-2/3 size).  Everything else of the Llama forward - RMSNorm (folded), the QKV / O / gate-up /
// down projections (SwiGLU epilogue), the final-token head - runs on the kernels the T5 path already uses.
//
// Semantics restated from hf: models/llama/modeling_llama.py: apply_rotary_pos_emb :137-160 (rotate_half pairs element i
// with i + head_dim/2), eager_attention_forward :192-214 (scaling = head_dim**-0.5, causal mask, fp32 softmax),
// repeat_kv :180-189 (query head h reads kv head h / (n_heads / n_kv_heads)).
#pragma once
#include "common.h"
#include "xcd_map.h"

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
  int n_seq, nqb;        // attn_causal128_dma_kernel (1-D grid): sequences, query blocks of 32 NW of the longest sequence
  int ko;                // attn_causal128_dma_kernel, measurement builds only (-DRK_MEASURE): timing knock-outs, see ATCD_KO
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

// The same attention by the recipe of the long-sequence T5 kernel (attention.h: attn_enc_long_kernel), for d = 128 (round 5;
// engine option llama_attn_dma = 1, the default).  The kernel above stages every 64-key tile through registers - K row-major,
// V through sixteen 4-byte transposing LDS writes per thread - between two barriers that leave the memory pipe idle while the
// tile is computed: 220 TF/s on a 1 536-token prompt.  Here the K and V rows of a 64-key chunk arrive by LDS-DMA
// (global_load_lds_dwordx4) into one of two 32-KiB stages while the previous chunk is computed (ONE barrier per chunk, behind a
// wait that has had a whole chunk of compute to be satisfied); a 128-column row is kept as TWO 64-column images in the T5
// kernels' layout ([64 keys][64 halves], the eight 16-byte slots of a row permuted by ATTD_SWZ on the SOURCE address), so the K
// fragments are conflict-free ds_read_b128 and the V^T fragments come from ds_read_b64_tr_b16 (attd_issue_vt) - no transpose
// is ever written.  S^T = K Q^T as before (a lane owns one query column), log2-domain online softmax per chunk with the
// packed-fp32 / v_max3 helpers of attention.h, the two half-rows of a query (lane, lane ^ 32) exchanged by
// v_permlane32_swap; the 64 accumulator registers are rescaled only when some lane's running maximum moved (alpha == 1 for
// every lane otherwise: the same bits); context rows leave as 16-byte stores.  64 KiB of LDS, <= 256 VGPRs: two workgroups per
// CU.  grid = ceil(B n_kv / 8) x 8 x (n_heads / n_kv) x ceil(maxL / 128) workgroups (see the mapping below); 256 threads = 4 waves x 32 queries.
// A row's arithmetic depends on its own sequence only (its position, its keys in chunks of 64 in order): batch-independent.
// timing-only knock-outs (measurement builds: hipcc ... -DRK_MEASURE, loaded through RK_ENGINE_LIB; results are garbage):
// AttnCausalArgs::ko bit 0 no K / V DMA inside the chunk loop, 1 no score MFMAs, 2 no softmax (scale, maximum, exp, merge),
// 3 no P V, 4 no workgroup barrier per chunk, 5 no context stores, 6 return at once, 7 return after the prologue.  tools/llama_attn_ko.py
#ifdef RK_MEASURE
#define ATCD_KO(bit) ((p.ko >> (bit)) & 1)
#else
#define ATCD_KO(bit) 0
#endif
#define ATCD_KEYS 64
#define ATCD_IMG_HALFS (ATCD_KEYS * 64)            // one image: 8 KiB
#define ATCD_STAGE_HALFS (4 * ATCD_IMG_HALFS)      // K columns 0-63 | K columns 64-127 | V columns 0-63 | V columns 64-127
#define ATCD_LDS_BYTES (2 * ATCD_STAGE_HALFS * 2)
// NW waves x 32 queries per workgroup (4 or 8): a query row's arithmetic does not depend on NW - the same chunks of 64 keys in the
// same order, the same per-wave decisions - so the host may pick it freely (8: two waves per SIMD inside ONE workgroup, each
// K / V chunk fetched once for 256 queries; 4: twice the workgroups, shorter critical path of the last query block)
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_causal128_dma_kernel(AttnCausalArgs p) {
  constexpr int ATCD_QUERIES = 32 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char atcd_smem[];
  half_t* const sbuf = (half_t*)atcd_smem;
  // workgroup -> (sequence, head, query block), XCD-aware: consecutive workgroups go to the 8 XCDs in turn, each with its own
  // L2, so workgroup i belongs to the (sequence, kv head) group 8 (i / 8 / W) + i % 8 - all W = heads-per-group x query-blocks
  // workgroups that read one K / V pair (0.8 MB at 1.5k tokens) meet in ONE L2 instead of each XCD pulling every pair through the
  // fabric - and walks that group's query blocks from the last (most keys) to the first, the heads of a kv head side by side
  const int hpg = p.n_heads / p.n_kv, W = hpg * p.nqb;
  int grp, w;
  if (!xcd_decode((int)blockIdx.x, p.n_seq * p.n_kv, W, grp, w)) return;   // uniform for the whole block (xcd_map.h)
  const int b = grp / p.n_kv, kvh = grp % p.n_kv, h = kvh * hpg + w % hpg, qb = p.nqb - 1 - w / hpg;
  const int tok0 = p.seq_off[b];
  const int L = p.seq_off[b + 1] - tok0;
  const int Q0 = qb * ATCD_QUERIES;
  if (Q0 >= L) return;                                       // uniform for the whole block
  if (ATCD_KO(6)) return;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int q0 = Q0 + wave * 32;
  const bool active = q0 < L;                                // wave-uniform
  const half_t* const kbase = p.qkv + (size_t)(p.n_heads + kvh) * 128;
  const half_t* const vbase = p.qkv + (size_t)(p.n_heads + p.n_kv + kvh) * 128;
  const int last_q = min(Q0 + ATCD_QUERIES - 1, L - 1);
  const int nch = (last_q >> 6) + 1;                         // chunks this block of queries can see (causal)
  auto opaque_lane = [&]() {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
  };
  struct LaneCtx { int hh, l31, qpos, kfo0, vfo0; };
  auto lane_ctx = [&](int lane) {
    LaneCtx c;
    c.hh = lane >> 5; c.l31 = lane & 31; c.qpos = q0 + c.l31;
    c.kfo0 = c.l31 * 64 + ((c.hh ^ ATTD_SWZ(c.l31)) << 3);
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    c.vfo0 = (4 * c.hh + (i16 >> 2)) * 64 + (((2 * g1 + ((i16 & 3) >> 1)) ^ ((((i16 >> 3) & 1) << 2) | c.hh)) << 3) + 4 * (i16 & 1);
    return c;
  };
  // chunk ch -> stage st: 32 pieces of 64 sixteen-byte slots (8 per image), 32 / NW per wave; keys beyond the sequence are
  // clamped copies of its last row (never visible to a valid query: causal)
  auto issue_chunk = [&](int lane, int ch, int st) {
#pragma unroll
    for (int k = 0; k < 32 / NW; ++k) {
      const int pid = wave + NW * k, img = pid >> 3, sub = pid & 7;
      const int slot = sub * 64 + lane, r = slot >> 3, c = slot & 7;
      int key = ch * ATCD_KEYS + r;
      key = key < L ? key : L - 1;
      const char* hb = (const char*)((img < 2 ? kbase : vbase) + (img & 1) * 64);
      const unsigned off = ((unsigned)(tok0 + key) * (unsigned)p.ld + ((c ^ ATTD_SWZ(r)) << 3)) * 2u;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(hb + off),
                                       (__attribute__((address_space(3))) void*)(sbuf + st * ATCD_STAGE_HALFS + img * ATCD_IMG_HALFS + sub * 512),
                                       16, 0, 0);
    }
  };

  // ---- prologue: this lane's Q fragments (8 k16 steps), chunk 0 ----
  half8 qf[8];
  {
    const LaneCtx c = lane_ctx(opaque_lane());
    const int qrow = c.qpos < L ? c.qpos : L - 1;
    const half_t* qptr = p.qkv + (size_t)(tok0 + qrow) * p.ld + h * 128 + 8 * c.hh;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *(const half8*)(qptr + 16 * s);
  }
  issue_chunk(opaque_lane(), 0, 0);
  // (the Q fragments are "used" here so that the compiler's wait for these tracked loads sits in front of the chunk loop and
  // not at their first MFMA inside it, where it would drain the DMA queue of every chunk)
  asm volatile("" :: "v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "v"(qf[4]), "v"(qf[5]), "v"(qf[6]), "v"(qf[7]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ATCD_KO(7)) return;

  f32x16 o[4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[f][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;                         // l_run: this lane's half of the row sum (the halves meet at the end)

  // scores of the chunk: 8 k16 steps (4 per image) x two 32-key halves; the K fragments of step i + 1 requested before the MFMAs of step i
  auto qk_chunk = [&](const LaneCtx& c, const half_t* kst, f32x16& s0, f32x16& s1) {
    const half_t* kb_ = kst + c.kfo0;
    half8 kf[2][2];
    auto fetch = [&](int i, half8 (&d)[2]) {
      const half_t* a = kb_ + (i >> 2) * ATCD_IMG_HALFS + ((c.kfo0 ^ ((i & 3) << 4)) - c.kfo0);
      d[0] = *(const half8*)a;
      d[1] = *(const half8*)(a + 32 * 64);
    };
    fetch(0, kf[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i + 1 < 8) fetch(i + 1, kf[(i + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (i == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][0], qf[0], z, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][1], qf[0], z, 0, 0, 0);
      } else {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i & 1][0], qf[i], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i & 1][1], qf[i], s1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // O^T += V^T P^T for one 64-column image of the chunk's V rows (the T5 kernels' pv_tile: transposing reads a k16 step ahead
  // of their MFMAs, counted lgkmcnt); oa / ob: output columns 0-31 / 32-63 of the image
  auto pv_image = [&](const LaneCtx& c, const half_t* vimg, const unsigned (&pp)[16], f32x16& oa, f32x16& ob) {
    const unsigned vb0 = (unsigned)(size_t)(const __attribute__((address_space(3))) half_t*)vimg;
    unsigned va[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) va[i] = vb0 + 2u * (unsigned)(c.vfo0 ^ (16 * i));
    half4 v[2][4];
    auto step = [&](auto gc, half4 (&d)[4]) {
      constexpr int g = decltype(gc)::value, sub = g >> 1, sp = g & 1;
      const attd_u32x4 pu = {pp[8 * sub + 4 * sp], pp[8 * sub + 4 * sp + 1], pp[8 * sub + 4 * sp + 2], pp[8 * sub + 4 * sp + 3]};
      const half8 pf = __builtin_bit_cast(half8, pu);
      if constexpr (g < 3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
      __builtin_amdgcn_sched_barrier(0);
      const half8 vf0 = {d[0][0], d[0][1], d[0][2], d[0][3], d[1][0], d[1][1], d[1][2], d[1][3]};
      const half8 vf1 = {d[2][0], d[2][1], d[2][2], d[2][3], d[3][0], d[3][1], d[3][2], d[3][3]};
      oa = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf0, pf, oa, 0, 0, 0);
      ob = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf1, pf, ob, 0, 0, 0);
    };
    using std::integral_constant;
    attd_issue_vt<0>(v[0], va);
    attd_issue_vt<1>(v[1], va);
    step(integral_constant<int, 0>{}, v[0]);
    attd_issue_vt<2>(v[0], va);
    step(integral_constant<int, 1>{}, v[1]);
    attd_issue_vt<3>(v[1], va);
    step(integral_constant<int, 2>{}, v[0]);
    step(integral_constant<int, 3>{}, v[1]);
  };

  // one chunk of a wave that sees some of its keys.  MASK: the chunk reaches past the wave's first query (the diagonal)
  auto chunk_body = [&](auto maskc, int ch, int st) {
    constexpr bool MASK = decltype(maskc)::value;
    const half_t* kst = sbuf + st * ATCD_STAGE_HALFS;
    f32x16 s0, s1;
    if (!ATCD_KO(1)) qk_chunk(lane_ctx(opaque_lane()), kst, s0, s1);
    else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      asm volatile("" : "+v"(s0), "+v"(s1));
    }
    __builtin_amdgcn_sched_barrier(0);
    float tmax = -1e30f;
    if (!ATCD_KO(2)) {
      const LaneCtx c1 = lane_ctx(opaque_lane());
      const int key_base = ch * ATCD_KEYS + 4 * c1.hh;       // register r <-> key key_base + (r & 3) + 8 (r >> 2) (+ 32 for s1)
      const f32x2 sc = {p.scale_log2e, p.scale_log2e};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t0 = f32x2{s0[r], s0[r + 1]} * sc, t1 = f32x2{s1[r], s1[r + 1]} * sc;
        s0[r] = t0[0]; s0[r + 1] = t0[1]; s1[r] = t1[0]; s1[r + 1] = t1[1];
        if (MASK) {
#pragma unroll
          for (int j = r; j < r + 2; ++j) {
            const int key0 = key_base + (j & 3) + 8 * (j >> 2);
            s0[j] = key0 <= c1.qpos ? s0[j] : -1e30f;
            s1[j] = key0 + 32 <= c1.qpos ? s1[j] : -1e30f;
          }
        }
        tmax = attn_max3(tmax, s0[r], s1[r]);
        tmax = attn_max3(tmax, s0[r + 1], s1[r + 1]);
      }
    }
    // (the chunk's first key is visible to every query of the wave, so the row maximum is a real score)
    const float m_c = attn_row_max(tmax);
    const float m_new = attn_max3(m_run, m_c, m_c);
    float psum = 0.f;
    if (!ATCD_KO(2)) attn_tile_exp(s0, s1, m_new, psum);
    unsigned pp[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const half2v a = {(half_t)s0[2 * i], (half_t)s0[2 * i + 1]};
      const half2v b2 = {(half_t)s1[2 * i], (half_t)s1[2 * i + 1]};
      pp[i] = __builtin_bit_cast(unsigned, a);
      pp[8 + i] = __builtin_bit_cast(unsigned, b2);
    }
    // online merge (m_run starts at -1e30: alpha = 0 and the zero accumulators stay zero on the first chunk); the 64 accumulator
    // registers are only touched when some lane's maximum moved - multiplying by 1.0f changes no bit
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (!ATCD_KO(2) && __builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
      const f32x2 a2 = {alpha, alpha};
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 x = f32x2{o[f][r], o[f][r + 1]} * a2;
          o[f][r] = x[0]; o[f][r + 1] = x[1];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const LaneCtx c3 = lane_ctx(opaque_lane());
    if (!ATCD_KO(3)) pv_image(c3, kst + 2 * ATCD_IMG_HALFS, pp, o[0], o[1]);
    __builtin_amdgcn_sched_barrier(0);
    if (!ATCD_KO(3)) pv_image(c3, kst + 3 * ATCD_IMG_HALFS, pp, o[2], o[3]);
    __builtin_amdgcn_sched_barrier(0);
  };

  using T = std::integral_constant<bool, true>; using F = std::integral_constant<bool, false>;
  for (int ch = 0; ch < nch; ++ch) {
    const int st = ch & 1;
    // the next chunk travels while this one is computed: its stage was last read before the barrier that ended chunk ch - 1
    if (ch + 1 < nch && !ATCD_KO(0)) issue_chunk(opaque_lane(), ch + 1, st ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    if (active && ch * ATCD_KEYS <= q0 + 31) {               // else: every key of the chunk lies after every query of this wave
      if (ch * ATCD_KEYS + ATCD_KEYS - 1 > q0) chunk_body(T{}, ch, st); else chunk_body(F{}, ch, st);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // chunk ch + 1 has landed (it had this whole chunk to do so)
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): every LDS read of this chunk retired
    if (!ATCD_KO(4)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- context rows: normalise, pack, 16-byte stores (v_permlane32_swap pairs the half-waves' quads into whole octets) ----
  if (active) {
    const float l_row = attn_row_sum(l_run);
    const float inv = 1.0f / l_row;                          // a query always sees its own key: l_row > 0
    const LaneCtx c4 = lane_ctx(opaque_lane());
    half_t* dst = p.ctx + (size_t)(tok0 + (c4.qpos < L ? c4.qpos : L - 1)) * p.ldctx + h * 128 + 16 * c4.hh;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      unsigned pk[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        half4 a;
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = f2h_sat(o[f][4 * q + j] * inv);
        const auto au = __builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, a);
        pk[2 * q] = au[0]; pk[2 * q + 1] = au[1];
      }
      const auto x0 = __builtin_amdgcn_permlane32_swap(pk[0], pk[4], false, false);
      const auto x1 = __builtin_amdgcn_permlane32_swap(pk[1], pk[5], false, false);
      const auto y0 = __builtin_amdgcn_permlane32_swap(pk[2], pk[6], false, false);
      const auto y1 = __builtin_amdgcn_permlane32_swap(pk[3], pk[7], false, false);
      const attd_u32x4 lo = {x0[0], x1[0], x0[1], x1[1]};
      const attd_u32x4 hi = {y0[0], y1[0], y0[1], y1[1]};
      if (c4.qpos < L && !ATCD_KO(5)) {
        *(attd_u32x4*)(dst + 32 * f) = lo;
        *(attd_u32x4*)(dst + 32 * f + 8) = hi;
      }
    }
  }
}
