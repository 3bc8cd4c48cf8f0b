// T5 attention kernels for gfx950 (d_kv = 64 only; every public T5 up to xl uses 64).
//
// Semantics restated from hf: models/t5/modeling_t5.py:144-173 with scaling = 1.0 (:196-197):
//   P = softmax(Q K^T + bias[h, bucket(j - i)] + mask),  ctx = P V
// bias comes from block 0's table, shared by all layers of a stack (:739-742); cross-attention has zero bias
// (:337-342).  Ragged batches: every sequence only ever sees its own keys, which equals HF's additive
// finfo.min padding mask (exp underflows to exactly 0).
#pragma once
#include "common.h"
#include <type_traits>
#include "xcd_map.h"

#ifndef ATTD_Q_EARLY
#define ATTD_Q_EARLY 1   // the next item's Q loads: 1 = right behind the score tiles (a whole softmax in front of the counted V wait), 0 = behind the V barrier
#endif
struct AttnEncArgs {
  const half_t* qkv;     // [T, ld]: q at column 0, k at column I, v at column 2I (output of the fused QKV GEMM)
  half_t* ctx;           // [T, ldctx]
  const int* seq_off;    // [B+1] token offsets of the packed batch
  const float* bias_lut; // [H][RK_LUT_N]
  int ld, ldctx, I;
  int heads_per_wg;      // DMA kernel only: (sequence, head) items per wave group, walked in turn
  int ko;                // DMA kernel, measurement builds only: timing knock-outs (ATTD_KO)
  int n_seq;             // DMA kernel only: sequences in the batch
  float* trace;          // DMA kernel, measurement builds only: phase time stamps of one workgroup (ATTD_STAMP), else nullptr
  int skip_long;         // tiled kernel: sequences longer than ATT_ROW_MAXL are left to attn_enc_long_kernel (round 5)
  int n_heads, nqb;      // attn_enc_long_kernel (1-D grid): heads, query blocks of 32 NW of the longest sequence
  int xcd_map;           // attn_enc_long_kernel: 1 = the workgroups of a (sequence, head) pair on one XCD, 0 = dealt over all eight
};

// Flash-style encoder self-attention.  grid = (ceil(maxL/128), H, B), 256 threads = 4 waves x 32 queries.
// Per 64-key tile: K tile row-major and V tile TRANSPOSED are staged in LDS;  S^T = K Q^T by MFMA 32x32x16
// (A = K rows, B = Q^T) so each lane owns ONE query column: the running max / sum / rescale are per-lane
// scalars and the fp16 P values are already in MFMA B-operand position for O^T = V^T P^T (A = V^T rows).
// The k-slot <-> key assignment is the same permutation for P (B operand) and V^T (A operand), so no
// cross-lane data movement is needed anywhere except one xor-32 shuffle for the row max and row sum.
// Pipeline: the global loads of tile t+1 are issued into registers before tile t is computed and written to the
// other LDS buffer afterwards (double buffer, ONE barrier per tile, memory latency hidden behind the MFMAs);
// V^T is written as key pairs (ds_write_b32); the output goes through LDS so stores are whole 128-byte rows.
#define ATT_KSTR 72   // sK row stride in halfs (144 B: 16-B aligned, conflict-free b128 reads)
#define ATT_VSTR 68   // sVt row stride in halfs (136 B: 8-B aligned, conflict-free b64 reads)

// Online-softmax step for one 64-key tile, shared by both encoder attention kernels so that they stay bit-identical
// (a sequence must score the same whichever kernel its batch selects).  s0 / s1: raw q.k of this lane's query against
// keys key_base + (r&3) + 8(r>>2) (+32 for s1); on return they hold the unnormalised probabilities relative to the new
// running maximum.  Everything is in the log2 domain: t = s * log2(e) + bias * log2(e) (one FMA, the tables are
// pre-multiplied) and p = exp2(t - m) (one v_exp_f32).  MASK: the tile contains keys >= L (only the last one can).
// FIRST: tile 0 - no previous maximum, no rescale of the (zero) output accumulators.  bias(r, sub) returns the table
// entry for register r of s0 (sub = 0) or s1 (sub = 1).
#define ATT_LOG2E 1.4426950408889634f
// CHUNKED: the table reads are fenced into groups of four registers (8 values in flight instead of 32) for kernels that
// run close to their register budget; the arithmetic and the order of every sum are the same.
// VALU economy (the DMA kernel is bound by its VALU slots: ~8 instructions per score): the FMA and the subtraction run as
// packed fp32 pairs (v_pk_fma_f32 / v_pk_add_f32: the same IEEE operations, two per slot) and the running maximum as
// v_max3_f32 through inline asm - fmaxf() makes the compiler canonicalise both inputs first (v_max x, x, x: three
// instructions per pair of scores instead of one); no NaN can reach here, and max is exact in any order.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float attn_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// First half of a tile's softmax: t = s * log2(e) + bias (keys >= L: -1e30) in place, and the running maximum of this lane's
// scores (tmax is read and updated: several tiles may share one maximum).
// A lane and its partner (lane ^ 32) hold the two halves of a query row's scores: v_permlane32_swap hands each the other's
// value without LDS (a and b are {own, partner} in one half-wave and {partner, own} in the other - max and + do not care).
// (inline asm: the builtin form with a float on both sides came back with ONE register for both results - clang 19, gfx950 -
// and the hazard recogniser does not look inside asm, hence the explicit wait states in front of the swap)
__device__ __forceinline__ void attn_swap32(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float attn_row_max(float x) {
  float a = x, b = x;
  attn_swap32(a, b);
  return attn_max3(a, b, b);
}
__device__ __forceinline__ float attn_row_sum(float x) {
  float a = x, b = x;
  attn_swap32(a, b);
  return a + b;
}
// attn_tile_exp with the scale and a CONSTANT bias folded into the exponent's argument: e = exp2(s log2(e) + c), c = bias - m
__device__ __forceinline__ void attn_tile_fma_exp(f32x16& s0, f32x16& s1, float c, float& psum) {
  const f32x2 c2 = {c, c}, l2e = {ATT_LOG2E, ATT_LOG2E};
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const f32x2 d0 = __builtin_elementwise_fma(f32x2{s0[r], s0[r + 1]}, l2e, c2), d1 = __builtin_elementwise_fma(f32x2{s1[r], s1[r + 1]}, l2e, c2);
    s0[r] = __builtin_amdgcn_exp2f(d0[0]); s0[r + 1] = __builtin_amdgcn_exp2f(d0[1]);
    s1[r] = __builtin_amdgcn_exp2f(d1[0]); s1[r + 1] = __builtin_amdgcn_exp2f(d1[1]);
    psum += s0[r] + s1[r];
    psum += s0[r + 1] + s1[r + 1];
  }
}
template <bool MASK, class BiasFn, bool CHUNKED = false>
__device__ __forceinline__ void attn_tile_bias_max(f32x16& s0, f32x16& s1, float& tmax, int key_base, int L, BiasFn bias) {
  // registers 4g .. 4g+3 of s0 / s1 hold keys tile0 + 8g (+32) .. +8 (both half-waves): a group that lies wholly inside the
  // sequence needs no select - decided per group on wave-uniform values, as a real scalar branch (the empty asm keeps the
  // compiler from turning it back into selects).  At L = 184 one group of eight is cut: 8 compare / select pairs instead
  // of 64 on every last tile.  Same values either way.
  const int tile0 = MASK ? (__builtin_amdgcn_readfirstlane(key_base) & ~63) : 0;
  // CHUNKED: the table entries of register group q + 1 are requested before group q is used (LDS reads return in order, so
  // the counted wait in front of group q leaves them in flight): 16 values in flight, one exposed LDS round trip per tile
  const f32x2 l2e = {ATT_LOG2E, ATT_LOG2E};
  f32x2 bq[2][2][2];                                  // [buffer][register pair of the group][s0 / s1]
  auto fetch_group = [&](int buf, int r0) {
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) bq[buf][jp][sub] = f32x2{bias(r0 + 2 * jp, sub), bias(r0 + 2 * jp + 1, sub)};
  };
  if (CHUNKED) fetch_group(0, 0);
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    if (CHUNKED && (r & 3) == 0) {
      __builtin_amdgcn_sched_barrier(0);
      if (r < 12) fetch_group(((r >> 2) + 1) & 1, r + 4);
      __builtin_amdgcn_sched_barrier(0);
    }
    const f32x2 b0 = CHUNKED ? bq[(r >> 2) & 1][(r >> 1) & 1][0] : f32x2{bias(r, 0), bias(r + 1, 0)};
    const f32x2 b1 = CHUNKED ? bq[(r >> 2) & 1][(r >> 1) & 1][1] : f32x2{bias(r, 1), bias(r + 1, 1)};
    const f32x2 t0 = __builtin_elementwise_fma(f32x2{s0[r], s0[r + 1]}, l2e, b0);
    const f32x2 t1 = __builtin_elementwise_fma(f32x2{s1[r], s1[r + 1]}, l2e, b1);
    s0[r] = t0[0]; s0[r + 1] = t0[1]; s1[r] = t1[0]; s1[r + 1] = t1[1];
    if (MASK && (r & 3) == 2) {
      const int g = r >> 2;
      if (tile0 + 8 * g + 8 > L) {
        asm volatile("");
#pragma unroll
        for (int j = 4 * g; j < 4 * g + 4; ++j) s0[j] = key_base + (j & 3) + 8 * g < L ? s0[j] : -1e30f;
      }
      if (tile0 + 8 * g + 40 > L) {
        asm volatile("");
#pragma unroll
        for (int j = 4 * g; j < 4 * g + 4; ++j) s1[j] = key_base + (j & 3) + 8 * g + 32 < L ? s1[j] : -1e30f;
      }
#pragma unroll
      for (int j = 4 * g; j < 4 * g + 4; ++j) tmax = attn_max3(tmax, s0[j], s1[j]);
    }
    if (!MASK) { tmax = attn_max3(tmax, s0[r], s1[r]); tmax = attn_max3(tmax, s0[r + 1], s1[r + 1]); }
  }
  if (CHUNKED) __builtin_amdgcn_sched_barrier(0);
}

// Second half: p = exp2(t - m) in place; psum += this lane's 32 probabilities, pairwise (s0[r] + s1[r]) in register order.
__device__ __forceinline__ void attn_tile_exp(f32x16& s0, f32x16& s1, float m, float& psum) {
  const f32x2 m2 = {m, m};
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const f32x2 d0 = f32x2{s0[r], s0[r + 1]} - m2, d1 = f32x2{s1[r], s1[r + 1]} - m2;
    s0[r] = __builtin_amdgcn_exp2f(d0[0]); s0[r + 1] = __builtin_amdgcn_exp2f(d0[1]);
    s1[r] = __builtin_amdgcn_exp2f(d1[0]); s1[r + 1] = __builtin_amdgcn_exp2f(d1[1]);
    psum += s0[r] + s1[r];
    psum += s0[r + 1] + s1[r + 1];
  }
}

// Online-softmax step of one 64-key tile (sequences longer than ATT_ROW_MAXL keys; shorter ones take the whole-row form below).
template <bool MASK, bool FIRST, class BiasFn, bool CHUNKED = false>
__device__ __forceinline__ void attn_tile_softmax(f32x16& s0, f32x16& s1, f32x16& o0, f32x16& o1, float& m_run, float& l_run,
                                                  int key_base, int L, BiasFn bias) {
  float tmax = -1e30f;
  attn_tile_bias_max<MASK, BiasFn, CHUNKED>(s0, s1, tmax, key_base, L, bias);
  tmax = attn_row_max(tmax);
  const float m_new = FIRST ? tmax : attn_max3(m_run, tmax, tmax);
  float psum = 0.f;
  attn_tile_exp(s0, s1, m_new, psum);
  psum = attn_row_sum(psum);
  if (FIRST) {
    l_run = psum;
  } else {
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
  }
  m_run = m_new;
}

// WHOLE-ROW softmax (sequences of at most ATT_ROW_MAXL = 192 keys, i.e. three key tiles - the pointwise prompts): all score
// tiles of a query row are formed first, ONE maximum and ONE sum are taken over the row (attn_tile_bias_max over the tiles in
// order, one exchange with the partner lane; attn_tile_exp over the tiles in order, one exchange), and the unnormalised
// probabilities go through P V without any rescaling.  Against the online form this drops two of three cross-lane
// exchanges per row, the alpha exponentials and the rescale of the output accumulators, and lets a wave issue the MFMAs
// of all its score tiles back to back.  Which form a sequence takes depends on ITS length only, so its bits never depend on
// the batch: the DMA kernel holds the three tiles in registers; the tiled kernel (batches that also hold a longer sequence)
// walks the key tiles of such a sequence twice - maximum first, then probabilities - and reproduces the same operations.
#define ATT_ROW_MAXL 192

// Compiled for two workgroups per CU (256-register budget): they overlap each other's load / softmax / MFMA phases.  (Rounds 2-4
// also shipped a key-split form for single long prompts and 1- / 3-per-CU builds; attn_enc_long_kernel superseded them in round
// 5 and they were removed in round 6 - git history and profiles/r05_attn_split_B1_B8.txt keep the measurements.)
#define ATT_GROUP_BYTES (2 * 64 * ATT_KSTR * 2 + 2 * 64 * ATT_VSTR * 2)
__global__ __launch_bounds__(256, 2) void attn_enc_kernel(AttnEncArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char sGroup[1][ATT_GROUP_BYTES];
  __shared__ float sLut[RK_LUT_N + 3];
  const int b = blockIdx.z, h = blockIdx.y, qt = blockIdx.x;
  const int tok0 = p.seq_off[b];
  const int L = p.seq_off[b + 1] - tok0;
  if (qt * 128 >= L) return;   // uniform for the whole block
  if (p.skip_long && L > ATT_ROW_MAXL) return;
  constexpr int grp = 0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;      // within the group
  half_t (*sK)[64 * ATT_KSTR] = (half_t (*)[64 * ATT_KSTR])sGroup[grp];
  half_t (*sVt)[64 * ATT_VSTR] = (half_t (*)[64 * ATT_VSTR])(sGroup[grp] + 2 * 64 * ATT_KSTR * 2);
  const int hh = lane >> 5, l31 = lane & 31;
  for (int i = threadIdx.x; i < RK_LUT_N; i += 256) sLut[i] = p.bias_lut[h * RK_LUT_N + i] * ATT_LOG2E;
  const int q0 = qt * 128 + wave * 32;
  const bool wave_active = q0 < L;
  const int qpos = q0 + l31;
  const int qrow = qpos < L ? qpos : L - 1;
  half8 qf[4];
  {
    const half_t* qptr = p.qkv + (size_t)(tok0 + qrow) * p.ld + h * 64 + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const half8*)(qptr + 16 * s);
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  const int nkt = (L + 63) >> 6;
  const int n_iter = nkt, kt_begin = 0, kt_end = nkt;
  // staging roles: K: thread takes rows (tid>>3) and 32+(tid>>3), chunk tid&7; V: key pair tid>>3, chunk tid&7
  const int srow = tid >> 3, scc = tid & 7;
  half8 rk0, rk1, rv0, rv1;
  auto load_tile = [&](int kt) {
    const int ka = min(kt * 64 + srow, L - 1), kb2 = min(kt * 64 + 32 + srow, L - 1);
    const half_t* base = p.qkv + p.I + h * 64 + scc * 8;
    rk0 = *(const half8*)(base + (size_t)(tok0 + ka) * p.ld);
    rk1 = *(const half8*)(base + (size_t)(tok0 + kb2) * p.ld);
    const int va = min(kt * 64 + 2 * srow, L - 1), vb = min(kt * 64 + 2 * srow + 1, L - 1);
    rv0 = *(const half8*)(base + p.I + (size_t)(tok0 + va) * p.ld);
    rv1 = *(const half8*)(base + p.I + (size_t)(tok0 + vb) * p.ld);
  };
  auto store_tile = [&](int buf) {
    *(half8*)(sK[buf] + srow * ATT_KSTR + scc * 8) = rk0;
    *(half8*)(sK[buf] + (32 + srow) * ATT_KSTR + scc * 8) = rk1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const half2v pr = {rv0[j], rv1[j]};
      *(half2v*)(sVt[buf] + (scc * 8 + j) * ATT_VSTR + 2 * srow) = pr;
    }
  };
  // Whole-row softmax (ATT_ROW_MAXL above) for a short sequence that shares its batch with a longer one: the key tiles are
  // walked TWICE - first pass: scores, bias and the row maximum only; second pass: the same scores again (same MFMA inputs,
  // same bits), probabilities against that maximum, P V without rescaling - which reproduces the DMA kernel's operations
  // and their order exactly.
  const bool row_form = L <= ATT_ROW_MAXL;
  const int n_total = row_form ? 2 * nkt : n_iter;
  auto tile_of = [&](int it) { return row_form ? (it >= nkt ? it - nkt : it) : kt_begin + it; };
  float row_tmax = -1e30f, row_m = 0.f, row_psum = 0.f;
  if (kt_begin < kt_end) {
    load_tile(kt_begin);
    store_tile(0);
  }
  __syncthreads();
  for (int it = 0; it < n_total; ++it) {
    const int kt = tile_of(it);
    const int cur = it & 1;
    const bool more = kt_begin < kt_end && it + 1 < (row_form ? n_total : kt_end - kt_begin);
    if (more) load_tile(tile_of(it + 1));       // in flight while this tile is computed
    if (wave_active && kt_begin < kt_end && kt < kt_end) {
      const half_t* kbuf = sK[cur];
      const half_t* vbuf = sVt[cur];
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const half8 k0 = *(const half8*)(kbuf + l31 * ATT_KSTR + 16 * s + 8 * hh);
        const half8 k1 = *(const half8*)(kbuf + (32 + l31) * ATT_KSTR + 16 * s + 8 * hh);
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[s], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[s], s1, 0, 0, 0);
      }
      {
        const int key_base = kt * 64 + 4 * hh;
        const bool last = kt == nkt - 1;
        // T5's bias is constant beyond +-max_distance (one bucket): a tile whose keys are all at least RK_LUT_R away from
        // all 32 queries of this wave adds the same table value to every score - no table reads, no index math (17 of the
        // 23 tiles of a 1.4k-token prompt).  Same fmaf inputs as the table path: bit-identical.
        const bool far_l = kt * 64 + 63 - q0 <= -RK_LUT_R, far_r = kt * 64 - (q0 + 31) >= RK_LUT_R;
        auto soft = [&](auto bias) {
          using BF = decltype(bias);
          if (row_form) {
            float tm = -1e30f;                      // second pass: the maximum is already known
            float& tmax = it < nkt ? row_tmax : tm;
            if (last) attn_tile_bias_max<true, BF>(s0, s1, tmax, key_base, L, bias);
            else attn_tile_bias_max<false, BF>(s0, s1, tmax, key_base, L, bias);
            if (it == nkt - 1) row_m = attn_row_max(row_tmax);
            if (it >= nkt) attn_tile_exp(s0, s1, row_m, row_psum);
          } else if (kt == kt_begin) {
            if (last) attn_tile_softmax<true, true>(s0, s1, o0, o1, m_run, l_run, key_base, L, bias);
            else attn_tile_softmax<false, true>(s0, s1, o0, o1, m_run, l_run, key_base, L, bias);
          } else {
            if (last) attn_tile_softmax<true, false>(s0, s1, o0, o1, m_run, l_run, key_base, L, bias);
            else attn_tile_softmax<false, false>(s0, s1, o0, o1, m_run, l_run, key_base, L, bias);
          }
        };
        if (far_l || far_r) {
          const float cb = far_l ? sLut[0] : sLut[RK_LUT_N - 1];
          soft([&](int, int) { return cb; });
        } else {
          soft([&](int r, int sub) {
            int rel = key_base + (r & 3) + 8 * (r >> 2) + 32 * sub - qpos;
            rel = rel < -RK_LUT_R ? -RK_LUT_R : (rel > RK_LUT_R ? RK_LUT_R : rel);
            return sLut[rel + RK_LUT_R];
          });
        }
      }
      if (!row_form || it >= nkt) {                  // (first pass of the whole-row form: no P V)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          half8 pf;
#pragma unroll
          for (int i = 0; i < 8; ++i) pf[i] = (half_t)(sub == 0 ? s0[8 * sp + i] : s1[8 * sp + i]);
          const int kb = sub * 32 + 16 * sp + 4 * hh;   // keys kb..kb+3 and kb+8..kb+11 <-> regs 8sp..8sp+7
          {
            const half_t* vr = vbuf + l31 * ATT_VSTR + kb;
            const half4 v0 = *(const half4*)vr, v1 = *(const half4*)(vr + 8);
            const half8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o0, 0, 0, 0);
          }
          {
            const half_t* vr = vbuf + (32 + l31) * ATT_VSTR + kb;
            const half4 v0 = *(const half4*)vr, v1 = *(const half4*)(vr + 8);
            const half8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o1, 0, 0, 0);
          }
        }
      }
      }
    }
    if (more) store_tile(cur ^ 1);                // buffer cur^1 was last read before the previous barrier
    __syncthreads();
  }
  if (row_form) l_run = attn_row_sum(row_psum);
  // all waves are past the last barrier: reuse sK[0] to turn per-lane 8-byte pieces into whole context rows
  if (wave_active) {
    const float inv = 1.0f / l_run;
    half_t* st = sK[0] + wave * (16 * ATT_KSTR);   // 2 passes of 16 query rows x 64 d per wave (4 x 16 rows = 64 rows)
#pragma unroll
    for (int half_i = 0; half_i < 2; ++half_i) {
      if ((l31 >> 4) == half_i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int d = 8 * q + 4 * hh;
          half4 a, c;
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] = f2h_sat(o0[4 * q + j] * inv); c[j] = f2h_sat(o1[4 * q + j] * inv); }
          *(half4*)(st + (l31 & 15) * ATT_KSTR + d) = a;
          *(half4*)(st + (l31 & 15) * ATT_KSTR + 32 + d) = c;
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 8) {
        const int row = r0 + (lane >> 3), ch = lane & 7;
        const int qq = q0 + half_i * 16 + row;
        if (qq < L)
          *(half8*)(p.ctx + (size_t)(tok0 + qq) * p.ldctx + h * 64 + ch * 8) = *(const half8*)(st + row * ATT_KSTR + ch * 8);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Short-sequence kernel (L <= ATT_ROW_MAXL = 192 for every sequence of the batch: the pointwise prompts) - the production path.
// History: rounds 1-2 ran a register-staged kernel (K and V global -> VGPR -> LDS, V through a scalar 16-bit transpose, one
// lock-step workgroup per CU, three workgroup barriers per head, online softmax per 64-key tile): 262 us per launch at
// 320 x 184 tokens, 8 % MFMA busy, 29 % of the kernel's HBM floor (profiles/r02d_*).  This kernel replaced it in round 3
// (the old ones are gone: two forms of the same arithmetic are two things to keep bit-identical):
//   * K and V rows go global -> LDS by DMA (global_load_lds_dwordx4), row-major, no staging registers and no LDS stores;
//     the bank-conflict swizzle is applied to the per-lane SOURCE address (the DMA image is lane-linear) and serves both
//     the ds_read_b128 K fragments and the transposing V reads;
//   * V^T fragments come straight out of the row-major image with ds_read_b64_tr_b16 (each 16-lane group reads a
//     [4 keys][16 d] block and every lane receives one d column of it);
//   * three row buffers rotate (K_h, V_h, K_h+1): the next head's K lands while this head is computed, V_h is fetched at
//     the head boundary and awaited - a COUNTED vmcnt - only when the row's probabilities are ready: two barriers per
//     head, neither behind an exposed memory round trip;
//   * WHOLE-ROW softmax (ATT_ROW_MAXL above): the 96 score registers of a query row are formed by 24 back-to-back MFMAs,
//     one maximum / one sum per row, P packed to fp16 as it is formed, P V without rescaling;
//   * the context rows leave the registers directly: v_permlane32_swap pairs the 8-byte pieces of the two half-waves into
//     16-byte stores (no LDS staging, no barrier);
//   * the launch is PERSISTENT: one workgroup per CU, the (sequence, head) items dealt out in contiguous runs per wave group
//     (5120 items at 320 sequences x 16 heads: 10 per group), prefetching across sequence boundaries (see the kernel).
// The tiled kernel above reproduces the same operations for such a sequence when its batch also holds a longer one.
#define ATTD_ROWS 192
// timing-only knock-outs of this kernel (measurement builds: hipcc ... -DRK_MEASURE, loaded through RK_ENGINE_LIB; results are
// garbage): AttnEncArgs::ko bit 0 no K / V DMA inside the head loop, 1 no score MFMAs, 2 no softmax, 4 no P V, 5 no context
// stores.  tools/attn_dma_check.py --ko=<mask>.  Round 3, 320 x 184: all five off 58 of 140 us - what a launch costs before any
// work: five rounds of workgroups with their prologues, the barriers and the Q / table loads of every head.
// ATTD_STAMP(k): wave `wave + 6 grp` of workgroup 7 records the 100 MHz wall clock (ticks since its start) at phase boundary k
// of every item: p.trace[(wave12 * 16 + item) * 16 + k]  (engine option attn_trace, rk_debug_read("attn_trace"))
#ifdef RK_MEASURE
#define ATTD_KO(bit) ((p.ko >> (bit)) & 1)
#define ATTD_STAMP(k) do { if (p.trace && blockIdx.x == 7 && n < 16) { const long long t_ = wall_clock64(); \
    if ((threadIdx.x & 63) == 0) p.trace[((wave + 6 * grp) * 16 + n) * 16 + (k)] = (float)(int)(t_ - attd_t0); } } while (0)
#else
#define ATTD_KO(bit) 0
#define ATTD_STAMP(k) do { } while (0)
#endif
#define ATTD_BUF_HALFS (ATTD_ROWS * 64)
#define ATTD_LUT_N (2 * ATTD_ROWS)
#define ATTD_LDS_BYTES (3 * ATTD_BUF_HALFS * 2 + 2 * ATTD_LUT_N * 4)
typedef __fp16 attd_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned attd_u32x4 __attribute__((ext_vector_type(4)));
#define ATTD_SWZ(r) (((((r) >> 1) & 1) << 2) | ((((r) >> 3) & 1) << 1) | (((r) >> 2) & 1))

// the four transposing reads of k16 step G of a key tile (keys 16 G .. 16 G + 16): va = byte addresses of this lane's share
// for {o0 first key quad, o0 second, o1 first, o1 second} at the tile's first key; see pv_tile
template <int G>
__device__ __forceinline__ void attd_issue_vt(half4 (&d)[4], const unsigned (&va)[4]) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d[0]) : "v"(va[0]), "n"(G * 2048));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d[1]) : "v"(va[1]), "n"(G * 2048 + 1024));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d[2]) : "v"(va[2]), "n"(G * 2048));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d[3]) : "v"(va[3]), "n"(G * 2048 + 1024));
}

#ifndef ATTD_MINW
#define ATTD_MINW 3
#endif
// NG = 2 (the production form): a 768-thread workgroup runs TWO such six-wave groups side by side, each on its own items
// and its own 75 KiB of LDS.  Why not two 384-thread workgroups per CU, which LDS (2 x 75 KiB) and registers (144 -> three
// waves per SIMD) allow and the occupancy API promises: the dispatcher places the six waves of a workgroup 2,2,1,1 over
// the four SIMDs from a start of its own choosing, a second workgroup then needs exactly the complementary 1,1,2,2, and
// it does not look for it - tools/probes/probe_resid.hip: 384-thread workgroups with >= 160 VGPRs run ONE per CU (the
// kernel measured the same at 256 and at 512 workgroups).  Twelve waves of ONE workgroup always fit three per SIMD.  The
// two groups share only the workgroup barriers (two per item in every form of the item body, whatever the item's sequence
// length - so groups on different sequences still meet the same barriers).
// (Tried and dropped on the whole-row form, round 3, each bit-identical and within +-1 % or worse: three issue priorities
// for the three waves of a SIMD (s_setprio), the next head's Q loads issued right after the score tiles instead of after
// the softmax, the context stores deferred behind the next head's DMA issues so that the counted V wait does not also
// wait for them: 138 against 132 us.)
// (Tried and dropped, end of round 3: running the second group HALF AN ITEM behind the first - it passes one workgroup
// barrier before its first item, the first group one after its last, so that one group's softmax meets the other's P V
// instead of its softmax; correct by construction and bit-identical, 114 -> 121 us per launch: the extra half item is paid
// and nothing comes back - the two groups' phases are not what limits an item; its ~190 KB of loads, DMA and stores per CU
// at the ~30 KB/us a CU gets while all 256 pull are more than half of its 11.7 us.)
// (Tried and dropped, round 3: a full-prefetch form of the one-group kernel - four row buffers, K AND V of the next head
// landing a whole head ahead, the next Q rows in registers of their own, context rows stored a head later, ONE barrier per
// head, 178 VGPRs - bit-identical, 177 us per launch against 161 for the one-group and 145 for the two-group form: the
// time of a head is the instruction latency of its own waves, not the memory waits the form removes.)
template <int NG>
__global__ __launch_bounds__(NG * 384, ATTD_MINW) void attn_enc_dma_kernel(AttnEncArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char attd_smem_all[];
  const int grp = NG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >= 384));
  unsigned char* const attd_smem = attd_smem_all + grp * ATTD_LDS_BYTES;
  half_t* const sbuf = (half_t*)attd_smem;                                  // three [192][64] row images
  float* const sLut = (float*)(attd_smem + 3 * ATTD_BUF_HALFS * 2);         // two bias tables (this head's / the next one's)
  const int wave = __builtin_amdgcn_readfirstlane(((int)threadIdx.x - grp * 384) >> 6);
  // PERSISTENT: the launch is one workgroup per CU; the (sequence, head) items t = sequence x H + head are dealt out in
  // contiguous runs of p.heads_per_wg per GROUP, and a group walks its run exactly as it used to walk the heads of one
  // sequence - the next item's K rows, Q rows and table entry travel while the current one is computed - so a sequence
  // boundary costs nothing and the workgroup prologue (first K / Q / table behind an exposed round trip, measured 3.4 us)
  // is paid once per launch instead of once per five heads.  Every group of a workgroup walks as many items as group 0 (the
  // same barriers for all waves); a group short of items repeats the last item of the batch without storing.
  struct Item { int h, tok0, L; };                                          // wave-uniform
  const int H = p.I >> 6, total = p.n_seq * H, per = p.heads_per_wg;
  const int t_wg = blockIdx.x * NG * per;
  const int n_walk = min(per, total - t_wg);
  if (n_walk <= 0) return;                                                  // uniform for the whole block
  const int t_first = t_wg + grp * per;
  const int n_own = max(0, min(per, total - t_first));
  // item n of this group = (sequence pb, head ph), stepped head by head from the group's first item (one division per
  // launch; everything stays in scalar registers: a per-item division would run on the VALU, put the sequence index in a
  // VGPR and turn the two offset loads into vector loads - whose wait drains the DMA queue)
  int pb = __builtin_amdgcn_readfirstlane(min(t_first, total - 1) / H), ph = min(t_first, total - 1) - pb * H;
  // The two token offsets of an item come by an explicit s_load_dwordx2 (inline asm, awaited with lgkmcnt(0) by item_done):
  // behind the kernel's own global stores the compiler no longer trusts the scalar cache and would fetch them with a VECTOR
  // load, whose vmcnt(0) drains the DMA queue in the middle of a head.  (seq_off is never written by this kernel.  An extra
  // SMEM operation in flight can only make the compiler's counted lgkmcnt waits stricter.)
  typedef int attd_i32x2 __attribute__((ext_vector_type(2)));
  auto item_issue = [&](attd_i32x2& raw, int& h_out) {     // request the item at (pb, ph), then step (clamped to the batch's last item)
    const int* src = p.seq_off + pb;
    asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(raw) : "s"(src) : "memory");
    h_out = ph;
    if (ph + 1 < H) ++ph;
    else if (pb + 1 < p.n_seq) { ++pb; ph = 0; }
  };
  auto item_done = [&](attd_i32x2& raw, int h) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(raw) :: "memory");
    Item it;
    it.h = h; it.tok0 = raw[0]; it.L = raw[1] - raw[0];
    return it;
  };
  const int q0 = wave * 32;
#ifdef RK_MEASURE
  const long long attd_t0 = wall_clock64();
#endif
  // the item being computed (updated by the item loop; the lambdas below read them)
  int tok0 = 0, L = 0;

  // Everything a lane derives from its lane number (fragment addresses, DMA source offsets ...) is ~25 registers.  Kept
  // across the head loop they would not fit beside the softmax at three waves per SIMD, so every head re-derives them from
  // an opaque copy of the lane number (a dozen integer instructions against ~1200 of softmax).
  struct LaneCtx {
    int hh, l31, tid, qpos;
    int lut_idx;         // entry of the head's bias table this thread converts for the next head
    int lut_q;           // sLut[lut_q + key - 4hh] = bias(key - qpos) * log2(e)
    int kfo0, vfo0;      // fragment offsets (halfs), see below
  };
  auto lane_ctx = [&](int lane) {
    LaneCtx c;
    c.hh = lane >> 5; c.l31 = lane & 31; c.tid = wave * 64 + lane; c.qpos = q0 + c.l31;
    int li = c.tid - (ATTD_ROWS - 1);                                       // table entry tid <-> key - query = tid - 191
    c.lut_idx = (li < -RK_LUT_R ? -RK_LUT_R : (li > RK_LUT_R ? RK_LUT_R : li)) + RK_LUT_R;
    c.lut_q = (ATTD_ROWS - 1) - c.qpos + 4 * c.hh;
    // K fragment of k16 step s: row l31 (+32, + 64 kt) of the GEMM-style image, chunk (2s + hh) ^ swizzle = chunk0 ^ 2s
    c.kfo0 = c.l31 * 64 + ((c.hh ^ ATTD_SWZ(c.l31)) << 3);
    // V: this lane's share of the transposing read - group g = lane>>4 covers d columns 16 (g&1) .. +16 of keys 4 hh .. +4;
    // lane i16 of the group supplies the address of key i16>>2, columns 4 (i16&3) .. +4.  key = 16 x + 8 sec + 4 hh +
    // (i16 >> 2): swizzle bit 2 = i16 bit 3, bit 0 = hh, bit 1 = sec (second read of a fragment: ^16 halfs); o1 (d + 32)
    // is chunk bit 2 flipped (^32 halfs)
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    c.vfo0 = (4 * c.hh + (i16 >> 2)) * 64 + (((2 * g1 + ((i16 & 3) >> 1)) ^ ((((i16 >> 3) & 1) << 2) | c.hh)) << 3) + 4 * (i16 & 1);
    return c;
  };
  // BYTE offset of this lane's part of its query row of item `it` from the head's column base (rows beyond the sequence:
  // its last row; an empty sequence: row 0 of whatever follows - loaded, never used)
  auto q_off = [&](const LaneCtx& c, const Item& it) {
    const int row = c.qpos < it.L ? c.qpos : max(it.L - 1, 0);
    return ((unsigned)(it.tok0 + row) * (unsigned)p.ld + 8 * c.hh) * 2u;
  };
  // DMA piece i of this wave fills LDS slots (4 wave + i) * 64 + lane: row r = slot >> 3, 16-B chunk c = slot & 7 (lane-
  // linear image); the global chunk it fetches is c ^ swizzle(r).
  // One swizzle serves both images: chunk ^= f(r), f = (r bit 1) << 2 | (r bit 3) << 1 | (r bit 2).  Over any 16
  // consecutive rows f takes each value twice (rows 2j, 2j+1, which sit in different halves of a 256-B bank row): the
  // ds_read_b128 K fragments are conflict-free like the GEMM's; and bit 2 follows key bit 1, so the four key rows of a
  // transposing V read occupy the four 64-B quarters of the bank row.
  // (every wave always issues its four pieces, whatever the sequence length: the counted vmcnt waits below rely on it)
  auto issue_rows = [&](int lane, int which, int buf, const Item& it) {   // which: 1 = K columns, 2 = V columns of the fused qkv rows
    const char* hb = (const char*)(p.qkv + which * p.I + it.h * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int slot = (4 * wave + i) * 64 + lane, r = slot >> 3, c = slot & 7;
      const unsigned off = ((unsigned)(it.tok0 + (r < it.L ? r : max(it.L - 1, 0))) * (unsigned)p.ld + ((c ^ ATTD_SWZ(r)) << 3)) * 2u;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(hb + off),
                                       (__attribute__((address_space(3))) void*)(sbuf + buf * ATTD_BUF_HALFS + (4 * wave + i) * 512),
                                       16, 0, 0);
    }
  };

  half8 qf[4];
  float lutreg, qwarm;
  // The Q rows and the table entry of the NEXT head travel through inline asm (invisible to the waitcnt pass, which would
  // otherwise drain the DMA queue at their first use): unconditional, issued and awaited in the same loop iteration, and
  // the wait's "+v" operands order every use behind it (tests/test_isa_guards.py).  Sixteen more live registers do not fit
  // beside the 96 score registers of a row at three waves per SIMD, so the Q fragments of the next head are loaded straight
  // into qf once the row's probabilities are packed (they have P V to arrive); what is issued a head ahead is one dword
  // per row (qwarm) that pulls the row's 128-byte line into L2, so that the real loads are short.
  auto issue_lut = [&](const LaneCtx& c, const Item& it) {
    const char* hb = (const char*)(p.qkv + it.h * 64);
    asm volatile("global_load_dword %0, %1, %2" : "=&v"(qwarm) : "v"(q_off(c, it)), "s"(hb) : "memory");
    const float* lb = p.bias_lut + it.h * RK_LUT_N;
    asm volatile("global_load_dword %0, %1, %2" : "=&v"(lutreg) : "v"(c.lut_idx * 4), "s"(lb) : "memory");
  };
  auto issue_q = [&](const LaneCtx& c, const Item& it) {
    const char* hb = (const char*)(p.qkv + it.h * 64);
    const unsigned qo = q_off(c, it);
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(qf[0]) : "v"(qo), "s"(hb) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=&v"(qf[1]) : "v"(qo), "s"(hb) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=&v"(qf[2]) : "v"(qo), "s"(hb) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:96" : "=&v"(qf[3]) : "v"(qo), "s"(hb) : "memory");
  };
  auto wait_q = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(lutreg), "+v"(qwarm) :: "memory");
  };

  // Scores of a whole row: NKT key tiles x 4 k16 steps, two MFMAs each (keys 0-31 / 32-63 of the tile).  The K fragments of
  // step i + 1 are requested before the MFMAs of step i (the sched_barriers pin that order; the compiler's own waitcnt pass
  // then emits the counted waits).
  auto qk_row = [&](auto nktc, const LaneCtx& c, const half_t* kbuf, f32x16 (&s)[decltype(nktc)::value][2]) {
    constexpr int NKT = decltype(nktc)::value;
    const half_t* kb_ = kbuf + c.kfo0;
    half8 kf[2][2];
    auto fetch = [&](int i, half8 (&d)[2]) {               // step i = 4 kt + k16 step
      const half_t* a = kb_ + (i >> 2) * 64 * 64 + ((c.kfo0 ^ ((i & 3) << 4)) - c.kfo0);
      d[0] = *(const half8*)a;
      d[1] = *(const half8*)(a + 32 * 64);
    };
    fetch(0, kf[0]);
#pragma unroll
    for (int i = 0; i < 4 * NKT; ++i) {
      if (i + 1 < 4 * NKT) fetch(i + 1, kf[(i + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int kt = i >> 2, ks = i & 3;
      if (ks == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        s[kt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i & 1][0], qf[ks], z, 0, 0, 0);
        s[kt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i & 1][1], qf[ks], z, 0, 0, 0);
      } else {
        s[kt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i & 1][0], qf[ks], s[kt][0], 0, 0, 0);
        s[kt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i & 1][1], qf[ks], s[kt][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // first half of the row softmax for key tile kt: bias, mask, running maximum (attn_tile_bias_max)
  auto bm_tile = [&](auto maskc, const LaneCtx& c, const float* lut, int kt, f32x16& s0, f32x16& s1, float& tmax) {
    const int key_base = kt * 64 + 4 * c.hh;
    // one opaque index per tile: all 32 table reads take their immediate offsets against ONE address register
    int lqi = c.lut_q + kt * 64;
    asm volatile("" : "+v"(lqi));
    const float* lq = lut + lqi;
    auto bias = [&](int r, int sub) { return lq[(r & 3) + 8 * (r >> 2) + 32 * sub]; };
    using BF = decltype(bias);
    attn_tile_bias_max<decltype(maskc)::value, BF, true>(s0, s1, tmax, key_base, L, bias);
  };
  // P V of one key tile.  The V^T fragments are read with inline-asm ds_read_b64_tr_b16 (the builtin form makes the waitcnt
  // pass drain the whole DMA queue - vmcnt(0) - in front of the first read: it cannot tell the K rows still in flight for
  // the next head from the V rows being read).  Four reads (one k16 step: first / second key quad for o0 and o1) are
  // issued one step ahead of the MFMAs that consume them; LDS operations return in order, so lgkmcnt(4) retires the
  // older four.  The sched_barriers around the tile keep compiler-issued LDS reads out of the counted span.
  auto pv_tile = [&](auto firstc, const LaneCtx& c, const half_t* vbuf, int kt, const unsigned (&pp)[16], f32x16& o0, f32x16& o1) {   // pp: P as packed halfs, s0's 16 then s1's
    constexpr bool FIRST_TILE = decltype(firstc)::value;   // key tile 0: the accumulators start from zero (C = 0 in the first MFMAs)
    const unsigned vb0 = (unsigned)(size_t)(const __attribute__((address_space(3))) half_t*)(vbuf + kt * 64 * 64) ;
    unsigned va[4];                                        // byte addresses: o0 first / second key quad, o1 first / second
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
      if constexpr (FIRST_TILE && g == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf0, pf, z, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf1, pf, z, 0, 0, 0);
      } else {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf0, pf, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf1, pf, o1, 0, 0, 0);
      }
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
  auto opaque_lane = [&]() {   // recomputed at every use (two instructions): no register holds the lane number across a phase
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
  };

  // ---- prologue: K rows, Q rows and table of the group's first item ---------------------------------------------------
  attd_i32x2 raw0, raw1;
  int ih0, ih1;
  item_issue(raw0, ih0);
  item_issue(raw1, ih1);
  Item cur = item_done(raw0, ih0), nxt = item_done(raw1, ih1);
  {
    const int lane = opaque_lane();
    const LaneCtx c = lane_ctx(lane);
    issue_rows(lane, 1, 0, cur);
    __builtin_amdgcn_sched_barrier(0);
    issue_lut(c, cur);
    issue_q(c, cur);
    wait_q();
    sLut[c.tid] = lutreg * ATT_LOG2E;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the table entry is written
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }

  int kb = 0, vb = 1, nb = 2;                             // buffers of K_h, V_h, K_h+1
  // NKT: the sequence's key tiles (1 .. 3) as a compile-time constant - the head body is straight-line code, only its last
  // tile carries the key mask
  // ACTIVE: this wave has query rows in THIS item (wave-uniform per item - items span sequences of different lengths in the
  // persistent launch, so `q0 < L` is re-evaluated per item; the two forms are separate straight-line bodies: guarded by a
  // run-time `if`, values defined in one guarded block and used in the next stayed allocated in between).  INVARIANT the
  // counted vmcnt(4) below relies on: the ACTIVE and the !ACTIVE body issue the SAME loads behind V_h - the line touch + table
  // entry (2), four K rows of the next item - and the four Q loads of the next item behind the wait, in any mix of lengths
  // (tests/test_isa_guards.py counts them per body; the ragged GPU test mixes L <= 32, 64 < L <= 128 and L = 192 in one launch)
  auto head = [&](auto lastc, auto nktc, auto activec, int n) {
    constexpr bool LAST = decltype(lastc)::value;
    constexpr int NKT = decltype(nktc)::value;
    constexpr bool ACTIVE = decltype(activec)::value;
    const int h = cur.h;
    const bool store_ok = n < n_own;
    const int lane = opaque_lane();
    const LaneCtx c = lane_ctx(lane);
    const half_t* kbuf = sbuf + kb * ATTD_BUF_HALFS;
    const half_t* vbuf = sbuf + vb * ATTD_BUF_HALFS;
    const float* lut = sLut + (n & 1) * ATTD_LUT_N;
    // everyone is past the barrier that ended head h-1: its K and V buffers are free
    ATTD_STAMP(0);
    if (!ATTD_KO(0)) issue_rows(lane, 2, vb, cur);
    if constexpr (!LAST) {
      __builtin_amdgcn_sched_barrier(0);
      issue_lut(c, nxt);
    }
    __builtin_amdgcn_sched_barrier(0);
    ATTD_STAMP(1);
    // ---- scores of the whole row: up to three key tiles, MFMAs back to back ----
    f32x16 s[NKT][2];
    if constexpr (ACTIVE) {
      if (!ATTD_KO(1)) qk_row(nktc, c, kbuf, s);
      else {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[kt][0][r] = 0.25f * r; s[kt][1][r] = 0.5f; }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#if ATTD_Q_EARLY
    if constexpr (!LAST) {
      issue_q(lane_ctx(opaque_lane()), nxt);              // qf has been dead since the score tiles (every wave, active or not)
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    ATTD_STAMP(2);
    // ---- whole-row softmax (attention.h: ATT_ROW_MAXL): one maximum, one sum, P packed to fp16 as it is formed ----
    // (every phase re-derives what it needs of the lane context from a fresh opaque lane number: nothing but the score
    // registers lives across the phases)
    unsigned pp[NKT][16];
    float l_row = 1.f;
    if constexpr (ACTIVE) {
      const LaneCtx c1 = lane_ctx(opaque_lane());
      float tmax = -1e30f;
      if (!ATTD_KO(2)) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          if (kt == NKT - 1) bm_tile(std::integral_constant<bool, true>{}, c1, lut, kt, s[kt][0], s[kt][1], tmax);
          else bm_tile(std::integral_constant<bool, false>{}, c1, lut, kt, s[kt][0], s[kt][1], tmax);
        }
      }
      const float m_row = attn_row_max(tmax);
      ATTD_STAMP(3);
      float psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
          if (!ATTD_KO(2)) attn_tile_exp(s[kt][0], s[kt][1], m_row, psum);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const half2v a = {(half_t)s[kt][0][2 * i], (half_t)s[kt][0][2 * i + 1]};
            const half2v b2 = {(half_t)s[kt][1][2 * i], (half_t)s[kt][1][2 * i + 1]};
            pp[kt][i] = __builtin_bit_cast(unsigned, a);
            pp[kt][8 + i] = __builtin_bit_cast(unsigned, b2);
          }
        }
      l_row = attn_row_sum(psum);
    }
    __builtin_amdgcn_sched_barrier(0);
    ATTD_STAMP(4);
    // The next item's K rows go out HERE, not at the top of the item beside V: the phase stamps of a measurement build
    // (tools/attn_trace.py) showed the twelve waves of a CU spending 1.6 us (first group) / 3.6 us (second, queued behind the
    // first) of a 12-us item just ISSUING their 96 DMA instructions - the address path takes ~30 KB/us per CU - and the first
    // group then waiting for the second at the barrier.  V alone (needed first) is half of that; K_i+1 is not needed before the
    // next item and queues behind the softmax instead: 117 -> 115 us per launch.  (The counted wait below is unchanged: V is
    // still the oldest.  Letting the second group issue its V rows behind its score tiles as well, so that the two groups'
    // bursts do not meet, changed nothing: the second group's waves are the younger ones and lose every arbitration, not
    // just this one - the first group waits ~3 us of every item for them at the barriers.)
    if constexpr (!LAST) {
      const int lane4 = opaque_lane();
      if (!ATTD_KO(0)) issue_rows(lane4, 1, nb, nxt);
    }
    __builtin_amdgcn_sched_barrier(0);
    // V_h: this wave's four DMA instructions are awaited with a COUNTED wait that holds whatever the order in which plain
    // loads and LDS-DMA loads retire relative to each other (round 4: they do not share one order - gemm.h, row factors).  In
    // flight behind V_h: the line touch + the table entry (two plain loads, issued an item ago) and the four K rows of the next
    // item (LDS-DMA: behind V_h in the DMA's own order).  At most 4 outstanding <=> at least 6 of those 10 retired, and the K rows
    // cannot be among them before V_h is: V_h has landed.  (Until round 4 the next item's four Q loads went out in front of this
    // wait and it read vmcnt(10): four plain loads overtaking V_h would have satisfied it.  They now follow the barrier below -
    // still a whole P V ahead of their use.  The context stores of the previous head are older, at most 4, and can only make
    // the wait stricter.)  Every wave, active or not, issues the same loads: see the ACTIVE note at the top of `head`.
    if constexpr (LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    ATTD_STAMP(5);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#if !ATTD_Q_EARLY
    if constexpr (!LAST) {
      issue_q(lane_ctx(opaque_lane()), nxt);              // qf has been dead since the score tiles (every wave, active or not)
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    ATTD_STAMP(6);
    f32x16 o0, o1;                                        // (local to the head: nothing of them lives across the softmax)
    if constexpr (ACTIVE) {
      const LaneCtx c3 = lane_ctx(opaque_lane());
      if (!ATTD_KO(4)) {
        pv_tile(std::integral_constant<bool, true>{}, c3, vbuf, 0, pp[0], o0, o1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NKT > 1) pv_tile(std::integral_constant<bool, false>{}, c3, vbuf, 1, pp[1], o0, o1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NKT > 2) pv_tile(std::integral_constant<bool, false>{}, c3, vbuf, 2, pp[2], o0, o1);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = __builtin_bit_cast(float, pp[0][r]); o1[r] = __builtin_bit_cast(float, pp[NKT - 1][r]); }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    ATTD_STAMP(7);
    unsigned pk[2][8];                                    // the context row pieces of this lane as packed halfs
    if constexpr (ACTIVE) {
      const float inv = 1.0f / l_row;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        half4 a, cc;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = f2h_sat(o0[4 * q + j] * inv); cc[j] = f2h_sat(o1[4 * q + j] * inv); }
        const auto au = __builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, a);
        const auto cu = __builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, cc);
        pk[0][2 * q] = au[0]; pk[0][2 * q + 1] = au[1];
        pk[1][2 * q] = cu[0]; pk[1][2 * q + 1] = cu[1];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const LaneCtx c4 = lane_ctx(opaque_lane());
    ATTD_STAMP(8);
    if constexpr (!LAST) {
      wait_q();                                           // K_h+1 (DMA) and the table entry landed long ago; Q_h+1 from L2
      sLut[((n + 1) & 1) * ATTD_LUT_N + c4.tid] = lutreg * ATT_LOG2E;
    }
    __builtin_amdgcn_sched_barrier(0);
    ATTD_STAMP(9);
    if constexpr (ACTIVE) {
      // piece q of o0 holds d = 8 q + 4 hh .. +4.  Swapping (piece 0 | hh=1) <-> (piece 2 | hh=0) and (1 | 1) <-> (3 | 0)
      // leaves d = 16 hh .. 16 hh + 16 contiguous in this lane: two 16-byte stores per 32-column half
      half_t* dst = p.ctx + (size_t)(tok0 + (c4.qpos < L ? c4.qpos : L - 1)) * p.ldctx + h * 64 + 16 * c4.hh;
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const auto x0 = __builtin_amdgcn_permlane32_swap(pk[o][0], pk[o][4], false, false);
        const auto x1 = __builtin_amdgcn_permlane32_swap(pk[o][1], pk[o][5], false, false);
        const auto y0 = __builtin_amdgcn_permlane32_swap(pk[o][2], pk[o][6], false, false);
        const auto y1 = __builtin_amdgcn_permlane32_swap(pk[o][3], pk[o][7], false, false);
        const attd_u32x4 lo = {x0[0], x1[0], x0[1], x1[1]};
        const attd_u32x4 hi = {y0[0], y1[0], y0[1], y1[1]};
        if (c4.qpos < L && store_ok && !ATTD_KO(5)) {
          *(attd_u32x4*)(dst + 32 * o) = lo;
          *(attd_u32x4*)(dst + 32 * o + 8) = hi;
        }
      }
    }
    ATTD_STAMP(10);
    if constexpr (!LAST) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): table written, every LDS read of this head retired
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      ATTD_STAMP(11);
      const int t = kb; kb = nb; nb = vb; vb = t;         // K_h+1 becomes K; V_h+1 goes where K_h was; K_h+2 where V_h was
    }
  };
  // the item loop: every item picks its head body by its own sequence length (key tiles; waves without query rows idle
  // through the same barriers)
  using T = std::integral_constant<bool, true>; using F = std::integral_constant<bool, false>;
  using N1 = std::integral_constant<int, 1>; using N2 = std::integral_constant<int, 2>; using N3 = std::integral_constant<int, 3>;
  for (int n = 0; n < n_walk; ++n) {
    tok0 = cur.tok0; L = cur.L;
    item_issue(raw0, ih0);                                // item n + 2: its offsets have the whole item to arrive
    const int nkt = (L + 63) >> 6;
    const bool active = q0 < L;
    if (n + 1 < n_walk) {
      if (!active) head(F{}, N1{}, F{}, n);
      else if (nkt == 3) head(F{}, N3{}, T{}, n);
      else if (nkt == 2) head(F{}, N2{}, T{}, n);
      else head(F{}, N1{}, T{}, n);
    } else {
      if (!active) head(T{}, N1{}, F{}, n);
      else if (nkt == 3) head(T{}, N3{}, T{}, n);
      else if (nkt == 2) head(T{}, N2{}, T{}, n);
      else head(T{}, N1{}, T{}, n);
    }
    cur = nxt; nxt = item_done(raw0, ih0);
  }
}

// Long sequences (L > ATT_ROW_MAXL: the setwise prompts, ~1.5k tokens; round 5).  The tiled kernel above stages K / V through
// registers (V through a 16-bit transpose), runs the online softmax per 64-key tile and reaches 13-15 % of the MFMA peak on
// eight 1 560-token prompts - a third of a lockstep setwise step.  This kernel brings the short-sequence kernel's recipe to
// long sequences: a 768-thread workgroup = twelve waves x 32 queries takes 384 queries of one (sequence, head) and walks the
// keys in chunks of 128: the chunk's K and V rows arrive by LDS-DMA (global_load_lds_dwordx4, swizzle on the source address,
// the same row-major image as the short kernel: conflict-free ds_read_b128 K fragments, V^T by ds_read_b64_tr_b16) into one
// of two stages while the previous chunk is computed (one barrier per chunk, behind a wait that has a whole chunk of compute to
// be satisfied); the 64 scores of a query row per chunk are formed by 16 back-to-back MFMAs, ONE maximum per chunk, the online
// merge (alpha, rescale of the 32 output accumulators) once per 128 keys instead of once per 64; the relative-position bias
// comes from a per-chunk table of key - query (built from the head's table in LDS), and a wave whose 32 queries are at least
// max_distance away from the whole chunk adds the bucket's constant instead (most chunks of a 1.5k prompt).
// A sequence's result depends on ITS length only: every sequence longer than ATT_ROW_MAXL takes this kernel, in any batch
// (attn_enc_kernel then only serves the batch's short sequences: AttnEncArgs::skip_long).  grid = ceil(n_seq H / 8) x 8 x ceil(maxL / (32 NW))
// workgroups, mapped to (sequence, head, query block) XCD-aware (below).
#define ATTL_KEYS 128
// NW waves x 32 queries per workgroup (12 / 6 / 4 / 3): a query row's arithmetic does not depend on NW - the same chunks of 128 keys in
// the same order, the same table values, the same near / far decision per wave - so the host picks NW from the batch (how many
// workgroups fill the chip, how much of the last query block is empty, how often a sequence's K / V are re-read) like a GEMM tile shape
#define ATTL_IMG_HALFS (ATTL_KEYS * 64)
#define ATTL_TAB_N 512
#define ATTL_LDS_BYTES (4 * ATTL_IMG_HALFS * 2 + 2 * ATTL_TAB_N * 4 + (RK_LUT_N + 3) * 4)
// ATTL_STAMP(k): the waves of workgroup ATTL_TRACE_WG record the shader clock (cycles since the kernel's start) at phase boundary k
// of every chunk: p.trace[(wave * 16 + chunk) * 16 + k]  (measurement builds, engine option attn_trace, tools/attn_long_trace.py)
#ifndef ATTL_TRACE_WG
#define ATTL_TRACE_WG 40
#endif
#ifdef RK_MEASURE
#define ATTL_STAMP(k) do { if (p.trace && blockIdx.x == ATTL_TRACE_WG && ch < 16 && wave < 12) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); \
    if ((threadIdx.x & 63) == 0) p.trace[(wave * 16 + ch) * 16 + (k)] = (float)(int)(t_ - attl_t0); } } while (0)
#else
#define ATTL_STAMP(k) do { } while (0)
#endif
template <int NW>
__global__ __launch_bounds__(64 * NW, NW >= 6 ? 3 : 2) void attn_enc_long_kernel(AttnEncArgs p) {
  constexpr int ATTL_QUERIES = 32 * NW, ATTL_THREADS = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char attl_smem[];
  half_t* const sbuf = (half_t*)attl_smem;                                      // [stage][K image | V image], [128][64] each
  float* const sTab = (float*)(attl_smem + 4 * ATTL_IMG_HALFS * 2);             // [stage][512]: bias(key - query) * log2(e) of a chunk
  float* const sLutH = sTab + 2 * ATTL_TAB_N;                                   // this head's table * log2(e)
  // workgroup -> (sequence, head, query block), XCD-aware: consecutive workgroups go to the 8 XCDs in turn, each with its own L2, so
  // workgroup i takes (sequence, head) pair 8 (i / 8 / nqb) + i % 8: the nqb workgroups that walk one pair's K / V rows (0.4 MB at
  // 1.5k tokens, read once per query block) meet in ONE L2 instead of every XCD pulling every pair through the fabric
  int pair, qb;
  if (p.xcd_map) { if (!xcd_decode((int)blockIdx.x, p.n_seq * p.n_heads, p.nqb, pair, qb)) return; }   // (xcd_map.h; uniform for the whole block)
  else { pair = (int)blockIdx.x / p.nqb; qb = (int)blockIdx.x % p.nqb; if (pair >= p.n_seq * p.n_heads) return; }
  const int b = pair / p.n_heads, h = pair % p.n_heads;
  const int tok0 = p.seq_off[b];
  const int L = p.seq_off[b + 1] - tok0;
  const int Q0 = qb * ATTL_QUERIES;
  if (L <= ATT_ROW_MAXL || Q0 >= L) return;                                     // uniform for the whole block
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  __builtin_assume(wave >= 0 && wave < NW);
  const int q0 = Q0 + wave * 32;
  const bool active = q0 < L;                                                   // wave-uniform
  const int nch = (L + ATTL_KEYS - 1) / ATTL_KEYS;
#ifdef RK_MEASURE
  const long long attl_t0 = (long long)__builtin_amdgcn_s_memtime();
#endif
  auto opaque_lane = [&]() {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
  };
  struct LaneCtx { int hh, l31, qpos, kfo0, vfo0, tab_q; };
  auto lane_ctx = [&](int lane) {
    LaneCtx c;
    c.hh = lane >> 5; c.l31 = lane & 31; c.qpos = q0 + c.l31;
    c.kfo0 = c.l31 * 64 + ((c.hh ^ ATTD_SWZ(c.l31)) << 3);
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    c.vfo0 = (4 * c.hh + (i16 >> 2)) * 64 + (((2 * g1 + ((i16 & 3) >> 1)) ^ ((((i16 >> 3) & 1) << 2) | c.hh)) << 3) + 4 * (i16 & 1);
    c.tab_q = (ATTL_QUERIES - 1) - (c.qpos - Q0) + 4 * c.hh;                    // sTab[tab_q + key - K0 - 4hh ...]: entry t <-> key - K0 - (query - Q0) + 383
    return c;
  };
  // K and V rows of chunk ch -> stage st: 32 pieces of 64 sixteen-byte slots (K image 16, V image 16) over the twelve waves
  constexpr int ATTL_NP = (32 + NW - 1) / NW;                                    // pieces per wave and chunk
  // byte offset of this lane's 16 bytes of piece k of chunk ch: row (tok0 + key) of the head's K / V columns, swizzled chunk.  (round 6)
  // The per-lane part (row within the chunk, swizzle) is formed ONCE (boff); a chunk wholly inside the sequence adds a uniform
  // stride - the address arithmetic of the eight pieces was ~110 of a chunk's ~590 VALU instructions.  The last chunk clamps its rows.
  unsigned boff[ATTL_NP];
  {
    const int lane = opaque_lane();
#pragma unroll
    for (int k = 0; k < ATTL_NP; ++k) {
      const int sub = (wave + NW * k) & 15, slot = sub * 64 + lane, r = slot >> 3, c = slot & 7;
      boff[k] = ((unsigned)(tok0 + r) * (unsigned)p.ld + ((c ^ ATTD_SWZ(r)) << 3)) * 2u;
    }
  }
  auto issue_piece = [&](auto insidec, int lane, int ch, int st, int k) {
    const int pid = wave + NW * k;
    if (pid < 32) {                                          // (decided at compile time wherever NW k + NW <= 32: wave < NW is assumed below)
      const int which = pid >> 4, sub = pid & 15;
      const char* hb = (const char*)(p.qkv + (1 + which) * p.I + h * 64);
      unsigned off;
      if constexpr (decltype(insidec)::value) {
        off = boff[k] + (unsigned)ch * (unsigned)(ATTL_KEYS * 2) * (unsigned)p.ld;
      } else {
        const int slot = sub * 64 + lane, r = slot >> 3, c = slot & 7;
        int key = ch * ATTL_KEYS + r;
        key = key < L ? key : L - 1;
        off = ((unsigned)(tok0 + key) * (unsigned)p.ld + ((c ^ ATTD_SWZ(r)) << 3)) * 2u;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(hb + off),
                                       (__attribute__((address_space(3))) void*)(sbuf + (2 * st + which) * ATTL_IMG_HALFS + sub * 512),
                                       16, 0, 0);
    }
  };
  auto issue_chunk = [&](int lane, int ch, int st) {
    if ((ch + 1) * ATTL_KEYS <= L) {                         // one uniform branch per chunk: every row of the chunk exists
#pragma unroll
      for (int k = 0; k < ATTL_NP; ++k) issue_piece(std::integral_constant<bool, true>{}, lane, ch, st, k);
    } else {
#pragma unroll
      for (int k = 0; k < ATTL_NP; ++k) issue_piece(std::integral_constant<bool, false>{}, lane, ch, st, k);
    }
  };
  // is any of the workgroup's queries within max_distance of chunk ch?  (then its table is built; uniform for the block)
  auto chunk_near_wg = [&](int ch) {
    const int lo = ch * ATTL_KEYS - (Q0 + ATTL_QUERIES - 1), hi = ch * ATTL_KEYS + ATTL_KEYS - 1 - Q0;     // range of key - query
    return hi > -RK_LUT_R && lo < RK_LUT_R;
  };
  auto build_table = [&](int ch, int st) {
    if (!chunk_near_wg(ch)) return;
    for (int t = threadIdx.x; t < ATTL_KEYS + ATTL_QUERIES - 1; t += ATTL_THREADS) {
      int rel = ch * ATTL_KEYS - Q0 + t - (ATTL_QUERIES - 1);
      rel = rel < -RK_LUT_R ? -RK_LUT_R : (rel > RK_LUT_R ? RK_LUT_R : rel);
      sTab[st * ATTL_TAB_N + t] = sLutH[rel + RK_LUT_R];
    }
  };

  // ---- prologue: the head's table, this lane's Q fragments, chunk 0 ----
  for (int i = threadIdx.x; i < RK_LUT_N; i += ATTL_THREADS) sLutH[i] = p.bias_lut[h * RK_LUT_N + i] * ATT_LOG2E;
  half8 qf[4];
  {
    const LaneCtx c = lane_ctx(opaque_lane());
    const int qrow = c.qpos < L ? c.qpos : L - 1;
    const half_t* qptr = p.qkv + (size_t)(tok0 + qrow) * p.ld + h * 64 + 8 * c.hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const half8*)(qptr + 16 * s);
  }
  issue_chunk(opaque_lane(), 0, 0);
  __syncthreads();                                          // sLutH written
  build_table(0, 0);
  // (the Q fragments are "used" here: the compiler's own wait for these tracked loads then sits in front of the chunk loop, not at
  // their first MFMA inside it, where - merged over the back edge - it would drain the DMA queue of every chunk)
  asm volatile("" :: "v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;                        // l_run: this lane's half of the row sum (the halves meet at the end)

  // scores of the chunk: two key tiles x 4 k16 steps, two MFMAs each; the K fragments of step i + 1 requested before the MFMAs of step i
  auto qk_chunk = [&](const LaneCtx& c, const half_t* kbuf, f32x16 (&s)[2][2]) {
    const half_t* kb_ = kbuf + c.kfo0;
    half8 kf[2][2];
    auto fetch = [&](int i, half8 (&d)[2]) {
      const half_t* a = kb_ + (i >> 2) * 64 * 64 + ((c.kfo0 ^ ((i & 3) << 4)) - c.kfo0);
      d[0] = *(const half8*)a;
      d[1] = *(const half8*)(a + 32 * 64);
    };
    fetch(0, kf[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i + 1 < 8) fetch(i + 1, kf[(i + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int kt = i >> 2, ks = i & 3;
      if (ks == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        s[kt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i & 1][0], qf[ks], z, 0, 0, 0);
        s[kt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i & 1][1], qf[ks], z, 0, 0, 0);
      } else {
        s[kt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i & 1][0], qf[ks], s[kt][0], 0, 0, 0);
        s[kt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i & 1][1], qf[ks], s[kt][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // P V of one key tile (the short kernel's pv_tile: asm transposing reads a k16 step ahead of their MFMAs, counted lgkmcnt)
  auto pv_tile = [&](const LaneCtx& c, const half_t* vbuf, int kt, const unsigned (&pp)[16]) {
    const unsigned vb0 = (unsigned)(size_t)(const __attribute__((address_space(3))) half_t*)(vbuf + kt * 64 * 64);
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
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf0, pf, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf1, pf, o1, 0, 0, 0);
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

  // one chunk of an active wave.  MASK: the chunk holds keys >= L (the last one); NEAR: bias from the chunk's table, else the
  // constant `far_bias` (every key of the chunk at least max_distance from every query of this wave)
  auto chunk_body = [&](auto maskc, auto nearc, int ch, int st, float far_bias) {
    constexpr bool MASK = decltype(maskc)::value, NEAR = decltype(nearc)::value;
    const half_t* kbuf = sbuf + (2 * st) * ATTL_IMG_HALFS;
    const half_t* vbuf = sbuf + (2 * st + 1) * ATTL_IMG_HALFS;
    f32x16 s[2][2];
    qk_chunk(lane_ctx(opaque_lane()), kbuf, s);
    __builtin_amdgcn_sched_barrier(0);
    ATTL_STAMP(1);
    float tmax = -1e30f;
    constexpr bool FARFOLD = !NEAR && !MASK;
    if constexpr (FARFOLD) {
      // (round 6) a far chunk wholly inside the sequence: ONE bias constant for all 128 x 32 scores, and t -> s log2(e) + bias is
      // monotonic, so the maximum is taken over the raw scores (max t = the same fma of max s, exactly) and the fma moves into the
      // exponent's argument: exp2(fma(s, log2(e), bias - m)) - 32 packed FMAs per chunk less (the kernel is bound by its VALU slots)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = attn_max3(tmax, s[kt][0][r], s[kt][1][r]);
      tmax = __builtin_fmaf(tmax, ATT_LOG2E, far_bias);
    } else {
      const LaneCtx c1 = lane_ctx(opaque_lane());
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int key_base = ch * ATTL_KEYS + kt * 64 + 4 * c1.hh;
        if constexpr (NEAR) {
          int lqi = c1.tab_q + kt * 64;
          asm volatile("" : "+v"(lqi));
          const float* lq = sTab + st * ATTL_TAB_N + lqi;
          auto bias = [&](int r, int sub) { return lq[(r & 3) + 8 * (r >> 2) + 32 * sub]; };
          attn_tile_bias_max<MASK, decltype(bias), true>(s[kt][0], s[kt][1], tmax, key_base, L, bias);
        } else {
          auto bias = [&](int, int) { return far_bias; };
          attn_tile_bias_max<MASK, decltype(bias), false>(s[kt][0], s[kt][1], tmax, key_base, L, bias);
        }
      }
    }
    const float m_c = attn_row_max(tmax);
    const float m_new = attn_max3(m_run, m_c, m_c);
    ATTL_STAMP(2);
    float psum = 0.f;
    unsigned pp[2][16];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if constexpr (FARFOLD) attn_tile_fma_exp(s[kt][0], s[kt][1], far_bias - m_new, psum);
      else attn_tile_exp(s[kt][0], s[kt][1], m_new, psum);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const half2v a = {(half_t)s[kt][0][2 * i], (half_t)s[kt][0][2 * i + 1]};
        const half2v b2 = {(half_t)s[kt][1][2 * i], (half_t)s[kt][1][2 * i + 1]};
        pp[kt][i] = __builtin_bit_cast(unsigned, a);
        pp[kt][8 + i] = __builtin_bit_cast(unsigned, b2);
      }
    }
    // online merge, once per chunk (m_run starts at -1e30: alpha = 0 and the zero accumulators stay zero on the first chunk)
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    const f32x2 a2 = {alpha, alpha};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 x0 = f32x2{o0[r], o0[r + 1]} * a2, x1 = f32x2{o1[r], o1[r + 1]} * a2;
      o0[r] = x0[0]; o0[r + 1] = x0[1]; o1[r] = x1[0]; o1[r + 1] = x1[1];
    }
    __builtin_amdgcn_sched_barrier(0);
    ATTL_STAMP(3);
    const LaneCtx c3 = lane_ctx(opaque_lane());
    pv_tile(c3, vbuf, 0, pp[0]);
    __builtin_amdgcn_sched_barrier(0);
    ATTL_STAMP(4);
    pv_tile(c3, vbuf, 1, pp[1]);
    __builtin_amdgcn_sched_barrier(0);
    ATTL_STAMP(5);
  };

  using T = std::integral_constant<bool, true>; using F = std::integral_constant<bool, false>;
  for (int ch = 0; ch < nch; ++ch) {
    const int st = ch & 1;
    // the next chunk travels while this one is computed: its stage was last read before the barrier that ended chunk ch - 1
    if (ch + 1 < nch) {
      issue_chunk(opaque_lane(), ch + 1, st ^ 1);
      build_table(ch + 1, st ^ 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    ATTL_STAMP(0);
    if (active) {
      const bool mask = (ch + 1) * ATTL_KEYS > L;
      // this wave's 32 queries against the chunk: range of key - query
      const int lo = ch * ATTL_KEYS - (q0 + 31), hi = ch * ATTL_KEYS + ATTL_KEYS - 1 - q0;
      const bool near = hi > -RK_LUT_R && lo < RK_LUT_R;
      const float far_bias = sLutH[hi <= -RK_LUT_R ? 0 : RK_LUT_N - 1];
      if (near) { if (mask) chunk_body(T{}, T{}, ch, st, 0.f); else chunk_body(F{}, T{}, ch, st, 0.f); }
      else { if (mask) chunk_body(T{}, F{}, ch, st, far_bias); else chunk_body(F{}, F{}, ch, st, far_bias); }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // chunk ch + 1 has landed (it had this whole chunk to do so)
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the next table is written, every LDS read of this chunk retired
    ATTL_STAMP(6);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ATTL_STAMP(7);
  }
  // ---- context rows: normalise, pack, store (the short kernel's permlane32_swap pairing: two 16-byte stores per 32-column half) ----
  if (active) {
    const float l_row = attn_row_sum(l_run);
    const float inv = 1.0f / l_row;
    unsigned pk[2][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      half4 a, cc;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = f2h_sat(o0[4 * q + j] * inv); cc[j] = f2h_sat(o1[4 * q + j] * inv); }
      const auto au = __builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, a);
      const auto cu = __builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, cc);
      pk[0][2 * q] = au[0]; pk[0][2 * q + 1] = au[1];
      pk[1][2 * q] = cu[0]; pk[1][2 * q + 1] = cu[1];
    }
    const LaneCtx c4 = lane_ctx(opaque_lane());
    half_t* dst = p.ctx + (size_t)(tok0 + (c4.qpos < L ? c4.qpos : L - 1)) * p.ldctx + h * 64 + 16 * c4.hh;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const auto x0 = __builtin_amdgcn_permlane32_swap(pk[o][0], pk[o][4], false, false);
      const auto x1 = __builtin_amdgcn_permlane32_swap(pk[o][1], pk[o][5], false, false);
      const auto y0 = __builtin_amdgcn_permlane32_swap(pk[o][2], pk[o][6], false, false);
      const auto y1 = __builtin_amdgcn_permlane32_swap(pk[o][3], pk[o][7], false, false);
      const attd_u32x4 lo = {x0[0], x1[0], x0[1], x1[1]};
      const attd_u32x4 hi = {y0[0], y1[0], y0[1], y1[1]};
      if (c4.qpos < L) {
        *(attd_u32x4*)(dst + 32 * o) = lo;
        *(attd_u32x4*)(dst + 32 * o + 8) = hi;
      }
    }
  }
}

// Decoder attention (self: causal + unidirectional bias; cross: zero bias, keys = encoder states of the same
// sequence).  One 256-thread workgroup per (query position, head, sequence): threads parallel over keys for the
// scores (q broadcast from LDS, 64 MACs per key in-lane), block reductions for max / sum, then the 4 waves split the
// keys for P V with lanes parallel over d, combined through LDS in a fixed order (bitwise reproducible).
// grid = (Lq, H, B); dynamic LDS = (64 + 4*64 + 8 + max_keys) floats.
#define ATTX_MAXK 192        // most keys per sequence of attn_dec_cross_mfma_kernel (six 32-key tiles: the whole score row in registers)
#define ATTX_MAXQ 64         // most decoder positions per sequence (two 32-query tiles, one wave each)
#define ATTX_KSTR 72         // halfs per K row in LDS (144 B: the 32 rows of a fragment read spread over the banks)
#define ATTX_VSTR 200        // halfs per V^T row in LDS (ATTX_MAXK keys + padding; 400 B)
#define ATTX_LDS_BYTES (ATTX_MAXK * ATTX_KSTR * 2 + 64 * ATTX_VSTR * 2 + (RK_LUT_N + 3) * 4)
struct AttnDecArgs {
  const half_t* q;  int ldq;     // query rows b*Lq + i, head columns h*64..
  const half_t* k;  const half_t* v;  int ldkv;   // key/value rows key_off + j
  const int* key_off;            // [B+1] (cross) or nullptr (self: keys are rows b*Lq .. b*Lq+Lq-1)
  half_t* ctx;      int ldctx;   // out rows b*Lq + i
  const float* bias_lut;         // [H][RK_LUT_N] or nullptr
  int Lq, causal, max_keys;
  // tree form of the causal self-attention (grid = (1, H, rows)): query row r sits at position tree_pos[r] and sees the
  // rows tree_keys[r * Lq + j], j = 0 .. tree_pos[r] - several continuations share the rows of their common prefix
  const int* tree_keys; const int* tree_pos;
  // cross-attention over the materialised K / V (long decoder prefixes, qlm): sequences of at most ATTX_MAXK keys are taken by
  // attn_dec_cross_mfma_kernel, the others by the staged kernels; which kernel computes a sequence follows from ITS key count
  // alone (its bits never depend on the batch): skip_short = 1 makes attn_dec_seq_kernel / attn_dec_kernel leave those to it
  int skip_short;
};

// The arithmetic of ONE query row of the decoder attention, shared by attn_dec_kernel (one workgroup per row: the tree form of
// rk_t5_greedy2 and sequences with more keys than attn_dec_seq_kernel stages) and attn_dec_seq_kernel (one workgroup per
// (head, sequence), a wave per row): explicit fma chains and fixed reduction trees, so that a row's context does not depend on
// which of the two kernels its call shape selects (tests/test_gpu_kernels.py: bit-identical).
__device__ __forceinline__ float dec_qk_dot(const float* sQ, const half_t* kr) {     // q . k over one head (64 columns)
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const half8 kk = *(const half8*)(kr + c * 8);
    const f32x4 q0 = *(const f32x4*)(sQ + c * 8), q1 = *(const f32x4*)(sQ + c * 8 + 4);
    s = __builtin_fmaf(q0[0], (float)kk[0], s); s = __builtin_fmaf(q0[1], (float)kk[1], s);
    s = __builtin_fmaf(q0[2], (float)kk[2], s); s = __builtin_fmaf(q0[3], (float)kk[3], s);
    s = __builtin_fmaf(q1[0], (float)kk[4], s); s = __builtin_fmaf(q1[1], (float)kk[5], s);
    s = __builtin_fmaf(q1[2], (float)kk[6], s); s = __builtin_fmaf(q1[3], (float)kk[7], s);
  }
  return s;
}
// Whole-wave sum / max for the decoder attention kernels without LDS round trips (wave_sum / wave_max of common.h go through six
// ds_bpermute each: ~1 us of a row's ~8 us): the 16 lanes of a DPP row by quad swaps and row rotations (row16_sum_f), then the four
// rows by v_readlane in a fixed order.  Every lane gets the result; ONE definition for both kernels (same bits).
__device__ __forceinline__ float dec_wave_sum(float v) {
  v = row16_sum_f(v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float dec_wave_max(float v) {
  v = row16_max(v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float dec_bias(const float* lut, int h, int j, int i) {
  int rel = j - i;
  rel = rel < -RK_LUT_R ? -RK_LUT_R : (rel > RK_LUT_R ? RK_LUT_R : rel);
  return lut[h * RK_LUT_N + rel + RK_LUT_R];
}
// sum_j P[j] V[j][d] over the keys j = w, w + 4, ... (the share of wave w of a four-wave workgroup), four interleaved chains
template <class VF>
__device__ __forceinline__ float dec_pv_part(const float* sP, int nk, int w, VF vf) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int j = w;
  for (; j + 12 < nk; j += 16) {
    a0 = __builtin_fmaf(sP[j], vf(j), a0);
    a1 = __builtin_fmaf(sP[j + 4], vf(j + 4), a1);
    a2 = __builtin_fmaf(sP[j + 8], vf(j + 8), a2);
    a3 = __builtin_fmaf(sP[j + 12], vf(j + 12), a3);
  }
  for (; j < nk; j += 4) a0 = __builtin_fmaf(sP[j], vf(j), a0);
  return (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void attn_dec_kernel(AttnDecArgs p) {
  extern __shared__ __attribute__((aligned(16))) float dec_smem[];
  float* sQ = dec_smem;            // [64]
  float* sPart = sQ + 64;          // [4][64]
  float* sRed = sPart + 256;       // [8]
  float* sP = sRed + 8;            // [max_keys]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.z, h = blockIdx.y;
  const int* tkeys = p.tree_keys ? p.tree_keys + (size_t)b * p.Lq : nullptr;
  const int i = tkeys ? p.tree_pos[b] : (int)blockIdx.x;
  int koff, Lk;
  if (tkeys) { koff = 0; Lk = i + 1; }
  else if (p.key_off) { koff = p.key_off[b]; Lk = p.key_off[b + 1] - koff; }
  else { koff = b * p.Lq; Lk = p.Lq; }
  if (p.skip_short && Lk <= ATTX_MAXK) return;                // attn_dec_cross_mfma_kernel's sequence (uniform for the workgroup)
  const int nk = p.causal ? (i + 1 < Lk ? i + 1 : Lk) : Lk;   // keys 0..nk-1 are visible
  const size_t qrow = tkeys ? (size_t)b : (size_t)(b * p.Lq + i);
  auto krow = [&](int j) { return tkeys ? (size_t)tkeys[j] : (size_t)(koff + j); };
  if (tid < 64) sQ[tid] = (float)p.q[qrow * p.ldq + h * 64 + tid];
  __syncthreads();
  float mx = -1e30f;
  for (int j = tid; j < nk; j += 256) {
    float s = dec_qk_dot(sQ, p.k + krow(j) * p.ldkv + h * 64);
    if (p.bias_lut) s += dec_bias(p.bias_lut, h, j, i);
    sP[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = dec_wave_max(mx);
  if (lane == 0) sRed[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3]));
  float sum = 0.f;
  for (int j = tid; j < nk; j += 256) {
    const float e = __expf(sP[j] - mx);
    sP[j] = e;
    sum += e;
  }
  sum = dec_wave_sum(sum);
  if (lane == 0) sRed[4 + wave] = sum;
  __syncthreads();
  sum = (sRed[4] + sRed[5]) + (sRed[6] + sRed[7]);
  // P V: wave w takes keys w, w+4, ...; lane = d
  const half_t* vb = p.v + h * 64 + lane;
  sPart[wave * 64 + lane] = dec_pv_part(sP, nk, wave, [&](int j) { return (float)vb[krow(j) * p.ldkv]; });
  __syncthreads();
  if (wave == 0) {
    const float acc = (sPart[lane] + sPart[64 + lane]) + (sPart[128 + lane] + sPart[192 + lane]);
    p.ctx[qrow * p.ldctx + h * 64 + lane] = f2h_sat(acc / sum);
  }
}

// The same attention with ONE workgroup per (head, sequence) for decoder passes of several positions (qlm: ~30 label
// positions, ref: llmrankers/pointwise.py:41-82; greedy prefixes): the per-row kernel above is launched as L_d x H x B
// workgroups that each re-read the head's K and V rows of their sequence (27 x 36 KB through L2 per (head, sequence) in a
// flan-t5-xl qlm call: 0.72 ms per launch, 34 of the 83 ms of a hits=100 query).  Here the K rows (144-byte stride: the lanes'
// 16-byte reads fall on different banks), the V rows and the head's bias table are staged in LDS once, and the four waves take the
// query rows in turn.  A row is computed with the arithmetic above - the four key shares of the per-row kernel's waves walked by
// one wave, same chains, same trees: BIT-IDENTICAL to the per-row kernel (tests).  Causal rows (self-attention: few keys) go one
// at a time; rows without a mask (cross-attention: every row sees all keys of the sequence) go FOUR AT A TIME - a K row, a V
// element and the four rows' probabilities (kept interleaved, one 16-byte broadcast read per key) are read from LDS once for the
// four rows, which is what the kernel's time is made of (round 5: LDS instructions per row 350 -> ~95).
// grid = (H, B); not for the tree form.
#define ATTS_KSTR 72
__host__ __device__ inline size_t attn_dec_seq_lds(int max_keys) {
  const size_t kp = ((size_t)max_keys + 3) & ~(size_t)3;
  return kp * ATTS_KSTR * 2 + kp * 64 * 2 + RK_LUT_N * sizeof(float) + 12 + 4 * (64 * 4 + kp * 4) * sizeof(float);
}
// Cross-attention of the long-prefix decoder (qlm: ~30 label positions per passage over its ~140 encoder rows; hf:
// modeling_t5.py:404-432 with the materialised K / V, no position bias, no mask) on the matrix cores (round 6).  The staged kernel
// below spent 128 us per launch on explicit fma chains (1.5 G fma per layer at flan-t5-xl dims, VALU-bound) for 114 MB of K / V -
// a 25-us read.  One workgroup of two waves per (head, sequence), each wave 32 decoder positions: K rows staged row-major, V
// TRANSPOSED (sVt[d][key]); S^T = K Q^T with mfma 32x32x16 - a lane owns ONE query column, its 16 registers per 32-key tile are
// keys (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the tile - the whole score row (<= 192 keys) stays in registers: one maximum, one
// sum (one exchange with the partner lane), P packed to fp16 in place; O^T = V^T P^T: registers 8 s .. 8 s + 7 of a tile ARE the
// B operand of its s-th 16-key step, and the matching V^T fragment is two 8-byte reads (keys 16 s + 4 hh + {0..3, 8..11}).
// Which sequences it takes follows from their own key count (<= ATTX_MAXK) and the call's position count (<= ATTX_MAXQ).
// The SAME kernel serves the causal self-attention of those prefixes (key_off = nullptr: the keys are the sequence's own Lq <= 64
// decoder rows; unidirectional position bias from the head's table, key > query masked; hf: modeling_t5.py:144-173).
__global__ __launch_bounds__(128) void attn_dec_cross_mfma_kernel(AttnDecArgs p) {
  extern __shared__ __attribute__((aligned(16))) float dec_smem[];
  half_t* sK = (half_t*)dec_smem;                   // [ATTX_MAXK][ATTX_KSTR]
  half_t* sVt = sK + ATTX_MAXK * ATTX_KSTR;         // [64][ATTX_VSTR]
  float* sLut = (float*)(sVt + 64 * ATTX_VSTR);     // [RK_LUT_N] (self-attention form only)
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x, b = blockIdx.y;
  const int koff = p.key_off ? p.key_off[b] : b * p.Lq, Lk = p.key_off ? p.key_off[b + 1] - koff : p.Lq;
  if (Lk > ATTX_MAXK || Lk <= 0) return;            // the staged kernel's sequence (uniform)
  const int nt = (Lk + 31) >> 5;                    // key tiles
  if (p.bias_lut)
    for (int idx = tid; idx < RK_LUT_N; idx += 128) sLut[idx] = p.bias_lut[h * RK_LUT_N + idx];
  // ---- stage K (row-major) and V (transposed); rows beyond Lk are zero: their scores are masked, their P is exactly 0 ----
  for (int idx = tid; idx < nt * 32 * 8; idx += 128) {
    const int r = idx >> 3, c = idx & 7;
    half8 kk = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kk;
    if (r < Lk) {
      const size_t g = (size_t)(koff + r) * p.ldkv + h * 64 + c * 8;
      kk = *(const half8*)(p.k + g);
      vv = *(const half8*)(p.v + g);
    }
    *(half8*)(sK + r * ATTX_KSTR + c * 8) = kk;
#pragma unroll
    for (int j = 0; j < 8; ++j) sVt[(c * 8 + j) * ATTX_VSTR + r] = vv[j];
  }
  __syncthreads();
  if (wave * 32 >= p.Lq) return;                    // a second wave without decoder positions
  // ---- Q fragments of this wave's 32 positions (B operand: k = 16 ks + 8 hh .. + 7 of query l31) ----
  const int qi = wave * 32 + l31;
  const size_t qrow = (size_t)b * p.Lq + (qi < p.Lq ? qi : p.Lq - 1);
  half8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const half8*)(p.q + qrow * p.ldq + h * 64 + ks * 16 + 8 * hh);
  // ---- scores: S^T tile t = K_t Q^T ----
  f32x16 sc[ATTX_MAXK / 32];
  float mx = -1e30f;
#pragma unroll
  for (int t = 0; t < ATTX_MAXK / 32; ++t) {
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 kf = *(const half8*)(sK + (t * 32 + l31) * ATTX_KSTR + ks * 16 + 8 * hh);
        sc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sc[t], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (p.bias_lut) {
          int rel = key - qi;
          rel = rel < -RK_LUT_R ? -RK_LUT_R : (rel > RK_LUT_R ? RK_LUT_R : rel);
          sc[t][r] += sLut[rel + RK_LUT_R];
        }
        if (key >= Lk || (p.causal && key > qi)) sc[t][r] = -1e30f;
        mx = fmaxf(mx, sc[t][r]);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  // ---- softmax over the row (this lane's keys + the partner lane's), P to fp16 in place of the scores ----
  float sum = 0.f;
  half8 pf[ATTX_MAXK / 32][2];
#pragma unroll
  for (int t = 0; t < ATTX_MAXK / 32; ++t) {
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __expf(sc[t][r] - mx);       // masked keys: exp(-1e30 - mx) = 0
        sum += e;
        pf[t][r >> 3][r & 7] = (half_t)e;
      }
    }
  }
  sum += __shfl_xor(sum, 32);
  // ---- O^T = V^T P^T: two 32-row d tiles ----
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
  for (int t = 0; t < ATTX_MAXK / 32; ++t) {
    if (t < nt) {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const half_t* vr = sVt + (dt * 32 + l31) * ATTX_VSTR + t * 32 + 16 * st + 4 * hh;
          const half4 v0 = *(const half4*)vr, v1 = *(const half4*)(vr + 8);
          const half8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[t][st], o[dt], 0, 0, 0);
        }
    }
  }
  // ---- context row of query l31: d = 32 dt + (r & 3) + 8 (r >> 2) + 4 hh -> 8-byte pieces ----
  if (qi < p.Lq) {
    const float inv = 1.0f / sum;
    half_t* dst = p.ctx + ((size_t)b * p.Lq + qi) * p.ldctx + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const half4 x = {f2h_sat(o[dt][4 * q] * inv), f2h_sat(o[dt][4 * q + 1] * inv), f2h_sat(o[dt][4 * q + 2] * inv), f2h_sat(o[dt][4 * q + 3] * inv)};
        *(half4*)(dst + dt * 32 + 8 * q + 4 * hh) = x;
      }
  }
}

__global__ __launch_bounds__(256) void attn_dec_seq_kernel(AttnDecArgs p) {
  extern __shared__ __attribute__((aligned(16))) float dec_smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int h = blockIdx.x, b = blockIdx.y;
  const int kp = (p.max_keys + 3) & ~3;
  half_t* sK = (half_t*)dec_smem;                  // [kp][ATTS_KSTR]
  half_t* sV = sK + (size_t)kp * ATTS_KSTR;        // [kp][64]
  float* sLut = (float*)(sV + (size_t)kp * 64);    // [RK_LUT_N (+3)]  this head's bias by clamp(key - query)
  float* sQ = sLut + RK_LUT_N + 3 + (size_t)wave * (64 * 4 + kp * 4);   // [4][64]   (wave-private from here on)
  float* sP = sQ + 64 * 4;                         // [kp][4]: the probabilities of the wave's (up to) four rows, interleaved
  int koff, Lk;
  if (p.key_off) { koff = p.key_off[b]; Lk = p.key_off[b + 1] - koff; }
  else { koff = b * p.Lq; Lk = p.Lq; }
  if (p.skip_short && Lk <= ATTX_MAXK) return;     // attn_dec_cross_mfma_kernel's sequence (uniform for the workgroup)
  for (int idx = tid; idx < Lk * 8; idx += 256) {
    const int r = idx >> 3, c = idx & 7;
    const size_t g = (size_t)(koff + r) * p.ldkv + h * 64 + c * 8;
    *(half8*)(sK + r * ATTS_KSTR + c * 8) = *(const half8*)(p.k + g);
    *(half8*)(sV + r * 64 + c * 8) = *(const half8*)(p.v + g);
  }
  if (p.bias_lut)
    for (int idx = tid; idx < RK_LUT_N; idx += 256) sLut[idx] = p.bias_lut[h * RK_LUT_N + idx];
  __syncthreads();
  auto bias_at = [&](int j, int i) {
    int rel = j - i;
    rel = rel < -RK_LUT_R ? -RK_LUT_R : (rel > RK_LUT_R ? RK_LUT_R : rel);
    return sLut[rel + RK_LUT_R];
  };
  const half_t* vb = sV + lane;
  if (p.causal || p.bias_lut) {
    // ---- one row at a time (self-attention: nk = i + 1 keys) ----
    for (int i = wave; i < p.Lq; i += 4) {
      const int nk = p.causal ? (i + 1 < Lk ? i + 1 : Lk) : Lk;
      const size_t qrow = (size_t)b * p.Lq + i;
      sQ[lane] = (float)p.q[qrow * p.ldq + h * 64 + lane];
      __builtin_amdgcn_wave_barrier();
      float mx = -1e30f;
      for (int j0 = lane; j0 < nk; j0 += 256) {
        float sc[4] = {0.f, 0.f, 0.f, 0.f};
        const half_t* kr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int j = j0 + 64 * u; kr[u] = sK + (j < nk ? j : j0) * ATTS_KSTR; }
        const int nu = (nk - (j0 - lane) + 63) >> 6;       // chains of this pass that hold a key for SOME lane (uniform)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const f32x4 q0 = *(const f32x4*)(sQ + c * 8), q1 = *(const f32x4*)(sQ + c * 8 + 4);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (u >= nu) continue;
            const half8 kk = *(const half8*)(kr[u] + c * 8);
            float s = sc[u];
            s = __builtin_fmaf(q0[0], (float)kk[0], s); s = __builtin_fmaf(q0[1], (float)kk[1], s);
            s = __builtin_fmaf(q0[2], (float)kk[2], s); s = __builtin_fmaf(q0[3], (float)kk[3], s);
            s = __builtin_fmaf(q1[0], (float)kk[4], s); s = __builtin_fmaf(q1[1], (float)kk[5], s);
            s = __builtin_fmaf(q1[2], (float)kk[6], s); s = __builtin_fmaf(q1[3], (float)kk[7], s);
            sc[u] = s;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = j0 + 64 * u;
          if (j < nk) {
            float s = sc[u];
            if (p.bias_lut) s += bias_at(j, i);
            sP[j] = s;
            mx = fmaxf(mx, s);
          }
        }
      }
      mx = dec_wave_max(mx);
      float ssum[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {                  // the per-row kernel's thread tid = 64 w + lane owns keys tid, tid + 256, ...
        float sum = 0.f;
        if (nk > 64 * w) {                           // (a share without keys sums to +0: the reduction tree below is unchanged)
          for (int j = w * 64 + lane; j < nk; j += 256) {
            const float e = __expf(sP[j] - mx);
            sP[j] = e;
            sum += e;
          }
          sum = dec_wave_sum(sum);
        }
        ssum[w] = sum;
      }
      const float sum = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
      __builtin_amdgcn_wave_barrier();
      float part[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) part[w] = dec_pv_part(sP, nk, w, [&](int j) { return (float)vb[j * 64]; });
      const float acc = (part[0] + part[1]) + (part[2] + part[3]);
      p.ctx[qrow * p.ldctx + h * 64 + lane] = f2h_sat(acc / sum);
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  // ---- four rows at a time (no mask, no bias: every row of the block sees keys 0 .. Lk-1) ----
  const int nk = Lk;
  for (int i0 = wave; i0 < p.Lq; i0 += 16) {
    int rows[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int i = i0 + 4 * r; rows[r] = i < p.Lq ? i : i0; }   // (a missing row repeats the first: computed, not stored)
#pragma unroll
    for (int r = 0; r < 4; ++r) sQ[r * 64 + lane] = (float)p.q[((size_t)b * p.Lq + rows[r]) * p.ldq + h * 64 + lane];
    __builtin_amdgcn_wave_barrier();
    float mx[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
    for (int j0 = lane; j0 < nk; j0 += 256) {
      float sc[4][4];                                   // [row][chain]
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) sc[r][u] = 0.f;
      const half_t* kr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int j = j0 + 64 * u; kr[u] = sK + (j < nk ? j : j0) * ATTS_KSTR; }
      const int nu = (nk - (j0 - lane) + 63) >> 6;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        half8 kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (u < nu) kk[u] = *(const half8*)(kr[u] + c * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 q0 = *(const f32x4*)(sQ + r * 64 + c * 8), q1 = *(const f32x4*)(sQ + r * 64 + c * 8 + 4);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (u >= nu) continue;
            float s = sc[r][u];
            s = __builtin_fmaf(q0[0], (float)kk[u][0], s); s = __builtin_fmaf(q0[1], (float)kk[u][1], s);
            s = __builtin_fmaf(q0[2], (float)kk[u][2], s); s = __builtin_fmaf(q0[3], (float)kk[u][3], s);
            s = __builtin_fmaf(q1[0], (float)kk[u][4], s); s = __builtin_fmaf(q1[1], (float)kk[u][5], s);
            s = __builtin_fmaf(q1[2], (float)kk[u][6], s); s = __builtin_fmaf(q1[3], (float)kk[u][7], s);
            sc[r][u] = s;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + 64 * u;
        if (j < nk) {
          const f32x4 v4 = {sc[0][u], sc[1][u], sc[2][u], sc[3][u]};
          *(f32x4*)(sP + j * 4) = v4;
#pragma unroll
          for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], sc[r][u]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = dec_wave_max(mx[r]);
    float ssum[4][4];                                    // [row][share]
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float sum[4] = {0.f, 0.f, 0.f, 0.f};
      if (nk > 64 * w) {
        for (int j = w * 64 + lane; j < nk; j += 256) {
          f32x4 v4 = *(const f32x4*)(sP + j * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float e = __expf(v4[r] - mx[r]); v4[r] = e; sum[r] += e; }
          *(f32x4*)(sP + j * 4) = v4;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] = dec_wave_sum(sum[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) ssum[r][w] = sum[r];
    }
    __builtin_amdgcn_wave_barrier();
    // P V: per row the four key shares x four chains of dec_pv_part; a V element and the four rows' probabilities are read once
    float a[4][4][4];                                    // [row][share][chain]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int c = 0; c < 4; ++c) a[r][w][c] = 0.f;
    int b0 = 0;
    for (; b0 + 15 < nk; b0 += 16) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int j = b0 + w + 4 * c;
          const float v = (float)vb[j * 64];
          const f32x4 p4 = *(const f32x4*)(sP + j * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r][w][c] = __builtin_fmaf(p4[r], v, a[r][w][c]);
        }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int j = b0 + w;
      for (; j + 12 < nk; j += 16) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float v = (float)vb[(j + 4 * c) * 64];
          const f32x4 p4 = *(const f32x4*)(sP + (j + 4 * c) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r][w][c] = __builtin_fmaf(p4[r], v, a[r][w][c]);
        }
      }
      for (; j < nk; j += 4) {
        const float v = (float)vb[j * 64];
        const f32x4 p4 = *(const f32x4*)(sP + j * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r][w][0] = __builtin_fmaf(p4[r], v, a[r][w][0]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * r;
      if (i < p.Lq) {
        const float sum = (ssum[r][0] + ssum[r][1]) + (ssum[r][2] + ssum[r][3]);
        float part[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) part[w] = (a[r][w][0] + a[r][w][1]) + (a[r][w][2] + a[r][w][3]);
        const float acc = (part[0] + part[1]) + (part[2] + part[3]);
        p.ctx[((size_t)b * p.Lq + i) * p.ldctx + h * 64 + lane] = f2h_sat(acc / sum);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Direct cross-attention for a handful of decoder rows (M = sequences x decoder positions <= 32).
// hf: T5LayerCrossAttention (modeling_t5.py:404-432) projects EVERY encoder state to K and V in every decoder layer
// (13.6 % of all FLOPs of the pointwise path).  With so few queries it is cheaper to move the projections to the
// query side - exact algebra, no approximation:
//     score[h][t] = q_h . (W_k,h e_t)            = (W_k,h^T q_h) . e_t
//     ctx_h       = sum_t p[h][t] (W_v,h e_t)    = W_v,h (sum_t p[h][t] e_t)
// so per layer: qk = W_k^T q (per-head skinny GEMM), this kernel pair (scores + softmax + weighted sum of the raw
// encoder states, split over 64-key chunks like flash-decoding), then ctx = W_v (.) (per-head skinny GEMM).
// 64x fewer FLOPs than the K/V projections and no [n_dec][T][2I] hand-off buffer.
struct XAttnArgs {
  const half_t* qk;      // [M, H, d]   W_k^T q per head (fp16)
  const half_t* enc;     // [T, d]      encoder output after the final norm (fp16)
  const int* seq_off;    // [B+1]
  float* part;           // [M, nch, H, d]  unnormalised partial sums, relative to the chunk's own max
  float* stat;           // [M, nch, H, 2]  (chunk max, chunk sum of exp)
  half_t* out;           // [M, H, d]   normalised  sum_t p[h][t] e_t  (fp16)
  int Ld, H, d, nch;
  int row0;              // decoder row of this pass's first query (blockIdx.y is relative to it; qk / part / stat / out are per pass)
  const int* row_seq;    // optional: decoder ROW -> encoder sequence, when the rows are not Ld per sequence (rk_t5_greedy2's
                         // candidate continuations of one prompt share their prefix rows); nullptr: row / Ld
};

// (Tried and dropped: ONE workgroup per row of a short sequence walking its <= 4 chunks with a running softmax - no fp32
// partial sums through HBM, no combine launch: 76 us per layer at 320 rows against 58 + 14 us for this pair; the row's
// chunks in sequence are slower than the same chunks side by side, and the pointwise step did not move.)
// grid = (nch, M, ceil(H/HPW)); 256 threads.  One 64-key chunk of one decoder row for a group of up to HPW heads
// (16: one workgroup per chunk and row - the 256-passage groups; 4: four times as many, lighter workgroups for the
// few-row setwise calls, where 48 workgroups of 16 heads left most of the chip idle).  Per-head arithmetic is the same.
template <int HPW>
__global__ __launch_bounds__(256) void xattn_part_kernel(XAttnArgs p) {
  constexpr int PR = 8;                        // key pairs per round of the weighted sums (16, and 16 instead of 4 score steps
                                               // in flight, changed nothing: 17.5 us either way for a 1.4k-token prompt)
  // softmax weights of the chunk as fp16 KEY PAIRS [32 pairs][16 heads]: the weighted sums below run on v_dot2_f32_f16
  // (two keys per instruction, fp32 accumulate) - half the VALU work of an fp32 FMA per key
  __shared__ __attribute__((aligned(16))) half2v sP2[32 * 16];
  __shared__ float sRed[2][4][16];
  const int ck = blockIdx.x, m = blockIdx.y, hg = blockIdx.z;
  const int b = p.row_seq ? p.row_seq[p.row0 + m] : (p.row0 + m) / p.Ld;
  const int tok0 = p.seq_off[b];
  const int L = p.seq_off[b + 1] - tok0;
  const int t0 = ck * 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, g = lane >> 4;
  const int nh = min(HPW, p.H - hg * HPW);
  float* stat = p.stat + (((size_t)m * p.nch + ck) * p.H + hg * HPW) * 2;
  if (t0 >= L) {     // this row has no keys here: mark the chunk empty for the combine step
    if (tid < nh) { stat[tid * 2] = -1e30f; stat[tid * 2 + 1] = 0.f; }
    return;
  }
  // ---- scores: S[h][t] = qk[h] . e_t  by MFMA 16x16x32 (A = the 16 head rows of qk, B = 16 encoder rows per wave) ----
  const int hrow = min(hg * HPW + l15, p.H - 1);
  const half_t* ap = p.qk + ((size_t)m * p.H + hrow) * p.d + 8 * g;
  const int t = t0 + wave * 16 + l15;
  const half_t* bp = p.enc + (size_t)(tok0 + min(t, L - 1)) * p.d + 8 * g;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int c0 = 0; c0 < p.d; c0 += 32)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const half8*)(ap + c0), *(const half8*)(bp + c0), acc, 0, 0, 0);
  // lane holds heads 4g..4g+3 for key t (C layout: col = lane&15, row = 4*(lane>>4) + r)
  float mx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (t >= L) acc[r] = -1e30f;
    mx[r] = row16_max(acc[r]);
  }
  if (l15 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sRed[0][wave][4 * g + r] = mx[r];
  }
  __syncthreads();
  float sm[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int h = 4 * g + r;
    mx[r] = fmaxf(fmaxf(sRed[0][0][h], sRed[0][1][h]), fmaxf(sRed[0][2][h], sRed[0][3][h]));
    const float e = __expf(acc[r] - mx[r]);
    {
      const int key = wave * 16 + l15;              // [key pair][head][key & 1]
      ((half_t*)sP2)[(((key >> 1) * 16 + h) << 1) + (key & 1)] = (half_t)e;
    }
    sm[r] = row16_sum_f(e);
  }
  if (l15 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sRed[1][wave][4 * g + r] = sm[r];
  }
  __syncthreads();
  if (tid < nh) {
    stat[tid * 2] = fmaxf(fmaxf(sRed[0][0][tid], sRed[0][1][tid]), fmaxf(sRed[0][2][tid], sRed[0][3][tid]));
    stat[tid * 2 + 1] = (sRed[1][0][tid] + sRed[1][1][tid]) + (sRed[1][2][tid] + sRed[1][3][tid]);
  }
  // ---- partial weighted sums of the raw encoder rows: thread owns 4 consecutive columns ----
  const int nvalid = min(64, L - t0);
  float* part = p.part + (((size_t)m * p.nch + ck) * p.H + hg * HPW) * p.d;
  for (int cb = tid * 4; cb < p.d; cb += 1024) {
    float a[HPW][4];
#pragma unroll
    for (int h = 0; h < HPW; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[h][j] = 0.f;
    const half_t* ep = p.enc + (size_t)tok0 * p.d + cb;
    // 8 key pairs per round: the 16 row loads go out together (the loop is bound by memory latency, not by the dots)
    const int npair = (nvalid + 1) >> 1;
    for (int tp0 = 0; tp0 < npair; tp0 += PR) {
      half4 ea[PR], eb[PR];
#pragma unroll
      for (int u = 0; u < PR; ++u) {
        const int r0 = min(t0 + 2 * (tp0 + u), L - 1), r1 = min(t0 + 2 * (tp0 + u) + 1, L - 1);   // clamped rows carry weight 0
        ea[u] = *(const half4*)(ep + (size_t)r0 * p.d);
        eb[u] = *(const half4*)(ep + (size_t)r1 * p.d);
      }
#pragma unroll
      for (int u = 0; u < PR; ++u) {
        if (tp0 + u < npair) {
          half2v e2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { e2[j][0] = ea[u][j]; e2[j][1] = eb[u][j]; }
#pragma unroll
          for (int hq = 0; hq < HPW / 4; ++hq) {
            const half8 w8 = *(const half8*)(sP2 + (tp0 + u) * 16 + hq * 4);   // LDS broadcast read: 4 heads x (key, key+1)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const half2v w2 = {w8[2 * r], w8[2 * r + 1]};
              const int h = hq * 4 + r;
#pragma unroll
              for (int j = 0; j < 4; ++j) a[h][j] = __builtin_amdgcn_fdot2(w2, e2[j], a[h][j], false);
            }
          }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < HPW; ++h)
      if (h < nh) { f32x4 o = {a[h][0], a[h][1], a[h][2], a[h][3]}; *(f32x4*)(part + (size_t)h * p.d + cb) = o; }
  }
}

// MFMA form of the chunk kernel (round 3; the production path).  The kernel above forms the weighted sums of the raw encoder
// rows on the VALU (v_dot2_f32_f16: 2048 dots per thread and chunk - at 320 rows x 3 chunks the pointwise decoder spent 58 us
// per layer here, its largest kernel).  They are a small GEMM - [16 heads x 64 keys] x [64 keys x d] - whose B operand is
// key-strided in memory.  Here the scores are computed exactly as above; then every wave walks ITS quarter of the columns in
// sub-slabs of 64: the [64 keys][64 columns] piece goes to a wave-private 8 KiB of LDS by DMA (eight 1-KiB instructions, the
// wave's own vmcnt(0), no workgroup barrier) and the sums run on mfma_f32_16x16x32_f16 with the E^T fragments read by
// ds_read_b64_tr_b16.  LDS image per wave: 128-byte rows, two per bank row; the 32-byte column pairs are XOR-swizzled by
// key bits (1, 3), which with the row parity puts the eight keys of one transposing read (4 + 4 of two k groups) into the
// eight 32-byte slots of the bank row.  35 KiB per workgroup: four per CU, whose DMA waits hide each other.
// (First form, dropped: one loader wave staging the whole [64][1024] chunk, 131 KiB = one workgroup per CU: 82 us per layer
// against 72 for the VALU pair - the scores phase lost the three co-resident workgroups that had hidden its load latency.
// Tried again at the end of round 3 with all four waves issuing the DMA and the scores reading the chunk from LDS - the rows
// then come from memory ONCE instead of twice (this kernel fetches 2 x 133 MB per launch for 120 MB of rows), bit-identical:
// dec_attn 0.170 -> 0.202 ms per step.  One workgroup per CU cannot overlap its own load and compute; four can.)
// Always 16 heads per workgroup (H > 16: blockIdx.z).  grid = (nch, M, ceil(H / 16)); 256 threads.
#define XAM_PSTR 72                                            // sP row stride in halfs: 144 B -> the 16 head rows start in 16 different 16-byte slots
#define XAM_LDS_TAIL (16 * XAM_PSTR * 2 + 2 * 4 * 16 * 4)                               // P, reduction scratch
template <bool FEW> constexpr int xam_lds_bytes() { return (FEW ? 8 : 4) * 64 * 64 * 2 + XAM_LDS_TAIL; }   // one / two [64 keys][64 columns] pieces per wave
#define XAM_F(key) ((((key) >> 1) & 1) | ((((key) >> 3) & 1) << 1))   // XOR on the 32-byte pair index (4 pairs per 128-byte row)
// FEW (round 6): the same arithmetic in the same order, scheduled for the few-row calls (one setwise compare: 23 chunks x 2 - 13 rows
// = 46 - 299 workgroups on 256 CUs, each a chain of exposed round trips: 15.2 us per launch): the score loop keeps sixteen k32 steps
// of operand loads in flight instead of four, and the wave's NEXT piece of encoder rows travels (asm LDS-DMA: the builtin makes the
// compiler drain the queue in front of every LDS read) while the current one is multiplied - two pieces per wave, counted vmcnt.
// The large pointwise launches (960 workgroups, four per CU hiding each other's latency) keep the one-piece form: twice the LDS
// would halve their residency.  Chosen from the launch's workgroup count; a row's bits do not depend on it.
template <bool FEW>
__global__ __launch_bounds__(256) void xattn_part_mfma_kernel(XAttnArgs p) {
  constexpr int NPIECE = FEW ? 2 : 1;
  __shared__ __attribute__((aligned(16))) unsigned char xam_smem[xam_lds_bytes<FEW>()];
  half_t* sP = (half_t*)(xam_smem + NPIECE * 4 * 64 * 64 * 2);
  float (*sRed)[4][16] = (float (*)[4][16])(xam_smem + NPIECE * 4 * 64 * 64 * 2 + 16 * XAM_PSTR * 2);
  const int ck = blockIdx.x, m = blockIdx.y, hg = blockIdx.z;
  const int b = p.row_seq ? p.row_seq[p.row0 + m] : (p.row0 + m) / p.Ld;
  const int tok0 = p.seq_off[b];
  const int L = p.seq_off[b + 1] - tok0;
  const int t0 = ck * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int nh = min(16, p.H - hg * 16);
  float* stat = p.stat + (((size_t)m * p.nch + ck) * p.H + hg * 16) * 2;
  if (t0 >= L) {     // this row has no keys here: mark the chunk empty for the combine step
    if (tid < nh) { stat[tid * 2] = -1e30f; stat[tid * 2 + 1] = 0.f; }
    return;
  }
  // ---- scores: S[h][t] = qk[h] . e_t  by MFMA 16x16x32 (A = the 16 head rows of qk, B = 16 encoder rows per wave) ----
  const int hrow = min(hg * 16 + l15, p.H - 1);
  const half_t* ap = p.qk + ((size_t)m * p.H + hrow) * p.d + 8 * g;
  const int t = t0 + wave * 16 + l15;
  const half_t* bp = p.enc + (size_t)(tok0 + min(t, L - 1)) * p.d + 8 * g;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (FEW) {
    for (int c1 = 0; c1 < p.d; c1 += 512) {                    // sixteen k32 steps of operand loads in flight (128 registers)
      half8 av[16], bv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int c0 = c1 + 32 * u < p.d ? c1 + 32 * u : c1;  // (d is a multiple of 32; a short tail re-reads the first step, unused)
        av[u] = *(const half8*)(ap + c0); bv[u] = *(const half8*)(bp + c0);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (c1 + 32 * u < p.d) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[u], bv[u], acc, 0, 0, 0);
    }
  } else {
#pragma unroll 4
    for (int c0 = 0; c0 < p.d; c0 += 32)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const half8*)(ap + c0), *(const half8*)(bp + c0), acc, 0, 0, 0);
  }
  // lane holds heads 4g..4g+3 for key t (C layout: col = lane&15, row = 4*(lane>>4) + r)
  float mx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (t >= L) acc[r] = -1e30f;
    mx[r] = row16_max(acc[r]);
  }
  if (l15 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sRed[0][wave][4 * g + r] = mx[r];
  }
  __syncthreads();
  float sm[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int h = 4 * g + r;
    mx[r] = fmaxf(fmaxf(sRed[0][0][h], sRed[0][1][h]), fmaxf(sRed[0][2][h], sRed[0][3][h]));
    const float e = __expf(acc[r] - mx[r]);
    sP[h * XAM_PSTR + wave * 16 + l15] = (half_t)e;            // [head][key of the chunk]
    sm[r] = row16_sum_f(e);
  }
  if (l15 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sRed[1][wave][4 * g + r] = sm[r];
  }
  __syncthreads();
  if (tid < nh) {
    stat[tid * 2] = fmaxf(fmaxf(sRed[0][0][tid], sRed[0][1][tid]), fmaxf(sRed[0][2][tid], sRed[0][3][tid]));
    stat[tid * 2 + 1] = (sRed[1][0][tid] + sRed[1][1][tid]) + (sRed[1][2][tid] + sRed[1][3][tid]);
  }
  // ---- partial weighted sums: O[h][c] = sum_key P[h][key] E[key][c] on the matrix cores ----
  float* part = p.part + (((size_t)m * p.nch + ck) * p.H + hg * 16) * p.d;
  half8 pa[2];                                                 // P fragments of the two k32 steps: head l15, keys 32 kk + 8 g .. +8
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) pa[kk] = *(const half8*)(sP + l15 * XAM_PSTR + 32 * kk + 8 * g);
  half_t* sE0 = (half_t*)xam_smem + wave * (NPIECE * 64 * 64);   // this wave's [64 keys][64 columns] piece(s)
  // DMA piece i fills LDS slots i * 64 + lane of the piece: key row r = slot >> 3, 16-byte chunk c = slot & 7; it fetches
  // global chunk c ^ (F(r) << 1).  Byte offsets of the eight source rows from the piece's first column:
  unsigned soff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int slot = i * 64 + lane, r = slot >> 3, c = slot & 7;
    soff[i] = ((unsigned)(tok0 + min(t0 + r, L - 1)) * (unsigned)p.d + ((c ^ (XAM_F(r) << 1)) << 3)) * 2u;   // clamped rows carry weight 0
  }
  // this lane's share of a transposing read: key 8 g + (i16 >> 2) (+ 32 kk, + 4 for the second read), columns 4 (i16 & 3)
  // .. +4 of a 16-column block; F depends on key bits 1 and 3 only: (kj >> 1) | (g & 1) << 1
  const int i16 = lane & 15, kj = i16 >> 2;
  const int fsw = ((kj >> 1) & 1) | ((g & 1) << 1);
  const int wcols = p.d >> 2;                                  // columns of this wave
  const unsigned lds_e = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) half_t*)sE0);
  auto fetch_piece = [&](int cs, int buf) {
    const char* src = (const char*)(p.enc + wave * wcols + cs);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (FEW) {
        // (s_nop 4 in front: a base pair that comes out of a spill lane is a VALU-written SGPR, five wait states before a VMEM read)
        const unsigned dst = lds_e + (unsigned)(buf * (64 * 64) + i * 512) * 2u;
        const unsigned o = soff[i];
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(o), "s"(src) : "memory", "m0");
#pragma clang diagnostic pop
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + soff[i]),
                                         (__attribute__((address_space(3))) void*)(sE0 + buf * (64 * 64) + i * 512), 16, 0, 0);
      }
    }
  };
  fetch_piece(0, 0);
  int buf = 0;
  for (int cs = 0; cs < wcols; cs += 64) {
    if constexpr (FEW) {
      // the NEXT piece travels while this one is multiplied (its buffer was last read two iterations ago: lgkmcnt(0) below).  Counted
      // wait: the eight instructions of the next piece may stay in flight - LDS-DMA instructions retire in order among themselves, and
      // the partial-sum stores in between only make the count stricter
      if (cs + 64 < wcols) {
        fetch_piece(cs + 64, buf ^ 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      if (cs > 0) fetch_piece(cs, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's piece has landed (wave-private: no barrier)
    }
    half_t* sE = sE0 + buf * (64 * 64);                        // (wave-private: no barrier)
#pragma unroll
    for (int cb = 0; cb < 64; cb += 16) {
      const int col = cb + 4 * (i16 & 3);                      // column (within the piece) this lane's read starts at
      const int coff = ((((col >> 4) ^ fsw) << 4) | (col & 15));
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const half_t* r0 = sE + (32 * kk + 8 * g + kj) * 64 + coff;
        const half4 e0 = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) attd_fp16x4*)r0));
        const half4 e1 = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) attd_fp16x4*)(r0 + 4 * 64)));
        const half8 eb = {e0[0], e0[1], e0[2], e0[3], e1[0], e1[1], e1[2], e1[3]};
        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[kk], eb, o, 0, 0, 0);
      }
      const int cout = wave * wcols + cs + cb + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r < nh) part[(size_t)(4 * g + r) * p.d + cout] = o[r];
    }
    if constexpr (FEW) buf ^= 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every read of the piece has returned before the next DMA overwrites it
  }
}

// grid = (H, M); 256 threads: merge the chunks of one (row, head) in chunk order and normalise.  The chunks that hold keys
// are exactly 0 .. ceil(L / 64) - 1, so the loops run over that prefix without per-chunk branches: the weights go through
// LDS once and the partial-sum loads of a thread are independent (16 in flight) - the branchy form paid one L2 round trip
// per chunk (17 us for 23 chunks of a 1.4k-token prompt).
__global__ __launch_bounds__(256) void xattn_combine_kernel(XAttnArgs p) {
  __shared__ float sW[1024], sS[1024];
  const int h = blockIdx.x, m = blockIdx.y, tid = threadIdx.x;
  const int b = p.row_seq ? p.row_seq[p.row0 + m] : (p.row0 + m) / p.Ld;
  const int L = p.seq_off[b + 1] - p.seq_off[b];
  const int nv = min(p.nch, (L + 63) >> 6);            // chunks with at least one key
  const float* stat = p.stat + ((size_t)m * p.nch * p.H + h) * 2;
  const size_t sstride = (size_t)p.H * 2;
  // (nv <= 1024: the host refuses sequences of more than 65536 tokens on this path)
  // the chunk statistics come in with one load per thread (a serial loop over them was one L2 round trip per few chunks);
  // the max and the denominator are then formed from LDS by every thread in chunk order, as before
  for (int i = tid; i < nv; i += 256) {
    const float2 ms = *(const float2*)(stat + i * sstride);
    sW[i] = ms.x; sS[i] = ms.y;
  }
  __syncthreads();
  float gmax = -1e30f;
  for (int ck = 0; ck < nv; ++ck) gmax = fmaxf(gmax, sW[ck]);
  float den = 0.f;
  for (int ck = 0; ck < nv; ++ck) den += __expf(sW[ck] - gmax) * sS[ck];
  const float inv = 1.0f / den;
  const float* part = p.part + ((size_t)m * p.nch * p.H + h) * p.d;
  const size_t pstride = (size_t)p.H * p.d;
  __syncthreads();                                     // everyone has read the maxima: sW becomes the weights
  for (int i = tid; i < nv; i += 256) sW[i] = __expf(sW[i] - gmax);
  __syncthreads();
  for (int cb = tid * 4; cb < p.d; cb += 1024) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < nv; i += 8) {                  // 8 chunk rows in flight; the last round is predicated, not serial
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(part + (size_t)min(i + u, nv - 1) * pstride + cb);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i + u < nv) { const float w = sW[i + u]; acc[0] += w * v[u][0]; acc[1] += w * v[u][1]; acc[2] += w * v[u][2]; acc[3] += w * v[u][3]; }
    }
    half4 o = {f2h_sat(acc[0] * inv), f2h_sat(acc[1] * inv), f2h_sat(acc[2] * inv), f2h_sat(acc[3] * inv)};
    *(half4*)(p.out + ((size_t)m * p.H + h) * p.d + cb) = o;
  }
}
