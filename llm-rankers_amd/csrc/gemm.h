// fp16 MFMA GEMM for gfx950:  C[m][n] (+)= sum_k A[m][k] * W[n][k]     (y = x W^T, nn.Linear without bias)
//
// Replaces the q/k/v/o and wi/wo Linear calls of hf: models/t5/modeling_t5.py:206-209, 106-123 that the
// reference reaches through self.llm(...) (ref: llmrankers/pointwise.py:117-119).
//
// Design (CDNA4): 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile as 2x2 MFMA 32x32x16 f16
// fragments with fp32 accumulators.  The WEIGHT tile is the MFMA A operand and the ACTIVATION tile the B
// operand, so a lane's 4 consecutive accumulator registers are 4 consecutive output columns n of one row m
// (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) -> 8/16-byte epilogue stores.
// Both tiles are K-contiguous in memory and are staged into LDS as [128 rows][64 halfs] (128-B rows) with the
// 16-B chunk index XOR-swizzled by (row>>1)&7, which makes every ds_read_b128 fragment read conflict-free
// (a 256-B bank row holds two tile rows; rows 2j,2j+1 share a swizzle value but sit in different halves).
// Staging is either direct-to-LDS DMA (global_load_lds_dwordx4; LDS image is lane-linear so the swizzle is
// applied to the per-lane SOURCE address) or through registers; double-buffered, one barrier per K step.
//
// Epilogues fuse what follows the Linear in T5: fp32 residual add (o / wo projections), GEGLU
// (gelu_new(wi_0 x) * wi_1 x with wi_0/wi_1 rows interleaved in groups of 32 by the weight packer), ReLU.
#pragma once
#include "common.h"
#include <type_traits>

enum GemmEpi { EPI_STORE_F16 = 0, EPI_RESID_F32 = 1, EPI_GEGLU_F16 = 2, EPI_RELU_F16 = 3, EPI_STORE_F32 = 4, EPI_SWIGLU_F16 = 5,
               EPI_ARGMAX_F32 = 6,     // weight-streaming kernel only: per 32-column block the row maximum and its first index
               EPI_LSE_F32 = 7 };      // tiled kernels only (qlm head): per 32-column block (max, sum exp(x - max)) + the label's logit
// gated epilogues: gate / up rows interleaved in groups of 32 by the weight packer, out = act(gate) * up
#define EPI_IS_GATED(E) ((E) == EPI_GEGLU_F16 || (E) == EPI_SWIGLU_F16)

struct GemmArgs {
  const half_t* A;   // [M, K] activations, row stride lda (halfs)
  const half_t* W;   // [N, K] weights, row stride ldw
  void* C;           // output, row stride ldc (elements of the output type)
  int lda, ldw, ldc;
  int M, N, K;
  int n_split;       // >0: output column n goes to block n / n_split at C + block*split_stride, column n % n_split
  long split_stride; // in elements
  float scale;       // multiplies the accumulator (1.0 normally)
  long bsA, bsW, bsC; // skinny kernel only: element strides between the blockIdx.y batches (per-head GEMMs)
  // folded RMSNorm.  Consumer: A is the un-normalised stream as fp16 x xs, W has the norm weight
  // folded into its columns, rowscale[m] = rsqrt(mean(x_m^2) + eps) / xs multiplies the accumulators of row m.
  const float* rowscale;
  // Producer (fp32 residual epilogue): besides C += acc, write fp16(C x xs) to xraw [M, ldx] and the sums of squares of
  // the new rows per 64-column block to ssq [M, nb]
  half_t* xraw; float* ssq; int ldx, nb; float xs;
  // Weight-streaming (decoder) form of the same fold: the consumer has no separate statistics kernel in front of it - it adds
  // the nb_in block sums of ssq_in [M, nb_in] (written by the producer GEMM before it, 32-column blocks) itself:
  // row factor = rsqrt(sum / K + eps_in) / xs
  const float* ssq_in; int nb_in; float eps_in;
  // EPI_ARGMAX_F32 (greedy head): C = float [M, ldc] block maxima, amax_idx = int [M, ldc] their first column (ldc = blocks)
  int* amax_idx;
  // EPI_LSE_F32 (qlm head): C = float2 [M, ldc] (block max, sum of exp(x - block max)) per 32-column block (ldc = blocks);
  // row m scores label lse_labels[m % lse_npos], whose logit goes to lse_xlab[m].  The logits never reach memory.
  const int* lse_labels; int lse_npos; float* lse_xlab;
  // ping-pong kernel: width (in tiles) of the column panels of the grouped tile order (gemm_tile_coords); host default GEMM_GROUP_N
  int group_n;
  // ping-pong kernel, K split over workgroups (SPLIT instantiations; gemm_pp2_kernel): ksplit >= 2 workgroups share an output tile,
  // each sums a contiguous K range; fp32 partial tiles in ks_slabs [ksplit * tiles][256 * 256], arrival tickets ks_cnt [tiles]
  // (zero between launches: the last arriver of a tile resets its ticket)
  int ksplit; float* ks_slabs; int* ks_cnt;
};

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_TILE_HALFS (128 * 64)
#define GEMM_LDS_BYTES (2 * 2 * GEMM_TILE_HALFS * 2)   // 2 stages x {A,W} x 16 KiB = 64 KiB

__device__ __forceinline__ float gelu_new_f(float x) {
  // hf: activations.py:59-66:  0.5 x (1 + tanh(u)),  u = sqrt(2/pi) (x + 0.044715 x^3).
  // 0.5 (1 + tanh u) == 1 / (1 + exp(-2u)) exactly; evaluated with the hardware exp2 / rcp (about 1e-6 relative,
  // far below the fp16 rounding of the result) instead of libm tanhf, which cost ~200 cycles per output here.
  const float u2 = -1.5957691216057308f * (x + 0.044715f * x * x * x);   // -2u
  // (the raw v_rcp_f32, 1 ulp: __frcp_rn expands to the ten-instruction IEEE division sequence on this target)
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(u2));
}

__device__ __forceinline__ float silu_f(float x) {
  // hf: activations.py SiLU (Llama's hidden_act): x * sigmoid(x), on the hardware exp2 / rcp like gelu_new_f
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
template <int EPI>
__device__ __forceinline__ float gate_act(float x) { return EPI == EPI_SWIGLU_F16 ? silu_f(x) : gelu_new_f(x); }

template <bool GLDS>
__device__ __forceinline__ void gemm_stage_tile(half_t* s_tile, const half_t* g, int ld, int row0, int rows_total,
                                                int k0, int wave, int lane, half8 (&regs)[4]) {
  // tile = 1024 16-byte slots; slot p = (row r = p>>3, chunk c = p&7).  Wave w owns slots [w*256, w*256+256).
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = (wave * 4 + i) * 64 + lane;
    const int r = p >> 3, c = p & 7;
    int grow = row0 + r;
    grow = grow < rows_total ? grow : rows_total - 1;   // clamp: rows beyond the matrix are never stored
    if (GLDS) {
      // LDS slot p receives global chunk c ^ f(r): the DMA writes lane-linear, so permute the source.
      const half_t* src = g + (size_t)grow * ld + k0 + ((c ^ ((r >> 1) & 7)) << 3);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(s_tile + (wave * 4 + i) * 512),
                                       16, 0, 0);
    } else {
      regs[i] = *(const half8*)(g + (size_t)grow * ld + k0 + (c << 3));
    }
  }
}

__device__ __forceinline__ void gemm_write_tile(half_t* s_tile, int wave, int lane, const half8 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = (wave * 4 + i) * 64 + lane;
    const int r = p >> 3, c = p & 7;
    *(half8*)(s_tile + r * 64 + ((c ^ ((r >> 1) & 7)) << 3)) = regs[i];
  }
}

__device__ __forceinline__ half8 gemm_frag(const half_t* s_tile, int r, int cc) {
  return *(const half8*)(s_tile + r * 64 + ((cc ^ ((r >> 1) & 7)) << 3));
}

// XCD-aware, L2-friendly tile order.  Block b runs on XCD b%8 and an XCD starts its blocks in increasing b, about
// one (v2) or two (v1) per CU at a time.  Each XCD is handed a contiguous run of a GROUPED tile list - column panels
// of GROUP_N tiles, row-major inside a panel - so the ~32 workgroups co-resident on one XCD form a compact
// (32/GROUP_N rows x GROUP_N cols) block of the output and share A / W panels through that XCD's private L2.
// Bijective for any grid size.
#define GEMM_GROUP_N 8
__device__ __forceinline__ void gemm_tile_coords(int bid, int tiles_m, int tiles_n, int& tm, int& tn, int group_n = GEMM_GROUP_N) {
  const int nwg = tiles_m * tiles_n;
  const int xcd = bid & 7, loc = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int per_panel = tiles_m * group_n;
  const int pn = lin / per_panel, rem = lin - pn * per_panel;
  const int w = min(group_n, tiles_n - pn * group_n);
  tm = rem / w;
  tn = pn * group_n + rem - tm * w;
}

// Epilogue shared by the tiled and the skinny kernel.  acc[ni][mi] are 32x32 MFMA C fragments of a 64(n) x 64(m)
// wave tile at (mbase, nbase); valid_mi/valid_ni limit the fragments a caller actually computed.
// The lane holds row m = mbase + mi*32 + l31 and, per register group q, 4 consecutive columns n.
template <int EPI, int NI = 2, int MI = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[NI][MI], int mbase, int nbase, int l31, int hh,
                                              float rowfac = 1.f) {
  const float sc = p.scale * rowfac;    // rowfac: folded-RMSNorm factor of this lane's row (MI == 1 callers), else 1
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mbase + mi * 32 + l31;
    if (m >= p.M) continue;
    if (EPI_IS_GATED(EPI)) {
      // acc[0] = gate rows, acc[1] = up rows of the same 32 output columns
      half_t* C = (half_t*)p.C;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = (nbase >> 1) + 8 * q + 4 * hh;
        if (col >= (p.N >> 1)) continue;
        half4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = f2h_sat(gate_act<EPI>(acc[0][mi][4 * q + j] * sc) * (acc[NI - 1][mi][4 * q + j] * sc));
        *(half4*)(C + (size_t)m * p.ldc + col) = o;
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int n = nbase + ni * 32 + 8 * q + 4 * hh;
          if (n >= p.N) continue;
          size_t base = 0;
          if (p.n_split > 0) {
            base = (size_t)(n / p.n_split) * (size_t)p.split_stride;
            n = n % p.n_split;
          }
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * q + j] * sc;
          if (EPI == EPI_STORE_F16 || EPI == EPI_RELU_F16) {
            half4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f2h_sat(EPI == EPI_RELU_F16 ? fmaxf(v[j], 0.f) : v[j]);
            *(half4*)((half_t*)p.C + base + (size_t)m * p.ldc + n) = o;
          } else if (EPI == EPI_RESID_F32) {
            float* c = (float*)p.C + base + (size_t)m * p.ldc + n;
            f32x4 old = *(f32x4*)c;
            f32x4 o = {old[0] + v[0], old[1] + v[1], old[2] + v[2], old[3] + v[3]};
            *(f32x4*)c = o;
          } else {  // EPI_STORE_F32
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *(f32x4*)((float*)p.C + base + (size_t)m * p.ldc + n) = o;
          }
        }
      }
    }
  }
}

// LDS-staged epilogue for the tiled kernels.  A lane's accumulators are 4-column pieces of 32 different rows, so
// storing them directly makes every store instruction touch 32+ cache lines with 8-byte pieces (measured: ~10 us
// of a 35 us tile).  Instead each wave drops one 32-row slab (32 x NI*32 outputs) at a time into a wave-private LDS
// region (the main-loop stages are dead by then) and writes it back with 16 bytes per lane along the rows, i.e.
// whole 64..256-byte row segments per instruction; the fp32 residual add reads the old row the same way.
// `stage` = this wave's LDS region, at least 32 * (NI*32*4 + 16) bytes.
// Residual epilogue, read side: the old fp32 rows are fetched in ACCUMULATOR layout at kernel start, so the read half
// of the read-modify-write travels while the main loop computes instead of after it (every tile of these single-round
// launches reaches its epilogue at the same moment, so un-overlapped epilogue traffic is pure added latency).
template <int NI, int MI>
__device__ __forceinline__ void gemm_prefetch_residual(const GemmArgs& p, f32x16 (&res)[NI][MI], int mbase, int nbase,
                                                       int l31, int hh) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int m = mbase + mi * 32 + l31;
    m = m < p.M ? m : p.M - 1;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int n = nbase + ni * 32 + 8 * q + 4 * hh;
        n = n < p.N ? n : p.N - 4;
        const f32x4 v = *(const f32x4*)((const float*)p.C + (size_t)m * p.ldc + n);
        res[ni][mi][4 * q] = v[0]; res[ni][mi][4 * q + 1] = v[1]; res[ni][mi][4 * q + 2] = v[2]; res[ni][mi][4 * q + 3] = v[3];
      }
  }
}

// ROWS = 32: one pass per 32-row slab; ROWS = 16: two passes of 16 rows (half the staging bytes per wave - the
// persistent ping-pong kernel stages beside the next tile's DMA targets and has only 48 KiB for it).
// rsc[mi]: the factor this lane's row of slab mi is multiplied by (p.scale, times the consumer-side RMSNorm row factor
// when the GEMM reads the un-normalised stream - GemmArgs::rowscale).
// DEPTH (fp32 residual epilogue): the OLD rows of DEPTH slabs ahead are already being fetched while a slab is staged,
// added and stored - the read half of the read-modify-write is a chain of HBM round trips otherwise (a wave has one
// slab = 4 loads of 1 KiB in flight, 32 KiB per CU: 3 TB/s over the chip however little else is running).
// Producer side of the folded RMSNorm (p.xraw != nullptr, fp32 residual epilogue only): next to the new fp32 rows the
// epilogue writes them once more as fp16 x p.xs (the next GEMM's A operand) and the sum of their squares per row and
// 64-column block (one wave = one block; fixed in-lane + DPP order, so any tile shape writes the same bits).
// qlm head (ref: llmrankers/pointwise.py:73-79: logits -> CrossEntropyLoss): instead of [rows, vocab] fp32 logits (written and
// read back: 2 x 4 x vocab bytes per row) the epilogue leaves, per row and 32-column MFMA fragment, the fragment's maximum
// and sum of exp(x - maximum), plus the logit of the row's label; qlm_lse_kernel (misc_kernels.h) merges the blocks of a
// row in block order.  A lane holds 16 of a fragment's 32 columns of its row, its partner (lane ^ 32) the others: fixed
// in-lane order + one exchange, so every tile shape writes the same bits.
template <int NI, int MI>
__device__ __forceinline__ void gemm_epilogue_lse(const GemmArgs& p, f32x16 (&acc)[NI][MI], int mbase, int nbase, int lane,
                                                  const float (&rsc)[MI]) {
  const int l31 = lane & 31, hh = lane >> 5;
  float2* stats = (float2*)p.C;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mbase + mi * 32 + l31;
    const bool row_ok = m < p.M;
    const int label = row_ok ? p.lse_labels[m % p.lse_npos] : -1;
    const float sc = rsc[mi];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n0 = nbase + ni * 32;
      if (n0 >= p.N) continue;                              // uniform
      float v[16], mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        v[r] = n < p.N ? acc[ni][mi][r] * sc : -INFINITY;
        mx = fmaxf(mx, v[r]);
        if (n == label) p.lse_xlab[m] = v[r];
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float se = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) se += __expf(v[r] - mx);
      se += __shfl_xor(se, 32);
      if (hh == 0 && row_ok) stats[(size_t)m * p.ldc + (n0 >> 5)] = make_float2(mx, se);
    }
  }
}

// Cache policy of the epilogue's memory traffic.  The outputs of the big GEMMs (120 - 360 MB per launch) and the old rows of the
// fp32 stream are touched once per launch; written / read with the default policy they pass through the 4 MiB L2 of the XCD
// and displace the A / W panels the co-resident tiles share.  Non-temporal hint (`nt`) on: bit 0 the read of the old fp32
// rows, bit 1 the fp32 rows written back, bit 2 the fp16 tile stores (QKV, GEGLU / ReLU outputs), bit 3 the fp16 copy of the
// stream.  Measured round 3 (separate libraries, one box, tools/ab_profile.py): bits 0-1 +0.4 %, bits 0-2 +1.6 % (7 485 ->
// 7 600 passages/s; QKV 0.864 -> 0.818, FFN-in 1.56 -> 1.51 ms per step), bit 3 nothing.  The LOADS of the panels must stay
// cacheable: `nt` on the A-panel DMA -4 %, on the W-panel DMA -8 % (GEMM_A_AUX / GEMM_W_AUX below); and the attention kernel's
// context stores (16-byte pieces, not whole lines) got 20 % slower with it.
#ifndef GEMM_EPI_NT
#define GEMM_EPI_NT 7
#endif
// timing-only knock-outs of the residual epilogue (A/B libraries of tools/ only; results are garbage): bit 0 the old fp32 rows are not
// read, 1 the new fp32 rows are not written, 2 the fp16 stream copy and its sums of squares are not written
#ifndef GEMM_EPI_KO
#define GEMM_EPI_KO 0
#endif
template <int EPI, int NI, int MI, bool RESID_IN_ACC = false, int ROWS = 32, int DEPTH = 0>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmArgs& p, f32x16 (&acc)[NI][MI], int mbase, int nbase,
                                                     int lane, unsigned char* stage, const float (&rsc)[MI]) {
  if constexpr (EPI == EPI_LSE_F32) {
    gemm_epilogue_lse<NI, MI>(p, acc, mbase, nbase, lane, rsc);
    return;
  }
  const int l31 = lane & 31, hh = lane >> 5;
  constexpr bool GEGLU = EPI_IS_GATED(EPI);
  constexpr bool F32 = EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32;
  constexpr bool RMW = EPI == EPI_RESID_F32 && !RESID_IN_ACC;  // the old rows are read here (not pre-added by the caller)
  constexpr int COLS = GEGLU ? NI * 16 : NI * 32;            // output columns of this wave
  constexpr int ELT = F32 ? 4 : 2;
  // padded LDS row (bytes).  fp32 rows: +16 (16-B accesses).  fp16 rows: +8 - a row stride of 8 x odd bytes puts the 32
  // lanes' 8-byte pieces on 32 different bank pairs (with +16 the stride is a multiple of 4 banks: 2-way conflicts,
  // 2-3 % of the kernel's cycles in PMC); rows are then only 8-B aligned, so they are read back as two 8-byte halves
  constexpr int ROWB = COLS * ELT + (F32 ? 16 : 8);
  constexpr int CHUNKS = COLS * ELT / 16;                      // 16-B pieces per row
  constexpr int ROWS_PER_PASS = 64 / CHUNKS;
  constexpr int HP = 32 / ROWS, NSLAB = MI * HP;
  constexpr int PASSES = (ROWS + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
  constexpr int NPRE = RMW ? NSLAB : 1;
  const int ncol0 = GEGLU ? (nbase >> 1) : nbase;
  const int nlimit = GEGLU ? (p.N >> 1) : p.N;
  const bool fold_out = EPI == EPI_RESID_F32 && p.xraw != nullptr;     // uniform
  // this lane's place in the write-back passes: row (within a pass) and 16-byte chunk
  const int prow = lane / CHUNKS, ch = lane % CHUNKS;
  const bool lane_on = prow < ROWS_PER_PASS;
  int n = ncol0 + ch * (16 / ELT);
  const bool n_ok = n < nlimit;
  size_t cbase = 0;
  if (p.n_split > 0) { cbase = (size_t)(n / p.n_split) * (size_t)p.split_stride; n = n % p.n_split; }
  f32x4 oldv[NPRE][PASSES];
  auto fetch_old = [&](int sl) {       // sl is a compile-time constant after unrolling
    const int mi = sl / HP, hp = sl % HP;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int row = i * ROWS_PER_PASS + prow;
      int m = mbase + mi * 32 + hp * ROWS + row;
      m = m < p.M ? m : p.M - 1;                                 // clamped rows are never stored
      const int nn = n_ok ? n : 0;
      const f32x4* src_ = (const f32x4*)((const float*)p.C + cbase + (size_t)m * p.ldc + nn);
      if constexpr ((GEMM_EPI_KO & 1) != 0) oldv[RMW ? sl : 0][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      else if constexpr (GEMM_EPI_NT != 0) oldv[RMW ? sl : 0][i] = __builtin_nontemporal_load(src_);
      else oldv[RMW ? sl : 0][i] = *src_;
    }
  };
  if constexpr (RMW && DEPTH > 0) {
#pragma unroll
    for (int sl = 0; sl < (DEPTH < NSLAB ? DEPTH : NSLAB); ++sl) fetch_old(sl);
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
  for (int hp = 0; hp < HP; ++hp) {
    const int sl = mi * HP + hp;
    if constexpr (RMW) {
      if constexpr (DEPTH > 0) { if (sl + DEPTH < NSLAB) fetch_old(sl + DEPTH); }
      else fetch_old(sl);
    }
    const float sc = rsc[mi];
    // ---- registers -> LDS (row l31 of the slab; with ROWS = 16 the lanes of the other half sit this pass out) ----
    unsigned char* myrow = stage + (l31 % ROWS) * ROWB;
    if (ROWS == 32 || (l31 / ROWS) == hp) {
    if (GEGLU) {
#pragma unroll
      for (int g = 0; g < NI / 2; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          half4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = f2h_sat(gate_act<EPI>(acc[2 * g][mi][4 * q + j] * sc) * (acc[2 * g + 1][mi][4 * q + j] * sc));
          *(half4*)(myrow + (g * 32 + 8 * q + 4 * hh) * 2) = o;
        }
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = ni * 32 + 8 * q + 4 * hh;
          if (F32) {
            f32x4 o = {acc[ni][mi][4 * q] * sc, acc[ni][mi][4 * q + 1] * sc, acc[ni][mi][4 * q + 2] * sc,
                       acc[ni][mi][4 * q + 3] * sc};
            *(f32x4*)(myrow + c * 4) = o;
          } else {
            half4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float v = acc[ni][mi][4 * q + j] * sc;
              o[j] = f2h_sat(EPI == EPI_RELU_F16 ? fmaxf(v, 0.f) : v);
            }
            *(half4*)(myrow + c * 2) = o;
          }
        }
    }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes are done (region is wave-private)
    __builtin_amdgcn_wave_barrier();
    // ---- LDS -> global, 16 B per lane along the rows ----
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int row = i * ROWS_PER_PASS + prow;
      const int m = mbase + mi * 32 + hp * ROWS + row;
      const bool ok = lane_on && row < ROWS && m < p.M && n_ok;
      const unsigned char* sp = stage + (row < ROWS ? row : 0) * ROWB + ch * 16;
      if (F32) {
        f32x4 v = *(const f32x4*)sp;
        if constexpr (RMW) { const f32x4 old = oldv[sl][i]; v = {old[0] + v[0], old[1] + v[1], old[2] + v[2], old[3] + v[3]}; }
        if (ok && !(RMW && (GEMM_EPI_KO & 2) != 0)) {
          f32x4* dst_ = (f32x4*)((float*)p.C + cbase + (size_t)m * p.ldc + n);
          if constexpr (RMW && (GEMM_EPI_NT & 2) != 0) __builtin_nontemporal_store(v, dst_);
          else *dst_ = v;
        }
        if constexpr (EPI == EPI_RESID_F32 && CHUNKS == 16) {   // one wave = one 64-column block (the host never picks the 192-wide tile here)
          if (fold_out) {
            // explicit fma chain: left to the compiler, a*a + b*b contracts differently in different instantiations of
            // this epilogue and the row statistics (hence every score) would depend on the tile shape
            float ss = ok ? __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], v[0] * v[0]))) : 0.f;
            ss = row16_sum_f(ss);                                // the 16 lanes of one row
            if (ok && (GEMM_EPI_KO & 4) == 0) {
              half4 xr = {f2h_sat(v[0] * p.xs), f2h_sat(v[1] * p.xs), f2h_sat(v[2] * p.xs), f2h_sat(v[3] * p.xs)};
              if constexpr ((GEMM_EPI_NT & 8) != 0) __builtin_nontemporal_store(xr, (half4*)(p.xraw + (size_t)m * p.ldx + n));
              else *(half4*)(p.xraw + (size_t)m * p.ldx + n) = xr;
              if (ch == 0) p.ssq[(size_t)m * p.nb + (ncol0 >> 6)] = ss;
            }
          }
        }
      } else if (ok) {
        const half4 lo = *(const half4*)sp, hi = *(const half4*)(sp + 8);
        const half8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if constexpr ((GEMM_EPI_NT & 4) != 0) __builtin_nontemporal_store(v8, (half8*)((half_t*)p.C + cbase + (size_t)m * p.ldc + n));
        else *(half8*)((half_t*)p.C + cbase + (size_t)m * p.ldc + n) = v8;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();      // slab fully read before the next one overwrites it
  }
}

// Row factors of a wave's MI slabs for the staged epilogue: p.scale, times the RMSNorm row factor when the GEMM reads the
// un-normalised stream (GemmArgs::rowscale; written by rowscale_kernel / embed_gather_kernel).
template <int MI>
__device__ __forceinline__ void gemm_row_factors(const GemmArgs& p, int mbase, int l31, float (&rsc)[MI]) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    rsc[mi] = p.scale;
    const int m = mbase + mi * 32 + l31;
    if (p.rowscale) rsc[mi] *= p.rowscale[m < p.M ? m : p.M - 1];
    else if (p.ssq_in) rsc[mi] *= rk_row_factor(p.ssq_in + (size_t)(m < p.M ? m : p.M - 1) * p.nb_in, p.nb_in, p.K, p.eps_in, p.xs);   // no statistics kernel ran (small M)
  }
}

template <int EPI, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  half_t* smem = (half_t*)gemm_smem;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int tiles_m = (p.M + GEMM_BM - 1) / GEMM_BM, tiles_n = (p.N + GEMM_BN - 1) / GEMM_BN;
  int tm, tn;
  gemm_tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / GEMM_BK;
  f32x16 res[2][2];
  if (EPI == EPI_RESID_F32) gemm_prefetch_residual<2, 2>(p, res, m0 + wm * 64, n0 + wn * 64, l31, hh);
  half8 ra[4], rw[4];
  // prologue: stage k-tile 0 into stage 0
  gemm_stage_tile<GLDS>(smem, p.A, p.lda, m0, p.M, 0, wave, lane, ra);
  gemm_stage_tile<GLDS>(smem + GEMM_TILE_HALFS, p.W, p.ldw, n0, p.N, 0, wave, lane, rw);
  if (!GLDS) {
    gemm_write_tile(smem, wave, lane, ra);
    gemm_write_tile(smem + GEMM_TILE_HALFS, wave, lane, rw);
  }
  for (int kt = 0; kt < nk; ++kt) {
    half_t* sA = smem + (kt & 1) * 2 * GEMM_TILE_HALFS;
    half_t* sW = sA + GEMM_TILE_HALFS;
    half_t* nA = smem + ((kt + 1) & 1) * 2 * GEMM_TILE_HALFS;
    half_t* nW = nA + GEMM_TILE_HALFS;
    __syncthreads();   // stage kt landed (glds drained by the barrier's vmcnt(0)); stage kt^1 free to overwrite
    const bool more = kt + 1 < nk;
    if (more) {
      gemm_stage_tile<GLDS>(nA, p.A, p.lda, m0, p.M, (kt + 1) * GEMM_BK, wave, lane, ra);
      gemm_stage_tile<GLDS>(nW, p.W, p.ldw, n0, p.N, (kt + 1) * GEMM_BK, wave, lane, rw);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cc = ks * 2 + hh;
      half8 wf[2], af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wf[i] = gemm_frag(sW, wn * 64 + i * 32 + l31, cc);
        af[i] = gemm_frag(sA, wm * 64 + i * 32 + l31, cc);
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
    if (!GLDS && more) {
      gemm_write_tile(nA, wave, lane, ra);
      gemm_write_tile(nW, wave, lane, rw);
    }
  }

  if (EPI == EPI_RESID_F32) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * p.scale + res[i][j][r];
  }
  __syncthreads();   // every wave is done reading the last stage: LDS becomes the epilogue staging area
  if (EPI == EPI_RESID_F32) {
    GemmArgs q = p; q.scale = 1.f;
    float rsc[2] = {1.f, 1.f};
    gemm_epilogue_staged<EPI, 2, 2, true>(q, acc, m0 + wm * 64, n0 + wn * 64, lane, gemm_smem + wave * (32 * (64 * 4 + 16)), rsc);
  } else {
    float rsc[2];
    gemm_row_factors<2>(p, m0 + wm * 64, l31, rsc);
    gemm_epilogue_staged<EPI, 2, 2>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, gemm_smem + wave * (32 * (64 * 4 + 16)), rsc);
  }
}

// ================================= small-M variant: 64x64 tiles ===============================================
// One setwise prompt is M ~ 1.5k rows: 256- or 128-row tiles give the O / FFN-out projections 24 - 96 workgroups for 256
// CUs.  This kernel cuts the output into 64x64 tiles (2 waves, each 32 rows x 64 columns = two MFMA 32x32 fragments) with
// the same LDS image, swizzle, DMA staging, K order and epilogues as the 128x128 kernel - so a row's result is bit-identical
// whichever variant runs.  Latency-bound by design: the point is 16x more workgroups in flight, not MFMA rate.
// NST LDS stages of 16 KiB (A + W tile): NST - 1 K tiles are in flight while one is consumed, waited for with counted vmcnt
// (every wave issues exactly 8 DMA instructions per stage).  With all tiles of such a GEMM resident at once the launch lasts
// nk x (time of one K step), and a K step of a 2-stage loop is one full L2 round trip (~1.2 us measured): 3 / 4 stages
// divide that by 2 / 3 at 48 / 64 KiB of LDS (three / two workgroups per CU).
template <int EPI, int NST>
__global__ __launch_bounds__(128, NST == 4 ? 2 : (NST == 3 ? 3 : 4)) void gemm_s64_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  half_t* smem = (half_t*)gemm_smem;
  constexpr int TILE = 64 * 64;                         // halfs per operand tile (8 KiB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int tiles_m = (p.M + 63) >> 6, tiles_n = (p.N + 63) >> 6;
  int tm, tn;
  gemm_tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * 64, n0 = tn * 64;
  // DMA staging: a 64-row tile is 512 16-byte slots; wave w owns slots [w*256, w*256+256): 4 instructions of 64 lanes
  const half_t* srcA[4];
  const half_t* srcW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pslot = (wave * 4 + i) * 64 + lane;
    const int r = pslot >> 3, c = pslot & 7;
    const int sw = (c ^ ((r >> 1) & 7)) << 3;
    const int ga = m0 + r < p.M ? m0 + r : p.M - 1, gw = n0 + r < p.N ? n0 + r : p.N - 1;
    srcA[i] = p.A + (size_t)ga * p.lda + sw;
    srcW[i] = p.W + (size_t)gw * p.ldw + sw;
  }
  auto stage = [&](int buf, int kt) {
    half_t* sA = smem + buf * 2 * TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + kt * 64),
                                       (__attribute__((address_space(3))) void*)(sA + (wave * 4 + i) * 512), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[i] + kt * 64),
                                       (__attribute__((address_space(3))) void*)(sA + TILE + (wave * 4 + i) * 512), 16, 0, 0);
  };
  f32x16 acc[2][1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
  const int nk = p.K >> 6;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) stage(s, s);
  int cur = 0, nxt = NST - 1;                           // stage consumed / stage refilled this step
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed when at most min(NST - 2, nk - 1 - kt) younger stages of this wave are still in flight
    const int younger = nk - 1 - kt < NST - 2 ? nk - 1 - kt : NST - 2;
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // ... everyone's has; the stage consumed in step kt - 1 is free
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NST - 1 < nk) stage(nxt, kt + NST - 1);
    const half_t* sA = smem + cur * 2 * TILE;
    const half_t* sW = sA + TILE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cc = ks * 2 + hh;
      const half8 af = gemm_frag(sA, wave * 32 + l31, cc);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[ni][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gemm_frag(sW, ni * 32 + l31, cc), af, acc[ni][0], 0, 0, 0);
    }
    cur = cur + 1 == NST ? 0 : cur + 1;
    nxt = nxt + 1 == NST ? 0 : nxt + 1;
  }
  __syncthreads();     // every wave is done reading the last stage: LDS becomes the epilogue staging area
  float rsc[1];
  gemm_row_factors<1>(p, m0 + wave * 32, l31, rsc);
  gemm_epilogue_staged<EPI, 2, 1>(p, acc, m0 + wave * 32, n0, lane, gemm_smem + wave * (32 * (64 * 4 + 16)), rsc);
}

// ---- skinny GEMM: few rows (the single-step decoder: M = sequences in the batch; 32 rows per blockIdx.z) --------
// Weight-streaming regime: every weight element is used once, so W is never staged in LDS.  One 512-thread workgroup
// owns NT consecutive 32-row weight tiles (NT = 2 for GEGLU: gate + up); its 8 waves take the K dimension in
// interleaved 16-wide steps, each wave streams its weight rows straight from HBM into MFMA A fragments (16 B per
// lane, 4 steps in flight) and reads the tiny activation matrix (L2-resident) as the B fragment.  Partial
// accumulators are combined through LDS by a FIXED binary tree (bitwise reproducible, unlike an atomic split-K)
// that needs only 16*NT KiB, so these workgroups still fit on a CU next to two resident encoder GEMM workgroups.
// (Tried and dropped: fusing the decoder's RMSNorms into this kernel's activation loads - every workgroup recomputing the
// row factors of its 32 rows and converting fp32 rows on the fly - removed three launches per layer but made the GEMMs
// slower by more than the norm kernels cost: pointwise -1.7 %, a setwise compare 6.83 -> 6.98 ms.  Likewise 8 instead of
// 4 K steps in flight per wave: the decoder GEMMs of a compare went 1.81 -> 2.13 ms.)
#define SKINNY_THREADS 512
template <int EPI, int NT>
__global__ __launch_bounds__(SKINNY_THREADS) void gemm_skinny_kernel(GemmArgs p) {
  __shared__ float red[4 * NT * 1024];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, hh = lane >> 5;
  const int n0 = blockIdx.x * 32 * NT;
  const int nsteps = p.K >> 4;
  const int mrow0 = blockIdx.z * 32;          // more than 32 rows: one workgroup per 32-row slab (weights re-read via L2)
  const int m = mrow0 + l31 < p.M ? mrow0 + l31 : p.M - 1;
  p.A += (size_t)blockIdx.y * p.bsA;          // batched form: one small GEMM per blockIdx.y (e.g. per attention head)
  p.W += (size_t)blockIdx.y * p.bsW;
  p.C = (char*)p.C + (size_t)blockIdx.y * p.bsC * ((EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32 || EPI == EPI_ARGMAX_F32) ? 4 : 2);
  const half_t* arow = p.A + (size_t)m * p.lda + 8 * hh;
  const half_t* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int n = n0 + t * 32 + l31;
    n = n < p.N ? n : p.N - 1;
    wrow[t] = p.W + (size_t)n * p.ldw + 8 * hh;
  }
  f32x16 acc[NT][1];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][0][r] = 0.f;
  // Wave 0 writes the result: what its epilogue reads besides the accumulators is requested now and travels under the
  // weight stream - the old fp32 rows of a residual add (registers), the block sums of squares of a folded RMSNorm (DMA
  // straight into LDS: no registers held across the main loop, the occupancy of this kernel is what hides its latency).
  __shared__ f32x4 sqs[8 * 64];                 // [instruction][lane]: float4 2i + (lane >> 5) of row (lane & 31)
  f32x4 old[EPI == EPI_RESID_F32 ? 4 : 1];
  float rowfac = 1.f;
  const bool row_ok = mrow0 + l31 < p.M;
  const bool sq_here = p.ssq_in && !p.rowscale;
  const bool sq_dma = sq_here && (p.nb_in & 3) == 0 && p.nb_in <= 64;
  if (wave == 0) {
    if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + 8 * q + 4 * hh;
        old[q] = (row_ok && n < p.N) ? *(const f32x4*)((const float*)p.C + (size_t)m * p.ldc + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (p.rowscale) rowfac = p.rowscale[m];
    else if (sq_dma) {
      const int nq = p.nb_in >> 2;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (2 * i < nq) {
          const int j = 2 * i + hh < nq ? 2 * i + hh : nq - 1;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.ssq_in + (size_t)m * p.nb_in + 4 * j),
                                           (__attribute__((address_space(3))) void*)(sqs + i * 64), 16, 0, 0);
        }
      }
    }
  }
  int s = wave;
  for (; s + 24 < nsteps; s += 32) {
    half8 bf[4], wf[4][NT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = (s + 8 * u) << 4;
      bf[u] = *(const half8*)(arow + k);
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[u][t] = *(const half8*)(wrow[t] + k);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[u][t], bf[u], acc[t][0], 0, 0, 0);
  }
  for (; s < nsteps; s += 8) {
    const int k = s << 4;
    const half8 bf = *(const half8*)(arow + k);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const half8 wf = *(const half8*)(wrow[t] + k);
      acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, bf, acc[t][0], 0, 0, 0);
    }
  }
  // fixed reduction tree: (w, w+4) -> w ; (w, w+2) -> w ; (0, 1) -> 0
#pragma unroll
  for (int half_n = 4; half_n >= 1; half_n >>= 1) {
    if (wave >= half_n && wave < 2 * half_n) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - half_n) * NT + t) * 1024 + r * 64 + lane] = acc[t][0][r];
    }
    __syncthreads();
    if (wave < half_n) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][0][r] += red[(wave * NT + t) * 1024 + r * 64 + lane];
    }
    __syncthreads();
  }
  if (wave != 0) return;
  if (sq_here) {
    // blocks added in increasing order, whatever the path (the DMA landed long ago: every barrier above drained vmcnt)
    float ssum = 0.f;
    if (sq_dma) {
      const int nq = p.nb_in >> 2;
      for (int j = 0; j < nq; ++j) {
        const f32x4 v = sqs[(j >> 1) * 64 + (j & 1) * 32 + l31];
        ssum += v[0]; ssum += v[1]; ssum += v[2]; ssum += v[3];
      }
    } else {
      for (int j = 0; j < p.nb_in; ++j) ssum += p.ssq_in[(size_t)m * p.nb_in + j];
    }
    rowfac = rsqrtf(ssum / (float)p.K + p.eps_in) / p.xs;
  }
  if constexpr (EPI == EPI_RESID_F32) {
    // C += acc (old rows prefetched above); with p.xraw also the fp16 copy of the new row and the sum of squares of this
    // workgroup's 32 columns (producer side of the next folded norm): a lane holds 16 of them, its partner (lane ^ 32) the rest
    const float sc = p.scale * rowfac;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 8 * q + 4 * hh;
      const bool ok = row_ok && n < p.N;
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = old[q][j] + acc[0][0][4 * q + j] * sc;
      if (ok) {
        *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = o;
        if (p.xraw) {
#pragma unroll
          for (int j = 0; j < 4; ++j) ss = __builtin_fmaf(o[j], o[j], ss);
          const half4 xr = {f2h_sat(o[0] * p.xs), f2h_sat(o[1] * p.xs), f2h_sat(o[2] * p.xs), f2h_sat(o[3] * p.xs)};
          *(half4*)(p.xraw + (size_t)m * p.ldx + n) = xr;
        }
      }
    }
    if (p.xraw) {
      ss += __shfl_xor(ss, 32);
      if (hh == 0 && row_ok) p.ssq[(size_t)m * p.nb + (n0 >> 5)] = ss;
    }
  } else if constexpr (EPI == EPI_ARGMAX_F32) {
    // the logits of this block never reach memory: largest value and its FIRST column (torch.argmax tie rule) among the
    // workgroup's 32 columns; argmax_blocks_kernel finishes the row
    const float sc = p.scale * rowfac;
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + 8 * q + 4 * hh + j;
        const float v = acc[0][0][4 * q + j] * sc;
        if (n < p.N && (v > best || (v == best && n < bi))) { best = v; bi = n; }
      }
    const float ov = __shfl_xor(best, 32);
    const int oi = __shfl_xor(bi, 32);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    if (hh == 0 && row_ok) {
      ((float*)p.C)[(size_t)m * p.ldc + blockIdx.x] = best;
      p.amax_idx[(size_t)m * p.ldc + blockIdx.x] = bi;
    }
  } else {
    gemm_epilogue<EPI, NT, 1>(p, acc, mrow0, n0, l31, hh, rowfac);
  }
}

// ================================= GEMM v2: 256-row tiles ==================================================
// Why: a 128x128 tile needs 8138/128 = 64 B/clk/CU of L2->LDS fill to run the MFMA pipe at peak, more than the ~56
// B/clk/CU a CU can pull (measured: MFMA busy 25 %, PMC in profiles/); a 256-wide tile halves that.  Structure:
//   * 512 threads = 8 waves as WM x WN, wave tile (MI*32) x (NI*32), fp32 accumulators in registers;
//   * K step of 64 (four MFMA k16 steps): tile rows are 128 B = one full cache line per DMA row (a 64-B row makes
//     every line travel L2->L1 twice), LDS image and XOR swizzle identical to the 128x128 kernel (conflict-free);
//   * two LDS stages; the DMA loads of k-tile t+1 are issued DURING the compute of tile t, one load after every
//     second MFMA of the first two k16 steps, so their issue cost hides under matrix-pipe time and they have the
//     rest of the step to land; one raw s_barrier per K step;
//   * XCD-aware tile order: block b runs on XCD b%8, and each XCD is handed a contiguous run of the tm-major tile
//     list, so the workgroups co-resident on one XCD share A / W panels through its private 4 MiB L2.
template <int EPI, int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(WM * WN * 64, WM * WN / 4) void gemm_v2_kernel(GemmArgs p) {
  constexpr int NW = WM * WN;                                    // waves per workgroup: 8 (2 per SIMD) or 16 (4 per SIMD)
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  constexpr int A_HALFS = BM * 64, W_HALFS = BN * 64, STAGE_HALFS = A_HALFS + W_HALFS;
  constexpr int A_INSTR = BM / 8, W_INSTR = BN / 8;            // one DMA instruction = 64 x 16 B = 8 tile rows
  constexpr int A_PW = A_INSTR / NW, W_PW = (W_INSTR + NW - 1) / NW;  // per wave per stage
  constexpr int LOADS = A_PW + W_PW;
  static_assert((NW == 8 || NW == 16) && A_INSTR % NW == 0, "8 or 16 waves; A tile must split evenly");
  constexpr int ISSUE_STRIDE = (3 * MI * NI) / LOADS;           // DMA issues spread over the first three k16 steps
  static_assert(ISSUE_STRIDE >= 1, "not enough MFMA slots to hide the DMA issues");
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  half_t* smem = (half_t*)gemm_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  int tm, tn;
  gemm_tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane DMA source pointers (k offset added per stage) and LDS destinations ----
  const half_t* src[LOADS];
  int dst[LOADS];
#pragma unroll
  for (int j = 0; j < LOADS; ++j) {
    const bool isA = j < A_PW;
    int instr = isA ? (wave * A_PW + j) : (wave * W_PW + (j - A_PW));
    if (!isA) instr = instr % W_INSTR;                           // uneven split: duplicate a load (same bytes)
    const int row = instr * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int grow = (isA ? m0 : n0) + row;
    const int lim = isA ? p.M : p.N;
    grow = grow < lim ? grow : lim - 1;
    src[j] = (isA ? p.A + (size_t)grow * p.lda : p.W + (size_t)grow * p.ldw) + chunk * 8;
    dst[j] = (isA ? 0 : A_HALFS) + instr * 512;
  }
  auto issue = [&](int j, int kt) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * 64),
                                     (__attribute__((address_space(3))) void*)(smem + (kt & 1) * STAGE_HALFS + dst[j]),
                                     16, 0, 0);
  };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K >> 6;
#pragma unroll
  for (int j = 0; j < LOADS; ++j) issue(j, 0);
  constexpr bool PREFETCH_RES = EPI == EPI_RESID_F32 && NI * MI <= 6 && NW == 8;   // register budget: 16 fp32 per fragment
  f32x16 res[PREFETCH_RES ? NI : 1][PREFETCH_RES ? MI : 1];
  if (PREFETCH_RES) gemm_prefetch_residual<PREFETCH_RES ? NI : 1, PREFETCH_RES ? MI : 1>(p, res, m0 + wm * MI * 32, n0 + wn * NI * 32, l31, hh);
  int arow[MI], wrow[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) arow[i] = (wm * MI * 32 + i * 32 + l31);
#pragma unroll
  for (int i = 0; i < NI; ++i) wrow[i] = (wn * NI * 32 + i * 32 + l31);

  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA for tile kt has landed
    __builtin_amdgcn_s_barrier();                       // ... everyone's has; stage (kt+1)&1 is free again
    __builtin_amdgcn_sched_barrier(0);
    const half_t* sA = smem + (kt & 1) * STAGE_HALFS;
    const half_t* sW = sA + A_HALFS;
    const bool more = kt + 1 < nk;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8 af[MI], wf[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = gemm_frag(sW, wrow[i], ks * 2 + hh);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = gemm_frag(sA, arow[i], ks * 2 + hh);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
          const int slot = ks * (MI * NI) + ni * MI + mi;     // compile-time after unrolling
          if (slot % ISSUE_STRIDE == 0 && slot / ISSUE_STRIDE < LOADS && more) issue(slot / ISSUE_STRIDE, kt + 1);
        }
    }
  }
  if (PREFETCH_RES) {
#pragma unroll
    for (int i = 0; i < (PREFETCH_RES ? NI : 1); ++i)
#pragma unroll
      for (int j = 0; j < (PREFETCH_RES ? MI : 1); ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * p.scale + res[i][j][r];
  }
  __syncthreads();   // every wave is done reading the last stage: LDS becomes the epilogue staging area
  if (PREFETCH_RES) {
    GemmArgs q = p; q.scale = 1.f;
    float rsc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) rsc[i] = 1.f;
    gemm_epilogue_staged<EPI, NI, MI, true>(q, acc, m0 + wm * MI * 32, n0 + wn * NI * 32, lane,
                                            gemm_smem + wave * (32 * (NI * 32 * 4 + 16)), rsc);
  } else {
    float rsc[MI];
    gemm_row_factors<MI>(p, m0 + wm * MI * 32, l31, rsc);
    gemm_epilogue_staged<EPI, NI, MI>(p, acc, m0 + wm * MI * 32, n0 + wn * NI * 32, lane,
                                      gemm_smem + wave * (32 * (NI * 32 * 4 + 16)), rsc);
  }
}

// ================================= GEMM v3: 256x256 ping-pong ================================================
// The v2 loop above leaves the matrix pipe idle while a wave waits for its fragment reads (every k16 step begins with
// ds_reads + lgkmcnt(0)) and drains the DMA queue once per K tile.  This kernel keeps the 256x256x64 tile, 8 waves and
// the LDS image (128-B rows, XOR swizzle, DMA with the swizzle on the source address) but changes the schedule:
//   * the two waves that share a SIMD (w and w+4) run HALF A PHASE APART: while one executes the MFMAs of a phase the
//     other issues its ds_reads for the next one, and they swap at every s_barrier (group 1 runs one barrier ahead);
//   * the tile rows are dealt to four 16 KiB HALF-TILES per K tile so that a wave's work splits into two super-phases
//     whose operands arrive separately, while its outputs still form ONE contiguous 128 x 64 block (A half h holds tile
//     rows 128*wm + 64h + [0,64), W half h rows 64*wn + 32h + [0,32)) - the epilogues (incl. the gate/up pairing of
//     GEGLU) are the v2 ones;
//   * DMA loads are issued BETWEEN the MFMAs of the issuing wave and waited for with COUNTED vmcnt (never 0 in the
//     loop), one super-phase before the data is read, followed by a barrier (the ordering LDS-DMA needs).
// What was measured on the way (tools/gemm_bench.py knock-out variants, 4096^3, profiles/r01e_gemm_pingpong.txt): an
// 8-phase form (8 MFMA per phase, DMA issued in the read section) ran 122 us = no better than v2; MFMA + barriers alone
// take 70 us with zero operands but 94 us with random operand bits (the chip clocks to its power budget), DMA alone
// 53-56 us, fragment reads alone 41 us; moving the DMA issue among the MFMAs gave 119 us, 16-MFMA phases 113-117 us
// (1.18-1.21 PF; DMA now costs ~12 us and the reads ~7 us over the 94 us MFMA floor).  Issuing the DMA BEFORE the reads
// of a phase was 10 % slower.
template <int N>
__device__ __forceinline__ void gemm_wait_vmcnt() {
  static_assert(N >= 0 && N <= 10 && N % 2 == 0, "counted vmcnt values used by the ping-pong kernel");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
}

// Per K tile t (LDS stage t&1):   SP0: read A0, W0, W1 (16 ds_read_b128) | 16 MFMA: (A0,W0) (A0,W1)
//                                 SP1: read A1 (8)                        | 16 MFMA: (A1,W0) (A1,W1)
// Half-tile order n = 4t + {A0, W0, W1, A1}; the MFMA section of super-phase g issues half-tiles 2g+6 and 2g+7 (two DMA
// instructions each per wave, after MFMA 1, 5, 9 and 13), so six half-tiles are in flight behind the one being computed;
// counted waits before the first barrier of g: vmcnt(4) for even g (A1(t) landed), vmcnt(2) for odd g (A0, W0, W1 of
// t+1 landed).  Buffer n mod 8 is re-filled in super-phase floor(n/2)+1 or later, one full barrier after its last
// reader (group 1) has its fragments in registers.  Every accumulator sees the K tiles, and the four k16 steps inside
// one, in increasing order with the same operand slots as the v1/v2 kernels: results are bit-identical to theirs.
// Needs K >= 128 (two K tiles).  The kernel is PERSISTENT (see set_tile below): launched with one workgroup per CU it
// walks its tiles and overlaps each tile's epilogue with the next tile's first loads (qkv at M = 23552: 170 -> 149 us,
// FFN-in 326 -> 308 us).  KO: timing-only knock-outs for bottleneck hunting (results are garbage): 1 = no DMA in
// the loop, 2 = no fragment reads, 4 = no MFMA, 8 = no W-panel DMA, 16 = no A-panel DMA (round 4: what a register-staged
// feed of one operand could buy at most).  Product code instantiates KO = 0 only.
#ifndef GEMM_A_AUX
#define GEMM_A_AUX 0   // cache-policy bits of the ping-pong kernel's A-panel / W-panel DMA loads (bit 1 = nt)
#endif
#ifndef GEMM_W_AUX
#define GEMM_W_AUX 0
#endif
#ifndef GEMM_KSPLIT_SKEW
#define GEMM_KSPLIT_SKEW 4      // K tiles split 0 of a two-way split takes more than split 1 (gemm_pp2_kernel<.., SPLIT>)
#endif
#ifndef GEMM_PP2_ISSUE_Q
#define GEMM_PP2_ISSUE_Q 1      // the DMA instruction of a k16 step goes out after its MFMA number ISSUE_Q (0..3)
#endif
#ifndef GEMM_PP2_SETPRIO
#define GEMM_PP2_SETPRIO 1
#endif
// LDS-DMA instructions of the ping-pong loop written out (round 4): saddr form - SGPR base + 32-bit lane offset, M0 set by hand.
// The builtin widens every lane offset to a 64-bit address: 16 VGPRs for the eight offsets of a wave (of 256) and a
// v_lshl_add_u64 per DMA instruction among the MFMAs.  Bit-identical; grouped pipeline +1.6 .. 2.0 % (7 220-7 290 -> 7 360-7 400
// passages/s in two alternations on one box, profiles/r04_gemm_feed_experiments.txt).
// Tried with it and dropped: a "mixed feed" - the W half-tiles through registers (global_load_dwordx4 in one super-phase,
// ds_write_b128 into the same LDS image in the next) instead of LDS-DMA, which the r03 probe and the r04 knock-outs (no W DMA
// at all: +6.5 .. 8.6 % on the loop) had suggested: 7-15 % SLOWER (QKV 405 -> 433 us, FFN-in 730 -> 840 us at M = 58 880).  A
// register load has to land within ONE super-phase (the 8 staging registers are all the kernel can spare; the DMA runs three
// super-phases ahead), so its wait sits at the head of every MFMA section.  Second form, with the 20 registers the asm DMA freed:
// two staging sets, two super-phases between a load and its ds_write (waits that are always satisfied): still 6 % slower in the
// pipeline (7 180 against 7 660-7 700 passages/s) - it is not the latency; eight more VMEM instructions and four ds_write_b128 per
// K tile among the MFMAs cost more than the DMA pieces they replace.  The patch (second form) is kept in
// profiles/r04_wreg_experiment.patch.
#ifndef GEMM_PP2_ASMDMA
#define GEMM_PP2_ASMDMA 1
#endif
// How many of a super-phase's four DMA instructions go out in its READ section (behind the fragment reads, in front of the
// counted wait) instead of among its MFMAs: 0, 2 (the first half-tile) or 4.  An MFMA section is 16 x 32 = 512 matrix-pipe cycles
// plus what its wave spends issuing DMA (~60 cycles a piece); the partner wave's read section next to it is ~300 cycles.
// Measured (r04, separately compiled libraries alternated on one box): 0 / 2 / 4 = 7 694-7 731 / 7 678-7 766 / 7 644-7 731 passages/s -
// no difference: it is not the issuing wave's stall that sets the 2 us of a K tile but the CU's LDS-DMA throughput beside a busy
// matrix pipe (64 pieces x ~50 cycles; probe r03), wherever the instructions sit.  Left at 0.
// Residual epilogue of the ping-pong kernel: the old fp32 rows are requested this many 16-row slabs ahead of their use (64 registers at
// 3; the operand fragments are dead by then).  Round 6 (profiles/r06_epilogue_knockouts.txt): O alone 216 -> 202 us at M = 58 880,
// FFN-out unchanged, the pipelined bench 7 698 -> 7 743 passages/s (six alternating runs each); all eight slabs ahead (7) spills and is
// slower (247 us).  The K-split instantiation (247 VGPRs) keeps 0.  Same bits.
#ifndef GEMM_PP2_EDEPTH
#define GEMM_PP2_EDEPTH 3
#endif
#ifndef GEMM_PP2_RISSUE
#define GEMM_PP2_RISSUE 0
#endif
// RS: consumer side of the folded RMSNorm - the accumulators of row m are multiplied by p.rowscale[m] (gemm_epilogue_staged)
// SPLIT (round 6; p.ksplit = 1 or 2): K split over TWO workgroups for the launches that have far fewer tiles than the chip has CUs and a long K (Llama-3-8B
// O / down projections of ONE setwise prompt: 96 tiles, K = 4 096 / 14 336; flan-t5-xl FFN-out at a few thousand rows).  The tile
// walk runs over 2 * tiles virtual tiles (split s of tile t = virtual tile s * tiles + t, K tiles [kt0, kt0 + nk); split 0 takes a
// few K tiles more).  The first arriver of a tile leaves its fp32 partial tile in a slab as WRITE-THROUGH 16-byte stores, every
// storing wave drains, one lane takes the tile's arrival ticket (guide G16 R1); the LAST arriver - it looks at the ticket first
// and, if the other is already there, publishes nothing - adds the other slab to its registers (a + b == b + a: the same bits
// whoever arrives last) and runs the epilogue.  Nobody spins.  (Three / four ways measured slower: every slab re-read.)  A split
// sums K in another association than one workgroup would: results differ from the unsplit launch in the last fp32 bits; whether a
// launch is split follows from (M, N, K) alone (host: choose_ksplit), so a call shape always gives the same bits.
template <int EPI, int KO = 0, bool RS = false, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void gemm_pp2_kernel(GemmArgs p) {
  constexpr int HALF = 128 * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  half_t* smem = (half_t*)gemm_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int grp = wave >> 2, wm = wave & 1, wn = (wave >> 1) & 3;
  const int tiles_m = (p.M + 255) >> 8, tiles_n = (p.N + 255) >> 8;
  const int nsplit = SPLIT ? p.ksplit : 1;
  const int ntiles_out = tiles_m * tiles_n, ntiles = ntiles_out * nsplit;   // output tiles | (virtual) tiles of the walk
  int m0 = 0, n0 = 0, kt0 = 0, nk = p.K >> 6;                                // current tile: origin, first K tile, K tiles
  unsigned off[4][2];   // kind 0 = A0, 1 = A1, 2 = W0, 3 = W1; byte offsets of this wave's two DMA instructions
  // PERSISTENT: a workgroup walks tiles blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8 or the whole grid, so
  // the tile -> XCD association of gemm_tile_coords holds) and issues the NEXT tile's first six half-tiles before it
  // runs the epilogue of the current one: pipeline fill and workgroup launch no longer sit between two tiles.
  auto set_tile = [&](int tile) {
    int tm, tn;
    if constexpr (SPLIT) {
      const int sp = tile / ntiles_out, nk_all = p.K >> 6, base_n = nk_all / nsplit, rem = nk_all - base_n * nsplit;
      tile -= sp * ntiles_out;
      kt0 = sp * base_n + (sp < rem ? sp : rem);
      nk = base_n + (sp < rem ? 1 : 0);
      // two splits: split 0 takes GEMM_KSPLIT_SKEW more K tiles than split 1, so that split 1 publishes its slab while split 0
      // still computes and split 0 - arriving last - finds the ticket taken and need not publish at all (below)
      if (base_n > 4 * GEMM_KSPLIT_SKEW) { if (sp == 0) nk += GEMM_KSPLIT_SKEW; else { kt0 += GEMM_KSPLIT_SKEW; nk -= GEMM_KSPLIT_SKEW; } }
    }
    gemm_tile_coords(tile, tiles_m, tiles_n, tm, tn, p.group_n);
    m0 = tm * 256; n0 = tn * 256;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wave * 2 + j) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int ga = m0 + 128 * (r >> 6) + 64 * h + (r & 63);
        ga = ga < p.M ? ga : p.M - 1;
        off[h][j] = ((unsigned)ga * (unsigned)p.lda + chunk * 8) * 2u;
        int gw = n0 + 64 * (r >> 5) + 32 * h + (r & 31);
        gw = gw < p.N ? gw : p.N - 1;
        off[2 + h][j] = ((unsigned)gw * (unsigned)p.ldw + chunk * 8) * 2u;
      }
    }
  };
  // buffer of (kind, stage) at (kind * 2 + stage) * 16 KiB
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)gemm_smem);
  auto issue1 = [&](auto kindc, int stage, int tile, auto jc) {
    constexpr int kind = decltype(kindc)::value, j = decltype(jc)::value;
    const char* base = (const char*)((kind < 2 ? p.A : p.W) + (kt0 + tile) * 64);
#if GEMM_PP2_ASMDMA
    const unsigned dst = lds0 + (unsigned)(((kind * 2 + stage) * HALF + (wave * 2 + j) * 512) * 2);
    const unsigned o = off[kind][j];                      // (asm operands cannot name a captured array element)
    // (M0 is written here: it is on the clobber list so that the compiler never keeps a value of its own in it across the statement)
    // (the clobber draws the note "reserved register M0 on the clobber list" per instantiation: silenced for THIS statement only - the
    // rest of the build keeps its inline-asm diagnostics; tests/test_isa_guards.py is the check that no compiler-owned M0 value lives
    // across the statement)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(o), "s"(base) : "memory", "m0");
#pragma clang diagnostic pop
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[kind][j]),
                                     (__attribute__((address_space(3))) void*)(smem + (kind * 2 + stage) * HALF + (wave * 2 + j) * 512),
                                     16, 0, kind < 2 ? GEMM_A_AUX : GEMM_W_AUX);
#endif
  };
  using std::integral_constant;
  using I0 = integral_constant<int, 0>; using I1 = integral_constant<int, 1>;
  using I2 = integral_constant<int, 2>; using I3 = integral_constant<int, 3>;

  const int xs = (l31 >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = ((ks * 2 + hh) ^ xs) << 3;
  const int a_lane = (wm * 64 + l31) * 64, w_lane = (wn * 32 + l31) * 64;
  half8 aF[2][4], w0F[4], w1F[4];
  if constexpr (KO != 0) {   // knock-out runs keep non-trivial operand bits (MFMA power depends on the data)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      union { unsigned u[4]; half8 h; } x;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const unsigned hsh = (unsigned)(lane * 2654435761u + ks * 40503u + i * 9973u + wave * 77u); x.u[i] = (hsh & 0x8FFF8FFFu) | 0x30003000u; }
      aF[0][ks] = x.h; w0F[ks] = x.h;
      x.u[0] ^= 0x80000000u; x.u[2] ^= 0x00008000u;
      aF[1][ks] = x.h; w1F[ks] = x.h;
    }
  }
  f32x16 acc[2][4];

  // one super-phase.  SP: 0 / 1;  ISSUE: the two half-tiles this phase's MFMA section prefetches exist;
  // WAIT: vmcnt count before the first barrier (-1 = none)
  auto sp = [&](auto spc, auto issuec, auto waitc, int t) {
    constexpr int SP = decltype(spc)::value, WAIT = decltype(waitc)::value;
    constexpr bool ISSUE = decltype(issuec)::value && !(KO & 1);
    const int st = t & 1;
    if constexpr (!(KO & 2)) {
      if constexpr (SP == 0) {
        const half_t* sw0 = smem + (4 + st) * HALF + w_lane;
        const half_t* sw1 = smem + (6 + st) * HALF + w_lane;
        const half_t* sa = smem + (0 + st) * HALF + a_lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w0F[ks] = *(const half8*)(sw0 + koff[ks]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { aF[0][ks] = *(const half8*)(sa + koff[ks]); aF[1][ks] = *(const half8*)(sa + 2048 + koff[ks]); }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w1F[ks] = *(const half8*)(sw1 + koff[ks]);
      } else {
        const half_t* sa = smem + (2 + st) * HALF + a_lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { aF[0][ks] = *(const half8*)(sa + koff[ks]); aF[1][ks] = *(const half8*)(sa + 2048 + koff[ks]); }
      }
    }
    constexpr int RISSUE = KO == 0 ? GEMM_PP2_RISSUE : 0;
    if constexpr (ISSUE && RISSUE > 0) {
      // this super-phase's first (RISSUE = 2) or both (4) half-tiles are requested here: their buffers were last read one
      // super-phase ago - those ds_reads were ISSUED before the barrier in front of this section, and a read issued earlier
      // returns the old bytes whatever lands later (MI355X_MICROARCH.md: LDS-DMA ordering)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SP == 0) {
        issue1(I3{}, st ^ 1, t + 1, I0{}); issue1(I3{}, st ^ 1, t + 1, I1{});
        if constexpr (RISSUE == 4) { issue1(I1{}, st ^ 1, t + 1, I0{}); issue1(I1{}, st ^ 1, t + 1, I1{}); }
      } else {
        issue1(I0{}, st, t + 2, I0{}); issue1(I0{}, st, t + 2, I1{});
        if constexpr (RISSUE == 4) { issue1(I2{}, st, t + 2, I0{}); issue1(I2{}, st, t + 2, I1{}); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // (knock-outs 8 / 16 drop the W / A half of the DMA: the counted waits follow the number of instructions still issued)
    constexpr int NA = (KO & 16) ? 0 : 2, NW = (KO & 8) ? 0 : 2;
    constexpr int WAITK = (WAIT == 4 ? NA + NW : (WAIT == 2 ? NA : WAIT)) + ((ISSUE && WAIT > 0) ? RISSUE : 0);
    if constexpr (WAIT >= 0 && !(KO & 1)) gemm_wait_vmcnt<WAITK >= 0 ? WAITK : 0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (GEMM_PP2_SETPRIO) __builtin_amdgcn_s_setprio(1);
    if constexpr ((KO & 4) != 0) {   // keep the fragment reads alive when the MFMAs are knocked out
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" :: "v"(aF[0][ks]), "v"(aF[1][ks]), "v"(w0F[ks]), "v"(w1F[ks]));
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ni = q >> 1, mi = q & 1;
        if constexpr (!(KO & 4))
          acc[ni][2 * SP + mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ni == 0 ? w0F[ks] : w1F[ks], aF[mi][ks], acc[ni][2 * SP + mi], 0, 0, 0);
        if constexpr (ISSUE) {
          if (q == GEMM_PP2_ISSUE_Q) {
            __builtin_amdgcn_sched_barrier(0);
            // SP0 of tile t prefetches W1(t+1), A1(t+1); SP1 prefetches A0(t+2), W0(t+2)  (stage of tile t+1 = st^1, t+2 = st)
            if constexpr (SP == 0) {
              if constexpr (!(KO & 8) && RISSUE < 2) { if (ks == 0) issue1(I3{}, st ^ 1, t + 1, I0{}); if (ks == 1) issue1(I3{}, st ^ 1, t + 1, I1{}); }
              if constexpr (!(KO & 16) && RISSUE < 4) { if (ks == 2) issue1(I1{}, st ^ 1, t + 1, I0{}); if (ks == 3) issue1(I1{}, st ^ 1, t + 1, I1{}); }
            } else {
              if constexpr (!(KO & 16) && RISSUE < 2) { if (ks == 0) issue1(I0{}, st, t + 2, I0{}); if (ks == 1) issue1(I0{}, st, t + 2, I1{}); }
              if constexpr (!(KO & 8) && RISSUE < 4) { if (ks == 2) issue1(I2{}, st, t + 2, I0{}); if (ks == 3) issue1(I2{}, st, t + 2, I1{}); }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    if (GEMM_PP2_SETPRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue of a tile: half-tiles 0..5 = K tile 0 complete, A0 and W0 of K tile 1 ----
  auto issue_prologue = [&]() {
    if constexpr (!(KO & 16)) { issue1(I0{}, 0, 0, I0{}); issue1(I0{}, 0, 0, I1{}); }
    if constexpr (!(KO & 8)) { issue1(I2{}, 0, 0, I0{}); issue1(I2{}, 0, 0, I1{}); }
    if constexpr (!(KO & 8)) { issue1(I3{}, 0, 0, I0{}); issue1(I3{}, 0, 0, I1{}); }
    if constexpr (!(KO & 16)) { issue1(I1{}, 0, 0, I0{}); issue1(I1{}, 0, 0, I1{}); }
    if constexpr (!(KO & 16)) { issue1(I0{}, 1, 1, I0{}); issue1(I0{}, 1, 1, I1{}); }
    if constexpr (!(KO & 8)) { issue1(I2{}, 1, 1, I0{}); issue1(I2{}, 1, 1, I1{}); }
  };
  using Yes = integral_constant<bool, true>; using No = integral_constant<bool, false>;
  using W4 = integral_constant<int, 4>; using W2 = integral_constant<int, 2>; using W0c = integral_constant<int, 0>;
  using WN = integral_constant<int, -1>;
  int tile = blockIdx.x;
  set_tile(tile);
  issue_prologue();
  while (true) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // at most 6 operations outstanding: the six oldest of the 12 prologue loads (A0, W0, W1 of K tile 0, this wave's
  // share) have landed - epilogue stores of the previous tile are younger and only make the wait stricter
  gemm_wait_vmcnt<((KO & 16) ? 0 : 4) + ((KO & 8) ? 0 : 2)>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier ahead: its MFMAs meet group 0's reads
  __builtin_amdgcn_sched_barrier(0);

  int t = 0;
  for (; t < nk - 2; ++t) {
    sp(I0{}, Yes{}, W4{}, t);
    sp(I1{}, Yes{}, W2{}, t);
  }
  // tile nk-2: its SP1 has nothing left to prefetch (would be tile nk)
  sp(I0{}, Yes{}, W4{}, t);
  sp(I1{}, No{}, W2{}, t);
  ++t;
  // tile nk-1: nothing to prefetch; A1 of this tile is the only load that can still be in flight
  sp(I0{}, No{}, W0c{}, t);
  // nothing is in flight and the loop has no counted wait left: the row factors of the folded RMSNorm travel under the
  // last 16 MFMAs.  Issued through inline asm (one 32-bit lane offset against the uniform base, immediate offsets for
  // the four 32-row slabs - the buffer is padded, rows beyond M read defined junk) so that neither 64-bit address
  // registers nor compiler-placed waits appear here: with 254 VGPRs live the tracked form became two load / vmcnt(0)
  // pairs in front of the last super-phase.  Waited for explicitly at the top of the epilogue (same loop iteration).
  float rsc[4] = {p.scale, p.scale, p.scale, p.scale};
  if constexpr (RS) {
    const unsigned roff = (unsigned)(m0 + wm * 128 + l31) * 4u;
    // (s_nop 4: should the base pair ever come out of a spill lane right here - v_readlane is a VALU write of an SGPR, a VMEM read of
    // it needs five wait states, and the hazard recognizer does not look inside inline asm; round 5's chained-launch kernel met exactly that)
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=&v"(rsc[0]) : "v"(roff), "s"(p.rowscale));
    asm volatile("global_load_dword %0, %1, %2 offset:128" : "=&v"(rsc[1]) : "v"(roff), "s"(p.rowscale));
    asm volatile("global_load_dword %0, %1, %2 offset:256" : "=&v"(rsc[2]) : "v"(roff), "s"(p.rowscale));
    asm volatile("global_load_dword %0, %1, %2 offset:384" : "=&v"(rsc[3]) : "v"(roff), "s"(p.rowscale));
  }
  __builtin_amdgcn_sched_barrier(0);
  sp(I1{}, No{}, WN{}, t);
  if (grp == 0) __builtin_amdgcn_s_barrier();  // re-align the two groups
  __syncthreads();   // every wave is done reading the stages
  // the next tile's first loads go out now and land under this tile's epilogue.  Their targets are the buffers of
  // (A0, W0, W1, A1 | stage 0) and (A0, W0 | stage 1); the epilogue stages its rows in the 48 KiB that are not:
  // W1 | stage 1 (112..128 KiB) and the 32 KiB above the eight buffers.
  const int mbase = m0 + wm * 128, nbase = n0 + wn * 64;
  const int next = tile + gridDim.x;
  if constexpr (RS) {
    // The four row-factor loads are awaited BEFORE the next tile's DMA goes out, with nothing else in flight (the last
    // super-phase drained the DMA queue): counting them against the 12 prologue instructions behind them (vmcnt(12)) assumed that
    // plain loads and LDS-DMA loads retire in one order - they do not always: with the DMA written out in asm the compiler no
    // longer put its own vmcnt(0) in front of the __syncthreads() above, and under load (decoder graphs replaying beside the
    // encoder) some rows were scaled by stale registers - scores off by up to 0.5, caught by bench.py --mode shard's
    // recomputation check, not by the serial GPU tests (round 4; tests/test_gpu_kernels.py now stresses it).
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rsc[0]), "+v"(rsc[1]), "+v"(rsc[2]), "+v"(rsc[3]) :: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) rsc[i] *= p.scale;
  }
  const int cur_tile = tile;
  if (next < ntiles) { set_tile(next); issue_prologue(); }
  bool finish = true;
  if constexpr (SPLIT) {
    if (nsplit > 1) {
      // publish this workgroup's partial tile (layout [wave][fragment][quarter][lane] x 16 B: every store instruction is 1 KiB
      // of consecutive bytes), take the ticket; the last arriver combines
      const int spl = cur_tile / ntiles_out, t_out = cur_tile - spl * ntiles_out;
      float* mine = p.ks_slabs + (size_t)cur_tile * 65536;
      int* flag = (int*)(gemm_smem + 163840 - 16);
      // the other split already there?  Then this workgroup is the last arriver whatever happens and its partial stays in registers
      if (tid == 0) *flag = __hip_atomic_load(p.ks_cnt + t_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1;
      __syncthreads();
      const bool skip_publish = *flag != 0;
      __syncthreads();                                      // (the flag word is written again below)
      if (!skip_publish) {
        // saddr form (uniform slab base + ONE 32-bit lane offset per fragment + immediate offsets): 64-bit store addresses per
        // (fragment, quarter) would be hoisted out of the tile walk and spilled - the kernel has no registers to spare
        unsigned voff = (unsigned)(wave * 32768 + lane * 16);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          const f32x16& a = acc[f >> 2][f & 3];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 x = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
            asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3 sc1\n\ts_nop 1" :: "v"(voff), "v"(x), "s"(mine), "n"(q * 1024) : "memory");
          }
          voff += 4096;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains (and the next tile's first loads land)
      }
      __syncthreads();
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(p.ks_cnt + t_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == 1;
        if (last) {
          __hip_atomic_store(p.ks_cnt + t_out, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nobody touches this ticket again
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *flag = last;
      }
      __syncthreads();
      finish = *flag != 0;
      if (finish) {
        // own registers + the other split's slab (a + b == b + a: the same bits whichever of the two arrives last); one fragment
        // at a time - 16 registers of loads in flight, not 128
        const float* src = p.ks_slabs + (size_t)((1 - spl) * ntiles_out + t_out) * 65536 + (size_t)(wave * 32 * 64 + lane) * 4;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          f32x4 x[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] = *(const f32x4*)(src + (size_t)((f * 4 + q) * 64) * 4);
          f32x16& a = acc[f >> 2][f & 3];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[4 * q + j] += x[q][j];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  // fp32 outputs: 16 rows per pass (16 x 272 B per wave); fp16 outputs fit whole 32-row slabs (32 x 144 B)
  constexpr bool F32OUT = EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32;
  constexpr int EROWS = F32OUT ? 16 : 32;
  if (finish) gemm_epilogue_staged<EPI, 2, 4, false, EROWS, SPLIT ? 0 : GEMM_PP2_EDEPTH>(p, acc, mbase, nbase, lane, gemm_smem + 114688 + wave * 4608, rsc);
  if (next >= ntiles) break;
  tile = next;
  __syncthreads();   // staging rows are read before the next tile's DMA wraps around to W1 | stage 1
  }
}
