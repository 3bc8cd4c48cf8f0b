// Fused projections around the query-side cross-attention of the few-row decoder (round 4).
//
// hf: T5LayerCrossAttention (modeling_t5.py:404-432) as restated in attention.h (XAttnArgs): per decoder layer
//     q = W_q norm(x);  qk_h = W_k,h^T q_h;  [scores / softmax / sum_t p e_t over 64-key chunks];  combine;  ctx_h = W_v,h (.)
// was FIVE launches: two per-head weight-streaming GEMMs of 64-wide K or N (5120 workgroups each at 320 rows), the chunk
// kernel, the combine kernel (H x rows workgroups) and the q projection before them.  The two pairs around the chunk kernel
// are row- AND head-local, so each pair is one kernel here, one workgroup per (head, slab of decoder rows), on the matrix cores:
//   dec_cross_qk_kernel:  q_h = rowfactor * (W_q,h x)  (K = d, split over the eight waves, fixed-order LDS tree) -> fp16 in LDS
//                         -> qk_h = W_k,h^T q_h (K = 64) -> qk [rows, H, d] fp16          (replaces the cq + ckT GEMMs)
//   dec_cross_cv_kernel:  merge the chunks' partial sums of a (row, head) in chunk order and normalise (the arithmetic of
//                         xattn_combine_kernel) -> fp16 in LDS -> ctx_h = W_v,h (.) (K = d) -> ctx [rows, H*64] fp16
//                                                                                         (replaces combine + the wv GEMM)
// A decoder layer at one position is 7 launches instead of 9 (9 instead of 11 beyond), the 5120-workgroup launches are gone.
// A row's result does not depend on which rows share its slab (MFMA columns are independent), so neither on the batch nor
// on the rows-per-workgroup the host picks.  Rounding points are the unfused path's (q, qk, the merged sums and ctx are fp16
// there too); only the fp32 summation order over K differs (contiguous eighths here, interleaved 16-steps there).
#pragma once
#include "common.h"

struct DecQKArgs {
  const half_t* x; int ldx;      // [M, ldx] decoder stream: fp16(x * xs) with the norm folded (row factor below) or the normalised rows
  const half_t* wq;              // [H*64, d] cross-attention q projection (norm weight folded in when x is the raw stream)
  const half_t* wkT;             // [H][d][64]  W_k regrouped per head and transposed (finalize)
  half_t* qk;                    // [M, H, d]
  int M, d, H;
  const float* rowscale;         // row factors, or
  const float* ssq_in; int nb_in; float eps, xs;   // block sums of squares -> rk_row_factor; both null: factor 1
  int R;                         // decoder rows per workgroup (<= 32; results do not depend on it)
  int CS;                        // workgroups per (head, slab): each recomputes q_h and takes 1 / CS of the output column pairs (few rows:
                                 // more workgroups stream the weights; d / 64 must be a multiple of CS; results do not depend on it)
};

#define DEC_NW 8            // waves per workgroup of both kernels: K is split eight ways, the partial tiles meet in a fixed-order LDS tree
__device__ __forceinline__ void dec_tree_reduce2(f32x16& a0, f32x16& a1, float* red, int wave, int lane) {
  // ((w0 + w4) + (w2 + w6)) + ((w1 + w5) + (w3 + w7)), fixed order; red: 4 x 2 x 16 x 64 floats = 32 KiB
#pragma unroll
  for (int half_n = DEC_NW / 2; half_n >= 1; half_n >>= 1) {
    if (wave >= half_n && wave < 2 * half_n) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { red[((wave - half_n) * 2 + 0) * 1024 + r * 64 + lane] = a0[r]; red[((wave - half_n) * 2 + 1) * 1024 + r * 64 + lane] = a1[r]; }
    }
    __syncthreads();
    if (wave < half_n) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { a0[r] += red[(wave * 2 + 0) * 1024 + r * 64 + lane]; a1[r] += red[(wave * 2 + 1) * 1024 + r * 64 + lane]; }
    }
    __syncthreads();
  }
}

// out[n][m] += sum_k W[n][k] X[m][k] over this wave's K range: two 32-row weight tiles (w0, w1 = w0 + 32 rows), activation
// rows either from global or from LDS (same pointer arithmetic).  The lane's pointers already carry its row and 8 * hh.
__device__ __forceinline__ void dec_mfma_krange(const half_t* w0, const half_t* w1, const half_t* xr, int kq, f32x16& a0, f32x16& a1) {
  int k = 0;
  for (; k + 128 <= kq; k += 128) {                                  // eight k16 steps (24 loads) in flight
    half8 xf[8], f0[8], f1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { f0[u] = *(const half8*)(w0 + k + 16 * u); f1[u] = *(const half8*)(w1 + k + 16 * u); xf[u] = *(const half8*)(xr + k + 16 * u); }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0[u], xf[u], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1[u], xf[u], a1, 0, 0, 0);
    }
  }
  for (; k < kq; k += 16) {
    const half8 xf = *(const half8*)(xr + k);
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)(w0 + k), xf, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)(w1 + k), xf, a1, 0, 0, 0);
  }
}

#define DECQ_STR 72      // sQ row stride in halfs (144 B: rows start in different 16-byte slots)
// grid = (H, ceil(M / R), CS); 512 threads.  d must be a multiple of 128 (eight K ranges of whole k16 steps).
__global__ __launch_bounds__(64 * DEC_NW) void dec_cross_qk_kernel(DecQKArgs p) {
  __shared__ __attribute__((aligned(16))) float red[DEC_NW * 32 * DECQ_STR / 2];   // 36 KiB: the tree's 32 KiB, later eight 4.5-KiB output slabs
  __shared__ __attribute__((aligned(16))) half_t sQ[32 * DECQ_STR];
  const int h = blockIdx.x, m0 = blockIdx.y * p.R;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int m = min(m0 + (l31 < p.R ? l31 : 0), p.M - 1);          // rows beyond the slab / the matrix are computed and never stored
  // the row factor of the folded norm is wave 0's alone and sits on the critical path behind the tree: request it first
  float rf = 1.f;
  if (wave == 0) {
    if (p.rowscale) rf = p.rowscale[m];
    else if (p.ssq_in) rf = rk_row_factor(p.ssq_in + (size_t)m * p.nb_in, p.nb_in, p.d, p.eps, p.xs);
  }
  // the W_k^T tiles of this wave's first two output tiles travel while the q projection is computed and reduced.  A wave
  // owns PAIRS of adjacent 32-column tiles (2 w, 2 w + 1), (2 w + 16, 2 w + 17), ...: 64 columns = one 128-byte line per row
  const int ntile = p.d >> 5;
  const int ppw = (ntile >> 1) / p.CS;                              // column pairs of this workgroup: [pair0, pair0 + ppw)
  const int pair0 = blockIdx.z * ppw;
  half8 wpre[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t = min(2 * (pair0 + min(wave, ppw - 1)) + i, ntile - 1);
    const half_t* wr = p.wkT + ((size_t)h * p.d + t * 32 + l31) * 64 + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wpre[i][ks] = *(const half8*)(wr + 16 * ks);
  }
  // ---- q_h [32 rows x 64] = W_q,h x : K = d in eight contiguous ranges, one per wave ----
  const int kq = p.d / DEC_NW;
  f32x16 a0, a1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  {
    const half_t* xr = p.x + (size_t)m * p.ldx + wave * kq + 8 * hh;
    const half_t* w0 = p.wq + (size_t)(h * 64 + l31) * p.d + wave * kq + 8 * hh;
    dec_mfma_krange(w0, w0 + (size_t)32 * p.d, xr, kq, a0, a1);
  }
  dec_tree_reduce2(a0, a1, red, wave, lane);
  if (wave == 0) {
    // C layout: lane holds activation row l31, weight rows (= q columns) 8 q + 4 hh + j of each 32-row tile
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      half4 o0, o1;
#pragma unroll
      for (int j = 0; j < 4; ++j) { o0[j] = f2h_sat(a0[4 * q + j] * rf); o1[j] = f2h_sat(a1[4 * q + j] * rf); }
      *(half4*)(sQ + l31 * DECQ_STR + 8 * q + 4 * hh) = o0;
      *(half4*)(sQ + l31 * DECQ_STR + 32 + 8 * q + 4 * hh) = o1;
    }
  }
  __syncthreads();                                                  // q_h is in LDS; the tree's scratch is free: it becomes the output staging
  // ---- qk_h [32 rows x d] = q_h W_k,h : K = 64.  A lane's accumulators are 4-column pieces of 32 different rows, so each
  //      pair of tiles goes through a wave-private LDS slab [32 rows][64 columns] (rows padded to 144 B) and leaves as whole
  //      128-byte row segments, 16 bytes per lane (stored straight from the accumulators the same bytes went out as 16-byte
  //      pieces of 32 rows per instruction: 20 us per launch at 320 rows, most of it the stores) ----
  half8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const half8*)(sQ + l31 * DECQ_STR + 16 * ks + 8 * hh);
  half_t* stage = (half_t*)red + wave * (32 * DECQ_STR);            // 4.5 KiB per wave of the 32 KiB tree scratch
  const int prow = lane >> 3, pch = lane & 7;                       // write-back: 8 rows x 8 16-byte pieces per instruction
  for (int pr = wave; pr < ppw; pr += DEC_NW) {
    const int t0 = 2 * (pair0 + pr);
    half8 wf[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (pr == wave) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[i][ks] = wpre[i][ks];
      } else {
        const int t = min(t0 + i, ntile - 1);
        const half_t* wr = p.wkT + ((size_t)h * p.d + t * 32 + l31) * 64 + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[i][ks] = *(const half8*)(wr + 16 * ks);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][ks], qf[ks], o, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const half4 v = {f2h_sat(o[4 * q]), f2h_sat(o[4 * q + 1]), f2h_sat(o[4 * q + 2]), f2h_sat(o[4 * q + 3])};
        *(half4*)(stage + l31 * DECQ_STR + 32 * i + 8 * q + 4 * hh) = v;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                             // lgkmcnt(0): the slab is written (wave-private: no barrier)
    __builtin_amdgcn_wave_barrier();
    const int ncols = min(64, p.d - t0 * 32);                       // (d = 32 x odd: the last pair holds one tile)
#pragma unroll
    for (int rr = 0; rr < 32; rr += 8) {
      const int row = rr + prow;
      const half8 v = *(const half8*)(stage + row * DECQ_STR + 8 * pch);
      if (row < p.R && m0 + row < p.M && 8 * pch < ncols)
        *(half8*)(p.qk + ((size_t)(m0 + row) * p.H + h) * p.d + t0 * 32 + 8 * pch) = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();                                // slab fully read before the next pair overwrites it
  }
}

struct DecCVArgs {
  const float* part;     // [rows, nch, H, d]  partial sums of the chunk kernel (this pass's rows)
  const float* stat;     // [rows, nch, H, 2]
  const int* seq_off; const int* row_seq; int Ld, row0;   // decoder row -> encoder sequence (as XAttnArgs)
  const half_t* wv;      // [H*64, d]  cross-attention W_v rows of this layer
  half_t* out;           // [rows, ldo]  ctx (fp16), column h*64 + n
  int nr, d, H, nch, ldo;
  int R;                 // decoder rows per workgroup (<= 32; the host picks it from the row count - results do not depend on it)
};
#define DECV_MAXCH 64    // chunks per row this kernel handles (sequences up to 4096 keys); longer: the unfused pair
// dynamic LDS: max(R x (d + 8) halfs, 32 KiB) [merged rows, later the reduction tree] + 2 x R x 64 floats [chunk tables] + 64 words
__host__ __device__ inline size_t dec_cv_rows_bytes(int d, int R) {
  size_t rows = (size_t)R * (d + 8) * 2;
  return rows < 32768 ? 32768 : rows;
}
__host__ __device__ inline size_t dec_cv_lds_bytes(int d, int R) { return dec_cv_rows_bytes(d, R) + (size_t)2 * R * DECV_MAXCH * 4 + 64 * 4; }
// grid = (H, ceil(nr / R)); 512 threads.  d must be a multiple of 128.
__global__ __launch_bounds__(64 * DEC_NW) void dec_cross_cv_kernel(DecCVArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char decv_smem[];
  const int wstr = p.d + 8;
  const size_t rows_bytes = dec_cv_rows_bytes(p.d, p.R);
  half_t* sWS = (half_t*)decv_smem;
  float* red = (float*)decv_smem;                                   // aliases sWS once every wave has its fragments
  float* sWt = (float*)(decv_smem + rows_bytes);                    // [R][64] chunk maxima, then weights
  float* sSm = sWt + p.R * DECV_MAXCH;                              // [R][64] chunk sums
  int* sNv = (int*)(sSm + p.R * DECV_MAXCH);                        // [32] chunks with keys
  float* sInv = (float*)(sNv + 32);                                 // [32] 1 / denominator
  const int h = blockIdx.x, mbase = blockIdx.y * p.R;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int kq = p.d / DEC_NW;
  // ---- chunk statistics of the workgroup's rows -> weights (xattn_combine_kernel's arithmetic, per row in chunk order) ----
  for (int idx = tid; idx < p.R * DECV_MAXCH; idx += 64 * DEC_NW) {
    const int r = idx / DECV_MAXCH, ck = idx % DECV_MAXCH, m = mbase + r;
    if (m < p.nr) {
      const int b = p.row_seq ? p.row_seq[p.row0 + m] : (p.row0 + m) / p.Ld;
      const int L = p.seq_off[b + 1] - p.seq_off[b];
      const int nv = min(p.nch, (L + 63) >> 6);
      if (ck < nv) {
        const float2 ms = *(const float2*)(p.stat + (((size_t)m * p.nch + ck) * p.H + h) * 2);
        sWt[r * DECV_MAXCH + ck] = ms.x; sSm[r * DECV_MAXCH + ck] = ms.y;
      }
      if (ck == 0) sNv[r] = nv;
    } else if (ck == 0) sNv[r] = 0;
  }
  __syncthreads();
  if (tid < p.R) {
    const int nv = sNv[tid];
    float* w = sWt + tid * DECV_MAXCH;
    const float* sm = sSm + tid * DECV_MAXCH;
    float gmax = -1e30f;
    for (int ck = 0; ck < nv; ++ck) gmax = fmaxf(gmax, w[ck]);
    float den = 0.f;
    for (int ck = 0; ck < nv; ++ck) den += __expf(w[ck] - gmax) * sm[ck];
    for (int ck = 0; ck < nv; ++ck) w[ck] = __expf(w[ck] - gmax);
    sInv[tid] = nv > 0 ? 1.0f / den : 0.f;
  }
  __syncthreads();
  // ---- merged, normalised sums of the raw encoder rows -> fp16 rows in LDS: wave w takes rows w, w + 8, ...; a lane owns
  //      columns 4 lane + 256 i (whole 1-KiB pieces per load instruction); four chunks x four pieces (16 loads) in flight,
  //      the last round of chunks predicated by a zero weight (clamped address), never serial ----
  const int npc = (p.d + 255) >> 8;                                 // 256-column pieces
  const int ngr = (npc + 3) >> 2;                                   // groups of four pieces (1024 columns)
  // work items (row, piece group) dealt to the waves; few rows with many chunks (a setwise prompt: 23) split a row's
  // columns over the waves as well: items (row, single piece)
  const bool fine = p.R * ngr < DEC_NW;
  const int nitem = fine ? p.R * npc : p.R * ngr;
  for (int it = wave; it < nitem; it += DEC_NW) {
    const int r = fine ? it / npc : it / ngr;
    const int c0 = fine ? it % npc : 4 * (it % ngr);
    const int cn = fine ? 1 : 4;                                    // pieces of this item
    const int m = mbase + r;
    if (m >= p.nr) continue;                                        // (rows beyond the pass: their MFMA columns are never stored)
    const int nv = sNv[r];
    const float* w = sWt + r * DECV_MAXCH;
    const float inv = sInv[r];
    const float* prow = p.part + (((size_t)m * p.nch) * p.H + h) * p.d + 4 * lane;
    const size_t cstride = (size_t)p.H * p.d;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!fine) {
      for (int ck0 = 0; ck0 < nv; ck0 += 4) {                       // 4 chunks x 4 pieces = 16 loads in flight
        f32x4 v[4][4];
        float wk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ck = min(ck0 + u, nv - 1);
          wk[u] = ck0 + u < nv ? w[ck] : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int col = (c0 + i) * 256 + 4 * lane;
            v[u][i] = col < p.d ? *(const f32x4*)(prow + ck * cstride + (c0 + i) * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(wk[u], v[u][i][j], acc[i][j]);   // explicit fma: the two item shapes must round alike
      }
    } else {
      const int col = c0 * 256 + 4 * lane;
      for (int ck0 = 0; ck0 < nv; ck0 += 16) {                      // one piece: 16 chunks in flight
        f32x4 v[16];
        float wk[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int ck = min(ck0 + u, nv - 1);
          wk[u] = ck0 + u < nv ? w[ck] : 0.f;
          v[u] = col < p.d ? *(const f32x4*)(prow + ck * cstride + c0 * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_fmaf(wk[u], v[u][j], acc[0][j]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = (c0 + i) * 256 + 4 * lane;
      if (i < cn && col < p.d) {
        const half4 o = {f2h_sat(acc[i][0] * inv), f2h_sat(acc[i][1] * inv), f2h_sat(acc[i][2] * inv), f2h_sat(acc[i][3] * inv)};
        *(half4*)(sWS + r * wstr + col) = o;
      }
    }
  }
  __syncthreads();
  // ---- ctx_h [32 rows x 64] = W_v,h (.) : K = d in eight contiguous ranges, one per wave; fixed-order LDS tree ----
  f32x16 a0, a1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  {
    const half_t* xr = sWS + (l31 < p.R ? l31 : 0) * wstr + wave * kq + 8 * hh;   // (columns beyond R repeat row 0: never stored)
    const half_t* w0 = p.wv + (size_t)(h * 64 + l31) * p.d + wave * kq + 8 * hh;
    dec_mfma_krange(w0, w0 + (size_t)32 * p.d, xr, kq, a0, a1);
  }
  __syncthreads();                                                  // every wave has read its sWS fragments: the tree may overwrite them
  dec_tree_reduce2(a0, a1, red, wave, lane);
  if (wave == 0 && l31 < p.R && mbase + l31 < p.nr) {
    half_t* orow = p.out + (size_t)(mbase + l31) * p.ldo + h * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const half4 o0 = {f2h_sat(a0[4 * q]), f2h_sat(a0[4 * q + 1]), f2h_sat(a0[4 * q + 2]), f2h_sat(a0[4 * q + 3])};
      const half4 o1 = {f2h_sat(a1[4 * q]), f2h_sat(a1[4 * q + 1]), f2h_sat(a1[4 * q + 2]), f2h_sat(a1[4 * q + 3])};
      *(half4*)(orow + 8 * q + 4 * hh) = o0;
      *(half4*)(orow + 32 + 8 * q + 4 * hh) = o1;
    }
  }
}
