// Few-row GEMV for the decoder of ONE setwise compare (round 6):  C[m][n] (+)= sum_k A[m][k] W[n][k]  for M <= 16 rows.
//
// Replaces the same Linears as gemm_skinny_kernel (hf: models/t5/modeling_t5.py:206-209, 106-123 inside T5Stack's decoder blocks)
// for the call shape of the reference's one-prompt generate / forward (ref: llmrankers/setwise.py:93-95, 184: decoder prefix
// "<pad> Passage", 2 rows; the second greedy step: 3).  Why a second kernel: the weight-streaming MFMA kernel cuts N into
// 32-column tiles - 32 workgroups for the 1 024-column projections, each streaming 64 KB in two dependent rounds, an LDS tree and
// an epilogue: 6.0 / 8.6 / 10.2 us per launch (store / residual / GEGLU; profiles/r06_compare_kernel_trace.txt) where a chip-wide
// launch that does nothing costs ~4 us.  tools/probes/probe_persist_gemv.hip measured the alternative: one wave per output column
// over ALL 256 CUs, the 64 lanes splitting K: 5.35 us per phase including its work (and a persistent pass with grid barriers: 13.9).
//
// Design: grid = min(CUs, ceil(columns / 4)) workgroups of four waves; a workgroup owns a contiguous block of output columns, a
// wave takes them round robin.  The M activation rows are staged once per workgroup in LDS (fp16, <= 90 KB); per column a lane
// loads its K / 64 weight elements (16-byte pieces at a 1 KiB stride: every load instruction is one contiguous KiB; the next
// column's pieces are requested before this column's arithmetic), multiplies with v_dot2_f32_f16 (fp32 accumulate) against the
// rows from LDS, and a fixed xor-shuffle tree adds the 64 lanes.  Lane m finishes row m: fp16 store with the folded-RMSNorm row
// factor (consumer side), fp32 residual add + fp16 stream copy + per-workgroup sums of squares (producer side: `ssq` holds ONE
// partial per producing workgroup, the consumer adds them in a fixed order), GEGLU / ReLU.
// Numerics: the same products as every other GEMM of the engine, another summation order (lane-strided K, then the tree); which
// family a decoder pass uses follows from its row count (<= option dec_gemv_rows: 4 by default, the measured cross-over against the
// MFMA kernel - profiles/r06_few_rows_ab.txt; this kernel takes up to 16) and position count (>= 2) - see DESIGN.md section 4.
#pragma once
#include "gemm.h"

#define GEMV_MAX_ROWS 16
#define GEMV_MAX_PIECES 6          // K <= 3072: 16-byte pieces per lane and weight row (piece i = elements lane * 8 + 512 i ..)

__host__ __device__ inline int gemv_grid(int n_out, int n_cu) { const int g = (n_out + 3) / 4; return g < n_cu ? g : n_cu; }

template <int EPI, int MR>
__global__ __launch_bounds__(256) void gemv_rows_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gemv_smem[];
  constexpr bool GATED = EPI_IS_GATED(EPI);
  half_t* sX = (half_t*)gemv_smem;                                   // [MR][K]
  float* sFac = (float*)(gemv_smem + (size_t)MR * p.K * 2);          // [MR] row factors
  float* sSq = sFac + GEMV_MAX_ROWS;                                 // [4 waves][MR] sums of squares of the new rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_out = GATED ? p.N >> 1 : p.N;
  const int G = gridDim.x, g = blockIdx.x;
  const int cb = (n_out + G - 1) / G;                                 // columns per workgroup
  const int c_begin = g * cb, c_end = min(n_out, c_begin + cb);
  const int npiece = (p.K + 511) >> 9;
  // weight row(s) of output column j
  auto wrow = [&](int j, int which) {
    const int r = GATED ? ((j >> 5) << 6) + (j & 31) + 32 * which : j;
    return p.W + (size_t)r * p.ldw;
  };
  half8 wcur[GATED ? 2 : 1][GEMV_MAX_PIECES], wnext[GATED ? 2 : 1][GEMV_MAX_PIECES];
  auto load_w = [&](int j, half8 (&dst)[GATED ? 2 : 1][GEMV_MAX_PIECES]) {
#pragma unroll
    for (int t = 0; t < (GATED ? 2 : 1); ++t) {
      const half_t* w = wrow(j < n_out ? j : n_out - 1, t);
#pragma unroll
      for (int i = 0; i < GEMV_MAX_PIECES; ++i) {
        const int k = lane * 8 + (i << 9);
        if (i < npiece) dst[t][i] = *(const half8*)(w + (k < p.K ? k : 0));
      }
    }
  };
  // the first column's weights travel while the rows are staged
  int j = c_begin + wave;
  if (j < c_end) load_w(j, wcur);
  for (int idx = tid; idx < MR * (p.K >> 3); idx += 256) {
    const int m = idx / (p.K >> 3), c = idx - m * (p.K >> 3);
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (m < p.M) v = *(const half8*)(p.A + (size_t)m * p.lda + c * 8);
    *(half8*)(sX + (size_t)m * p.K + c * 8) = v;
  }
  // consumer side of the folded RMSNorm: row factor = given, or rsqrt(sum of the producers' partial sums / K + eps) / xs with the
  // partials added in a FIXED order (lane-strided, then the xor tree) - wave w takes rows w, w + 4, ...
  for (int m = wave; m < MR; m += 4) {
    float f = 1.f;
    if (m < p.M) {
      if (p.rowscale) f = p.rowscale[m];
      else if (p.ssq_in) {
        const int per = (p.nb_in + 63) >> 6;
        float s = 0.f;
        for (int q = 0; q < per; ++q) { const int b = lane * per + q; if (b < p.nb_in) s += p.ssq_in[(size_t)m * p.nb_in + b]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        f = rsqrtf(s / (float)p.K + p.eps_in) / p.xs;
      }
    }
    if (lane == 0) sFac[m] = f * p.scale;
  }
  __syncthreads();
  float ssq_acc = 0.f;                                                // lane m: sum of squares of row m over this wave's columns
  for (; j < c_end; j += 4) {
    const int jn = j + 4;
    if (jn < c_end) load_w(jn, wnext);                                // the next column's pieces under this column's arithmetic
    float acc[GATED ? 2 : 1][MR];
#pragma unroll
    for (int t = 0; t < (GATED ? 2 : 1); ++t)
#pragma unroll
      for (int m = 0; m < MR; ++m) acc[t][m] = 0.f;
#pragma unroll
    for (int i = 0; i < GEMV_MAX_PIECES; ++i) {
      const int k = lane * 8 + (i << 9);
      if (i < npiece && k < p.K) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          const half8 xv = *(const half8*)(sX + (size_t)m * p.K + k);
#pragma unroll
          for (int t = 0; t < (GATED ? 2 : 1); ++t) {
            float a = acc[t][m];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
              const half2v w2 = {wcur[t][i][2 * e2], wcur[t][i][2 * e2 + 1]}, x2 = {xv[2 * e2], xv[2 * e2 + 1]};
              a = __builtin_amdgcn_fdot2(w2, x2, a, false);
            }
            acc[t][m] = a;
          }
        }
      }
    }
    // the 64 lanes' partial sums: fixed xor tree, every lane ends up with the total
#pragma unroll
    for (int t = 0; t < (GATED ? 2 : 1); ++t)
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        float a = acc[t][m];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        acc[t][m] = a;
      }
    // lane m finishes row m of column j
    float mine0 = 0.f, mine1 = 0.f;
#pragma unroll
    for (int m = 0; m < MR; ++m) if (lane == m) { mine0 = acc[0][m]; mine1 = acc[GATED ? 1 : 0][m]; }
    if (lane < p.M) {
      const float sc = sFac[lane];
      if constexpr (GATED) {
        ((half_t*)p.C)[(size_t)lane * p.ldc + j] = f2h_sat(gate_act<EPI>(mine0 * sc) * (mine1 * sc));
      } else if constexpr (EPI == EPI_STORE_F16 || EPI == EPI_RELU_F16) {
        const float v = mine0 * sc;
        ((half_t*)p.C)[(size_t)lane * p.ldc + j] = f2h_sat(EPI == EPI_RELU_F16 ? fmaxf(v, 0.f) : v);
      } else if constexpr (EPI == EPI_RESID_F32) {
        float* c = (float*)p.C + (size_t)lane * p.ldc + j;
        const float nv = *c + mine0 * sc;
        *c = nv;
        if (p.xraw) {
          p.xraw[(size_t)lane * p.ldx + j] = f2h_sat(nv * p.xs);
          ssq_acc = __builtin_fmaf(nv, nv, ssq_acc);
        }
      } else {
        ((float*)p.C)[(size_t)lane * p.ldc + j] = mine0 * sc;
      }
    }
#pragma unroll
    for (int t = 0; t < (GATED ? 2 : 1); ++t)
#pragma unroll
      for (int i = 0; i < GEMV_MAX_PIECES; ++i) wcur[t][i] = wnext[t][i];
  }
  if constexpr (EPI == EPI_RESID_F32) {
    if (p.xraw) {
      // producer side: ONE partial sum of squares per row and workgroup - the four waves' shares added in wave order
      if (lane < MR) sSq[wave * MR + lane] = ssq_acc;
      __syncthreads();
      if (tid < p.M) p.ssq[(size_t)tid * p.nb + g] = ((sSq[tid] + sSq[MR + tid]) + sSq[2 * MR + tid]) + sSq[3 * MR + tid];
    }
  }
}
