// What would a PERSISTENT few-row decoder pass get on this chip?  (round-5 review item 2: "build the two-layer prototype and commit
// its trace".)  The flan-t5-large decoder layer of one setwise compare is a chain of eight dependent few-row GEMVs (M = 2 rows;
// q|k|v 3072x1024, o 1024x1024, cross-q 1024x1024, W_k^T 1024x1024, W_v 1024x1024, cross-o 1024x1024, FFN-in 5632x1024, FFN-out
// 1024x2816: 33 MB of fp16 weights) - every phase needs the COMPLETE vector the previous one produced (an all-to-all edge).  This
// probe runs that chain for LAYERS layers three ways and prints microseconds per phase:
//   launches : one kernel per phase, captured in a hipGraph and replayed (what the engine does today, minus attention)
//   persist  : ONE launch, 256 workgroups (one per CU), a grid barrier between phases (monotonic counter, lane-0 release fence
//              before the arrive, relaxed sc1 polls with s_sleep, acquire fence after; bounded spin -> error word)
//   prefetch : the same, and every workgroup loads ITS slice of the NEXT phase's weights into registers before it arrives at the
//              barrier (weights never depend on the previous phase) - the best case of "request the next phase's weights early"
//   barrier  : the persistent launch with the GEMV bodies removed (what the barriers alone cost)
// Arithmetic: fp16 weights, fp32 accumulate, activations squashed to keep the chain bounded; all three forms must produce the same
// final vector (checked).  Build: hipcc --offload-arch=gfx950 -O3 -o exp/probe_persist_gemv tools/probes/probe_persist_gemv.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define NPH 8
struct Phase { int N, K; size_t woff; };                 // y[2][N] = x[2][K] W[N][K]^T
struct Chain { Phase ph[NPH]; int layers; size_t wlayer; };

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one output column n of phase p for both rows: the 64 lanes split K in half8 pieces (K multiple of 512)
template <bool PRE>
__device__ __forceinline__ void gemv_col(const half_t* __restrict__ w, const half_t* x, int K, int lane, float& y0, float& y1, const half8* pre) {
  float a0 = 0.f, a1 = 0.f;
  int i = 0;
  for (int k = lane * 8; k < K; k += 512, ++i) {
    const half8 wv = PRE ? pre[i] : *(const half8*)(w + k);
    const half8 x0 = *(const half8*)(x + k), x1 = *(const half8*)(x + K + k);
#pragma unroll
    for (int j = 0; j < 8; ++j) { a0 += (float)wv[j] * (float)x0[j]; a1 += (float)wv[j] * (float)x1[j]; }
  }
  y0 = wave_sum_f(a0); y1 = wave_sum_f(a1);
}

__device__ __forceinline__ half_t squash(float v) { return (half_t)(v / (1.0f + fabsf(v))); }

// columns of phase p owned by (workgroup g of G, wave w of 4): a contiguous range
__device__ __forceinline__ void col_range(int N, int g, int G, int w, int& c0, int& c1) {
  const int per = (N + G * 4 - 1) / (G * 4);
  c0 = (g * 4 + w) * per; c1 = c0 + per < N ? c0 + per : N; if (c0 > N) c0 = N;
}

__global__ __launch_bounds__(256) void phase_kernel(const half_t* W, Phase ph, const half_t* x, half_t* y) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int c0, c1; col_range(ph.N, blockIdx.x, gridDim.x, w, c0, c1);
  for (int n = c0; n < c1; ++n) {
    float y0, y1;
    gemv_col<false>(W + ph.woff + (size_t)n * ph.K, x, ph.K, lane, y0, y1, nullptr);
    if (lane == 0) { y[n] = squash(y0); y[ph.N + n] = squash(y1); }
  }
}

#define MAXCOLS 6      // columns per wave (5632 / 1024 = 5.5)
#define MAXPRE 6       // half8 pieces per lane and column (2816 / 512 = 5.5)
template <int MODE>    // 0 persist, 1 prefetch, 2 barrier only
__global__ __launch_bounds__(256) void persist_kernel(const half_t* W, Chain ch, half_t* act /* [2][2][8192] ping-pong */, unsigned* bar, unsigned* err) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = blockIdx.x, G = gridDim.x;
  unsigned gen = 0;
  half8 pre[MAXCOLS][MAXPRE];
  auto prefetch = [&](const Phase& ph, const half_t* wl) {
    int c0, c1; col_range(ph.N, g, G, w, c0, c1);
#pragma unroll
    for (int c = 0; c < MAXCOLS; ++c)
#pragma unroll
      for (int i = 0; i < MAXPRE; ++i) {
        const int n = c0 + c, k = lane * 8 + 512 * i;
        if (n < c1 && k < ph.K) pre[c][i] = *(const half8*)(wl + ph.woff + (size_t)n * ph.K + k);
      }
  };
  if (MODE == 1) prefetch(ch.ph[0], W);
  int cur = 0;
  for (int l = 0; l < ch.layers; ++l) {
    const half_t* wl = W + (size_t)l * ch.wlayer;
    for (int p = 0; p < NPH; ++p) {
      const Phase ph = ch.ph[p];
      const half_t* x = act + cur * 16384;
      half_t* y = act + (cur ^ 1) * 16384;
      if (MODE != 2) {
        int c0, c1; col_range(ph.N, g, G, w, c0, c1);
#pragma unroll
        for (int c = 0; c < MAXCOLS; ++c) {
          const int n = c0 + c;
          if (n < c1) {
            float y0, y1;
            if (MODE == 1) gemv_col<true>(nullptr, x, ph.K, lane, y0, y1, pre[c]);
            else gemv_col<false>(wl + ph.woff + (size_t)n * ph.K, x, ph.K, lane, y0, y1, nullptr);
            if (lane == 0) { y[n] = squash(y0); y[ph.N + n] = squash(y1); }
          }
        }
      }
      // the next phase's weights do not depend on this phase: request them BEFORE the barrier
      if (MODE == 1) {
        const int np = p + 1 < NPH ? p + 1 : 0;
        const int nl = p + 1 < NPH ? l : l + 1;
        if (nl < ch.layers) prefetch(ch.ph[np], W + (size_t)nl * ch.wlayer);
      }
      // ---- grid barrier: monotonic counter ----
      __syncthreads();
      if (threadIdx.x == 0) {
        ++gen;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = gen * (unsigned)G;
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > 4000000u) { *err = 1u; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      cur ^= 1;
    }
  }
}

int main() {
  const int layers = 24, G = 256;
  const int dims[NPH][2] = {{3072, 1024}, {1024, 1024}, {1024, 1024}, {1024, 1024}, {1024, 1024}, {1024, 1024}, {5632, 1024}, {1024, 2816}};
  Chain ch{}; ch.layers = layers;
  size_t off = 0;
  for (int p = 0; p < NPH; ++p) { ch.ph[p] = {dims[p][0], dims[p][1], off}; off += (size_t)dims[p][0] * dims[p][1]; }
  ch.wlayer = off;
  // the chain's shapes must connect: K of phase p+1 <= N of phase p (the probe feeds the first K outputs on)
  std::vector<half_t> hw(off * layers);
  unsigned s = 12345u;
  for (size_t i = 0; i < hw.size(); ++i) { s = s * 1664525u + 1013904223u; hw[i] = (half_t)(((int)(s >> 16) % 2001 - 1000) * 4e-5f * 16.f); }
  half_t *dW, *act; unsigned *bar, *err;
  hipMalloc(&dW, hw.size() * 2); hipMemcpy(dW, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&act, 2 * 16384 * 2); hipMalloc(&bar, 4); hipMalloc(&err, 4);
  std::vector<half_t> x0(16384);
  for (int i = 0; i < 16384; ++i) { s = s * 1664525u + 1013904223u; x0[i] = (half_t)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto reset = [&]() { hipMemcpy(act, x0.data(), 16384 * 2, hipMemcpyHostToDevice); hipMemset(act + 16384, 0, 16384 * 2); hipMemset(bar, 0, 4); hipMemset(err, 0, 4); };
  std::vector<half_t> ref(16384), got(16384);
  // ---- launches, as a graph ----
  reset();
  hipGraph_t graph; hipGraphExec_t exec;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  {
    int cur = 0;
    for (int l = 0; l < layers; ++l)
      for (int p = 0; p < NPH; ++p) {
        hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, st, dW + (size_t)l * ch.wlayer, ch.ph[p], act + cur * 16384, act + (cur ^ 1) * 16384);
        cur ^= 1;
      }
  }
  hipStreamEndCapture(st, &graph); hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    reset(); hipEventRecord(e0, st); hipGraphLaunch(exec, st); hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  hipMemcpy(ref.data(), act, 16384 * 2, hipMemcpyDeviceToHost);     // 192 phases: even -> result in buffer 0
  printf("launches (hipGraph, %d kernels)      : %8.1f us total = %6.2f us per phase, %5.1f us per layer\n", layers * NPH, ms * 1e3, ms * 1e3 / (layers * NPH), ms * 1e3 / layers);
  auto run = [&](auto kern, const char* name, bool check) {
    float t = 0; unsigned herr = 0;
    for (int rep = 0; rep < 3; ++rep) {
      reset(); hipEventRecord(e0, st);
      hipLaunchKernelGGL(kern, dim3(G), dim3(256), 0, st, dW, ch, act, bar, err);
      hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&t, e0, e1);
    }
    hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    hipMemcpy(got.data(), act, 16384 * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    if (check) for (int i = 0; i < 1024; ++i) bad += (got[i] != ref[i]) + (got[1024 + i] != ref[1024 + i]);
    printf("%-38s: %8.1f us total = %6.2f us per phase, %5.1f us per layer%s%s\n", name, t * 1e3, t * 1e3 / (layers * NPH), t * 1e3 / layers,
           herr ? "  BARRIER TIMEOUT" : "", check ? (bad ? "  MISMATCH vs launches" : "  == launches (bit-exact)") : "");
  };
  run(persist_kernel<2>, "persistent, barriers only", false);
  run(persist_kernel<0>, "persistent, GEMV after the barrier", true);
  run(persist_kernel<1>, "persistent, next weights prefetched", true);
  printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
