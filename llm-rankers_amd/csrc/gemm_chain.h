// Chained GEMM launch for gfx950: TWO row-local GEMMs of an encoder layer in ONE persistent launch, the second reading row
// panels of the first's output as soon as they are complete (row-panel ready flags).
//
// Replaces, per encoder layer (hf: models/t5/modeling_t5.py:304,367 + 106-123 via ref: llmrankers/pointwise.py:117-119), the launch
// pairs   {O projection + residual, rowscale_kernel, FFN-in (+ GEGLU / ReLU)}   and
//         {FFN-out projection + residual, rowscale_kernel, next layer's QKV projection}.
// Why: the fp32 read-modify-write epilogue of the residual GEMMs is HBM-bound (640 KB per 256 x 256 tile; 18 us when all 256
// CUs run it at once - they do, a launch of equal tiles runs in lock step) while their main loops are MFMA-bound: the matrix
// pipe and HBM took turns inside those launches and re-synchronised at every kernel boundary (O 25 %, FFN-out 36 % of the MFMA
// peak for two rounds of reviews).  Here the producer tiles are spread evenly among the consumer tiles (gemm_chain_sched.h), so
// at any moment ~15-25 % of the CUs are in a producer epilogue and the others in main loops, the tail of the producer GEMM is
// filled with consumer tiles, and two kernel boundaries + the statistics kernel per pair are gone.
//
// The tile loop is gemm_pp2_kernel's (gemm.h: 256 x 256 x 64 ping-pong, asm LDS-DMA, counted vmcnt) - same K order, same
// epilogue code: results are BIT-IDENTICAL to the separate launches (tests).  What is new:
//   * work distribution: 8 queues (one per XCD: block b's home queue is b & 7 - a speed hint only), each an ordered list of its
//     tiles; a workgroup claims the next position of its queue with one atomic add and steals from the other queues when its
//     own is empty.  A queue is self-contained and dependency-ordered, and only RUNNING workgroups hold claims, so the launch
//     completes with any number of resident workgroups (two encoder streams can interleave two such launches on the chip);
//   * hand-off (MI355X_MICROARCH.md "inter-workgroup visibility", form R1): the producer epilogue writes the fp16 stream copy and
//     the block sums of squares as write-through (sc1) stores; every wave drains (asm vmcnt(0)), the workgroup barriers, ONE lane
//     takes an arrival ticket of the row panel; the LAST arriver turns the panel's block sums into row factors (rk_row_factor's
//     own addition order), stores them sc1, drains, and ONE lane stores the panel's flag (= the launch's epoch: nothing is
//     zeroed between launches; the ticket counter resets itself).  A consumer tile's flag is requested one tile AHEAD (an sc1 load
//     under the last MFMAs of the tile before), so a ready panel costs nothing; an unready one (rare: the producers run 2-3
//     blocks ahead) is waited for AFTER the current epilogue by one polling lane with s_sleep and a wall-clock timeout that sets
//     an error word instead of hanging the GPU;
//   * the consumer side issues `buffer_inv sc1` (this CU's L1) before it touches a panel.  The XCD's L2 cannot hold a stale line of a
//     panel: every line of it is written (write-through, dropped from the writer's L2) before any CU of this launch reads it, and
//     kernel boundaries invalidate the L2s.
#pragma once
#include "gemm.h"
#include "gemm_chain_sched.h"

struct ChainArgs {
  GemmArgs prod;        // fp32 residual GEMM, folded-norm producer (xraw, ssq set)
  GemmArgs cons;        // consumer: A = prod.xraw, rowscale = `rowscale` below (RS epilogue)
  int* heads;           // [CHAIN_QUEUES] next position of every queue; zero before the launch
  int* cnt;             // [row panels] arrival tickets (self-resetting)
  unsigned* flag;       // [row panels] == epoch: panel published
  unsigned epoch;       // unique per launch on these flag words, never 0
  int* err;             // pinned host word: non-zero after a flag wait timed out (results are garbage then; the host checks it)
  float* rowscale;      // [M (padded)] row factors, written by the last-arriving producer tile of a panel
  int lead_blocks;      // producer lead in blocks of CHAIN_R panels
  unsigned long long* trace;   // measurement builds: per workgroup and tile {start, end of main loop, end of epilogue} wall clock
  int debug;            // measurement builds: knock-outs for fault / bottleneck hunting (results are garbage): 1 no row-factor loads, 2 no flag
                        // load / wait (always "ready"), 4 no publish, 8 no epilogues, 16 no DMA, 32 claims through the synchronous path only
};
#ifdef RK_MEASURE
#define CHAIN_DBG(bit) ((a.debug & (bit)) != 0)
#else
#define CHAIN_DBG(bit) false
#endif

#define CHAIN_TIMEOUT_TICKS 20000000LL    // 0.2 s of the 100 MHz wall clock

// one lane: wait until *flag == epoch.  Bounded: a lost hand-off becomes an error word, not a hung GPU.
__device__ __forceinline__ void chain_spin(const unsigned* flag, unsigned epoch, int* err) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
    __builtin_amdgcn_s_sleep(16);
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;       // someone already gave up: do not wait again
    if (wall_clock64() - t0 > CHAIN_TIMEOUT_TICKS) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
  }
}

// rk_row_factor (common.h) with agent-scope loads: the block sums were written by other workgroups of this launch
__device__ __forceinline__ float chain_row_factor(const float* src, int nb, int d, float eps, float xs) {
  float s = 0.f;
  for (int j = 0; j < nb; ++j) s += __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return rsqrtf(s / (float)d + eps) / xs;
}

template <int EPI_C>
__global__ __launch_bounds__(512, 2) void gemm_chain_kernel(ChainArgs a) {
  constexpr int HALF = 128 * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  half_t* smem = (half_t*)gemm_smem;
  volatile int* sched = (volatile int*)(gemm_smem + 163840 - 64);   // [0] tile after next, [1] next tile ready, [2] last arriver
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int grp = wave >> 2, wm = wave & 1, wn = (wave >> 1) & 3;
  const int M = a.prod.M;
  const int tiles_m = (M + 255) >> 8;
  const int NP = (a.prod.N + 255) >> 8, NC = (a.cons.N + 255) >> 8;
  // ---- per-tile state (uniform) ----
  int m0 = 0, n0 = 0, nk = 2;
  bool is_prod = false;
  const half_t* curA = a.prod.A;
  const half_t* curW = a.prod.W;
  unsigned off[4][2];   // kind 0 = A0, 1 = A1, 2 = W0, 3 = W1; byte offsets of this wave's two DMA instructions
  auto set_tile = [&](int code) {
    is_prod = chain_is_prod(code);
    m0 = chain_tm(code) * 256; n0 = chain_tn(code) * 256;
    const int lda = is_prod ? a.prod.lda : a.cons.lda, ldw = is_prod ? a.prod.ldw : a.cons.ldw;
    const int N = is_prod ? a.prod.N : a.cons.N;
    curA = is_prod ? a.prod.A : a.cons.A;
    curW = is_prod ? a.prod.W : a.cons.W;
    nk = (is_prod ? a.prod.K : a.cons.K) >> 6;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wave * 2 + j) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int ga = m0 + 128 * (r >> 6) + 64 * h + (r & 63);
        ga = ga < M ? ga : M - 1;
        off[h][j] = ((unsigned)ga * (unsigned)lda + chunk * 8) * 2u;
        int gw = n0 + 64 * (r >> 5) + 32 * h + (r & 31);
        gw = gw < N ? gw : N - 1;
        off[2 + h][j] = ((unsigned)gw * (unsigned)ldw + chunk * 8) * 2u;
      }
    }
  };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)gemm_smem);
  auto issue1 = [&](auto kindc, int stage, int tile, auto jc) {
    constexpr int kind = decltype(kindc)::value, j = decltype(jc)::value;
    const char* base = (const char*)((kind < 2 ? curA : curW) + tile * 64);
    const unsigned dst = lds0 + (unsigned)(((kind * 2 + stage) * HALF + (wave * 2 + j) * 512) * 2);
    const unsigned o = off[kind][j];
    if (CHAIN_DBG(16)) return;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(o), "s"(base) : "memory", "m0");
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
  f32x16 acc[2][4];

  // one super-phase of gemm_pp2_kernel's loop (gemm.h: same reads, same MFMA order, same DMA issue points and counted waits)
  auto sp = [&](auto spc, auto issuec, auto waitc, int t) {
    constexpr int SP = decltype(spc)::value, WAIT = decltype(waitc)::value;
    constexpr bool ISSUE = decltype(issuec)::value;
    const int st = t & 1;
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
    if constexpr (WAIT >= 0) gemm_wait_vmcnt<WAIT>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (GEMM_PP2_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ni = q >> 1, mi = q & 1;
        acc[ni][2 * SP + mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ni == 0 ? w0F[ks] : w1F[ks], aF[mi][ks], acc[ni][2 * SP + mi], 0, 0, 0);
        if constexpr (ISSUE) {
          if (q == GEMM_PP2_ISSUE_Q) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SP == 0) {
              if (ks == 0) issue1(I3{}, st ^ 1, t + 1, I0{}); if (ks == 1) issue1(I3{}, st ^ 1, t + 1, I1{});
              if (ks == 2) issue1(I1{}, st ^ 1, t + 1, I0{}); if (ks == 3) issue1(I1{}, st ^ 1, t + 1, I1{});
            } else {
              if (ks == 0) issue1(I0{}, st, t + 2, I0{}); if (ks == 1) issue1(I0{}, st, t + 2, I1{});
              if (ks == 2) issue1(I2{}, st, t + 2, I0{}); if (ks == 3) issue1(I2{}, st, t + 2, I1{});
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
  auto issue_prologue = [&]() {   // half-tiles 0..5 = K tile 0 complete, A0 and W0 of K tile 1
    issue1(I0{}, 0, 0, I0{}); issue1(I0{}, 0, 0, I1{});
    issue1(I2{}, 0, 0, I0{}); issue1(I2{}, 0, 0, I1{});
    issue1(I3{}, 0, 0, I0{}); issue1(I3{}, 0, 0, I1{});
    issue1(I1{}, 0, 0, I0{}); issue1(I1{}, 0, 0, I1{});
    issue1(I0{}, 1, 1, I0{}); issue1(I0{}, 1, 1, I1{});
    issue1(I2{}, 1, 1, I0{}); issue1(I2{}, 1, 1, I1{});
  };
  using Yes = integral_constant<bool, true>; using No = integral_constant<bool, false>;
  using W4 = integral_constant<int, 4>; using W2 = integral_constant<int, 2>; using W0c = integral_constant<int, 0>;
  using WN = integral_constant<int, -1>;

  // ---- work distribution (wave 0 claims, everyone reads the result from LDS) ----
  int qcur = blockIdx.x & (CHAIN_QUEUES - 1), tried = 0;
  auto pull_sync = [&]() -> int {            // wave-uniform; a returning atomic per attempt
    int tm_ = tiles_m, np_ = NP, nc_ = NC;
    asm volatile("" : "+s"(tm_), "+s"(np_), "+s"(nc_));
    while (tried < CHAIN_QUEUES) {
      const int len = chain_queue_len(qcur, tm_, np_, nc_);
      int idx = len;
      if (len > 0) {
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(a.heads + qcur, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idx = __builtin_amdgcn_readfirstlane(v);
      }
      if (idx < len) return chain_decode(qcur, idx, tm_, np_, nc_, a.lead_blocks);
      qcur = (qcur + 1) & (CHAIN_QUEUES - 1);
      ++tried;
    }
    return -1;
  };
  // everyone: block until the tile's inputs are published (consumer tiles), then drop this CU's L1
  auto wait_ready = [&](int code) {
    if (!chain_is_prod(code) && !CHAIN_DBG(2)) {
      if (tid == 0) chain_spin(a.flag + chain_tm(code), a.epoch, a.err);
      __syncthreads();
    }
    asm volatile("buffer_inv sc1" ::: "memory");
  };

  if (wave == 0) {
    const int c0 = pull_sync();
    const int c1 = c0 >= 0 ? pull_sync() : -1;
    if (lane == 0) { sched[0] = c0; sched[1] = c1; }
  }
  __syncthreads();
  int cur = __builtin_amdgcn_readfirstlane(sched[0]), next = __builtin_amdgcn_readfirstlane(sched[1]);   // (uniform: tile state lives in SGPRs)
  if (cur < 0) return;
  wait_ready(cur);
  set_tile(cur);
  issue_prologue();
#ifdef RK_MEASURE
  int n_done = 0;
#endif
  while (true) {
#ifdef RK_MEASURE
    unsigned long long t_start = 0;
    if (a.trace && tid == 0) t_start = wall_clock64();
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    gemm_wait_vmcnt<6>();
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
    sp(I0{}, Yes{}, W4{}, t);
    sp(I1{}, No{}, W2{}, t);
    ++t;
    sp(I0{}, No{}, W0c{}, t);
    // nothing is in flight: under the last 16 MFMAs travel (a) the row factors of a consumer tile, (b) wave 0: the ready flag of
    // the NEXT tile and the claim of the tile after it - all awaited together below, before anything else is issued
    const float tile_scale = is_prod ? a.prod.scale : a.cons.scale;
    float rsc[4] = {tile_scale, tile_scale, tile_scale, tile_scale};
    if (!is_prod && !CHAIN_DBG(1)) {
      const unsigned roff = (unsigned)(m0 + wm * 128 + l31) * 4u;
      // (s_nop 4: the base pair may have just been restored from a spill lane by v_readlane - a VALU write of an SGPR needs five
      // wait states before a VMEM instruction reads it, and the compiler's hazard recognizer does not look inside inline asm;
      // without it the GEGLU instantiation loaded through a stale pointer half: memory faults on the first launch)
      asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=&v"(rsc[0]) : "v"(roff), "s"(a.rowscale));
      asm volatile("global_load_dword %0, %1, %2 offset:128" : "=&v"(rsc[1]) : "v"(roff), "s"(a.rowscale));
      asm volatile("global_load_dword %0, %1, %2 offset:256" : "=&v"(rsc[2]) : "v"(roff), "s"(a.rowscale));
      asm volatile("global_load_dword %0, %1, %2 offset:384" : "=&v"(rsc[3]) : "v"(roff), "s"(a.rowscale));
    }
    unsigned flagv = a.epoch;
    int idxv = 0x7fffffff;
    const bool claim = wave == 0 && next >= 0 && tried < CHAIN_QUEUES;
    if (wave == 0) {
      if (next >= 0 && !chain_is_prod(next) && !CHAIN_DBG(2)) {
        const unsigned foff = (unsigned)chain_tm(next) * 4u;
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 sc1" : "=&v"(flagv) : "v"(foff), "s"(a.flag));
      }
      if (claim && lane == 0 && !CHAIN_DBG(32)) {
        const unsigned hoff = (unsigned)qcur * 4u;
        const int one = 1;
        asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=&v"(idxv) : "v"(hoff), "v"(one), "s"(a.heads) : "memory");
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    sp(I1{}, No{}, WN{}, t);
    if (grp == 0) __builtin_amdgcn_s_barrier();  // re-align the two groups
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rsc[0]), "+v"(rsc[1]), "+v"(rsc[2]), "+v"(rsc[3]), "+v"(flagv), "+v"(idxv) :: "memory");
    if (wave == 0) {
      int n2 = -1;
      if (claim && CHAIN_DBG(32)) n2 = pull_sync();
      else if (claim) {
        const int idx = __builtin_amdgcn_readfirstlane(idxv);
        // (the geometry goes through an opaque statement: the reciprocals of the decode's divisions are loop invariants that
        // would otherwise be formed in front of the tile loop and kept - spilled - across the main loop)
        int tm_ = tiles_m, np_ = NP, nc_ = NC;
        asm volatile("" : "+s"(tm_), "+s"(np_), "+s"(nc_));
        if (idx < chain_queue_len(qcur, tm_, np_, nc_)) n2 = chain_decode(qcur, idx, tm_, np_, nc_, a.lead_blocks);
        else { qcur = (qcur + 1) & (CHAIN_QUEUES - 1); ++tried; n2 = pull_sync(); }        // own queue empty: steal (end of the launch)
      }
      const int rdy = (next < 0 || chain_is_prod(next) || __builtin_amdgcn_readfirstlane(flagv) == a.epoch) ? 1 : 0;
      if (lane == 0) { sched[0] = n2; sched[1] = rdy; }
    }
    __syncthreads();   // every wave is done reading the stages; the scheduler words are visible
    const int next2 = __builtin_amdgcn_readfirstlane(sched[0]);
    const bool ready = __builtin_amdgcn_readfirstlane(sched[1]) != 0;
#ifdef RK_MEASURE
    unsigned long long t_main = 0;
    if (a.trace && tid == 0) t_main = wall_clock64();
#endif
    const int mbase = m0 + wm * 128, nbase = n0 + wn * 64, tm_cur = m0 >> 8;
    const bool was_prod = is_prod;
    if (!was_prod) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rsc[i] *= tile_scale;
    }
    // the next tile's first loads go out now and land under this tile's epilogue (gemm_pp2_kernel: targets and staging area)
    if (next >= 0 && ready) {
      asm volatile("buffer_inv sc1" ::: "memory");
      set_tile(next);
      issue_prologue();
    }
    // (the lane number is re-derived here through an opaque statement: with TWO epilogues in the kernel their loop-invariant lane
    // arithmetic - row / chunk of the write-back passes, staging addresses - was hoisted in front of the tile loop and lived, in
    // part spilled, across the main loop; this way it is formed after the last MFMA and the loop keeps gemm_pp2_kernel's registers)
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    if (was_prod) {
      if (!CHAIN_DBG(8))
      gemm_epilogue_staged<EPI_RESID_F32, 2, 4, false, 16, 0, true>(a.prod, acc, mbase, nbase, lane_e, gemm_smem + 114688 + wave * 4608, rsc);
      // ---- publish: write-through stores drained by every wave, then ONE arrival ticket for the row panel ----
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) sched[2] = CHAIN_DBG(4) ? 0 : (__hip_atomic_fetch_add(a.cnt + tm_cur, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NP - 1) ? 1 : 0;
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(sched[2]) != 0) {            // last producer tile of the panel: block sums -> row factors, then the flag
        asm volatile("buffer_inv sc1" ::: "memory");
        const int m = tm_cur * 256 + tid;
        if (tid < 256 && m < M)
          __hip_atomic_store(a.rowscale + m, chain_row_factor(a.prod.ssq + (size_t)m * a.prod.nb, a.prod.nb, a.prod.N, a.prod.eps_in, a.prod.xs),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          __hip_atomic_store(a.cnt + tm_cur, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(a.flag + tm_cur, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    } else if (!CHAIN_DBG(8)) {
      gemm_epilogue_staged<EPI_C, 2, 4, false, 32>(a.cons, acc, mbase, nbase, lane_e, gemm_smem + 114688 + wave * 4608, rsc);
    }
#ifdef RK_MEASURE
    if (a.trace && tid == 0 && n_done < 64) {
      unsigned long long* tr = a.trace + ((size_t)blockIdx.x * 64 + n_done) * 4;
      tr[0] = t_start; tr[1] = t_main; tr[2] = wall_clock64(); tr[3] = (unsigned long long)(unsigned)cur | ((unsigned long long)(ready ? 1 : 0) << 32);
    }
    ++n_done;
#endif
    if (next < 0) break;
    if (!ready) {                     // rare: the panel was not published when this tile's main loop ended
      wait_ready(next);
      set_tile(next);
      issue_prologue();
    }
    cur = next;
    next = next2;
    __syncthreads();   // staging rows are read before the next tile's DMA wraps around to W1 | stage 1
  }
}
