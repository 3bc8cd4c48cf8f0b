// Stream-K form of the tiled fp16 MFMA GEMM for the small-M regime (round 6).
//
// Replaces the same nn.Linear calls as gemm.h (hf: models/t5/modeling_t5.py:206-209, 106-123; Llama projections) when the
// reference scores ONE call at a time: one setwise compare (ref: llmrankers/setwise.py:93-95, 184), one Llama compare
// (setwise.py:168-172), one rank's 12 - 13 passage share of a doc-sharded query (pointwise.py:102-127 over 1 / 8 of the
// candidates).  At M = 1.4k - 2.4k rows the 1 024-column projections are 24 - 40 tiles of 256 x 256 for 256 CUs, and the tile
// shapes small enough to fill the chip pay one LDS-DMA instruction per MFMA (64 x 64: issue-bound at a third of the big
// kernel's per-CU rate).
//
// Design: ONE persistent workgroup per CU (512 threads = 8 waves, two per SIMD).  The output is cut into 128 x 128 tiles; the
// K loop of every tile into K / 64 steps; the (tile, step) units are laid out in one line - tiles in the XCD-grouped order of
// gemm_tile_coords - and dealt to the workgroups in EQUAL contiguous runs, whatever the tile count (stream-K).  A run covers
// the tail of one tile, whole tiles, the head of another.  Whole tiles go straight through the epilogues of gemm.h (same MFMA
// order and operand slots as every other tile kernel: same bits).  A tile shared by several workgroups: every contributor
// leaves its fp32 partial accumulators in a slab of the workspace as write-through stores (guide G16 R1: sc1 payload, every
// storing wave drains, barrier, ONE lane takes an arrival ticket), and the LAST arriver adds the partials in run order = K
// order - a fixed order, whoever arrives last - and runs the epilogue.  Nobody spins: no residency assumption, no deadlock.
// The result of a split tile differs from the unsplit kernels' in the last fp32 bits (another association of the K sum); which
// tiles are split is a function of (M, N, K, grid) only, so a call shape always gives the same bits.
//
// Main loop: wave tile 32 (m) x 64 (n) = two 32x32x16 accumulators, the LDS image / XOR swizzle / DMA-with-swizzled-source of
// gemm.h, four stages of 32 KiB filled by asm LDS-DMA (saddr form, M0 by hand - see gemm_pp2_kernel) three steps ahead with
// counted vmcnt.  PING-PONG like gemm_pp2_kernel: the two waves of a SIMD (w, w + 4) run one barrier apart - while one group
// executes the eight MFMAs of a step (and issues its four DMA instructions between them) the other issues its twelve fragment
// reads.  (Measured on the first, lock-step form of this kernel with timing knock-outs, M = 2 392, N = 5 632: 0.81 us per step
// = 0.41 us of fragment reads + 0.42 us of MFMAs, one after the other: LDS and matrix pipe took turns.)  The next run segment's
// first three stages go out before the epilogue of the current one; the epilogue stages its rows in the fourth stage buffer and
// the 32 KiB above the stages.
#pragma once
#include "gemm.h"

struct SkArgs {
  float* slabs;   // [2 * grid][128 * 128] fp32 partial tiles: slot 2v = the segment that continues a tile somebody else began, 2v + 1 = the one that begins a tile
  int* cnt;       // [tiles] arrival tickets, zero between launches (the last arriver resets its tile's)
};

#define SK_NST 4
#define SK_STAGE_BYTES 32768                 // A tile [128][64] + W tile [128][64] halfs
#define SK_EPI_WAVE_BYTES 4608               // staged epilogue: one region per wave (gemm_epilogue_staged)
#define SK_SLAB_FLOATS (128 * 128)
#define SK_LDS_BYTES 163840                  // four stages + 32 KiB: waves 0-6 stage their epilogue rows above the stages, wave 7 in stage 3; flag word at the top

// start of virtual workgroup v's run in the unit line / the workgroup whose run holds unit u
__device__ __forceinline__ long sk_start(long v, long U, long G) { return v * U / G; }
__device__ __forceinline__ int sk_owner(long u, long U, long G) { return (int)(((u + 1) * G - 1) / U); }

// KO: timing-only knock-outs for bottleneck hunting (results are garbage; -DRK_MEASURE builds only): 1 = no DMA in the loop, 2 = no
// fragment reads, 4 = no MFMA
template <int EPI, int KO = 0>
__global__ __launch_bounds__(512, 2) void gemm_sk_kernel(GemmArgs p, SkArgs s) {
  constexpr int NST = SK_NST;
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  half_t* smem = (half_t*)gemm_smem;
  int* s_flag = (int*)(gemm_smem + SK_LDS_BYTES - 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int grp = wave >> 2, wm = wave & 1, wn = (wave >> 1) & 1, wq = wm + 2 * grp;   // group g holds row quarters 2g, 2g + 1 (x two column halves)
  unsigned char* epi_stage = wave < 7 ? gemm_smem + NST * SK_STAGE_BYTES + wave * SK_EPI_WAVE_BYTES : gemm_smem + (NST - 1) * SK_STAGE_BYTES;
  const int tiles_m = (p.M + 127) >> 7, tiles_n = (p.N + 127) >> 7, nk = p.K >> 6;
  const long U = (long)tiles_m * tiles_n * nk, G = gridDim.x;
  // virtual index: the workgroups of one XCD (blockIdx.x % 8) take a CONTIGUOUS stretch of the line, so the tiles an XCD works on
  // at one time form a compact block of the output (A / W panels shared through its L2) and the workgroups that share a split
  // tile sit on one XCD except at the seven seams
  int v;
  {
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, q = (int)G >> 3, r = (int)G & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const long u0 = sk_start(v, U, G), u1 = sk_start(v + 1, U, G);
  if (u0 >= u1) return;

  // ---- per-tile state: output origin and this wave's four DMA source offsets (two A, two W instructions per stage) ----
  int m0 = 0, n0 = 0;
  unsigned offA[2], offW[2];
  auto set_tile = [&](int t) {
    const int per_panel = tiles_m * p.group_n;
    const int pn = t / per_panel, rem = t - pn * per_panel;
    const int w = min(p.group_n, tiles_n - pn * p.group_n);
    const int tm = rem / w, tn = pn * p.group_n + rem - tm * w;
    m0 = tm << 7; n0 = tn << 7;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wave * 2 + j) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((r >> 1) & 7);
      const int ga = m0 + r < p.M ? m0 + r : p.M - 1, gw = n0 + r < p.N ? n0 + r : p.N - 1;
      offA[j] = ((unsigned)ga * (unsigned)p.lda + chunk * 8) * 2u;
      offW[j] = ((unsigned)gw * (unsigned)p.ldw + chunk * 8) * 2u;
    }
  };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)gemm_smem);
  // one LDS-DMA instruction: 64 lanes x 16 B = 8 tile rows.  (M0 on the clobber list: the compiler keeps nothing of its own in it.
  // s_nop 3 in front: this kernel has SGPR spills, and a base pair that comes out of a spill lane right here is a VALU write
  // (v_readlane) of an SGPR a VMEM instruction reads - five wait states the hazard recognizer does not see inside inline asm.)
  auto dma = [&](unsigned dst, unsigned o, const char* base) {
    asm volatile("s_nop 3\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(o), "s"(base) : "memory", "m0");
  };
  auto issue_j = [&](int buf, int kt_abs, int j) {        // j = 0, 1: A instructions; 2, 3: W instructions
    const unsigned dst = lds0 + (unsigned)(buf * SK_STAGE_BYTES + (j >> 1) * 16384 + (wave * 2 + (j & 1)) * 1024);
    const char* base = (const char*)((j >> 1) ? p.W : p.A) + (size_t)kt_abs * 128;
    dma(dst, (j >> 1) ? offW[j & 1] : offA[j & 1], base);
  };
  auto issue_stage = [&](int buf, int kt_abs) {
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_j(buf, kt_abs, j);
  };

  const int xs = (l31 >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = ((ks * 2 + hh) ^ xs) << 3;
  const int a_lane = (wq * 32 + l31) * 64, w_lane = 8192 + (wn * 64 + l31) * 64;

  // ---- the run, segment by segment ----
  long u = u0;
  int t = (int)(u / nk), kb = (int)(u - (long)t * nk);
  int len = (int)((long)(nk - kb) < u1 - u ? (long)(nk - kb) : u1 - u);
  set_tile(t);
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < len) issue_stage(i, kb + i);
  while (true) {
    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    // stage 0 of this segment has landed when only the younger prologue stages (4 instructions each) are still in flight
    if constexpr (!(KO & 1)) {
      if (len >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (len == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier behind: its reads meet group 0's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    int cur = 0, nxt = NST - 1;
    for (int kt = 0; kt < len; ++kt) {
      // ---- read section: the step's twelve fragments; then this wave's share of stage kt + 1 must have landed (stages up to
      // kt + 2 are out: one younger stage may still be in flight).  The barrier after it plus the one the other group adds order
      // the landed bytes for every reader (LDS-DMA: only the issuing wave's vmcnt + a barrier the reader has passed do) ----
      const half_t* sb = smem + cur * (SK_STAGE_BYTES / 2);
      half8 af[4], w0[4], w1[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if constexpr (!(KO & 2)) {
          af[ks] = *(const half8*)(sb + a_lane + koff[ks]);
          w0[ks] = *(const half8*)(sb + w_lane + koff[ks]);
          w1[ks] = *(const half8*)(sb + w_lane + 2048 + koff[ks]);
        } else {
          union { unsigned u[4]; half8 h; } x;
#pragma unroll
          for (int i = 0; i < 4; ++i) x.u[i] = ((unsigned)(lane * 2654435761u + ks * 40503u + i * 9973u + wave * 77u) & 0x8FFF8FFFu) | 0x30003000u;
          af[ks] = x.h; w0[ks] = x.h; x.u[0] ^= 0x80000000u; w1[ks] = x.h;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(KO & 1)) {
        if (kt + 2 < len) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA section; stage kt + 3 goes into the buffer of stage kt - 1 (last read two barriers ago by either group) ----
      const bool more = kt + NST - 1 < len && !(KO & 1);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if constexpr (!(KO & 4)) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[ks], af[ks], acc[0][0], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[ks], af[ks], acc[1][0], 0, 0, 0);
        } else {
          asm volatile("" :: "v"(af[ks]), "v"(w0[ks]), "v"(w1[ks]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue_j(nxt, kb + kt + NST - 1, ks);    // one DMA instruction behind every pair of MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      cur = cur + 1 == NST ? 0 : cur + 1;
      nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();           // re-align the two groups
    __syncthreads();                                      // every wave is done reading the stages
    // ---- the next segment's first stages travel under this segment's epilogue / hand-off ----
    const int c_t = t, c_kb = kb, c_len = len, c_m0 = m0, c_n0 = n0;
    u += len;
    const bool have_next = u < u1;
    if (have_next) {
      t = (int)(u / nk); kb = 0;                          // a later segment always begins a tile
      len = (int)((long)nk < u1 - u ? (long)nk : u1 - u);
      set_tile(t);
#pragma unroll
      for (int i = 0; i < NST - 1; ++i)
        if (i < len) issue_stage(i, i);
    }
    bool finish = true;
    if (c_kb != 0 || c_len != nk) {
      // ---- split tile: publish the partial, take a ticket; the last arriver combines in run order ----
      const long tb = (long)c_t * nk;
      const int c_lo = sk_owner(tb, U, G), c_hi = sk_owner(tb + nk - 1, U, G);
      float* mine = s.slabs + (size_t)(2 * v + (c_kb != 0 ? 0 : 1)) * SK_SLAB_FLOATS;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 x = {acc[ni][0][4 * q], acc[ni][0][4 * q + 1], acc[ni][0][4 * q + 2], acc[ni][0][4 * q + 3]};
          const float* dst = mine + (size_t)(((wave * 2 + ni) * 4 + q) * 64 + lane) * 4;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(x) : "memory");
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // EVERY storing wave drains (also the DMA it just issued: they are needed next anyway)
      __syncthreads();
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(s.cnt + c_t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == c_hi - c_lo;
        if (last) {
          __hip_atomic_store(s.cnt + c_t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nobody else touches this tile's ticket any more
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *s_flag = last;
      }
      __syncthreads();
      finish = *s_flag != 0;
      if (finish) {
        f32x16 tot[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[i][r] = 0.f;
        for (int c = c_lo; c <= c_hi; ++c) {
          if (c == v) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int r = 0; r < 16; ++r) tot[i][r] += acc[i][0][r];
          } else {
            const long cs = sk_start(c, U, G);
            const float* src = s.slabs + (size_t)(2 * c + (cs > tb ? 0 : 1)) * SK_SLAB_FLOATS;   // c continues the tile (its run began inside it) or begins it
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 x = *(const f32x4*)(src + (size_t)(((wave * 2 + ni) * 4 + q) * 64 + lane) * 4);
                tot[ni][4 * q] += x[0]; tot[ni][4 * q + 1] += x[1]; tot[ni][4 * q + 2] += x[2]; tot[ni][4 * q + 3] += x[3];
              }
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][0] = tot[i];
      }
    }
    if (finish) {
      float rsc[1];
      gemm_row_factors<1>(p, c_m0 + wq * 32, l31, rsc);
      constexpr bool F32OUT = EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32;
      gemm_epilogue_staged<EPI, 2, 1, false, F32OUT ? 16 : 32>(p, acc, c_m0 + wq * 32, c_n0 + wn * 64, lane, epi_stage, rsc);
    }
    if (!have_next) break;
    __syncthreads();                                      // s_flag and the staging rows are free again
  }
}
