// Tile order of the chained GEMM launch (gemm_chain.h): plain integer arithmetic, shared by the kernel and by a host-side
// test (tests/test_chain_sched.py compiles this header with g++ and checks that every queue lists each of its tiles exactly
// once, producers ahead of the consumers that read them).
//
// The launch runs TWO row-local GEMMs of an encoder layer back to back (hf: models/t5/modeling_t5.py:304,367 -> 106-123: the
// attention output projection + residual, then the FFN input projections; or the FFN output projection + residual, then the
// next layer's q/k/v projections): the producer P[M, NP tiles] writes the residual stream, the consumer C[M, NC tiles] reads
// row panel m of it once all NP producer tiles of panel m are done.  Row panels (256 rows) are dealt to 8 QUEUES in
// contiguous ranges - one per XCD, so the tiles that share an A / W panel meet in one L2; a queue is self-contained (the
// producers of a panel sit in the queue of its consumers, earlier) - and a queue's tiles are listed as a MERGE of its
// producer stream (panel-major) with its consumer stream (blocks of CHAIN_R panels, column panels of CHAIN_G tiles inside a
// block: co-resident tiles form compact (rows x 8 columns) pieces of the output like gemm_tile_coords' order), the producer
// stream running `lead` tiles ahead and the two interleaved evenly (Bresenham) - so that at any time about NP / (NP + NC) of
// the CUs are in the HBM-bound read-modify-write epilogue of a producer tile while the others are in main loops.
#pragma once

#if defined(__HIPCC__)
#define RK_HD __host__ __device__ __forceinline__
#else
#define RK_HD inline
#endif

#define CHAIN_R 4          // row panels per block of the consumer stream
#define CHAIN_G 8          // column-panel width (tiles) of the consumer stream
#define CHAIN_QUEUES 8

// tile code: bit 30 = producer tile, bits 10..29 = row panel, bits 0..9 = column tile;  -1 = none
RK_HD int chain_code(bool prod, int tm, int tn) { return (prod ? (1 << 30) : 0) | (tm << 10) | tn; }
RK_HD bool chain_is_prod(int code) { return ((code >> 30) & 1) != 0; }
RK_HD int chain_tm(int code) { return (code >> 10) & 0xFFFFF; }
RK_HD int chain_tn(int code) { return code & 1023; }

// row panels [p0, p0 + np) of queue q
RK_HD void chain_queue(int q, int tiles_m, int& p0, int& np) {
  const int base = tiles_m / CHAIN_QUEUES, rem = tiles_m % CHAIN_QUEUES;
  np = base + (q < rem ? 1 : 0);
  p0 = q * base + (q < rem ? q : rem);
}
RK_HD int chain_queue_len(int q, int tiles_m, int NP, int NC) {
  int p0, np;
  chain_queue(q, tiles_m, p0, np);
  return np * (NP + NC);
}

// position u (0 <= u < chain_queue_len) of queue q -> tile code.  lead_blocks: how many blocks of CHAIN_R panels the producer
// stream is ahead of the consumer stream (>= 1; 2-3 keeps a consumer's inputs a few tile times old when it starts).
RK_HD int chain_decode(int q, int u, int tiles_m, int NP, int NC, int lead_blocks) {
  int p0, np;
  chain_queue(q, tiles_m, p0, np);
  const int PO = np * NP;
  int lead = lead_blocks * CHAIN_R * NP;
  lead = lead < PO ? lead : PO;
  int o = -1, f = -1;
  if (u < lead) {
    o = u;
  } else {
    const int v = u - lead, S = NP + NC, rest = PO - lead;      // rest: producer tiles inside the merged region
    const int g0 = (v * NP) / S, g1 = ((v + 1) * NP) / S;       // producer tiles among the merged positions [0, v) / [0, v]
    if (g1 > g0 && g0 < rest) o = lead + g0;
    else f = v - (g0 < rest ? g0 : rest);
  }
  if (o >= 0) {
    const int r = o / NP;
    return chain_code(true, p0 + r, o - r * NP);
  }
  const int per = CHAIN_R * NC;
  const int rb = f / per, g2 = f - rb * per;
  int rows = np - rb * CHAIN_R;
  rows = rows < CHAIN_R ? rows : CHAIN_R;
  const int ncp = (NC + CHAIN_G - 1) / CHAIN_G;
  int cp = g2 / (rows * CHAIN_G);
  cp = cp < ncp - 1 ? cp : ncp - 1;
  const int rem = g2 - cp * rows * CHAIN_G;
  int w = NC - cp * CHAIN_G;
  w = w < CHAIN_G ? w : CHAIN_G;
  const int r = rem / w;
  return chain_code(false, p0 + rb * CHAIN_R + r, cp * CHAIN_G + rem - r * w);
}
