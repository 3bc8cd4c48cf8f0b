// librk_engine.so — host side of the MI355X reranking engine: C ABI (include/rk_engine.h), weight repacking,
// workspace management and the launch sequence of the T5 encoder-decoder forward.
//
// What it replaces in the reference: T5ForConditionalGeneration.from_pretrained + .forward + .generate as
// called from llmrankers/pointwise.py:20-24,73-75,117-119 and llmrankers/setwise.py:46-59,93-95,184.
// The arithmetic restated here lives in hf: transformers/models/t5/modeling_t5.py (cited per kernel).
//
// gfx950 only; no CPU fallback, no CUDA/HIP dual paths.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is dlopen'ed on first use (rk_comm_*), never linked

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/rk_engine.h"
#include "attention.h"
#include "decoder_kernels.h"
#include "gemm.h"
#include "gemv_rows.h"
#include "llama_kernels.h"
#include "misc_kernels.h"

namespace {

thread_local std::string g_create_error;

enum ProfClass {
  PC_ENC_GEMM_QKV = 0, PC_ENC_GEMM_O, PC_ENC_GEMM_FFN_IN, PC_ENC_GEMM_FFN_OUT, PC_ENC_ATTN, PC_GEMM_CROSS_KV,
  PC_NORM, PC_EMBED, PC_DEC_GEMM, PC_DEC_ATTN, PC_HEAD, PC_OTHER,
  PC_COUNT
};
const char* kProfNames[PC_COUNT] = {"enc_gemm_qkv", "enc_gemm_o", "enc_gemm_ffn_in", "enc_gemm_ffn_out", "enc_attn",
                                    "gemm_cross_kv", "norm", "embed", "dec_gemm", "dec_attn", "head", "other"};

struct HostTensor {
  std::vector<half_t> h;   // 2-D matrices (fp16, the reference's accelerator dtype)
  std::vector<float> f;    // 1-D norm weights and the relative-attention tables (kept fp32)
  std::vector<int64_t> shape;
};

struct EncLayerW {
  half_t *qkv = nullptr, *o = nullptr, *ffn_in = nullptr, *ffn_out = nullptr; float *ln0 = nullptr, *ln1 = nullptr;
  half_t *qkv_f = nullptr, *ffn_in_f = nullptr;   // the same with the RMSNorm weight folded into the columns (W[n][k] * ln[k])
};
struct DecLayerW {
  half_t *qkv = nullptr, *o = nullptr, *cq = nullptr, *co = nullptr, *ffn_in = nullptr, *ffn_out = nullptr;
  half_t* ckT = nullptr;    // cross-attention W_k regrouped per head and transposed: [H][d_model][64] (direct path)
  half_t* ov = nullptr;     // self-attention W_o W_v [d_model, d_model]: the whole sub-layer at L_d = 1
  float *ln0 = nullptr, *ln1 = nullptr, *ln2 = nullptr;
  half_t *qkv_f = nullptr, *ov_f = nullptr, *cq_f = nullptr, *ffn_in_f = nullptr;   // norm weight folded in (W[n][k] * ln[k]), as in the encoder
};

// Llama-family decoder layer (hf: modeling_llama.py:291-330): fused q|k|v and interleaved gate|up carry the RMSNorm weights
struct LlamaLayerW { half_t *qkv_f = nullptr, *o = nullptr, *gu_f = nullptr, *down = nullptr; };

struct ProfRec { hipEvent_t a, b; int cls; };

}  // namespace

#define RK_SLOTS 2

// Everything that belongs to ONE batch in flight: own activation workspace and own pair of HIP streams.  Two slots
// let (a) the latency-bound decoder chain of one batch hide under the MFMA-bound encoder of the next and (b) the
// encoder kernels of both batches co-run, so workgroups of one fill the tile-quantisation tail of the other
// (1104 GEMM tiles on 512 resident slots is 3 rounds alone but 2.16 rounds of work).
struct Slot {
  hipStream_t se = nullptr, sd = nullptr;   // this slot's encoder chain (MFMA-bound) | decoder chain (latency-bound)
  float* hidden = nullptr; half_t *xn = nullptr, *qkv = nullptr, *ctx = nullptr, *ffh = nullptr, *enc_out = nullptr;
  half_t* xraw = nullptr; float *ssq = nullptr, *rowscale = nullptr;   // folded RMSNorm: fp16 stream x RK_XRAW_SCALE, block sums of squares, row factors
  int* d_tokens = nullptr; int* d_seq_off = nullptr;
  int n_seq = 0, T = 0, maxL = 0, minL = 0; bool staged = false; int last_n_out = 0;
  half_t* cross_kv = nullptr;                                  // [n_dec][max_tokens][2I] encoder -> decoder hand-off
  int *d_dec_ids = nullptr, *d_last_rows = nullptr, *d_out_ids = nullptr, *d_labels = nullptr, *d_argmax = nullptr, *d_row_seq = nullptr, *d_tree_keys = nullptr, *d_tree_pos = nullptr;
  float* dhidden = nullptr; half_t *dxn = nullptr, *dqkv = nullptr, *dctx = nullptr, *dq = nullptr, *dffh = nullptr, *dlast = nullptr;
  half_t* dxraw[2] = {nullptr, nullptr}; float* dssq[2] = {nullptr, nullptr}; float* drowscale = nullptr;   // folded decoder norms (run_decoder)
  float* dssq_few[2] = {nullptr, nullptr};   // the same for the few-row GEMV family (gemv_rows.h): one partial per producing workgroup
  half_t *xqk = nullptr, *xctx = nullptr;                      // direct cross-attention: [32][H*d] each
  float *xpart = nullptr, *xstat = nullptr; bool have_cross_kv = false;
  float* d_scores = nullptr; float* h_scores = nullptr;
  int* h_small = nullptr;                                      // pinned staging for small int uploads
  std::vector<int> cache_dec, cache_out, cache_rows;
  std::vector<int> cache_g2; unsigned long dec_epoch = 0, g2_epoch = ~0ul;   // rk_t5_greedy2's five index arrays, valid while no other path wrote the decoder id / row buffers (dec_epoch)
  hipEvent_t ev_enc = nullptr, ev_dec = nullptr; bool dec_pending = false;
};

struct rk_engine {
  rk_model_desc d{};
  int dev = 0;
  std::string err;
  bool finalized = false;
  int inner = 0;
  std::map<std::string, HostTensor> host;
  std::vector<void*> allocs;
  // weights
  half_t *emb = nullptr, *lm_head = nullptr, *cross_kv_w = nullptr;
  std::vector<EncLayerW> enc;
  std::vector<DecLayerW> dec;
  float *enc_final_ln = nullptr, *dec_final_ln = nullptr, *lut_enc = nullptr, *lut_dec = nullptr;
  float* logits = nullptr; size_t logits_cap = 0;              // qlm head: per-block (max, sum exp) pairs [rows, vocab/32] + label logits [rows]; slot 0 only
  const int* lse_labels = nullptr; int lse_npos = 0; float* lse_xlab = nullptr;   // arguments of the next EPI_LSE_F32 launch
  float* amax_val = nullptr; int* amax_idx = nullptr; size_t amax_rows = 0;   // greedy head: per-row block maxima / first columns
  size_t scores_cap = 0;
  Slot slots[RK_SLOTS];
  // options / measurement
  // Engine options (rk_engine_set_option; table kOptions below: key, range, meaning).  Every option selects between TESTED
  // implementations of the same arithmetic - the on-device cross-check of a default path (tests/test_gpu_kernels.py compares them
  // bit for bit or within the stated tolerance) - or is a measurement knob of tools/; none is an unfinished experiment.
  struct Options {
    int glds = 1, skinny = 0x3F, overlap = 1, gemm_variant = 0, attn_short = 5, xattn_direct = 1, attn_heads_per_wg = 0, attn_ko = 0,
        gemm_persistent = 1, fold_norm = 1, s64_stages = 0, dec_fold_norm = 1, greedy_spec = 160, consumer_stats = 1, xattn_mfma = 1,
        dec_ffn_tiled = 1, gemm_split = 1, dec_fuse = 1, dec_fuse_rows = 0, dec_attn_seq = 1, attn_long = 1, attn_long_nw = 0,
        llama_attn_dma = 1, attn_long_xcd = 1, llama_attn_nw = 0, dec_graph = 1, gemm_sk = 1, dec_cross_mfma = 1, dec_gemv = 1, dec_gemv_rows = 4;
  } opt;
  float* attn_trace = nullptr;   // measurement builds only (option attn_trace)
  int n_cu = 256;
  // K-split ping-pong GEMM (gemm.h: SPLIT): partial-tile slabs and arrival tickets, one set per stream (launches on different
  // streams overlap)
  struct SkWs { hipStream_t st = nullptr; float* slabs = nullptr; int* cnt = nullptr; };
  SkWs sk_ws[2 * RK_SLOTS];
  hipEvent_t t0 = nullptr, t1 = nullptr, t_tmp = nullptr;
  bool prof_on = false;
  std::vector<ProfRec> prof_recs; size_t prof_used = 0;
  double prof_flops[PC_COUNT] = {0}, prof_bytes[PC_COUNT] = {0}; int64_t prof_n[PC_COUNT] = {0};
  // decoder-only family (rk_llama_*): family 1 reuses `d` for the shared fields (vocab, d_model = hidden, d_ff =
  // intermediate, eps, capacities) so that the helpers below serve both families
  int family = 0; rk_llama_desc ld{};
  float rope_factor = 0.f, rope_low = 1.f, rope_high = 4.f; int rope_orig = 0;   // rope type llama3 when rope_factor > 0
  std::vector<LlamaLayerW> ll; float *l_final_ln = nullptr, *rope_cos = nullptr, *rope_sin = nullptr; int* d_pos = nullptr;
  // decoder chains as HIP graphs: key = everything the launch parameters of a chain depend on
  struct GraphEntry { int seen = 0; bool failed = false; hipGraphExec_t exec = nullptr; };
  std::map<std::vector<int>, GraphEntry> graphs; int opt_epoch = 0;
  // score collection across GPUs (K9): one RCCL communicator per engine = per process = per GPU
  ncclComm_t comm = nullptr; int comm_rank = 0, comm_world = 1;
  float* d_gather[RK_SLOTS] = {nullptr}; float* h_gather[RK_SLOTS] = {nullptr}; size_t gather_cap = 0;
  hipEvent_t ev_gather[RK_SLOTS] = {nullptr}; bool gather_pending[RK_SLOTS] = {false}; int gather_n[RK_SLOTS] = {0};
  // appended form (a rank's share scored in several engine calls): send buffer [gather_cap], result [world][gather_cap]
  float *d_gsend = nullptr, *d_gall = nullptr, *h_gall = nullptr, *h_gstage = nullptr;
  hipEvent_t ev_gall = nullptr, ev_append = nullptr; bool gall_pending = false, append_foreign = false; int gall_n = 0;
};

namespace {

int fail(rk_engine* e, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf; else g_create_error = buf;
  return code;
}

#define HIPCHK(E, call)                                                                              \
  do {                                                                                               \
    hipError_t _s = (call);                                                                          \
    if (_s != hipSuccess) return fail((E), RK_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_s)); \
  } while (0)

template <class T>
int dalloc(rk_engine* e, T** p, size_t n) {
  void* q = nullptr;
  HIPCHK(e, hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
  e->allocs.push_back(q);
  *p = (T*)q;
  return RK_OK;
}

template <class T>
int upload(rk_engine* e, T** p, const T* src, size_t n) {
  int rc = dalloc(e, p, n);
  if (rc) return rc;
  HIPCHK(e, hipMemcpy(*p, src, n * sizeof(T), hipMemcpyHostToDevice));
  return RK_OK;
}

// ---- profiling-aware launch bracket ---------------------------------------------------------------------
struct Bracket {
  rk_engine* e; hipStream_t st; int idx = -1;
  Bracket(rk_engine* e_, hipStream_t st_, int cls, double flops, double bytes) : e(e_), st(st_) {
    if (!e->prof_on) return;
    if (e->prof_used == e->prof_recs.size()) {
      ProfRec r{};
      if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
      e->prof_recs.push_back(r);
    }
    idx = (int)e->prof_used++;
    e->prof_recs[idx].cls = cls;
    e->prof_flops[cls] += flops; e->prof_bytes[cls] += bytes; e->prof_n[cls]++;
    hipEventRecord(e->prof_recs[idx].a, st);
  }
  ~Bracket() { if (idx >= 0) hipEventRecord(e->prof_recs[idx].b, st); }
};

// overlap = 0 puts every launch of every slot on ONE stream (serial timeline, used for per-kernel event timing)
// encoders alternate between TWO streams however many slots there are (more concurrent GEMM chains only thrash);
// the extra slots exist to lengthen the distance between a decoder and the next encoder that reuses its buffers
hipStream_t enc_stream(rk_engine* e, Slot& sl) { return e->opt.overlap ? e->slots[(&sl - e->slots) & 1].se : e->slots[0].se; }
hipStream_t dec_stream(rk_engine* e, Slot& sl) { return e->opt.overlap ? sl.sd : e->slots[0].se; }

// ---- kernel launch helpers ------------------------------------------------------------------------------
// More than 64 KiB of dynamic LDS needs hipFuncSetAttribute, which applies per DEVICE: done once per (kernel, device),
// whichever engine / thread launches it there first (engines on different GPUs may live in one process).
inline void ensure_dynamic_lds(const void* fn, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (rc != hipSuccess) fprintf(stderr, "[rk_engine] hipFuncSetAttribute(%d B LDS) failed: %s\n", bytes, hipGetErrorString(rc));
  done.fetch_or(bit, std::memory_order_release);
}

template <int EPI, int WM, int WN, int MI, int NI>
void launch_v2(hipStream_t st, const GemmArgs& a) {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  constexpr int smem_stages = 2 * (BM + BN) * 64 * 2, smem_epi = WM * WN * 32 * (NI * 32 * 4 + 16);
  constexpr int smem = smem_stages > smem_epi ? smem_stages : smem_epi;
  static std::atomic<uint64_t> attr_done{0};
  ensure_dynamic_lds((const void*)gemm_v2_kernel<EPI, WM, WN, MI, NI>, smem, attr_done);
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm_v2_kernel<EPI, WM, WN, MI, NI>), dim3(tiles), dim3(WM * WN * 64), smem, st, a);
}

#define KSPLIT_MAX_SLABS 256      // partial tiles (256 x 256 fp32 = 256 KiB each) per stream: 64 MiB

template <int EPI, int KO = 0, bool RS = false, bool SPLIT = false>
void launch_pp2(hipStream_t st, const GemmArgs& a, int max_wgs) {
  constexpr int smem = 2 * 4 * 128 * 64 * 2 + 32768;   // 8 half-tile buffers + 32 KiB epilogue staging = all 160 KiB
  static std::atomic<uint64_t> attr_done{0};
  ensure_dynamic_lds((const void*)gemm_pp2_kernel<EPI, KO, RS, SPLIT>, smem, attr_done);
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256) * (SPLIT ? a.ksplit : 1);
  // persistent: one workgroup per CU walks the tiles (max_wgs = CUs rounded down to a multiple of 8 keeps the tile -> XCD
  // association); max_wgs <= 0: one workgroup per tile
  // (measured, r03: trimming the grid to the fewest workgroups with the same number of rounds - 232 instead of 256 for the
  // 920 tiles of O / FFN-out, to leave 24 CUs to the decoder stream for the whole GEMM - changes nothing: 7350-7370 against
  // 7367-7376 passages/s; the decoder kernels are not waiting for CUs, they share the memory system.  Likewise starting the
  // workgroups that walk one tile less 12 / 20 / 28 us late, so that their read-modify-write epilogues fall into the others'
  // MFMA phases at no cost to the critical path: o 0.496 -> 0.484 ms per step in the serial profile, 7302-7340 against
  // 7338-7376 passages/s in the pipeline - the epilogue's cost is not a shared-HBM burst that de-phasing would spread)
  const int grid = max_wgs > 0 && tiles > max_wgs ? max_wgs : tiles;
  hipLaunchKernelGGL((gemm_pp2_kernel<EPI, KO, RS, SPLIT>), dim3(grid), dim3(512), smem, st, a);
}

// Tile-shape choice.  variant: 0 = auto, 1 = 128x128 (v1, two workgroups per CU), 2 = 256x256, 3 = 256x192,
// 4 = 256x128 (v2 kernels, one workgroup per CU), 5 = 256x256 ping-pong (v3).  auto = cheapest under a measured model:
// time ~ rounds over the resident slots x the variant's time for one round of K = 1024 (us, MI355X, tools/gemm_bench.py
// at M = 736 .. 23552, profiles/r01c_gemm_bench.txt, r01e_gemm_pingpong.txt).  All variants sum K in the same order, so
// the choice never changes a result bit.  GEGLU pairs gate/up inside 64-row wave tiles: no 192-wide tile for it.
// K split of the ping-pong kernel (gemm.h: SPLIT): only the fp32 residual projections (O / FFN-out, Llama o / down), only TWO ways,
// only when the launch has at most half as many 256 x 256 tiles as the chip has CUs and K >= 6 144 (96 K tiles).  Measured
// (profiles/r06_gemm_ksplit.txt, M = 1 536, N = 4 096): K = 14 336 281.6 -> 212.0 us, K = 8 192 155.9 -> 123.6 us, K = 4 096
// 76.9 -> 75.1 us (not worth a different rounding); three / four ways lose (every slab is published and re-read).  A function of
// (M, N, K) alone.  flan-t5-large / -xl never qualify (K <= 5 120).
const rk_engine::SkWs* ksplit_workspace(const rk_engine* e, hipStream_t st) {
  for (const auto& w : e->sk_ws) if (w.st == st && w.slabs) return &w;
  return nullptr;
}
int choose_ksplit(const rk_engine* e, int epi, int M, int N, int K) {
  if (!e->opt.gemm_sk || !(epi == EPI_RESID_F32 || epi == EPI_STORE_F32) || K % 64) return 1;
  const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
  const int wgs = e->n_cu & ~7, nk = K / 64;
  if (e->opt.gemm_sk == 2)                                             // tests / measurement: two ways wherever they fit
    return (nk >= 4 && tiles * 2 <= KSPLIT_MAX_SLABS) ? 2 : 1;
  if (epi != EPI_RESID_F32 || tiles < 1 || tiles * 2 > wgs || nk < 96) return 1;
  return 2;
}

int choose_variant(const rk_engine* e, int epi, int M, int N, int K, bool fold_producer = false, double* cost_out = nullptr) {
  if (cost_out) *cost_out = 0;
  if (e->opt.gemm_variant) return (e->opt.gemm_variant == 3 && (EPI_IS_GATED(epi) || fold_producer)) ? 2 : e->opt.gemm_variant;
  struct V { int id, bm, bn, slots; double round_us; };
  static const V vs[5] = {{5, 256, 256, 256, 25.5}, {2, 256, 256, 256, 29.8}, {3, 256, 192, 256, 24.7}, {4, 256, 128, 256, 19.0}, {1, 128, 128, 512, 17.3}};
  // 64x64 tiles (variant 6) win only while the larger tiles leave most of the chip idle (tools/gemm_bench.py, r02: O / FFN-out
  // of one or two setwise prompts, M = 1450 / 2900: 12.7 / 15.0 us against 15.7 / 17.4 us on 128x128 tiles; from M = 5888 on,
  // or for the wide QKV / FFN-in outputs, they lose): at most 192 tiles of 128x128
  if ((long)((M + 127) / 128) * ((N + 127) / 128) <= 192 && K >= 64) { if (cost_out) *cost_out = 15.0; return 6; }
  double best = 1e30; int bv = 1;
  for (const V& v : vs) {
    if (v.id == 3 && (EPI_IS_GATED(epi) || fold_producer)) continue;   // (the folded-norm producer needs 64-column wave tiles)
    if (v.id == 5 && K < 128) continue;
    const long tiles = (long)((M + v.bm - 1) / v.bm) * ((N + v.bn - 1) / v.bn);
    double cost = (double)((tiles + v.slots - 1) / v.slots) * v.round_us;
    if (v.id == 5 && e->opt.gemm_sk == 1 && choose_ksplit(e, epi, M, N, K) > 1) cost = 0.0;   // the K-split launch wins wherever it is eligible (measured)
    if (cost < best - 1e-9) { best = cost; bv = v.id; }
  }
  if (cost_out) *cost_out = best;
  return bv;
}

// Launch plan of one tiled GEMM: rows [0, m_pp2) on the persistent ping-pong kernel in WHOLE rounds over the CUs, the rest
// (m_pp2 = 0: everything) on `variant`.  The ping-pong kernel pays a full round for a partial one: the 100 passages of one
// query (M = 18 400: 72 row panels) are 288 tiles of the O / FFN-out projections = 1.1 rounds paid as 2, 864 of QKV = 3.4 as
// 4.  All tile variants produce the same bits (K order, epilogue statistics: tests), so the rows beyond the last whole round
// go to the cheapest fill-in variant as a second launch - same model as choose_variant.  The grouped bench launches (M = 58 880)
// keep one launch: their last round is 60-98 % full and the model says so.
struct GemmPlan { int m_pp2; int variant; };
GemmPlan choose_plan(const rk_engine* e, int epi, int M, int N, int K, bool fold_producer) {
  double whole = 0;
  GemmPlan plan{0, choose_variant(e, epi, M, N, K, fold_producer, &whole)};
  if (!e->opt.gemm_split || e->opt.gemm_variant || K < 128 || e->opt.gemm_persistent != 1) return plan;
  if (!(epi == EPI_STORE_F16 || epi == EPI_RESID_F32 || EPI_IS_GATED(epi) || epi == EPI_RELU_F16)) return plan;   // (the heads index rows from 0)
  const int wgs = e->n_cu & ~7, tiles_n = (N + 255) / 256, tiles_m = (M + 255) / 256;
  const long rounds = (long)tiles_m * tiles_n / wgs;
  if (rounds < 1 || (long)tiles_m * tiles_n % wgs == 0) return plan;
  const int panels = (int)(rounds * wgs / tiles_n);          // whole row panels inside the whole rounds
  if (panels < 1 || panels >= tiles_m) return plan;
  const long used = (long)panels * tiles_n;
  double rest = 0;
  const int v_rest = choose_variant(e, epi, M - panels * 256, N, K, fold_producer, &rest);
  const double split = (double)((used + wgs - 1) / wgs) * 25.5 + rest + 1.5;   // + a kernel boundary
  if (split < whole - 1e-9) { plan.m_pp2 = panels * 256; plan.variant = v_rest; }
  return plan;
}

// does a folded-norm consumer GEMM of this shape run the persistent ping-pong kernel (row factors from rowscale_kernel)?
bool consumer_uses_pp2(const rk_engine* e, int epi, int M, int N, int K) {
  const GemmPlan pl = choose_plan(e, epi, M, N, K, false);
  int v = pl.variant;
  if (v > 6) v = 5;
  return pl.m_pp2 > 0 || (v == 5 && K >= 128);
}

template <int EPI>
void launch_gemm_epi(rk_engine* e, hipStream_t st, const GemmArgs& a_in, int force_variant = 0) {
  GemmArgs a = a_in;
  int variant = force_variant;
  if (!variant) {
    const GemmPlan pl = choose_plan(e, EPI, a.M, a.N, a.K, a.xraw != nullptr);
    variant = pl.variant;
    if (pl.m_pp2 > 0) {
      // whole rounds on the ping-pong kernel, then the remaining rows on the fill-in variant (row-offset arguments)
      GemmArgs head = a;
      head.M = pl.m_pp2;
      launch_gemm_epi<EPI>(e, st, head, 5);
      const size_t r = (size_t)pl.m_pp2;
      constexpr size_t celt = (EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32) ? 4 : 2;
      a.A += r * a.lda;
      a.C = (char*)a.C + r * (size_t)a.ldc * celt;
      if (a.rowscale) a.rowscale += r;
      if (a.xraw) a.xraw += r * a.ldx;
      if (a.ssq) a.ssq += r * a.nb;
      if (a.ssq_in) a.ssq_in += r * a.nb_in;
      a.M -= pl.m_pp2;
    }
  }
#ifdef RK_MEASURE
  if constexpr (EPI == EPI_STORE_F16) {                              // timing-only knock-outs (gemm_variant 80 + mask)
    if (variant > 80 && variant <= 96 && a.K >= 128) {
      switch (variant - 80) {
        case 8: launch_pp2<EPI, 8>(st, a, (e->opt.gemm_persistent == 1 ? (e->n_cu & ~7) : (e->opt.gemm_persistent & ~7))); return;    // no W-panel DMA
        case 16: launch_pp2<EPI, 16>(st, a, (e->opt.gemm_persistent == 1 ? (e->n_cu & ~7) : (e->opt.gemm_persistent & ~7))); return;  // no A-panel DMA
        case 1: launch_pp2<EPI, 1>(st, a, (e->opt.gemm_persistent == 1 ? (e->n_cu & ~7) : (e->opt.gemm_persistent & ~7))); return;
        case 2: launch_pp2<EPI, 2>(st, a, (e->opt.gemm_persistent == 1 ? (e->n_cu & ~7) : (e->opt.gemm_persistent & ~7))); return;
        case 3: launch_pp2<EPI, 3>(st, a, (e->opt.gemm_persistent == 1 ? (e->n_cu & ~7) : (e->opt.gemm_persistent & ~7))); return;
        case 4: launch_pp2<EPI, 4>(st, a, (e->opt.gemm_persistent == 1 ? (e->n_cu & ~7) : (e->opt.gemm_persistent & ~7))); return;
        case 5: launch_pp2<EPI, 5>(st, a, (e->opt.gemm_persistent == 1 ? (e->n_cu & ~7) : (e->opt.gemm_persistent & ~7))); return;
        default: launch_pp2<EPI, 6>(st, a, (e->opt.gemm_persistent == 1 ? (e->n_cu & ~7) : (e->opt.gemm_persistent & ~7))); return;
      }
    }
  }
#endif
  if (variant == 6) {
    const int tiles = ((a.M + 63) / 64) * ((a.N + 63) / 64);
    // stages: as many as keep every tile resident at once (4 -> 2 workgroups per CU, 3 -> 3, 2 -> 4)
    int nst = e->opt.s64_stages;
    if (nst < 2 || nst > 4) nst = tiles <= 2 * e->n_cu ? 4 : (tiles <= 3 * e->n_cu ? 3 : 2);
    if (nst == 4) {
      static std::atomic<uint64_t> attr_done{0};
      ensure_dynamic_lds((const void*)gemm_s64_kernel<EPI, 4>, 65536, attr_done);
      hipLaunchKernelGGL((gemm_s64_kernel<EPI, 4>), dim3(tiles), dim3(128), 65536, st, a);
    } else if (nst == 3) {
      static std::atomic<uint64_t> attr_done{0};
      ensure_dynamic_lds((const void*)gemm_s64_kernel<EPI, 3>, 49152, attr_done);
      hipLaunchKernelGGL((gemm_s64_kernel<EPI, 3>), dim3(tiles), dim3(128), 49152, st, a);
    } else {
      static std::atomic<uint64_t> attr_done{0};
      ensure_dynamic_lds((const void*)gemm_s64_kernel<EPI, 2>, 32768, attr_done);
      hipLaunchKernelGGL((gemm_s64_kernel<EPI, 2>), dim3(tiles), dim3(128), 32768, st, a);
    }
    return;
  }
  if (variant > 6) variant = 5;
  if (variant == 5 && a.K < 128) variant = 2;                       // the ping-pong kernel needs two K tiles
  if (variant == 5) {
    const int wgs = e->opt.gemm_persistent == 1 ? (e->n_cu & ~7) : (e->opt.gemm_persistent & ~7);
    if constexpr (EPI == EPI_STORE_F16 || EPI_IS_GATED(EPI) || EPI == EPI_RELU_F16) {
      if (a.rowscale) { launch_pp2<EPI, 0, true>(st, a, wgs); return; }   // consumer side of the folded RMSNorm
    }
    if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32) {
      const int ks = e->opt.gemm_persistent == 1 ? choose_ksplit(e, EPI, a.M, a.N, a.K) : 1;
      const rk_engine::SkWs* w = ks > 1 ? ksplit_workspace(e, st) : nullptr;
      if (w) {
        GemmArgs b = a;
        b.ksplit = ks; b.ks_slabs = w->slabs; b.ks_cnt = w->cnt;
        launch_pp2<EPI, 0, false, true>(st, b, wgs);
        return;
      }
    }
    launch_pp2<EPI>(st, a, wgs);
    return;
  }
  if (variant == 2) { launch_v2<EPI, 2, 4, 4, 2>(st, a); return; }
  if constexpr (!EPI_IS_GATED(EPI)) { if (variant == 3) { launch_v2<EPI, 4, 2, 2, 3>(st, a); return; } }
  if (variant == 4) { launch_v2<EPI, 4, 2, 2, 2>(st, a); return; }
  // (a 16-wave 256x256 form, launch_v2<EPI, 4, 4, 2, 2>, measured 3-9 % slower than the 8-wave one: not instantiated)
  const int tiles = ((a.M + GEMM_BM - 1) / GEMM_BM) * ((a.N + GEMM_BN - 1) / GEMM_BN);
  if (e->opt.glds)
    hipLaunchKernelGGL((gemm_f16_kernel<EPI, true>), dim3(tiles), dim3(256), GEMM_LDS_BYTES, st, a);
  else
    hipLaunchKernelGGL((gemm_f16_kernel<EPI, false>), dim3(tiles), dim3(256), GEMM_LDS_BYTES, st, a);
}

// Folded RMSNorm hooks of one GEMM launch (GemmArgs): consumer side = rowscale, producer side = xraw + ssq.
#define RK_XRAW_SCALE 0.0625f   // the fp16 copy of the fp32 residual stream is stored x 2^-4: head-room for the outlier
                                // channels of real T5 checkpoints (fp16 max 65504 -> 1.0e6), exact (power of two)
struct GemmFold { const float* rowscale = nullptr; half_t* xraw = nullptr; float* ssq = nullptr; const float* ssq_in = nullptr; int nb_in = 0;
                  bool few = false; };   // few: the few-row GEMV family (gemv_rows.h; run_decoder decides from the pass's rows / positions)

GemmArgs make_gemm_args(const rk_engine* e, const half_t* A, int lda, const half_t* W, int ldw, void* C, int ldc, int M, int N, int K,
                        int n_split, long split_stride, float scale, long bsA, long bsW, long bsC, const GemmFold& fold) {
  GemmArgs a{A, W, C, lda, ldw, ldc, M, N, K, n_split, split_stride, scale, bsA, bsW, bsC};
  a.rowscale = fold.rowscale; a.xraw = fold.xraw; a.ssq = fold.ssq; a.ldx = N; a.nb = (N + 63) / 64; a.xs = RK_XRAW_SCALE;
  a.ssq_in = fold.ssq_in; a.nb_in = fold.nb_in ? fold.nb_in : (K + 63) / 64; a.eps_in = e->d.eps;       // (tiled producers: 64-column blocks)
  a.group_n = GEMM_GROUP_N;
  return a;
}

void gemm(rk_engine* e, hipStream_t st, int cls, int epi, const half_t* A, int lda, const half_t* W, int ldw, void* C,
          int ldc, int M, int N, int K, int n_split = 0, long split_stride = 0, float scale = 1.f,
          int batch = 1, long bsA = 0, long bsW = 0, long bsC = 0, bool weight_streaming = false, GemmFold fold = GemmFold()) {
  if (M <= 0) return;
  GemmArgs a = make_gemm_args(e, A, lda, W, ldw, C, ldc, M, N, K, n_split, split_stride, scale, bsA, bsW, bsC, fold);
  const double flops = 2.0 * M * (double)N * K * batch;
  const double out_elems = EPI_IS_GATED(epi) ? (double)M * N / 2 : (double)M * N;
  const double bytes = 2.0 * ((double)M * K + (double)N * K) +
                       out_elems * (epi == EPI_RESID_F32 ? 8.0 : (epi == EPI_STORE_F32 ? 4.0 : 2.0));
  Bracket br(e, st, cls, flops, bytes);
  // Few-row GEMV family (gemv_rows.h): the decoder pass of ONE setwise compare (run_decoder sets fold.few from the pass's row and
  // position counts): one wave per output column over all CUs instead of 32-column MFMA tiles.
  if (fold.few && weight_streaming && batch == 1 && n_split == 0 && M <= GEMV_MAX_ROWS && K % 8 == 0 && K <= 512 * GEMV_MAX_PIECES &&
      (epi == EPI_STORE_F16 || epi == EPI_RESID_F32 || epi == EPI_GEGLU_F16 || epi == EPI_RELU_F16 || epi == EPI_STORE_F32)) {
    const int n_out = EPI_IS_GATED(epi) ? N / 2 : N;
    const int grid = gemv_grid(n_out, e->n_cu);
    a.nb = gemv_grid(N, e->n_cu);                                   // producer: one partial sum of squares per workgroup
    const int mr = M <= 2 ? 2 : (M <= 4 ? 4 : (M <= 8 ? 8 : 16));
    const int smem = mr * K * 2 + (GEMV_MAX_ROWS + 4 * GEMV_MAX_ROWS) * 4;
    auto go = [&](auto kern) {
      static std::atomic<uint64_t> attr_done{0};
      if (smem > 65536) ensure_dynamic_lds((const void*)kern, 160 * 1024, attr_done);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, a);
    };
#define RK_GEMV_CASE(E)                                                                                                   \
    case E: if (mr == 2) go(gemv_rows_kernel<E, 2>); else if (mr == 4) go(gemv_rows_kernel<E, 4>);                        \
            else if (mr == 8) go(gemv_rows_kernel<E, 8>); else go(gemv_rows_kernel<E, 16>); break;
    switch (epi) {
      RK_GEMV_CASE(EPI_STORE_F16) RK_GEMV_CASE(EPI_RESID_F32) RK_GEMV_CASE(EPI_GEGLU_F16) RK_GEMV_CASE(EPI_STORE_F32)
      default: if (mr == 2) go(gemv_rows_kernel<EPI_RELU_F16, 2>); else if (mr == 4) go(gemv_rows_kernel<EPI_RELU_F16, 4>);
               else if (mr == 8) go(gemv_rows_kernel<EPI_RELU_F16, 8>); else go(gemv_rows_kernel<EPI_RELU_F16, 16>); break;
    }
#undef RK_GEMV_CASE
    return;
  }
  // Kernel family is chosen by the CALLER's regime, never by M: a row's result must not depend on how many other rows
  // share the launch (the split-K weight-streaming kernel and the tiled kernels sum K in different orders).
  if ((weight_streaming || batch > 1) && n_split == 0 && (((e->opt.skinny >> epi) & 1) || batch > 1 || epi == EPI_ARGMAX_F32)) {
    // (a form where one workgroup takes up to 8 row slabs - 8x fewer, fatter workgroups - was bit-identical but made
    // the step 6 % slower: what the decoder costs the concurrent encoder GEMMs is the serial LENGTH of its chain, every
    // kernel delaying some tile of the GEMM in flight, not its CU-time; so: many short workgroups)
    const dim3 b(SKINNY_THREADS);
    const unsigned gy = (unsigned)batch, gz = (unsigned)((M + 31) / 32);
    a.nb = (N + 31) / 32; a.nb_in = fold.nb_in ? fold.nb_in : (K + 31) / 32;   // this kernel's own producer blocks are 32 columns wide
    // (tried and dropped, round 3: a 1-D launch that runs all row slabs of a column block on ONE XCD, so that a weight row is
    // fetched into one L2 only - dec_gemm 0.337 vs 0.328 ms per step at 320 rows: the slabs are not bound by weight traffic)
    switch (epi) {
      case EPI_STORE_F16: hipLaunchKernelGGL((gemm_skinny_kernel<EPI_STORE_F16, 1>), dim3((N + 31) / 32, gy, gz), b, 0, st, a); break;
      case EPI_RESID_F32: hipLaunchKernelGGL((gemm_skinny_kernel<EPI_RESID_F32, 1>), dim3((N + 31) / 32, gy, gz), b, 0, st, a); break;
      case EPI_GEGLU_F16: hipLaunchKernelGGL((gemm_skinny_kernel<EPI_GEGLU_F16, 2>), dim3((N + 63) / 64, gy, gz), b, 0, st, a); break;
      case EPI_SWIGLU_F16: hipLaunchKernelGGL((gemm_skinny_kernel<EPI_SWIGLU_F16, 2>), dim3((N + 63) / 64, gy, gz), b, 0, st, a); break;
      case EPI_RELU_F16: hipLaunchKernelGGL((gemm_skinny_kernel<EPI_RELU_F16, 1>), dim3((N + 31) / 32, gy, gz), b, 0, st, a); break;
      case EPI_ARGMAX_F32: a.amax_idx = e->amax_idx; hipLaunchKernelGGL((gemm_skinny_kernel<EPI_ARGMAX_F32, 1>), dim3((N + 31) / 32, gy, gz), b, 0, st, a); break;
      default: hipLaunchKernelGGL((gemm_skinny_kernel<EPI_STORE_F32, 1>), dim3((N + 31) / 32, gy, gz), b, 0, st, a); break;
    }
    return;
  }
  switch (epi) {
    case EPI_STORE_F16: launch_gemm_epi<EPI_STORE_F16>(e, st, a); break;
    case EPI_RESID_F32: launch_gemm_epi<EPI_RESID_F32>(e, st, a); break;
    case EPI_GEGLU_F16: launch_gemm_epi<EPI_GEGLU_F16>(e, st, a); break;
    case EPI_SWIGLU_F16: launch_gemm_epi<EPI_SWIGLU_F16>(e, st, a); break;
    case EPI_RELU_F16: launch_gemm_epi<EPI_RELU_F16>(e, st, a); break;
    case EPI_LSE_F32: a.lse_labels = e->lse_labels; a.lse_npos = e->lse_npos; a.lse_xlab = e->lse_xlab; launch_gemm_epi<EPI_LSE_F32>(e, st, a); break;
    default: launch_gemm_epi<EPI_STORE_F32>(e, st, a); break;
  }
}

void rmsnorm(rk_engine* e, hipStream_t st, const float* x, const float* w, half_t* out, const int* row_map, int rows, float scale = 1.f) {
  if (rows <= 0) return;
  Bracket br(e, st, PC_NORM, 3.0 * rows * e->d.d_model, (double)rows * e->d.d_model * 6.0);
  const int dm = e->d.d_model;
  const dim3 g((rows + 3) / 4), b(256);
  if (dm <= 1024) hipLaunchKernelGGL(rmsnorm_kernel<4>, g, b, 0, st, x, w, out, row_map, rows, dm, e->d.eps, scale);
  else if (dm <= 2048) hipLaunchKernelGGL(rmsnorm_kernel<8>, g, b, 0, st, x, w, out, row_map, rows, dm, e->d.eps, scale);
  else hipLaunchKernelGGL(rmsnorm_kernel<16>, g, b, 0, st, x, w, out, row_map, rows, dm, e->d.eps, scale);
}

void embed(rk_engine* e, hipStream_t st, const int* ids, float* out, int rows, half_t* xraw = nullptr, float* rowscale = nullptr) {
  if (rows <= 0) return;
  Bracket br(e, st, PC_EMBED, 0, (double)rows * e->d.d_model * 6.0);
  hipLaunchKernelGGL(embed_gather_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, ids, e->emb, out, rows,
                     e->d.d_model, e->d.vocab, xraw, rowscale, RK_XRAW_SCALE, e->d.eps);
}

// folded RMSNorm: block sums of squares (left by the residual GEMM epilogue) -> row factors
void rowscale(rk_engine* e, hipStream_t st, const float* ssq, float* out, int rows, int nb = 0) {   // nb: block sums per row (0: 64-column blocks)
  if (rows <= 0) return;
  if (nb <= 0) nb = (e->d.d_model + 63) / 64;
  Bracket br(e, st, PC_NORM, 0, (double)rows * (nb + 1) * 4.0);
  hipLaunchKernelGGL(rowscale_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, ssq, out, rows, nb, e->d.d_model, e->d.eps, RK_XRAW_SCALE);
}

// ---- relative position bucket (hf: modeling_t5.py:216-262), float32 like torch ------------------------------
int rel_bucket(int rel, bool bidirectional, int num_buckets, int max_distance) {
  int ret = 0;
  if (bidirectional) {
    num_buckets /= 2;
    if (rel > 0) ret += num_buckets;
    rel = rel < 0 ? -rel : rel;
  } else {
    rel = rel < 0 ? -rel : 0;
  }
  const int max_exact = num_buckets / 2;
  if (rel < max_exact) return ret + rel;
  const float num = logf((float)rel / (float)max_exact);
  const float den = (float)std::log((double)max_distance / (double)max_exact);
  int large = max_exact + (int)(num / den * (float)(num_buckets - max_exact));
  if (large > num_buckets - 1) large = num_buckets - 1;
  return ret + large;
}

// ---- weight lookup -------------------------------------------------------------------------------------------
const HostTensor* need(rk_engine* e, const std::string& name, int64_t r, int64_t c, std::string* missing) {
  auto it = e->host.find(name);
  if (it == e->host.end()) { *missing += (missing->empty() ? "" : ", ") + name; return nullptr; }
  const HostTensor& t = it->second;
  const bool ok = (c < 0) ? (t.shape.size() == 1 && t.shape[0] == r) : (t.shape.size() == 2 && t.shape[0] == r && t.shape[1] == c);
  if (!ok) { *missing += (missing->empty() ? "" : ", ") + name + "(bad shape)"; return nullptr; }
  return &t;
}

int set_device(rk_engine* e) {
  HIPCHK(e, hipSetDevice(e->dev));
  return RK_OK;
}

// Upload a small int array through pinned memory unless it equals what is already on the device.
int upload_small(rk_engine* e, Slot& sl, hipStream_t st, std::vector<int>* cache, int* dptr, int pin_slot, const int* src, int n) {
  if ((int)cache->size() == n && (n == 0 || memcmp(cache->data(), src, n * sizeof(int)) == 0)) return RK_OK;
  HIPCHK(e, hipStreamSynchronize(st));   // the pinned slot may still be in flight
  ++sl.dec_epoch;
  if (n > 8192) {                        // larger than a pinned slot: plain synchronous copy
    HIPCHK(e, hipMemcpy(dptr, src, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    cache->assign(src, src + n);
    return RK_OK;
  }
  int* pin = sl.h_small + pin_slot * 8192;
  memcpy(pin, src, n * sizeof(int));
  HIPCHK(e, hipMemcpyAsync(dptr, pin, n * sizeof(int), hipMemcpyHostToDevice, st));
  cache->assign(src, src + n);
  return RK_OK;
}

// ---- forward passes -----------------------------------------------------------------------------------------
// hf: modeling_t5.py:663-750 (T5Stack.forward, encoder) over the slot's staged ragged batch, then the stacked
// cross-attention K/V projections of all decoder layers (:325-326 with key_value_states = encoder output).
#define XA_MAX_ROWS 512     // decoder rows (sequences x positions) per pass of the direct cross-attention path
#define XA_MAX_CHUNKS 4096  // rows x 64-key chunks of partial-sum workspace per pass
#define XA_MAX_LD 16        // decoder positions per sequence up to which the query-side form is used

// Query-side cross-attention (attention.h) or materialised K/V: the two round at different points, so the choice must
// not depend on what else shares the call (a row's logits have to be the same in any batch, on any rank): it is made
// from the decoder LENGTH of the call alone (the query-side form costs L_d x L x H x d flops per sequence against
// L x 2I x d for the projections: cheaper up to L_d ~ 64, and far fewer bytes below 16).  More rows than the workspace
// holds are taken in passes (run_decoder).
bool use_xattn_direct(const rk_engine* e, const Slot&, int max_ld) {
  return e->opt.xattn_direct && max_ld <= XA_MAX_LD;
}

int run_encoder(rk_engine* e, Slot& sl, bool need_cross_kv) {
  const rk_model_desc& d = e->d;
  hipStream_t st = enc_stream(e, sl);
  const int T = sl.T, I = e->inner, dm = d.d_model, F = d.d_ff;
  // Folded RMSNorm (default): the GEMMs that follow a norm read the un-normalised stream as fp16 (written by the
  // producer of the stream: embedding / residual epilogue), their weights carry the norm weight, and their epilogue
  // applies the row factor - the two norm kernels per layer (re-reading the fp32 stream) are gone.
  const bool fold = e->opt.fold_norm != 0;
  GemmFold cons, cons_ssq, prod;
  if (fold) { cons.rowscale = sl.rowscale; cons_ssq.ssq_in = sl.ssq; prod.xraw = sl.xraw; prod.ssq = sl.ssq; }
  // The persistent ping-pong GEMM takes its row factors ready-made (loaded under its last MFMAs): a rowscale_kernel runs in
  // front of it.  The fill-in tile variants of small launches (one setwise prompt) add the block sums themselves in their
  // epilogue (gemm_row_factors, same rk_row_factor -> same bits): two 5-us launches per layer less where launches are what costs.
  const bool qkv_pp2 = !e->opt.consumer_stats || consumer_uses_pp2(e, EPI_STORE_F16, T, 3 * I, dm);
  const bool ffn_pp2 = !e->opt.consumer_stats || consumer_uses_pp2(e, d.gated_gelu ? EPI_GEGLU_F16 : EPI_RELU_F16, T, d.gated_gelu ? 2 * F : F, dm);
  embed(e, st, sl.d_tokens, sl.hidden, T, fold ? sl.xraw : nullptr, fold ? sl.rowscale : nullptr);
  for (int l = 0; l < d.n_enc_layers; ++l) {
    const EncLayerW& w = e->enc[l];
    if (fold) {
      gemm(e, st, PC_ENC_GEMM_QKV, EPI_STORE_F16, sl.xraw, dm, w.qkv_f, dm, sl.qkv, 3 * I, T, 3 * I, dm, 0, 0, 1.f, 1, 0, 0, 0, false,
           (l == 0 || qkv_pp2) ? cons : cons_ssq);            // layer 0: the embedding kernel wrote the row factors
    } else {
      rmsnorm(e, st, sl.hidden, w.ln0, sl.xn, nullptr, T);
      gemm(e, st, PC_ENC_GEMM_QKV, EPI_STORE_F16, sl.xn, dm, w.qkv, dm, sl.qkv, 3 * I, T, 3 * I, dm);
    }
    {
      // Every sequence of the batch at most ATT_ROW_MAXL keys: the DMA kernel (attn_short = 5, the default: two six-wave groups
      // per 768-thread workgroup; 6: one group per workgroup); otherwise, or with attn_short = 0, the tiled kernel.  The two
      // compute a sequence bit-identically (attention.h: ATT_ROW_MAXL).
      AttnEncArgs a{sl.qkv, sl.ctx, sl.d_seq_off, e->lut_enc, 3 * I, I, I, 1, e->opt.attn_ko};
#ifdef RK_MEASURE
      a.trace = e->attn_trace;
#endif
      const double att_flops = 4.0 * (double)sl.maxL * T * I;   // exact for uniform lengths, upper bound if ragged
      Bracket br(e, st, PC_ENC_ATTN, att_flops, (double)T * 4 * I * 2.0);
      if (sl.maxL <= ATT_ROW_MAXL && e->opt.attn_short) {
        const int ng = e->opt.attn_short == 6 ? 1 : 2;
        // persistent launch: at most one workgroup per CU, the (sequence, head) items dealt out evenly in contiguous runs per
        // wave group (320 sequences x 16 heads on 256 CUs x 2 groups: 10 items each)
        const long total = (long)sl.n_seq * d.n_heads, groups = (long)e->n_cu * ng;
        const int per = e->opt.attn_heads_per_wg > 0 ? e->opt.attn_heads_per_wg : (int)((total + groups - 1) / groups);
        a.heads_per_wg = per;
        a.n_seq = sl.n_seq;
        const dim3 grid((unsigned)((total + (long)ng * per - 1) / ((long)ng * per)));
        if (ng == 2) {
          static std::atomic<uint64_t> attr_done{0};
          ensure_dynamic_lds((const void*)attn_enc_dma_kernel<2>, 2 * ATTD_LDS_BYTES, attr_done);
          hipLaunchKernelGGL(attn_enc_dma_kernel<2>, grid, dim3(768), 2 * ATTD_LDS_BYTES, st, a);
        } else {
          static std::atomic<uint64_t> attr_done{0};
          ensure_dynamic_lds((const void*)attn_enc_dma_kernel<1>, ATTD_LDS_BYTES, attr_done);
          hipLaunchKernelGGL(attn_enc_dma_kernel<1>, grid, dim3(384), ATTD_LDS_BYTES, st, a);
        }
      }
      else if (e->opt.attn_long) {
        // every sequence longer than ATT_ROW_MAXL keys: the chunked LDS-DMA kernel (round 5); the batch's short sequences (if any):
        // the tiled kernel, which reproduces the short kernel's bits - a sequence's result depends on ITS length only
        if (sl.minL <= ATT_ROW_MAXL) {
          a.skip_long = 1;
          hipLaunchKernelGGL(attn_enc_kernel, dim3((ATT_ROW_MAXL + 127) / 128, d.n_heads, sl.n_seq), dim3(256), 0, st, a);
        }
        // waves per workgroup (same bits for every choice): option attn_long_nw, or from the batch - enough workgroups for the chip
        int nw = e->opt.attn_long_nw;
        // (measured, one to eight 1 560-token prompts: 4 waves = 128 queries per workgroup, two workgroups per CU, wins everywhere -
        // 36.9 / 49.1 / 157.5 us per layer at 1 / 2 / 8 prompts against 60 / 60 / 180 at twelve waves, 48 / 85 / 212 at six (384-thread
        // workgroups of this register size run one per CU) and 39.6 / 67.6 / 224.9 for the tiled kernel; profiles/r05_attn_long.jsonl)
        if (nw != 12 && nw != 6 && nw != 4 && nw != 3) nw = 4;
        a.n_seq = sl.n_seq; a.n_heads = d.n_heads; a.nqb = (sl.maxL + 32 * nw - 1) / (32 * nw); a.xcd_map = e->opt.attn_long_xcd;
        const dim3 grid(xcd_grid(sl.n_seq * d.n_heads, a.nqb));
        static std::atomic<uint64_t> attr12{0}, attr6{0}, attr4{0}, attr3{0};
        if (nw == 12) { ensure_dynamic_lds((const void*)attn_enc_long_kernel<12>, ATTL_LDS_BYTES, attr12); hipLaunchKernelGGL(attn_enc_long_kernel<12>, grid, dim3(768), ATTL_LDS_BYTES, st, a); }
        else if (nw == 6) { ensure_dynamic_lds((const void*)attn_enc_long_kernel<6>, ATTL_LDS_BYTES, attr6); hipLaunchKernelGGL(attn_enc_long_kernel<6>, grid, dim3(384), ATTL_LDS_BYTES, st, a); }
        else if (nw == 4) { ensure_dynamic_lds((const void*)attn_enc_long_kernel<4>, ATTL_LDS_BYTES, attr4); hipLaunchKernelGGL(attn_enc_long_kernel<4>, grid, dim3(256), ATTL_LDS_BYTES, st, a); }
        else { ensure_dynamic_lds((const void*)attn_enc_long_kernel<3>, ATTL_LDS_BYTES, attr3); hipLaunchKernelGGL(attn_enc_long_kernel<3>, grid, dim3(192), ATTL_LDS_BYTES, st, a); }
      }
      else   // option attn_long = 0: the tiled kernel for every length (the on-device cross-check of the two DMA kernels)
        hipLaunchKernelGGL(attn_enc_kernel, dim3((sl.maxL + 127) / 128, d.n_heads, sl.n_seq), dim3(256), 0, st, a);
    }
    if (fold) {
      gemm(e, st, PC_ENC_GEMM_O, EPI_RESID_F32, sl.ctx, I, w.o, I, sl.hidden, dm, T, dm, I, 0, 0, 1.f, 1, 0, 0, 0, false, prod);
      if (ffn_pp2) rowscale(e, st, sl.ssq, sl.rowscale, T);
      if (d.gated_gelu)
        gemm(e, st, PC_ENC_GEMM_FFN_IN, EPI_GEGLU_F16, sl.xraw, dm, w.ffn_in_f, dm, sl.ffh, F, T, 2 * F, dm, 0, 0, 1.f, 1, 0, 0, 0, false, ffn_pp2 ? cons : cons_ssq);
      else
        gemm(e, st, PC_ENC_GEMM_FFN_IN, EPI_RELU_F16, sl.xraw, dm, w.ffn_in_f, dm, sl.ffh, F, T, F, dm, 0, 0, 1.f, 1, 0, 0, 0, false, ffn_pp2 ? cons : cons_ssq);
      const bool last = l + 1 == d.n_enc_layers;   // the final norm reads the fp32 stream itself
      gemm(e, st, PC_ENC_GEMM_FFN_OUT, EPI_RESID_F32, sl.ffh, F, w.ffn_out, F, sl.hidden, dm, T, dm, F, 0, 0, 1.f, 1, 0, 0, 0, false, last ? GemmFold() : prod);
      if (!last && qkv_pp2) rowscale(e, st, sl.ssq, sl.rowscale, T);
      continue;
    }
    gemm(e, st, PC_ENC_GEMM_O, EPI_RESID_F32, sl.ctx, I, w.o, I, sl.hidden, dm, T, dm, I);
    rmsnorm(e, st, sl.hidden, w.ln1, sl.xn, nullptr, T);
    if (d.gated_gelu)
      gemm(e, st, PC_ENC_GEMM_FFN_IN, EPI_GEGLU_F16, sl.xn, dm, w.ffn_in, dm, sl.ffh, F, T, 2 * F, dm);
    else
      gemm(e, st, PC_ENC_GEMM_FFN_IN, EPI_RELU_F16, sl.xn, dm, w.ffn_in, dm, sl.ffh, F, T, F, dm);
    gemm(e, st, PC_ENC_GEMM_FFN_OUT, EPI_RESID_F32, sl.ffh, F, w.ffn_out, F, sl.hidden, dm, T, dm, F);
  }
  rmsnorm(e, st, sl.hidden, e->enc_final_ln, sl.enc_out, nullptr, T);
  // the stacked K/V projections are only materialised when the decoder has too many rows for the query-side form
  if (need_cross_kv)
    gemm(e, st, PC_GEMM_CROSS_KV, EPI_STORE_F16, sl.enc_out, dm, e->cross_kv_w, dm, sl.cross_kv, 2 * I, T,
         d.n_dec_layers * 2 * I, dm, 2 * I, (long)d.max_tokens * 2 * I);
  sl.have_cross_kv = need_cross_kv;
  HIPCHK(e, hipGetLastError());
  return RK_OK;
}

// hf: modeling_t5.py:663-750 (decoder stack) for Ld teacher-forced positions per sequence (ids already on the
// device in sl.d_dec_ids, row = b*Ld + t).  Leaves the residual stream in sl.dhidden.
// tree (rk_t5_greedy2): the decoder rows are not Ld per sequence - several continuations of a prompt share the rows of their
// common prefix.  rows = row count, Ld = longest position count; device arrays: keys[r * Ld + j] = row at position j of
// row r's sequence, pos[r] = position of row r, seq[r] = its encoder sequence.  Query-side cross-attention only.
struct DecTree { int rows; const int* keys; const int* pos; const int* seq; };
// (Tried and dropped, round 3: the single-position pass of 320 rows as TWO or THREE chains of 32-row-aligned row ranges on
// helper streams, fork / join by events (parallel branches of the decoder graph) - bit-identical, but 6.4-6.6k passages/s
// against 7.3k: what the decoder costs the encoder running beside it is every one of its kernels delaying the persistent
// GEMM it meets, so more, smaller decoder kernels cost more, not less.  The lever is fewer and shorter decoder kernels.)
int run_decoder(rk_engine* e, Slot& sl, int Ld, const DecTree* tree = nullptr) {
  const rk_model_desc& d = e->d;
  hipStream_t st = dec_stream(e, sl);
  const int B = sl.n_seq, M = tree ? tree->rows : B * Ld, I = e->inner, dm = d.d_model, F = d.d_ff;
  if (tree && (sl.have_cross_kv || Ld < 2)) return fail(e, RK_ERR_STATE, "the tree form needs the query-side cross-attention and L_d >= 2");
  const bool ws = Ld <= 4;   // few decoder positions: weight-streaming GEMMs (any number of sequences); else tiled
  // Folded RMSNorm on the weight-streaming path (as in the encoder, minus the statistics kernel): the residual GEMMs leave
  // the new rows as fp16 (dxraw) with their sums of squares per 32-column block (dssq), the GEMM behind the norm reads
  // those with the norm weight folded into its matrix and forms the row factor itself (gemm.h: GemmArgs::ssq_in) - three
  // launches per layer less.  A producer never writes the buffer a workgroup of the same launch may still read: two of each.
  const bool dfold = ws && e->opt.dec_fold_norm && (e->opt.skinny & 0x3F) == 0x3F;
  int cur = 0; bool from_embed = true;
  // (the producers of dssq are weight-streaming GEMMs: 32-column blocks, whichever kernel family consumes them)
  // Few-row GEMV family (round 6, gemv_rows.h): the pass of ONE setwise / pairwise prompt - a handful of rows at two or more
  // positions ("<pad> Passage": 2 rows; the second greedy step: 3) - runs its plain projections one wave per output column over all
  // CUs.  Decided from the PASS (rows, positions), so every row of a pass takes one family; a row scored alone and the same row in
  // a lockstep call of more than eight prompts differ in the last bits.  The one-position pointwise decoder
  // (any row count) never takes it: its batch independence stays bit-exact.
  // Row limit (option dec_gemv_rows, default 4): measured cross-over at flan-t5-large dims and 1 450-token prompts, whole calls,
  // GEMV against weight-streaming MFMA family: 2 rows 4.87 / 5.20 ms, 4 rows 6.38 / 6.67, 6 rows 8.30 / 8.25, 8 rows 9.41 / 9.39,
  // 12 rows 12.86 / 12.36, 16 rows 15.18 / 14.58 (profiles/r06_few_rows_ab.txt) - every workgroup stages ALL rows in LDS and the
  // per-column VALU work grows with the rows; rk_t5_greedy2's 13-row tree pass stays on the matrix cores.
  const bool fuse_any = (e->opt.dec_fuse == 2 || (e->opt.dec_fuse == 1 && Ld == 1));
  const bool few = dfold && e->opt.dec_gemv && Ld >= 2 && M <= e->opt.dec_gemv_rows && !fuse_any && dm <= 512 * GEMV_MAX_PIECES && F <= 512 * GEMV_MAX_PIECES &&
                   dm % 8 == 0 && F % 8 == 0 && I % 8 == 0;
  const int nb_few = gemv_grid(dm, e->n_cu);
  auto cons = [&]() {
    GemmFold f; f.few = few;
    if (from_embed) f.rowscale = sl.drowscale;
    else if (few) { f.ssq_in = sl.dssq_few[cur]; f.nb_in = nb_few; }
    else { f.ssq_in = sl.dssq[cur]; f.nb_in = (dm + 31) / 32; }
    return f;
  };
  auto with_prod = [&](GemmFold f) { f.few = few; f.xraw = sl.dxraw[cur ^ 1]; f.ssq = few ? sl.dssq_few[cur ^ 1] : sl.dssq[cur ^ 1]; return f; };
  auto flip = [&]() { cur ^= 1; from_embed = false; };
  embed(e, st, sl.d_dec_ids, sl.dhidden, M, dfold ? sl.dxraw[0] : nullptr, dfold ? sl.drowscale : nullptr);
  const size_t smem_self = (64 + 256 + 8 + (size_t)Ld) * sizeof(float);
  const size_t smem_cross = (64 + 256 + 8 + (size_t)sl.maxL) * sizeof(float);
  for (int l = 0; l < d.n_dec_layers; ++l) {
    const DecLayerW& w = e->dec[l];
    if (!dfold) rmsnorm(e, st, sl.dhidden, w.ln0, sl.dxn, nullptr, M);
    if (Ld == 1) {
      // one decoder position: softmax over a single key is 1, so self-attention is exactly o(v(x)) — the q/k
      // projections, scores and bias are dead (hf: modeling_t5.py:448-509 at L_d = 1)
      // ... and o(v(x)) = (W_o W_v) x: one GEMM with the product matrix formed once at finalize
      if (dfold) {
        gemm(e, st, PC_DEC_GEMM, EPI_RESID_F32, sl.dxraw[cur], dm, w.ov_f, dm, sl.dhidden, dm, M, dm, dm, 0, 0, 1.f, 1, 0, 0, 0, ws, with_prod(cons()));
        flip();
      } else {
        gemm(e, st, PC_DEC_GEMM, EPI_RESID_F32, sl.dxn, dm, w.ov, dm, sl.dhidden, dm, M, dm, dm, 0, 0, 1.f, 1, 0, 0, 0, ws);
      }
    } else {
      if (dfold) gemm(e, st, PC_DEC_GEMM, EPI_STORE_F16, sl.dxraw[cur], dm, w.qkv_f, dm, sl.dqkv, 3 * I, M, 3 * I, dm, 0, 0, 1.f, 1, 0, 0, 0, ws, cons());
      else gemm(e, st, PC_DEC_GEMM, EPI_STORE_F16, sl.dxn, dm, w.qkv, dm, sl.dqkv, 3 * I, M, 3 * I, dm, 0, 0, 1.f, 1, 0, 0, 0, ws);
      AttnDecArgs a{sl.dqkv, 3 * I, sl.dqkv + I, sl.dqkv + 2 * I, 3 * I, nullptr, sl.dctx, I, e->lut_dec, Ld, 1, Ld};
      {
        Bracket br(e, st, PC_DEC_ATTN, 4.0 * M * Ld * I, 0);
        if (tree) {
          a.tree_keys = tree->keys; a.tree_pos = tree->pos;
          hipLaunchKernelGGL(attn_dec_kernel, dim3(1, d.n_heads, M), dim3(256), smem_self, st, a);
        } else if (e->opt.dec_cross_mfma && Ld > XA_MAX_LD && Ld <= ATTX_MAXQ) {
          // long prefixes (the materialised-K / V regime, qlm): causal self-attention on the matrix cores too (round 6); the choice
          // follows from the call's position count alone, like the cross-attention's
          hipLaunchKernelGGL(attn_dec_cross_mfma_kernel, dim3(d.n_heads, B), dim3(128), ATTX_LDS_BYTES, st, a);
        } else if (e->opt.dec_attn_seq && attn_dec_seq_lds(Ld) <= 160 * 1024) {   // one workgroup per (head, sequence): K / V staged once (same bits)
          hipLaunchKernelGGL(attn_dec_seq_kernel, dim3(d.n_heads, B), dim3(256), attn_dec_seq_lds(Ld), st, a);
        } else {
          hipLaunchKernelGGL(attn_dec_kernel, dim3(Ld, d.n_heads, B), dim3(256), smem_self, st, a);
        }
      }
      gemm(e, st, PC_DEC_GEMM, EPI_RESID_F32, sl.dctx, I, w.o, I, sl.dhidden, dm, M, dm, I, 0, 0, 1.f, 1, 0, 0, 0, ws, dfold ? with_prod(GemmFold()) : GemmFold());
      if (dfold) flip();
    }
    // Query-side cross-attention with the projections around it fused per (head, row slab) - decoder_kernels.h: the q
    // projection + W_k^T q in one launch, the chunk merge + W_v in another (dec_fuse = 1, the default): 3 launches instead of 5
    // Measured (r04): at 100-320 rows (pointwise, one decoder position) the fused pair is 9 us per layer faster and the grouped
    // pipeline gains 1.2 %; at the 13 rows x 23 chunks of a setwise compare it is 4 us per layer SLOWER (the separate GEMMs spread
    // the cold weights of a layer over 512 + 5120 workgroups).  The family follows from the CALL SHAPE, never from the batch
    // (the two round differently): fused for one decoder position, separate beyond (dec_fuse = 2 forces the fused form: tests)
    const bool fuse = (e->opt.dec_fuse == 2 || (e->opt.dec_fuse == 1 && Ld == 1)) && !sl.have_cross_kv && dm % 128 == 0;   // (eight K ranges of whole k16 steps per workgroup)
    if (!dfold) rmsnorm(e, st, sl.dhidden, w.ln1, sl.dxn, nullptr, M);
    if (!fuse) {
      if (dfold) gemm(e, st, PC_DEC_GEMM, EPI_STORE_F16, sl.dxraw[cur], dm, w.cq_f, dm, sl.dq, I, M, I, dm, 0, 0, 1.f, 1, 0, 0, 0, ws, cons());
      else gemm(e, st, PC_DEC_GEMM, EPI_STORE_F16, sl.dxn, dm, w.cq, dm, sl.dq, I, M, I, dm, 0, 0, 1.f, 1, 0, 0, 0, ws);
    }
    if (!sl.have_cross_kv) {
      // query-side cross-attention: qk = W_k^T q per head; scores/softmax/weighted sum over the raw encoder states;
      // ctx = W_v (.) per head  (attention.h: XAttnArgs)
      const int H = d.n_heads, nch = (sl.maxL + 63) / 64;
      const int blk = std::max(1, std::min(XA_MAX_ROWS, XA_MAX_CHUNKS / nch));
      const half_t* wv = e->cross_kv_w + ((size_t)l * 2 * I + I) * dm;
      for (int r0 = 0; r0 < M; r0 += blk) {
        const int nr = std::min(blk, M - r0);
        if (fuse) {
          const GemmFold cf = dfold ? cons() : GemmFold();
          DecQKArgs qa{(dfold ? sl.dxraw[cur] : sl.dxn) + (size_t)r0 * dm, dm, dfold ? w.cq_f : w.cq, w.ckT, sl.xqk, nr, dm, H,
                       cf.rowscale ? cf.rowscale + r0 : nullptr, cf.ssq_in ? cf.ssq_in + (size_t)r0 * cf.nb_in : nullptr, cf.nb_in, d.eps, RK_XRAW_SCALE, 32, 1};
          if (e->opt.dec_fuse_rows > 0) qa.R = std::min(32, e->opt.dec_fuse_rows);
          else if (nr <= 16) qa.R = 16;   // (a setwise pass: 13 rows - half the MFMA columns, half the x rows; measured at 320 rows: 32 > 16 > 8)
          // few rows: several workgroups per (head, slab) share the output columns, each streaming 1 / CS of W_k^T (and all of W_q,h)
          while (qa.CS < 8 && (dm / 64) % (2 * qa.CS) == 0 && (long)((nr + qa.R - 1) / qa.R) * H * qa.CS < e->n_cu / 2) qa.CS *= 2;
          Bracket br(e, st, PC_DEC_GEMM, 2.0 * nr * (double)dm * I * 2, 2.0 * ((double)I * dm * 2 + (double)nr * H * dm));
          hipLaunchKernelGGL(dec_cross_qk_kernel, dim3(H, (nr + qa.R - 1) / qa.R, qa.CS), dim3(64 * DEC_NW), 0, st, qa);
        } else {
          gemm(e, st, PC_DEC_GEMM, EPI_STORE_F16, sl.dq + (size_t)r0 * I, I, w.ckT, 64, sl.xqk, H * dm, nr, dm, 64, 0, 0, 1.f, H, 64, (long)dm * 64, dm);
        }
        XAttnArgs xa{sl.xqk, sl.enc_out, sl.d_seq_off, sl.xpart, sl.xstat, sl.xctx, Ld, H, dm, nch, r0, tree ? tree->seq : nullptr};
        const bool fuse_cv = fuse && nch <= DECV_MAXCH;
        {
          Bracket br(e, st, PC_DEC_ATTN, 4.0 * nr * (double)sl.maxL * H * dm, (double)sl.T * dm * 2.0 * 2);
          // MFMA form (weighted sums on the matrix cores, the chunk's encoder rows staged in LDS by a loader wave) whenever
          // the model width allows its LDS image; the VALU form otherwise.  The choice depends on the MODEL only, never on the
          // batch (the two round differently).
          if (e->opt.xattn_mfma && dm % 256 == 0) {             // (every wave takes whole 64-column pieces of its quarter)
            // few workgroups (a setwise compare): the latency-scheduled form, two pieces per wave (same bits; attention.h)
            if ((long)nch * nr * ((H + 15) / 16) <= 2 * e->n_cu) hipLaunchKernelGGL(xattn_part_mfma_kernel<true>, dim3(nch, nr, (H + 15) / 16), dim3(256), 0, st, xa);
            else hipLaunchKernelGGL(xattn_part_mfma_kernel<false>, dim3(nch, nr, (H + 15) / 16), dim3(256), 0, st, xa);
          } else if ((long)nch * nr * ((H + 15) / 16) >= 2 * e->n_cu)
            hipLaunchKernelGGL(xattn_part_kernel<16>, dim3(nch, nr, (H + 15) / 16), dim3(256), 0, st, xa);
          else
            hipLaunchKernelGGL(xattn_part_kernel<4>, dim3(nch, nr, (H + 3) / 4), dim3(256), 0, st, xa);
          if (!fuse_cv) hipLaunchKernelGGL(xattn_combine_kernel, dim3(H, nr), dim3(256), 0, st, xa);
        }
        if (fuse_cv) {
          // rows per workgroup: the largest slab that still gives about half the chip a workgroup (results do not depend on it)
          // (16 rows: 41 KiB of LDS at d = 1024, three workgroups per CU hide each other's load latency)
          int R = 16;
          while (R > 2 && (long)((nr + R - 1) / R) * H < e->n_cu / 2) R >>= 1;
          DecCVArgs ca{sl.xpart, sl.xstat, sl.d_seq_off, tree ? tree->seq : nullptr, Ld, r0, wv, sl.dctx + (size_t)r0 * I, nr, dm, H, nch, I, R};
          const size_t lds = dec_cv_lds_bytes(dm, R);
          static std::atomic<uint64_t> attr_done{0};
          ensure_dynamic_lds((const void*)dec_cross_cv_kernel, 160 * 1024, attr_done);
          Bracket br(e, st, PC_DEC_GEMM, 2.0 * nr * (double)dm * I, 2.0 * (double)I * dm + 4.0 * (double)nr * nch * H * dm);
          hipLaunchKernelGGL(dec_cross_cv_kernel, dim3(H, (nr + R - 1) / R), dim3(64 * DEC_NW), lds, st, ca);
        } else {
          gemm(e, st, PC_DEC_GEMM, EPI_STORE_F16, sl.xctx, H * dm, wv, dm, sl.dctx + (size_t)r0 * I, I, nr, 64, dm, 0, 0, 1.f, H, dm, (long)64 * dm, 64);
        }
      }
    } else {
      const half_t* kv = sl.cross_kv + (size_t)l * d.max_tokens * 2 * I;
      AttnDecArgs a{sl.dq, I, kv, kv + I, 2 * I, sl.d_seq_off, sl.dctx, I, nullptr, Ld, 0, sl.maxL};
      Bracket br(e, st, PC_DEC_ATTN, 4.0 * Ld * (double)sl.T * I, (double)sl.T * 2 * I * 2.0);
      // sequences of at most ATTX_MAXK keys (every pointwise prompt): the matrix-core kernel; longer ones: the staged kernels.  The
      // call's position count decides whether the MFMA kernel runs at all, a sequence's own key count which kernel takes it.
      const bool mfma = e->opt.dec_cross_mfma && Ld >= 2 && Ld <= ATTX_MAXQ;
      if (mfma) hipLaunchKernelGGL(attn_dec_cross_mfma_kernel, dim3(d.n_heads, B), dim3(128), ATTX_LDS_BYTES, st, a);
      a.skip_short = mfma ? 1 : 0;
      if (mfma && sl.maxL <= ATTX_MAXK) {
        // nothing left for the staged kernels
      } else if (e->opt.dec_attn_seq && Ld >= 2 && attn_dec_seq_lds(sl.maxL) <= 160 * 1024)
        hipLaunchKernelGGL(attn_dec_seq_kernel, dim3(d.n_heads, B), dim3(256), attn_dec_seq_lds(sl.maxL), st, a);
      else
        hipLaunchKernelGGL(attn_dec_kernel, dim3(Ld, d.n_heads, B), dim3(256), smem_cross, st, a);
    }
    if (dfold) {
      gemm(e, st, PC_DEC_GEMM, EPI_RESID_F32, sl.dctx, I, w.co, I, sl.dhidden, dm, M, dm, I, 0, 0, 1.f, 1, 0, 0, 0, ws, with_prod(GemmFold()));
      flip();
      {
        // ONE decoder position (pointwise yes_no, MonoT5): FFN-in runs on the TILED kernels whatever the number of rows.  Its
        // 5632 output columns are 176 column blocks x (rows / 32) workgroups for the weight-streaming kernel - 1760 at the
        // bench's 320 rows, 29.9 us per layer - against 440 tiles of 64x64 on the matrix cores, 14.5 us (dec_gemm 0.32 ->
        // 0.28 ms per step, +0.9 % passages/s).  The family follows from the call shape (L_d == 1), never from the batch, so a
        // row's bits still do not depend on what shares its launch; the other projections of the layer (1024 columns: 80 tiles)
        // measured the same on either family and stay where they were.
        const bool tiled_in = e->opt.dec_ffn_tiled && Ld == 1;
        const int epi_in = d.gated_gelu ? EPI_GEGLU_F16 : EPI_RELU_F16, n_in = d.gated_gelu ? 2 * F : F;
        GemmFold cf = cons();
        if (tiled_in && cf.ssq_in && consumer_uses_pp2(e, epi_in, M, n_in, dm)) {
          // (many rows, or a forced tile shape: the persistent ping-pong kernel takes its row factors ready-made - same block
          // sums, same rk_row_factor, same bits as the fill-in kernels form in their epilogue)
          rowscale(e, st, cf.ssq_in, sl.drowscale, M, cf.nb_in);
          cf = GemmFold(); cf.rowscale = sl.drowscale;
        }
        gemm(e, st, PC_DEC_GEMM, epi_in, sl.dxraw[cur], dm, w.ffn_in_f, dm, sl.dffh, F, M, n_in, dm, 0, 0, 1.f, 1, 0, 0, 0, tiled_in ? false : ws, cf);
      }
      const bool last = l + 1 == d.n_dec_layers;   // the final norm (head kernels) reads the fp32 stream itself
      // (the same switch for FFN-out - 80 tiles of 64x64 with 44 K steps each - took 9 us per layer off the serial profile and
      // nothing measurable off the pipeline: left on the weight-streaming kernel)
      GemmFold plain; plain.few = few;
      gemm(e, st, PC_DEC_GEMM, EPI_RESID_F32, sl.dffh, F, w.ffn_out, F, sl.dhidden, dm, M, dm, F, 0, 0, 1.f, 1, 0, 0, 0, ws, last ? plain : with_prod(GemmFold()));
      if (!last) flip();
      continue;
    }
    gemm(e, st, PC_DEC_GEMM, EPI_RESID_F32, sl.dctx, I, w.co, I, sl.dhidden, dm, M, dm, I, 0, 0, 1.f, 1, 0, 0, 0, ws);
    rmsnorm(e, st, sl.dhidden, w.ln2, sl.dxn, nullptr, M);
    if (d.gated_gelu)
      gemm(e, st, PC_DEC_GEMM, EPI_GEGLU_F16, sl.dxn, dm, w.ffn_in, dm, sl.dffh, F, M, 2 * F, dm, 0, 0, 1.f, 1, 0, 0, 0, ws);
    else
      gemm(e, st, PC_DEC_GEMM, EPI_RELU_F16, sl.dxn, dm, w.ffn_in, dm, sl.dffh, F, M, F, dm, 0, 0, 1.f, 1, 0, 0, 0, ws);
    gemm(e, st, PC_DEC_GEMM, EPI_RESID_F32, sl.dffh, F, w.ffn_out, F, sl.dhidden, dm, M, dm, F, 0, 0, 1.f, 1, 0, 0, 0, ws);
  }
  HIPCHK(e, hipGetLastError());
  return RK_OK;
}

// Encoder on s_enc, decoder on s_dec, ordered by events; the decoder of this slot's PREVIOUS batch must have
// finished reading cross_kv before the encoder overwrites it.
int encoder_then_handoff(rk_engine* e, Slot& sl, int max_ld) {
  hipStream_t se = enc_stream(e, sl), sd = dec_stream(e, sl);
  if (sl.dec_pending && sd != se) HIPCHK(e, hipStreamWaitEvent(se, sl.ev_dec, 0));
  int rc = run_encoder(e, sl, !use_xattn_direct(e, sl, max_ld));
  if (rc) return rc;
  if (sd != se) {
    HIPCHK(e, hipEventRecord(sl.ev_enc, se));
    HIPCHK(e, hipStreamWaitEvent(sd, sl.ev_enc, 0));
  }
  return RK_OK;
}

int mark_decoder_done(rk_engine* e, Slot& sl) {
  HIPCHK(e, hipEventRecord(sl.ev_dec, dec_stream(e, sl)));
  sl.dec_pending = true;
  return RK_OK;
}

int sync_all(rk_engine* e) {
  for (Slot& sl : e->slots) {
    HIPCHK(e, hipStreamSynchronize(sl.se));
    HIPCHK(e, hipStreamSynchronize(sl.sd));
  }
  return RK_OK;
}

int ensure_logits(rk_engine* e, size_t rows) {
  // fused qlm head: rows x ceil(vocab / 32) float2 block statistics, then rows floats of label logits (the [rows, vocab]
  // fp32 logits this buffer used to hold are never materialised)
  const size_t need_elems = rows * (2 * ((size_t)(e->d.vocab + 31) / 32) + 1);
  if (need_elems <= e->logits_cap) return RK_OK;
  int rc = sync_all(e);
  if (rc) return rc;
  if (e->logits) HIPCHK(e, hipFree(e->logits));
  e->logits = nullptr; e->logits_cap = 0;
  HIPCHK(e, hipMalloc((void**)&e->logits, need_elems * sizeof(float)));
  e->logits_cap = need_elems;
  return RK_OK;
}

float head_scale(const rk_engine* e) {   // hf: modeling_t5.py:1044-1045 (scale_decoder_outputs)
  return e->d.tied_head ? 1.0f / std::sqrt((float)e->d.d_model) : 1.0f;
}

int check_batch(rk_engine* e, Slot& sl, const int32_t* tokens, const int32_t* off, int n_seq) {
  if (!e->finalized) return fail(e, RK_ERR_STATE, "engine not finalized");
  if (!tokens || !off || n_seq <= 0) return fail(e, RK_ERR_INVALID, "empty batch (n_seq=%d)", n_seq);
  if (n_seq > e->d.max_seqs) return fail(e, RK_ERR_CAPACITY, "n_seq %d > max_seqs %d", n_seq, e->d.max_seqs);
  if (off[0] != 0) return fail(e, RK_ERR_INVALID, "seq_offsets[0] must be 0");
  int maxL = 0, minL = 1 << 30;
  for (int b = 0; b < n_seq; ++b) {
    const int L = off[b + 1] - off[b];
    if (L <= 0) return fail(e, RK_ERR_INVALID, "sequence %d is empty", b);
    maxL = std::max(maxL, L);
    minL = std::min(minL, L);
  }
  const int T = off[n_seq];
  if (T > e->d.max_tokens) return fail(e, RK_ERR_CAPACITY, "%d tokens > max_tokens %d", T, e->d.max_tokens);
  for (int t = 0; t < T; ++t)
    if (tokens[t] < 0 || tokens[t] >= e->d.vocab) return fail(e, RK_ERR_INVALID, "token id %d out of range at %d", tokens[t], t);
  if ((64 + 256 + 8 + (size_t)maxL) * sizeof(float) > 160 * 1024 || maxL > 65536)
    return fail(e, RK_ERR_CAPACITY, "sequence of %d tokens exceeds the cross-attention LDS budget", maxL);
  sl.maxL = maxL; sl.minL = minL; sl.T = T; sl.n_seq = n_seq;
  return RK_OK;
}

int check_ids(rk_engine* e, const int32_t* ids, int n, const char* what) {
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= e->d.vocab) return fail(e, RK_ERR_INVALID, "%s id %d out of range", what, ids[i]);
  return RK_OK;
}

int upload_dec_ids_shared(rk_engine* e, Slot& sl, const int32_t* prefix, int Ld) {
  std::vector<int> ids((size_t)sl.n_seq * Ld);
  for (int b = 0; b < sl.n_seq; ++b) memcpy(&ids[(size_t)b * Ld], prefix, Ld * sizeof(int));
  hipStream_t st = dec_stream(e, sl);
  if (ids.size() > 8192) {   // larger than a pinned slot: plain synchronous copy
    HIPCHK(e, hipStreamSynchronize(st));
    HIPCHK(e, hipMemcpy(sl.d_dec_ids, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice));
    sl.cache_dec.clear(); ++sl.dec_epoch;
    return RK_OK;
  }
  return upload_small(e, sl, st, &sl.cache_dec, sl.d_dec_ids, 0, ids.data(), (int)ids.size());
}

int stage_slot(rk_engine* e, int slot, const int32_t* tokens, const int32_t* seq_offsets, int n_seq) {
  if (slot < 0 || slot >= RK_SLOTS) return fail(e, RK_ERR_INVALID, "slot %d out of range", slot);
  if (e->family != 0) return fail(e, RK_ERR_STATE, "T5 entry point called on a Llama engine (use rk_llama_*)");
  int rc = set_device(e);
  if (rc) return rc;
  Slot& sl = e->slots[slot];
  sl.staged = false;
  if ((rc = check_batch(e, sl, tokens, seq_offsets, n_seq))) return rc;
  // the slot's previous batch (encoder reads tokens, decoder reads seq_off) must be done before overwriting
  if (sl.dec_pending) { HIPCHK(e, hipEventSynchronize(sl.ev_dec)); sl.dec_pending = false; }
  HIPCHK(e, hipStreamSynchronize(enc_stream(e, sl)));
  HIPCHK(e, hipMemcpy(sl.d_tokens, tokens, (size_t)sl.T * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(sl.d_seq_off, seq_offsets, (size_t)(n_seq + 1) * sizeof(int), hipMemcpyHostToDevice));
  sl.staged = true;
  return RK_OK;
}

// The decoder chain of a call is ~300 dependent launches of kernels that run for a few microseconds each: issued eagerly
// it is bound by the host's launch rate (about 10 us per launch end to end), replayed as ONE HIP graph by the GPU's own
// dependent-kernel boundary (1-2 us).  `body` enqueues the chain on `st`; the second time a key is seen the chain is
// captured, instantiated and cached, from then on it is replayed.  The key holds every value the launch parameters
// depend on (shapes, options epoch); buffers are per-slot and never move.  Profiling runs stay eager (per-kernel events).
template <class F>
int run_graphed(rk_engine* e, hipStream_t st, std::vector<int> key, F&& body) {
  key.push_back(e->opt_epoch);
  if (!e->opt.dec_graph || e->prof_on) return body();
  auto& g = e->graphs[key];
  if (g.exec) { HIPCHK(e, hipGraphLaunch(g.exec, st)); return RK_OK; }
  if (g.failed || g.seen++ == 0) return body();          // first sighting: eager (also does the one-off kernel attribute calls)
  if (e->graphs.size() > 256) {                           // bounded cache: drop everything but this key's slot
    for (auto& kv : e->graphs) if (kv.second.exec && &kv.second != &g) { hipGraphExecDestroy(kv.second.exec); kv.second.exec = nullptr; kv.second.seen = 0; }
  }
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); g.failed = true; return body(); }
  const int rc = body();
  const hipError_t ec = hipStreamEndCapture(st, &graph);
  if (rc != RK_OK || ec != hipSuccess || !graph) {
    (void)hipGetLastError();
    if (graph) hipGraphDestroy(graph);
    g.failed = true;
    return rc != RK_OK ? rc : body();                     // nothing was executed during the capture: run it now
  }
  const hipError_t ei = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (ei != hipSuccess || !g.exec) { (void)hipGetLastError(); g.exec = nullptr; g.failed = true; return body(); }
  HIPCHK(e, hipGraphLaunch(g.exec, st));
  return RK_OK;
}

int score_slot(rk_engine* e, int slot, const int32_t* dec_prefix, int dec_len, const int32_t* out_token_ids, int n_out) {
  if (slot < 0 || slot >= RK_SLOTS) return fail(e, RK_ERR_INVALID, "slot %d out of range", slot);
  int rc = set_device(e);
  if (rc) return rc;
  Slot& sl = e->slots[slot];
  if (!sl.staged) return fail(e, RK_ERR_STATE, "no staged batch in slot %d", slot);
  if (!dec_prefix || dec_len <= 0 || dec_len > e->d.max_dec_len) return fail(e, RK_ERR_CAPACITY, "dec_len %d out of range (max %d)", dec_len, e->d.max_dec_len);
  if (!out_token_ids || n_out <= 0 || n_out > 64) return fail(e, RK_ERR_INVALID, "n_out must be in 1..64 (got %d)", n_out);
  if ((rc = check_ids(e, dec_prefix, dec_len, "decoder")) || (rc = check_ids(e, out_token_ids, n_out, "output"))) return rc;
  hipStream_t sd = dec_stream(e, sl);
  if ((rc = upload_dec_ids_shared(e, sl, dec_prefix, dec_len))) return rc;
  if ((rc = upload_small(e, sl, sd, &sl.cache_out, sl.d_out_ids, 1, out_token_ids, n_out))) return rc;   // n_out <= 64 (checked)
  std::vector<int> rows(sl.n_seq);
  for (int b = 0; b < sl.n_seq; ++b) rows[b] = b * dec_len + dec_len - 1;
  if ((rc = upload_small(e, sl, sd, &sl.cache_rows, sl.d_last_rows, 2, rows.data(), sl.n_seq))) return rc;
  if ((rc = encoder_then_handoff(e, sl, dec_len))) return rc;
#ifdef RK_MEASURE   // measurement builds only (never what build() ships): encoder-chain floor, scores are garbage
  static const bool skip_dec = getenv("RK_DEBUG_SKIP_DECODER") != nullptr;
#else
  constexpr bool skip_dec = false;
#endif
  rc = run_graphed(e, sd, {0, slot, sl.n_seq, dec_len, sl.have_cross_kv ? sl.maxL : (sl.maxL + 63) / 64, (int)sl.have_cross_kv, n_out, (int)skip_dec}, [&]() -> int {
    int r = RK_OK;
    if (!skip_dec && (r = run_decoder(e, sl, dec_len))) return r;
    rmsnorm(e, sd, sl.dhidden, e->dec_final_ln, sl.dlast, sl.d_last_rows, sl.n_seq, head_scale(e));
    Bracket br(e, sd, PC_HEAD, 2.0 * sl.n_seq * n_out * e->d.d_model, 0);
    hipLaunchKernelGGL(head_rows_kernel, dim3((sl.n_seq * n_out + 3) / 4), dim3(256), 0, sd, sl.dlast, e->lm_head,
                       sl.d_out_ids, sl.d_scores, sl.n_seq, n_out, e->d.d_model);
    return RK_OK;
  });
  if (rc) return rc;
  HIPCHK(e, hipMemcpyAsync(sl.h_scores, sl.d_scores, (size_t)sl.n_seq * n_out * sizeof(float), hipMemcpyDeviceToHost, sd));
  HIPCHK(e, hipGetLastError());
  sl.last_n_out = n_out;
  return mark_decoder_done(e, sl);
}


// ---- RCCL, loaded on first use: a 1-GPU process never needs it, and when PyTorch is in the process the SONAME
// librccl.so.1 resolves to the copy torch already mapped (one RCCL per process, like the HIP runtime) ----------------
struct RcclApi {
  void* h = nullptr; bool tried = false; std::string err;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;   // optional
};
RcclApi g_rccl;

// Which librccl: ONE per process.  When PyTorch is in the process its wheel's own copy (torch/lib/librccl.so, 2.26.6 in this
// image against 2.27.7 under /opt/rocm) is usually mapped already - a second, different RCCL beside it would bring its own
// HSA / IPC state.  So: (1) the copy already mapped into this process (first "librccl" entry of /proc/self/maps), by its
// exact path; (2) the SONAME through the loader's search path; (3) the ROCm install.  rk_comm_library_info reports the
// path and version actually bound, and bench.py / run.py log it.
std::string mapped_rccl_path() {
  FILE* f = fopen("/proc/self/maps", "r");
  if (!f) return "";
  char line[4096];
  std::string found;
  while (fgets(line, sizeof line, f)) {
    const char* p = strstr(line, "librccl");
    if (!p) continue;
    const char* path = strchr(line, '/');
    if (!path) continue;
    found.assign(path);
    while (!found.empty() && (found.back() == '\n' || found.back() == ' ')) found.pop_back();
    break;
  }
  fclose(f);
  return found;
}

const RcclApi* rccl_api() {
  if (g_rccl.tried) return g_rccl.h ? &g_rccl : nullptr;
  g_rccl.tried = true;
  const std::string mapped = mapped_rccl_path();
  if (!mapped.empty()) g_rccl.h = dlopen(mapped.c_str(), RTLD_NOW | RTLD_GLOBAL);
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    if (g_rccl.h) break;
    g_rccl.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!g_rccl.h) { g_rccl.err = std::string("dlopen(librccl.so.1) failed: ") + (dlerror() ? dlerror() : "?"); return nullptr; }
#define RK_SYM(field, sym)                                                              \
  g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.h, sym);                        \
  if (!g_rccl.field) { g_rccl.err = std::string("librccl lacks ") + sym; dlclose(g_rccl.h); g_rccl.h = nullptr; return nullptr; }
  RK_SYM(GetUniqueId, "ncclGetUniqueId") RK_SYM(CommInitRank, "ncclCommInitRank") RK_SYM(CommDestroy, "ncclCommDestroy")
  RK_SYM(AllGather, "ncclAllGather") RK_SYM(GetErrorString, "ncclGetErrorString")
#undef RK_SYM
  g_rccl.GetVersion = (decltype(g_rccl.GetVersion))dlsym(g_rccl.h, "ncclGetVersion");
  return &g_rccl;
}

void comm_release(rk_engine* e) {
  if (e->comm) { if (const RcclApi* r = rccl_api()) r->CommDestroy(e->comm); e->comm = nullptr; }
  for (int i = 0; i < RK_SLOTS; ++i) {
    if (e->d_gather[i]) { hipFree(e->d_gather[i]); e->d_gather[i] = nullptr; }
    if (e->h_gather[i]) { hipHostFree(e->h_gather[i]); e->h_gather[i] = nullptr; }
    if (e->ev_gather[i]) { hipEventDestroy(e->ev_gather[i]); e->ev_gather[i] = nullptr; }
    e->gather_pending[i] = false;
  }
  if (e->d_gsend) { hipFree(e->d_gsend); e->d_gsend = nullptr; }
  if (e->d_gall) { hipFree(e->d_gall); e->d_gall = nullptr; }
  if (e->h_gall) { hipHostFree(e->h_gall); e->h_gall = nullptr; }
  if (e->h_gstage) { hipHostFree(e->h_gstage); e->h_gstage = nullptr; }
  if (e->ev_gall) { hipEventDestroy(e->ev_gall); e->ev_gall = nullptr; }
  if (e->ev_append) { hipEventDestroy(e->ev_append); e->ev_append = nullptr; }
  e->gall_pending = false; e->append_foreign = false; e->gall_n = 0;
  e->gather_cap = 0; e->comm_world = 1; e->comm_rank = 0;
}

// the gather / send / staging buffers of a communicator just created (rk_comm_init); on an error the caller releases everything
int comm_alloc_buffers(rk_engine* e) {
  const int world = e->comm_world;
  for (int i = 0; i < RK_SLOTS; ++i) {
    HIPCHK(e, hipMalloc((void**)&e->d_gather[i], e->gather_cap * world * sizeof(float)));
    HIPCHK(e, hipHostMalloc((void**)&e->h_gather[i], e->gather_cap * world * sizeof(float), hipHostMallocDefault));
    HIPCHK(e, hipEventCreateWithFlags(&e->ev_gather[i], hipEventDisableTiming));
  }
  HIPCHK(e, hipMalloc((void**)&e->d_gsend, e->gather_cap * sizeof(float)));
  HIPCHK(e, hipMemset(e->d_gsend, 0, e->gather_cap * sizeof(float)));
  HIPCHK(e, hipMalloc((void**)&e->d_gall, e->gather_cap * world * sizeof(float)));
  HIPCHK(e, hipHostMalloc((void**)&e->h_gall, e->gather_cap * world * sizeof(float), hipHostMallocDefault));
  HIPCHK(e, hipHostMalloc((void**)&e->h_gstage, e->gather_cap * sizeof(float), hipHostMallocDefault));
  HIPCHK(e, hipEventCreateWithFlags(&e->ev_gall, hipEventDisableTiming));
  HIPCHK(e, hipEventCreateWithFlags(&e->ev_append, hipEventDisableTiming));
  return RK_OK;
}

}  // namespace

// =============================================== C ABI =======================================================
extern "C" {

int rk_abi_version(void) { return 1; }

int rk_rel_bucket(int relative_position, int bidirectional, int num_buckets, int max_distance) {
  return rel_bucket(relative_position, bidirectional != 0, num_buckets, max_distance);
}

const char* rk_last_error(const rk_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int rk_profile_num_classes(void) { return PC_COUNT; }
const char* rk_profile_class_name(int cls) { return (cls >= 0 && cls < PC_COUNT) ? kProfNames[cls] : ""; }

int rk_engine_create(const rk_model_desc* desc, int device_ordinal, rk_engine** out) {
  if (!desc || !out) return fail(nullptr, RK_ERR_INVALID, "null argument");
  *out = nullptr;
  const rk_model_desc& d = *desc;
  if (d.d_kv != 64) return fail(nullptr, RK_ERR_INVALID, "d_kv=%d unsupported: the gfx950 attention kernels are built for d_kv=64", d.d_kv);
  if (d.d_model % 64 || (d.n_heads * d.d_kv) % 64 || d.d_ff % 64 || d.vocab % 4)
    return fail(nullptr, RK_ERR_INVALID, "d_model, n_heads*d_kv, d_ff must be multiples of 64 and vocab of 4");
  if (d.max_distance > RK_LUT_R || d.n_buckets < 4 || d.n_buckets > 256)
    return fail(nullptr, RK_ERR_INVALID, "relative attention config unsupported (max_distance<=%d)", RK_LUT_R);
  if (d.max_tokens <= 0 || d.max_seqs <= 0 || d.max_dec_len <= 0 || d.n_enc_layers <= 0 || d.n_dec_layers <= 0)
    return fail(nullptr, RK_ERR_INVALID, "capacities and layer counts must be positive");
  if ((long)d.max_seqs * d.max_dec_len > 8192 * 8) return fail(nullptr, RK_ERR_INVALID, "max_seqs*max_dec_len too large");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(nullptr, RK_ERR_NO_DEVICE, "no HIP device visible: this engine has no CPU path");
  if (device_ordinal < 0 || device_ordinal >= n_dev)
    return fail(nullptr, RK_ERR_NO_DEVICE, "device ordinal %d out of range (%d devices)", device_ordinal, n_dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess)
    return fail(nullptr, RK_ERR_HIP, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, RK_ERR_NO_DEVICE, "device %d is %s; kernels are built for gfx950 (MI355X) only", device_ordinal, prop.gcnArchName);
  rk_engine* e = new rk_engine();
  e->d = d; e->dev = device_ordinal; e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; e->inner = d.n_heads * d.d_kv;
  int prio_lo = 0, prio_hi = 0;
  bool ok = hipSetDevice(device_ordinal) == hipSuccess;
  ok = ok && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess;
  // the decoder chain is a long sequence of tiny dependent kernels: give it dispatch priority over the encoder's
  // chip-filling GEMM grids so it progresses while they run
  for (int i = 0; ok && i < RK_SLOTS; ++i)
    ok = hipStreamCreateWithPriority(&e->slots[i].se, hipStreamNonBlocking, prio_lo) == hipSuccess &&
         hipStreamCreateWithPriority(&e->slots[i].sd, hipStreamNonBlocking, prio_hi) == hipSuccess;
  ok = ok && hipEventCreate(&e->t0) == hipSuccess && hipEventCreate(&e->t1) == hipSuccess &&
       hipEventCreateWithFlags(&e->t_tmp, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < RK_SLOTS; ++i)
    ok = hipEventCreateWithFlags(&e->slots[i].ev_enc, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&e->slots[i].ev_dec, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < 2 * RK_SLOTS; ++i) {
    rk_engine::SkWs& w = e->sk_ws[i];
    w.st = (i & 1) ? e->slots[i >> 1].sd : e->slots[i >> 1].se;
    ok = hipMalloc((void**)&w.slabs, (size_t)KSPLIT_MAX_SLABS * 65536 * sizeof(float)) == hipSuccess &&
         hipMalloc((void**)&w.cnt, KSPLIT_MAX_SLABS * sizeof(int)) == hipSuccess &&
         hipMemset(w.cnt, 0, KSPLIT_MAX_SLABS * sizeof(int)) == hipSuccess;
  }
  if (!ok) {
    for (auto& w : e->sk_ws) { if (w.slabs) hipFree(w.slabs); if (w.cnt) hipFree(w.cnt); }
    delete e;
    return fail(nullptr, RK_ERR_HIP, "stream/event/workspace creation failed");
  }
  *out = e;
  return RK_OK;
}

void rk_engine_destroy(rk_engine* e) {
  if (!e) return;
  hipSetDevice(e->dev);
  for (auto& sl : e->slots) {
    if (sl.se) hipStreamSynchronize(sl.se);
    if (sl.sd) hipStreamSynchronize(sl.sd);
  }
  comm_release(e);
  for (auto& kv : e->graphs) if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
  e->graphs.clear();
  for (void* p : e->allocs) hipFree(p);
  if (e->logits) hipFree(e->logits);
  if (e->amax_val) hipFree(e->amax_val);
  if (e->amax_idx) hipFree(e->amax_idx);
  for (auto& w : e->sk_ws) { if (w.slabs) hipFree(w.slabs); if (w.cnt) hipFree(w.cnt); }
  for (auto& sl : e->slots) {
    if (sl.h_scores) hipHostFree(sl.h_scores);
    if (sl.h_small) hipHostFree(sl.h_small);
    if (sl.ev_enc) hipEventDestroy(sl.ev_enc);
    if (sl.ev_dec) hipEventDestroy(sl.ev_dec);
  }
  for (auto& r : e->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  if (e->t0) hipEventDestroy(e->t0);
  if (e->t1) hipEventDestroy(e->t1);
  if (e->t_tmp) hipEventDestroy(e->t_tmp);
  for (auto& sl : e->slots) {
    if (sl.se) hipStreamDestroy(sl.se);
    if (sl.sd) hipStreamDestroy(sl.sd);
  }
  delete e;
}

int rk_engine_load_tensor(rk_engine* e, const char* hf_name, const void* data, int dtype, const int64_t* shape, int ndim) {
  if (!e || !hf_name || !data || !shape) return fail(e, RK_ERR_INVALID, "null argument");
  if (e->finalized) return fail(e, RK_ERR_STATE, "engine already finalized");
  if (ndim < 1 || ndim > 2) return RK_OK;   // nothing on the path has another rank
  std::string name(hf_name);
  // duplicates of shared.weight in HF checkpoints
  if (name == "encoder.embed_tokens.weight" || name == "decoder.embed_tokens.weight") return RK_OK;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  const bool keep_f32 = ndim == 1 || name.find("relative_attention_bias") != std::string::npos;
  auto get = [&](size_t i) -> float {
    switch (dtype) {
      case RK_F32: return ((const float*)data)[i];
      case RK_F16: return (float)((const half_t*)data)[i];
      case RK_BF16: { uint32_t u = (uint32_t)((const uint16_t*)data)[i] << 16; float f; memcpy(&f, &u, 4); return f; }
      default: return 0.f;
    }
  };
  if (dtype < RK_F32 || dtype > RK_BF16) return fail(e, RK_ERR_INVALID, "unknown dtype %d", dtype);
  if (keep_f32) { t.f.resize(n); for (size_t i = 0; i < n; ++i) t.f[i] = get(i); }
  else if (dtype == RK_F16) { t.h.assign((const half_t*)data, (const half_t*)data + n); }
  else { t.h.resize(n); for (size_t i = 0; i < n; ++i) t.h[i] = (half_t)get(i); }
  e->host[name] = std::move(t);
  return RK_OK;
}

static int llama_finalize(rk_engine* e);

int rk_engine_finalize(rk_engine* e) {
  if (!e) return RK_ERR_INVALID;
  if (e->finalized) return fail(e, RK_ERR_STATE, "already finalized");
  int rc = set_device(e);
  if (rc) return rc;
  if (e->family == 1) return llama_finalize(e);
  const rk_model_desc& d = e->d;
  const int I = e->inner, dm = d.d_model, F = d.d_ff, V = d.vocab;
  std::string missing;
  auto N2 = [&](const std::string& n, int64_t r, int64_t c) { return need(e, n, r, c, &missing); };
  auto N1 = [&](const std::string& n, int64_t r) { return need(e, n, r, -1, &missing); };

  // pass 1: presence / shape check of everything so the error lists all problems at once
  N2("shared.weight", V, dm);
  if (!d.tied_head) N2("lm_head.weight", V, dm);
  N1("encoder.final_layer_norm.weight", dm); N1("decoder.final_layer_norm.weight", dm);
  N2("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", d.n_buckets, d.n_heads);
  N2("decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", d.n_buckets, d.n_heads);
  auto ffn_names = [&](const std::string& p, std::vector<std::string>* v) {
    if (d.gated_gelu) { v->push_back(p + ".wi_0.weight"); v->push_back(p + ".wi_1.weight"); } else v->push_back(p + ".wi.weight");
  };
  for (int l = 0; l < d.n_enc_layers; ++l) {
    const std::string p = "encoder.block." + std::to_string(l) + ".layer";
    for (const char* m : {"q", "k", "v"}) N2(p + ".0.SelfAttention." + m + ".weight", I, dm);
    N2(p + ".0.SelfAttention.o.weight", dm, I);
    N1(p + ".0.layer_norm.weight", dm); N1(p + ".1.layer_norm.weight", dm);
    std::vector<std::string> fn; ffn_names(p + ".1.DenseReluDense", &fn);
    for (auto& s : fn) N2(s, F, dm);
    N2(p + ".1.DenseReluDense.wo.weight", dm, F);
  }
  for (int l = 0; l < d.n_dec_layers; ++l) {
    const std::string p = "decoder.block." + std::to_string(l) + ".layer";
    for (const char* a : {".0.SelfAttention.", ".1.EncDecAttention."}) {
      for (const char* m : {"q", "k", "v"}) N2(p + a + m + ".weight", I, dm);
      N2(p + a + "o.weight", dm, I);
    }
    N1(p + ".0.layer_norm.weight", dm); N1(p + ".1.layer_norm.weight", dm); N1(p + ".2.layer_norm.weight", dm);
    std::vector<std::string> fn; ffn_names(p + ".2.DenseReluDense", &fn);
    for (auto& s : fn) N2(s, F, dm);
    N2(p + ".2.DenseReluDense.wo.weight", dm, F);
  }
  if (!missing.empty()) return fail(e, RK_ERR_MISSING, "missing or mis-shaped tensors: %s", missing.c_str());

  auto H = [&](const std::string& n) -> const std::vector<half_t>& { return e->host[n].h; };
  auto Fv = [&](const std::string& n) -> const std::vector<float>& { return e->host[n].f; };
  auto up_h = [&](half_t** dst, const std::vector<half_t>& v) { return upload(e, dst, v.data(), v.size()); };
  auto up_f = [&](float** dst, const std::vector<float>& v) { return upload(e, dst, v.data(), v.size()); };
  auto cat3 = [&](const std::string& p) {
    std::vector<half_t> v;
    v.reserve((size_t)3 * I * dm);
    for (const char* m : {"q", "k", "v"}) { const auto& s = H(p + m + ".weight"); v.insert(v.end(), s.begin(), s.end()); }
    return v;
  };
  // wi_0 | wi_1 interleaved in groups of 32 output rows: the GEGLU epilogue finds gate and up of one output
  // column in the same lane / same accumulator index of two adjacent 32x32 MFMA fragments.
  auto ffn_in = [&](const std::string& p) {
    if (!d.gated_gelu) return H(p + ".wi.weight");
    const auto& g = H(p + ".wi_0.weight"); const auto& u = H(p + ".wi_1.weight");
    std::vector<half_t> v((size_t)2 * F * dm);
    for (int blk = 0; blk < F / 32; ++blk) {
      memcpy(&v[((size_t)blk * 64) * dm], &g[((size_t)blk * 32) * dm], (size_t)32 * dm * sizeof(half_t));
      memcpy(&v[((size_t)blk * 64 + 32) * dm], &u[((size_t)blk * 32) * dm], (size_t)32 * dm * sizeof(half_t));
    }
    return v;
  };
#define RC(x) do { rc = (x); if (rc) return rc; } while (0)
  RC(up_h(&e->emb, H("shared.weight")));
  if (d.tied_head) e->lm_head = e->emb; else RC(up_h(&e->lm_head, H("lm_head.weight")));
  RC(up_f(&e->enc_final_ln, Fv("encoder.final_layer_norm.weight")));
  RC(up_f(&e->dec_final_ln, Fv("decoder.final_layer_norm.weight")));
  for (int stack = 0; stack < 2; ++stack) {
    const auto& tab = Fv(std::string(stack ? "decoder" : "encoder") + ".block.0.layer.0.SelfAttention.relative_attention_bias.weight");
    std::vector<float> lut((size_t)d.n_heads * RK_LUT_N);
    for (int rel = -RK_LUT_R; rel <= RK_LUT_R; ++rel) {
      const int bkt = rel_bucket(rel, stack == 0, d.n_buckets, d.max_distance);
      for (int h = 0; h < d.n_heads; ++h) lut[(size_t)h * RK_LUT_N + rel + RK_LUT_R] = tab[(size_t)bkt * d.n_heads + h];
    }
    RC(up_f(stack ? &e->lut_dec : &e->lut_enc, lut));
  }
  e->enc.resize(d.n_enc_layers);
  for (int l = 0; l < d.n_enc_layers; ++l) {
    const std::string p = "encoder.block." + std::to_string(l) + ".layer";
    EncLayerW& w = e->enc[l];
    RC(up_h(&w.qkv, cat3(p + ".0.SelfAttention.")));
    RC(up_h(&w.o, H(p + ".0.SelfAttention.o.weight")));
    RC(up_h(&w.ffn_in, ffn_in(p + ".1.DenseReluDense")));
    RC(up_h(&w.ffn_out, H(p + ".1.DenseReluDense.wo.weight")));
    RC(up_f(&w.ln0, Fv(p + ".0.layer_norm.weight")));
    RC(up_f(&w.ln1, Fv(p + ".1.layer_norm.weight")));
    // folded RMSNorm: W'[n][k] = fp16(W[n][k] * ln[k])  (the norm weight multiplies the GEMM's input channels)
    auto folded = [&](const std::vector<half_t>& wm, const std::vector<float>& ln) {
      std::vector<half_t> v(wm.size());
      const size_t rows = wm.size() / dm;
      for (size_t r = 0; r < rows; ++r)
        for (int k = 0; k < dm; ++k) v[r * dm + k] = (half_t)((float)wm[r * dm + k] * ln[k]);
      return v;
    };
    RC(up_h(&w.qkv_f, folded(cat3(p + ".0.SelfAttention."), Fv(p + ".0.layer_norm.weight"))));
    RC(up_h(&w.ffn_in_f, folded(ffn_in(p + ".1.DenseReluDense"), Fv(p + ".1.layer_norm.weight"))));
  }
  e->dec.resize(d.n_dec_layers);
  std::vector<half_t> ckv;
  ckv.reserve((size_t)d.n_dec_layers * 2 * I * dm);
  for (int l = 0; l < d.n_dec_layers; ++l) {
    const std::string p = "decoder.block." + std::to_string(l) + ".layer";
    DecLayerW& w = e->dec[l];
    RC(up_h(&w.qkv, cat3(p + ".0.SelfAttention.")));
    RC(up_h(&w.o, H(p + ".0.SelfAttention.o.weight")));
    RC(up_h(&w.cq, H(p + ".1.EncDecAttention.q.weight")));
    RC(up_h(&w.co, H(p + ".1.EncDecAttention.o.weight")));
    {   // W_k regrouped per head and transposed: ckT[h][c][j] = W_k[h*64 + j][c]
      const auto& wk = H(p + ".1.EncDecAttention.k.weight");
      std::vector<half_t> t((size_t)I * dm);
      for (int h = 0; h < d.n_heads; ++h)
        for (int c = 0; c < dm; ++c)
          for (int j = 0; j < 64; ++j) t[((size_t)h * dm + c) * 64 + j] = wk[((size_t)h * 64 + j) * dm + c];
      RC(up_h(&w.ckT, t));
    }
    RC(up_h(&w.ffn_in, ffn_in(p + ".2.DenseReluDense")));
    RC(up_h(&w.ffn_out, H(p + ".2.DenseReluDense.wo.weight")));
    RC(up_f(&w.ln0, Fv(p + ".0.layer_norm.weight")));
    RC(up_f(&w.ln1, Fv(p + ".1.layer_norm.weight")));
    RC(up_f(&w.ln2, Fv(p + ".2.layer_norm.weight")));
    {
      auto folded = [&](const std::vector<half_t>& wm, const std::vector<float>& ln) {
        std::vector<half_t> v(wm.size());
        const size_t rows = wm.size() / dm;
        for (size_t r = 0; r < rows; ++r)
          for (int k = 0; k < dm; ++k) v[r * dm + k] = (half_t)((float)wm[r * dm + k] * ln[k]);
        return v;
      };
      RC(up_h(&w.qkv_f, folded(cat3(p + ".0.SelfAttention."), Fv(p + ".0.layer_norm.weight"))));
      RC(up_h(&w.cq_f, folded(H(p + ".1.EncDecAttention.q.weight"), Fv(p + ".1.layer_norm.weight"))));
      RC(up_h(&w.ffn_in_f, folded(ffn_in(p + ".2.DenseReluDense"), Fv(p + ".2.layer_norm.weight"))));
    }
    for (const char* m : {"k", "v"}) { const auto& s = H(p + ".1.EncDecAttention." + m + ".weight"); ckv.insert(ckv.end(), s.begin(), s.end()); }
  }
  RC(up_h(&e->cross_kv_w, ckv));
  {
    // W_ov[n][k] = sum_j W_o[n][j] W_v[j][k]  via the engine GEMM: A = W_o [d, I], W = W_v^T [d, I]  (fp32 out)
    half_t* d_vT = nullptr; float* d_ov32 = nullptr;
    RC(dalloc(e, &d_vT, (size_t)dm * I)); RC(dalloc(e, &d_ov32, (size_t)dm * dm));
    std::vector<half_t> vT((size_t)dm * I), ov16((size_t)dm * dm);
    std::vector<float> ov32((size_t)dm * dm);
    const int saved_variant = e->opt.gemm_variant;
    for (int l = 0; l < d.n_dec_layers; ++l) {
      const std::string p = "decoder.block." + std::to_string(l) + ".layer.0.SelfAttention.";
      const auto& wv = H(p + "v.weight");
      for (int j = 0; j < I; ++j)
        for (int k = 0; k < dm; ++k) vT[(size_t)k * I + j] = wv[(size_t)j * dm + k];
      HIPCHK(e, hipMemcpy(d_vT, vT.data(), vT.size() * 2, hipMemcpyHostToDevice));
      gemm(e, e->slots[0].se, PC_OTHER, EPI_STORE_F32, e->dec[l].o, I, d_vT, I, d_ov32, dm, dm, dm, I);
      HIPCHK(e, hipStreamSynchronize(e->slots[0].se));
      HIPCHK(e, hipMemcpy(ov32.data(), d_ov32, ov32.size() * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < ov32.size(); ++i) ov16[i] = (half_t)ov32[i];
      RC(up_h(&e->dec[l].ov, ov16));
      const auto& ln0 = Fv("decoder.block." + std::to_string(l) + ".layer.0.layer_norm.weight");
      for (size_t i = 0; i < ov32.size(); ++i) ov16[i] = (half_t)(ov32[i] * ln0[i % dm]);
      RC(up_h(&e->dec[l].ov_f, ov16));
    }
    e->opt.gemm_variant = saved_variant;
    HIPCHK(e, hipGetLastError());
  }
  e->host.clear();

  // workspaces, sized once for the 288 GB part: nothing is allocated on the hot path afterwards
  const size_t Tc = d.max_tokens, Bc = d.max_seqs, Mc = (size_t)d.max_seqs * d.max_dec_len;
  e->scores_cap = Bc * 64;
  for (Slot& sl : e->slots) {
    RC(dalloc(e, &sl.hidden, Tc * dm)); RC(dalloc(e, &sl.xn, Tc * dm)); RC(dalloc(e, &sl.qkv, Tc * 3 * I));
    RC(dalloc(e, &sl.ctx, Tc * I)); RC(dalloc(e, &sl.ffh, Tc * F)); RC(dalloc(e, &sl.enc_out, Tc * dm));
    RC(dalloc(e, &sl.xraw, Tc * dm)); RC(dalloc(e, &sl.ssq, Tc * ((dm + 63) / 64))); RC(dalloc(e, &sl.rowscale, Tc + 512));   // padded: the ping-pong GEMM reads the row factors of a whole 256-row tile
    HIPCHK(e, hipMemset(sl.rowscale, 0, (Tc + 512) * sizeof(float)));
    RC(dalloc(e, &sl.d_tokens, Tc)); RC(dalloc(e, &sl.d_seq_off, Bc + 1));
    RC(dalloc(e, &sl.cross_kv, (size_t)d.n_dec_layers * Tc * 2 * I));
    RC(dalloc(e, &sl.d_dec_ids, Mc)); RC(dalloc(e, &sl.d_last_rows, Bc)); RC(dalloc(e, &sl.d_out_ids, 8192));
    RC(dalloc(e, &sl.d_labels, (size_t)d.max_dec_len)); RC(dalloc(e, &sl.d_argmax, Bc));
    RC(dalloc(e, &sl.d_row_seq, Mc)); RC(dalloc(e, &sl.d_tree_keys, Mc * (size_t)d.max_dec_len)); RC(dalloc(e, &sl.d_tree_pos, Mc));
    RC(dalloc(e, &sl.dhidden, Mc * dm)); RC(dalloc(e, &sl.dxn, Mc * dm)); RC(dalloc(e, &sl.dqkv, Mc * 3 * I));
    RC(dalloc(e, &sl.dctx, Mc * I)); RC(dalloc(e, &sl.dq, Mc * I)); RC(dalloc(e, &sl.dffh, Mc * F));
    RC(dalloc(e, &sl.dlast, Bc * dm));
    for (int i = 0; i < 2; ++i) { RC(dalloc(e, &sl.dxraw[i], Mc * dm)); RC(dalloc(e, &sl.dssq[i], Mc * ((dm + 31) / 32))); RC(dalloc(e, &sl.dssq_few[i], (size_t)GEMV_MAX_ROWS * e->n_cu)); }
    RC(dalloc(e, &sl.drowscale, Mc));
    RC(dalloc(e, &sl.xqk, (size_t)XA_MAX_ROWS * d.n_heads * dm)); RC(dalloc(e, &sl.xctx, (size_t)XA_MAX_ROWS * d.n_heads * dm));
    RC(dalloc(e, &sl.xpart, (size_t)XA_MAX_CHUNKS * d.n_heads * dm)); RC(dalloc(e, &sl.xstat, (size_t)XA_MAX_CHUNKS * d.n_heads * 2));
    RC(dalloc(e, &sl.d_scores, e->scores_cap));
    HIPCHK(e, hipHostMalloc((void**)&sl.h_scores, e->scores_cap * sizeof(float), hipHostMallocDefault));
    HIPCHK(e, hipHostMalloc((void**)&sl.h_small, 4 * 8192 * sizeof(int), hipHostMallocDefault));
  }
#undef RC
  // dynamic-LDS opt-in for the kernels that may exceed the 64 KiB default
  const int dec_smem_max = (int)((64 + 256 + 8 + (size_t)std::max(d.max_tokens, d.max_dec_len)) * sizeof(float));
  if (dec_smem_max > 160 * 1024) { /* checked per call against maxL */ }
  // best effort: only kernels asking for more than the default dynamic-LDS window need the opt-in
  (void)hipFuncSetAttribute((const void*)attn_dec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            std::min(dec_smem_max, 160 * 1024));
  (void)hipFuncSetAttribute((const void*)attn_dec_seq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define GEMM_ATTR(EPI)                                                                                              \
  (void)hipFuncSetAttribute((const void*)gemm_f16_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES); \
  (void)hipFuncSetAttribute((const void*)gemm_f16_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
  GEMM_ATTR(EPI_STORE_F16) GEMM_ATTR(EPI_RESID_F32) GEMM_ATTR(EPI_GEGLU_F16) GEMM_ATTR(EPI_RELU_F16) GEMM_ATTR(EPI_STORE_F32) GEMM_ATTR(EPI_SWIGLU_F16)
#undef GEMM_ATTR
  (void)hipGetLastError();
  HIPCHK(e, hipDeviceSynchronize());
  e->finalized = true;
  return RK_OK;
}

int rk_t5_stage_slot(rk_engine* e, int slot, const int32_t* tokens, const int32_t* seq_offsets, int n_seq) {
  if (!e) return RK_ERR_INVALID;
  return stage_slot(e, slot, tokens, seq_offsets, n_seq);
}
int rk_t5_stage(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq) {
  return rk_t5_stage_slot(e, 0, tokens, seq_offsets, n_seq);
}

int rk_t5_score_slot(rk_engine* e, int slot, const int32_t* dec_prefix, int dec_len, const int32_t* out_token_ids, int n_out) {
  if (!e) return RK_ERR_INVALID;
  return score_slot(e, slot, dec_prefix, dec_len, out_token_ids, n_out);
}
int rk_t5_score_staged(rk_engine* e, const int32_t* dec_prefix, int dec_len, const int32_t* out_token_ids, int n_out) {
  return rk_t5_score_slot(e, 0, dec_prefix, dec_len, out_token_ids, n_out);
}

int rk_engine_sync(rk_engine* e) {
  if (!e) return RK_ERR_INVALID;
  int rc = set_device(e);
  if (rc) return rc;
  return sync_all(e);
}

int rk_t5_read_scores_slot(rk_engine* e, int slot, float* out_logits, int n_floats) {
  if (!e || !out_logits || slot < 0 || slot >= RK_SLOTS) return RK_ERR_INVALID;
  Slot& sl = e->slots[slot];
  if (n_floats > sl.n_seq * sl.last_n_out) return fail(e, RK_ERR_INVALID, "asked for %d floats, have %d", n_floats, sl.n_seq * sl.last_n_out);
  if (sl.dec_pending) { HIPCHK(e, hipEventSynchronize(sl.ev_dec)); sl.dec_pending = false; }
  memcpy(out_logits, sl.h_scores, (size_t)n_floats * sizeof(float));
  return RK_OK;
}
int rk_t5_read_scores(rk_engine* e, float* out_logits, int n_floats) { return rk_t5_read_scores_slot(e, 0, out_logits, n_floats); }

int rk_t5_scores_device_ptr(rk_engine* e, void** out_ptr) {
  if (!e || !out_ptr) return RK_ERR_INVALID;
  *out_ptr = e->slots[0].d_scores;
  return RK_OK;
}

int rk_engine_num_slots(void) { return RK_SLOTS; }

int rk_t5_score(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq, const int32_t* dec_prefix,
                int dec_len, const int32_t* out_token_ids, int n_out, float* out_logits) {
  int rc;
  if ((rc = rk_t5_stage(e, tokens, seq_offsets, n_seq))) return rc;
  if ((rc = rk_t5_score_staged(e, dec_prefix, dec_len, out_token_ids, n_out))) return rc;
  return rk_t5_read_scores(e, out_logits, n_seq * n_out);
}

int rk_t5_qlm(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq, const int32_t* labels,
              int n_labels, float* out_scores) {
  int rc;
  if ((rc = rk_t5_stage(e, tokens, seq_offsets, n_seq))) return rc;
  Slot& sl = e->slots[0];
  if (!labels || n_labels <= 0 || n_labels > e->d.max_dec_len) return fail(e, RK_ERR_CAPACITY, "n_labels %d out of range (max %d)", n_labels, e->d.max_dec_len);
  if ((rc = check_ids(e, labels, n_labels, "label"))) return rc;
  // decoder input = shift_right(labels): [decoder_start(0), labels[:-1]]  (hf: modeling_t5.py:618-637)
  std::vector<int> dec_in(n_labels);
  dec_in[0] = 0;
  for (int t = 1; t < n_labels; ++t) dec_in[t] = labels[t - 1];
  hipStream_t sd = dec_stream(e, sl);
  if ((rc = upload_dec_ids_shared(e, sl, dec_in.data(), n_labels))) return rc;
  HIPCHK(e, hipStreamSynchronize(sd));
  HIPCHK(e, hipMemcpy(sl.d_labels, labels, n_labels * sizeof(int), hipMemcpyHostToDevice));
  const int M = n_seq * n_labels;
  if ((rc = ensure_logits(e, M))) return rc;
  if ((rc = encoder_then_handoff(e, sl, n_labels))) return rc;
  if ((rc = run_decoder(e, sl, n_labels))) return rc;
  rmsnorm(e, sd, sl.dhidden, e->dec_final_ln, sl.dxn, nullptr, M, head_scale(e));
  // head GEMM with the log-sum-exp fused into its epilogue: per row and 32-column block (max, sum exp) + the label's logit
  const int nblk = (e->d.vocab + 31) / 32;
  e->lse_labels = sl.d_labels; e->lse_npos = n_labels; e->lse_xlab = e->logits + (size_t)M * nblk * 2;
  gemm(e, sd, PC_HEAD, EPI_LSE_F32, sl.dxn, e->d.d_model, e->lm_head, e->d.d_model, e->logits, nblk, M, e->d.vocab, e->d.d_model);
  hipLaunchKernelGGL(qlm_lse_kernel, dim3(n_seq), dim3(256), 0, sd, (const float2*)e->logits, nblk, e->lse_xlab, n_labels, sl.d_scores);
  HIPCHK(e, hipMemcpyAsync(sl.h_scores, sl.d_scores, (size_t)n_seq * sizeof(float), hipMemcpyDeviceToHost, sd));
  if ((rc = mark_decoder_done(e, sl))) return rc;
  if ((rc = sync_all(e))) return rc;
  sl.dec_pending = false;
  HIPCHK(e, hipGetLastError());
  memcpy(out_scores, sl.h_scores, (size_t)n_seq * sizeof(float));
  return RK_OK;
}

// Greedy head: full-vocabulary logits of `rows` final-normed rows (x: [rows, d_model] fp16) reduced to their first arg-max
// WITHOUT writing the logits: the weight-streaming GEMM keeps per 32-column block the maximum and its first column
// (gemm.h: EPI_ARGMAX_F32), argmax_blocks_kernel picks per row.  hf: modeling_t5.py:1044-1047 + torch.argmax.
static int ensure_amax(rk_engine* e, size_t rows, int vocab) {
  if (rows <= e->amax_rows) return RK_OK;
  int rc = sync_all(e);
  if (rc) return rc;
  if (e->amax_val) HIPCHK(e, hipFree(e->amax_val));
  if (e->amax_idx) HIPCHK(e, hipFree(e->amax_idx));
  e->amax_val = nullptr; e->amax_idx = nullptr; e->amax_rows = 0;
  const size_t nblk = (size_t)(vocab + 31) / 32;
  HIPCHK(e, hipMalloc((void**)&e->amax_val, rows * nblk * sizeof(float)));
  HIPCHK(e, hipMalloc((void**)&e->amax_idx, rows * nblk * sizeof(int)));
  e->amax_rows = rows;
  return RK_OK;
}
static void head_argmax(rk_engine* e, hipStream_t st, const half_t* x, int rows, int d_model, int vocab, int* d_out) {
  const int nblk = (vocab + 31) / 32;
  gemm(e, st, PC_HEAD, EPI_ARGMAX_F32, x, d_model, e->lm_head, d_model, e->amax_val, nblk, rows, vocab, d_model, 0, 0, 1.f, 1, 0, 0, 0, true);
  hipLaunchKernelGGL(argmax_blocks_kernel, dim3(rows), dim3(256), 0, st, e->amax_val, e->amax_idx, nblk, d_out);
}

// One greedy step over the staged batch (encoder done): decoder over rows[b] (Ld ids per sequence), final norm of the last
// position, full-vocabulary head, arg-max -> amax[b].  Synchronous (the caller decides the next ids on the host).
static int greedy_step(rk_engine* e, Slot& sl, const std::vector<std::vector<int>>& rows, int Ld, std::vector<int>& amax) {
  const int n_seq = (int)rows.size();
  hipStream_t sd = dec_stream(e, sl);
  std::vector<int> flat((size_t)n_seq * Ld), rowmap(n_seq);
  for (int b = 0; b < n_seq; ++b) { memcpy(&flat[(size_t)b * Ld], rows[b].data(), Ld * sizeof(int)); rowmap[b] = b * Ld + Ld - 1; }
  HIPCHK(e, hipStreamSynchronize(sd));
  HIPCHK(e, hipMemcpy(sl.d_dec_ids, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(sl.d_last_rows, rowmap.data(), n_seq * sizeof(int), hipMemcpyHostToDevice));
  sl.cache_dec.clear(); sl.cache_rows.clear(); ++sl.dec_epoch;
  int rc = run_graphed(e, sd, {1, 0, n_seq, Ld, sl.have_cross_kv ? sl.maxL : (sl.maxL + 63) / 64, (int)sl.have_cross_kv, (int)e->amax_rows}, [&]() -> int {
    int r = run_decoder(e, sl, Ld);
    if (r) return r;
    rmsnorm(e, sd, sl.dhidden, e->dec_final_ln, sl.dlast, sl.d_last_rows, n_seq, head_scale(e));
    head_argmax(e, sd, sl.dlast, n_seq, e->d.d_model, e->d.vocab, sl.d_argmax);
    return RK_OK;
  });
  if (rc) return rc;
  amax.resize(n_seq);
  HIPCHK(e, hipMemcpyAsync(amax.data(), sl.d_argmax, n_seq * sizeof(int), hipMemcpyDeviceToHost, sd));
  HIPCHK(e, hipStreamSynchronize(sd));
  HIPCHK(e, hipGetLastError());
  return RK_OK;
}

int rk_t5_greedy(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq, const int32_t* dec_prefix,
                 int dec_len, int max_new, int eos_id, int pad_id, int32_t* out_tokens, int32_t* out_steps) {
  int rc;
  if ((rc = rk_t5_stage(e, tokens, seq_offsets, n_seq))) return rc;
  Slot& sl = e->slots[0];
  if (!dec_prefix || dec_len <= 0 || max_new <= 0 || dec_len + max_new - 1 > e->d.max_dec_len)
    return fail(e, RK_ERR_CAPACITY, "dec_len %d + max_new %d exceeds max_dec_len %d", dec_len, max_new, e->d.max_dec_len);
  if ((rc = check_ids(e, dec_prefix, dec_len, "decoder"))) return rc;
  if ((rc = ensure_amax(e, (size_t)n_seq, e->d.vocab))) return rc;
  if ((rc = encoder_then_handoff(e, sl, dec_len + max_new - 1))) return rc;
  hipStream_t sd = dec_stream(e, sl);
  // Per-row decoder ids grow by one token per step; the tiny decoder is recomputed over the whole prefix each
  // step (cross K/V are reused), which equals HF's KV-cached greedy loop (hf: generation/utils.py:2868-2935).
  std::vector<std::vector<int>> rows(n_seq, std::vector<int>(dec_prefix, dec_prefix + dec_len));
  std::vector<char> done(n_seq, 0);
  std::vector<int> amax(n_seq);
  for (int b = 0; b < n_seq; ++b)
    for (int t = 0; t < max_new; ++t) out_tokens[b * max_new + t] = pad_id;
  int steps = 0;
  for (int t = 0; t < max_new; ++t) {
    if ((rc = greedy_step(e, sl, rows, dec_len + t, amax))) return rc;
    ++steps;
    bool all_done = true;
    for (int b = 0; b < n_seq; ++b) {
      const int tok = done[b] ? pad_id : amax[b];      // finished rows emit pad (hf: generation/utils.py:2927-2929)
      out_tokens[b * max_new + t] = tok;
      rows[b].push_back(tok);
      if (tok == eos_id) done[b] = 1;
      all_done = all_done && done[b];
    }
    if (all_done) break;
  }
  if ((rc = mark_decoder_done(e, sl))) return rc;
  if ((rc = sync_all(e))) return rc;
  sl.dec_pending = false;
  if (out_steps) *out_steps = steps;
  return RK_OK;
}

// Two greedy tokens in ONE decoder pass (the setwise `generation` compare: ref llmrankers/setwise.py:113-121 runs
// generate(max_new_tokens=2) after "<pad> Passage").  The first new token is almost always one of a few label tokens, and a
// decoder pass over a handful of rows costs what its ~270 launches cost, so the second step is computed for EVERY candidate
// at once: the prefix rows of a prompt once, plus one row per candidate at position dec_len that attends to the prefix rows
// and itself (attn_dec_kernel's tree form) and to the prompt (XAttnArgs::row_seq).  The last prefix row gives token 1 over
// the full vocabulary; the row of the candidate that IS token 1 gives token 2.  Rows are independent of the batch they run in (tests), so both tokens are bit-identical to
// rk_t5_greedy(max_new = 2); a first token outside the candidates, or a batch too large for the workspace, takes that path.
int rk_t5_greedy2(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq, const int32_t* dec_prefix,
                  int dec_len, const int32_t* cand_ids, int n_cand, int eos_id, int pad_id, int32_t* out_tokens, int32_t* out_steps) {
  if (!e) return RK_ERR_INVALID;
  const int Ld = dec_len + 1;
  const long per_seq = (long)dec_len + n_cand;                         // rows of one prompt: the prefix once, one row per candidate
  const long M = (long)n_seq * per_seq, R = (long)n_seq * (1 + n_cand);
  const bool fits = cand_ids && n_cand > 0 && dec_prefix && dec_len > 0 && Ld <= e->d.max_dec_len && R <= e->d.max_seqs &&
                    M <= (long)e->d.max_seqs * e->d.max_dec_len && M <= XA_MAX_ROWS && M <= e->opt.greedy_spec &&
                    use_xattn_direct(e, e->slots[0], Ld);
  if (!fits) return rk_t5_greedy(e, tokens, seq_offsets, n_seq, dec_prefix, dec_len, 2, eos_id, pad_id, out_tokens, out_steps);
  int rc;
  if ((rc = check_ids(e, cand_ids, n_cand, "candidate"))) return rc;
  if ((rc = rk_t5_stage(e, tokens, seq_offsets, n_seq))) return rc;
  Slot& sl = e->slots[0];
  if ((rc = check_ids(e, dec_prefix, dec_len, "decoder"))) return rc;
  if ((rc = ensure_amax(e, (size_t)R, e->d.vocab))) return rc;
  if ((rc = encoder_then_handoff(e, sl, Ld))) return rc;
  hipStream_t sd = dec_stream(e, sl);
  // row layout of prompt b: [prefix position 0 .. dec_len-1][candidate 0 .. n_cand-1 at position dec_len]
  std::vector<int> ids((size_t)M), rows((size_t)R), rseq((size_t)M), rpos((size_t)M), keys((size_t)M * Ld, 0), amax((size_t)R);
  for (int b = 0; b < n_seq; ++b) {
    const int r0 = (int)(b * per_seq);
    for (int i = 0; i < dec_len; ++i) {
      const int r = r0 + i;
      ids[r] = dec_prefix[i]; rseq[r] = b; rpos[r] = i;
      for (int j = 0; j <= i; ++j) keys[(size_t)r * Ld + j] = r0 + j;
    }
    rows[b] = r0 + dec_len - 1;                                       // token 1: the last prefix row
    for (int c = 0; c < n_cand; ++c) {
      const int r = r0 + dec_len + c;
      ids[r] = cand_ids[c]; rseq[r] = b; rpos[r] = dec_len;
      for (int j = 0; j < dec_len; ++j) keys[(size_t)r * Ld + j] = r0 + j;
      keys[(size_t)r * Ld + dec_len] = r;
      rows[n_seq + b * n_cand + c] = r;                                // token 2 if token 1 was candidate c
    }
  }
  // the five index arrays are the same for every compare of a query (same prefix, candidates and prompt count): uploaded
  // only when they differ from what this path left on the device and nobody else wrote the shared id / row buffers since
  std::vector<int> sig;
  sig.reserve(ids.size() + rows.size() + rseq.size() + rpos.size() + keys.size() + 2);
  sig.push_back(n_seq); sig.push_back(Ld);
  for (const std::vector<int>* v : {&ids, &rows, &rseq, &rpos, &keys}) sig.insert(sig.end(), v->begin(), v->end());
  if (sl.g2_epoch != sl.dec_epoch || sig != sl.cache_g2) {
    HIPCHK(e, hipStreamSynchronize(sd));
    HIPCHK(e, hipMemcpy(sl.d_dec_ids, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(sl.d_last_rows, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(sl.d_row_seq, rseq.data(), rseq.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(sl.d_tree_pos, rpos.data(), rpos.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(sl.d_tree_keys, keys.data(), keys.size() * sizeof(int), hipMemcpyHostToDevice));
    sl.cache_dec.clear(); sl.cache_rows.clear();
    sl.cache_g2.swap(sig);
    sl.g2_epoch = ++sl.dec_epoch;
  }
  const DecTree tree{(int)M, sl.d_tree_keys, sl.d_tree_pos, sl.d_row_seq};
  rc = run_graphed(e, sd, {2, 0, n_seq, Ld, (sl.maxL + 63) / 64, n_cand, (int)e->amax_rows}, [&]() -> int {
    int r = run_decoder(e, sl, Ld, &tree);
    if (r) return r;
    rmsnorm(e, sd, sl.dhidden, e->dec_final_ln, sl.dlast, sl.d_last_rows, (int)R, head_scale(e));
    head_argmax(e, sd, sl.dlast, (int)R, e->d.d_model, e->d.vocab, sl.d_argmax);
    return RK_OK;
  });
  if (rc) return rc;
  HIPCHK(e, hipMemcpyAsync(amax.data(), sl.d_argmax, (size_t)R * sizeof(int), hipMemcpyDeviceToHost, sd));
  HIPCHK(e, hipStreamSynchronize(sd));
  HIPCHK(e, hipGetLastError());
  bool all_done = true, miss = false;
  for (int b = 0; b < n_seq; ++b) {
    const int t1 = amax[b];
    out_tokens[b * 2] = t1;
    out_tokens[b * 2 + 1] = pad_id;                                    // finished rows emit pad (hf: generation/utils.py:2927-2929)
    if (t1 == eos_id) continue;
    all_done = false;
    int c = 0;
    while (c < n_cand && cand_ids[c] != t1) ++c;
    if (c == n_cand) miss = true;
    else out_tokens[b * 2 + 1] = amax[n_seq + b * n_cand + c];
  }
  if (miss) {
    // a first token outside the candidates: the ordinary second step (same encoder output, one more decoder pass)
    std::vector<std::vector<int>> seq_rows(n_seq, std::vector<int>(dec_prefix, dec_prefix + dec_len));
    for (int b = 0; b < n_seq; ++b) seq_rows[b].push_back(out_tokens[b * 2]);
    std::vector<int> a2;
    if ((rc = greedy_step(e, sl, seq_rows, Ld, a2))) return rc;
    for (int b = 0; b < n_seq; ++b)
      if (out_tokens[b * 2] != eos_id) out_tokens[b * 2 + 1] = a2[b];
  }
  if ((rc = mark_decoder_done(e, sl))) return rc;
  if ((rc = sync_all(e))) return rc;
  sl.dec_pending = false;
  if (out_steps) *out_steps = all_done ? 1 : 2;
  return RK_OK;
}

// =============================================== Llama family ================================================
// Decoder-only setwise scoring (ref: llmrankers/setwise.py:60-69, 159-177): prefill of the whole prompt, then the
// arg-max of the last position's logits (generate(max_new_tokens=1, do_sample=False)).  hf: models/llama/modeling_llama.py.
int rk_llama_create(const rk_llama_desc* desc, int device_ordinal, rk_engine** out) {
  if (!desc || !out) return fail(nullptr, RK_ERR_INVALID, "null argument");
  *out = nullptr;
  const rk_llama_desc& l = *desc;
  if (l.head_dim != 128) return fail(nullptr, RK_ERR_INVALID, "head_dim=%d unsupported: the gfx950 causal attention kernel is built for head_dim=128", l.head_dim);
  if (l.n_kv_heads <= 0 || l.n_heads % l.n_kv_heads) return fail(nullptr, RK_ERR_INVALID, "n_heads must be a multiple of n_kv_heads");
  if (l.hidden % 64 || l.intermediate % 64 || l.vocab % 4 || l.hidden > 4096) return fail(nullptr, RK_ERR_INVALID, "hidden / intermediate must be multiples of 64 (hidden <= 4096), vocab of 4");
  if (l.max_tokens <= 0 || l.max_seqs <= 0 || l.n_layers <= 0) return fail(nullptr, RK_ERR_INVALID, "capacities and layer count must be positive");
  rk_model_desc d{};
  d.vocab = l.vocab; d.d_model = l.hidden; d.n_heads = l.n_heads; d.d_kv = 64; d.d_ff = l.intermediate;   // d_kv only passes the T5 checks
  d.n_enc_layers = l.n_layers; d.n_dec_layers = 1; d.n_buckets = 32; d.max_distance = 128; d.gated_gelu = 1; d.tied_head = l.tied_head;
  d.eps = l.eps; d.max_tokens = l.max_tokens; d.max_seqs = l.max_seqs; d.max_dec_len = 1;
  if ((long)d.n_heads * 64 % 64) return fail(nullptr, RK_ERR_INVALID, "bad head count");
  int rc = rk_engine_create(&d, device_ordinal, out);
  if (rc) return rc;
  (*out)->family = 1; (*out)->ld = l; (*out)->inner = l.n_heads * l.head_dim;
  return RK_OK;
}

static int llama_finalize(rk_engine* e) {
  const rk_llama_desc& l = e->ld;
  const int dm = l.hidden, Q = l.n_heads * 128, KV = l.n_kv_heads * 128, F = l.intermediate, V = l.vocab;
  std::string missing;
  auto N2 = [&](const std::string& n, int64_t r, int64_t c) { return need(e, n, r, c, &missing); };
  auto N1 = [&](const std::string& n, int64_t r) { return need(e, n, r, -1, &missing); };
  N2("model.embed_tokens.weight", V, dm);
  if (!l.tied_head) N2("lm_head.weight", V, dm);
  N1("model.norm.weight", dm);
  for (int i = 0; i < l.n_layers; ++i) {
    const std::string p = "model.layers." + std::to_string(i);
    N2(p + ".self_attn.q_proj.weight", Q, dm); N2(p + ".self_attn.k_proj.weight", KV, dm); N2(p + ".self_attn.v_proj.weight", KV, dm);
    N2(p + ".self_attn.o_proj.weight", dm, Q);
    N2(p + ".mlp.gate_proj.weight", F, dm); N2(p + ".mlp.up_proj.weight", F, dm); N2(p + ".mlp.down_proj.weight", dm, F);
    N1(p + ".input_layernorm.weight", dm); N1(p + ".post_attention_layernorm.weight", dm);
  }
  if (!missing.empty()) return fail(e, RK_ERR_MISSING, "missing or mis-shaped tensors: %s", missing.c_str());
  auto H = [&](const std::string& n) -> const std::vector<half_t>& { return e->host[n].h; };
  auto Fv = [&](const std::string& n) -> const std::vector<float>& { return e->host[n].f; };
  int rc = RK_OK;
#define RC(x) do { rc = (x); if (rc) return rc; } while (0)
  RC(upload(e, &e->emb, H("model.embed_tokens.weight").data(), H("model.embed_tokens.weight").size()));
  if (l.tied_head) e->lm_head = e->emb; else RC(upload(e, &e->lm_head, H("lm_head.weight").data(), H("lm_head.weight").size()));
  RC(upload(e, &e->l_final_ln, Fv("model.norm.weight").data(), (size_t)dm));
  e->ll.resize(l.n_layers);
  std::vector<half_t> buf;
  for (int i = 0; i < l.n_layers; ++i) {
    const std::string p = "model.layers." + std::to_string(i);
    LlamaLayerW& w = e->ll[i];
    {   // q | k | v rows with the input RMSNorm weight folded into the columns
      const auto& ln = Fv(p + ".input_layernorm.weight");
      buf.resize((size_t)(Q + 2 * KV) * dm);
      size_t r0 = 0;
      for (const char* m : {"q_proj", "k_proj", "v_proj"}) {
        const auto& src = H(p + ".self_attn." + m + ".weight");
        const size_t rows = src.size() / dm;
        for (size_t r = 0; r < rows; ++r)
          for (int k = 0; k < dm; ++k) buf[(r0 + r) * dm + k] = (half_t)((float)src[r * dm + k] * ln[k]);
        r0 += rows;
      }
      RC(upload(e, &w.qkv_f, buf.data(), buf.size()));
    }
    RC(upload(e, &w.o, H(p + ".self_attn.o_proj.weight").data(), (size_t)dm * Q));
    {   // gate | up interleaved in groups of 32 rows (the SwiGLU epilogue pairs them in one lane), post-attention norm folded
      const auto& ln = Fv(p + ".post_attention_layernorm.weight");
      const auto& g = H(p + ".mlp.gate_proj.weight"); const auto& u = H(p + ".mlp.up_proj.weight");
      buf.resize((size_t)2 * F * dm);
      for (int blk = 0; blk < F / 32; ++blk)
        for (int r = 0; r < 32; ++r)
          for (int k = 0; k < dm; ++k) {
            buf[((size_t)blk * 64 + r) * dm + k] = (half_t)((float)g[((size_t)blk * 32 + r) * dm + k] * ln[k]);
            buf[((size_t)blk * 64 + 32 + r) * dm + k] = (half_t)((float)u[((size_t)blk * 32 + r) * dm + k] * ln[k]);
          }
      RC(upload(e, &w.gu_f, buf.data(), buf.size()));
    }
    RC(upload(e, &w.down, H(p + ".mlp.down_proj.weight").data(), (size_t)dm * F));
  }
  e->host.clear();
  {   // rotary tables, float32 like hf: modeling_llama.py:94-127: freq_i = theta^(-2i/128), then the llama3 rope type's
      // wavelength-dependent scaling (hf: modeling_rope_utils.py _compute_llama3_parameters) when it was asked for
    const size_t Tc = l.max_tokens;
    std::vector<float> c(Tc * 64), sn(Tc * 64);
    for (int i = 0; i < 64; ++i) {
      float inv = 1.0f / powf(l.rope_theta, (float)(2 * i) / 128.0f);
      if (e->rope_factor > 0.f) {
        const float orig = (float)e->rope_orig, wavelen = 6.283185307179586f / inv;
        const float scaled = wavelen > orig / e->rope_low ? inv / e->rope_factor : inv;
        const float smooth = (orig / wavelen - e->rope_low) / (e->rope_high - e->rope_low);
        const bool medium = !(wavelen < orig / e->rope_high) && !(wavelen > orig / e->rope_low);
        inv = medium ? (1.0f - smooth) * scaled / e->rope_factor + smooth * scaled : scaled;
      }
      for (size_t t = 0; t < Tc; ++t) { const float a = (float)t * inv; c[t * 64 + i] = cosf(a); sn[t * 64 + i] = sinf(a); }
    }
    RC(upload(e, &e->rope_cos, c.data(), c.size())); RC(upload(e, &e->rope_sin, sn.data(), sn.size()));
  }
  const size_t Tc = l.max_tokens, Bc = l.max_seqs;
  Slot& sl = e->slots[0];
  RC(dalloc(e, &sl.hidden, Tc * dm)); RC(dalloc(e, &sl.xraw, Tc * dm)); RC(dalloc(e, &sl.ssq, Tc * ((dm + 63) / 64)));
  RC(dalloc(e, &sl.rowscale, Tc + 512)); HIPCHK(e, hipMemset(sl.rowscale, 0, (Tc + 512) * sizeof(float)));
  RC(dalloc(e, &sl.qkv, Tc * (Q + 2 * KV))); RC(dalloc(e, &sl.ctx, Tc * Q)); RC(dalloc(e, &sl.ffh, Tc * F));
  RC(dalloc(e, &sl.d_tokens, Tc)); RC(dalloc(e, &e->d_pos, Tc)); RC(dalloc(e, &sl.d_seq_off, Bc + 1));
  RC(dalloc(e, &sl.d_last_rows, Bc)); RC(dalloc(e, &sl.d_out_ids, 8192)); RC(dalloc(e, &sl.d_argmax, Bc)); RC(dalloc(e, &sl.dlast, Bc * dm));
  e->scores_cap = Bc * 64;
  RC(dalloc(e, &sl.d_scores, e->scores_cap));
  HIPCHK(e, hipHostMalloc((void**)&sl.h_scores, e->scores_cap * sizeof(float), hipHostMallocDefault));
  HIPCHK(e, hipHostMalloc((void**)&sl.h_small, 4 * 8192 * sizeof(int), hipHostMallocDefault));
#undef RC
  HIPCHK(e, hipDeviceSynchronize());
  e->finalized = true;
  return RK_OK;
}

// prefill of the ragged batch; leaves the final-normed LAST hidden state of every sequence in sl.dlast [n_seq, hidden]
static int llama_prefill(rk_engine* e, const int32_t* tokens, const int32_t* off, int n_seq) {
  if (!e || e->family != 1) return fail(e, RK_ERR_STATE, "not a Llama engine");
  int rc = set_device(e);
  if (rc) return rc;
  Slot& sl = e->slots[0];
  if ((rc = check_batch(e, sl, tokens, off, n_seq))) return rc;
  const rk_llama_desc& l = e->ld;
  const int T = sl.T, dm = l.hidden, Q = l.n_heads * 128, KV = l.n_kv_heads * 128, F = l.intermediate, ldq = Q + 2 * KV;
  hipStream_t st = sl.se;
  HIPCHK(e, hipStreamSynchronize(st));
  std::vector<int> pos(T), last(n_seq);
  for (int b = 0; b < n_seq; ++b) {
    for (int t = off[b]; t < off[b + 1]; ++t) pos[t] = t - off[b];
    last[b] = off[b + 1] - 1;
  }
  HIPCHK(e, hipMemcpy(sl.d_tokens, tokens, (size_t)T * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(e->d_pos, pos.data(), (size_t)T * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(sl.d_seq_off, off, (size_t)(n_seq + 1) * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(sl.d_last_rows, last.data(), (size_t)n_seq * sizeof(int), hipMemcpyHostToDevice));
  GemmFold cons, prod;
  cons.rowscale = sl.rowscale; prod.xraw = sl.xraw; prod.ssq = sl.ssq;
  embed(e, st, sl.d_tokens, sl.hidden, T, sl.xraw, sl.rowscale);
  const float scale_log2e = (1.0f / std::sqrt(128.0f)) * 1.4426950408889634f;
  for (int i = 0; i < l.n_layers; ++i) {
    const LlamaLayerW& w = e->ll[i];
    gemm(e, st, PC_ENC_GEMM_QKV, EPI_STORE_F16, sl.xraw, dm, w.qkv_f, dm, sl.qkv, ldq, T, ldq, dm, 0, 0, 1.f, 1, 0, 0, 0, false, cons);
    {
      Bracket br(e, st, PC_OTHER, 0, (double)T * (Q + KV) * 4.0);
      hipLaunchKernelGGL(rope128_kernel, dim3(T), dim3(256), 0, st, sl.qkv, e->d_pos, e->rope_cos, e->rope_sin, ldq, l.n_heads + l.n_kv_heads);
    }
    {
      AttnCausalArgs a{sl.qkv, sl.ctx, sl.d_seq_off, ldq, Q, l.n_heads, l.n_kv_heads, scale_log2e, 0, 0, e->opt.attn_ko};
      Bracket br(e, st, PC_ENC_ATTN, 2.0 * (double)sl.maxL * T * Q, (double)T * (2 * Q + 2 * KV) * 2.0);   // causal: half of 4 L T Q
      if (e->opt.llama_attn_dma) {     // K / V chunks by LDS-DMA, V^T by transposing reads (round 5); chosen by the option alone: batch-independent
        static std::atomic<uint64_t> attr_done{0};
        static std::atomic<uint64_t> attr_done8{0};
        int lds = ATCD_LDS_BYTES, lds_max = ATCD_LDS_BYTES;
#ifdef RK_MEASURE
        lds_max += 49152;
        if (e->opt.attn_ko & 256) lds += 49152;          // residency probe: 112 KiB per workgroup = ONE per CU for certain
#endif
        const int nw = e->opt.llama_attn_nw == 8 ? 8 : 4;   // same bits either way
        a.n_seq = n_seq; a.nqb = (sl.maxL + 32 * nw - 1) / (32 * nw);
        const dim3 grid(xcd_grid(n_seq * l.n_kv_heads, (l.n_heads / l.n_kv_heads) * a.nqb));
        if (nw == 8) { ensure_dynamic_lds((const void*)attn_causal128_dma_kernel<8>, lds_max, attr_done8); hipLaunchKernelGGL(attn_causal128_dma_kernel<8>, grid, dim3(512), lds, st, a); }
        else { ensure_dynamic_lds((const void*)attn_causal128_dma_kernel<4>, lds_max, attr_done); hipLaunchKernelGGL(attn_causal128_dma_kernel<4>, grid, dim3(256), lds, st, a); }
      } else {
        hipLaunchKernelGGL(attn_causal128_kernel, dim3((sl.maxL + 127) / 128, l.n_heads, n_seq), dim3(256), 0, st, a);
      }
    }
    gemm(e, st, PC_ENC_GEMM_O, EPI_RESID_F32, sl.ctx, Q, w.o, Q, sl.hidden, dm, T, dm, Q, 0, 0, 1.f, 1, 0, 0, 0, false, prod);
    rowscale(e, st, sl.ssq, sl.rowscale, T);
    gemm(e, st, PC_ENC_GEMM_FFN_IN, EPI_SWIGLU_F16, sl.xraw, dm, w.gu_f, dm, sl.ffh, F, T, 2 * F, dm, 0, 0, 1.f, 1, 0, 0, 0, false, cons);
    const bool lastl = i + 1 == l.n_layers;
    gemm(e, st, PC_ENC_GEMM_FFN_OUT, EPI_RESID_F32, sl.ffh, F, w.down, F, sl.hidden, dm, T, dm, F, 0, 0, 1.f, 1, 0, 0, 0, false, lastl ? GemmFold() : prod);
    if (!lastl) rowscale(e, st, sl.ssq, sl.rowscale, T);
  }
  rmsnorm(e, st, sl.hidden, e->l_final_ln, sl.dlast, sl.d_last_rows, n_seq);
  HIPCHK(e, hipGetLastError());
  return RK_OK;
}

int rk_llama_last_logits(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq,
                         const int32_t* out_token_ids, int n_out, float* out_logits) {
  if (!e || !out_logits) return RK_ERR_INVALID;
  if (!out_token_ids || n_out <= 0 || n_out > 64) return fail(e, RK_ERR_INVALID, "n_out must be in 1..64 (got %d)", n_out);
  int rc = check_ids(e, out_token_ids, n_out, "output");
  if (rc) return rc;
  if ((rc = llama_prefill(e, tokens, seq_offsets, n_seq))) return rc;
  Slot& sl = e->slots[0];
  hipStream_t st = sl.se;
  HIPCHK(e, hipMemcpyAsync(sl.d_out_ids, out_token_ids, n_out * sizeof(int), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(head_rows_kernel, dim3((n_seq * n_out + 3) / 4), dim3(256), 0, st, sl.dlast, e->lm_head, sl.d_out_ids,
                     sl.d_scores, n_seq, n_out, e->ld.hidden);
  HIPCHK(e, hipMemcpyAsync(sl.h_scores, sl.d_scores, (size_t)n_seq * n_out * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(e, hipStreamSynchronize(st));
  HIPCHK(e, hipGetLastError());
  memcpy(out_logits, sl.h_scores, (size_t)n_seq * n_out * sizeof(float));
  return RK_OK;
}

int rk_llama_set_rope_scaling(rk_engine* e, float factor, float low_freq_factor, float high_freq_factor, int original_max_pos) {
  if (!e) return RK_ERR_INVALID;
  if (e->family != 1) return fail(e, RK_ERR_STATE, "rope scaling applies to Llama engines (rk_llama_create)");
  if (e->finalized) return fail(e, RK_ERR_STATE, "rk_llama_set_rope_scaling must precede rk_engine_finalize (the rotary tables are built there)");
  if (!(factor > 0.f) || !(high_freq_factor > low_freq_factor) || !(low_freq_factor > 0.f) || original_max_pos <= 0)
    return fail(e, RK_ERR_INVALID, "bad llama3 rope scaling (factor %g, low %g, high %g, original_max_position_embeddings %d)", factor, low_freq_factor, high_freq_factor, original_max_pos);
  e->rope_factor = factor; e->rope_low = low_freq_factor; e->rope_high = high_freq_factor; e->rope_orig = original_max_pos;
  return RK_OK;
}

int rk_llama_greedy1(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq, int32_t* out_tokens) {
  if (!e || !out_tokens) return RK_ERR_INVALID;
  int rc = llama_prefill(e, tokens, seq_offsets, n_seq);
  if (rc) return rc;
  if ((rc = ensure_amax(e, (size_t)n_seq, e->ld.vocab))) return rc;
  Slot& sl = e->slots[0];
  hipStream_t st = sl.se;
  // full-vocabulary head on the n_seq last rows: weight-streaming GEMM, then the first arg-max (torch.argmax tie rule)
  head_argmax(e, st, sl.dlast, n_seq, e->ld.hidden, e->ld.vocab, sl.d_argmax);
  std::vector<int> amax(n_seq);
  HIPCHK(e, hipMemcpyAsync(amax.data(), sl.d_argmax, n_seq * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(e, hipStreamSynchronize(st));
  HIPCHK(e, hipGetLastError());
  for (int b = 0; b < n_seq; ++b) out_tokens[b] = amax[b];
  return RK_OK;
}

// ---- K9: score collection across the GPUs of a node, RCCL over xGMI, straight from the slot's device score buffer --
int rk_comm_unique_id(uint8_t* out_id, int n_bytes) {
  if (!out_id || n_bytes != RK_COMM_ID_BYTES) return fail(nullptr, RK_ERR_INVALID, "unique id buffer must be %d bytes", RK_COMM_ID_BYTES);
  static_assert(RK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rk_engine.h and rccl.h disagree on the id size");
  const RcclApi* r = rccl_api();
  if (!r) return fail(nullptr, RK_ERR_HIP, "%s", g_rccl.err.c_str());
  ncclUniqueId id;
  const ncclResult_t rc = r->GetUniqueId(&id);
  if (rc != ncclSuccess) return fail(nullptr, RK_ERR_HIP, "ncclGetUniqueId: %s", r->GetErrorString(rc));
  memcpy(out_id, id.internal, RK_COMM_ID_BYTES);
  return RK_OK;
}

int rk_comm_init(rk_engine* e, const uint8_t* id_bytes, int n_bytes, int rank, int world, int max_floats_per_rank) {
  if (!e || !id_bytes || n_bytes != RK_COMM_ID_BYTES) return fail(e, RK_ERR_INVALID, "bad unique id");
  if (world < 1 || rank < 0 || rank >= world || max_floats_per_rank <= 0) return fail(e, RK_ERR_INVALID, "bad rank %d / world %d / capacity %d", rank, world, max_floats_per_rank);
  if (!e->finalized) return fail(e, RK_ERR_STATE, "engine not finalized");
  int rc = set_device(e);
  if (rc) return rc;
  const RcclApi* r = rccl_api();
  if (!r) return fail(e, RK_ERR_HIP, "%s", g_rccl.err.c_str());
  comm_release(e);
  ncclUniqueId id;
  memcpy(id.internal, id_bytes, RK_COMM_ID_BYTES);
  const ncclResult_t nrc = r->CommInitRank(&e->comm, world, id, rank);   // collective over all ranks
  if (nrc != ncclSuccess) { e->comm = nullptr; return fail(e, RK_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, r->GetErrorString(nrc)); }
  e->comm_rank = rank; e->comm_world = world; e->gather_cap = (size_t)max_floats_per_rank;
  if ((rc = comm_alloc_buffers(e))) {     // a half-built communicator must not report a capacity: tear it down, keep the message
    const std::string why = e->err;
    comm_release(e);
    e->err = why;
    return rc;
  }
  return RK_OK;
}

int rk_comm_world(const rk_engine* e, int* out_rank, int* out_world) {
  if (!e) return RK_ERR_INVALID;
  if (out_rank) *out_rank = e->comm ? e->comm_rank : 0;
  if (out_world) *out_world = e->comm ? e->comm_world : 1;
  return RK_OK;
}

int rk_comm_capacity(const rk_engine* e) { return (e && e->comm) ? (int)e->gather_cap : 0; }

int rk_comm_library_info(char* buf, int n_bytes) {
  if (!buf || n_bytes <= 1) return RK_ERR_INVALID;
  const RcclApi* r = rccl_api();
  if (!r) return fail(nullptr, RK_ERR_HIP, "%s", g_rccl.err.c_str());
  Dl_info info{};
  const char* path = (dladdr((void*)r->AllGather, &info) && info.dli_fname) ? info.dli_fname : "?";
  int ver = 0;
  if (r->GetVersion) r->GetVersion(&ver);
  const int n = snprintf(buf, (size_t)n_bytes, "%s|%d", path, ver);
  return n < n_bytes ? n : n_bytes - 1;
}

int rk_comm_all_gather_slot(rk_engine* e, int slot, int n_floats) {
  if (!e || slot < 0 || slot >= RK_SLOTS) return RK_ERR_INVALID;
  if (!e->comm) return fail(e, RK_ERR_STATE, "rk_comm_init has not been called");
  if (n_floats <= 0 || (size_t)n_floats > e->gather_cap || (size_t)n_floats > e->scores_cap) return fail(e, RK_ERR_CAPACITY, "n_floats %d out of range (capacity %zu, score buffer %zu)", n_floats, e->gather_cap, e->scores_cap);
  int rc = set_device(e);
  if (rc) return rc;
  Slot& sl = e->slots[slot];
  hipStream_t sd = dec_stream(e, sl);   // the stream the slot's scores are produced on: the gather simply follows them
  if (e->gather_pending[slot]) { HIPCHK(e, hipEventSynchronize(e->ev_gather[slot])); e->gather_pending[slot] = false; }
  const ncclResult_t nrc = rccl_api()->AllGather(sl.d_scores, e->d_gather[slot], (size_t)n_floats, ncclFloat, e->comm, sd);
  if (nrc != ncclSuccess) return fail(e, RK_ERR_HIP, "ncclAllGather: %s", rccl_api()->GetErrorString(nrc));
  HIPCHK(e, hipMemcpyAsync(e->h_gather[slot], e->d_gather[slot], (size_t)n_floats * e->comm_world * sizeof(float), hipMemcpyDeviceToHost, sd));
  HIPCHK(e, hipEventRecord(e->ev_gather[slot], sd));
  e->gather_pending[slot] = true; e->gather_n[slot] = n_floats;
  return mark_decoder_done(e, sl);      // the slot's buffers stay busy until the gather has read them
}

int rk_comm_read_gathered_slot(rk_engine* e, int slot, float* out, int n_floats_total) {
  if (!e || !out || slot < 0 || slot >= RK_SLOTS) return RK_ERR_INVALID;
  if (!e->comm) return fail(e, RK_ERR_STATE, "rk_comm_init has not been called");
  if (n_floats_total != e->gather_n[slot] * e->comm_world) return fail(e, RK_ERR_INVALID, "asked for %d floats, the last gather of slot %d holds %d", n_floats_total, slot, e->gather_n[slot] * e->comm_world);
  if (e->gather_pending[slot]) { HIPCHK(e, hipEventSynchronize(e->ev_gather[slot])); e->gather_pending[slot] = false; }
  memcpy(out, e->h_gather[slot], (size_t)n_floats_total * sizeof(float));
  return RK_OK;
}

// Appended form: a rank whose share of a query's candidates needs several engine calls (more sequences or tokens than one
// call holds) copies each call's scores behind the ones before - device to device, on the stream that produced them - and
// ONE all_gather ships the whole share.  Every rank issues exactly one collective per query whatever its chunk count.
int rk_comm_append_scores_slot(rk_engine* e, int slot, int n_floats, int dst_offset) {
  if (!e || slot < 0 || slot >= RK_SLOTS) return RK_ERR_INVALID;
  if (!e->comm) return fail(e, RK_ERR_STATE, "rk_comm_init has not been called");
  if (n_floats < 0 || dst_offset < 0 || (size_t)n_floats + (size_t)dst_offset > e->gather_cap || (size_t)n_floats > e->scores_cap)
    return fail(e, RK_ERR_CAPACITY, "append of %d floats at %d exceeds the send buffer (%zu) or the score buffer (%zu)", n_floats, dst_offset, e->gather_cap, e->scores_cap);
  int rc = set_device(e);
  if (rc) return rc;
  if (n_floats == 0) return RK_OK;
  Slot& sl = e->slots[slot];
  hipStream_t sd = dec_stream(e, sl);
  // the previous gather still reads the send buffer until its event has passed
  if (e->gall_pending) { HIPCHK(e, hipEventSynchronize(e->ev_gall)); e->gall_pending = false; }
  HIPCHK(e, hipMemcpyAsync(e->d_gsend + dst_offset, sl.d_scores, (size_t)n_floats * sizeof(float), hipMemcpyDeviceToDevice, sd));
  if (sd != dec_stream(e, e->slots[0])) { HIPCHK(e, hipEventRecord(e->ev_append, sd)); e->append_foreign = true; }
  return mark_decoder_done(e, sl);      // the slot's score buffer stays busy until the copy has read it
}

int rk_comm_append_host(rk_engine* e, const float* values, int n_floats, int dst_offset) {
  if (!e || (!values && n_floats > 0)) return RK_ERR_INVALID;
  if (!e->comm) return fail(e, RK_ERR_STATE, "rk_comm_init has not been called");
  if (n_floats < 0 || dst_offset < 0 || (size_t)n_floats + (size_t)dst_offset > e->gather_cap)
    return fail(e, RK_ERR_CAPACITY, "append of %d host floats at %d exceeds the send buffer (%zu)", n_floats, dst_offset, e->gather_cap);
  int rc = set_device(e);
  if (rc) return rc;
  if (n_floats == 0) return RK_OK;
  // the previous gather still reads the send buffer (and its staging copy may be in flight) until its event has passed
  if (e->gall_pending) { HIPCHK(e, hipEventSynchronize(e->ev_gall)); e->gall_pending = false; }
  hipStream_t s0 = dec_stream(e, e->slots[0]);     // the stream rk_comm_all_gather_appended runs on: the copy precedes it in order
  // one staging region per destination range: regions of one query do not overlap, the next query starts behind the gather's event
  memcpy(e->h_gstage + dst_offset, values, (size_t)n_floats * sizeof(float));
  HIPCHK(e, hipMemcpyAsync(e->d_gsend + dst_offset, e->h_gstage + dst_offset, (size_t)n_floats * sizeof(float), hipMemcpyHostToDevice, s0));
  return RK_OK;
}

int rk_comm_all_gather_appended(rk_engine* e, int n_floats) {
  if (!e) return RK_ERR_INVALID;
  if (!e->comm) return fail(e, RK_ERR_STATE, "rk_comm_init has not been called");
  if (n_floats <= 0 || (size_t)n_floats > e->gather_cap) return fail(e, RK_ERR_CAPACITY, "n_floats %d out of range (capacity %zu)", n_floats, e->gather_cap);
  int rc = set_device(e);
  if (rc) return rc;
  hipStream_t s0 = dec_stream(e, e->slots[0]);
  if (e->gall_pending) { HIPCHK(e, hipEventSynchronize(e->ev_gall)); e->gall_pending = false; }
  if (e->append_foreign) { HIPCHK(e, hipStreamWaitEvent(s0, e->ev_append, 0)); e->append_foreign = false; }
  const ncclResult_t nrc = rccl_api()->AllGather(e->d_gsend, e->d_gall, (size_t)n_floats, ncclFloat, e->comm, s0);
  if (nrc != ncclSuccess) return fail(e, RK_ERR_HIP, "ncclAllGather: %s", rccl_api()->GetErrorString(nrc));
  HIPCHK(e, hipMemcpyAsync(e->h_gall, e->d_gall, (size_t)n_floats * e->comm_world * sizeof(float), hipMemcpyDeviceToHost, s0));
  HIPCHK(e, hipEventRecord(e->ev_gall, s0));
  e->gall_pending = true; e->gall_n = n_floats;
  return RK_OK;
}

int rk_comm_read_appended(rk_engine* e, float* out, int n_floats_total) {
  if (!e || !out) return RK_ERR_INVALID;
  if (!e->comm) return fail(e, RK_ERR_STATE, "rk_comm_init has not been called");
  if (n_floats_total != e->gall_n * e->comm_world) return fail(e, RK_ERR_INVALID, "asked for %d floats, the last appended gather holds %d", n_floats_total, e->gall_n * e->comm_world);
  if (e->gall_pending) { HIPCHK(e, hipEventSynchronize(e->ev_gall)); e->gall_pending = false; }
  memcpy(out, e->h_gall, (size_t)n_floats_total * sizeof(float));
  return RK_OK;
}

int rk_comm_destroy(rk_engine* e) {
  if (!e) return RK_ERR_INVALID;
  if (set_device(e) == RK_OK) sync_all(e);
  comm_release(e);
  return RK_OK;
}

int rk_timer_begin(rk_engine* e) {
  if (!e) return RK_ERR_INVALID;
  int rc = set_device(e);
  if (rc) return rc;
  // callers synchronise before a timed region; make every other stream start behind the start event anyway
  hipStream_t s0 = e->slots[0].se;
  HIPCHK(e, hipEventRecord(e->t0, s0));
  for (Slot& sl : e->slots) {
    if (sl.se != s0) HIPCHK(e, hipStreamWaitEvent(sl.se, e->t0, 0));
    HIPCHK(e, hipStreamWaitEvent(sl.sd, e->t0, 0));
  }
  return RK_OK;
}

int rk_timer_end(rk_engine* e, float* out_ms) {
  if (!e || !out_ms) return RK_ERR_INVALID;
  int rc = set_device(e);
  if (rc) return rc;
  // the stop event must follow the work of ALL streams: chain them into slot 0's encoder stream
  hipStream_t s0 = e->slots[0].se;
  for (Slot& sl : e->slots) {
    if (sl.se != s0) { HIPCHK(e, hipEventRecord(e->t_tmp, sl.se)); HIPCHK(e, hipStreamWaitEvent(s0, e->t_tmp, 0)); }
    HIPCHK(e, hipEventRecord(e->t_tmp, sl.sd)); HIPCHK(e, hipStreamWaitEvent(s0, e->t_tmp, 0));
  }
  HIPCHK(e, hipEventRecord(e->t1, s0));
  HIPCHK(e, hipEventSynchronize(e->t1));
  HIPCHK(e, hipEventElapsedTime(out_ms, e->t0, e->t1));
  return RK_OK;
}

int rk_profile_enable(rk_engine* e, int on) { if (!e) return RK_ERR_INVALID; e->prof_on = on != 0; return RK_OK; }

int rk_profile_reset(rk_engine* e) {
  if (!e) return RK_ERR_INVALID;
  int rc = set_device(e);
  if (rc) return rc;
  if ((rc = sync_all(e))) return rc;
  e->prof_used = 0;
  for (int c = 0; c < PC_COUNT; ++c) { e->prof_flops[c] = 0; e->prof_bytes[c] = 0; e->prof_n[c] = 0; }
  return RK_OK;
}

int rk_profile_get(rk_engine* e, int cls, double* total_ms, int64_t* launches, double* flops, double* bytes) {
  if (!e || cls < 0 || cls >= PC_COUNT) return RK_ERR_INVALID;
  int rc = set_device(e);
  if (rc) return rc;
  if ((rc = sync_all(e))) return rc;
  double ms = 0;
  for (size_t i = 0; i < e->prof_used; ++i) {
    if (e->prof_recs[i].cls != cls) continue;
    float t = 0;
    HIPCHK(e, hipEventElapsedTime(&t, e->prof_recs[i].a, e->prof_recs[i].b));
    ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = e->prof_n[cls];
  if (flops) *flops = e->prof_flops[cls];
  if (bytes) *bytes = e->prof_bytes[cls];
  return RK_OK;
}

}  // extern "C" (reopened behind the option table)

// Option table: key -> field, the values it accepts (lo..hi, or the listed set), meaning.  A value outside the range is an error
// (RK_ERR_INVALID), as include/rk_engine.h promises - a sweep can never record a value that was not applied.
namespace {
struct OptionDesc { const char* key; int rk_engine::Options::*field; int lo, hi; const char* allowed; const char* what; };
const OptionDesc kOptions[] = {
  {"dec_graph", &rk_engine::Options::dec_graph, 0, 1, nullptr, "decoder chains replayed as HIP graphs (1) or launched eagerly (0)"},
  {"gemm_glds", &rk_engine::Options::glds, 0, 1, nullptr, "128x128 GEMM staging by LDS-DMA (1) or through registers (0); same bits"},
  {"gemm_skinny", &rk_engine::Options::skinny, 0, 0x3F, nullptr, "bit per epilogue kind: few-row GEMMs on the weight-streaming kernel (1 = all)"},
  {"dec_ffn_tiled", &rk_engine::Options::dec_ffn_tiled, 0, 1, nullptr, "one-position decoder: FFN-in on the tiled kernels (1) or the weight-streaming kernel (0)"},
  {"gemm_persistent", &rk_engine::Options::gemm_persistent, 0, 1024, nullptr, "ping-pong GEMM: 1 = one workgroup per CU walks the tiles, 0 = one per tile, n > 1 = n workgroups"},
  {"gemm_s64_stages", &rk_engine::Options::s64_stages, 0, 4, "0,2,3,4", "LDS stages of the 64x64 GEMM (0 = from the tile count); same bits"},
  {"consumer_stats", &rk_engine::Options::consumer_stats, 0, 1, nullptr, "encoder row factors formed by the non-persistent consumer GEMMs themselves (1) or always by rowscale_kernel (0); same bits"},
  {"greedy_spec", &rk_engine::Options::greedy_spec, 0, 65536, nullptr, "rk_t5_greedy2: most decoder rows of a speculative pass; 0 = never speculate; same tokens"},
  {"dec_fold_norm", &rk_engine::Options::dec_fold_norm, 0, 1, nullptr, "decoder RMSNorms folded into the weight-streaming GEMMs (1) or separate kernels (0)"},
  {"fold_norm", &rk_engine::Options::fold_norm, 0, 1, nullptr, "encoder RMSNorm folded into the GEMMs (1) or separate kernels (0)"},
  {"xattn_mfma", &rk_engine::Options::xattn_mfma, 0, 1, nullptr, "query-side cross-attention: weighted sums on the matrix cores (1) or the VALU form (0)"},
  {"attn_heads_per_wg", &rk_engine::Options::attn_heads_per_wg, 0, 4096, nullptr, "short-sequence attention: (sequence, head) items per wave group, 0 = dealt evenly; same bits"},
  {"xattn_direct", &rk_engine::Options::xattn_direct, 0, 1, nullptr, "decoder prefixes <= 16: query-side cross-attention (1) or materialised K / V (0)"},
  {"attn_short", &rk_engine::Options::attn_short, 0, 6, "0,5,6", "sequences <= 192 keys: DMA kernel with two (5) / one (6) wave group per workgroup, or the tiled kernel (0); same bits"},
  {"gemm_variant", &rk_engine::Options::gemm_variant, 0, 120, nullptr, "tile variant: 0 auto, 1..6 see choose_variant; measurement builds: 80+ / 100+ knock-outs"},
  {"dec_fuse_rows", &rk_engine::Options::dec_fuse_rows, 0, 32, nullptr, "rows per workgroup of dec_cross_qk_kernel (0 = auto); same bits"},
  {"dec_fuse", &rk_engine::Options::dec_fuse, 0, 2, nullptr, "few-row decoder: projections around the query-side cross-attention fused at one position (1), always (2), never (0)"},
  {"llama_attn_nw", &rk_engine::Options::llama_attn_nw, 0, 8, "0,4,8", "waves per workgroup of the Llama LDS-DMA attention kernel (0 = default 8); same bits"},
  {"llama_attn_dma", &rk_engine::Options::llama_attn_dma, 0, 1, nullptr, "Llama causal attention: LDS-DMA kernel (1) or the register-staged first kernel (0); differ within fp16 noise"},
  {"attn_long_xcd", &rk_engine::Options::attn_long_xcd, 0, 1, nullptr, "long-sequence attention: workgroups of a (sequence, head) pair on one XCD (1) or dealt over all eight (0); same bits"},
  {"attn_long_nw", &rk_engine::Options::attn_long_nw, 0, 12, "0,3,4,6,12", "waves per workgroup of the long-sequence attention kernel (0 = default 4); same bits"},
  {"attn_long", &rk_engine::Options::attn_long, 0, 1, nullptr, "sequences > 192 keys: the chunked LDS-DMA kernel (1) or the tiled kernel (0)"},
  {"dec_gemv", &rk_engine::Options::dec_gemv, 0, 1, nullptr, "decoder pass of at most 16 rows at >= 2 positions (one setwise compare): plain projections on the wave-per-column GEMV kernel (1) or the weight-streaming MFMA kernel (0); differ within fp32 summation-order noise"},
  {"dec_gemv_rows", &rk_engine::Options::dec_gemv_rows, 1, GEMV_MAX_ROWS, nullptr, "largest row count of a decoder pass that takes the few-row GEMV family (default 4 = the measured cross-over; the kernel takes up to 16)"},
  {"dec_cross_mfma", &rk_engine::Options::dec_cross_mfma, 0, 1, nullptr, "long decoder prefixes (qlm), cross-attention over the materialised K / V: matrix-core kernel for sequences <= 192 keys (1) or the staged fma-chain kernels (0); differ within fp16 noise"},
  {"dec_attn_seq", &rk_engine::Options::dec_attn_seq, 0, 1, nullptr, "decoder attention at several positions: one workgroup per (head, sequence) (1) or per query row (0); same bits"},
  {"gemm_sk", &rk_engine::Options::gemm_sk, 0, 2, nullptr, "ping-pong GEMM, fp32 residual projections with few tiles and a long K: K split over two workgroups (1: choose_ksplit), never (0), wherever it fits (2: tests)"},
  {"gemm_split", &rk_engine::Options::gemm_split, 0, 1, nullptr, "rows beyond the ping-pong kernel's last whole round on a fill-in tile variant (1) or one launch (0); same bits"},
#ifdef RK_MEASURE
  {"attn_ko", &rk_engine::Options::attn_ko, 0, 1 << 20, nullptr, "timing-only knock-outs of the attention kernels (measurement builds)"},
#endif
};
bool option_value_ok(const OptionDesc& o, int value) {
  if (value < o.lo || value > o.hi) return false;
  if (!o.allowed) return true;
  for (const char* p = o.allowed; *p;) {
    if (atoi(p) == value) return true;
    while (*p && *p != ',') ++p;
    if (*p == ',') ++p;
  }
  return false;
}
}  // namespace

extern "C" {

int rk_engine_set_option(rk_engine* e, const char* key, int value) {
  if (!e || !key) return RK_ERR_INVALID;
  if (!strcmp(key, "overlap")) {    // 1: decoder chain on its own stream (default); 0: everything on one stream
    if (value < 0 || value > 1) return fail(e, RK_ERR_INVALID, "option overlap: 0..1");
    if (set_device(e) || sync_all(e)) return RK_ERR_HIP;
    for (Slot& sl : e->slots) sl.dec_pending = false;
    e->opt.overlap = value;
    e->opt_epoch++;
    return RK_OK;
  }
#ifdef RK_MEASURE
  if (!strcmp(key, "attn_trace")) {   // phase time stamps of one workgroup of the DMA attention kernel (attention.h: ATTD_STAMP)
    if (value && !e->attn_trace) { if (hipMalloc(&e->attn_trace, 12 * 16 * 16 * sizeof(float)) != hipSuccess) return RK_ERR_HIP; hipMemset(e->attn_trace, 0, 12 * 16 * 16 * sizeof(float)); }
    if (!value && e->attn_trace) { hipFree(e->attn_trace); e->attn_trace = nullptr; }
    return RK_OK;
  }
#endif
  for (const OptionDesc& o : kOptions) {
    if (strcmp(key, o.key)) continue;
    if (!strcmp(key, "gemm_skinny") && value == 1) value = 0x3F;
    if (!option_value_ok(o, value))
      return fail(e, RK_ERR_INVALID, "option %s: value %d outside %d..%d%s%s (%s)", key, value, o.lo, o.hi, o.allowed ? ", allowed: " : "", o.allowed ? o.allowed : "", o.what);
    e->opt.*(o.field) = value;
    e->opt_epoch++;                                           // cached decoder graphs were captured under the old options
    return RK_OK;
  }
  return fail(e, RK_ERR_INVALID, "unknown option %s", key);
}

int rk_debug_gemm(rk_engine* e, const uint16_t* A, const uint16_t* W, float* C, int M, int N, int K, int use_glds) {
  if (!e || !A || !W || !C) return RK_ERR_INVALID;
  int rc = set_device(e);
  if (rc) return rc;
  if (K % 64 || N % 4) return fail(e, RK_ERR_INVALID, "debug gemm needs K%%64==0 and N%%4==0");
  half_t *dA = nullptr, *dW = nullptr; float* dC = nullptr;
  HIPCHK(e, hipMalloc((void**)&dA, (size_t)M * K * 2)); HIPCHK(e, hipMalloc((void**)&dW, (size_t)N * K * 2));
  HIPCHK(e, hipMalloc((void**)&dC, (size_t)M * N * 4));
  HIPCHK(e, hipMemcpy(dA, A, (size_t)M * K * 2, hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(dW, W, (size_t)N * K * 2, hipMemcpyHostToDevice));
  const int saved = e->opt.glds;
  e->opt.glds = use_glds != 0;
  GemmFold dbg_fold; dbg_fold.few = use_glds == 3;                 // 3: the few-row GEMV family (M <= 16)
  gemm(e, e->slots[0].se, PC_OTHER, EPI_STORE_F32, dA, K, dW, K, dC, N, M, N, K, 0, 0, 1.f, 1, 0, 0, 0, /*weight_streaming=*/use_glds >= 2, dbg_fold);
  e->opt.glds = saved;
  HIPCHK(e, hipStreamSynchronize(e->slots[0].se));
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
  hipFree(dA); hipFree(dW); hipFree(dC);
  return RK_OK;
}

// debug/measurement: time `iters` back-to-back launches of the engine's GEMM at one shape (random fp16 operands on
// the device, epilogue `epi` as in GemmEpi).  *out_ms = average per launch from HIP events on the encoder stream.
int rk_debug_gemm_bench(rk_engine* e, int M, int N, int K, int epi, int iters, float* out_ms) {
  if (!e || !out_ms || M <= 0 || N <= 0 || K <= 0 || iters <= 0) return RK_ERR_INVALID;
  int rc = set_device(e);
  if (rc) return rc;
  if (K % 64 || N % 4) return fail(e, RK_ERR_INVALID, "gemm bench needs K%%64==0 and N%%4==0");
  // (RK_BENCH_PAD: extra halfs per operand row - a leading dimension that is not a power of two; measurement of the L2 channel spread)
  const char* pad_env = getenv("RK_BENCH_PAD");
  const int pad = pad_env ? atoi(pad_env) : 0;
  const int ldk = K + pad;
  const size_t na = (size_t)M * ldk, nw = (size_t)N * ldk, nc = (size_t)M * N;
  half_t *dA = nullptr, *dW = nullptr; void* dC = nullptr;
  HIPCHK(e, hipMalloc((void**)&dA, na * 2)); HIPCHK(e, hipMalloc((void**)&dW, nw * 2));
  HIPCHK(e, hipMalloc(&dC, nc * 4));
  {
    std::vector<half_t> h(std::max(na, nw));
    uint32_t x = 12345u;
    for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; h[i] = (half_t)(((int)(x >> 16) % 2001 - 1000) * 1e-3f); }
    HIPCHK(e, hipMemcpy(dA, h.data(), na * 2, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(dW, h.data(), nw * 2, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemset(dC, 0, nc * 4));
  }
  const int ldc = EPI_IS_GATED(epi) ? N / 2 : N;
  // (RK_BENCH_FOLD: the residual epilogue as the encoder runs it - producer side of the folded RMSNorm: fp16 stream copy + sums of squares)
  GemmFold fold;
  half_t* dX = nullptr; float* dS = nullptr;
  if (getenv("RK_BENCH_FOLD") && atoi(getenv("RK_BENCH_FOLD")) && epi == EPI_RESID_F32) {
    HIPCHK(e, hipMalloc((void**)&dX, nc * 2)); HIPCHK(e, hipMalloc((void**)&dS, (size_t)M * ((N + 31) / 32) * 4));
    fold.xraw = dX; fold.ssq = dS;
  }
  for (int i = 0; i < 2; ++i) gemm(e, e->slots[0].se, PC_OTHER, epi, dA, ldk, dW, ldk, dC, ldc, M, N, K, 0, 0, 1.f, 1, 0, 0, 0, false, fold);
  HIPCHK(e, hipEventRecord(e->t0, e->slots[0].se));
  for (int i = 0; i < iters; ++i) gemm(e, e->slots[0].se, PC_OTHER, epi, dA, ldk, dW, ldk, dC, ldc, M, N, K, 0, 0, 1.f, 1, 0, 0, 0, false, fold);
  HIPCHK(e, hipEventRecord(e->t1, e->slots[0].se));
  HIPCHK(e, hipEventSynchronize(e->t1));
  float ms = 0;
  HIPCHK(e, hipEventElapsedTime(&ms, e->t0, e->t1));
  HIPCHK(e, hipGetLastError());
  *out_ms = ms / iters;
  hipFree(dA); hipFree(dW); hipFree(dC); if (dX) hipFree(dX); if (dS) hipFree(dS);
  return RK_OK;
}

int64_t rk_debug_read(rk_engine* e, const char* name, float* out, int64_t max_floats) {
  if (!e || !name || !out) return RK_ERR_INVALID;
  if (set_device(e)) return RK_ERR_HIP;
  if (sync_all(e)) return RK_ERR_HIP;
  if (!strcmp(name, "occupancy")) {   // resident workgroups per CU the runtime computes for the main kernels
    if (max_floats < 6) return RK_ERR_INVALID;
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_enc_kernel, 256, 0); out[0] = (float)n;
    out[1] = 0.f;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_f16_kernel<EPI_STORE_F16, true>, 256, GEMM_LDS_BYTES); out[2] = (float)n;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_pp2_kernel<EPI_STORE_F16, 0>, 512, 163840); out[3] = (float)n;
    out[4] = 0.f;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rmsnorm_kernel<4>, 256, 0); out[5] = (float)n;
    if (max_floats >= 10) {
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_enc_dma_kernel<1>, 384, ATTD_LDS_BYTES); out[6] = (float)n;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_enc_dma_kernel<2>, 768, 2 * ATTD_LDS_BYTES); out[7] = (float)n;
      out[8] = out[9] = 0.f;
      return 10;
    }
    return 6;
  }
  const Slot& sl = e->slots[0];
  const std::string n(name);
  const int I = e->inner, dm = e->d.d_model;
  const void* src = nullptr; int64_t cnt = 0; bool is_half = true;
  if (n == "enc_hidden") { src = sl.hidden; cnt = (int64_t)sl.T * dm; is_half = false; }
#ifdef RK_MEASURE
  else if (n == "attn_trace" && e->attn_trace) { src = e->attn_trace; cnt = 12 * 16 * 16; is_half = false; }
#endif
  else if (n == "enc_out") { src = sl.enc_out; cnt = (int64_t)sl.T * dm; }
  else if (n == "qkv") { src = sl.qkv; cnt = (int64_t)sl.T * 3 * I; }
  else if (n == "ctx") { src = sl.ctx; cnt = (int64_t)sl.T * I; }
  else if (n == "xn") { src = sl.xn; cnt = (int64_t)sl.T * dm; }
  else if (n == "dec_hidden") { src = sl.dhidden; cnt = (int64_t)sl.n_seq * e->d.max_dec_len * dm; is_half = false; }
  else return fail(e, RK_ERR_INVALID, "unknown buffer %s", name);
  cnt = std::min(cnt, max_floats);
  if (is_half) {
    std::vector<half_t> tmp(cnt);
    if (hipMemcpy(tmp.data(), src, cnt * 2, hipMemcpyDeviceToHost) != hipSuccess) return fail(e, RK_ERR_HIP, "copy failed");
    for (int64_t i = 0; i < cnt; ++i) out[i] = (float)tmp[i];
  } else if (hipMemcpy(out, src, cnt * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(e, RK_ERR_HIP, "copy failed");
  return cnt;
}

}  // extern "C"
