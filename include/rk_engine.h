/* rk_engine.h — C ABI of the MI355X-native reranking engine (librk_engine.so, gfx950 only).
 *
 * This is the drop-in boundary for ONE hot path of ielab/llm-rankers: the batched T5 encoder-decoder forward
 * behind PointwiseLlmRanker.rerank and SetwiseLlmRanker.compare.  The reference has no FFI of its own; the
 * inner boundary these entry points replace is exactly three HuggingFace calls plus two attributes
 * (SURVEY.md section 8b):
 *
 *   self.llm(input_ids, attention_mask, decoder_input_ids=...).logits   ref: llmrankers/pointwise.py:117-119,
 *                                                                             llmrankers/setwise.py:184
 *   self.llm(input_ids, attention_mask, labels=...).logits               ref: llmrankers/pointwise.py:73-75
 *   self.llm.generate(input_ids, decoder_input_ids=..., max_new_tokens=2) ref: llmrankers/setwise.py:93-95,128-130
 *   T5ForConditionalGeneration.from_pretrained(...)                       ref: llmrankers/pointwise.py:20-24
 *
 * Conventions: plain C, every function returns 0 on success or a negative rk_status; rk_last_error() gives the
 * message of the last failure on that engine (or of the last failed rk_engine_create when engine == NULL).
 * No exceptions, Python objects or torch types cross this boundary.  Inputs are RAGGED: the B token sequences
 * of a batch are concatenated without padding, seq_offsets[B+1] gives the boundaries (the reference right-pads
 * to the batch's longest sequence and masks; results are identical).  The caller owns all host
 * buffers; the engine owns device memory and one HIP stream.  One engine per device; calls on one engine must
 * be serialised by the caller; different engines may be driven from different threads.
 * There is NO CPU fallback: with no usable gfx950 device rk_engine_create fails with RK_ERR_NO_DEVICE.
 */
#ifndef RK_ENGINE_H
#define RK_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rk_engine rk_engine;

typedef enum rk_status {
  RK_OK = 0,
  RK_ERR_INVALID = -1,    /* bad argument / unsupported model dimensions */
  RK_ERR_NO_DEVICE = -2,  /* no gfx950 GPU visible — the engine never falls back to the CPU */
  RK_ERR_HIP = -3,        /* a HIP runtime call failed */
  RK_ERR_STATE = -4,      /* call order violated (e.g. score before finalize) */
  RK_ERR_MISSING = -5,    /* finalize: a required tensor was never loaded */
  RK_ERR_CAPACITY = -6    /* batch exceeds max_tokens / max_seqs / max_dec_len of the engine */
} rk_status;

typedef enum rk_dtype { RK_F32 = 0, RK_F16 = 1, RK_BF16 = 2 } rk_dtype;

/* Mirrors the fields of HF's T5Config the path depends on (hf: models/t5/configuration_t5.py). */
typedef struct rk_model_desc {
  int32_t vocab, d_model, n_heads, d_kv, d_ff;
  int32_t n_enc_layers, n_dec_layers;
  int32_t n_buckets, max_distance;   /* relative_attention_num_buckets / _max_distance */
  int32_t gated_gelu;                /* 1: feed_forward_proj = "gated-gelu" (T5 v1.1 / flan); 0: "relu" */
  int32_t tied_head;                 /* 1: lm_head = shared and logits scaled by d_model^-0.5 (T5 v1.0, monoT5) */
  float eps;                         /* layer_norm_epsilon */
  int32_t max_tokens;                /* capacity: encoder tokens per call (sum over the batch) */
  int32_t max_seqs;                  /* capacity: sequences per call */
  int32_t max_dec_len;               /* capacity: decoder positions per sequence */
} rk_model_desc;

/* ---- lifetime & weights (replaces from_pretrained, ref: pointwise.py:20-24 / setwise.py:46-50) ---- */
int rk_engine_create(const rk_model_desc* desc, int device_ordinal, rk_engine** out);
void rk_engine_destroy(rk_engine* e);
const char* rk_last_error(const rk_engine* e);
/* One call per HF state-dict entry (names as in the checkpoint, e.g. "encoder.block.0.layer.0.SelfAttention.q.weight").
 * Data is converted to fp16 (the reference's accelerator dtype).  Names the path does not need are ignored. */
int rk_engine_load_tensor(rk_engine* e, const char* hf_name, const void* data, int dtype, const int64_t* shape, int ndim);
/* Repack (fused QKV, interleaved wi_0|wi_1, stacked cross K/V, bias tables), upload, free host copies. */
int rk_engine_finalize(rk_engine* e);

/* ---- blocking calls on host buffers: what the Python rankers use ---- */
/* logits of the LAST decoder position for out_token_ids (n_out > 0) -> out_logits[n_seq][n_out] fp32.
 * dec_prefix is shared by all sequences ([0] for yes_no, [0, "Passage"] for setwise likelihood).
 * replaces: self.llm(input_ids, attention_mask, decoder_input_ids).logits[:, -1, ids]  (pointwise.py:117-121, setwise.py:184-186) */
int rk_t5_score(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq,
                const int32_t* dec_prefix, int dec_len, const int32_t* out_token_ids, int n_out, float* out_logits);
/* out_scores[b] = -sum_t CE(logits[b,t,:], labels[t]) with decoder input = shift_right(labels).
 * replaces: self.llm(input_ids, attention_mask, labels=...).logits + CrossEntropyLoss(reduction="none") (pointwise.py:73-79) */
int rk_t5_qlm(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq,
              const int32_t* labels, int n_labels, float* out_scores);
/* Greedy continuation of dec_prefix by up to max_new tokens -> out_tokens[n_seq][max_new] (pad_id after EOS;
 * stops early when every row has finished, remaining columns = pad_id); *out_steps = decoder steps executed.
 * replaces: self.llm.generate(input_ids, decoder_input_ids=..., max_new_tokens=2)  (setwise.py:93-95, 128-130) */
int rk_t5_greedy(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq,
                 const int32_t* dec_prefix, int dec_len, int max_new, int eos_id, int pad_id,
                 int32_t* out_tokens, int32_t* out_steps);

/* rk_t5_greedy with max_new = 2 and a hint: the first new token is expected to be one of cand_ids[n_cand] (the passage
 * labels of a setwise compare).  Both steps then run in ONE decoder pass - the second speculatively for every candidate -
 * and out_tokens[n_seq][2] / *out_steps are bit-identical to rk_t5_greedy's; any other first token, or a batch that does
 * not fit the workspace, falls back to rk_t5_greedy itself.
 * replaces: self.llm.generate(input_ids, decoder_input_ids=..., max_new_tokens=2)  (setwise.py:113-115) */
int rk_t5_greedy2(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq,
                  const int32_t* dec_prefix, int dec_len, const int32_t* cand_ids, int n_cand, int eos_id, int pad_id,
                  int32_t* out_tokens, int32_t* out_steps);

/* ---- staged / asynchronous form: inputs resident in HBM, used by bench.py and the multi-GPU driver ---- */
int rk_t5_stage(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq);   /* H2D, synchronous */
/* enqueue encoder + decoder + head on the engine stream for the staged batch; scores land in an engine-owned
 * device buffer and are copied to pinned host memory asynchronously. Returns without synchronising. */
int rk_t5_score_staged(rk_engine* e, const int32_t* dec_prefix, int dec_len, const int32_t* out_token_ids, int n_out);
int rk_engine_sync(rk_engine* e);
int rk_t5_read_scores(rk_engine* e, float* out_logits, int n_floats);   /* waits for the batch's decoder, then copies */
/* The same three calls for batch slot `slot` in [0, rk_engine_num_slots()).  Slots are independent batches in flight:
 * the latency-bound decoder chain of one slot runs on its own high-priority HIP stream while the MFMA-bound encoder
 * of the next slot occupies the chip (events order encoder -> decoder per slot).  The un-suffixed calls use slot 0. */
int rk_engine_num_slots(void);
int rk_t5_stage_slot(rk_engine* e, int slot, const int32_t* tokens, const int32_t* seq_offsets, int n_seq);
int rk_t5_score_slot(rk_engine* e, int slot, const int32_t* dec_prefix, int dec_len, const int32_t* out_token_ids, int n_out);
int rk_t5_read_scores_slot(rk_engine* e, int slot, float* out_logits, int n_floats);
/* device address of the fp32 score buffer [n_seq][n_out] of the last rk_t5_score_staged (for RCCL gathers) */
int rk_t5_scores_device_ptr(rk_engine* e, void** out_ptr);

/* ---- decoder-only Llama family: the model call of the reference's setwise ranker for `model_type == 'llama'`
 * (ref: llmrankers/setwise.py:60-69, 159-177): self.llm.generate(input_ids, do_sample=False, max_new_tokens=1) = prefill
 * of the prompt + arg-max of the last position's logits.  hf: models/llama/modeling_llama.py (RMSNorm, RoPE with
 * rope_theta, grouped-query causal attention with head_dim 128, SwiGLU).  Weights go through rk_engine_load_tensor with
 * the HF Llama names ("model.layers.0.self_attn.q_proj.weight", ...) and rk_engine_finalize; rk_engine_destroy frees. */
typedef struct rk_llama_desc {
  int32_t vocab, hidden, n_heads, n_kv_heads, head_dim, intermediate, n_layers;
  int32_t tied_head;                 /* tie_word_embeddings */
  float eps, rope_theta;             /* rms_norm_eps, rope_theta (default rope type) */
  int32_t max_tokens, max_seqs;      /* capacity: prompt tokens per call (sum), prompts per call */
} rk_llama_desc;
int rk_llama_create(const rk_llama_desc* desc, int device_ordinal, rk_engine** out);
/* rope type "llama3" (Llama-3.1 / 3.2 checkpoints; hf: modeling_rope_utils.py _compute_llama3_parameters, reached through
 * AutoModelForCausalLM.from_pretrained at ref: llmrankers/setwise.py:65-69): inverse frequencies whose wavelength exceeds
 * original_max_pos / low_freq_factor are divided by factor, those between original_max_pos / high_freq_factor and that are
 * interpolated.  Call between rk_llama_create and rk_engine_finalize (the rotary tables are built there); never calling
 * it = the default rope type. */
int rk_llama_set_rope_scaling(rk_engine* e, float factor, float low_freq_factor, float high_freq_factor, int original_max_pos);
/* next token of every prompt: first arg-max over the whole vocabulary of the logits at its last position */
int rk_llama_greedy1(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq, int32_t* out_tokens);
/* the same logits for a few vocabulary rows only -> out_logits[n_seq][n_out] fp32 (label scoring, tests) */
int rk_llama_last_logits(rk_engine* e, const int32_t* tokens, const int32_t* seq_offsets, int n_seq,
                         const int32_t* out_token_ids, int n_out, float* out_logits);

/* ---- score collection across the GPUs of one node: one process per GPU, one engine per process.
 * The reference has no counterpart (its multi-GPU mode is accelerate's layer placement, ref: pointwise.py:21); this
 * replaces the torch.distributed round trip of a data-parallel caller.  RCCL (librccl.so.1) is dlopen'ed on first use.
 * rank 0 makes the id, the caller ships its 128 bytes to every rank out of band (torchrun's store, a file, MPI ...),
 * then EVERY rank calls rk_comm_init (collective).  max_floats_per_rank bounds the later gathers. */
#define RK_COMM_ID_BYTES 128
int rk_comm_unique_id(uint8_t* out_id, int n_bytes);
int rk_comm_init(rk_engine* e, const uint8_t* id_bytes, int n_bytes, int rank, int world, int max_floats_per_rank);
int rk_comm_world(const rk_engine* e, int* out_rank, int* out_world);
/* max_floats_per_rank of the live communicator (0 without one): the bound every rank checks BEFORE it enters a gather */
int rk_comm_capacity(const rk_engine* e);
/* which RCCL the process bound (dlopen'ed on first use): "<path of the mapped library>|<ncclGetVersion code>" into buf,
 * NUL-terminated; returns the length written or a negative status.  For logs: the torch wheel bundles its own librccl. */
int rk_comm_library_info(char* buf, int n_bytes);
/* ONE ncclAllGather of the slot's device score buffer (the first n_floats fp32 of what rk_t5_score_slot / rk_t5_qlm
 * left there; every rank passes the same n_floats, ranks with fewer scores are read up to their own count by the
 * caller), enqueued on the stream that produces the scores, followed by an async copy to pinned host memory.
 * Returns without synchronising.  rk_comm_read_gathered_slot waits and copies out[world][n_floats]. */
int rk_comm_all_gather_slot(rk_engine* e, int slot, int n_floats);
int rk_comm_read_gathered_slot(rk_engine* e, int slot, float* out, int n_floats_total);
/* Appended form for a rank whose share of the candidates takes several engine calls (more than max_seqs sequences or
 * max_tokens tokens): after each blocking call (rk_t5_score / rk_t5_qlm: scores in slot 0) or rk_t5_score_slot, copy that
 * call's n_floats scores to offset dst_offset of the engine's send buffer (device to device, on the producing stream);
 * then ONE ncclAllGather of the first n_floats of the send buffer (every rank passes the same n_floats = the largest
 * share; capacity = rk_comm_init's max_floats_per_rank); rk_comm_read_appended waits and copies out[world][n_floats].
 * Every rank issues exactly one collective per query, whatever its number of engine calls. */
int rk_comm_append_scores_slot(rk_engine* e, int slot, int n_floats, int dst_offset);
/* the same for n_floats HOST values (small per-passage side data that has to reach every rank with the scores, e.g. the
 * token counts behind the reference's prompt-token counter, ref: llmrankers/pointwise.py:105-114): staged through pinned
 * memory, copied to the send buffer on the stream the gather will run on; the caller's buffer is free on return */
int rk_comm_append_host(rk_engine* e, const float* values, int n_floats, int dst_offset);
int rk_comm_all_gather_appended(rk_engine* e, int n_floats);
int rk_comm_read_appended(rk_engine* e, float* out, int n_floats_total);
int rk_comm_destroy(rk_engine* e);

/* ---- measurement (HIP events on the engine's own stream) ---- */
int rk_timer_begin(rk_engine* e);                 /* record start event on the engine stream */
int rk_timer_end(rk_engine* e, float* out_ms);    /* record stop, synchronise, elapsed ms */
/* per-kernel-class profiling: when enabled every launch is bracketed by an event pair (perturbs throughput;
 * use in a separate pass).  Classes are listed by rk_profile_class_name. */
int rk_profile_enable(rk_engine* e, int on);
int rk_profile_reset(rk_engine* e);
int rk_profile_num_classes(void);
const char* rk_profile_class_name(int cls);
int rk_profile_get(rk_engine* e, int cls, double* total_ms, int64_t* launches, double* flops, double* bytes);
/* Engine options: the defaults are the fast path; every option selects between TESTED implementations of the same arithmetic (the
 * on-device cross-check of a default path: "fold_norm", "gemm_glds", "attn_short", "attn_long", "dec_attn_seq", "dec_cross_mfma",
 * "llama_attn_dma", ...) or is a measurement knob ("overlap", "dec_graph", "gemm_variant", "gemm_split", "gemm_sk", ...).  ONE
 * table holds key, accepted range / set and meaning: kOptions next to rk_engine_set_option in csrc/rk_engine.hip; INTEGRATION.md
 * lists them by purpose.  Returns RK_ERR_INVALID for an unknown key or a value outside the key's range (nothing is applied). */
int rk_engine_set_option(rk_engine* e, const char* key, int value);

/* ---- host-only helpers (no device needed) ---- */
int rk_abi_version(void);
/* T5 relative-position bucket (hf: modeling_t5.py:216-262) as used to build the device bias tables */
int rk_rel_bucket(int relative_position, int bidirectional, int num_buckets, int max_distance);
/* debug: run one GEMM through the engine's kernel on host data (A[M,K] fp16, W[N,K] fp16 -> C[M,N] fp32);
 * use_glds: 1 = tiled kernel with LDS-DMA staging, 0 = register staging, 2 = the weight-streaming (decoder) kernel,
 * 3 = the few-row GEMV kernel (M <= 16, K <= 3072; falls back to 2 otherwise) */
int rk_debug_gemm(rk_engine* e, const uint16_t* A, const uint16_t* W, float* C, int M, int N, int K, int use_glds);
/* measurement: average ms per launch of the engine's GEMM kernel at one shape (epi = 0 store f16, 1 residual f32,
 * 2 GEGLU, 3 ReLU, 4 store f32), random operands, `iters` back-to-back launches timed with HIP events */
int rk_debug_gemm_bench(rk_engine* e, int M, int N, int K, int epi, int iters, float* out_ms);
/* debug: copy an internal activation buffer to the host as fp32. name: "enc_hidden" [T,d], "enc_out" [T,d],
 * "qkv" [T,3I], "ctx" [T,I], "dec_hidden" [B*Ld,d]. Returns number of floats written or a negative status. */
int64_t rk_debug_read(rk_engine* e, const char* name, float* out, int64_t max_floats);

#ifdef __cplusplus
}
#endif
#endif /* RK_ENGINE_H */
