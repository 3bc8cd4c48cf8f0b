"""Deterministic synthetic T5 checkpoints (no network, no real weights anywhere offline).

A counter-based generator (splitmix64 -> Box-Muller) so that the build container, the
GPU box, the oracle and the HIP engine can all regenerate bit-identical weights from
(dims, seed) without shipping multi-GB files.  Tensor names and shapes follow the
HuggingFace T5 state-dict (hf: models/t5/modeling_t5.py:590-616 gives the init scales
we imitate so activations stay well conditioned).

All values are rounded to fp16-representable numbers: the reference's accelerator path
loads the checkpoint as fp16 (ref: llmrankers/pointwise.py:20-24), so "the model" both
the oracle (fp32 math) and the engine (fp16 MFMA inputs) see is the same set of numbers.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, Iterator, Optional, Tuple

import numpy as np


@dataclass(frozen=True)
class T5Dims:
    vocab: int
    d_model: int
    n_heads: int
    d_kv: int
    d_ff: int
    n_enc: int
    n_dec: int
    gated: bool = True          # T5 v1.1 / flan: gated-gelu; v1.0 / monoT5: relu
    tied_head: bool = False     # v1.0 ties lm_head to shared and scales by d_model**-0.5
    n_buckets: int = 32
    max_distance: int = 128
    eps: float = 1e-6

    @property
    def inner(self) -> int:
        return self.n_heads * self.d_kv

    def to_hf_config(self) -> dict:
        """config.json content HF's T5Config understands (used by tools/ and oracle/hf_path)."""
        return {
            "architectures": ["T5ForConditionalGeneration"],
            "model_type": "t5",
            "vocab_size": self.vocab,
            "d_model": self.d_model,
            "d_kv": self.d_kv,
            "d_ff": self.d_ff,
            "num_heads": self.n_heads,
            "num_layers": self.n_enc,
            "num_decoder_layers": self.n_dec,
            "relative_attention_num_buckets": self.n_buckets,
            "relative_attention_max_distance": self.max_distance,
            "layer_norm_epsilon": self.eps,
            "feed_forward_proj": "gated-gelu" if self.gated else "relu",
            "tie_word_embeddings": bool(self.tied_head),
            "dropout_rate": 0.0,
            "initializer_factor": 1.0,
            "is_encoder_decoder": True,
            "use_cache": True,
            "pad_token_id": 0,
            "eos_token_id": 1,
            "decoder_start_token_id": 0,
        }

    @staticmethod
    def from_hf_config(cfg: dict) -> "T5Dims":
        ffp = cfg.get("feed_forward_proj", "relu")
        if ffp not in ("relu", "gated-gelu"):
            raise NotImplementedError(f"feed_forward_proj={ffp!r} is not supported by the MI355X engine")
        return T5Dims(
            vocab=cfg["vocab_size"], d_model=cfg["d_model"], n_heads=cfg["num_heads"], d_kv=cfg["d_kv"],
            d_ff=cfg["d_ff"], n_enc=cfg["num_layers"],
            n_dec=cfg.get("num_decoder_layers") or cfg["num_layers"],
            gated=ffp.startswith("gated"),
            tied_head=bool(cfg.get("tie_word_embeddings", True)),
            n_buckets=cfg.get("relative_attention_num_buckets", 32),
            max_distance=cfg.get("relative_attention_max_distance", 128),
            eps=cfg.get("layer_norm_epsilon", 1e-6),
        )

    def asdict(self) -> dict:
        return asdict(self)


# Public model-card dimensions.
FLAN_T5_SMALL = T5Dims(vocab=32128, d_model=512, n_heads=6, d_kv=64, d_ff=1024, n_enc=8, n_dec=8)
FLAN_T5_BASE = T5Dims(vocab=32128, d_model=768, n_heads=12, d_kv=64, d_ff=2048, n_enc=12, n_dec=12)
FLAN_T5_LARGE = T5Dims(vocab=32128, d_model=1024, n_heads=16, d_kv=64, d_ff=2816, n_enc=24, n_dec=24)
FLAN_T5_XL = T5Dims(vocab=32128, d_model=2048, n_heads=32, d_kv=64, d_ff=5120, n_enc=24, n_dec=24)
# Toy models for fixtures: kernel-friendly (multiples of 64) but with inner != d_model.
TOY_GATED_UNTIED = T5Dims(vocab=256, d_model=128, n_heads=3, d_kv=64, d_ff=256, n_enc=2, n_dec=2)
TOY_RELU_TIED = T5Dims(vocab=256, d_model=128, n_heads=3, d_kv=64, d_ff=256, n_enc=2, n_dec=2,
                       gated=False, tied_head=True)
# monoT5-like (T5 v1.0: relu, tied + scaled head) with a vocabulary that covers the fixed 'true' / 'false' token ids
# 1176 / 6136 the reference hard-codes (ref: llmrankers/pointwise.py:177-178)
TOY_MONOT5 = T5Dims(vocab=6144, d_model=128, n_heads=3, d_kv=64, d_ff=256, n_enc=2, n_dec=2, gated=False, tied_head=True)

@dataclass(frozen=True)
class LlamaDims:
    """Decoder-only Llama family (hf: models/llama/configuration_llama.py): RMSNorm, RoPE, grouped-query attention, SwiGLU."""
    vocab: int
    hidden: int
    n_heads: int
    n_kv_heads: int
    head_dim: int
    intermediate: int
    n_layers: int
    rope_theta: float = 500000.0
    eps: float = 1e-5
    tied_head: bool = False
    bos_token_id: int = 1
    eos_token_id: int = 2
    # rope_type "llama3" (Llama-3.1 / 3.2 checkpoints; hf: modeling_rope_utils.py _compute_llama3_parameters):
    # (factor, low_freq_factor, high_freq_factor, original_max_position_embeddings), None = default rope type
    rope_scaling: Optional[Tuple[float, float, float, int]] = None

    def to_hf_config(self) -> dict:
        scaling = None
        if self.rope_scaling is not None:
            f, lo, hi, orig = self.rope_scaling
            scaling = {"rope_type": "llama3", "factor": float(f), "low_freq_factor": float(lo), "high_freq_factor": float(hi),
                       "original_max_position_embeddings": int(orig)}
        return {
            "architectures": ["LlamaForCausalLM"], "model_type": "llama", "vocab_size": self.vocab,
            "hidden_size": self.hidden, "intermediate_size": self.intermediate, "num_hidden_layers": self.n_layers,
            "num_attention_heads": self.n_heads, "num_key_value_heads": self.n_kv_heads, "head_dim": self.head_dim,
            "hidden_act": "silu", "rms_norm_eps": self.eps, "rope_theta": self.rope_theta, "rope_scaling": scaling,
            "max_position_embeddings": 8192, "attention_bias": False, "mlp_bias": False,
            "tie_word_embeddings": bool(self.tied_head), "bos_token_id": self.bos_token_id, "eos_token_id": self.eos_token_id,
            "use_cache": True,
        }

    @staticmethod
    def from_hf_config(cfg: dict) -> "LlamaDims":
        if cfg.get("hidden_act", "silu") != "silu" or cfg.get("attention_bias") or cfg.get("mlp_bias"):
            raise NotImplementedError("only bias-free SwiGLU Llama configurations are supported by the MI355X engine")
        rope = cfg.get("rope_parameters") or {}
        scaling = cfg.get("rope_scaling") or ({k: v for k, v in rope.items() if k != "rope_theta"} if rope.get("rope_type", "default") != "default" else None)
        rs = None
        if scaling and scaling.get("rope_type", scaling.get("type", "default")) != "default":
            if scaling.get("rope_type", scaling.get("type")) != "llama3":
                raise NotImplementedError(f"rope scaling {scaling} is not supported by the MI355X engine (default and llama3 rope types are)")
            rs = (float(scaling["factor"]), float(scaling["low_freq_factor"]), float(scaling["high_freq_factor"]),
                  int(scaling.get("original_max_position_embeddings") or cfg.get("max_position_embeddings")))
        heads = cfg["num_attention_heads"]
        eos = cfg.get("eos_token_id", 2)
        return LlamaDims(vocab=cfg["vocab_size"], hidden=cfg["hidden_size"], n_heads=heads,
                         n_kv_heads=cfg.get("num_key_value_heads") or heads,
                         head_dim=cfg.get("head_dim") or cfg["hidden_size"] // heads,
                         intermediate=cfg["intermediate_size"], n_layers=cfg["num_hidden_layers"],
                         rope_theta=float(cfg.get("rope_theta") or rope.get("rope_theta") or 10000.0),
                         eps=cfg.get("rms_norm_eps", 1e-6), tied_head=bool(cfg.get("tie_word_embeddings", False)),
                         bos_token_id=cfg.get("bos_token_id", 1) or 1, eos_token_id=eos[0] if isinstance(eos, list) else eos,
                         rope_scaling=rs)


LLAMA_3_8B = LlamaDims(vocab=128256, hidden=4096, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336, n_layers=32,
                       bos_token_id=128000, eos_token_id=128001)
# toy: kernel-friendly (head_dim 128 like every Llama-3), grouped-query (4 q heads on 2 kv heads), q width != hidden
TOY_LLAMA = LlamaDims(vocab=256, hidden=256, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=512, n_layers=2)
# the same with Llama-3.1's rope type; a short original context and a small base so that most of the 64 frequencies fall in
# the scaled and the interpolated bands at toy sequence lengths
TOY_LLAMA3ROPE = LlamaDims(vocab=256, hidden=256, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=512, n_layers=2,
                           rope_theta=10000.0, rope_scaling=(8.0, 1.0, 4.0, 32))


def llama_tensor_specs(d: "LlamaDims") -> Iterator[Tuple[str, Tuple[int, ...], float, bool]]:
    """(hf_name, shape, std, is_norm_weight) of a LlamaForCausalLM checkpoint, in a fixed order."""
    yield "model.embed_tokens.weight", (d.vocab, d.hidden), 1.0, False
    for i in range(d.n_layers):
        p = f"model.layers.{i}"
        yield f"{p}.self_attn.q_proj.weight", (d.n_heads * d.head_dim, d.hidden), d.hidden ** -0.5, False
        yield f"{p}.self_attn.k_proj.weight", (d.n_kv_heads * d.head_dim, d.hidden), d.hidden ** -0.5, False
        yield f"{p}.self_attn.v_proj.weight", (d.n_kv_heads * d.head_dim, d.hidden), d.hidden ** -0.5, False
        yield f"{p}.self_attn.o_proj.weight", (d.hidden, d.n_heads * d.head_dim), (d.n_heads * d.head_dim) ** -0.5, False
        yield f"{p}.input_layernorm.weight", (d.hidden,), 0.1, True
        yield f"{p}.mlp.gate_proj.weight", (d.intermediate, d.hidden), d.hidden ** -0.5, False
        yield f"{p}.mlp.up_proj.weight", (d.intermediate, d.hidden), d.hidden ** -0.5, False
        yield f"{p}.mlp.down_proj.weight", (d.hidden, d.intermediate), d.intermediate ** -0.5, False
        yield f"{p}.post_attention_layernorm.weight", (d.hidden,), 0.1, True
    yield "model.norm.weight", (d.hidden,), 0.1, True
    if not d.tied_head:
        yield "lm_head.weight", (d.vocab, d.hidden), d.hidden ** -0.5, False


NAMED_DIMS = {
    "flan-t5-small": FLAN_T5_SMALL, "flan-t5-base": FLAN_T5_BASE, "flan-t5-large": FLAN_T5_LARGE,
    "flan-t5-xl": FLAN_T5_XL, "toy-gated-untied": TOY_GATED_UNTIED, "toy-relu-tied": TOY_RELU_TIED, "toy-monot5": TOY_MONOT5,
    "llama-3-8b": LLAMA_3_8B, "toy-llama": TOY_LLAMA, "toy-llama3rope": TOY_LLAMA3ROPE,
}

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def counter_normal(n: int, stream: int, seed: int) -> np.ndarray:
    """n standard normals, a pure function of (stream, seed, index). float64 Box-Muller."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed * 0x1000003 + stream], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64)
        a = _splitmix64(idx * np.uint64(2) + base)
        b = _splitmix64(idx * np.uint64(2) + np.uint64(1) + base)
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740993.0)  # (0,1)
    u2 = (b >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)          # [0,1)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _fp16_round(x: np.ndarray) -> np.ndarray:
    return x.astype(np.float16).astype(np.float32)


def tensor_specs(d: T5Dims) -> Iterator[Tuple[str, Tuple[int, ...], float, bool]]:
    """(hf_name, shape, std, is_norm_weight) for every tensor of the checkpoint, in a fixed order."""
    yield "shared.weight", (d.vocab, d.d_model), 1.0, False
    for stack, n_layers in (("encoder", d.n_enc), ("decoder", d.n_dec)):
        for i in range(n_layers):
            p = f"{stack}.block.{i}.layer"
            attn_layers = [(0, "SelfAttention")] + ([(1, "EncDecAttention")] if stack == "decoder" else [])
            for li, an in attn_layers:
                yield f"{p}.{li}.{an}.q.weight", (d.inner, d.d_model), (d.d_model * d.d_kv) ** -0.5, False
                yield f"{p}.{li}.{an}.k.weight", (d.inner, d.d_model), d.d_model ** -0.5, False
                yield f"{p}.{li}.{an}.v.weight", (d.inner, d.d_model), d.d_model ** -0.5, False
                yield f"{p}.{li}.{an}.o.weight", (d.d_model, d.inner), d.inner ** -0.5, False
                if i == 0 and li == 0:
                    yield (f"{p}.0.SelfAttention.relative_attention_bias.weight", (d.n_buckets, d.n_heads),
                           0.5, False)   # HF uses d_model**-0.5; 0.5 makes the bias matter in tests
                yield f"{p}.{li}.layer_norm.weight", (d.d_model,), 0.1, True
            fl = 2 if stack == "decoder" else 1
            if d.gated:
                yield f"{p}.{fl}.DenseReluDense.wi_0.weight", (d.d_ff, d.d_model), d.d_model ** -0.5, False
                yield f"{p}.{fl}.DenseReluDense.wi_1.weight", (d.d_ff, d.d_model), d.d_model ** -0.5, False
            else:
                yield f"{p}.{fl}.DenseReluDense.wi.weight", (d.d_ff, d.d_model), d.d_model ** -0.5, False
            yield f"{p}.{fl}.DenseReluDense.wo.weight", (d.d_model, d.d_ff), d.d_ff ** -0.5, False
            yield f"{p}.{fl}.layer_norm.weight", (d.d_model,), 0.1, True
        yield f"{stack}.final_layer_norm.weight", (d.d_model,), 0.1, True
    if not d.tied_head:
        yield "lm_head.weight", (d.vocab, d.d_model), d.d_model ** -0.5, False


def _make_tensor(args):
    stream, name, shape, std, is_norm, seed, gain = args
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    step = 1 << 22                      # chunked so peak memory stays small and threads interleave well
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed * 0x1000003 + stream], dtype=np.uint64))[0]
    for s0 in range(0, n, step):
        s1 = min(n, s0 + step)
        with np.errstate(over="ignore"):
            idx = np.arange(s0, s1, dtype=np.uint64)
            a = _splitmix64(idx * np.uint64(2) + base)
            b = _splitmix64(idx * np.uint64(2) + np.uint64(1) + base)
        u1 = ((a >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740993.0)
        u2 = (b >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
        w = (1.0 + std * z) if is_norm else (gain if std < 1.0 else 1.0) * std * z
        out[s0:s1] = w.astype(np.float32)
    return name, _fp16_round(out.reshape(shape))


def synth_tensors(d, seed: int = 929, gain: float = 1.0, threads: int = 0) -> Iterator[Tuple[str, np.ndarray]]:
    """Yield (hf_name, fp32 array with fp16-representable values).  A pure function of (dims, seed, gain):
    value i of tensor #s is Box-Muller(splitmix64 counters) — identical on every machine and thread count."""
    specs = llama_tensor_specs(d) if isinstance(d, LlamaDims) else tensor_specs(d)
    jobs = [(stream, name, shape, std, is_norm, seed, gain)
            for stream, (name, shape, std, is_norm) in enumerate(specs)]
    if threads <= 1:
        for j in jobs:
            yield _make_tensor(j)
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as ex:       # numpy ufuncs release the GIL
        for res in ex.map(_make_tensor, jobs):
            yield res


def synth_tensors_pool(d, seed: int = 929, pool_log2: int = 25) -> Iterator[Tuple[str, np.ndarray]]:
    """TIMING-ONLY weights for the multi-billion-parameter bench legs (flan-t5-xl, Llama-3-8B): the same names, shapes and
    per-tensor standard deviations as synth_tensors, but every tensor is a window into ONE pool of 2^pool_log2 counter-generated
    standard normals (pre-scaled per distinct std, fp16), at a per-tensor offset.  Statistically the same operand bits for the
    matrix cores (their power draw depends on the data) at a fraction of the host time: Llama-3-8B in seconds instead of two
    minutes.  NOT for parity work - tensors overlap, and goldens are made from synth_tensors."""
    specs = list(llama_tensor_specs(d) if isinstance(d, LlamaDims) else tensor_specs(d))
    n_pool = 1 << pool_log2
    from concurrent.futures import ThreadPoolExecutor
    import os
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:       # numpy ufuncs release the GIL
        z = np.concatenate([r[1] for r in ex.map(_make_tensor, [(1000 + i, "pool", (1 << 22,), 1.0, False, seed, 1.0)
                                                                 for i in range(n_pool >> 22)])])
    pools = {}
    for stream, (name, shape, std, is_norm) in enumerate(specs):
        n = int(np.prod(shape))
        if is_norm:
            yield name, _fp16_round(1.0 + std * z[(stream * 7919) % 4096:][:n])
            continue
        if std not in pools:
            pools[std] = (z * np.float32(std)).astype(np.float16)
        pool = pools[std]
        off = (stream * 2654435761) % (n_pool - min(n, n_pool - 1))
        if n <= n_pool - off:
            yield name, pool[off:off + n].reshape(shape)
        else:                                           # a tensor larger than the pool (embedding / head of a 128k vocabulary)
            out, pos, r = np.empty(n, dtype=np.float16), 0, 0
            while pos < n:                                  # pool-sized pieces, each starting at its own offset (wrapping)
                o = (off * (r + 1) + r * 104729) % n_pool
                m = min(n_pool - o, n - pos)
                out[pos:pos + m] = pool[o:o + m]
                pos, r = pos + m, r + 1
            yield name, out.reshape(shape)


def synth_state_dict(d, seed: int = 929, gain: float = 1.0, threads: int = 0) -> Dict[str, np.ndarray]:
    return dict(synth_tensors(d, seed, gain, threads))


OUTLIER_CHANNELS, OUTLIER_HIDDEN, OUTLIER_GAIN_WO, OUTLIER_GAIN_WI1 = (17, 300, 777), (5, 1200, 2001, 2777), 60.0, 100.0


def with_outlier_channels(state: Dict[str, np.ndarray], d: T5Dims) -> Dict[str, np.ndarray]:
    """The trained-checkpoint shape the N(0, init-std) weights lack: a few residual-stream channels and a few FFN hidden units
    two orders of magnitude above the rest (real T5 checkpoints have them; the engine's 1 / 16-scaled fp16 copy of the fp32 stream
    and its fp16 saturation exist for them).  In every block the `wo` rows of OUTLIER_CHANNELS are multiplied by OUTLIER_GAIN_WO
    (those stream channels then run in the hundreds to thousands and dominate every RMSNorm) and the `wi_1` (or `wi`) rows of
    OUTLIER_HIDDEN by OUTLIER_GAIN_WI1; values stay fp16-representable like every synthetic weight.  Returns a new dict."""
    out = dict(state)
    ch = [c for c in OUTLIER_CHANNELS if c < d.d_model]
    hid = [h for h in OUTLIER_HIDDEN if h < d.d_ff]
    for name, w in state.items():
        if name.endswith("DenseReluDense.wo.weight"):
            w = np.array(w, dtype=np.float32, copy=True)
            w[ch] = _fp16_round(w[ch] * np.float32(OUTLIER_GAIN_WO))
            out[name] = w
        elif name.endswith("DenseReluDense.wi_1.weight") or name.endswith("DenseReluDense.wi.weight"):
            w = np.array(w, dtype=np.float32, copy=True)
            w[hid] = _fp16_round(w[hid] * np.float32(OUTLIER_GAIN_WI1))
            out[name] = w
    return out


def synth_token_batch(n_seq: int, min_len: int, max_len: int, vocab: int, seed: int):
    """Ragged synthetic prompts: ids ~ U{3..vocab-29}, last id = 1 (EOS). Returns list of int32 arrays."""
    rs = np.random.RandomState(seed)   # MT19937: stable across numpy versions
    hi = max(4, vocab - 28)
    out = []
    for _ in range(n_seq):
        n = int(rs.randint(min_len, max_len + 1))
        ids = rs.randint(3, hi, size=n).astype(np.int32)
        ids[-1] = 1
        out.append(ids)
    return out


def write_checkpoint(path: str, spec: dict, tokenizer_dir: str = None) -> None:
    """Materialise a HF-layout checkpoint dir (config.json + model.safetensors [+ tokenizer files]) from a
    regeneration recipe {dims, seed, gain[, boost_ids, boost]} — see tests/golden/ckpts.json."""
    import json
    import os
    import shutil
    from safetensors.numpy import save_file
    dims = NAMED_DIMS[spec["dims"]]
    os.makedirs(path, exist_ok=True)
    sd = synth_state_dict(dims, seed=spec["seed"], gain=spec.get("gain", 1.0))
    if spec.get("boost_ids"):
        if isinstance(dims, LlamaDims):
            key = "model.embed_tokens.weight" if dims.tied_head else "lm_head.weight"
        else:
            key = "shared.weight" if dims.tied_head else "lm_head.weight"
        w = sd[key].copy()
        ids = np.asarray(spec["boost_ids"], dtype=np.int64)
        w[ids] = _fp16_round(w[ids] * np.float32(spec["boost"]))
        if spec.get("boost2_ids"):                      # a second group with its own factor (e.g. EOS a little below the labels)
            ids2 = np.asarray(spec["boost2_ids"], dtype=np.int64)
            w[ids2] = _fp16_round(w[ids2] * np.float32(spec["boost2"]))
        sd[key] = w
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(dims.to_hf_config(), f, indent=1)
    if tokenizer_dir:
        for fn in os.listdir(tokenizer_dir):
            shutil.copy(os.path.join(tokenizer_dir, fn), os.path.join(path, fn))


def checkpoint_sha256(path: str) -> str:
    import hashlib
    import os
    from safetensors.numpy import load_file
    sd = load_file(os.path.join(path, "model.safetensors"))
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()
