"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not a product path, not a fallback.

A plain numpy fp32 restatement of the decoder-only Llama forward the reference reaches through
`self.llm.generate(input_ids, do_sample=False, max_new_tokens=1)` in its setwise ranker (ref: llmrankers/setwise.py:60-69,
159-177).  As for T5, the arithmetic lives in the third-party dependency **transformers** (installed: 5.15.0):
hf: models/llama/modeling_llama.py — LlamaRMSNorm :52-67, rotary embedding :70-127 and apply_rotary_pos_emb :130-160,
LlamaMLP :163-177, grouped-query attention :180-288 (repeat_kv, scaling = head_dim**-0.5, causal mask), decoder layer
:291-330, final norm + lm_head.  Pinned by tests/test_oracle_golden.py against logits of HF's LlamaForCausalLM and
against end-to-end cases of the reference's own SetwiseLlmRanker (tools/make_goldens.py --only-llama).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def rmsnorm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """hf: modeling_llama.py:61-66."""
    var = np.mean(x.astype(np.float32) ** 2, axis=-1, keepdims=True)
    return w * (x * (1.0 / np.sqrt(var + np.float32(eps))))


def rope_inv_freq(head_dim: int, theta: float, scaling=None) -> np.ndarray:
    """Inverse frequencies [head_dim / 2], fp32.  scaling = (factor, low_freq_factor, high_freq_factor,
    original_max_position_embeddings): rope type "llama3" (hf: modeling_rope_utils.py _compute_llama3_parameters) -
    wavelengths above original/low are divided by factor, those between original/high and original/low interpolated;
    None: the default rope type (hf: modeling_llama.py:94-127)."""
    inv_freq = (1.0 / (np.float32(theta) ** (np.arange(0, head_dim, 2, dtype=np.float32) / np.float32(head_dim)))).astype(np.float32)
    if scaling is not None:
        factor, low, high, orig = (np.float32(scaling[0]), np.float32(scaling[1]), np.float32(scaling[2]), np.float32(scaling[3]))
        wavelen = (np.float32(2.0 * np.pi) / inv_freq).astype(np.float32)
        scaled = np.where(wavelen > orig / low, inv_freq / factor, inv_freq).astype(np.float32)
        smooth = ((orig / wavelen - low) / (high - low)).astype(np.float32)
        smoothed = ((np.float32(1.0) - smooth) * scaled / factor + smooth * scaled).astype(np.float32)
        medium = ~(wavelen < orig / high) & ~(wavelen > orig / low)
        inv_freq = np.where(medium, smoothed, scaled).astype(np.float32)
    return inv_freq


def rope_tables(n_pos: int, head_dim: int, theta: float, scaling=None):
    """hf: modeling_llama.py:94-127: cos / sin [n_pos, head_dim], fp32 like the reference's CPU path."""
    inv_freq = rope_inv_freq(head_dim, theta, scaling)
    freqs = np.arange(n_pos, dtype=np.float32)[:, None] * inv_freq[None, :]
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)


def rotate_half(x: np.ndarray) -> np.ndarray:
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


class LlamaOracle:
    def __init__(self, dims, state: Dict[str, np.ndarray]):
        self.d = dims
        self.w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in state.items()}

    def _lin(self, x, name):
        return x @ self.w[name].T

    def hidden_states(self, ids: Sequence[int]) -> np.ndarray:
        """[L, hidden] after the final norm, one unpadded sequence, full causal prefill."""
        d = self.d
        L = len(ids)
        h = self.w["model.embed_tokens.weight"][np.asarray(ids, dtype=np.int64)]
        cos, sin = rope_tables(L, d.head_dim, d.rope_theta, getattr(d, "rope_scaling", None))
        causal = np.where(np.arange(L)[None, :] > np.arange(L)[:, None], np.float32(np.finfo(np.float32).min), np.float32(0.0))
        rep = d.n_heads // d.n_kv_heads
        for i in range(d.n_layers):
            p = f"model.layers.{i}"
            x = rmsnorm(h, self.w[p + ".input_layernorm.weight"], d.eps)
            q = self._lin(x, p + ".self_attn.q_proj.weight").reshape(L, d.n_heads, d.head_dim).transpose(1, 0, 2)
            k = self._lin(x, p + ".self_attn.k_proj.weight").reshape(L, d.n_kv_heads, d.head_dim).transpose(1, 0, 2)
            v = self._lin(x, p + ".self_attn.v_proj.weight").reshape(L, d.n_kv_heads, d.head_dim).transpose(1, 0, 2)
            q = q * cos[None] + rotate_half(q) * sin[None]
            k = k * cos[None] + rotate_half(k) * sin[None]
            k = np.repeat(k, rep, axis=0)                      # hf: repeat_kv :180-189
            v = np.repeat(v, rep, axis=0)
            s = (q @ k.transpose(0, 2, 1)) * np.float32(d.head_dim ** -0.5) + causal[None]
            s = s - s.max(axis=-1, keepdims=True)
            pr = np.exp(s)
            pr = pr / pr.sum(axis=-1, keepdims=True)
            ctx = (pr @ v).transpose(1, 0, 2).reshape(L, d.n_heads * d.head_dim)
            h = h + self._lin(ctx, p + ".self_attn.o_proj.weight")
            x = rmsnorm(h, self.w[p + ".post_attention_layernorm.weight"], d.eps)
            g = self._lin(x, p + ".mlp.gate_proj.weight")
            act = g / (1.0 + np.exp(-g))                       # silu
            h = h + self._lin(act * self._lin(x, p + ".mlp.up_proj.weight"), p + ".mlp.down_proj.weight")
        return rmsnorm(h, self.w["model.norm.weight"], d.eps)

    def last_logits(self, seqs: List[Sequence[int]], out_ids=None) -> np.ndarray:
        """logits of the LAST position of every sequence, [B, V] or [B, n_out]."""
        head = self.w["model.embed_tokens.weight"] if self.d.tied_head else self.w["lm_head.weight"]
        rows = []
        for ids in seqs:
            lg = self.hidden_states(ids)[-1] @ head.T
            rows.append(lg if out_ids is None else lg[np.asarray(out_ids, dtype=np.int64)])
        return np.stack(rows).astype(np.float32)

    def greedy1(self, seqs: List[Sequence[int]]) -> np.ndarray:
        """generate(do_sample=False, max_new_tokens=1): arg-max (first maximum) of the last position's logits."""
        return np.argmax(self.last_logits(seqs), axis=-1).astype(np.int32)
