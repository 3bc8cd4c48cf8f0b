"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not a product path, not a fallback.

A plain numpy fp32 restatement of the arithmetic the reference's hot path executes.  The
reference itself (ielab/llm-rankers) contains no arithmetic: every FLOP of
`self.llm(input_ids, attention_mask, decoder_input_ids=...)` (ref: llmrankers/pointwise.py:117-119,
:73-75; llmrankers/setwise.py:93-95, :184) runs inside the third-party dependency
**transformers** (reference floor `>=4.31.0`, ref: setup.py:17-19; installed and pinned here:
5.15.0) — `T5ForConditionalGeneration.forward`, hf: models/t5/modeling_t5.py:939-1066.  Each
function below cites the hf: line it restates.

Parity pin: tools/make_goldens.py imports the real reference against HF in the build container,
captures inputs/activations/logits/rankings into tests/golden/, and tests/test_oracle_golden.py checks
this file against every one of them (fp32 tolerance 2e-5).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

Ragged execution: sequences are processed one at a time without padding.  HF pads to the batch's
longest and masks keys with finfo.min; exp() of that underflows to exactly 0 in fp32, and T5
positions are index based, so the unpadded result equals the padded one (SURVEY.md section 8a probe:
4.8e-7; re-checked by the golden test which uses padded HF batches).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np


def relative_position_bucket(rel: np.ndarray, bidirectional: bool, num_buckets: int = 32,
                             max_distance: int = 128) -> np.ndarray:
    """hf: modeling_t5.py:216-262 (T5Attention._relative_position_bucket). rel = key_pos - query_pos."""
    rel = rel.astype(np.int64)
    ret = np.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        ret += (rel > 0).astype(np.int64) * num_buckets
        rel = np.abs(rel)
    else:
        rel = -np.minimum(rel, 0)
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    # float32 log like torch; clamp the argument so log(0) never fires (masked by is_small anyway)
    relf = np.maximum(rel, 1).astype(np.float32)
    large = max_exact + (
        np.log(relf / np.float32(max_exact)) / np.float32(math.log(max_distance / max_exact))
        * np.float32(num_buckets - max_exact)
    ).astype(np.int64)
    large = np.minimum(large, num_buckets - 1)
    ret += np.where(is_small, rel, large)
    return ret


def rmsnorm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """hf: modeling_t5.py:59-72 (T5LayerNorm.forward): no mean subtraction, no bias."""
    var = np.mean(x.astype(np.float32) ** 2, axis=-1, keepdims=True)
    return w * (x * (1.0 / np.sqrt(var + np.float32(eps))))


def gelu_new(x: np.ndarray) -> np.ndarray:
    """hf: activations.py:59-66 (NewGELUActivation)."""
    c = np.float32(math.sqrt(2.0 / math.pi))
    return np.float32(0.5) * x * (np.float32(1.0) + np.tanh(c * (x + np.float32(0.044715) * x * x * x)))


def softmax_lastdim(s: np.ndarray) -> np.ndarray:
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m)
    return e / e.sum(axis=-1, keepdims=True)


class T5Oracle:
    """fp32 T5 encoder-decoder forward on a HF-named state dict (numpy arrays)."""

    def __init__(self, dims, state: Dict[str, np.ndarray]):
        self.d = dims
        self.w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in state.items()}
        self.capture: Optional[dict] = None     # set to {} to record intermediates of the next call

    # -- helpers -------------------------------------------------------------------------------
    def _cap(self, key, val):
        if self.capture is not None:
            self.capture[key] = np.array(val, copy=True)

    def _lin(self, x, name):
        return x @ self.w[name].T           # nn.Linear(bias=False): y = x W^T

    def _heads(self, x):                    # [L, inner] -> [H, L, d_kv]
        L = x.shape[0]
        return x.reshape(L, self.d.n_heads, self.d.d_kv).transpose(1, 0, 2)

    def _attn(self, xq, xkv, p, bias):
        """hf: modeling_t5.py:144-173 with scaling=1.0 (:196-197)."""
        q = self._heads(self._lin(xq, p + ".q.weight"))
        k = self._heads(self._lin(xkv, p + ".k.weight"))
        v = self._heads(self._lin(xkv, p + ".v.weight"))
        s = q @ k.transpose(0, 2, 1)
        if bias is not None:
            s = s + bias
        ctx = softmax_lastdim(s) @ v                                   # [H, Lq, d_kv]
        ctx = ctx.transpose(1, 0, 2).reshape(xq.shape[0], self.d.inner)
        return self._lin(ctx, p + ".o.weight")

    def _ffn(self, x, p):
        """hf: modeling_t5.py:75-123 (T5DenseActDense / T5DenseGatedActDense)."""
        if self.d.gated:
            h = gelu_new(self._lin(x, p + ".wi_0.weight")) * self._lin(x, p + ".wi_1.weight")
        else:
            h = np.maximum(self._lin(x, p + ".wi.weight"), 0.0)
        return self._lin(h, p + ".wo.weight")

    def _bias(self, stack: str, lq: int, lk: int) -> np.ndarray:
        """hf: modeling_t5.py:264-279 (compute_bias); table owned by block 0 (:739-742)."""
        tab = self.w[f"{stack}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
        rel = np.arange(lk)[None, :] - np.arange(lq)[:, None]
        b = relative_position_bucket(rel, bidirectional=(stack == "encoder"),
                                     num_buckets=self.d.n_buckets, max_distance=self.d.max_distance)
        return tab[b].transpose(2, 0, 1)                               # [H, lq, lk]

    # -- stacks --------------------------------------------------------------------------------
    def encode(self, ids: Sequence[int]) -> np.ndarray:
        """hf: modeling_t5.py:663-750 (T5Stack.forward, encoder) for one unpadded sequence."""
        d = self.d
        h = self.w["shared.weight"][np.asarray(ids, dtype=np.int64)]
        self._cap("enc.embed", h)
        bias = self._bias("encoder", len(ids), len(ids))
        self._cap("enc.bias", bias)
        for i in range(d.n_enc):
            p = f"encoder.block.{i}.layer"
            x = rmsnorm(h, self.w[p + ".0.layer_norm.weight"], d.eps)
            h = h + self._attn(x, x, p + ".0.SelfAttention", bias)
            self._cap(f"enc.{i}.attn", h)
            x = rmsnorm(h, self.w[p + ".1.layer_norm.weight"], d.eps)
            h = h + self._ffn(x, p + ".1.DenseReluDense")
            self._cap(f"enc.{i}.ffn", h)
        out = rmsnorm(h, self.w["encoder.final_layer_norm.weight"], d.eps)
        self._cap("enc.final", out)
        return out

    def decode(self, enc: np.ndarray, dec_ids: Sequence[int]) -> np.ndarray:
        """hf: modeling_t5.py:663-750 (decoder) + :1044-1047 (scale, lm_head). Returns logits [Ld, V]."""
        d = self.d
        ld = len(dec_ids)
        h = self.w["shared.weight"][np.asarray(dec_ids, dtype=np.int64)]
        causal = np.where(np.arange(ld)[None, :] > np.arange(ld)[:, None], np.float32(np.finfo(np.float32).min),
                          np.float32(0.0))
        bias = self._bias("decoder", ld, ld) + causal[None]
        for i in range(d.n_dec):
            p = f"decoder.block.{i}.layer"
            x = rmsnorm(h, self.w[p + ".0.layer_norm.weight"], d.eps)
            h = h + self._attn(x, x, p + ".0.SelfAttention", bias)
            self._cap(f"dec.{i}.self", h)
            x = rmsnorm(h, self.w[p + ".1.layer_norm.weight"], d.eps)
            h = h + self._attn(x, enc, p + ".1.EncDecAttention", None)   # zero bias, hf :337-342
            self._cap(f"dec.{i}.cross", h)
            x = rmsnorm(h, self.w[p + ".2.layer_norm.weight"], d.eps)
            h = h + self._ffn(x, p + ".2.DenseReluDense")
            self._cap(f"dec.{i}.ffn", h)
        out = rmsnorm(h, self.w["decoder.final_layer_norm.weight"], d.eps)
        if d.tied_head:
            out = out * np.float32(d.d_model ** -0.5)
            head = self.w["shared.weight"]
        else:
            head = self.w["lm_head.weight"]
        self._cap("dec.final", out)
        return out @ head.T

    # -- the three call shapes of the hot path (SURVEY.md section 8b inner boundary) -------------
    def score_last(self, seqs: List[Sequence[int]], dec_prefix: Sequence[int],
                   out_ids: Optional[Sequence[int]] = None) -> np.ndarray:
        """logits of the LAST decoder position, [B, n_out] (ref: pointwise.py:117-121, setwise.py:184)."""
        rows = []
        for ids in seqs:
            lg = self.decode(self.encode(ids), dec_prefix)[-1]
            rows.append(lg if out_ids is None else lg[np.asarray(out_ids, dtype=np.int64)])
        return np.stack(rows).astype(np.float32)

    def qlm(self, seqs: List[Sequence[int]], labels: Sequence[int]) -> np.ndarray:
        """-sum_t CE(logits_t, label_t) with decoder input = shift_right(labels)
        (ref: pointwise.py:73-79; hf: modeling_t5.py:618-637, 1026-1028)."""
        labels = np.asarray(labels, dtype=np.int64)
        dec_in = np.concatenate([[0], labels[:-1]])
        out = []
        for ids in seqs:
            lg = self.decode(self.encode(ids), dec_in).astype(np.float32)
            m = lg.max(axis=-1, keepdims=True)
            lse = (m + np.log(np.exp(lg - m).sum(axis=-1, keepdims=True)))[:, 0]
            out.append(-(lse - lg[np.arange(len(labels)), labels]).sum())
        return np.asarray(out, dtype=np.float32)

    def greedy(self, seqs: List[Sequence[int]], dec_prefix: Sequence[int], max_new: int,
               eos_id: int = 1, pad_id: int = 0) -> np.ndarray:
        """greedy continuation, [B, max_new] (ref: setwise.py:93-95; hf: generation/utils.py greedy:
        argmax = first max index; after EOS a row emits pad; stop when every row finished)."""
        res = np.full((len(seqs), max_new), pad_id, dtype=np.int32)
        for b, ids in enumerate(seqs):
            enc = self.encode(ids)
            cur = list(dec_prefix)
            for t in range(max_new):
                nxt = int(np.argmax(self.decode(enc, cur)[-1]))
                res[b, t] = nxt
                cur.append(nxt)
                if nxt == eos_id:
                    break
        return res
