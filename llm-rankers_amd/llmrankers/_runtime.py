"""Checkpoint loading and the per-process engine handle shared by the rankers.

Replaces `T5ForConditionalGeneration.from_pretrained(..., device_map='auto', torch_dtype=fp16)` of
ref: llmrankers/pointwise.py:20-24 and llmrankers/setwise.py:46-50: reads a HuggingFace-layout checkpoint
directory (config.json + *.safetensors), hands every tensor to the C ABI (rk_engine_load_tensor) and finalises
the MI355X engine.  No torch model is ever built and there is no CPU path: device='cpu' raises.
"""
from __future__ import annotations

import json
import os
from typing import Iterator, List, Sequence, Tuple

import numpy as np

from . import _synth
from ._engine import RkEngine


def parse_device(device) -> int:
    """'cuda' (the reference's default, ref: run.py:222), 'cuda:N', 'hip', 'hip:N' or an int -> ordinal."""
    if isinstance(device, int):
        return device
    s = str(device).lower()
    if s == "cpu":
        raise RuntimeError("the MI355X engine has no CPU path (device='cpu'); use the HuggingFace reference for CPU runs")
    if s in ("cuda", "hip", "gpu"):
        return int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("RK_DEVICE_FROM_LOCAL_RANK") else 0
    for pre in ("cuda:", "hip:"):
        if s.startswith(pre):
            return int(s[len(pre):])
    raise ValueError(f"unrecognised device {device!r}")


def read_config(model_dir: str) -> dict:
    path = os.path.join(model_dir, "config.json")
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: the engine loads local HuggingFace-layout checkpoints (no network in this build)")
    with open(path) as f:
        return json.load(f)


def iter_checkpoint_tensors(model_dir: str) -> Iterator[Tuple[str, np.ndarray]]:
    """Yield (hf_name, ndarray) from model.safetensors or a sharded safetensors checkpoint."""
    from safetensors import safe_open
    idx = os.path.join(model_dir, "model.safetensors.index.json")
    if os.path.exists(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    elif os.path.exists(os.path.join(model_dir, "model.safetensors")):
        files = ["model.safetensors"]
    else:
        raise FileNotFoundError(f"no model.safetensors[.index.json] in {model_dir} (pytorch_model.bin is not supported)")
    for fn in files:
        path = os.path.join(model_dir, fn)
        try:
            with safe_open(path, framework="np") as f:
                for k in f.keys():
                    yield k, f.get_tensor(k)
        except TypeError:
            # bf16 has no numpy dtype: read through torch (plumbing only) and hand raw bf16 bits to the engine
            import torch
            with safe_open(path, framework="pt") as f:
                for k in f.keys():
                    t = f.get_tensor(k)
                    if t.dtype == torch.bfloat16:
                        yield k, t.view(torch.int16).numpy().view(np.uint16)
                    else:
                        yield k, t.float().numpy()


class T5Runtime:
    """Engine + chunking so a call may exceed the engine's token capacity (results are batch-independent)."""

    def __init__(self, model_name_or_path: str, device, max_tokens: int = 49152, max_seqs: int = 256,
                 max_dec_len: int = 136):
        cfg = read_config(model_name_or_path)
        self.model_type = cfg.get("model_type")
        if self.model_type != "t5":
            raise NotImplementedError(f"Model type {self.model_type} is not supported yet by the MI355X engine")
        self.config = cfg
        self.dims = _synth.T5Dims.from_hf_config(cfg)
        self.decoder_start_token_id = cfg.get("decoder_start_token_id", 0)
        self.max_tokens, self.max_seqs = max_tokens, max_seqs
        self.engine = RkEngine(self.dims, parse_device(device), max_tokens, max_seqs, max_dec_len)
        self.engine.load_state(iter_checkpoint_tensors(model_name_or_path))

    def _chunks(self, seqs: Sequence[Sequence[int]]) -> Iterator[List[Sequence[int]]]:
        cur, tok = [], 0
        for s in seqs:
            if len(s) > self.max_tokens:
                raise ValueError(f"a prompt of {len(s)} tokens exceeds the engine capacity {self.max_tokens}")
            if cur and (tok + len(s) > self.max_tokens or len(cur) == self.max_seqs):
                yield cur
                cur, tok = [], 0
            cur.append(s)
            tok += len(s)
        if cur:
            yield cur

    def score(self, seqs, dec_prefix, out_ids) -> np.ndarray:
        return np.concatenate([self.engine.score(c, dec_prefix, out_ids) for c in self._chunks(seqs)], axis=0)

    def score_batches(self, batches, dec_prefix, out_ids) -> List[np.ndarray]:
        """Score several independent batches, keeping the engine's batch slots full: batch i+1 is staged and its
        encoder enqueued while the decoder chain of batch i still runs (rk_t5_stage_slot / rk_t5_score_slot).
        Results are identical to calling score() per batch; only the waiting is overlapped."""
        eng = self.engine
        n_slots = eng.num_slots
        # The reference's batch_size only shapes its host loop; results do not depend on batch composition (ragged
        # execution), so consecutive batches are merged up to the engine's capacity: one launch sequence then covers
        # ~one query's candidates - better GEMM tile quantisation and ONE decoder chain instead of one per batch.
        flat = [s for b in batches for s in b]
        work = list(self._chunks(flat))
        parts = []
        pending = []                                                   # slots in submission order
        for k, chunk in enumerate(work):
            slot = k % n_slots
            if len(pending) == n_slots:                                # the slot we are about to reuse must be drained
                parts.append(eng.read_scores(pending.pop(0)))
            eng.stage(chunk, slot=slot)
            eng.score_staged(dec_prefix, out_ids, slot=slot)
            pending.append(slot)
        for s0 in pending:
            parts.append(eng.read_scores(s0))
        allsc = np.concatenate(parts, axis=0) if parts else np.zeros((0, len(out_ids)), np.float32)
        out, pos = [], 0
        for b in batches:
            out.append(allsc[pos:pos + len(b)])
            pos += len(b)
        return out

    def qlm(self, seqs, labels) -> np.ndarray:
        return np.concatenate([self.engine.qlm(c, labels) for c in self._chunks(seqs)], axis=0)

    def greedy(self, seqs, dec_prefix, max_new, eos_id=1, pad_id=0) -> np.ndarray:
        """[B, max_new] new tokens; columns after the step at which every row had finished hold -1."""
        parts = []
        for c in self._chunks(seqs):
            toks, steps = self.engine.greedy(c, dec_prefix, max_new, eos_id, pad_id)
            toks = toks.copy()
            toks[:, steps:] = -1
            parts.append(toks)
        return np.concatenate(parts, axis=0)
