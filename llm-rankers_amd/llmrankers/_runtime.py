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
from ._engine import RkEngine, RkLlamaEngine


def parse_device(device) -> int:
    """'cuda' (the reference's default, ref: run.py:222), 'cuda:N', 'hip', 'hip:N' or an int -> ordinal."""
    if isinstance(device, int):
        return device
    s = str(device).lower()
    if s == "cpu":
        raise RuntimeError("the MI355X engine has no CPU path (device='cpu'); use the HuggingFace reference for CPU runs")
    if s in ("cuda", "hip", "gpu"):
        # one process per GPU under torchrun: 'cuda' means "this rank's GPU" (LOCAL_RANK), ordinal 0 otherwise
        return int(os.environ.get("LOCAL_RANK", "0") or 0)
    for pre in ("cuda:", "hip:"):
        if s.startswith(pre):
            return int(s[len(pre):])
    raise ValueError(f"unrecognised device {device!r}")


def resolve_checkpoint(model_name_or_path: str, cache_dir=None) -> str:
    """A local HuggingFace-layout directory, or a hub id ('google/flan-t5-large', as the reference's README and
    run.py use) resolved through huggingface_hub's cache (downloading when the machine is online) — the same lookup
    from_pretrained does (ref: pointwise.py:15-24)."""
    if os.path.isdir(model_name_or_path):
        return model_name_or_path
    try:
        from huggingface_hub import snapshot_download
        st = ["config.json", "*.safetensors", "*.safetensors.index.json"]
        pt = ["config.json", "pytorch_model*.bin", "pytorch_model.bin.index.json"]

        def has_weights(d):
            names = os.listdir(d)
            return any(n.endswith(".safetensors") or (n.startswith("pytorch_model") and n.endswith(".bin")) for n in names)
        # safetensors first; the .bin files only when the repository has none (hub models that ship both are not fetched
        # twice).  A cache hit that holds config.json but no weights falls through to the online download.
        for kw in ({"local_files_only": True}, {}):
            for patterns in (st, pt):
                try:
                    d = snapshot_download(model_name_or_path, cache_dir=cache_dir, allow_patterns=patterns, **kw)
                except Exception:
                    continue
                if has_weights(d):
                    return d
        raise FileNotFoundError("no weight files (model*.safetensors / pytorch_model*.bin) found locally or on the hub")
    except Exception as exc:
        raise FileNotFoundError(
            f"{model_name_or_path!r} is neither a local checkpoint directory nor a hub model reachable from here ({exc})") from None


def read_config(model_dir: str) -> dict:
    path = os.path.join(model_dir, "config.json")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found: not a HuggingFace-layout checkpoint directory")
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
        yield from _iter_torch_bin(model_dir)
        return
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


def _iter_torch_bin(model_dir: str) -> Iterator[Tuple[str, np.ndarray]]:
    """pytorch_model.bin checkpoints (e.g. castorini/monot5-*): read with torch.load — loader plumbing only."""
    import torch
    idx = os.path.join(model_dir, "pytorch_model.bin.index.json")
    if os.path.exists(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    elif os.path.exists(os.path.join(model_dir, "pytorch_model.bin")):
        files = ["pytorch_model.bin"]
    else:
        raise FileNotFoundError(f"no model.safetensors[.index.json] or pytorch_model.bin[.index.json] in {model_dir}")
    for fn in files:
        sd = torch.load(os.path.join(model_dir, fn), map_location="cpu", weights_only=True)
        for k, t in sd.items():
            if t.dtype == torch.bfloat16:
                yield k, t.contiguous().view(torch.int16).numpy().view(np.uint16)
            elif t.dtype == torch.float16:
                yield k, t.numpy()
            else:
                yield k, t.float().numpy()


class T5Runtime:
    """Engine + chunking so a call may exceed the engine's token capacity (results are batch-independent)."""

    def __init__(self, model_name_or_path: str, device, max_tokens: int = 49152, max_seqs: int = 256,
                 max_dec_len: int = 136, cache_dir=None):
        model_name_or_path = resolve_checkpoint(model_name_or_path, cache_dir)
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

    @classmethod
    def from_engine(cls, engine: RkEngine, dims=None) -> "T5Runtime":
        """Wrap an engine that is already loaded (bench.py, tools/): no checkpoint directory involved."""
        self = cls.__new__(cls)
        self.dims = dims if dims is not None else engine.dims
        self.config, self.model_type, self.decoder_start_token_id = self.dims.to_hf_config(), "t5", 0
        self.max_tokens, self.max_seqs = int(engine.desc.max_tokens), int(engine.desc.max_seqs)
        self.engine = engine
        return self

    # -- multi-GPU: scores are collected by the engine's own RCCL communicator (rk_comm_*) --------------------
    COMM_FLOATS_PER_RANK = 65536      # send-buffer capacity (256 KB per rank): hits / world x outputs per passage

    def comm_ready(self) -> bool:
        """an engine communicator exists (any world size: one rank gathers with itself over the same calls)"""
        return self.comm_capacity > 0

    def comm_rank_world(self):
        """(rank, world) of the engine communicator"""
        return int(getattr(self.engine, "comm_rank", 0)), int(getattr(self.engine, "comm_world", 1))

    def comm_init_from_process_group(self, max_floats_per_rank: int = 0):
        """One process per GPU under torchrun: take rank / world from the initialised torch.distributed group, use it
        ONLY to hand rank 0's RCCL id to the other ranks, and build the engine's communicator (collective)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        ids = [self.engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        self.engine.comm_init(ids[0], rank, world, max_floats_per_rank or self.COMM_FLOATS_PER_RANK)

    @property
    def comm_capacity(self) -> int:
        """Floats per rank the live communicator can ship - the ENGINE's figure, whoever built the communicator (this
        runtime, bench.py or a tool through engine.comm_init): one value for the check and the message on every rank."""
        return int(getattr(self.engine, "comm_capacity", 0) or 0)

    def ensure_comm(self) -> bool:
        """Called by a candidate-sharding ranker before its first sharded query: under an initialised process group of
        more than one rank the engine communicator is built once (every rank gets here - the call is collective).
        Returns comm_ready().  A failure to bring RCCL up raises: there is no silent host-side substitute on a GPU run."""
        if self.comm_ready():
            return True
        try:
            import torch.distributed as dist
        except Exception:
            return False
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return False
        self.comm_init_from_process_group()
        return self.comm_ready()

    def sharded_scores(self, kind: str, seqs, arg, out_ids, width_floats: int, tail=None, tail_offset: int = 0):
        """This rank's share of one query -> (local raw outputs, allv [world, width_floats]) with ONE RCCL all_gather.
        The share may take several engine calls (more than max_seqs sequences / max_tokens tokens): each call's scores are
        appended to the engine's send buffer on the device (rk_comm_append_scores_slot), then the whole share is shipped.
        Every rank issues exactly one collective per query whatever its chunk count (also with an empty share).
        `tail`: host floats that travel in the same gather at `tail_offset` of this rank's row (the per-passage token counts
        behind the reference's counters).  width_floats is the same on every rank (it follows from the candidate count and
        the world size alone), so the capacity check below fails on ALL ranks or none - never inside the collective."""
        if width_floats > self.comm_capacity:
            raise ValueError(f"{width_floats} floats per rank exceed the communicator's send buffer ({self.comm_capacity}); "
                             "build it with comm_init_from_process_group(max_floats_per_rank=...)")
        k = len(out_ids) if kind == "score" else 1
        if tail is not None and len(tail):
            self.engine.comm_append_host(np.asarray(tail, dtype=np.float32), tail_offset)
        parts, off = [], 0
        for chunk in self._chunks(seqs):
            part = self.engine.qlm(chunk, arg) if kind == "qlm" else self.engine.score(chunk, arg, out_ids)
            n = len(chunk) * k
            self.engine.comm_append(n, off, slot=0)              # blocking calls leave their scores in slot 0
            parts.append(np.asarray(part, dtype=np.float32).reshape(-1))
            off += n
        local = np.concatenate(parts) if parts else np.zeros(0, np.float32)
        return local, self.engine.comm_all_gather_appended(width_floats)

    def all_gather_last_scores(self, n_floats: int) -> np.ndarray:
        """[world, n_floats]: the first n_floats of slot 0's device score buffer of every rank (one RCCL all_gather; a
        share that fits ONE engine call - bench.py's grouped launches; rankers use sharded_scores)."""
        self.engine.comm_all_gather(n_floats, slot=0)
        return self.engine.comm_read_gathered(0)

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

    def score_stream(self, groups, dec_prefix, out_ids) -> np.ndarray:
        """Scores [n, len(out_ids)] of the token sequences that the iterable `groups` yields (lists of sequences, e.g. one
        query's prompts at a time), in order.  `groups` is consumed LAZILY: as soon as the sequences pulled so far fill an
        engine call (max_seqs / max_tokens - the same greedy cut as _chunks over the flat list) the call is staged and
        enqueued on the next batch slot, and the host goes on pulling - i.e. tokenising - the following sequences while the
        GPU works (rk_t5_stage_slot / rk_t5_score_slot return at once; only reading a slot's scores waits).  A call of six
        queries thus leaves the tokenisation of the first launch sequence exposed instead of all of it.  Same bits as
        score() on the flat list: a sequence's scores do not depend on what shares its call."""
        eng = self.engine
        n_slots = eng.num_slots
        parts, pending = [], []                                         # slots in submission order
        launched = 0
        cur, tok = [], 0

        def launch(chunk):
            nonlocal launched
            slot = launched % n_slots
            if len(pending) == n_slots:                                # the slot we are about to reuse must be drained
                parts.append(eng.read_scores(pending.pop(0)))
            eng.stage(chunk, slot=slot)
            eng.score_staged(dec_prefix, out_ids, slot=slot)
            pending.append(slot)
            launched += 1

        try:
            for group in groups:
                for s in group:
                    if len(s) > self.max_tokens:
                        raise ValueError(f"a prompt of {len(s)} tokens exceeds the engine capacity {self.max_tokens}")
                    if cur and (tok + len(s) > self.max_tokens or len(cur) == self.max_seqs):
                        launch(cur)
                        cur, tok = [], 0
                    cur.append(s)
                    tok += len(s)
            if cur:
                launch(cur)
        finally:
            # whatever happened above (a prompt too long, the caller's generator raising): no slot stays in flight
            for s0 in pending:
                parts.append(eng.read_scores(s0))
        return np.concatenate(parts, axis=0) if parts else np.zeros((0, len(out_ids)), np.float32)

    def score_batches(self, batches, dec_prefix, out_ids) -> List[np.ndarray]:
        """Score several independent batches, keeping the engine's batch slots full: batch i+1 is staged and its
        encoder enqueued while the decoder chain of batch i still runs (rk_t5_stage_slot / rk_t5_score_slot).
        Results are identical to calling score() per batch; only the waiting is overlapped."""
        # The reference's batch_size only shapes its host loop; results do not depend on batch composition (ragged
        # execution), so consecutive batches are merged up to the engine's capacity: one launch sequence then covers
        # ~one query's candidates - better GEMM tile quantisation and ONE decoder chain instead of one per batch.
        batches = list(batches)
        allsc = self.score_stream(batches, dec_prefix, out_ids)
        out, pos = [], 0
        for b in batches:
            out.append(allsc[pos:pos + len(b)])
            pos += len(b)
        return out

    def score_async(self, seqs, dec_prefix, out_ids, slot: int):
        """Enqueue ONE engine call for `seqs` on batch slot `slot` and return at once -> a handle for score_collect, or None
        when the sequences do not fit one call (the caller then uses score(), which cuts them).  The caller owns the slot
        until it has collected: nothing else may be launched on it, and no blocking call may run, in between."""
        if not (0 < len(seqs) <= self.max_seqs and sum(len(s) for s in seqs) <= self.max_tokens):
            return None
        self.engine.stage(seqs, slot=slot)
        self.engine.score_staged(dec_prefix, out_ids, slot=slot)
        return slot

    def score_collect(self, handle) -> np.ndarray:
        """[n, len(out_ids)] scores of the call score_async enqueued (waits for it)."""
        return self.engine.read_scores(handle)

    def qlm(self, seqs, labels) -> np.ndarray:
        return np.concatenate([self.engine.qlm(c, labels) for c in self._chunks(seqs)], axis=0)

    def qlm_batches(self, batches, labels) -> List[np.ndarray]:
        """qlm scores of several batches of ONE query (same labels): the reference's batch_size only shapes its host loop and a
        passage's score does not depend on what shares its call, so the batches go to the engine merged up to its capacity -
        one encoder / decoder / head sequence over the query's candidates instead of one per batch of 32."""
        batches = list(batches)
        allsc = self.qlm([s for b in batches for s in b], labels) if batches else np.zeros(0, np.float32)
        out, pos = [], 0
        for b in batches:
            out.append(allsc[pos:pos + len(b)])
            pos += len(b)
        return out

    supports_greedy_candidates = True

    def greedy(self, seqs, dec_prefix, max_new, eos_id=1, pad_id=0, candidates=None) -> np.ndarray:
        """[B, max_new] new tokens; columns after the step at which every row had finished hold -1.  `candidates`: see
        RkEngine.greedy (a hint that never changes the result)."""
        parts = []
        for c in self._chunks(seqs):
            toks, steps = self.engine.greedy(c, dec_prefix, max_new, eos_id, pad_id, candidates)
            toks = toks.copy()
            toks[:, steps:] = -1
            parts.append(toks)
        return np.concatenate(parts, axis=0)


class LlamaRuntime:
    """Decoder-only (Llama family) counterpart of T5Runtime: checkpoint directory -> rk_llama_* engine.  Replaces
    `AutoModelForCausalLM.from_pretrained(..., device_map='auto', torch_dtype=fp16)` of ref: llmrankers/setwise.py:65-69."""

    def __init__(self, model_name_or_path: str, device, max_tokens: int = 32768, max_seqs: int = 16, cache_dir=None):
        model_name_or_path = resolve_checkpoint(model_name_or_path, cache_dir)
        cfg = read_config(model_name_or_path)
        self.model_type = cfg.get("model_type")
        if self.model_type != "llama":
            raise NotImplementedError(f"Model type {self.model_type} is not supported yet by the MI355X engine")
        self.config = cfg
        self.dims = _synth.LlamaDims.from_hf_config(cfg)
        self.max_tokens, self.max_seqs = max_tokens, max_seqs
        self.engine = RkLlamaEngine(self.dims, parse_device(device), max_tokens, max_seqs)
        self.engine.load_state(iter_checkpoint_tensors(model_name_or_path))

    @classmethod
    def from_engine(cls, engine: RkLlamaEngine, dims=None) -> "LlamaRuntime":
        self = cls.__new__(cls)
        self.dims = dims if dims is not None else engine.dims
        self.config, self.model_type = self.dims.to_hf_config(), "llama"
        self.max_tokens, self.max_seqs = int(engine.desc.max_tokens), int(engine.desc.max_seqs)
        self.engine = engine
        return self

    _chunks = T5Runtime._chunks

    def greedy1(self, seqs) -> np.ndarray:
        """next token (first arg-max of the last position's logits) of every prompt"""
        return np.concatenate([self.engine.greedy1(c) for c in self._chunks(seqs)], axis=0)

    def last_logits(self, seqs, out_ids) -> np.ndarray:
        return np.concatenate([self.engine.last_logits(c, out_ids) for c in self._chunks(seqs)], axis=0)


def load_runtime(model_name_or_path: str, device, cache_dir=None):
    """T5Runtime or LlamaRuntime by the checkpoint's config.model_type (ref: setwise.py:40-71 dispatches the same way);
    anything else raises NotImplementedError like the reference."""
    path = resolve_checkpoint(model_name_or_path, cache_dir)
    mt = read_config(path).get("model_type")
    if mt == "t5":
        return T5Runtime(path, device, cache_dir=cache_dir)
    if mt == "llama":
        return LlamaRuntime(path, device, cache_dir=cache_dir)
    raise NotImplementedError(f"Model type {mt} is not supported yet by the MI355X engine")
