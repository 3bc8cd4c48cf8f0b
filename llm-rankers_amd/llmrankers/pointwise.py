"""Pointwise rankers on the MI355X engine: yes_no, qlm and monoT5 scoring.

Drop-in for ref: llmrankers/pointwise.py — same constructor signature, attributes, counters, in-place score
mutation and stable descending sort.  What differs is underneath: prompts are tokenised once, batches are
ragged (no padding, no DataLoader workers) and each batch is ONE call into the HIP engine, which returns only
the vocabulary rows the method reads (yes/no ids) instead of materialising [B, 1, 32128] logits.
"""
from typing import List

import numpy as np

from ._batching import batches, padded_token_count, tokenize_prompts
from .rankers import LlmRanker, SearchResult

YES_NO_PROMPT = "Passage: {text}\nQuery: {query}\nDoes the passage answer the query? Answer 'Yes' or 'No'"
QLM_PROMPT = "Passage: {text}\nPlease write a question based on this passage."
MONOT5_PROMPT = "Query: {query} Document: {document} Relevant:"


def _softmax_first(a: np.ndarray, b: np.ndarray, fp16: bool = False) -> np.ndarray:
    """softmax([a, b])[0] in float32, the way torch.softmax evaluates it (max-subtracted).  fp16=True reproduces the
    reference's accelerator path (ref: pointwise.py:22-23: fp16 model on 'cuda'): logits rounded to fp16, softmax
    evaluated in fp32 and rounded to fp16 as torch does for half tensors - scores then take the reference's quantised
    values (0.9995, 1.0, ...) and exact ties are resolved by input order exactly as there."""
    if fp16:
        a, b = a.astype(np.float16), b.astype(np.float16)
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    m = np.maximum(a, b)
    ea, eb = np.exp(a - m), np.exp(b - m)
    p = ea / (ea + eb)
    return p.astype(np.float16).astype(np.float32) if fp16 else p


class PointwiseLlmRanker(LlmRanker):
    # Scores are fp32 softmaxes of the engine's fp32 logits (closest to the reference's CPU ground truth).  Set to True to
    # get the reference's 'cuda' score values instead: fp16 logits and fp16 probabilities, hence its exact ties.
    fp16_scores = False

    def __init__(self, model_name_or_path, tokenizer_name_or_path, device, method="qlm", batch_size=1, cache_dir=None,
                 shard_candidates=False):
        # ref: pointwise.py:13-34: tokenizer + model from the checkpoint; here the model is the HIP engine
        from transformers import T5Tokenizer
        from ._runtime import T5Runtime
        tokenizer = T5Tokenizer.from_pretrained(
            tokenizer_name_or_path if tokenizer_name_or_path is not None else model_name_or_path, cache_dir=cache_dir)
        runtime = T5Runtime(model_name_or_path, device, cache_dir=cache_dir)   # NotImplementedError for non-T5 models
        self._setup(runtime, tokenizer, device, method, batch_size, shard_candidates)

    @classmethod
    def from_runtime(cls, runtime, tokenizer, device="cuda", method="qlm", batch_size=1, shard_candidates=False):
        """Build the ranker around an existing runtime (an engine that is already loaded, or a test double) and
        tokenizer instead of loading a checkpoint: several rankers can share one engine."""
        self = cls.__new__(cls)
        self._setup(runtime, tokenizer, device, method, batch_size, shard_candidates)
        return self

    def _setup(self, runtime, tokenizer, device, method, batch_size, shard_candidates):
        self.tokenizer = tokenizer
        self.llm = runtime
        self.config = getattr(runtime, "config", None)
        self.device = device
        self.method = method
        self.batch_size = batch_size
        # opt-in data parallelism (one process per GPU under torchrun): each rank scores a contiguous chunk of the
        # candidate list, one all_gather collects the scores AND the prompts' token counts (llmrankers/_dist.py), so
        # scores, rank order and the three counters are those of a single-GPU run on every rank.
        self.shard_candidates = shard_candidates
        self.total_compare = 0
        self.total_completion_tokens = 0
        self.total_prompt_tokens = 0

    def _reset(self):
        self.total_compare = 0
        self.total_completion_tokens = 0
        self.total_prompt_tokens = 0

    # -- what a method scores: prompts, the engine call shape, and how raw engine output becomes a score ---------
    def _spec(self, query: str, docs: List[SearchResult]):
        """-> (prompts, kind, arg, out_ids, dec_len, finish) or None for a method the reference ignores.
        kind 'score': arg = decoder prefix, raw = logits [n, len(out_ids)];  kind 'qlm': arg = labels, raw = [n]."""
        if self.method == "qlm":
            # ref: pointwise.py:41-82 — score = -sum_t CE(label_t), labels = "<pad> {query}" without specials
            labels = self.tokenizer.encode(f"<pad> {query}", add_special_tokens=False)
            prompts = [QLM_PROMPT.format(text=doc.text) for doc in docs]
            return prompts, "qlm", labels, None, len(labels), lambda raw: np.asarray(raw, dtype=np.float32).reshape(-1)
        if self.method == "yes_no":
            # ref: pointwise.py:84-127 — score = softmax([logit_yes, logit_no])[0] at the first decoder step
            yes_id = self.tokenizer.encode("Yes", add_special_tokens=False)[0]
            no_id = self.tokenizer.encode("No", add_special_tokens=False)[0]
            prompts = [YES_NO_PROMPT.format(text=doc.text, query=query) for doc in docs]
            return prompts, "score", [self.tokenizer.pad_token_id], [yes_id, no_id], 1, \
                lambda raw: _softmax_first(raw[:, 0], raw[:, 1], self.fp16_scores)
        return None   # any other method: the reference silently leaves the scores untouched and still sorts (ref :129)

    def _counted_batches(self, prompts: List[str], dec_len: int):
        """Tokenise once, cut into the reference's batches and keep its counters (ref: pointwise.py:105-114)."""
        seqs = tokenize_prompts(self.tokenizer, prompts)
        out = []
        for s, e in batches(len(seqs), self.batch_size):
            chunk = seqs[s:e]
            self.total_compare += 1
            self.total_prompt_tokens += padded_token_count(chunk)
            self.total_prompt_tokens += len(chunk) * dec_len     # decoder inputs count as prompt (ref :68,114)
            out.append(chunk)
        return out

    def _count_from_lengths(self, lengths, dec_len: int):
        """The same counters from the prompts' token counts alone (sharded runs: every rank tokenises only its share and
        learns the other lengths from the gather): one compare per reference batch, batch-longest padding (ref :105-114)."""
        for s, e in batches(len(lengths), self.batch_size):
            self.total_compare += 1
            self.total_prompt_tokens += (e - s) * int(max(lengths[s:e])) + (e - s) * dec_len

    def _raw(self, chunks, kind, arg, out_ids):
        """Raw engine output for all batches of one query, concatenated in passage order."""
        if not chunks:
            return np.zeros((0, len(out_ids)) if kind == "score" else (0,), np.float32)
        if kind == "qlm":
            if hasattr(self.llm, "qlm_batches"):                 # one engine call for the query's batches (batch independence)
                return np.concatenate([np.asarray(x, dtype=np.float32) for x in self.llm.qlm_batches(chunks, arg)])
            return np.concatenate([np.asarray(self.llm.qlm(c, arg), dtype=np.float32) for c in chunks])
        if hasattr(self.llm, "score_batches"):                   # pipelined across the engine's batch slots
            return np.concatenate(self.llm.score_batches(chunks, arg, out_ids), axis=0)
        return np.concatenate([self.llm.score(c, arg, out_ids) for c in chunks], axis=0)

    def _rerank_sharded(self, query: str, ranking: List[SearchResult]) -> List[SearchResult]:
        """One process per GPU: this rank scores its contiguous chunk of the candidates, ONE all_gather collects the
        raw engine outputs of all ranks - issued by the engine itself over RCCL (T5Runtime.ensure_comm builds the
        communicator on the first sharded query; a share that needs several engine calls is appended on the device and
        shipped whole), or through torch.distributed when the runtime has no communicator (the CPU test doubles) - and
        every rank finishes and sorts identically.  Replaces accelerate's device_map='auto' layer placement of
        ref: llmrankers/pointwise.py:20-24."""
        from . import _dist
        rank, ws = _dist.world()
        if ws == 1 and getattr(self.llm, "comm_ready", lambda: False)():
            rank, ws = self.llm.comm_rank_world()               # a communicator built without a torch process group
        bounds = _dist.shard_bounds(len(ranking), ws)
        s, e = bounds[rank]
        width = max(b - a for a, b in bounds)
        self._reset()
        if width == 0:                                           # no candidates at all: nothing to score or gather on any rank
            return []
        spec = self._spec(query, ranking[s:e])
        if spec is None:
            return sorted(ranking, key=lambda x: x.score, reverse=True)
        prompts, kind, arg, out_ids, dec_len, finish = spec
        flat = tokenize_prompts(self.tokenizer, prompts)
        k = len(out_ids) if kind == "score" else 1
        # a rank's row of the gather: its scores [width * k], then its prompts' token counts [width] (exact in fp32) - the
        # counters of the whole query (ref: pointwise.py:105-114) need every passage's length, and every rank reports them
        lens = np.asarray([len(q) for q in flat], dtype=np.float32)
        row = width * (k + 1)
        # the engine builds its RCCL communicator on the first sharded query (collective: every rank is here)
        if getattr(self.llm, "ensure_comm", lambda: False)():
            local, allv = self.llm.sharded_scores(kind, flat, arg, out_ids, row, tail=lens, tail_offset=width * k)
            assert len(local) == (e - s) * k
            allv = np.asarray(allv, dtype=np.float32).reshape(ws, row)
        else:
            chunks = [flat[i:j] for i, j in batches(len(flat), self.batch_size)]
            local = np.zeros(row, np.float32)
            local[:(e - s) * k] = self._raw(chunks, kind, arg, out_ids).reshape(-1)
            local[width * k:width * k + (e - s)] = lens
            # a runtime without a communicator of its own (the CPU tests' oracle-backed doubles) brings its gather along; the real
            # runtime never gets here (T5Runtime.ensure_comm builds the engine's RCCL communicator or raises)
            gather = getattr(self.llm, "host_all_gather", None)
            if gather is None:
                raise RuntimeError("candidate sharding needs a runtime with a communicator (T5Runtime) or a host_all_gather of its own")
            allv = gather(local, row)
        raw = np.concatenate([allv[r, :(b - a) * k] for r, (a, b) in enumerate(bounds)])
        all_lens = np.concatenate([allv[r, width * k:width * k + (b - a)] for r, (a, b) in enumerate(bounds)])
        self._count_from_lengths(np.rint(all_lens).astype(np.int64), dec_len)
        scores = finish(raw.reshape(-1, k) if kind == "score" else raw)
        for doc, sc in zip(ranking, scores):
            doc.score = float(sc)
        return sorted(ranking, key=lambda x: x.score, reverse=True)

    def _rerank_sharded_many(self, items):
        """Candidate sharding for SEVERAL queries per engine launch sequence (round 6): every rank takes its contiguous share of
        EVERY query (100 candidates over 8 ranks: 13,13,13,13,12,12,12,12 each), scores all the shares in one launch sequence -
        sixteen queries are 208 passages per rank instead of 13, the regime the GEMMs are built for - and ONE all_gather carries
        the raw outputs and the prompts' token counts of all of them.  A rank's row: its scores packed in query order
        [sum_q n_rq * k], then at offset W * k (W = sum_q of the widest share) the token counts [sum_q n_rq].  Rankings, scores
        and counters are those of `rerank` query by query (a passage's score does not depend on what shares its call).
        Returns None when the queries cannot share a call (mixed call shapes, a row beyond the communicator's capacity)."""
        from . import _dist
        rank, ws = _dist.world()
        if ws == 1 and getattr(self.llm, "comm_ready", lambda: False)():
            rank, ws = self.llm.comm_rank_world()
        plans, W = [], 0
        for query, ranking in items:
            bounds = _dist.shard_bounds(len(ranking), ws)
            s, e = bounds[rank]
            spec = self._spec(query, ranking[s:e])
            if spec is None or spec[1] != "score" or (plans and (spec[2], spec[3]) != (plans[0][2][2], plans[0][2][3])):
                return None
            plans.append((bounds, (s, e), spec))
            W += max(b - a for a, b in bounds)
        if not plans or W == 0:
            return None
        arg, out_ids, dec_len = plans[0][2][2], plans[0][2][3], plans[0][2][4]
        k = len(out_ids)
        row = W * (k + 1)
        have_comm = getattr(self.llm, "ensure_comm", lambda: False)()
        if have_comm and row > self.llm.comm_capacity:
            return None
        flat = [q for _, _, spec in plans for q in tokenize_prompts(self.tokenizer, spec[0])]
        lens = np.asarray([len(q) for q in flat], dtype=np.float32)
        if have_comm:
            local, allv = self.llm.sharded_scores("score", flat, arg, out_ids, row, tail=lens, tail_offset=W * k)
            assert len(local) == len(flat) * k
            allv = np.asarray(allv, dtype=np.float32).reshape(ws, row)
        else:
            gather = getattr(self.llm, "host_all_gather", None)
            if gather is None:
                raise RuntimeError("candidate sharding needs a runtime with a communicator (T5Runtime) or a host_all_gather of its own")
            chunks = [flat[i:j] for i, j in batches(len(flat), self.batch_size)]
            local = np.zeros(row, np.float32)
            local[:len(flat) * k] = self._raw(chunks, "score", arg, out_ids).reshape(-1)
            local[W * k:W * k + len(flat)] = lens
            allv = gather(local, row)
        out, counters = [], []
        pos = [0] * ws                                                    # passages of rank r consumed so far
        for (query, ranking), (bounds, _, spec) in zip(items, plans):
            raw = np.concatenate([allv[r, pos[r] * k:(pos[r] + b - a) * k] for r, (a, b) in enumerate(bounds)])
            all_lens = np.concatenate([allv[r, W * k + pos[r]:W * k + pos[r] + b - a] for r, (a, b) in enumerate(bounds)])
            for r, (a, b) in enumerate(bounds):
                pos[r] += b - a
            self._reset()
            self._count_from_lengths(np.rint(all_lens).astype(np.int64), dec_len)
            counters.append((self.total_compare, self.total_prompt_tokens, self.total_completion_tokens))
            for doc, sc in zip(ranking, spec[5](raw.reshape(-1, k))):
                doc.score = float(sc)
            out.append(sorted(ranking, key=lambda x: x.score, reverse=True))
        return out, counters

    def rerank(self, query: str, ranking: List[SearchResult]) -> List[SearchResult]:
        if self.shard_candidates:
            from . import _dist
            # more than one rank - or an engine communicator that already exists (a one-rank communicator walks the same
            # append / gather path: that is how the GPU tests reach it on a one-GPU box)
            if _dist.world()[1] > 1 or getattr(self.llm, "comm_ready", lambda: False)():
                return self._rerank_sharded(query, ranking)
        self._reset()
        spec = self._spec(query, ranking)
        if spec is not None:
            prompts, kind, arg, out_ids, dec_len, finish = spec
            raw = self._raw(self._counted_batches(prompts, dec_len), kind, arg, out_ids)
            for doc, sc in zip(ranking, finish(raw)):
                doc.score = float(sc)
        return sorted(ranking, key=lambda x: x.score, reverse=True)

    def rerank_many(self, items):
        """Several queries per engine launch sequence: `items` = [(query, ranking), ...] -> (rankings, counters) where
        rankings[i] is exactly what `rerank(*items[i])` returns (same objects scored in place, same order) and counters[i]
        = (total_compare, total_prompt_tokens, total_completion_tokens) that call would have left.  The reference ranks one
        query at a time (ref: run.py:183-201); a passage's score does not depend on what shares its engine call (ragged
        execution, bit-exact), so the batches of all the queries go to the engine together - its GEMMs then run on several
        hundred passages instead of a hundred and the decoder chain runs once (grouped launches).
        Candidate-sharded runs group the same way (_rerank_sharded_many: every rank's share of every query in one launch
        sequence, ONE all_gather); qlm (per-query labels) and unknown methods take the one-by-one path."""
        items = list(items)
        specs = []
        if self.shard_candidates and items:
            from . import _dist
            if _dist.world()[1] > 1 or getattr(self.llm, "comm_ready", lambda: False)():
                res = self._rerank_sharded_many(items)                # every rank's share of every query in one launch sequence
                if res is not None:
                    return res
        grouped = not self.shard_candidates
        for query, ranking in items:
            spec = self._spec(query, ranking) if grouped else None
            specs.append(spec)
            if spec is None or spec[1] != "score" or (specs[0] is not None and (spec[2], spec[3]) != (specs[0][2], specs[0][3])):
                grouped = False
        if not grouped or not items:
            out, counters = [], []
            for query, ranking in items:
                out.append(self.rerank(query, ranking))
                counters.append((self.total_compare, self.total_prompt_tokens, self.total_completion_tokens))
            return out, counters
        counters, sizes = [], []

        def tokenised_queries():
            # one query's prompts at a time: the runtime launches as soon as an engine call is full and keeps pulling, so the
            # later queries are tokenised while the GPU already scores the first ones (T5Runtime.score_stream)
            for (query, ranking), spec in zip(items, specs):
                self._reset()
                prompts, _, _, _, dec_len, _ = spec
                chunks = self._counted_batches(prompts, dec_len)
                counters.append((self.total_compare, self.total_prompt_tokens, self.total_completion_tokens))
                sizes.append(len(prompts))
                yield chunks

        if hasattr(self.llm, "score_stream"):
            raw = self.llm.score_stream(([s for c in chunks for s in c] for chunks in tokenised_queries()), specs[0][2], specs[0][3])
        else:                                                        # a runtime without batch slots (test doubles): batch by batch
            raw = self._raw([c for chunks in tokenised_queries() for c in chunks], "score", specs[0][2], specs[0][3])
        out, pos = [], 0
        for (query, ranking), spec, n in zip(items, specs, sizes):
            for doc, sc in zip(ranking, spec[5](raw[pos:pos + n])):
                doc.score = float(sc)
            pos += n
            out.append(sorted(ranking, key=lambda x: x.score, reverse=True))
        return out, counters

    def truncate(self, text, length):
        return self.tokenizer.convert_tokens_to_string(self.tokenizer.tokenize(text)[:length])


class MonoT5LlmRanker(PointwiseLlmRanker):
    """ref: pointwise.py:136-186 — softmax over the fixed ids of 'false'/'true', decoder start token as input;
    the reference's rerank ignores `method`."""
    FALSE_ID, TRUE_ID = 6136, 1176

    def _spec(self, query: str, docs: List[SearchResult]):
        prompts = [MONOT5_PROMPT.format(query=query, document=doc.text) for doc in docs]
        return prompts, "score", [self.llm.decoder_start_token_id], [self.FALSE_ID, self.TRUE_ID], 1, \
            lambda raw: _softmax_first(raw[:, 1], raw[:, 0], self.fp16_scores)
