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


def _softmax_first(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """softmax([a, b])[0] in float32, the way torch.softmax evaluates it (max-subtracted)."""
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    m = np.maximum(a, b)
    ea, eb = np.exp(a - m), np.exp(b - m)
    return ea / (ea + eb)


class PointwiseLlmRanker(LlmRanker):

    def __init__(self, model_name_or_path, tokenizer_name_or_path, device, method="qlm", batch_size=1, cache_dir=None,
                 _runtime=None, _tokenizer=None, shard_candidates=False):
        # ref: pointwise.py:13-34.  `_runtime` / `_tokenizer` are test seams; production builds the HIP engine.
        if _tokenizer is None:
            from transformers import T5Tokenizer
            _tokenizer = T5Tokenizer.from_pretrained(
                tokenizer_name_or_path if tokenizer_name_or_path is not None else model_name_or_path,
                cache_dir=cache_dir)
        self.tokenizer = _tokenizer
        if _runtime is None:
            from ._runtime import T5Runtime
            _runtime = T5Runtime(model_name_or_path, device)   # raises NotImplementedError for non-T5 models
        self.llm = _runtime
        self.config = getattr(_runtime, "config", None)
        self.device = device
        self.method = method
        self.batch_size = batch_size
        # opt-in data parallelism (one process per GPU under torchrun): each rank scores a contiguous chunk of the
        # candidate list, one all_gather collects the scores (llmrankers/_dist.py).  Counters then describe the
        # local chunk; scores and rank order are identical to a single-GPU run.
        self.shard_candidates = shard_candidates
        self.total_compare = 0
        self.total_completion_tokens = 0
        self.total_prompt_tokens = 0

    def _reset(self):
        self.total_compare = 0
        self.total_completion_tokens = 0
        self.total_prompt_tokens = 0

    def _scored_batches(self, prompts: List[str], dec_len: int):
        """Yield (start, token lists) per batch while keeping the reference's counters."""
        seqs = tokenize_prompts(self.tokenizer, prompts)
        for s, e in batches(len(seqs), self.batch_size):
            chunk = seqs[s:e]
            self.total_compare += 1
            self.total_prompt_tokens += padded_token_count(chunk)
            self.total_prompt_tokens += len(chunk) * dec_len     # decoder inputs count as prompt (ref :68,114)
            yield s, chunk

    def _rerank_sharded(self, query: str, ranking: List[SearchResult]) -> List[SearchResult]:
        from . import _dist

        def score_chunk(chunk):
            if not len(chunk):
                return []
            saved, self.shard_candidates = self.shard_candidates, False
            try:
                self.rerank(query, list(chunk))          # mutates the chunk's objects in place
            finally:
                self.shard_candidates = saved
            return [d.score for d in chunk]

        scores = _dist.sharded_scores(score_chunk, ranking)
        for doc, sc in zip(ranking, scores):
            doc.score = float(sc)
        return sorted(ranking, key=lambda x: x.score, reverse=True)

    def _score_all(self, chunks, dec, out_ids):
        """All batches of one query through the engine; pipelined across its batch slots when the runtime can."""
        if hasattr(self.llm, "score_batches"):
            return self.llm.score_batches(chunks, dec, out_ids)
        return [self.llm.score(c, dec, out_ids) for c in chunks]

    def rerank(self, query: str, ranking: List[SearchResult]) -> List[SearchResult]:
        if self.shard_candidates:
            from . import _dist
            if _dist.world()[1] > 1:
                return self._rerank_sharded(query, ranking)
        self._reset()
        if self.method == "qlm":
            # ref: pointwise.py:41-82 — score = -sum_t CE(label_t), labels = "<pad> {query}" without specials
            labels = self.tokenizer.encode(f"<pad> {query}", add_special_tokens=False)
            prompts = [QLM_PROMPT.format(text=doc.text) for doc in ranking]
            for s, chunk in self._scored_batches(prompts, len(labels)):
                scores = self.llm.qlm(chunk, labels)
                for i, sc in enumerate(scores):
                    ranking[s + i].score = float(sc)
        elif self.method == "yes_no":
            # ref: pointwise.py:84-127 — score = softmax([logit_yes, logit_no])[0] at the first decoder step
            yes_id = self.tokenizer.encode("Yes", add_special_tokens=False)[0]
            no_id = self.tokenizer.encode("No", add_special_tokens=False)[0]
            prompts = [YES_NO_PROMPT.format(text=doc.text, query=query) for doc in ranking]
            dec = [self.tokenizer.pad_token_id]
            todo = list(self._scored_batches(prompts, 1))
            for (s, chunk), lg in zip(todo, self._score_all([c for _, c in todo], dec, [yes_id, no_id])):
                p_yes = _softmax_first(lg[:, 0], lg[:, 1])
                for i, sc in enumerate(p_yes):
                    ranking[s + i].score = float(sc)
        # any other method: the reference silently leaves the scores untouched and still sorts (ref :129)
        return sorted(ranking, key=lambda x: x.score, reverse=True)

    def truncate(self, text, length):
        return self.tokenizer.convert_tokens_to_string(self.tokenizer.tokenize(text)[:length])


class MonoT5LlmRanker(PointwiseLlmRanker):
    """ref: pointwise.py:136-186 — softmax over the fixed ids of 'false'/'true', decoder start token as input."""
    FALSE_ID, TRUE_ID = 6136, 1176

    def rerank(self, query: str, ranking: List[SearchResult]) -> List[SearchResult]:
        self._reset()
        prompts = [MONOT5_PROMPT.format(query=query, document=doc.text) for doc in ranking]
        dec = [self.llm.decoder_start_token_id]
        todo = list(self._scored_batches(prompts, 1))
        for (s, chunk), lg in zip(todo, self._score_all([c for _, c in todo], dec, [self.FALSE_ID, self.TRUE_ID])):
            p_true = _softmax_first(lg[:, 1], lg[:, 0])
            for i, sc in enumerate(p_true):
                ranking[s + i].score = float(sc)
        return sorted(ranking, key=lambda x: x.score, reverse=True)
