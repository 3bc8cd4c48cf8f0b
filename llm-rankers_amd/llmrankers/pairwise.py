"""Pairwise ranking prompting (PRP) on the MI355X engine: allpair / heapsort / bubblesort over "which of the two
passages is more relevant" calls.

Drop-in for ref: llmrankers/pairwise.py:29-295 (PairwiseLlmRanker, T5 family) — same constructor, prompt, `compare()`
contract (two prompts, A/B and B/A, decoded generations returned), counters and result assembly.  The model call is the
same primitive the setwise ranker uses: greedy continuation of "<pad> Passage" by two tokens (rk_t5_greedy), here for
the two orderings of a pair in one engine call, and for `allpair` for as many pairs as fit the engine's capacity at
once (the reference's batch_size only shapes its host loop and the padded-shape counters, reproduced arithmetically).
Llama-family checkpoints (ref: pairwise.py:60-77, 104-129) take the decoder-only path of the setwise ranker: chat-template
prompt + " Passage:", prefill and ONE greedy token per ordering (rk_llama_greedy1), outputs "Passage <token>"; `allpair`
is T5-only there too (it reads `self.decoder_input_ids`, which the reference only sets for T5: AttributeError).
"""
from itertools import combinations
from typing import List

import numpy as np

from ._batching import batches, padded_token_count, tokenize_prompts
from .rankers import LlmRanker, SearchResult

PROMPT = ('Given a query "{query}", which of the following two passages is more relevant to the query?\n\n'
          'Passage A: "{doc1}"\n\nPassage B: "{doc2}"\n\nOutput Passage A or Passage B:')
WIN = ["Passage A", "Passage B"]       # the first passage of the pair wins iff the A/B prompt says A and the B/A prompt says B


class PairwiseLlmRanker(LlmRanker):

    def __init__(self, model_name_or_path, tokenizer_name_or_path, device, method="allpair", batch_size=2, k=10,
                 cache_dir=None):
        # ref: pairwise.py:30-82: T5 or Llama family by config.model_type, NotImplementedError otherwise
        from ._runtime import load_runtime
        try:
            runtime = load_runtime(model_name_or_path, device, cache_dir=cache_dir)
        except NotImplementedError as exc:
            raise NotImplementedError(f"{exc} (pairwise)") from None
        if runtime.model_type == "llama":
            from transformers import AutoTokenizer
            from .setwise import VICUNA_TEMPLATE
            tokenizer = AutoTokenizer.from_pretrained(model_name_or_path, cache_dir=cache_dir)   # (the reference ignores tokenizer_name_or_path here)
            tokenizer.use_default_system_prompt = False
            if 'v1.5' in model_name_or_path:       # the reference's `'vicuna' and 'v1.5' in name` (ref :62)
                tokenizer.chat_template = VICUNA_TEMPLATE
            tokenizer.pad_token = "[PAD]"          # ref :65-66 (never used: both orderings of a pair are scored unpadded)
            tokenizer.padding_side = "left"
        else:
            from transformers import T5Tokenizer
            tokenizer = T5Tokenizer.from_pretrained(
                tokenizer_name_or_path if tokenizer_name_or_path is not None else model_name_or_path, cache_dir=cache_dir)
        self._setup(runtime, tokenizer, device, method, batch_size, k)

    @classmethod
    def from_runtime(cls, runtime, tokenizer, device="cuda", method="allpair", batch_size=2, k=10):
        """Build the ranker around an existing runtime (a loaded engine, or a test double) and tokenizer."""
        self = cls.__new__(cls)
        self._setup(runtime, tokenizer, device, method, batch_size, k)
        return self

    def _setup(self, runtime, tokenizer, device, method, batch_size, k):
        self.device, self.method, self.batch_size, self.k = device, method, batch_size, k
        self.prompt = PROMPT
        self.llm, self.tokenizer = runtime, tokenizer
        self.config = getattr(runtime, "config", None)
        self.model_type = getattr(runtime, "model_type", "t5")
        if self.model_type == "t5":                         # (ref :53-56: only the T5 branch has a decoder prompt)
            self.decoder_input_ids = self.tokenizer.encode("<pad> Passage", add_special_tokens=False)
            # the two labels a generation normally starts with: hint for the one-pass two-token greedy (rk_t5_greedy2)
            self._label_ids = [self.tokenizer.encode(f"<pad> Passage {c}", add_special_tokens=False)[-1] for c in "AB"]
        self.total_compare = 0
        self.total_completion_tokens = 0
        self.total_prompt_tokens = 0

    # -- the model call --------------------------------------------------------------------------------------
    def _generate(self, token_lists: List[List[int]]):
        """Greedy, max_new_tokens=2, continuing "<pad> Passage" for every prompt.  Returns (texts, new_lens): the decoded
        generation of each row (prefix included, specials skipped - what batch_decode gives the reference) and the
        number of new tokens each row needed (EOS included), from which the reference's batch-level output length
        follows: HF stops a batch when all its rows have finished."""
        eos, pad = self.tokenizer.eos_token_id, self.tokenizer.pad_token_id
        if getattr(self.llm, "supports_greedy_candidates", False):
            new = np.asarray(self.llm.greedy(token_lists, self.decoder_input_ids, 2, eos, pad, candidates=self._label_ids))
        else:
            new = np.asarray(self.llm.greedy(token_lists, self.decoder_input_ids, 2, eos, pad))
        texts, lens = [], []
        for row in new:
            toks = [int(t) for t in row if t >= 0]
            n = toks.index(eos) + 1 if eos in toks else 2
            lens.append(n)
            texts.append(self.tokenizer.decode(list(self.decoder_input_ids) + toks[:n], skip_special_tokens=True))
        return texts, lens

    def compare(self, query: str, docs: List):
        # ref: pairwise.py:84-131 — docs = the two passage TEXTS; both orderings in one call
        self.total_compare += 1
        texts = [self.prompt.format(query=query, doc1=docs[0], doc2=docs[1]),
                 self.prompt.format(query=query, doc1=docs[1], doc2=docs[0])]
        if self.model_type == "llama":
            # ref :104-129: chat template + " Passage:", one greedy token per ordering.  The reference tokenises the two prompts
            # into ONE tensor without padding - they hold the same words in another order and normally the same number of
            # tokens (it raises when they do not; here the counters then take the longer one, as a padded batch would)
            ids = []
            for t in texts:
                prompt = self.tokenizer.apply_chat_template([{"role": "user", "content": t}], tokenize=False, add_generation_prompt=True)
                ids.append(list(self.tokenizer(prompt + " Passage:")["input_ids"]))
            width = max(len(x) for x in ids)
            self.total_prompt_tokens += len(ids) * width
            toks = self.llm.greedy1(ids)
            self.total_completion_tokens += len(ids) * (width + 1)      # generate() returns prompt + new token
            return [f"Passage {self.tokenizer.decode([int(t)], skip_special_tokens=True).strip().upper()}" for t in toks]
        ids = tokenize_prompts(self.tokenizer, texts)
        self.total_prompt_tokens += padded_token_count(ids)                 # padding='longest' (ref :93-97)
        out, lens = self._generate(ids)
        self.total_completion_tokens += len(ids) * (len(self.decoder_input_ids) + max(lens))
        return out

    def _first_wins(self, query, a_text, b_text) -> bool:
        return self.compare(query, [a_text, b_text]) == WIN

    # -- sort drivers --------------------------------------------------------------------------------------------
    def _heapify(self, arr, n, i, gt):
        # ref: pairwise.py:133-147 (binary max-heap sift-down; at most two comparisons per level, in this order)
        while True:
            largest, left, right = i, 2 * i + 1, 2 * i + 2
            if left < n and gt(arr[left], arr[i]):
                largest = left
            if right < n and gt(arr[right], arr[largest]):
                largest = right
            if largest == i:
                return
            arr[i], arr[largest] = arr[largest], arr[i]
            i = largest

    def _heap_sort(self, arr, k, gt):
        # ref: pairwise.py:149-162
        n = len(arr)
        for i in range(n // 2, -1, -1):
            self._heapify(arr, n, i, gt)
        ranked = 0
        for i in range(n - 1, 0, -1):
            arr[i], arr[0] = arr[0], arr[i]
            ranked += 1
            if ranked == k:
                break
            self._heapify(arr, i, 0, gt)

    def _allpair(self, query, ranking):
        # ref: pairwise.py:169-216 — every unordered pair in both orders, one generation each; a win needs both
        # orderings to agree, anything else is half a point each
        pairs = list(combinations(ranking, 2))
        prompts = []
        for d1, d2 in pairs:
            prompts.append(self.prompt.format(query=query, doc1=d1.text, doc2=d2.text))
            prompts.append(self.prompt.format(query=query, doc1=d2.text, doc2=d1.text))
        seqs = tokenize_prompts(self.tokenizer, prompts)
        outputs, lens = self._generate(seqs) if seqs else ([], [])
        for s, e in batches(len(seqs), self.batch_size):        # the reference's batches only shape its counters
            self.total_compare += 1
            self.total_prompt_tokens += padded_token_count(seqs[s:e])
            self.total_completion_tokens += (e - s) * (len(self.decoder_input_ids) + max(lens[s:e]))
        scores = {}
        for i, (d1, d2) in enumerate(pairs):
            o1, o2 = outputs[2 * i], outputs[2 * i + 1]
            if o1 == "Passage A" and o2 == "Passage B":
                scores[d1.docid] = scores.get(d1.docid, 0.0) + 1
            elif o1 == "Passage B" and o2 == "Passage A":
                scores[d2.docid] = scores.get(d2.docid, 0.0) + 1
            else:
                scores[d1.docid] = scores.get(d1.docid, 0.0) + 0.5
                scores[d2.docid] = scores.get(d2.docid, 0.0) + 0.5
        # documents enter in the order they first score (dict order) and the sort is stable, as in the reference
        return sorted([SearchResult(docid=d, score=s, text=None) for d, s in scores.items()], key=lambda x: x.score, reverse=True)

    def rerank(self, query: str, ranking: List[SearchResult]) -> List[SearchResult]:
        # ref: pairwise.py:164-295
        original_docids = [doc.docid for doc in ranking]      # (the reference deep-copies the list; only the docid order is read)
        self.total_compare = 0
        self.total_completion_tokens = 0
        self.total_prompt_tokens = 0
        if self.method == "allpair":
            ranking = self._allpair(query, ranking)
        elif self.method == "heapsort":
            arr = list(ranking)
            self._heap_sort(arr, self.k, lambda a, b: self._first_wins(query, a.text, b.text))
            ranking = [SearchResult(docid=d.docid, score=-i, text=None) for i, d in enumerate(reversed(arr))]
        elif self.method == "bubblesort":
            # ref: pairwise.py:246-269 — the reference's variant that skips pairs already known to be in order
            k = min(self.k, len(ranking))
            last_end = len(ranking) - 1
            for i in range(k):
                cur, changed = last_end, False
                while cur > i:
                    if self._first_wins(query, ranking[cur].text, ranking[cur - 1].text):
                        ranking[cur - 1], ranking[cur] = ranking[cur], ranking[cur - 1]
                        if not changed:
                            changed = True
                            if last_end != len(ranking) - 1:
                                last_end += 1
                    if not changed:
                        last_end -= 1
                    cur -= 1
        else:
            raise NotImplementedError(f'Method {self.method} is not implemented.')
        results, top, rank = [], set(), 1
        for doc in ranking[:self.k]:
            top.add(doc.docid)
            results.append(SearchResult(docid=doc.docid, score=-rank, text=None))
            rank += 1
        for docid in original_docids:
            if docid not in top:
                results.append(SearchResult(docid=docid, score=-rank, text=None))
                rank += 1
        return results

    def truncate(self, text, length):
        return self.tokenizer.convert_tokens_to_string(self.tokenizer.tokenize(text)[:length])
