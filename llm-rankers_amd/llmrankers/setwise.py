"""Setwise ranker on the MI355X engine: c-ary heapsort / bubblesort over an LLM "which passage is most relevant" call.

Drop-in for ref: llmrankers/setwise.py:21-316 (SetwiseLlmRanker) — same constructor, `compare()` contract,
counters, fallbacks for malformed model output and result assembly.  Each compare is one encoder pass over a
single long prompt plus either two greedy decoder steps (`scoring='generation'`, engine call rk_t5_greedy) or
one label-row read at decoder position 1 (`scoring='likelihood'`, rk_t5_score with the 23 label ids).
Llama-family models (ref: setwise.py:60-69,159-177): chat-template prompt + " Passage:", prefill and ONE greedy token
(rk_llama_greedy1); `likelihood` scoring raises NotImplementedError for them exactly as in the reference.
"""
import random
from collections import Counter
from typing import List

import numpy as np

from ._batching import tokenize_prompts
from .rankers import LlmRanker, SearchResult

random.seed(929)   # same import-time seeding as the reference (ref: setwise.py:18): permutation voting depends on it

# the chat template the reference installs for vicuna-v1.5 checkpoints (ref: setwise.py:63-64) — data, quoted as is
VICUNA_TEMPLATE = ("{% if messages[0]['role'] == 'system' %}{% set loop_messages = messages[1:] %}{% set system_message = messages[0]['content'] %}"
                   "{% else %}{% set loop_messages = messages %}{% set system_message = 'A chat between a curious user and an artificial intelligence "
                   "assistant. The assistant gives helpful, detailed, and polite answers to the user\\'s questions.' %}{% endif %}"
                   "{% for message in loop_messages %}{% if (message['role'] == 'user') != (loop.index0 % 2 == 0) %}"
                   "{{ raise_exception('Conversation roles must alternate user/assistant/user/assistant/...') }}{% endif %}"
                   "{% if loop.index0 == 0 %}{{ system_message }}{% endif %}{% if message['role'] == 'user' %}{{ ' USER: ' + message['content'].strip() }}"
                   "{% elif message['role'] == 'assistant' %}{{ ' ASSISTANT: ' + message['content'].strip() + eos_token }}{% endif %}{% endfor %}"
                   "{% if add_generation_prompt %}{{ ' ASSISTANT:' }}{% endif %}")

QUESTION = 'Given a query "{query}", which of the following passages is the most relevant one to the query?\n\n'
INSTRUCTION = '\n\nOutput only the passage label of the most relevant passage:'


class SetwiseLlmRanker(LlmRanker):
    # "Passage X" / "Passage Y" tokenize into 3 tokens with the T5 vocabulary, hence 23 labels (ref: setwise.py:22-23)
    CHARACTERS = ["A", "B", "C", "D", "E", "F", "G", "H", "I", "J", "K", "L",
                  "M", "N", "O", "P", "Q", "R", "S", "T", "U", "V", "W"]

    def __init__(self, model_name_or_path, tokenizer_name_or_path, device, num_child=3, k=10, scoring='generation',
                 method="heapsort", num_permutation=1, cache_dir=None):
        # ref: setwise.py:25-77: T5 or Llama family by config.model_type, NotImplementedError otherwise
        from ._runtime import load_runtime
        try:
            runtime = load_runtime(model_name_or_path, device, cache_dir=cache_dir)
        except NotImplementedError as exc:   # same message shape as ref: setwise.py:71
            raise NotImplementedError(f"{exc} (setwise)") from None
        if runtime.model_type == "llama":
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(model_name_or_path, cache_dir=cache_dir)   # (the reference ignores tokenizer_name_or_path here)
            tokenizer.use_default_system_prompt = False
            if 'v1.5' in model_name_or_path:       # the reference's `'vicuna' and 'v1.5' in name` (ref :63)
                tokenizer.chat_template = VICUNA_TEMPLATE
        else:
            from transformers import T5Tokenizer
            tokenizer = T5Tokenizer.from_pretrained(
                tokenizer_name_or_path if tokenizer_name_or_path is not None else model_name_or_path, cache_dir=cache_dir)
        self._setup(runtime, tokenizer, device, num_child, k, scoring, method, num_permutation)

    @classmethod
    def from_runtime(cls, runtime, tokenizer, device="cuda", num_child=3, k=10, scoring='generation', method="heapsort",
                     num_permutation=1):
        """Build the ranker around an existing runtime (a loaded engine, or a test double) and tokenizer."""
        self = cls.__new__(cls)
        self._setup(runtime, tokenizer, device, num_child, k, scoring, method, num_permutation)
        return self

    def _setup(self, runtime, tokenizer, device, num_child, k, scoring, method, num_permutation):
        self.device = device
        self.num_child = num_child
        self.num_permutation = num_permutation
        self.k = k
        self.llm = runtime
        self.config = getattr(runtime, "config", None)
        self.tokenizer = tokenizer
        self.model_type = getattr(runtime, "model_type", "t5")
        if self.model_type == "t5":
            # decoder prompt "<pad> Passage" and the last token of "<pad> Passage {label}" (ref: setwise.py:51-59)
            self.decoder_input_ids = self.tokenizer.encode("<pad> Passage", add_special_tokens=False)
            self.target_token_ids = [self.tokenizer.encode(f"<pad> Passage {c}", add_special_tokens=False)[-1]
                                     for c in self.CHARACTERS]
        self.scoring = scoring
        self.method = method
        # heapsort build phase: advance the independent sift-downs of a tree level in one engine call each (same
        # result, compares and counters as the reference's one-by-one order; 9 of the 29 compares at hits=100, c=10)
        self.batch_independent_compares = True
        self.total_compare = 0
        self.total_completion_tokens = 0
        self.total_prompt_tokens = 0

    # ------------------------------------------------------------------------------------------------------
    def _prompt(self, query: str, labels: List[str], texts: List[str]) -> str:
        passages = "\n\n".join(f'Passage {lab}: "{txt}"' for lab, txt in zip(labels, texts))
        return QUESTION.format(query=query) + passages + INSTRUCTION

    def _generate(self, token_lists: List[List[int]]) -> List[List[int]]:
        """Greedy, max_new_tokens=2, continuing "<pad> Passage".  Returns full output id rows the way
        HF generate does: prefix + new tokens, all rows cut at the step where every row had finished."""
        eos, pad = self.tokenizer.eos_token_id, self.tokenizer.pad_token_id
        if getattr(self.llm, "supports_greedy_candidates", False):
            # hint: the first new token is normally a passage label -> both steps in one decoder pass (rk_t5_greedy2)
            new = self.llm.greedy(token_lists, self.decoder_input_ids, 2, eos, pad,
                                  candidates=self.target_token_ids[:self.num_child + 1])
        else:
            new = self.llm.greedy(token_lists, self.decoder_input_ids, 2, eos, pad)
        out = []
        for row in np.asarray(new):
            out.append(list(self.decoder_input_ids) + [int(t) for t in row if t >= 0])
        return out

    def compare(self, query: str, docs: List):
        # ref: setwise.py:79-198
        self.total_compare += 1 if self.num_permutation == 1 else self.num_permutation
        n = len(docs)
        if self.model_type == "llama":
            # ref: setwise.py:159-177 — generation only; num_permutation is ignored (it only entered total_compare above)
            if self.scoring != 'generation':
                if self.scoring == 'likelihood':
                    raise NotImplementedError
                raise UnboundLocalError("local variable 'output' referenced before assignment")
            ids = self._llama_prompt_ids(self._prompt(query, self.CHARACTERS[:n], [d.text for d in docs]))
            self.total_prompt_tokens += len(ids)
            tok = int(self.llm.greedy1([ids])[0])
            self.total_completion_tokens += len(ids) + 1        # generate() returns prompt + new token for a decoder-only model
            output = self.tokenizer.decode([tok], skip_special_tokens=True).strip().upper()
            if not (len(output) == 1 and output in self.CHARACTERS):
                print(f"Unexpected output: {output}")
            return output
        if self.scoring == 'generation':
            if self.num_permutation == 1:
                text = self._prompt(query, self.CHARACTERS[:n], [d.text for d in docs])
                ids = tokenize_prompts(self.tokenizer, [text])
                self.total_prompt_tokens += len(ids[0])
                output_ids = self._generate(ids)[0]
                self.total_completion_tokens += len(output_ids)
                output = self.tokenizer.decode(output_ids, skip_special_tokens=True).strip()
                output = output[-1]
            else:
                id_passage = [(i, p) for i, p in enumerate(docs)]
                labels = [self.CHARACTERS[i] for i in range(n)]
                perms = []
                for _ in range(self.num_permutation):   # two draws per permutation, in this order (ref :107-109)
                    perms.append([random.sample(id_passage, len(id_passage)), random.sample(labels, len(labels))])
                refs, texts = [], []
                for shuffled, chars in perms:
                    refs.append(([p[0] for p in shuffled], list(chars)))
                    texts.append(self._prompt(query, list(chars), [p[1].text for p in shuffled]))
                ids = tokenize_prompts(self.tokenizer, texts)
                # return_tensors="pt" without padding requires equal lengths; permuting passages keeps them equal
                self.total_prompt_tokens += len(ids[0]) * len(ids)
                rows = self._generate(ids)
                plen = len(self.decoder_input_ids)
                decoded = self.tokenizer.batch_decode([r[plen:] for r in rows], skip_special_tokens=True)
                candidates = []
                for (docids, chars), result in zip(refs, decoded):
                    result = result.strip().upper()
                    if len(result) != 1 or result not in chars:
                        print(f"Unexpected output: {result}")
                        continue
                    candidates.append(docids[chars.index(result)])
                if len(candidates) == 0:
                    print(f"Unexpected voting: {decoded}")
                    output = "Unexpected voting."
                else:
                    counts = Counter(candidates)
                    top = max(counts.values())
                    winners = [c for c, v in counts.items() if v == top]
                    output = self.CHARACTERS[winners[0] if len(winners) == 1 else random.choice(winners)]
        elif self.scoring == 'likelihood':
            text = self._prompt(query, self.CHARACTERS[:n], [d.text for d in docs])
            ids = tokenize_prompts(self.tokenizer, [text])
            self.total_prompt_tokens += len(ids[0])
            # softmax over the vocabulary is monotone, so the best label is the arg-max of the label logits;
            # stable descending sort = first maximum wins (ref: setwise.py:184-188)
            if n == 0:
                raise IndexError("list index out of range")   # ranked[0] on an empty list in the reference (:188)
            lg = self.llm.score(ids, self.decoder_input_ids, self.target_token_ids[:n])[0]
            output = self.CHARACTERS[int(np.argmax(lg))]
        else:
            raise UnboundLocalError("local variable 'output' referenced before assignment")  # what the reference does

        if not (len(output) == 1 and output in self.CHARACTERS):
            print(f"Unexpected output: {output}")
        return output

    def _llama_prompt_ids(self, input_text: str) -> List[int]:
        """chat template + " Passage:" -> token ids, as ref: setwise.py:160-165 builds them"""
        prompt = self.tokenizer.apply_chat_template([{"role": "user", "content": input_text}], tokenize=False, add_generation_prompt=True)
        return list(self.tokenizer(prompt + " Passage:")["input_ids"])

    def _compare_many(self, query: str, doc_lists: List[List]) -> List[str]:
        """Independent compares in ONE engine call.  Same outputs and counters as `compare()` on each window in turn
        (num_permutation == 1 only: no random draws are involved); the engine's results do not depend on which
        prompts share a call (ragged execution, bit-exact batch independence)."""
        outs, prompt_tokens, completion_tokens = self._compare_windows([query] * len(doc_lists), doc_lists)
        self.total_compare += len(doc_lists)
        self.total_prompt_tokens += sum(prompt_tokens)
        self.total_completion_tokens += sum(completion_tokens)
        return outs

    def _compare_windows(self, queries: List[str], doc_lists: List[List]):
        """The engine call behind `_compare_many`, for windows that may belong to different queries (`rerank_many`):
        -> (labels, prompt tokens per window, completion tokens per window); touches no counter."""
        assert self.num_permutation == 1
        texts = [self._prompt(q, self.CHARACTERS[:len(docs)], [d.text for d in docs]) for q, docs in zip(queries, doc_lists)]
        if self.model_type == "llama":
            if self.scoring != 'generation':
                raise NotImplementedError
            ids = [self._llama_prompt_ids(t) for t in texts]
            toks = self.llm.greedy1(ids)
            outs = [self.tokenizer.decode([int(tok)], skip_special_tokens=True).strip().upper() for tok in toks]
            for output in outs:
                if not (len(output) == 1 and output in self.CHARACTERS):
                    print(f"Unexpected output: {output}")
            return outs, [len(seq) for seq in ids], [len(seq) + 1 for seq in ids]
        ids = tokenize_prompts(self.tokenizer, texts)
        prompt_tokens = [len(i) for i in ids]
        outs, completion_tokens = [], [0] * len(ids)
        if self.scoring == 'generation':
            eos = self.tokenizer.eos_token_id
            for r, row in enumerate(self._generate(ids)):
                new = row[len(self.decoder_input_ids):]
                if eos in new:                                  # alone, this row would have stopped at its own EOS
                    new = new[:new.index(eos) + 1]
                row = list(self.decoder_input_ids) + new
                completion_tokens[r] = len(row)
                outs.append(self.tokenizer.decode(row, skip_special_tokens=True).strip()[-1])
        elif self.scoring == 'likelihood':
            if any(len(docs) == 0 for docs in doc_lists):
                raise IndexError("list index out of range")
            nmax = max(len(docs) for docs in doc_lists)
            lg = np.asarray(self.llm.score(ids, self.decoder_input_ids, self.target_token_ids[:nmax]))
            outs = [self.CHARACTERS[int(np.argmax(lg[r, :len(docs)]))] for r, docs in enumerate(doc_lists)]
        else:
            raise UnboundLocalError("local variable 'output' referenced before assignment")
        for output in outs:
            if not (len(output) == 1 and output in self.CHARACTERS):
                print(f"Unexpected output: {output}")
        return outs, prompt_tokens, completion_tokens

    # ---- the same compare, launched and collected separately (likelihood scoring on a runtime with batch slots) ----------
    def _can_alternate(self) -> bool:
        """Two groups of lockstep queries can alternate over the engine's batch slots: T5 likelihood scoring (one engine call
        per step, no host decision inside it) on a runtime that launches without waiting (T5Runtime.score_async)."""
        return (self.model_type != "llama" and self.scoring == "likelihood" and getattr(self, "alternate_groups", True)
                and hasattr(self.llm, "score_async") and getattr(getattr(self.llm, "engine", None), "num_slots", 1) >= 2)

    def _launch_windows(self, queries: List[str], doc_lists: List[List], slot: int):
        """First half of `_compare_windows` (likelihood): prompts, tokens, ONE engine call enqueued on `slot`; returns at once.
        None when the windows do not fit one engine call (the caller then takes the blocking path)."""
        if any(len(docs) == 0 for docs in doc_lists):
            raise IndexError("list index out of range")
        texts = [self._prompt(q, self.CHARACTERS[:len(docs)], [d.text for d in docs]) for q, docs in zip(queries, doc_lists)]
        ids = tokenize_prompts(self.tokenizer, texts)
        nmax = max(len(docs) for docs in doc_lists)
        handle = self.llm.score_async(ids, self.decoder_input_ids, self.target_token_ids[:nmax], slot)
        if handle is None:
            return None
        return handle, [len(docs) for docs in doc_lists], [len(i) for i in ids]

    def _collect_windows(self, launched):
        """Second half: waits for the slot, -> (labels, prompt tokens per window, completion tokens per window)."""
        handle, sizes, prompt_tokens = launched
        lg = np.asarray(self.llm.score_collect(handle))
        outs = [self.CHARACTERS[int(np.argmax(lg[r, :n]))] for r, n in enumerate(sizes)]
        return outs, prompt_tokens, [0] * len(sizes)

    def _batched_ok(self) -> bool:
        # level-wise batching needs compare() to be ours (no subclass / instance override) and draw-free
        return (getattr(self, "batch_independent_compares", False) and self.num_permutation == 1
                and "compare" not in self.__dict__ and type(self).compare is SetwiseLlmRanker.compare)

    def _build_heap_batched(self, arr, n, query):
        """Build phase of the c-ary heapsort with the sift-downs of one tree level advanced together.  The reference
        walks i = n//c .. 0 (ref: setwise.py:221-223), i.e. level by level from the deepest internal one; nodes of a
        level root disjoint subtrees, so their sift-down chains touch disjoint array slots and commute: the array,
        the set of compares and every counter end up identical, only the order of compares inside a level differs."""
        c = self.num_child
        levels = {}
        for i in range(n // c, -1, -1):
            if c * i + 1 < n:
                depth, first = 0, 0
                while i >= first + c ** depth:      # nodes of depth d occupy [first, first + c^d)
                    first += c ** depth
                    depth += 1
                levels.setdefault(depth, []).append(i)
        for depth in sorted(levels, reverse=True):
            active = levels[depth]
            while active:
                windows = [[i] + list(range(c * i + 1, min(c * (i + 1) + 1, n))) for i in active]
                outs = self._compare_many(query, [[arr[j] for j in inds] for inds in windows])
                nxt = []
                for i, inds, out in zip(active, windows, outs):
                    best = self._pick(out)
                    largest = inds[best] if best < len(inds) else i
                    if largest != i:
                        arr[i], arr[largest] = arr[largest], arr[i]
                        if c * largest + 1 < n:
                            nxt.append(largest)
                active = nxt

    # ---- sort drivers: pure index logic, must reproduce the reference's comparisons exactly ---------------
    def _pick(self, output: str) -> int:
        try:
            return self.CHARACTERS.index(output)
        except ValueError:
            return 0                              # malformed output -> first document wins (ref :206-209)

    def heapify(self, arr, n, i, query):
        # ref: setwise.py:200-217, written as a loop instead of tail recursion
        c = self.num_child
        while c * i + 1 < n:
            hi = min(c * (i + 1) + 1, n)
            inds = [i] + list(range(c * i + 1, hi))
            best = self._pick(self.compare(query, [arr[j] for j in inds]))
            largest = inds[best] if best < len(inds) else i   # label beyond the window keeps the parent (ref :210-213)
            if largest == i:
                return
            arr[i], arr[largest] = arr[largest], arr[i]
            i = largest

    def heapSort(self, arr, query, k):
        # ref: setwise.py:219-232
        n = len(arr)
        if self._batched_ok():
            self._build_heap_batched(arr, n, query)
        else:
            for i in range(n // self.num_child, -1, -1):
                self.heapify(arr, n, i, query)
        ranked = 0
        for i in range(n - 1, 0, -1):
            arr[i], arr[0] = arr[0], arr[i]
            ranked += 1
            if ranked == k:
                break
            self.heapify(arr, i, 0, query)

    def _bubblesort(self, ranking, query):
        # ref: setwise.py:243-273 — sliding window of num_child+1 bubbling the best document to position i,
        # with the reference's `last_start` shortcut that skips windows already known to be in order.
        c = self.num_child
        full = len(ranking) - (c + 1)
        last_start = full
        for i in range(self.k):
            start, end = last_start, last_start + (c + 1)
            changed = False
            while True:
                if start < i:
                    start = i
                window = ranking[start:end]
                best = self._pick(self.compare(query, window))
                if best != 0:
                    # no guard here in the reference either: an out-of-window label raises IndexError
                    ranking[start], ranking[start + best] = ranking[start + best], ranking[start]
                    if not changed:
                        changed = True
                        if last_start != full and best == len(window) - 1:
                            last_start += len(window) - 1
                if start == i:
                    break
                if not changed:
                    last_start -= c
                start -= c
                end -= c

    def rerank(self, query: str, ranking: List[SearchResult]) -> List[SearchResult]:
        # ref: setwise.py:234-313.  NB: like the reference, the caller's list is re-ordered in place.
        original_docids = [doc.docid for doc in ranking]     # (the reference deep-copies the whole list: 3 ms for 100 passages; only the docid order is read)
        self.total_compare = 0
        self.total_completion_tokens = 0
        self.total_prompt_tokens = 0
        if self.method == "heapsort":
            self.heapSort(ranking, query, self.k)
            ranking = list(reversed(ranking))
        elif self.method == "bubblesort":
            self._bubblesort(ranking, query)
        else:
            raise NotImplementedError(f'Method {self.method} is not implemented.')
        results, top_doc_ids, rank = [], set(), 1
        for doc in ranking[:self.k]:
            top_doc_ids.add(doc.docid)
            results.append(SearchResult(docid=doc.docid, score=-rank, text=None))
            rank += 1
        for docid in original_docids:
            if docid not in top_doc_ids:
                results.append(SearchResult(docid=docid, score=-rank, text=None))
                rank += 1
        return results

    # ---- several queries at once ---------------------------------------------------------------------------
    def _heapsort_steps(self, arr, k):
        """heapSort(arr, query, k) as a generator: yields lists of windows whose compares are independent of each other
        (one tree level of the build phase, then one sift-down step at a time) and is sent their labels.  Same array
        updates, compares and order as `_build_heap_batched` + `heapify` (the shipped single-query path)."""
        c, n = self.num_child, len(arr)
        levels = {}
        for i in range(n // c, -1, -1):
            if c * i + 1 < n:
                depth, first = 0, 0
                while i >= first + c ** depth:
                    first += c ** depth
                    depth += 1
                levels.setdefault(depth, []).append(i)
        for depth in sorted(levels, reverse=True):
            active = levels[depth]
            while active:
                windows = [[i] + list(range(c * i + 1, min(c * (i + 1) + 1, n))) for i in active]
                outs = yield [[arr[j] for j in inds] for inds in windows]
                nxt = []
                for i, inds, out in zip(active, windows, outs):
                    best = self._pick(out)
                    largest = inds[best] if best < len(inds) else i
                    if largest != i:
                        arr[i], arr[largest] = arr[largest], arr[i]
                        if c * largest + 1 < n:
                            nxt.append(largest)
                active = nxt
        ranked = 0
        for m in range(n - 1, 0, -1):
            arr[m], arr[0] = arr[0], arr[m]
            ranked += 1
            if ranked == k:
                break
            i = 0
            while c * i + 1 < m:
                inds = [i] + list(range(c * i + 1, min(c * (i + 1) + 1, m)))
                (out,) = yield [[arr[j] for j in inds]]
                best = self._pick(out)
                largest = inds[best] if best < len(inds) else i
                if largest == i:
                    break
                arr[i], arr[largest] = arr[largest], arr[i]
                i = largest

    def _bubblesort_steps(self, ranking):
        """`_bubblesort` as a generator (one window per step; same windows, swaps and `last_start` shortcut)."""
        c = self.num_child
        full = len(ranking) - (c + 1)
        last_start = full
        for i in range(self.k):
            start, end = last_start, last_start + (c + 1)
            changed = False
            while True:
                if start < i:
                    start = i
                window = ranking[start:end]
                (out,) = yield [window]
                best = self._pick(out)
                if best != 0:
                    ranking[start], ranking[start + best] = ranking[start + best], ranking[start]
                    if not changed:
                        changed = True
                        if last_start != full and best == len(window) - 1:
                            last_start += len(window) - 1
                if start == i:
                    break
                if not changed:
                    last_start -= c
                start -= c
                end -= c

    def rerank_many(self, items):
        """Several queries at once: `items` = [(query, ranking), ...] -> (results, counters); results[i] and counters[i] =
        (total_compare, total_prompt_tokens, total_completion_tokens) are exactly what `rerank(*items[i])` gives, and the
        callers' lists end up re-ordered the same way.  The compares of ONE query are a dependency chain (each sift-down step
        needs the previous label), but the chains of different queries are independent: their pending compares go to the
        engine together, one call per step of all the chains - several ~900-token prompts per launch sequence instead of one
        (a compare's result does not depend on what shares its engine call: ragged execution, bit-exact).
        heapsort and bubblesort with the draw-free default settings; anything else takes the one-by-one path."""
        items = list(items)
        if self.method not in ("heapsort", "bubblesort") or not self._batched_ok() or len(items) < 2:
            out, counters = [], []
            for query, ranking in items:
                out.append(self.rerank(query, ranking))
                counters.append((self.total_compare, self.total_prompt_tokens, self.total_completion_tokens))
            return out, counters
        originals = [[doc.docid for doc in ranking] for _, ranking in items]
        counts = [[0, 0, 0] for _ in items]
        heap = self.method == "heapsort"
        gens = [self._heapsort_steps(ranking, self.k) if heap else self._bubblesort_steps(ranking) for _, ranking in items]
        pending = {}
        for q, gen in enumerate(gens):
            try:
                pending[q] = next(gen)
            except StopIteration:
                pass

        def windows_of(group):
            order = sorted(group)
            return order, [items[q][0] for q in order for _ in group[q]], [w for q in order for w in group[q]]

        def advance(group, order, outs, ptok, ctok):
            """hand every chain of the group its labels -> the group's next pending windows"""
            pos, nxt = 0, {}
            for q in order:
                n = len(group[q])
                counts[q][0] += n
                counts[q][1] += sum(ptok[pos:pos + n])
                counts[q][2] += sum(ctok[pos:pos + n])
                try:
                    nxt[q] = gens[q].send(outs[pos:pos + n])
                except StopIteration:
                    pass
                pos += n
            return nxt

        if len(pending) >= 4 and self._can_alternate():
            # Two groups of chains alternate over the engine's two batch slots: while one group's call is on the GPU the host
            # advances the other group's heaps, builds and tokenises its prompts and launches them - the launch-bound decoder
            # chain of one call runs under the encoder of the next, and the host part of a step is hidden.  A chain sees the
            # same labels as alone (batch independence), so rankings and counters do not change (tests).
            groups = [{}, {}]
            for i, q in enumerate(sorted(pending)):
                groups[i % 2][q] = pending[q]
            inflight = []                                            # [(group index, order, launched)] oldest first

            def submit(g) -> bool:
                order, queries, windows = windows_of(groups[g])
                launched = self._launch_windows(queries, windows, slot=g)
                if launched is None:                                 # does not fit one engine call: back to the blocking loop below
                    return False
                inflight.append((g, order, launched))
                return True

            ok = True
            try:
                for g in (0, 1):
                    if ok and groups[g]:
                        ok = submit(g)
                while ok and inflight:
                    g, order, launched = inflight.pop(0)
                    groups[g] = advance(groups[g], order, *self._collect_windows(launched))
                    if groups[g]:
                        ok = submit(g)
                while inflight:                                      # (only after a call that did not fit)
                    g, order, launched = inflight.pop(0)
                    groups[g] = advance(groups[g], order, *self._collect_windows(launched))
            finally:
                # whatever raised above (a tokenizer error, a capacity error of the other group's launch): a call still queued on
                # its slot is collected before the exception leaves - score_async's contract is that nothing else runs on the
                # engine until then, and the caller's next rerank() would
                for _, _, launched in inflight:
                    try:
                        self._collect_windows(launched)
                    except Exception:
                        pass
            pending = {**groups[0], **groups[1]}
        while pending:
            order, queries, windows = windows_of(pending)
            pending = advance(pending, order, *self._compare_windows(queries, windows))
        results = []
        for (query, ranking), original in zip(items, originals):
            ordered = list(reversed(ranking)) if heap else ranking
            res, top, rank = [], set(), 1
            for doc in ordered[:self.k]:
                top.add(doc.docid)
                res.append(SearchResult(docid=doc.docid, score=-rank, text=None))
                rank += 1
            for docid in original:
                if docid not in top:
                    res.append(SearchResult(docid=docid, score=-rank, text=None))
                    rank += 1
            results.append(res)
        counters = [tuple(c) for c in counts]
        if counters:
            self.total_compare, self.total_prompt_tokens, self.total_completion_tokens = counters[-1]
        return results, counters

    def truncate(self, text, length):
        return self.tokenizer.convert_tokens_to_string(self.tokenizer.tokenize(text)[:length])
