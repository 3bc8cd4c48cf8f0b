"""Host-side batching that reproduces the reference's tokenise-all-then-pad-per-batch behaviour WITHOUT padding.

The reference tokenises every prompt at once with no truncation (ref: llmrankers/pairwise.py:17-26, EOS is
appended by the tokenizer), then a DataLoader + DataCollatorWithPadding(padding='longest') right-pads each
batch (ref: llmrankers/pointwise.py:90-101), forking 4 worker processes per rerank() call.  The engine takes
ragged token lists, so nothing is padded and nothing is forked; only the integer bookkeeping the reference
derives from the padded shape (total_prompt_tokens, ref: pointwise.py:107,114) is reproduced here.
"""
import json
import re
import os
import warnings
import weakref
from itertools import chain
from typing import Iterator, List, Sequence, Tuple


def _tokenize_full(tokenizer, prompts: Sequence[str]) -> List[List[int]]:
    return [list(ids) for ids in tokenizer(list(prompts))["input_ids"]]


_PLAIN_RE = re.compile(r"[\x20-\x7e\n]*")      # printable ASCII and newlines (one C-level scan per prompt)


class _MemoMismatch(Exception):
    pass


# normalisers that cannot act across a word boundary of printable-ASCII text (character-level maps); a Replace only in the
# T5 form that collapses runs of blanks (the words are the same) or with a blank-free literal pattern
_SAFE_NORMALIZERS = {"NFC", "NFD", "NFKC", "NFKD", "Precompiled", "Nmt", "Lowercase", "StripAccents"}


def _normalizer_is_word_local(norm) -> bool:
    if norm is None:
        return True
    t = norm.get("type")
    if t == "Sequence":
        return all(_normalizer_is_word_local(n) for n in norm.get("normalizers", []))
    if t in _SAFE_NORMALIZERS:
        return True
    if t == "Replace":
        pat = norm.get("pattern") or {}
        if pat.get("Regex") == " {2,}" and norm.get("content") == " ":
            return True
        lit = pat.get("String")
        return isinstance(lit, str) and lit != "" and not any(c.isspace() for c in lit) and not any(c.isspace() for c in norm.get("content", ""))
    return False


class WordSpliceTokenizer:
    """Exact memo of a whitespace-pre-tokenised sentencepiece tokenizer (host tokenisation).

    A setwise query re-tokenises the same hundred passages in ~30 different ten-passage prompts, one ~900-token prompt per
    `compare` call (ref: llmrankers/setwise.py:113-115): 1.4 ms of host time each, a quarter of the query once the
    forward takes 4 ms.  The T5 tokenizers split the normalised text at whitespace FIRST (`WhitespaceSplit`), prepend
    the metaspace to every piece and run the Unigram model per piece, then append `</s>`: the ids of a prompt are the
    concatenation of the ids of its words, so each distinct word is tokenised once and a prompt becomes dictionary
    look-ups.  Only tokenizers whose backend says exactly that are handled (anything else - BPE with a regex
    pre-tokenizer, a Metaspace without the whitespace split - goes to the tokenizer as before), and the result is
    checked as it runs: (a) every NEW word is verified IN CONTEXT the first time it is seen - the window {previous word,
    the word, next word} of the first prompt that contains it goes to the tokenizer in the same call that tokenises the
    word alone, and its ids must be the concatenation of the three words' ids; (b) whole prompts are compared with the
    tokenizer's own on the first prompts and then at a fixed stride.  The first difference switches the memo off for good
    and the pending call is re-tokenised by the tokenizer itself, so a call never returns ids the check has just
    contradicted; the warning says how many prompts had been spliced since the last whole-prompt comparison.
    `RK_TOKEN_CACHE=0` disables the memo.
    """
    # words are cut at ' ' and '\n' only: separators both before and after the normaliser; other blanks stay inside a "word"
    MAX_WORDS = 2_000_000

    def __init__(self, tokenizer, verify_first: int = 4, verify_every: int = 128):
        self.tokenizer = tokenizer
        self.words = {}
        self.verify_first, self.verify_every = verify_first, verify_every
        self.seen = 0
        self.since_check = 0            # prompts returned since the last whole-prompt comparison (reported on a mismatch)
        self.suffix: List[int] = []
        self.enabled = os.environ.get("RK_TOKEN_CACHE", "1") != "0" and self._eligible(tokenizer)
        if self.enabled:
            try:
                self.suffix = list(tokenizer("")["input_ids"])
                probe = "memo probe: \"two words\"\n\nnext"
                self.enabled = self._splice([probe]) == _tokenize_full(tokenizer, [probe])
            except Exception:
                self.enabled = False

    @staticmethod
    def _eligible(tokenizer) -> bool:
        backend = getattr(tokenizer, "backend_tokenizer", None)
        if backend is None or not hasattr(backend, "to_str"):
            return False
        try:
            cfg = json.loads(backend.to_str())
            pre = cfg.get("pre_tokenizer") or {}
            seq = pre.get("pretokenizers") if pre.get("type") == "Sequence" else [pre]
            if not seq or seq[0].get("type") != "WhitespaceSplit":
                return False
            for p in seq[1:]:
                if p.get("type") != "Metaspace" or p.get("prepend_scheme", "always") != "always":
                    return False
            if (cfg.get("model") or {}).get("type") != "Unigram":
                return False
            if not _normalizer_is_word_local(cfg.get("normalizer")):
                return False        # e.g. a regex Replace that can match across a blank: words are not independent
            post = cfg.get("post_processor")
            if post is not None:
                single = post.get("single") if post.get("type") == "TemplateProcessing" else None
                if not single or "Sequence" not in single[0] or any("SpecialToken" not in x for x in single[1:]):
                    return False        # specials in front of the text, or something this memo does not model
            return True
        except Exception:
            return False

    def _splice(self, prompts: Sequence[str]) -> List[List[int]]:
        # (spliced prompts are printable ASCII + newlines - _plain - so str.split() cuts exactly at the blanks and newlines the
        # pre-tokenizer cuts at, in C)
        pieces = [p.split() for p in prompts]
        words = self.words
        suffix = self.suffix
        try:                                       # the common case: every word is known - one pass, no membership scan
            return [list(chain.from_iterable(map(words.__getitem__, ws))) + suffix for ws in pieces]
        except KeyError:
            pass
        missing = {w for ws in pieces for w in ws if w not in words}
        if len(words) + len(missing) > self.MAX_WORDS:
            words.clear()
            missing = {w for ws in pieces for w in ws}
        order = list(missing)
        # the first context of every new word: (previous word, the word, next word) of the first prompt that holds it
        windows, todo = [], set(missing)
        for ws in pieces:
            if not todo:
                break
            for i, w in enumerate(ws):
                if w in todo:
                    todo.discard(w)
                    if len(ws) > 1:
                        windows.append(ws[max(0, i - 1):i + 2])
        got = self.tokenizer(order + [" ".join(win) for win in windows], add_special_tokens=False)["input_ids"]
        for w, ids in zip(order, got):
            words[w] = tuple(ids)
        for win, ids in zip(windows, got[len(order):]):
            if list(chain.from_iterable(map(words.__getitem__, win))) != list(ids):
                raise _MemoMismatch(" ".join(win))
        return [list(chain.from_iterable(map(words.__getitem__, ws))) + suffix for ws in pieces]

    @staticmethod
    def _plain(prompt: str) -> bool:
        """Only prompts of printable ASCII and newlines are spliced: for them the tokenizers' normaliser (NFKC, the
        control-character rules) is the identity and cannot reach across a word boundary; anything else - combining marks,
        other blanks, NUL ... - goes to the tokenizer itself."""
        return _PLAIN_RE.fullmatch(prompt) is not None

    def __call__(self, prompts: Sequence[str]) -> List[List[int]]:
        if not self.enabled:
            return _tokenize_full(self.tokenizer, prompts)
        plain = [self._plain(p) for p in prompts]
        if not all(plain):
            rest = [p for p, ok in zip(prompts, plain) if not ok]
            rest_ids = iter(_tokenize_full(self.tokenizer, rest))
            plain_ids = iter(self([p for p, ok in zip(prompts, plain) if ok]) if any(plain) else [])
            return [next(plain_ids) if ok else next(rest_ids) for ok in plain]
        try:
            out = self._splice(prompts)
            for i in range(len(prompts)):
                n = self.seen + i
                if n < self.verify_first or n % self.verify_every == 0:
                    if _tokenize_full(self.tokenizer, [prompts[i]])[0] != out[i]:
                        raise _MemoMismatch(prompts[i][:80])
                    self.since_check = -i - 1          # (prompts of this call behind i are counted below)
        except _MemoMismatch as exc:
            warnings.warn(f"word-level token memo disagrees with the tokenizer ({str(exc)!r}); switched off - this call is "
                          f"tokenised by the tokenizer itself; {max(self.since_check, 0)} prompt(s) of earlier calls had been "
                          "spliced since the last whole-prompt comparison")
            self.enabled = False
            self.words.clear()
            return _tokenize_full(self.tokenizer, prompts)
        self.seen += len(prompts)
        self.since_check += len(prompts)
        return out


_MEMOS = weakref.WeakKeyDictionary()


def tokenize_prompts(tokenizer, prompts: Sequence[str]) -> List[List[int]]:
    """`tokenizer(prompts)["input_ids"]` (no truncation, specials added), through the exact word memo when it applies."""
    if not prompts:
        return []
    try:
        memo = _MEMOS.get(tokenizer)
        if memo is None:
            memo = _MEMOS[tokenizer] = WordSpliceTokenizer(tokenizer)
    except TypeError:                   # not hashable / not weak-referenceable: plain path
        return _tokenize_full(tokenizer, prompts)
    return memo(prompts)


def batches(n_items: int, batch_size: int) -> Iterator[Tuple[int, int]]:
    """[start, end) ranges in order; the last one may be short (drop_last=False, shuffle=False)."""
    for s in range(0, n_items, batch_size):
        yield s, min(s + batch_size, n_items)


def padded_token_count(seqs: Sequence[Sequence[int]]) -> int:
    """B x L_longest: what `input_ids.shape[0] * input_ids.shape[1]` is after the reference's collator."""
    return len(seqs) * max(len(s) for s in seqs) if seqs else 0


def default_queries_per_call(kind: str, hits: int) -> int:
    """How many queries run.py hands to a ranker's rerank_many at once when --queries_per_call is left at 0 (auto).
    Results, caller lists and counters are those of one query at a time (tests); what changes is what one engine launch
    sequence holds.  pointwise: enough queries for >= 1 600 passages, at most 16 - the runtime cuts them into launch sequences of
    up to 256 prompts that pipeline over the engine's two slots AND launches while the later queries are still being tokenised
    (T5Runtime.score_stream), so per call only the first launch sequence's tokenisation and the last one's decoder chain are
    exposed (tools/per_call_sweep.py at hits=100, one box, round 4: 6 / 8 / 12 / 16 queries per call = 7 430 / 7 432 / 7 649 / 7 749
    passages/s; before the streaming launch 6 / 8 = 7 098 / 6 992 and the default was 6); setwise: thirty-two heapsorts in lockstep
    (tools/bench_setwise_query.py, configs[2] shape at ~0.9k-token prompts, rounds 2-4: 1 / 4 / 6 / 8 queries = 106 / 47.7 / 41.0 /
    37.4 ms per query, `likelihood`; at the stated size, ~1.56k-token prompts, round 5: 1 / 8 / 16 queries = 127 / 55.7 / 48.1 ms
    `likelihood`, 129 / 59.6 / 52.0 ms `generation`; round 6, profiles/r06_lockstep_sweep.txt: 16 / 24 / 32 / 48 queries = 46.8 /
    45.4 / 44.4 / 41.7 ms `likelihood`, 50.7 / 47.7 / 46.0 / 44.6 ms `generation` - the more prompts per sift-down step, the
    better the encoder GEMMs' rounds over the CUs are filled.  Thirty-two = two alternating groups of sixteen prompts, the most
    that still fits ONE engine call of either runtime: 25k tokens of T5Runtime's 49k, the 16 sequences of LlamaRuntime); anything
    else one query at a time."""
    if kind == "pointwise":
        return max(1, min(16, -(-1600 // max(1, int(hits)))))
    if kind == "setwise":
        return 32
    return 1
