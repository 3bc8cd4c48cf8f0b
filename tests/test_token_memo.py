"""The exact word-level token memo of llmrankers._batching: ids identical to the tokenizer's own for
every prompt shape the rankers build, untouched tokenizers it does not model, and the self-check that switches it off."""
import json
import os
import random
import warnings

import pytest

from conftest import GOLD
from llmrankers import _batching
from llmrankers._batching import WordSpliceTokenizer, tokenize_prompts


@pytest.fixture(scope="module")
def tok():
    from transformers import T5Tokenizer
    return T5Tokenizer.from_pretrained(os.path.join(GOLD, "tok"))


def full(tok, prompts):
    return [list(x) for x in tok(list(prompts))["input_ids"]]


def test_memo_is_exact_on_ranker_prompts_and_odd_text(tok):
    with open(os.path.join(GOLD, "rerank_cases.json")) as f:
        cases = json.load(f)["cases"]
    texts = [t for c in cases for _, _, t in c["input"]][:200]
    rs = random.Random(5)
    prompts = []
    for i in range(40):                                                    # the three prompt templates (ref: pointwise.py:41-47, setwise.py:106-111)
        docs = rs.sample(texts, 4)
        q = rs.choice(texts)[:40]
        prompts.append(f"Passage: {docs[0]}\nQuery: {q}\nDoes the passage answer the query? Answer 'Yes' or 'No'")
        body = "\n\n".join(f'Passage {chr(65 + j)}: "{d}"' for j, d in enumerate(docs))
        prompts.append(f'Given a query "{q}", which of the following passages is the most relevant one to the query?\n\n{body}'
                       f"\n\nOutput only the passage label of the most relevant passage:")
        prompts.append(f"Passage: {docs[1]}\nPlease write a question based on this passage.")
    prompts += ["", " ", "\n", "a", "  leading and   double  spaces ", "tab\tinside word", "new\nline\n\nblocks\n",
                "unicode: café naïve 中文 \U0001F600 nb sp zero​width", "punct!? (x) [y] {z} a-b_c 3.14 1,000",
                "</s> inside <pad> text <extra_id_0> end", "\"quoted\" 'single' ``ticks''", "x" * 300]
    memo = WordSpliceTokenizer(tok, verify_first=0, verify_every=10 ** 9)  # no self-check: the memo alone must be right
    assert memo.enabled
    assert memo(prompts) == full(tok, prompts)
    assert memo(prompts[::-1]) == full(tok, prompts[::-1])                  # warm: every word from the dictionary
    assert memo.enabled and len(memo.words) > 100


def test_module_entry_point_uses_one_memo_per_tokenizer_and_honours_the_switch(tok, monkeypatch):
    prompts = ["Passage: alpha beta\nQuery: gamma", "Passage A: \"alpha\"\n\nPassage B: \"beta gamma\""]
    assert tokenize_prompts(tok, prompts) == full(tok, prompts)
    assert tokenize_prompts(tok, []) == []
    memo = _batching._MEMOS[tok]
    assert memo.enabled and "alpha" in memo.words
    assert tokenize_prompts(tok, prompts[:1]) == full(tok, prompts[:1]) and _batching._MEMOS[tok] is memo
    monkeypatch.setenv("RK_TOKEN_CACHE", "0")
    assert not WordSpliceTokenizer(tok).enabled


def test_other_tokenizer_families_go_to_the_tokenizer(tok):
    from transformers import AutoTokenizer
    llama = AutoTokenizer.from_pretrained(os.path.join(GOLD, "tok_llama"))
    memo = WordSpliceTokenizer(llama)
    assert not memo.enabled                                                 # byte-level BPE: not word-separable as modelled
    p = ["Passage A: \"x y\"\n\nPassage B: \"z\""]
    assert memo(p) == full(llama, p)

    class Plain:                                                            # no tokenizers backend at all (test stubs)
        def __call__(self, texts, **kw):
            return {"input_ids": [[len(t)] for t in texts]}
    assert tokenize_prompts(Plain(), ["ab", "c"]) == [[2], [1]]


def test_self_check_switches_the_memo_off_on_the_first_difference(tok):
    class ContextDependent:
        """Claims the T5 backend but glues a marker to the ids of any text that contains 'magic' somewhere else."""
        backend_tokenizer = tok.backend_tokenizer

        def __call__(self, texts, **kw):
            single = isinstance(texts, str)
            out = tok([texts] if single else list(texts), **kw)["input_ids"]
            out = [list(ids) + ([7] if ("magic" in t and " " in t.strip()) else []) for ids, t in zip(out, [texts] if single else texts)]
            return {"input_ids": out[0] if single else out}

    fake = ContextDependent()
    memo = WordSpliceTokenizer(fake, verify_first=2, verify_every=4)
    assert memo.enabled
    assert memo(["plain words only"]) == fake(["plain words only"])["input_ids"]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = memo(["some magic here", "more words"])
    assert got == fake(["some magic here", "more words"])["input_ids"]      # the caller still gets the tokenizer's ids
    assert not memo.enabled and w and "switched off" in str(w[0].message)
    assert memo(["some magic here"]) == fake(["some magic here"])["input_ids"]


def test_non_plain_prompts_bypass_the_memo():
    """Only printable-ASCII prompts are spliced from the word memo; a prompt with a combining mark, a no-break space or a tab
    goes to the tokenizer itself (the normaliser may act across a word boundary there) - results equal the tokenizer's for
    every prompt of a mixed batch, in order."""
    from transformers import T5Tokenizer
    from llmrankers._batching import WordSpliceTokenizer, _tokenize_full
    tok = T5Tokenizer.from_pretrained(os.path.join(GOLD, "tok"))
    memo = WordSpliceTokenizer(tok)
    assert memo.enabled
    prompts = ["neural ranking model", "cafe\u0301 river water", "search\u00a0engine index", "music\tart film", "plain again\nnext line"]
    assert [memo._plain(p) for p in prompts] == [True, False, False, False, True]
    calls = []
    orig = memo._splice
    memo._splice = lambda ps: (calls.append(list(ps)), orig(ps))[1]
    assert memo(prompts) == _tokenize_full(tok, prompts)
    assert calls and all(memo._plain(p) for c in calls for p in c)


def test_every_new_word_is_verified_in_its_first_context(tok):
    """Whole-prompt sampling switched off entirely: a word whose ids depend on having a neighbour is caught the first time it
    appears, by the (previous, word, next) window that travels with the new words - the call that would have been wrong is
    re-tokenised by the tokenizer, and the memo stays off."""
    class NeighbourDependent:
        """T5 backend; any text that holds the word 'magic' next to another word gets an extra id after it."""
        backend_tokenizer = tok.backend_tokenizer

        def __call__(self, texts, **kw):
            single = isinstance(texts, str)
            tl = [texts] if single else list(texts)
            out = []
            for ids, t in zip(tok(tl, **kw)["input_ids"], tl):
                ids = list(ids)
                ws = t.split()
                if "magic" in ws and len(ws) > 1:
                    k = len(tok(" ".join(ws[:ws.index("magic") + 1]), add_special_tokens=False)["input_ids"])
                    ids.insert(k, 7)
                out.append(ids)
            return {"input_ids": out[0] if single else out}

    fake = NeighbourDependent()
    memo = WordSpliceTokenizer(fake, verify_first=0, verify_every=10 ** 9)
    assert memo.enabled
    warm = ["plain words only", "more plain words here"]
    assert memo(warm) == fake(warm)["input_ids"] and memo.enabled
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = memo(["plain words", "some magic here", "words only"])
    assert got == fake(["plain words", "some magic here", "words only"])["input_ids"]
    assert not memo.enabled and w and "switched off" in str(w[0].message) and "1 prompt(s)" in str(w[0].message)   # (prompt 0 was a whole-prompt comparison)


def test_a_normaliser_that_can_reach_across_a_blank_makes_the_tokenizer_ineligible(tok, tmp_path):
    """Adversarial backend: the T5 pipeline plus a regex Replace that glues 'river water' into one word - the ids of 'water'
    then depend on the word in front of it, and both words may be known before they first meet.  Such a tokenizer is not
    spliced at all (the eligibility gate reads the normaliser); the T5 forms - none, NFKC / Precompiled, the blank-collapsing
    Replace - still are."""
    from tokenizers import Regex, Tokenizer, normalizers
    from transformers import PreTrainedTokenizerFast
    backend = Tokenizer.from_str(tok.backend_tokenizer.to_str())
    backend.normalizer = normalizers.Sequence([normalizers.NFKC(), normalizers.Replace(Regex("river water"), "riverwater")])
    glued = PreTrainedTokenizerFast(tokenizer_object=backend, eos_token="</s>", pad_token="<pad>", unk_token="<unk>")
    memo = WordSpliceTokenizer(glued)
    assert not memo.enabled
    prompts = ["water", "river", "the river water flows", "river  water"]
    assert memo(prompts) == full(glued, prompts)
    assert full(glued, ["the river water flows"]) != [full(glued, ["the river"])[0][:-1] + full(glued, ["water flows"])[0]]   # (it IS context dependent)
    backend.normalizer = normalizers.Sequence([normalizers.NFKC(), normalizers.Replace(Regex(" {2,}"), " ")])
    t5like = PreTrainedTokenizerFast(tokenizer_object=backend, eos_token="</s>", pad_token="<pad>", unk_token="<unk>")
    memo = WordSpliceTokenizer(t5like, verify_first=0, verify_every=10 ** 9)
    assert memo.enabled
    assert memo(prompts) == full(t5like, prompts) and memo.enabled
