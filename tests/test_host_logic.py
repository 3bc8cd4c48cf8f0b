"""Host logic of the drop-in rankers vs goldens recorded from the REAL reference (tools/make_goldens.py).

CPU only: the model is the oracle-backed stub (tests/_stub.py) so what is under test here is prompting,
tokenise-then-batch, counters, score extraction, sort drivers, fallbacks and result assembly
(ref: llmrankers/pointwise.py:36-133, llmrankers/setwise.py:79-316).  The same cases run against the HIP
engine in tests/test_gpu_rerank.py."""
import contextlib
import io
import json
import os
import random

import numpy as np
import pytest

from conftest import GOLD, load_state
from _stub import OracleRuntime
from llmrankers.rankers import SearchResult, LlmRanker
from llmrankers.pointwise import PointwiseLlmRanker, MonoT5LlmRanker
from llmrankers.setwise import SetwiseLlmRanker


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(GOLD, "rerank_cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def runtimes(ckpt_dirs):
    from transformers import T5Tokenizer
    out = {}
    for name, path in ckpt_dirs.items():
        dims, state = load_state(path)
        if hasattr(dims, "n_enc"):                      # the T5-family fixtures (the Llama one has its own test below)
            out[name] = (OracleRuntime(dims, state), T5Tokenizer.from_pretrained(path))
    return out


def build_ranker(case, rt, tok):
    if case["kind"] == "pointwise":
        return PointwiseLlmRanker.from_runtime(rt, tok, method=case["method"], batch_size=case["batch_size"])
    return SetwiseLlmRanker.from_runtime(rt, tok, num_child=case["num_child"], k=case["k"], scoring=case["scoring"],
                            method=case["method"], num_permutation=case["num_permutation"])


def check_case(case, ranker, score_tol):
    ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
    random.seed(929)
    sink = io.StringIO()
    if case.get("raises"):
        with pytest.raises(IndexError), contextlib.redirect_stdout(sink):
            ranker.rerank(case["query"], ranking)
        return
    with contextlib.redirect_stdout(sink):
        res = ranker.rerank(case["query"], ranking)
    want = case["result"]
    assert [r.docid for r in res] == [d for d, _ in want]
    if case["kind"] == "pointwise":
        np.testing.assert_allclose([r.score for r in res], [s for _, s in want], atol=score_tol, rtol=score_tol)
        assert all(a is b for a, b in zip(sorted(res, key=id), sorted(ranking, key=id)))   # same objects, mutated in place
    else:
        assert [r.score for r in res] == [s for _, s in want]
        assert all(r.text is None for r in res)
        assert [r.docid for r in ranking] == case["caller_list_after"]
    assert [ranker.total_compare, ranker.total_prompt_tokens, ranker.total_completion_tokens] == case["counters"]


def test_all_reference_cases_cpu(cases, runtimes):
    assert len(cases["cases"]) >= 30
    for case in cases["cases"]:
        rt, tok = runtimes[case["ckpt"]]
        check_case(case, build_ranker(case, rt, tok), score_tol=2e-5)


@pytest.mark.parametrize("scoring", ["likelihood", "generation"])
def test_level_batched_build_heap_equals_one_by_one(runtimes, scoring):
    """The build phase submits the independent sift-downs of a heap level in one engine call: same ranking, caller
    list and counters as the reference's one-compare-at-a-time order (three internal levels here)."""
    rt, tok = runtimes["ckpt_gated_untied"]
    rs = random.Random(7)
    words = ["alpha", "beta", "gamma", "delta", "eps", "zeta", "eta", "theta"]
    docs = [(f"d{i}", float(40 - i), " ".join(rs.choice(words) for _ in range(rs.randint(3, 9)))) for i in range(40)]
    outs = []
    for batched in (False, True):
        rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=3, k=5, scoring=scoring, method="heapsort")
        rk.batch_independent_compares = batched
        calls = []
        many = rk._compare_many
        rk._compare_many = lambda q, dl, _m=many, _c=calls: (_c.append(len(dl)), _m(q, dl))[1]
        ranking = [SearchResult(docid=d, score=sc, text=t) for d, sc, t in docs]
        with contextlib.redirect_stdout(io.StringIO()):
            res = rk.rerank("which greek letter", ranking)
        outs.append(([(r.docid, r.score) for r in res], [r.docid for r in ranking],
                     [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens]))
        assert (len(calls) > 0 and max(calls) > 1) == batched
    assert outs[0] == outs[1]


def test_monot5_reference_cases_cpu(ckpt_dirs):
    """MonoT5LlmRanker vs the reference's own class (ref: pointwise.py:136-186): 'Query: .. Document: .. Relevant:' prompt,
    decoder_start_token_id as decoder input, softmax over the fixed ids (false 6136, true 1176) -> P(true), counters."""
    from transformers import T5Tokenizer
    with open(os.path.join(GOLD, "monot5_cases.json")) as f:
        mcases = json.load(f)["cases"]
    assert len(mcases) >= 6
    dims, state = load_state(ckpt_dirs["ckpt_monot5"])
    assert dims.tied_head and not dims.gated and dims.vocab > 6136
    rt, tok = OracleRuntime(dims, state), T5Tokenizer.from_pretrained(ckpt_dirs["ckpt_monot5"])
    for case in mcases:
        rk = MonoT5LlmRanker.from_runtime(rt, tok, method="yes_no", batch_size=case["batch_size"])
        ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
        res = rk.rerank(case["query"], ranking)
        assert [r.docid for r in res] == [d for d, _ in case["result"]]
        np.testing.assert_allclose([r.score for r in res], [s for _, s in case["result"]], atol=2e-5, rtol=2e-5)
        assert [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens] == case["counters"]


def test_pairwise_reference_cases_cpu(ckpt_dirs):
    """PairwiseLlmRanker (PRP) vs the reference's own class (ref: pairwise.py:29-295): allpair scoring incl. conflicts
    and dict-order ties, the binary heapsort and the shortcut bubblesort, every compare's two decoded generations,
    padded-shape prompt counters and batch-level completion counters."""
    from transformers import T5Tokenizer
    from llmrankers.pairwise import PairwiseLlmRanker
    with open(os.path.join(GOLD, "pairwise_cases.json")) as f:
        pcases = json.load(f)["cases"]
    assert len(pcases) >= 14
    rts = {}
    for case in pcases:
        if case["ckpt"] not in rts:
            dims, state = load_state(ckpt_dirs[case["ckpt"]])
            rts[case["ckpt"]] = (OracleRuntime(dims, state), T5Tokenizer.from_pretrained(ckpt_dirs[case["ckpt"]]))
        rt, tok = rts[case["ckpt"]]
        rk = PairwiseLlmRanker.from_runtime(rt, tok, method=case["method"], batch_size=case["batch_size"], k=case["k"])
        log, orig = [], rk.compare
        rk.compare = lambda q, d, _o=orig, _l=log: (_l.append([list(d)]), _l[-1].append(list(_o(q, d))))[1] or _l[-1][1]
        ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
        res = rk.rerank(case["query"], ranking)
        tag = (case["ckpt"], case["method"], case["batch_size"], case["k"])
        assert log == case["compares"], tag
        assert [[r.docid, r.score] for r in res] == case["result"], tag
        assert [r.docid for r in ranking] == case["caller_list_after"], tag
        assert [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens] == case["counters"], tag
        assert all(r.text is None for r in res)
    with pytest.raises(NotImplementedError):
        PairwiseLlmRanker.from_runtime(rt, tok, method="quicksort").rerank("q", [SearchResult("a", 1.0, "x")])


def test_llama_setwise_reference_cases_cpu(ckpt_dirs):
    """SetwiseLlmRanker on a Llama-family checkpoint vs the reference's own class (ref: setwise.py:60-69, 159-177): chat
    template + " Passage:" prompt ids, one greedy token decoded / stripped / upper-cased, prompt + 1 completion tokens,
    num_permutation only counted, likelihood -> NotImplementedError, the bubblesort IndexError of an out-of-window label."""
    from transformers import AutoTokenizer
    from _stub import OracleLlamaRuntime
    with open(os.path.join(GOLD, "llama_cases.json")) as f:
        gold = json.load(f)
    dims, state = load_state(ckpt_dirs["ckpt_llama"])
    tok = AutoTokenizer.from_pretrained(ckpt_dirs["ckpt_llama"])
    rt = OracleLlamaRuntime(dims, state)
    probe = SetwiseLlmRanker.from_runtime(rt, tok)
    assert probe._llama_prompt_ids("hello world") == gold["prompt_probe"]["ids"]
    n = 0
    for case in gold["cases"]:
        rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=case["num_child"], k=case["k"], scoring=case["scoring"],
                                           method=case["method"], num_permutation=case["num_permutation"])
        rk.batch_independent_compares = False
        log, orig = [], rk.compare
        rk.compare = lambda q, d, _o=orig, _l=log: (_l.append([[x.docid for x in d]]), _l[-1].append(_o(q, d)))[1] or _l[-1][1]
        ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
        random.seed(929)
        sink = io.StringIO()
        if case["raises"]:
            with pytest.raises({"IndexError": IndexError, "NotImplementedError": NotImplementedError}[case["raises"]]), contextlib.redirect_stdout(sink):
                rk.rerank(case["query"], ranking)
            assert [c for c in log if len(c) == 2] == case["compares"]
        else:
            with contextlib.redirect_stdout(sink):
                res = rk.rerank(case["query"], ranking)
            assert log == case["compares"]
            assert [[r.docid, r.score] for r in res] == case["result"]
            assert [r.docid for r in ranking] == case["caller_list_after"]
        assert [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens] == case["counters"], case["scoring"]
        n += 1
    assert n >= 10
    # the level-batched build phase gives the same result for a Llama model too
    case = next(c for c in gold["cases"] if c["method"] == "heapsort" and c["num_child"] == 3 and not c["raises"])
    rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=3, k=case["k"], scoring="generation", method="heapsort")
    assert rk._batched_ok()
    ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
    with contextlib.redirect_stdout(io.StringIO()):
        res = rk.rerank(case["query"], ranking)
    assert [[r.docid, r.score] for r in res] == case["result"]
    assert [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens] == case["counters"]


def test_llama_pairwise_reference_cases_cpu(ckpt_dirs):
    """PairwiseLlmRanker on a Llama-family checkpoint vs the reference's own class (ref: pairwise.py:60-77, 104-129): every
    compare's two outputs ("Passage <token>", decoded / stripped / upper-cased), rankings and counters of heapsort and
    bubblesort queries; `allpair` is T5-only in the reference (AttributeError) and here."""
    from transformers import AutoTokenizer
    from _stub import OracleLlamaRuntime
    from llmrankers.pairwise import PairwiseLlmRanker
    with open(os.path.join(GOLD, "llama_pairwise_cases.json")) as f:
        gold = json.load(f)
    dims, state = load_state(ckpt_dirs["ckpt_llama"])
    tok = AutoTokenizer.from_pretrained(ckpt_dirs["ckpt_llama"])
    rt = OracleLlamaRuntime(dims, state)
    n = 0
    for case in gold["cases"]:
        rk = PairwiseLlmRanker.from_runtime(rt, tok, method=case["method"], batch_size=2, k=case["k"])
        log, orig = [], rk.compare
        rk.compare = lambda q, d, _o=orig, _l=log: (_l.append([list(d)]), _l[-1].append(_o(q, d)))[1] or _l[-1][1]
        ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
        if case["raises"]:
            with pytest.raises(AttributeError):
                rk.rerank(case["query"], ranking)
            continue
        res = rk.rerank(case["query"], ranking)
        assert log == case["compares"], case["method"]
        assert [[r.docid, r.score] for r in res] == case["result"]
        assert [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens] == case["counters"]
        n += 1
    assert n >= 4


def test_fp16_score_mode_reproduces_the_reference_cuda_quantisation(runtimes):
    """PointwiseLlmRanker.fp16_scores: logits and probabilities rounded the way the reference's fp16 'cuda' path rounds
    them (torch: half logits -> softmax in fp32 -> half) - saturated scores tie exactly and keep their input order."""
    import torch
    rt, tok = runtimes["ckpt_gated_untied"]

    class Fixed:
        model_type, config, decoder_start_token_id = "t5", None, 0

        def score(self, seqs, dec, ids):
            return np.array([[9.31, -1.2], [9.6, -1.3], [0.2, 0.1], [12.0, -3.0]], np.float32)[:len(seqs)]

    docs = [SearchResult(docid=f"d{i}", score=0.0, text=f"text {i}") for i in range(4)]
    rk = PointwiseLlmRanker.from_runtime(Fixed(), tok, method="yes_no", batch_size=4)
    rk.fp16_scores = True
    res = rk.rerank("q", list(docs))
    lg = torch.tensor([[9.31, -1.2], [9.6, -1.3], [0.2, 0.1], [12.0, -3.0]]).half()
    want = torch.softmax(lg, dim=1)[:, 0].float().tolist()            # what the reference computes on an fp16 model
    got = {r.docid: r.score for r in res}
    assert [got[f"d{i}"] for i in range(4)] == want
    assert got["d0"] == got["d1"] == got["d3"] == 1.0                  # saturated: exact ties ...
    assert [r.docid for r in res] == ["d0", "d1", "d3", "d2"]           # ... resolved by input order (stable sort)
    rk.fp16_scores = False
    res = rk.rerank("q", [SearchResult(docid=f"d{i}", score=0.0, text=f"text {i}") for i in range(4)])
    assert [r.docid for r in res] == ["d3", "d1", "d0", "d2"]           # fp32 scores separate them


def test_truncate(cases, runtimes):
    rt, tok = runtimes["ckpt_gated_untied"]
    pw = PointwiseLlmRanker.from_runtime(rt, tok, method="yes_no", batch_size=2)
    sw = SetwiseLlmRanker.from_runtime(rt, tok)
    for text, n, want in cases["truncate"]:
        assert pw.truncate(text, n) == want
        assert sw.truncate(text, n) == want


def test_sort_drivers_match_reference_traces():
    """Reference heapsort/bubblesort (incl. malformed-output fallbacks) replayed with the recorded comparator."""
    with open(os.path.join(GOLD, "sort_traces.json")) as f:
        traces = json.load(f)
    assert len(traces) == 28
    for tr in traces:
        rk = SetwiseLlmRanker.__new__(SetwiseLlmRanker)
        rk.num_child, rk.k, rk.method, rk.num_permutation = tr["c"], tr["k"], tr["method"], 1
        rel, calls = tr["rel"], []

        def fake(query, docs, _rel=rel, _calls=calls, _mode=tr["mode"], _rk=rk):
            _rk.total_compare += 1
            idx = [int(d.docid[1:]) for d in docs]
            _calls.append(idx)
            if not idx:
                return "A"
            best = max(range(len(docs)), key=lambda j: _rel[idx[j]])
            if _mode == "garbage" and len(_calls) % 3 == 0:
                return ["?", "Z", "zz"][len(_calls) % 9 // 3]
            if _mode == "garbage" and len(_calls) % 7 == 0 and _rk.method == "heapsort":
                return "W"
            return SetwiseLlmRanker.CHARACTERS[best]

        rk.compare = fake
        n = tr["n"]
        ranking = [SearchResult(docid=f"d{i}", score=float(n - i), text=f"t{i}") for i in range(n)]
        res = rk.rerank("q", ranking)
        assert calls == tr["calls"], (tr["n"], tr["c"], tr["k"], tr["method"], tr["mode"])
        assert [[r.docid, r.score] for r in res] == tr["result"]
        assert [r.docid for r in ranking] == tr["caller_list_after"]
        assert rk.total_compare == tr["total_compare"]


def test_api_surface():
    import dataclasses
    import inspect
    assert [f.name for f in dataclasses.fields(SearchResult)] == ["docid", "score", "text"]
    assert list(inspect.signature(PointwiseLlmRanker.__init__).parameters)[:7] == [
        "self", "model_name_or_path", "tokenizer_name_or_path", "device", "method", "batch_size", "cache_dir"]
    sig = inspect.signature(SetwiseLlmRanker.__init__)
    assert list(sig.parameters)[:10] == ["self", "model_name_or_path", "tokenizer_name_or_path", "device", "num_child",
                                         "k", "scoring", "method", "num_permutation", "cache_dir"]
    assert (sig.parameters["num_child"].default, sig.parameters["k"].default, sig.parameters["scoring"].default,
            sig.parameters["method"].default, sig.parameters["num_permutation"].default) == (3, 10, "generation", "heapsort", 1)
    assert len(SetwiseLlmRanker.CHARACTERS) == 23
    assert issubclass(MonoT5LlmRanker, PointwiseLlmRanker) and issubclass(SetwiseLlmRanker, LlmRanker)
    with pytest.raises(NotImplementedError):
        LlmRanker().rerank("q", [])


def test_unknown_sort_method_and_cpu_device(runtimes, ckpt_dirs):
    rt, tok = runtimes["ckpt_gated_untied"]
    sw = SetwiseLlmRanker.from_runtime(rt, tok, method="quicksort")
    with pytest.raises(NotImplementedError):
        sw.rerank("q", [SearchResult("a", 1.0, "x"), SearchResult("b", 0.5, "y")])
    # no silent CPU fallback in the product path: device='cpu' must fail loudly
    with pytest.raises((RuntimeError, FileNotFoundError)):
        PointwiseLlmRanker(ckpt_dirs["ckpt_gated_untied"], None, "cpu", method="yes_no", batch_size=2)


def test_generation_passes_label_candidates_to_runtimes_that_take_them(runtimes):
    """SetwiseLlmRanker / PairwiseLlmRanker hand the label token ids to a runtime that advertises
    `supports_greedy_candidates` (the engine then runs both greedy steps in one decoder pass); a runtime that does not
    is called exactly as before.  The hint never changes what is returned."""
    from llmrankers.pairwise import PairwiseLlmRanker
    rt, tok = runtimes["ckpt_labelboost"]
    seen = []

    class Hinted:
        supports_greedy_candidates = True
        config = rt.config

        def greedy(self, seqs, dec_prefix, max_new, eos_id=1, pad_id=0, candidates=None):
            seen.append(list(candidates) if candidates is not None else None)
            return rt.greedy(seqs, dec_prefix, max_new, eos_id, pad_id)

        def score(self, *a, **k):
            return rt.score(*a, **k)

    docs = [SearchResult(docid=f"d{i}", score=0.0, text=f"text {i} about things") for i in range(4)]
    plain = SetwiseLlmRanker.from_runtime(rt, tok, num_child=3, k=2, scoring="generation")
    hinted = SetwiseLlmRanker.from_runtime(Hinted(), tok, num_child=3, k=2, scoring="generation")
    with contextlib.redirect_stdout(io.StringIO()):
        assert hinted.compare("a query", docs) == plain.compare("a query", docs)
    assert seen == [hinted.target_token_ids[:4]]
    pw = PairwiseLlmRanker.from_runtime(Hinted(), tok, method="bubblesort", k=2)
    pw0 = PairwiseLlmRanker.from_runtime(rt, tok, method="bubblesort", k=2)
    with contextlib.redirect_stdout(io.StringIO()):
        assert pw.compare("a query", ["one text", "another text"]) == pw0.compare("a query", ["one text", "another text"])
    assert seen[-1] == pw._label_ids and len(seen[-1]) == 2


@pytest.mark.parametrize("scoring", ["generation", "likelihood"])
def test_setwise_rerank_many_equals_one_query_at_a_time(runtimes, scoring):
    """SetwiseLlmRanker.rerank_many advances the heapsorts of several queries in lockstep, their pending compares in one
    engine call per step: per query the result, the caller's re-ordered list and the counters are those of rerank();
    engine calls carry compares of several queries; other settings fall back to one query at a time."""
    rt, tok = runtimes["ckpt_labelboost"]
    calls = []

    class Spy:
        config = rt.config

        def score(self, seqs, *a, **k):
            calls.append(len(seqs))
            return rt.score(seqs, *a, **k)

        def greedy(self, seqs, *a, **k):
            calls.append(len(seqs))
            return rt.greedy(seqs, *a, **k)

    rs = random.Random(17)
    words = "alpha beta gamma delta river water neural model search index music film bank money".split()

    def make(n, seed):
        r = random.Random(seed)
        return [SearchResult(docid=f"q{seed}d{i}", score=float(n - i), text=" ".join(r.choice(words) for _ in range(6))) for i in range(n)]

    sizes = [14, 1, 9, 23, 0, 5]
    queries = [" ".join(rs.choice(words) for _ in range(3)) for _ in sizes]
    for num_child, k in ((3, 4), (2, 10)):
        one = SetwiseLlmRanker.from_runtime(rt, tok, num_child=num_child, k=k, scoring=scoring)
        want, want_lists, want_counters = [], [], []
        with contextlib.redirect_stdout(io.StringIO()):
            for q, (query, n) in enumerate(zip(queries, sizes)):
                ranking = make(n, q)
                res = one.rerank(query, ranking)
                want.append([(d.docid, d.score, d.text) for d in res])
                want_lists.append([d.docid for d in ranking])
                want_counters.append((one.total_compare, one.total_prompt_tokens, one.total_completion_tokens))
        del calls[:]
        many = SetwiseLlmRanker.from_runtime(Spy(), tok, num_child=num_child, k=k, scoring=scoring)
        rankings = [make(n, q) for q, n in enumerate(sizes)]
        with contextlib.redirect_stdout(io.StringIO()):
            got, counters = many.rerank_many(list(zip(queries, rankings)))
        assert [[(d.docid, d.score, d.text) for d in res] for res in got] == want
        assert [[d.docid for d in r] for r in rankings] == want_lists
        assert counters == want_counters
        assert (many.total_compare, many.total_prompt_tokens, many.total_completion_tokens) == want_counters[-1]
        assert sum(calls) == sum(c[0] for c in want_counters) and len(calls) < sum(calls) / 2   # same compares, far fewer engine calls
    # bubblesort: the same lockstep driver (one window per query and step)
    for num_child, k in ((2, 2), (3, 3)):
        ref = SetwiseLlmRanker.from_runtime(rt, tok, num_child=num_child, k=k, scoring=scoring, method="bubblesort")
        bub = SetwiseLlmRanker.from_runtime(Spy(), tok, num_child=num_child, k=k, scoring=scoring, method="bubblesort")
        bsizes = [9, 6, 12, 5]
        want, want_lists, want_counters = [], [], []
        with contextlib.redirect_stdout(io.StringIO()):
            for q, n in enumerate(bsizes):
                ranking = make(n, q)
                want.append([(d.docid, d.score) for d in ref.rerank(queries[q], ranking)])
                want_lists.append([d.docid for d in ranking])
                want_counters.append((ref.total_compare, ref.total_prompt_tokens, ref.total_completion_tokens))
            del calls[:]
            rankings = [make(n, q) for q, n in enumerate(bsizes)]
            got, counters = bub.rerank_many(list(zip(queries, rankings)))
        assert [[(d.docid, d.score) for d in r] for r in got] == want and counters == want_counters
        assert [[d.docid for d in r] for r in rankings] == want_lists
        assert max(calls) > 1                                                       # windows of several queries shared a call
    # permutation voting draws random numbers: the one-by-one path
    perm = SetwiseLlmRanker.from_runtime(rt, tok, num_child=2, k=2, scoring=scoring, num_permutation=2)
    ref = SetwiseLlmRanker.from_runtime(rt, tok, num_child=2, k=2, scoring=scoring, num_permutation=2)
    with contextlib.redirect_stdout(io.StringIO()):
        random.seed(5)
        got, _ = perm.rerank_many([(queries[0], make(6, 0)), (queries[2], make(5, 2))])
        random.seed(5)
        assert [[d.docid for d in r] for r in got] == [[d.docid for d in ref.rerank(queries[0], make(6, 0))], [d.docid for d in ref.rerank(queries[2], make(5, 2))]]


def test_default_queries_per_call():
    """run.py --queries_per_call 0 (auto): pointwise enough queries for 1 600 passages per call, capped at 16; setwise thirty-two
    heapsorts in lockstep (two alternating groups of sixteen prompts); everything else one query at a time (llmrankers/_batching.py)."""
    from llmrankers._batching import default_queries_per_call as d
    assert [d("pointwise", h) for h in (2000, 1000, 512, 100, 64, 20, 4, 0)] == [1, 2, 4, 16, 16, 16, 16, 16]
    assert d("setwise", 100) == 32 and d("pairwise", 100) == 1 and d("other", 100) == 1


class _EventEngine:
    """DryEngine (tools/dry_engine.py) that records the order of engine calls - to see WHEN the runtime launches."""

    def __new__(cls, events, **kw):
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
        from dry_engine import DryEngine
        from llmrankers import _synth

        class Rec(DryEngine):
            def stage(self, seqs, slot=0):
                events.append(("stage", slot, len(seqs)))
                return super().stage(seqs, slot)

            def read_scores(self, slot=0):
                events.append(("read", slot))
                return super().read_scores(slot)
        return Rec(_synth.TOY_GATED_UNTIED, 0, **kw)


def test_score_stream_launches_while_the_caller_still_produces_sequences():
    """T5Runtime.score_stream: same scores and the same cuts as the flat list, launches as soon as a call is full (the groups
    behind it are pulled - tokenised - after the launch), and no slot stays in flight when the producer raises."""
    from llmrankers._runtime import T5Runtime
    from llmrankers import _synth
    events = []
    eng = _EventEngine(events, max_tokens=400, max_seqs=5, max_dec_len=4)
    rt = T5Runtime.from_engine(eng, _synth.TOY_GATED_UNTIED)
    rs = random.Random(4)
    groups = [[[rs.randrange(3, 90) for _ in range(rs.randrange(1, 120))] for _ in range(n)] for n in (4, 3, 0, 6, 1, 2)]
    flat = [s for g in groups for s in g]

    def produce():
        for i, g in enumerate(groups):
            events.append(("pull", i))
            yield g
    got = rt.score_stream(produce(), [0], [7, 9])
    want = rt.score(flat, [0], [7, 9])
    assert np.array_equal(got, want) and got.shape == (len(flat), 2)
    cuts = [len(c) for c in rt._chunks(flat)]
    assert [e[2] for e in events if e[0] == "stage"] == cuts and len(cuts) >= 4
    order = [e[0] for e in events]
    first_stage = order.index("stage")
    assert ("pull", len(groups) - 1) in events[first_stage:], "the last group was produced before the first launch"
    # two slots: a slot is read before it is staged again, and every staged slot is read exactly once
    staged = [e[1] for e in events if e[0] == "stage"]
    assert staged == [i % eng.num_slots for i in range(len(staged))]
    assert sum(e[0] == "read" for e in events) == len(staged)
    # score_batches = the same stream, cut back into the caller's batches (empty ones included)
    parts = rt.score_batches(groups, [0], [7, 9])
    assert [len(p) for p in parts] == [len(g) for g in groups] and np.array_equal(np.concatenate(parts), want)
    assert rt.score_stream(iter([]), [0], [7, 9]).shape == (0, 2)

    # a producer that fails mid-way: the exception arrives, the launches already made are drained
    del events[:]

    def bad():
        yield groups[0]
        yield groups[1]
        raise RuntimeError("tokeniser fell over")
    with pytest.raises(RuntimeError, match="fell over"):
        rt.score_stream(bad(), [0], [7, 9])
    assert sum(e[0] == "stage" for e in events) == sum(e[0] == "read" for e in events)
    with pytest.raises(ValueError, match="exceeds the engine capacity"):
        rt.score_stream([[list(range(3, 90)) * 6]], [0], [7, 9])


def test_pointwise_rerank_many_streams_and_equals_one_query_at_a_time():
    """PointwiseLlmRanker.rerank_many over the streaming runtime: rankings, scores and counters of rerank() per query; the
    first engine launch happens before the last query has been tokenised."""
    from transformers import T5Tokenizer
    from llmrankers._runtime import T5Runtime
    from llmrankers import _synth
    events = []
    eng = _EventEngine(events, max_tokens=4000, max_seqs=24, max_dec_len=4)
    rt = T5Runtime.from_engine(eng, _synth.TOY_GATED_UNTIED)
    tok = T5Tokenizer.from_pretrained(os.path.join(GOLD, "tok"))

    rk = PointwiseLlmRanker.from_runtime(rt, tok, method="yes_no", batch_size=4)
    counted = rk._counted_batches

    def noted(prompts, dec_len):                           # when a query's prompts are tokenised and cut into batches
        events.append(("tokenise",))
        return counted(prompts, dec_len)
    rk._counted_batches = noted
    rs = random.Random(1)
    words = "alpha beta gamma delta search engine ranking neural index passage".split()

    def items():
        r = random.Random(2)
        return [(" ".join(r.choice(words) for _ in range(5)),
                 [SearchResult(docid=f"q{q}d{i}", score=float(20 - i), text=" ".join(r.choice(words) for _ in range(r.randrange(3, 40))))
                  for i in range(n)]) for q, n in enumerate((10, 7, 10, 9, 10))]
    many, counters = rk.rerank_many(items())
    order = [e[0] for e in events]
    assert "stage" in order and order.index("stage") < len(order) - 1 - order[::-1].index("tokenise"), \
        "every query was tokenised before the first launch"
    one = []
    for q, ranking in items():
        res = rk.rerank(q, ranking)
        one.append(([(d.docid, d.score) for d in res], (rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens)))
    assert [[(d.docid, d.score) for d in res] for res in many] == [o[0] for o in one]
    assert counters == [o[1] for o in one]


@pytest.mark.parametrize("method,sizes", [("heapsort", (14, 9, 14, 12, 13, 5, 14)), ("bubblesort", (14, 9, 14, 12, 13, 5, 14)),
                                          ("heapsort", (14, 9, 14, 12, 13, 5, 14, 11, 14, 7, 10, 14, 13, 6, 12, 14))])   # 16 = run.py's default
def test_setwise_lockstep_alternates_two_groups_over_the_slots(method, sizes):
    """SetwiseLlmRanker.rerank_many, likelihood scoring on a runtime with batch slots: the chains run as two groups that
    alternate over slots 0 and 1 (a launch on one slot happens while the other is still uncollected), rankings and counters
    are those of rerank() one query at a time; a call that does not fit the engine sends the rest down the blocking loop."""
    from transformers import T5Tokenizer
    from llmrankers._runtime import T5Runtime
    from llmrankers import _synth
    tok = T5Tokenizer.from_pretrained(os.path.join(GOLD, "tok"))
    words = "alpha beta gamma delta search engine ranking neural index passage".split()

    def items():
        r = random.Random(7)
        return [(" ".join(r.choice(words) for _ in range(4)),
                 [SearchResult(docid=f"q{q}d{i}", score=float(50 - i), text=" ".join(r.choice(words) for _ in range(r.randrange(3, 12))))
                  for i in range(n)]) for q, n in enumerate(sizes)]

    def run(max_tokens, alternate=True):
        events = []
        eng = _EventEngine(events, max_tokens=max_tokens, max_seqs=64, max_dec_len=4)
        rt = T5Runtime.from_engine(eng, _synth.TOY_GATED_UNTIED)
        rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=3, k=4, scoring="likelihood", method=method)
        rk.alternate_groups = alternate
        with contextlib.redirect_stdout(io.StringIO()):
            many, counters = rk.rerank_many(items())
            one = []
            for q, ranking in items():
                res = rk.rerank(q, ranking)
                one.append(([(d.docid, d.score) for d in res], (rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens)))
        assert [[(d.docid, d.score) for d in res] for res in many] == [o[0] for o in one]
        assert counters == [o[1] for o in one]
        return events, eng

    events, eng = run(100000)
    stages = [e for e in events if e[0] == "stage"]
    assert any(e[1] == 1 for e in stages), "slot 1 was never used: the groups did not alternate"
    # somewhere a slot is staged while the other slot's scores have not been read yet
    outstanding, overlapped = set(), False
    for e in events:
        if e[0] == "stage":
            overlapped |= bool(outstanding - {e[1]})
            outstanding.add(e[1])
        elif e[0] == "read":
            outstanding.discard(e[1])
    assert overlapped
    ev_single, _ = run(100000, alternate=False)
    assert all(e[1] == 0 for e in ev_single if e[0] == "stage")
    # an engine too small for a group's call: same results through the blocking loop (asserted inside run)
    run(260)
