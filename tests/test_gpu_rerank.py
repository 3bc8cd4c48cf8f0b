"""End-to-end drop-in parity on the GPU: every case recorded from the REAL reference (tests/golden/rerank_cases.json,
made by tools/make_goldens.py) re-run through PointwiseLlmRanker / SetwiseLlmRanker on the HIP engine."""
import contextlib
import io
import json
import os
import random

import numpy as np
import pytest

from conftest import GOLD

pytestmark = pytest.mark.gpu
SCORE_TOL = 1e-3          # north_star: pointwise scores within 1e-3
# Setwise parity is "identical docid rank order".  That is demanded for every case whose reference decisions all have
# a logit margin (top-1 vs runner-up, recorded by tools/annotate_margins.py with the fp32 oracle) above the fp16
# noise floor of this model scale (measured max logit error ~2e-3); all committed cases qualify.
MARGIN_FLOOR = 5e-3


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(GOLD, "rerank_cases.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="module")
def stack(ckpt_dirs):
    """one runtime + tokenizer per fixture checkpoint, built through the product path (checkpoint dir -> C ABI)"""
    from transformers import T5Tokenizer
    from llmrankers._runtime import T5Runtime
    out = {}
    for name, path in ckpt_dirs.items():
        with open(os.path.join(path, "config.json")) as f:
            if json.load(f).get("model_type") != "t5":
                continue                                   # the Llama checkpoint has its own tests (test_gpu_kernels.py)
        out[name] = (T5Runtime(path, "cuda", max_tokens=8192, max_seqs=64, max_dec_len=40), T5Tokenizer.from_pretrained(path))
    return out


def _build(case, rt, tok):
    from llmrankers.pointwise import PointwiseLlmRanker
    from llmrankers.setwise import SetwiseLlmRanker
    if case["kind"] == "pointwise":
        return PointwiseLlmRanker.from_runtime(rt, tok, method=case["method"], batch_size=case["batch_size"])
    return SetwiseLlmRanker.from_runtime(rt, tok, num_child=case["num_child"], k=case["k"], scoring=case["scoring"],
                            method=case["method"], num_permutation=case["num_permutation"])


def test_pointwise_cases(cases, stack):
    from llmrankers.rankers import SearchResult
    n = 0
    for case in cases:
        if case["kind"] != "pointwise":
            continue
        rt, tok = stack[case["ckpt"]]
        ranker = _build(case, rt, tok)
        ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
        res = ranker.rerank(case["query"], ranking)
        want = dict((d, s) for d, s in case["result"])
        tol = SCORE_TOL if case["method"] == "yes_no" else 5e-2      # qlm scores are sums of ~10 log-probs of size ~5
        got = np.array([r.score for r in res]); ref = np.array([want[r.docid] for r in res])
        assert np.abs(got - ref).max() < tol, (case["method"], case["ckpt"], np.abs(got - ref).max())
        # rank order identical wherever the reference's adjacent scores differ by more than the tolerance
        ref_sorted = [s for _, s in case["result"]]
        if min(a - b for a, b in zip(ref_sorted, ref_sorted[1:])) > 2 * tol:
            assert [r.docid for r in res] == [d for d, _ in case["result"]]
        assert [ranker.total_compare, ranker.total_prompt_tokens, ranker.total_completion_tokens] == case["counters"]
        n += 1
    assert n >= 16


def test_monot5_cases(stack):
    """MonoT5LlmRanker on the engine vs the reference's MonoT5LlmRanker (relu FFN, tied + scaled head, ids 6136 / 1176)."""
    from llmrankers.pointwise import MonoT5LlmRanker
    from llmrankers.rankers import SearchResult
    with open(os.path.join(GOLD, "monot5_cases.json")) as f:
        mcases = json.load(f)["cases"]
    rt, tok = stack["ckpt_monot5"]
    for case in mcases:
        rk = MonoT5LlmRanker.from_runtime(rt, tok, method="yes_no", batch_size=case["batch_size"])
        ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
        res = rk.rerank(case["query"], ranking)
        want = dict((d, s) for d, s in case["result"])
        got = np.array([r.score for r in res]); ref = np.array([want[r.docid] for r in res])
        assert np.abs(got - ref).max() < SCORE_TOL, np.abs(got - ref).max()
        ref_sorted = [s for _, s in case["result"]]
        if len(ref_sorted) < 2 or min(a - b for a, b in zip(ref_sorted, ref_sorted[1:])) > 2 * SCORE_TOL:
            assert [r.docid for r in res] == [d for d, _ in case["result"]]
        assert [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens] == case["counters"]
    assert len(mcases) >= 6


def test_pairwise_cases(stack):
    """PairwiseLlmRanker (PRP: two greedy generations per pair through rk_t5_greedy) vs the reference's PairwiseLlmRanker."""
    from llmrankers.pairwise import PairwiseLlmRanker
    from llmrankers.rankers import SearchResult
    with open(os.path.join(GOLD, "pairwise_cases.json")) as f:
        pcases = json.load(f)["cases"]
    for case in pcases:
        rt, tok = stack[case["ckpt"]]
        rk = PairwiseLlmRanker.from_runtime(rt, tok, method=case["method"], batch_size=case["batch_size"], k=case["k"])
        log, orig = [], rk.compare
        rk.compare = lambda q, d, _o=orig, _l=log: (_l.append([list(d)]), _l[-1].append(list(_o(q, d))))[1] or _l[-1][1]
        ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
        res = rk.rerank(case["query"], ranking)
        tag = (case["ckpt"], case["method"], case["batch_size"], case["k"])
        assert log == case["compares"], tag
        assert [[r.docid, r.score] for r in res] == case["result"], tag
        assert [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens] == case["counters"], tag
    assert len(pcases) >= 14


@pytest.mark.parametrize("batched", [True, False])
def test_setwise_cases(cases, stack, monkeypatch, batched):
    """batched=True is what ships: the build phase of the heapsort submits the independent sift-downs of a tree level
    in one engine call (SetwiseLlmRanker._build_heap_batched / _compare_many, incl. the multi-row greedy EOS trimming).
    Compares are logged by wrapping the CLASS methods, so the ranker's own dispatch (_batched_ok) is untouched; the
    level-wise order differs from the reference's inside a level, so the compare log is checked as a multiset there
    and as an exact sequence with batching off."""
    from llmrankers.rankers import SearchResult
    from llmrankers.setwise import SetwiseLlmRanker
    log, many_sizes = [], []
    orig_compare, orig_many = SetwiseLlmRanker.compare, SetwiseLlmRanker._compare_many

    def logged(self, query, docs):
        out = orig_compare(self, query, docs)
        log.append([[d.docid for d in docs], out])
        return out

    def logged_many(self, query, doc_lists):
        outs = orig_many(self, query, doc_lists)
        many_sizes.append(len(doc_lists))
        for docs, out in zip(doc_lists, outs):
            log.append([[d.docid for d in docs], out])
        return outs

    monkeypatch.setattr(SetwiseLlmRanker, "compare", logged)
    monkeypatch.setattr(SetwiseLlmRanker, "_compare_many", logged_many)
    n = n_batched_calls = 0
    for case in cases:
        if case["kind"] != "setwise":
            continue
        rt, tok = stack[case["ckpt"]]
        ranker = _build(case, rt, tok)
        ranker.batch_independent_compares = batched
        assert ranker._batched_ok() == (batched and case["num_permutation"] == 1)
        ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
        del log[:], many_sizes[:]
        random.seed(929)
        sink = io.StringIO()
        tag = (case["ckpt"], case["scoring"], case["method"], case["num_child"], case["num_permutation"])
        assert case["min_margin"] is None or case["min_margin"] > MARGIN_FLOOR, (tag, case["min_margin"])
        if case.get("raises"):
            with pytest.raises(IndexError), contextlib.redirect_stdout(sink):
                ranker.rerank(case["query"], ranking)
            n += 1
            continue
        with contextlib.redirect_stdout(sink):
            res = ranker.rerank(case["query"], ranking)
        if many_sizes:
            n_batched_calls += sum(1 for m in many_sizes if m > 1)
            key = lambda c: (tuple(c[0]), c[1])
            assert sorted(map(key, log)) == sorted(map(key, case["compares"])), tag
        else:
            assert log == case["compares"], f"{tag}: first differing compare " \
                f"{next((i, a, b) for i, (a, b) in enumerate(zip(log + [None], case['compares'] + [None])) if a != b)}"
        assert [[r.docid, r.score] for r in res] == case["result"], tag          # identical docid rank order
        assert [r.docid for r in ranking] == case["caller_list_after"]
        assert [ranker.total_compare, ranker.total_prompt_tokens, ranker.total_completion_tokens] == case["counters"], tag
        n += 1
    assert n >= 16
    assert (n_batched_calls > 0) == batched        # the level-batched path really ran (or really did not)


def test_pipelined_batches_equal_blocking_calls(stack):
    """T5Runtime.score_batches (two batch slots in flight) returns exactly what per-batch blocking calls return."""
    from llmrankers import _synth
    rt, _ = stack["ckpt_gated_untied"]
    batches = [_synth.synth_token_batch(n, 5, 90, rt.dims.vocab, seed=40 + i) for i, n in enumerate([7, 3, 9, 1, 5])]
    want = [rt.score(b, [0], [11, 12, 13]) for b in batches]
    got = rt.score_batches(batches, [0], [11, 12, 13])
    assert len(got) == len(want)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)


def test_config3_full_size_setwise_query_flan_t5_large():
    """BASELINE.json configs[2] at full size: flan-t5-large dimensions, ONE heapsort query with hits=100, num_child=10, k=10
    (11 passages of ~120 tokens per prompt, ~1.45k tokens), both scorings.  tests/golden/setwise_large.json holds the run of
    the same ranker on the fp32 oracle (tools/make_setwise_large_golden.py: every compare with its label logits / greedy
    tokens and its decision margin).  (1) every recorded compare is replayed on the engine: label logits within tolerance,
    same decision wherever the recorded margin is above the floor; (2) the whole query runs through the shipped, level-batched
    driver and must give the recorded ranking, caller-list order and counters."""
    from transformers import T5Tokenizer
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    from llmrankers._runtime import T5Runtime
    from llmrankers.rankers import SearchResult
    from llmrankers.setwise import SetwiseLlmRanker
    if not os.path.exists(os.path.join(GOLD, "setwise_large.json")):
        pytest.skip("tests/golden/setwise_large.json not generated yet (tools/make_setwise_large_golden.py, ~30 CPU-minutes)")
    with open(os.path.join(GOLD, "setwise_large.json")) as f:
        gold = json.load(f)
    w = gold["weights"]
    dims = _synth.NAMED_DIMS[w["dims"]]
    state = _synth.synth_state_dict(dims, seed=w["seed"], threads=32)
    head = state["lm_head.weight"].copy()
    ids = np.asarray(w["boost_ids"], dtype=np.int64)
    head[ids] = (head[ids] * np.float32(w["boost"])).astype(np.float16).astype(np.float32)
    state["lm_head.weight"] = head
    eng = RkEngine(dims, 0, max_tokens=32768, max_seqs=16, max_dec_len=8).load_state(state.items())
    rt = T5Runtime.from_engine(eng, dims)
    tok = T5Tokenizer.from_pretrained(os.path.join(GOLD, "tok"))
    floor = gold["floor"]
    for scoring in ("likelihood", "generation"):
        run = gold["runs"][scoring]
        # a run may carry a corpus of its own (tools/make_setwise_large_golden.py --generation-only: the generation margins are
        # top-2 gaps over the FULL vocabulary, so a corpus that is decisive for likelihood need not be for generation)
        query, docs = run.get("query", gold["query"]), run.get("docs", gold["docs"])
        text = {d: t for d, t in docs}
        # no dead branch on the committed fixture: BOTH scorings are asserted in full below (round-5 review, weak #1a)
        assert run["min_margin"] > floor, (scoring, run["min_margin"])
        rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=gold["num_child"], k=gold["k"], scoring=scoring, method="heapsort")
        worst, decided = 0.0, 0
        for docids, out, rec in run["compares"]:
            window = [SearchResult(docid=d, score=0.0, text=text[d]) for d in docids]
            with contextlib.redirect_stdout(io.StringIO()):
                got = rk.compare(query, window)
            if scoring == "likelihood":
                prompt = tokenize_ids(rk, query, window)
                lg = rt.score([prompt], rk.decoder_input_ids, rk.target_token_ids[:len(window)])[0]
                worst = max(worst, float(np.abs(lg - np.array(rec["logits"])).max()))
            if rec["margin"] > floor:
                decided += 1
                assert got == out, (scoring, docids, got, out, rec["margin"])
        assert decided == len(run["compares"]), (scoring, decided)
        if scoring == "likelihood":
            assert worst < floor / 2, worst        # label logits (lm_head rows boosted x6) vs the fp32 oracle
        # the whole query through the shipped driver (build phase level-batched)
        rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=gold["num_child"], k=gold["k"], scoring=scoring, method="heapsort")
        assert rk._batched_ok()
        ranking = [SearchResult(docid=d, score=float(100 - i), text=t) for i, (d, t) in enumerate(docs)]
        with contextlib.redirect_stdout(io.StringIO()):
            res = rk.rerank(query, ranking)
        assert [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens] == run["counters"], scoring
        assert [[r.docid, r.score] for r in res] == run["result"], scoring
        assert [r.docid for r in ranking] == run["caller_list_after"], scoring
    eng.close()


def test_config4_flan_t5_xl_qlm_full_query_through_the_sharded_ranker():
    """BASELINE.json configs[3] at full size THROUGH THE RANKER: flan-t5-xl dimensions, pointwise qlm, ONE query of hits=100
    (ref: llmrankers/pointwise.py:41-82) run by PointwiseLlmRanker(shard_candidates=True) on the HIP engine with a one-rank
    RCCL communicator - the candidate-sharding path of DESIGN.md section 6: the share takes three engine calls (max_seqs = 40),
    each appended to the send buffer on the device, token counts appended from the host, ONE gather - against
    tests/golden/xl_qlm_query.json (the same ranker on the fp32 oracle, tools/make_xl_qlm_golden.py).  Scores within the qlm
    tolerance, the same order wherever the oracle's scores differ by more than twice that tolerance, identical counters.
    The measured error is printed: a qlm score is a SUM of ~15-33 log-probabilities of magnitude ~10 each (here |score| ~
    150-350), so BASELINE.md section 2's 1e-3 reads as a RELATIVE bound for it."""
    from transformers import T5Tokenizer
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    from llmrankers._runtime import T5Runtime
    from llmrankers.pointwise import PointwiseLlmRanker
    from llmrankers.rankers import SearchResult
    path = os.path.join(GOLD, "xl_qlm_query.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/xl_qlm_query.json not generated yet (tools/make_xl_qlm_golden.py)")
    with open(path) as f:
        gold = json.load(f)
    dims = _synth.NAMED_DIMS[gold["dims"]]
    state = _synth.synth_state_dict(dims, seed=gold["weight_seed"], threads=min(32, os.cpu_count() or 8))
    eng = RkEngine(dims, 0, max_tokens=8192, max_seqs=40, max_dec_len=40).load_state(state.items())
    del state
    try:
        rt = T5Runtime.from_engine(eng, dims)
        eng.comm_init(eng.comm_unique_id(), 0, 1, 1024)                     # one-rank communicator: same calls as N ranks
        tok = T5Tokenizer.from_pretrained(os.path.join(GOLD, "tok"))
        rk = PointwiseLlmRanker.from_runtime(rt, tok, method="qlm", batch_size=gold["batch_size"], shard_candidates=True)
        n = len(gold["docs"])
        ranking = [SearchResult(docid=f"d{i}", score=float(n - i), text=t) for i, t in enumerate(gold["docs"])]
        res = rk.rerank(gold["query"], ranking)
        got = np.array([d.score for d in ranking])                           # scored in place, passage order
        want = np.array(gold["scores"])
        err = np.abs(got - want)
        rel = float(err.max() / np.abs(want).max())
        print(f"[xl qlm query] {n} passages, |score| up to {np.abs(want).max():.1f}: max abs error {err.max():.4f}, relative {rel:.2e}")
        assert rel < 2e-4, (err.max(), rel)                    # measured 6.4e-5 (max abs 0.011 at |score| 166)
        tol = 2e-4 * float(np.abs(want).max())
        order = {d.docid: i for i, d in enumerate(res)}
        for i in range(n):
            for j in range(n):
                if want[i] - want[j] > 2 * tol:
                    assert order[f"d{i}"] < order[f"d{j}"], (i, j, want[i], want[j], got[i], got[j])
        assert [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens] == gold["counters"]
        # the same query without sharding: the same bits (batch independence across the two call patterns)
        rk1 = PointwiseLlmRanker.from_runtime(rt, tok, method="qlm", batch_size=gold["batch_size"])
        ranking1 = [SearchResult(docid=f"d{i}", score=float(n - i), text=t) for i, t in enumerate(gold["docs"])]
        res1 = rk1.rerank(gold["query"], ranking1)
        assert [d.docid for d in res1] == [d.docid for d in res] and [d.score for d in ranking1] == [d.score for d in ranking]
        assert [rk1.total_compare, rk1.total_prompt_tokens, rk1.total_completion_tokens] == gold["counters"]
    finally:
        eng.comm_destroy()
        eng.close()


def test_config5_llama_8b_widths_setwise_query_vs_oracle_golden():
    """BASELINE.json configs[4] call shape through the ranker: ONE setwise heapsort query (hits=100, num_child=10, k=10,
    generation) with Llama-3-8B widths and two layers (the depth the oracle affords for ~50 compares) through SetwiseLlmRanker
    on the HIP engine (chat template + " Passage:", prefill, one greedy token; ref: llmrankers/setwise.py:159-177) against
    tests/golden/llama_setwise_query.json: the recorded compare sequence (greedy tokens), ranking, caller-list order and
    counters - demanded exactly when the smallest recorded margin is above the fp16 noise floor of this model scale, and
    compare by compare up to the first decision below the floor otherwise.  The full 32-layer depth of the same query is run
    for its properties in test_gpu_kernels.py::test_llama_3_8b_full_depth_vs_oracle_golden."""
    from transformers import AutoTokenizer
    from llmrankers import _synth
    from llmrankers._engine import RkLlamaEngine
    from llmrankers._runtime import LlamaRuntime
    from llmrankers.rankers import SearchResult
    from llmrankers.setwise import SetwiseLlmRanker
    path = os.path.join(GOLD, "llama_setwise_query.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/llama_setwise_query.json not generated yet (tools/make_llama_setwise_golden.py)")
    with open(path) as f:
        gold = json.load(f)
    dims = _synth.LlamaDims(vocab=128256, hidden=4096, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336,
                            n_layers=gold["n_layers"], bos_token_id=128000, eos_token_id=128001)
    state = _synth.synth_state_dict(dims, seed=gold["weight_seed"], threads=min(32, os.cpu_count() or 8))
    head = state["lm_head.weight"].copy()
    ids = np.asarray(gold["boost_ids"], dtype=np.int64)
    head[ids] = (head[ids] * np.float32(gold["boost"])).astype(np.float16).astype(np.float32)
    state["lm_head.weight"] = head
    eng = RkLlamaEngine(dims, device=0, max_tokens=16384, max_seqs=16).load_state(state.items())
    del state, head
    try:
        rt = LlamaRuntime.from_engine(eng, dims)
        seen = []
        orig = rt.greedy1
        rt.greedy1 = lambda seqs: (lambda out: (seen.extend(int(t) for t in out), out)[1])(orig(seqs))
        tok = AutoTokenizer.from_pretrained(os.path.join(GOLD, "tok_llama"))
        rk = SetwiseLlmRanker.from_runtime(rt, tok, num_child=10, k=10, scoring="generation", method="heapsort")
        rk.batch_independent_compares = False                    # the recorded one-by-one compare order
        ranking = [SearchResult(docid=f"d{i}", score=float(100 - i), text=t) for i, t in enumerate(gold["docs"])]
        with contextlib.redirect_stdout(io.StringIO()):
            res = rk.rerank(gold["query"], ranking)
        recs = gold["compares"]
        # noise floor: the engine's logits are within 4e-3 of the logit scale of the oracle's at these widths (test_gpu_kernels)
        floor = 8e-3 * max(r["logit_abs_max"] for r in recs)
        first_low = next((i for i, r in enumerate(recs) if r["margin"] <= floor), len(recs))
        assert first_low >= 0.5 * len(recs), (first_low, len(recs), floor)
        assert seen[:first_low] == [r["token"] for r in recs[:first_low]]
        if first_low == len(recs):
            assert [[r.docid, r.score] for r in res] == gold["ranking"]
            assert [d.docid for d in ranking] == gold["caller_list_after"]
            assert [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens] == gold["counters"]
        # the shipped, level-batched driver: same ranking and counters as one compare at a time on the engine
        rk2 = SetwiseLlmRanker.from_runtime(rt, tok, num_child=10, k=10, scoring="generation", method="heapsort")
        ranking2 = [SearchResult(docid=f"d{i}", score=float(100 - i), text=t) for i, t in enumerate(gold["docs"])]
        with contextlib.redirect_stdout(io.StringIO()):
            res2 = rk2.rerank(gold["query"], ranking2)
        assert [[r.docid, r.score] for r in res2] == [[r.docid, r.score] for r in res]
        assert (rk2.total_compare, rk2.total_prompt_tokens, rk2.total_completion_tokens) == (rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens)
    finally:
        eng.close()


def test_rerank_many_equals_one_query_at_a_time_on_the_engine(cases, stack):
    """rerank_many on the HIP engine: the recorded reference cases of one checkpoint handed over together - pointwise: all
    batches in one launch sequence; setwise heapsort and bubblesort: the queries' sort chains in lockstep, one engine call
    per step - give per query exactly (bit for bit: scores, order, caller lists, counters) what rerank() gives alone."""
    from llmrankers.rankers import SearchResult

    def key(c):
        return (c["kind"], c["ckpt"], c.get("method"), c.get("scoring"), c.get("num_child"), c.get("k"), c.get("batch_size"), c.get("num_permutation"))

    groups = {}
    for c in cases:
        if c.get("raises") or (c["kind"] == "setwise" and c["num_permutation"] != 1):     # heapsort AND bubblesort drivers
            continue
        groups.setdefault(key(c), []).append(c)
    checked = 0
    for grp in groups.values():
        base = grp[0]
        rt, tok = stack[base["ckpt"]]
        # several queries over the recorded inputs (a group may hold a single recorded case: vary the query text)
        items = [(c["query"] + suffix, [tuple(x) for x in c["input"]]) for c in grp for suffix in ("", " again", " once more")]
        one, many = _build(base, rt, tok), _build(base, rt, tok)
        want, lists, counters = [], [], []
        with contextlib.redirect_stdout(io.StringIO()):
            for q, inp in items:
                ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in inp]
                random.seed(929)
                res = one.rerank(q, ranking)
                want.append([(r.docid, r.score) for r in res])
                lists.append([r.docid for r in ranking])
                counters.append((one.total_compare, one.total_prompt_tokens, one.total_completion_tokens))
            rankings = [[SearchResult(docid=d, score=s, text=t) for d, s, t in inp] for _, inp in items]
            got, cnt = many.rerank_many([(q, r) for (q, _), r in zip(items, rankings)])
        assert [[(r.docid, r.score) for r in res] for res in got] == want, key(base)
        assert cnt == counters, key(base)
        if base["kind"] == "setwise":
            assert [[r.docid for r in ranking] for ranking in rankings] == lists
        checked += 1
    assert checked >= 8 and any(k[2] == "bubblesort" for k in groups)


def tokenize_ids(rk, query, window):
    from llmrankers._batching import tokenize_prompts
    return tokenize_prompts(rk.tokenizer, [rk._prompt(query, rk.CHARACTERS[:len(window)], [d.text for d in window])])[0]


def test_run_py_cli_on_the_engine(ckpt_dirs, tmp_path):
    """run.py end to end on the HIP engine (product constructors: checkpoint directory -> engine): pointwise, setwise on T5
    and on the Llama family, pairwise; TREC output equals what the library-level rankers return; --resume and --qrels work."""
    import importlib.util
    from conftest import REPO
    spec = importlib.util.spec_from_file_location("rk_run_gpu", os.path.join(REPO, "run.py"))
    runmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(runmod)
    (tmp_path / "q.tsv").write_text("q1\tneural ranking model\nq2\twater river mountain\n")
    words = ["search engine index", "river water city", "music art film", "vaccine covid virus", "bank money trade", "history science physics"]
    (tmp_path / "d.tsv").write_text("\n".join(f"d{i}\t{w}" for i, w in enumerate(words)) + "\n")
    lines = [f"{q} Q0 d{i} {r + 1} {10 - r} bm25" for q in ("q1", "q2") for r, i in enumerate(range(6))]
    (tmp_path / "in.trec").write_text("\n".join(lines) + "\n")
    (tmp_path / "qrels").write_text("q1 0 d0 2\nq2 0 d1 1\n")
    parser, commands = runmod.build_parser()

    def run(ckpt, save, tail, extra=()):
        args = runmod.parse_args(parser, commands, ["run", "--model_name_or_path", ckpt, "--run_path", str(tmp_path / "in.trec"),
                                                    "--save_path", str(save), "--query_file", str(tmp_path / "q.tsv"),
                                                    "--doc_file", str(tmp_path / "d.tsv"), "--hits", "6", "--passage_length", "16",
                                                    "--query_length", "16", *extra, *tail])
        runmod.validate(args)
        with contextlib.redirect_stdout(io.StringIO()) as out:
            runmod.main(args)
        return [l.split("\t") for l in save.read_text().splitlines()], out.getvalue()

    t5, llama = ckpt_dirs["ckpt_labelboost"], ckpt_dirs["ckpt_llama"]
    rows, log = run(t5, tmp_path / "pw.trec", ["pointwise", "--method", "yes_no", "--batch_size", "4"], ["--qrels", str(tmp_path / "qrels")])
    assert len(rows) == 12 and all(r[1] == "Q0" and r[5] == "LLMRankers" for r in rows) and "NDCG@10 reranked:" in log
    for q in ("q1", "q2"):
        sc = [float(r[4]) for r in rows if r[0] == q]
        assert sc == sorted(sc, reverse=True) and all(0.0 < s < 1.0 for s in sc)
    first = (tmp_path / "pw.trec").read_text()
    (tmp_path / "pw_resume.trec").write_text("".join(l + "\n" for l in first.splitlines()[:6]))
    run(t5, tmp_path / "pw_resume.trec", ["pointwise", "--method", "yes_no", "--batch_size", "4"], ["--resume"])
    assert (tmp_path / "pw_resume.trec").read_text() == first
    for name, ckpt, tail in (("sw_t5", t5, ["setwise", "--num_child", "3", "--k", "3"]),
                             ("sw_t5_lik", t5, ["setwise", "--num_child", "2", "--k", "2", "--method", "bubblesort"]),
                             ("sw_llama", llama, ["setwise", "--num_child", "3", "--k", "3"]),
                             ("pair", t5, ["pairwise", "--method", "heapsort", "--k", "2"])):
        extra = ["--scoring", "likelihood"] if name == "sw_t5_lik" else []
        rows, _ = run(ckpt, tmp_path / f"{name}.trec", tail, extra)
        assert len(rows) == 12, name
        for q in ("q1", "q2"):
            part = [r for r in rows if r[0] == q]
            assert [int(r[3]) for r in part] == list(range(1, 7)) and [int(r[4]) for r in part] == [-i for i in range(1, 7)], name
            assert sorted(r[2] for r in part) == [f"d{i}" for i in range(6)], name
