"""Multi-process candidate sharding on CPU: world_size 2, gloo backend, rendezvous on 127.0.0.1.

Exercises the N>1 path of the pointwise ranker (llmrankers/_dist.py): contiguous shards, one all_gather, identical
scores and rank order on every rank and identical to the single-process result.  The model is the oracle-backed
test stub (no GPU here); on the GPU box the same code runs with backend 'nccl' (= RCCL over xGMI)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import GOLD, REPO


def test_shard_bounds():
    from llmrankers._dist import shard_bounds
    assert [e - s for s, e in shard_bounds(100, 8)] == [13, 13, 13, 13, 12, 12, 12, 12]
    assert shard_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert shard_bounds(0, 2) == [(0, 0), (0, 0)]
    for n in (1, 7, 100, 101):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:]))


WORKER = r'''
import json, os, sys
sys.path[:0] = [os.path.join(sys.argv[1], "llm-rankers_amd"), sys.argv[1], os.path.join(sys.argv[1], "tests")]
import torch.distributed as dist
from conftest import load_state
from _stub import OracleRuntime
from llmrankers import _synth
from llmrankers.rankers import SearchResult
from llmrankers.pointwise import PointwiseLlmRanker
from transformers import T5Tokenizer
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=int(sys.argv[4]))
ck = sys.argv[5]
dims, state = load_state(ck)
case = json.load(open(sys.argv[6]))
tok = T5Tokenizer.from_pretrained(ck)
rk = PointwiseLlmRanker.from_runtime(OracleRuntime(dims, state), tok, method=case["method"], batch_size=case["batch_size"], shard_candidates=True)
ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
res = rk.rerank(case["query"], ranking)
print("RESULT " + json.dumps([[r.docid, r.score] for r in res]))
dist.destroy_process_group()
'''


def test_two_rank_sharded_rerank_matches_reference(ckpt_dirs, tmp_path):
    with open(os.path.join(GOLD, "rerank_cases.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["kind"] == "pointwise" and c["ckpt"] == "ckpt_gated_untied"]
    case = next(c for c in cases if c["method"] == "yes_no" and len(c["input"]) == 13)   # 13 docs -> shards of 7 and 6
    cpath = tmp_path / "case.json"
    cpath.write_text(json.dumps(case))
    wpath = tmp_path / "worker.py"
    wpath.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(wpath), REPO, str(port), str(r), "2", ckpt_dirs["ckpt_gated_untied"], str(cpath)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(next(l for l in out.splitlines() if l.startswith("RESULT "))[7:]))
    assert outs[0] == outs[1]                                          # every rank holds the same final ranking
    assert [d for d, _ in outs[0]] == [d for d, _ in case["result"]]   # = the reference's order
    np.testing.assert_allclose([s for _, s in outs[0]], [s for _, s in case["result"]], atol=2e-5)


WORKER_ENGINE = r'''
import json, os, sys
sys.path[:0] = [os.path.join(sys.argv[1], "llm-rankers_amd"), sys.argv[1], os.path.join(sys.argv[1], "tests")]
import torch.distributed as dist
from conftest import load_state
from _stub import FakeCommEngine
from llmrankers._runtime import T5Runtime
from llmrankers.rankers import SearchResult
from llmrankers.pointwise import PointwiseLlmRanker
from transformers import T5Tokenizer
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=int(sys.argv[4]))
ck = sys.argv[5]
dims, state = load_state(ck)
case = json.load(open(sys.argv[6]))
tok = T5Tokenizer.from_pretrained(ck)
eng = FakeCommEngine(dims, state, max_seqs=3)
rt = T5Runtime.from_engine(eng, dims)                 # the REAL runtime: ensure_comm / sharded_scores / chunking
rk = PointwiseLlmRanker.from_runtime(rt, tok, method=case["method"], batch_size=case["batch_size"], shard_candidates=True)
out = []
for rep in range(2):                                  # two queries: the communicator is built once, one gather each
    ranking = [SearchResult(docid=d, score=s, text=t) for d, s, t in case["input"]]
    res = rk.rerank(case["query"], ranking)
    out.append([[r.docid, r.score] for r in res])
print("RESULT " + json.dumps({"rankings": out, "calls": eng.calls, "empty": rk.rerank(case["query"], [])}))
dist.destroy_process_group()
'''


def _run_two_ranks(tmp_path, worker_src, ckpt, case):
    cpath = tmp_path / "case.json"
    cpath.write_text(json.dumps(case))
    wpath = tmp_path / "worker.py"
    wpath.write_text(worker_src)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(wpath), REPO, str(port), str(r), "2", ckpt, str(cpath)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(next(l for l in out.splitlines() if l.startswith("RESULT "))[7:]))
    return outs


def test_two_rank_engine_gather_path_with_multi_chunk_shards(ckpt_dirs, tmp_path):
    """The branch a GPU run takes: PointwiseLlmRanker(shard_candidates=True) -> T5Runtime.ensure_comm builds the engine
    communicator from the process group on the first sharded query -> T5Runtime.sharded_scores scores the local share in
    SEVERAL engine calls (max_seqs = 3 here, shares of 7 + 6 and 5 + 5 passages), appends each call's scores to the send buffer and
    ships the whole share with ONE gather.  (Round 2's path gathered slot 0's buffer after the last call only: every chunk
    but the last was lost.)  Same ranking and scores as the reference on every rank."""
    with open(os.path.join(GOLD, "rerank_cases.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["kind"] == "pointwise" and c["ckpt"] == "ckpt_gated_untied"]
    for method, n_docs in (("yes_no", 13), ("qlm", 10)):
        case = next(c for c in cases if c["method"] == method and len(c["input"]) == n_docs)
        outs = _run_two_ranks(tmp_path, WORKER_ENGINE, ckpt_dirs["ckpt_gated_untied"], case)
        assert outs[0]["rankings"] == outs[1]["rankings"]
        for rank, o in enumerate(outs):
            n_local = (n_docs + 1 - rank) // 2               # 13 -> 7, 6;  10 -> 5, 5
            n_chunks = -(-n_local // 3)
            assert o["calls"]["init"] == 1 and o["calls"]["gather"] == 2, o["calls"]
            assert o["calls"]["append"] == 2 * n_chunks and o["calls"]["qlm" if method == "qlm" else "score"] == 2 * n_chunks, o["calls"]
            assert o["empty"] == []
            for ranking in o["rankings"]:
                assert [d for d, _ in ranking] == [d for d, _ in case["result"]]
                np.testing.assert_allclose([s for _, s in ranking], [s for _, s in case["result"]], atol=2e-5, rtol=1e-5)


WORKER_MANY = r'''
import json, os, sys
sys.path[:0] = [os.path.join(sys.argv[1], "llm-rankers_amd"), sys.argv[1], os.path.join(sys.argv[1], "tests")]
import torch.distributed as dist
from conftest import load_state
from _stub import FakeCommEngine, OracleRuntime
from llmrankers._runtime import T5Runtime
from llmrankers.rankers import SearchResult
from llmrankers.pointwise import PointwiseLlmRanker
from transformers import T5Tokenizer
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=int(sys.argv[4]))
ck = sys.argv[5]
dims, state = load_state(ck)
cases = json.load(open(sys.argv[6]))
tok = T5Tokenizer.from_pretrained(ck)
out = {}
for name in ("double", "engine"):
    eng = FakeCommEngine(dims, state, max_seqs=4)
    rt = OracleRuntime(dims, state) if name == "double" else T5Runtime.from_engine(eng, dims)
    rk = PointwiseLlmRanker.from_runtime(rt, tok, method="yes_no", batch_size=cases[0]["batch_size"], shard_candidates=True)
    items = [(c["query"], [SearchResult(docid=d, score=s, text=t) for d, s, t in c["input"]]) for c in cases]
    items.append((cases[0]["query"], []))                     # a query without candidates rides along
    ranked, counters = rk.rerank_many(items)
    one = []
    for c in cases:                                            # the same queries one at a time: rankings, scores, counters
        r = rk.rerank(c["query"], [SearchResult(docid=d, score=s, text=t) for d, s, t in c["input"]])
        one.append([[[x.docid, x.score] for x in r], [rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens]])
    out[name] = {"many": [[[x.docid, x.score] for x in r] for r in ranked], "counters": [list(c) for c in counters], "one": one,
                 "calls": dict(eng.calls)}
print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


def test_two_rank_sharded_rerank_many_groups_the_shares_of_several_queries(ckpt_dirs, tmp_path):
    """Round 6: PointwiseLlmRanker.rerank_many under candidate sharding - every rank's share of EVERY query in one launch sequence,
    ONE gather for all of them (scores + token counts) - against the same queries one at a time and the reference's recorded
    results: rankings, scores, counters; through the test double's host gather and through the real T5Runtime (engine
    communicator path: one init, ONE gather for the group, the scores appended chunk by chunk)."""
    with open(os.path.join(GOLD, "rerank_cases.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["kind"] == "pointwise" and c["ckpt"] == "ckpt_gated_untied" and c["method"] == "yes_no"]
    bs = cases[0]["batch_size"]
    cases = [c for c in cases if c["batch_size"] == bs][:3]
    assert len(cases) >= 2
    outs = _run_two_ranks(tmp_path, WORKER_MANY, ckpt_dirs["ckpt_gated_untied"], cases)
    assert outs[0] == outs[1] or all(outs[0][k]["many"] == outs[1][k]["many"] for k in outs[0])   # every rank holds the same rankings
    for rank, o in enumerate(outs):
        for name in ("double", "engine"):
            r = o[name]
            assert r["many"][-1] == [] and r["counters"][-1] == [0, 0, 0]
            for c, many, cnt, (one, one_cnt) in zip(cases, r["many"], r["counters"], r["one"]):
                assert [d for d, _ in many] == [d for d, _ in c["result"]] == [d for d, _ in one]
                np.testing.assert_allclose([s for _, s in many], [s for _, s in c["result"]], atol=2e-5, rtol=1e-5)
                np.testing.assert_allclose([s for _, s in many], [s for _, s in one], atol=1e-6)
                assert cnt == one_cnt == list(c["counters"]) if "counters" in c else cnt == one_cnt
        calls = o["engine"]["calls"]
        # grouped call: ONE gather; then the three single queries: one gather each
        assert calls["init"] == 1 and calls["gather"] == 1 + len(cases), calls
