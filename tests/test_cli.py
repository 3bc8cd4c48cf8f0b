"""run.py: reference command-line surface (ref: run.py:20-38, 41-49, 206-259) — CPU only."""
import importlib.util
import os
import sys

import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def runmod():
    spec = importlib.util.spec_from_file_location("rk_run", os.path.join(REPO, "run.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_argument_surface_and_defaults(runmod):
    parser, commands = runmod.build_parser()
    a = runmod.parse_args(parser, commands, ["run", "--model_name_or_path", "m", "--run_path", "r", "--save_path", "s",
                                             "pointwise", "--method", "yes_no", "--batch_size", "32"])
    runmod.validate(a)
    assert (a.run.hits, a.run.query_length, a.run.passage_length, a.run.device, a.run.scoring) == (100, 128, 128, "cuda", "generation")
    assert (a.pointwise.method, a.pointwise.batch_size) == ("yes_no", 32) and a.setwise is None
    b = runmod.parse_args(parser, commands, ["run", "--model_name_or_path", "m", "setwise"])
    assert (b.setwise.num_child, b.setwise.method, b.setwise.k, b.setwise.num_permutation) == (3, "heapsort", 10, 1)
    with pytest.raises(ValueError):
        runmod.validate(runmod.parse_args(parser, commands, ["run", "--model_name_or_path", "m"]))
    with pytest.raises(ValueError):
        runmod.validate(runmod.parse_args(parser, commands, ["run", "pointwise", "setwise"]))
    with pytest.raises(ValueError):
        runmod.validate(runmod.parse_args(parser, commands, ["run", "--ir_dataset_name", "x", "--pyserini_index", "y", "pointwise"]))


def test_trec_round_trip_with_file_sources(runmod, tmp_path, ckpt_dirs, monkeypatch):
    from conftest import load_state
    from _stub import OracleRuntime
    from transformers import T5Tokenizer
    from llmrankers.pointwise import PointwiseLlmRanker
    ck = ckpt_dirs["ckpt_gated_untied"]
    dims, state = load_state(ck)
    tok = T5Tokenizer.from_pretrained(ck)
    monkeypatch.setattr(runmod, "build_ranker", lambda args: PointwiseLlmRanker.from_runtime(OracleRuntime(dims, state), tok, method=args.pointwise.method, batch_size=args.pointwise.batch_size))
    (tmp_path / "q.tsv").write_text("q1\tneural ranking model\nq2\twater river mountain\n")
    (tmp_path / "d.jsonl").write_text("\n".join(
        '{"docid": "d%d", "title": "topic", "text": "%s"}' % (i, w) for i, w in enumerate(
            ["search engine index", "river water city", "music art film", "vaccine covid virus", "bank money trade"])) + "\n")
    lines = [f"{q} Q0 d{i} {r + 1} {10 - r} bm25" for q in ("q1", "q2") for r, i in enumerate([0, 1, 2, 3, 4])]
    (tmp_path / "in.trec").write_text("\n".join(lines) + "\n")
    parser, commands = runmod.build_parser()
    args = runmod.parse_args(parser, commands, ["run", "--model_name_or_path", ck, "--run_path", str(tmp_path / "in.trec"),
                                                "--save_path", str(tmp_path / "out.trec"), "--query_file", str(tmp_path / "q.tsv"),
                                                "--doc_file", str(tmp_path / "d.jsonl"), "--hits", "4", "--query_length", "32",
                                                "--passage_length", "16", "pointwise", "--method", "yes_no", "--batch_size", "3"])
    runmod.validate(args)
    runmod.main(args)
    out = (tmp_path / "out.trec").read_text().splitlines()
    assert len(out) == 8                                        # hits=4 per query
    for k, line in enumerate(out):
        qid, q0, docid, rank, score, tag = line.split("\t")
        assert q0 == "Q0" and tag == "LLMRankers" and int(rank) == k % 4 + 1 and 0.0 < float(score) < 1.0
    for q in ("q1", "q2"):
        sc = [float(l.split("\t")[4]) for l in out if l.startswith(q + "\t")]
        assert sc == sorted(sc, reverse=True)


def test_ndcg_and_shards(runmod, tmp_path):
    import math
    rels = {"a": 3, "b": 0, "c": 1, "d": 2}
    got = runmod.ndcg_at_k(["b", "a", "x", "c"], rels, k=10)
    dcg = 3 / math.log2(3) + 1 / math.log2(5)
    idcg = 3 / math.log2(2) + 2 / math.log2(3) + 1 / math.log2(4)
    assert abs(got - dcg / idcg) < 1e-12
    assert runmod.ndcg_at_k(["a", "d", "c"], rels) == 1.0 and runmod.ndcg_at_k(["x"], rels) == 0.0
    assert runmod.ndcg_at_k(["a"], {"a": 0}) == 0.0                              # no relevant document: 0, not a division by zero
    assert runmod.ndcg_at_k(list("abcd") * 5, rels, k=2) == runmod.ndcg_at_k(["a", "b"], rels, k=2)
    (tmp_path / "q.rels").write_text("q1 0 a 3\nq1 0 b 0\nq2 0 c 1\nq3 0 z 1\n")
    q = runmod.read_qrels(str(tmp_path / "q.rels"))
    assert q == {"q1": {"a": 3, "b": 0}, "q2": {"c": 1}, "q3": {"z": 1}}
    assert abs(runmod.mean_ndcg({"q1": ["a", "b"], "q2": ["x", "c"], "q9": ["a"]}, q) - (1.0 + 1 / math.log2(3)) / 2) < 1e-12
    assert [len(x) for x in runmod.split_into_shards(list(range(10)), 4)] == [3, 3, 2, 2]
    assert sum(runmod.split_into_shards(list(range(10)), 4), []) == list(range(10))


def test_resume_appends_and_skips_done_queries(runmod, tmp_path, ckpt_dirs, monkeypatch):
    """--resume: each query's lines are appended as soon as it is ranked; a restarted run skips the qids already in
    the file and ends with the same file as an uninterrupted run (ref: Rank-R1/run_setwise.py:79-87, 284-291)."""
    from conftest import load_state
    from _stub import OracleRuntime
    from transformers import T5Tokenizer
    from llmrankers.pointwise import PointwiseLlmRanker
    ck = ckpt_dirs["ckpt_gated_untied"]
    dims, state = load_state(ck)
    tok = T5Tokenizer.from_pretrained(ck)
    calls = []

    def make(args):
        rk = PointwiseLlmRanker.from_runtime(OracleRuntime(dims, state), tok, method=args.pointwise.method, batch_size=args.pointwise.batch_size)
        orig = rk.rerank
        rk.rerank = lambda q, r: (calls.append(q), orig(q, r))[1]
        return rk

    monkeypatch.setattr(runmod, "build_ranker", make)
    (tmp_path / "q.tsv").write_text("q1\tneural ranking model\nq2\twater river mountain\nq3\tmusic art film\n")
    (tmp_path / "d.tsv").write_text("\n".join(f"d{i}\t{w}" for i, w in enumerate(
        ["search engine index", "river water city", "music art film", "vaccine covid virus"])) + "\n")
    lines = [f"{q} Q0 d{i} {r + 1} {10 - r} bm25" for q in ("q1", "q2", "q3") for r, i in enumerate([0, 1, 2, 3])]
    (tmp_path / "in.trec").write_text("\n".join(lines) + "\n")
    (tmp_path / "qrels").write_text("q1 0 d0 2\nq2 0 d1 1\nq3 0 d2 3\n")
    parser, commands = runmod.build_parser()

    def run(save, extra=()):
        args = runmod.parse_args(parser, commands, ["run", "--model_name_or_path", ck, "--run_path", str(tmp_path / "in.trec"),
                                                    "--save_path", str(save), "--query_file", str(tmp_path / "q.tsv"),
                                                    "--doc_file", str(tmp_path / "d.tsv"), "--hits", "4", "--qrels", str(tmp_path / "qrels"),
                                                    "--queries_per_call", "1",   # (this test counts rerank() calls: one query at a time)
                                                    *extra, "pointwise", "--method", "yes_no", "--batch_size", "2"])
        runmod.validate(args)
        runmod.main(args)

    run(tmp_path / "full.trec")
    full = (tmp_path / "full.trec").read_text()
    assert len(full.splitlines()) == 12 and len(calls) == 3
    part = tmp_path / "part.trec"
    part.write_text("".join(l + "\n" for l in full.splitlines()[:4]))          # q1 was finished before the "crash"
    del calls[:]
    run(part, ["--resume"])
    assert part.read_text() == full and len(calls) == 2                          # only q2 and q3 were ranked again
    del calls[:]
    run(part, ["--resume"])
    assert part.read_text() == full and calls == []                              # nothing left to do
    # killed in the middle of an append: q2's block is short and its last line is cut - q2 is NOT done, its fragment goes
    fl = full.splitlines()
    part.write_text("".join(l + "\n" for l in fl[:6]) + fl[6][:9])
    del calls[:]
    run(part, ["--resume"])
    assert part.read_text() == full and len(calls) == 2                          # q2 (again) and q3
    # part files are `<save_path>.rank<N>` only; a complete block has as many well-formed lines as the query has candidates
    for name in ("x.trec.rank0", "x.trec.rank12", "x.trec.rank0.bak", "x.trec.rankings"):
        (tmp_path / name).write_text("")
    assert [os.path.basename(f) for f in runmod.part_files(str(tmp_path / "x.trec"))] == ["x.trec.rank0", "x.trec.rank12"]
    (tmp_path / "x.trec.rank0").write_text("".join(l + "\n" for l in fl[:7]))
    assert list(runmod.complete_run_blocks(str(tmp_path / "x.trec.rank0"), {"q1": 4, "q2": 4})) == ["q1"]
    # query shards: two processes, each its half, together the whole run
    run(tmp_path / "s0.trec", ["--dataset_number_of_shards", "2", "--dataset_shard_index", "0"])
    run(tmp_path / "s1.trec", ["--dataset_number_of_shards", "2", "--dataset_shard_index", "1"])
    assert (tmp_path / "s0.trec").read_text() + (tmp_path / "s1.trec").read_text() == full


def test_passage_truncation_is_cached_per_docid(runmod, tmp_path, ckpt_dirs, monkeypatch):
    """The first-stage run lists the same passages for several queries; their tokenise / cut / detokenise round trip
    (ref: pointwise.py:132-133) runs once per docid."""
    from conftest import load_state
    from _stub import OracleRuntime
    from transformers import T5Tokenizer
    from llmrankers.pointwise import PointwiseLlmRanker
    ck = ckpt_dirs["ckpt_gated_untied"]
    dims, state = load_state(ck)
    tok = T5Tokenizer.from_pretrained(ck)
    calls = []

    def make(args):
        rk = PointwiseLlmRanker.from_runtime(OracleRuntime(dims, state), tok, method="yes_no", batch_size=4)
        orig = rk.truncate
        rk.truncate = lambda text, n: (calls.append(text), orig(text, n))[1]
        return rk

    monkeypatch.setattr(runmod, "build_ranker", make)
    (tmp_path / "q.tsv").write_text("q1\tneural ranking model\nq2\twater river mountain\nq3\tart history book\n")
    docs = ["search engine index", "river water city", "music art film", "vaccine covid virus"]
    (tmp_path / "d.tsv").write_text("\n".join(f"d{i}\t{w}" for i, w in enumerate(docs)) + "\n")
    lines = [f"{q} Q0 d{i} {r + 1} {10 - r} bm25" for q in ("q1", "q2", "q3") for r, i in enumerate([0, 1, 2, 3])]
    (tmp_path / "in.trec").write_text("\n".join(lines) + "\n")
    parser, commands = runmod.build_parser()
    args = runmod.parse_args(parser, commands, ["run", "--model_name_or_path", ck, "--run_path", str(tmp_path / "in.trec"),
                                                "--save_path", str(tmp_path / "out.trec"), "--query_file", str(tmp_path / "q.tsv"),
                                                "--doc_file", str(tmp_path / "d.tsv"), "--hits", "4", "pointwise", "--method", "yes_no",
                                                "--batch_size", "4"])
    runmod.validate(args)
    runmod.main(args)
    assert sorted(c for c in calls if c in docs) == sorted(docs)          # 4 passages truncated once each, not 12 times
    assert len((tmp_path / "out.trec").read_text().splitlines()) == 12


def test_queries_per_call_gives_the_same_run_and_statistics(runmod, tmp_path, ckpt_dirs, monkeypatch, capsys):
    """--queries_per_call N (pointwise): N queries per engine launch sequence through PointwiseLlmRanker.rerank_many - the
    TREC file and the printed averages (comparisons, prompt / completion tokens) are those of one query at a time, also
    with --resume and with a last, shorter group; rerank_many itself returns per query what rerank returns (scores, order,
    counters), for yes_no and MonoT5, and takes the one-by-one path for qlm."""
    from conftest import load_state
    from _stub import OracleRuntime
    from transformers import T5Tokenizer
    from llmrankers.pointwise import PointwiseLlmRanker, MonoT5LlmRanker
    from llmrankers.rankers import SearchResult
    ck = ckpt_dirs["ckpt_gated_untied"]
    dims, state = load_state(ck)
    tok = T5Tokenizer.from_pretrained(ck)
    engine_calls = []

    class Counting(OracleRuntime):
        def score_batches(self, batches, dec_prefix, out_ids):           # like T5Runtime: consecutive batches merged into one engine call
            flat = [s for b in batches for s in b]
            engine_calls.append(len(flat))
            return [self.score(flat, dec_prefix, out_ids)]

    monkeypatch.setattr(runmod, "build_ranker", lambda args: PointwiseLlmRanker.from_runtime(
        Counting(dims, state), tok, method=args.pointwise.method, batch_size=args.pointwise.batch_size))
    words = ["search engine index", "river water city", "music art film", "vaccine covid virus", "bank money trade"]
    (tmp_path / "q.tsv").write_text("".join(f"q{i}\t{w}\n" for i, w in enumerate(words)))
    (tmp_path / "d.tsv").write_text("".join(f"d{i}\t{w} {words[(i + 2) % 5]}\n" for i, w in enumerate(words)))
    lines = [f"q{q} Q0 d{(q + r) % 5} {r + 1} {10 - r} bm25" for q in range(5) for r in range(4)]
    (tmp_path / "in.trec").write_text("\n".join(lines) + "\n")
    parser, commands = runmod.build_parser()

    def run(save, extra=()):
        args = runmod.parse_args(parser, commands, ["run", "--model_name_or_path", ck, "--run_path", str(tmp_path / "in.trec"),
                                                    "--save_path", str(save), "--query_file", str(tmp_path / "q.tsv"),
                                                    "--doc_file", str(tmp_path / "d.tsv"), "--hits", "4", *extra,
                                                    "pointwise", "--method", "yes_no", "--batch_size", "3"])
        runmod.validate(args)
        capsys.readouterr()
        runmod.main(args)
        return [l for l in capsys.readouterr().out.splitlines() if l.startswith("Avg") and "time" not in l]

    stats1 = run(tmp_path / "one.trec", ["--queries_per_call", "1"])               # the reference's loop: one query at a time
    n1 = len(engine_calls)
    del engine_calls[:]
    stats_auto = run(tmp_path / "auto.trec")                                       # default = auto (enough queries for 1 600 passages, at most 16 per call)
    assert (tmp_path / "auto.trec").read_text() == (tmp_path / "one.trec").read_text() and stats_auto == stats1
    assert engine_calls == [20]                                                    # all 5 queries x 4 passages in ONE engine call
    del engine_calls[:]
    stats3 = run(tmp_path / "three.trec", ["--queries_per_call", "3"])             # groups of 3 + 2
    assert (tmp_path / "three.trec").read_text() == (tmp_path / "one.trec").read_text()
    assert stats3 == stats1 and len(stats1) == 3
    assert n1 == 5 and engine_calls == [12, 8]                                      # 3 + 2 queries of 4 passages per engine call
    run(tmp_path / "res.trec", ["--queries_per_call", "2", "--resume"])
    assert (tmp_path / "res.trec").read_text() == (tmp_path / "one.trec").read_text()

    def ranking(q):
        return [SearchResult(docid=f"d{(q + r) % 5}", score=float(10 - r), text=f"{words[(q + r) % 5]} and {words[q]}") for r in range(4)]

    for cls, kw in ((PointwiseLlmRanker, {"method": "yes_no"}), (PointwiseLlmRanker, {"method": "qlm"})):
        rk = cls.from_runtime(OracleRuntime(dims, state), tok, batch_size=3, **kw)
        single, counters = [], []
        for q in range(3):
            res = rk.rerank(words[q], ranking(q))
            single.append([(d.docid, d.score) for d in res])
            counters.append((rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens))
        many, cnt = rk.rerank_many([(words[q], ranking(q)) for q in range(3)])
        assert [[(d.docid, d.score) for d in res] for res in many] == single and cnt == counters
        assert (rk.total_compare, rk.total_prompt_tokens, rk.total_completion_tokens) == counters[-1]
    assert PointwiseLlmRanker.from_runtime(OracleRuntime(dims, state), tok, method="yes_no", batch_size=3).rerank_many([]) == ([], [])
    assert MonoT5LlmRanker.rerank_many is PointwiseLlmRanker.rerank_many


RUN_WORKER = r'''
import importlib.util, json, os, sys
repo = sys.argv[1]
sys.path[:0] = [os.path.join(repo, "llm-rankers_amd"), repo, os.path.join(repo, "tests")]
spec = importlib.util.spec_from_file_location("rk_run", os.path.join(repo, "run.py"))
runmod = importlib.util.module_from_spec(spec); spec.loader.exec_module(runmod)
from conftest import load_state
from _stub import FakeCommEngine
from llmrankers._runtime import T5Runtime
from llmrankers.pointwise import PointwiseLlmRanker
from transformers import T5Tokenizer
ck = sys.argv[2]
dims, state = load_state(ck)
tok = T5Tokenizer.from_pretrained(ck)
engines = []
def make(args):
    eng = FakeCommEngine(dims, state, max_seqs=2)
    engines.append(eng)
    return PointwiseLlmRanker.from_runtime(T5Runtime.from_engine(eng, dims), tok, method=args.pointwise.method, batch_size=args.pointwise.batch_size,
                                           shard_candidates=runmod.candidates_sharded(args, runmod.dist_env()[1]))
runmod.build_ranker = make
parser, commands = runmod.build_parser()
args = runmod.parse_args(parser, commands, json.loads(sys.argv[3]))
runmod.validate(args)
runmod.main(args)
print("CALLS " + json.dumps(engines[0].calls))
'''


@pytest.mark.parametrize("shard_flag", ["1", "0"])
def test_two_rank_run_py_matches_single_process(runmod, tmp_path, ckpt_dirs, shard_flag):
    """run.py under a 2-rank launcher environment (gloo): candidate sharding (--shard_candidates 1: every query's candidates
    split over the ranks, engine gather path, several engine calls per share) and query replicas (0: ranks take whole
    queries, rank 0 collects) both write exactly the run file of a single process; only rank 0 writes."""
    import json
    import socket
    import subprocess
    ck = ckpt_dirs["ckpt_gated_untied"]
    (tmp_path / "q.tsv").write_text("q1\tneural ranking model\nq2\twater river mountain\nq3\tmusic art film\n")
    (tmp_path / "d.tsv").write_text("\n".join(f"d{i}\t{w}" for i, w in enumerate(
        ["search engine index", "river water city", "music art film", "vaccine covid virus", "bank money trade",
         "neural model answer", "mountain river water", "film music topic"])) + "\n")
    lines = [f"{q} Q0 d{i} {r + 1} {10 - r} bm25" for q in ("q1", "q2", "q3") for r, i in enumerate([0, 1, 2, 3, 4, 5, 6])]
    (tmp_path / "in.trec").write_text("\n".join(lines) + "\n")
    (tmp_path / "worker.py").write_text(RUN_WORKER)

    def argv(save):
        return ["run", "--model_name_or_path", ck, "--run_path", str(tmp_path / "in.trec"), "--save_path", str(save),
                "--query_file", str(tmp_path / "q.tsv"), "--doc_file", str(tmp_path / "d.tsv"), "--hits", "7",
                "--shard_candidates", shard_flag, "pointwise", "--method", "yes_no", "--batch_size", "3"]

    base_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    single = subprocess.run([sys.executable, str(tmp_path / "worker.py"), REPO, ck, json.dumps(argv(tmp_path / "single.trec"))],
                            capture_output=True, text=True, env=dict(base_env, OMP_NUM_THREADS="2"), timeout=600)
    assert single.returncode == 0, single.stderr[-2000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, str(tmp_path / "worker.py"), REPO, ck, json.dumps(argv(tmp_path / "multi.trec"))],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(base_env, OMP_NUM_THREADS="2", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))) for r in range(2)]
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        outs.append(out)
    assert "Avg comparisons" in outs[0] and "Avg comparisons" not in outs[1]       # rank 0 reports

    def avgs(text):
        return [l for l in text.splitlines() if l.startswith("Avg") and "time" not in l]
    # ... the single-process (= the reference's) statistics: under candidate sharding every rank learns all token counts from
    # the gather, under query replicas rank 0 sums the ranks' counters
    assert avgs(outs[0]) == avgs(single.stdout) and len(avgs(single.stdout)) == 3
    a = [l.split("\t") for l in (tmp_path / "single.trec").read_text().splitlines()]
    b = [l.split("\t") for l in (tmp_path / "multi.trec").read_text().splitlines()]
    assert [x[:4] + x[5:] for x in a] == [x[:4] + x[5:] for x in b] and len(a) == 21
    assert max(abs(float(x[4]) - float(y[4])) for x, y in zip(a, b)) < 1e-6
    calls = [json.loads(next(l for l in o.splitlines() if l.startswith("CALLS "))[6:]) for o in outs]
    if shard_flag == "1":
        # 7 candidates -> shares of 4 and 3 per query.  Round 6: run.py's default (--queries_per_call auto) hands the three queries to
        # rerank_many together and candidate sharding now GROUPS them: every rank's shares of all three queries in one launch
        # sequence (12 / 9 passages, two per engine call here: 6 / 5 calls), ONE gather for all of them
        assert all(c["init"] == 1 and c["gather"] == 1 for c in calls), calls
        assert [c["append"] for c in calls] == [6, 5], calls
    else:                       # replicas: no communicator, no gather
        assert all(c["init"] == 0 and c["gather"] == 0 for c in calls), calls


def test_resume_hygiene_cut_block_in_a_part_file_and_foreign_blocks(runmod, tmp_path):
    """Round-5 advisor finding: a kill in the middle of an append leaves a cut block in <save_path>.rank<N>; the restarted rank
    used to append the re-ranked query right behind its own fragment (the cut line swallowed the new block's first line), the
    merge then rejected the qid and the query was silently missing.  Now every part file (and --save_path) is rewritten without
    fragments BEFORE anybody appends, the last complete block of a qid counts, and complete-looking blocks that disagree with
    the expected length (another --hits) are refused instead of deleted."""
    expected = {"q1": 3, "q2": 3}
    blk = lambda q, tag="LLMRankers": [f"{q}\tQ0\td{i}\t{i + 1}\t{0.9 - 0.1 * i}\t{tag}\n" for i in range(3)]
    part = tmp_path / "run.trec.rank0"
    part.write_text("".join(blk("q2")) + "".join(blk("q1")[:1]) + blk("q1")[1][:11])   # q2 done, q1 cut in its second line
    kept, cut, mism = runmod.scan_run_blocks(str(part), expected)
    assert list(kept) == ["q2"] and cut == ["q1"] and mism == []
    # the old failure: append behind the fragment -> the joined line spoils the new block too
    spoiled = tmp_path / "spoiled.rank0"
    spoiled.write_text(part.read_text() + "".join(blk("q1")))
    assert "q1" not in runmod.complete_run_blocks(str(spoiled), expected)
    # the fix: clean first, then append
    assert list(runmod.clean_run_file(str(part), expected)) == ["q2"]
    assert part.read_text() == "".join(blk("q2")) and not (tmp_path / "run.trec.rank0.clean").exists()
    with open(part, "a") as f:
        f.writelines(blk("q1"))
    assert set(runmod.complete_run_blocks(str(part), expected)) == {"q1", "q2"}
    # a short block in the MIDDLE of a file with well-formed lines is not a fragment: other settings wrote it - refuse, keep the file
    foreign = tmp_path / "foreign.trec"
    foreign.write_text("".join(blk("q1")[:2]) + "".join(blk("q2")))
    before = foreign.read_text()
    with pytest.raises(ValueError, match="refusing"):
        runmod.clean_run_file(str(foreign), expected)
    assert foreign.read_text() == before
    # several complete blocks of one qid (an earlier run was merged twice): the last one counts
    twice = tmp_path / "twice.trec"
    twice.write_text("".join(blk("q1", "old")) + "".join(blk("q2")) + "".join(blk("q1", "new")))
    kept = runmod.complete_run_blocks(str(twice), expected)
    assert list(kept) == ["q2", "q1"] and kept["q1"][0].endswith("new\n")
    # a missing file is an empty one
    assert runmod.scan_run_blocks(str(tmp_path / "none"), expected) == ({}, [], [])


def test_two_rank_replicas_resume_is_durable(runmod, tmp_path, ckpt_dirs):
    """--resume under query replicas (--shard_candidates 0, 2 ranks): every rank appends its finished queries to
    <save_path>.rank<N> as it goes; a restart skips what ANY part file (or --save_path) already holds - here q3 sits in a part
    file a 'crashed' run left behind - and rank 0 merges everything into --save_path in first-stage order and removes the parts."""
    import json
    import socket
    import subprocess
    ck = ckpt_dirs["ckpt_gated_untied"]
    (tmp_path / "q.tsv").write_text("q1\tneural ranking model\nq2\twater river mountain\nq3\tmusic art film\n")
    (tmp_path / "d.tsv").write_text("\n".join(f"d{i}\t{w}" for i, w in enumerate(
        ["search engine index", "river water city", "music art film", "vaccine covid virus", "bank money trade",
         "neural model answer", "mountain river water", "film music topic"])) + "\n")
    lines = [f"{q} Q0 d{i} {r + 1} {10 - r} bm25" for q in ("q1", "q2", "q3") for r, i in enumerate([0, 1, 2, 3, 4, 5, 6])]
    (tmp_path / "in.trec").write_text("\n".join(lines) + "\n")
    (tmp_path / "worker.py").write_text(RUN_WORKER)

    def argv(save, extra=()):
        return ["run", "--model_name_or_path", ck, "--run_path", str(tmp_path / "in.trec"), "--save_path", str(save),
                "--query_file", str(tmp_path / "q.tsv"), "--doc_file", str(tmp_path / "d.tsv"), "--hits", "7",
                "--shard_candidates", "0", "--queries_per_call", "1", *extra, "pointwise", "--method", "yes_no", "--batch_size", "3"]

    base_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    single = subprocess.run([sys.executable, str(tmp_path / "worker.py"), REPO, ck, json.dumps(argv(tmp_path / "single.trec"))],
                            capture_output=True, text=True, env=dict(base_env, OMP_NUM_THREADS="2"), timeout=600)
    assert single.returncode == 0, single.stderr[-2000:]
    want = (tmp_path / "single.trec").read_text()
    # what a killed earlier run left: q3 finished by (then) rank 1, nothing merged yet
    (tmp_path / "multi.trec.rank1").write_text("".join(l + "\n" for l in want.splitlines() if l.startswith("q3\t")))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, str(tmp_path / "worker.py"), REPO, ck, json.dumps(argv(tmp_path / "multi.trec", ["--resume"]))],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(base_env, OMP_NUM_THREADS="2", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))) for r in range(2)]
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        outs.append(out)
    a = [l.split("\t") for l in want.splitlines()]
    b = [l.split("\t") for l in (tmp_path / "multi.trec").read_text().splitlines()]
    assert [x[:4] + x[5:] for x in a] == [x[:4] + x[5:] for x in b] and len(b) == 21
    assert not list(tmp_path.glob("multi.trec.rank*"))                           # merged and removed
    calls = [json.loads(next(l for l in o.splitlines() if l.startswith("CALLS "))[6:]) for o in outs]
    # shards: rank 0 = (q1, q2), rank 1 = (q3,): q3 was done, so rank 1 scored nothing; rank 0 scored 2 queries x 4 engine calls
    assert calls[1]["score"] == 0 and calls[0]["score"] == 8, calls
    # a --save_path written with other settings (a complete-looking 2-line block in the middle): rank 0 refuses to clean it and
    # EVERY rank stops with that message - nobody is left waiting at a barrier (the refusal travels in the broadcast)
    foreign = tmp_path / "foreign.trec"
    fl = want.splitlines()
    foreign.write_text("".join(l + "\n" for l in fl[:2]) + "".join(l + "\n" for l in fl[7:14]))
    before = foreign.read_text()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, str(tmp_path / "worker.py"), REPO, ck, json.dumps(argv(foreign, ["--resume"]))],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(base_env, OMP_NUM_THREADS="2", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode != 0 and "refusing" in err, err[-1500:]
    assert foreign.read_text() == before
