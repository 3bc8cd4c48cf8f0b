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
