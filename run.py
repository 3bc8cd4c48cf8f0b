#!/usr/bin/env python
"""run.py — experiment driver with the reference's command line (ref: run.py:206-259), on the MI355X engine.

    python run.py run --model_name_or_path <ckpt dir> --run_path first_stage.trec --save_path out.trec \
                      --ir_dataset_name msmarco-passage/trec-dl-2019 --hits 100 --query_length 32 --passage_length 128 \
           pointwise --method yes_no --batch_size 32

Same sub-commands (`run` + one of `pointwise` / `setwise`), flags, defaults, TREC run input/output and the four
averages printed at the end (ref: run.py:198-201).  pairwise / listwise / OpenAI rankers are outside the hot
path this build accelerates and are rejected with a clear message.  The data back-ends
(ir_datasets / pyserini) are imported lazily; because neither exists offline, two plain-file sources are
accepted as well:  --query_file (TSV `qid<TAB>text` or JSONL {"qid"|"query_id"|"_id", "text"|"query"}) and
--doc_file (TSV `docid<TAB>text` or JSONL {"docid"|"doc_id"|"_id", "text"|"contents", ["title"]}).
"""
import argparse
import json
import logging
import random
import sys
import time

random.seed(929)          # ref: run.py:16 (governs --shuffle_ranking random and setwise permutation voting)
logger = logging.getLogger(__name__)

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "llm-rankers_amd"))
from llmrankers.rankers import SearchResult            # noqa: E402


def parse_args(parser, commands, argv=None):
    """Split argv at sub-command names so `run ... pointwise ...` fills two namespaces (ref: run.py:20-38)."""
    argv = sys.argv[1:] if argv is None else argv
    groups = [[]]
    for tok in argv:
        if tok in commands.choices:
            groups.append([tok])
        else:
            groups[-1].append(tok)
    args = argparse.Namespace(**{c: None for c in commands.choices})
    parser.parse_args(groups[0], namespace=args)
    for g in groups[1:]:
        ns = argparse.Namespace()
        setattr(args, g[0], ns)
        parser.parse_args(g, namespace=ns)
    return args


def write_run_file(path, results, tag):
    # ref: run.py:41-49
    with open(path, "w") as f:
        for qid, _, ranking in results:
            for rank, doc in enumerate(ranking, start=1):
                f.write(f"{qid}\tQ0\t{doc.docid}\t{rank}\t{doc.score}\t{tag}\n")


def _read_kv_file(path, id_keys, text_keys):
    out = {}
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            if line.lstrip().startswith("{"):
                rec = json.loads(line)
                key = next(str(rec[k]) for k in id_keys if k in rec)
                text = next(rec[k] for k in text_keys if k in rec)
                if rec.get("title"):
                    text = f'{rec["title"]} {text}'
            else:
                key, text = line.split("\t", 1)
            out[key] = text
    return out


def build_ranker(args):
    if args.pointwise:
        from llmrankers.pointwise import MonoT5LlmRanker, PointwiseLlmRanker
        cls = MonoT5LlmRanker if "monot5" in args.run.model_name_or_path else PointwiseLlmRanker
        return cls(model_name_or_path=args.run.model_name_or_path, tokenizer_name_or_path=args.run.tokenizer_name_or_path,
                   device=args.run.device, cache_dir=args.run.cache_dir, method=args.pointwise.method,
                   batch_size=args.pointwise.batch_size)
    if args.setwise:
        if args.run.openai_key:
            raise NotImplementedError("OpenAI rankers are remote HTTP calls, not part of the MI355X hot path; use the reference")
        from llmrankers.setwise import SetwiseLlmRanker
        return SetwiseLlmRanker(model_name_or_path=args.run.model_name_or_path, tokenizer_name_or_path=args.run.tokenizer_name_or_path,
                                device=args.run.device, cache_dir=args.run.cache_dir, num_child=args.setwise.num_child,
                                scoring=args.run.scoring, method=args.setwise.method,
                                num_permutation=args.setwise.num_permutation, k=args.setwise.k)
    if args.pairwise or args.listwise:
        raise NotImplementedError("pairwise / listwise rankers are outside the path this engine accelerates (DESIGN.md); use the reference")
    raise ValueError("Must specify either --pointwise, --setwise, --pairwise or --listwise.")


def load_queries_and_docs(args, ranker):
    r = args.run
    if r.query_file or r.doc_file:
        if not (r.query_file and r.doc_file):
            raise ValueError("--query_file and --doc_file must be given together")
        queries = _read_kv_file(r.query_file, ("qid", "query_id", "_id", "id"), ("text", "query", "title"))
        docs = _read_kv_file(r.doc_file, ("docid", "doc_id", "_id", "id"), ("text", "contents"))
        return {q: ranker.truncate(t, r.query_length) for q, t in queries.items()}, docs.__getitem__
    if r.ir_dataset_name is not None:
        import ir_datasets                                   # lazy: absent offline
        dataset = ir_datasets.load(r.ir_dataset_name)
        query_map = {q.query_id: ranker.truncate(q.text, r.query_length) for q in dataset.queries_iter()}
        store = dataset.docs_store()

        def get(docid):
            d = store.get(docid)
            return f"{d.title} {d.text}" if "title" in dir(d) else d.text
        return query_map, get
    from pyserini.search.lucene import LuceneSearcher         # lazy: absent offline
    from pyserini.search._base import get_topics
    topics = get_topics(r.pyserini_index + "-test")
    query_map = {str(t): ranker.truncate(topics[t]["title"], r.query_length) for t in topics}
    searcher = LuceneSearcher.from_prebuilt_index(r.pyserini_index + ".flat")

    def get(docid):
        data = json.loads(searcher.doc(docid).raw())
        return f'{data["title"]} {data["text"]}' if "title" in data else data["text"]
    return query_map, get


def main(args):
    ranker = build_ranker(args)
    query_map, get_doc = load_queries_and_docs(args, ranker)
    logger.info(f"Loading first stage run from {args.run.run_path}.")
    first_stage, cur_qid, cur = [], None, []
    with open(args.run.run_path) as f:
        for line in f:
            qid, _, docid, _, score, _ = line.strip().split()      # 6 whitespace-separated fields (ref: run.py:157)
            if qid != cur_qid:
                if cur_qid is not None:
                    first_stage.append((cur_qid, query_map[cur_qid], cur[:args.run.hits]))
                cur, cur_qid = [], qid
            if len(cur) >= args.run.hits:
                continue
            cur.append(SearchResult(docid=docid, score=float(score), text=ranker.truncate(get_doc(docid), args.run.passage_length)))
        if cur_qid is not None:
            first_stage.append((cur_qid, query_map[cur_qid], cur[:args.run.hits]))

    results, n_cmp, n_prompt, n_compl = [], 0, 0, 0
    tic = time.time()
    for qid, query, ranking in first_stage:
        if args.run.shuffle_ranking == "random":
            random.shuffle(ranking)
        elif args.run.shuffle_ranking == "inverse":
            ranking = ranking[::-1]
        results.append((qid, query, ranker.rerank(query, ranking)))
        n_cmp += ranker.total_compare
        n_prompt += ranker.total_prompt_tokens
        n_compl += ranker.total_completion_tokens
    toc = time.time()
    n = max(len(results), 1)
    print(f"Avg comparisons: {n_cmp / n}")
    print(f"Avg prompt tokens: {n_prompt / n}")
    print(f"Avg completion tokens: {n_compl / n}")
    print(f"Avg time per query: {(toc - tic) / n}")
    write_run_file(args.run.save_path, results, "LLMRankers")


def build_parser():
    parser = argparse.ArgumentParser()
    commands = parser.add_subparsers(title="sub-commands")
    rp = commands.add_parser("run")
    rp.add_argument("--run_path", type=str, help="Path to the first stage run file (TREC format) to rerank.")
    rp.add_argument("--save_path", type=str, help="Path to save the reranked run file (TREC format).")
    rp.add_argument("--model_name_or_path", type=str, help="Local HuggingFace-layout checkpoint directory.")
    rp.add_argument("--tokenizer_name_or_path", type=str, default=None)
    rp.add_argument("--ir_dataset_name", type=str, default=None)
    rp.add_argument("--pyserini_index", type=str, default=None)
    rp.add_argument("--query_file", type=str, default=None, help="offline source: TSV/JSONL of queries")
    rp.add_argument("--doc_file", type=str, default=None, help="offline source: TSV/JSONL of documents")
    rp.add_argument("--hits", type=int, default=100)
    rp.add_argument("--query_length", type=int, default=128)
    rp.add_argument("--passage_length", type=int, default=128)
    rp.add_argument("--device", type=str, default="cuda")
    rp.add_argument("--cache_dir", type=str, default=None)
    rp.add_argument("--openai_key", type=str, default=None)
    rp.add_argument("--scoring", type=str, default="generation", choices=["generation", "likelihood"])
    rp.add_argument("--shuffle_ranking", type=str, default=None, choices=["inverse", "random"])
    pw = commands.add_parser("pointwise")
    pw.add_argument("--method", type=str, default="yes_no", choices=["qlm", "yes_no"])
    pw.add_argument("--batch_size", type=int, default=2)
    pa = commands.add_parser("pairwise")
    pa.add_argument("--method", type=str, default="allpair", choices=["allpair", "heapsort", "bubblesort"])
    pa.add_argument("--batch_size", type=int, default=2)
    pa.add_argument("--k", type=int, default=10)
    sw = commands.add_parser("setwise")
    sw.add_argument("--num_child", type=int, default=3)
    sw.add_argument("--method", type=str, default="heapsort", choices=["heapsort", "bubblesort"])
    sw.add_argument("--k", type=int, default=10)
    sw.add_argument("--num_permutation", type=int, default=1)
    lw = commands.add_parser("listwise")
    lw.add_argument("--window_size", type=int, default=3)
    lw.add_argument("--step_size", type=int, default=1)
    lw.add_argument("--num_repeat", type=int, default=1)
    return parser, commands


def validate(args):
    # ref: run.py:253-258
    if args.run is not None and args.run.ir_dataset_name is not None and args.run.pyserini_index is not None:
        raise ValueError("Must specify either --ir_dataset_name or --pyserini_index, not both.")
    d = vars(args)
    if d["run"] is None or sum(d[a] is not None for a in d) != 2:
        raise ValueError("Need to set --run and can only set one of --pointwise, --pairwise, --setwise, --listwise")


if __name__ == "__main__":
    _parser, _commands = build_parser()
    _args = parse_args(_parser, _commands)
    validate(_args)
    main(_args)
