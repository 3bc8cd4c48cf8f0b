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

Multi-GPU (one process per GPU; the reference's only multi-GPU mode is accelerate's device_map='auto' layer placement,
ref: llmrankers/pointwise.py:20-24, README.md:357): `--num_gpus N` re-executes the command under torch.distributed.run with N
ranks on this node (or run it under torchrun yourself).  Pointwise rankers then shard every query's CANDIDATES over the ranks
(hits=100 over 8 GPUs -> 13,13,13,13,12,12,12,12) and the engine collects the scores with one RCCL all_gather per query
(`--shard_candidates`, the default for pointwise under more than one rank); setwise / pairwise sorts are dependency chains, so
their QUERIES are dealt to the ranks instead (replicas) and rank 0 collects the rankings.  Rank 0 writes the run file and
prints the averages.

Long runs (the reference's Rank-R1 driver, ref: Rank-R1/run_setwise.py:79-87, 266-300): `--resume` appends each query's
ranking to --save_path as soon as it is done and skips the qids already in the file when restarted;
`--dataset_number_of_shards` / `--dataset_shard_index` cut the query list into contiguous shards (one process per GPU,
replicas only - the right multi-GPU mode for setwise).  `--qrels` prints NDCG@10 of the input run and of the reranked
run with a self-contained NDCG (trec_eval / pyserini are not available offline).
"""
import argparse
import json
import logging
import os
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


def write_run_file(path, results, tag, mode="w"):
    # ref: run.py:41-49; mode "a" = the append-per-query form of ref: Rank-R1/run_setwise.py:79-87
    with open(path, mode) as f:
        for qid, _, ranking in results:
            for rank, doc in enumerate(ranking, start=1):
                f.write(f"{qid}\tQ0\t{doc.docid}\t{rank}\t{doc.score}\t{tag}\n")


def part_files(save_path):
    """Per-rank part files of a --resume run in query-replica mode (`<save_path>.rank<N>`, N decimal), any world size.
    Anything else that happens to match the prefix (`.rank0.bak`, editor backups) is not a part file."""
    import glob
    import re
    pat = re.compile(re.escape(save_path) + r"\.rank[0-9]+$")
    return sorted(f for f in glob.glob(glob.escape(save_path) + ".rank*") if pat.match(f))


def scan_run_blocks(path, expected):
    """A (partial) run file as blocks of consecutive lines of one qid, classified.  Returns (kept, cut, mismatched):
    kept {qid: [raw lines]} - COMPLETE blocks: exactly expected[qid] well-formed lines (six fields), every one newline-terminated
    (qids that `expected` does not know - another query set - are kept as they are); if a qid has several complete blocks the last
    one counts.  cut [qid] - what a process killed in the middle of an append leaves: a malformed / unterminated line anywhere in
    the block, or a short block at the very end of the file.  mismatched [(qid, lines, expected)] - well-formed blocks whose
    length disagrees with `expected` (a run made with another --hits or another first-stage run): never deleted silently."""
    blocks = []
    try:
        with open(path) as f:
            for line in f:
                parts = line.split()
                if not parts:
                    continue
                if blocks and blocks[-1][0] == parts[0]:
                    blocks[-1][1].append(line)
                else:
                    blocks.append((parts[0], [line]))
    except FileNotFoundError:
        pass
    kept, cut, mismatched = {}, [], []
    for i, (qid, lines) in enumerate(blocks):
        well_formed = all(l.endswith("\n") and len(l.split()) == 6 for l in lines)
        if well_formed and (qid not in expected or len(lines) == expected[qid]):
            kept.pop(qid, None)                                   # (file order of the block that counts)
            kept[qid] = lines
        elif not well_formed or (i == len(blocks) - 1 and len(lines) < expected[qid]):
            cut.append(qid)
        else:
            mismatched.append((qid, len(lines), expected[qid]))
    return kept, cut, mismatched


def complete_run_blocks(path, expected):
    """{qid: [raw lines]} of the queries whose block in a (partial) run file is COMPLETE (scan_run_blocks).  A query with only a
    cut block is not done - it is ranked again and its fragment is dropped (never merged)."""
    return scan_run_blocks(path, expected)[0]


def clean_run_file(path, expected):
    """Resume hygiene: rewrite `path` atomically without the fragments a killed append left behind, so that a re-ranked query is
    never appended behind its own fragment (a cut last line would swallow the first line of the new block).  Complete-looking
    blocks that disagree with `expected` are NOT deleted: the run was made with other settings - refuse.  Returns the kept blocks."""
    kept, cut, mismatched = scan_run_blocks(path, expected)
    if mismatched:
        qid, have, want = mismatched[0]
        raise ValueError(f"--resume: {path} holds {have} lines for query {qid} but this run expects {want} (another --hits or first-stage "
                         f"run?); {len(mismatched)} such queries - refusing to delete them, move the file away or match the settings")
    if cut:
        tmp = path + ".clean"
        with open(tmp, "w") as f:
            for lines in kept.values():
                f.writelines(lines)
        os.replace(tmp, path)
    return kept


def _auto_per_call(kind, hits):
    from llmrankers._batching import default_queries_per_call
    return default_queries_per_call(kind, hits)


def split_into_shards(data, num_shards):
    """Contiguous shards whose sizes differ by at most one (ref: Rank-R1/run_setwise.py:90-92)."""
    base, extra = divmod(len(data), num_shards)
    out, s = [], 0
    for i in range(num_shards):
        e = s + base + (1 if i < extra else 0)
        out.append(data[s:e])
        s = e
    return out


def read_qrels(path):
    """TREC qrels: `qid iter docid rel` (or 3 columns) -> {qid: {docid: int rel}}."""
    out = {}
    with open(path) as f:
        for line in f:
            parts = line.split()
            if len(parts) >= 4:
                qid, docid, rel = parts[0], parts[2], parts[3]
            elif len(parts) == 3:
                qid, docid, rel = parts
            else:
                continue
            out.setdefault(qid, {})[docid] = int(float(rel))
    return out


def ndcg_at_k(ranked_docids, rels, k=10):
    """NDCG@k with trec_eval's ndcg_cut convention: gain = rel (linear), discount = 1 / log2(rank + 1), ideal ranking from
    the judged documents of the query; 0 for a query without relevant documents."""
    import math
    dcg = sum(max(rels.get(d, 0), 0) / math.log2(i + 2) for i, d in enumerate(ranked_docids[:k]))
    ideal = sorted((r for r in rels.values() if r > 0), reverse=True)[:k]
    idcg = sum(r / math.log2(i + 2) for i, r in enumerate(ideal))
    return dcg / idcg if idcg > 0 else 0.0


def mean_ndcg(run, qrels, k=10):
    """run: {qid: [docid, ...] best first}.  Mean over the queries that have qrels (trec_eval -c off)."""
    vals = [ndcg_at_k(docs, qrels[q], k) for q, docs in run.items() if q in qrels]
    return sum(vals) / len(vals) if vals else 0.0


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


def dist_env():
    """(rank, world) of the launcher's environment (torchrun / torch.distributed.run), (0, 1) without one."""
    return int(os.environ.get("RANK", "0") or 0), int(os.environ.get("WORLD_SIZE", "1") or 1)


def self_spawn(num_gpus, argv=None):
    """`run.py run --num_gpus N ...` without a launcher: start the N ranks ourselves (one process per GPU, rendezvous on
    127.0.0.1), exactly like bench.py --gpus N."""
    if num_gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={num_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv))
    print("[run] --num_gpus %d without a launcher: re-executing as\n      %s" % (num_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    os.execv(sys.executable, cmd)


def candidates_sharded(args, world):
    """Pointwise under more than one rank shards the candidates of every query (one RCCL gather per query) unless
    --shard_candidates 0 asks for query-level replicas; the sorting rankers are always replicas."""
    flag = getattr(args.run, "shard_candidates", None)
    return bool(args.pointwise) and world > 1 and (flag is None or int(flag) != 0)


def build_ranker(args):
    if args.pointwise:
        from llmrankers.pointwise import MonoT5LlmRanker, PointwiseLlmRanker
        cls = MonoT5LlmRanker if "monot5" in args.run.model_name_or_path else PointwiseLlmRanker
        return cls(model_name_or_path=args.run.model_name_or_path, tokenizer_name_or_path=args.run.tokenizer_name_or_path,
                   device=args.run.device, cache_dir=args.run.cache_dir, method=args.pointwise.method,
                   batch_size=args.pointwise.batch_size, shard_candidates=candidates_sharded(args, dist_env()[1]))
    if args.setwise:
        if args.run.openai_key:
            raise NotImplementedError("OpenAI rankers are remote HTTP calls, not part of the MI355X hot path; use the reference")
        from llmrankers.setwise import SetwiseLlmRanker
        return SetwiseLlmRanker(model_name_or_path=args.run.model_name_or_path, tokenizer_name_or_path=args.run.tokenizer_name_or_path,
                                device=args.run.device, cache_dir=args.run.cache_dir, num_child=args.setwise.num_child,
                                scoring=args.run.scoring, method=args.setwise.method,
                                num_permutation=args.setwise.num_permutation, k=args.setwise.k)
    if args.pairwise:
        if args.pairwise.method != "allpair":                    # ref: run.py:88-90
            args.pairwise.batch_size = 2
            logger.info("Setting batch_size to 2.")
        if args.run.openai_key or "duot5" in args.run.model_name_or_path:
            raise NotImplementedError("OpenAI / duoT5 pairwise rankers are not part of this build; use the reference")
        from llmrankers.pairwise import PairwiseLlmRanker
        return PairwiseLlmRanker(model_name_or_path=args.run.model_name_or_path, tokenizer_name_or_path=args.run.tokenizer_name_or_path,
                                 device=args.run.device, cache_dir=args.run.cache_dir, method=args.pairwise.method,
                                 batch_size=args.pairwise.batch_size, k=args.pairwise.k)
    if args.listwise:
        raise NotImplementedError("listwise rankers are outside the path this engine accelerates (DESIGN.md); use the reference")
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
    rank, world = dist_env()
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():                     # host-side control plane only; scores travel over the engine's RCCL
            dist.init_process_group(backend="gloo")
    ranker = build_ranker(args)
    query_map, get_doc = load_queries_and_docs(args, ranker)
    logger.info(f"Loading first stage run from {args.run.run_path}.")
    first_stage, cur_qid, cur = [], None, []
    # docid-keyed cache of the tokenise -> cut -> detokenise round trip (ref: pointwise.py:132-133, run.py:174): the same
    # passage is retrieved for many queries, its truncated text is a pure function of (text, passage_length)
    truncated = {}

    def passage(docid):
        if docid not in truncated:
            truncated[docid] = ranker.truncate(get_doc(docid), args.run.passage_length)
        return truncated[docid]

    with open(args.run.run_path) as f:
        for line in f:
            qid, _, docid, _, score, _ = line.strip().split()      # 6 whitespace-separated fields (ref: run.py:157)
            if qid != cur_qid:
                if cur_qid is not None:
                    first_stage.append((cur_qid, query_map[cur_qid], cur[:args.run.hits]))
                cur, cur_qid = [], qid
            if len(cur) >= args.run.hits:
                continue
            cur.append(SearchResult(docid=docid, score=float(score), text=passage(docid)))
        if cur_qid is not None:
            first_stage.append((cur_qid, query_map[cur_qid], cur[:args.run.hits]))

    if getattr(args.run, "dataset_number_of_shards", 1) > 1:        # one process per GPU, each takes a contiguous shard of the queries
        first_stage = split_into_shards(first_stage, args.run.dataset_number_of_shards)[args.run.dataset_shard_index]
    shard_cands = candidates_sharded(args, world)
    replicas = world > 1 and not shard_cands                         # ranks take whole queries; rank 0 collects the rankings
    all_qids = [qid for qid, _, _ in first_stage]
    all_shards = None
    if replicas:
        all_shards = split_into_shards(first_stage, world)
        first_stage = all_shards[rank]
    writer = rank == 0                                               # candidate sharding: every rank holds every ranking
    resume = bool(getattr(args.run, "resume", False))
    # a query is done only if its block is complete (as many lines as it has candidates, the last one newline-terminated)
    expected = {qid: len(ranking) for qid, _, ranking in (item for shard in (all_shards or [first_stage]) for item in shard)}
    done = set()
    # query replicas under --resume: every rank appends ITS finished queries to <save_path>.rank<N> after every call (durable
    # like the single-process form, ref: Rank-R1/run_setwise.py:79-87); a restart - with any number of ranks - skips what any
    # part file holds, and rank 0 merges the parts into --save_path at the end
    my_part = f"{args.run.save_path}.rank{rank}" if (resume and replicas) else None
    if resume:
        # a cut block (killed mid-append) in --save_path or in ANY part file is removed before anybody appends: rank 0 rewrites
        # the files, the others wait for it
        refusal = None
        if writer:
            try:
                clean_run_file(args.run.save_path, expected)
                if replicas:
                    for pf in part_files(args.run.save_path):
                        clean_run_file(pf, expected)
            except ValueError as exc:                                # foreign blocks: every rank must stop, not only rank 0
                refusal = str(exc)
        if world > 1:
            import torch.distributed as dist
            box = [refusal]
            dist.broadcast_object_list(box, src=0)                   # (also the barrier: nobody appends before rank 0 is done)
            refusal = box[0]
        if refusal:
            raise ValueError(refusal)
        done = set(complete_run_blocks(args.run.save_path, expected))
        if replicas:
            for pf in part_files(args.run.save_path):
                done.update(complete_run_blocks(pf, expected))
    if done:
        print(f"{args.run.save_path} exists. Continue ranking ({len(done)} queries done)")
    first_stage_order = {qid: [d.docid for d in ranking] for qid, _, ranking in first_stage}

    results, n_cmp, n_prompt, n_compl = [], 0, 0, 0
    # --queries_per_call N: N queries go to the engine together (PointwiseLlmRanker.rerank_many: all their batches in one launch
    # sequence; SetwiseLlmRanker.rerank_many: their heapsorts advance in lockstep, one engine call per step of all the chains) -
    # same rankings and counters as one query at a time, the engine's batched throughput instead of its per-query one
    per_call = int(getattr(args.run, "queries_per_call", 0) or 0)
    if per_call <= 0:                                                # auto: the engine's grouped throughput by default
        from llmrankers._batching import default_queries_per_call
        per_call = default_queries_per_call("pointwise" if args.pointwise else ("setwise" if args.setwise else "other"), args.run.hits)
    if per_call > 1 and not hasattr(ranker, "rerank_many"):
        per_call = 1
    if per_call > 1 and args.run.shuffle_ranking == "random" and getattr(ranker, "num_permutation", 1) > 1:
        # permutation voting draws from the same global RNG as --shuffle_ranking random: queued queries would be shuffled
        # before earlier ones have drawn their permutations - a different random sequence than one query at a time
        per_call = 1
    pending = []

    def flush():
        nonlocal n_cmp, n_prompt, n_compl
        if not pending:
            return
        if len(pending) == 1:
            qid, query, ranking = pending[0]
            ranked = [ranker.rerank(query, ranking)]
            counters = [(ranker.total_compare, ranker.total_prompt_tokens, ranker.total_completion_tokens)]
        else:
            ranked, counters = ranker.rerank_many([(query, ranking) for _, query, ranking in pending])
        for (qid, query, _), res, (c, p, t) in zip(pending, ranked, counters):
            results.append((qid, query, res))
            n_cmp += c
            n_prompt += p
            n_compl += t
        if resume and my_part:                                       # durable after every call, one part file per rank
            write_run_file(my_part, results[-len(pending):], "LLMRankers", mode="a")
        elif resume and writer:                                      # (ref: Rank-R1/run_setwise.py:79-87)
            write_run_file(args.run.save_path, results[-len(pending):], "LLMRankers", mode="a")
        pending.clear()

    # --shuffle_ranking random draws from ONE global sequence (random.seed(929), ref: run.py:16,187): under query replicas every
    # rank walks ALL queries in first-stage order and shuffles each of them, so that its own queries see exactly the draws a
    # single process would have made for them.  (Rankers that draw random numbers themselves - setwise permutation voting -
    # cannot be advanced past the queries of other ranks: their multi-rank runs differ from a single process, and say so.)
    mine = {qid for qid, _, _ in first_stage}
    walk = first_stage
    if replicas and args.run.shuffle_ranking == "random":
        walk = [item for shard in all_shards for item in shard]
        if getattr(ranker, "num_permutation", 1) > 1 and rank == 0:
            print("[run] note: permutation voting draws from the global RNG inside rerank(); with --num_gpus > 1 the draws of a "
                  "query differ from a single-process run", file=sys.stderr)
    tic = time.time()
    for qid, query, ranking in walk:
        if qid in done:
            continue
        if args.run.shuffle_ranking == "random":
            random.shuffle(ranking)
        elif args.run.shuffle_ranking == "inverse":
            ranking = ranking[::-1]
        if qid not in mine:
            continue
        pending.append((qid, query, ranking))
        if len(pending) >= per_call:
            flush()
    flush()
    toc = time.time()
    if replicas:
        # every rank ranked its share of the queries: rank 0 collects (rankings are small: docid + score per line)
        import torch.distributed as dist
        mine = ([(qid, query, [(d.docid, d.score) for d in res]) for qid, query, res in results], n_cmp, n_prompt, n_compl, toc - tic)
        parts = [None] * world if rank == 0 else None
        dist.gather_object(mine, parts, dst=0)
        if rank == 0:
            by_qid, n_cmp, n_prompt, n_compl, wall = {}, 0, 0, 0, 0.0
            for rows, c, p_, t_, w in parts:
                for qid, query, docs in rows:
                    by_qid[qid] = (qid, query, [SearchResult(docid=d, score=sc, text=None) for d, sc in docs])
                n_cmp, n_prompt, n_compl, wall = n_cmp + c, n_prompt + p_, n_compl + t_, max(wall, w)
            results = [by_qid[q] for q in all_qids if q in by_qid]          # first-stage order, like a single process
            tic, toc = 0.0, wall
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if not writer:
        return
    n = max(len(results), 1)
    print(f"Avg comparisons: {n_cmp / n}")
    print(f"Avg prompt tokens: {n_prompt / n}")
    print(f"Avg completion tokens: {n_compl / n}")
    print(f"Avg time per query: {(toc - tic) / n}")
    if replicas and resume:
        # merge: what --save_path already held, then every rank's part file (all flushed: the barrier above), in first-stage order
        merged = complete_run_blocks(args.run.save_path, expected)
        parts_now = part_files(args.run.save_path)
        for pf in parts_now:
            for qid, lines in complete_run_blocks(pf, expected).items():
                merged.setdefault(qid, lines)
        qset = set(all_qids)
        order = [q for q in merged if q not in qset] + [q for q in all_qids if q in merged]
        tmp = args.run.save_path + ".merge"
        with open(tmp, "w") as f:
            for q in order:
                f.writelines(merged[q])
        os.replace(tmp, args.run.save_path)
        for pf in parts_now:
            os.remove(pf)
    elif not resume:
        write_run_file(args.run.save_path, results, "LLMRankers")
    if getattr(args.run, "qrels", None):
        qrels = read_qrels(args.run.qrels)
        reranked = {}
        with open(args.run.save_path) as f:
            for line in f:
                parts = line.split()
                reranked.setdefault(parts[0], []).append(parts[2])
        print(f"NDCG@10 first stage: {mean_ndcg(first_stage_order, qrels):.4f}")
        print(f"NDCG@10 reranked: {mean_ndcg(reranked, qrels):.4f}")


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
    rp.add_argument("--resume", action="store_true", help="append every query's ranking to --save_path at once and skip qids already there")
    rp.add_argument("--dataset_number_of_shards", type=int, default=1)
    rp.add_argument("--dataset_shard_index", type=int, default=0)
    rp.add_argument("--qrels", type=str, default=None, help="TREC qrels file: print NDCG@10 of the input and the reranked run")
    rp.add_argument("--num_gpus", type=int, default=1,
                    help="ranks (one process per GPU) on this node; > 1 without a launcher re-executes under torch.distributed.run")
    rp.add_argument("--shard_candidates", type=int, default=None, choices=[0, 1],
                    help="pointwise under several ranks: 1 (default) shards every query's candidates and gathers the scores over "
                         "RCCL, 0 deals whole queries to the ranks")
    rp.add_argument("--queries_per_call", type=int, default=0,
                    help="pointwise / setwise: queries handed to the engine together (same rankings and counters as one at a time); "
                         "0 = auto (llmrankers._batching.default_queries_per_call): pointwise enough queries for >= 1600 passages "
                         f"per call, at most 16 ({_auto_per_call('pointwise', 100)} at hits=100), setwise {_auto_per_call('setwise', 100)} "
                         "heapsorts in lockstep; with --resume the run file is appended once per call, i.e. every that many queries; "
                         "1 = the reference's one query at a time (and its per-query flush)")
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
    self_spawn(int(getattr(_args.run, "num_gpus", 1) or 1))
    main(_args)
