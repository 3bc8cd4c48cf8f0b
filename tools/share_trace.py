#!/usr/bin/env python
"""Kernel timeline of one rank's 13-passage share (bench.py shard_share_leg) from a rocprofv3 kernel trace: run under
`rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/share_trace.py run`, then `... summarize DIR`:
per kernel name launches, average duration, and per stream (queue) the busy time of the last steps."""
import csv, glob, json, os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "run":
    sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
    import torch  # noqa
    import bench
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    dims = _synth.FLAN_T5_LARGE
    eng = RkEngine(dims, device=0, max_tokens=32 * 184, max_seqs=32, max_dec_len=4)
    eng.load_state(_synth.synth_state_dict(dims, seed=929, threads=32).items())
    for kv in os.environ.get("RK_OPTS", "").split(","):
        if kv:
            eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    print(json.dumps(bench.shard_share_leg(eng, dims, 184, steps=16)))
else:
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-len(rows) // 3:]                                # the last third: steady state
    t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
    print(f"window {(t1 - t0) / 1e3:.1f} us, {len(rows)} kernels")
    by_q = collections.defaultdict(list)
    for r in rows:
        by_q[r.get("Queue_Id", "?")].append(r)
    for q, rs in by_q.items():
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
        print(f"queue {q}: {len(rs)} kernels, busy {busy:.1f} us ({busy / ((t1 - t0) / 1e3) * 100:.0f} % of the window)")
        agg = collections.OrderedDict()
        for r in rs:
            a = agg.setdefault(r["Kernel_Name"][:70], [0, 0.0])
            a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"    {a[0]:5d} x {a[1] / a[0]:7.2f} us = {a[1]:8.1f}  {k}")
