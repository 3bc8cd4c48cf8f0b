#!/usr/bin/env python
"""Timeline of ONE setwise compare from a rocprofv3 kernel trace: per kernel name the launches, the average duration and the
average gap to the previous kernel's end (run: rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/compare_trace.py run;
then python tools/compare_trace.py summarize DIR)."""
import csv, glob, json, os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "run":
    sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd"), REPO]
    import torch  # noqa
    from llmrankers import _synth
    from llmrankers._engine import RkEngine
    L, B = int(os.environ.get("RK_L", "1450")), int(os.environ.get("RK_B", "1"))
    dims = _synth.FLAN_T5_LARGE
    eng = RkEngine(dims, 0, max_tokens=32768, max_seqs=16, max_dec_len=8).load_state(_synth.synth_tensors(dims, seed=929, threads=32))
    for kv in os.environ.get("RK_OPTS", "").split(","):
        if kv:
            eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    seqs = _synth.synth_token_batch(B, L, L, dims.vocab, seed=7)
    for _ in range(6):
        eng.score(seqs, [0, 5454], list(range(71, 82)))
    eng.sync()
else:
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last call = the kernels after the last embed_gather pair ... simply take the last 1/6 of the rows
    n = len(rows) // 6
    rows = rows[-n:]
    agg = collections.OrderedDict()
    prev_end = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = r["Kernel_Name"][:60]
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e3
        if prev_end is not None:
            a[2] += max(0, s - prev_end) / 1e3
        prev_end = e
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    print(f"kernels {len(rows)} span {span:.1f} us busy {sum(a[1] for a in agg.values()):.1f} us gaps {sum(a[2] for a in agg.values()):.1f} us")
    for k, a in agg.items():
        print(f"{a[0]:4d} x {a[1]/a[0]:7.2f} us  gap before {a[2]/a[0]:6.2f} us  total {a[1]+a[2]:8.1f}  {k}")
