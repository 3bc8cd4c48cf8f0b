#!/bin/bash
# step time vs launch-group size (batches of 32 passages per engine launch sequence)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for g in ${GROUPS_LIST:-4 5 8}; do
  timeout 300 python bench.py --group $g --steps $((g*16)) --warmup $((g*2)) --no_cpu_baseline ${BENCH_EXTRA} > gpurun_out/bench_g$g.json 2> gpurun_out/bench_g$g.err
  echo "group=$g rc=$?"; python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/bench_g$g.json").read().strip().splitlines()[-1])
    print({k:j[k] for k in ("value","ms_per_step")}, j.get("roofline",{}).get("achieved"), j.get("max_score_diff_vs_fp32") or j.get("config",{}).get("max_abs_prob_diff_vs_fp32"))
except Exception as ex:
    print("parse error", ex); print(open("gpurun_out/bench_g$g.err").read()[-1500:])
PY
done
