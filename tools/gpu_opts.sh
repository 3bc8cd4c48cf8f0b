#!/bin/bash
# A/B an engine option through bench.py: OPTS="attn_heads_per_wg=1 attn_heads_per_wg=4" bash tools/gpu_opts.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for o in ${OPTS}; do
  timeout 300 python bench.py --no_cpu_baseline --steps ${STEPS:-64} --warmup 16 --opt $o ${BENCH_EXTRA} > gpurun_out/bench_opt.json 2> gpurun_out/bench_opt.err; rc=$?
  python - "$o" $rc <<'PY'
import json, sys
try:
    j=json.loads(open("gpurun_out/bench_opt.json").read().strip().splitlines()[-1])
    pc=j["roofline"]["per_class"]
    print(sys.argv[1], {k:j[k] for k in ("value","ms_per_step")}, "attn", pc["enc_attn"]["ms_per_step"], "norm", pc["norm"]["ms_per_step"], "gemm", j["roofline"]["achieved"])
except Exception as ex:
    print(sys.argv[1], "rc", sys.argv[2], "error", ex); print(open("gpurun_out/bench_opt.err").read()[-800:])
PY
done
