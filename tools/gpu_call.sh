cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c17
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/c17/pytest_gpu.log 2>&1; grep -E "passed|failed|Error|assert " gpurun_out/c17/pytest_gpu.log | head -20
timeout 900 python tools/bench_setwise_query.py > gpurun_out/c17/setwise.json 2>/dev/null; RK_WORDS=140 RK_QUERY_WORDS=24 RK_MANY=8 timeout 900 python tools/bench_setwise_query.py 2>/dev/null | tail -1 > gpurun_out/c17/setwise_s3.json; python - <<'PY'
import json
j=json.loads(open('gpurun_out/c17/setwise_s3.json').read().strip().splitlines()[-1])
print({k:(v['ms_per_query'], v.get('frac_of_mfma_peak')) for k,v in j.items() if isinstance(v,dict)})
PY
