#!/bin/bash
# scratch: the driver's bench command on the current tree (the setwise leg now runs run.py's default of 32 queries in lockstep)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/x; mkdir -p $O
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_many32.json 2> $O/bench_many32.err; echo "rc=$?"; tail -3 $O/bench_many32.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/x/bench_many32.json').read().strip().splitlines()[-1])
print(j['value'], j['roofline']['frac'], j['roofline'].get('frac_rocprof') is not None)
print({k:(v['ms_per_query'], v['frac_of_mfma_peak']) for k,v in j['config']['setwise_query'].items() if isinstance(v,dict)})
PY
timeout 600 python -m pytest tests/test_gpu_rerank.py -q -m gpu -k "setwise or lockstep or many" 2>&1 | tail -2
