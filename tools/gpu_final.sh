#!/bin/bash
# round evidence: GPU test suite, the driver's bench command (complete line), rocprofv3 kernel stats + PMC of the same
# configuration (one stream), the config-3 setwise query timing.  Everything lands in gpurun_out/final/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log; fi   # SKIP_TESTS=1: the suite ran in its own call
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench_driver.json
timeout 600 python bench.py --steps 128 --warmup 16 --no_cpu_baseline --no_per_query > $OUT/bench_128.json 2> $OUT/bench_128.err; python -c "
import json; j=json.loads(open('$OUT/bench_128.json').read().strip().splitlines()[-1]); print('128 steps:', j['value'], j['config']['whole_path_frac_of_mfma_peak'], j['roofline']['frac'], j['roofline']['family']['frac'])"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --mode shard --no_cpu_baseline --no_per_query --no_extras > $OUT/bench_shard_n1.json 2> /dev/null; python -c "
import json; j=json.loads(open('$OUT/bench_shard_n1.json').read().strip().splitlines()[-1]); print('shard mode, 1 rank:', j['value'], j['config']['gather_check'])"
timeout 600 python bench.py --steps 20 --warmup 5 --group 1 --no_cpu_baseline --no_per_query --no_profile > $OUT/bench_group1.json 2> /dev/null; python -c "
import json; j=json.loads(open('$OUT/bench_group1.json').read().strip().splitlines()[-1]); print('group 1:', j['value'])"
timeout 900 python tools/bench_setwise_query.py 2>/dev/null | tail -1 > $OUT/setwise_query.json; cat $OUT/setwise_query.json
timeout 300 python tools/profile_compare.py 2>/dev/null | tail -1 > $OUT/compare_profile.json; cat $OUT/compare_profile.json
timeout 1200 python tools/bench_llama.py 2> $OUT/llama_bench.err | tail -1 > $OUT/llama_bench.json; cat $OUT/llama_bench.json; tail -2 $OUT/llama_bench.err
bash tools/gpu_prof.sh > $OUT/prof_stdout.txt 2>&1; tail -22 $OUT/prof_stdout.txt | cut -c1-220
cp gpurun_out/prof/bench_kernel_stats.csv $OUT/ 2>/dev/null; cp gpurun_out/prof/pmc_summary.json $OUT/ 2>/dev/null
