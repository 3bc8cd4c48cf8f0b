#!/bin/bash
# Round evidence, ONE lease: box id, the GPU test suite, the bench configuration under rocprofv3 (kernel stats + PMC passes, one
# stream), then the driver's bench command (complete line, un-profiled) back to back, and the side legs.  Everything lands in gpurun_out/final/; copy what is to be judged into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
{ echo "hostname: $(hostname)"; echo "date: $(date -u +%FT%TZ)"; rocm-smi --showuniqueid --showserial 2>/dev/null | grep -i "unique\|serial"; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4; } > $OUT/box.txt 2>&1; cat $OUT/box.txt
if [ -z "$SKIP_TESTS" ]; then timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log; fi   # SKIP_TESTS=1: the suite ran in its own call
# the profiled passes FIRST: their summaries become profiles/*_latest on this box, so that the bench line of the SAME lease
# quotes this lease's rocprofv3 statistics (roofline.frac_rocprof) and PMC traffic (roofline.traffic)
bash tools/gpu_prof.sh > $OUT/prof_stdout.txt 2>&1; tail -22 $OUT/prof_stdout.txt | cut -c1-220
cp gpurun_out/prof/bench_kernel_stats.csv $OUT/ 2>/dev/null; cp gpurun_out/prof/pmc_summary.json $OUT/ 2>/dev/null
cp gpurun_out/prof/bench_kernel_stats.csv profiles/bench_kernel_stats_latest.csv 2>/dev/null; cp gpurun_out/prof/pmc_summary.json profiles/pmc_summary_latest.json 2>/dev/null
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"; tail -c 1200 $OUT/bench_driver.json; tail -4 $OUT/bench_driver.err
if [ -z "$SKIP_SIDE" ]; then
timeout 600 python bench.py --steps 128 --warmup 16 --no_cpu_baseline --no_per_query --no_extras > $OUT/bench_128.json 2> $OUT/bench_128.err; python -c "
import json; j=json.loads(open('$OUT/bench_128.json').read().strip().splitlines()[-1]); print('128 steps:', j['value'], j['config']['whole_path_frac_of_mfma_peak'], j['roofline']['frac'], j['roofline']['family']['frac'])"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --mode shard --no_cpu_baseline --no_per_query --no_extras > $OUT/bench_shard_n1.json 2> /dev/null; python -c "
import json; j=json.loads(open('$OUT/bench_shard_n1.json').read().strip().splitlines()[-1]); print('shard mode, 1 rank:', j['value'], j['config']['gather_check'])"
timeout 600 python bench.py --steps 20 --warmup 5 --group 1 --no_cpu_baseline --no_per_query --no_profile --no_extras > $OUT/bench_group1.json 2> /dev/null; python -c "
import json; j=json.loads(open('$OUT/bench_group1.json').read().strip().splitlines()[-1]); print('group 1:', j['value'])"
timeout 300 python tools/profile_compare.py 2>/dev/null | tail -1 > $OUT/compare_profile.json; cat $OUT/compare_profile.json
fi
