cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/lease4; rm -rf $OUT; mkdir -p $OUT
{ echo "hostname: $(hostname)"; echo "date: $(date -u +%FT%TZ)"; rocm-smi --showuniqueid --showserial 2>/dev/null | grep -i "unique\|serial"; } > $OUT/box.txt 2>&1; cat $OUT/box.txt
timeout 330 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench_driver.json; tail -3 $OUT/bench_driver.err
timeout 660 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $OUT/pytest_full.log | tail -5 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
