cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c8
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/c8/pytest_gpu.log; cat gpurun_out/c8/pytest_gpu.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c8/qlm_prof -o qlm -- python $GRAFT_REPO_ROOT/tools/bench_qlm_xl.py > $GRAFT_REPO_ROOT/gpurun_out/c8/qlm_prof_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/c8/qlm_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/c8/qlm_xl_kernel_stats.csv; head -16 gpurun_out/c8/qlm_xl_kernel_stats.csv | cut -c1-200
find gpurun_out/c8/qlm_prof -name "*kernel_trace.csv" -delete
RK_L=1560 timeout 300 python tools/profile_compare.py 2>/dev/null | tail -1 > gpurun_out/c8/compare_profile_1560.json; cat gpurun_out/c8/compare_profile_1560.json
