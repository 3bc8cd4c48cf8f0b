#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/profile_compare.py 2>/dev/null | tail -1 > gpurun_out/r2j_compare.json; cat gpurun_out/r2j_compare.json
RK_B=9 timeout 300 python tools/profile_compare.py 2>/dev/null | tail -1 > gpurun_out/r2j_compare_b9.json; cat gpurun_out/r2j_compare_b9.json
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2j_prof -o cmp -- python $GRAFT_REPO_ROOT/tools/profile_compare.py > /dev/null 2>&1
head -30 $GRAFT_REPO_ROOT/gpurun_out/r2j_prof/cmp_kernel_stats.csv | cut -c1-160
find $GRAFT_REPO_ROOT/gpurun_out/r2j_prof -name "*kernel_trace.csv" -delete
