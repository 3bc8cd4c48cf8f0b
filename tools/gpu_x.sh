#!/bin/bash
# scratch: rocprofv3 kernel statistics of one lockstep setwise call shape (eight 1 560-token prompts, likelihood + generation)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
R=$PWD; O=$PWD/gpurun_out/x/lockstep_stats; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" 2> $O/build.log
cd /tmp && RK_B=8 RK_L=1560 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ls -- python $R/tools/profile_compare.py > $O/stdout.txt 2>&1
cd $R; head -25 $O/ls_kernel_stats.csv | cut -c1-200; find $O -name "*kernel_trace.csv" -size +2M -delete
