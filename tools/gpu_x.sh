#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/x; mkdir -p $O
: > $O/depth_ab.txt
for rep in 1 2 3; do for n in cur d3; do echo "== $n rep $rep" >> $O/depth_ab.txt; RK_ENGINE_LIB=exp/librk_$n.so timeout 600 python tools/ab_profile.py base: 2>/dev/null | grep -v amdgpu | cut -c1-700 >> $O/depth_ab.txt; done; done
cat $O/depth_ab.txt
