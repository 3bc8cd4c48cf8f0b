cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c3
L=gpurun_out/c3/debug.log; : > $L
run() { echo "== $*" >> $L; RK_ENGINE_LIB=exp/librk_engine_measure.so timeout 120 python tools/chain_debug.py "$@" >> $L 2>&1; echo "rc=$?" >> $L; }
# flan-t5-small dims, 40 x 184 tokens (29 panels): everything knocked out -> one piece at a time
run flan-t5-small 40 1 63
run flan-t5-small 40 1 47   # + DMA
run flan-t5-small 40 1 39   # + DMA + epilogues
run flan-t5-small 40 1 35   # + publish
run flan-t5-small 40 1 33   # + flag load / wait
run flan-t5-small 40 1 32   # + row-factor loads
run flan-t5-small 40 1 0    # + async claim = product form, pair A only
run flan-t5-small 40 2 0    # pair B only
run flan-t5-small 40 0 0    # both
run flan-t5-large 100 0 0
grep -v "amdgpu.ids\|^\[tools\]" $L | cut -c1-400
