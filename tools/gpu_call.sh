cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c6
RK_ENGINE_LIB=exp/librk_engine_measure.so RK_ONLY=1,2 RK_DEBUG=0,6,14 RK_LEAD=3 timeout 600 python tools/chain_trace.py > gpurun_out/c6/trace.jsonl 2> gpurun_out/c6/trace.err; echo "trace rc=$?"
cut -c1-1500 gpurun_out/c6/trace.jsonl; tail -3 gpurun_out/c6/trace.err
timeout 600 python tools/chain_check.py > gpurun_out/c6/chain_check.jsonl 2> gpurun_out/c6/chain_check.err; echo "chain_check rc=$?"
grep -v '"shape"' gpurun_out/c6/chain_check.jsonl | cut -c1-1300; grep -c '"lead3_bit_identical": true' gpurun_out/c6/chain_check.jsonl
tail -3 gpurun_out/c6/chain_check.err
