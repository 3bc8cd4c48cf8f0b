cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c5
RK_ENGINE_LIB=exp/librk_engine_measure.so RK_ONLY=1,2 RK_DEBUG=0,4,2,1 RK_LEAD=3 timeout 600 python tools/chain_trace.py > gpurun_out/c5/trace.jsonl 2> gpurun_out/c5/trace.err; echo "trace rc=$?"
RK_ENGINE_LIB=exp/librk_engine_measure.so RK_ONLY=1 RK_DEBUG=0 RK_LEAD=1,6 timeout 300 python tools/chain_trace.py >> gpurun_out/c5/trace.jsonl 2>> gpurun_out/c5/trace.err; echo "trace rc=$?"
cut -c1-1400 gpurun_out/c5/trace.jsonl; tail -3 gpurun_out/c5/trace.err
