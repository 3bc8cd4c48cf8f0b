cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c12
timeout 900 python tools/sweep.py "G=10,steps=40,warmup=10,rep=2" "G=10,steps=40,warmup=10,rep=2,gemm_split=2" "G=10,steps=40,warmup=10,rep=2,gemm_split=3" "G=10,steps=40,warmup=10,rep=2" "G=10,steps=40,warmup=10,rep=2,gemm_split=2" "G=10,steps=40,warmup=10,rep=2,gemm_split=3" 2>/dev/null > gpurun_out/c12/split_sweep.jsonl; cat gpurun_out/c12/split_sweep.jsonl
