cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c9
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "per_sequence or qlm or two_token" 2>&1 | tail -3
timeout 300 python tools/bench_qlm_xl.py 2>/dev/null | tail -1 > gpurun_out/c9/qlm_xl.json; cut -c1-1500 gpurun_out/c9/qlm_xl.json
timeout 900 python tools/sweep.py "G=10,steps=40,warmup=10,rep=2" "G=10,steps=40,warmup=10,rep=2,gemm_epi_depth=3" "G=10,steps=40,warmup=10,rep=2,gemm_epi_depth=3,gemm_stagger_us=12" "G=10,steps=40,warmup=10,rep=2,gemm_epi_depth=3,gemm_stagger_us=18" "G=10,steps=40,warmup=10,rep=2,gemm_epi_depth=3,gemm_stagger_us=25" "G=10,steps=40,warmup=10,rep=2,gemm_stagger_us=18" "G=10,steps=40,warmup=10,rep=2" "G=10,steps=40,warmup=10,rep=2,overlap=0" "G=10,steps=40,warmup=10,rep=2,overlap=0,gemm_epi_depth=3,gemm_stagger_us=18" 2>/dev/null > gpurun_out/c9/stagger_sweep.jsonl; cat gpurun_out/c9/stagger_sweep.jsonl
