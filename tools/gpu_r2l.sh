#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm or fold or flan_t5_large or setwise_shape" 2>&1 | tail -12 > gpurun_out/r2l_pytest.log; cat gpurun_out/r2l_pytest.log
for M in 1450 2900 5888 13050; do RK_BENCH_M=$M RK_GEMM_VARIANTS=1,4,5,6 timeout 300 python tools/gemm_bench.py 20 2>/dev/null | grep -v JSON; done
for o in "gemm_variant=0" "gemm_variant=6"; do echo "== $o"; RK_OPTS=$o timeout 300 python tools/profile_compare.py 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['likelihood_ms'], j['generation_ms'], {k:v[0] for k,v in j['likelihood_classes_ms'].items()})"; done
