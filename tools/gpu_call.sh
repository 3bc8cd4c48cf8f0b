cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c11
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rerank.py -q -x -p no:cacheprovider -k "per_sequence or qlm or two_token or greedy or setwise or generation" > gpurun_out/c11/pytest.log 2>&1; grep -E "passed|failed|error|Error|assert" gpurun_out/c11/pytest.log | head -20
