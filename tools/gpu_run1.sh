#!/bin/bash
# First GPU visit: runtime-order probe, smoke, GPU parity tests, a short bench with profile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{ rocminfo | grep -E "Marketing Name|gfx9" | head -4; nproc; free -g | head -2; } > gpurun_out/box.txt 2>&1
RK_IMPORT_TORCH_FIRST=0 timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_notorch.log 2>&1
echo "smoke (system HIP runtime, no torch) rc=$?"; tail -3 gpurun_out/smoke_notorch.log
timeout 600 python -c "import torch; import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_torch.log 2>&1
echo "smoke (torch imported first) rc=$?"; tail -3 gpurun_out/smoke_torch.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -60 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
timeout 300 python bench.py --group 1 --no_cpu_baseline > gpurun_out/bench_nooverlap.json 2>> gpurun_out/bench.err
echo "bench(group=1) rc=$?"; cat gpurun_out/bench_nooverlap.json
