#!/bin/bash
# the whole -m gpu suite on the current tree (log -> gpurun_out/suite/pytest_gpu.log)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/suite; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" 2> $O/build.log
timeout 3000 python -m pytest tests/ -q -m gpu ${PYTEST_X:-} --durations=15 > $O/pytest_gpu.log 2>&1
tail -n 40 $O/pytest_gpu.log
