#!/bin/bash
# scratch: whole GPU suite on the current tree
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
PYTEST_X= bash tools/gpu_suite.sh
