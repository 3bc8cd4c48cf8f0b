#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
RK_OPTS=";overlap=0;dec_graph=0;;overlap=0" timeout 300 python tools/share_probe.py 2>/dev/null | tee gpurun_out/x/share_probe.txt
