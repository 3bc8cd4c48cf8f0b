#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/r2k_pytest.log; cat gpurun_out/r2k_pytest.log
timeout 900 python tools/sweep.py "G=8,steps=64,warmup=16,rep=2,dec_fuse_norm=0" "G=8,steps=64,warmup=16,rep=2,dec_fuse_norm=1" "G=10,steps=20,warmup=5,rep=3,dec_fuse_norm=0" "G=10,steps=20,warmup=5,rep=3,dec_fuse_norm=1" > gpurun_out/r2k_sweep.jsonl 2> gpurun_out/r2k_sweep.err
cat gpurun_out/r2k_sweep.jsonl; tail -3 gpurun_out/r2k_sweep.err
for o in "dec_fuse_norm=0" "dec_fuse_norm=1" "attn_tiled_occ=2" "attn_tiled_occ=3"; do echo "== $o"; RK_OPTS=$o timeout 300 python tools/profile_compare.py 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['likelihood_ms'], j['generation_ms'], {k:v[0] for k,v in j['likelihood_classes_ms'].items()})"; done
echo "== B=9"; for o in "attn_tiled_occ=1" "attn_tiled_occ=2" "attn_tiled_occ=3"; do RK_B=9 RK_OPTS=$o timeout 300 python tools/profile_compare.py 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$o', j['likelihood_ms'], j['generation_ms'], j['likelihood_classes_ms']['enc_attn'])"; done
