cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c14
for o in "attn_split=1" "attn_split=0" "attn_split=0,attn_tiled_occ=3" "attn_split=1,attn_tiled_occ=1"; do
for b in 8 1; do
echo "== B=$b $o"; RK_OPTS=$o RK_L=1560 RK_B=$b timeout 300 python tools/profile_compare.py 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print(j['likelihood_ms'], j['likelihood_classes_ms']['enc_attn'])"
done; done 2>&1 | tee gpurun_out/c14/attn_split.txt
