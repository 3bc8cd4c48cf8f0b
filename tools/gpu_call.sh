cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c2
timeout 600 python tools/chain_check.py > gpurun_out/c2/chain_check.jsonl 2> gpurun_out/c2/chain_check.err; echo "chain_check rc=$?"
cat gpurun_out/c2/chain_check.jsonl | cut -c1-1500
tail -5 gpurun_out/c2/chain_check.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline --no_per_query --no_extras > gpurun_out/c2/bench_chain1.json 2> gpurun_out/c2/bench_chain1.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline --no_per_query --no_extras --opt chain=0 > gpurun_out/c2/bench_chain0.json 2> gpurun_out/c2/bench_chain0.err; echo "bench rc=$?"
python - <<'PY'
import json
for n in ("chain1", "chain0"):
    try:
        j = json.loads(open(f"gpurun_out/c2/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, j["value"], j["ms_per_step"], j["config"]["whole_path_frac_of_mfma_peak"], j["config"]["timed_regions_ms"], {k: v["ms_per_step"] for k, v in j["roofline"]["per_class"].items()})
    except Exception as exc:
        print(n, "failed", exc); print(open(f"gpurun_out/c2/bench_{n}.err").read()[-1500:])
PY
