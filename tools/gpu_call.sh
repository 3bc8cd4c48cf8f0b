mkdir -p gpurun_out/c24
timeout 400 python tools/llama_gemm_variants.py > gpurun_out/c24/variants.jsonl 2> gpurun_out/c24/err.log
echo "rc=$?"
cat gpurun_out/c24/variants.jsonl
tail -5 gpurun_out/c24/err.log
