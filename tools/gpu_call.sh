mkdir -p gpurun_out/c20
RK_LAYERS=8 timeout 420 python tools/llama_attn_check.py > gpurun_out/c20/llama_attn.jsonl 2> gpurun_out/c20/err.log
echo "rc=$?"
cat gpurun_out/c20/llama_attn.jsonl
tail -5 gpurun_out/c20/err.log
