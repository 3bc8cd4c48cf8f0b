mkdir -p gpurun_out/c28
RK_LAYERS=8 timeout 300 python tools/llama_attn_check.py > gpurun_out/c28/llama_attn.jsonl 2> gpurun_out/c28/err.log
echo "rc=$?"
cat gpurun_out/c28/llama_attn.jsonl
RK_KO=0,256 RK_ENGINE_LIB=exp/librk_engine_measure.so timeout 200 python tools/llama_attn_ko.py > gpurun_out/c28/ko4.jsonl 2>> gpurun_out/c28/err.log
RK_NW=8 RK_KO=0,256 RK_ENGINE_LIB=exp/librk_engine_measure.so timeout 200 python tools/llama_attn_ko.py > gpurun_out/c28/ko8.jsonl 2>> gpurun_out/c28/err.log
cat gpurun_out/c28/ko4.jsonl gpurun_out/c28/ko8.jsonl
grep -v "amdgpu.ids\|tools\]" gpurun_out/c28/err.log | tail -3
