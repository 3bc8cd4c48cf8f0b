mkdir -p gpurun_out/c30
timeout 200 python -c "
import sys, json
sys.path[:0]=['tools','llm-rankers_amd','.']
import bench_setwise_query as b
out=b.run(reps=2, many=16, one_by_one=False, words=140, query_words=24)
for k,v in out.items():
    v.pop('top10', None)
print(json.dumps(out))
" > gpurun_out/c30/many16.json 2> gpurun_out/c30/err.log; echo "rc=$?"
cat gpurun_out/c30/many16.json; grep -v "amdgpu.ids" gpurun_out/c30/err.log | tail -3
