#!/usr/bin/env python
"""Key numbers of one evidence lease (tools/gpu_final.sh output directory or profiles/<prefix>) for DESIGN.md section 5 / README."""
import json, sys, os
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final"
pre = sys.argv[2] if len(sys.argv) > 2 else ""
def J(n):
    return json.loads(open(os.path.join(d, pre + n)).read().strip().splitlines()[-1])
b = J("bench_driver.json"); c = b["config"]; r = b["roofline"]
print("box:", [l for l in open(os.path.join(d, pre + "box.txt")) if "Unique ID:" in l][0].split("Unique ID:")[1].strip())
print(f"driver: {b['value']} p/s, {b['ms_per_step']} ms/step, whole-path {c['whole_path_frac_of_mfma_peak']}, regions {c['timed_regions_ms']}")
fr = r.get("frac_rocprof")
print(f"ffn_in: events {r['avg_launch_us']} us {r['achieved']} TF frac {r['frac']}; rocprof {fr if not isinstance(fr, dict) else (fr['avg_launch_us'], fr['frac'])}; traffic {r['traffic']}")
print("per_class:", {k: (v['ms_per_step'], v['tflops']) for k, v in r['per_class'].items()})
print("128 steps:", J("bench_128.json")["value"], "| shard n1:", J("bench_shard_n1.json")["value"], "| group1:", J("bench_group1.json")["value"])
print("per_query:", {k: (v['ms_per_query'], v['passages_per_s']) for k, v in c['per_query'].items()})
print("ragged:", c['ragged']['passages_per_s'], c['ragged']['frac_of_mfma_peak'])
ss = c['shard_share']; print("shard_share:", ss['ms_per_query'], ss['frac_of_mfma_peak_per_gpu'], ss['predicted_8gpu_strong_scaling_passages_per_s'], "| grouped16:", ss['grouped16']['ms_per_query'], ss['grouped16']['frac_of_mfma_peak_per_gpu'], ss['grouped16']['predicted_8gpu_strong_scaling_passages_per_s'])
print("predicted_8gpu weak:", c['predicted_8gpu']['weak_passages_per_s'])
sq = c['setwise_query']; print("setwise:", {k: (v['ms_per_query'], v['frac_of_mfma_peak']) for k, v in sq.items() if isinstance(v, dict)})
q = c['qlm_xl']; print("qlm_xl:", q['ms_per_query'], q['passages_per_s'], q['frac_of_mfma_peak'], {k: v[0] for k, v in q['classes_ms'].items()}, "share:", q.get('share_of_8_ranks', {}).get('ms_per_query'))
l = c['llama_compare']; print("llama:", l['B1']['ms_per_compare'], l['B1']['frac_of_mfma_peak'], l['B4']['ms_per_compare'], l['B4']['frac_of_mfma_peak'], l['classes_ms_B1'])
cb = b['cpu_baseline']; print("cpu:", cb['value'], cb['cores'], cb['max_abs_score_diff_vs_gpu'])
cp = J("compare_profile.json"); print("compare:", cp['likelihood_ms'], cp['generation_ms'], cp['likelihood_classes_ms'])
