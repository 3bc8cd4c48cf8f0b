bash tools/gpu_final.sh
