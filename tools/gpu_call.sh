SKIP_TESTS=1 bash tools/gpu_final.sh
