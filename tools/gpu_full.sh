bash tools/gpu_check.sh r01f
bash tools/gpu_pmc.sh r01f
