tag="${1:-r01g}"
bash tools/gpu_check.sh $tag
bash tools/gpu_pmc.sh $tag
