cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp faster-voxelpose_amd/libfvp_hip.so /tmp/base.so
(
for i in 1 2; do
cp /tmp/base.so faster-voxelpose_amd/libfvp_hip.so
echo "=== base per-op"; python tools/bench_conv.py --ops 3,9,16 2>&1 | grep " op"
echo "=== base s3"; python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-prof | cut -c80-130
cp tools/ab/libfvp_hip_nt.so faster-voxelpose_amd/libfvp_hip.so
echo "=== nt per-op"; python tools/bench_conv.py --ops 3,9,16 2>&1 | grep " op"
echo "=== nt s3"; python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-prof | cut -c80-130
done
cp /tmp/base.so faster-voxelpose_amd/libfvp_hip.so
) > gpurun_out/diag35.log 2>&1
