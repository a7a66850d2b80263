cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp faster-voxelpose_amd/libfvp_hip.so /tmp/base.so
(
for i in 1 2; do
cp /tmp/base.so faster-voxelpose_amd/libfvp_hip.so
echo "=== scalar per-op"; python tools/bench_conv.py --ops 3,4,9,16 2>&1 | grep " op"
echo "=== scalar s1"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 --no-prof | cut -c80-130
echo "=== scalar s3"; python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-prof | cut -c80-130
cp tools/ab/libfvp_hip_pk.so faster-voxelpose_amd/libfvp_hip.so
echo "=== packed per-op"; python tools/bench_conv.py --ops 3,4,9,16 2>&1 | grep " op"
echo "=== packed s1"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 --no-prof | cut -c80-130
echo "=== packed s3"; python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-prof | cut -c80-130
done
cp /tmp/base.so faster-voxelpose_amd/libfvp_hip.so
) > gpurun_out/diag41.log 2>&1
