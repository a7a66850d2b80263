"""MI355X-native Faster-VoxelPose inference hot path (heatmaps -> HDN -> JLN -> 3D joints).

Host side mirrors the reference's ``lib/models`` operator API; all arithmetic runs in
hand-written HIP kernels (``csrc/``) behind the C ABI declared in ``include/fvp.h``.
"""
__version__ = "0.1.0"
