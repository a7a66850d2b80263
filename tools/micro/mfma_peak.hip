// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 / 16x16x4 rate on this GPU (diagnostics).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, bool MASK>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float a0, float b0, int w) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        float bb = b;
        if (MASK) bb = (unsigned(int(threadIdx.x) + u + i) < unsigned(w)) ? b : 0.f;
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[i], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> void run(const char* name, F launch, double flop_per_block_iter, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-40s blocks %5d  %8.1f us  %7.1f TFLOP/s\n", name, blocks, ms * 1e3, flop_per_block_iter * blocks * iters / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 2000;
  for (int blocks : {256, 512, 1024, 2048}) {
    run("32x32x2 4acc", [&] { hipLaunchKernelGGL((k32<4, false>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f, 64); }, 4.0 * 8 * 4 * 4096, blocks, iters);
    run("32x32x2 4acc masked", [&] { hipLaunchKernelGGL((k32<4, true>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f, 64); }, 4.0 * 8 * 4 * 4096, blocks, iters);
    run("32x32x2 2acc", [&] { hipLaunchKernelGGL((k32<2, false>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f, 64); }, 4.0 * 8 * 2 * 4096, blocks, iters);
    run("16x16x4 4acc", [&] { hipLaunchKernelGGL((k16<4>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4.0 * 8 * 4 * 2048, blocks, iters);
    run("16x16x4 8acc", [&] { hipLaunchKernelGGL((k16<8>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4.0 * 8 * 8 * 2048, blocks, iters);
  }
  return 0;
}
