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
// short workgroups (like one conv tile: 576 MFMAs per wave) with a big dynamic LDS allocation and
// barriers every 144 MFMAs: measures workgroup launch / turnover cost
template <bool BAR>
__global__ void __launch_bounds__(256) kshort(float* out, int chunks, float a0, float b0) {
  extern __shared__ float lds[];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  lds[threadIdx.x] = a;
  for (int c = 0; c < chunks; ++c) {
    for (int it = 0; it < 36; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    if (BAR) __syncthreads();
  }
  float s = lds[(threadIdx.x + 1) & 255];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the conv inner loop in isolation: per step 1 A-word + 4 B-words from LDS (software pipelined,
// explicit wait), 4 masked MFMAs; 144 MFMAs per chunk, barrier per chunk
template <int MODE>
__global__ void __launch_bounds__(256, 2) kloop(float* out, int chunks, int W, int stride) {
  extern __shared__ float lds[];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, half = lane >> 5;
  for (int i = t; i < 8192; i += 256) lds[i] = float(i & 255) * 0.001f;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int poff[4], txs[4];
  for (int pb = 0; pb < 4; ++pb) { poff[pb] = (pb * 32 + l31) + half * stride; txs[pb] = l31 + pb - 1; }
  const float* ws = lds + 4096 + l31 + half * 288;
  for (int c = 0; c < chunks; ++c) {
    float av[2], bv[2][4];
    av[0] = ws[0];
    for (int pb = 0; pb < 4; ++pb) bv[0][pb] = lds[poff[pb]];
    for (int ci = 0; ci < 4; ++ci) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int cur = (ci + tap) & 1, nxt = cur ^ 1;   // (parity approximated; ci even/odd alternate)
        if (MODE >= 1) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); }
        const int ky = (tap + 1) / 3, kx = (tap + 1) % 3;
        av[nxt] = ws[(tap + 1) * 32 + ci * 576];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) bv[nxt][pb] = lds[poff[pb] + ky * W + kx + ci * 2 * stride];
        if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
          const float bb = (MODE >= 2 ? unsigned(txs[pb] + (tap % 3)) < unsigned(W) : true) ? bv[cur][pb] : 0.f;
          acc[pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur], bb, acc[pb], 0, 0, 0);
        }
        if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + t] = s;
}
int main() {
  {
    float* o3; hipMalloc(&o3, 8192 * 256 * 4);
    hipFuncSetAttribute((const void*)kloop<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)kloop<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)kloop<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int blocks : {512, 1920}) {
      run("conv-like loop, compiler schedule", [&] { hipLaunchKernelGGL((kloop<0>), dim3(blocks), dim3(256), 76 * 1024, 0, o3, 4, 64, 640); }, 4.0 * 144 * 4 * 4096, blocks, 1);
      run("conv-like loop, pinned swp", [&] { hipLaunchKernelGGL((kloop<1>), dim3(blocks), dim3(256), 76 * 1024, 0, o3, 4, 64, 640); }, 4.0 * 144 * 4 * 4096, blocks, 1);
      run("conv-like loop, pinned swp + mask", [&] { hipLaunchKernelGGL((kloop<2>), dim3(blocks), dim3(256), 76 * 1024, 0, o3, 4, 64, 640); }, 4.0 * 144 * 4 * 4096, blocks, 1);
    }
  }

  {
    float* o2; hipMalloc(&o2, 8192 * 256 * 4);
    hipFuncSetAttribute((const void*)kshort<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)kshort<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int lds : {1024, 60 * 1024, 76 * 1024}) for (int blocks : {512, 1920, 3840}) {
      char nm[64]; snprintf(nm, 64, "short WG 576 MFMA/wave bar lds %dK", lds / 1024);
      run(nm, [&] { hipLaunchKernelGGL((kshort<true>), dim3(blocks), dim3(256), lds, 0, o2, 4, 1.f, 2.f); }, 4.0 * 144 * 4 * 4096 / 1.0 / 2000.0 * 2000.0, blocks, 1);
      snprintf(nm, 64, "short WG 576 MFMA/wave nobar lds %dK", lds / 1024);
      run(nm, [&] { hipLaunchKernelGGL((kshort<false>), dim3(blocks), dim3(256), lds, 0, o2, 4, 1.f, 2.f); }, 4.0 * 144 * 4 * 4096, blocks, 1);
    }
  }

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
