// Micro-benchmark (diagnostics): do the MFMA bursts of one wave and the VALU / LDS work of the OTHER wave of the same SIMD
// overlap?  One 512-thread workgroup per CU (two waves per SIMD, as k_conv_wino): waves 0-3 issue bursts of 16 independent
// v_mfma_f32_16x16x4_f32, waves 4-7 run a VALU chain (+ optional LDS reads).  Times: MFMA alone, VALU alone, both.
//   hipcc --offload-arch=gfx950 -O3 -o coexec.bin coexec.hip && ./coexec.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: waves 0-3 do MFMA bursts; bit 1: waves 4-7 do VALU work; ACC_AGPR: accumulators in AGPRs (inline asm)
template <int MODE, bool ACC_AGPR, bool LDS_READS, bool SAME_WAVE>
__global__ void __launch_bounds__(512, 2) kco(float* out, int iters, float a0, float b0) {
  __shared__ float lds[4096];
  const int t = threadIdx.x, wave = t >> 6;
  for (int i = t; i < 4096; i += 512) lds[i] = float(i) * 1e-3f;
  __syncthreads();
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + t, b = b0 - t;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a0 * i + t;
  const bool do_m = (MODE & 1) && (SAME_WAVE || wave < 4);
  const bool do_v = (MODE & 2) && (SAME_WAVE || wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (ACC_AGPR) {
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        } else {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
      }
    }
    if (do_v) {
      // ~64 dependent-free VALU instructions (8 chains x 8) + optionally 8 LDS reads: roughly the fetch / transform phase
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        if (LDS_READS) v[r] += lds[(t * 4 + r * 64 + it) & 4095];
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + t] = s;
}

template <typename F> float run(const char* name, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  printf("%-64s %8.1f us\n", name, ms * 1e3);
  return ms;
}

#define RUN(M, A, L, S, name) run(name, [&] { hipLaunchKernelGGL((kco<M, A, L, S>), dim3(blocks), dim3(512), 0, 0, out, iters, 1.0f, 2.0f); })

int main() {
  float* out;
  const int blocks = 256, iters = 4000;
  hipMalloc(&out, blocks * 512 * sizeof(float));
  printf("one 512-thread workgroup per CU, %d iterations; per iteration: 16 MFMA 16x16x4 f32 (waves 0-3) / 64 v_fma (waves 4-7)\n", iters);
  printf("MFMA floor per iteration: 16 x 32 = 512 cycles -> %.1f us at 2.4 GHz\n", iters * 512 / 2.4e3);
  RUN(1, false, false, false, "MFMA only (waves 0-3), accumulators in VGPRs");
  RUN(2, false, false, false, "VALU only (waves 4-7)");
  RUN(3, false, false, false, "MFMA (waves 0-3) + VALU (waves 4-7), VGPR accumulators");
  RUN(1, true, false, false, "MFMA only, accumulators in AGPRs");
  RUN(3, true, false, false, "MFMA + VALU, AGPR accumulators");
  RUN(2, false, true, false, "VALU + LDS reads only (waves 4-7)");
  RUN(3, false, true, false, "MFMA + VALU + LDS reads, VGPR accumulators");
  RUN(3, true, true, false, "MFMA + VALU + LDS reads, AGPR accumulators");
  RUN(1, false, false, true, "all 8 waves: MFMA only (two bursts per SIMD and iteration)");
  RUN(2, false, false, true, "all 8 waves: VALU only");
  RUN(3, false, false, true, "all 8 waves: MFMA burst then VALU block each (k_conv_wino's shape), VGPR");
  RUN(3, true, false, true, "all 8 waves: MFMA burst then VALU block each, AGPR");
  hipFree(out);
  return 0;
}
