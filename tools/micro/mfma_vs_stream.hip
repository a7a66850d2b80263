// Micro-benchmark (diagnostics): does the fp32 MFMA work of one wave overlap with the global-memory streaming of the OTHER
// wave of the same SIMD?  One 512-thread workgroup per CU (two waves per SIMD): waves 0-3 run bursts of independent
// v_mfma_f32_32x32x2_f32 (operands in registers), waves 4-7 stream: read N bytes (res-like, float2 per lane, 256-byte runs
// per half wave at a 16 KB row stride) and write N bytes.  Times: MFMA alone, stream alone, both; and the k_conv_reg-like
// form where EVERY wave alternates an MFMA burst with a load / store batch (phases of a wave are serial, two waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_vs_stream.bin mfma_vs_stream.hip && ./mfma_vs_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bit 0: MFMA, bit 1: stream.  SPLIT: roles by wave (0-3 / 4-7); else every wave does both, one after the other
template <int MODE, bool SPLIT, int PRIO = 0, bool AGPR = false, int KIND = 0>
__global__ void __launch_bounds__(512, 1) kms(const float* __restrict__ src, float* __restrict__ dst, float* out, int tiles_per_wave,
                                               int mfma_per_tile, int rows, size_t row_stride, float a0) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const long long t0 = clock64();
  const bool do_m = (MODE & 1) && (!SPLIT || wave < 4);
  const bool do_s = (MODE & 2) && (!SPLIT || wave >= 4);
  if (PRIO == 1 && SPLIT && wave >= 4) __builtin_amdgcn_s_setprio(3);   // stream waves above the MFMA waves
  if (PRIO == 2 && SPLIT && wave < 4) __builtin_amdgcn_s_setprio(3);    // MFMA waves above the stream waves
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + lane, b = a0 - lane;
  float2 keep = make_float2(0.f, 0.f);
  // a wave's stream tile: `rows` rows of 64 lanes x float2 (512 B contiguous per row), rows row_stride floats apart
  const int sw = SPLIT ? wave - 4 : wave, nsw = SPLIT ? 4 : 8;
  for (int it = 0; it < tiles_per_wave; ++it) {
    if (do_m) {
      for (int m = 0; m < mfma_per_tile; m += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
          else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
      }
    }
    if (do_s) {
      // KIND 0: loads + one add per value + stores; 1: loads and stores only (scalar row pointers + one lane offset: no
      // vector ALU instruction in the loop); 2: loads only; 3: stores only
      const size_t tile = (size_t(blockIdx.x) * tiles_per_wave + it) * nsw + __builtin_amdgcn_readfirstlane(sw);
      const float* s = src + tile * 128;
      float* d = dst + tile * 128;
      const unsigned lo = 2 * lane;
      float2 v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = make_float2(a, b);
      for (int r0 = 0; r0 < rows; r0 += 16) {
        if (KIND != 3) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = *reinterpret_cast<const float2*>(s + size_t(r0 + r) * row_stride + lo);
        }
        if (KIND == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(v[r].x), "v"(v[r].y));
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (KIND == 0) v[r].x += 1.f;
            *reinterpret_cast<float2*>(d + size_t(r0 + r) * row_stride + lo) = v[r];
          }
        }
      }
    }
  }
  float sum = keep.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) sum += acc[i][0];
  if (sum == 123.456f) out[0] = sum;
  if (blockIdx.x == 7 && lane == 0) out[8 + wave] = float(clock64() - t0);
}

template <int MODE, bool SPLIT, int PRIO = 0, bool AGPR = false, int KIND = 0>
static float run(const float* src, float* dst, float* out, int tpw, int mpt, int rows, size_t rs) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kms<MODE, SPLIT, PRIO, AGPR, KIND>), dim3(256), dim3(512), 0, 0, src, dst, out, tpw, mpt, rows, rs, 1.0f);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((kms<MODE, SPLIT, PRIO, AGPR, KIND>), dim3(256), dim3(512), 0, 0, src, dst, out, tpw, mpt, rows, rs, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / 5;
}

// Interleaved form: every wave, per tile, issues ONE load (next tile's row) and ONE store (previous tile's row) after every
// group of four MFMAs: do memory instructions issue in the shadow of the wave's own MFMAs?
template <int MODE>
__global__ void __launch_bounds__(512, 1) kint(const float* __restrict__ src, float* __restrict__ dst, float* out, int tiles_per_wave,
                                                size_t row_stride, float a0) {
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + lane, b = a0 - lane;
  float2 cur[32], nxt[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) cur[r] = nxt[r] = make_float2(a, b);
  const unsigned lo = 2 * lane;
  for (int it = 0; it < tiles_per_wave; ++it) {
    const size_t tile = (size_t(blockIdx.x) * tiles_per_wave + it) * 8 + wave;
    const float* s = src + (tile + 8) * 128;          // next tile of this wave
    float* d = dst + tile * 128;
#pragma unroll
    for (int g = 0; g < 32; ++g) {                    // 32 groups of 4 MFMAs = 128 per tile
      if (MODE & 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
      if (MODE & 2) {
        nxt[g] = *reinterpret_cast<const float2*>(s + size_t(g) * row_stride + lo);
        *reinterpret_cast<float2*>(d + size_t(g) * row_stride + lo) = cur[g];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) cur[r] = nxt[r];
  }
  float sum = cur[0].x;
#pragma unroll
  for (int i = 0; i < 4; ++i) sum += acc[i][0];
  if (sum == 123.456f) out[0] = sum;
}

template <int MODE>
static float run_int(const float* src, float* dst, float* out, int tpw, size_t rs) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kint<MODE>), dim3(256), dim3(512), 0, 0, src, dst, out, tpw, rs, 1.0f);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((kint<MODE>), dim3(256), dim3(512), 0, 0, src, dst, out, tpw, rs, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / 5;
}

int main() {
  // rows x 512 B per tile; 256 WGs x tpw x (4 or 8) tiles: choose ~126 MB read + 126 MB written in the split form
  const int rows = 32, tpw = 30;
  const size_t row_stride = 4096 * 4;                 // floats between rows (64 KB: like cout planes of a 128x128 map)
  const size_t tiles = size_t(256) * tpw * 8;
  const size_t n = tiles * 128 + size_t(rows) * row_stride + 1024;
  float *src, *dst, *out;
  hipMalloc(&src, n * 4);
  hipMalloc(&dst, n * 4);
  hipMalloc(&out, 64);
  hipMemset(src, 0, n * 4);
  const double mb_split = 256.0 * tpw * 4 * rows * 512 / 1e6, mb_all = 2 * mb_split;
  {
    const int mpt = 128;
    printf("stream waves at s_setprio 3: both %.1f us;  MFMA waves at s_setprio 3: both %.1f us\n",
           run<3, true, 1>(src, dst, out, tpw, mpt, rows, row_stride), run<3, true, 2>(src, dst, out, tpw, mpt, rows, row_stride));
    printf("AGPR accumulators: mfma %.1f us, both %.1f us, both with stream waves at prio 3 %.1f us\n",
           run<1, true, 0, true>(src, dst, out, tpw, mpt, rows, row_stride), run<3, true, 0, true>(src, dst, out, tpw, mpt, rows, row_stride),
           run<3, true, 1, true>(src, dst, out, tpw, mpt, rows, row_stride));
  }
  {
    const int mpt = 128;
    printf("no VALU in the stream loop: stream %.1f us, both %.1f us\n", run<2, true, 0, false, 1>(src, dst, out, tpw, mpt, rows, row_stride),
           run<3, true, 0, false, 1>(src, dst, out, tpw, mpt, rows, row_stride));
    printf("loads only: stream %.1f us, both %.1f us\n", run<2, true, 0, false, 2>(src, dst, out, tpw, mpt, rows, row_stride),
           run<3, true, 0, false, 2>(src, dst, out, tpw, mpt, rows, row_stride));
    printf("stores only: stream %.1f us, both %.1f us\n", run<2, true, 0, false, 3>(src, dst, out, tpw, mpt, rows, row_stride),
           run<3, true, 0, false, 3>(src, dst, out, tpw, mpt, rows, row_stride));
  }
  printf("interleaved (every wave: 4 MFMAs, 1 load, 1 store, ...; 128 MFMA + 32 rows per tile): mfma %.1f us, stream %.1f us, both %.1f us\n",
         run_int<1>(src, dst, out, tpw, row_stride), run_int<2>(src, dst, out, tpw, row_stride), run_int<3>(src, dst, out, tpw, row_stride));
  {
    run<3, true>(src, dst, out, tpw, 128, rows, row_stride);
    float h[16];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("split roles, both: cycles (s_memtime, 100 MHz) per wave of workgroup 7: mfma waves %.0f %.0f %.0f %.0f   stream waves %.0f %.0f %.0f %.0f\n",
           h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
  }
  for (int mpt : {128}) {
    const float tm = run<1, true>(src, dst, out, tpw, mpt, rows, row_stride);
    const float ts = run<2, true>(src, dst, out, tpw, mpt, rows, row_stride);
    const float tb = run<3, true>(src, dst, out, tpw, mpt, rows, row_stride);
    printf("split roles, %3d MFMA/tile: mfma %.1f us  stream %.1f us (%.0f MB each way, %.2f TB/s)  both %.1f us  (sum %.1f, max %.1f)\n", mpt, tm, ts,
           mb_split, 2 * mb_split / ts / 1e6 * 1e0, tb, tm + ts, tm > ts ? tm : ts);
    const float am = run<1, false>(src, dst, out, tpw, mpt, rows, row_stride);
    const float as = run<2, false>(src, dst, out, tpw, mpt, rows, row_stride);
    const float ab = run<3, false>(src, dst, out, tpw, mpt, rows, row_stride);
    printf("every wave both, %3d MFMA/tile: mfma %.1f us  stream %.1f us (%.0f MB each way, %.2f TB/s)  both %.1f us  (sum %.1f, max %.1f)\n", mpt, am, as,
           mb_all, 2 * mb_all / as / 1e6, ab, am + as, am > as ? am : as);
  }
  return 0;
}
