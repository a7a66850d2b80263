// Micro-benchmark (diagnostics): what does the texture-addresser / L1 charge a gather for?  A wave instruction
// (global_load_dwordx4, 64 lanes x 16 B) reads G groups of L contiguous lanes; each group reads L*16 contiguous bytes at a
// pseudo-random pixel of a 240 x 128 x 64 B map (L1 / L2 resident, as the tri-plane kernel's taps):
//   A  L = 4  : 16 segments of  64 B per instruction (the quad-lane form: one tap of 16 channels)
//   B  L = 8  :  8 segments of 128 B, 128-byte aligned (nw + ne of an even x0)
//   C  L = 8  :  8 segments of 128 B starting at an odd pixel (straddling two lines)
//   D  L = 16 :  4 segments of 256 B
// Reported: time per instruction-equivalent and bytes/s.   hipcc --offload-arch=gfx950 -O3 -o gather_lines.bin gather_lines.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef WIN
#define WIN 2048u
#endif

template <int L, int ODD>
__global__ void __launch_bounds__(512) kg(const float4* __restrict__ map, int npix, int iters, float* out) {
  const int t = threadIdx.x, lane = t & 63;
  const int grp = (blockIdx.x * 8 + (t >> 6)) * (64 / L) + lane / L, sub = lane % L;
  unsigned s = 0x9E3779B9u * unsigned(grp + 1);
  float4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    // neighbouring groups of a wave hit nearby pixels (a 4 x 4 voxel patch): a small window around a wave-level base
    unsigned base = (blockIdx.x * 977u + it * 131u) % unsigned(npix - 4096);
    unsigned px = base + (s >> 20) % WIN;
    if (L >= 8) px = (px & ~1u) + ODD;               // 128-byte aligned (or deliberately odd)
    const float4 v = map[size_t(px) * 4 + sub];        // pixel = 4 float4 (64 B); L lanes read L*16 contiguous bytes
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  out[blockIdx.x * 512 + t] = acc.x + acc.y + acc.z + acc.w;
}

template <int L, int ODD> void run(const char* name, const float4* map, int npix, float* out) {
  const int blocks = 256 * 4, iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kg<L, ODD>), dim3(blocks), dim3(512), 0, 0, map, npix, iters, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((kg<L, ODD>), dim3(blocks), dim3(512), 0, 0, map, npix, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double instr = double(blocks) * 8 * iters;                  // wave instructions
  printf("%-44s %8.1f us  %6.2f ns/instr/CU  %6.1f TB/s\n", name, ms * 1e3, ms * 1e6 / (instr / 256), instr * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
  const int npix = 240 * 128 * 5 * 8;                 // 8 frames x 5 views of a 240 x 128 map, 64 B per pixel = 78 MB
  float4* map; float* out;
  hipMalloc(&map, size_t(npix) * 64); hipMalloc(&out, 256 * 4 * 512 * 4);
  hipMemset(map, 0, size_t(npix) * 64);
  run<4, 0>("A: 16 x 64 B segments per instruction", map, npix, out);
  run<8, 0>("B:  8 x 128 B segments, aligned", map, npix, out);
  run<8, 1>("C:  8 x 128 B segments, odd pixel (2 lines)", map, npix, out);
  run<16, 0>("D:  4 x 256 B segments", map, npix, out);
  return 0;
}
