// Micro-test (diagnostics): buffer_load_dwordx4 ... lds on gfx950 - does it exist, where do the 16 bytes of lane l land,
// and what do lanes whose offset fails the range check write?   hipcc --offload-arch=gfx950 -O3 -o buflds.bin buflds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) k(const float* src, float* out, unsigned nbytes, unsigned soff) {
  __shared__ float lds[1024];
  const int t = threadIdx.x;
  for (int i = t; i < 1024; i += 64) lds[i] = -1.0f;
  __syncthreads();
  // raw buffer: base, stride 0, num_records = nbytes, dword3 = 0x00020000 (CK's gfx94x / gfx950 value)
  i32x4 rsrc;
  const uint64_t b = reinterpret_cast<uint64_t>(src);
  rsrc[0] = __builtin_amdgcn_readfirstlane(int(uint32_t(b)));
  rsrc[1] = __builtin_amdgcn_readfirstlane(int(uint32_t(b >> 32) & 0xffff));
  rsrc[2] = __builtin_amdgcn_readfirstlane(int(nbytes));
  rsrc[3] = __builtin_amdgcn_readfirstlane(0x00020000);
  // lanes 0..47: valid offsets (16 B each, reversed order to see the lane -> LDS mapping); 48..55: offset 0x80000000;
  // 56..63: just past num_records
  unsigned voff = (47 - t) * 16;
  if (t >= 48) voff = 0x80000000u;
  if (t >= 56) voff = nbytes - 8;
  const unsigned la = __builtin_amdgcn_readfirstlane(unsigned(size_t((__attribute__((address_space(3))) float*)lds)) + 256 * 4);
  const unsigned so = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_waitcnt vmcnt(0)"
               :
               : "s"(la), "v"(voff), "s"(rsrc), "s"(so)
               : "memory", "m0");
  __syncthreads();
  for (int i = t; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
  const int n = 4096;
  float *src, *out, h[1024], hs[n];
  for (int i = 0; i < n; ++i) hs[i] = float(i);
  hipMalloc(&src, n * sizeof(float));
  hipMalloc(&out, 1024 * sizeof(float));
  hipMemcpy(src, hs, n * sizeof(float), hipMemcpyHostToDevice);
  for (unsigned soff : {0u, 1024u}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, 2048u * 4u, soff);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("soffset %u: lds[252..] (written region starts at 256):\n", soff);
    for (int l : {0, 1, 46, 47, 48, 55, 56, 63}) printf("  lane %2d -> lds[%4d..] = %g %g %g %g\n", l, 256 + 4 * l, h[256 + 4 * l], h[257 + 4 * l], h[258 + 4 * l], h[259 + 4 * l]);
    printf("  before %g after %g\n", h[255], h[256 + 256]);
  }
  return 0;
}
