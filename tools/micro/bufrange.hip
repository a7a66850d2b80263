// Raw-buffer range check on gfx950 for ORDINARY buffer loads / stores (k_conv_reg): is the scalar offset part of it?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/bufrange.hip -o tools/micro/bufrange.bin && tools/micro/bufrange.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_store(float* dst, unsigned nrec, unsigned so) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(dst, 0, int(nrec), 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 1.0f + threadIdx.x), r, int(threadIdx.x * 4), int(so), 0);
}
__global__ void k_load(const float* src, float* out, unsigned nrec, unsigned so) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, int(nrec), 0x00020000);
  out[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, int(threadIdx.x * 4), int(so), 0));
}
// __builtin_amdgcn_raw_buffer_load_b64 in hipcc of ROCm 7.2: both elements come back as element 0 (it lowers to the i32 intrinsic)
__global__ void k_load64(const float* src, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4096 * 4, 0x00020000);
  auto v = __builtin_amdgcn_raw_buffer_load_b64(r, int(threadIdx.x * 8), 0, 0);
  out[threadIdx.x] = __builtin_bit_cast(float, v[1]) - __builtin_bit_cast(float, v[0]);
}
int main() {
  float *d, *o;
  hipMalloc(&d, 4096 * 4);
  hipMalloc(&o, 64 * 4);
  std::vector<float> h(4096), ho(64);
  // store: 64 lanes x 4 B at voffset 0..252, soffset 1024, num_records 1152 (= soffset + 128): lanes 0..31 inside iff
  // the scalar offset counts; num_records 128: lanes 0..31 inside iff it does NOT count
  for (unsigned nrec : {1152u, 128u}) {
    hipMemset(d, 0, 4096 * 4);
    k_store<<<1, 64>>>(d, nrec, 1024);
    hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
    int n = 0, last = -1;
    for (int i = 0; i < 64; ++i) if (h[256 + i] != 0.f) { ++n; last = i; }
    printf("store  num_records %4u soffset 1024: %2d lanes written (last lane %d)\n", nrec, n, last);
  }
  for (int i = 0; i < 4096; ++i) h[i] = float(i);
  hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (unsigned nrec : {1152u, 128u}) {
    k_load<<<1, 64>>>(d, o, nrec, 1024);
    hipMemcpy(ho.data(), o, 64 * 4, hipMemcpyDeviceToHost);
    int n = 0, last = -1;
    for (int i = 0; i < 64; ++i) if (ho[i] != 0.f) { ++n; last = i; }
    printf("load   num_records %4u soffset 1024: %2d lanes non-zero (last lane %d, lane 0 = %.0f)\n", nrec, n, last, ho[0]);
  }
  k_load64<<<1, 64>>>(d, o);
  hipMemcpy(ho.data(), o, 64 * 4, hipMemcpyDeviceToHost);
  printf("raw_buffer_load_b64: element 1 - element 0 = %.0f (1 expected; 0 = the builtin returned element 0 twice)\n", ho[5]);
  return 0;
}
