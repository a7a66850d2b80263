/*
 * hipemu - a minimal HIP execution-model emulator for the CPU.  TEST INFRASTRUCTURE ONLY.
 *
 * The build container has no GPU.  To check the *logic* of the gfx950 kernels in
 * faster-voxelpose_amd/csrc (indexing, LDS tiling, barriers, wave shuffles, MFMA fragment
 * layouts) before spending GPU minutes, the unmodified .hip sources are compiled a second
 * time with the host clang against this header (it shadows <hip/hip_runtime.h>) into
 * tests/hipemu/libfvp_emu.so.  Only `-m "not gpu"` tests load that library, by explicit
 * path, with numpy buffers standing in for device memory.  The product package never
 * loads it and has no CPU execution path.
 *
 * Model: each workgroup runs as blockDim cooperative fibers on one OS thread (workgroups
 * are spread over OS threads); __syncthreads and the wave-collective operations
 * (__shfl*, MFMA) are rendezvous points.  `__shared__` becomes `static thread_local`.
 * v_mfma_f32_32x32x2_f32 follows the lane layout documented for gfx950: A[i=l&31][k=l>>5],
 * B[k=l>>5][j=l&31], D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), computed as the
 * k-ordered fmaf chain the hardware produces.
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define HIPEMU 1
#define FVP_OPAQUE(x) ((void)0)
#define FVP_OPAQUE_V(x) ((void)0)
#define FVP_OPAQUE_PAIR(s0, s1) ((void)0)
#define FVP_OPAQUE_PAIR_AFTER(s0, s1, vdep) ((void)0)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::dyn_smem());
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };

namespace hipemu {
struct Ctx {
  dim3 tid, bid, bdim, gdim;
  int lane, wave;
};
extern thread_local Ctx* cur;
char* dyn_smem();
void block_barrier();
// wave rendezvous: every lane deposits `n` 32-bit words, then may read any lane's words
void wave_exchange(const uint32_t* mine, int n, uint32_t (*all)[4]);
void wave_exchange_done();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::cur->bid)
#define blockDim (hipemu::cur->bdim)
#define gridDim (hipemu::cur->gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::block_barrier(); }

template <typename... KArgs, typename... Args>
static inline void hipLaunchKernelGGL(void (*k)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t,
                                      Args... args) {
  hipemu::launch(grid, block, shmem, [=]() { k(args...); });
}
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
// a 3-CU device: persistent kernels walk several work units per workgroup in the emulated runs
enum { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 3; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }

// ---- cross-lane -------------------------------------------------------------------------
template <typename T>
static inline T hipemu_shfl_from(T v, int src) {
  static_assert(sizeof(T) == 4, "32-bit shuffles only");
  uint32_t mine[1];
  memcpy(mine, &v, 4);
  uint32_t all[64][4];
  hipemu::wave_exchange(mine, 1, all);
  T r;
  memcpy(&r, &all[src & 63][0], 4);
  hipemu::wave_exchange_done();
  return r;
}
template <typename T> static inline T __shfl_xor(T v, int m, int = 64) { return hipemu_shfl_from(v, hipemu::cur->lane ^ m); }
template <typename T> static inline T __shfl_down(T v, int d, int = 64) {
  int s = hipemu::cur->lane + d;
  return hipemu_shfl_from(v, s > 63 ? hipemu::cur->lane : s);
}
template <typename T> static inline T __shfl(T v, int src, int = 64) { return hipemu_shfl_from(v, src); }

typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x16 hipemu_mfma_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
  uint32_t mine[2];
  memcpy(&mine[0], &a, 4);
  memcpy(&mine[1], &b, 4);
  uint32_t all[64][4];
  hipemu::wave_exchange(mine, 2, all);
  const int l = hipemu::cur->lane, col = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      memcpy(&av, &all[row + 32 * k][0], 4);
      memcpy(&bv, &all[col + 32 * k][1], 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  hipemu::wave_exchange_done();
  return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_32x32x2f32
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D reg r: row 4*(l>>4)+r, col l&15
static inline hipemu_f32x4 hipemu_mfma_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
  uint32_t mine[2];
  memcpy(&mine[0], &a, 4);
  memcpy(&mine[1], &b, 4);
  uint32_t all[64][4];
  hipemu::wave_exchange(mine, 2, all);
  const int l = hipemu::cur->lane, col = l & 15, g = l >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * g + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, &all[row + 16 * k][0], 4);
      memcpy(&bv, &all[col + 16 * k][1], 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  hipemu::wave_exchange_done();
  return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_16x16x4f32
// lanes of a wave run in lock-step on the hardware; the emulator's fibers do not, so code that passes
// data between lanes of one wave through LDS marks the hand-over with a wave barrier
static inline void hipemu_wave_sync() {
  uint32_t mine[1] = {0}, all[64][4];
  hipemu::wave_exchange(mine, 1, all);
  hipemu::wave_exchange_done();
}
#define __builtin_amdgcn_wave_barrier() hipemu_wave_sync()
// v_mfma_f32_32x32x16_bf16: A[i=l&31][k = 8*(l>>5) + e], B[k][j=l&31], 8 bf16 per lane and operand
// (passed as four 32-bit words); D as the f32 32x32 form.  Products are exact in fp32; the
// accumulation order of the hardware is not specified -- k-ascending fp32 adds here.
static inline hipemu_f32x16 hipemu_mfma_32x32x16_bf16(const uint32_t* a, const uint32_t* b, hipemu_f32x16 c) {
  uint32_t all[2][64][4];                              // the exchange carries 4 words per lane: A, then B
  hipemu::wave_exchange(a, 4, all[0]);
  hipemu::wave_exchange_done();
  hipemu::wave_exchange(b, 4, all[1]);
  const int l = hipemu::cur->lane, col = l & 31, hi = l >> 5;
  auto elem = [&](int lane, int which, int e) {
    const uint32_t w = all[which][lane][e >> 1];
    const uint32_t h = (e & 1) ? (w >> 16) : (w & 0xffffu);
    const uint32_t u = h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
  };
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int k = 0; k < 16; ++k) acc += elem(row + 32 * (k >> 3), 0, k & 7) * elem(col + 32 * (k >> 3), 1, k & 7);
    c[r] = acc;
  }
  hipemu::wave_exchange_done();
  return c;
}

typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
static inline hipemu_f32x16 hipemu_mfma_bf16_v(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c, int, int, int) {
  uint32_t aw[4], bw[4];
  memcpy(aw, &a, 16);
  memcpy(bw, &b, 16);
  return hipemu_mfma_32x32x16_bf16(aw, bw, c);
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 hipemu_mfma_bf16_v

// LDS-DMA: wave-uniform LDS base + lane * size, per-lane global source
static inline void hipemu_glds(const __attribute__((address_space(1))) void* g, __attribute__((address_space(3))) void* l,
                               unsigned size, int off, unsigned) {
  char* dst = (char*)(void*)l + off + size_t(hipemu::cur->lane) * size;
  memcpy(dst, (const char*)(const void*)g + off, size);
}
#define __builtin_amdgcn_global_load_lds hipemu_glds

// ---- the product's inline-asm primitives (faster-voxelpose_amd/csrc/fvp_asm.h), emulated ----------------------------
// LDS byte addresses are offsets from the workgroup's dynamic LDS base.
#define FVP_ASM_PRIMITIVES_PROVIDED 1
#define FVP_LDS_BYTE_ADDRESS(arr) unsigned((const char*)(arr) - (const char*)hipemu::dyn_smem())
namespace fvp {
typedef int fvp_i32x4 __attribute__((ext_vector_type(4)));
// global_load_lds_dwordx4: lane l copies 16 bytes from its global address to LDS byte address lds_addr + 16 l
static inline void asm_global_load_lds16(const float* g, unsigned lds_addr) {
  memcpy(hipemu::dyn_smem() + lds_addr + size_t(hipemu::cur->lane) * 16, g, 16);
}
// buffer_load_dwordx4 ... offen lds (raw buffer, stride 0): lane l copies 16 bytes from base + voffset + soffset to LDS
// byte address `la` + 16 l; every dword whose offset fails the range check (offset + 4 > num_records, the SGPR offset
// included) is written as zero - what tools/micro/buflds.hip measured on gfx950.
static inline void asm_buffer_load_lds16(unsigned la, unsigned vo, const fvp_i32x4& rs, unsigned so) {
  const unsigned long long base = (unsigned long long)(unsigned)rs[0] | ((unsigned long long)((unsigned)rs[1] & 0xffffu) << 32);
  const unsigned long long nrec = (unsigned)rs[2];
  char* dst = hipemu::dyn_smem() + la + size_t(hipemu::cur->lane) * 16;
  for (int d = 0; d < 4; ++d) {
    const unsigned long long off = (unsigned long long)vo + so + 4ull * d;
    float v = 0.0f;
    if (off + 4 <= nrec) memcpy(&v, (const char*)base + off, 4);
    memcpy(dst + 4 * d, &v, 4);
  }
}
// the Winograd epilogue's asm buffer accesses: descriptor as four ints (base lo, base hi[15:0], num_records, flags)
typedef float fvp_f32x2 __attribute__((ext_vector_type(2)));
static inline char* hipemu_rs_base(const fvp_i32x4& rs) {
  return (char*)((unsigned long long)(unsigned)rs[0] | ((unsigned long long)((unsigned)rs[1] & 0xffffu) << 32));
}
static inline fvp_f32x2 asm_buffer_load_f32x2(unsigned vo, const fvp_i32x4& rs, unsigned so) {
  float t[2] = {0.0f, 0.0f};
  const unsigned long long off = (unsigned long long)vo + so, nrec = (unsigned)rs[2];
  for (int d = 0; d < 2; ++d)
    if (off + 4ull * d + 4 <= nrec) memcpy(&t[d], hipemu_rs_base(rs) + off + 4 * d, 4);
  return fvp_f32x2{t[0], t[1]};
}
static inline void asm_buffer_store_f32x2(fvp_f32x2 v, unsigned vo, const fvp_i32x4& rs, unsigned so) {
  const unsigned long long off = (unsigned long long)vo + so, nrec = (unsigned)rs[2];
  const float t[2] = {v[0], v[1]};
  for (int d = 0; d < 2; ++d)
    if (off + 4ull * d + 4 <= nrec) memcpy(hipemu_rs_base(rs) + off + 4 * d, &t[d], 4);
}
static inline void asm_buffer_store_f32(float v, unsigned vo, const fvp_i32x4& rs, unsigned so) {
  const unsigned long long off = (unsigned long long)vo + so, nrec = (unsigned)rs[2];
  if (off + 4 <= nrec) memcpy(hipemu_rs_base(rs) + off, &v, 4);
}
// raw buffer loads / stores (k_conv_reg): dword-granular range check against num_records, scalar offset included
struct fvp_rsrc {
  char* base;
  unsigned long long nrec;
};
static inline fvp_rsrc make_rsrc(const void* p, unsigned bytes) { return fvp_rsrc{(char*)p, bytes}; }
static inline float buf_load_f32(fvp_rsrc r, unsigned vo, unsigned so) {
  const unsigned long long off = (unsigned long long)vo + so;
  float v = 0.0f;
  if (off + 4 <= r.nrec) memcpy(&v, r.base + off, 4);
  return v;
}
static inline float2 buf_load_f32x2(fvp_rsrc r, unsigned vo, unsigned so) {
  return make_float2(buf_load_f32(r, vo, so), buf_load_f32(r, vo + 4, so));
}
static inline void buf_store_f32(float x, fvp_rsrc r, unsigned vo, unsigned so) {
  const unsigned long long off = (unsigned long long)vo + so;
  if (off + 4 <= r.nrec) memcpy(r.base + off, &x, 4);
}
static inline void buf_store_f32x2(float2 x, fvp_rsrc r, unsigned vo, unsigned so) {
  buf_store_f32(x.x, r, vo, so);
  buf_store_f32(x.y, r, vo + 4, so);
}
}  // namespace fvp

#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_barrier() hipemu::block_barrier()
// DPP: quad_perm (dpp_ctrl < 0x100): lane l reads lane (l & ~3) | sel[l & 3]; row_ror:n (0x121..0x12f): lane l
// reads lane (l - n) mod 16 of its row; row_mirror (0x140) / row_half_mirror (0x141): reversed 16 / 8 lanes
static inline int hipemu_update_dpp(int old, int src, int ctrl, int, int, bool) {
  (void)old;
  const int l = hipemu::cur->lane;
  int from;
  if (ctrl < 0x100) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
  else if (ctrl >= 0x121 && ctrl <= 0x12f) from = (l & ~15) | ((l - (ctrl - 0x120)) & 15);
  else if (ctrl == 0x140) from = (l & ~15) | (15 - (l & 15));
  else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));
  else { fprintf(stderr, "hipemu: unsupported dpp_ctrl 0x%x\n", ctrl); abort(); }
  return hipemu_shfl_from(src, from);
}
#define __builtin_amdgcn_update_dpp hipemu_update_dpp
// v_permlane32_swap_b32 vdst, src: lanes 32-63 of vdst <-> lanes 0-31 of src; returns {vdst, src}
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
static inline hipemu_u32x2 hipemu_permlane32_swap(unsigned vdst, unsigned src, bool, bool) {
  const int l = hipemu::cur->lane;
  const unsigned src_lo = (unsigned)hipemu_shfl_from((int)src, l & 31);          // src of lane l - 32 (for upper lanes)
  const unsigned dst_hi = (unsigned)hipemu_shfl_from((int)vdst, (l & 31) + 32);   // vdst of lane l + 32 (for lower lanes)
  hipemu_u32x2 r;
  r[0] = l < 32 ? vdst : src_lo;
  r[1] = l < 32 ? dst_hi : src;
  return r;
}
#define __builtin_amdgcn_permlane32_swap hipemu_permlane32_swap
// the value is wave-uniform wherever the kernels use it
static inline int hipemu_readfirstlane(int v) { return v; }
#define __builtin_amdgcn_readfirstlane hipemu_readfirstlane

// ---- scalar intrinsics -------------------------------------------------------------------
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return unsigned((uint64_t(a) * b) >> 32); }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline long long __double_as_longlong(double d) { long long i; memcpy(&i, &d, 8); return i; }
static inline double __longlong_as_double(long long i) { double d; memcpy(&d, &i, 8); return d; }
static inline int atomicMax(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
