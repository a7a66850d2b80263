// Shared host-side helpers for the C ABI (include/fvp.h).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fvp.h"

// Make a wave-uniform integer opaque to the optimiser at this point (stops loop-invariant code
// motion from hoisting dozens of derived LDS addresses out of a hot loop into live registers).
#ifndef FVP_OPAQUE
#define FVP_OPAQUE(x) asm volatile("" : "+s"(x))
// two wave-uniform values made opaque together, optionally ordered after the computation of a vector value
// the same for a per-lane value (keeps loop-invariant unpacking of a packed register out of the live set)
#define FVP_OPAQUE_V(x) asm volatile("" : "+v"(x))
#define FVP_OPAQUE_PAIR(s0, s1) asm volatile("" : "+s"(s0), "+s"(s1))
#define FVP_OPAQUE_PAIR_AFTER(s0, s1, vdep) asm volatile("" : "+s"(s0), "+s"(s1) : "v"(vdep))
#endif

// The shipped library reads NO environment variable.  Every kernel-selection, tuning and ablation switch documented in
// DESIGN.md exists only in the diagnostics build (-DFVP_DIAG=1 -> tests/diag/libfvp_hip_diag.so, loaded by tests and
// tools, never by the package): in the product every switch has its default value and the ablation masks handed to the
// kernels are 0, so a stray variable in a user's shell cannot change a result.
#ifndef FVP_DIAG
#define FVP_DIAG 0
#endif
#if FVP_DIAG
#include <cstdlib>
#endif

namespace fvp {

inline const char* diag_env(const char* name) {
#if FVP_DIAG
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Conv epilogue affine (bias, then eval-BatchNorm scale/shift) with a pinned operation order so
// that every kernel variant produces the same bits: one rounding for acc + bias, one fma.
__device__ __forceinline__ float bn_affine(float acc, float bias, float scale, float shift) {
  return __fmaf_rn(acc + bias, scale, shift);
}

// s_waitcnt vmcnt(n), n = 0..63, with lgkmcnt / expcnt left alone (gfx9 simm16: vmcnt[3:0] | expcnt 7<<4 | lgkmcnt 15<<8 |
// vmcnt[5:4] << 14); n is wave-uniform, the switch compiles to a scalar jump
#define FVP_VMCNT_CASE(n) case n: __builtin_amdgcn_s_waitcnt(0x0f70 | ((n) & 15) | (((n) >> 4) << 14)); break;
#define FVP_VMCNT_CASE4(n) FVP_VMCNT_CASE(n) FVP_VMCNT_CASE(n + 1) FVP_VMCNT_CASE(n + 2) FVP_VMCNT_CASE(n + 3)
#define FVP_VMCNT_CASE16(n) FVP_VMCNT_CASE4(n) FVP_VMCNT_CASE4(n + 4) FVP_VMCNT_CASE4(n + 8) FVP_VMCNT_CASE4(n + 12)
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
    FVP_VMCNT_CASE16(0) FVP_VMCNT_CASE16(16) FVP_VMCNT_CASE16(32) FVP_VMCNT_CASE16(48)
    default: __builtin_amdgcn_s_waitcnt(0x0f70); break;
  }
}

// the same for n = 0..8 in a handful of scalar instructions (the 64-way form is a cascade of ~30 compares and branches,
// paid by every wave at every chunk barrier); larger n fall through to the general form
__device__ __forceinline__ void wait_vmcnt_small(int n) {
  switch (n) {
    FVP_VMCNT_CASE4(0) FVP_VMCNT_CASE4(4) FVP_VMCNT_CASE(8)
    default: wait_vmcnt(n < 63 ? n : 63); break;
  }
}

inline hipStream_t as_stream(fvp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Kernel-class timing used by bench.py's roofline leg (fvp_prof_*).
// level 0 = off, 1 = coarse (one event pair per conv-stack run, per launch for the other classes),
// 2 = fine (one pair per conv launch; perturbs a ~100-launch step by >10 %)
int prof_level();
void prof_begin(int cls, hipStream_t s);
void prof_end(int cls, hipStream_t s, double flops, long launches);

struct ProfScope {
  int cls;
  hipStream_t s;
  double flops;
  long launches;
  bool on;
  ProfScope(int c, hipStream_t st, double f = 0.0, long n = 1, bool enable = true)
      : cls(c), s(st), flops(f), launches(n), on(enable) {
    if (on) prof_begin(cls, s);
  }
  ~ProfScope() {
    if (on) prof_end(cls, s, flops, launches);
  }
};

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : static_cast<int>(e);
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Opt a kernel into more than 64 KB of dynamic LDS.  The attribute is per (function, device): one
// flag bit per device ordinal, so a second GPU used by the same process gets its own opt-in.  The opt-in is
// remembered, not its size, so it always asks for the CU's whole 160 KB (the attribute is a cap on what a launch
// may request, not an allocation): a later launch of the same kernel with a larger footprint (another joint
// count / cube size in the same process) stays legal.
struct LdsOptIn {
  unsigned long long done = 0;   // bit d: set on device d (benign race: setting twice is harmless)
};
inline int lds_opt_in(LdsOptIn& st, const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (unsigned(dev) & 63u);
  if (__atomic_load_n(&st.done, __ATOMIC_RELAXED) & bit) return 0;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return static_cast<int>(e);
  __atomic_fetch_or(&st.done, bit, __ATOMIC_RELAXED);
  return 0;
}

}  // namespace fvp

#define FVP_REQUIRE(cond)            \
  do {                               \
    if (!(cond)) return FVP_EINVAL;  \
  } while (0)
#define FVP_LIMIT(cond)              \
  do {                               \
    if (!(cond)) return FVP_ELIMIT;  \
  } while (0)
