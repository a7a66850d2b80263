// Shared host-side helpers for the C ABI (include/fvp.h).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fvp.h"

namespace fvp {

inline hipStream_t as_stream(fvp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Kernel-class timing used by bench.py's roofline leg (fvp_prof_*).
void prof_begin(int cls, hipStream_t s);
void prof_end(int cls, hipStream_t s, double flops);

struct ProfScope {
  int cls;
  hipStream_t s;
  double flops;
  ProfScope(int c, hipStream_t st, double f = 0.0) : cls(c), s(st), flops(f) { prof_begin(cls, s); }
  ~ProfScope() { prof_end(cls, s, flops); }
};

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : static_cast<int>(e);
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace fvp

#define FVP_REQUIRE(cond)            \
  do {                               \
    if (!(cond)) return FVP_EINVAL;  \
  } while (0)
#define FVP_LIMIT(cond)              \
  do {                               \
    if (!(cond)) return FVP_ELIMIT;  \
  } while (0)
