// Library-level entry points of the C ABI: version, error strings and the per-kernel-class
// HIP-event timing used by bench.py's roofline leg.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "fvp_common.h"

namespace fvp {

struct ProfState {
  std::mutex mu;
  int level = 0;
  struct Pair { hipEvent_t a, b; int cls; double flops; long launches; };
  std::vector<Pair> open;      // recorded, not yet read back
  std::vector<hipEvent_t> pool;
  double ms[FVP_K_COUNT] = {0};
  long long launches[FVP_K_COUNT] = {0};
  double flops[FVP_K_COUNT] = {0};
  hipEvent_t pending[FVP_K_COUNT] = {nullptr};
};
static ProfState g_prof;

static hipEvent_t get_event() {
  if (!g_prof.pool.empty()) {
    hipEvent_t e = g_prof.pool.back();
    g_prof.pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

int prof_level() { return g_prof.level; }

void prof_begin(int cls, hipStream_t s) {
  if (!g_prof.level) return;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  hipEvent_t a = get_event();
  (void)hipEventRecord(a, s);
  g_prof.pending[cls] = a;
}

void prof_end(int cls, hipStream_t s, double flops, long launches) {
  if (!g_prof.level) return;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  if (!g_prof.pending[cls]) return;
  hipEvent_t b = get_event();
  (void)hipEventRecord(b, s);
  g_prof.open.push_back({g_prof.pending[cls], b, cls, flops, launches});
  g_prof.pending[cls] = nullptr;
}

static void drain() {
  for (auto& p : g_prof.open) {
    (void)hipEventSynchronize(p.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, p.a, p.b);
    g_prof.ms[p.cls] += ms;
    g_prof.launches[p.cls] += p.launches;
    g_prof.flops[p.cls] += p.flops;
    g_prof.pool.push_back(p.a);
    g_prof.pool.push_back(p.b);
  }
  g_prof.open.clear();
}

}  // namespace fvp

using namespace fvp;

extern "C" int fvp_version(void) { return FVP_ABI_VERSION; }
extern "C" int fvp_diag_build(void) { return FVP_DIAG; }

extern "C" int fvp_sizeof(int what) {
  return what == 0 ? int(sizeof(FvpGeom)) : what == 1 ? int(sizeof(FvpConvOp)) : what == 2 ? int(sizeof(FvpBbOp)) : FVP_EINVAL;
}

extern "C" const char* fvp_error_string(int code) {
  switch (code) {
    case 0: return "success";
    case FVP_EINVAL: return "fvp: invalid argument (null pointer or inconsistent sizes)";
    case FVP_ELIMIT: return "fvp: size outside the compiled limits of this kernel";
    default: return hipGetErrorString(static_cast<hipError_t>(code));
  }
}

extern "C" int fvp_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.level = on < 0 ? 0 : on;
  return 0;
}

extern "C" int fvp_prof_read(int cls, double* ms, int64_t* launches, double* flops) {
  if (cls < 0 || cls >= FVP_K_COUNT) return FVP_EINVAL;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  drain();
  if (ms) *ms = g_prof.ms[cls];
  if (launches) *launches = g_prof.launches[cls];
  if (flops) *flops = g_prof.flops[cls];
  return 0;
}

extern "C" int fvp_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  drain();
  for (int i = 0; i < FVP_K_COUNT; ++i) {
    g_prof.ms[i] = 0;
    g_prof.launches[i] = 0;
    g_prof.flops[i] = 0;
  }
  return 0;
}
