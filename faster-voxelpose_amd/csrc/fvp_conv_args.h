// Launch arguments and small host / device helpers shared by the conv translation units
// (fvp_conv.hip: direct and register-direct kernels; fvp_conv_wino.hip: the Winograd kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "fvp_asm.h"
#include "fvp_common.h"

namespace fvp {

// Division by a launch-constant: q = umulhi(x, magic), magic = floor(2^32/d) + 1 (exact for
// 0 <= x < 2^32/d; d == 1 is flagged by magic == 0).  Runtime integer division costs ~40 VALU
// instructions on gfx950; the index arithmetic of a workgroup used to contain ~90 of them.
__device__ __forceinline__ int fdiv(int x, unsigned magic) { return magic ? int(__umulhi(unsigned(x), magic)) : x; }
// the same without a branch or select on `magic == 0` (wave-uniform operands otherwise compile to scalar branches: the
// Winograd kernel's cursor code had one per division)
__device__ __forceinline__ int fdiv_nb(int x, unsigned magic) {
  return int(__umulhi(unsigned(x), magic) + (unsigned(x) & (0u - unsigned(magic == 0u))));
}
typedef fvp_i32x4 i32x4;

inline unsigned make_magic(int d) { return d <= 1 ? 0u : unsigned((1ull << 32) / unsigned(d)) + 1u; }

// 1: k_conv_dma's chunk DMA uses buffer addressing (no vector instruction per item and chunk), 0: per-lane global addresses
#ifndef FVP_CONV_BUF_DMA
#define FVP_CONV_BUF_DMA 1
#endif

struct ConvArgs {
  const float* src;
  float* dst;
  const float* res;
  const float* wts;   // packed [cinp][KK][coutp] (+ tap-major blocks for transposed conv)
  const float* epi;   // bias | scale | shift, each coutp
  const uint8_t* plane_valid;
  const float* zeros; // >= 16 bytes of zeros in device memory (head of the params blob)
  int valid_div;
  int planes, cin, cinp, cout, coutp;
  int H, W;           // input spatial size
  int OH, OW;         // output spatial size
  int osy, osx;       // output stride (2 for transposed conv, else 1)
  int TN, TH, TW;     // tile
  int tiles_x, tiles_y;
  int CC;             // input channels per LDS chunk (even)
  int flags;
  int ablate;         // diagnostics only (FVP_CONV_ABLATE): 1 skip input staging, 2 skip weight staging,
                      // 4 skip the MFMA loop, 8 skip the epilogue stores
  int dma;            // 1: k_conv_dma (pipelined LDS-DMA staging), needs vec
  int vec;            // 1: full-width tile with W % 4 == 0 -> 16-byte staging, margin layout
  int ntapT;          // 1, or number of transposed-conv taps (blockIdx.z)
  int tapT_w;         // taps along x for the transposed conv (2), 1-D: 2, rows: ntapT / tapT_w
  unsigned m_w, m_thw, m_qpr, m_rpc, m_thp;   // fdiv magics: TW, TH*TW, W/4, TN*(TH+KH-1), TH+KH-1
  int tpp, tpr;       // Winograd: 2x2 tiles per plane band of a workgroup, tiles per row
  int wino_ni;        // Winograd: input DMA rounds (of 512 x 16 B) per chunk
  int nunits, ysplit; // Winograd: work units (plane group x row band x cout block), cout blocks
  unsigned m_ys, m_ty; // fdiv magics: ysplit, tiles_y
  unsigned long long* dbg;   // Winograd, diagnostics build (-DFVP_WINO_TIMING=1): phase cycle sums
  float* pool_dst;    // Winograd: if set, max_pool(2,2) of the output is written here too (one value per 2x2 tile)
  int wrow;           // k_conv_dma: floats per packed weight row (coutp, or 2*coutp for the paired transposed conv)
  unsigned m_tpp, m_tpr;
  // paired transposed conv with 32 couts: the 1x1 conv that consumes its output (P2PNet's output layer,
  // cnns_2d.py:142) applied in the epilogue; the 32-channel map itself is then not stored
  const float* w2;    // packed [32][coutp2 = 32] weights of that conv
  const float* epi2;  // its bias | scale | shift
  float* dst2;        // its output [planes][cout2][OH][OW]
  int cout2, flags2;
  int nflags;         // Winograd: entries of plane_valid staged in LDS (0: every unit is valid)
  unsigned m_vd;      // Winograd: fdiv magic of valid_div
  int epi_off;        // k_conv_dma: float offset of the BN vectors' copy in dynamic LDS (behind slots and epilogue scratch)
};


// Kernel arguments re-read at the point of use.  A by-value kernel argument is an invariant load from the kernarg
// segment: hipcc hoists all of them to the top of the kernel and keeps ~60 SGPRs alive across the K loop (the spills of
// round 4).  Behind an opaque copy of the segment pointer the loads stay where the source puts them.
#if defined(HIPEMU)
#define FVP_FRESH_ARGS(a) (&(a))
typedef const ConvArgs* KArgsPtr;
#else
typedef const __attribute__((address_space(4))) ConvArgs* KArgsPtr;
__device__ __forceinline__ KArgsPtr fresh_args_ptr() {
  KArgsPtr p = (KArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
#define FVP_FRESH_ARGS(a) fresh_args_ptr()
#endif

// The LDS-DMA of k_conv_dma / k_conv_wino addresses a work unit's input and weights with 32-bit BYTE offsets against a
// raw buffer descriptor whose num_records is 0x7ffffff0: per-lane offset (up to (TN + 1) planes of the plane group) plus
// the scalar chunk offset (up to one plane).  An offset that wrapped or failed the range check would make the hardware
// write zeros - a silently wrong result - so shapes outside the range are refused with FVP_ELIMIT by the planners.
inline bool buf_dma_range_ok(int TN, int cin, int h, int w, double weight_floats) {
  const double lim = double(0x7ffffff0u);
  return (double(TN) + 2.0) * cin * h * w * 4.0 < lim && weight_floats * 4.0 < lim;
}

// diagnostics-build knob (FVP_* environment variable); the product compiles diag_env() to nullptr
inline size_t env_size(const char* name, size_t dflt) {
  const char* v = fvp::diag_env(name);
  return v ? size_t(atol(v)) : dflt;
}

// one persistent workgroup per CU (FVP_WINO_WGS overrides the count in the diagnostics build)
int persistent_workgroups();

// ---- Winograd F(2x2,3x3) path (fvp_conv_wino.hip) ----
// shapes the Winograd kernel takes (a SHAPE rule, never the number of planes)
bool wino_shape_ok(int h, int w, int cinp, int coutp);
// plans and launches one 3x3 conv; `a` carries src / dst / res / epi / plane_valid / shape / flags / pool_dst
int wino_plan_and_launch(const FvpConvOp& op, ConvArgs a, const float* params, int planes, hipStream_t s);
// state_dict weight [cout][cin][3][3] -> Winograd-domain copy at params + op.wino_off
int wino_pack(const float* weight, const FvpConvOp& op, float* params, hipStream_t s);

}  // namespace fvp
