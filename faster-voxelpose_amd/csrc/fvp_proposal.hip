// Proposal path (gfx950): 3x3 NMS + exact top-k, gathers at the selected cells, z arg-max and
// proposal packing.  Integer / index results are bit-exact with the reference
// (lib/core/proposal.py:13-33, lib/models/human_detection_net.py:44-65, :85-102).
// Tie rule (torch.topk leaves ties unspecified): value descending, then lowest flat index.
#include <hip/hip_runtime.h>

#include "fvp_common.h"

namespace fvp {

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
  return v > bv || (v == bv && i < bi);
}

// One workgroup (1024 threads) per frame.  The map is copied to LDS once (coalesced), the 3x3 NMS reads its nine
// neighbours from there, then N rounds of block arg-max over the kept values.  REG = true (X * Y <= 16 384, every
// shipped config): every thread owns <= 16 cells in registers, a round is a register scan + shuffle tree + one
// 16-entry LDS stage.  REG = false (larger maps, up to the LDS limit of ~40 000 cells, e.g. 200 x 200): the keep
// decisions are taken for all cells first (one bit per cell in two registers), then the map is overwritten in
// place by keep * x and the rounds scan the thread's cells in LDS.  Same results either way.
constexpr int kNmsThreads = 1024, kNmsCells = 16, kNmsMaxCellsLds = 64;
template <bool REG>
__global__ void __launch_bounds__(kNmsThreads)
k_nms_topk(const float* __restrict__ hm, int X, int Y, int N, float* __restrict__ vals, long long* __restrict__ idx,
           long long* __restrict__ flat) {
  HIP_DYNAMIC_SHARED(float, raw)                // [X*Y] raw map | 16 values | 16 indices
  const int b = blockIdx.x, t = threadIdx.x, n = X * Y;
  const float* m = hm + size_t(b) * n;
  float* wv = raw + n;
  int* wi = reinterpret_cast<int*>(raw + n + 16);
  for (int i = t; i < n; i += kNmsThreads) raw[i] = m[i];
  __syncthreads();
  // keep = (x == max).float(); keep * x   (core/proposal.py:23-25)
  auto kept = [&](int i) {
    const int x = i / Y, y = i - x * Y;
    const float cv = raw[i];
    float mx = cv;
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy) {
        const int xx = x + dx, yy = y + dy;
        if (xx >= 0 && xx < X && yy >= 0 && yy < Y) mx = fmaxf(mx, raw[xx * Y + yy]);
      }
    return cv == mx;
  };
  // kept value of this thread's cells i = t + 1024 c
  float kv[kNmsCells];
  if (REG) {
#pragma unroll
    for (int c = 0; c < kNmsCells; ++c) {
      const int i = t + c * kNmsThreads;
      kv[c] = -INFINITY;
      if (i < n) kv[c] = __fmul_rn(kept(i) ? 1.0f : 0.0f, raw[i]);
    }
  } else {
    unsigned long long bits = 0;                // host: n <= kNmsMaxCellsLds * 1024
    for (int c = 0, i = t; i < n; ++c, i += kNmsThreads) bits |= (unsigned long long)(kept(i) ? 1 : 0) << c;
    __syncthreads();                            // every neighbourhood has been read: overwrite in place
    for (int c = 0, i = t; i < n; ++c, i += kNmsThreads) raw[i] = __fmul_rn(((bits >> c) & 1) ? 1.0f : 0.0f, raw[i]);
    // (each thread re-reads only its own cells below: no barrier needed)
  }
  for (int k = 0; k < N; ++k) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (REG) {
#pragma unroll
      for (int c = 0; c < kNmsCells; ++c) {
        const int i = t + c * kNmsThreads;
        if (i < n && better(kv[c], i, bv, bi)) { bv = kv[c]; bi = i; }
      }
    } else {
      for (int i = t; i < n; i += kNmsThreads)
        if (better(raw[i], i, bv, bi)) { bv = raw[i]; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if ((t & 63) == 0) { wv[t >> 6] = bv; wi[t >> 6] = bi; }
    __syncthreads();
    // every thread reduces the 16 wave winners (broadcast reads) so the winner is known everywhere without a second barrier
    bv = wv[0];
    bi = wi[0];
    for (int w = 1; w < kNmsThreads / 64; ++w)
      if (better(wv[w], wi[w], bv, bi)) { bv = wv[w]; bi = wi[w]; }
    if (bi == 0x7fffffff) bi = 0;               // N > number of finite cells: degenerate, pick cell 0 (value -inf)
    if (t == 0) {
      vals[size_t(b) * N + k] = bv;
      flat[size_t(b) * N + k] = bi;
      // the reference unravels with shape[1] = X for both coordinates (core/proposal.py:16-17)
      idx[(size_t(b) * N + k) * 2 + 0] = bi / X;
      idx[(size_t(b) * N + k) * 2 + 1] = bi % X;
    }
    if (REG) {
#pragma unroll
      for (int c = 0; c < kNmsCells; ++c)
        if (t + c * kNmsThreads == bi) kv[c] = -INFINITY;   // the owner retires the winner
    } else if ((bi & (kNmsThreads - 1)) == t) {
      raw[bi] = -INFINITY;                       // the owner's own cell: visible to its next scan without a barrier
    }
    __syncthreads();                              // wv / wi are rewritten next round
  }
}

__global__ void __launch_bounds__(256)
k_gather(const float* __restrict__ bbox_map, const float* __restrict__ cubes, const long long* __restrict__ flat, int B,
         int J, int XY, int Z, int N, float* __restrict__ bbox_flat, float* __restrict__ match_bbox,
         float* __restrict__ feat1d) {
  const long i = long(blockIdx.x) * 256 + threadIdx.x;
  if (bbox_flat && i < long(B) * XY * 2) {      // [B][XY][2] <- [B][2][XY]
    const int c = int(i % 2);
    const long r = i / 2;
    const int cell = int(r % XY), b = int(r / XY);
    bbox_flat[i] = bbox_map[(size_t(b) * 2 + c) * XY + cell];
  }
  if (i < long(B) * N * 2) {
    const int c = int(i % 2);
    const long r = i / 2;
    const int b = int(r / N);
    match_bbox[i] = bbox_map[(size_t(b) * 2 + c) * XY + flat[r]];
  }
  if (cubes && i < long(B) * N * J * Z) {       // feat1d[b*N+k][j][z] = cubes[b][j][flat][z]
    const int z = int(i % Z);
    long r = i / Z;
    const int j = int(r % J);
    r /= J;
    const int b = int(r / N);
    feat1d[i] = cubes[((size_t(b) * J + j) * XY + flat[r]) * Z + z];
  }
}

__global__ void __launch_bounds__(64)
k_proposals(const float* __restrict__ hm1d, const float* __restrict__ conf2d, const long long* __restrict__ idx2d,
            const float* __restrict__ match_bbox, const float* __restrict__ sb, float min_score, int BN, int Z,
            long long* __restrict__ topk_index, float* __restrict__ centers, unsigned char* __restrict__ valid) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= BN) return;
  const float* h = hm1d + size_t(i) * Z;
  float bv = h[0];
  int bz = 0;
  for (int z = 1; z < Z; ++z)
    if (h[z] > bv) { bv = h[z]; bz = z; }       // first maximum
  const long long ix = idx2d[size_t(i) * 2], iy = idx2d[size_t(i) * 2 + 1];
  if (topk_index) {
    topk_index[size_t(i) * 3 + 0] = ix;
    topk_index[size_t(i) * 3 + 1] = iy;
    topk_index[size_t(i) * 3 + 2] = bz;
  }
  const float conf = __fmul_rn(conf2d[i], bv);  // human_detection_net.py:101
  float* c = centers + size_t(i) * 7;
  // idx.float() * scale + bias: two roundings, no fma (human_detection_net.py:49)
  c[0] = __fadd_rn(__fmul_rn(float(ix), sb[0]), sb[3]);
  c[1] = __fadd_rn(__fmul_rn(float(iy), sb[1]), sb[4]);
  c[2] = __fadd_rn(__fmul_rn(float(bz), sb[2]), sb[5]);
  c[3] = (conf > min_score ? 1.0f : 0.0f) - 1.0f;
  c[4] = conf;
  c[5] = match_bbox[size_t(i) * 2];
  c[6] = match_bbox[size_t(i) * 2 + 1];
  if (valid) valid[i] = c[3] >= 0.0f;            // faster_voxelpose.py:45 (mask = proposal_centers[:, :, 3] >= 0)
}

// ProposalLayer.forward on its own (human_detection_net.py:44-65, eval branch): same arithmetic as the tail of k_proposals
__global__ void __launch_bounds__(64)
k_proposal_layer(const long long* __restrict__ topk_index, const float* __restrict__ topk_confs,
                 const float* __restrict__ match_bbox, const float* __restrict__ sb, float min_score, int BN,
                 float* __restrict__ centers) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= BN) return;
  const float conf = topk_confs[i];
  float* c = centers + size_t(i) * 7;
#pragma unroll
  for (int a = 0; a < 3; ++a) c[a] = __fadd_rn(__fmul_rn(float(topk_index[size_t(i) * 3 + a]), sb[a]), sb[3 + a]);
  c[3] = (conf > min_score ? 1.0f : 0.0f) - 1.0f;
  c[4] = conf;
  c[5] = match_bbox[size_t(i) * 2];
  c[6] = match_bbox[size_t(i) * 2 + 1];
}

}  // namespace fvp

using namespace fvp;

extern "C" int fvp_nms_topk(const float* hm2d, int B, int X, int Y, int N, float* vals, int64_t* idx, int64_t* flat,
                            fvp_stream_t s) {
  FVP_REQUIRE(hm2d && vals && idx && flat && B >= 0 && X > 0 && Y > 0 && N > 0);
  const size_t lds = size_t(X) * Y * 4 + 128;
  // limit: the map + 32 words must fit the CU's 160 KB of LDS (X * Y <= 40 928, e.g. 200 x 200)
  FVP_LIMIT(lds <= 160 * 1024 && N <= X * Y && X * Y <= kNmsMaxCellsLds * kNmsThreads);
  const bool reg = X * Y <= kNmsCells * kNmsThreads;
  auto k = reg ? &k_nms_topk<true> : &k_nms_topk<false>;
  static LdsOptIn optin[2];
  if (lds_opt_in(optin[reg], reinterpret_cast<const void*>(k), lds)) return FVP_ELIMIT;
  if (B == 0) return 0;
  ProfScope ps(FVP_K_OTHER, as_stream(s));
  hipLaunchKernelGGL(k, dim3(B), dim3(kNmsThreads), lds, as_stream(s), hm2d, X, Y, N, vals,
                     reinterpret_cast<long long*>(idx), reinterpret_cast<long long*>(flat));
  return launch_status();
}

extern "C" int fvp_gather_proposals(const float* bbox_map, const float* cubes, const int64_t* flat, int B, int J,
                                    int X, int Y, int Z, int N, float* bbox_flat, float* match_bbox, float* feat1d,
                                    fvp_stream_t s) {
  FVP_REQUIRE(bbox_map && flat && match_bbox && B >= 0 && (cubes != nullptr) == (feat1d != nullptr));
  if (B == 0) return 0;
  long total = cubes ? long(B) * N * J * Z : 0;
  if (bbox_flat && long(B) * X * Y * 2 > total) total = long(B) * X * Y * 2;
  if (long(B) * N * 2 > total) total = long(B) * N * 2;
  ProfScope ps(FVP_K_OTHER, as_stream(s));
  hipLaunchKernelGGL(k_gather, dim3(unsigned((total + 255) / 256)), dim3(256), 0, as_stream(s), bbox_map, cubes,
                     reinterpret_cast<const long long*>(flat), B, J, X * Y, Z, N, bbox_flat, match_bbox, feat1d);
  return launch_status();
}

extern "C" int fvp_proposals(const float* hm1d, const float* conf2d, const int64_t* idx2d, const float* match_bbox,
                             const float* sb, float min_score, int B, int N, int Z, int64_t* topk_index,
                             float* centers, uint8_t* valid, fvp_stream_t s) {
  FVP_REQUIRE(hm1d && conf2d && idx2d && match_bbox && sb && centers && B >= 0 && N > 0 && Z > 0);
  if (B == 0) return 0;
  ProfScope ps(FVP_K_OTHER, as_stream(s));
  hipLaunchKernelGGL(k_proposals, dim3(ceil_div(B * N, 64)), dim3(64), 0, as_stream(s), hm1d, conf2d,
                     reinterpret_cast<const long long*>(idx2d), match_bbox, sb, min_score, B * N, Z,
                     reinterpret_cast<long long*>(topk_index), centers, valid);
  return launch_status();
}

extern "C" int fvp_proposal_layer(const int64_t* topk_index, const float* topk_confs, const float* match_bbox,
                                  const float* sb, float min_score, int B, int N, float* centers, fvp_stream_t s) {
  FVP_REQUIRE(topk_index && topk_confs && match_bbox && sb && centers && B >= 0 && N > 0);
  if (B == 0) return 0;
  ProfScope ps(FVP_K_OTHER, as_stream(s));
  hipLaunchKernelGGL(k_proposal_layer, dim3(ceil_div(B * N, 64)), dim3(64), 0, as_stream(s),
                     reinterpret_cast<const long long*>(topk_index), topk_confs, match_bbox, sb, min_score, B * N, centers);
  return launch_status();
}
