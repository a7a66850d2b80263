// Device-side geometry shared by every back-projection kernel.
//
// One voxel centre (wx,wy,wz) -> normalised grid_sample coordinate in one view, with
// exactly the fp32 operation order of the reference chain
//   cameras.project_point (lib/utils/cameras.py:30-56)
//   -> clamp [-1, max(ori)] (lib/models/project_whole.py:51)
//   -> affine_transform_pts_cuda (lib/utils/transforms.py:59-63)
//   -> * [w,h] / IMAGE_SIZE, / [w-1,h-1] * 2 - 1, clamp +-1.1 (project_whole.py:53-59).
// The two small matrix products (R (x-T), the 2x3 affine) are k-ordered fma chains, which
// is what the reference's sgemm produces on the CPU (checked bit-for-bit in the build
// container); every other step is a separately rounded IEEE op, hence the
// explicit __f*_rn intrinsics (no contraction).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fvp.h"

namespace fvp {

struct Cam {  // FVP_CAM_FLOATS layout
  float R[9], T[3], f[2], c[2], k[3], p[2], pad[3];
};
static_assert(sizeof(Cam) == FVP_CAM_FLOATS * sizeof(float), "camera record size");

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__device__ __forceinline__ void project_norm(const Cam& cm, const FvpGeom& g, float wx, float wy, float wz,
                                             float& gx, float& gy) {
  const float d0 = __fsub_rn(wx, cm.T[0]), d1 = __fsub_rn(wy, cm.T[1]), d2 = __fsub_rn(wz, cm.T[2]);
  const float xc0 = __fmaf_rn(cm.R[2], d2, __fmaf_rn(cm.R[1], d1, __fmul_rn(cm.R[0], d0)));
  const float xc1 = __fmaf_rn(cm.R[5], d2, __fmaf_rn(cm.R[4], d1, __fmul_rn(cm.R[3], d0)));
  const float xc2 = __fmaf_rn(cm.R[8], d2, __fmaf_rn(cm.R[7], d1, __fmul_rn(cm.R[6], d0)));
  const float den = __fadd_rn(xc2, 1e-5f);
  const float y0 = __fdiv_rn(xc0, den), y1 = __fdiv_rn(xc1, den);
  const float r = __fadd_rn(__fmul_rn(y0, y0), __fmul_rn(y1, y1));
  // d = 1 + k0 r + k1 r r + k2 r r r   (left to right)
  float d = __fadd_rn(1.0f, __fmul_rn(cm.k[0], r));
  d = __fadd_rn(d, __fmul_rn(__fmul_rn(cm.k[1], r), r));
  d = __fadd_rn(d, __fmul_rn(__fmul_rn(__fmul_rn(cm.k[2], r), r), r));
  // u = y0 d + 2 p0 y0 y1 + p1 (r + 2 y0 y0) ;  v = y1 d + 2 p1 y0 y1 + p0 (r + 2 y1 y1)
  float u = __fadd_rn(__fmul_rn(y0, d), __fmul_rn(__fmul_rn(__fmul_rn(2.0f, cm.p[0]), y0), y1));
  u = __fadd_rn(u, __fmul_rn(cm.p[1], __fadd_rn(r, __fmul_rn(__fmul_rn(2.0f, y0), y0))));
  float v = __fadd_rn(__fmul_rn(y1, d), __fmul_rn(__fmul_rn(__fmul_rn(2.0f, cm.p[1]), y0), y1));
  v = __fadd_rn(v, __fmul_rn(cm.p[0], __fadd_rn(r, __fmul_rn(__fmul_rn(2.0f, y1), y1))));
  float px = __fadd_rn(__fmul_rn(cm.f[0], u), cm.c[0]);
  float py = __fadd_rn(__fmul_rn(cm.f[1], v), cm.c[1]);
  // torch.clamp(x, lo, hi) = min(max(x, lo), hi); NaN is undefined behaviour in the reference
  px = clampf(px, -1.0f, g.clamp_max);
  py = clampf(py, -1.0f, g.clamp_max);
  const float ax = __fmaf_rn(g.rt[2], 1.0f, __fmaf_rn(g.rt[1], py, __fmul_rn(g.rt[0], px)));
  const float ay = __fmaf_rn(g.rt[5], 1.0f, __fmaf_rn(g.rt[4], py, __fmul_rn(g.rt[3], px)));
  float sx = __fdiv_rn(__fmul_rn(ax, g.hm_w), g.img_w);
  float sy = __fdiv_rn(__fmul_rn(ay, g.hm_h), g.img_h);
  sx = __fsub_rn(__fmul_rn(__fdiv_rn(sx, __fsub_rn(g.hm_w, 1.0f)), 2.0f), 1.0f);
  sy = __fsub_rn(__fmul_rn(__fdiv_rn(sy, __fsub_rn(g.hm_h, 1.0f)), 2.0f), 1.0f);
  gx = clampf(sx, -1.1f, 1.1f);
  gy = clampf(sy, -1.1f, 1.1f);
}

// Bilinear taps of F.grid_sample(align_corners=True, padding 'zeros'): tap order nw, ne, sw,
// se; weights are the opposite-corner areas.  `inside` bit i set <=> tap i is in the image.
struct Taps {
  int off[4];    // element offsets (y*W + x) into one channels-last plane, valid where inside
  float w[4];
  int inside;
};

__device__ __forceinline__ Taps bilinear_taps(float gx, float gy, int W, int H) {
  const float ix = __fmul_rn(__fadd_rn(gx, 1.0f), 0.5f * float(W - 1));
  const float iy = __fmul_rn(__fadd_rn(gy, 1.0f), 0.5f * float(H - 1));
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float x1f = __fadd_rn(x0f, 1.0f), y1f = __fadd_rn(y0f, 1.0f);
  Taps t;
  t.w[0] = __fmul_rn(__fsub_rn(x1f, ix), __fsub_rn(y1f, iy));
  t.w[1] = __fmul_rn(__fsub_rn(ix, x0f), __fsub_rn(y1f, iy));
  t.w[2] = __fmul_rn(__fsub_rn(x1f, ix), __fsub_rn(iy, y0f));
  t.w[3] = __fmul_rn(__fsub_rn(ix, x0f), __fsub_rn(iy, y0f));
  const int x0 = int(x0f), y0 = int(y0f), x1 = x0 + 1, y1 = y0 + 1;
  const bool x0in = x0 >= 0 && x0 < W, x1in = x1 >= 0 && x1 < W;
  const bool y0in = y0 >= 0 && y0 < H, y1in = y1 >= 0 && y1 < H;
  t.inside = (x0in && y0in ? 1 : 0) | (x1in && y0in ? 2 : 0) | (x0in && y1in ? 4 : 0) | (x1in && y1in ? 8 : 0);
  t.off[0] = y0 * W + x0;
  t.off[1] = y0 * W + x1;
  t.off[2] = y1 * W + x0;
  t.off[3] = y1 * W + x1;
  return t;
}

// acc[c] (c < 4*NV) += bilinear sample of channels-last plane `cl` ([H*W][JP], JP = 4*NV)
// with the reference's accumulation order: nw*w, then fma(ne), fma(sw), fma(se).
template <int NV>
__device__ __forceinline__ void sample_view(const float* __restrict__ cl, const Taps& t, float (&out)[4 * NV]) {
  float4 v[4][NV];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = (t.inside >> k) & 1;
    const float4* p = reinterpret_cast<const float4*>(cl + size_t(in ? t.off[k] : 0) * (4 * NV));
#pragma unroll
    for (int q = 0; q < NV; ++q) v[k][q] = in ? p[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    float a0 = __fmul_rn(v[0][q].x, t.w[0]), a1 = __fmul_rn(v[0][q].y, t.w[0]);
    float a2 = __fmul_rn(v[0][q].z, t.w[0]), a3 = __fmul_rn(v[0][q].w, t.w[0]);
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      a0 = __fmaf_rn(v[k][q].x, t.w[k], a0);
      a1 = __fmaf_rn(v[k][q].y, t.w[k], a1);
      a2 = __fmaf_rn(v[k][q].z, t.w[k], a2);
      a3 = __fmaf_rn(v[k][q].w, t.w[k], a3);
    }
    out[4 * q + 0] = a0;
    out[4 * q + 1] = a1;
    out[4 * q + 2] = a2;
    out[4 * q + 3] = a3;
  }
}

}  // namespace fvp
