// Back-projection kernels (gfx950): heatmap staging, sampling grids, whole-space cubes with
// fused z-max, per-person cubes, orthographic tri-plane maxima, and the fused
// project+tri-plane fast path.  Reference sites: lib/models/project_whole.py:49-88,
// lib/models/project_individual.py:60-136, lib/models/joint_localization_net.py:80-81,
// lib/models/cnns_2d.py:174.
//
// Data layout in HBM
//   heat     [B][V][J][H][W]        NCHW, as the reference hands it over
//   heat_cl  [B][V][H*W][JP]        channels-last, JP = 4*ceil(J/4): one bilinear tap of one
//                                   (voxel, view) is JP contiguous floats = JP/4 dwordx4 loads
//                                   shared by all joints (the reference gathers J planes)
//   cubes    [B][J][X][Y][Z]        z fastest (reference layout)
//   planes   [nP][3][J][C][C]       xy | xz | yz maxima of one person's cube
// All kernels are HBM/L2-bound gathers; lanes run along z so loads of neighbouring voxels hit
// neighbouring heatmap pixels and every store is a contiguous 256-byte row.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fvp_common.h"
#include "fvp_geom.h"

namespace fvp {

// ---------------------------------------------------------------------------------------------
// NCHW -> channels-last staging: one thread per pixel, coalesced reads per channel plane,
// JP/4 dwordx4 stores per thread.
template <int NV>
__global__ void __launch_bounds__(256) k_heat_to_cl(const float* __restrict__ heat, float* __restrict__ cl,
                                                    int J, int HW) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const size_t bv = blockIdx.y;
  if (pix >= HW) return;
  const float* src = heat + bv * size_t(J) * HW + pix;
  float v[4 * NV];
#pragma unroll
  for (int j = 0; j < 4 * NV; ++j) v[j] = (j < J) ? src[size_t(j) * HW] : 0.0f;
  float4* dst = reinterpret_cast<float4*>(cl + (bv * HW + pix) * size_t(4 * NV));
#pragma unroll
  for (int q = 0; q < NV; ++q) dst[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// ---------------------------------------------------------------------------------------------
// Sampling grid for the drop-in cache / parity tests: one thread per (view, voxel).
__global__ void __launch_bounds__(256) k_sample_grid(const float* __restrict__ ax, const float* __restrict__ ay,
                                                     const float* __restrict__ az, int nx, int ny, int nz,
                                                     const Cam* __restrict__ cams, FvpGeom g,
                                                     float* __restrict__ grid) {
  const int n = nx * ny * nz;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int v = blockIdx.y;
  if (i >= n) return;
  const int iz = i % nz, iy = (i / nz) % ny, ix = i / (nz * ny);
  float gx, gy;
  project_norm(cams[v], g, ax[ix], ay[iy], az[iz], gx, gy);
  grid[(size_t(v) * n + i) * 2 + 0] = gx;
  grid[(size_t(v) * n + i) * 2 + 1] = gy;
}

// mean over views of the bilinear samples of one world point, clamped to [0,1]
// (project_whole.py:83,86 / project_individual.py:130,134): sum views in order, divide by V.
template <int NV>
__device__ __forceinline__ void backproject_point(const float* __restrict__ heat_cl_frame,
                                                  const Cam* __restrict__ cams, const FvpGeom& g, float wx,
                                                  float wy, float wz, float (&acc)[4 * NV]) {
  const size_t view_stride = size_t(g.H) * g.W * (4 * NV);
  for (int v = 0; v < g.V; ++v) {
    float gx, gy;
    project_norm(cams[v], g, wx, wy, wz, gx, gy);
    const Taps t = bilinear_taps(gx, gy, g.W, g.H);
    float s[4 * NV];
    if (t.inside) {
      sample_view<NV>(heat_cl_frame + v * view_stride, t, s);
    } else {
#pragma unroll
      for (int c = 0; c < 4 * NV; ++c) s[c] = 0.0f;
    }
#pragma unroll
    for (int c = 0; c < 4 * NV; ++c) acc[c] = (v == 0) ? s[c] : __fadd_rn(acc[c], s[c]);
  }
  const float nv = float(g.V);
#pragma unroll
  for (int c = 0; c < 4 * NV; ++c) acc[c] = clampf(__fdiv_rn(acc[c], nv), 0.0f, 1.0f);
}

// ---------------------------------------------------------------------------------------------
// Whole-space cubes (+ fused z-max).  A workgroup owns CPB = 256/Z complete z-columns so the
// z-max never leaves the workgroup: values go through LDS [JP][256] and CPB*J threads each
// reduce one column of one joint.
template <int NV>
__global__ void __launch_bounds__(256)
k_project_whole(const float* __restrict__ heat_cl, const Cam* __restrict__ cams, const int* __restrict__ frame_set,
                const float* __restrict__ ax, const float* __restrict__ ay, const float* __restrict__ az, int X,
                int Y, int Z, FvpGeom g, float* __restrict__ cubes, float* __restrict__ zmax) {
  __shared__ float sm[4 * NV][256];
  const int cpb = 256 / Z;
  const int b = blockIdx.y;
  const int t = threadIdx.x;
  const int cl_ = t / Z, z = t - cl_ * Z;
  const int col = blockIdx.x * cpb + cl_;
  const int ncol = X * Y;
  const bool active = cl_ < cpb && col < ncol;
  const int J = g.J;
  float acc[4 * NV];
#pragma unroll
  for (int c = 0; c < 4 * NV; ++c) acc[c] = 0.0f;
  if (active) {
    const int x = col / Y, y = col - x * Y;
    const float* frame = heat_cl + size_t(b) * g.V * g.H * g.W * (4 * NV);
    backproject_point<NV>(frame, cams + size_t(frame_set[b]) * g.V, g, ax[x], ay[y], az[z], acc);
    if (cubes) {
      const size_t vox = size_t(col) * Z + z;
      const size_t nvox = size_t(ncol) * Z;
#pragma unroll
      for (int c = 0; c < 4 * NV; ++c)
        if (c < J) cubes[(size_t(b) * J + c) * nvox + vox] = acc[c];
    }
  }
  if (zmax) {
#pragma unroll
    for (int c = 0; c < 4 * NV; ++c) sm[c][t] = acc[c];
    __syncthreads();
    for (int item = t; item < cpb * J; item += 256) {
      const int c = item / cpb, k = item - c * cpb;
      const int cc = blockIdx.x * cpb + k;
      if (cc < ncol) {
        float m = sm[c][k * Z];
        for (int zz = 1; zz < Z; ++zz) m = fmaxf(m, sm[c][k * Z + zz]);
        zmax[(size_t(b) * J + c) * ncol + cc] = m;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Per-person cube, materialised (drop-in for project_individual.ProjectLayer.forward).
// boxes[p] = tl[3], start[3], end[3] in fine-grid indices (fvp_person_boxes).
template <int NV>
__global__ void __launch_bounds__(256)
k_project_individual(const float* __restrict__ heat_cl, const Cam* __restrict__ cams,
                     const int* __restrict__ frame_set, const int* __restrict__ person_frame,
                     const uint8_t* __restrict__ person_valid, const int* __restrict__ boxes,
                     const float* __restrict__ fx, const float* __restrict__ fy, const float* __restrict__ fz, int C,
                     FvpGeom g, float* __restrict__ cubes) {
  const int p = blockIdx.y;
  const int lv = blockIdx.x * 256 + threadIdx.x;
  const int nvox = C * C * C;
  if (lv >= nvox) return;
  const int J = g.J;
  const int lz = lv % C, ly = (lv / C) % C, lx = lv / (C * C);
  const int* bx = boxes + p * 9;
  const int gx_ = bx[0] + lx, gy_ = bx[1] + ly, gz_ = bx[2] + lz;
  bool in = (!person_valid || person_valid[p]);
  in = in && bx[3] < bx[6] && bx[4] < bx[7] && bx[5] < bx[8];   // empty window -> all zero (:125)
  in = in && gx_ >= bx[3] && gx_ < bx[6] && gy_ >= bx[4] && gy_ < bx[7] && gz_ >= bx[5] && gz_ < bx[8];
  float acc[4 * NV];
#pragma unroll
  for (int c = 0; c < 4 * NV; ++c) acc[c] = 0.0f;
  if (in) {
    const int b = person_frame[p];
    const float* frame = heat_cl + size_t(b) * g.V * g.H * g.W * (4 * NV);
    backproject_point<NV>(frame, cams + size_t(frame_set[b]) * g.V, g, fx[gx_], fy[gy_], fz[gz_], acc);
  }
#pragma unroll
  for (int c = 0; c < 4 * NV; ++c)
    if (c < J) cubes[(size_t(p) * J + c) * nvox + lv] = acc[c];
}

// ---------------------------------------------------------------------------------------------
// Orthographic maxima of a materialised cube: one workgroup per (person, joint), x-slabs
// streamed through LDS; yz keeps a running max in registers.  PER = (y,z) cells per thread, a power of two >= C*C/256
// (as a run-time bound on a 64-deep unrolled loop the 64 `i < per` predicates were kept as SGPR pairs: 78 spills, round 4).
template <int PER>
__global__ void __launch_bounds__(256)
k_triplane_max(const float* __restrict__ cubes, float* __restrict__ planes, int J, int C) {
  HIP_DYNAMIC_SHARED(float, slab)                     // [C][C+1]
  const int j = blockIdx.x, p = blockIdx.y, t = threadIdx.x;
  const int CC = C * C, ld = C + 1;
  const float* cube = cubes + (size_t(p) * J + j) * CC * C;
  float* pxy = planes + ((size_t(p) * 3 + 0) * J + j) * CC;
  float* pxz = planes + ((size_t(p) * 3 + 1) * J + j) * CC;
  float* pyz = planes + ((size_t(p) * 3 + 2) * J + j) * CC;
  float run[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) run[i] = -INFINITY;
  for (int x = 0; x < C; ++x) {
    __syncthreads();
    int ncell = CC;
    FVP_OPAQUE(ncell);                                // (the PER bounds tests stay in the loop: hoisted, each is an SGPR pair)
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int cell = t + i * 256;
      if (cell < ncell) {
        const float v = cube[size_t(x) * CC + cell];
        slab[(cell / C) * ld + (cell % C)] = v;
        run[i] = fmaxf(run[i], v);
      }
    }
    __syncthreads();
    for (int r = t; r < 2 * C; r += 256) {
      if (r < C) {                                     // xy[x][y=r] = max_z
        float m = slab[r * ld];
        for (int z = 1; z < C; ++z) m = fmaxf(m, slab[r * ld + z]);
        pxy[x * C + r] = m;
      } else {                                         // xz[x][z] = max_y
        const int z = r - C;
        float m = slab[z];
        for (int y = 1; y < C; ++y) m = fmaxf(m, slab[y * ld + z]);
        pxz[x * C + z] = m;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int cell = t + i * 256;
    if (cell < CC) pyz[cell] = run[i];
  }
}

// ---------------------------------------------------------------------------------------------
// Fused fast path: sample every voxel of the person's window and emit the three maxima
// without writing the cube.
//
// The gather is bound by the texture-addresser / L1 line rate (GRBM_TA_BUSY 95 % in the first
// version, which gave every lane a whole 64-byte pixel = 4 dwordx4 loads touching 64 different
// cache lines per instruction).  Here four consecutive lanes share one voxel: lane q of the quad
// loads channels 4q..4q+3 of each tap, so one dwordx4 instruction covers 16 voxels x 64
// contiguous bytes (4x fewer lines per instruction).  The projection arithmetic is shared inside
// the quad: lane q projects view q and the tap descriptors are broadcast with DPP quad_perm.
//
// Workgroup = (person, 256 (y,z) cells) x 4 lanes = 1024 threads; lanes of a wave: 16 consecutive
// z x 4 channel quads; loop over x inside the thread.
//   yz[y][z] = max_x : running max in registers, owned by this workgroup -> plain store
//   xy[x][y] = max_z : shuffle over the 16 z-lanes of a wave, then LDS across the row's waves
//   xz[x][z] = max_y : LDS over the workgroup's rows, then integer atomicMax across workgroups
//                      (values are clamped to [0,1] => non-negative => int order == float order;
//                      planes are pre-zeroed, so the result is order-independent and bit-equal
//                      to the materialised path)
// Workgroups are numbered so that all persons of one frame run on one XCD (block id mod 8 is the
// XCD): an XCD's L2 then serves one frame's heatmaps instead of all of them.
struct TapD {   // tap descriptor of one (voxel, view): 4 offsets, 4 weights, inside mask
  int off[4];
  float w[4];
  int inside;
};

template <int SRC>
__device__ __forceinline__ int quad_bcast_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, SRC * 0x55, 0xf, 0xf, false);   // quad_perm [SRC,SRC,SRC,SRC]
}
template <int SRC>
__device__ __forceinline__ TapD quad_bcast(const TapD& t) {
  TapD r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r.off[k] = quad_bcast_i<SRC>(t.off[k]);
    r.w[k] = __int_as_float(quad_bcast_i<SRC>(__float_as_int(t.w[k])));
  }
  r.inside = quad_bcast_i<SRC>(t.inside);
  return r;
}

__device__ __forceinline__ TapD make_taps(const Cam& cm, const FvpGeom& g, float wx, float wy, float wz) {
  float gx, gy;
  project_norm(cm, g, wx, wy, wz, gx, gy);
  const Taps t = bilinear_taps(gx, gy, g.W, g.H);
  TapD d;
#pragma unroll
  for (int k = 0; k < 4; ++k) { d.off[k] = t.off[k]; d.w[k] = t.w[k]; }
  d.inside = t.inside;
  return d;
}

// acc[i] (+)= bilinear sample of channels ch0..ch0+3 (reference accumulation order: nw*w then fma)
__device__ __forceinline__ void sample4(const float* __restrict__ cl, int JP, int ch0, const TapD& t, bool first,
                                        float (&acc)[4]) {
  float4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = (t.inside >> k) & 1;
    v[k] = *reinterpret_cast<const float4*>(cl + size_t(in ? t.off[k] : 0) * JP + ch0);
    if (!in) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float s[4] = {__fmul_rn(v[0].x, t.w[0]), __fmul_rn(v[0].y, t.w[0]), __fmul_rn(v[0].z, t.w[0]),
                __fmul_rn(v[0].w, t.w[0])};
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    s[0] = __fmaf_rn(v[k].x, t.w[k], s[0]);
    s[1] = __fmaf_rn(v[k].y, t.w[k], s[1]);
    s[2] = __fmaf_rn(v[k].z, t.w[k], s[2]);
    s[3] = __fmaf_rn(v[k].w, t.w[k], s[3]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = first ? s[i] : __fadd_rn(acc[i], s[i]);
}

// Whole-space cubes + fused z-max, quad-lane form (the default): 4 lanes per voxel, lane q
// gathers channels [16n + 4q, +4) of every tap, so the four lanes of a voxel read one contiguous
// 64-byte run per tap (16 cache lines per wave instruction instead of 64 -- the gather is bound by
// the texture-addresser line rate) and lane q computes the projection of view q for the quad
// (DPP broadcast).  Same per-channel arithmetic as k_project_whole (bit-equal cubes).
// the per-voxel body of the quad-lane form: mean over views of the bilinear samples of channels [16n + 4q, +4),
// clamped to [0,1] (shared by the whole-space kernel and the proposal-column kernel, so both give the same bits)
template <int NVL>
__device__ __forceinline__ void whole_point_quad(const float* __restrict__ frame, const Cam* __restrict__ cm,
                                                 const FvpGeom& g, int q, bool active, float wx, float wy, float wz,
                                                 float (&out)[NVL][4]) {
  const int JP = g.JP;
  const size_t view_stride = size_t(g.H) * g.W * JP;
  float acc[NVL][4];
#pragma unroll
  for (int n = 0; n < NVL; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[n][i] = 0.0f;
  TapD mine;
  mine.inside = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { mine.off[k] = 0; mine.w[k] = 0.0f; }
  if (active && q < g.V) mine = make_taps(cm[q], g, wx, wy, wz);
  auto add_view = [&](const TapD& tv, int v) {
#pragma unroll
    for (int n = 0; n < NVL; ++n) {
      const int ch0 = 16 * n + 4 * q;
      if (ch0 < JP) {
        if (tv.inside) sample4(frame + v * view_stride, JP, ch0, tv, v == 0, acc[n]);
        else if (v == 0) { acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.0f; }
      }
    }
  };
  { const TapD tv = quad_bcast<0>(mine); if (active) add_view(tv, 0); }
  if (g.V > 1) { const TapD tv = quad_bcast<1>(mine); if (active) add_view(tv, 1); }
  if (g.V > 2) { const TapD tv = quad_bcast<2>(mine); if (active) add_view(tv, 2); }
  if (g.V > 3) { const TapD tv = quad_bcast<3>(mine); if (active) add_view(tv, 3); }
  for (int v = 4; v < g.V; ++v)
    if (active) add_view(make_taps(cm[v], g, wx, wy, wz), v);
  const float nv = float(g.V);
#pragma unroll
  for (int n = 0; n < NVL; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[n][i] = clampf(__fdiv_rn(acc[n][i], nv), 0.0f, 1.0f);
}

template <int NVL>
__global__ void __launch_bounds__(1024)
k_project_whole_q(const float* __restrict__ heat_cl, const Cam* __restrict__ cams,
                  const int* __restrict__ frame_set, const float* __restrict__ ax, const float* __restrict__ ay,
                  const float* __restrict__ az, int X, int Y, int Z, int B, int nblk, FvpGeom g,
                  float* __restrict__ cubes, float* __restrict__ zmax) {
  __shared__ float sm[16 * NVL][256];
  const int cpb = 256 / Z;
  // 1-D grid of B * nblk workgroups.  Consecutive workgroup ids go round-robin over the 8 XCDs (each with
  // its own L2), so with B % 8 == 0 frame f is pinned to XCD f % 8: an XCD's L2 then holds the heatmaps of
  // one frame at a time instead of all B (HBM-side fetch was 3.5x the algorithmic bytes without it).
  int b, bxi;
  if (B % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    b = xcd + 8 * (j / nblk);
    bxi = j % nblk;
  } else {
    b = blockIdx.x / nblk;
    bxi = blockIdx.x % nblk;
  }
  const int t = threadIdx.x, q = t & 3, vl = t >> 2;           // vl = voxel inside the workgroup
  const int cl_ = vl / Z, z = vl - cl_ * Z;
  const int col = bxi * cpb + cl_;
  const int ncol = X * Y;
  const bool active = cl_ < cpb && col < ncol;
  const int J = g.J, JP = g.JP;
  const size_t view_stride = size_t(g.H) * g.W * JP;
  const float* frame = heat_cl + size_t(b) * g.V * view_stride;
  const Cam* cm = cams + size_t(frame_set[b]) * g.V;
  float wx = 0.0f, wy = 0.0f, wz = 0.0f;
  if (active) {
    const int x = col / Y, y = col - x * Y;
    wx = ax[x];
    wy = ay[y];
    wz = az[z];
  }
  float val[NVL][4];
  whole_point_quad<NVL>(frame, cm, g, q, active, wx, wy, wz, val);
  const size_t vox = size_t(col) * Z + z;
  const size_t nvox = size_t(ncol) * Z;
#pragma unroll
  for (int n = 0; n < NVL; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 16 * n + 4 * q + i;
      const float v = val[n][i];
      if (zmax) sm[c][vl] = active ? v : 0.0f;
      if (cubes && active && c < J) cubes[(size_t(b) * J + c) * nvox + vox] = v;
    }
  if (zmax) {
    __syncthreads();
    for (int item = t; item < cpb * J; item += 1024) {
      const int c = item / cpb, k = item - c * cpb;
      const int cc = bxi * cpb + k;
      if (cc < ncol) {
        float m = sm[c][k * Z];
        for (int zz = 1; zz < Z; ++zz) m = fmaxf(m, sm[c][k * Z + zz]);
        zmax[(size_t(b) * J + c) * ncol + cc] = m;
      }
    }
  }
}

// Proposal z-columns only (human_detection_net.py:92-93 gathers N columns of the cubes per frame and nothing else
// of them is used by the fused forward): feat1d[b*N + k][c][z] = cubes[b][c][flat[b][k]][z], computed by the very
// device function of k_project_whole_q (same bits as the materialised cubes), 61 MB of cube writes per 8 frames never
// happen.  Workgroup = one proposal column: Z voxels x 4 lanes.
template <int NVL>
__global__ void __launch_bounds__(1024)
k_project_columns_q(const float* __restrict__ heat_cl, const Cam* __restrict__ cams, const int* __restrict__ frame_set,
                    const float* __restrict__ ax, const float* __restrict__ ay, const float* __restrict__ az, int Y,
                    int Z, int ncol, int N, FvpGeom g, const long long* __restrict__ flat, float* __restrict__ feat1d) {
  const int pi = blockIdx.x, b = pi / N;
  const int t = threadIdx.x, q = t & 3, z = t >> 2;
  const long long col = flat[pi];
  const bool active = z < Z && col >= 0 && col < ncol;
  const int J = g.J;
  const size_t view_stride = size_t(g.H) * g.W * g.JP;
  const float* frame = heat_cl + size_t(b) * g.V * view_stride;
  const Cam* cm = cams + size_t(frame_set[b]) * g.V;
  float wx = 0.0f, wy = 0.0f, wz = 0.0f;
  if (active) {
    const int x = int(col) / Y, y = int(col) - x * Y;
    wx = ax[x];
    wy = ay[y];
    wz = az[z];
  }
  float val[NVL][4];
  whole_point_quad<NVL>(frame, cm, g, q, active, wx, wy, wz, val);
#pragma unroll
  for (int n = 0; n < NVL; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 16 * n + 4 * q + i;
      if (z < Z && c < J) feat1d[(size_t(pi) * J + c) * Z + z] = active ? val[n][i] : 0.0f;
    }
}

#if FVP_DIAG   // the round-1 gather form of the fused projection: diagnostics build only (FVP_TRIPLANE_GATHER=1)
template <int NVL>   // channel quads per lane: ceil(JP/16)
__global__ void __launch_bounds__(1024)
k_project_triplane(const float* __restrict__ heat_cl, const Cam* __restrict__ cams, const int* __restrict__ frame_set,
                   const int* __restrict__ person_frame, const uint8_t* __restrict__ person_valid,
                   const int* __restrict__ boxes, const float* __restrict__ fx, const float* __restrict__ fy,
                   const float* __restrict__ fz, int C, int nP, int chunks, int ppf, FvpGeom g,
                   float* __restrict__ planes) {
  __shared__ float sm_xz[4 * NVL * 4][256];       // [channel][cell]
  __shared__ float sm_xy[16][4 * NVL * 4];        // [wave][channel]
  // ---- block -> (person, chunk), frame-major per XCD when the frame count allows it
  int p, chunk;
  {
    const int id = blockIdx.x;
    const int bpf = ppf * chunks;                 // blocks per frame
    const int nframes = nP / ppf;
    if (nframes % 8 == 0) {
      const int xcd = id & 7, j = id >> 3;
      const int frame = xcd + 8 * (j / bpf), r = j % bpf;
      p = frame * ppf + r / chunks;
      chunk = r % chunks;
    } else {
      p = id / chunks;
      chunk = id % chunks;
    }
  }
  if (person_valid && !person_valid[p]) return;
  const int* bx = boxes + p * 9;
  const int tl0 = bx[0], tl1 = bx[1], tl2 = bx[2];
  const int s0 = bx[3], s1 = bx[4], s2 = bx[5], e0 = bx[6], e1 = bx[7], e2 = bx[8];
  if (s0 >= e0 || s1 >= e1 || s2 >= e2) return;
  const int t = threadIdx.x, q = t & 3, lane = t & 63, wave = t >> 6;
  const int J = g.J, JP = g.JP, CC = C * C;
  const int rows = 256 / C > 0 ? 256 / C : 1;
  const int cell_l = t >> 2;                       // 0..255 within the workgroup
  const int cell = chunk * 256 + cell_l;
  const int y = cell / C, z = cell - y * C;
  const int y_first = (chunk * 256) / C;
  if (tl1 + y_first >= e1 || tl1 + y_first + rows - 1 < s1) return;   // chunk outside the y window
  const int gy_ = tl1 + y, gz_ = tl2 + z;
  const bool yz_in = y < C && gy_ >= s1 && gy_ < e1 && gz_ >= s2 && gz_ < e2;
  const int b = person_frame[p];
  const size_t view_stride = size_t(g.H) * g.W * JP;
  const float* frame = heat_cl + size_t(b) * g.V * view_stride;
  const Cam* cm = cams + size_t(frame_set[b]) * g.V;
  const float wy = yz_in ? fy[gy_] : 0.0f, wz = yz_in ? fz[gz_] : 0.0f;
  float* pxy = planes + (size_t(p) * 3 + 0) * J * CC;
  float* pxz = planes + (size_t(p) * 3 + 1) * J * CC;
  float* pyz = planes + (size_t(p) * 3 + 2) * J * CC;
  const int wpr = C >= 16 ? C / 16 : 1;           // waves per y row
  float run[NVL][4];
#pragma unroll
  for (int n = 0; n < NVL; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i) run[n][i] = 0.0f;
  const float nv = float(g.V);
  const int lx0 = s0 - tl0, lx1 = e0 - tl0;
  for (int lx = lx0; lx < lx1; ++lx) {
    const float wx = fx[tl0 + lx];
    float acc[NVL][4];
#pragma unroll
    for (int n = 0; n < NVL; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[n][i] = 0.0f;
    // lane q projects view q; descriptors are shared inside the quad by DPP
    TapD mine;
    mine.inside = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { mine.off[k] = 0; mine.w[k] = 0.0f; }
    if (yz_in && q < g.V) mine = make_taps(cm[q], g, wx, wy, wz);
    auto add_view = [&](const TapD& tv, int v) {
#pragma unroll
      for (int n = 0; n < NVL; ++n) {
        const int ch0 = 16 * n + 4 * q;
        if (ch0 < JP) {
          if (tv.inside) sample4(frame + v * view_stride, JP, ch0, tv, v == 0, acc[n]);
          else if (v == 0) { acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.0f; }
        }
      }
    };
    { const TapD tv = quad_bcast<0>(mine); if (yz_in) add_view(tv, 0); }
    if (g.V > 1) { const TapD tv = quad_bcast<1>(mine); if (yz_in) add_view(tv, 1); }
    if (g.V > 2) { const TapD tv = quad_bcast<2>(mine); if (yz_in) add_view(tv, 2); }
    if (g.V > 3) { const TapD tv = quad_bcast<3>(mine); if (yz_in) add_view(tv, 3); }
    for (int v = 4; v < g.V; ++v)
      if (yz_in) add_view(make_taps(cm[v], g, wx, wy, wz), v);
#pragma unroll
    for (int n = 0; n < NVL; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[n][i] = clampf(__fdiv_rn(acc[n][i], nv), 0.0f, 1.0f);
        run[n][i] = fmaxf(run[n][i], acc[n][i]);
      }
    // ---- reductions
    __syncthreads();                               // previous iteration's LDS readers are done
#pragma unroll
    for (int n = 0; n < NVL; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 16 * n + 4 * q + i;
        sm_xz[c][cell_l] = acc[n][i];
        float m = acc[n][i];
        const int zw = C < 16 ? C : 16;            // z lanes of this row inside the wave
        for (int o = (zw >> 1) * 4; o >= 4; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (C >= 16) { if ((lane >> 2) == 0) sm_xy[wave][c] = m; }
        else if (((lane >> 2) & (zw - 1)) == 0 && c < J && y < C && m > 0.0f) pxy[size_t(c) * CC + lx * C + y] = m;
      }
    __syncthreads();
    if (C >= 16) {                                 // xy: combine the waves of each row
      const int nitem = rows * J;
      if (t < nitem) {
        const int r = t / J, c = t - r * J;
        float m = sm_xy[r * wpr][c];
        for (int k = 1; k < wpr; ++k) m = fmaxf(m, sm_xy[r * wpr + k][c]);
        const int yy = y_first + r;
        if (yy < C && m > 0.0f) pxy[size_t(c) * CC + lx * C + yy] = m;
      }
    }
    for (int item = t; item < J * C; item += 1024) {   // xz: max over this workgroup's rows
      const int c = item / C, zz = item - c * C;
      float m = sm_xz[c][zz];
      for (int r = 1; r < rows; ++r) m = fmaxf(m, sm_xz[c][r * C + zz]);
      if (m > 0.0f) {
        if (chunks > 1) atomicMax(reinterpret_cast<int*>(&pxz[size_t(c) * CC + lx * C + zz]), __float_as_int(m));
        else pxz[size_t(c) * CC + lx * C + zz] = m;
      }
    }
  }
  if (y < C) {
#pragma unroll
    for (int n = 0; n < NVL; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 16 * n + 4 * q + i;
        if (c < J && run[n][i] > 0.0f) pyz[size_t(c) * CC + y * C + z] = run[n][i];
      }
  }
}

#endif  // FVP_DIAG

}  // namespace fvp
#include "fvp_project_lds.h"
namespace fvp {

// z-max of materialised cubes: one thread per (b,j,x,y) column.
__global__ void __launch_bounds__(256) k_zmax(const float* __restrict__ cubes, float* __restrict__ zmax, long n, int Z) {
  const long i = long(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const float* c = cubes + i * Z;
  float m = c[0];
  for (int z = 1; z < Z; ++z) m = fmaxf(m, c[z]);
  zmax[i] = m;
}

// ---------------------------------------------------------------------------------------------
// fvp_person_boxes: integer window arithmetic of project_individual.py:110-121.
__global__ void __launch_bounds__(64)
k_person_boxes(const float* __restrict__ centers, int n, const float* __restrict__ consts, int f0, int f1, int f2,
               int c0, int c1, int c2, int* __restrict__ boxes, float* __restrict__ offset) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float* pc = centers + size_t(i) * 7;
  const int fine[3] = {f0, f1, f2}, cube[3] = {c0, c1, c2};
  int tl[3], m[3];
  for (int a = 0; a < 3; ++a) {
    // round(c*scale + bias): two roundings then half-to-even, .int() truncation (exact here)
    const float v = __fadd_rn(__fmul_rn(pc[a], consts[a]), consts[3 + a]);
    tl[a] = int(rintf(v));
    // offset = tl / (fine-1) * whole - whole/2 + ind/2   (left to right, :111)
    const float whole = consts[6 + a], ind = consts[9 + a];
    float o = __fmul_rn(__fdiv_rn(float(tl[a]), float(fine[a] - 1)), whole);
    o = __fadd_rn(__fsub_rn(o, __fdiv_rn(whole, 2.0f)), __fdiv_rn(ind, 2.0f));
    offset[size_t(i) * 3 + a] = o;
  }
  for (int a = 0; a < 2; ++a) {
    // ((1 - bbox) / 2 * (C - 1)).int(), negatives -> 0 (:114-115); z margin is 0 (:117)
    const float r = __fmul_rn(__fdiv_rn(__fsub_rn(1.0f, pc[5 + a]), 2.0f), float(cube[a] - 1));
    m[a] = int(r);
    if (m[a] < 0) m[a] = 0;
  }
  m[2] = 0;
  for (int a = 0; a < 3; ++a) {
    const int s = tl[a] + m[a], e = tl[a] + cube[a] - m[a];
    boxes[i * 9 + a] = tl[a];
    boxes[i * 9 + 3 + a] = s >= 0 ? s : 0;
    boxes[i * 9 + 6 + a] = e <= fine[a] ? e : fine[a];
  }
}

}  // namespace fvp

// =============================================================================================
// C ABI
// =============================================================================================
using namespace fvp;

#define FVP_NV_SWITCH(nv, CALL)     \
  switch (nv) {                     \
    case 1: { CALL(1); } break;     \
    case 2: { CALL(2); } break;     \
    case 3: { CALL(3); } break;     \
    case 4: { CALL(4); } break;     \
    case 5: { CALL(5); } break;     \
    case 6: { CALL(6); } break;     \
    case 7: { CALL(7); } break;     \
    case 8: { CALL(8); } break;     \
    default: return FVP_ELIMIT;     \
  }

static int check_geom(const FvpGeom* g) {
  if (!g) return FVP_EINVAL;
  if (g->V < 1 || g->V > FVP_MAX_VIEWS || g->J < 1 || g->J > FVP_MAX_JOINTS) return FVP_ELIMIT;
  if (g->JP != 4 * ((g->J + 3) / 4) || g->W < 2 || g->H < 2) return FVP_EINVAL;
  return 0;
}

extern "C" int fvp_heatmaps_to_cl(const float* heat, float* heat_cl, int B, const FvpGeom* g, fvp_stream_t s) {
  FVP_REQUIRE(heat && heat_cl && B >= 0);
  if (int e = check_geom(g)) return e;
  if (B == 0) return 0;
  const int HW = g->H * g->W;
  ProfScope ps(FVP_K_OTHER, as_stream(s));
#define CALL(NV)                                                                                             \
  auto k = &k_heat_to_cl<NV>;                                                                                \
  hipLaunchKernelGGL(k, dim3(ceil_div(HW, 256), B * g->V), dim3(256), 0, as_stream(s), heat, heat_cl, g->J, HW);
  FVP_NV_SWITCH(g->JP / 4, CALL)
#undef CALL
  return launch_status();
}

extern "C" int fvp_sample_grid(const float* ax, const float* ay, const float* az, int nx, int ny, int nz,
                               const float* cams, const FvpGeom* g, float* grid, fvp_stream_t s) {
  FVP_REQUIRE(ax && ay && az && cams && grid && nx > 0 && ny > 0 && nz > 0);
  if (int e = check_geom(g)) return e;
  const int n = nx * ny * nz;
  ProfScope ps(FVP_K_OTHER, as_stream(s));
  hipLaunchKernelGGL(k_sample_grid, dim3(ceil_div(n, 256), g->V), dim3(256), 0, as_stream(s), ax, ay, az, nx, ny,
                     nz, reinterpret_cast<const Cam*>(cams), *g, grid);
  return launch_status();
}

extern "C" int fvp_project_whole(const float* heat_cl, const float* cams, const int32_t* frame_set,
                                 const float* ax, const float* ay, const float* az, int X, int Y, int Z, int B,
                                 const FvpGeom* g, float* cubes, float* zmax, fvp_stream_t s) {
  FVP_REQUIRE(heat_cl && cams && frame_set && ax && ay && az && (cubes || zmax) && X > 0 && Y > 0 && Z > 0);
  if (int e = check_geom(g)) return e;
  FVP_LIMIT(Z <= 256);
  if (B == 0) return 0;
  const int cpb = 256 / Z;
  ProfScope ps(FVP_K_PROJECT_WHOLE, as_stream(s));
  static const bool no_quad = fvp::diag_env("FVP_WHOLE_NO_QUAD") != nullptr;
  const int nvl = ceil_div(g->JP, 16);
  if (!no_quad && nvl <= 2) {
    const int nblk = ceil_div(X * Y, cpb);
    if (nvl == 1)
      hipLaunchKernelGGL(k_project_whole_q<1>, dim3(nblk * B), dim3(1024), 0, as_stream(s), heat_cl,
                         reinterpret_cast<const Cam*>(cams), frame_set, ax, ay, az, X, Y, Z, B, nblk, *g, cubes, zmax);
    else
      hipLaunchKernelGGL(k_project_whole_q<2>, dim3(nblk * B), dim3(1024), 0, as_stream(s), heat_cl,
                         reinterpret_cast<const Cam*>(cams), frame_set, ax, ay, az, X, Y, Z, B, nblk, *g, cubes, zmax);
    return launch_status();
  }
#define CALL(NV)                                                                                              \
  auto k = &k_project_whole<NV>;                                                                              \
  hipLaunchKernelGGL(k, dim3(ceil_div(X * Y, cpb), B), dim3(256), 0, as_stream(s), heat_cl,                   \
                     reinterpret_cast<const Cam*>(cams), frame_set, ax, ay, az, X, Y, Z, *g, cubes, zmax);
  FVP_NV_SWITCH(g->JP / 4, CALL)
#undef CALL
  return launch_status();
}

extern "C" int fvp_project_columns(const float* heat_cl, const float* cams, const int32_t* frame_set, const float* ax,
                                   const float* ay, const float* az, int X, int Y, int Z, int B, const FvpGeom* g,
                                   const int64_t* flat, int N, float* feat1d, fvp_stream_t s) {
  FVP_REQUIRE(heat_cl && cams && frame_set && ax && ay && az && flat && feat1d && X > 0 && Y > 0 && Z > 0 && N > 0);
  if (int e = check_geom(g)) return e;
  FVP_LIMIT(Z <= 256 && g->JP <= 32);
  if (B == 0) return 0;
  ProfScope ps(FVP_K_PROJECT_WHOLE, as_stream(s));
  const int threads = ceil_div(4 * Z, 64) * 64;
  if (g->JP <= 16)
    hipLaunchKernelGGL(k_project_columns_q<1>, dim3(B * N), dim3(threads), 0, as_stream(s), heat_cl,
                       reinterpret_cast<const Cam*>(cams), frame_set, ax, ay, az, Y, Z, X * Y, N, *g,
                       reinterpret_cast<const long long*>(flat), feat1d);
  else
    hipLaunchKernelGGL(k_project_columns_q<2>, dim3(B * N), dim3(threads), 0, as_stream(s), heat_cl,
                       reinterpret_cast<const Cam*>(cams), frame_set, ax, ay, az, Y, Z, X * Y, N, *g,
                       reinterpret_cast<const long long*>(flat), feat1d);
  return launch_status();
}

extern "C" int fvp_zmax(const float* cubes, float* zmax, long n, int Z, fvp_stream_t s) {
  FVP_REQUIRE(cubes && zmax && n >= 0 && Z > 0);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_zmax, dim3(unsigned((n + 255) / 256)), dim3(256), 0, as_stream(s), cubes, zmax, n, Z);
  return launch_status();
}

extern "C" int fvp_person_boxes(const float* centers, int n, const float* consts, const int32_t* fine_cube,
                                int32_t* boxes, float* offset, fvp_stream_t s) {
  FVP_REQUIRE(centers && consts && fine_cube && boxes && offset && n >= 0);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_person_boxes, dim3(ceil_div(n, 64)), dim3(64), 0, as_stream(s), centers, n, consts,
                     fine_cube[0], fine_cube[1], fine_cube[2], fine_cube[3], fine_cube[4], fine_cube[5], boxes,
                     offset);
  return launch_status();
}

static int check_cube(int C) {
  if (C < 4 || C > 256 || (C & (C - 1)) != 0) return FVP_ELIMIT;   // power of two (row segments)
  return 0;
}

extern "C" int fvp_project_individual(const float* heat_cl, const float* cams, const int32_t* frame_set,
                                      const int32_t* person_frame, const uint8_t* person_valid,
                                      const int32_t* boxes, const float* fx, const float* fy, const float* fz,
                                      const int32_t* fine, int C, int nP, const FvpGeom* g, float* cubes,
                                      fvp_stream_t s) {
  FVP_REQUIRE(heat_cl && cams && frame_set && person_frame && boxes && fx && fy && fz && cubes && nP >= 0);
  (void)fine;
  if (int e = check_geom(g)) return e;
  if (int e = check_cube(C)) return e;
  if (nP == 0) return 0;
  ProfScope ps(FVP_K_OTHER, as_stream(s));
#define CALL(NV)                                                                                                 \
  auto k = &k_project_individual<NV>;                                                                            \
  hipLaunchKernelGGL(k, dim3(ceil_div(C * C * C, 256), nP), dim3(256), 0, as_stream(s), heat_cl,                 \
                     reinterpret_cast<const Cam*>(cams), frame_set, person_frame, person_valid, boxes, fx, fy, fz, \
                     C, *g, cubes);
  FVP_NV_SWITCH(g->JP / 4, CALL)
#undef CALL
  return launch_status();
}

extern "C" int fvp_triplane_max(const float* cubes, float* planes, int nP, int J, int C, fvp_stream_t s) {
  FVP_REQUIRE(cubes && planes && nP >= 0 && J > 0);
  if (int e = check_cube(C)) return e;
  FVP_LIMIT(C <= 128);
  if (nP == 0) return 0;
  ProfScope ps(FVP_K_OTHER, as_stream(s));
  const int per = (C * C + 255) / 256;                 // <= 64 for C <= 128
  const dim3 grid(J, nP);
  const size_t lds = size_t(C) * (C + 1) * sizeof(float);
#define FVP_TRIMAX(PER_) hipLaunchKernelGGL(k_triplane_max<PER_>, grid, dim3(256), lds, as_stream(s), cubes, planes, J, C)
  if (per <= 1) FVP_TRIMAX(1);
  else if (per <= 4) FVP_TRIMAX(4);
  else if (per <= 16) FVP_TRIMAX(16);
  else FVP_TRIMAX(64);
#undef FVP_TRIMAX
  return launch_status();
}

extern "C" int fvp_project_individual_triplane(const float* heat_cl, const float* cams, const int32_t* frame_set,
                                               const int32_t* person_frame, const uint8_t* person_valid,
                                               const int32_t* boxes, const float* fx, const float* fy,
                                               const float* fz, const int32_t* fine, int C, int nP, const FvpGeom* g,
                                               float* planes, int persons_per_frame, const float* fine_grid,
                                               fvp_stream_t s) {
  FVP_REQUIRE(heat_cl && cams && frame_set && person_frame && boxes && fx && fy && fz && planes && nP >= 0);
  FVP_REQUIRE(!fine_grid || fine);
  if (int e = check_geom(g)) return e;
  if (int e = check_cube(C)) return e;
  if (nP == 0) return 0;
  // persons_per_frame > 0 promises person p belongs to frame p / persons_per_frame (used only to
  // place the workgroups of one frame on one XCD); 0 = unknown
  const int ppf = (persons_per_frame > 0 && nP % persons_per_frame == 0) ? persons_per_frame : nP;
  [[maybe_unused]] const int chunks = ceil_div(C * C, 256);      // (the diagnostics build's gather form)
  const int nvl = ceil_div(g->JP, 16);
  ProfScope ps(FVP_K_PROJECT_TRIPLANE, as_stream(s));
  // The shipped form for every joint count (JP <= 32) is the UNSTAGED compact-block kernel k_project_triplane_blk (round 4:
  // Panoptic 494 -> 348 us, Shelf 709 -> 592 us, Campus 161 -> 158 us against the LDS-staged forms, same bits; lanes whose
  // channel quad lies beyond JP idle, so miniature joint counts run it too).  The LDS-staged quad / lane-per-voxel forms
  // and the round-1 gather form exist in the diagnostics build only (round 6), behind FVP_TRIPLANE_STAGED / _LANE /
  // _QUAD / _GATHER; they give the same bits (tests/test_emu_kernels.py, tools/gpu_switch_matrix.sh).
  const int F0 = fine_grid ? fine[0] : 0, F1 = fine_grid ? fine[1] : 0, F2 = fine_grid ? fine[2] : 0;
  const int nby = ceil_div(C, kBY);
#if FVP_DIAG
  const int nbx = ceil_div(C, kBX);
  static const bool gather = fvp::diag_env("FVP_TRIPLANE_GATHER") != nullptr;
  // FVP_TRIPLANE_QUAD=1: the round-2 form of the LDS-staged kernel (four lanes per voxel) instead of one lane per voxel
  const bool quad_form = fvp::diag_env("FVP_TRIPLANE_QUAD") != nullptr;            // read per call: tests switch it
  // diagnostics (wrong results): 1 no sampling, 2 no tile DMA, 4 no plane maxima, 8 no global-gather fallback
  static const int tri_ablate_env = fvp::diag_env("FVP_TRI_ABLATE") ? atoi(fvp::diag_env("FVP_TRI_ABLATE")) : 0;
  // bit 16 (NOT a diagnostic: results are identical either way): rectangles that fit the two tiles together are
  // staged across both when their turn comes instead of being gathered from global memory.  On for the
  // lane-per-voxel form (its gather reads 64 lines per instruction), off for the quad form (measured 495 -> 515 us:
  // the quad form's gather runs on the otherwise idle texture path beside the LDS reads of the staged rectangles).
  // FVP_TRI_TWO_TILE=0/1 overrides.
  const char* tt_env = fvp::diag_env("FVP_TRI_TWO_TILE");
  const bool staged = fvp::diag_env("FVP_TRIPLANE_STAGED") != nullptr;
  const bool force_lane = fvp::diag_env("FVP_TRIPLANE_LANE") != nullptr;           // read per call: tests switch it
  // lane-per-voxel staged form: JP = 4 .. 12 and 20 (JP = 16 stages through the quad form; JP 24 .. 32 would spill)
  const bool lane_form = !gather && !quad_form && g->JP <= 20 && ((staged && g->JP != 16) || force_lane);
  const int tri_ablate = (tri_ablate_env & ~16) | ((tt_env ? atoi(tt_env) != 0 : lane_form) ? 16 : 0);
  // tests: FVP_TRI_CAP_PX lowers the rectangle size the kernel treats as fitting a tile (the allocation is unchanged),
  // so that one-tile, two-tile and global-gather rectangles all occur on small fixtures; read per call
  auto cap_lim = [](int cap) { const char* e = fvp::diag_env("FVP_TRI_CAP_PX"); const int v = e ? atoi(e) : cap; return v > 0 && v < cap ? v : cap; };
  // two tiles per workgroup, two workgroups (512 threads, <= 128 VGPRs) per CU: 4 x 36 KB + state
  const int cap_px = int((36 * 1024) / (size_t(g->JP) * 4));
  const size_t lds = 2 * size_t(cap_px) * g->JP * 4 + 64;
  if (lane_form) {
    const int lq = (g->JP / 4) | 1;                                           // tile pixel pitch in quads (odd)
    const int cap_px = int((36 * 1024) / (size_t(lq) * 16));
    const size_t lds = 2 * size_t(cap_px) * lq * 16 + 64;
    FVP_LIMIT(size_t(cap_px) * lq * 4 >= size_t(kBX * kBY + (kBX + kBY) * 32) * g->JP);   // the block's plane cells alias tile 0
#define CALL2(NQ_, CACHED_)                                                                                       \
  {                                                                                                                \
    static LdsOptIn optin;                                                                                         \
    auto k = &k_project_triplane_lds2<NQ_, CACHED_>;                                                              \
    if (int e = lds_opt_in(optin, reinterpret_cast<const void*>(k), lds)) return e;                                \
    hipLaunchKernelGGL(k, dim3(nbx * nby * nP), dim3(kTriThreads), lds, as_stream(s), heat_cl,                     \
                       reinterpret_cast<const Cam*>(cams), frame_set, person_frame, person_valid, boxes, fx, fy, fz, C, \
                       nP, nbx, nby, ppf, cap_px, *g, fine_grid, F0, F1, F2, planes, tri_ablate, cap_lim(cap_px));                  \
  }
#define CASE2(NQ_) case NQ_: if (fine_grid) CALL2(NQ_, true) else CALL2(NQ_, false) break;
    switch (g->JP / 4) {
      CASE2(1) CASE2(2) CASE2(3) CASE2(4) CASE2(5)
      default: return FVP_ELIMIT;
    }
#undef CASE2
#undef CALL2
    return launch_status();
  }
  const bool blk_form = !gather && !quad_form && !staged;
#else
  constexpr bool blk_form = true;
#endif
  // JP = 16 (Panoptic): the compact-block form WITHOUT LDS staging (k_project_triplane_blk, round 4).  FVP_TRIPLANE_STAGED=1
  // (diagnostics build) keeps the staged quad form for comparison; both give the same bits.
  if (blk_form && nvl <= 2) {
    const int nbx2 = ceil_div(C, kBlkBX);
    const int bz = nvl == 2 ? 16 : kBlkBZ;
    // z-resident cells (ZRES, diagnostics build only: FVP_TRI_ZRES=1).  Measured in round 5, same box, same bits: Panoptic
    // 345 -> 389 us, Shelf 587 -> 654 us, Campus 155 -> 187 us.  The 35-43 KB of cells per workgroup cap a CU at 3-4
    // workgroups where the per-z-block cells (10 KB) let the register count decide (5), and the barriers it removes were
    // never the bound: the phases of the other workgroups on the CU already fill them.
    int zsh = 0;
    while ((1 << zsh) < C) ++zsh;
    const size_t lds_z = (size_t(kBlkBX + kBY) << zsh) * (g->JP + 1) * 4;
#if FVP_DIAG
    const char* zr_env = fvp::diag_env("FVP_TRI_ZRES");
    const bool zres = zr_env && atoi(zr_env) != 0 && (1 << zsh) >= bz && lds_z <= kTriZresMaxLds;
#else
    constexpr bool zres = false;
#endif
    const size_t lds_b = zres ? lds_z : size_t(kBlkBX * kBY + (kBlkBX + kBY) * bz) * (g->JP + 1) * 4;   // cell pitch JP + 1
#define CALLB(NVL_, CACHED_, ZRES_)                                                                                          \
  hipLaunchKernelGGL((k_project_triplane_blk<NVL_, CACHED_, ZRES_>), dim3(nbx2 * nby * nP), dim3(kBlkThreads), lds_b,       \
                     as_stream(s), heat_cl, reinterpret_cast<const Cam*>(cams), frame_set, person_frame, person_valid, boxes, \
                     fx, fy, fz, C, nP, nbx2, nby, ppf, *g, fine_grid, F0, F1, F2, planes, zsh)
#if FVP_DIAG
#define CALLB2(NVL_)                                                                     \
  {                                                                                      \
    if (fine_grid) { if (zres) CALLB(NVL_, true, true); else CALLB(NVL_, true, false); } \
    else { if (zres) CALLB(NVL_, false, true); else CALLB(NVL_, false, false); }         \
  }
#else
#define CALLB2(NVL_)                                                                     \
  {                                                                                      \
    if (fine_grid) CALLB(NVL_, true, false); else CALLB(NVL_, false, false);             \
  }
#endif
    // JP = 20 (J = 17: Shelf / Campus): five channel quads - the fifth sampled for four voxels in one pass (Q5, round 6).
    // FVP_TRI_NO_Q5=1 (diagnostics build) keeps the two-group form for comparison; same bits.
    const bool q5 = nvl == 2 && g->JP == 20 && !zres && fvp::diag_env("FVP_TRI_NO_Q5") == nullptr;
    if (q5) {
      if (fine_grid)
        hipLaunchKernelGGL((k_project_triplane_blk<2, true, false, true>), dim3(nbx2 * nby * nP), dim3(kBlkThreads), lds_b, as_stream(s),
                           heat_cl, reinterpret_cast<const Cam*>(cams), frame_set, person_frame, person_valid, boxes, fx, fy, fz, C, nP,
                           nbx2, nby, ppf, *g, fine_grid, F0, F1, F2, planes, zsh);
      else
        hipLaunchKernelGGL((k_project_triplane_blk<2, false, false, true>), dim3(nbx2 * nby * nP), dim3(kBlkThreads), lds_b, as_stream(s),
                           heat_cl, reinterpret_cast<const Cam*>(cams), frame_set, person_frame, person_valid, boxes, fx, fy, fz, C, nP,
                           nbx2, nby, ppf, *g, fine_grid, F0, F1, F2, planes, zsh);
    } else if (nvl == 2) CALLB2(2) else CALLB2(1)
#undef CALLB2
#undef CALLB
    return launch_status();
  }
#if FVP_DIAG
  if (!gather && nvl <= 2) {
    FVP_LIMIT(cap_px >= kBX * kBY + (kBX + kBY) * (nvl == 1 ? 32 : 16));      // the block's plane cells alias tile 0
#define CALL(NVL_, CACHED_)                                                                                        \
  {                                                                                                                \
    static LdsOptIn optin;                                                                                         \
    auto k = &k_project_triplane_lds<NVL_, CACHED_>;                                                               \
    if (int e = lds_opt_in(optin, reinterpret_cast<const void*>(k), lds)) return e;                                \
    hipLaunchKernelGGL(k, dim3(nbx * nby * nP), dim3(kTriThreads), lds, as_stream(s), heat_cl,                     \
                       reinterpret_cast<const Cam*>(cams), frame_set, person_frame, person_valid, boxes, fx, fy, fz, C, \
                       nP, nbx, nby, ppf, cap_px, *g, fine_grid, F0, F1, F2, planes, tri_ablate, cap_lim(cap_px));                  \
  }
    if (nvl == 1) {
      if (fine_grid) CALL(1, true) else CALL(1, false)
    } else {
      if (fine_grid) CALL(2, true) else CALL(2, false)
    }
#undef CALL
    return launch_status();
  }
  if (nvl == 1) {
    auto k = &k_project_triplane<1>;
    hipLaunchKernelGGL(k, dim3(chunks * nP), dim3(1024), 0, as_stream(s), heat_cl, reinterpret_cast<const Cam*>(cams),
                       frame_set, person_frame, person_valid, boxes, fx, fy, fz, C, nP, chunks, ppf, *g, planes);
  } else if (nvl == 2) {
    auto k = &k_project_triplane<2>;
    hipLaunchKernelGGL(k, dim3(chunks * nP), dim3(1024), 0, as_stream(s), heat_cl, reinterpret_cast<const Cam*>(cams),
                       frame_set, person_frame, person_valid, boxes, fx, fy, fz, C, nP, chunks, ppf, *g, planes);
  } else {
    return FVP_ELIMIT;
  }
  return launch_status();
#else
  return FVP_ELIMIT;                                     // more than 32 joint channels
#endif
}
