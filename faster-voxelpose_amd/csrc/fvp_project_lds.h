// Fused per-person back-projection + xy/xz/yz maxima with the heatmap footprint STAGED IN LDS
// (included by fvp_project.hip).  Replaces project_individual.ProjectLayer.forward :124-134 +
// joint_localization_net.py:80-81, same arithmetic per sample as every other projection kernel
// (project_norm / bilinear weights / tap order / view-ordered sum), hence bit-equal planes.
//
// Why: one person's window (~40 x 40 x 64 voxels x 5 views x 4 taps) reads 335 MB of taps out of a
// footprint of ~1.4 MB: neighbouring voxels re-read the same heatmap pixels ~240 times.  The first
// fused kernel gathered every tap through the texture path and sat at the L1 line rate (36 of
// 39 TB/s, 738 us for 80 people).  Here a workgroup owns a block of 8 x 8 x 32 voxels; per view it
//   1. projects the block's voxels (lane q of a voxel quad projects voxels q, q+4 and shares the
//      result with DPP quad_perm), reduces the bounding rectangle of their taps over the workgroup,
//   2. copies that rectangle of the channels-last heatmap (~14 x 44 px x JP floats) into LDS with the
//      LDS-DMA (coalesced 16-byte items, no VGPRs),
//   3. samples all 2048 voxels from LDS (ds_read_b128: the four lanes of a voxel read the 64
//      contiguous bytes of a pixel),
// accumulating the view sum in registers in view order.  After the last view: divide, clamp, and
// reduce the block's values into its 8x8 (xy), 8x32 (xz) and 8x32 (yz) cells through LDS integer
// atomicMax, then into the global planes with integer atomicMax (values are clamped to [0,1]:
// non-negative floats order like ints; the planes are pre-zeroed; zeros are skipped).
//
// Two workgroups (1024 threads, <= 76 KB of LDS) share a CU: one samples while the other waits for
// its DMA.  A rectangle that does not fit the tile falls back to global gathers for that view.
#pragma once
#include <type_traits>
#include <climits>

namespace fvp {

typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef FVP_TRI_PACKED
#define FVP_TRI_PACKED 1
#endif
#ifndef FVP_TRI_EXPLICIT_AS
#define FVP_TRI_EXPLICIT_AS 1   // 0: FLAT loads everywhere, 1: split by address space where it pays (NVL > 1), 2: everywhere
#endif

typedef float f32x4v __attribute__((ext_vector_type(4)));
// 16-byte loads with the address space spelled out (a generic pointer would become a FLAT load)
__device__ __forceinline__ float4 lds_ld4(const float* p) {
  const f32x4v v = *(const __attribute__((address_space(3))) f32x4v*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 glb_ld4(const float* p) {
  const f32x4v v = *(const __attribute__((address_space(1))) f32x4v*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}

struct TapL {      // tap descriptor of one (voxel, view), relative to the staged rectangle
  int base;        // float offset of the nw tap's pixel (clamped into the rectangle), channel 0, in the LDS tile
  int dx, dy;      // float offsets nw -> ne and nw -> sw (0 where the neighbour is clamped onto the same pixel)
  float w[4];      // bilinear weights, ZERO for taps outside the image (0 * pixel = +0: the reference's zero padding)
};

template <int SRC>
__device__ __forceinline__ TapL quad_bcast_l(const TapL& t) {
  TapL r;
  r.base = quad_bcast_i<SRC>(t.base);
  r.dx = quad_bcast_i<SRC>(t.dx);
  r.dy = quad_bcast_i<SRC>(t.dy);
#pragma unroll
  for (int k = 0; k < 4; ++k) r.w[k] = __int_as_float(quad_bcast_i<SRC>(__float_as_int(t.w[k])));
  return r;
}

// pixel of the nw tap and the bilinear weights of a normalised coordinate (same expressions as bilinear_taps)
__device__ __forceinline__ void tap_origin(float gx, float gy, int W, int H, int& x0, int& y0, float (&w)[4], int& inside) {
  const float ix = __fmul_rn(__fadd_rn(gx, 1.0f), 0.5f * float(W - 1));
  const float iy = __fmul_rn(__fadd_rn(gy, 1.0f), 0.5f * float(H - 1));
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float x1f = __fadd_rn(x0f, 1.0f), y1f = __fadd_rn(y0f, 1.0f);
  w[0] = __fmul_rn(__fsub_rn(x1f, ix), __fsub_rn(y1f, iy));
  w[1] = __fmul_rn(__fsub_rn(ix, x0f), __fsub_rn(y1f, iy));
  w[2] = __fmul_rn(__fsub_rn(x1f, ix), __fsub_rn(iy, y0f));
  w[3] = __fmul_rn(__fsub_rn(ix, x0f), __fsub_rn(iy, y0f));
  x0 = int(x0f);
  y0 = int(y0f);
  const bool x0in = x0 >= 0 && x0 < W, x1in = x0 + 1 >= 0 && x0 + 1 < W;
  const bool y0in = y0 >= 0 && y0 < H, y1in = y0 + 1 >= 0 && y0 + 1 < H;
  inside = (x0in && y0in ? 1 : 0) | (x1in && y0in ? 2 : 0) | (x0in && y1in ? 4 : 0) | (x1in && y1in ? 8 : 0);
}

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
// max over the 16 lanes of a DPP row (all lanes of the row receive it)
__device__ __forceinline__ int row_max(int v) {
  v = imax(v, dpp_i<0xB1>(v));        // quad_perm [1,0,3,2]
  v = imax(v, dpp_i<0x4E>(v));        // quad_perm [2,3,0,1]
  v = imax(v, dpp_i<0x141>(v));       // row_half_mirror
  v = imax(v, dpp_i<0x140>(v));       // row_mirror
  return v;
}

// Plane cell update: integer atomicMax on the non-negative float (fire and forget).  Reading the cell first and issuing
// the atomic only when it would win was measured slower (490 -> 533 us): the phase is latency-bound, not atomic-bound.
__device__ __forceinline__ void plane_max(float* cell, int vv) { atomicMax(reinterpret_cast<int*>(cell), vv); }

constexpr int kBX = 8, kBY = 4;                 // voxel block of a workgroup in x, y; in z: 32 (one channel quad per
                                               // lane) or 16 (two: J = 17), so that both forms stay within 128 VGPRs
constexpr int kTriThreads = kBX * kBY * 4 * 4; // (x, y, zs) slots x 4 channel-quad lanes = 512

#if FVP_DIAG   // the LDS-staged forms are diagnostics-build code (round 6): no shipped shape launches them
// CACHED: sampling coordinates come from the per-sequence cache `fgrid` ([nsets][V][F0*F1*F2][2], the reference's
// cached grid) instead of being recomputed: 2 coalesced 8-byte loads per lane and view replace ~230 VALU
// instructions (the projection was ~45 % of the kernel's VALU work).
template <int NVL, bool CACHED>   // NVL = channel quads per lane: ceil(JP/16)
__global__ void __launch_bounds__(kTriThreads, 4)     // <= 128 VGPRs: two workgroups (2 x 8 waves) per CU
k_project_triplane_lds(const float* __restrict__ heat_cl, const Cam* __restrict__ cams, const int* __restrict__ frame_set,
                       const int* __restrict__ person_frame, const uint8_t* __restrict__ person_valid,
                       const int* __restrict__ boxes, const float* __restrict__ fx, const float* __restrict__ fy,
                       const float* __restrict__ fz, int C, int nP, int nbx, int nby, int ppf, int cap_px, FvpGeom g,
                       const float* __restrict__ fgrid, int F0, int F1, int F2, float* __restrict__ planes, int ablate, int cap_lim) {
  constexpr int BZ = NVL == 1 ? 32 : 16;            // z extent of the voxel block
  constexpr int VPT = BZ / 4;                        // voxels per thread: z = zs + 4 i
  constexpr int OWN = VPT / 4;                       // voxels a lane projects per view (i = q + 4 k)
  HIP_DYNAMIC_SHARED(float, smem)
  // LDS: two tiles of cap_px * JP floats (view v lives in tile v & 1; tile 0 is aliased by the block's plane
  // cells after the last view) | 3 x 4 ints of rectangle state (view v uses slot v % 3)
  constexpr int NT = kTriThreads;
  const int J = g.J, JP = g.JP, CC = C * C, V = g.V, W = g.W, H = g.H;
  const int tile_sz = cap_px * JP;
  int* rect = reinterpret_cast<int*>(smem + 2 * size_t(tile_sz));        // [3][4]: -minx, maxx, -miny, maxy
  // ---- block id -> (person, x block, y block); frame-major per XCD when the frame count allows it
  int p, blk;
  {
    const int id = blockIdx.x;
    const int bpp = nbx * nby, bpf = ppf * bpp;
    const int nframes = nP / ppf;
    if (nframes % 8 == 0) {
      const int xcd = id & 7, j = id >> 3;
      const int frame = xcd + 8 * (j / bpf), r = j % bpf;
      p = frame * ppf + r / bpp;
      blk = r % bpp;
    } else {
      p = id / bpp;
      blk = id % bpp;
    }
  }
  if (person_valid && !person_valid[p]) return;
  const int* bx = boxes + p * 9;
  const int tl0 = bx[0], tl1 = bx[1], tl2 = bx[2];
  const int s0 = bx[3], s1 = bx[4], s2 = bx[5], e0 = bx[6], e1 = bx[7], e2 = bx[8];
  if (s0 >= e0 || s1 >= e1 || s2 >= e2) return;
  const int xb = blk / nby, yb = blk - xb * nby;
  const int gx0 = s0 + xb * kBX, gy0 = s1 + yb * kBY;                   // blocks are aligned to the window start
  if (gx0 >= e0 || gy0 >= e1) return;

  const int t = threadIdx.x, q = t & 3, lane = t & 63, wave = t >> 6;
  const int slot = t >> 2, zs = slot & 3, yy = (slot >> 2) & (kBY - 1), xx = slot / (4 * kBY);
  const int gxi = gx0 + xx, gyi = gy0 + yy;                              // fine-grid indices of this thread's column
  const bool col_in = gxi < e0 && gyi < e1;
  const int b = person_frame[p];
  const size_t view_stride = size_t(H) * W * JP;
  const float* frame = heat_cl + size_t(b) * V * view_stride;
  const Cam* cm = cams + size_t(frame_set[b]) * V;
  const float wx = col_in ? fx[gxi] : 0.0f, wy = col_in ? fy[gyi] : 0.0f;
  const size_t nfine = size_t(F0) * F1 * F2;
  // cached coordinates of this thread's column (z = 0) in view 0 of the frame's camera set
  const float2* gcol = CACHED ? reinterpret_cast<const float2*>(fgrid) + size_t(frame_set[b]) * V * nfine +
                                    (size_t(col_in ? gxi : 0) * F1 + (col_in ? gyi : 0)) * F2
                              : nullptr;
  float* pxy = planes + (size_t(p) * 3 + 0) * J * CC;
  float* pxz = planes + (size_t(p) * 3 + 1) * J * CC;
  float* pyz = planes + (size_t(p) * 3 + 2) * J * CC;
  const float nv = float(V);
  const int qn = JP >> 2;                                                // channel quads per pixel
  const unsigned m_qn = unsigned((1ull << 32) / unsigned(qn)) + 1u;

  // this lane's two voxels (i = q and q + 4) of view v: projection, tap origin; contribution to view v's rectangle
  struct Own { int xy[OWN]; int inside[OWN]; float w[OWN][4]; };
  // CACHED: coordinates of view v are loaded one iteration ahead (load_coords) so their latency hides behind
  // the sampling of the previous view
  float2 crd[OWN];
  auto load_coords = [&](int v, int gz0) {
#pragma unroll
    for (int k = 0; k < OWN; ++k) {
      const int gzi = gz0 + zs + 4 * (q + 4 * k);
      crd[k] = gcol[size_t(v) * nfine + (gzi < F2 ? gzi : F2 - 1)];
    }
  };
  auto project = [&](int v, int gz0, Own& o) {
    int mnx = INT_MIN, mxx = INT_MIN, mny = INT_MIN, mxy = INT_MIN;     // (-min, max) pairs
#pragma unroll
    for (int k = 0; k < OWN; ++k) {
      const int gzi = gz0 + zs + 4 * (q + 4 * k);
      const bool vin = col_in && gzi < e2;
      o.inside[k] = 0;
      o.xy[k] = 0;
      o.w[k][0] = o.w[k][1] = o.w[k][2] = o.w[k][3] = 0.0f;
      if (vin) {
        float sx, sy;
        if (CACHED) {
          sx = crd[k].x;
          sy = crd[k].y;
        } else {
          project_norm(cm[v], g, wx, wy, fz[gzi], sx, sy);
        }
        int x0, y0;
        tap_origin(sx, sy, W, H, x0, y0, o.w[k], o.inside[k]);
        o.xy[k] = int((unsigned(y0) << 16) | (unsigned(x0) & 0xffffu));
        if (o.inside[k]) {
          mnx = imax(mnx, -imax(x0, 0)); mxx = imax(mxx, imin(x0 + 1, W - 1));
          mny = imax(mny, -imax(y0, 0)); mxy = imax(mxy, imin(y0 + 1, H - 1));
        }
      }
    }
    mnx = row_max(mnx); mxx = row_max(mxx); mny = row_max(mny); mxy = row_max(mxy);
    if ((lane & 15) == 0 && mxx != INT_MIN) {
      int* rc = rect + 4 * (v % 3);
      atomicMax(&rc[0], mnx); atomicMax(&rc[1], mxx); atomicMax(&rc[2], mny); atomicMax(&rc[3], mxy);
    }
  };
  // staged: fits one tile (DMA issued one view ahead); big: fits the two tiles together (staged when its turn comes,
  // nothing overlapped); neither: sampled from global memory
  struct Rect { int x0, y0, w, h, pitch; bool any, staged, big, lds; };
  auto read_rect = [&](int v) {
    const int* rc = rect + 4 * (v % 3);
    // workgroup-uniform values: keep them in scalar registers
    const int c0 = __builtin_amdgcn_readfirstlane(rc[0]), c1 = __builtin_amdgcn_readfirstlane(rc[1]);
    const int c2 = __builtin_amdgcn_readfirstlane(rc[2]), c3 = __builtin_amdgcn_readfirstlane(rc[3]);
    Rect r;
    r.any = c1 != INT_MIN;                                               // some tap of the block hits this view's image
    r.x0 = r.any ? -c0 : 0;
    r.y0 = r.any ? -c2 : 0;
    r.w = r.any ? c1 - r.x0 + 1 : 0;
    r.h = r.any ? c3 - r.y0 + 1 : 0;
    r.pitch = r.w | 1;                                                   // odd pixel pitch: rows start in different bank quarters
    r.staged = r.any && r.h * r.pitch <= cap_lim;                        // cap_lim = cap_px (tests may lower it)
    r.big = r.any && !r.staged && r.h * r.pitch <= 2 * cap_lim && (ablate & 16);
    r.lds = r.staged || r.big;
    return r;
  };
  // rectangle -> LDS tile, rows of pitch * JP floats.  A wave copies rows wave, wave + NW, ...: one LDS-DMA
  // instruction per 64 quads of a row (lane = quad, so no index division; the pad column re-reads the last pixel)
  auto issue_dma = [&](int v, const Rect& r) {
    if (!r.lds || (ablate & 2)) return;
    float* tile = r.big ? smem : smem + (v & 1) * tile_sz;
    const float* plane = frame + size_t(v) * view_stride + (size_t(r.y0) * W + r.x0) * JP;
    const int pq = r.pitch * qn;                                         // quads per tile row
    for (int c0 = 0; c0 < pq; c0 += 64) {
      const int col = c0 + lane;
      int px = qn == 4 ? col >> 2 : (qn == 1 ? col : int(__umulhi(unsigned(col), m_qn)));
      const int cq = col - px * qn;
      px = px < r.w ? px : r.w - 1;
      const float* src0 = plane + size_t(px) * JP + 4 * cq;
      for (int row = wave; row < r.h; row += NT / 64) {
        if (col < pq)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + size_t(row) * W * JP),
                                           (__attribute__((address_space(3))) void*)(tile + size_t(row * pq + c0) * 4), 16, 0, 0);
      }
    }
  };
  // descriptor of own voxel k relative to view v's rectangle (staged) or to the global plane (fallback: pitch = W)
  auto finish = [&](const Own& o, int k, const Rect& r) {
    TapL d;
    const int x0 = (o.xy[k] << 16) >> 16, y0 = o.xy[k] >> 16;
    const int rx0 = r.lds ? r.x0 : 0, ry0 = r.lds ? r.y0 : 0;
    const int rx1 = r.lds ? r.x0 + r.w - 1 : W - 1, ry1 = r.lds ? r.y0 + r.h - 1 : H - 1;
    const int pitch = r.lds ? r.pitch : W;
    const int cx0 = imin(imax(x0, rx0), rx1), cx1 = imin(imax(x0 + 1, rx0), rx1);
    const int cy0 = imin(imax(y0, ry0), ry1), cy1 = imin(imax(y0 + 1, ry0), ry1);
    d.base = ((cy0 - ry0) * pitch + (cx0 - rx0)) * JP;
    d.dx = (cx1 - cx0) * JP;
    d.dy = (cy1 - cy0) * pitch * JP;
#pragma unroll
    for (int c = 0; c < 4; ++c) d.w[c] = ((o.inside[k] >> c) & 1) ? o.w[k][c] : 0.0f;
    return d;
  };

  for (int gz0 = s2; gz0 < e2; gz0 += BZ) {                             // z blocks of the window
    if (t < 12) rect[t] = INT_MIN;
    __syncthreads();
    float acc[VPT][NVL][4];
#pragma unroll
    for (int i = 0; i < VPT; ++i)
#pragma unroll
      for (int n = 0; n < NVL; ++n)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][n][c] = 0.0f;

    Own cur, nxt;
    if (CACHED) load_coords(0, gz0);
    project(0, gz0, cur);
    __syncthreads();
    Rect rcur = read_rect(0);
    if (rcur.staged) issue_dma(0, rcur);                                 // (a two-tile rectangle is staged inside the loop)
    if (CACHED && V > 1) load_coords(1, gz0);
    for (int v = 0; v < V; ++v) {
      if (v + 1 < V) project(v + 1, gz0, nxt);                           // overlaps view v's DMA
      wait_vmcnt(0);                                                     // this wave's items of view v have landed
      __syncthreads();                                                   // view v staged for everybody; rectangle v+1 complete;
                                                                         // tile (v+1)&1 is free (view v-1 has been sampled)
      Rect rnxt = rcur;
      if (v + 1 < V) rnxt = read_rect(v + 1);
      if (rcur.big) {
        // view v's rectangle needs both tiles: they are free now (view v-1 has been sampled, view v+1 is not issued)
        issue_dma(v, rcur);
        if (CACHED && v + 2 < V) load_coords(v + 2, gz0);
        wait_vmcnt(0);
        __syncthreads();
      } else if (v + 1 < V) {
        if (rnxt.staged) issue_dma(v + 1, rnxt);                         // overlaps the sampling of view v
        if (CACHED && v + 2 < V) load_coords(v + 2, gz0);                // consumed by the next iteration's project()
      }
      if (t < 4) rect[4 * (v % 3) + t] = INT_MIN;                        // slot of view v (= view v+3): all its readers passed the barrier
      // ---- sample the thread's 8 voxels of view v (owner lane i & 3 holds the descriptor of voxel i)
      if (rcur.any && !(ablate & 1) && (rcur.lds || !(ablate & 8))) {   // (8: skip the global-gather fallback)
        // Two pointers, one derived from the LDS array only and one global, and a uniform branch around the four tap
        // loads: a single pointer selected at run time (rounds 1-2) forced FLAT loads, which take the texture-addresser
        // path even when they land in LDS, plus a full vmcnt(0) wait per voxel - staging bought nothing (490 us with
        // every rectangle gathered, 490 staged).  Only the loads are duplicated; duplicating the whole sampling code
        // cost 37 spilled registers (645 us).
        const bool from_lds = rcur.lds;
        const float* lsrc = smem + ((rcur.big || !(v & 1)) ? 0 : tile_sz);
        const float* gsrc = frame + size_t(v) * view_stride;
        // NVL == 1 (JP = 16) keeps the single run-time pointer (FLAT loads): measured 490 us against 496 with the split
        // loads (12 spilled registers); NVL == 2 gains from the split (Shelf 764 -> 674 us, no spills)
        constexpr bool EXPLICIT_AS = (FVP_TRI_EXPLICIT_AS == 1) ? NVL > 1 : (FVP_TRI_EXPLICIT_AS > 1);
        const float* src = from_lds ? lsrc : gsrc;
        auto sample = [&](const TapL& tv, float (&a)[NVL][4]) {
#pragma unroll
          for (int n = 0; n < NVL; ++n) {
            const int ch0 = 16 * n + 4 * q;
            if (ch0 < JP) {
              float4 v0, v1, v2, v3;
              if constexpr (!EXPLICIT_AS) {
                const float* p0 = src + tv.base + ch0;
                v0 = *reinterpret_cast<const float4*>(p0);
                v1 = *reinterpret_cast<const float4*>(p0 + tv.dx);
                v2 = *reinterpret_cast<const float4*>(p0 + tv.dy);
                v3 = *reinterpret_cast<const float4*>(p0 + tv.dy + tv.dx);
              } else if (from_lds) {
                const float* p0 = lsrc + tv.base + ch0;
                v0 = lds_ld4(p0);
                v1 = lds_ld4(p0 + tv.dx);
                v2 = lds_ld4(p0 + tv.dy);
                v3 = lds_ld4(p0 + tv.dy + tv.dx);
              } else {
                const float* p0 = gsrc + tv.base + ch0;
                v0 = glb_ld4(p0);
                v1 = glb_ld4(p0 + tv.dx);
                v2 = glb_ld4(p0 + tv.dy);
                v3 = glb_ld4(p0 + tv.dy + tv.dx);
              }
#if FVP_TRI_PACKED
              // packed fp32 on register pairs: IEEE per element, the same operation order per channel (same bits)
              const f32x2 w0 = f32x2{tv.w[0], tv.w[0]}, w1 = f32x2{tv.w[1], tv.w[1]}, w2 = f32x2{tv.w[2], tv.w[2]},
                          w3 = f32x2{tv.w[3], tv.w[3]};
              f32x2 lo = f32x2{v0.x, v0.y} * w0, hi = f32x2{v0.z, v0.w} * w0;
              lo = __builtin_elementwise_fma(f32x2{v1.x, v1.y}, w1, lo);
              hi = __builtin_elementwise_fma(f32x2{v1.z, v1.w}, w1, hi);
              lo = __builtin_elementwise_fma(f32x2{v2.x, v2.y}, w2, lo);
              hi = __builtin_elementwise_fma(f32x2{v2.z, v2.w}, w2, hi);
              lo = __builtin_elementwise_fma(f32x2{v3.x, v3.y}, w3, lo);
              hi = __builtin_elementwise_fma(f32x2{v3.z, v3.w}, w3, hi);
              const f32x2 alo = f32x2{a[n][0], a[n][1]} + lo, ahi = f32x2{a[n][2], a[n][3]} + hi;
              a[n][0] = alo.x; a[n][1] = alo.y; a[n][2] = ahi.x; a[n][3] = ahi.y;
#else
              float s0_ = __fmul_rn(v0.x, tv.w[0]), s1_ = __fmul_rn(v0.y, tv.w[0]);
              float s2_ = __fmul_rn(v0.z, tv.w[0]), s3_ = __fmul_rn(v0.w, tv.w[0]);
              s0_ = __fmaf_rn(v1.x, tv.w[1], s0_); s1_ = __fmaf_rn(v1.y, tv.w[1], s1_);
              s2_ = __fmaf_rn(v1.z, tv.w[1], s2_); s3_ = __fmaf_rn(v1.w, tv.w[1], s3_);
              s0_ = __fmaf_rn(v2.x, tv.w[2], s0_); s1_ = __fmaf_rn(v2.y, tv.w[2], s1_);
              s2_ = __fmaf_rn(v2.z, tv.w[2], s2_); s3_ = __fmaf_rn(v2.w, tv.w[2], s3_);
              s0_ = __fmaf_rn(v3.x, tv.w[3], s0_); s1_ = __fmaf_rn(v3.y, tv.w[3], s1_);
              s2_ = __fmaf_rn(v3.z, tv.w[3], s2_); s3_ = __fmaf_rn(v3.w, tv.w[3], s3_);
              a[n][0] = __fadd_rn(a[n][0], s0_); a[n][1] = __fadd_rn(a[n][1], s1_);
              a[n][2] = __fadd_rn(a[n][2], s2_); a[n][3] = __fadd_rn(a[n][3], s3_);
#endif
            }
          }
        };
#pragma unroll
        for (int k = 0; k < OWN; ++k) {
          const TapL mine = finish(cur, k, rcur);
          { const TapL tv = quad_bcast_l<0>(mine); sample(tv, acc[4 * k + 0]); }
          { const TapL tv = quad_bcast_l<1>(mine); sample(tv, acc[4 * k + 1]); }
          { const TapL tv = quad_bcast_l<2>(mine); sample(tv, acc[4 * k + 2]); }
          { const TapL tv = quad_bcast_l<3>(mine); sample(tv, acc[4 * k + 3]); }
        }
      }
      if (rcur.big) {
        __syncthreads();                                                 // everybody is done with the two-tile rectangle
        if (v + 1 < V && rnxt.staged) issue_dma(v + 1, rnxt);            // (not overlapped: the rare path)
      }
      cur = nxt;
      rcur = rnxt;
    }
    // ---- mean over views, clamp, block maxima through LDS (tile 0 is free: every sampler passes the barrier
    //      below), then into the global planes
    __syncthreads();
    if (ablate & 4) continue;                                            // diagnostics (FVP_TRI_ABLATE): uniform
    int* cxy = reinterpret_cast<int*>(smem);                             // [kBX * kBY columns][JP]
    int* cxz = cxy + kBX * kBY * JP;                                     // [kBX][BZ][JP]
    int* cyz = cxz + kBX * BZ * JP;                                     // [kBY][BZ][JP]
    const int ncell = (kBX * kBY + (kBX + kBY) * BZ) * JP;              // <= cap_px * JP (checked by the host)
    for (int i = t; i < ncell; i += NT) cxy[i] = 0;
    __syncthreads();
    // The block's maxima are taken over the RAW view sums (clamped at 0 so that they order like ints); mean and clamp
    // are applied once per plane cell below: v -> clamp(v / V, 0, 1) is monotone, so max commutes with it and the planes
    // are bit-equal to dividing every voxel first (16 384 IEEE divisions per block pass became 6 240).
#pragma unroll
    for (int n = 0; n < NVL; ++n)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ch = 16 * n + 4 * q + c;
        float mz = 0.0f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
          const float val = fmaxf(acc[i][n][c], 0.0f);
          mz = fmaxf(mz, val);
          if (val > 0.0f && ch < JP) {
            const int z = zs + 4 * i;
            atomicMax(&cxz[(xx * BZ + z) * JP + ch], __float_as_int(val));
            atomicMax(&cyz[(yy * BZ + z) * JP + ch], __float_as_int(val));
          }
        }
        // the four zs lanes of a column sit 4 lanes apart inside one DPP row: rotate by 4 and 8
        int mi = __float_as_int(mz);                                     // non-negative floats order like ints
        mi = imax(mi, dpp_i<0x124>(mi));
        mi = imax(mi, dpp_i<0x128>(mi));
        if (zs == 0 && mi > 0 && ch < JP) cxy[(xx * kBY + yy) * JP + ch] = mi;
      }
    __syncthreads();
    const int lz0 = gz0 - tl2;
    for (int i = t; i < kBX * kBY * J; i += NT) {                        // xy: cell (x, y) of channel ch
      const int ch = i / (kBX * kBY), col = i - ch * (kBX * kBY), cx = col / kBY, cy = col - cx * kBY;
      const int raw = cxy[col * JP + ch];
      if (raw > 0 && gx0 + cx < e0 && gy0 + cy < e1) {
        const int vv = __float_as_int(clampf(__fdiv_rn(__int_as_float(raw), nv), 0.0f, 1.0f));
        if (vv > 0) plane_max(&pxy[size_t(ch) * CC + (gx0 + cx - tl0) * C + (gy0 + cy - tl1)], vv);
      }
    }
    for (int i = t; i < (kBX + kBY) * BZ * J; i += NT) {                // xz then yz rows: z fastest
      const int ch = i / ((kBX + kBY) * BZ), r = i - ch * ((kBX + kBY) * BZ), a = r / BZ, z = r - a * BZ;
      if (gz0 + z < e2) {
        const int raw = cxz[r * JP + ch];                                // rows kBX.. continue into cyz
        const int vv = raw > 0 ? __float_as_int(clampf(__fdiv_rn(__int_as_float(raw), nv), 0.0f, 1.0f)) : 0;
        if (vv > 0) {
          if (a < kBX) {
            if (gx0 + a < e0) plane_max(&pxz[size_t(ch) * CC + (gx0 + a - tl0) * C + lz0 + z], vv);
          } else if (gy0 + a - kBX < e1) {
            plane_max(&pyz[size_t(ch) * CC + (gy0 + a - kBX - tl1) * C + lz0 + z], vv);
          }
        }
      }
    }
    __syncthreads();                                                     // the tiles are reused by the next z block
  }
}

#endif  // FVP_DIAG

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the quad form WITHOUT LDS staging.  Measured in round 3: with every rectangle forced onto the global-gather
// path the staged quad kernel takes 493 us against 490 - its sampling is bound by the instruction stream, and the compact
// 8 x 4 x 32 voxel block (not the LDS tile) is what made it faster than the round-1 gather kernel: neighbouring voxels'
// taps hit the CU's L1 / the XCD's L2.  So the staging machinery - rectangle reductions (DPP + LDS atomics), the tile DMA,
// one workgroup barrier per view, two 36 KB tiles - buys the JP = 16 form nothing and costs the "skeleton" time.  This
// kernel keeps the block shape, the lane mapping, the coordinate cache and the plane-maxima phase, and samples every tap
// with global_load_dwordx4 (address-space-qualified: no FLAT): no barrier inside the view loop, waves run free, LDS only
// for the block's plane cells.  Same arithmetic per sample in the same order: bit-equal planes.
#ifndef FVP_TRI_BLK_OCC
#define FVP_TRI_BLK_OCC 4
#endif
#ifndef FVP_TRI_BLK_BX
#define FVP_TRI_BLK_BX 4            // x extent of the voxel block: 8 -> 512 threads, 4 -> 256 threads (finer phase interleaving)
#endif
// Block shape, measured (80 people, same box): 8 x 4 x 32 voxels / 512 threads 396 us; 4 x 4 x 32 / 256 threads 363 us (more,
// smaller workgroups per CU: the plane-maxima phase of one lies beside the sampling of the others); 2 x 4 x 32 410 us;
// 4 x 4 x 16 346-352 us.  Ablations of the 8 x 4 x 32 form: 395 us = 247 us sampling (10.5 GB of 64-byte taps: the L1's
// 64 B/clk/CU line rate gives 268 us) + 81 us plane maxima + 67 us coordinates / tap descriptors.
#ifndef FVP_TRI_BLK_BZ
#define FVP_TRI_BLK_BZ 16
#endif
constexpr int kBlkBX = FVP_TRI_BLK_BX, kBlkBZ = FVP_TRI_BLK_BZ;
constexpr int kBlkThreads = kBlkBX * kBY * 4 * 4;
#ifndef FVP_TRI_ZRES_MAX_KB
#define FVP_TRI_ZRES_MAX_KB 48
#endif
constexpr size_t kTriZresMaxLds = size_t(FVP_TRI_ZRES_MAX_KB) * 1024;   // z-resident plane cells per workgroup (ZRES form)
// ZRES (round 5): the x-z / y-z cells of the workgroup's WHOLE z range (2^zsh >= C cells per row) stay in LDS across its
// z blocks and the x-y maxima stay in registers (the same lane owns a column in every z block): one zero fill, no barrier
// inside the z loop, one merge into the global planes at the end - instead of zero fill + 3 barriers + merge per z block.
// max is order-independent: same bits.
// Q5 (round 6; JP = 20: Shelf / Campus, J = 17): the channels are FIVE quads.  With two quad groups per lane (NVL = 2) the second
// group holds one quad, so its sampling pass ran with three of a voxel's four lanes idle - 8 tap passes per four voxels where 5
// would do.  Here the four broadcast passes sample channels 0-15 as for JP = 16, and ONE more pass samples channels 16-19 of all
// four voxels at once: lane q takes voxel q's fifth quad with the tap descriptor it computed itself (no broadcast).  Same
// arithmetic per sample, so the same bits; the accumulators of the fifth quad belong to (voxel 4k + q), which the plane-maxima
// phase accounts for (its x-y maximum is reduced over the q lanes too).
template <int NVL, bool CACHED, bool ZRES, bool Q5 = false>
__global__ void __launch_bounds__(kBlkThreads, FVP_TRI_BLK_OCC)
k_project_triplane_blk(const float* __restrict__ heat_cl, const Cam* __restrict__ cams, const int* __restrict__ frame_set,
                       const int* __restrict__ person_frame, const uint8_t* __restrict__ person_valid,
                       const int* __restrict__ boxes, const float* __restrict__ fx, const float* __restrict__ fy,
                       const float* __restrict__ fz, int C, int nP, int nbx, int nby, int ppf, FvpGeom g,
                       const float* __restrict__ fgrid, int F0, int F1, int F2, float* __restrict__ planes, int zsh) {
  static_assert(!Q5 || (NVL == 2 && !ZRES), "the five-quad form replaces the two-group form of the per-z-block kernel");
  constexpr int NVA = Q5 ? 1 : NVL;                  // quad groups sampled by the broadcast passes
  constexpr int BZ = NVL == 1 ? kBlkBZ : 16;
  constexpr int VPT = BZ / 4;
  constexpr int OWN = VPT / 4;
  HIP_DYNAMIC_SHARED(float, smem)                     // the block's plane cells (ints)
  constexpr int NT = kBlkThreads;
  constexpr int kBX = kBlkBX;                        // (shadows the staged kernels' block extent)
  const int J = g.J, JP = g.JP, CC = C * C, V = g.V, W = g.W, H = g.H;
  int p, blk;
  {
    const int id = blockIdx.x;
    const int bpp = nbx * nby, bpf = ppf * bpp;
    const int nframes = nP / ppf;
    if (nframes % 8 == 0) {
      const int xcd = id & 7, j = id >> 3;
      const int frame = xcd + 8 * (j / bpf), r = j % bpf;
      p = frame * ppf + r / bpp;
      blk = r % bpp;
    } else {
      p = id / bpp;
      blk = id % bpp;
    }
  }
  if (person_valid && !person_valid[p]) return;
  const int* bx = boxes + p * 9;
  const int tl0 = bx[0], tl1 = bx[1], tl2 = bx[2];
  const int s0 = bx[3], s1 = bx[4], s2 = bx[5], e0 = bx[6], e1 = bx[7], e2 = bx[8];
  if (s0 >= e0 || s1 >= e1 || s2 >= e2) return;
  const int xb = blk / nby, yb = blk - xb * nby;
  const int gx0 = s0 + xb * kBX, gy0 = s1 + yb * kBY;
  if (gx0 >= e0 || gy0 >= e1) return;

  const int t = threadIdx.x, q = t & 3;
  const int slot = t >> 2, zs = slot & 3, yy = (slot >> 2) & (kBY - 1), xx = slot / (4 * kBY);
  const int gxi = gx0 + xx, gyi = gy0 + yy;
  const bool col_in = gxi < e0 && gyi < e1;
  const int b = person_frame[p];
  const size_t view_stride = size_t(H) * W * JP;
  const float* frame = heat_cl + size_t(b) * V * view_stride;
  const Cam* cm = cams + size_t(frame_set[b]) * V;
  const float wx = col_in ? fx[gxi] : 0.0f, wy = col_in ? fy[gyi] : 0.0f;
  const size_t nfine = size_t(F0) * F1 * F2;
  const float2* gcol = CACHED ? reinterpret_cast<const float2*>(fgrid) + size_t(frame_set[b]) * V * nfine +
                                    (size_t(col_in ? gxi : 0) * F1 + (col_in ? gyi : 0)) * F2
                              : nullptr;
  float* pxy = planes + (size_t(p) * 3 + 0) * J * CC;
  float* pxz = planes + (size_t(p) * 3 + 1) * J * CC;
  float* pyz = planes + (size_t(p) * 3 + 2) * J * CC;
  const float nv = float(V);
  const int JPp = JP + 1;                                                // cell pitch (see below)
  [[maybe_unused]] int mxy[NVL][4];                                      // ZRES: running x-y maxima of this lane's column
  [[maybe_unused]] const int zt = 1 << zsh;
  if constexpr (ZRES) {
#pragma unroll
    for (int n = 0; n < NVL; ++n)
#pragma unroll
      for (int c = 0; c < 4; ++c) mxy[n][c] = 0;
    int* cz = reinterpret_cast<int*>(smem);                              // [kBX + kBY rows][zt][JPp]
    const int ncell = ((kBX + kBY) << zsh) * JPp;
    for (int i = t; i < ncell; i += NT) cz[i] = 0;
    __syncthreads();
  }

  for (int gz0 = s2; gz0 < e2; gz0 += BZ) {
    float acc[VPT][NVA][4];
#pragma unroll
    for (int i = 0; i < VPT; ++i)
#pragma unroll
      for (int n = 0; n < NVA; ++n)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][n][c] = 0.0f;
    [[maybe_unused]] float accx[OWN][1][4];                              // Q5: channels 16-19 of voxel 4 k + q
#pragma unroll
    for (int kk = 0; kk < OWN; ++kk)
#pragma unroll
      for (int c = 0; c < 4; ++c) accx[kk][0][c] = 0.0f;
    float2 crd[OWN];
    auto load_coords = [&](int v) {
#pragma unroll
      for (int k = 0; k < OWN; ++k) {
        const int gzi = gz0 + zs + 4 * (q + 4 * k);
        crd[k] = gcol[size_t(v) * nfine + (gzi < F2 ? gzi : F2 - 1)];
      }
    };
    if (CACHED) load_coords(0);
    for (int v = 0; v < V; ++v) {
      // this lane's own voxels (i = q + 4 k) of view v: tap descriptor relative to the view's plane
      TapL mine[OWN];
#pragma unroll
      for (int k = 0; k < OWN; ++k) {
        const int gzi = gz0 + zs + 4 * (q + 4 * k);
        const bool vin = col_in && gzi < e2;
        TapL d;
        d.base = d.dx = d.dy = 0;
        d.w[0] = d.w[1] = d.w[2] = d.w[3] = 0.0f;
        if (vin) {
          float sx, sy;
          if (CACHED) {
            sx = crd[k].x;
            sy = crd[k].y;
          } else {
            project_norm(cm[v], g, wx, wy, fz[gzi], sx, sy);
          }
          int x0, y0, inside;
          float w4[4];
          tap_origin(sx, sy, W, H, x0, y0, w4, inside);
          const int cx0 = imin(imax(x0, 0), W - 1), cx1 = imin(imax(x0 + 1, 0), W - 1);
          const int cy0 = imin(imax(y0, 0), H - 1), cy1 = imin(imax(y0 + 1, 0), H - 1);
          d.base = (cy0 * W + cx0) * JP;
          d.dx = (cx1 - cx0) * JP;
          d.dy = (cy1 - cy0) * W * JP;
#pragma unroll
          for (int c = 0; c < 4; ++c) d.w[c] = ((inside >> c) & 1) ? w4[c] : 0.0f;
        }
        mine[k] = d;
      }
      if (CACHED && v + 1 < V) load_coords(v + 1);     // next view's coordinates under this view's sampling
#ifdef FVP_TRI_BLK_NOSAMPLE
      if (mine[0].base != -12345) { acc[0][0][0] += mine[0].w[0] + mine[OWN - 1].w[3] + float(mine[0].base); continue; }
#endif
      const float* gsrc = frame + size_t(v) * view_stride;
      auto sample = [&](const TapL& tv, auto& a, auto own) {
        constexpr bool kOwn = decltype(own)::value;                      // Q5's extra pass: channels 16-19, every lane
#pragma unroll
        for (int n = 0; n < (kOwn ? 1 : NVA); ++n) {
          const int ch0 = kOwn ? 16 : 16 * n + 4 * q;
          if (kOwn || ch0 < JP) {
            const float* p0 = gsrc + tv.base + ch0;
            const float4 v0 = glb_ld4(p0), v1 = glb_ld4(p0 + tv.dx), v2 = glb_ld4(p0 + tv.dy), v3 = glb_ld4(p0 + tv.dy + tv.dx);
            const f32x2 w0 = f32x2{tv.w[0], tv.w[0]}, w1 = f32x2{tv.w[1], tv.w[1]}, w2 = f32x2{tv.w[2], tv.w[2]},
                        w3 = f32x2{tv.w[3], tv.w[3]};
            f32x2 lo = f32x2{v0.x, v0.y} * w0, hi = f32x2{v0.z, v0.w} * w0;
            lo = __builtin_elementwise_fma(f32x2{v1.x, v1.y}, w1, lo);
            hi = __builtin_elementwise_fma(f32x2{v1.z, v1.w}, w1, hi);
            lo = __builtin_elementwise_fma(f32x2{v2.x, v2.y}, w2, lo);
            hi = __builtin_elementwise_fma(f32x2{v2.z, v2.w}, w2, hi);
            lo = __builtin_elementwise_fma(f32x2{v3.x, v3.y}, w3, lo);
            hi = __builtin_elementwise_fma(f32x2{v3.z, v3.w}, w3, hi);
            const f32x2 alo = f32x2{a[n][0], a[n][1]} + lo, ahi = f32x2{a[n][2], a[n][3]} + hi;
            a[n][0] = alo.x; a[n][1] = alo.y; a[n][2] = ahi.x; a[n][3] = ahi.y;
          }
        }
      };
#pragma unroll
      for (int k = 0; k < OWN; ++k) {
        { const TapL tv = quad_bcast_l<0>(mine[k]); sample(tv, acc[4 * k + 0], std::false_type{}); }
        { const TapL tv = quad_bcast_l<1>(mine[k]); sample(tv, acc[4 * k + 1], std::false_type{}); }
        { const TapL tv = quad_bcast_l<2>(mine[k]); sample(tv, acc[4 * k + 2], std::false_type{}); }
        { const TapL tv = quad_bcast_l<3>(mine[k]); sample(tv, acc[4 * k + 3], std::false_type{}); }
        if constexpr (Q5) sample(mine[k], accx[k], std::true_type{});
      }
    }
#ifdef FVP_TRI_BLK_NOMAX          // (ablation variants, tools/build_variant.sh: wrong results)
    if (acc[0][0][0] != 1.2345e-30f) continue;
#endif
    // ---- block maxima through LDS, then into the global planes (as k_project_triplane_lds)
    // Cell rows are JPp = JP + 1 words apart (round 5): with the natural pitch of 16 the lanes of an atomic instruction -
    // (channel quad q, z phase zs, y) -> word (cell * 16 + 4 q + c) - fall on 8 of the 32 banks (z phases 0 / 2 and the
    // y pairs of a 32-lane group share them): SQ_LDS_BANK_CONFLICT was 7.6 cycles per LDS instruction of this kernel.
    // With the odd pitch the 32 lanes of a group hit 32 different banks; only the same-address pairs of the x-z cells
    // (the two y lanes of a group) still serialise.
    if constexpr (ZRES) {
      int* rxz = reinterpret_cast<int*>(smem);                           // [kBX][zt][JPp]
      int* ryz = rxz + ((kBX << zsh) * JPp);                             // [kBY][zt][JPp]
      const int zb = gz0 - s2 + zs;
#pragma unroll
      for (int n = 0; n < NVL; ++n)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int ch = 16 * n + 4 * q + c;
          float mz = 0.0f;
#pragma unroll
          for (int i = 0; i < VPT; ++i) {
            const float val = fmaxf(acc[i][n][c], 0.0f);
            mz = fmaxf(mz, val);
            if (val > 0.0f && ch < JP) {
              const int z = zb + 4 * i;
              atomicMax(&rxz[((xx << zsh) + z) * JPp + ch], __float_as_int(val));
              atomicMax(&ryz[((yy << zsh) + z) * JPp + ch], __float_as_int(val));
            }
          }
          int mi = __float_as_int(mz);
          mi = imax(mi, dpp_i<0x124>(mi));
          mi = imax(mi, dpp_i<0x128>(mi));
          mxy[n][c] = imax(mxy[n][c], mi);
        }
      continue;
    }
    int* cxy = reinterpret_cast<int*>(smem);                             // [kBX * kBY columns][JPp]
    int* cxz = cxy + kBX * kBY * JPp;                                    // [kBX][BZ][JPp]
    int* cyz = cxz + kBX * BZ * JPp;                                     // [kBY][BZ][JPp]
    const int ncell = (kBX * kBY + (kBX + kBY) * BZ) * JPp;
    __syncthreads();                                                     // (the previous z block's readers are done)
    for (int i = t; i < ncell; i += NT) cxy[i] = 0;
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NVA; ++n)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ch = 16 * n + 4 * q + c;
        float mz = 0.0f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
          const float val = fmaxf(acc[i][n][c], 0.0f);
          mz = fmaxf(mz, val);
          if (val > 0.0f && ch < JP) {
            const int z = zs + 4 * i;
            atomicMax(&cxz[(xx * BZ + z) * JPp + ch], __float_as_int(val));
            atomicMax(&cyz[(yy * BZ + z) * JPp + ch], __float_as_int(val));
          }
        }
        // (reducing the four y lanes of an (x, z) cell with DPP row rotates before one of them touches LDS - lanes as
        // q + 4 y + 16 zs - measured slower: 371 vs 361 us)
        int mi = __float_as_int(mz);
        mi = imax(mi, dpp_i<0x124>(mi));
        mi = imax(mi, dpp_i<0x128>(mi));
        if (zs == 0 && mi > 0 && ch < JP) cxy[(xx * kBY + yy) * JPp + ch] = mi;
      }
    if constexpr (Q5) {
      // the fifth quad: this lane holds channels 16-19 of voxel i = 4 k + q (z = zs + 4 i) of its column; the column's x-y
      // maximum is over all 16 lanes of the DPP row (q and zs)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ch = 16 + c;
        float mz = 0.0f;
#pragma unroll
        for (int kk = 0; kk < OWN; ++kk) {
          const float val = fmaxf(accx[kk][0][c], 0.0f);
          mz = fmaxf(mz, val);
          if (val > 0.0f) {
            const int z = zs + 4 * (4 * kk + q);
            atomicMax(&cxz[(xx * BZ + z) * JPp + ch], __float_as_int(val));
            atomicMax(&cyz[(yy * BZ + z) * JPp + ch], __float_as_int(val));
          }
        }
        int mi = __float_as_int(mz);
        mi = imax(mi, dpp_i<0x121>(mi));
        mi = imax(mi, dpp_i<0x122>(mi));
        mi = imax(mi, dpp_i<0x124>(mi));
        mi = imax(mi, dpp_i<0x128>(mi));
        if (zs == 0 && q == 0 && mi > 0) cxy[(xx * kBY + yy) * JPp + ch] = mi;
      }
    }
    __syncthreads();
    const int lz0 = gz0 - tl2;
    for (int i = t; i < kBX * kBY * J; i += NT) {
      const int ch = i / (kBX * kBY), col = i - ch * (kBX * kBY), cx = col / kBY, cy = col - cx * kBY;
      const int raw = cxy[col * JPp + ch];
      if (raw > 0 && gx0 + cx < e0 && gy0 + cy < e1) {
        const int vv = __float_as_int(clampf(__fdiv_rn(__int_as_float(raw), nv), 0.0f, 1.0f));
        if (vv > 0) plane_max(&pxy[size_t(ch) * CC + (gx0 + cx - tl0) * C + (gy0 + cy - tl1)], vv);
      }
    }
    for (int i = t; i < (kBX + kBY) * BZ * J; i += NT) {
      const int ch = i / ((kBX + kBY) * BZ), r = i - ch * ((kBX + kBY) * BZ), a = r / BZ, z = r - a * BZ;
      if (gz0 + z < e2) {
        const int raw = cxz[r * JPp + ch];
        const int vv = raw > 0 ? __float_as_int(clampf(__fdiv_rn(__int_as_float(raw), nv), 0.0f, 1.0f)) : 0;
        if (vv > 0) {
          if (a < kBX) {
            if (gx0 + a < e0) plane_max(&pxz[size_t(ch) * CC + (gx0 + a - tl0) * C + lz0 + z], vv);
          } else if (gy0 + a - kBX < e1) {
            plane_max(&pyz[size_t(ch) * CC + (gy0 + a - kBX - tl1) * C + lz0 + z], vv);
          }
        }
      }
    }
  }
  if constexpr (ZRES) {
    __syncthreads();                                                     // every wave's cell maxima are in
    if (zs == 0 && col_in) {
#pragma unroll
      for (int n = 0; n < NVL; ++n)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int ch = 16 * n + 4 * q + c;
          if (ch < J && mxy[n][c] > 0) {
            const int vv = __float_as_int(clampf(__fdiv_rn(__int_as_float(mxy[n][c]), nv), 0.0f, 1.0f));
            if (vv > 0) plane_max(&pxy[size_t(ch) * CC + (gxi - tl0) * C + (gyi - tl1)], vv);
          }
        }
    }
    const int* cz = reinterpret_cast<const int*>(smem);
    const int nz = e2 - s2, lz0 = s2 - tl2;
    const int rows = (kBX + kBY) << zsh;                                 // x-z rows then y-z rows, z fastest
    for (int i = t; i < rows * J; i += NT) {
      static_assert(kBX + kBY == 8, "row count is a power of two");
      const int ch = i >> (zsh + 3), r = i & (rows - 1), a = r >> zsh, z = r & (zt - 1);
      if (z < nz) {
        const int raw = cz[r * JPp + ch];
        const int vv = raw > 0 ? __float_as_int(clampf(__fdiv_rn(__int_as_float(raw), nv), 0.0f, 1.0f)) : 0;
        if (vv > 0) {
          if (a < kBX) {
            if (gx0 + a < e0) plane_max(&pxz[size_t(ch) * CC + (gx0 + a - tl0) * C + lz0 + z], vv);
          } else if (gy0 + a - kBX < e1) {
            plane_max(&pyz[size_t(ch) * CC + (gy0 + a - kBX - tl1) * C + lz0 + z], vv);
          }
        }
      }
    }
  }
}

#if FVP_DIAG
// ---------------------------------------------------------------------------------------------------------------
// Round 3: the same kernel with ONE LANE PER VOXEL (all JP channels) instead of four lanes per voxel (a channel quad
// each).  Ablations of the quad form (80 people, FVP_TRI_ABLATE): 494 us complete = 225 us sampling + 104 us
// plane maxima + ~175 us skeleton (projection, rectangles, barriers); the tile DMA is hidden (15 us).  The sampling is
// issue-bound, and most of what it issues is not arithmetic: a (voxel, view) costs 4 lanes x (4 ds_read_b128 + 20 VALU +
// 7 DPP broadcasts of the tap descriptor + address adds) = ~150 lane-instructions.  The quad form was right for the
// round-1 kernel that gathered through the texture path (a quad reads 64 contiguous bytes: 4x fewer cache lines per
// instruction); from LDS a ds_read_b128 costs the same whichever lanes issue it.  With a lane per voxel the descriptor
// stays in the lane that computed it (no broadcasts), the four taps of a channel quad are packed-fp32 operations on
// register pairs (v_pk_mul / v_pk_fma / v_pk_add: IEEE per element, same operation order per channel, so the same
// bits), and a (voxel, view) is NQ x (4 ds_read_b128 + 10 packed VALU) + a few address adds = ~60 lane-instructions.
// Thread t: z16 = t & 15, y = (t >> 4) & 3, x = t >> 6 (= wave), voxels z16 and z16 + 16 of the 8 x 4 x 32 block.
template <int NQ, bool CACHED>   // NQ = JP / 4 channel quads per pixel
__global__ void __launch_bounds__(kTriThreads, 4)
k_project_triplane_lds2(const float* __restrict__ heat_cl, const Cam* __restrict__ cams, const int* __restrict__ frame_set,
                        const int* __restrict__ person_frame, const uint8_t* __restrict__ person_valid,
                        const int* __restrict__ boxes, const float* __restrict__ fx, const float* __restrict__ fy,
                        const float* __restrict__ fz, int C, int nP, int nbx, int nby, int ppf, int cap_px, FvpGeom g,
                        const float* __restrict__ fgrid, int F0, int F1, int F2, float* __restrict__ planes, int ablate, int cap_lim) {
  constexpr int BZ = 32, VPT = 2, JP = 4 * NQ;
  // pixel pitch in the LDS tile: an ODD number of 16-byte quads (JP = 16: 80 bytes, the fifth quad is padding), so that
  // the 64 lanes of a ds_read_b128 - each at its own pixel - spread over all bank groups (with the 64-byte pitch of
  // the global layout they hit 4 of the 16: measured 668 us against 490 for the quad form)
  constexpr int LQ = NQ | 1, LP = 4 * LQ;
  HIP_DYNAMIC_SHARED(float, smem)
  constexpr int NT = kTriThreads;
  const int J = g.J, CC = C * C, V = g.V, W = g.W, H = g.H;
  const int tile_sz = cap_px * LP;
  int* rect = reinterpret_cast<int*>(smem + 2 * size_t(tile_sz));        // [3][4]: -minx, maxx, -miny, maxy
  int p, blk;
  {
    const int id = blockIdx.x;
    const int bpp = nbx * nby, bpf = ppf * bpp;
    const int nframes = nP / ppf;
    if (nframes % 8 == 0) {
      const int xcd = id & 7, j = id >> 3;
      const int frame = xcd + 8 * (j / bpf), r = j % bpf;
      p = frame * ppf + r / bpp;
      blk = r % bpp;
    } else {
      p = id / bpp;
      blk = id % bpp;
    }
  }
  if (person_valid && !person_valid[p]) return;
  const int* bx = boxes + p * 9;
  const int tl0 = bx[0], tl1 = bx[1], tl2 = bx[2];
  const int s0 = bx[3], s1 = bx[4], s2 = bx[5], e0 = bx[6], e1 = bx[7], e2 = bx[8];
  if (s0 >= e0 || s1 >= e1 || s2 >= e2) return;
  const int xb = blk / nby, yb = blk - xb * nby;
  const int gx0 = s0 + xb * kBX, gy0 = s1 + yb * kBY;
  if (gx0 >= e0 || gy0 >= e1) return;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int z16 = t & 15, yy = (t >> 4) & (kBY - 1), xx = t >> 6;
  const int gxi = gx0 + xx, gyi = gy0 + yy;
  const bool col_in = gxi < e0 && gyi < e1;
  const int b = person_frame[p];
  const size_t view_stride = size_t(H) * W * JP;
  const float* frame = heat_cl + size_t(b) * V * view_stride;
  const Cam* cm = cams + size_t(frame_set[b]) * V;
  const float wx = col_in ? fx[gxi] : 0.0f, wy = col_in ? fy[gyi] : 0.0f;
  const size_t nfine = size_t(F0) * F1 * F2;
  const float2* gcol = CACHED ? reinterpret_cast<const float2*>(fgrid) + size_t(frame_set[b]) * V * nfine +
                                    (size_t(col_in ? gxi : 0) * F1 + (col_in ? gyi : 0)) * F2
                              : nullptr;
  float* pxy = planes + (size_t(p) * 3 + 0) * J * CC;
  float* pxz = planes + (size_t(p) * 3 + 1) * J * CC;
  float* pyz = planes + (size_t(p) * 3 + 2) * J * CC;
  const float nv = float(V);

  struct Own { int xy[VPT]; int inside[VPT]; float w[VPT][4]; };
  float2 crd[VPT];
  auto load_coords = [&](int v, int gz0) {
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const int gzi = gz0 + z16 + 16 * k;
      crd[k] = gcol[size_t(v) * nfine + (gzi < F2 ? gzi : F2 - 1)];
    }
  };
  auto project = [&](int v, int gz0, Own& o) {
    int mnx = INT_MIN, mxx = INT_MIN, mny = INT_MIN, mxy = INT_MIN;     // (-min, max) pairs
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const int gzi = gz0 + z16 + 16 * k;
      const bool vin = col_in && gzi < e2;
      o.inside[k] = 0;
      o.xy[k] = 0;
      o.w[k][0] = o.w[k][1] = o.w[k][2] = o.w[k][3] = 0.0f;
      if (vin) {
        float sx, sy;
        if (CACHED) {
          sx = crd[k].x;
          sy = crd[k].y;
        } else {
          project_norm(cm[v], g, wx, wy, fz[gzi], sx, sy);
        }
        int x0, y0;
        tap_origin(sx, sy, W, H, x0, y0, o.w[k], o.inside[k]);
        o.xy[k] = int((unsigned(y0) << 16) | (unsigned(x0) & 0xffffu));
        if (o.inside[k]) {
          mnx = imax(mnx, -imax(x0, 0)); mxx = imax(mxx, imin(x0 + 1, W - 1));
          mny = imax(mny, -imax(y0, 0)); mxy = imax(mxy, imin(y0 + 1, H - 1));
        }
      }
    }
    mnx = row_max(mnx); mxx = row_max(mxx); mny = row_max(mny); mxy = row_max(mxy);
    if ((lane & 15) == 0 && mxx != INT_MIN) {
      int* rc = rect + 4 * (v % 3);
      atomicMax(&rc[0], mnx); atomicMax(&rc[1], mxx); atomicMax(&rc[2], mny); atomicMax(&rc[3], mxy);
    }
  };
  // staged: fits one tile (DMA issued one view ahead); big: fits the two tiles together (staged when its turn comes,
  // nothing overlapped); neither: sampled from global memory
  struct Rect { int x0, y0, w, h, pitch; bool any, staged, big, lds; };
  auto read_rect = [&](int v) {
    const int* rc = rect + 4 * (v % 3);
    const int c0 = __builtin_amdgcn_readfirstlane(rc[0]), c1 = __builtin_amdgcn_readfirstlane(rc[1]);
    const int c2 = __builtin_amdgcn_readfirstlane(rc[2]), c3 = __builtin_amdgcn_readfirstlane(rc[3]);
    Rect r;
    r.any = c1 != INT_MIN;
    r.x0 = r.any ? -c0 : 0;
    r.y0 = r.any ? -c2 : 0;
    r.w = r.any ? c1 - r.x0 + 1 : 0;
    r.h = r.any ? c3 - r.y0 + 1 : 0;
    r.pitch = r.w | 1;
    r.staged = r.any && r.h * r.pitch <= cap_lim;                        // cap_lim = cap_px (tests may lower it)
    r.big = r.any && !r.staged && r.h * r.pitch <= 2 * cap_lim && (ablate & 16);
    r.lds = r.staged || r.big;
    return r;
  };
  auto issue_dma = [&](int v, const Rect& r) {
    if (!r.lds || (ablate & 2)) return;
    float* tile = r.big ? smem : smem + (v & 1) * tile_sz;
    const float* plane = frame + size_t(v) * view_stride + (size_t(r.y0) * W + r.x0) * JP;
    const int pq = r.pitch * LQ;                                         // quads per tile row (incl. the padding quads)
    for (int c0 = 0; c0 < pq; c0 += 64) {
      const int col = c0 + lane;
      int px = col / LQ;                                                 // LQ is a compile-time constant
      int cq = col - px * LQ;
      cq = cq < NQ ? cq : NQ - 1;                                        // padding quad: re-read the last one
      px = px < r.w ? px : r.w - 1;
      const float* src0 = plane + size_t(px) * JP + 4 * cq;
      for (int row = wave; row < r.h; row += NT / 64) {
        if (col < pq)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + size_t(row) * W * JP),
                                           (__attribute__((address_space(3))) void*)(tile + size_t(row * pq + c0) * 4), 16, 0, 0);
      }
    }
  };
  auto finish = [&](const Own& o, int k, const Rect& r) {
    TapL d;
    const int x0 = (o.xy[k] << 16) >> 16, y0 = o.xy[k] >> 16;
    const int rx0 = r.lds ? r.x0 : 0, ry0 = r.lds ? r.y0 : 0;
    const int rx1 = r.lds ? r.x0 + r.w - 1 : W - 1, ry1 = r.lds ? r.y0 + r.h - 1 : H - 1;
    const int pitch = r.lds ? r.pitch : W;
    const int ps = r.lds ? LP : JP;                                   // floats per pixel: LDS tile / global plane
    const int cx0 = imin(imax(x0, rx0), rx1), cx1 = imin(imax(x0 + 1, rx0), rx1);
    const int cy0 = imin(imax(y0, ry0), ry1), cy1 = imin(imax(y0 + 1, ry0), ry1);
    d.base = ((cy0 - ry0) * pitch + (cx0 - rx0)) * ps;
    d.dx = (cx1 - cx0) * ps;
    d.dy = (cy1 - cy0) * pitch * ps;
#pragma unroll
    for (int c = 0; c < 4; ++c) d.w[c] = ((o.inside[k] >> c) & 1) ? o.w[k][c] : 0.0f;
    return d;
  };

  for (int gz0 = s2; gz0 < e2; gz0 += BZ) {
    if (t < 12) rect[t] = INT_MIN;
    __syncthreads();
    f32x2 acc[VPT][2 * NQ];                                              // channel pairs
#pragma unroll
    for (int k = 0; k < VPT; ++k)
#pragma unroll
      for (int c = 0; c < 2 * NQ; ++c) acc[k][c] = f32x2{0.0f, 0.0f};

    Own cur, nxt;
    if (CACHED) load_coords(0, gz0);
    project(0, gz0, cur);
    __syncthreads();
    Rect rcur = read_rect(0);
    if (rcur.staged) issue_dma(0, rcur);                                 // (a two-tile rectangle is staged inside the loop)
    if (CACHED && V > 1) load_coords(1, gz0);
    for (int v = 0; v < V; ++v) {
      if (v + 1 < V) project(v + 1, gz0, nxt);
      wait_vmcnt(0);
      __syncthreads();
      Rect rnxt = rcur;
      if (v + 1 < V) rnxt = read_rect(v + 1);
      if (rcur.big) {
        // view v's rectangle needs both tiles: they are free now (view v-1 has been sampled, view v+1 is not issued)
        issue_dma(v, rcur);
        if (CACHED && v + 2 < V) load_coords(v + 2, gz0);
        wait_vmcnt(0);
        __syncthreads();
      } else if (v + 1 < V) {
        if (rnxt.staged) issue_dma(v + 1, rnxt);                         // overlaps the sampling of view v
        if (CACHED && v + 2 < V) load_coords(v + 2, gz0);                // consumed by the next iteration's project()
      }
      if (t < 4) rect[4 * (v % 3) + t] = INT_MIN;                        // slot of view v (= view v+3): all its readers passed the barrier
      if (rcur.any && !(ablate & 1) && (rcur.lds || !(ablate & 8))) {   // (8: skip the global-gather fallback)
        const float* src = rcur.big ? smem : (rcur.staged ? smem + (v & 1) * tile_sz : frame + size_t(v) * view_stride);
        // (one run-time pointer = FLAT loads, unlike the quad form: with the loads split by address space the
        //  scheduler hoists all 4 NQ of them per voxel in both copies and spills 160-255 registers - 2.7 ms on Shelf)
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
          const TapL d = finish(cur, k, rcur);
          const float* p0 = src + d.base;
          const f32x2 w0 = f32x2{d.w[0], d.w[0]}, w1 = f32x2{d.w[1], d.w[1]}, w2 = f32x2{d.w[2], d.w[2]},
                      w3 = f32x2{d.w[3], d.w[3]};
#pragma unroll
          for (int n = 0; n < NQ; ++n) {
            const float4 v0 = *reinterpret_cast<const float4*>(p0 + 4 * n);
            const float4 v1 = *reinterpret_cast<const float4*>(p0 + d.dx + 4 * n);
            const float4 v2 = *reinterpret_cast<const float4*>(p0 + d.dy + 4 * n);
            const float4 v3 = *reinterpret_cast<const float4*>(p0 + d.dy + d.dx + 4 * n);
            // per channel: nw * w0, then fma(ne), fma(sw), fma(se), then the view-ordered add (the reference's order)
            f32x2 lo = f32x2{v0.x, v0.y} * w0, hi = f32x2{v0.z, v0.w} * w0;
            lo = __builtin_elementwise_fma(f32x2{v1.x, v1.y}, w1, lo);
            hi = __builtin_elementwise_fma(f32x2{v1.z, v1.w}, w1, hi);
            lo = __builtin_elementwise_fma(f32x2{v2.x, v2.y}, w2, lo);
            hi = __builtin_elementwise_fma(f32x2{v2.z, v2.w}, w2, hi);
            lo = __builtin_elementwise_fma(f32x2{v3.x, v3.y}, w3, lo);
            hi = __builtin_elementwise_fma(f32x2{v3.z, v3.w}, w3, hi);
            acc[k][2 * n] = acc[k][2 * n] + lo;
            acc[k][2 * n + 1] = acc[k][2 * n + 1] + hi;
          }
        }
      }
      if (rcur.big) {
        __syncthreads();                                                 // everybody is done with the two-tile rectangle
        if (v + 1 < V && rnxt.staged) issue_dma(v + 1, rnxt);            // (not overlapped: the rare path)
      }
      cur = nxt;
      rcur = rnxt;
    }
    __syncthreads();
    if (ablate & 4) continue;
    int* cxy = reinterpret_cast<int*>(smem);                             // [kBX * kBY columns][JP]
    int* cxz = cxy + kBX * kBY * JP;                                     // [kBX][BZ][JP]
    int* cyz = cxz + kBX * BZ * JP;                                      // [kBY][BZ][JP]
    constexpr int ncell = (kBX * kBY + (kBX + kBY) * BZ) * JP;           // <= cap_px * JP (checked by the host)
    for (int i = t; i < ncell; i += NT) cxy[i] = 0;
    __syncthreads();
    // maxima over the raw view sums (clamped at 0); mean + clamp once per plane cell below (monotone: same bits)
#pragma unroll
    for (int c2 = 0; c2 < 2 * NQ; ++c2)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ch = 2 * c2 + e;
        float mz = 0.0f;
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
          const float val = fmaxf(e ? acc[k][c2].y : acc[k][c2].x, 0.0f);
          mz = fmaxf(mz, val);
          if (val > 0.0f) {
            const int z = z16 + 16 * k;
            atomicMax(&cxz[(xx * BZ + z) * JP + ch], __float_as_int(val));
            atomicMax(&cyz[(yy * BZ + z) * JP + ch], __float_as_int(val));
          }
        }
        const int mi = row_max(__float_as_int(mz));                      // the 16 z-lanes of a column are one DPP row
        if (z16 == 0 && mi > 0) cxy[(xx * kBY + yy) * JP + ch] = mi;
      }
    __syncthreads();
    const int lz0 = gz0 - tl2;
    for (int i = t; i < kBX * kBY * J; i += NT) {
      const int ch = i / (kBX * kBY), col = i - ch * (kBX * kBY), cx = col / kBY, cy = col - cx * kBY;
      const int raw = cxy[col * JP + ch];
      if (raw > 0 && gx0 + cx < e0 && gy0 + cy < e1) {
        const int vv = __float_as_int(clampf(__fdiv_rn(__int_as_float(raw), nv), 0.0f, 1.0f));
        if (vv > 0) plane_max(&pxy[size_t(ch) * CC + (gx0 + cx - tl0) * C + (gy0 + cy - tl1)], vv);
      }
    }
    for (int i = t; i < (kBX + kBY) * BZ * J; i += NT) {
      const int ch = i / ((kBX + kBY) * BZ), r = i - ch * ((kBX + kBY) * BZ), a = r / BZ, z = r - a * BZ;
      if (gz0 + z < e2) {
        const int raw = cxz[r * JP + ch];
        const int vv = raw > 0 ? __float_as_int(clampf(__fdiv_rn(__int_as_float(raw), nv), 0.0f, 1.0f)) : 0;
        if (vv > 0) {
          if (a < kBX) {
            if (gx0 + a < e0) plane_max(&pxz[size_t(ch) * CC + (gx0 + a - tl0) * C + lz0 + z], vv);
          } else if (gy0 + a - kBX < e1) {
            plane_max(&pyz[size_t(ch) * CC + (gy0 + a - kBX - tl1) * C + lz0 + z], vv);
          }
        }
      }
    }
    __syncthreads();
  }
}

#endif  // FVP_DIAG

}  // namespace fvp
