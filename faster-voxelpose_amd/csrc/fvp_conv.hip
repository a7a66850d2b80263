// Conv stacks on the fp32 matrix cores (gfx950): CenterNet / C2CNet / P2PNet.
// Reference sites: lib/models/cnns_2d.py:12-178, lib/models/cnns_1d.py:10-132.
//
// Every conv is an implicit GEMM  D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],
// k = (cin, ky, kx), on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain):
//   A operand = weights  (row i = cout,  lane l holds A[l&31][k = l>>5])
//   B operand = pixels   (col j = pixel, lane l holds B[k = l>>5][l&31])
//   D         : lane l holds pixel column l&31 and cout rows (r&3) + 8*(r>>2) + 4*(l>>5)
// so for a fixed accumulator register 32 lanes store 32 consecutive pixels of one output
// channel: activations stay NCHW ([planes][C][H][W], the reference layout) end to end and
// both global loads and stores are contiguous rows.
//
// Workgroup = 4 waves.  Tile = TN planes x TH rows x TW cols = 128*PB pixels x all couts
// (coutp = 32*CB); wave w owns pixel blocks [w*PB, (w+1)*PB).  Per chunk of CC input
// channels the workgroup stages
//   Xs[CC][TN][TH+KH-1][TW+KW-1]   zero-padded halo tile (planar per channel => lanes of a
//                                  pixel block read consecutive LDS words, conflict-free)
//   Ws[CC][KH*KW][coutp]           a contiguous slice of the packed weights
// and runs (CC/2)*KH*KW*CB*PB MFMAs per wave with one LDS word per operand.
// BatchNorm (eval), bias, residual add and ReLU are fused into the store epilogue.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "fvp_conv_args.h"

// waves per SIMD the 1x1 / transposed-conv kernels are compiled for: they are HBM-bound, three waves (<= 168 VGPRs)
// measured 11 us faster than two on the 64 -> 32 transposed conv, four spill
#ifndef FVP_CONV_1X1_OCC
#define FVP_CONV_1X1_OCC 3
#endif
namespace fvp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kStageU = 4;   // independent 16-byte loads in flight per thread while staging
constexpr int kStageS = 8;   // same for the 4-byte generic path

// ---------------------------------------------------------------------------------------------
// Shared epilogue: bias, BN scale/shift, residual, ReLU; coalesced NCHW stores.  Residual values
// are fetched with unconditional (address-clamped) loads, 16 per accumulator tile, before any
// arithmetic, so they cost one memory latency per tile instead of one per element.
template <int CB, int PB, bool HAS_RES>
__device__ __forceinline__ void conv_epilogue(KArgsPtr a, const float* epi, f32x16 (&acc)[CB][PB], int wave, int l31,
                                              int half, int plane0, int y0, int x0, int co0, int tapT) {
  const float* bias = epi;
  const float* scale = epi + a->coutp;
  const float* shift = epi + 2 * a->coutp;
  const int OHW = a->OH * a->OW;
  const int dy = (a->ntapT > 1) ? tapT / a->tapT_w : 0, dx = (a->ntapT > 1) ? tapT % a->tapT_w : 0;
  const bool relu = a->flags & FVP_EPI_RELU;
  const bool res_after = a->flags & FVP_EPI_RES_AFTER_RELU;
  const int tile_px = a->TN * a->TH * a->TW;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int q = (wave * PB + pb) * 32 + l31;
    const int qc = q < tile_px ? q : 0;
    const int n = qc / (a->TH * a->TW), r2 = qc - n * (a->TH * a->TW);
    const int ty = r2 / a->TW, tx = r2 - ty * a->TW;
    const int plane = plane0 + n, y = y0 + ty, x = x0 + tx;
    const bool pix_ok = q < tile_px && plane < a->planes && y < a->H && x < a->W;
    const size_t opix = pix_ok ? size_t(y * a->osy + dy) * a->OW + (x * a->osx + dx) : 0;
    const size_t pbase = pix_ok ? size_t(plane) * a->cout : 0;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      float rv[16];
      unsigned o[16];
      bool ok[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        ok[r] = pix_ok && co < a->cout;
        o[r] = unsigned((pbase + (ok[r] ? co : 0)) * OHW + opix);
        if (HAS_RES) rv[r] = a->res[o[r]];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;     // < coutp: epi vectors are padded
        float v = bn_affine(acc[cb][pb][r], bias[co], scale[co], shift[co]);
        if (HAS_RES && !res_after) v += rv[r];
        if (relu) v = fmaxf(v, 0.0f);
        if (HAS_RES && res_after) v += rv[r];
        if (ok[r]) a->dst[o[r]] = v;
      }
    }
  }
}

// Branch-free staging helpers: every thread computes kU addresses, issues kU independent
// loads (out-of-range items read a valid dummy address and are masked afterwards -- a branch
// around a load makes hipcc wait for each one separately), then writes LDS.
template <int KH, int KW, int CB, int PB>
__global__ void __launch_bounds__(256) k_conv(ConvArgs a) {
  HIP_DYNAMIC_SHARED(float, smem)
  constexpr int KK = KH * KW;
  constexpr int CBW = 32 * CB;               // couts of this workgroup
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  // LDS row stride: vector layout = 4 floats margin | TW | 4 floats margin (16-byte aligned rows)
  const int THp = a.TH + KH - 1, TWp = a.vec ? a.TW + 8 : a.TW + KW - 1;
  const int xbase = a.vec ? 4 - (KW - 1) / 2 : 0;   // LDS column of image x0 - padW
  const int plane_sz = THp * TWp;            // one channel of one plane in LDS
  const int CS = a.TN * plane_sz;            // channel stride in Xs
  float* Xs = smem;
  float* Ws = smem + ((a.CC * CS + 3) & ~3);   // keep the weight slice 16-byte aligned

  int tile = blockIdx.x;
  const int tx_i = tile % a.tiles_x;
  tile /= a.tiles_x;
  const int ty_i = tile % a.tiles_y;
  const int pg = tile / a.tiles_y;
  const int plane0 = pg * a.TN;
  const int y0 = ty_i * a.TH, x0 = tx_i * a.TW;
  const int co0 = blockIdx.y * CBW;
  if (a.plane_valid && a.TN == 1 && !a.plane_valid[plane0 / a.valid_div]) return;

  const int tapT = blockIdx.z;               // transposed-conv tap (0 otherwise)
  const float* wts = a.wts + size_t(tapT) * a.cinp * KK * a.coutp + co0;

  // per-lane LDS offset of each of this wave's pixels (B operand)
  const int tile_px = a.TN * a.TH * a.TW;
  int poff[PB];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int q = (wave * PB + pb) * 32 + l31;
    if (q < tile_px) {
      const int n = q / (a.TH * a.TW), r = q - n * (a.TH * a.TW);
      const int ty = r / a.TW, tx = r - ty * a.TW;
      poff[pb] = n * plane_sz + ty * TWp + tx + xbase;
    } else {
      poff[pb] = 0;                          // reads valid LDS, result never stored
    }
  }

  f32x16 acc[CB][PB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][pb][r] = 0.0f;

  const int padH = (KH - 1) / 2, padW = (KW - 1) / 2;
  const int HW = a.H * a.W;
  const int nrows = a.CC * a.TN * THp;
  const int rows_per_ch = a.TN * THp;

  // margins of the vector layout are never written by the per-chunk staging: clear once
  if (a.vec) {
    const int total = a.CC * CS;
    for (int e = t; e < total; e += 256) Xs[e] = 0.0f;
  }

  for (int c0 = 0; c0 < a.cinp; c0 += a.CC) {
    __syncthreads();
    // (staging reads its launch arguments afresh: kept in SGPRs across the MFMA loop they spilled, round 5)
    const KArgsPtr sa = FVP_FRESH_ARGS(a);
    if (sa->ablate & 1) {
    } else if (sa->vec) {
      // full-width tile, W % 4 == 0: image rows are contiguous 16-byte-aligned runs
      const int qpr = sa->W >> 2;                       // quads per row
      const int nitems = nrows * qpr;
      for (int it0 = t; it0 < nitems; it0 += 256 * kStageU) {
        float4 v[kStageU];
        int dst[kStageU];
        bool ok[kStageU];
#pragma unroll
        for (int u = 0; u < kStageU; ++u) {
          const int it = it0 + 256 * u;
          const bool live = it < nitems;
          const int itc = live ? it : 0;
          const int row = itc / qpr, q = itc - row * qpr;
          const int ci = row / rows_per_ch;
          const int rem = row - ci * rows_per_ch;
          const int n = rem / THp, ry = rem - n * THp;
          const int c = c0 + ci, plane = plane0 + n, y = y0 + ry - padH;
          ok[u] = live && c < sa->cin && plane < sa->planes && y >= 0 && y < sa->H;
          dst[u] = live ? ci * CS + n * plane_sz + ry * TWp + 4 + 4 * q : -1;
          const size_t off = ok[u] ? (size_t(plane) * sa->cin + c) * HW + size_t(y) * sa->W + 4 * q : 0;
          v[u] = *reinterpret_cast<const float4*>(sa->src + off);
        }
#pragma unroll
        for (int u = 0; u < kStageU; ++u)
          if (dst[u] >= 0)
            *reinterpret_cast<float4*>(Xs + dst[u]) = ok[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      // generic path (narrow / odd-width maps): element-wise, halo written explicitly
      const int nitems = nrows * TWp;
      for (int it0 = t; it0 < nitems; it0 += 256 * kStageS) {
        float v[kStageS];
        int dst[kStageS];
        bool ok[kStageS];
#pragma unroll
        for (int u = 0; u < kStageS; ++u) {
          const int it = it0 + 256 * u;
          const bool live = it < nitems;
          const int itc = live ? it : 0;
          const int row = itc / TWp, col = itc - row * TWp;
          const int ci = row / rows_per_ch;
          const int rem = row - ci * rows_per_ch;
          const int n = rem / THp, ry = rem - n * THp;
          const int c = c0 + ci, plane = plane0 + n, y = y0 + ry - padH, x = x0 + col - padW;
          ok[u] = live && c < sa->cin && plane < sa->planes && y >= 0 && y < sa->H && x >= 0 && x < sa->W;
          dst[u] = live ? itc + ci * (CS - rows_per_ch * TWp) : -1;   // = ci*CS + n*plane_sz + ry*TWp + col
          const size_t off = ok[u] ? (size_t(plane) * sa->cin + c) * HW + size_t(y) * sa->W + x : 0;
          v[u] = sa->src[off];
        }
#pragma unroll
        for (int u = 0; u < kStageS; ++u)
          if (dst[u] >= 0) Xs[dst[u]] = ok[u] ? v[u] : 0.0f;
      }
    }
    // ---- weight slice: rows (ci, tap) of CBW floats at stride coutp; zero beyond cinp
    if (!(sa->ablate & 2)) {
      constexpr int QPR = CBW / 4;
      const int nitems = sa->CC * KK * QPR;
      const int avail_rows = (sa->cinp - c0) * KK;
      const float* gw = wts + size_t(c0) * KK * sa->coutp;
      for (int it0 = t; it0 < nitems; it0 += 256 * kStageU) {
        float4 v[kStageU];
        int dst[kStageU];
        bool ok[kStageU];
#pragma unroll
        for (int u = 0; u < kStageU; ++u) {
          const int it = it0 + 256 * u;
          const bool live = it < nitems;
          const int itc = live ? it : 0;
          const int row = itc / QPR, q = itc - row * QPR;
          ok[u] = live && row < avail_rows;
          dst[u] = live ? itc * 4 : -1;
          v[u] = *reinterpret_cast<const float4*>(gw + (ok[u] ? size_t(row) * sa->coutp + 4 * q : 0));
        }
#pragma unroll
        for (int u = 0; u < kStageU; ++u)
          if (dst[u] >= 0)
            *reinterpret_cast<float4*>(Ws + dst[u]) = ok[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __syncthreads();
    // ---- MFMA over the chunk
    for (int ci = (a.ablate & 4) ? a.CC : 0; ci < a.CC; ci += 2) {
      const float* xs = Xs + (ci + half) * CS;
      const float* ws = Ws + (ci + half) * KK * CBW + l31;
#pragma unroll
      for (int ky = 0; ky < KH; ++ky) {
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
          float av[CB], bv[PB];
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) av[cb] = ws[(ky * KW + kx) * CBW + cb * 32];
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) bv[pb] = xs[poff[pb] + ky * TWp + kx];
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
              acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb], bv[pb], acc[cb][pb], 0, 0, 0);
        }
      }
    }
  }

  if (a.ablate & 8) return;
  if (a.flags & FVP_EPI_RES)
    conv_epilogue<CB, PB, true>(FVP_FRESH_ARGS(a), a.epi, acc, wave, l31, half, plane0, y0, x0, co0, tapT);
  else
    conv_epilogue<CB, PB, false>(FVP_FRESH_ARGS(a), a.epi, acc, wave, l31, half, plane0, y0, x0, co0, tapT);
}

// Epilogue of the PAIR layout: accumulator rows co / 16+co of a lane are pixels 2j / 2j+1 of the
// same cout -> one float2 store per cout, 256 contiguous bytes per 32 lanes.
template <int PB, bool HAS_RES>
__device__ __forceinline__ void conv_epilogue_pair(KArgsPtr a, const float* epi, f32x16 (&acc)[PB], int wave, int l31,
                                                   int half, int plane0, int y0) {
  const float* bias = epi;
  const float* scale = epi + a->coutp;
  const float* shift = epi + 2 * a->coutp;
  const bool relu = a->flags & FVP_EPI_RELU;
  const bool res_after = a->flags & FVP_EPI_RES_AFTER_RELU;
  const int Wq = a->W >> 1;
  const int tile_px = a->TN * a->TH * Wq;
  const int HW = a->H * a->W;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int q = (wave * PB + pb) * 32 + l31;
    const int qc = q < tile_px ? q : 0;
    const int n = fdiv(qc, a->m_thw), r2 = qc - n * (a->TH * Wq);
    const int ty = fdiv(r2, a->m_w), tx = r2 - ty * Wq;
    const int plane = plane0 + n, y = y0 + ty;
    const bool pix_ok = q < tile_px && plane < a->planes && y < a->H;
    const unsigned pix = pix_ok ? unsigned(y * a->W + 2 * tx) : 0u;
    const unsigned pbase = pix_ok ? unsigned(plane) * a->cout : 0u;
    float2 rv[8];
    unsigned o[8];
    bool ok[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * half;           // 0..15
      ok[r] = pix_ok && co < a->cout;
      o[r] = (pbase + (ok[r] ? co : 0)) * unsigned(HW) + pix;
      if (HAS_RES) rv[r] = *reinterpret_cast<const float2*>(a->res + o[r]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v[2] = {acc[pb][r], acc[pb][r + 8]};
      const float rr[2] = {HAS_RES ? rv[r].x : 0.f, HAS_RES ? rv[r].y : 0.f};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float x = bn_affine(v[e], bias[co], scale[co], shift[co]);
        if (HAS_RES && !res_after) x += rr[e];
        if (relu) x = fmaxf(x, 0.0f);
        if (HAS_RES && res_after) x += rr[e];
        v[e] = x;
      }
      if (ok[r]) *reinterpret_cast<float2*>(a->dst + o[r]) = make_float2(v[0], v[1]);
    }
  }
}

// v_permlane32_swap: the value of lane l lands in lane l ^ 32 for the half that needs it - from_upper = 0: lanes
// 32-63 receive lanes 0-31 (the other half keeps its own value), from_upper = 1: lanes 0-31 receive lanes 32-63.
__device__ __forceinline__ float hop32(float v, int from_upper) {
  const unsigned u = unsigned(__float_as_int(v));
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __int_as_float(int(from_upper ? r[1] : r[0]));
}

// Epilogue of the paired transposed conv: blocks cb / cb + CB/2 are output columns 2x / 2x+1.
template <int CB, int PB, bool HAS_RES, bool FUSE2 = false, class AP = KArgsPtr>
__device__ __forceinline__ void conv_epilogue_tpair(AP a, const float* epi, f32x16 (&acc)[CB][PB], int wave, int l31,
                                                    int half, int plane0, int y0, int dy, const float* w2s = nullptr,
                                                    const float* epi2 = nullptr) {
  constexpr int CH = CB / 2;
  static_assert(!FUSE2 || CH == 1, "the fused 1x1 conv needs all 32 couts of a pixel in one lane pair");
  const float* bias = epi;
  const float* scale = epi + a->coutp;
  const float* shift = epi + 2 * a->coutp;
  const bool relu = a->flags & FVP_EPI_RELU;
  const bool res_after = a->flags & FVP_EPI_RES_AFTER_RELU;
  const int tile_px = a->TN * a->TH * a->W;
  const int OHW = a->OH * a->OW;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int q = (wave * PB + pb) * 32 + l31;
    const int qc = q < tile_px ? q : 0;
    const int n = fdiv(qc, a->m_thw), r2 = qc - n * (a->TH * a->W);
    const int ty = fdiv(r2, a->m_w), tx = r2 - ty * a->W;
    const int plane = plane0 + n, y = y0 + ty;
    const bool pix_ok = q < tile_px && plane < a->planes && y < a->H;
    const unsigned pix = pix_ok ? unsigned((2 * y + dy) * a->OW + 2 * tx) : 0u;
    const unsigned pbase = pix_ok ? unsigned(plane) * a->cout : 0u;
#pragma unroll
    for (int cb = 0; cb < CH; ++cb) {
      float2 rv[16];
      unsigned o[16];
      bool ok[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        ok[r] = pix_ok && co < a->cout;
        o[r] = (pbase + (ok[r] ? co : 0)) * unsigned(OHW) + pix;
        if (HAS_RES) rv[r] = *reinterpret_cast<const float2*>(a->res + o[r]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;     // < coutp: epi vectors are padded
        float v[2] = {acc[cb][pb][r], acc[cb + CH][pb][r]};
        const float rr[2] = {HAS_RES ? rv[r].x : 0.f, HAS_RES ? rv[r].y : 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float x = bn_affine(v[e], bias[co], scale[co], shift[co]);
          if (HAS_RES && !res_after) x += rr[e];
          if (relu) x = fmaxf(x, 0.0f);
          if (HAS_RES && res_after) x += rr[e];
          v[e] = x;
        }
        if (FUSE2) {
          acc[cb][pb][r] = v[0];                     // keep the finished values in the accumulator registers
          acc[cb + CH][pb][r] = v[1];
        } else if (ok[r]) {
          *reinterpret_cast<float2*>(a->dst + o[r]) = make_float2(v[0], v[1]);
        }
      }
    }
    if (FUSE2) {
      // out[j] = sum_c W2[c][j] * y[c] for this lane's two pixels: a lane holds 16 of the 32 channels (channel
      // (r&3) + 8(r>>2) + 4 half in register r), its partner lane ^ 32 the other 16; W2 sits in LDS as [c][32]
      const float* bias2 = epi2;
      const float* scale2 = epi2 + 32;
      const float* shift2 = epi2 + 64;
      const bool relu2 = a->flags2 & FVP_EPI_RELU;
      const unsigned pb2 = pix_ok ? unsigned(plane) * a->cout2 : 0u;
#pragma unroll
      for (int j0 = 0; j0 < 16; j0 += 4) {
        // ONE fma chain over the 32 channels in channel order - what the standalone 1x1 kernel's MFMA sequence (and a
        // plain fp32 dot product) computes, so the fused form is bit-identical to the two-kernel form.  Channels
        // 4g..4g+3 live in the lane with half == (g & 1) at registers 4 (g >> 1) + q: the running sums hop between
        // the two lanes of a pair after every group (v_permlane32_swap); the lane that does not own a group
        // computes a value nobody reads.  (Summing each lane's 16 channels separately and adding the halves is
        // 30 us per step cheaper but moved the conditioned fixture from 3.9e-4 to 5.0e-4 mm.)
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = 4 * (g >> 1) + q;
            const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float4 w = *reinterpret_cast<const float4*>(w2s + co * 32 + j0);
            const float y0v = acc[0][pb][r], y1v = acc[CH][pb][r];
            s0[0] = fmaf(w.x, y0v, s0[0]);  s1[0] = fmaf(w.x, y1v, s1[0]);
            s0[1] = fmaf(w.y, y0v, s0[1]);  s1[1] = fmaf(w.y, y1v, s1[1]);
            s0[2] = fmaf(w.z, y0v, s0[2]);  s1[2] = fmaf(w.z, y1v, s1[2]);
            s0[3] = fmaf(w.w, y0v, s0[3]);  s1[3] = fmaf(w.w, y1v, s1[3]);
          }
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {             // hand the chains to the partner lane (owner of group g + 1)
            s0[jj] = hop32(s0[jj], g & 1);
            s1[jj] = hop32(s1[jj], g & 1);
          }
        }
        // after the last hop the lanes with half == 0 hold the finished sums (group 7 belongs to half == 1)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = j0 + jj;
          const float t0 = s0[jj], t1 = s1[jj];
          if (half == 0 && pix_ok && j < a->cout2) {
            float x0 = bn_affine(t0, bias2[j], scale2[j], shift2[j]), x1 = bn_affine(t1, bias2[j], scale2[j], shift2[j]);
            if (relu2) {
              x0 = fmaxf(x0, 0.0f);
              x1 = fmaxf(x1, 0.0f);
            }
            *reinterpret_cast<float2*>(a->dst2 + (pb2 + j) * unsigned(OHW) + pix) = make_float2(x0, x1);
          }
        }
      }
    }
  }
}

// Wide epilogue for stride-1 outputs of the pipelined kernel: each 32x32 accumulator tile goes
// through a 4 KB per-wave LDS scratch so that a lane ends up with 4 consecutive pixels of one
// channel -> dwordx4 residual loads and dwordx4 stores in 128-byte runs (the MFMA layout gives a
// lane 16 different channels of ONE pixel, i.e. 4-byte stores).
template <int CB, int PB, bool HAS_RES>
__device__ __forceinline__ void conv_epilogue_wide(KArgsPtr a, const float* epi, f32x16 (&acc)[CB][PB], float* scratch,
                                                   int wave, int lane, int plane0, int y0, int co0) {
  const float* bias = epi;
  const float* scale = epi + a->coutp;
  const float* shift = epi + 2 * a->coutp;
  const bool relu = a->flags & FVP_EPI_RELU;
  const bool res_after = a->flags & FVP_EPI_RES_AFTER_RELU;
  const int tile_px = a->TN * a->TH * a->TW;
  const int l31 = lane & 31, half = lane >> 5;
  const int HW = a->H * a->W;
  float* sc = scratch + wave * 1024;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[cb][pb][r];
      __syncthreads();
      float4 v[4], rv[4];
      unsigned off[4];
      bool ok[4];
      int co[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        const int col = i >> 3, qd = i & 7;
        v[j] = *reinterpret_cast<const float4*>(sc + col * 32 + qd * 4);
        co[j] = co0 + cb * 32 + col;
        const int q = (wave * PB + pb) * 32 + qd * 4;
        const int qc = q < tile_px ? q : 0;
        const int n = fdiv(qc, a->m_thw), r2 = qc - n * (a->TH * a->TW);
        const int ty = fdiv(r2, a->m_w), tx = r2 - ty * a->TW;
        const int plane = plane0 + n, y = y0 + ty;
        ok[j] = q < tile_px && plane < a->planes && y < a->H && co[j] < a->cout;
        off[j] = ok[j] ? unsigned((plane * a->cout + co[j]) * HW + y * a->W + tx) : 0u;
        if (HAS_RES) rv[j] = *reinterpret_cast<const float4*>(a->res + off[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float b = bias[co[j]], sc_ = scale[co[j]], sh = shift[co[j]];
        float o[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        const float rr[4] = {HAS_RES ? rv[j].x : 0.f, HAS_RES ? rv[j].y : 0.f, HAS_RES ? rv[j].z : 0.f,
                             HAS_RES ? rv[j].w : 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = bn_affine(o[e], b, sc_, sh);
          if (HAS_RES && !res_after) x += rr[e];
          if (relu) x = fmaxf(x, 0.0f);
          if (HAS_RES && res_after) x += rr[e];
          o[e] = x;
        }
        if (ok[j]) *reinterpret_cast<float4*>(a->dst + off[j]) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Pipelined conv for full-width tiles with W % 4 == 0 (every 2-D layer of the three nets):
// chunk k+1 is copied HBM/L2 -> LDS by the LDS-DMA (global_load_lds, 16 B per lane, no VGPRs)
// while the matrix cores work on chunk k; two LDS buffers, one barrier per chunk.
//   Xs[buf][CC][TN][TH+KH-1][4 + W]  every row slot starts with a 16-byte zero margin (its DMA
//                                  lanes read the zero page), so the slots stay dense for the DMA
//                                  (base + lane*16) and the left / right halo taps read zeros
//                                  without any per-operand masking; rows above / below the image,
//                                  channel padding and missing planes also read the zero page
//   Ws[buf][CC][KH*KW][32*CB]
//
// PAIR (layers with cout <= 16, i.e. the 7x7 front conv 15 -> 16): instead of padding the couts
// to the 32 rows of the MFMA tile, rows 16..31 hold the SAME couts with the kernel shifted one
// tap to the right, and a tile column is a pixel PAIR: row co computes pixel 2j, row 16+co pixel
// 2j+1, over KW+1 taps (the extra tap of each row set has a zero weight, which leaves the fma
// chain bit-identical).  7/8 of the MFMA work is useful instead of 1/2.
//
// TPAIR (ConvTranspose k2 s2, 2-D): blockIdx.z = output row parity dy; the weight rows of the two
// column taps are concatenated ([dy][cinp][dx*coutp + co]), so accumulator block cb and
// cb + CB/2 of a lane are output pixels (2x, 2x+1) of the same cout: the input tile is read once
// for both taps and the outputs leave as float2 (the tap-per-launch form reads it four times and
// stores single floats at stride 2).
//
// KS (split K; small maps with long reductions: CenterNet's 3x3 layers on the 20x20 / 40x40 levels, 64-128
// channels): with a handful of planes such a layer has fewer pixel blocks than the chip has SIMDs, and a wave's
// output is ONE dependent chain of cinp*9/2 = 288-576 MFMAs (64 cycles each: 8-15 us before the first store).
// Here the four waves of a workgroup share one pixel block and each takes every fourth channel pair of a chunk;
// the four partial tiles are added in the fixed order ((w0 + w1) + w2) + w3 through LDS and wave 0 runs the
// epilogue.  Chosen from the layer SHAPE only, so a frame's result does not depend on its batch.
template <int KH, int KW, int CB, int PB, bool PAIR = false, bool TPAIR = false, bool KS = false>
__global__ void __launch_bounds__(256, (KH * KW == 1 && !(TPAIR && CB * PB == 4 && CB == 2)) ? FVP_CONV_1X1_OCC : 2) k_conv_dma(ConvArgs a) {
  HIP_DYNAMIC_SHARED(float, smem)
  constexpr int KT = PAIR ? KW + 1 : KW;             // taps per kernel row in the packed layout
  constexpr int KK = KH * KT;
  constexpr int CBW = 32 * CB;
  constexpr int padH = (KH - 1) / 2, padW = (KW - 1) / 2;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int THp = a.TH + KH - 1, W = a.W, WP = W + 4;
  const int plane_sz = THp * WP;
  const int CS = a.TN * plane_sz;
  const int xs_sz = a.CC * CS + 4, ws_sz = a.CC * KK * CBW, buf_sz = xs_sz + ws_sz;   // floats, all % 4 == 0

  int tile = blockIdx.x;
  const int pg = tile / a.tiles_y;                   // scalar (uniform) division
  const int ty_i = tile - pg * a.tiles_y;
  const int plane0 = pg * a.TN;
  const int y0 = ty_i * a.TH;
  const int co0 = blockIdx.y * CBW;
  if (a.plane_valid && a.TN == 1 && !a.plane_valid[plane0 / a.valid_div]) return;
  const int tapT = blockIdx.z;
  const float* wts = a.wts + size_t(tapT) * a.cinp * KK * a.wrow + co0;

  const int Wq = PAIR ? W >> 1 : W;                  // tile columns per image row (pixels or pixel pairs)
  const int tile_px = a.TN * a.TH * Wq;
  int poff[PB];
  const int pwave = KS ? 0 : wave;                   // KS: every wave works on the workgroup's first PB pixel blocks
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int q = (pwave * PB + pb) * 32 + l31;
    const int qc = q < tile_px ? q : 0;
    const int n = fdiv(qc, a.m_thw), r = qc - n * (a.TH * Wq);
    const int ty = fdiv(r, a.m_w), tx = r - ty * Wq;
    poff[pb] = n * plane_sz + ty * WP + 4 + (PAIR ? 2 * tx : tx) - padW;
  }

  f32x16 acc[CB][PB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][pb][r] = 0.0f;

  const int HW = a.H * W;
  const int qpr = (W >> 2) + 1;                      // quads per row slot, quad 0 = zero margin
  const int rows_per_ch = a.TN * THp;
  const int nin = a.CC * rows_per_ch * qpr + 1;      // 16-byte items of the input tile (+ tail zero quad)
  const int nwq = a.CC * KK * (CBW / 4);             // 16-byte items of the weight slice
  const int nchunks = (a.cinp + a.CC - 1) / a.CC;

  // The (row, quad) decomposition of this lane's staging items does not depend on the chunk:
  // do the integer divisions once, keep {source offset within the chunk, channel} per item.
  constexpr int kMaxIn = 8;                          // host guarantees nin <= kMaxIn * 256
#if FVP_CONV_BUF_DMA
  // (the paired transposed conv keeps the global-address form: with the fused 1x1 head it sits at its 168-register cap and
  // the buffer form spilled 179 dwords there)
  constexpr bool kBuf = !TPAIR;
#else
  constexpr bool kBuf = false;
#endif
  constexpr unsigned kOOB = 0x80000000u;             // an offset that fails the buffer range check: the DMA writes zeros
  unsigned vin[kMaxIn];                              // kBuf: byte offset of item j from the plane group's first element, or kOOB
  unsigned ci_pack[kMaxIn / 4] = {};                 // kBuf: channel-in-chunk of item j, 8 bits each (ragged last chunk)
  int in_off[kMaxIn], in_ci[kMaxIn];                 // !kBuf: float offset (-1: zero page) and channel-in-chunk
#pragma unroll
  for (int j = 0; j < kMaxIn; ++j) {
    const int it = (wave + 4 * j) * 64 + lane;
    int off = -1, cij = 0;
    if (it < nin) {
      const int row = fdiv(it, a.m_qpr), q = it - row * qpr;
      const int ci = fdiv(row, a.m_rpc);
      const int rem = row - ci * rows_per_ch;
      const int n = fdiv(rem, a.m_thp), ry = rem - n * THp;
      const int plane = plane0 + n, y = y0 + ry - padH;
      cij = ci;
      if (q > 0 && ci < a.CC && plane < a.planes && y >= 0 && y < a.H)
        off = (n * a.cin + ci) * HW + y * W + 4 * (q - 1);
    }
    if constexpr (kBuf) {
      vin[j] = off >= 0 ? unsigned(off) * 4u : kOOB;
      ci_pack[j >> 2] |= unsigned(cij & 255) << (8 * (j & 3));
    } else {
      in_off[j] = off;
      in_ci[j] = cij;
    }
  }
  const float* src_tile = a.src + size_t(plane0) * a.cin * HW;
  // Chunk DMA through buffer addressing (as in k_conv_wino, round 3): the fp32 MFMA shares the vector ALUs, so the ~8
  // vector instructions per item of the global-address form (64-bit select against the zero page, pointer add) were matrix
  // time of every wave of the SIMD.  Descriptors (SGPRs): the workgroup's plane group / its packed weight block; per-lane
  // 32-bit byte offsets, chunk-invariant: one per input item, ONE for all weight items (item j + 1 lies 256 quads = a whole
  // number of rows further: a uniform offset); the chunk offset is scalar.  Lanes outside the image carry an offset that
  // fails the range check: the hardware writes zeros (tools/micro/buflds.hip).  Only a ragged last chunk (cin or cinp not
  // a multiple of CC) masks per item.
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = FVP_LDS_BYTE_ADDRESS(smem);
  auto make_rsrc = [](const float* p) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane(int(unsigned(b)));
    rs[1] = __builtin_amdgcn_readfirstlane(int(unsigned(b >> 32) & 0xffffu));
    rs[2] = 0x7ffffff0;
    rs[3] = 0x00020000;
    return rs;
  };
  const i32x4 rs_in = make_rsrc(src_tile), rs_w = make_rsrc(wts);
  constexpr int QPRW = CBW / 4;                       // 16-byte quads per packed weight row
  constexpr int RPI = 256 / QPRW;                     // weight rows between item j and item j + 1 of a lane
  static_assert(256 % QPRW == 0, "item j + 1 of a lane must lie a whole number of packed weight rows further (CB in {1,2,4})");
  const int wit0 = wave * 64 + lane;
  const int wrow0 = wit0 / QPRW;
  const unsigned vw0 = unsigned(wrow0 * a.wrow + 4 * (wit0 % QPRW)) * 4u;
  auto buf_dma16 = [&](unsigned vo, const i32x4& rs, unsigned so, unsigned la) { asm_buffer_load_lds16(la, vo, rs, so); };
  auto stage_buf = [&](int k, int buf) {
    const int c0 = k * a.CC;
    const unsigned la_x = lds0 + 4u * unsigned(4 + buf * buf_sz + wave_s * 256);
    const unsigned la_w = la_x + 4u * unsigned(xs_sz);
    const unsigned so_in = unsigned(c0) * unsigned(HW) * 4u;
    const bool ragged = c0 + a.CC > a.cin || c0 + a.CC > a.cinp;      // uniform: the trailing chunk(s) that reach past cin / cinp
#pragma unroll
    for (int j = 0; j < kMaxIn; ++j) {
      const int g = wave + 4 * j;
      if (g * 64 + lane < nin) {
        unsigned vo = vin[j];
        if (ragged && c0 + int((ci_pack[j >> 2] >> (8 * (j & 3))) & 255u) >= a.cin) vo = kOOB;
        buf_dma16(vo, rs_in, so_in, la_x + unsigned(j) * 4096u);
      }
    }
    const int avail_rows = (a.cinp - c0) * KK;
    const unsigned so_w = unsigned(c0) * unsigned(KK) * unsigned(a.wrow) * 4u;
    const unsigned step_w = unsigned(RPI) * unsigned(a.wrow) * 4u;
    int jj = 0;
    for (int g = wave; g * 64 < nwq; g += 4, ++jj) {
      if (g * 64 + lane < nwq) {
        unsigned vo = vw0;
        if (ragged && wrow0 + jj * RPI >= avail_rows) vo = kOOB;
        buf_dma16(vo, rs_w, so_w + unsigned(jj) * step_w, la_w + unsigned(jj) * 4096u);
      }
    }
  };
  auto stage_glb = [&](int k, int buf) {
    float* xs = smem + 4 + buf * buf_sz;
    float* ws = xs + xs_sz;
    const int c0 = k * a.CC;
#pragma unroll
    for (int j = 0; j < kMaxIn; ++j) {
      const int g = wave + 4 * j;
      if (g * 64 + lane < nin) {
        const bool ok = in_off[j] >= 0 && c0 + in_ci[j] < a.cin;
        const float* src = ok ? src_tile + size_t(c0) * HW + in_off[j] : a.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(xs + g * 256), 16, 0, 0);
      }
    }
    const int avail_rows = (a.cinp - c0) * KK;
    const float* gw = wts + size_t(c0) * KK * a.wrow;
    for (int g = wave; g * 64 < nwq; g += 4) {
      const int it = g * 64 + lane;
      if (it < nwq) {
        const int row = it / (CBW / 4), q = it - row * (CBW / 4);
        const float* src = row < avail_rows ? gw + size_t(row) * a.wrow + 4 * q : a.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ws + g * 256), 16, 0, 0);
      }
    }
  };

  auto stage = [&](int k, int buf) {
    if constexpr (kBuf) stage_buf(k, buf);
    else stage_glb(k, buf);
  };

  // bias | scale | shift of every cout (+ the fused 1x1 conv's three 32-vectors) staged behind everything else in LDS: the
  // epilogues read them with ds_read.  As global loads they queued behind the previous element's stores in the in-order
  // vmcnt counter, and behind a conditional store the compiler can only wait with vmcnt(0): every output element of a
  // lane paid a store round trip plus a load round trip (16-32 chained round trips per wave; round 3, found in the ISA)
  const float* const epi_s = smem + a.epi_off;
  {
    float* e = smem + a.epi_off;
    for (int i = t; i < 3 * a.coutp; i += 256) e[i] = a.epi[i];
    if (TPAIR && a.w2 && t < 96) e[3 * a.coutp + t] = a.epi2[t];
  }
  if (!(a.ablate & 3)) stage(0, 0);
  wait_vmcnt(0);                                     // (the DMA is inline asm: the barrier's fence does not know it)
  __syncthreads();
  for (int k = 0; k < nchunks; ++k) {
    const int buf = k & 1;
    if (k + 1 < nchunks && !(a.ablate & 3)) stage(k + 1, buf ^ 1);
    const float* Xs = smem + 4 + buf * buf_sz;
    const float* Ws = Xs + xs_sz;
    // Software-pipelined operand fetch: the LDS reads of step s+1 (one tap of one channel pair:
    // CB A-words + PB B-words) are issued before the CB*PB MFMAs of step s; sched_barriers pin
    // that order (left alone, hipcc sinks every ds_read next to its MFMA and each MFMA eats a
    // full LDS latency).  Two register sets; KH is odd for every kernel shape, so consecutive
    // channel pairs alternate the starting set (template parameter P).
    // Pipeline step = one kernel ROW of one channel pair (KW taps): its KW*(CB+PB) operand words are
    // fetched while the previous row's KW*CB*PB MFMAs run.  Fetching a whole row per step lets the
    // compiler pair neighbouring taps into ds_read2_b32 (B: adjacent floats; A: CBW apart) and needs
    // one address add per (row, pixel block) instead of one per tap.
    float av[2][KT][CB], bv[2][KT][PB];
    constexpr int CSTEP = KS ? 8 : 2;                // channel distance between a wave's consecutive channel pairs
    const int cfirst = KS ? 2 * wave : 0;            // (host: CC % 8 == 0 for KS)
    auto fetch = [&](int set, int ci, int ky, int wp) {
      const int cic = ci < a.CC ? ci : a.CC - 2;              // last prefetch of a chunk: harmless re-read
      const float* xs = Xs + (cic + half) * CS + ky * wp;
      const float* ws = Ws + (cic + half) * KK * CBW + ky * KT * CBW + l31;
#pragma unroll
      for (int kx = 0; kx < KT; ++kx) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) av[set][kx][cb] = ws[kx * CBW + cb * 32];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) bv[set][kx][pb] = xs[poff[pb] + kx];
      }
    };
    auto block = [&](auto parity, int ci) {
      constexpr int P = decltype(parity)::value;
      int wp = WP;
      FVP_OPAQUE(wp);                                  // row addresses are recomputed per channel pair, not kept live
#pragma unroll
      for (int ky = 0; ky < KH; ++ky) {
        const int cur = (P + ky) & 1, nxt = cur ^ 1;
        // wait for this step's operands (issued one step ago) BEFORE issuing the next step's reads:
        // hipcc only emits lgkmcnt(0), which placed after the new reads would expose their latency
        __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0), vmcnt/expcnt untouched
        __builtin_amdgcn_sched_barrier(0);
        if (ky + 1 < KH) fetch(nxt, ci, ky + 1, wp);
        else fetch(nxt, ci + CSTEP, 0, wp);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kx = 0; kx < KT; ++kx)
#pragma unroll
          for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
              acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][kx][cb], bv[cur][kx][pb], acc[cb][pb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (!(a.ablate & 4)) {
      fetch(0, cfirst, 0, WP);
      for (int ci = cfirst; ci < a.CC; ci += 2 * CSTEP) {
        block(std::integral_constant<int, 0>{}, ci);
        if (ci + CSTEP < a.CC) block(std::integral_constant<int, 1>{}, ci + CSTEP);
      }
    }
    wait_vmcnt(0);                                   // chunk k + 1 has landed (this wave's items; the barrier: everybody's)
    __syncthreads();
  }
  if (a.ablate & 8) return;
  if constexpr (KS) {
    // the chunk loop ended with a barrier: the slots are free.  red[w - 1][cb][pb][r][lane]
    float* red = smem + 4;
    if (wave > 0) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((((wave - 1) * CB + cb) * PB + pb) * 16 + r) * 64 + lane] = acc[cb][pb][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 1; w < 4; ++w)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[cb][pb][r] += red[((((w - 1) * CB + cb) * PB + pb) * 16 + r) * 64 + lane];
    if (a.flags & FVP_EPI_RES)
      conv_epilogue<CB, PB, true>(FVP_FRESH_ARGS(a), epi_s, acc, 0, l31, half, plane0, y0, 0, co0, tapT);
    else
      conv_epilogue<CB, PB, false>(FVP_FRESH_ARGS(a), epi_s, acc, 0, l31, half, plane0, y0, 0, co0, tapT);
    return;
  }
  if (PAIR) {
    if (a.flags & FVP_EPI_RES)
      conv_epilogue_pair<PB, true>(FVP_FRESH_ARGS(a), epi_s, acc[0], wave, l31, half, plane0, y0);
    else
      conv_epilogue_pair<PB, false>(FVP_FRESH_ARGS(a), epi_s, acc[0], wave, l31, half, plane0, y0);
  } else if (TPAIR) {
    if constexpr (CB == 2) {
      if (a.w2) {                                      // fused 1x1 output conv: its weights [32][32] through LDS
        float* w2s = smem + 4;                         // (the chunk loop ended with a barrier: the slots are free)
        for (int i = threadIdx.x; i < 32 * 32 / 4; i += 256)
          reinterpret_cast<float4*>(w2s)[i] = reinterpret_cast<const float4*>(a.w2)[i];
        __syncthreads();
        if (a.flags & FVP_EPI_RES)
          conv_epilogue_tpair<CB, PB, true, true>(&a, epi_s, acc, wave, l31, half, plane0, y0, tapT, w2s, epi_s + 3 * a.coutp);
        else
          conv_epilogue_tpair<CB, PB, false, true>(&a, epi_s, acc, wave, l31, half, plane0, y0, tapT, w2s, epi_s + 3 * a.coutp);
        return;
      }
    }
    if (a.flags & FVP_EPI_RES)
      conv_epilogue_tpair<CB, PB, true>(FVP_FRESH_ARGS(a), epi_s, acc, wave, l31, half, plane0, y0, tapT);
    else
      conv_epilogue_tpair<CB, PB, false>(FVP_FRESH_ARGS(a), epi_s, acc, wave, l31, half, plane0, y0, tapT);
  } else if (a.ntapT > 1 || (a.ablate & 16)) {          // transposed conv: strided outputs, scalar stores
    if (a.flags & FVP_EPI_RES)
      conv_epilogue<CB, PB, true>(FVP_FRESH_ARGS(a), epi_s, acc, wave, l31, half, plane0, y0, 0, co0, tapT);
    else
      conv_epilogue<CB, PB, false>(FVP_FRESH_ARGS(a), epi_s, acc, wave, l31, half, plane0, y0, 0, co0, tapT);
  } else if (a.flags & FVP_EPI_RES) {
    conv_epilogue_wide<CB, PB, true>(FVP_FRESH_ARGS(a), epi_s, acc, smem + 4, wave, lane, plane0, y0, co0);
  } else {
    conv_epilogue_wide<CB, PB, false>(FVP_FRESH_ARGS(a), epi_s, acc, smem + 4, wave, lane, plane0, y0, co0);
  }
}

}  // namespace fvp
#include "fvp_conv_reg.h"
namespace fvp {

// max_pool(2,2) / max_pool1d(2): one thread per output element.
__global__ void __launch_bounds__(256) k_pool2(const float* __restrict__ src, float* __restrict__ dst, long total,
                                               int H, int W, const uint8_t* __restrict__ plane_valid, int valid_div,
                                               int C) {
  const long i = long(blockIdx.x) * 256 + threadIdx.x;
  if (i >= total) return;
  const int OW = W / 2, OH = H > 1 ? H / 2 : 1;
  const int ox = int(i % OW);
  const int oy = int((i / OW) % OH);
  const long pc = i / (long(OW) * OH);
  if (plane_valid && !plane_valid[(pc / C) / valid_div]) return;
  const float* s = src + pc * long(H) * W + long(oy) * (H > 1 ? 2 : 1) * W + ox * 2;
  float m = fmaxf(s[0], s[1]);
  if (H > 1) m = fmaxf(m, fmaxf(s[W], s[W + 1]));
  dst[i] = m;
}

// state_dict tensors -> packed [tapT][cinp][KK][coutp] + bias|scale|shift.
__global__ void __launch_bounds__(256)
k_pack_conv(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ gamma,
            const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var, float eps,
            int transposed, int cin, int cout, int cinp, int coutp, int kh, int kw, float* __restrict__ wdst,
            float* __restrict__ edst) {
  const int KK = kh * kw;
  const int ntap = transposed ? KK : 1, kk = transposed ? 1 : KK;
  const int nw = ntap * cinp * kk * coutp;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nw) {
    const int co = i % coutp;
    int r = i / coutp;
    const int tap = r % kk;
    r /= kk;
    const int ci = r % cinp, tt = r / cinp;
    float v = 0.0f;
    if (co < cout && ci < cin) {
      // Conv: weight[co][ci][ky][kx];  ConvTranspose: weight[ci][co][dy][dx]
      v = transposed ? w[(size_t(ci) * cout + co) * KK + tt] : w[(size_t(co) * cin + ci) * KK + tap];
    }
    wdst[i] = v;
  }
  if (i < coutp) {
    float bb = 0.0f, sc = 1.0f, sh = 0.0f;
    if (i < cout) {
      bb = b ? b[i] : 0.0f;
      if (gamma) {
        // eval BatchNorm: y = x * alpha + (beta - mean * alpha), alpha = gamma / sqrt(var + eps)
        sc = __fdiv_rn(gamma[i], sqrtf(__fadd_rn(var[i], eps)));
        sh = __fsub_rn(beta[i], __fmul_rn(mean[i], sc));
      }
    }
    edst[i] = bb;
    edst[coutp + i] = sc;
    edst[2 * coutp + i] = sh;
  }
}

}  // namespace fvp
#include "fvp_conv7.h"
namespace fvp {

// PAIR layout of a KHxKW conv with cout <= 16: [cinp][KH][KW+1][32], row co = w[co][ci][ky][kx]
// (kx < KW), row 16+co = w[co][ci][ky][kx-1] (kx >= 1), zero elsewhere.
__global__ void __launch_bounds__(256)
k_pack_pair(const float* __restrict__ w, int cin, int cout, int cinp, int kh, int kw, float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int kt = kw + 1;
  if (i >= cinp * kh * kt * 32) return;
  const int row = i & 31;
  int r = i >> 5;
  const int kx = r % kt;
  r /= kt;
  const int ky = r % kh, ci = r / kh;
  const int co = row & 15, sx = row < 16 ? kx : kx - 1;
  float v = 0.0f;
  if (co < cout && ci < cin && sx >= 0 && sx < kw) v = w[((size_t(co) * cin + ci) * kh + ky) * kw + sx];
  dst[i] = v;
}

// Paired layout of ConvTranspose(k2,s2) weights [cin][cout][2][2]: [dy][cinp][dx*coutp + co].
__global__ void __launch_bounds__(256)
k_pack_tpair(const float* __restrict__ w, int cin, int cout, int cinp, int coutp, float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * cinp * 2 * coutp) return;
  const int co = i % coutp;
  int r = i / coutp;
  const int dx = r & 1;
  r >>= 1;
  const int ci = r % cinp, dy = r / cinp;
  dst[i] = (co < cout && ci < cin) ? w[((size_t(ci) * cout + co) * 2 + dy) * 2 + dx] : 0.0f;
}

template <int KH, int KW, int CB, int PB>
static int launch_conv(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  if (a.dma) {
    auto k = &k_conv_dma<KH, KW, CB, PB>;
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  } else {
    auto k = &k_conv<KH, KW, CB, PB>;
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  }
  return launch_status();
}

template <int KH, int KW>
static int dispatch_tile(int CB, int PB, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  const int key = CB * 10 + PB;
  switch (key) {
    case 11: return launch_conv<KH, KW, 1, 1>(a, grid, lds, s);
    case 12: return launch_conv<KH, KW, 1, 2>(a, grid, lds, s);
    case 14: return launch_conv<KH, KW, 1, 4>(a, grid, lds, s);
    case 22: return launch_conv<KH, KW, 2, 2>(a, grid, lds, s);
    case 41: return launch_conv<KH, KW, 4, 1>(a, grid, lds, s);
    default: return FVP_ELIMIT;
  }
}

static int dispatch_conv(int kh, int kw, int CB, int PB, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  if (kh == 7 && kw == 7) return dispatch_tile<7, 7>(CB, PB, a, grid, lds, s);
  if (kh == 3 && kw == 3) return dispatch_tile<3, 3>(CB, PB, a, grid, lds, s);
  if (kh == 1 && kw == 1) return dispatch_tile<1, 1>(CB, PB, a, grid, lds, s);
  if (kh == 1 && kw == 7) return dispatch_tile<1, 7>(CB, PB, a, grid, lds, s);
  if (kh == 1 && kw == 3) return dispatch_tile<1, 3>(CB, PB, a, grid, lds, s);
  return FVP_ELIMIT;
}

// tuning knobs (diagnostics): FVP_CONV_LDS_KB, FVP_CONV_ABLATE, FVP_CONV_PB
static const size_t kLdsBudget = env_size("FVP_CONV_LDS_KB", 64) * 1024;
static const int kAblate = int(env_size("FVP_CONV_ABLATE", 0));
static const int kForcePB = int(env_size("FVP_CONV_PB", 0));
static const int kNoDma = int(env_size("FVP_CONV_NO_DMA", 0));

static const int kNoWino = int(env_size("FVP_CONV_NO_WINO", 0));
static const int kNoPair = int(env_size("FVP_CONV_NO_PAIR", 0));
static const int kNoK7 = int(env_size("FVP_CONV_NO_K7", 0));        // diagnostics: 7x7 convs on the pixel-pair form of k_conv_dma
static const size_t kK7LdsPad = env_size("FVP_K7_LDS_KB", 0) * 1024;  // diagnostics: pad k_conv7's LDS request (fewer workgroups per CU)
static const int kNoPoolFuse = int(env_size("FVP_CONV_NO_POOL_FUSE", 0));
static const int kNoHeadFuse = int(env_size("FVP_CONV_NO_HEAD_FUSE", 0));
static const int kNoReg = int(env_size("FVP_CONV_NO_REG", 0));     // diagnostics: 1x1 / transposed convs on k_conv_dma
// (read per call in the diagnostics build, so that a test can run the same stack through both kernels; a constant in the product)
// Transposed convs gain from 200 tiles on (B = 1: 128 -> 64 22.6 -> 18.6 us at 240 tiles, 64 -> 32 17.2 -> 12.5 us at 960; CenterNet at
// B = 8: 16.3 -> 9.0 us at 400), 1x1 convs only from ~1 000 (64 -> 128 at 240 tiles: 7.4 -> 13.1 us).
static long reg_min_tiles(bool transposed) { return long(env_size("FVP_CONV_REG_MIN_TILES", transposed ? 200 : 1024)); }
static const int kNoKSplit = int(env_size("FVP_CONV_NO_KSPLIT", 0));   // diagnostics: no split-K form for the small-map 3x3 layers
template <int K, int NB, int MODE>
static int launch_reg(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static LdsOptIn optin[2];
  if (a.flags & FVP_EPI_RES) {
    auto k = &k_conv_reg<K, NB, MODE, true>;
    if (int e = lds_opt_in(optin[1], reinterpret_cast<const void*>(k), lds)) return e;
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  } else {
    auto k = &k_conv_reg<K, NB, MODE, false>;
    if (int e = lds_opt_in(optin[0], reinterpret_cast<const void*>(k), lds)) return e;
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  }
  return launch_status();
}

// k_conv_reg (fvp_conv_reg.h) for 1x1 convs and paired transposed convs whose planes are whole 32-pixel tiles and whose
// weights fit LDS.  Returns -1 when the layer is not of that kind (the caller goes on to k_conv_dma).  The two kernels
// produce the same bits, so the choice may depend on the amount of work: below reg_min_tiles() tiles (a few planes of
// CenterNet's maps) the 256-pixel tiles of k_conv_dma spread over more CUs.
static bool reg_shape_ok(const FvpConvOp& op, int planes, bool tr) {
  const int hw = op.h * op.w;
  if (kNoReg || op.cin != op.cinp || op.h <= 1 || hw % 32) return false;
  if ((op.flags & FVP_EPI_RES) && op.cout % 8) return false;      // (residual rows are read through a uniform row pointer)
  if (tr && (op.pair_off <= 0 || kNoPair)) return false;
  return long(planes) * (hw / 32) >= reg_min_tiles(tr);
}
// the 64 -> 32 transposed conv with its fused 1x1 head runs on k_conv_reg (whose MFMA head takes up to 32 output channels;
// k_conv_dma's fma-chain head stops at 16)
static bool reg_takes_fused_head(const FvpConvOp& op, int planes) {
  return op.kind == FVP_OP_CONVT2 && op.cinp == 64 && op.coutp == 32 && reg_shape_ok(op, planes, true);
}

static int plan_and_launch_reg(const FvpConvOp& op, ConvArgs a, const float* params, int planes, hipStream_t s, bool tr) {
  const int hw = op.h * op.w;
  if (!reg_shape_ok(op, planes, tr)) return -1;
  const long tiles = long(planes) * (hw / 32);
  int mode = 0, NB = op.coutp / 32;
  a.wrow = op.coutp;
  if (tr) {
    mode = a.w2 ? 2 : 1;
    NB = 2 * op.coutp / 32;
    a.wts = params + op.pair_off;
    a.wrow = 2 * op.coutp;
    a.ntapT = 2;
    a.tapT_w = 1;
  } else if (op.kh != 1 || op.kw != 1 || a.w2) {
    return -1;
  }
  const int ny = 1;
  // instances exist for these (cinp, NB, mode) only; the key is injective on that domain (cinp = 56,
  // coutp = 2688 used to alias 6440)
  if (NB < 1 || NB > 4 || (op.cinp != 16 && op.cinp != 32 && op.cinp != 64 && op.cinp != 128)) return -1;
  // raw-buffer addressing: a plane's input (K rows) and output (cout rows, + 4 for the upper half wave) are addressed with
  // 32-bit byte offsets and an `int` num_records
  if (double(op.cinp) * hw * 4.0 >= 2147483648.0 || (double(op.coutp) + 4.0) * hw * (tr ? 4.0 : 1.0) * 4.0 >= 2147483648.0) return -1;
  const int key = op.cinp * 100 + NB * 10 + mode;
  if (key != 1610 && key != 3210 && key != 3220 && key != 6440 && key != 12841 && key != 6421 && key != 6422) return -1;
  a.m_tpp = make_magic(hw / 32);
  a.m_w = make_magic(op.w);
  a.zeros = params;
  a.ablate = kAblate;
  const size_t lds = (size_t(op.cinp) * 32 * NB + 3 * size_t(op.coutp) + (mode == 2 ? 32 * 32 + 96 : 0)) * sizeof(float);
  static const int kRegWgs = int(env_size("FVP_CONV_REG_WGS", 0));   // diagnostics: workgroups per CU
  const int occ = kRegWgs ? kRegWgs : (op.cinp * NB <= 128 ? 3 : 2);
  const int per_cu = std::max(1, std::min(occ, int((160 * 1024) / (lds + 256))));
  const int nz = tr ? 2 : 1;
  const long want = (tiles + 3) / 4;
  const int gx = int(std::min<long>(want, std::max(1, persistent_workgroups() * per_cu / (nz * ny))));
  dim3 grid(gx, ny, nz);
  ProfScope ps(FVP_K_CONV, s, 2.0 * op.cin * op.cout * (tr ? 4.0 : 1.0) * hw * planes, 1, prof_level() >= 2);
  switch (key) {
    case 1610: return launch_reg<16, 1, 0>(a, grid, lds, s);
    case 3210: return launch_reg<32, 1, 0>(a, grid, lds, s);
    case 3220: return launch_reg<32, 2, 0>(a, grid, lds, s);
    case 6440: return launch_reg<64, 4, 0>(a, grid, lds, s);
    case 12841: return launch_reg<128, 4, 1>(a, grid, lds, s);
    case 6421: return launch_reg<64, 2, 1>(a, grid, lds, s);
    default: return launch_reg<64, 2, 2>(a, grid, lds, s);
  }
}

// ConvTranspose(k2,s2) in the paired form: CB = 2*coutp/32 accumulator blocks (both column taps),
// grid.z = output row parity.
static int plan_and_launch_tpair(const FvpConvOp& op, ConvArgs a, const float* params, int planes, hipStream_t s) {
  const int CB = 2 * op.coutp / 32, PB = 4 / CB;         // (2,2) or (4,1)
  a.wts = params + op.pair_off;
  a.wrow = 2 * op.coutp;
  a.ntapT = 2;
  a.tapT_w = 1;
  a.ablate = kAblate;
  const int hw = op.h * op.w, TP = PB * 128;
  a.TW = op.w;
  if (hw <= TP) {
    a.TH = op.h;
    a.TN = TP / hw;
  } else if (op.w <= TP) {
    a.TH = TP / op.w;
    a.TN = 1;
  } else {
    return FVP_ELIMIT;
  }
  a.m_w = make_magic(a.TW);
  a.m_thw = make_magic(a.TH * a.TW);
  a.tiles_x = 1;
  a.tiles_y = ceil_div(op.h, a.TH);
  a.vec = a.dma = 1;
  a.zeros = params;
  const size_t per_ch = (size_t(a.TN) * a.TH * (a.TW + 4) + size_t(32) * CB) * sizeof(float);
  const size_t epi_bytes = (size_t(3) * op.coutp + 96) * sizeof(float);     // BN vectors (+ the fused 1x1 conv's) in LDS
  int CC = int((kLdsBudget - 64 - epi_bytes) / (per_ch * 2)) & ~1;
  if (CC > op.cinp) CC = op.cinp;
  if (CC < 2) return FVP_ELIMIT;
  while (CC > 2 && size_t(CC) * a.TN * a.TH * (a.TW / 4 + 1) + 1 > 2048) CC -= 2;
  for (int d = CC; d >= 2 && d * 2 > CC; d -= 2)
    if (op.cinp % d == 0) { CC = d; break; }
  a.CC = CC;
  a.m_qpr = make_magic(a.TW / 4 + 1);
  a.m_rpc = make_magic(a.TN * a.TH);
  a.m_thp = make_magic(a.TH);
  const size_t lds0 = (std::max<size_t>(16 + 2 * (per_ch * CC + 16), 16 + 16384) + 15) & ~size_t(15);
  a.epi_off = int(lds0 / sizeof(float));
  const size_t lds = lds0 + epi_bytes;
  dim3 grid(a.tiles_y * ceil_div(planes, a.TN), 1, 2);
  ProfScope ps(FVP_K_CONV, s, 2.0 * op.cin * op.cout * 4.0 * hw * planes, 1, prof_level() >= 2);
  if (CB == 2)
    hipLaunchKernelGGL((k_conv_dma<1, 1, 2, 2, false, true>), grid, dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL((k_conv_dma<1, 1, 4, 1, false, true>), grid, dim3(256), lds, s, a);
  return launch_status();
}

// Tile selection: all couts per workgroup (CB = coutp/32), PB so that CB*PB <= 8 accumulator
// tiles per wave, the tile shaped to cover full image rows where possible.
static int plan_and_launch(const FvpConvOp& op, const float* params, float* const* bufs, int planes,
                           const uint8_t* plane_valid, int valid_div, hipStream_t s, float* pool_dst = nullptr,
                           const FvpConvOp* head = nullptr) {
  ConvArgs a{};
  a.pool_dst = pool_dst;
  if (head) {                                          // (eligibility checked by the caller)
    a.w2 = params + head->w_off;
    a.epi2 = params + head->e_off;
    a.dst2 = bufs[head->dst];
    a.cout2 = head->cout;
    a.flags2 = head->flags;
  }
  const bool tr = op.kind == FVP_OP_CONVT2;
  const int kh = tr ? 1 : op.kh, kw = tr ? 1 : op.kw;
  a.src = bufs[op.src];
  a.dst = bufs[op.dst];
  a.res = op.res >= 0 ? bufs[op.res] : nullptr;
  a.wts = params + op.w_off;
  a.epi = params + op.e_off;
  a.plane_valid = plane_valid;
  a.valid_div = valid_div > 0 ? valid_div : 1;
  a.m_vd = make_magic(a.valid_div);
  a.planes = planes;
  a.cin = op.cin;
  a.cinp = op.cinp;
  a.cout = op.cout;
  a.coutp = op.coutp;
  a.H = op.h;
  a.W = op.w;
  a.flags = op.flags;
  if (tr) {
    a.osx = 2;
    a.osy = op.h > 1 ? 2 : 1;
    a.ntapT = op.h > 1 ? 4 : 2;
    a.tapT_w = 2;
  } else {
    a.osx = a.osy = 1;
    a.ntapT = 1;
    a.tapT_w = 1;
  }
  a.OH = op.h * a.osy;
  a.OW = op.w * a.osx;
  if (double(planes) * std::max(op.cin, op.cout) * a.OH * a.OW >= 2147483648.0) return FVP_ELIMIT;
  a.wrow = op.coutp;
  if (tr || (kh == 1 && kw == 1)) {
    const int rc = plan_and_launch_reg(op, a, params, planes, s, tr);
    if (rc != -1) return rc;
  }
  if (tr && op.h > 1 && op.pair_off > 0 && !kNoPair && op.w % 4 == 0 && (op.coutp == 32 || op.coutp == 64))
    return plan_and_launch_tpair(op, a, params, planes, s);
  const int CBfull = op.coutp / 32;
  if (CBfull != 1 && CBfull != 2 && CBfull != 4) return FVP_ELIMIT;
  a.ablate = kAblate;
  if (!tr && op.wino_off > 0 && !kNoWino && op.cin == op.cinp && op.cout % 32 == 0 && op.cout == op.coutp) return wino_plan_and_launch(op, a, params, planes, s);
  // 7x7 front conv on 16x16x4 tiles (k_conv7): a SHAPE rule (map width, channels), never the number of planes.  Maps of
  // 64 / 128 columns (P2PNet: 30 planes per frame); CenterNet's 80 x 80 map (one plane per frame) stays on the pixel-pair
  // form - 8 planes: 25.9 us there, 28.9-29.9 us here (160 one-tile workgroups of 784 chained MFMAs per wave).
  if (!tr && kh == 7 && kw == 7 && op.pair_off > 0 && !kNoK7 && op.cout <= 16 && !(op.flags & FVP_EPI_RES) && !pool_dst && !head &&
      (op.w == 64 || op.w == 128) && op.cin <= 20 && op.h >= 4 &&
      double(4 * (op.cin <= 16 ? 4 : 5)) * op.h * op.w * 4.0 < 2147483648.0) {   // 32-bit byte offsets of all 4 * NCG channel slots (channels >= cin
                                                                                 // must stay out of range, never wrap: ADVICE round 5)
    const int ncg = op.cin <= 16 ? 4 : 5;
    a.wts = params + op.pair_off + size_t(op.cinp) * 7 * 8 * 32;       // the k-grouped copy behind the pixel-pair copy
    a.tiles_y = ceil_div(op.h, kK7Rows);
    const dim3 grid(planes * a.tiles_y);
    ProfScope ps(FVP_K_CONV, s, 2.0 * op.cin * op.cout * 49.0 * op.h * op.w * planes, 1, prof_level() >= 2);
#define FVP_K7(W_, NCG_)                                                                                      \
  {                                                                                                            \
    static LdsOptIn optin;                                                                                     \
    auto k = &k_conv7<W_, NCG_>;                                                                               \
    const size_t lds = std::max(kK7LdsPad, size_t(4 * NCG_) * conv7_cs(W_, kK7Rows) * sizeof(float));          \
    if (int e = lds_opt_in(optin, reinterpret_cast<const void*>(k), lds)) return e;                            \
    hipLaunchKernelGGL(k, grid, dim3(W_ / 16 * 64), lds, s, a);                                                \
  }
    if (ncg == 4) {
      if (op.w == 64) FVP_K7(64, 4) else FVP_K7(128, 4)
    } else {
      if (op.w == 64) FVP_K7(64, 5) else FVP_K7(128, 5)
    }
#undef FVP_K7
    return launch_status();
  }
  // Accumulator budget: CB*PB = 4 tiles of 32x32 per wave (~141 registers, 3 waves/SIMD).
  // Large grids keep all couts in one workgroup (input tile staged once); small grids split
  // couts over blockIdx.y and shrink the pixel tile so that more CUs get work.
  // PAIR layout (cout <= 16): tile columns are pixel pairs
  const bool pair = !tr && op.pair_off > 0 && !kNoPair && op.w % 4 == 0;
  const int wq = pair ? op.w / 2 : op.w;           // tile columns per image row
  const int hw = op.h * wq;
  const long px_total = long(hw) * planes;
  int CB = CBfull, PB = 4 / CBfull;
  if (px_total / (128 * PB) < 512) {
    CB = 1;
    PB = 4;
    while (PB > 1 && (px_total + 128 * PB - 1) / (128 * PB) * CBfull < 512) PB >>= 1;
  }
  if (kForcePB && CB * kForcePB <= 4) PB = kForcePB;
  // split K over the four waves (k_conv_dma<..., KS>): 3x3 layers with >= 64 channels on maps of 256 .. 576 pixels
  // (CenterNet's 20x20 level: reductions of 576-1152 terms, 13-52 pixel blocks per plane).  Measured on the MI355X:
  // 128->128 @20x20 32.3 -> 18.7 us at one plane (33.3 -> 31.0 at eight); the 40x40 level gains nothing at one plane
  // (18.9 -> 18.2 us) and doubles at eight planes (every workgroup stages the same weight slice), so it stays on the
  // plain form.  A SHAPE rule (map size, channels), never the number of planes: a frame's result does not depend on
  // the batch it is computed in.
  const bool ksplit = !tr && !pair && !kNoKSplit && kh == 3 && kw == 3 && op.w % 4 == 0 && op.w <= 32 && op.h * op.w >= 256 &&
                      op.h * op.w <= 576 && op.cinp >= 64 && op.cinp % 8 == 0 && !kNoDma;
  if (ksplit) {
    CB = 1;
    PB = 1;                                               // one image row (plus masked lanes) per workgroup
  }
  a.ablate = kAblate;
  const int TP = ksplit ? PB * 32 : PB * 128;
  if (hw <= TP) {                       // whole planes per tile
    a.TW = op.w;
    a.TH = op.h;
    a.TN = TP / hw > 0 ? TP / hw : 1;
  } else if (wq <= TP) {                // full-width row bands
    a.TW = op.w;
    a.TH = TP / wq;
    a.TN = 1;
  } else {
    a.TW = TP;
    a.TH = 1;
    a.TN = 1;
  }
  a.m_w = make_magic(pair ? wq : a.TW);
  a.m_thw = make_magic(a.TH * (pair ? wq : a.TW));
  a.tiles_x = ceil_div(op.w, a.TW);
  a.tiles_y = ceil_div(op.h, a.TH);
  const int pgroups = ceil_div(planes, a.TN);
  // channel chunk: largest even CC that fits the LDS budget
  a.vec = (a.TW == op.w && op.w % 4 == 0) ? 1 : 0;
  a.dma = (a.vec && !kNoDma) ? 1 : 0;
  a.zeros = params;
  const int twp = a.dma ? a.TW + 4 : (a.vec ? a.TW + 8 : a.TW + kw - 1);
  const int kt = pair ? kw + 1 : kw;               // taps per kernel row in the packed layout
  const size_t per_ch = (size_t(a.TN) * (a.TH + kh - 1) * twp + size_t(kh) * kt * 32 * CB) * sizeof(float);
  // the pipelined kernel keeps two chunks in LDS
  const size_t epi_bytes = a.dma ? (size_t(3) * op.coutp + 96) * sizeof(float) : 0;   // BN vectors in LDS (k_conv_dma)
  int CC = int((kLdsBudget - 64 - epi_bytes) / (per_ch * (a.dma ? 2 : 1))) & ~1;
  if (CC > op.cinp) CC = op.cinp;
  if (CC < 2) return FVP_ELIMIT;
  if (a.dma)                                          // k_conv_dma keeps <= 8 staging items per lane
    while (CC > 2 && size_t(CC) * a.TN * (a.TH + kh - 1) * (a.TW / 4 + 1) + 1 > 2048) CC -= 2;
  for (int d = CC; d >= 2 && d * 2 > CC; d -= 2)      // avoid a ragged last chunk when a close divisor exists
    if (op.cinp % d == 0) { CC = d; break; }
  if (ksplit) {                                       // whole groups of four channel pairs; no ragged chunk
    if (!a.dma || CC < 8) return FVP_ELIMIT;
    CC &= ~7;
    while (CC > 8 && op.cinp % CC) CC -= 8;
  }
  a.CC = CC;
  if (a.dma && !buf_dma_range_ok(a.TN, op.cin, op.h, op.w, double(op.cinp) * kh * kt * a.wrow)) return FVP_ELIMIT;
  // stage_buf packs an item's channel-in-chunk into 8 bits
  if (a.dma && CC > 255) return FVP_ELIMIT;
  a.m_qpr = make_magic(a.dma ? a.TW / 4 + 1 : a.TW / 4);
  a.m_rpc = make_magic(a.TN * (a.TH + kh - 1));
  a.m_thp = make_magic(a.TH + kh - 1);
  const size_t xs_floats = (size_t(CC) * a.TN * (a.TH + kh - 1) * twp + 3) & ~size_t(3);
  size_t lds = a.dma ? std::max<size_t>(16 + 2 * (per_ch * CC + 16), 16 + 16384)
                     : (xs_floats + size_t(CC) * kh * kt * 32 * CB) * sizeof(float);
  if (ksplit) lds = std::max<size_t>(lds, 16 + size_t(3) * PB * 16 * 64 * sizeof(float));   // partial tiles of waves 1..3
  if (a.dma) {
    lds = (lds + 15) & ~size_t(15);
    a.epi_off = int(lds / sizeof(float));
    lds += epi_bytes;
  }
  dim3 grid(a.tiles_x * a.tiles_y * pgroups, CBfull / CB, a.ntapT);
  // algorithmic FLOPs (2*MAC on the true channel counts)
  const double taps = tr ? double(a.ntapT) : double(op.kh * op.kw);
  ProfScope ps(FVP_K_CONV, s, 2.0 * op.cin * op.cout * taps * op.h * op.w * planes, 1, prof_level() >= 2);
  if (ksplit) {
    hipLaunchKernelGGL((k_conv_dma<3, 3, 1, 1, false, false, true>), grid, dim3(256), lds, s, a);
    return launch_status();
  }
  if (pair) {
    if (!a.dma || CB != 1 || kh != 7 || kw != 7) return FVP_ELIMIT;
    a.wts = params + op.pair_off;
    switch (PB) {
      case 1: hipLaunchKernelGGL((k_conv_dma<7, 7, 1, 1, true>), grid, dim3(256), lds, s, a); break;
      case 2: hipLaunchKernelGGL((k_conv_dma<7, 7, 1, 2, true>), grid, dim3(256), lds, s, a); break;
      case 4: hipLaunchKernelGGL((k_conv_dma<7, 7, 1, 4, true>), grid, dim3(256), lds, s, a); break;
      default: return FVP_ELIMIT;
    }
    return launch_status();
  }
  return dispatch_conv(kh, kw, CB, PB, a, grid, lds, s);
}

}  // namespace fvp

using namespace fvp;

extern "C" int fvp_conv_stack_run(const FvpConvOp* ops, int nops, const float* params, float* const* bufs, int nbufs,
                                  int planes, const uint8_t* plane_valid, int valid_div, fvp_stream_t s) {
  FVP_REQUIRE(ops && params && bufs && nops >= 0 && planes >= 0);
  if (planes == 0) return 0;
  // coarse profiling: one event pair around the whole stack (the max-pool launches inside are
  // counted in the time, not in the FLOPs)
  double flops = 0.0;
  long nconv = 0;
  for (int i = 0; i < nops; ++i) {
    const FvpConvOp& op = ops[i];
    if (op.kind == FVP_OP_POOL2) continue;
    const double taps = op.kind == FVP_OP_CONVT2 ? (op.h > 1 ? 4.0 : 2.0) : double(op.kh * op.kw);
    flops += 2.0 * op.cin * op.cout * taps * op.h * op.w * planes;
    nconv += 1;
  }
  ProfScope stack_scope(FVP_K_CONV, as_stream(s), flops, nconv, prof_level() == 1);
  // a 2x2 max-pool that reads the output of a Winograd conv is produced by that conv's epilogue
  // (one pooled value per 2x2 output tile): the pool launch and its re-read of the map disappear
  unsigned long long pooled = 0;                       // bit j: pool op j already done
  for (int i = 0; i < nops; ++i) {
    const FvpConvOp& op = ops[i];
    FVP_REQUIRE(op.src >= 0 && op.src < nbufs && op.dst >= 0 && op.dst < nbufs && op.res < nbufs);
    int rc;
    if (op.kind == FVP_OP_POOL2 && i < 64 && ((pooled >> i) & 1)) continue;
    if (op.kind == FVP_OP_POOL2) {
      FVP_REQUIRE(op.w % 2 == 0 && (op.h == 1 || op.h % 2 == 0));
      const long total = long(planes) * op.cin * (op.h > 1 ? op.h / 2 : 1) * (op.w / 2);
      ProfScope ps(FVP_K_OTHER, as_stream(s), 0.0, 1, prof_level() >= 2);
      hipLaunchKernelGGL(k_pool2, dim3(unsigned((total + 255) / 256)), dim3(256), 0, as_stream(s),
                         (const float*)bufs[op.src], bufs[op.dst], total, op.h, op.w, plane_valid,
                         valid_div > 0 ? valid_div : 1, op.cin);
      rc = launch_status();
    } else if (op.kind == FVP_OP_CONV || op.kind == FVP_OP_CONVT2) {
      float* pool_dst = nullptr;
      if (op.kind == FVP_OP_CONV && op.wino_off > 0 && op.cin == op.cinp && op.cout % 32 == 0 && op.cout == op.coutp && !kNoWino && !kNoPoolFuse) {
        for (int j = i + 1; j < nops && j < 64; ++j)
          if (ops[j].kind == FVP_OP_POOL2 && ops[j].src == op.dst && ops[j].h == op.h && ops[j].w == op.w &&
              ops[j].h > 1 && ops[j].cin == op.cout && ops[j].dst >= 0 && ops[j].dst < nbufs) {
            pool_dst = bufs[ops[j].dst];
            pooled |= 1ull << j;
            break;
          }
      }
      // a 32-cout paired transposed conv whose only consumer is the next op, a plain 1x1 conv (P2PNet's output
      // layer): that conv runs in the transposed conv's epilogue and the 32-channel map is never stored
      const FvpConvOp* head = nullptr;
      if (op.kind == FVP_OP_CONVT2 && op.h > 1 && op.pair_off > 0 && !kNoPair && !kNoHeadFuse && op.w % 4 == 0 && op.coutp == 32 &&
          i + 1 < nops) {
        const FvpConvOp& nx = ops[i + 1];
        bool only = nx.kind == FVP_OP_CONV && nx.kh == 1 && nx.kw == 1 && nx.src == op.dst && nx.res < 0 && nx.cin == op.cout &&
                    nx.cinp == 32 && nx.coutp == 32 && (nx.cout <= 16 || reg_takes_fused_head(op, planes)) && nx.h == 2 * op.h && nx.w == 2 * op.w && nx.dst != op.dst &&
                    nx.dst >= 0 && nx.dst < nbufs;
        for (int j = i + 2; j < nops && only; ++j) only = ops[j].src != op.dst && ops[j].res != op.dst;
        if (only) head = &nx;
      }
      rc = plan_and_launch(op, params, bufs, planes, plane_valid, valid_div, as_stream(s), pool_dst, head);
      if (!rc && head) ++i;
    } else {
      rc = FVP_EINVAL;
    }
    if (rc) return rc;
  }
  return 0;
}

extern "C" int fvp_pack_conv(const float* weight, const float* bias, const float* bn_gamma, const float* bn_beta,
                             const float* bn_mean, const float* bn_var, float eps, int transposed,
                             const FvpConvOp* op, float* params, fvp_stream_t s) {
  FVP_REQUIRE(weight && op && params);
  FVP_REQUIRE(!bn_gamma || (bn_beta && bn_mean && bn_var));
  const int KK = op->kh * op->kw;
  const int nw = op->cinp * KK * op->coutp;
  const int n = nw > op->coutp ? nw : op->coutp;
  hipLaunchKernelGGL(k_pack_conv, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(s), weight, bias, bn_gamma,
                     bn_beta, bn_mean, bn_var, eps, transposed, op->cin, op->cout, op->cinp, op->coutp, op->kh,
                     op->kw, params + op->w_off, params + op->e_off);
  if (op->pair_off > 0 && transposed) {
    FVP_REQUIRE(op->kh == 2 && op->kw == 2);
    hipLaunchKernelGGL(k_pack_tpair, dim3(ceil_div(4 * op->cinp * op->coutp, 256)), dim3(256), 0, as_stream(s), weight,
                       op->cin, op->cout, op->cinp, op->coutp, params + op->pair_off);
  } else if (op->pair_off > 0) {
    FVP_REQUIRE(op->cout <= 16 && op->coutp == 32);
    hipLaunchKernelGGL(k_pack_pair, dim3(ceil_div(op->cinp * op->kh * (op->kw + 1) * 32, 256)), dim3(256), 0,
                       as_stream(s), weight, op->cin, op->cout, op->cinp, op->kh, op->kw, params + op->pair_off);
    if (op->kh == 7 && op->kw == 7) {                                  // k-grouped copy for k_conv7, behind the pixel-pair copy
      const int ncg = op->cin <= 16 ? 4 : ceil_div(op->cin, 4);        // (k_conv7 runs four groups for every cin <= 16)
      hipLaunchKernelGGL(k_pack_k7, dim3(ceil_div(ncg * kK7GroupFloats, 256)), dim3(256), 0, as_stream(s), weight, op->cin, op->cout, ncg,
                         params + op->pair_off + size_t(op->cinp) * 7 * 8 * 32);
    }
  }
  if (op->wino_off > 0) {
    FVP_REQUIRE(!transposed && op->kh == 3 && op->kw == 3);
    if (int rc = wino_pack(weight, *op, params, as_stream(s))) return rc;
  }
  return launch_status();
}
