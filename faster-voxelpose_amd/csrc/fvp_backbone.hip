// Pose-ResNet backbone (lib/models/resnet.py:98-215) in bf16 on the matrix cores: every conv /
// transposed conv is an implicit GEMM  D[pixel][cout] = sum_k X[pixel][k] * W[cout][k]  over NHWC
// bf16 activations, k = (tap, cin), on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; eval
// BatchNorm (folded to scale / shift), residual add and ReLU run in the epilogue.
//
//   A operand = pixels  (row i = pixel: lane l holds X[l&31][8*(l>>5) .. +7] of a 16-wide k step)
//   B operand = weights (col j = cout,  same k mapping), packed [cout][k] so that both operands are
//               read from LDS as one ds_read_b128 per lane
//   D         : lane l holds cout column l&31 and pixel rows (r&3) + 8*(r>>2) + 4*(l>>5): 32 lanes
//               store 32 consecutive channels of one pixel = 64 contiguous bytes of NHWC
//
// Workgroup = 4 waves (2 x 2), tile 128 pixels x 128 couts (64 for the 64-channel layers), k chunks
// of 64 through one LDS buffer (37 KB: four workgroups per CU) with a 144-byte row pitch (conflict-free
// b128 reads); the next chunk's global loads (branch-free, address-clamped) are in flight in registers
// while the current one is multiplied.  A per-launch tap table
// (dy, dx) covers strided convs and the four parity classes of ConvTranspose(k4, s2, p1) with the
// same kernel; the last layer writes fp32 heatmaps directly in the channels-last layout the
// projection kernels read (and / or NCHW, the reference's layout).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "fvp_common.h"

#ifndef FVP_BB_NBUF
#define FVP_BB_NBUF 1
#endif

namespace fvp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct alignas(16) Bf8 { uint32_t w[4]; };          // 8 bf16

__device__ __forceinline__ uint16_t f2bf(float f) {   // round to nearest even
  uint32_t u = uint32_t(__float_as_int(f));
  u += 0x7fffu + ((u >> 16) & 1u);
  return uint16_t(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __int_as_float(int(uint32_t(h) << 16)); }

__device__ __forceinline__ f32x16 mfma_bf16(const Bf8& a, const Bf8& b, f32x16 c) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct BbConvArgs {
  const uint16_t* in;
  uint16_t* out;
  float* out_cl;          // fp32 [N][ROH*ROW][out_jp] (last layer) or null
  float* out_nchw;        // fp32 [N][Cout][ROH][ROW] (last layer) or null
  const uint16_t* res;
  const uint16_t* w;      // [Coutp][K]
  const float* epi;       // scale | shift, each Coutp
  int N, H, W, Cinp, cin_log2;
  int OH, OW;             // grid the GEMM rows walk (output grid; input grid for a transposed-conv class)
  int ROH, ROW, Cbuf;     // real output tensor dims (NHWC, Cbuf channels)
  int Cout, Coutp;
  int stride, stride_x;   // input pixel = (oy*stride + dy, ox*stride_x + dx)
  int os, py, px;         // output pixel = o*os + p
  int ntaps, K, relu, out_jp;
  int ncls;               // 1, or 4 parity classes of a transposed conv on blockIdx.z (tap tables / weights per class)
  signed char dy[64], dx[64];
};

// pixel index -> (image, row, col).  Plain integer division: the operand reaches N*OH*OW (millions), far
// beyond the exact range of a 32-bit reciprocal multiply, and the decode runs a handful of times per thread.
__device__ __forceinline__ void bb_decode(int m, int OW, int OHW, int& n, int& oy, int& ox) {
  n = m / OHW;
  const int r = m - n * OHW;
  oy = r / OW;
  ox = r - oy * OW;
}

template <int BN>
__global__ void __launch_bounds__(256) k_bb_conv(BbConvArgs a) {
  constexpr int BM = 128, BK = 64, LP = BK + 8;       // LDS row pitch in bf16 (144 bytes: conflict-free b128 reads)
  constexpr int WN = BN / 2;                          // couts per wave: 64 or 32
  constexpr int NJ = WN / 32;                         // cout tiles per wave
  constexpr int KS = BK / 16;                         // MFMA k steps per chunk
  constexpr int NG = BK / 8;                          // 16-byte groups per row and chunk
  constexpr int AU = BM * NG / 256;                   // A vectors per thread and chunk (4)
  constexpr int BU = BN * NG / 256;                   // B vectors per thread and chunk (4 / 2)
  HIP_DYNAMIC_SHARED(uint16_t, smem)                  // As[2][BM][LP] | Bs[2][BN][LP] | taps
  constexpr int NBUF = FVP_BB_NBUF;
  uint16_t(*As)[BM][LP] = reinterpret_cast<uint16_t(*)[BM][LP]>(smem);
  uint16_t(*Bs)[BN][LP] = reinterpret_cast<uint16_t(*)[BN][LP]>(smem + NBUF * BM * LP);
  signed char* tdy = reinterpret_cast<signed char*>(smem + NBUF * BM * LP + NBUF * BN * LP);
  signed char* tdx = tdy + 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = a.N * a.OH * a.OW;
  const int m0 = blockIdx.x * BM, co0 = blockIdx.y * BN;
  const int cls = blockIdx.z;                          // parity class of a transposed conv (0 otherwise)
  const int opy = a.ncls > 1 ? cls >> 1 : a.py, opx = a.ncls > 1 ? (cls & 1) : a.px;
  const uint16_t* wcls = a.w + size_t(cls) * a.Coutp * a.K;
  if (t < 64) {
    tdy[t] = a.dy[(t + cls * a.ntaps) & 63];
    tdx[t] = a.dx[(t + cls * a.ntaps) & 63];
  }

  // ---- this thread's staging duty: AU groups of one pixel row pair ... every vector = (row, 8-wide k group)
  // vector v = t + 256*u: row = v / NG, group = v % NG  (NG = 8: a wave reads 8 rows x 128 contiguous bytes)
  int a_iy0[AU], a_ix0[AU], a_base[AU];               // per vector: o*stride per axis, image offset n*H*W
  bool a_ok[AU];
#pragma unroll
  for (int u = 0; u < AU; ++u) {
    const int v = t + 256 * u, row = v / NG;
    const int am = m0 + row;
    a_ok[u] = am < M;
    const int mm = a_ok[u] ? am : 0;
    int n, oy, ox;
    bb_decode(mm, a.OW, a.OH * a.OW, n, oy, ox);
    a_iy0[u] = oy * a.stride;
    a_ix0[u] = ox * a.stride_x;
    a_base[u] = n * a.H * a.W;
  }
  const int ag = t % NG;                              // the same k group for all of this thread's vectors
  __syncthreads();                                    // tap table visible

  Bf8 ra[AU], rb[BU];
  auto gload = [&](int chunk) {
    const int kk = chunk * BK + ag * 8;
    const bool k_ok = kk < a.K;
    const int kc = k_ok ? kk : 0;
    const int tap = kc >> a.cin_log2, c0 = kc & (a.Cinp - 1);
    const int dy = tdy[tap], dx = tdx[tap];
#pragma unroll
    for (int u = 0; u < AU; ++u) {                    // unconditional, address-clamped loads; masked afterwards
      const int iy = a_iy0[u] + dy, ix = a_ix0[u] + dx;
      const bool ok = k_ok && a_ok[u] && unsigned(iy) < unsigned(a.H) && unsigned(ix) < unsigned(a.W);
      const size_t off = ok ? (size_t(a_base[u] + iy * a.W + ix) * a.Cinp + c0) : 0;
      ra[u] = *reinterpret_cast<const Bf8*>(a.in + off);
      if (!ok) ra[u] = Bf8{{0u, 0u, 0u, 0u}};
    }
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const int row = (t + 256 * u) / NG;
      rb[u] = *reinterpret_cast<const Bf8*>(wcls + size_t(co0 + row) * a.K + kc);
      if (!k_ok) rb[u] = Bf8{{0u, 0u, 0u, 0u}};
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < AU; ++u) *reinterpret_cast<Bf8*>(&As[buf][(t + 256 * u) / NG][ag * 8]) = ra[u];
#pragma unroll
    for (int u = 0; u < BU; ++u) *reinterpret_cast<Bf8*>(&Bs[buf][(t + 256 * u) / NG][ag * 8]) = rb[u];
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nchunks = (a.K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = NBUF == 2 ? (c & 1) : 0;
    if (c + 1 < nchunks) gload(c + 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Bf8 fa[2], fb[NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const Bf8*>(&As[buf][wm * 64 + i * 32 + l31][ks * 16 + 8 * half]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const Bf8*>(&Bs[buf][wn * WN + j * 32 + l31][ks * 16 + 8 * half]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma_bf16(fa[i], fb[j], acc[i][j]);
    }
    if (NBUF == 1) __syncthreads();                    // single buffer: everyone done reading before the refill
    if (c + 1 < nchunks) lstore(NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

  // ---- epilogue: BN scale / shift, residual, ReLU
  const float* scale = a.epi;
  const float* shift = a.epi + a.Coutp;
  if (a.out && !a.out_cl && !a.out_nchw && (a.Cbuf & 7) == 0) {
    // bf16 NHWC output: the wave's tile goes through LDS as fp32 so that a lane ends up with 8 consecutive
    // channels of one pixel: 16-byte residual loads and 16-byte stores
    // (32 couts at a time: 9 KB per wave, so the epilogue does not raise the kernel's LDS footprint)
    constexpr int EP = 32 + 4;
    float* et = reinterpret_cast<float*>(smem) + wave * (64 * EP);
    __syncthreads();                                                   // every wave is done with As / Bs
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float sc = scale[co0 + wn * WN + j * 32 + l31], sh = shift[co0 + wn * WN + j * 32 + l31];
      __builtin_amdgcn_wave_barrier();                                  // previous block's readers are done
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * EP + l31] = acc[i][j][r] * sc + sh;
      __builtin_amdgcn_s_waitcnt(0xc07f);                               // own LDS writes landed (wave-private tile)
      __builtin_amdgcn_wave_barrier();
      constexpr int NV = 64 * 4 / 64;                                   // 64 rows x 4 groups of 8 channels
      size_t pixv[NV];
      bool okv[NV];
      Bf8 resv[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) {                                   // unconditional, clamped residual loads first
        const int idx = lane + 64 * v, row = idx >> 2, g = idx & 3;
        const int m = m0 + wm * 64 + row;
        const int co = co0 + wn * WN + j * 32 + g * 8;
        okv[v] = m < M && co < a.Cbuf;
        const int mm = okv[v] ? m : 0;
        size_t pix = size_t(mm);
        if (a.os != 1) {
          int n_, oy, ox;
          bb_decode(mm, a.OW, a.OH * a.OW, n_, oy, ox);
          pix = (size_t(n_) * a.ROH + oy * a.os + opy) * a.ROW + ox * a.os + opx;
        }
        pixv[v] = pix * a.Cbuf + (okv[v] ? co : 0);
        if (a.res) resv[v] = *reinterpret_cast<const Bf8*>(a.res + pixv[v]);
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int idx = lane + 64 * v, row = idx >> 2, g = idx & 3;
        const float4 lo = *reinterpret_cast<const float4*>(et + row * EP + g * 8);
        const float4 hi = *reinterpret_cast<const float4*>(et + row * EP + g * 8 + 4);
        float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        Bf8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v0 = x[2 * e], v1 = x[2 * e + 1];
          if (a.res) {
            v0 += bf2f(uint16_t(resv[v].w[e] & 0xffffu));
            v1 += bf2f(uint16_t(resv[v].w[e] >> 16));
          }
          if (a.relu) {
            v0 = fmaxf(v0, 0.0f);
            v1 = fmaxf(v1, 0.0f);
          }
          o.w[e] = uint32_t(f2bf(v0)) | (uint32_t(f2bf(v1)) << 16);
        }
        if (okv[v]) *reinterpret_cast<Bf8*>(a.out + pixv[v]) = o;
      }
    }
    return;
  }
  // fp32 heatmaps of the last layer / odd channel counts (small): element-wise stores
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int co = co0 + wn * WN + j * 32 + l31;
    const float sc = scale[co], sh = shift[co];        // epi vectors are padded to Coutp
    const bool co_ok = co < a.Cout;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= M) continue;
        const int n_ = m / (a.OH * a.OW);
        const size_t pix = size_t(m);                  // the heatmap layer is a plain 1x1 conv (os = 1)
        float v = acc[i][j][r] * sc + sh;
        if (a.res && co_ok) v += bf2f(a.res[pix * a.Cbuf + co]);
        if (a.relu) v = fmaxf(v, 0.0f);
        if (a.out && co < a.Cbuf) a.out[pix * a.Cbuf + co] = co_ok ? f2bf(v) : uint16_t(0);
        if (a.out_cl && co < a.out_jp) a.out_cl[pix * a.out_jp + co] = co_ok ? v : 0.0f;
        if (a.out_nchw && co_ok) {
          const size_t hw = size_t(a.ROH) * a.ROW;
          a.out_nchw[(size_t(n_) * a.Cout + co) * hw + (pix - size_t(n_) * hw)] = v;
        }
      }
    }
  }
}

// Large-tile variant for layers with >= 256 couts and enough pixels: 8 waves, 256 pixels x 256 couts per
// workgroup, 64 x 128 per wave (2 x 4 MFMA tiles = 128 accumulator registers, two waves per SIMD), so a k step
// needs 6 ds_read_b128 for 8 MFMAs instead of 4 for 4 -- the 128 x 128 kernel is bound by LDS read bandwidth.
// Both operand tiles are copied by the LDS-DMA (global_load_lds, 16 B per lane, no staging registers) into two
// 72 KB slots: chunk c+1 streams in while chunk c is multiplied.  Row pitch 144 bytes = 9 DMA lanes per row, the
// ninth reading a zero page; out-of-image taps, rows beyond M and k beyond K read the zero page as well.
__global__ void __launch_bounds__(512, 2) k_bb_conv_big(BbConvArgs a, const uint16_t* __restrict__ zeros) {
  constexpr int BM = 256, BN = 256, BK = 64, LP = BK + 8, QPR = 9;   // quads (16 B) per LDS row incl. the pad quad
  constexpr int SLOT = (BM + BN) * LP;                               // bf16 elements per slot
  constexpr int NITEM = BM * QPR;                                    // DMA items of one operand tile (2304)
  constexpr int NR = (NITEM + 511) / 512;                            // DMA rounds per operand (5; the last one half full)
  HIP_DYNAMIC_SHARED(uint16_t, smem)
  signed char* tdy = reinterpret_cast<signed char*>(smem + 2 * SLOT);
  signed char* tdx = tdy + 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = a.N * a.OH * a.OW;
  const int m0 = blockIdx.x * BM, co0 = blockIdx.y * BN;
  const int cls = blockIdx.z;                          // parity class of a transposed conv (0 otherwise)
  const int opy = a.ncls > 1 ? cls >> 1 : a.py, opx = a.ncls > 1 ? (cls & 1) : a.px;
  const uint16_t* wcls = a.w + size_t(cls) * a.Coutp * a.K;
  if (t < 64) {
    tdy[t] = a.dy[(t + cls * a.ntaps) & 63];
    tdx[t] = a.dx[(t + cls * a.ntaps) & 63];
  }
  // ---- this lane's DMA items (fixed over the k loop): A item -> (pixel row, k group), B item -> (cout row, k group)
  int a_iy0[NR], a_ix0[NR], a_base[NR], qv[NR];       // qv = k group 0..7, or -1: pad quad / item beyond the tile
  bool a_ok[NR];                                      // pixel row inside M
  int b_row[NR];                                      // element offset of the weight row
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int it = (wave + 8 * j) * 64 + lane;
    const int row = it / QPR, q = it - row * QPR;
    qv[j] = (it < NITEM && q < 8) ? q : -1;
    a_ok[j] = false;
    a_iy0[j] = a_ix0[j] = a_base[j] = 0;
    b_row[j] = (co0 + (row < BN ? row : 0)) * a.K;    // packed weights are padded to Coutp rows
    const int am = m0 + row;
    if (qv[j] >= 0 && am < M) {
      int n, oy, ox;
      bb_decode(am, a.OW, a.OH * a.OW, n, oy, ox);
      a_iy0[j] = oy * a.stride;
      a_ix0[j] = ox * a.stride_x;
      a_base[j] = n * a.H * a.W;
      a_ok[j] = true;
    }
  }
  __syncthreads();                                    // tap table visible
  const int nchunks = (a.K + BK - 1) / BK;
  auto stage = [&](int chunk, int slot) {
    uint16_t* As = smem + slot * SLOT;
    uint16_t* Bs = As + BM * LP;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int g = wave + 8 * j;
      if (g * 64 < NITEM) {                           // wave-uniform: the last round is issued by half of the waves
        const int kk = chunk * BK + (qv[j] < 0 ? 0 : qv[j]) * 8;
        const bool k_ok = qv[j] >= 0 && kk < a.K;
        const int kc = k_ok ? kk : 0;
        const int tap = kc >> a.cin_log2, c0 = kc & (a.Cinp - 1);
        const int iy = a_iy0[j] + tdy[tap], ix = a_ix0[j] + tdx[tap];
        const bool ok = a_ok[j] && k_ok && unsigned(iy) < unsigned(a.H) && unsigned(ix) < unsigned(a.W);
        const uint16_t* src = ok ? a.in + (size_t(a_base[j] + iy * a.W + ix) * a.Cinp + c0) : zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(As + g * 512), 16, 0, 0);
        const uint16_t* wsrc = k_ok ? wcls + size_t(b_row[j]) + kc : zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc,
                                         (__attribute__((address_space(3))) void*)(Bs + g * 512), 16, 0, 0);
      }
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  stage(0, 0);
  wait_vmcnt(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int slot = c & 1;
    if (c + 1 < nchunks) stage(c + 1, slot ^ 1);
    const uint16_t* As = smem + slot * SLOT;
    const uint16_t* Bs = As + BM * LP;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      Bf8 fa[2], fb[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const Bf8*>(As + (wm * 64 + i * 32 + l31) * LP + ks * 16 + 8 * half);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const Bf8*>(Bs + (wn * 128 + j * 32 + l31) * LP + ks * 16 + 8 * half);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16(fa[i], fb[j], acc[i][j]);
    }
    wait_vmcnt(0);                                    // this wave's share of chunk c+1 has landed
    __syncthreads();                                  // ... everybody's; and everybody is done reading chunk c
  }

  // ---- epilogue: as in k_bb_conv (bf16 NHWC output through wave-private fp32 LDS tiles, 32 couts at a time)
  const float* scale = a.epi;
  const float* shift = a.epi + a.Coutp;
  constexpr int EP = 32 + 4;
  float* et = reinterpret_cast<float*>(smem) + wave * (64 * EP);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float sc = scale[co0 + wn * 128 + j * 32 + l31], sh = shift[co0 + wn * 128 + j * 32 + l31];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * EP + l31] = acc[i][j][r] * sc + sh;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    size_t pixv[4];
    bool okv[4];
    Bf8 resv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int idx = lane + 64 * v, row = idx >> 2, g = idx & 3;
      const int m = m0 + wm * 64 + row;
      const int co = co0 + wn * 128 + j * 32 + g * 8;
      okv[v] = m < M && co < a.Cbuf;
      const int mm = okv[v] ? m : 0;
      size_t pix = size_t(mm);
      if (a.os != 1) {
        int n_, oy, ox;
        bb_decode(mm, a.OW, a.OH * a.OW, n_, oy, ox);
        pix = (size_t(n_) * a.ROH + oy * a.os + opy) * a.ROW + ox * a.os + opx;
      }
      pixv[v] = pix * a.Cbuf + (okv[v] ? co : 0);
      if (a.res) resv[v] = *reinterpret_cast<const Bf8*>(a.res + pixv[v]);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int idx = lane + 64 * v, row = idx >> 2, g = idx & 3;
      const float4 lo = *reinterpret_cast<const float4*>(et + row * EP + g * 8);
      const float4 hi = *reinterpret_cast<const float4*>(et + row * EP + g * 8 + 4);
      float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      Bf8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v0 = x[2 * e], v1 = x[2 * e + 1];
        if (a.res) {
          v0 += bf2f(uint16_t(resv[v].w[e] & 0xffffu));
          v1 += bf2f(uint16_t(resv[v].w[e] >> 16));
        }
        if (a.relu) {
          v0 = fmaxf(v0, 0.0f);
          v1 = fmaxf(v1, 0.0f);
        }
        o.w[e] = uint32_t(f2bf(v0)) | (uint32_t(f2bf(v1)) << 16);
      }
      if (okv[v]) *reinterpret_cast<Bf8*>(a.out + pixv[v]) = o;
    }
  }
}

// MaxPool2d(3, stride 2, padding 1) on NHWC bf16: one thread per (output pixel, 8-channel group).
__global__ void __launch_bounds__(256)
k_bb_maxpool(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int N, int H, int W, int C, int OH, int OW) {
  const long i = long(blockIdx.x) * 256 + threadIdx.x;
  const int cg = C / 8;
  if (i >= long(N) * OH * OW * cg) return;
  const int g = int(i % cg);
  long p = i / cg;
  const int ox = int(p % OW);
  p /= OW;
  const int oy = int(p % OH), n = int(p / OH);
  float m[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
      if (unsigned(iy) >= unsigned(H) || unsigned(ix) >= unsigned(W)) continue;
      const Bf8 v = *reinterpret_cast<const Bf8*>(in + (size_t(n * H + iy) * W + ix) * C + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], bf2f(uint16_t(v.w[e >> 1] >> (16 * (e & 1)))));
    }
  Bf8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o.w[e] = uint32_t(f2bf(m[2 * e])) | (uint32_t(f2bf(m[2 * e + 1])) << 16);
  *reinterpret_cast<Bf8*>(out + (size_t(n * OH + oy) * OW + ox) * C + g * 8) = o;
}

// Images NCHW fp32 [N][C<=4][H][W] -> NHWC bf16 with 4 channels per pixel (channel 3 zero for RGB).  Read as
// [N][H][W/2][8] this is the input of the stem conv in its pixel-pair form (below).
__global__ void __launch_bounds__(256)
k_bb_input(const float* __restrict__ img, uint16_t* __restrict__ out, int N, int C, int H, int W) {
  const long i = long(blockIdx.x) * 256 + threadIdx.x;     // one thread per pixel PAIR
  const long hw = long(H) * W, hw2 = hw / 2;
  if (i >= long(N) * hw2) return;
  const int n = int(i / hw2);
  const long p = (i - n * hw2) * 2;
  uint16_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = 0; e < 2; ++e)
    for (int c = 0; c < C && c < 4; ++c) v[4 * e + c] = f2bf(img[(size_t(n) * C + c) * hw + p + e]);
  Bf8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o.w[e] = uint32_t(v[2 * e]) | (uint32_t(v[2 * e + 1]) << 16);
  *reinterpret_cast<Bf8*>(out + size_t(i) * 8) = o;
}

// Stem weights [Cout][Cin<=4][7][7] (stride 2, pad 3) in the pixel-pair form: the image is read as
// [H][W/2] pixel pairs of 8 channels (2 pixels x 4), where the conv has stride (2, 1) and 7 x 4 taps:
// pair tap p in 0..3 sits at pair offset p - 2 and holds kernel columns kw = 2p - 1 + e (e = pixel of the pair;
// kw = -1 does not exist -> zero).  K = 7 * 4 * 8 = 224 instead of 7 * 7 * 8 = 392 with per-pixel padding.
__global__ void __launch_bounds__(256)
k_bb_pack_stem(const float* __restrict__ w, int cin, int cout, int coutp, uint16_t* __restrict__ dst) {
  const int K = 7 * 4 * 8;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= coutp * K) return;
  const int co = i / K, kk = i - co * K;
  const int tap = kk >> 3, e = (kk >> 2) & 1, c = kk & 3;
  const int kh = tap >> 2, p = tap & 3, kw = 2 * p - 1 + e;
  float v = 0.0f;
  if (co < cout && c < cin && kw >= 0) v = w[((size_t(co) * cin + c) * 7 + kh) * 7 + kw];
  dst[i] = f2bf(v);
}

// Weights -> [cls][Coutp][ntaps*Cinp] bf16.  Conv: weight [Cout][Cin][KH][KW], taps (kh, kw) row-major.
// ConvTranspose(k4,s2,p1): weight [Cin][Cout][4][4]; class (py, px) uses taps (a, b) in {0,1}^2 with
// kh = 1 - py + 2a ... see tap tables on the host: kh = khs[py][a], kw = khs[px][b].
__global__ void __launch_bounds__(256)
k_bb_pack_w(const float* __restrict__ w, int transposed, int cin, int cout, int cinp, int coutp, int kh, int kw,
            uint16_t* __restrict__ dst) {
  const int ncls = transposed ? 4 : 1;
  const int ntaps = transposed ? 4 : kh * kw;
  const int K = ntaps * cinp;
  const long total = long(ncls) * coutp * K;
  const long i = long(blockIdx.x) * 256 + threadIdx.x;
  if (i >= total) return;
  const int kk = int(i % K);
  long r = i / K;
  const int co = int(r % coutp), cls = int(r / coutp);
  const int tap = kk / cinp, ci = kk - tap * cinp;
  float v = 0.0f;
  if (co < cout && ci < cin) {
    if (!transposed) {
      v = w[((size_t(co) * cin + ci) * kh + tap / kw) * kw + tap % kw];
    } else {
      const int py = cls >> 1, px = cls & 1, ta = tap >> 1, tb = tap & 1;
      const int ky = py ? 2 * ta : 1 + 2 * ta;        // oy = 2y + py gathers input rows y + dy via kernel row ky
      const int kx = px ? 2 * tb : 1 + 2 * tb;
      v = w[((size_t(ci) * cout + co) * kh + ky) * kw + kx];
    }
  }
  dst[i] = f2bf(v);
}

// conv bias + eval BatchNorm -> scale | shift (y = acc * scale + shift), each coutp
__global__ void __launch_bounds__(256)
k_bb_pack_epi(const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
              const float* __restrict__ mean, const float* __restrict__ var, float eps, int cout, int coutp,
              float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= coutp) return;
  float sc = 1.0f, sh = 0.0f;
  if (i < cout) {
    const float b = bias ? bias[i] : 0.0f;
    if (gamma) {
      sc = gamma[i] / sqrtf(var[i] + eps);
      sh = beta[i] + (b - mean[i]) * sc;
    } else {
      sh = b;
    }
  } else {
    sc = 0.0f;
  }
  dst[i] = sc;
  dst[coutp + i] = sh;
}

}  // namespace fvp

using namespace fvp;

extern "C" int fvp_bb_input(const float* images, uint16_t* nhwc8, int N, int C, int H, int W, fvp_stream_t s) {
  FVP_REQUIRE(images && nhwc8 && N >= 0 && C >= 1 && C <= 8 && H > 0 && W > 0);
  if (N == 0) return 0;
  FVP_REQUIRE(W % 2 == 0 && C <= 4);
  const long n = long(N) * H * W / 2;
  hipLaunchKernelGGL(k_bb_input, dim3(unsigned((n + 255) / 256)), dim3(256), 0, as_stream(s), images, nhwc8, N, C, H, W);
  return launch_status();
}

extern "C" int fvp_bb_pack(const float* weight, const float* bias, const float* bn_gamma, const float* bn_beta,
                           const float* bn_mean, const float* bn_var, float eps, const FvpBbOp* op, uint16_t* wblob,
                           float* eblob, fvp_stream_t s) {
  FVP_REQUIRE(weight && op && wblob && eblob && (!bn_gamma || (bn_beta && bn_mean && bn_var)));
  FVP_REQUIRE(op->kind == FVP_BB_CONV || op->kind == FVP_BB_DECONV);
  const int tr = op->kind == FVP_BB_DECONV;
  FVP_REQUIRE(!tr || (op->kh == 4 && op->kw == 4 && op->stride == 2 && op->pad == 1));
  if (op->flags & FVP_BB_STEM) {
    FVP_REQUIRE(!tr && op->kh == 7 && op->kw == 7 && op->stride == 2 && op->pad == 3 && op->cin <= 4 && op->cinp == 8);
    hipLaunchKernelGGL(k_bb_pack_stem, dim3(ceil_div(op->coutp * 224, 256)), dim3(256), 0, as_stream(s), weight, op->cin,
                       op->cout, op->coutp, wblob + op->w_off);
    hipLaunchKernelGGL(k_bb_pack_epi, dim3(ceil_div(op->coutp, 256)), dim3(256), 0, as_stream(s), bias, bn_gamma, bn_beta,
                       bn_mean, bn_var, eps, op->cout, op->coutp, eblob + op->e_off);
    return launch_status();
  }
  const int ntaps = tr ? 4 : op->kh * op->kw;
  const long total = long(tr ? 4 : 1) * op->coutp * ntaps * op->cinp;
  hipLaunchKernelGGL(k_bb_pack_w, dim3(unsigned((total + 255) / 256)), dim3(256), 0, as_stream(s), weight, tr, op->cin,
                     op->cout, op->cinp, op->coutp, op->kh, op->kw, wblob + op->w_off);
  hipLaunchKernelGGL(k_bb_pack_epi, dim3(ceil_div(op->coutp, 256)), dim3(256), 0, as_stream(s), bias, bn_gamma, bn_beta,
                     bn_mean, bn_var, eps, op->cout, op->coutp, eblob + op->e_off);
  return launch_status();
}

template <int BN>
static int bb_launch(const BbConvArgs& a, dim3 grid, hipStream_t s) {
  constexpr size_t lds_ab = (FVP_BB_NBUF * 128 * 72 + FVP_BB_NBUF * BN * 72) * sizeof(uint16_t) + 128;
  constexpr size_t lds_ep = 4 * 64 * 36 * sizeof(float);               // epilogue tiles reuse the same memory
  constexpr size_t lds = lds_ab > lds_ep ? lds_ab : lds_ep;
  static LdsOptIn optin;
  auto k = &k_bb_conv<BN>;
  if (int e = lds_opt_in(optin, reinterpret_cast<const void*>(k), lds)) return e;
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  return launch_status();
}

static int bb_launch_conv(const FvpBbOp& op, BbConvArgs a, hipStream_t s, const uint16_t* zeros) {
  const int M = a.N * a.OH * a.OW;
  static const bool no_big = getenv("FVP_BB_NO_BIG") != nullptr;
  static const long big_min = getenv("FVP_BB_BIG_MIN_TILES") ? atol(getenv("FVP_BB_BIG_MIN_TILES")) : 100;   // (tests: 1)
  // large tiles whenever the layer has >= 256 couts (measured faster than the 128 x 128 kernel down to ~150
  // workgroups: 685 vs 455 TFLOP/s on the 512->512 3x3 at 16x30) and the output is plain bf16 NHWC
  if (!no_big && op.coutp % 256 == 0 && long(ceil_div(M, 256)) * (op.coutp / 256) * a.ncls >= big_min && a.out && !a.out_cl &&
      !a.out_nchw && (a.Cbuf & 7) == 0 && size_t(op.coutp) * a.K < (1u << 30)) {
    constexpr size_t lds = 2 * (256 + 256) * 72 * sizeof(uint16_t) + 128;
    static LdsOptIn optin;
    if (int e = lds_opt_in(optin, reinterpret_cast<const void*>(&k_bb_conv_big), lds)) return e;
    hipLaunchKernelGGL(k_bb_conv_big, dim3(ceil_div(M, 256), op.coutp / 256, a.ncls), dim3(512), lds, s, a, zeros);
    return launch_status();
  }
  const bool wide = op.coutp % 128 == 0;
  dim3 grid(ceil_div(M, 128), op.coutp / (wide ? 128 : 64), a.ncls);
  return wide ? bb_launch<128>(a, grid, s) : bb_launch<64>(a, grid, s);
}

extern "C" int fvp_bb_run(const FvpBbOp* ops, int nops, const uint16_t* wblob, const float* eblob, void* const* bufs,
                          int nbufs, int N, float* heat_cl, int heat_jp, float* heat_nchw, fvp_stream_t s) {
  FVP_REQUIRE(ops && wblob && eblob && bufs && nops >= 0 && N >= 0);
  if (N == 0) return 0;
  double flops = 0.0;
  for (int i = 0; i < nops; ++i)
    if (ops[i].kind != FVP_BB_MAXPOOL)
      flops += 2.0 * ops[i].cin * ops[i].cout * (ops[i].kind == FVP_BB_DECONV ? 4.0 : double(ops[i].kh * ops[i].kw)) *
               ops[i].oh * ops[i].ow * N;
  ProfScope ps(FVP_K_BACKBONE, as_stream(s), flops, nops);
  for (int i = 0; i < nops; ++i) {
    const FvpBbOp& op = ops[i];
    FVP_REQUIRE(op.src >= 0 && op.src < nbufs && op.dst < nbufs && op.res < nbufs);
    if (op.kind == FVP_BB_MAXPOOL) {
      const long n = long(N) * op.oh * op.ow * (op.cinp / 8);
      hipLaunchKernelGGL(k_bb_maxpool, dim3(unsigned((n + 255) / 256)), dim3(256), 0, as_stream(s),
                         (const uint16_t*)bufs[op.src], (uint16_t*)bufs[op.dst], N, op.h, op.w, op.cinp, op.oh, op.ow);
      if (int rc = launch_status()) return rc;
      continue;
    }
    FVP_REQUIRE(op.kind == FVP_BB_CONV || op.kind == FVP_BB_DECONV);
    FVP_LIMIT(op.cinp >= 8 && (op.cinp & (op.cinp - 1)) == 0 && op.coutp % 64 == 0);
    BbConvArgs a{};
    a.in = (const uint16_t*)bufs[op.src];
    a.out = op.dst >= 0 ? (uint16_t*)bufs[op.dst] : nullptr;
    a.res = op.res >= 0 ? (const uint16_t*)bufs[op.res] : nullptr;
    a.epi = eblob + op.e_off;
    a.N = N;
    a.H = op.h;
    a.W = op.w;
    a.Cinp = op.cinp;
    a.cin_log2 = __builtin_ctz(unsigned(op.cinp));
    a.Cout = op.cout;
    a.Coutp = op.coutp;
    a.Cbuf = op.cbuf;
    a.ROH = op.oh;
    a.ROW = op.ow;
    a.relu = (op.flags & FVP_EPI_RELU) ? 1 : 0;
    if (op.flags & FVP_BB_OUT_HEAT) {
      FVP_REQUIRE((heat_cl && heat_jp >= op.cout) || heat_nchw);
      a.out_cl = heat_cl;
      a.out_jp = heat_jp;
      a.out_nchw = heat_nchw;
    }
    if (op.kind == FVP_BB_CONV) {
      FVP_LIMIT(op.kh * op.kw <= 64);
      a.OH = op.oh;
      a.OW = op.ow;
      a.stride = a.stride_x = op.stride;
      a.os = 1;
      a.ntaps = op.kh * op.kw;
      for (int t = 0; t < a.ntaps; ++t) {
        a.dy[t] = (signed char)(t / op.kw - op.pad);
        a.dx[t] = (signed char)(t % op.kw - op.pad);
      }
      if (op.flags & FVP_BB_STEM) {                     // pixel-pair form: input [H][W/2] x 8 channels, 7 x 4 taps
        a.W = op.w / 2;
        a.stride_x = 1;
        a.ntaps = 28;
        for (int t = 0; t < 28; ++t) {
          a.dy[t] = (signed char)(t / 4 - 3);
          a.dx[t] = (signed char)(t % 4 - 2);
        }
      }
      a.K = a.ntaps * op.cinp;
      a.ncls = 1;
      a.w = wblob + op.w_off;
      if (int rc = bb_launch_conv(op, a, as_stream(s), reinterpret_cast<const uint16_t*>(eblob))) return rc;
    } else {
      // ConvTranspose(k4, s2, p1): output (2y + py, 2x + px) gathers input rows y + dy: py = 0 -> (ky 1, dy 0),
      // (ky 3, dy -1); py = 1 -> (ky 0, dy +1), (ky 2, dy 0)
      a.OH = op.h;
      a.OW = op.w;
      a.stride = a.stride_x = 1;
      a.os = 2;
      a.ntaps = 4;
      a.K = 4 * op.cinp;
      a.ncls = 4;                                       // one launch, the parity class on blockIdx.z
      for (int cls = 0; cls < 4; ++cls) {
        const int py = cls >> 1, px = cls & 1;
        for (int t = 0; t < 4; ++t) {
          const int ta = t >> 1, tb = t & 1;
          a.dy[cls * 4 + t] = (signed char)(py ? (ta ? 0 : 1) : (ta ? -1 : 0));
          a.dx[cls * 4 + t] = (signed char)(px ? (tb ? 0 : 1) : (tb ? -1 : 0));
        }
      }
      a.w = wblob + op.w_off;
      if (int rc = bb_launch_conv(op, a, as_stream(s), reinterpret_cast<const uint16_t*>(eblob))) return rc;
    }
  }
  return 0;
}
