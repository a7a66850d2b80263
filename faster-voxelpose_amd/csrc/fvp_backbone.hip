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
// Two kernels share that formulation:
//   k_bb_conv<BN>      4 waves (2 x 2), tile 128 pixels x 128 couts (64 for the 64-channel layers), k chunks of 64
//                      through one LDS buffer (37 KB: four workgroups per CU) with a 144-byte row pitch (conflict-free
//                      b128 reads); the next chunk's global loads (branch-free, address-clamped) are in flight in
//                      registers while the current one is multiplied.  Stem, 64-cout layers, unfused heatmap layer.
//   k_bb_conv_dma<BN>  8 waves, 256 pixels x 256 / 128 couts, both operand tiles through the LDS-DMA into two
//                      swizzled slots, per-lane addressing hoisted out of the k loop (every layer with >= 64 stored
//                      input channels and >= 128 couts; comment at the kernel).
// A per-launch tap table (dy, dx) covers strided convs and the four parity classes of ConvTranspose(k4, s2, p1) with
// the same kernels; the heatmap layer writes fp32 heatmaps directly in the channels-last layout the projection
// kernels read (and / or NCHW, the reference's layout) - from the epilogue of the last transposed conv when fused.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "fvp_common.h"

// ring slots of the LDS-DMA kernel's 128-cout configurations (chunks in flight = slots - 1).  Measured on the MI355X
// (40 images, round 3, with the plain chunk barrier): 32-wide chunks with three slots (72 KB, inside the epilogue's 73.7 KB)
// K-heavy 1x1 layers -5 %, backbone pass 9.04 -> 8.93 ms - but with two batches in flight the images -> joints leg of
// bench.py then came out bimodal (361-440 instead of 690 frames/s in 3 of 7 fresh processes, the serial rate unchanged;
// cause not established), so both configurations stay at two slots; four slots (96 KB: one workgroup per CU) 9.35 ms;
// 64-wide chunks with three slots (144 KB) cost the big transposed conv 7 %.
#ifndef FVP_BB_SLOTS_128_64
#define FVP_BB_SLOTS_128_64 2
#endif
#ifndef FVP_BB_SLOTS_128_32
#define FVP_BB_SLOTS_128_32 2
#endif
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
  int toff[32];           // k_bb_conv_dma: element offset (dy * W + dx) * Cinp of tap t (class-major for transposed convs)
  // k_bb_conv_dma<.., FUSE>: the 1x1 heatmap layer applied to this layer's output tile (which is then not stored)
  const uint16_t* w2;     // [Coutp2 >= 32][K2] packed weights of the heatmap layer, K2 = this layer's Coutp
  const float* epi2;      // its scale | shift, each Coutp2
  int K2, Cout2, Coutp2;
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

// Large-tile variant (every layer whose input has >= 64 stored channels and whose output is plain bf16 NHWC):
// 8 waves, 256 pixels x BN couts per workgroup (BN = 256 / 128 / 64), 64 x BN/2 per wave, so a k step needs
// 2 + BN/64 ds_read_b128 for BN/16 MFMAs -- the 128 x 128 kernel above needs one read per MFMA and is bound by
// LDS read bandwidth.  Both operand tiles are copied by the LDS-DMA (global_load_lds, 16 B per lane, no staging
// registers) into two slots: chunk c+1 streams in while chunk c is multiplied.
//   * LDS rows are 128 bytes (one 64-wide k chunk) without padding; the 16-byte k group g of row r is stored at
//     position g ^ ((r >> 1) & 7), which makes the MFMA operand fetches conflict-free: a ds_read_b128 is served in
//     the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32), and within each the 8 even and the 8 odd rows
//     must land on 8 different positions (r & 7 does not do that: rows 12 and 20 collide, measured as 48 % of the
//     LDS cycles).  The DMA writes lane-contiguous LDS, so lane (row, position p) reads source group p ^ swz(row).
//   * a k chunk lies inside one tap (Cinp % 64 == 0), so everything per-lane is hoisted out of the k loop: the
//     lane's source pointers at tap (0, 0), one validity bit per tap, the weight row pointers.  Per chunk a lane
//     adds one wave-uniform offset and selects the zero page for invalid taps: ~6 VALU instructions per 16-byte
//     item instead of ~40 (the address arithmetic used to cost as much issue time as the MFMAs).
//   * workgroup id -> (pixel tile, cout tile) keeps the cout tiles of one pixel tile on one XCD, back to back:
//     the activation tile is fetched into that XCD's L2 once.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {          // v_cvt_pk_bf16_f32 (RNE)
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

template <int BN, int BK, int NSLOT, bool FUSE = false>
__global__ void __launch_bounds__(512, BN == 128 ? 4 : 2) k_bb_conv_dma(BbConvArgs a, const uint16_t* __restrict__ zeros) {
  constexpr int BM = 256, ROWB = 2 * BK;                             // LDS row = one k chunk of a pixel / cout: 128 or 64 bytes
  constexpr int GPR = BK / 8;                                        // 16-byte k groups per row
  constexpr int SLOTB = (BM + BN) * ROWB;                            // bytes per slot
  constexpr int NRA = BM * GPR / 512, NRB = BN * GPR / 512;          // DMA rounds (512 lanes x 16 B) per operand tile
  constexpr int LPC = NRA + NRB;                                     // DMA instructions per wave and chunk
  constexpr int D = NSLOT - 1;                                       // chunks in flight ahead of the one being multiplied
  constexpr int WN = BN / 2, NJ = WN / 32;                           // couts / MFMA tiles per wave
  static_assert(NRA >= 1 && NRB >= 1 && (BK == 64 || BK == 32), "tile shape");
  HIP_DYNAMIC_SHARED(uint16_t, smem16)
  char* smem = reinterpret_cast<char*>(smem16);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = a.N * a.OH * a.OW;
  // workgroup id -> XCD (id & 7) -> that XCD's pixel tiles, cout tiles innermost
  const int nct = a.Coutp / BN, nin = nct * a.ncls;    // (class, cout tile) innermost: they share the activation tile
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int inner = seq % nin;
  const int mt = (seq / nin) * 8 + xcd, ct = inner % nct;
  if (mt * BM >= M) return;
  const int m0 = mt * BM, co0 = ct * BN;
  const int cls = inner / nct;                         // parity class of a transposed conv (0 otherwise)
  const int opy = a.ncls > 1 ? cls >> 1 : a.py, opx = a.ncls > 1 ? (cls & 1) : a.px;
  const uint16_t* wcls = a.w + size_t(cls) * a.Coutp * a.K;
  const int tap0 = cls * a.ntaps;
  // the 16-byte k group g of row r sits at position g ^ swz(r): conflict-free operand fetches
  auto swz = [](int r) { return BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); };

  // ---- this lane's DMA items (fixed over the k loop)
  const uint16_t* ap[NRA];                            // source of this item's k group at tap (0, 0), channel 0
  uint32_t amask[NRA];                                // bit t: tap t of this pixel lies inside the image
  const uint16_t* bp[NRB];
#pragma unroll
  for (int j = 0; j < NRA; ++j) {
    const int it = (wave + 8 * j) * 64 + lane;
    const int row = it / GPR, g = (it % GPR) ^ swz(row);
    const int am = m0 + row;
    const bool ok = am < M;
    int n, oy, ox;
    bb_decode(ok ? am : 0, a.OW, a.OH * a.OW, n, oy, ox);
    const int iy0 = oy * a.stride, ix0 = ox * a.stride_x;
    ap[j] = a.in + ((size_t(n) * a.H + iy0) * a.W + ix0) * a.Cinp + g * 8;
    uint32_t mk = 0;
    for (int tp = 0; tp < a.ntaps; ++tp) {
      const int iy = iy0 + a.dy[(tap0 + tp) & 63], ix = ix0 + a.dx[(tap0 + tp) & 63];
      if (ok && unsigned(iy) < unsigned(a.H) && unsigned(ix) < unsigned(a.W)) mk |= 1u << tp;
    }
    amask[j] = mk;
  }
#pragma unroll
  for (int j = 0; j < NRB; ++j) {
    const int it = (wave + 8 * j) * 64 + lane;
    const int row = it / GPR, g = (it % GPR) ^ swz(row);
    bp[j] = wcls + size_t(co0 + row) * a.K + g * 8;   // packed weights are padded to Coutp rows
  }
  const int nchunks = a.K / BK;
  // k order: channel block outermost, taps innermost -- consecutive chunks re-read the same activation rows shifted
  // by one tap, while they are still in the XCD's L2 (tap-major order touched them again only Cinp / 64 chunks of
  // 32 CUs x 64 KB later: beyond a 4 MB L2)
  // A 32-wide chunk is one half of a (64-channel block, tap) step, halves innermost: every tile configuration then
  // adds the products of an output in exactly the same order (the per-op choice of fvp_bb_tune never changes a bit).
  int st_tap = 0, st_c0 = 0, st_half = 0;
  auto stage = [&](int chunk) {
    const int tap = st_tap, c0 = st_c0 + st_half * BK;
    if (BK == 64 || ++st_half == 64 / BK) {
      st_half = 0;
      if (++st_tap == a.ntaps) {
        st_tap = 0;
        st_c0 += 64;
      }
    }
    const int kk = tap * a.Cinp + c0;                 // position in the packed weights [cout][tap][cin]
    const int soff = a.toff[(tap0 + tap) & 31] + c0;  // wave-uniform element offset of (tap, c0)
    const uint32_t bit = 1u << tap;
    char* As = smem + (chunk % NSLOT) * SLOTB;
    char* Bs = As + BM * ROWB;
#pragma unroll
    for (int j = 0; j < NRA; ++j) {
      const uint16_t* src = (amask[j] & bit) ? ap[j] + soff : zeros;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(As + (wave + 8 * j) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NRB; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bp[j] + kk),
                                       (__attribute__((address_space(3))) void*)(Bs + (wave + 8 * j) * 1024), 16, 0, 0);
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // operand fetch offsets: row (tile row + l31), k group 2 ks + half
  const int a_rd = (wm * 64 + l31) * ROWB, b_rd = (BM + wn * WN + l31) * ROWB;
  int xo[BK / 16];
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) xo[ks] = ((2 * ks + half) ^ swz(l31)) << 4;

  // ring of NSLOT slots, D = NSLOT - 1 chunks in flight ahead of the MFMAs.  Used with two 64-wide slots: four
  // 32-wide slots (three chunks in flight, same LDS) measured 4-8 % slower -- twice the barriers -- so the DMA
  // stream is not latency-starved; ablations put the k loop at MFMA 45 %, DMA issue / landing 30 %, operand
  // fetches 15 %, barriers 10 % of the time (deconv 256->256 at 64x120)
#pragma unroll
  for (int c = 0; c < D; ++c)
    if (c < nchunks) stage(c);
  for (int c = 0; c < nchunks; ++c) {
    const int ahead = nchunks - 1 - c;                // chunks issued after c (at most D - 1 before this iteration's stage)
    wait_vmcnt((ahead < D - 1 ? ahead : D - 1) * LPC);  // this wave's share of chunk c has landed
    // ... everybody's; and everybody is done reading chunk c - 1.  A plain s_barrier behind this wave's lgkmcnt(0):
    // __syncthreads() is fence + barrier, and the fence drains EVERY outstanding LDS-DMA (vmcnt(0)), i.e. the chunks
    // requested for later iterations too - with it a ring deeper than two slots never had more than one chunk in flight
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    if (c + D < nchunks) stage(c + D);                // into the slot of chunk c - 1
    const char* S = smem + (c % NSLOT) * SLOTB;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      Bf8 fa[2], fb[NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const Bf8*>(S + a_rd + i * 32 * ROWB + xo[ks]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const Bf8*>(S + b_rd + j * 32 * ROWB + xo[ks]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma_bf16(fa[i], fb[j], acc[i][j]);
    }
  }
  __syncthreads();                                    // every wave is done with the slots: the epilogue reuses them

  // ---- epilogue: bf16 NHWC output through wave-private fp32 LDS tiles (32 couts at a time) so that a lane
  // ends up with 8 consecutive channels of one pixel: 16-byte residual loads and 16-byte stores
  const float* scale = a.epi;
  const float* shift = a.epi + a.Coutp;
  constexpr int EP = 32 + 4;
  float* et = reinterpret_cast<float*>(smem) + wave * (64 * EP);
  if constexpr (FUSE) {
    // ---- fused heatmap layer (final_layer, resnet.py:199): heat[pixel][joint] = sum_c bf16(relu(bn(acc)))[pixel][c] *
    // W2[joint][c] over the workgroup's 256 channels (one cout tile: co0 = 0).  The staged 64 x 32 block is read back
    // in the MFMA A layout (row l31, 8 channels per half), rounded to bf16 exactly as the stored activation would be,
    // and multiplied with W2 (B operand: joint l31); the two cout halves (wn) are summed through LDS.
    f32x16 hacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) hacc[i][r] = 0.0f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int cj = wn * WN + j * 32;
      const float sc = scale[cj + l31], sh = shift[cj + l31];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * EP + l31] = acc[i][j][r] * sc + sh;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const Bf8 fb = *reinterpret_cast<const Bf8*>(a.w2 + size_t(l31) * a.K2 + cj + 16 * ks + 8 * half);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float* src = et + (i * 32 + l31) * EP + 16 * ks + 8 * half;
          const float4 lo = *reinterpret_cast<const float4*>(src);
          const float4 hi = *reinterpret_cast<const float4*>(src + 4);
          float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          Bf8 fa;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v0 = x[2 * e], v1 = x[2 * e + 1];
            if (a.relu) {
              v0 = fmaxf(v0, 0.0f);
              v1 = fmaxf(v1, 0.0f);
            }
            fa.w[e] = pack_bf16x2(v0, v1);
          }
          hacc[i] = mfma_bf16(fa, fb, hacc[i]);
        }
      }
    }
    float* red = reinterpret_cast<float*>(smem + 8 * 64 * EP * sizeof(float)) + wm * (64 * 32);   // behind the staging tiles
    if (wn == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = hacc[i][r];
    }
    __syncthreads();
    if (wn == 1) return;
    const float sc2 = a.epi2[l31], sh2 = a.epi2[a.Coutp2 + l31];
    const size_t hw = size_t(a.ROH) * a.ROW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {                    // four consecutive pixel rows per register quad
        const int rb = i * 32 + 8 * q + 4 * half;
        int n_, oy, ox;
        const int mb = m0 + wm * 64 + rb;
        bb_decode(mb < M ? mb : 0, a.OW, a.OH * a.OW, n_, oy, ox);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (mb + e < M) {
            const size_t pix = (size_t(n_) * a.ROH + oy * a.os + opy) * a.ROW + ox * a.os + opx;
            const float v = (hacc[i][4 * q + e] + red[(rb + e) * 32 + l31]) * sc2 + sh2;
            if (a.out_cl && l31 < a.out_jp) a.out_cl[pix * a.out_jp + l31] = l31 < a.Cout2 ? v : 0.0f;
            if (a.out_nchw && l31 < a.Cout2) a.out_nchw[(size_t(n_) * a.Cout2 + l31) * hw + (pix - size_t(n_) * hw)] = v;
          }
          if (++ox == a.OW) {                          // next pixel of the GEMM row walk
            ox = 0;
            if (++oy == a.OH) {
              oy = 0;
              ++n_;
            }
          }
        }
      }
    return;
  }
  size_t pixv[4];                                     // this lane's four (pixel row, channel group) vectors
  bool rowok[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = (lane + 64 * v) >> 2;
    const int m = m0 + wm * 64 + row;
    rowok[v] = m < M;
    const int mm = rowok[v] ? m : 0;
    size_t pix = size_t(mm);
    if (a.os != 1) {
      int n_, oy, ox;
      bb_decode(mm, a.OW, a.OH * a.OW, n_, oy, ox);
      pix = (size_t(n_) * a.ROH + oy * a.os + opy) * a.ROW + ox * a.os + opx;
    }
    pixv[v] = pix * a.Cbuf;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int cj = co0 + wn * WN + j * 32;
    const float sc = scale[cj + l31], sh = shift[cj + l31];
    const int co = cj + (lane & 3) * 8;
    const bool co_ok = co < a.Cbuf;
    Bf8 resv[4];
    if (a.res) {                                      // unconditional, clamped residual loads first
#pragma unroll
      for (int v = 0; v < 4; ++v) resv[v] = *reinterpret_cast<const Bf8*>(a.res + pixv[v] + (co_ok ? co : 0));
    }
    __builtin_amdgcn_wave_barrier();                  // the previous block's readers are done
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * EP + l31] = acc[i][j][r] * sc + sh;
    __builtin_amdgcn_s_waitcnt(0xc07f);               // own LDS writes landed (wave-private tile)
    __builtin_amdgcn_wave_barrier();
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
        o.w[e] = pack_bf16x2(v0, v1);
      }
      if (rowok[v] && co_ok) *reinterpret_cast<Bf8*>(a.out + pixv[v] + co) = o;
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

#include "fvp_backbone_fused.h"

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

static int bb_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    n = cus;
  }
  return n;
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

template <int BN, int BK, int NSLOT, bool FUSE = false>
static int bb_launch_dma(const BbConvArgs& a, int M, hipStream_t s, const uint16_t* zeros) {
  constexpr size_t lds_max = size_t(NSLOT) * (256 + BN) * 2 * BK;
  constexpr size_t lds_ep = 8 * 64 * 36 * sizeof(float) + (FUSE ? 4 * 64 * 32 * sizeof(float) : 0);   // staging (+ wn reduction)
  // a short k loop touches only its first chunks' slots: less LDS = more workgroups per CU for the 1x1 expansions
  const size_t used = size_t(std::min(NSLOT, a.K / BK)) * (256 + BN) * 2 * BK;
  const size_t lds = std::max(used, lds_ep);
  static LdsOptIn optin;
  auto k = &k_bb_conv_dma<BN, BK, NSLOT, FUSE>;
  if (int e = lds_opt_in(optin, reinterpret_cast<const void*>(k), lds_max > lds_ep ? lds_max : lds_ep)) return e;
  const int nmt = ceil_div(M, 256), nct = a.Coutp / BN;
  hipLaunchKernelGGL(k, dim3(unsigned(8 * ceil_div(nmt, 8) * nct * a.ncls)), dim3(512), lds, s, a, zeros);
  return launch_status();
}

// Diagnostic switches (tests flip them between calls): read ONCE per fvp_bb_run call, not per launch - a
// backbone pass is ~60 launches, and getenv racing a setenv from another thread is undefined behaviour.
struct BbSwitches {
  bool no_big;          // FVP_BB_NO_BIG: the register-staged kernel only
  int bn_cap;           // FVP_BB_DMA_BN: 128-cout tiles only
  bool no_fuse_final;   // FVP_BB_NO_FUSE_FINAL: heatmap layer as its own launch
  bool no_fuse_stem;    // FVP_BB_NO_FUSE_STEM: stem conv and max-pool as two launches
  bool no_fuse_block;   // FVP_BB_NO_FUSE_BLOCK: layer1's bottlenecks layer by layer
  int ablate;           // FVP_BB_ABLATE: phase ablations of k_bb_bottleneck64 (wrong results)
  static BbSwitches read() {
    const char* bn = fvp::diag_env("FVP_BB_DMA_BN");
    return {fvp::diag_env("FVP_BB_NO_BIG") != nullptr, bn ? atoi(bn) : 256, fvp::diag_env("FVP_BB_NO_FUSE_FINAL") != nullptr,
            fvp::diag_env("FVP_BB_NO_FUSE_STEM") != nullptr, fvp::diag_env("FVP_BB_NO_FUSE_BLOCK") != nullptr,
            fvp::diag_env("FVP_BB_ABLATE") ? atoi(fvp::diag_env("FVP_BB_ABLATE")) : 0};
  }
};

static int bb_launch_conv(const FvpBbOp& op, BbConvArgs a, hipStream_t s, const uint16_t* zeros, const BbSwitches& sw) {
  const int M = a.N * a.OH * a.OW;
  const bool no_big = sw.no_big;
  const int bn_cap = sw.bn_cap;
  // LDS-DMA kernel: a k chunk of 64 inside one tap, plain bf16 NHWC output, offsets within 32 bits
  if (a.w2) {                                          // heatmap layer fused into this layer (eligibility: fvp_bb_run)
    for (int i = 0; i < a.ntaps * a.ncls; ++i) a.toff[i] = (int(a.dy[i]) * a.W + int(a.dx[i])) * a.Cinp;
    return bb_launch_dma<256, 64, 2, true>(a, M, s, zeros);
  }
  if (!no_big && op.cinp % 64 == 0 && op.coutp % 128 == 0 && a.ntaps * a.ncls <= 32 && a.out && !a.out_cl && !a.out_nchw && (a.Cbuf & 7) == 0 &&
      size_t(op.coutp) * a.K < (1u << 30) && size_t(a.N) * a.H * a.W * a.Cinp < (1u << 30)) {
    for (int i = 0; i < a.ntaps * a.ncls; ++i) a.toff[i] = (int(a.dy[i]) * a.W + int(a.dx[i])) * a.Cinp;
    // 256 couts per workgroup unless that leaves the chip badly filled: W workgroups run in ceil(W / 256) rounds, a
    // 128-wide tile needs ~1.2x the time per output but halves the granule (3x3 256->256 at 32x60: 300 workgroups =
    // two rounds, the second one 17 % full); the single-chunk expansions with a residual are HBM-bound and gain from
    // two resident workgroups per CU (less LDS, 112 registers)
    int bn = 128;
    if (op.coutp % 256 == 0 && bn_cap >= 256) {
      const long W = long(ceil_div(M, 256)) * (op.coutp / 256) * a.ncls;
      const double r256 = double(ceil_div(W, 256L)), r128 = double(ceil_div(2 * W, 256L)) * 0.5 * 1.2;
      const bool small_k = a.K == 64 && a.res;
      if (!(r128 < r256 || small_k)) bn = 256;
    }
    // tile configuration chosen by fvp_bb_tune for this op (bits 8-9 of flags), else the heuristic above.
    // 1: 256 couts, 64-wide chunks; 2: 128 couts, 64-wide chunks; 3: 128 couts, 32-wide chunks (73.7 KB of LDS and
    // <= 128 VGPRs: two workgroups per CU, one's epilogue overlaps the other's k loop).  All three walk k in the same
    // order with the same MFMA: the choice never changes a result bit.
    int cfg = (op.flags >> FVP_BB_CFG_SHIFT) & 3;
    if (cfg == 1 && op.coutp % 256 != 0) cfg = 0;
    if (cfg == 0) cfg = bn == 256 ? 1 : 2;
    return cfg == 1 ? bb_launch_dma<256, 64, 2>(a, M, s, zeros)
                    : cfg == 2 ? bb_launch_dma<128, 64, FVP_BB_SLOTS_128_64>(a, M, s, zeros)
                               : bb_launch_dma<128, 32, FVP_BB_SLOTS_128_32>(a, M, s, zeros);
  }
  const bool wide = op.coutp % 128 == 0;
  dim3 grid(ceil_div(M, 128), op.coutp / (wide ? 128 : 64), a.ncls);
  return wide ? bb_launch<128>(a, grid, s) : bb_launch<64>(a, grid, s);
}

// A 64-plane bottleneck at ops[i..]: [downsample 1x1 (no ReLU),] conv1 1x1 -> 64 + ReLU, conv2 3x3 s1 p1 64 -> 64 + ReLU,
// conv3 1x1 64 -> 256 + residual + ReLU (resnet.py:57-95), all stride 1 on one map, the intermediates read by nobody else.
// Returns the number of ops the fused kernel covers (0 = not this pattern).
static int bb_match_bottleneck64(const FvpBbOp* ops, int nops, int i, int N) {
  auto plain = [](const FvpBbOp& o, int k) {
    return o.kind == FVP_BB_CONV && o.kh == k && o.kw == k && o.stride == 1 && o.pad == (k == 3 ? 1 : 0) && o.dst >= 0 &&
           !(o.flags & (FVP_BB_STEM | FVP_BB_OUT_HEAT)) && o.cin == o.cinp && o.cout == o.coutp && o.cbuf == o.cout;
  };
  int j = i;
  bool ds = false;
  if (j + 3 < nops && plain(ops[j], 1) && !(ops[j].flags & FVP_EPI_RELU) && ops[j].res < 0 && ops[j].cout == 256 && ops[j].cin == 64 &&
      ops[j + 1].src == ops[j].src && ops[j + 3].res == ops[j].dst) {
    ds = true;
    ++j;
  }
  if (j + 2 >= nops) return 0;
  const FvpBbOp &c1 = ops[j], &c2 = ops[j + 1], &c3 = ops[j + 2];
  if (!(plain(c1, 1) && plain(c2, 3) && plain(c3, 1))) return 0;
  if (!((c1.flags & FVP_EPI_RELU) && (c2.flags & FVP_EPI_RELU) && (c3.flags & FVP_EPI_RELU))) return 0;
  if (c1.cout != 64 || c2.cin != 64 || c2.cout != 64 || c3.cin != 64 || c3.cout != 256) return 0;
  if (c1.res >= 0 || c2.res >= 0 || c2.src != c1.dst || c3.src != c2.dst) return 0;
  if (c1.h != c2.h || c1.w != c2.w || c1.h != c3.h || c1.w != c3.w || c1.oh != c1.h || c1.ow != c1.w) return 0;
  if (ds ? (c1.cin != 64 || c3.res != ops[i].dst || ops[i].h != c1.h || ops[i].w != c1.w)
         : (c1.cin != 256 || c3.res != c1.src)) return 0;
  if (c3.dst == c1.src) return 0;                                   // (the kernel reads neighbours' halos of x after writing its tile)
  if (size_t(N) * c1.h * c1.w * 256 >= (size_t(1) << 31)) return 0;
  // intermediates (and the downsample tensor) must have no other reader
  const int n = j + 3 - i;
  for (int k = i + n; k < nops; ++k)
    for (int m = i; m < i + n - 1; ++m)
      if (ops[k].src == ops[m].dst || ops[k].res == ops[m].dst) return 0;
  return n;
}

template <int CIN, bool DS>
static int bb_launch_bottleneck64(const BbBlockArgs& a, hipStream_t s) {
  constexpr size_t lds = bb_block_lds(CIN, DS);
  static LdsOptIn optin;
  auto k = &k_bb_bottleneck64<CIN, DS>;
  if (int e = lds_opt_in(optin, reinterpret_cast<const void*>(k), lds)) return e;
  hipLaunchKernelGGL(k, dim3(unsigned(std::min(a.ntiles, bb_num_cus()))), dim3(kBkThreads), lds, s, a);
  return launch_status();
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
  const BbSwitches sw = BbSwitches::read();
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
    // a whole 64-plane bottleneck (layer1) as one kernel: k_bb_bottleneck64
    if (!sw.no_fuse_block && !sw.no_big) {
      if (const int nb = bb_match_bottleneck64(ops, nops, i, N)) {
        const bool ds = nb == 4;
        const FvpBbOp &c1 = ops[i + (ds ? 1 : 0)], &c2 = ops[i + (ds ? 2 : 1)], &c3 = ops[i + nb - 1];
        for (int k = i; k < i + nb; ++k) FVP_REQUIRE(ops[k].src >= 0 && ops[k].src < nbufs && ops[k].dst < nbufs && ops[k].res < nbufs);
        BbBlockArgs ba{};
        ba.x = (const uint16_t*)bufs[c1.src];
        ba.out = (uint16_t*)bufs[c3.dst];
        ba.w1 = wblob + c1.w_off;
        ba.w2 = wblob + c2.w_off;
        ba.w3 = wblob + c3.w_off;
        ba.wd = ds ? wblob + op.w_off : nullptr;
        ba.e1 = eblob + c1.e_off;
        ba.e2 = eblob + c2.e_off;
        ba.e3 = eblob + c3.e_off;
        ba.ed = ds ? eblob + op.e_off : nullptr;
        ba.N = N;
        ba.H = c1.h;
        ba.W = c1.w;
        ba.tiles_x = ceil_div(c1.w, kBkTW);
        ba.tiles_y = ceil_div(c1.h, kBkTH);
        ba.ntiles = ba.tiles_x * ba.tiles_y * N;
        ba.ablate = int(sw.ablate);
        if (int rc = ds ? bb_launch_bottleneck64<64, true>(ba, as_stream(s)) : bb_launch_bottleneck64<256, false>(ba, as_stream(s))) return rc;
        i += nb - 1;
        continue;
      }
    }
    // stem conv + bn + ReLU + MaxPool2d(3, 2, 1) as one kernel (k_bb_stem_pool): the stem's output - the largest tensor of
    // the network after the deconv head - is neither written nor read.  Eligible when the pooling is the only reader.
    if ((op.flags & FVP_BB_STEM) && (op.flags & FVP_EPI_RELU) && !sw.no_fuse_stem && i + 1 < nops && op.res < 0 && op.dst >= 0 &&
        op.coutp == 64 && op.cout == 64 && op.cinp == 8 && op.w % 2 == 0) {
      const FvpBbOp& nx = ops[i + 1];
      bool only_reader = nx.kind == FVP_BB_MAXPOOL && nx.src == op.dst && nx.cinp == 64 && nx.dst >= 0 && nx.h == op.oh && nx.w == op.ow &&
                         nx.oh == (op.oh + 1) / 2 && nx.ow == (op.ow + 1) / 2;
      for (int k = i + 2; k < nops && only_reader; ++k) only_reader = ops[k].src != op.dst && ops[k].res != op.dst;
      if (only_reader && size_t(N) * op.h * op.w * 4 < (1u << 31)) {
        BbStemArgs sa{};
        sa.in = (const uint16_t*)bufs[op.src];
        sa.out = (uint16_t*)bufs[nx.dst];
        sa.w = wblob + op.w_off;
        sa.epi = eblob + op.e_off;
        sa.N = N;
        sa.H = op.h;
        sa.W2 = op.w / 2;
        sa.CH = op.oh;
        sa.CW = op.ow;
        sa.PH = nx.oh;
        sa.PW = nx.ow;
        sa.tiles_x = ceil_div(nx.ow, kStQ);
        sa.tiles_y = ceil_div(nx.oh, kStR);
        sa.ntiles = sa.tiles_x * sa.tiles_y * N;
        static LdsOptIn optin;
        if (int e = lds_opt_in(optin, reinterpret_cast<const void*>(&k_bb_stem_pool), kStLds)) return e;
        hipLaunchKernelGGL(k_bb_stem_pool, dim3(unsigned(std::min(sa.ntiles, 2 * bb_num_cus()))), dim3(kStThreads), kStLds,
                           as_stream(s), sa);
        if (int rc = launch_status()) return rc;
        ++i;
        continue;
      }
    }
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
    // the 1x1 heatmap layer behind a 256-cout layer is applied in that layer's epilogue: its 256-channel output
    // (the largest activation of the network) is never written or read
    bool fused_heat = false;
    if (i + 1 < nops && !sw.no_fuse_final && !sw.no_big) {
      const FvpBbOp& nx = ops[i + 1];
      if (nx.kind == FVP_BB_CONV && (nx.flags & FVP_BB_OUT_HEAT) && nx.src == op.dst && nx.res < 0 && nx.kh == 1 && nx.kw == 1 &&
          nx.stride == 1 && nx.pad == 0 && nx.cinp == 256 && op.coutp == 256 && op.cout == 256 && op.cinp % 64 == 0 && op.res < 0 &&
          nx.cout <= 32 && nx.coutp >= 32 && (op.kind == FVP_BB_DECONV || op.kh * op.kw <= 9) &&
          size_t(N) * op.h * op.w * op.cinp < (1u << 30) && ((heat_cl && heat_jp >= nx.cout && heat_jp <= 32) || (!heat_cl && heat_nchw))) {
        fused_heat = true;
        a.w2 = wblob + nx.w_off;
        a.epi2 = eblob + nx.e_off;
        a.K2 = nx.cinp;
        a.Cout2 = nx.cout;
        a.Coutp2 = nx.coutp;
        a.out_cl = heat_cl;
        a.out_jp = heat_jp;
        a.out_nchw = heat_nchw;
      }
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
      if (int rc = bb_launch_conv(op, a, as_stream(s), reinterpret_cast<const uint16_t*>(eblob), sw)) return rc;
      if (fused_heat) ++i;
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
      if (int rc = bb_launch_conv(op, a, as_stream(s), reinterpret_cast<const uint16_t*>(eblob), sw)) return rc;
      if (fused_heat) ++i;
    }
  }
  return 0;
}

// Per-op choice of the LDS-DMA kernel's tile configuration by measurement: every conv / transposed conv that the
// kernel serves is run alone with each configuration (one warm-up + three timed launches, HIP events on `s`) and
// the fastest goes into bits 8-9 of its flags.  The activation buffers are only scratch here.  Which tile shape
// wins depends on how the launch fills the 256 CUs and on whether the op is HBM- or MFMA-bound; the candidates
// compute identical bits, so tuning never changes results.
extern "C" int fvp_bb_tune(FvpBbOp* ops, int nops, const uint16_t* wblob, const float* eblob, void* const* bufs, int nbufs,
                           int N, fvp_stream_t s) {
  FVP_REQUIRE(ops && wblob && eblob && bufs && nops >= 0 && N >= 0);
  if (N == 0) return 0;
  hipStream_t st = as_stream(s);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return launch_status() ? launch_status() : FVP_EINVAL;
  int rc = 0;
  for (int i = 0; i < nops && !rc; ++i) {
    FvpBbOp& op = ops[i];
    op.flags &= ~(3 << FVP_BB_CFG_SHIFT);
    if ((op.kind != FVP_BB_CONV && op.kind != FVP_BB_DECONV) || (op.flags & (FVP_BB_OUT_HEAT | FVP_BB_STEM)) || op.dst < 0 ||
        op.cinp % 64 != 0 || op.coutp % 128 != 0)
      continue;
    float best = 0.0f;
    int best_cfg = 0;
    for (int cfg = 1; cfg <= 3 && !rc; ++cfg) {
      if (cfg == 1 && op.coutp % 256 != 0) continue;
      FvpBbOp one = op;
      one.flags |= cfg << FVP_BB_CFG_SHIFT;
      rc = fvp_bb_run(&one, 1, wblob, eblob, bufs, nbufs, N, nullptr, 0, nullptr, s);            // warm-up
      if (rc) break;
      if (hipEventRecord(e0, st) != hipSuccess) rc = FVP_EINVAL;
      for (int r = 0; r < 3 && !rc; ++r) rc = fvp_bb_run(&one, 1, wblob, eblob, bufs, nbufs, N, nullptr, 0, nullptr, s);
      if (hipEventRecord(e1, st) != hipSuccess) rc = FVP_EINVAL;
      if (!rc && hipEventSynchronize(e1) != hipSuccess) rc = FVP_EINVAL;
      if (rc) break;
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = FVP_EINVAL; break; }
      if (best_cfg == 0 || ms < best) {
        best = ms;
        best_cfg = cfg;
      }
    }
    if (!rc) op.flags |= best_cfg << FVP_BB_CFG_SHIFT;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}
