// Register-direct conv for the layers with a SHORT reduction and no halo: the 1x1 skip / output convs and
// ConvTranspose(k2, s2) of cnns_2d.py:40-72,74-112 (reference: lib/models/cnns_2d.py Res.skip_con :52-58, Upsample :62-72).
//
// Same implicit GEMM and the same v_mfma_f32_32x32x2_f32 chain as k_conv_dma (A = weights, B = pixels, channels in
// ascending order from a zero accumulator: results are bit-identical to it), but nothing of the activations goes through
// LDS: a wave owns 32 consecutive pixels of one plane and loads its B operands - X[k][pixel], k = 2 s + (lane >> 5) -
// straight from the NCHW map into registers (128-byte rows per half wave), all K / 2 of them before the first MFMA.
// The weights of the whole layer sit in LDS once per workgroup (persistent workgroups walk the tile list), so the main
// loop has no barrier, no DMA bookkeeping and no staging arithmetic: per MFMA one half of an 8-byte LDS read.
// The next tile's operands are requested before this tile's epilogue (and the first tile's before the weights are staged),
// the small instances also fetch a tile's residual values before its MFMA loop.
// Measured on the MI355X, B = 8 (240 planes), k_conv_dma -> this kernel: 1x1 skips 40.6 / 29.1-32.6 / 25.1 -> 34.3 / 23.1 /
// 19.3-21 us, transposed 128 -> 64: 72.9 -> 61 us, transposed 64 -> 32 + fused 1x1 head: 125 -> 76-80 us (the head as MFMAs
// instead of a 32-step fma chain with lane hops).  What is left (128 -> 64: loads 10 + MFMA 29-33 + epilogue 21 us, measured
// one phase at a time) still ADDS UP: one, two or three waves per SIMD, distinct issue priorities per wave slot, a staggered
// start of every other workgroup and a cout split over two workgroups all measured within +-3 us of each other.
//
//   Wl[q][h][n] = float2{ W[4 q + h][n], W[4 q + 2 + h][n] }   (h = lane >> 5): one ds_read_b64 feeds MFMA steps 2 q and
//   2 q + 1; a half wave reads 256 contiguous bytes: conflict-free without padding.
//
// MODE 0: 1x1 conv.  MODE 1: transposed conv, blockIdx.z = output row parity, accumulator blocks cb / cb + NB/2 = output
// columns 2x / 2x + 1 (float2 stores, 256-byte runs per half wave).  MODE 2: MODE 1 with 32 couts plus the 1x1 conv that
// consumes them (P2PNet's output layer) as a second MFMA chain: the finished 32 channels of a pixel sit in one lane pair
// (channel (r & 3) + 8 (r >> 2) + 4 (lane >> 5) in accumulator register r); one v_permlane32_swap per register pair turns
// them into B operands for channel pairs (2 t, 2 t + 1), consumed in ascending t: the chain of the standalone 1x1 kernel.
#pragma once

// prefetch policy (bit 0: the next tile's B operands before this tile's epilogue, bit 1: this tile's residual values before
// its MFMA loop) and waves per SIMD, for the instances with >= 256 / < 256 accumulator-plus-operand registers
#ifndef FVP_REG_PF_BIG
#define FVP_REG_PF_BIG 1
#endif
#ifndef FVP_REG_PF_SMALL
#define FVP_REG_PF_SMALL 3
#endif
#ifndef FVP_REG_OCC_SMALL
#define FVP_REG_OCC_SMALL 3
#endif

namespace fvp {

template <int K, int NB, int MODE, bool HAS_RES>
__global__ void __launch_bounds__(256, (K * NB <= 128 ? FVP_REG_OCC_SMALL : 2)) k_conv_reg(ConvArgs a) {
  constexpr int PF = (K * NB >= 256) ? FVP_REG_PF_BIG : FVP_REG_PF_SMALL;
  constexpr bool PFB = PF & 1, PFR = HAS_RES && (PF & 2);
  HIP_DYNAMIC_SHARED(float, smem)
  static_assert(K % 4 == 0 && K >= 4, "channel quads");
  static_assert(MODE == 0 || NB % 2 == 0, "the transposed conv keeps both column taps of a cout block");
  static_assert(MODE != 2 || NB == 2, "the fused 1x1 conv needs all 32 couts of a pixel in one lane pair");
  constexpr int NT = 32 * NB;                         // floats per packed weight row
  constexpr int KQ = K / 4;
  constexpr int NR = MODE ? NB / 2 : NB;              // residual / output register blocks of a tile
  const int t = threadIdx.x, lane = t & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int dy = MODE ? int(blockIdx.z) : 0;
  const int co0 = blockIdx.y * (MODE ? NT / 2 : NT);  // first cout of this workgroup
  const int W = a.W, HW = a.H * W;
  const int tpp = HW >> 5;                            // tiles per plane (host: HW % 32 == 0)
  const int ntiles = a.planes * tpp;
  const int stride = gridDim.x * 4;

  // tiles of masked planes are skipped (k_conv_dma: whole workgroups return); the tile index stays wave-uniform for the
  // compiler too (the flag comes back from a vector load), or every descriptor below would be built in a waterfall loop
  auto next_valid = [&](int tl) {
    if (a.plane_valid)
      while (tl < ntiles && !__builtin_amdgcn_readfirstlane(int(a.plane_valid[fdiv(fdiv(tl, a.m_tpp), a.m_vd)]))) tl += stride;
    return __builtin_amdgcn_readfirstlane(tl);
  };
  // Every global access of the loop is raw-buffer addressed (round 5): descriptor of the tile's first row (4 SGPRs), ONE
  // per-lane byte offset for all K / 2 loads resp. all rows of the tile, one scalar byte offset per row (a multiple of the
  // row stride, recomputed where it is used: FVP_OPAQUE keeps hipcc from hoisting 64 of them out of the tile loop).
  // Round 4 had a 64-bit scalar pointer per row: an SGPR pair per row in flight, 143-628 SGPR spills per instance.
  const unsigned voff_in = unsigned(half * HW + l31) * 4u;
  float b[K / 2];
#pragma unroll
  for (int s = 0; s < K / 2; ++s) b[s] = 0.0f;           // (only read un-loaded under the diagnostics ablation)
  auto load_b = [&](int tl) {
    const int pl = fdiv(tl, a.m_tpp);
    const fvp_rsrc rs = make_rsrc(a.src + size_t(pl) * K * HW + (tl - pl * tpp) * 32, 0x7ffffff0u);
    unsigned hw8 = unsigned(HW) * 8u;                   // two channels further
    FVP_OPAQUE(hw8);
#pragma unroll
    for (int s = 0; s < K / 2; ++s) b[s] = buf_load_f32(rs, voff_in, unsigned(s) * hw8);
  };
  // the first tile's operands are on their way while the weights are staged
  int tile = next_valid(blockIdx.x * 4 + wave);
  if (PFB && tile < ntiles && !(a.ablate & 1)) load_b(tile);

  float2* Wl = reinterpret_cast<float2*>(smem);
  float* epi_s = smem + K * NT;
  float2* W2l = reinterpret_cast<float2*>(epi_s + 3 * a.coutp);
  float* epi2_s = epi_s + 3 * a.coutp + 32 * 32;
  {
    const float* wts = a.wts + size_t(dy) * K * a.wrow;
    constexpr int NQ4 = NT / 4;
#pragma unroll 4
    for (int i = t; i < ((a.ablate & 2) ? 0 : KQ * 2 * NQ4); i += 256) {
      const int n4 = i % NQ4, qh = i / NQ4, h = qh & 1, q = qh >> 1;
      // column of packed row element n: this workgroup's cout block (blockIdx.y) of the 1x1 conv, or of each column tap
      const int n = 4 * n4, col = MODE ? (n / (NT / 2)) * a.coutp + co0 + n % (NT / 2) : co0 + n;
      const float4 u = *reinterpret_cast<const float4*>(wts + (4 * q + h) * a.wrow + col);
      const float4 v = *reinterpret_cast<const float4*>(wts + (4 * q + 2 + h) * a.wrow + col);
      float4* d = reinterpret_cast<float4*>(Wl + qh * NT + 4 * n4);
      d[0] = make_float4(u.x, v.x, u.y, v.y);
      d[1] = make_float4(u.z, v.z, u.w, v.w);
    }
    for (int i = t; i < 3 * a.coutp; i += 256) epi_s[i] = a.epi[i];
    if (MODE == 2) {
      for (int i = t; i < 8 * 2 * 32; i += 256) {
        const int n = i & 31, qh = i >> 5, h = qh & 1, q = qh >> 1;
        W2l[i] = make_float2(a.w2[(4 * q + h) * 32 + n], a.w2[(4 * q + 2 + h) * 32 + n]);
      }
      if (t < 96) epi2_s[t] = a.epi2[t];
    }
  }
  __syncthreads();

  const float* bias = epi_s;
  const float* scale = epi_s + a.coutp;
  const float* shift = epi_s + 2 * a.coutp;
  const bool relu = a.flags & FVP_EPI_RELU;
  const bool res_after = a.flags & FVP_EPI_RES_AFTER_RELU;
  const float2* wl = Wl + half * NT + l31;
  const int OHW = a.OH * a.OW;

  while (tile < ntiles) {
    const int plane = fdiv(tile, a.m_tpp);
    const int px = (tile - plane * tpp) * 32 + l31;
    // per-lane part of an output address (bytes): the pixel, and 4 rows further for the upper half wave; row (r, nb) adds
    // the scalar (nb * 32 + (r & 3) + 8 (r >> 2)) * ostep4.  The output / residual descriptors start at cout co0 of the
    // plane and end behind its last cout: padded rows fail the range check (loads return 0, stores are dropped).
    unsigned ostep4, voff;
    if (MODE == 0) {
      ostep4 = unsigned(HW) * 4u;
      voff = unsigned(px) * 4u + 4u * unsigned(half) * ostep4;
    } else {
      const int y = fdiv(px, a.m_w), x = px - y * W;
      ostep4 = unsigned(OHW) * 4u;
      voff = unsigned((2 * y + dy) * a.OW + 2 * x) * 4u + 4u * unsigned(half) * ostep4;
    }
    FVP_OPAQUE(ostep4);
    const size_t pbase = (size_t(plane) * a.cout + co0) * (ostep4 >> 2);
    const unsigned obytes = a.cout > co0 ? unsigned(a.cout - co0) * ostep4 : 0u;
    const fvp_rsrc rd = make_rsrc(a.dst + pbase, obytes);
    typedef typename std::conditional<MODE == 0, float, float2>::type res_t;
    res_t rv[PFR ? NR : 1][16];
    auto load_res = [&](int nb, res_t (&dst)[16]) {
      const fvp_rsrc rr = make_rsrc(a.res + pbase, obytes);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned so = unsigned(nb * 32 + (r & 3) + 8 * (r >> 2)) * ostep4;
        if constexpr (MODE == 0) dst[r] = buf_load_f32(rr, voff, so);
        else dst[r] = buf_load_f32x2(rr, voff, so);
      }
    };
    if (PFR) {
#pragma unroll
      for (int nb = 0; nb < NR; ++nb) load_res(nb, rv[nb]);
    }
    if (!PFB && !(a.ablate & 1)) load_b(tile);

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    // two operand sets: the LDS reads of quad q + 1 are issued before the 2 NB MFMAs of quad q (see k_conv_dma)
    float2 av[2][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) av[0][nb] = wl[nb * 32];
    if (!(a.ablate & 4))
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int cur = q & 1;
      __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0), vmcnt untouched
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < KQ) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) av[cur ^ 1][nb] = wl[(q + 1) * 2 * NT + nb * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][nb].x, b[2 * q], acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][nb].y, b[2 * q + 1], acc[nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next tile's operands travel during this tile's epilogue (and the other waves' MFMA loops)
    const int nxt = next_valid(tile + stride);
    if (PFB && nxt < ntiles && !(a.ablate & 1)) load_b(nxt);
    __builtin_amdgcn_sched_barrier(0);
    if (a.ablate & 8) {
      tile = nxt;
      continue;
    }

    if constexpr (MODE == 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        res_t rl[16];
        if (HAS_RES && !PFR) load_res(nb, rl);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;     // < coutp: epi vectors are padded
          float v = bn_affine(acc[nb][r], bias[co], scale[co], shift[co]);
          const float rr = HAS_RES ? (PFR ? rv[PFR ? nb : 0][r] : rl[r]) : 0.f;
          if (HAS_RES && !res_after) v += rr;
          if (relu) v = fmaxf(v, 0.0f);
          if (HAS_RES && res_after) v += rr;
          buf_store_f32(v, rd, voff, unsigned(nb * 32 + (r & 3) + 8 * (r >> 2)) * ostep4);
        }
      }
    } else {
      constexpr int CH = NB / 2;
#pragma unroll
      for (int cb = 0; cb < CH; ++cb) {
        res_t rl[16];
        if (HAS_RES && !PFR) load_res(cb, rl);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          float v[2] = {acc[cb][r], acc[cb + CH][r]};
          const res_t r2 = HAS_RES ? (PFR ? rv[PFR ? cb : 0][r] : rl[r]) : res_t{};
          const float rr[2] = {r2.x, r2.y};
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float xv = bn_affine(v[e], bias[co], scale[co], shift[co]);
            if (HAS_RES && !res_after) xv += rr[e];
            if (relu) xv = fmaxf(xv, 0.0f);
            if (HAS_RES && res_after) xv += rr[e];
            v[e] = xv;
          }
          if (MODE == 2) {
            acc[cb][r] = v[0];                         // B operands of the fused 1x1 conv
            acc[cb + CH][r] = v[1];
          } else {
            buf_store_f32x2(make_float2(v[0], v[1]), rd, voff, unsigned(cb * 32 + (r & 3) + 8 * (r >> 2)) * ostep4);
          }
        }
      }
      if constexpr (MODE == 2) {
        // registers (4 g, 4 g + 1) and (4 g + 2, 4 g + 3): upper half of the first <-> lower half of the second.  Afterwards
        // register 4 g + {0, 2, 1, 3} holds channels (2 t, 2 t + 1) in its (lower, upper) half for t = 4 g + {0, 1, 2, 3}.
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const auto sw = __builtin_amdgcn_permlane32_swap(unsigned(__float_as_int(acc[e][r])),
                                                             unsigned(__float_as_int(acc[e][r + 1])), false, false);
            acc[e][r] = __int_as_float(int(sw[0]));
            acc[e][r + 1] = __int_as_float(int(sw[1]));
          }
        f32x16 h0, h1;
#pragma unroll
        for (int r = 0; r < 16; ++r) h0[r] = h1[r] = 0.0f;
        const float2* w2 = W2l + half * 32 + l31;
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) {
          const int g = tt >> 2, si = tt & 3;
          const int r = 4 * g + ((si & 1) << 1 | (si >> 1));
          const float2 a2 = w2[(tt >> 1) * 64];
          const float aw = (tt & 1) ? a2.y : a2.x;
          h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, acc[0][r], h0, 0, 0, 0);
          h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, acc[1][r], h1, 0, 0, 0);
        }
        const float* bias2 = epi2_s;
        const float* scale2 = epi2_s + 32;
        const float* shift2 = epi2_s + 64;
        const bool relu2 = a.flags2 & FVP_EPI_RELU;
        const fvp_rsrc rd2 = make_rsrc(a.dst2 + size_t(plane) * a.cout2 * (ostep4 >> 2), unsigned(a.cout2) * ostep4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
          float x0 = bn_affine(h0[r], bias2[j], scale2[j], shift2[j]), x1 = bn_affine(h1[r], bias2[j], scale2[j], shift2[j]);
          if (relu2) {
            x0 = fmaxf(x0, 0.0f);
            x1 = fmaxf(x1, 0.0f);
          }
          buf_store_f32x2(make_float2(x0, x1), rd2, voff, unsigned((r & 3) + 8 * (r >> 2)) * ostep4);
        }
      }
    }
    tile = nxt;
  }
}

}  // namespace fvp
