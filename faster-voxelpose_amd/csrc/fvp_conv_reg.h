// Register-direct conv for the layers with a SHORT reduction and no halo: the 1x1 skip / output convs and
// ConvTranspose(k2, s2) of cnns_2d.py:40-72,74-112 (reference: lib/models/cnns_2d.py Res.skip_con :52-58, Upsample :62-72).
//
// Same implicit GEMM and the same v_mfma_f32_32x32x2_f32 chain as k_conv_dma (A = weights, B = pixels, channels in
// ascending order from a zero accumulator: results are bit-identical to it), but nothing of the activations goes through
// LDS: a wave owns 32 consecutive pixels of one plane and loads its B operands - X[k][pixel], k = 2 s + (lane >> 5) -
// straight from the NCHW map into registers (128-byte rows per half wave), all K / 2 of them before the first MFMA.
// The weights of the whole layer sit in LDS once per workgroup (persistent workgroups walk the tile list), so the main
// loop has no barrier, no DMA bookkeeping and no staging arithmetic: per MFMA one half of an 8-byte LDS read.
// k_conv_dma on these layers is three serial phases per tile (stage, MFMA, epilogue: 17 + 27 + 25 us on the 128 -> 64
// transposed conv); here the phases of a SIMD's two or three waves overlap freely.
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

namespace fvp {

template <int K, int NB, int MODE, bool HAS_RES>
__global__ void __launch_bounds__(256, (K * NB <= 128 ? 3 : 2)) k_conv_reg(ConvArgs a) {
  HIP_DYNAMIC_SHARED(float, smem)
  static_assert(K % 4 == 0 && K >= 4, "channel quads");
  static_assert(MODE == 0 || NB % 2 == 0, "the transposed conv keeps both column taps of a cout block");
  static_assert(MODE != 2 || NB == 2, "the fused 1x1 conv needs all 32 couts of a pixel in one lane pair");
  constexpr int NT = 32 * NB;                         // floats per packed weight row
  constexpr int KQ = K / 4;
  const int t = threadIdx.x, lane = t & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int dy = MODE ? int(blockIdx.z) : 0;

  float2* Wl = reinterpret_cast<float2*>(smem);
  float* epi_s = smem + K * NT;
  float2* W2l = reinterpret_cast<float2*>(epi_s + 3 * a.coutp);
  float* epi2_s = epi_s + 3 * a.coutp + 32 * 32;
  {
    const float* wts = a.wts + size_t(dy) * K * a.wrow;
    for (int i = t; i < KQ * 2 * NT; i += 256) {
      const int n = i % NT, qh = i / NT, h = qh & 1, q = qh >> 1;
      Wl[i] = make_float2(wts[(4 * q + h) * a.wrow + n], wts[(4 * q + 2 + h) * a.wrow + n]);
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

  const int W = a.W, HW = a.H * W;
  const int tpp = HW >> 5;                            // tiles per plane (host: HW % 32 == 0)
  const int ntiles = a.planes * tpp;
  const float* bias = epi_s;
  const float* scale = epi_s + a.coutp;
  const float* shift = epi_s + 2 * a.coutp;
  const bool relu = a.flags & FVP_EPI_RELU;
  const bool res_after = a.flags & FVP_EPI_RES_AFTER_RELU;
  const float2* wl = Wl + half * NT + l31;

  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int plane = fdiv(tile, a.m_tpp);
    if (a.plane_valid && !a.plane_valid[plane / a.valid_div]) continue;
    const int px = (tile - plane * tpp) * 32 + l31;
    const float* xp = a.src + (size_t(plane) * K + half) * HW + px;
    float b[K / 2];
#pragma unroll
    for (int s = 0; s < K / 2; ++s) b[s] = xp[size_t(2 * s) * HW];

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    // two operand sets: the LDS reads of quad q + 1 are issued before the 2 NB MFMAs of quad q (see k_conv_dma)
    float2 av[2][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) av[0][nb] = wl[nb * 32];
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

    if constexpr (MODE == 0) {
      const unsigned obase = unsigned(plane) * unsigned(a.cout) * unsigned(HW) + unsigned(px);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float rv[16];
        unsigned o[16];
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          ok[r] = co < a.cout;
          o[r] = obase + unsigned(ok[r] ? co : 0) * unsigned(HW);
          if (HAS_RES) rv[r] = a.res[o[r]];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;     // < coutp: epi vectors are padded
          float v = bn_affine(acc[nb][r], bias[co], scale[co], shift[co]);
          if (HAS_RES && !res_after) v += rv[r];
          if (relu) v = fmaxf(v, 0.0f);
          if (HAS_RES && res_after) v += rv[r];
          if (ok[r]) a.dst[o[r]] = v;
        }
      }
    } else {
      constexpr int CH = NB / 2;
      const int OHW = a.OH * a.OW;
      const int y = fdiv(px, a.m_w), x = px - y * W;
      const unsigned pix = unsigned((2 * y + dy) * a.OW + 2 * x);
      const unsigned pbase = unsigned(plane) * unsigned(a.cout);
#pragma unroll
      for (int cb = 0; cb < CH; ++cb) {
        float2 rv[16];
        unsigned o[16];
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          ok[r] = co < a.cout;
          o[r] = (pbase + unsigned(ok[r] ? co : 0)) * unsigned(OHW) + pix;
          if (HAS_RES) rv[r] = *reinterpret_cast<const float2*>(a.res + o[r]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          float v[2] = {acc[cb][r], acc[cb + CH][r]};
          const float rr[2] = {HAS_RES ? rv[r].x : 0.f, HAS_RES ? rv[r].y : 0.f};
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
          } else if (ok[r]) {
            *reinterpret_cast<float2*>(a.dst + o[r]) = make_float2(v[0], v[1]);
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
        const unsigned pb2 = unsigned(plane) * unsigned(a.cout2);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
          float x0 = bn_affine(h0[r], bias2[j], scale2[j], shift2[j]), x1 = bn_affine(h1[r], bias2[j], scale2[j], shift2[j]);
          if (relu2) {
            x0 = fmaxf(x0, 0.0f);
            x1 = fmaxf(x1, 0.0f);
          }
          if (j < a.cout2) *reinterpret_cast<float2*>(a.dst2 + (pb2 + unsigned(j)) * unsigned(OHW) + pix) = make_float2(x0, x1);
        }
      }
    }
  }
}

}  // namespace fvp
