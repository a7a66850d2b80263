// 3x3 stride-1 'same' conv as Winograd F(2x2,3x3) on the fp32 matrix cores (included by
// fvp_conv.hip; P2PNet's res-blocks, lib/models/cnns_2d.py:12-71, are >90 % of the path's FLOPs).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      2x2 outputs from a 4x4 input patch,
//
// i.e. 16 independent GEMMs  M_p[cout][tile] = sum_ci U_p[cout][ci] * V_p[ci][tile]  (p = 4*xi+nu):
// 16 multiplies per 4 outputs instead of 36.  Mapping on v_mfma_f32_16x16x4_f32 (4 channels per
// instruction, 4 accumulator registers per 16x16 tile):
//   A operand = U_p   lane l: cout l&15, channel ci + (l>>4)     (pre-transformed by k_pack_wino)
//   B operand = V_p   lane l: tile l&15, channel ci + (l>>4)     (transformed in registers from the
//                                                                 lane's own 4x4 patch in LDS)
//   D         = M_p   lane l: tile l&15, couts 4*(l>>4) + r
// A wave owns 32 couts x 16 tiles: 2 x 16 accumulator tiles = 128 registers, so two waves fit a
// SIMD and one wave's patch transform / LDS reads overlap the other's MFMAs.  For a fixed
// (cout, tile) all 16 M_p sit in the same lane and register slot: the output transform and the
// bias/BN/residual/ReLU epilogue are pure per-lane arithmetic, stored as float2 rows.
//
// Workgroup = 8 waves = WC cout blocks (32) x WT tile blocks (16).  LDS per chunk of CC channels
// (double buffered, filled by the LDS-DMA exactly like k_conv_dma):
//   Xs[CC][TN][TH+2][4 + W]   zero-margin dense rows (halo reads need no masking)
//   Ws[CC][32*WC][16]         quad q of row `co` stored at quad q ^ ((co>>2)&3): the four
//                             ds_read_b128 of a lane (xi = 0..3) are bank-conflict free unpadded
#pragma once

namespace fvp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WC, int WT, bool HAS_RES, int DIAG = 0>
__global__ void __launch_bounds__(512, 2) k_conv_wino(ConvArgs a) {
  HIP_DYNAMIC_SHARED(float, smem)
  static_assert(WC * WT == 8, "8 waves");
  constexpr int CBW = 32 * WC;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int k4 = lane >> 4, l15 = lane & 15;
  const int wc = wave % WC, wt = wave / WC;
  const int THp = a.TH + 2, W = a.W, WP = W + 4;
  const int plane_sz = THp * WP;
  const int CS = a.TN * plane_sz;
  const int xs_sz = a.CC * CS + 4, ws_sz = a.CC * CBW * 16, buf_sz = xs_sz + ws_sz;   // floats, all % 4 == 0

  const int tile = blockIdx.x;
  const int pg = tile / a.tiles_y;
  const int ty_i = tile - pg * a.tiles_y;
  const int plane0 = pg * a.TN;
  const int y0 = ty_i * a.TH;
  const int co0 = blockIdx.y * CBW;
  if (a.plane_valid && a.TN == 1 && !a.plane_valid[plane0 / a.valid_div]) return;
  const float* wts = a.wts + size_t(co0) * 16;       // [cinp][coutp][16]

  // this lane's 2x2 output tile inside the workgroup tile (exact cover: 16*WT = TN * tpp)
  const int q = wt * 16 + l15;
  const int tn = fdiv(q, a.m_tpp), trem = q - tn * a.tpp;
  const int ty = trem >> a.tpr_log2, tx = trem & ((1 << a.tpr_log2) - 1);
  // LDS row 0 of the tile is image row y0 - 1; column 4 of a row slot is image x = 0
  const int poff = tn * plane_sz + 2 * ty * WP + 3 + 2 * tx + k4 * CS;
  const int swz = (l15 >> 2) & 3;
  int aoff[4];                                       // cout block cb adds 16 rows = 256 floats
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) aoff[xi] = ((k4 * CBW + wc * 32 + l15) * 4 + (xi ^ swz)) * 4;

  f32x4 acc[2][16];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[cb][p][r] = 0.0f;

  const int HW = a.H * W;
  const int qpr = (W >> 2) + 1;
  const int rows_per_ch = a.TN * THp;
  const int nin = a.CC * rows_per_ch * qpr + 1;
  const int nwq = a.CC * CBW * 4;
  const int nchunks = (a.cinp + a.CC - 1) / a.CC;

  constexpr int kMaxIn = 4;                          // host guarantees nin <= kMaxIn * 512
  int in_off[kMaxIn], in_ci[kMaxIn];
#pragma unroll
  for (int j = 0; j < kMaxIn; ++j) {
    const int it = (wave + 8 * j) * 64 + lane;
    in_off[j] = -1;
    in_ci[j] = 0;
    if (it < nin) {
      const int row = fdiv(it, a.m_qpr), qd = it - row * qpr;
      const int ci = fdiv(row, a.m_rpc);
      const int rem = row - ci * rows_per_ch;
      const int n = fdiv(rem, a.m_thp), ry = rem - n * THp;
      const int plane = plane0 + n, y = y0 + ry - 1;
      in_ci[j] = ci;
      if (qd > 0 && ci < a.CC && plane < a.planes && y >= 0 && y < a.H)
        in_off[j] = (n * a.cin + ci) * HW + y * W + 4 * (qd - 1);
    }
  }
  const float* src_tile = a.src + size_t(plane0) * a.cin * HW;
  auto stage = [&](int k, int buf) {
    float* xs = smem + 4 + buf * buf_sz;
    float* ws = xs + xs_sz;
    const int c0 = k * a.CC;
#pragma unroll
    for (int j = 0; j < kMaxIn; ++j) {
      const int g = wave + 8 * j;
      if (g * 64 + lane < nin) {
        const bool ok = in_off[j] >= 0 && c0 + in_ci[j] < a.cin;
        const float* src = ok ? src_tile + size_t(c0) * HW + in_off[j] : a.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(xs + g * 256), 16, 0, 0);
      }
    }
    const float* gw = wts + size_t(c0) * a.coutp * 16;
    for (int g = wave; g * 64 < nwq; g += 8) {       // nwq is a multiple of 64
      const int it = g * 64 + lane;
      const int ci = it / (CBW * 4), qd = it - ci * (CBW * 4);
      const float* src = c0 + ci < a.cinp ? gw + size_t(ci) * a.coutp * 16 + 4 * qd : a.zeros;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(ws + g * 256), 16, 0, 0);
    }
  };

  if (!(a.ablate & 1)) stage(0, 0);
  __syncthreads();
  for (int k = 0; k < nchunks; ++k) {
    const int buf = k & 1;
    if (k + 1 < nchunks && !(a.ablate & 1)) stage(k + 1, buf ^ 1);
    const float* Xs = smem + 4 + buf * buf_sz;
    const float* Ws = Xs + xs_sz;
    // One step = 4 channels = two half-steps (cout block 0 / 1) of 16 MFMAs each.  The weight
    // quads of the NEXT half-step and the patch of the NEXT step are in flight while the
    // current MFMAs run; the patch transform (32 adds) is interleaved with them.
    float4 av[2][4];
    float dv[4][4];
    float vv[4][4];
    auto fetch_a = [&](int cb, int ci) {
      const int cic = ci < a.CC ? ci : a.CC - 4;     // last prefetch of a chunk: harmless re-read
      const float* ws = Ws + cic * (CBW * 16) + cb * 256;
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) av[cb][xi] = *reinterpret_cast<const float4*>(ws + aoff[xi]);
    };
    auto fetch_d = [&](int ci, int wp) {
      const int cic = ci < a.CC ? ci : a.CC - 4;
      const float* xs = Xs + cic * CS + poff;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* row = xs + r * wp;
        const float2 mid = *reinterpret_cast<const float2*>(row + 1);   // 8-byte aligned: 4 + 2*tx
        dv[r][0] = row[0];
        dv[r][1] = mid.x;
        dv[r][2] = mid.y;
        dv[r][3] = row[3];
      }
    };
    auto mfma16 = [&](int cb) {
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        acc[cb][4 * xi + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].x, vv[xi][0], acc[cb][4 * xi + 0], 0, 0, 0);
        acc[cb][4 * xi + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].y, vv[xi][1], acc[cb][4 * xi + 1], 0, 0, 0);
        acc[cb][4 * xi + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].z, vv[xi][2], acc[cb][4 * xi + 2], 0, 0, 0);
        acc[cb][4 * xi + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].w, vv[xi][3], acc[cb][4 * xi + 3], 0, 0, 0);
      }
    };
    fetch_a(0, 0);
    fetch_d(0, WP);
    for (int ci = (a.ablate & 4) ? a.CC : 0; ci < a.CC; ci += 4) {
      int wp = WP;
      FVP_OPAQUE(wp);
      // ---- half-step 0: transform the patch, cout block 0
      __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): av[0] and dv have landed
      __builtin_amdgcn_sched_barrier(0);
      if (!(DIAG & 2)) fetch_a(1, ci);
      __builtin_amdgcn_sched_barrier(0);             // issue the reads now: left alone hipcc sinks them below the MFMAs
      float tt[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {                  // B^T d
        tt[0][c] = dv[0][c] - dv[2][c];
        tt[1][c] = dv[1][c] + dv[2][c];
        tt[2][c] = dv[2][c] - dv[1][c];
        tt[3][c] = dv[1][c] - dv[3][c];
      }
      if (!(DIAG & 2)) fetch_d(ci + 4, wp);                           // dv is dead from here: refill for the next step
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {               // (B^T d) B
        if (DIAG & 1) {
          vv[xi][0] = dv[xi][0]; vv[xi][1] = dv[xi][1]; vv[xi][2] = dv[xi][2]; vv[xi][3] = dv[xi][3];
        } else {
        vv[xi][0] = tt[xi][0] - tt[xi][2];
        vv[xi][1] = tt[xi][1] + tt[xi][2];
        vv[xi][2] = tt[xi][2] - tt[xi][1];
        vv[xi][3] = tt[xi][1] - tt[xi][3];
        }
      }
      mfma16(0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- half-step 1: cout block 1
      __builtin_amdgcn_s_waitcnt(0xc07f);            // av[1] (and the next patch) have landed
      __builtin_amdgcn_sched_barrier(0);
      if (!(DIAG & 2)) fetch_a(0, ci + 4);
      __builtin_amdgcn_sched_barrier(0);
      mfma16(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  if (a.ablate & 8) return;
  // ---- output transform + epilogue, per lane: tile (plane, y, x), 8 couts
  const float* bias = a.epi;
  const float* scale = a.epi + a.coutp;
  const float* shift = a.epi + 2 * a.coutp;
  const bool relu = a.flags & FVP_EPI_RELU;
  const bool res_after = a.flags & FVP_EPI_RES_AFTER_RELU;
  const int plane = plane0 + tn, y = y0 + 2 * ty, x = 2 * tx;
  const bool tile_ok = plane < a.planes;
  const unsigned pix = tile_ok ? unsigned(y * W + x) : 0u;
  const unsigned cbase = tile_ok ? unsigned(plane) * a.cout : 0u;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    unsigned off[4];
    bool ok[4];
    int co[4];
    float2 r0[4], r1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      co[r] = co0 + wc * 32 + cb * 16 + 4 * k4 + r;
      ok[r] = tile_ok && co[r] < a.cout;
      off[r] = (cbase + (ok[r] ? co[r] : 0)) * unsigned(HW) + pix;
      if (HAS_RES) {
        r0[r] = *reinterpret_cast<const float2*>(a.res + off[r]);
        r1[r] = *reinterpret_cast<const float2*>(a.res + off[r] + W);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s[4][2];
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        const float m0 = acc[cb][4 * xi][r], m1 = acc[cb][4 * xi + 1][r], m2 = acc[cb][4 * xi + 2][r],
                    m3 = acc[cb][4 * xi + 3][r];
        s[xi][0] = (m0 + m1) + m2;
        s[xi][1] = (m1 - m2) - m3;
      }
      float o[2][2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        o[0][e] = (s[0][e] + s[1][e]) + s[2][e];
        o[1][e] = (s[1][e] - s[2][e]) - s[3][e];
      }
      const float b = bias[co[r]], sc = scale[co[r]], sh = shift[co[r]];
      const float rr[2][2] = {{HAS_RES ? r0[r].x : 0.f, HAS_RES ? r0[r].y : 0.f},
                              {HAS_RES ? r1[r].x : 0.f, HAS_RES ? r1[r].y : 0.f}};
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float v = bn_affine(o[i][e], b, sc, sh);
          if (HAS_RES && !res_after) v += rr[i][e];
          if (relu) v = fmaxf(v, 0.0f);
          if (HAS_RES && res_after) v += rr[i][e];
          o[i][e] = v;
        }
      if (ok[r]) {
        *reinterpret_cast<float2*>(a.dst + off[r]) = make_float2(o[0][0], o[0][1]);
        *reinterpret_cast<float2*>(a.dst + off[r] + W) = make_float2(o[1][0], o[1][1]);
      }
    }
  }
}

// state_dict weight [cout][cin][3][3] -> Winograd-domain U = G g G^T, layout [cinp][coutp][16]
// with quad xi of row `co` stored at quad xi ^ ((co>>2)&3).
__global__ void __launch_bounds__(256)
k_pack_wino(const float* __restrict__ w, int cin, int cout, int cinp, int coutp, float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cinp * coutp) return;
  const int co = i % coutp, ci = i / coutp;
  float g[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      g[ky][kx] = (co < cout && ci < cin) ? w[(size_t(co) * cin + ci) * 9 + ky * 3 + kx] : 0.0f;
  float gg[4][3];                                   // G g
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const float e = g[0][kx] + g[2][kx];
    gg[0][kx] = g[0][kx];
    gg[1][kx] = 0.5f * (e + g[1][kx]);
    gg[2][kx] = 0.5f * (e - g[1][kx]);
    gg[3][kx] = g[2][kx];
  }
  float* out = dst + size_t(i) * 16;
  const int swz = (co >> 2) & 3;
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) {                   // (G g) G^T
    const float e = gg[xi][0] + gg[xi][2];
    float* o = out + 4 * (xi ^ swz);
    o[0] = gg[xi][0];
    o[1] = 0.5f * (e + gg[xi][1]);
    o[2] = 0.5f * (e - gg[xi][1]);
    o[3] = gg[xi][2];
  }
}

}  // namespace fvp
