// k_conv7 (round 5): the 7x7 front conv of P2PNet / CenterNet (cnns_2d.py:119-121: Basic2DBlock(cin, 16, 7), cin = joints)
// on v_mfma_f32_16x16x4_f32 with the reduction ordered (channel group of 4, kernel row, kernel column):
//
//   D[pixel m][cout n] += sum_{g < 4} X[4 c + g][y + ky][x0 + m + kx] * Wt[n][4 c + g][ky][kx]
//
//   A operand = activations: lane (m = l & 15, g = l >> 4) reads X of channel 4c + g, pixel x0 + m, from the LDS tile.  A
//               step (c, input row r, kx) is ONE ds_read_b32 at `lane base + immediate`; it is multiplied into every
//               output row j of the tile that the input row reaches (ky = r - j in 0..6), so an activation is read once per
//               R output rows: (R + 6) * 7 reads per 49 R MFMAs.
//   B operand = weights: lane (n = l & 15, g) holds Wt[n][4c + g][tap]: the 49 taps of a channel group live in 49
//               registers, loaded with coalesced 256-byte global loads from the k-grouped copy [c][tap][g][n] that
//               fvp_pack_conv writes behind the pixel-pair copy; the next group's 49 arrive under this group's MFMAs.
//   D         : lane holds pixels 4 g .. 4 g + 3 of cout n: one 16-byte store per output row.
//
// Against the pixel-pair form of k_conv_dma (32x32x2 tiles: rows 16..31 = the same couts one tap to the right, 8 tap
// columns per kernel row, channel PAIRS): 4 NCG x 49 k-steps instead of 8 x 56 double-width ones - 0.875 of the matrix
// work -, no weights in LDS (50 KB), no barrier inside the reduction, one wave per 16-pixel column block and a tile of R = 4
// rows: 15 tiles per CU for 240 planes of 64 x 64 (the 16-row tiles came to 3.75 per CU: a quarter of the last round idle).
// The input tile - every channel, R + 6 rows - is copied by ONE round of LDS-DMA; rows start with a 16-byte zero margin that
// serves as left halo and as the previous row's right halo (as in k_conv_dma), rows outside the image, channels >= cin and
// the pad behind each channel plane are out-of-range offsets: the DMA writes zeros.  The channel planes are CS words apart
// with CS = 16 (mod 64): the four channel groups of an A read fall on four different bank quarters.
// The order of the sum of an output - c, ky, kx ascending, four channels inside the MFMA - depends on nothing but the layer.
#pragma once

namespace fvp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int conv7_cs(int W, int R) {                 // words between channel planes of the LDS tile
  const int raw = (R + 6) * (W + 4);
  return raw + ((16 - raw % 64) + 64) % 64 + (((16 - raw % 64) + 64) % 64 < 4 ? 64 : 0);
}

#ifndef FVP_K7_R
#define FVP_K7_R 4                                     // output rows per tile (and per wave)
#endif
constexpr int kK7Rows = FVP_K7_R;
#ifndef FVP_K7_OCC
#define FVP_K7_OCC 3                                    // waves per SIMD the four-group instances are compiled for
#endif
#ifndef FVP_K7_ABLATE
#define FVP_K7_ABLATE 0                                // variant builds only (wrong results): 1 no tile DMA, 2 no LDS reads, 4 no weight loads
#endif

template <int W, int NCG, int R = kK7Rows>
__global__ void __launch_bounds__(W / 16 * 64, (NCG == 4 && R <= 4) ? FVP_K7_OCC : 2) k_conv7(ConvArgs a) {
  HIP_DYNAMIC_SHARED(float, smem)
  constexpr int NW = W / 16, NT = NW * 64;
  constexpr int THp = R + 6, WP = W + 4, QPR = WP / 4;
  constexpr int CS = conv7_cs(W, R), QPC = CS / 4;
  constexpr int NCH = 4 * NCG;
  constexpr int NITEMS = NCH * QPC;                    // 16-byte items of the tile (pads included: they must read zero)
  constexpr int NIT = (NITEMS + NT - 1) / NT;
  static_assert(CS % 64 == 16 && CS - THp * WP >= 4 && W % 16 == 0, "tile layout");
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, m = lane & 15;

  // tile id -> (plane, row band); eighths of the grid are contiguous per XCD (neighbouring bands share halo rows in L2)
  const int ntiles = gridDim.x;
  int tile = blockIdx.x;
  if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
  const int plane = tile / a.tiles_y;
  const int y0 = (tile - plane * a.tiles_y) * R;
  if (a.plane_valid && !a.plane_valid[plane / a.valid_div]) return;
  const int H = a.H, HW = H * W;

  // epilogue vectors and the first channel group's weights: in flight under the tile DMA
  const int n = m;                                      // this lane's cout
  const float e_bias = a.epi[n], e_scale = a.epi[a.coutp + n], e_shift = a.epi[2 * a.coutp + n];
  const float* wl = a.wts + lane;
  float bw[2][49];
#pragma unroll
  for (int tp = 0; tp < 49; ++tp) bw[0][tp] = wl[tp * 64];

  // ---- input tile: [NCH][THp rows of (margin quad + W/4 quads)] + pad, one DMA round
  if (!(FVP_K7_ABLATE & 1)) {
    fvp_i32x4 rs;
    const unsigned long long b = reinterpret_cast<unsigned long long>(a.src + size_t(plane) * a.cin * HW);
    rs[0] = __builtin_amdgcn_readfirstlane(int(unsigned(b)));
    rs[1] = __builtin_amdgcn_readfirstlane(int(unsigned(b >> 32) & 0xffffu));
    rs[2] = a.cin * HW * 4;                             // channels >= cin fail the range check
    rs[3] = 0x00020000;
    const unsigned lds0 = FVP_LDS_BYTE_ADDRESS(smem);
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int it = (wave + NW * j) * 64 + lane;
      if (it < NITEMS) {
        const int ch = it / QPC, rem = it - ch * QPC;
        const int row = rem / QPR, q = rem - row * QPR;
        const int y = y0 - 3 + row;
        const bool ok = row < THp && q > 0 && y >= 0 && y < H;
        const unsigned vo = ok ? unsigned((ch * H + y) * W + 4 * (q - 1)) * 4u : 0x80000000u;
        asm_buffer_load_lds16(lds0 + unsigned(wave + NW * j) * 1024u, vo, rs, 0u);
      }
    }
  }
  wait_vmcnt(0);
  __syncthreads();

  f32x4 acc[R];
#pragma unroll
  for (int j = 0; j < R; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  const float* xs = smem + g * CS + wave * 16 + m + 1;  // channel g of a group, pixel x0 + m, tap kx = 0 (margin 4 - pad 3)
  float av[2][7];
  auto fetch = [&](int set, int c, int r) {
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      if (FVP_K7_ABLATE & 2) {
        av[set][kx] = float(lane + kx);
        FVP_OPAQUE_V(av[set][kx]);
      } else {
        av[set][kx] = xs[c * 4 * CS + r * WP + kx];
      }
    }
  };
  fetch(0, 0, 0);
#pragma unroll
  for (int c = 0; c < NCG; ++c) {
    if (c + 1 < NCG && !(FVP_K7_ABLATE & 4)) {
#pragma unroll
      for (int tp = 0; tp < 49; ++tp) bw[(c + 1) & 1][tp] = wl[((c + 1) * 49 + tp) * 64];
    }
#pragma unroll
    for (int r = 0; r < THp; ++r) {
      const int cur = (c * THp + r) & 1, nxt = cur ^ 1;
      __builtin_amdgcn_s_waitcnt(0xc07f);               // this row's activations (issued one step ago) have landed
      __builtin_amdgcn_sched_barrier(0);
      if (r + 1 < THp) fetch(nxt, c, r + 1);
      else if (c + 1 < NCG) fetch(nxt, c + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kx = 0; kx < 7; ++kx)
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int ky = r - j;
          if (ky >= 0 && ky < 7)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][kx], bw[(FVP_K7_ABLATE & 4) ? 0 : (c & 1)][ky * 7 + kx], acc[j], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: bias, BatchNorm, ReLU; 16-byte stores (4 pixels of one cout per lane and row)
  if (n < a.cout) {
    const bool relu = a.flags & FVP_EPI_RELU;
    float* drow = a.dst + (size_t(plane) * a.cout + n) * HW + wave * 16 + 4 * g;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int y = y0 + j;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = bn_affine(acc[j][e], e_bias, e_scale, e_shift);
        if (relu) x = fmaxf(x, 0.0f);
        o[e] = x;
      }
      if (y < H) *reinterpret_cast<float4*>(drow + size_t(y) * W) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// k-grouped copy of a 7x7 conv's weights [cout][cin][7][7] for k_conv7: [c][tap][g][n] = w[n][4 c + g][tap], zero for
// channels >= cin and couts >= cout.
__global__ void __launch_bounds__(256) k_pack_k7(const float* __restrict__ w, int cin, int cout, int ncg, float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ncg * 49 * 64) return;
  const int n = i & 15, g = (i >> 4) & 3, r = i >> 6;
  const int tp = r % 49, c = r / 49;
  const int ch = 4 * c + g;
  dst[i] = (ch < cin && n < cout) ? w[(size_t(n) * cin + ch) * 49 + tp] : 0.0f;
}

}  // namespace fvp
