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
//               registers (13 quads), loaded with 13 coalesced global_load_dwordx4 from the k-grouped copy
//               [c][tap quad][g][n][4] that fvp_pack_conv writes behind the pixel-pair copy; the next group's arrive under
//               this group's MFMAs.
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
  // 49 taps = 13 quads of 4 (the last one holds one tap): 13 global_load_dwordx4 per channel group instead of 49 dword
  // loads - every vector instruction of a wave costs its SIMD ~12 cycles of matrix time
  const float4* wl = reinterpret_cast<const float4*>(a.wts) + lane;
  float4 bw[2][13];
#pragma unroll
  for (int tq = 0; tq < 13; ++tq) bw[0][tq] = wl[tq * 64];

  // ---- input tile: [NCH][THp rows of (margin quad + W/4 quads)] + pad, one DMA round
  if (!(FVP_K7_ABLATE & 1)) {
    fvp_i32x4 rs;
    const unsigned long long b = reinterpret_cast<unsigned long long>(a.src + size_t(plane) * a.cin * HW);
    rs[0] = __builtin_amdgcn_readfirstlane(int(unsigned(b)));
    rs[1] = __builtin_amdgcn_readfirstlane(int(unsigned(b >> 32) & 0xffffu));
    rs[2] = a.cin * HW * 4;                             // channels >= cin fail the range check
    rs[3] = 0x00020000;
    const unsigned lds0 = FVP_LDS_BYTE_ADDRESS(smem);
    // item j of this lane = item j - 1 + NT: (channel, row, quad) advance by compile-time steps with two carries instead of
    // two divisions per item (the set-up in front of 784 MFMAs is matrix time too)
    constexpr int kDCh = NT / QPC, kDRem = NT % QPC, kDRow = kDRem / QPR, kDQ = kDRem % QPR;
    constexpr int kPadRow = QPC / QPR, kPadQ = QPC % QPR;       // a channel plane = kPadRow rows + kPadQ quads
    const int it0 = wave * 64 + lane;
    int ch = it0 / QPC, rem0 = it0 - ch * QPC;
    int row = rem0 / QPR, q = rem0 - row * QPR;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      if ((wave + NW * j) * 64 + lane < NITEMS) {
        const int y = y0 - 3 + row;
        const bool ok = row < THp && q > 0 && y >= 0 && y < H;
        const unsigned vo = ok ? unsigned((ch * H + y) * W + 4 * (q - 1)) * 4u : 0x80000000u;
        asm_buffer_load_lds16(lds0 + unsigned(wave + NW * j) * 1024u, vo, rs, 0u);
      }
      ch += kDCh;
      row += kDRow;
      q += kDQ;
      if (q >= QPR) {
        q -= QPR;
        ++row;
      }
      // past the end of the channel plane (row * QPR + q >= QPC): next channel
      if (row > kPadRow || (row == kPadRow && q >= kPadQ)) {
        ++ch;
        row -= kPadRow;
        q -= kPadQ;
        if (q < 0) {
          q += QPR;
          --row;
        }
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
  // One opaque word offset per three rows: the reads of those rows are `base + small constant` (ds_read2_b32 reaches 255
  // words); left to itself hipcc rebuilds a base for almost every read pair (191 v_add_u32 per tile, 0.24 per MFMA).
  int boff = 0;
  auto fetch = [&](int set, int c, int r) {
    if (r % 3 == 0) {
      boff = c * 4 * CS + r * WP;
      FVP_OPAQUE_V(boff);
    }
    const float* rowp = xs + boff + (r % 3) * WP;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      if (FVP_K7_ABLATE & 2) {
        av[set][kx] = float(lane + kx);
        FVP_OPAQUE_V(av[set][kx]);
      } else {
        av[set][kx] = rowp[kx];
      }
    }
  };
  fetch(0, 0, 0);
#pragma unroll
  for (int c = 0; c < NCG; ++c) {
    if (c + 1 < NCG && !(FVP_K7_ABLATE & 4)) {
#pragma unroll
      for (int tq = 0; tq < 13; ++tq) bw[(c + 1) & 1][tq] = wl[((c + 1) * 13 + tq) * 64];
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
          if (ky >= 0 && ky < 7) {
            const int tp = ky * 7 + kx;
            const float4& wq = bw[(FVP_K7_ABLATE & 4) ? 0 : (c & 1)][tp >> 2];
            const float wv = (tp & 3) == 0 ? wq.x : (tp & 3) == 1 ? wq.y : (tp & 3) == 2 ? wq.z : wq.w;
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][kx], wv, acc[j], 0, 0, 0);
          }
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

// k-grouped copy of a 7x7 conv's weights [cout][cin][7][7] for k_conv7: [c][tap quad tq][lane = g * 16 + n][e] =
// w[n][4 c + g][tap 4 tq + e], zero for channels >= cin, couts >= cout and taps >= 49 (13 quads per channel group).
constexpr int kK7GroupFloats = 13 * 64 * 4;
__global__ void __launch_bounds__(256) k_pack_k7(const float* __restrict__ w, int cin, int cout, int ncg, float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ncg * kK7GroupFloats) return;
  const int e = i & 3, n = (i >> 2) & 15, g = (i >> 6) & 3, r = i >> 8;
  const int tq = r % 13, c = r / 13;
  const int ch = 4 * c + g, tp = 4 * tq + e;
  dst[i] = (ch < cin && n < cout && tp < 49) ? w[(size_t(n) * cin + ch) * 49 + tp] : 0.0f;
}

}  // namespace fvp
