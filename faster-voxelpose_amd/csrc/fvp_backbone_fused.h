// Fused kernels of the bf16 backbone (round 6): layers that used to round-trip their largest tensors through HBM
// as separate launches.  Included by fvp_backbone.hip (inside namespace fvp; Bf8, f32x16, mfma_bf16, pack_bf16x2
// come from there).
//
// k_bb_stem_pool: conv1 (7x7, stride 2, 3 -> 64) + bn1 + ReLU + MaxPool2d(3, 2, 1) (resnet.py:103-106, :185-188) in ONE
// kernel.  As two launches the stem wrote its 64-channel 256 x 480 map (629 MB for 40 images) and the pooling read it
// back: 429 + 165 us.  Here a workgroup owns a tile of 4 x 15 pooled pixels = 9 x 32 conv pixels (the 3 x 3 / 2 windows
// of neighbouring tiles overlap by one conv row / column, recomputed: 9/8 x 32/30), keeps the conv tile in LDS as bf16
// and stores only the pooled map.
//   * the conv is the same GEMM as in k_bb_conv's stem path (pixel-pair form: K = 7 rows x 4 pair taps x 8 = 224, 14
//     steps of v_mfma_f32_32x32x16_bf16 in the same k order), transposed: A = weights (row = cout), B = pixels (column =
//     conv pixel), so that a lane ends up with 4 consecutive couts of ONE pixel per accumulator quad - BN + ReLU, one
//     8-byte LDS write per quad, no transposition through LDS;
//   * wave w computes conv row w of the tile (32 pixels x 64 couts: two accumulator tiles); both operands come from LDS:
//     the input patch (23 rows x 35 pixel pairs x 16 B, zero outside the image = the conv's padding) at
//     (2 i + kh, j + p) - 32 consecutive 16-byte words per half wave, conflict-free - and the 28 KB of weights, resident
//     for the workgroup's whole life (persistent workgroups, two per CU, walk the tiles x-fastest so that neighbours'
//     halos meet in the XCD's L2);
//   * conv pixels outside the image are written as 0: every pooling window holds at least one real pixel and ReLU
//     outputs are >= 0, so 0 is the identity of the maximum (MaxPool2d pads with -inf); non-negative bf16 order like
//     unsigned integers, so the 3 x 3 maximum is v_pk_max_u16 on the packed pairs;
//   * the next tile's input patch is requested into registers before this tile's MFMAs.
constexpr int kStR = 4, kStQ = 15;                  // pooled rows / columns per tile
constexpr int kStCR = 2 * kStR + 1, kStCC = 32;     // conv rows / columns per tile (column 31 is computed, never used)
constexpr int kStIR = 2 * kStCR + 5;                // input rows of the patch: 23
constexpr int kStIC = kStCC + 3, kStIP = 36;        // pixel pairs per patch row (35) and their pitch
constexpr int kStWP = 232;                          // weight row pitch in bf16 (464 B: 16 rows land on 16 different bank quads)
constexpr int kStOP = 72;                           // conv-tile pixel pitch in bf16 (144 B)
constexpr int kStThreads = kStCR * 64;              // one wave per conv row: 576
constexpr int kStItems = kStIR * kStIC;             // 16-byte items of an input patch: 805
constexpr size_t kStLds = size_t(64) * kStWP * 2 + size_t(kStCR) * kStCC * kStOP * 2 + 128 * 4;   // weights | conv tile (aliases the patch) | scale, shift
static_assert(size_t(kStIR) * kStIP * 16 <= size_t(kStCR) * kStCC * kStOP * 2, "the input patch aliases the conv tile");
static_assert(kStItems <= 2 * kStThreads, "two patch items per thread");

struct BbStemArgs {
  const uint16_t* in;       // [N][H][W2] pixel pairs of 8 bf16 (fvp_bb_input)
  uint16_t* out;            // [N][PH][PW][64] bf16
  const uint16_t* w;        // [64][224] (k_bb_pack_stem)
  const float* epi;         // scale[64] | shift[64]
  int N, H, W2, CH, CW, PH, PW, tiles_x, tiles_y, ntiles;
};

__global__ void __launch_bounds__(kStThreads, 5) k_bb_stem_pool(BbStemArgs a) {
  HIP_DYNAMIC_SHARED(uint16_t, smem)
  uint16_t* wts = smem;                                  // [64][kStWP]
  uint16_t* tile = smem + 64 * kStWP;                    // input patch [23][36][8], then the conv tile [9][32][kStOP]
  float* epi_s = reinterpret_cast<float*>(tile + kStCR * kStCC * kStOP);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  // ---- resident weights and BN vectors
  for (int it = t; it < 64 * 28; it += kStThreads) {
    const int row = it / 28, g = it - row * 28;
    *reinterpret_cast<Bf8*>(wts + row * kStWP + g * 8) = *reinterpret_cast<const Bf8*>(a.w + row * 224 + g * 8);
  }
  if (t < 128) epi_s[t] = a.epi[t];

  // ---- input patch of a tile -> registers (two 16-byte items per thread; zeros outside the image)
  Bf8 pre[2];
  auto request = [&](int tid_) {
    const int tx = tid_ % a.tiles_x, r = tid_ / a.tiles_x, ty = r % a.tiles_y, n = r / a.tiles_y;
    const int iy0 = 4 * kStR * ty - 5, ix0 = 2 * kStQ * tx - 3;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = t + u * kStThreads;
      const int ar = it / kStIC, b = it - ar * kStIC;
      const int iy = iy0 + ar, ix = ix0 + b;
      const bool ok = it < kStItems && unsigned(iy) < unsigned(a.H) && unsigned(ix) < unsigned(a.W2);
      const size_t off = ok ? (size_t(n) * a.H + iy) * a.W2 + ix : 0;
      pre[u] = *reinterpret_cast<const Bf8*>(a.in + off * 8);
      if (!ok) pre[u] = Bf8{{0u, 0u, 0u, 0u}};
    }
  };
  int tid = blockIdx.x;
  if (tid < a.ntiles) request(tid);
  for (; tid < a.ntiles; tid += gridDim.x) {
    const int tx = tid % a.tiles_x, r0 = tid / a.tiles_x, ty = r0 % a.tiles_y, n = r0 / a.tiles_y;
    __syncthreads();                                     // the previous tile's pooling has read the conv tile (same memory)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = t + u * kStThreads;
      if (it < kStItems) {
        const int ar = it / kStIC, b = it - ar * kStIC;
        *reinterpret_cast<Bf8*>(tile + (ar * kStIP + b) * 8) = pre[u];
      }
    }
    __syncthreads();
    if (tid + int(gridDim.x) < a.ntiles) request(tid + int(gridDim.x));

    // ---- conv row `wave` of the tile: D[cout][pixel] over K = 224 in 14 steps (kh = s >> 1, pair taps 2 (s & 1) + half)
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const uint16_t* xrow = tile + ((2 * wave) * kStIP + l31 + half) * 8;
    const uint16_t* wrow = wts + l31 * kStWP + half * 8;
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const Bf8 xb = *reinterpret_cast<const Bf8*>(xrow + ((s >> 1) * kStIP + 2 * (s & 1)) * 8);
      const Bf8 w0 = *reinterpret_cast<const Bf8*>(wrow + s * 16);
      const Bf8 w1 = *reinterpret_cast<const Bf8*>(wrow + 32 * kStWP + s * 16);
      acc[0] = mfma_bf16(w0, xb, acc[0]);
      acc[1] = mfma_bf16(w1, xb, acc[1]);
    }
    __syncthreads();                                     // every wave has read the patch: the conv tile may overwrite it

    // ---- BN + ReLU, 0 outside the image, bf16, this lane's pixel (wave, l31): 8 quads of 4 consecutive couts
    {
      const int cy = 2 * kStR * ty - 1 + wave, cx = 2 * kStQ * tx - 1 + l31;
      const bool ok = unsigned(cy) < unsigned(a.CH) && unsigned(cx) < unsigned(a.CW);
      uint16_t* dst = tile + (wave * kStCC + l31) * kStOP;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = 32 * cb + 8 * q + 4 * half;
          const float4 sc = *reinterpret_cast<const float4*>(epi_s + c0);
          const float4 sh = *reinterpret_cast<const float4*>(epi_s + 64 + c0);
          float v0 = fmaxf(acc[cb][4 * q + 0] * sc.x + sh.x, 0.0f), v1 = fmaxf(acc[cb][4 * q + 1] * sc.y + sh.y, 0.0f);
          float v2 = fmaxf(acc[cb][4 * q + 2] * sc.z + sh.z, 0.0f), v3 = fmaxf(acc[cb][4 * q + 3] * sc.w + sh.w, 0.0f);
          typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 o = {ok ? pack_bf16x2(v0, v1) : 0u, ok ? pack_bf16x2(v2, v3) : 0u};
          *reinterpret_cast<u32x2*>(dst + c0) = o;
        }
    }
    __syncthreads();

    // ---- 3 x 3 / 2 maximum: thread = (pooled row r, pooled column q, 8-channel group g)
    if (t < kStR * kStQ * 8) {
      typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
      const int r = t / (kStQ * 8), rem = t - r * (kStQ * 8), q = rem >> 3, g = rem & 7;
      const int py = kStR * ty + r, px = kStQ * tx + q;
      if (py < a.PH && px < a.PW) {
        const uint16_t* src = tile + ((2 * r) * kStCC + 2 * q) * kStOP + g * 8;
        u16x8 m = *reinterpret_cast<const u16x8*>(src);
#pragma unroll
        for (int di = 0; di < 3; ++di)
#pragma unroll
          for (int dj = 0; dj < 3; ++dj)
            if (di | dj) m = __builtin_elementwise_max(m, *reinterpret_cast<const u16x8*>(src + (di * kStCC + dj) * kStOP));
        *reinterpret_cast<u16x8*>(a.out + ((size_t(n) * a.PH + py) * a.PW + px) * 64 + g * 8) = m;
      }
    }
  }
}
