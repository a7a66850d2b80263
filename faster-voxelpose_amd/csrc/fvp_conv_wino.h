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
// (three slots filled by the LDS-DMA two chunks ahead; layout as in k_conv_dma):
//   Xs[CC][TN][TH+2][4 + W]   zero-margin dense rows (halo reads need no masking)
//   Ws[CC][32*WC][16]         quad q of row `co` stored at quad q ^ ((co>>2)&3): the four
//                             ds_read_b128 of a lane (xi = 0..3) are bank-conflict free unpadded
#pragma once

// cache policy of the input-tile DMA (aux bits of global_load_lds); nt (= 2) measured 1 % slower
#ifndef FVP_WINO_TIMING
#define FVP_WINO_TIMING 0
#endif
// Chunk barrier of the K loop.  __syncthreads() is a workgroup FENCE + barrier: the fence makes hipcc drain every
// outstanding LDS-DMA (s_waitcnt vmcnt(0)) - including the chunk requested a moment ago - so the counted vmcnt in front
// of it was dead code and the ring never had a chunk in flight across a barrier (found in the ISA in round 3).  The
// plain s_barrier leaves the counters to the explicit waits: lgkmcnt(0) (this wave's reads of the slot are done) and
// vmcnt(nps) (its items of chunk g+1 have landed; chunk g+2 stays in flight).
#ifndef FVP_WINO_FENCE_BARRIER
#define FVP_WINO_FENCE_BARRIER 0
#endif
#if FVP_WINO_FENCE_BARRIER
#define FVP_WINO_BARRIER() __syncthreads()
#else
#define FVP_WINO_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#ifndef FVP_WINO_YOUNG_PRIO
#define FVP_WINO_YOUNG_PRIO 0
#endif
#ifndef FVP_WINO_STORES_IN_FLIGHT
#define FVP_WINO_STORES_IN_FLIGHT 1
#endif
// Ablation switches other than 1 (no DMA) and 8 (no epilogue) - FVP_CONV_ABLATE bits 4, 16, 32, 64, 128, 256, 512, 1024 -
// only exist in a diagnostics build
#ifndef FVP_WINO_DIAG
#define FVP_WINO_DIAG 0
#endif
#ifndef FVP_WINO_EPI_FAST
#define FVP_WINO_EPI_FAST 1
#endif
// 1: the K loop's LDS-DMA uses buffer addressing (no vector instruction per chunk), 0: per-lane global addresses
#ifndef FVP_WINO_BUF_DMA
#define FVP_WINO_BUF_DMA 1
#endif
#ifndef FVP_WINO_ZERO_C
#define FVP_WINO_ZERO_C 1
#endif
#ifndef FVP_WINO_ASM_DMA
#define FVP_WINO_ASM_DMA 1
#endif
#ifndef FVP_WINO_IN_AUX
#define FVP_WINO_IN_AUX 0
#endif

namespace fvp {

// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to 1 KB of LDS at `l` (wave-uniform).
// FVP_WINO_ASM_DMA: the instruction is emitted as inline asm.  With the builtin, hipcc knows that LDS is written behind
// its back and puts a full `s_waitcnt vmcnt(0)` in front of the first LDS read that follows a pending DMA - i.e. at
// the top of every chunk of the K loop, right after the chunk two ahead was requested: every chunk paid the whole
// L2 -> LDS latency and the counted waits of the source were dead code (round 3, found in the ISA; the K loop compiled
// without the DMA has no vmcnt wait at all).  As asm the DMA is invisible to the waitcnt pass; ordering is what the
// source says: s_waitcnt vmcnt(n) counted per chunk + s_barrier.
// The LDS destination is given as (array, float index): the generic -> LDS cast of the bare array folds to a constant; a
// cast of a pointer VARIABLE makes hipcc emit a null check that this compiler version mis-selects ("Illegal instruction").
#if FVP_WINO_ASM_DMA
#define FVP_WINO_LDS_DMA16(g, lds, idx, aux) \
  asm_global_load_lds16(g, __builtin_amdgcn_readfirstlane(FVP_LDS_BYTE_ADDRESS(lds) + 4u * unsigned(idx)))
#else
#define FVP_WINO_LDS_DMA16(g, lds, idx, aux)                                                          \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g),                \
                                   (__attribute__((address_space(3))) void*)(const_cast<float*>(lds) + (idx)), 16, 0, aux)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Column pass of the input transform on register pairs E = (t0, t3), M = (t1, t2):
//   v03 = (t0 - t2, t1 - t3)   v12 = (t1 + t2, t2 - t1)
// Plain scalar adds.  (Round 1 used two hand-written v_pk_add_f32 with op_sel / neg modifiers here.  Inline
// asm hides the instruction from the compiler's hazard recognizer, and with one wave of the workgroup per SIMD
// (the 4-wave tiling) under concurrent kernels the packed result was sporadically consumed by the following MFMA
// before it was valid: timing-dependent wrong nu = 0 columns.  Scalar adds measured the same speed:
// MI355X_MICROARCH.md lists packed f32 VALU beside MFMAs as an anti-lever anyway.)
__device__ __forceinline__ void wino_cols(f32x2 E, f32x2 M, f32x2& v03, f32x2& v12) {
  v03 = f32x2{E.x - M.y, M.x - E.y};
  v12 = f32x2{M.x + M.y, M.y - M.x};
}

// RESW: the whole Winograd-domain weight tensor of the workgroup's cout block ([cinp][CBW][16], <= 64 KB)
// stays resident in LDS behind the three input slots (loaded once per persistent workgroup) instead of
// streaming through the slots chunk by chunk: for the 32-channel layers the weight chunks were more than
// half of the LDS-DMA traffic of a unit.
template <int WC, int WT, int CC, bool HAS_RES, bool RESW = false>
__global__ void __launch_bounds__(WC * WT * 64, 2) k_conv_wino(ConvArgs a) {
  HIP_DYNAMIC_SHARED(float, smem)
  constexpr int NWV = WC * WT;                       // waves per workgroup: 8 (one workgroup per CU) or 4 (two per CU)
  static_assert(NWV == 8 || NWV == 4, "4 or 8 waves");
  static_assert(CC == 4 || CC == 8, "chunk");
  constexpr int CBW = 32 * WC;
  constexpr int WCH = CC * CBW * 16;                 // floats of one weight chunk
  constexpr int WS_SZ = RESW ? 0 : WCH;              // ... streamed through a slot
  constexpr int NW = RESW ? 0 : CC * WC * 2 / NWV;   // weight DMA instructions per wave per chunk
  constexpr int S = CC / 4;                          // steps (4 channels) per chunk
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int k4 = lane >> 4, l15 = lane & 15;
  const int wc = wave % WC, wt = wave / WC;
  const int THp = a.TH + 2, W = a.W, WP = W + 4;
  const int plane_sz = THp * WP;
  const int CS = a.TN * plane_sz;
  const int xs_sz = a.wino_ni * (NWV * 256);         // input slot, padded to whole DMA rounds (floats)
  const int buf_sz = xs_sz + WS_SZ;

  // Persistent workgroups: unit u = (plane group, row band, cout block); workgroup b walks
  // u = b, b + G, b + 2G, ... (G = gridDim.x <= number of CUs).  The chunk stream (DMA two chunks
  // ahead) runs across unit boundaries, so a unit's first chunks land while the previous unit is
  // still computing and there is no workgroup relaunch between tiles.
  const int G = gridDim.x, nunits = a.nunits;
  auto unit_valid = [&](int u) {
    if (!a.plane_valid || a.TN != 1) return true;
    const int pg = fdiv(fdiv(u, a.m_ys), a.m_ty);
    return a.plane_valid[pg / a.valid_div] != 0;
  };
  auto next_unit = [&](int u) {
    while (u < nunits && !unit_valid(u)) u += G;
    return u;
  };
  int u = next_unit(blockIdx.x);
  if (u >= nunits) return;
#if FVP_WINO_YOUNG_PRIO
  // the second-dispatched half of an 8-wave workgroup loses the VALU arbitration on every segment (age): static priority
  if (NWV == 8 && wave >= NWV / 2) __builtin_amdgcn_s_setprio(FVP_WINO_YOUNG_PRIO);
#endif

  // this lane's 2x2 output tile inside the workgroup tile: TN planes x TR rows x tpr tiles; lanes beyond that
  // product (row lengths that do not divide 16*WT) compute on tile 0's data and store nothing
  const int q0 = wt * 16 + l15;
  const bool q_ok = q0 < a.TN * a.tpp;
  const int q = q_ok ? q0 : 0;
  const int tn = fdiv(q, a.m_tpp), trem = q - tn * a.tpp;
  const int ty = fdiv(trem, a.m_tpr), tx = trem - ty * a.tpr;
  // LDS row 0 of the tile is image row y0 - 1; column 4 of a row slot is image x = 0
  const int poff = tn * plane_sz + 2 * ty * WP + 3 + 2 * tx + k4 * CS;
  const int swz = (l15 >> 2) & 3;
  int aoff[4];                                       // cout block cb adds 16 rows = 256 floats
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) aoff[xi] = ((k4 * CBW + wc * 32 + l15) * 4 + (xi ^ swz)) * 4;
  const float* const wres = smem + 4 + 3 * buf_sz;   // RESW: resident weights [cinp][CBW][16]
  // first weight float of chunk k living in slot `slot` (streamed) or in the resident copy
  auto wchunk = [&](const float* slot, int k) { return RESW ? wres + k * WCH : slot + xs_sz; };

  f32x4 acc[2][16];
#if !FVP_WINO_ZERO_C
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[cb][p][r] = 0.0f;
#endif

  const int HW = a.H * W;
  const int qpr = (W >> 2) + 1;
  const int rows_per_ch = a.TN * THp;
  const int nin = CC * rows_per_ch * qpr + 1;        // + the zero quad behind the last row
  const int nchunks = a.cinp / CC;
  const int nps = a.wino_ni + NW;                    // DMA instructions per wave per chunk (uniform)

  constexpr int kMaxIn = 4;                          // host guarantees wino_ni <= kMaxIn
  const size_t in_step = size_t(CC) * HW, w_step = size_t(CC) * a.coutp * 16;
  // this lane's weight item j: channel ci0 + j * DCI of the chunk, quad qd0 of the cout block's row
  constexpr int DCI = NWV * 64 / (CBW * 4);
  const int wit = wave * 64 + lane;
  const size_t woff0 = size_t(wit / (CBW * 4)) * a.coutp * 16 + 4 * (wit % (CBW * 4));
  const size_t wdj = size_t(DCI) * a.coutp * 16;
  int su = u, sk = 0;                                // DMA cursor (unit su, chunk sk)
#if FVP_WINO_BUF_DMA
  // ---- DMA through buffer addressing (round 3).  On this part the fp32 MFMA runs on the vector ALUs: a VALU
  // instruction of EITHER wave of a SIMD takes matrix time away (tools/micro/coexec.hip: MFMA bursts of one wave + a
  // VALU stream of the other = 0.98 + 0.7 x 0.50 ms, not max), and the global-address form spent ~8 VALU instructions per
  // DMA item and chunk (64-bit select between the image and the zero page, pointer add, readfirstlane for M0).  As a
  // buffer load the item is: descriptor (SGPRs: the unit's base, set when the cursor enters a unit) + per-lane 32-bit byte
  // offset (VGPR, per unit) + chunk offset (SGPR) -> NO vector instruction per chunk; lanes outside the image carry an
  // offset that fails the range check and the hardware writes zeros to LDS (tools/micro/buflds.hip), so the zero page
  // and the select are gone too.
  constexpr unsigned kOOB = 0x80000000u;             // + any chunk offset (< 2^31) still fails the range check
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = FVP_LDS_BYTE_ADDRESS(smem);
  unsigned voff[kMaxIn];                             // this lane's input items: byte offset from (unit base - one row), or kOOB
  const unsigned woffb = unsigned(woff0) * 4u;       // this lane's weight item 0 (bytes from the unit's cout block, chunk 0)
  i32x4 rs_in = {0, 0, 0x7ffffff0, 0x00020000}, rs_w = {0, 0, 0x7ffffff0, 0x00020000};   // raw buffers, stride 0
  auto set_base = [](i32x4& rs, const float* p) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    rs[0] = __builtin_amdgcn_readfirstlane(int(unsigned(b)));
    rs[1] = __builtin_amdgcn_readfirstlane(int(unsigned(b >> 32) & 0xffffu));
  };
  auto enter_unit = [&](int su_) {
    const int st = fdiv(su_, a.m_ys), sy = su_ - st * a.ysplit;
    const int spg = fdiv(st, a.m_ty), sty = st - spg * a.tiles_y;
    const int splane0 = spg * a.TN, sy0 = sty * a.TH;
    // row 0 of a slot is image row sy0 - 1: the descriptor starts one row above the band so that offsets are >= 0
    set_base(rs_in, a.src + size_t(splane0) * a.cin * HW + sy0 * W - W);
    set_base(rs_w, a.wts + size_t(sy) * (CBW * 16));
#pragma unroll
    for (int j = 0; j < kMaxIn; ++j) {
      voff[j] = kOOB;
      const int it = (wave + NWV * j) * 64 + lane;
      if (j < a.wino_ni && it < nin) {
        const int row = fdiv(it, a.m_qpr), qd = it - row * qpr;
        const int ci = fdiv(row, a.m_rpc);
        const int rem = row - ci * rows_per_ch;
        const int n = fdiv(rem, a.m_thp), ry = rem - n * THp;
        if (qd > 0 && ci < CC && unsigned(sy0 + ry - 1) < unsigned(a.H) && splane0 + n < a.planes)
          voff[j] = unsigned((n * a.cin + ci) * HW + ry * W + 4 * (qd - 1)) * 4u;
      }
    }
  };
  auto buf_dma16 = [&](unsigned vo, const i32x4& rs, unsigned so, unsigned la) { asm_buffer_load_lds16(la, vo, rs, so); };
  // every wave issues exactly nps DMA instructions per chunk (counted s_waitcnt vmcnt below)
  auto stage = [&](int k, int boff) {
    const unsigned so_in = unsigned(k) * unsigned(in_step) * 4u;
    const unsigned la0 = lds0 + 4u * unsigned(4 + boff + wave_s * 256);
#pragma unroll
    for (int j = 0; j < kMaxIn; ++j)
      if (j < a.wino_ni) buf_dma16(voff[j], rs_in, so_in, la0 + unsigned(NWV * j) * 1024u);
    const unsigned so_w = unsigned(k) * unsigned(w_step) * 4u;
#pragma unroll
    for (int j = 0; j < NW; ++j)
      buf_dma16(woffb, rs_w, so_w + unsigned(j) * unsigned(wdj) * 4u, la0 + unsigned(xs_sz + NWV * j * 256) * 4u);
  };
#else
  // unit-invariant part of this lane's input DMA items: offset relative to the unit's first
  // (plane, channel chunk, row band) and {row in slot, plane in group, channel in chunk}
  int rel_off[kMaxIn], meta[kMaxIn];
#pragma unroll
  for (int j = 0; j < kMaxIn; ++j) {
    const int it = (wave + NWV * j) * 64 + lane;
    rel_off[j] = 0;
    meta[j] = -1;                                    // zero page: margins, padding items
    if (it < nin) {
      const int row = fdiv(it, a.m_qpr), qd = it - row * qpr;
      const int ci = fdiv(row, a.m_rpc);
      const int rem = row - ci * rows_per_ch;
      const int n = fdiv(rem, a.m_thp), ry = rem - n * THp;
      if (qd > 0 && ci < CC) {
        rel_off[j] = (n * a.cin + ci) * HW + (ry - 1) * W + 4 * (qd - 1);
        meta[j] = ry | (n << 8);
      }
    }
  }
  // ---- DMA cursor (unit su, chunk sk).  Everything that depends on the unit only - the base pointers
  // and which of this lane's items fall inside the image - is computed when the cursor ENTERS a unit;
  // per chunk a DMA item then costs a pointer add and a select (the address code used to be ~600
  // instructions per chunk and wave, issued between the MFMA bursts of the other wave of the SIMD).
  // The host guarantees cin % CC == 0 (no partial channel chunks).
  const float* ubase = a.src;                        // uniform: the cursor unit's (plane group, row band), channel 0
  const float* gwbase = a.wts;                       // uniform: the cursor unit's cout block, channel 0
  unsigned okmask = 0;                               // per lane: bit j <=> item j reads the image, else the zero page
  auto enter_unit = [&](int su_) {
    const int st = fdiv(su_, a.m_ys), sy = su_ - st * a.ysplit;
    const int spg = fdiv(st, a.m_ty), sty = st - spg * a.tiles_y;
    const int splane0 = spg * a.TN, sy0 = sty * a.TH;
    ubase = a.src + size_t(splane0) * a.cin * HW + sy0 * W;
    gwbase = a.wts + size_t(sy) * (CBW * 16);
    okmask = 0;
#pragma unroll
    for (int j = 0; j < kMaxIn; ++j) {
      const int ry = meta[j] & 255, n = (meta[j] >> 8) & 255;
      const bool ok = meta[j] >= 0 && unsigned(sy0 + ry - 1) < unsigned(a.H) && splane0 + n < a.planes;
      okmask |= ok ? (1u << j) : 0u;
    }
  };
  // every wave issues exactly nps DMA instructions per chunk (counted s_waitcnt vmcnt below)
  auto stage = [&](int k, int boff) {
    const float* bk = ubase + size_t(k) * in_step;
#pragma unroll
    for (int j = 0; j < kMaxIn; ++j) {
      if (j < a.wino_ni) {
        const int g = wave + NWV * j;
        const float* src = ((okmask >> j) & 1u) ? bk + rel_off[j] : a.zeros;
        FVP_WINO_LDS_DMA16(src, smem, 4 + boff + g * 256, FVP_WINO_IN_AUX);
      }
    }
    const float* wk = gwbase + size_t(k) * w_step + woff0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int g = wave + NWV * j;
      FVP_WINO_LDS_DMA16(wk + j * wdj, smem, 4 + boff + xs_sz + g * 256, 0);
    }
  };
#endif
  enter_unit(su);
  auto advance_cursor = [&]() {
    if (++sk == nchunks) {
      sk = 0;
      su = next_unit(su + G);
      if (su < nunits) enter_unit(su);
    }
  };
  auto stage_next = [&](int boff) {
    if (su >= nunits) return false;
    stage(sk, boff);
    advance_cursor();
    return true;
  };
  // ---- operand fetch / transform / MFMA building blocks
  float4 av[2][4];
  f32x2 dM[4], dE[4];                                // patch rows as pairs (d1,d2) and (d0,d3)
  f32x2 v03[4], v12[4];                              // V[xi][0],V[xi][3] and V[xi][1],V[xi][2]
  auto fetch_a = [&](int cb, const float* wbase, int s) {     // wbase = weights of the chunk, s = step in chunk
    if (FVP_WINO_DIAG && (a.ablate & 256)) return;                      // diagnostics: no A-operand reads
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
      av[cb][xi] = *reinterpret_cast<const float4*>(wbase + aoff[xi] + (s * 4 * CBW * 16 + cb * 256));
  };
  auto fetch_d = [&](const float* base, int s, int wp) {
    const float* xs = base + poff + s * 4 * CS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* row = xs + r * wp;
      dM[r] = *reinterpret_cast<const f32x2*>(row + 1);       // 8-byte aligned: column 4 + 2*tx
      dE[r] = (FVP_WINO_DIAG && (a.ablate & 64)) ? dM[r] : f32x2{row[0], row[3]};   // (bit 64, diagnostics: no single-float patch reads)
    }
  };
  auto transform_rows = [&](f32x2 (&tM)[4], f32x2 (&tE)[4]) {  // B^T d
    tM[0] = dM[0] - dM[2];  tE[0] = dE[0] - dE[2];
    tM[1] = dM[1] + dM[2];  tE[1] = dE[1] + dE[2];
    tM[2] = dM[2] - dM[1];  tE[2] = dE[2] - dE[1];
    tM[3] = dM[1] - dM[3];  tE[3] = dE[1] - dE[3];
  };
  // first = the unit's first step: the MFMAs take the constant 0 as C.  (Clearing the 128 accumulator registers between
  // units cost 128 vector moves per wave and unit - and on this part a vector instruction of either wave of a SIMD is
  // matrix time lost, see the DMA note above.)
  auto mfma16 = [&](int cb, bool first) {
    if (FVP_WINO_DIAG && (a.ablate & 4)) return;                        // diagnostics: no MFMA
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    if (first) {
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        acc[cb][4 * xi + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].x, v03[xi].x, z, 0, 0, 0);
        acc[cb][4 * xi + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].y, v12[xi].x, z, 0, 0, 0);
        acc[cb][4 * xi + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].z, v12[xi].y, z, 0, 0, 0);
        acc[cb][4 * xi + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].w, v03[xi].y, z, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        acc[cb][4 * xi + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].x, v03[xi].x, acc[cb][4 * xi + 0], 0, 0, 0);
        acc[cb][4 * xi + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].y, v12[xi].x, acc[cb][4 * xi + 1], 0, 0, 0);
        acc[cb][4 * xi + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].z, v12[xi].y, acc[cb][4 * xi + 2], 0, 0, 0);
        acc[cb][4 * xi + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][xi].w, v03[xi].y, acc[cb][4 * xi + 3], 0, 0, 0);
      }
    }
  };

  // ---- three chunk slots: chunk g+2 streams in while chunk g is consumed
  const bool dma = !(a.ablate & 1);
  if (RESW) {                                        // cinp * CBW * 4 quads, NWV * 64 per round
    const int rounds = (a.cinp * CBW * 4) / (NWV * 64);
    for (int j = 0; j < rounds; ++j) {
      const int g = wave + NWV * j;
      FVP_WINO_LDS_DMA16(a.wts + size_t(g * 64 + lane) * 4, smem, 4 + 3 * buf_sz + g * 256, 0);
    }
    if (!dma) wait_vmcnt(0);
  }
  // bias | scale | shift of every cout, [3][coutp], behind the slots (and the resident weights): the epilogue reads them
  // with ds_read (lgkmcnt).  As global loads they sat in the in-order vmcnt queue behind the previous cout's stores, and
  // every one of the 8 couts of a lane paid a store round trip plus a load round trip (round 3, found in the ISA).
  {
    float* const e = const_cast<float*>(smem) + 4 + 3 * buf_sz + (RESW ? a.cinp * CBW * 16 : 0);
    for (int i = t; i < 3 * a.coutp; i += NWV * 64) e[i] = a.epi[i];
  }
  if (dma) {
    stage_next(0);
    const bool second = stage_next(buf_sz);
    wait_vmcnt(second ? nps : 0);
  }
  __syncthreads();
#if FVP_WINO_TIMING
  // diagnostics build: s_memtime stamps around the phases of a step; the stamps are consumed right after the next
  // lgkmcnt(0) wait of the loop itself, so they add no waits of their own
  unsigned long long tacc[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tV = 0, tBar = 0;
  unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tEE = 0, tF = 0, tG = 0;
  bool tvalid = false;
  const unsigned long long tstart = __builtin_readcyclecounter();
#define FVP_TS(x) x = __builtin_readcyclecounter()
#else
#define FVP_TS(x)
#endif
  int cur_off = 0;
  int st_pending = 0;                                // 1: the previous unit's epilogue drained the DMA queue (its stores may still be in flight)
  fetch_a(0, wchunk(smem + 4, 0), 0);
  fetch_d(smem + 4, 0, WP);
  while (true) {
  // counted wait + barrier of a chunk (see the ring description above); first = the unit's first chunk
  auto chunk_barrier = [&](auto firstc, bool more) {
    constexpr bool kFirstB = decltype(firstc)::value;
    if (dma) {
      if constexpr (kFirstB || !FVP_WINO_ZERO_C) {
        // a unit's first chunk: behind an epilogue that drained the DMA queue (st_pending, see there) the chunk this
        // barrier guards has already landed and the epilogue's stores may stay in flight: no vmcnt wait at all
        if (!st_pending) wait_vmcnt_small(more ? nps : 0);
        st_pending = 0;
      } else {
        wait_vmcnt_small(more ? nps : 0);          // nps <= 8: a handful of scalar instructions instead of ~30
      }
    }
    if (!(FVP_WINO_DIAG && (a.ablate & 128))) FVP_WINO_BARRIER();     // (bit 128, diagnostics: no chunk barrier)
  };
  // the chunk body exists twice: the unit's first chunk (its first step's MFMAs take C = 0) and every other one
  auto chunk = [&](int k, auto firstc) {
    constexpr bool kFirst = decltype(firstc)::value;
    const int nxt_off = cur_off + buf_sz >= 3 * buf_sz ? 0 : cur_off + buf_sz;
    const int nn_off = nxt_off + buf_sz >= 3 * buf_sz ? 0 : nxt_off + buf_sz;
#if FVP_WINO_TIMING
    const unsigned long long tS0 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
#endif
    const bool more = dma && stage_next(nn_off);
#if FVP_WINO_TIMING
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long tS1 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    tacc[10] += tS1 - tS0;
#endif
    const float* cur = smem + 4 + cur_off;
    const float* nxt = smem + 4 + nxt_off;
    int wp = WP;
    FVP_OPAQUE(wp);
#pragma unroll
    for (int s = 0; s < S; ++s) {
      // ---- half-step 0: patch transform, cout block 0
#if FVP_WINO_TIMING
      const unsigned long long tA2 = __builtin_readcyclecounter();
#endif
      __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): av[0] and the patch have landed
      __builtin_amdgcn_sched_barrier(0);
#if FVP_WINO_TIMING
      if (tvalid) {                                  // previous step: every stamp has returned (wait above)
        tacc[0] += tB - tA; tacc[1] += tC - tB; tacc[2] += tD - tC; tacc[3] += tEE - tD; tacc[4] += tF - tEE;
        tacc[5] += tG - tF; tacc[6] += tA2 - tG;
      }
      tvalid = true;
      tA = tA2;
      FVP_TS(tB);
      __builtin_amdgcn_sched_barrier(0);
#endif
      fetch_a(1, wchunk(cur, k), s);
      __builtin_amdgcn_sched_barrier(0);             // issue the reads now: left alone hipcc sinks them below the MFMAs
      f32x2 tM[4], tE[4];
      if (FVP_WINO_DIAG && (a.ablate & 512)) {                          // diagnostics: no input transform
#pragma unroll
        for (int r = 0; r < 4; ++r) { tM[r] = dM[r]; tE[r] = dE[r]; }
      } else {
        transform_rows(tM, tE);
      }
      if (s + 1 < S) fetch_d(cur, s + 1, wp);        // the patch registers are dead: refill for the next step
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        if (FVP_WINO_DIAG && (a.ablate & 512)) { v03[xi] = tE[xi]; v12[xi] = tM[xi]; }
        else wino_cols(tE[xi], tM[xi], v03[xi], v12[xi]);
      }
#if FVP_WINO_TIMING
      __builtin_amdgcn_sched_barrier(0);
      FVP_TS(tC);
      __builtin_amdgcn_sched_barrier(0);
#endif
      mfma16(0, kFirst && s == 0);
      __builtin_amdgcn_sched_barrier(0);
      FVP_TS(tD);
      // ---- half-step 1: cout block 1; the last one of a chunk crosses into the next slot
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_sched_barrier(0);
      FVP_TS(tEE);
      if (s + 1 < S) {
        fetch_a(0, wchunk(cur, k), s + 1);
      } else {
        // all reads of this slot are complete (lgkmcnt above); once every wave is here the slot
        // may be overwritten by the DMA of chunk g+3, and chunk g+1 has landed for everybody
        chunk_barrier(std::integral_constant<bool, kFirst>{}, more);
        if (k + 1 < nchunks) {                       // (a unit's last chunk: the epilogue needs the registers)
          fetch_a(0, wchunk(nxt, k + 1), 0);
          fetch_d(nxt, 0, wp);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      FVP_TS(tF);
      mfma16(1, kFirst && s == 0);
      __builtin_amdgcn_sched_barrier(0);
      FVP_TS(tG);
    }
    cur_off = nxt_off;
  };
  if (FVP_WINO_ZERO_C) chunk(0, std::integral_constant<bool, true>{});
  for (int k = FVP_WINO_ZERO_C ? 1 : 0; k < nchunks; ++k) chunk(k, std::integral_constant<bool, false>{});

  // ---- unit finished: output transform + epilogue, then the next unit of this workgroup
#if FVP_WINO_TIMING
  const unsigned long long tE0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
#endif
  int ue = u;
  FVP_OPAQUE(ue);                                    // keeps the epilogue's address math out of the K loop's live set
  const int ut = fdiv(ue, a.m_ys), uy = ue - ut * a.ysplit;
  const int pg = fdiv(ut, a.m_ty), ty_i = ut - pg * a.tiles_y;
  const int plane0 = pg * a.TN, y0 = ty_i * a.TH, co0 = uy * CBW;
  if (!(a.ablate & 8)) {
  // per lane: tile (plane, y, x), 8 couts
  const bool relu = a.flags & FVP_EPI_RELU;
  const bool res_after = a.flags & FVP_EPI_RES_AFTER_RELU;
  const int plane = plane0 + tn, y = y0 + 2 * ty, x = 2 * tx;
  const bool tile_ok = q_ok && plane < a.planes && y < a.H;
  const unsigned pix = tile_ok ? unsigned(y * W + x) : 0u;
  const unsigned cbase = tile_ok ? unsigned(plane) * a.cout : 0u;
  const unsigned ppix = unsigned((y >> 1) * (W >> 1) + tx);
  // vmcnt is in-order and counts stores: a load issued behind a store waits for the store's whole round trip, and the
  // compiler may not move loads above stores itself (dst and res are not known to be distinct).  So every residual load
  // of the lane (2 cout blocks x 4 couts x 2 rows) is issued before the first store.
  int bs = buf_sz;
  FVP_OPAQUE(bs);
  const float* const epi_s = smem + 4 + 3 * bs + (RESW ? a.cinp * CBW * 16 : 0);
  // element offset of (cout co4 + r, this lane's tile); padded couts and masked tiles read a valid address and store nothing
  const unsigned omask = (FVP_WINO_DIAG && (a.ablate & 1024)) ? 0x3ffffu : ~0u;   // (bit 1024, diagnostics: epilogue traffic stays inside 1 MB)
  auto out_off = [&](int co) { return ((cbase + (tile_ok && co < a.cout ? co : 0)) * unsigned(HW) + pix) & omask; };
  float2 r0[2][4], r1[2][4];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int co4 = co0 + wc * 32 + cb * 16 + 4 * k4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned off = out_off(co4 + r);
      if (HAS_RES) {
        if (FVP_WINO_DIAG && (a.ablate & 16)) {       // diagnostics: no residual loads
          r0[cb][r] = r1[cb][r] = make_float2(0.f, 0.f);
        } else {
          r0[cb][r] = *reinterpret_cast<const float2*>(a.res + off);
          r1[cb][r] = *reinterpret_cast<const float2*>(a.res + off + W);
        }
      }
    }
  }
  // output transform A^T M A of the 8 couts while the residual loads are in flight (the accumulators die here)
  float o[2][4][2][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
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
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        o[cb][r][0][e] = (s[0][e] + s[1][e]) + s[2][e];
        o[cb][r][1][e] = (s[1][e] - s[2][e]) - s[3][e];
      }
    }
  // ONE wait for all residual loads.  The stores below are conditional (masked tiles), so behind the first of them the
  // compiler's counter no longer knows how many younger operations are in the queue and every later use of a loaded
  // value would get a full vmcnt(0) - i.e. wait for the stores issued so far.
  // The same wait (taken by the kernels without a residual too) is what makes "stores stay in flight" safe BY
  // CONSTRUCTION: the only DMA chunk still in the in-order vmcnt queue here is the one requested at the top of this
  // unit's last chunk - the chunk the NEXT unit's first barrier has to see landed.  After vmcnt(0) it has landed, so
  // that barrier needs no vmcnt wait at all, whatever the number of store instructions hipcc emits below (round 3
  // counted them - 16 + 8 - and a miscount would have let the barrier pass early: ADVICE round 3).
  __builtin_amdgcn_sched_barrier(0);
  if (HAS_RES || (FVP_WINO_STORES_IN_FLIGHT && dma)) wait_vmcnt(0);
  __builtin_amdgcn_sched_barrier(0);
  // every P2PNet / CenterNet layer on this kernel is BN (+ residual) -> ReLU: that order gets its own copy of the loop (as
  // run-time flags the two selects per value were a quarter of the epilogue's instructions)
  auto finalize = [&](auto fast) {
    // kFast: BN (+ residual) -> ReLU and every cout of the block exists (cout % 32 == 0): one predicate (the lane's tile)
    // for all stores, no per-cout compare, no select in the addresses (masked lanes compute on plane 0 / pixel 0)
    constexpr bool kFast = decltype(fast)::value;
    float v[2][4][2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int co4 = co0 + wc * 32 + cb * 16 + 4 * k4;
      f32x4 bn[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) bn[i] = *reinterpret_cast<const f32x4*>(epi_s + i * a.coutp + co4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float b = bn[0][r], sc = bn[1][r], sh = bn[2][r];
        const float rr[2][2] = {{HAS_RES ? r0[cb][r].x : 0.f, HAS_RES ? r0[cb][r].y : 0.f},
                                {HAS_RES ? r1[cb][r].x : 0.f, HAS_RES ? r1[cb][r].y : 0.f}};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float x = bn_affine(o[cb][r][i][e], b, sc, sh);
            if (HAS_RES && (kFast || !res_after)) x += rr[i][e];
            if (kFast || relu) x = fmaxf(x, 0.0f);
            if (HAS_RES && !kFast && res_after) x += rr[i][e];
            v[cb][r][i][e] = x;
          }
        if (!kFast && tile_ok && co4 + r < a.cout &&
            (!(FVP_WINO_DIAG && (a.ablate & 32)) || v[cb][r][0][0] == 1.2345e-30f)) {   // (bit 32, diagnostics: no stores)
          const unsigned off = out_off(co4 + r);
          *reinterpret_cast<float2*>(a.dst + off) = make_float2(v[cb][r][0][0], v[cb][r][0][1]);
          *reinterpret_cast<float2*>(a.dst + off + W) = make_float2(v[cb][r][1][0], v[cb][r][1][1]);
          if (a.pool_dst)                            // fused max_pool(2,2): this lane's tile is one pooled pixel
            a.pool_dst[(cbase + co4 + r) * unsigned(HW >> 2) + ppix] =
                fmaxf(fmaxf(v[cb][r][0][0], v[cb][r][0][1]), fmaxf(v[cb][r][1][0], v[cb][r][1][1]));
        }
      }
    }
    if (kFast && tile_ok) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wc * 32 + cb * 16 + 4 * k4 + r;
          const unsigned off = (cbase + unsigned(co)) * unsigned(HW) + pix;
          *reinterpret_cast<float2*>(a.dst + off) = make_float2(v[cb][r][0][0], v[cb][r][0][1]);
          *reinterpret_cast<float2*>(a.dst + off + W) = make_float2(v[cb][r][1][0], v[cb][r][1][1]);
          if (a.pool_dst)
            a.pool_dst[(cbase + unsigned(co)) * unsigned(HW >> 2) + ppix] =
                fmaxf(fmaxf(v[cb][r][0][0], v[cb][r][0][1]), fmaxf(v[cb][r][1][0], v[cb][r][1][1]));
        }
    }
  };
  const bool fast = FVP_WINO_EPI_FAST && relu && !res_after && (a.cout & 31) == 0 && !(FVP_WINO_DIAG && (a.ablate & 32));
  if (fast) finalize(std::integral_constant<bool, true>{});
  else finalize(std::integral_constant<bool, false>{});
  // The stores above may stay in flight across the next unit's first chunk barrier: the chunk that barrier guards has
  // landed (vmcnt(0) above), so its wait is lifted to the counter's maximum - no dependence on how many stores exist.
  st_pending = (FVP_WINO_STORES_IN_FLIGHT && dma) ? 1 : 0;
  }
#if !FVP_WINO_ZERO_C
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[cb][p][r] = 0.0f;
#endif
#if FVP_WINO_TIMING
  {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long tE1 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    tacc[11] += tE1 - tE0;
  }
#endif
  u = next_unit(u + G);
  if (u >= nunits) break;
  fetch_a(0, wchunk(smem + 4 + cur_off, 0), 0);
  fetch_d(smem + 4 + cur_off, 0, WP);
  }
#if FVP_WINO_TIMING
  if (a.dbg && lane == 0) {
    tacc[7] = __builtin_readcyclecounter() - tstart;
    unsigned long long* d = a.dbg + (wave >= NWV / 2 ? 16 : 0);
    for (int i = 0; i < 15; ++i) atomicAdd(d + i, tacc[i]);
    atomicAdd(d + 15, 1ull);
  }
#endif
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
