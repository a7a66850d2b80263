// 3x3 stride-1 'same' conv as Winograd F(2x2,3x3) on the fp32 matrix cores (P2PNet's res-blocks,
// lib/models/cnns_2d.py:12-71, are >90 % of the path's FLOPs).  Its own translation unit since round 5
// (compiled with -fno-slp-vectorize: no packed-f32 VALU beside the MFMAs, MI355X_MICROARCH.md "price of one filler").
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
// Workgroup = 8 (or 4) waves = WC cout blocks (32) x WT tile blocks (16).  LDS per chunk of CC channels
// (three slots filled by the LDS-DMA two chunks ahead):
//   Xs[CC][TN][TH+2][4 + W]   zero-margin dense rows (halo reads need no masking)
//   Ws[CC][32*WC][16]         quad q of row `co` stored at quad q ^ ((co>>2)&3): the four
//                             ds_read_b128 of a lane (xi = 0..3) are bank-conflict free unpadded
// (That is the A operand.  The B operand's patch reads are NOT conflict-free: a patch starts at image x - 1 = an odd column
// of the row slot, the 16 tile lanes step by two floats and the four channel groups by a multiple of four, so an instruction's
// 64 lanes share the 32 odd banks - SQ_LDS_BANK_CONFLICT reads 2.9 cycles per LDS instruction in the 64-128-channel variants;
// DESIGN.md section 8 says why the layout stays.)
//
// Round 5 rewrite of the control structure (same arithmetic, same bits).  The round-4 kernel spilled 20-64 SGPRs in
// every variant (243 v_readlane in the 64-cout form), carried ~470 scalar compares / branches and 159 s_waitcnt per chunk
// loop (run-time DMA item counts, 64-way vmcnt ladders) and packed-f32 adds in the K loop.  Now:
//   * the number of DMA items per wave and chunk (NI input + NW weight instructions) is a template parameter: the chunk
//     body is straight-line code and every counted wait is an immediate;
//   * the per-unit part of a DMA item is 'inside the image or not' only: the lane's byte offset relative to the unit's
//     descriptor base never changes, the validity is bit 31 of that offset (fails the buffer range check -> the
//     hardware writes zeros), recomputed per unit from the lane number (a dozen vector instructions per item, no
//     exec-mask branches, nothing kept in registers for it);
//   * arguments that only the epilogue or the cursor needs are re-read from the kernarg segment where they are used
//     (fresh_args) instead of living in SGPRs across the K loop;
//   * the validity flags of the plane groups (persons) are staged in LDS: the cursor's look-ahead no longer issues a
//     global load (vmcnt) in the middle of the DMA ring;
//   * both chunks staged by the prologue are waited for before the first barrier, so the first chunk barrier of EVERY
//     unit needs no vmcnt wait (the epilogue's single vmcnt(0) covers it, see there);
//   * the epilogue's residual loads / stores are raw-buffer accesses (one per-lane offset + a running scalar row offset);
//   * CW = 1 forms (one 16-cout block per wave): quarter-size units for launches that cannot fill the chip (B = 1), and -
//     diagnostics build only, measured 10-20 % slower - the 16-wave / four-waves-per-SIMD form.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <type_traits>

#include "fvp_conv_args.h"

// Ablation switches of the kernel (FVP_CONV_ABLATE bits: 1 no DMA, 4 no MFMA, 8 no epilogue, 16 no residual loads, 32 no
// stores, 128 no chunk barrier, 256 no A reads, 512 no input transform, 1024 epilogue traffic inside 1 MB) exist only in a
// variant built with -DFVP_WINO_ABLATE=1 (tools/build_variant.sh): as run-time flags they cost the diagnostics build 43-87
// spilled VGPRs per instance (round 5: its 32-channel layers ran at half speed, which distorted every A/B made through it).
#ifndef FVP_WINO_ABLATE
#define FVP_WINO_ABLATE 0
#endif

namespace fvp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// s_waitcnt vmcnt(N) with an immediate (lgkmcnt / expcnt untouched)
template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {
  static_assert(N >= 0 && N < 64, "vmcnt");
  __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}

constexpr unsigned kWinoOOB = 0x80000000u;   // offset bit that fails the buffer range check (num_records 0x7ffffff0)

// RESW: the whole Winograd-domain weight tensor of the workgroup's cout block ([cinp][CBW][16], <= 64 KB)
// stays resident in LDS behind the three input slots (loaded once per persistent workgroup) instead of
// streaming through the slots chunk by chunk: for the 32-channel layers the weight chunks were more than
// half of the LDS-DMA traffic of a unit.
// CW = cout blocks (of 16) per wave: 2 = the 128-accumulator wave tile above (two waves per SIMD); 1 = 16 couts x 16 tiles,
// 64 accumulators, <= 128 registers: a 16-wave workgroup, FOUR waves per SIMD (round 5: more waves to cover each other's
// LDS / DMA / barrier waits, at twice the patch transforms per MFMA).  WC = wave groups along the couts.
template <int WC, int WT, int CC, int NI, bool HAS_RES, bool RESW, int CW = 2>
__global__ void __launch_bounds__(WC * WT * 64, (WC * WT == 16 ? 4 : 2)) k_conv_wino(ConvArgs a) {
  HIP_DYNAMIC_SHARED(float, smem)
  constexpr int NWV = WC * WT;                       // waves per workgroup: 8 / 16 (one workgroup per CU) or 4 (two per CU)
  constexpr int NT = NWV * 64;
  static_assert(NWV == 16 || NWV == 8 || NWV == 4, "4, 8 or 16 waves");
  static_assert(CW == 1 || CW == 2, "cout blocks per wave");
  static_assert(CC == 4 || CC == 8, "chunk");
  static_assert(NI >= 1 && NI <= 4, "input DMA rounds");
  constexpr int CBW = 16 * CW * WC;                  // couts of the workgroup
  constexpr int WCH = CC * CBW * 16;                 // floats of one weight chunk
  constexpr int WS_SZ = RESW ? 0 : WCH;              // ... streamed through a slot
  constexpr int NW = RESW ? 0 : CC * CBW * 4 / NT;   // weight DMA instructions per wave per chunk
  static_assert(RESW || (CC * CBW * 4) % NT == 0, "whole weight DMA rounds");
  constexpr int NPS = NI + NW;                       // DMA instructions per wave per chunk
  constexpr int S = CC / 4;                          // steps (4 channels) per chunk
  constexpr int XS_SZ = NI * NT * 4;                 // input slot: NI rounds of one 16-byte item per thread (floats)
  constexpr int BUF_SZ = XS_SZ + WS_SZ;              // one ring slot
  constexpr bool kDiag = FVP_WINO_ABLATE != 0;       // ablation switches: a variant build only (see FVP_WINO_ABLATE)
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int k4 = lane >> 4, l15 = lane & 15;
  const int wc = wave % WC, wt = wave / WC;
  const int W = a.W, THp = a.TH + 2, WP = W + 4;
  const int plane_sz = THp * WP;
  const int CS = a.TN * plane_sz;
  const int ablate = kDiag ? a.ablate : 0;
  const bool dma = !(ablate & 1);

  // ---- LDS map (floats): [4 pad][3 ring slots][resident weights][bias | scale | shift][plane-group flags]
  const int epi_off = 4 + 3 * BUF_SZ + (RESW ? a.cinp * CBW * 16 : 0);
  const unsigned char* const vflag = reinterpret_cast<const unsigned char*>(smem + epi_off + 3 * a.coutp);

  // Persistent workgroups: unit u = (plane group, row band, cout block); workgroup b walks
  // u = b, b + G, b + 2G, ... (G = gridDim.x <= number of CUs).  The chunk stream (DMA two chunks
  // ahead) runs across unit boundaries, so a unit's first chunks land while the previous unit is
  // still computing and there is no workgroup relaunch between tiles.  Units of invalid plane groups
  // (persons below the score threshold) are skipped: flags from global memory while the prologue runs,
  // from their LDS copy afterwards (a global load inside the K loop would sit in the vmcnt queue of the ring).
  const int G = gridDim.x, nunits = a.nunits;
  auto next_unit = [&](int u, auto from_lds) {
    if (a.nflags > 0) {
      while (u < nunits) {
        const int pg = fdiv_nb(fdiv_nb(u, a.m_ys), a.m_ty);
        const int f = fdiv_nb(pg, a.m_vd);
        const int ok = decltype(from_lds)::value ? int(vflag[f]) : int(a.plane_valid[f]);
        if (__builtin_amdgcn_readfirstlane(ok)) break;
        u += G;
      }
    }
    return u;
  };
  using LdsFlags = std::integral_constant<bool, true>;
  using GlobalFlags = std::integral_constant<bool, false>;
  int u = next_unit(int(blockIdx.x), GlobalFlags{});
  if (u >= nunits) return;

  // this lane's 2x2 output tile inside the workgroup tile: TN planes x TR rows x tpr tiles; lanes beyond that
  // product (row lengths that do not divide 16*WT) compute on tile 0's data and store nothing
  const int q0 = wt * 16 + l15;
  int poff;
  {
    const bool q_ok = q0 < a.TN * a.tpp;
    const int q = q_ok ? q0 : 0;
    const int tn = fdiv_nb(q, a.m_tpp), trem = q - tn * a.tpp;
    const int ty = fdiv_nb(trem, a.m_tpr), tx = trem - ty * a.tpr;
    // LDS row 0 of the tile is image row y0 - 1; column 4 of a row slot is image x = 0
    poff = tn * plane_sz + 2 * ty * WP + 3 + 2 * tx + k4 * CS;
  }
  const int swz = (l15 >> 2) & 3;
  // A operand of quad xi: float index ((row * 4 + (xi ^ swz)) * 4 = a0 ^ (xi << 2), a0 = row * 16 + swz * 4 (cout block cb adds
  // 16 rows = 256 floats).  The two-block form keeps the four offsets in registers; the one-block form (128 registers in all)
  // keeps a0 and pays three XORs per fetch.
  const int a0 = (k4 * CBW + wc * (16 * CW) + l15) * 16 + swz * 4;
  int aoff[4];
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) aoff[xi] = a0 ^ (xi << 2);
  const float* const wres = smem + 4 + 3 * BUF_SZ;   // RESW: resident weights [cinp][CBW][16]
  // first weight float of chunk k living in slot `slot` (streamed) or in the resident copy
  auto wchunk = [&](const float* slot, int k) { return RESW ? wres + k * WCH : slot + XS_SZ; };

  f32x4 acc[CW][16];

  const int HW = a.H * W;
  const int nchunks = a.cinp / CC;

  // ---- LDS-DMA items.  Input item j of this lane: quad `qd` of row `row` of the slot (rows = channel-major, then
  // plane, then row of the band; quad 0 is the left zero margin; the item behind the last row is the zero quad the
  // halo reads of the last row run into).  Its byte offset from the unit's descriptor base (plane group's first plane,
  // channel 0 of the chunk, one row above the band) is unit-independent; whether it lies inside the image depends on
  // the unit's band (top / bottom rows) and plane group (last, partial one).  Bit 31 = outside: the buffer range check
  // fails and the hardware writes zeros to LDS (tools/micro/buflds.hip) - no zero page, no select, no vector
  // instruction per chunk.  The item's (row, plane) is recomputed from the lane number when the cursor enters a unit
  // (a dozen vector instructions per item and unit) rather than kept in a register.
  unsigned voff[NI];
  // (row in band, plane in group) of input item j of this lane and whether the item can lie inside the image at all
  auto item_pos = [&](int j, int& ry, int& n, bool& inside, int& ci, int& qd) {
    const int qpr = (W >> 2) + 1;
    const int rows_per_ch = a.TN * THp;
    const int nin = CC * rows_per_ch * qpr + 1;      // + the zero quad behind the last row
    int ln = lane;
    FVP_OPAQUE_V(ln);                                // (recomputed where it is used: per unit, not kept in registers)
    const int it = (wave + NWV * j) * 64 + ln;
    const int row = fdiv_nb(it, a.m_qpr);
    qd = it - row * qpr;
    ci = fdiv_nb(row, a.m_rpc);
    const int rem = row - ci * rows_per_ch;
    n = fdiv_nb(rem, a.m_thp);
    ry = rem - n * THp;
    inside = it < nin && qd > 0 && ci < CC;
  };
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    int ry, n, ci, qd;
    bool inside;
    item_pos(j, ry, n, inside, ci, qd);
    voff[j] = inside ? unsigned((n * a.cin + ci) * HW + ry * W + 4 * (qd - 1)) * 4u : 0u;
    __builtin_amdgcn_sched_barrier(0);
  }
  // this lane's weight item j: channel ci0 + j * DCI of the chunk, quad qd0 of the cout block's row
  constexpr int DCI = NT / (CBW * 4);
  const unsigned woffb = (unsigned(t / (CBW * 4)) * unsigned(a.coutp) * 16u + 4u * unsigned(t % (CBW * 4))) * 4u;
  const unsigned wdj4 = unsigned(DCI) * unsigned(a.coutp) * 64u;        // bytes between weight items j and j + 1
  const unsigned in_step4 = unsigned(CC) * unsigned(HW) * 4u;           // bytes per chunk: input, weights
  const unsigned w_step4 = unsigned(CC) * unsigned(a.coutp) * 64u;
  const unsigned lds0 = FVP_LDS_BYTE_ADDRESS(smem) + 16u + unsigned(wave) * 1024u;   // this wave's first item of slot 0
  i32x4 rs_in = {0, 0, 0x7ffffff0, 0x00020000}, rs_w = {0, 0, 0x7ffffff0, 0x00020000};   // raw buffers, stride 0
  auto set_base = [](i32x4& rs, const float* p) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    rs[0] = __builtin_amdgcn_readfirstlane(int(unsigned(b)));
    rs[1] = __builtin_amdgcn_readfirstlane(int(unsigned(b >> 32) & 0xffffu));
  };
  int su = u, sk = 0;                                // DMA cursor (unit su, chunk sk)
  auto enter_unit = [&](int su_) {
    const KArgsPtr ka = FVP_FRESH_ARGS(a);
    const int ys = ka->ysplit, tys = ka->tiles_y;
    const int st = fdiv_nb(su_, ka->m_ys), sy = su_ - st * ys;
    const int spg = fdiv_nb(st, ka->m_ty), sty = st - spg * tys;
    const int splane0 = spg * ka->TN, sy0 = sty * ka->TH;
    const int H = ka->H, planes = ka->planes;
    // row 0 of a slot is image row sy0 - 1: the descriptor starts one row above the band so that offsets are >= 0
    set_base(rs_in, ka->src + size_t(splane0) * ka->cin * HW + sy0 * W - W);
    set_base(rs_w, ka->wts + size_t(sy) * (CBW * 16));
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int ry, n, ci, qd;
      bool inside;
      item_pos(j, ry, n, inside, ci, qd);
      const bool ok = inside && unsigned(sy0 + ry - 1) < unsigned(H) && splane0 + n < planes;
      voff[j] = (voff[j] & 0x7fffffffu) | (ok ? 0u : kWinoOOB);
      __builtin_amdgcn_sched_barrier(0);             // one item at a time: the items' temporaries must not pile up (16-wave form: 128 registers)
    }
  };
  // every wave issues exactly NPS DMA instructions per chunk (counted s_waitcnt vmcnt below)
  auto stage = [&](int k, int slot) {
    const unsigned la0 = lds0 + unsigned(slot) * unsigned(BUF_SZ * 4);
    const unsigned so_in = unsigned(k) * in_step4;
#pragma unroll
    for (int j = 0; j < NI; ++j) asm_buffer_load_lds16(la0 + unsigned(NWV * j) * 1024u, voff[j], rs_in, so_in);
    const unsigned so_w = unsigned(k) * w_step4;
#pragma unroll
    for (int j = 0; j < NW; ++j)
      asm_buffer_load_lds16(la0 + unsigned(XS_SZ * 4 + NWV * j * 1024), woffb, rs_w, so_w + unsigned(j) * wdj4);
  };
  enter_unit(su);
  // stage the cursor's chunk into `slot` and advance; false once every unit has been requested
  auto stage_next = [&](int slot, auto from_lds) {
    if (su >= nunits) return false;
    stage(sk, slot);
    if (++sk == nchunks) {
      sk = 0;
      su = next_unit(su + G, from_lds);
      if (su < nunits) enter_unit(su);
    }
    return true;
  };

  // ---- operand fetch / transform / MFMA building blocks
  float4 av[CW][4];
  float d[4][4];                                     // the lane's 4x4 patch
  float v[4][4];                                     // V[xi][nu] = B^T d B
  auto fetch_a = [&](int cb, const float* wbase, int s) {     // wbase = weights of the chunk, s = step in chunk
    if (kDiag && (ablate & 256)) return;                      // diagnostics: no A-operand reads
    int a0o = a0;
    if (CW == 1) FVP_OPAQUE_V(a0o);                  // (recomputed per fetch: hipcc would hoist the four offsets again)
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
      av[cb][xi] = *reinterpret_cast<const float4*>(wbase + (CW == 1 ? (a0o ^ (xi << 2)) : aoff[xi]) + (s * 4 * CBW * 16 + cb * 256));
  };
  auto fetch_d = [&](const float* base, int s, int wp) {
    const float* xs = base + poff + s * 4 * CS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* row = xs + r * wp;
      const float2 m = *reinterpret_cast<const float2*>(row + 1);       // 8-byte aligned: column 4 + 2*tx
      d[r][0] = row[0];
      d[r][1] = m.x;
      d[r][2] = m.y;
      d[r][3] = row[3];
    }
  };
  // V = B^T d B as 32 plain scalar adds (rows, then columns); the patch registers die in the row pass
  auto transform = [&](float (&tr)[4][4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      tr[0][c] = d[0][c] - d[2][c];
      tr[1][c] = d[1][c] + d[2][c];
      tr[2][c] = d[2][c] - d[1][c];
      tr[3][c] = d[1][c] - d[3][c];
    }
  };
  auto columns = [&](const float (&tr)[4][4]) {
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      v[xi][0] = tr[xi][0] - tr[xi][2];
      v[xi][1] = tr[xi][1] + tr[xi][2];
      v[xi][2] = tr[xi][2] - tr[xi][1];
      v[xi][3] = tr[xi][1] - tr[xi][3];
    }
  };
  // first = the unit's first step: the MFMAs take the constant 0 as C (clearing the 128 accumulator registers between
  // units cost 128 vector moves per wave and unit - and on this part a vector instruction of either wave of a SIMD is
  // matrix time lost: tools/micro/coexec.hip)
  auto mfma16 = [&](int cb, auto firstc) {
    if (kDiag && (ablate & 4)) return;                        // diagnostics: no MFMA
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      const float aw[4] = {av[cb][xi].x, av[cb][xi].y, av[cb][xi].z, av[cb][xi].w};
#pragma unroll
      for (int nu = 0; nu < 4; ++nu)
        acc[cb][4 * xi + nu] = __builtin_amdgcn_mfma_f32_16x16x4f32(
            aw[nu], v[xi][nu], decltype(firstc)::value ? z : acc[cb][4 * xi + nu], 0, 0, 0);
    }
  };

  // ---- prologue: resident weights, BN vectors, validity flags, the first two chunks
  if (RESW) {                                        // cinp * CBW * 4 quads, NT per round
    const int rounds = (a.cinp * CBW * 4) / NT;
    for (int j = 0; j < rounds; ++j)
      asm_global_load_lds16(a.wts + size_t((wave + NWV * j) * 64 + lane) * 4,
                            __builtin_amdgcn_readfirstlane(FVP_LDS_BYTE_ADDRESS(smem) +
                                                           4u * unsigned(4 + 3 * BUF_SZ + (wave + NWV * j) * 256)));
  }
  // bias | scale | shift of every cout, [3][coutp], behind the slots (and the resident weights): the epilogue reads them
  // with ds_read (lgkmcnt).  As global loads they sat in the in-order vmcnt queue behind the previous cout's stores.
  {
    float* const e = const_cast<float*>(smem) + epi_off;
    for (int i = t; i < 3 * a.coutp; i += NT) e[i] = a.epi[i];
    unsigned char* const f = const_cast<unsigned char*>(vflag);
    for (int i = t; i < a.nflags; i += NT) f[i] = a.plane_valid[i];
  }
  if (dma) {
    stage_next(0, GlobalFlags{});
    stage_next(1, GlobalFlags{});
  }
  wait_vmcnt_imm<0>();                               // both chunks (and the resident weights) have landed: see chunk_barrier
  __syncthreads();

  int cur = 0;                                       // ring slot of the chunk being consumed
  auto slot_ptr = [&](int slot) { return smem + 4 + slot * BUF_SZ; };
  fetch_a(0, wchunk(slot_ptr(0), 0), 0);
  fetch_d(slot_ptr(0), 0, WP);
  while (true) {
    // The chunk body exists twice: the unit's first chunk (its first step's MFMAs take C = 0, its barrier needs no vmcnt
    // wait) and every other one.
    auto chunk = [&](int k, auto firstc) {
      constexpr bool kFirst = decltype(firstc)::value;
      const int nxt = cur == 2 ? 0 : cur + 1;
      const int nn = nxt == 2 ? 0 : nxt + 1;
      const bool more = dma && stage_next(nn, LdsFlags{});
      const float* const cs = slot_ptr(cur);
      const float* const ns = slot_ptr(nxt);
      int wp = WP;
      FVP_OPAQUE(wp);
      if constexpr (CW == 2) {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        // ---- half-step 0: patch transform, cout block 0
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): av[0] and the patch have landed
        __builtin_amdgcn_sched_barrier(0);
        fetch_a(1, wchunk(cs, k), s);
        __builtin_amdgcn_sched_barrier(0);           // issue the reads now: left alone hipcc sinks them below the MFMAs
        float tr[4][4];
        if (kDiag && (ablate & 512)) {               // diagnostics: no input transform
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[r][c] = d[r][c];
        } else {
          transform(tr);
        }
        if (s + 1 < S) fetch_d(cs, s + 1, wp);       // the patch registers are dead: refill for the next step
        __builtin_amdgcn_sched_barrier(0);
        if (!(kDiag && (ablate & 512))) columns(tr);
        if (kFirst && s == 0) mfma16(0, std::integral_constant<bool, true>{});
        else mfma16(0, std::integral_constant<bool, false>{});
        __builtin_amdgcn_sched_barrier(0);
        // ---- half-step 1: cout block 1; the last one of a chunk crosses into the next slot
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < S) {
          fetch_a(0, wchunk(cs, k), s + 1);
        } else {
          // All reads of this slot are complete (lgkmcnt above); once every wave is here the slot may be overwritten by
          // the DMA of chunk g+3, and chunk g+1 has landed for everybody: every wave waits for ITS items of chunk g+1
          // (vmcnt(NPS): only the NPS instructions of chunk g+2 may still be in flight) before the barrier.  A unit's
          // first chunk needs no wait: chunk g+1 was requested before the previous unit's epilogue, whose vmcnt(0) (or the
          // prologue's) it has passed - and the epilogue's stores may stay in flight across this barrier.
          if (!kFirst && dma) {
            if (more) wait_vmcnt_imm<NPS>();
            else wait_vmcnt_imm<0>();
          }
          if (!(kDiag && (ablate & 128))) __builtin_amdgcn_s_barrier();     // plain barrier: no fence, the counters are ours
          if (k + 1 < nchunks) {                     // (a unit's last chunk: the epilogue needs the registers)
            fetch_a(0, wchunk(ns, k + 1), 0);
            fetch_d(ns, 0, wp);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kFirst && s == 0) mfma16(1, std::integral_constant<bool, true>{});
        else mfma16(1, std::integral_constant<bool, false>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      } else {
        // ---- one cout block per wave (four waves per SIMD): per step  wait - transform - 16 MFMAs - request the next
        // step's operands.  Nothing is double-buffered (patch, V and the next patch share 16 registers: with the next
        // patch in flight beside V the wave would need 128 registers before anything else): the other three waves of the
        // SIMD cover the LDS latency.
#pragma unroll
        for (int s = 0; s < S; ++s) {
          __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): av[0] and the patch have landed
          __builtin_amdgcn_sched_barrier(0);
          {
            float tr[4][4];
            transform(tr);
            columns(tr);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (kFirst && s == 0) mfma16(0, std::integral_constant<bool, true>{});
          else mfma16(0, std::integral_constant<bool, false>{});
          __builtin_amdgcn_sched_barrier(0);
          if (s + 1 < S) {
            fetch_a(0, wchunk(cs, k), s + 1);
            fetch_d(cs, s + 1, wp);
          } else {
            // (chunk barrier: see the two-block form above; no LDS read of this slot is pending here)
            if (!kFirst && dma) {
              if (more) wait_vmcnt_imm<NPS>();
              else wait_vmcnt_imm<0>();
            }
            __builtin_amdgcn_s_barrier();
            if (k + 1 < nchunks) {
              fetch_a(0, wchunk(ns, k + 1), 0);
              fetch_d(ns, 0, wp);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      cur = nxt;
    };
    chunk(0, std::integral_constant<bool, true>{});
    for (int k = 1; k < nchunks; ++k) chunk(k, std::integral_constant<bool, false>{});

    // ---- unit finished: output transform + epilogue, then the next unit of this workgroup.  Everything the epilogue
    // needs from the launch arguments is read here, not kept in SGPRs across the K loop.
    if (!(ablate & 8)) {
      const KArgsPtr ka = FVP_FRESH_ARGS(a);
      const int ys = ka->ysplit, tys = ka->tiles_y;
      const int ut = fdiv_nb(u, ka->m_ys), uy = u - ut * ys;
      const int pg = fdiv_nb(ut, ka->m_ty), ty_i = ut - pg * tys;
      const int plane0 = pg * ka->TN, y0 = ty_i * ka->TH, co0 = uy * CBW;
      const int cout = ka->cout, coutp = ka->coutp;
      const int flags = ka->flags;
      float* const dst = ka->dst;
      const float* const res = ka->res;
      float* const pool_dst = ka->pool_dst;
      // per lane: tile (plane, y, x), 8 (or 4) couts; the tile coordinates are recomputed from the lane's tile number here
      // instead of living in three registers across the K loop
      int le = lane;
      FVP_OPAQUE_V(le);
      const int k4 = le >> 4;
      const int qe = wt * 16 + (le & 15);
      const bool q_ok = qe < ka->TN * ka->tpp;
      const int qq = q_ok ? qe : 0;
      const int tn = fdiv_nb(qq, ka->m_tpp), trem = qq - tn * ka->tpp;
      const int ty = fdiv_nb(trem, ka->m_tpr), tx = trem - ty * ka->tpr;
      const bool relu = flags & FVP_EPI_RELU;
      const bool res_after = flags & FVP_EPI_RES_AFTER_RELU;
      const int plane = plane0 + tn, y = y0 + 2 * ty, x = 2 * tx;
      const bool tile_ok = q_ok && plane < ka->planes && y < ka->H;
      const unsigned ppix = unsigned((y >> 1) * (W >> 1) + tx);
      const float* const epi_s = smem + epi_off;
      // Epilogue addressing (round 5): raw descriptors of the unit's first plane (output, residual, pooled output) in SGPRs,
      // ONE per-lane byte offset - cout 4 k4 of the wave's block, the lane's tile; bit 31 (range check: loads return 0,
      // stores are dropped) for masked tiles - and a scalar byte offset per (cout, row): no address arithmetic and no
      // predicate per access (the per-access 64-bit pointer adds were ~15 % of the epilogue's vector instructions).
      // vmcnt is in-order and counts stores: every residual load of the lane is issued before the first store.
      const bool fast = relu && !res_after;            // (every cout of the block exists: the planner takes cout % 32 == 0 only)
      unsigned HW4 = unsigned(HW) * 4u, W4 = unsigned(W) * 4u, HWq4 = unsigned(HW >> 2) * 4u;
      FVP_OPAQUE(HW4);                                 // (the 8-16 scalar row offsets are formed here, per unit: hoisted out
      FVP_OPAQUE(W4);                                  // of the K loop as multiples of the loop-invariant HW they spilled)
      FVP_OPAQUE(HWq4);
      // (one descriptor register set, re-pointed per phase - residual loads, output stores, pooled stores: three sets at
      // once do not fit the SGPR file beside the K loop's state)
      i32x4 rs = {0, 0, 0x7ffffff0, 0x00020000};
      if (HAS_RES) set_base(rs, res + size_t(plane0) * cout * HW);
      const unsigned lco = unsigned(tn * cout + co0 + wc * (16 * CW) + 4 * k4);     // this lane's first cout, as a row of the unit
      const unsigned omask = (kDiag && (ablate & 1024)) ? 0x3ffffu : 0x7fffffffu;   // (bit 1024, diagnostics: epilogue traffic stays inside 1 MB)
      const unsigned voff0 = tile_ok ? ((lco * unsigned(HW) + unsigned(y * W + x)) * 4u) & omask : kWinoOOB;
      const unsigned voffp = tile_ok ? (lco * unsigned(HW >> 2) + ppix) * 4u : kWinoOOB;
      fvp_f32x2 r0[CW][4], r1[CW][4];
      if (HAS_RES && !(kDiag && (ablate & 16))) {      // (bit 16, diagnostics: no residual loads)
        unsigned so = 0;                               // scalar row offset (cb * 16 + r) * HW4, advanced as it is used: formed
#pragma unroll                                         // up front, the 16 offsets of a lane's accesses are 16 more SGPRs
        for (int cb = 0; cb < CW; ++cb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            r0[cb][r] = asm_buffer_load_f32x2(voff0, rs, so);
            r1[cb][r] = asm_buffer_load_f32x2(voff0, rs, so + W4);
            so += HW4;
            FVP_OPAQUE(so);
          }
          so += 12u * HW4;
        }
      } else {
#pragma unroll
        for (int cb = 0; cb < CW; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) r0[cb][r] = r1[cb][r] = fvp_f32x2{0.f, 0.f};
      }
      // output transform A^T M A of the 8 couts while the residual loads are in flight (the accumulators die here)
      float o[CW][4][2][2];
#pragma unroll
      for (int cb = 0; cb < CW; ++cb)
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
      // that barrier needs no vmcnt wait at all, whatever the number of store instructions hipcc emits below.
      __builtin_amdgcn_sched_barrier(0);
      wait_vmcnt_imm<0>();
      __builtin_amdgcn_sched_barrier(0);
      // every P2PNet / CenterNet layer on this kernel is BN (+ residual) -> ReLU: that order gets its own copy of the loop (as
      // run-time flags the two selects per value were a quarter of the epilogue's instructions)
      auto finalize = [&](auto fastc) {
        // kFast: BN (+ residual) -> ReLU, the order of every P2PNet / CenterNet layer on this kernel
        constexpr bool kFast = decltype(fastc)::value;
        set_base(rs, dst + size_t(plane0) * cout * HW);
        float pm[CW][4];                             // fused max_pool(2,2): this lane's tile is one pooled pixel
        unsigned sso = 0;
#pragma unroll
        for (int cb = 0; cb < CW; ++cb) {
          const int co4 = co0 + wc * (16 * CW) + cb * 16 + 4 * k4;
          f32x4 bn[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) bn[i] = *reinterpret_cast<const f32x4*>(epi_s + i * coutp + co4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float b = bn[0][r], sc = bn[1][r], sh = bn[2][r];
            const float rr[2][2] = {{r0[cb][r].x, r0[cb][r].y}, {r1[cb][r].x, r1[cb][r].y}};
            float vv[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                float xv = bn_affine(o[cb][r][i][e], b, sc, sh);
                if (HAS_RES && (kFast || !res_after)) xv += rr[i][e];
                if (kFast || relu) xv = fmaxf(xv, 0.0f);
                if (HAS_RES && !kFast && res_after) xv += rr[i][e];
                vv[i][e] = xv;
              }
            pm[cb][r] = fmaxf(fmaxf(vv[0][0], vv[0][1]), fmaxf(vv[1][0], vv[1][1]));
            if (kDiag && (ablate & 32) && vv[0][0] != 1.2345e-30f) { sso += HW4; continue; }   // (bit 32, diagnostics: no stores)
            asm_buffer_store_f32x2(fvp_f32x2{vv[0][0], vv[0][1]}, voff0, rs, sso);
            asm_buffer_store_f32x2(fvp_f32x2{vv[1][0], vv[1][1]}, voff0, rs, sso + W4);
            sso += HW4;
            FVP_OPAQUE(sso);
          }
          sso += 12u * HW4;
        }
        if (pool_dst && !(kDiag && (ablate & 32))) {
          set_base(rs, pool_dst + size_t(plane0) * cout * (HW >> 2));
#pragma unroll
          for (int cb = 0; cb < CW; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              asm_buffer_store_f32(pm[cb][r], voffp, rs, unsigned(cb * 16 + r) * HWq4);
        }
      };
      if (fast) finalize(std::integral_constant<bool, true>{});
      else finalize(std::integral_constant<bool, false>{});
    } else {
      wait_vmcnt_imm<0>();                           // (diagnostics, no epilogue: the ring invariant still needs the drain)
    }
    u = next_unit(u + G, LdsFlags{});
    if (u >= nunits) break;
    fetch_a(0, wchunk(slot_ptr(cur), 0), 0);
    fetch_d(slot_ptr(cur), 0, WP);
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

// ---------------------------------------------------------------------------------------------------------------------
// host side
static const size_t kWinoLdsBudget = env_size("FVP_WINO_LDS_KB", 152) * 1024;
static const int kWinoGeneric = int(env_size("FVP_WINO_GENERIC", 0));
constexpr int kWinoMaskedMinW = 40;     // narrowest non-power-of-two row that takes masked Winograd tiles (netspec.WINO_MASKED_MIN_W)
static const int kWinoWC1 = int(env_size("FVP_WINO_WC1", 0));        // diagnostics: 32-cout blocks for every layer
// 4-wave workgroups (32 couts x 64 tiles, <= 78 KB of LDS, two per CU) instead of one 8-wave workgroup per CU:
// half-size work units.  Slower per FLOP when the launch has plenty of units (more LDS-DMA traffic per MFMA), but
// a small batch (B = 1: 30 planes) has only 60-120 full-size units for 256 CUs.  FVP_WINO_HALF: 0 = automatic
// (half-size units when the full-size ones cannot fill the CUs), 1 = always, 2 = never.  Both tilings perform the
// same arithmetic in the same order, so the result does not depend on the choice (i.e. on the batch).
static const int kWinoHalf = int(env_size("FVP_WINO_HALF", 0));
static const int kWinoNoResW = int(env_size("FVP_WINO_NO_RESW", 0)); // diagnostics: stream the weights of the 32-channel layers too
static const int kWinoAblate = int(env_size("FVP_CONV_ABLATE", 0));
#ifndef FVP_WINO_W16_DEFAULT
#define FVP_WINO_W16_DEFAULT 0
#endif

int persistent_workgroups() {
  static int n = 0;
  if (!n) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    n = int(env_size("FVP_WINO_WGS", size_t(cus)));
  }
  return n;
}

// Shapes the Winograd kernel covers: 3x3, even H, W a power of two in [8, 64*4] with W/2 dividing
// a wave's 32 tiles or vice versa.  Decided from the layer SHAPE only (never from the number of
// planes), so a frame's result does not depend on the batch it is computed in.
static bool wino_tiling(int h, int w, int cinp, int coutp, int* WC, int* WT, int* TN, int* TR, bool half = false) {
  if (h < 2 || (h & 1) || w < 8 || (w & 3) || (coutp != 32 && coutp % 64 != 0) || cinp % 4 != 0) return false;
  // Maps whose rows do not divide the workgroup tile (CenterNet's 80x80 / 40x40 / 20x20 levels) run with masked tiles.
  // Until round 5 they stayed on the direct kernel (582 vs 564 us for CenterNet at B = 8 in round 3: launch-latency bound
  // either way).  Since round 5's rewrite of this kernel the masked form wins on the 80- and 40-wide levels at every batch
  // (round 6, same box: 3x3 layers 23 -> 14.7 / 19.5 -> 13.8 us at B = 8, 11.5 -> 8 / 18.5 -> 13.2 us at B = 1; CenterNet
  // 496 -> ~420 us per pass at B = 8, 379 -> ~330 us at B = 1), while the 20-wide level is faster on the split-K direct form
  // at B = 1 (17 vs 22 us; one or two workgroups of tiles).  So: rows of >= 40 columns take masked tiles, narrower ones the
  // direct kernel - a SHAPE rule, never the batch.  The detection map moves by fp32 rounding only: the GPU suite (top-k
  // indices, proposal centres and valid flags exact on every golden and sweep) is green with it and the joints do not change
  // by a bit (the joint stage reads nothing of CenterNet's float values).  FVP_WINO_GENERIC=1 (diagnostics build) takes
  // every width.
  if ((w & (w - 1)) && w < kWinoMaskedMinW && !kWinoGeneric) return false;
  *WC = (coutp == 32 || kWinoWC1 || half) ? 1 : 2;
  *WT = (half ? 4 : 8) / *WC;
  // a unit = TN planes x TR tile rows x (w/2) tiles <= the workgroup's 16*WT tiles; tiles beyond that product
  // (maps whose row length does not divide the workgroup tile: 80x80, 40x40, 20x20) are masked lanes
  const int tpr = w / 2, TT = 16 * *WT, rows = h / 2;
  if (tpr > TT) return false;
  if (rows * tpr >= TT) {
    *TN = 1;
    *TR = TT / tpr;
  } else {
    *TN = TT / (rows * tpr);
    *TR = rows;
  }
  return true;
}

bool wino_shape_ok(int h, int w, int cinp, int coutp) {
  int WC, WT, TN, TR;
  return wino_tiling(h, w, cinp, coutp, &WC, &WT, &TN, &TR);
}

template <int WC, int WT, int CC, int NI, bool RES, bool RESW, int CW = 2>
static int launch_wino3(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static LdsOptIn optin;
  auto k = &k_conv_wino<WC, WT, CC, NI, RES, RESW, CW>;
  if (int e = lds_opt_in(optin, reinterpret_cast<const void*>(k), 160 * 1024)) return e;
  hipLaunchKernelGGL(k, grid, dim3(WC * WT * 64), lds, s, a);
  return launch_status();
}
template <int WC, int WT, int CC, bool RESW, int CW = 2>
static int launch_wino2(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  const bool res = a.flags & FVP_EPI_RES;
  switch (a.wino_ni) {
    case 1: return res ? launch_wino3<WC, WT, CC, 1, true, RESW, CW>(a, grid, lds, s) : launch_wino3<WC, WT, CC, 1, false, RESW, CW>(a, grid, lds, s);
    case 2: return res ? launch_wino3<WC, WT, CC, 2, true, RESW, CW>(a, grid, lds, s) : launch_wino3<WC, WT, CC, 2, false, RESW, CW>(a, grid, lds, s);
    case 3: return res ? launch_wino3<WC, WT, CC, 3, true, RESW, CW>(a, grid, lds, s) : launch_wino3<WC, WT, CC, 3, false, RESW, CW>(a, grid, lds, s);
    case 4: return res ? launch_wino3<WC, WT, CC, 4, true, RESW, CW>(a, grid, lds, s) : launch_wino3<WC, WT, CC, 4, false, RESW, CW>(a, grid, lds, s);
  }
  return FVP_ELIMIT;
}
#if FVP_DIAG
// 16-wave form (one 16-cout block per wave, four waves per SIMD): CC = 8, one or two input DMA rounds
template <int WC, int WT, bool RESW>
static int launch_wino16(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  const bool res = a.flags & FVP_EPI_RES;
  if (a.wino_ni == 1)
    return res ? launch_wino3<WC, WT, 8, 1, true, RESW, 1>(a, grid, lds, s) : launch_wino3<WC, WT, 8, 1, false, RESW, 1>(a, grid, lds, s);
  if (a.wino_ni == 2)
    return res ? launch_wino3<WC, WT, 8, 2, true, RESW, 1>(a, grid, lds, s) : launch_wino3<WC, WT, 8, 2, false, RESW, 1>(a, grid, lds, s);
  return FVP_ELIMIT;
}
#endif
template <int WC, int WT>
static int launch_wino(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s, bool resw) {
  if (WC == 1 && WT == 8 && resw && a.CC == 8) return launch_wino2<1, 8, 8, true>(a, grid, lds, s);
  if (a.CC == 8) return launch_wino2<WC, WT, 8, false>(a, grid, lds, s);
  return launch_wino2<WC, WT, 4, false>(a, grid, lds, s);
}

int wino_plan_and_launch(const FvpConvOp& op, ConvArgs a, const float* params, int planes, hipStream_t s) {
  int WC, WT, TN, TR;
  // whole 32-cout blocks only, and no padded couts: the epilogue has no per-cout predicate and its descriptor spans the
  // plane group, so a block of padding couts (cout = 96, coutp = 128) would be written into the next plane
  if (op.cout % 32 != 0 || op.cout != op.coutp) return FVP_EINVAL;
  if (!wino_tiling(op.h, op.w, op.cinp, op.coutp, &WC, &WT, &TN, &TR, kWinoHalf == 1)) return FVP_EINVAL;
  if (kWinoHalf == 0) {
    // full-size units: (rows bands) x (plane groups) x (cout blocks); switch to half-size ones when they cannot fill the CUs
    const long units = long(ceil_div(op.h / 2, TR)) * ceil_div(planes, TN) * (op.coutp / (32 * WC));
    int wc, wt, tn, tr;
    if (units < persistent_workgroups() && wino_tiling(op.h, op.w, op.cinp, op.coutp, &wc, &wt, &tn, &tr, true)) {
      WC = wc; WT = wt; TN = tn; TR = tr;
    }
  }
  // 16-wave form (one 16-cout block per wave, four waves per SIMD) for full-size units: instantiated in the diagnostics
  // build only (FVP_WINO_W16=1).  Measured in round 5, same box, P2PNet at B = 8, per launch: 64 -> 64 @32x32 98 -> 109 us,
  // 128 -> 128 @16x16 85 -> 99 us, 32 -> 32 @64x64 with residual 119 -> 143 us, whole pass 1 916 -> 2 113 us, pipelined
  // 3 150 -> 2 960 frames/s: twice the patch transforms and A reads per MFMA cost more than four waves per SIMD hide.
#if FVP_DIAG
  static const int kW16 = int(env_size("FVP_WINO_W16", FVP_WINO_W16_DEFAULT));
#else
  constexpr int kW16 = 0;
#endif
  const bool w16 = kW16 && WC * WT == 8 && op.cinp % 8 == 0;
  // Quarter-size units (round 5): when even the half-size units (32 couts x 64 tiles) fill less than a quarter of the
  // chip's 512 slots - B = 1: the 128-channel 16x16 layers have 120 of them - a 4-wave workgroup takes 16 couts x 64 tiles
  // (one 16-cout block per wave: the accumulation chain of a (cout, tile) is the same, so the bits are) and the launch's
  // critical path, one unit, halves: 30 -> 23 us per launch, B = 1 serial 795 -> 822 frames/s (1.26 -> 1.22 ms) at 1 915 -> 1 883
  // with four batches in flight.  Applied to every layer below half the slots (240 units: the 32- / 64-channel layers at
  // B = 1 too) it reaches 825 serial but 1 845 in flight and costs B = 2 3 % (twice the patch transforms per MFMA on a chip that
  // IS full then): threshold 1/4.
  // FVP_WINO_QUARTER (diagnostics build): 0 = never, 1 = below a quarter of the slots (default), 2 = below half.
  static const int kQuarter = int(env_size("FVP_WINO_QUARTER", 1));
  bool quarter = false;
  if (kQuarter && !w16 && WC * WT == 4 && op.cinp % 4 == 0) {
    const long units_half = long(ceil_div(op.h / 2, TR)) * ceil_div(planes, TN) * (op.coutp / 32);
    quarter = units_half * (kQuarter == 2 ? 1 : 2) < persistent_workgroups();
  }
  const int CW = (w16 || quarter) ? 1 : 2;
  if (w16) WC *= 2;                                  // wave groups along the couts: 16 couts each
  a.ablate = kWinoAblate;
  a.wts = params + op.wino_off;
  a.TN = TN;
  a.TH = 2 * TR;
  a.TW = op.w;
  a.tiles_x = 1;
  a.tiles_y = ceil_div(op.h / 2, TR);
  a.tpp = TR * (op.w / 2);
  a.m_tpp = make_magic(a.tpp);
  a.tpr = op.w / 2;
  a.m_tpr = make_magic(a.tpr);
  a.vec = a.dma = 1;
  a.zeros = params;
  const int CBW = 16 * CW * WC;
  // channels per chunk: 8 when it divides cinp and three slots fit, else 4
  // resident weights: one 32-cout block covers all couts and [cinp][32][16] fits beside the three input slots
  const size_t resw_bytes = size_t(op.cinp) * CBW * 64;
  const size_t budget = WC * WT == 4 ? std::min<size_t>(kWinoLdsBudget, 78 * 1024) : kWinoLdsBudget;   // two workgroups per CU
  const size_t epi_bytes = size_t(3) * op.coutp * 4;     // bias | scale | shift in LDS
  // validity flags of the plane groups (one byte each, only consulted for one-plane units)
  a.nflags = (a.plane_valid && TN == 1) ? ceil_div(planes, a.valid_div) : 0;
  a.m_vd = make_magic(a.valid_div);
  const size_t flag_bytes = size_t(a.nflags + 15) & ~size_t(15);
  bool resw = CBW == 32 && WT == 8 && op.coutp == 32 && op.cinp % 8 == 0 && resw_bytes <= 64 * 1024 && !kWinoNoResW;
  auto slot_bytes = [&](int cc, int* ni, bool rw) {
    const size_t quads = size_t(cc) * TN * (a.TH + 2) * (op.w / 4 + 1) + 1;
    const size_t per_round = size_t(WC) * WT * 64;       // one 16-byte item per thread and round
    *ni = int((quads + per_round - 1) / per_round);
    return size_t(*ni) * per_round * 16 + (rw ? 0 : size_t(cc) * CBW * 64);
  };
  const size_t fixed = 64 + epi_bytes + flag_bytes;
  int CC = op.cinp % 8 == 0 ? 8 : 4, ni = 0;
  size_t slot = slot_bytes(CC, &ni, resw);
  if (resw && (3 * slot + resw_bytes + fixed > budget || ni > 4)) {
    resw = false;
    slot = slot_bytes(CC, &ni, false);
  }
  if (CC == 8 && !resw && (3 * slot + fixed > budget || ni > 4)) {
    CC = 4;
    slot = slot_bytes(CC, &ni, false);
  }
  if (3 * slot + (resw ? resw_bytes : 0) + fixed > budget || ni > 4) return FVP_ELIMIT;
  a.CC = CC;
  a.wino_ni = ni;
  if (!buf_dma_range_ok(TN, op.cin, op.h, op.w, double(op.cinp) * op.coutp * 16)) return FVP_ELIMIT;
  // the epilogue's per-lane byte offset spans the unit's TN planes of the output (bit 31 is the 'masked' flag)
  if ((double(TN) + 1.0) * op.cout * op.h * op.w * 4.0 >= 2147483648.0) return FVP_ELIMIT;
  a.m_qpr = make_magic(op.w / 4 + 1);
  a.m_rpc = make_magic(TN * (a.TH + 2));
  a.m_thp = make_magic(a.TH + 2);
  const size_t lds = 16 + 3 * slot + (resw ? resw_bytes : 0) + epi_bytes + flag_bytes;
  a.ysplit = op.coutp / CBW;
  a.nunits = a.tiles_y * ceil_div(planes, TN) * a.ysplit;
  a.m_ys = make_magic(a.ysplit);
  a.m_ty = make_magic(a.tiles_y);
  // One workgroup per slot.  FVP_WINO_BALANCED=1 (diagnostics build): ceil(units / rounds) workgroups that all do the same
  // number of units (240 instead of 256 for 240 planes), leaving 16 CUs to the other streams for the whole launch.
  static const int kBalanced = int(env_size("FVP_WINO_BALANCED", 0));
  const int slots = persistent_workgroups() * (WC * WT == 4 ? 2 : 1);
  dim3 grid(kBalanced ? ceil_div(a.nunits, ceil_div(a.nunits, slots)) : std::min(a.nunits, slots), 1, 1);
  ProfScope ps(a.nunits < slots ? FVP_K_CONV_WINO_SMALL : FVP_K_CONV_WINO, s, 2.0 * op.cin * op.cout * 9.0 * op.h * op.w * planes, 1,
               prof_level() >= 2);
#if FVP_DIAG
  if (w16) {
    if (CC != 8 || ni > 2) return FVP_ELIMIT;        // (shapes outside the 16-wave instances)
    if (CBW == 32) return resw ? launch_wino16<2, 8, true>(a, grid, lds, s) : launch_wino16<2, 8, false>(a, grid, lds, s);
    return launch_wino16<4, 4, false>(a, grid, lds, s);
  }
#endif
  if (quarter) return a.CC == 8 ? launch_wino2<1, 4, 8, false, 1>(a, grid, lds, s) : launch_wino2<1, 4, 4, false, 1>(a, grid, lds, s);
  if (WC * WT == 4) return launch_wino<1, 4>(a, grid, lds, s, false);
  return WC == 1 ? launch_wino<1, 8>(a, grid, lds, s, resw) : launch_wino<2, 4>(a, grid, lds, s, false);
}

int wino_pack(const float* weight, const FvpConvOp& op, float* params, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_wino, dim3(ceil_div(op.cinp * op.coutp, 256)), dim3(256), 0, s, weight, op.cin, op.cout,
                     op.cinp, op.coutp, params + op.wino_off);
  return launch_status();
}

}  // namespace fvp
