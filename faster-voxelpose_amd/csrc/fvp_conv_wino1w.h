// Winograd F(2x2,3x3) with ONE WAVE PER SIMD (round 4; included by fvp_conv.hip after fvp_conv_wino.h, whose building
// blocks - LDS layout, DMA ring, weight packing, output transform - it shares).  Layers with cout % 64 == 0.
//
// Why.  k_conv_wino puts two waves on every SIMD (32 couts x 16 tiles each, 128 accumulator registers) so that one
// wave's patch transform and LDS reads run under the other's MFMAs.  On this part that cover is thin: the fp32 MFMA
// and the vector ALU share the SIMD's issue (tools/micro/coexec.hip), the two waves are re-aligned by a barrier every
// chunk and then both start with their VALU phase, and the younger wave loses the arbitration on every segment
// (s_memtime stamps, DESIGN.md: matrix pipe ~54 % busy inside the K loop).  Here a workgroup is 4 waves, one per SIMD,
// and a wave owns 64 couts x 16 tiles = 4 x 16 accumulator tiles = 256 registers (the 512-register budget of a lone
// wave: accumulators in AGPRs).  Per 4-channel step a wave issues 64 MFMAs for ONE patch transform (the two-wave form:
// 32 MFMAs per transform and wave), i.e. half the vector instructions per MFMA, and the loop is software-pipelined by
// hand inside the wave - nothing depends on a partner:
//     half-step cb = 0..3 (16 MFMAs each, cout block cb):   A(cb + 1) is requested at the top of half-step cb;
//     half-step 3 additionally requests the patch of step s + 1 and runs its transform BETWEEN its MFMAs;
//     the chunk's LDS-DMA instructions are spread over the half-steps of the chunk's first step (3 per half-step);
//     the chunk barrier sits between half-steps 2 and 3 of the chunk's last step: A(cb 3) is in registers by then, so
//     no wave reads the slot afterwards, and the first patch of the next chunk is transformed under half-step 3.
// The DMA ring (3 slots, counted vmcnt), the unit walk (persistent workgroups), the resident BN vectors, the epilogue
// with all residual loads up front and stores left in flight are those of k_conv_wino.  Same arithmetic in the same
// order per output (channel order of the MFMA chain, output transform, BN / residual / ReLU): the result is bit-identical
// to k_conv_wino's (tested), so the choice of kernel cannot change a frame's result.
#pragma once

namespace fvp {

template <int CC, bool HAS_RES>
__global__ void __launch_bounds__(256, 1) k_conv_wino1w(ConvArgs a) {
  HIP_DYNAMIC_SHARED(float, smem)
  constexpr int NWV = 4, CPW = 4, CBW = 64;
  static_assert(CC == 4 || CC == 8, "chunk");
  constexpr int WCH = CC * CBW * 16;                 // floats of one weight chunk
  constexpr int NW = CC;                             // weight DMA instructions per wave per chunk (one round = one channel)
  constexpr int S = CC / 4;                          // steps (4 channels) per chunk
  constexpr int kMaxIn = 4;                          // host guarantees wino_ni <= kMaxIn
  constexpr int kSlots = 12;                         // DMA issue slots in a chunk's first step (3 per half-step)
  static_assert(kMaxIn + NW <= kSlots, "a chunk's DMA instructions must fit the issue slots of its first step");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int k4 = lane >> 4, l15 = lane & 15;
  const int wt = wave;
  const int THp = a.TH + 2, W = a.W, WP = W + 4;
  const int plane_sz = THp * WP;
  const int CS = a.TN * plane_sz;
  const int xs_sz = a.wino_ni * (NWV * 256);         // input slot, padded to whole DMA rounds (floats)
  const int buf_sz = xs_sz + WCH;

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

  const int q0 = wt * 16 + l15;
  const bool q_ok = q0 < a.TN * a.tpp;
  const int q = q_ok ? q0 : 0;
  const int tn = fdiv(q, a.m_tpp), trem = q - tn * a.tpp;
  const int ty = fdiv(trem, a.m_tpr), tx = trem - ty * a.tpr;
  const int poff = tn * plane_sz + 2 * ty * WP + 3 + 2 * tx + k4 * CS;
  const int swz = (l15 >> 2) & 3;
  int aoff[4];                                       // cout block cb adds 16 rows = 256 floats
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) aoff[xi] = ((k4 * CBW + l15) * 4 + (xi ^ swz)) * 4;

  f32x4 acc[CPW][16];

  const int HW = a.H * W;
  const int qpr = (W >> 2) + 1;
  const int rows_per_ch = a.TN * THp;
  const int nin = CC * rows_per_ch * qpr + 1;        // + the zero quad behind the last row
  const int nchunks = a.cinp / CC;
  const int nps = a.wino_ni + NW;                    // DMA instructions per wave per chunk (uniform)
  const size_t in_step = size_t(CC) * HW, w_step = size_t(CC) * a.coutp * 16;
  const unsigned woffb = unsigned(wave * 64 + lane) * 16u;   // this lane's quad of a channel's [64][16] weight block (bytes)
  const unsigned wdjb = unsigned(a.coutp) * 64u;             // next channel (bytes)
  int su = u, sk = 0;                                // DMA cursor (unit su, chunk sk)
  constexpr unsigned kOOB = 0x80000000u;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = FVP_LDS_BYTE_ADDRESS(smem);
  unsigned voff[kMaxIn];
  i32x4 rs_in = {0, 0, 0x7ffffff0, 0x00020000}, rs_w = {0, 0, 0x7ffffff0, 0x00020000};
  auto set_base = [](i32x4& rs, const float* p) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    rs[0] = __builtin_amdgcn_readfirstlane(int(unsigned(b)));
    rs[1] = __builtin_amdgcn_readfirstlane(int(unsigned(b >> 32) & 0xffffu));
  };
  auto enter_unit = [&](int su_) {
    const int st = fdiv(su_, a.m_ys), sy = su_ - st * a.ysplit;
    const int spg = fdiv(st, a.m_ty), sty = st - spg * a.tiles_y;
    const int splane0 = spg * a.TN, sy0 = sty * a.TH;
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
  // ---- DMA of one chunk as nps separate items: begin (captures the cursor), item(j), end (advances the cursor)
  unsigned st_so_in = 0, st_so_w = 0, st_la0 = 0;
  bool st_on = false;
  auto stage_begin = [&](int boff) {
    st_on = su < nunits;
    st_so_in = unsigned(sk) * unsigned(in_step) * 4u;
    st_so_w = unsigned(sk) * unsigned(w_step) * 4u;
    st_la0 = lds0 + 4u * unsigned(4 + boff + wave_s * 256);
    return st_on;
  };
  auto stage_item = [&](int j) {                     // j wave-uniform
    if (!st_on || j >= nps) return;
    if (j < a.wino_ni) {
      // (voff is indexed by a uniform run-time j through this switch: no dynamic register indexing)
      unsigned vo = voff[0];
      if (j == 1) vo = voff[1];
      if (j == 2) vo = voff[2];
      if (j == 3) vo = voff[3];
      asm_buffer_load_lds16(st_la0 + unsigned(NWV * j) * 1024u, vo, rs_in, st_so_in);
    } else {
      const int jw = j - a.wino_ni;
      asm_buffer_load_lds16(st_la0 + unsigned(xs_sz + NWV * jw * 256) * 4u, woffb, rs_w, st_so_w + unsigned(jw) * wdjb);
    }
  };
  auto stage_end = [&]() {
    if (!st_on) return;
    if (++sk == nchunks) {
      sk = 0;
      su = next_unit(su + G);
      if (su < nunits) enter_unit(su);
    }
  };
  auto stage_all = [&](int boff) {
    const bool on = stage_begin(boff);
    for (int j = 0; j < kSlots; ++j) stage_item(j);
    stage_end();
    return on;
  };
  enter_unit(su);

  // ---- operand fetch / transform / MFMA building blocks
  float4 av[2][4];                                   // A operands of cout block cb live in av[cb & 1]
  f32x2 dM[4], dE[4];                                // patch rows as pairs (d1,d2) and (d0,d3)
  f32x2 v03[4], v12[4];                              // V of the CURRENT step
  f32x2 n03[4], n12[4];                              // V of the NEXT step (built under half-step 3)
  auto fetch_a = [&](int cb, const float* wbase, int s) {
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
      av[cb & 1][xi] = *reinterpret_cast<const float4*>(wbase + aoff[xi] + (s * 4 * CBW * 16 + cb * 256));
  };
  auto fetch_d = [&](const float* base, int s, int wp) {
    const float* xs = base + poff + s * 4 * CS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* row = xs + r * wp;
      dM[r] = *reinterpret_cast<const f32x2*>(row + 1);
      dE[r] = f32x2{row[0], row[3]};
    }
  };
  // (B^T d) B for one xi: rows first (shared), then the column pass of row xi
  f32x2 tM[4], tE[4];
  auto transform_rows = [&]() {
    tM[0] = dM[0] - dM[2];  tE[0] = dE[0] - dE[2];
    tM[1] = dM[1] + dM[2];  tE[1] = dE[1] + dE[2];
    tM[2] = dM[2] - dM[1];  tE[2] = dE[2] - dE[1];
    tM[3] = dM[1] - dM[3];  tE[3] = dE[1] - dE[3];
  };
  auto mfma4 = [&](int cb, int xi, bool first) {     // the four MFMAs of Winograd row xi, cout block cb
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    const float4 w = av[cb & 1][xi];
    if (first) {
      acc[cb][4 * xi + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, v03[xi].x, z, 0, 0, 0);
      acc[cb][4 * xi + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, v12[xi].x, z, 0, 0, 0);
      acc[cb][4 * xi + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, v12[xi].y, z, 0, 0, 0);
      acc[cb][4 * xi + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, v03[xi].y, z, 0, 0, 0);
    } else {
      acc[cb][4 * xi + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, v03[xi].x, acc[cb][4 * xi + 0], 0, 0, 0);
      acc[cb][4 * xi + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, v12[xi].x, acc[cb][4 * xi + 1], 0, 0, 0);
      acc[cb][4 * xi + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, v12[xi].y, acc[cb][4 * xi + 2], 0, 0, 0);
      acc[cb][4 * xi + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, v03[xi].y, acc[cb][4 * xi + 3], 0, 0, 0);
    }
  };

  // ---- prologue: BN vectors -> LDS, chunks 0 and 1 requested, first operands and first V
  {
    float* const e = const_cast<float*>(smem) + 4 + 3 * buf_sz;
    for (int i = t; i < 3 * a.coutp; i += NWV * 64) e[i] = a.epi[i];
  }
  {
    stage_all(0);
    const bool second = stage_all(buf_sz);
    wait_vmcnt(second ? nps : 0);
  }
  __syncthreads();
  int cur_off = 0;
  int st_pending = 0;                                // 1: the previous epilogue drained the DMA queue (stores may be in flight)
  // V of the stream's first step + A(cb 0): exposed once per kernel (later ones are built under half-step 3)
  fetch_a(0, smem + 4 + xs_sz, 0);
  fetch_d(smem + 4, 0, WP);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  transform_rows();
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) wino_cols(tE[xi], tM[xi], v03[xi], v12[xi]);

  while (true) {
    // the chunk body exists twice: the unit's first chunk (its first step's MFMAs take C = 0) and every other one
    auto chunk = [&](int k, auto firstc) {
      constexpr bool kFirst = decltype(firstc)::value;
      const int nxt_off = cur_off + buf_sz >= 3 * buf_sz ? 0 : cur_off + buf_sz;
      const int nn_off = nxt_off + buf_sz >= 3 * buf_sz ? 0 : nxt_off + buf_sz;
      const bool more = stage_begin(nn_off);           // chunk g + 2: its items are issued inside step 0 below
      const float* cur = smem + 4 + cur_off;
      const float* nxt = smem + 4 + nxt_off;
      const float* wcur = cur + xs_sz;
      int wp = WP;
      FVP_OPAQUE(wp);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const bool first = kFirst && s == 0;
        const bool last_step = s + 1 == S;
#pragma unroll
        for (int cb = 0; cb < CPW; ++cb) {
          __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): A(cb) has landed
          __builtin_amdgcn_sched_barrier(0);
          if (cb + 1 < CPW) {
            fetch_a(cb + 1, wcur, s);                  // next cout block of this step
          } else {
            // half-step 3: first operands of the next step.  Across a chunk boundary they come from the next slot,
            // which is complete for everybody (barrier below, taken before this half-step).
            if (!last_step) {
              fetch_a(0, wcur, s + 1);
              fetch_d(cur, s + 1, wp);
            } else {
              fetch_a(0, nxt + xs_sz, 0);
              fetch_d(nxt, 0, wp);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int xi = 0; xi < 4; ++xi) {
            mfma4(cb, xi, first);
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0 && xi < 3) {                    // this chunk's share of the DMA issue: one item per MFMA group
              stage_item(3 * cb + xi);
              __builtin_amdgcn_sched_barrier(0);
            }
            if (cb == CPW - 1) {
              // the next step's patch transform between the MFMA groups of half-step 3: the patch was requested at the
              // top of this half-step (two MFMA groups = ~256 cycles ago when the rows are needed)
              if (xi == 1) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_sched_barrier(0);
                transform_rows();
                wino_cols(tE[0], tM[0], n03[0], n12[0]);
                wino_cols(tE[1], tM[1], n03[1], n12[1]);
                __builtin_amdgcn_sched_barrier(0);
              } else if (xi == 2) {
                wino_cols(tE[2], tM[2], n03[2], n12[2]);
                wino_cols(tE[3], tM[3], n03[3], n12[3]);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          if (s == 0 && cb == CPW - 1) stage_end();    // every item of chunk g + 2 is issued: advance the cursor
          if (last_step && cb == CPW - 2) {
            // ---- chunk barrier, between half-steps 2 and 3 of the chunk's last step: A(cb 3) is in registers
            //      (lgkmcnt(0)), so this wave is done with the slot; chunk g + 1 has landed (counted vmcnt)
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if constexpr (kFirst) {
              if (!st_pending) wait_vmcnt_small(more ? nps : 0);
              st_pending = 0;
            } else {
              wait_vmcnt_small(more ? nps : 0);
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // the V built under half-step 3 becomes the current one
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) { v03[xi] = n03[xi]; v12[xi] = n12[xi]; }
      }
      cur_off = nxt_off;
    };
    chunk(0, std::integral_constant<bool, true>{});
    for (int k = 1; k < nchunks; ++k) chunk(k, std::integral_constant<bool, false>{});

    // ---- unit finished: output transform + epilogue (as k_conv_wino), then the next unit of this workgroup.
    //      av[0] (A of cout block 0) and V of the next unit's first step stay live across it.
    int ue = u;
    FVP_OPAQUE(ue);
    const int ut = fdiv(ue, a.m_ys), uy = ue - ut * a.ysplit;
    const int pg = fdiv(ut, a.m_ty), ty_i = ut - pg * a.tiles_y;
    const int plane0 = pg * a.TN, y0 = ty_i * a.TH, co0 = uy * CBW;
    {
      const bool relu = a.flags & FVP_EPI_RELU;
      const bool res_after = a.flags & FVP_EPI_RES_AFTER_RELU;
      const int plane = plane0 + tn, y = y0 + 2 * ty, x = 2 * tx;
      const bool tile_ok = q_ok && plane < a.planes && y < a.H;
      const unsigned pix = tile_ok ? unsigned(y * W + x) : 0u;
      const unsigned cbase = tile_ok ? unsigned(plane) * a.cout : 0u;
      const unsigned ppix = unsigned((y >> 1) * (W >> 1) + tx);
      int bs = buf_sz;
      FVP_OPAQUE(bs);
      const float* const epi_s = smem + 4 + 3 * bs;
      auto out_off = [&](int co) { return (cbase + (tile_ok && co < a.cout ? co : 0)) * unsigned(HW) + pix; };
      // the epilogue runs in two halves of two cout blocks each (32 couts): residual loads of a half first, then its
      // output transform, one wait, its stores
#pragma unroll
      for (int hb = 0; hb < CPW; hb += 2) {
        float2 r0[2][4], r1[2][4];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const int co4 = co0 + (hb + c2) * 16 + 4 * k4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned off = out_off(co4 + r);
            if (HAS_RES) {
              r0[c2][r] = *reinterpret_cast<const float2*>(a.res + off);
              r1[c2][r] = *reinterpret_cast<const float2*>(a.res + off + W);
            }
          }
        }
        float o[2][4][2][2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sx[4][2];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
              const float m0 = acc[hb + c2][4 * xi][r], m1 = acc[hb + c2][4 * xi + 1][r], m2 = acc[hb + c2][4 * xi + 2][r],
                          m3 = acc[hb + c2][4 * xi + 3][r];
              sx[xi][0] = (m0 + m1) + m2;
              sx[xi][1] = (m1 - m2) - m3;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              o[c2][r][0][e] = (sx[0][e] + sx[1][e]) + sx[2][e];
              o[c2][r][1][e] = (sx[1][e] - sx[2][e]) - sx[3][e];
            }
          }
        // ONE wait: the residual rows of this half - and, in the first half, the DMA chunk requested during this unit's
        // last chunk, which is what lets the stores stay in flight across the next chunk barrier (see k_conv_wino)
        __builtin_amdgcn_sched_barrier(0);
        if (HAS_RES || hb == 0) wait_vmcnt(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const int co4 = co0 + (hb + c2) * 16 + 4 * k4;
          f32x4 bn[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) bn[i] = *reinterpret_cast<const f32x4*>(epi_s + i * a.coutp + co4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float b = bn[0][r], sc = bn[1][r], sh = bn[2][r];
            const float rr[2][2] = {{HAS_RES ? r0[c2][r].x : 0.f, HAS_RES ? r0[c2][r].y : 0.f},
                                    {HAS_RES ? r1[c2][r].x : 0.f, HAS_RES ? r1[c2][r].y : 0.f}};
            float v[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                float xv = bn_affine(o[c2][r][i][e], b, sc, sh);
                if (HAS_RES && !res_after) xv += rr[i][e];
                if (relu) xv = fmaxf(xv, 0.0f);
                if (HAS_RES && res_after) xv += rr[i][e];
                v[i][e] = xv;
              }
            if (tile_ok && co4 + r < a.cout) {
              const unsigned off = (cbase + unsigned(co4 + r)) * unsigned(HW) + pix;
              *reinterpret_cast<float2*>(a.dst + off) = make_float2(v[0][0], v[0][1]);
              *reinterpret_cast<float2*>(a.dst + off + W) = make_float2(v[1][0], v[1][1]);
              if (a.pool_dst)                          // fused max_pool(2,2): this lane's tile is one pooled pixel
                a.pool_dst[(cbase + unsigned(co4 + r)) * unsigned(HW >> 2) + ppix] =
                    fmaxf(fmaxf(v[0][0], v[0][1]), fmaxf(v[1][0], v[1][1]));
            }
          }
        }
      }
      // Stores (and, with a residual, nothing else) may stay in flight: the first half's vmcnt(0) came after the last
      // DMA chunk this unit requested, so the next unit's first chunk barrier needs no vmcnt wait.  The second half's
      // vmcnt(0) (HAS_RES) also waits for the first half's stores: accepted (their round trip overlaps the transform).
      st_pending = 1;
    }
    u = next_unit(u + G);
    if (u >= nunits) break;
  }
}

}  // namespace fvp
