// Whole 1-D conv stack (C2CNet, lib/models/cnns_1d.py:112-132) in ONE kernel, one workgroup per
// proposal column.  The generic interpreter needs 25 dependent launches of 8-40 us for a net
// that is 4.9 MFLOP per proposal; here all activations (<= 128 channels x <= 24 positions) stay in
// LDS, weights stream from L2 straight into the MFMA A operand through a register prefetch
// ring, and layers are separated by one workgroup barrier instead of a kernel boundary.
//
// LDS: every channel of an activation buffer is a row slot of `slotw` floats = 4 zero columns |
// L data | zero tail (slotw >= L + 8, <= 32); the MFMA pixel lane j is slot column j, so a tap
// reads column j + k - pad and the halo is the zero margin.  Epilogues write whole slots (zeros
// outside the data columns), which lets buffers share LDS space by live range.
// Wave (cb, ks) owns cout block cb (32 couts) and every kKS-th channel pair of a weight chunk (split K: the
// dependent MFMA chain of an output is 1/kKS as long, 4 * kKS waves work instead of <= 4); at the end of an
// op the kKS partial tiles are added in the fixed order ks = 0, 1, ... through the just-consumed ring slot.
// Measured (80 columns): 270 us with one wave per cout block, 238 us with kKS = 4; a deeper ring of smaller
// slots (kNB = 4 x 24 KB, kKS = 2) is slower (269 us): the cost is per chunk (~3 us: scalar descriptor loads,
// 16-wave rendezvous), not DMA latency.  Ablations at 80 columns (231 us): without the vmcnt waits 230 (the weight
// DMA is fully hidden), without the MFMA loop 134, skeleton alone (DMA + rendezvous) 41; requesting the operands
// of the next channel pair before the MFMAs of the current one changes nothing (236): what is left is the 25
// dependent op steps (K-slice reduction + epilogue, ~4 us each) and the 32-wide MFMA on 5-20 useful columns.
// Round 3: (i) the operand words of up to four channel pairs are fetched as one batch in front of their MFMAs (the
// per-MFMA LDS latency was most of the "MFMA loop" above); (ii) the K-slice reduction
// and the epilogue are spread over all sixteen waves (slice ks finishes 4 of the 16 accumulator registers of its cout
// block) instead of four finishers doing all the work while twelve waves wait.
// Deterministic; differs from the generic interpreter's single k-ordered chain only in rounding.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "fvp_common.h"

namespace fvp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMaxOps = 32, kMaxBufs = 40, kMaxChunks = 112;
constexpr int kWFloats = 12288;                       // floats per weight buffer (LDS-DMA ring)
constexpr int kNB = 2;                               // ring depth: chunk c+1 streams in while chunk c computes
constexpr int kKS = 4;                               // K split across waves
constexpr int kNT1 = 256 * kKS;                      // threads per workgroup (4 cout blocks x kKS)
constexpr int kEpiFloats = 512;                      // LDS floats per epilogue-vector slot (3 * coutp <= 384, whole DMA pieces)

// One pipeline step of the fused stack: `nrows` weight rows (row = (ci, tap), coutp floats each) of
// op `op` starting at row0, or a weight-less step (pool / slot clearing) when nrows == 0.
struct Chunk {
  short op, tap;             // tap: transposed-conv tap (0 / 1), else 0
  short row0, nrows;
  short first, last;         // first / last chunk of this (op, tap)
};

struct Fused1dArgs {
  FvpConvOp ops[kMaxOps];
  Chunk chunks[kMaxChunks];
  int buf_off[kMaxBufs];     // LDS float offset of each activation buffer
  int buf_w[kMaxBufs];       // slot width (floats) of each buffer
  int nops, nchunks, planes, cin, L, lds_floats;
  int ablate;                // diagnostics (FVP_C1D_ABLATE): 1 no MFMA loop, 2 no reduction / epilogue, 4 no DMA wait, 8 no DMA (wrong results)
  const float* params;
  const float* in;           // [planes][cin][L]
  float* out;                // [planes][cout_last][L_last]
};

// MFMAs of one weight chunk for wave (cb, ks): channel pairs ks, ks + kKS, ... in batches of two pairs (k7: one).  All
// operand words of a batch (6-7 A-words and B-words) are requested before its first MFMA, so a chunk costs one
// LDS latency instead of one per MFMA (a wave has only 12-14 MFMAs per chunk: the fetch latency was most of the loop).
// Same k order per slice as before: results are unchanged.
template <int KW>
__device__ __forceinline__ void conv_chunk(const Fused1dArgs& a, const FvpConvOp& op, const Chunk& ch,
                                           const float* lds, const float* wbuf, f32x16& acc0, int lane, int cb, int ks) {
  const int half = lane >> 5, l31 = lane & 31;
  constexpr int pad = (KW - 1) / 2;
  const int sw = a.buf_w[op.src];
  const float* in = lds + a.buf_off[op.src] + l31 - pad + (2 * (ch.row0 / (2 * KW)) + half) * sw;
  const float* ws = wbuf + cb * 32 + l31 + half * KW * op.coutp;
  const int np = ch.nrows / (2 * KW);
  constexpr int NB = KW >= 7 ? 1 : 2;                  // pairs per batch: 12-14 operand registers (16 waves: 128 VGPRs, no spills)
  for (int p0 = ks; p0 < np; p0 += NB * kKS) {
    float av[NB][KW], bv[NB][KW];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int p = p0 + u * kKS;
      if (p < np) {                                    // wave-uniform
#pragma unroll
        for (int t = 0; t < KW; ++t) {
          av[u][t] = ws[(2 * p * KW + t) * op.coutp];
          bv[u][t] = in[2 * p * sw + t];
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): the whole batch has landed
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int p = p0 + u * kKS;
      if (p < np) {
#pragma unroll
        for (int t = 0; t < KW; ++t) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u][t], acc0, 0, 0, 0);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(kNT1) k_conv1d_fused(Fused1dArgs a) {
  HIP_DYNAMIC_SHARED(float, lds)
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wave = wv & 3, ks = wv >> 2;             // cout block, K slice
  const int l31 = lane & 31, half = lane >> 5;
  const int plane = blockIdx.x;
  float* const wbufs = lds + a.lds_floats;           // two weight buffers behind the activation arena
  float* const epis = wbufs + kNB * kWFloats;        // two slots of epilogue vectors (op parity)

  // LDS-DMA of chunk c's weight rows into ring slot c % kNB; returns this wave's instruction count
  auto stage = [&](int c) -> int {
    if (c >= a.nchunks || (a.ablate & 8)) return 0;
    const Chunk& ch = a.chunks[c];
    if (ch.nrows == 0) return 0;
    const FvpConvOp& op = a.ops[ch.op];
    const float* src = a.params + op.w_off + size_t(ch.tap) * op.cinp * op.coutp + size_t(ch.row0) * op.coutp;
    float* dst = wbufs + (c % kNB) * kWFloats;
    const int nq = ch.nrows * op.coutp / 4;
    int n = 0;
    for (int g = wv; g * 64 < nq; g += 4 * kKS) {
      const int it = g * 64 + lane;
      ++n;
      if (it < nq)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + it * 4),
                                         (__attribute__((address_space(3))) void*)(dst + g * 256), 16, 0, 0);
    }
    // the op's bias | scale | shift vectors (3 * coutp <= 384 floats) ride along with its first weight chunk: the
    // epilogue used to fetch them from global memory after the last MFMA, ~2 us of exposed latency per op
    if (ch.first && wv == 4 * kKS - 1) {
      const int ne = 3 * op.coutp / 4;                // quads
      float* edst = epis + (ch.op & 1) * kEpiFloats;
      for (int g = 0; g * 64 < ne; ++g) {
        const int it = g * 64 + lane;
        ++n;
        if (it < ne)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.params + op.e_off + it * 4),
                                           (__attribute__((address_space(3))) void*)(edst + g * 256), 16, 0, 0);
      }
    }
    return n;
  };

  // in flight: chunks c+1 .. c+kNB-1 while chunk c computes; ninfl[k] = this wave's DMA count of c+1+k
  int ninfl[kNB - 1];
  stage(0);
#pragma unroll
  for (int k = 0; k < kNB - 1; ++k) ninfl[k] = stage(1 + k);
  // the arena starts zeroed: channel-padding rows and out-of-slot halo reads must hit finite values
  for (int i = t; i < a.lds_floats; i += kNT1) lds[i] = 0.0f;
  __syncthreads();
  {                                                  // input -> buffer 0 (whole slots: zero margins)
    const int sw = a.buf_w[0];
    float* b0 = lds + a.buf_off[0];
    const float* src = a.in + size_t(plane) * a.cin * a.L;
    for (int i = t; i < a.cin * sw; i += kNT1) {
      const int c = i / sw, j = i - c * sw;
      b0[i] = (j >= 4 && j < 4 + a.L) ? src[c * a.L + (j - 4)] : 0.0f;
    }
  }
  __syncthreads();

  f32x16 acc0;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.0f;
  for (int c = 0; c < a.nchunks; ++c) {
    // ring slot (c + kNB - 1) % kNB was last read during chunk c - 1, which every wave has left
    if (c > 0) {
#pragma unroll
      for (int k = 0; k < kNB - 2; ++k) ninfl[k] = ninfl[k + 1];
      ninfl[kNB - 2] = stage(c + kNB - 1);
    }
    // end-of-chunk rendezvous: LDS writes done, chunk c+1 landed, younger DMAs may stay in flight
    auto rendezvous = [&]() {
      int keep = 0;
#pragma unroll
      for (int k = 1; k < kNB - 1; ++k) keep += ninfl[k];
      __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0)
      if (!(a.ablate & 4)) wait_vmcnt(keep);
      __builtin_amdgcn_s_barrier();
    };
    const Chunk& ch = a.chunks[c];
    const FvpConvOp& op = a.ops[ch.op];
    const int Lin = op.w;
    if (ch.nrows == 0) {
      if (op.kind == FVP_OP_POOL2) {
        const int sw = a.buf_w[op.src], dw = a.buf_w[op.dst], Lo = Lin / 2;
        const float* s = lds + a.buf_off[op.src];
        float* d = lds + a.buf_off[op.dst];
        for (int i = t; i < op.cin * dw; i += kNT1) {
          const int cc = i / dw, j = i - cc * dw;
          float v = 0.0f;
          if (j >= 4 && j < 4 + Lo) v = fmaxf(s[cc * sw + 4 + 2 * (j - 4)], s[cc * sw + 4 + 2 * (j - 4) + 1]);
          d[i] = v;
        }
      } else {                                       // transposed conv: clear the slots before the scatter
        float* d = lds + a.buf_off[op.dst];
        const int n = op.cout * a.buf_w[op.dst];
        for (int i = t; i < n; i += kNT1) d[i] = 0.0f;
      }
      rendezvous();
      continue;
    }
    const bool tr = op.kind == FVP_OP_CONVT2;
    const int ncb = op.coutp / 32;
    const bool active = wave < ncb;
    float* const wcur = wbufs + (c % kNB) * kWFloats;
    if (active && !(a.ablate & 1)) {
      if (ch.first) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.0f;
      }
      if (tr || op.kw == 1) conv_chunk<1>(a, op, ch, lds, wcur, acc0, lane, wave, ks);
      else if (op.kw == 3) conv_chunk<3>(a, op, ch, lds, wcur, acc0, lane, wave, ks);
      else conv_chunk<7>(a, op, ch, lds, wcur, acc0, lane, wave, ks);
    }
    if (ch.last && !(a.ablate & 2)) {
      // ---- K-slice reduction + epilogue, spread over ALL waves of a cout block: slice ks finishes accumulator
      //      registers 4 ks .. 4 ks + 3 (couts cb * 32 + 8 ks + 4 half + 0..3).  Every slice parks the 12 registers it
      //      does not finish in the ring slot this chunk just consumed (free until the next iteration's DMA):
      //      part[cb][slice][12][lane]; the owner then adds the four slices in the fixed order 0, 1, 2, 3 (its own
      //      from registers) - the order the single-finisher form used, so results are unchanged by the spreading.
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();                    // every wave is done reading the slot's weights
      f32x16& acc = acc0;
      if (active) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((r >> 2) != ks) wcur[((wave * kKS + ks) * 12 + (r < 4 * ks ? r : r - 4)) * 64 + lane] = acc[r];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      if (active) {
        float own[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                  // wave-uniform selects: no dynamic register indexing
          own[i] = ks == 0 ? acc[i] : (ks == 1 ? acc[4 + i] : (ks == 2 ? acc[8 + i] : acc[12 + i]));
        }
        float v4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * ks + i;
          float v = 0.0f;
#pragma unroll
          for (int k2 = 0; k2 < kKS; ++k2) {
            const float x = k2 == ks ? own[i] : wcur[((wave * kKS + k2) * 12 + (r < 4 * k2 ? r : r - 4)) * 64 + lane];
            v = k2 == 0 ? x : v + x;
          }
          v4[i] = v;
        }
        // ---- epilogue: slot column j = l31; data columns [4, 4 + Lin)
        const int Lo = tr ? 2 * Lin : Lin;
        const int dw = a.buf_w[op.dst];
        const bool final_op = ch.op == a.nops - 1;
        float* dst = lds + a.buf_off[op.dst];
        const float* bias = epis + (ch.op & 1) * kEpiFloats;   // staged with the op's first chunk
        const float* scale = bias + op.coutp;
        const float* shift = bias + 2 * op.coutp;
        const bool relu = op.flags & FVP_EPI_RELU, has_res = op.flags & FVP_EPI_RES;
        const bool res_after = op.flags & FVP_EPI_RES_AFTER_RELU;
        const float* res = has_res ? lds + a.buf_off[op.res] : nullptr;
        const int rw = has_res ? a.buf_w[op.res] : 0;
        const int x = l31 - 4;
        const bool data = x >= 0 && x < Lin;
        const int oc = tr ? 4 + 2 * x + ch.tap : l31;  // output slot column
        const int co0 = wave * 32 + 8 * ks + 4 * half; // this lane's four consecutive couts
        float b4[4], s4[4], h4[4], r4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                  // all LDS operands first (epi vectors are padded to coutp)
          b4[i] = bias[co0 + i];
          s4[i] = scale[co0 + i];
          h4[i] = shift[co0 + i];
          r4[i] = (has_res && data && co0 + i < op.cout) ? res[(co0 + i) * rw + oc] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int co = co0 + i;
          if (co < op.cout) {
            float v = bn_affine(v4[i], b4[i], s4[i], h4[i]);
            if (data) {
              if (has_res && !res_after) v += r4[i];
              if (relu) v = fmaxf(v, 0.0f);
              if (has_res && res_after) v += r4[i];
            } else {
              v = 0.0f;
            }
            if (final_op) {
              if (data) a.out[(size_t(plane) * op.cout + co) * Lo + (tr ? 2 * x + ch.tap : x)] = v;
            } else if (tr) {
              if (data) dst[co * dw + oc] = v;
            } else if (l31 < dw) {
              dst[co * dw + l31] = v;
            }
          }
        }
      }
    }
    rendezvous();                                    // chunk consumed, next weights landed, outputs visible
  }
}

}  // namespace fvp

using namespace fvp;

extern "C" int fvp_conv_stack_run_fused_1d(const FvpConvOp* ops, int nops, const float* params, const float* in,
                                           float* out, int planes, fvp_stream_t s) {
  FVP_REQUIRE(ops && params && in && out && nops > 0 && planes >= 0);
  FVP_LIMIT(nops <= kMaxOps);
  if (planes == 0) return 0;
  Fused1dArgs a;
  std::memset(&a, 0, sizeof(a));
  // ---- validate the stack and find buffer geometry (channels, length) + live ranges
  int nbufs = 0, chan[kMaxBufs], len[kMaxBufs], first[kMaxBufs], lastuse[kMaxBufs];
  for (int i = 0; i < kMaxBufs; ++i) { chan[i] = 0; len[i] = 0; first[i] = -1; lastuse[i] = -1; }
  double flops = 0.0;
  for (int i = 0; i < nops; ++i) {
    const FvpConvOp& op = ops[i];
    FVP_REQUIRE(op.h == 1 && op.src >= 0 && op.dst >= 0);
    FVP_LIMIT(op.src < kMaxBufs && op.dst < kMaxBufs && op.res < kMaxBufs && op.w <= 24 && op.coutp <= 128);
    const int Lo = op.kind == FVP_OP_CONVT2 ? 2 * op.w : (op.kind == FVP_OP_POOL2 ? op.w / 2 : op.w);
    FVP_LIMIT(Lo <= 24);
    if (op.kind == FVP_OP_CONV) FVP_LIMIT(op.kh == 1 && (op.kw == 1 || op.kw == 3 || op.kw == 7));
    FVP_LIMIT(op.e_off % 4 == 0 && op.coutp % 4 == 0);   // the epilogue vectors are staged by 16-byte LDS-DMA items
    chan[op.src] = op.cin;
    len[op.src] = op.w;
    chan[op.dst] = op.cout;
    len[op.dst] = Lo;
    if (first[op.src] < 0) first[op.src] = -1;       // only the stack input is never defined
    if (first[op.dst] < 0) first[op.dst] = i;
    lastuse[op.src] = i;
    if (op.res >= 0) lastuse[op.res] = i;
    if (op.dst + 1 > nbufs) nbufs = op.dst + 1;
    if (op.src + 1 > nbufs) nbufs = op.src + 1;
    if (op.kind != FVP_OP_POOL2)
      flops += 2.0 * op.cin * op.cout * (op.kind == FVP_OP_CONVT2 ? 2.0 : double(op.kw)) * op.w * planes;
    a.ops[i] = op;
  }
  // ---- LDS placement: first-fit over live ranges [def, last use]
  int off[kMaxBufs], size[kMaxBufs];
  int total = 0;
  for (int b = 0; b < nbufs; ++b) {
    const int sw = ((len[b] + 8 + 3) / 4) * 4;        // 4 margin | L | >= 4 tail, multiple of 4
    FVP_LIMIT(sw <= 32);
    a.buf_w[b] = sw;
    size[b] = chan[b] * sw;
  }
  for (int b = 0; b < nbufs; ++b) {
    if (chan[b] == 0) { off[b] = 0; continue; }
    int pos = 0;
    bool moved = true;
    while (moved) {
      moved = false;
      for (int o = 0; o < b; ++o) {
        if (chan[o] == 0) continue;
        const bool live_overlap = !(lastuse[o] < first[b] || lastuse[b] < first[o]);
        if (live_overlap && pos < off[o] + size[o] && off[o] < pos + size[b]) {
          pos = off[o] + size[o];
          moved = true;
        }
      }
    }
    off[b] = pos;
    if (pos + size[b] > total) total = pos + size[b];
  }
  for (int b = 0; b < nbufs; ++b) a.buf_off[b] = off[b];
  // ---- chunk list: weight rows in LDS-DMA pieces of <= kWFloats, whole channel pairs per piece
  int nch = 0;
  auto push = [&](int op, int tap, int row0, int nrows, int first, int last) {
    if (nch < kMaxChunks) {
      Chunk& c = a.chunks[nch];
      c.op = short(op); c.tap = short(tap); c.row0 = short(row0); c.nrows = short(nrows);
      c.first = short(first); c.last = short(last);
    }
    ++nch;
  };
  for (int i = 0; i < nops; ++i) {
    const FvpConvOp& op = ops[i];
    if (op.kind == FVP_OP_POOL2) { push(i, 0, 0, 0, 1, 1); continue; }
    const bool tr = op.kind == FVP_OP_CONVT2;
    const int kk = tr ? 1 : op.kw;
    const int rows = op.cinp * kk;
    int per = (kWFloats / op.coutp) / (2 * kk) * (2 * kk);
    FVP_LIMIT(per >= 2 * kk);
    if (tr) push(i, 0, 0, 0, 1, 1);                    // slot clearing before the strided scatter
    for (int tap = 0; tap < (tr ? 2 : 1); ++tap)
      for (int r0 = 0; r0 < rows; r0 += per)
        push(i, tap, r0, rows - r0 < per ? rows - r0 : per, r0 == 0, r0 + per >= rows);
  }
  FVP_LIMIT(nch <= kMaxChunks);
  a.nchunks = nch;
  a.nops = nops;
  a.planes = planes;
  a.cin = ops[0].cin;
  a.L = ops[0].w;
  total = (total + 64 + 3) & ~3;                     // slack for channel-padding rows; keeps the weight buffers 16-B aligned
  a.lds_floats = total;
  static const int kAblate1d = fvp::diag_env("FVP_C1D_ABLATE") ? atoi(fvp::diag_env("FVP_C1D_ABLATE")) : 0;
  a.ablate = kAblate1d;
  a.params = params;
  a.in = in;
  a.out = out;
  const size_t lds = size_t(total + kNB * kWFloats + 2 * kEpiFloats) * sizeof(float);
  FVP_LIMIT(lds <= 160 * 1024);
  static LdsOptIn optin;
  if (lds_opt_in(optin, reinterpret_cast<const void*>(&k_conv1d_fused), lds > 64 * 1024 ? 160 * 1024 : 0)) return FVP_ELIMIT;
  ProfScope ps(FVP_K_CONV, as_stream(s), flops, nops, prof_level() >= 1);
  static_assert(kKS * 4 * 12 * 64 <= kWFloats, "the K-slice partials must fit one ring slot");
  hipLaunchKernelGGL(k_conv1d_fused, dim3(planes), dim3(kNT1), lds, as_stream(s), a);
  return launch_status();
}
