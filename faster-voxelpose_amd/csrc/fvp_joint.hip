// JLN tail (gfx950): soft-argmax + WeightNet per (person, plane, joint) map, then offset add,
// confidence-weighted fusion and scatter-back.  Reference sites:
// lib/models/joint_localization_net.py:15-33 (SoftArgmaxLayer), :44-62 (fuse_pose_preds),
// :84-98; lib/models/weight_net.py:48-80; lib/models/faster_voxelpose.py:102-103.
//
// One workgroup per map: the C x C map is read once from HBM into LDS and feeds both the
// softmax expectation (beta = 100 makes it ill-conditioned, so sums are carried in fp64)
// and WeightNet's 1->F 3x3 conv + BN + 2x2 max-pool + ReLU + global average + MLP.
#include <hip/hip_runtime.h>

#include "fvp_common.h"

namespace fvp {

constexpr int kMaxF = 32;

__device__ __forceinline__ double shfl_xor_f64(double v, int o) {
  long long b = __double_as_longlong(v);
  int lo = int(b & 0xffffffffLL), hi = int(b >> 32);
  lo = __shfl_xor(lo, o);
  hi = __shfl_xor(hi, o);
  return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// wn layout: conv_w[F][9] | conv_b[F] | bn_scale[F] | bn_shift[F] | fc1_w[Hd][F] | fc1_b[Hd] | fc2_w[Hd] | fc2_b
__global__ void __launch_bounds__(256)
k_softargmax_weightnet(const float* __restrict__ feat, const float* __restrict__ center_grid,
                       const float* __restrict__ wn, float beta, int J, int C, int F, int Hd,
                       const uint8_t* __restrict__ person_valid, float* __restrict__ pose2d,
                       float* __restrict__ pmax, float* __restrict__ wgt) {
  HIP_DYNAMIC_SHARED(float, smem)                 // map[C*C] | red[...]
  const int j = blockIdx.x, plane = blockIdx.y, p = blockIdx.z;
  if (person_valid && !person_valid[p]) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int CC = C * C;
  float* map = smem;
  double* redd = reinterpret_cast<double*>(smem + ((CC + 1) & ~1));   // [4][3] doubles
  float* redf = reinterpret_cast<float*>(redd + 12);                  // [4][kMaxF] + [kMaxF] avg + [Hd] hidden
  const size_t mi = (size_t(p) * 3 + plane) * J + j;
  const float* src = feat + mi * CC;

  // ---- load map, running max of beta*x
  float lmax = -INFINITY;
  for (int i = t; i < CC; i += 256) {
    const float v = src[i];
    map[i] = v;
    lmax = fmaxf(lmax, __fmul_rn(beta, v));
  }
  for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
  if (lane == 0) redf[wave] = lmax;
  __syncthreads();
  const float m = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  __syncthreads();

  // ---- softmax(beta x): e = exp(beta x - m); S = sum e; pose = sum (e/S) * grid
  double s_e = 0.0;
  for (int i = t; i < CC; i += 256) s_e += double(expf(__fsub_rn(__fmul_rn(beta, map[i]), m)));
  for (int o = 32; o > 0; o >>= 1) s_e += shfl_xor_f64(s_e, o);
  if (lane == 0) redd[wave] = s_e;
  __syncthreads();
  const float S = float(redd[0] + redd[1] + redd[2] + redd[3]);
  __syncthreads();
  const float* grid = center_grid + size_t(plane) * CC * 2;
  double sx = 0.0, sy = 0.0;
  for (int i = t; i < CC; i += 256) {
    const float pr = __fdiv_rn(expf(__fsub_rn(__fmul_rn(beta, map[i]), m)), S);
    sx += double(pr) * double(grid[2 * i]);
    sy += double(pr) * double(grid[2 * i + 1]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    sx += shfl_xor_f64(sx, o);
    sy += shfl_xor_f64(sy, o);
  }
  if (lane == 0) { redd[wave * 3 + 1] = sx; redd[wave * 3 + 2] = sy; }
  __syncthreads();
  if (t == 0) {
    pose2d[mi * 2 + 0] = float(redd[1] + redd[4] + redd[7] + redd[10]);
    pose2d[mi * 2 + 1] = float(redd[2] + redd[5] + redd[8] + redd[11]);
    pmax[mi] = __fdiv_rn(1.0f, S);               // the maximum cell has e = exp(0) = 1
  }

  // ---- WeightNet: conv 1->F k3 (zero pad) + BN + maxpool2 + ReLU, summed over the map
  const float* cw = wn;
  const float* cb = wn + F * 9;
  const float* bs = cb + F;
  const float* bh = bs + F;
  const float* w1 = bh + F;
  const float* b1 = w1 + size_t(Hd) * F;
  const float* w2 = b1 + Hd;
  const float* b2 = w2 + Hd;
  const int PW = C / 2, NWIN = PW * PW;
  float sum[kMaxF];
#pragma unroll
  for (int f = 0; f < kMaxF; ++f) sum[f] = 0.0f;
  // (Round 5 tried to get rid of this loop's 24 SGPR spills - 32 `f < F` predicates kept as SGPR pairs plus 12 F hoisted
  // scalars: a template on F == kMaxF without the predicates spills 787 SGPRs (hipcc then hoists every scalar); re-reading
  // the scalars behind an opaque table pointer per window or per feature group, ordered after an earlier feature's result,
  // compiles to 0 spills but measured 252-299 us against 177 us and was not bit-stable with several batches in flight.
  // The spills are the cheaper form: tests/test_kernel_resources.py records the count.)
  for (int w = t; w < NWIN; w += 256) {
    const int wy = w / PW, wx = w - wy * PW;
    float patch[4][4];                             // rows 2wy-1 .. 2wy+2, cols 2wx-1 .. 2wx+2
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int yy = 2 * wy - 1 + r, xx = 2 * wx - 1 + c;
        patch[r][c] = (yy >= 0 && yy < C && xx >= 0 && xx < C) ? map[yy * C + xx] : 0.0f;
      }
    // the two outputs of a window row (ox = 0, 1) run as one packed fma chain: same fmaf per
    // element, half the VALU instructions
    f32x2 pp[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) pp[r][c] = f32x2{patch[r][c], patch[r][c + 1]};
#pragma unroll
    for (int f = 0; f < kMaxF; ++f) {
      if (f < F) {
        float k[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) k[q] = cw[f * 9 + q];
        f32x2 best = f32x2{-INFINITY, -INFINITY};
#pragma unroll
        for (int oy = 0; oy < 2; ++oy) {
          f32x2 a = f32x2{0.0f, 0.0f};
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
              a = __builtin_elementwise_fma(pp[oy + ky][kx], f32x2{k[ky * 3 + kx], k[ky * 3 + kx]}, a);
          a = (a + cb[f]) * bs[f] + bh[f];
          best = __builtin_elementwise_max(best, a);
        }
        sum[f] += fmaxf(fmaxf(best.x, best.y), 0.0f);
      }
    }
  }
#pragma unroll
  for (int f = 0; f < kMaxF; ++f) {
    if (f < F) {
      float v = sum[f];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) redf[wave * kMaxF + f] = v;
    }
  }
  __syncthreads();
  float* avg = redf + 4 * kMaxF;
  float* hid = avg + kMaxF;
  if (t < F) avg[t] = (redf[t] + redf[kMaxF + t] + redf[2 * kMaxF + t] + redf[3 * kMaxF + t]) / float(NWIN);
  __syncthreads();
  for (int h = t; h < Hd; h += 256) {
    float a = b1[h];
    for (int f = 0; f < F; ++f) a = fmaf(w1[size_t(h) * F + f], avg[f], a);
    hid[h] = fmaxf(a, 0.0f);
  }
  __syncthreads();
  if (wave == 0) {
    float a = 0.0f;
    for (int h = lane; h < Hd; h += 64) a = fmaf(w2[h], hid[h], a);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) wgt[mi] = 1.0f / (1.0f + expf(-(a + b2[0])));
  }
}

__global__ void __launch_bounds__(256)
k_pack_weightnet(const float* cw, const float* cb, const float* gamma, const float* beta, const float* mean,
                 const float* var, float eps, const float* w1, const float* b1, const float* w2, const float* b2,
                 int F, int Hd, float* wn) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float* o_cb = wn + F * 9;
  float* o_bs = o_cb + F;
  float* o_bh = o_bs + F;
  float* o_w1 = o_bh + F;
  float* o_b1 = o_w1 + size_t(Hd) * F;
  float* o_w2 = o_b1 + Hd;
  float* o_b2 = o_w2 + Hd;
  if (i < F * 9) wn[i] = cw[i];
  if (i < F) {
    o_cb[i] = cb[i];
    const float sc = __fdiv_rn(gamma[i], sqrtf(__fadd_rn(var[i], eps)));
    o_bs[i] = sc;
    o_bh[i] = __fsub_rn(beta[i], __fmul_rn(mean[i], sc));
  }
  if (i < Hd * F) o_w1[i] = w1[i];
  if (i < Hd) { o_b1[i] = b1[i]; o_w2[i] = w2[i]; }
  if (i == 0) o_b2[0] = b2[0];
}

// offsets, fusion, scatter-back: one thread per (person, joint)
__global__ void __launch_bounds__(256)
k_fuse(const float* __restrict__ pose2d, const float* __restrict__ pmax, const float* __restrict__ wgt,
       const float* __restrict__ offset, const uint8_t* __restrict__ person_valid, int nP, int J,
       float* __restrict__ centers, float* __restrict__ fused, float* __restrict__ planes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nP * J) return;
  const int p = i / J, j = i - p * J;
  const bool valid = !person_valid || person_valid[p];
  float* fo = fused + size_t(i) * 5;
  const float flag = centers[size_t(p) * 7 + 3];
  if (!valid) {
    fo[0] = fo[1] = fo[2] = 0.0f;
    fo[3] = flag;
    fo[4] = centers[size_t(p) * 7 + 4];
    for (int k = 0; k < 3; ++k) {
      planes[((size_t(k) * nP + p) * J + j) * 2 + 0] = 0.0f;
      planes[((size_t(k) * nP + p) * J + j) * 2 + 1] = 0.0f;
    }
    return;
  }
  const float ox = offset[p * 3], oy = offset[p * 3 + 1], oz = offset[p * 3 + 2];
  const size_t m0 = (size_t(p) * 3 + 0) * J + j, m1 = (size_t(p) * 3 + 1) * J + j, m2 = (size_t(p) * 3 + 2) * J + j;
  const float xy0 = pose2d[m0 * 2] + ox, xy1 = pose2d[m0 * 2 + 1] + oy;   // :88
  const float xz0 = pose2d[m1 * 2] + ox, xz1 = pose2d[m1 * 2 + 1] + oz;   // :89
  const float yz0 = pose2d[m2 * 2] + oy, yz1 = pose2d[m2 * 2 + 1] + oz;   // :90
  const float wxy = wgt[m0], wxz = wgt[m1], wyz = wgt[m2];
  const float sx = wxy + wxz, sy = wxy + wyz, sz = wxz + wyz;             // :50-55 normalise pairs
  const float x = (wxy / sx) * xy0 + (wxz / sx) * xz0;                    // :57
  const float y = (wxy / sy) * xy1 + (wyz / sy) * yz0;                    // :58
  const float z = (wxz / sz) * xz1 + (wyz / sz) * yz1;                    // :59
  // confidence: mean over (plane, joint) of the max softmax probability (:27-28)
  float cs = 0.0f;
  for (int k = 0; k < 3; ++k)
    for (int jj = 0; jj < J; ++jj) cs += pmax[(size_t(p) * 3 + k) * J + jj];
  const float conf = cs / float(3 * J);
  fo[0] = x;
  fo[1] = y;
  fo[2] = z;
  fo[3] = flag;
  fo[4] = conf;
  planes[((size_t(0) * nP + p) * J + j) * 2 + 0] = xy0;
  planes[((size_t(0) * nP + p) * J + j) * 2 + 1] = xy1;
  planes[((size_t(1) * nP + p) * J + j) * 2 + 0] = xz0;
  planes[((size_t(1) * nP + p) * J + j) * 2 + 1] = xz1;
  planes[((size_t(2) * nP + p) * J + j) * 2 + 0] = yz0;
  planes[((size_t(2) * nP + p) * J + j) * 2 + 1] = yz1;
  if (j == 0) centers[size_t(p) * 7 + 4] = conf;                         // :98 (in place)
}

}  // namespace fvp

using namespace fvp;

extern "C" int fvp_softargmax_weightnet(const float* feat, const float* center_grid, const float* wn, float beta,
                                        int nP, int J, int C, int F, int Hd, const uint8_t* person_valid,
                                        float* pose2d, float* pmax, float* wgt, fvp_stream_t s) {
  FVP_REQUIRE(feat && center_grid && wn && pose2d && pmax && wgt && nP >= 0 && J > 0);
  FVP_LIMIT(F >= 1 && F <= kMaxF && Hd >= 1 && Hd <= 1024 && C >= 2 && C % 2 == 0 && C <= 192);
  if (nP == 0) return 0;
  const size_t lds = (size_t((C * C + 1) & ~1)) * 4 + 12 * 8 + (5 * kMaxF + Hd) * 4;
  FVP_LIMIT(lds <= 160 * 1024);
  static LdsOptIn optin;                               // jln128: the 64 KB map needs the large-LDS opt-in
  if (int e = lds_opt_in(optin, reinterpret_cast<const void*>(&k_softargmax_weightnet), lds > 64 * 1024 ? 160 * 1024 : 0))
    return e;
  // algorithmic FLOPs of WeightNet's conv (2*9*F per pixel) + MLP, for the profile hook
  ProfScope ps(FVP_K_SOFTARGMAX, as_stream(s), double(nP) * 3 * J * (2.0 * 9 * F * C * C + 2.0 * F * Hd + 2.0 * Hd));
  hipLaunchKernelGGL(k_softargmax_weightnet, dim3(J, 3, nP), dim3(256), lds, as_stream(s), feat, center_grid, wn,
                     beta, J, C, F, Hd, person_valid, pose2d, pmax, wgt);
  return launch_status();
}

extern "C" int fvp_pack_weightnet(const float* conv_w, const float* conv_b, const float* bn_gamma,
                                  const float* bn_beta, const float* bn_mean, const float* bn_var, float eps,
                                  const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                                  int F, int Hd, float* wn, fvp_stream_t s) {
  FVP_REQUIRE(conv_w && conv_b && bn_gamma && bn_beta && bn_mean && bn_var && fc1_w && fc1_b && fc2_w && fc2_b && wn);
  FVP_LIMIT(F >= 1 && F <= kMaxF && Hd >= 1);
  const int n = Hd * F > F * 9 ? Hd * F : F * 9;
  hipLaunchKernelGGL(k_pack_weightnet, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(s), conv_w, conv_b, bn_gamma,
                     bn_beta, bn_mean, bn_var, eps, fc1_w, fc1_b, fc2_w, fc2_b, F, Hd, wn);
  return launch_status();
}

extern "C" int fvp_fuse_poses(const float* pose2d, const float* pmax, const float* wgt, const float* offset,
                              const uint8_t* person_valid, int nP, int J, float* centers, float* fused_poses,
                              float* plane_poses, fvp_stream_t s) {
  FVP_REQUIRE(pose2d && pmax && wgt && offset && centers && fused_poses && plane_poses && nP >= 0 && J > 0);
  if (nP == 0) return 0;
  ProfScope ps(FVP_K_OTHER, as_stream(s));
  hipLaunchKernelGGL(k_fuse, dim3(ceil_div(nP * J, 256)), dim3(256), 0, as_stream(s), pose2d, pmax, wgt, offset,
                     person_valid, nP, J, centers, fused_poses, plane_poses);
  return launch_status();
}
