// Input heatmaps rasterised from 2-D detections (the reference's "precomputed-heatmap path"):
// lib/dataset/JointsDataset.py:271-338 (generate_input_heatmap, eval branch) on the GPU.
//
// One workgroup per (image, band of RB rows): the band of ALL joints lives in LDS
// (tile[J][RB][W]); the people of the image are stamped one after the other (windows of different
// people overlap; the element-wise max makes the order irrelevant, and exp(.) <= 1 makes the
// reference's clip(0, 1) a no-op), each wave taking every fourth joint.  The band is then written
// once, coalesced, in both layouts: NCHW rows and the channels-last staging copy.  All scalar
// arithmetic of the reference is float64 (numpy promotes the float32 arange against float64
// scalars) and is kept in float64 here: integer truncations, floor division and the IEEE sqrt /
// division are exact, so window positions are bit-identical; exp(double) is rounded to float32 once.
#include <hip/hip_runtime.h>

#include "fvp_common.h"

namespace fvp {

__global__ void __launch_bounds__(256)
k_rasterise(const double* __restrict__ joints, const int* __restrict__ num_people, int P, int J, int W, int H, int RB,
            double fsx, double fsy, double sigma, float* __restrict__ nchw, float* __restrict__ cl, int JP) {
  HIP_DYNAMIC_SHARED(float, tile)                     // [J][RB][W]
  constexpr int kMaxPJ = 1024;                        // people x joints handled per pass (host-checked)
  __shared__ double q[kMaxPJ][2];                     // joints / feat_stride
  __shared__ double prm[64][4];                       // per person: tmp_size, x0, den, ng
  const int img = blockIdx.y, y0 = blockIdx.x * RB, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rows = H - y0 < RB ? H - y0 : RB;
  const int plane = RB * W;
  for (int i = t; i < J * plane; i += 256) tile[i] = 0.0f;
  const int np_ = num_people[img] < P ? num_people[img] : P;
  // ---- per-joint quotients and per-person Gaussian parameters, once per workgroup
  const double* pj = joints + size_t(img) * P * J * 2;
  for (int i = t; i < np_ * J; i += 256) {
    q[i][0] = pj[2 * i] / fsx;
    q[i][1] = pj[2 * i + 1] / fsy;
  }
  __syncthreads();
  if (t < np_) {
    // compute_human_scale (:197-203) with every joint visible
    const double(*qp)[2] = q + t * J;
    double minx = qp[0][0], maxx = minx, miny = qp[0][1], maxy = miny;
    for (int k = 1; k < J; ++k) {
      minx = fmin(minx, qp[k][0]);
      maxx = fmax(maxx, qp[k][0]);
      miny = fmin(miny, qp[k][1]);
      maxy = fmax(maxy, qp[k][1]);
    }
    const double ext = fmax(maxy - miny, maxx - minx);
    double hs = ext * ext;
    hs = fmin(fmax(hs, 1.0 / 4 * 96 * 96), 4.0 * 96 * 96);
    hs = 2 * hs;
    const double cur_sigma = sigma * sqrt(hs / (96.0 * 96.0));
    const double tmp_size = cur_sigma * 3;
    const double size = 2 * tmp_size + 1;
    prm[t][0] = tmp_size;
    prm[t][1] = floor(size / 2);                      // x0 = y0 = size // 2
    prm[t][2] = 2 * (cur_sigma * cur_sigma);
    prm[t][3] = ceil(size);                           // len(np.arange(0, size, 1))
  }
  for (int n = 0; n < np_; ++n) {
    __syncthreads();                                  // parameters ready; previous person's LDS stores done
    const double tmp_size = prm[n][0], x0 = prm[n][1], den = prm[n][2];
    const int ng = int(prm[n][3]);
    for (int j = wave; j < J; j += 4) {
      const int mu_x = int(q[n * J + j][0]), mu_y = int(q[n * J + j][1]);           // trunc toward zero, as int()
      const int ul0 = int(mu_x - tmp_size), ul1 = int(mu_y - tmp_size);
      const int br0 = int(mu_x + tmp_size + 1), br1 = int(mu_y + tmp_size + 1);
      if (ul0 >= W || ul1 >= H || br0 < 0 || br1 < 0) continue;
      const int gx0 = ul0 < 0 ? -ul0 : 0, gx1 = (br0 < W ? br0 : W) - ul0;
      const int gy0 = ul1 < 0 ? -ul1 : 0, gy1 = (br1 < H ? br1 : H) - ul1;
      const int ix0 = ul0 > 0 ? ul0 : 0, iy0 = ul1 > 0 ? ul1 : 0;
      // numpy slicing clamps the stop index to the array length
      const int wx = (gx1 < ng ? gx1 : ng) - gx0, wy = (gy1 < ng ? gy1 : ng) - gy0;
      if (wx <= 0 || wy <= 0) continue;
      // rows of the window inside this band
      const int ya = iy0 > y0 ? iy0 : y0, yb = (iy0 + wy < y0 + rows ? iy0 + wy : y0 + rows);
      if (ya >= yb) continue;
      float* tj = tile + j * plane;
      for (int i = lane; i < (yb - ya) * wx; i += 64) {
        const int ry = i / wx, xx = i - ry * wx;
        const int yy = ya + ry - iy0;                 // row inside the window slice
        const double dx = double(gx0 + xx) - x0, dy = double(gy0 + yy) - x0;
        const float g = float(exp(-(dx * dx + dy * dy) / den));
        float* d = tj + (ya + ry - y0) * W + ix0 + xx;
        *d = fmaxf(*d, g);
      }
    }
  }
  __syncthreads();
  const int HW = H * W;
  if (nchw)
    for (int i = t; i < J * rows * W; i += 256) {
      const int j = i / (rows * W), r = i - j * (rows * W);
      nchw[(size_t(img) * J + j) * HW + y0 * W + r] = tile[j * plane + r];
    }
  if (cl)
    for (int i = t; i < rows * W * JP; i += 256) {
      const int px = i / JP, c = i - px * JP;
      cl[(size_t(img) * HW + y0 * W + px) * JP + c] = c < J ? tile[c * plane + px] : 0.0f;
    }
}

}  // namespace fvp

using namespace fvp;

extern "C" int fvp_rasterise_heatmaps(const double* joints, const int32_t* num_people, int nimg, int P, int J, int W,
                                      int H, double feat_stride_x, double feat_stride_y, double sigma,
                                      float* heat_nchw, float* heat_cl, int JP, fvp_stream_t s) {
  FVP_REQUIRE(joints && num_people && (heat_nchw || heat_cl) && nimg >= 0 && P >= 0 && J > 0 && W > 0 && H > 0);
  FVP_REQUIRE(feat_stride_x > 0 && feat_stride_y > 0 && sigma > 0 && (!heat_cl || JP >= J));
  if (nimg == 0) return 0;
  FVP_LIMIT(size_t(J) * W * sizeof(float) <= 32 * 1024 && P <= 64 && P * J <= 1024);
  int RB = 4;                                          // rows per band: the band of all joints fits 64 KB of LDS
  while (RB > 1 && size_t(J) * RB * W * sizeof(float) > 44 * 1024) RB >>= 1;   // + 18 KB of static LDS
  ProfScope ps(FVP_K_OTHER, as_stream(s));
  hipLaunchKernelGGL(k_rasterise, dim3(ceil_div(H, RB), nimg), dim3(256), size_t(J) * RB * W * sizeof(float),
                     as_stream(s), joints, num_people, P, J, W, H, RB, feat_stride_x, feat_stride_y, sigma, heat_nchw,
                     heat_cl, JP);
  return launch_status();
}
