// Image -> network input on the device: prep_im_for_blob (lib/utils/blob.py:36-50) and one level of
// prep_im_for_blob_cfm's pyramid (:53-85) -- mean subtraction, cv2.resize(INTER_LINEAR), HWC -> CHW, zero padding to the
// blob's plane size -- in one pass over the output.  HBM-bound elementwise/gather work (a 600x1000 output reads <= 2.2 MB
// of uint8 pixels and writes 7.2 MB); no LDS staging: neighbouring lanes read neighbouring source pixels.
//
// Bit-exactness contract (this file is compiled with -ffp-contract=off): the arithmetic is the host path's
// (mnc_amd/lib/utils/blob.py: resize_linear), operation by operation in float32 -- pixel - mean in float64 rounded once to
// float32 (numpy's `float32_array -= float64_array`), the horizontal pass a*(1-ax) + b*ax, then the vertical pass
// top*(1-ay) + bottom*ay.  The tap tables (source index + fraction per output column / row) are computed by the SAME host
// function for both paths and handed over as arrays.
#include "mnc_internal.h"

namespace mnc {

__global__ __launch_bounds__(256) void prep_image_kernel(const unsigned char* __restrict__ im, int H, int W, double m0,
                                                         double m1, double m2, const int* __restrict__ x0,
                                                         const float* __restrict__ ax, int OW, const int* __restrict__ y0,
                                                         const float* __restrict__ ay, int OH, float* __restrict__ out,
                                                         int PH, int PW) {
  const long plane = (long)PH * PW;
  const double mean[3] = {m0, m1, m2};
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < plane; idx += (long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % PW), y = (int)(idx / PW);
    if (x >= OW || y >= OH) {                            // im_list_to_blob's zero padding (blob.py:17-33)
      out[idx] = 0.f;
      out[plane + idx] = 0.f;
      out[2 * plane + idx] = 0.f;
      continue;
    }
    const int xa = x0[x], xb = min(xa + 1, W - 1);
    const int ya = y0[y], yb = min(ya + 1, H - 1);
    const float fx = ax[x], fy = ay[y];
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const unsigned char* p00 = im + ((long)ya * W + xa) * 3;
    const unsigned char* p01 = im + ((long)ya * W + xb) * 3;
    const unsigned char* p10 = im + ((long)yb * W + xa) * 3;
    const unsigned char* p11 = im + ((long)yb * W + xb) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a = (float)((double)p00[c] - mean[c]), b = (float)((double)p01[c] - mean[c]);
      const float d = (float)((double)p10[c] - mean[c]), e = (float)((double)p11[c] - mean[c]);
      const float top = a * gx + b * fx;
      const float bot = d * gx + e * fx;
      out[c * plane + idx] = top * gy + bot * fy;
    }
  }
}

}  // namespace mnc

using namespace mnc;

int mnc_prep_image(mnc_ctx* ctx, const unsigned char* d_bgr, int H, int W, const double* means, const int* d_x0,
                   const float* d_ax, int OW, const int* d_y0, const float* d_ay, int OH, float* d_out, int PH, int PW) {
  MNC_REQUIRE(ctx && d_bgr && means && d_x0 && d_ax && d_y0 && d_ay && d_out, "mnc_prep_image: null pointer");
  MNC_REQUIRE(H > 0 && W > 0 && OH > 0 && OW > 0 && PH >= OH && PW >= OW, "mnc_prep_image: bad shape");
  const long plane = (long)PH * PW;
  long g = (plane + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  LaunchScope ls(ctx, "prep_image", 0.0, 12.0 * (double)plane + 3.0 * (double)H * W);
  hipLaunchKernelGGL(prep_image_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, d_bgr, H, W, means[0], means[1],
                     means[2], d_x0, d_ax, OW, d_y0, d_ay, OH, d_out, PH, PW);
  return ls.finish("prep_image_kernel");
}
