// Per-RoI layers of the MNC heads for gfx950: ROIWarping, MaskResize, MaskPooling and the MAX 2x2/2 poolings that
// follow them (test.prototxt:479-505, 558-582, 631-650, 809-820, 885-909, 958-977).
//
// The arithmetic of the three MNC-specific Caffe layers lives in the un-vendored caffe-mnc submodule; these kernels
// follow oracle/SPEC.md (every convention that had to be chosen is tagged SPEC-CHOICE there and here) and mirror
// oracle/mnc_oracle.c operation by operation.
//
// Layouts: conv5_3 in c8 [C/8][H][W][8]; per-RoI features [R][PH][PW][C] -- one RoI is one K-contiguous GEMM row, and
// every kernel below reads/writes 16-byte vectors along C.  All of this is HBM/L2-bound gather/elementwise work: no LDS
// staging is needed because conv5_3 (4.9 MB at 600x1000) stays L2/Infinity-Cache resident across the 300 RoIs; ROIWarping
// first re-lays it out pixel-major so that every bilinear tap of a wave is one contiguous kilobyte.
#include <cfloat>

#include "mnc_internal.h"

namespace mnc {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

// conv5_3 c8 [C/8][H][W][8] -> pixel-major [H][W][C]: the gather below reads whole pixels (all C channels of a tap are one
// contiguous run), which c8 scatters over C/8 planes.  4.9 MB at 600x1000: the transposition costs a few microseconds.
__global__ __launch_bounds__(256) void c8_to_hwc_kernel(const float* __restrict__ in, float* __restrict__ out, int CB,
                                                        long HW) {
  const long total = HW * CB * 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int half = (int)(idx & 1);
    const long t = idx >> 1;
    const int cb = (int)(t % CB);
    const long p = t / CB;
    *reinterpret_cast<float4*>(out + (p * CB + cb) * 8 + half * 4) = ld4(in + ((long)cb * HW + p) * 8 + half * 4);
  }
}

// One bilinear sample of 4 channels at feature-map position (sx, sy); taps outside the map contribute 0.
// SPEC.md section 1: w00*f00 + w01*f01 + w10*f10 + w11*f11 in that order.  `px` = hwc feature map + the lane's channel offset.
__device__ __forceinline__ float4 warp_sample(const float* __restrict__ px, int H, int W, int C, float sx, float sy) {
  const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
  const float ax = sx - (float)x0, ay = sy - (float)y0;
  const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;
  const bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* p00 = px + ((long)y0 * W + x0) * C;
  const float4 a00 = (vy0 && vx0) ? ld4(p00) : z;
  const float4 a01 = (vy0 && vx1) ? ld4(p00 + C) : z;
  const float* p10 = p00 + (long)W * C;
  const float4 a10 = (vy1 && vx0) ? ld4(p10) : z;
  const float4 a11 = (vy1 && vx1) ? ld4(p10 + C) : z;
#define MNC_BL(f) (w00 * a00.f + w01 * a01.f + w10 * a10.f + w11 * a11.f)
  return make_float4(MNC_BL(x), MNC_BL(y), MNC_BL(z), MNC_BL(w));
#undef MNC_BL
}

// thread = (roi, ph, pw, 4-channel group); channels fastest -> a wave reads 1 KB contiguous per tap and writes 1 KB.
// SPEC-CHOICE (SPEC.md 1): un-rounded edges x*scale; roi_w = max(x2s-x1s+1, 1); bin = roi_w/PWs; sample at x1s + pw*bin.
// POOL2: the warp grid is (2PH)x(2PW) and each output is the max of its 2x2 samples (the fused Pooling layer).
template <int POOL2>
__global__ __launch_bounds__(256) void roi_warp_kernel(const float* __restrict__ feat_hwc, int C, int H, int W,
                                                       const float* __restrict__ rois, int R, int PH, int PW, float scale,
                                                       float* __restrict__ out) {
  const int C4 = C >> 2;
  const long total = (long)R * PH * PW * C4;
  const int GH = POOL2 ? 2 * PH : PH, GW = POOL2 ? 2 * PW : PW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    long t = idx / C4;
    const int pw = (int)(t % PW);
    t /= PW;
    const int ph = (int)(t % PH);
    const int r = (int)(t / PH);
    const float* roi = rois + (long)r * 5;
    const float x1s = roi[1] * scale, y1s = roi[2] * scale, x2s = roi[3] * scale, y2s = roi[4] * scale;
    const float rw = fmaxf(x2s - x1s + 1.0f, 1.0f), rh = fmaxf(y2s - y1s + 1.0f, 1.0f);
    const float bw = rw / (float)GW, bh = rh / (float)GH;
    const float* px = feat_hwc + c4 * 4;
    float4 o;
    if (POOL2) {
      o = warp_sample(px, H, W, C, x1s + (float)(2 * pw) * bw, y1s + (float)(2 * ph) * bh);
      o = max4(o, warp_sample(px, H, W, C, x1s + (float)(2 * pw + 1) * bw, y1s + (float)(2 * ph) * bh));
      o = max4(o, warp_sample(px, H, W, C, x1s + (float)(2 * pw) * bw, y1s + (float)(2 * ph + 1) * bh));
      o = max4(o, warp_sample(px, H, W, C, x1s + (float)(2 * pw + 1) * bw, y1s + (float)(2 * ph + 1) * bh));
    } else {
      o = warp_sample(px, H, W, C, x1s + (float)pw * bw, y1s + (float)ph * bh);
    }
    *reinterpret_cast<float4*>(out + idx * 4) = o;      // == (((r*PH + ph)*PW + pw)*C + c4*4)
  }
}

// [R][PH][PW][C] -> [R][PH/2][PW/2][C], MAX 2x2/2 (PH, PW even on this path: 28->14, 14->7)
__global__ __launch_bounds__(256) void maxpool2_rhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                            int PH, int PW, int C4) {
  const int OH = PH / 2, OW = PW / 2;
  const long total = (long)R * OH * OW * C4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    long t = idx / C4;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const long r = t / OH;
    const float* p = in + (((r * PH + 2 * oh) * PW + 2 * ow) * C4 + c4) * 4;
    float4 m = max4(ld4(p), ld4(p + (long)C4 * 4));
    m = max4(m, ld4(p + (long)PW * C4 * 4));
    m = max4(m, ld4(p + (long)(PW + 1) * C4 * 4));
    *reinterpret_cast<float4*>(out + idx * 4) = m;
  }
}

// SPEC-CHOICE (SPEC.md 2): ratio = in/out, source = dst*ratio (top-left aligned), floor + bilinear, nearest on the last
// source row/column -- the author's own convention in lib/nms/mv_kernel.cu:193-240.
__global__ void mask_resize_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int IH, int IW, int OH,
                                   int OW) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * OH * OW) return;
  const int w = idx % OW, h = (idx / OW) % OH, r = idx / (OW * OH);
  const float rh = (float)IH / (float)OH, rw = (float)IW / (float)OW;
  const float ix = (float)w * rw, iy = (float)h * rh;
  const int sx = (int)floorf(ix), sy = (int)floorf(iy);
  const float* m = in + (long)r * IH * IW;
  float v;
  if (sx == IW - 1 || sy == IH - 1) {
    v = m[sy * IW + sx];
  } else {
    const float fx = ix - (float)sx, fy = iy - (float)sy;
    v = (1.0f - fx) * (1.0f - fy) * m[sy * IW + sx] + fx * (1.0f - fy) * m[sy * IW + sx + 1] +
        (1.0f - fx) * fy * m[(sy + 1) * IW + sx] + fx * fy * m[(sy + 1) * IW + sx + 1];
  }
  out[idx] = v;
}

// SPEC-CHOICE (SPEC.md 3): feature * continuous mask, broadcast over channels; POOL2 fuses the MAX 2x2/2 that follows.
template <int POOL2>
__global__ __launch_bounds__(256) void mask_pool_kernel(const float* __restrict__ feat, const float* __restrict__ mask,
                                                        float* __restrict__ out, int R, int PH, int PW, int C4) {
  const int OH = POOL2 ? PH / 2 : PH, OW = POOL2 ? PW / 2 : PW;
  const long total = (long)R * OH * OW * C4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    long t = idx / C4;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const long r = t / OH;
    auto prod = [&](int h, int w) {
      const float mk = mask[(r * PH + h) * PW + w];
      const float4 f = ld4(feat + (((r * PH + h) * PW + w) * C4 + c4) * 4);
      return make_float4(f.x * mk, f.y * mk, f.z * mk, f.w * mk);
    };
    float4 v;
    if (POOL2) {
      v = max4(max4(prod(2 * oh, 2 * ow), prod(2 * oh, 2 * ow + 1)), max4(prod(2 * oh + 1, 2 * ow), prod(2 * oh + 1, 2 * ow + 1)));
    } else {
      v = prod(oh, ow);
    }
    *reinterpret_cast<float4*>(out + idx * 4) = v;
  }
}

// [R][C][P] <-> [R][P][C]  (P = PH*PW)
__global__ void rchw_to_rhwc_kernel(const float* __restrict__ in, float* __restrict__ out, long R, int C, int P) {
  const long total = R * C * P;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const long t = idx / C;
    const int p = (int)(t % P);
    const long r = t / P;
    out[idx] = in[(r * C + c) * P + p];
  }
}
__global__ void rhwc_to_rchw_kernel(const float* __restrict__ in, float* __restrict__ out, long R, int C, int P) {
  const long total = R * C * P;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int p = (int)(idx % P);
    const long t = idx / P;
    const int c = (int)(t % C);
    const long r = t / C;
    out[idx] = in[(r * P + p) * C + c];
  }
}

static int grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 256 * 64) g = 256 * 64;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace mnc

using namespace mnc;

extern "C" {

int mnc_roi_warp(mnc_ctx* ctx, const float* d_feat, int C, int H, int W, const float* d_rois, int R, int PH, int PW,
                 float scale, int pool2, float* d_out) {
  MNC_REQUIRE(ctx && d_feat && d_out && (R == 0 || d_rois), "mnc_roi_warp: null pointer");
  MNC_REQUIRE(C > 0 && C % 8 == 0 && H > 0 && W > 0 && R >= 0 && PH > 0 && PW > 0, "mnc_roi_warp: bad shape");
  if (R == 0) return MNC_OK;
  const long total = (long)R * PH * PW * (C / 4);
  const double samples = pool2 ? 4.0 : 1.0;
  // pixel-major copy of the feature map in the context's scratch arena (same stream: ordered after any earlier user)
  int rc = ensure_scratch(ctx, (size_t)C * H * W * 4);
  if (rc) return rc;
  float* d_hwc = (float*)ctx->scratch;
  {
    LaunchScope lt(ctx, "c8_to_hwc", 0.0, 8.0 * C * (double)H * W);
    hipLaunchKernelGGL(c8_to_hwc_kernel, dim3(grid_for((long)H * W * (C / 8) * 2)), dim3(256), 0, ctx->stream, d_feat, d_hwc,
                       C / 8, (long)H * W);
    rc = lt.finish("c8_to_hwc_kernel");
    if (rc) return rc;
  }
  LaunchScope ls(ctx, pool2 ? "roi_warp_pool2" : "roi_warp", 0.0, 4.0 * ((double)R * PH * PW * C * (1.0 + 4.0 * samples)));
  if (pool2)
    hipLaunchKernelGGL(roi_warp_kernel<1>, dim3(grid_for(total)), dim3(256), 0, ctx->stream, d_hwc, C, H, W, d_rois, R, PH,
                       PW, scale, d_out);
  else
    hipLaunchKernelGGL(roi_warp_kernel<0>, dim3(grid_for(total)), dim3(256), 0, ctx->stream, d_hwc, C, H, W, d_rois, R, PH,
                       PW, scale, d_out);
  return ls.finish("roi_warp_kernel");
}

// ---- ROIPooling (Fast R-CNN max pooling over integer bins; models/VGG16/cfm/test.prototxt:397-407, 446-456) ---------
// One thread per (roi, ph, pw, group of 8 channels), channel group fastest: the [R][PH][PW][C] output is written in
// contiguous 32-byte pieces and each read is one 32-byte c8 pixel.  `v > max` comparisons (not fmaxf) so that NaN features
// are skipped exactly as in Caffe's kernel; an empty bin yields 0.
__global__ __launch_bounds__(256) void roi_pool_kernel(const float* __restrict__ feat, int N, int C8, int H, int W,
                                                       const float* __restrict__ rois, int PH, int PW, float scale,
                                                       long total, float* __restrict__ out) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % C8);
    long t = idx / C8;
    const int pw = (int)(t % PW);
    t /= PW;
    const int ph = (int)(t % PH);
    const long r = t / PH;
    const float* roi = rois + r * 5;
    int b = (int)roi[0];
    b = b < 0 ? 0 : (b >= N ? N - 1 : b);               // Caffe CHECKs the index on its CPU path only; stay in bounds
    const int x1 = (int)roundf(roi[1] * scale), y1 = (int)roundf(roi[2] * scale);
    const int x2 = (int)roundf(roi[3] * scale), y2 = (int)roundf(roi[4] * scale);
    const int rw = max(x2 - x1 + 1, 1), rh = max(y2 - y1 + 1, 1);
    const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    int hs = (int)floorf((float)ph * bh), he = (int)ceilf((float)(ph + 1) * bh);
    int ws = (int)floorf((float)pw * bw), we = (int)ceilf((float)(pw + 1) * bw);
    hs = min(max(hs + y1, 0), H);
    he = min(max(he + y1, 0), H);
    ws = min(max(ws + x1, 0), W);
    we = min(max(we + x1, 0), W);
    const bool empty = he <= hs || we <= ws;
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = empty ? 0.f : -FLT_MAX;
    const float* plane = feat + (((long)b * C8 + c8) * H) * (long)W * 8;
    for (int h = hs; h < he; ++h)
      for (int w = ws; w < we; ++w) {
        const float4 a = *reinterpret_cast<const float4*>(plane + ((long)h * W + w) * 8);
        const float4 c = *reinterpret_cast<const float4*>(plane + ((long)h * W + w) * 8 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (v[k] > m[k]) m[k] = v[k];
      }
    float* o = out + idx * 8;
    *reinterpret_cast<float4*>(o) = make_float4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(m[4], m[5], m[6], m[7]);
  }
}

int mnc_roi_pool(mnc_ctx* ctx, const float* d_feat, int N, int C, int H, int W, const float* d_rois, int R, int PH, int PW,
                 float scale, float* d_out) {
  MNC_REQUIRE(ctx && d_feat && d_out && (R == 0 || d_rois), "mnc_roi_pool: null pointer");
  MNC_REQUIRE(N > 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0 && R >= 0 && PH > 0 && PW > 0, "mnc_roi_pool: bad shape");
  if (R == 0) return MNC_OK;
  const long total = (long)R * PH * PW * (C / 8);
  LaunchScope ls(ctx, "roi_pool", 0.0, 4.0 * (double)R * PH * PW * C * 2.0);
  hipLaunchKernelGGL(roi_pool_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, d_feat, N, C / 8, H, W, d_rois, PH, PW,
                     scale, total, d_out);
  return ls.finish("roi_pool_kernel");
}

int mnc_maxpool2_rhwc(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int PH, int PW, int C) {
  MNC_REQUIRE(ctx && d_in && d_out && R >= 0 && PH > 0 && PW > 0 && PH % 2 == 0 && PW % 2 == 0 && C > 0 && C % 4 == 0,
              "mnc_maxpool2_rhwc: bad argument (PH, PW must be even, C%%4==0)");
  if (R == 0) return MNC_OK;
  LaunchScope ls(ctx, "maxpool2_rhwc", 0.0, 4.0 * R * (double)C * PH * PW * 1.25);
  hipLaunchKernelGGL(maxpool2_rhwc_kernel, dim3(grid_for((long)R * (PH / 2) * (PW / 2) * (C / 4))), dim3(256), 0,
                     ctx->stream, d_in, d_out, R, PH, PW, C / 4);
  return ls.finish("maxpool2_rhwc_kernel");
}

int mnc_mask_resize(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int IH, int IW, int OH, int OW) {
  MNC_REQUIRE(ctx && d_in && d_out && R >= 0 && IH > 1 && IW > 1 && OH > 0 && OW > 0, "mnc_mask_resize: bad argument");
  if (R == 0) return MNC_OK;
  LaunchScope ls(ctx, "mask_resize");
  hipLaunchKernelGGL(mask_resize_kernel, dim3(cdiv((long)R * OH * OW, 256)), dim3(256), 0, ctx->stream, d_in, d_out, R, IH,
                     IW, OH, OW);
  return ls.finish("mask_resize_kernel");
}

int mnc_mask_pool(mnc_ctx* ctx, const float* d_feat, const float* d_mask, float* d_out, int R, int PH, int PW, int C,
                  int pool2) {
  MNC_REQUIRE(ctx && d_feat && d_mask && d_out && R >= 0 && PH > 0 && PW > 0 && C > 0 && C % 4 == 0,
              "mnc_mask_pool: bad argument");
  MNC_REQUIRE(!pool2 || (PH % 2 == 0 && PW % 2 == 0), "mnc_mask_pool: pool2 needs even PH, PW");
  if (R == 0) return MNC_OK;
  const int OH = pool2 ? PH / 2 : PH, OW = pool2 ? PW / 2 : PW;
  LaunchScope ls(ctx, pool2 ? "mask_pool_pool2" : "mask_pool", 0.0, 4.0 * R * (double)C * (PH * PW + OH * OW));
  const long total = (long)R * OH * OW * (C / 4);
  if (pool2)
    hipLaunchKernelGGL(mask_pool_kernel<1>, dim3(grid_for(total)), dim3(256), 0, ctx->stream, d_feat, d_mask, d_out, R, PH,
                       PW, C / 4);
  else
    hipLaunchKernelGGL(mask_pool_kernel<0>, dim3(grid_for(total)), dim3(256), 0, ctx->stream, d_feat, d_mask, d_out, R, PH,
                       PW, C / 4);
  return ls.finish("mask_pool_kernel");
}

int mnc_rchw_to_rhwc(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int C, int PH, int PW) {
  MNC_REQUIRE(ctx && d_in && d_out && R >= 0 && C > 0 && PH > 0 && PW > 0, "mnc_rchw_to_rhwc: bad argument");
  if (R == 0) return MNC_OK;
  LaunchScope ls(ctx, "rchw_to_rhwc");
  hipLaunchKernelGGL(rchw_to_rhwc_kernel, dim3(grid_for((long)R * C * PH * PW)), dim3(256), 0, ctx->stream, d_in, d_out,
                     (long)R, C, PH * PW);
  return ls.finish("rchw_to_rhwc_kernel");
}

int mnc_rhwc_to_rchw(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int C, int PH, int PW) {
  MNC_REQUIRE(ctx && d_in && d_out && R >= 0 && C > 0 && PH > 0 && PW > 0, "mnc_rhwc_to_rchw: bad argument");
  if (R == 0) return MNC_OK;
  LaunchScope ls(ctx, "rhwc_to_rchw");
  hipLaunchKernelGGL(rhwc_to_rchw_kernel, dim3(grid_for((long)R * C * PH * PW)), dim3(256), 0, ctx->stream, d_in, d_out,
                     (long)R, C, PH * PW);
  return ls.finish("rhwc_to_rchw_kernel");
}

}  // extern "C"
