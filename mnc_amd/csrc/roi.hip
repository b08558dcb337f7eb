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
#include <cstdlib>
#include <type_traits>

#include "mnc_internal.h"
#include "x3_split.h"

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

// sm_store4 / sm_store8 (x3_split.h): the second output of the per-RoI producers, the tensor in the stage-major 2-byte form the
// reduced-precision InnerProducts multiply from (mnc_hip.h: mnc_fc_{f16,bf16x3}_pre).

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
template <int POOL2, int SM>
__global__ __launch_bounds__(256) void roi_warp_kernel(const float* __restrict__ feat_hwc, int C, int H, int W,
                                                       const float* __restrict__ rois, int R, int PH, int PW, float scale,
                                                       float* __restrict__ out, void* __restrict__ sm) {
  const unsigned C4 = (unsigned)C >> 2;
  // 32-bit index arithmetic (the launcher checks total < 2^31): three 64-bit divisions by run-time values were ~400 of the ~700
  // VALU instructions of an iteration of this kernel
  const unsigned total = (unsigned)R * PH * PW * C4;
  const int GH = POOL2 ? 2 * PH : PH, GW = POOL2 ? 2 * PW : PW;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    unsigned t = idx / C4;
    const int pw = (int)(t % (unsigned)PW);
    t /= (unsigned)PW;
    const int ph = (int)(t % (unsigned)PH);
    const int r = (int)(t / (unsigned)PH);
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
    if (!SM || out) *reinterpret_cast<float4*>(out + (long)idx * 4) = o;      // == (((r*PH + ph)*PW + pw)*C + c4*4); (SM: optional)
    if (SM) sm_store4<SM>(sm, R, r, ((long)ph * PW + pw) * C + c4 * 4, o);
  }
}

// ---- ROIWarping with the SPEC-CHOICEs as run-time switches (mnc_layer_conventions, include/mnc_hip.h; oracle/SPEC.md section 6) ----
// The kernels above and the wave kernels below are the SPEC's conventions compiled in.  When a context carries any other
// convention the launcher takes this kernel instead: same thread mapping as roi_warp_kernel, the edge rounding, width rule,
// sample position inside the bin and border rule evaluated from the struct, in oracle/mnc_oracle.c:orc_roi_warp_ex's operation
// order (bit-exact with it for every combination, tests/test_gpu_ops.py).  Untuned on purpose: it exists so that swapping a
// convention is configuration the day the caffe-mnc source can be read; the chosen one then moves into the tuned kernels.
__device__ __forceinline__ float warp_coord(float lo, float bin, int g, int sample) {
  return sample == 0 ? lo + (float)g * bin : sample == 1 ? lo + ((float)g + 0.5f) * bin : lo + ((float)g + 0.5f) * bin - 0.5f;
}

__device__ __forceinline__ float4 warp_sample_conv(const float* __restrict__ px, int H, int W, int C, float sx, float sy, int oob) {
  const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
  const float ax = sx - (float)x0, ay = sy - (float)y0;
  const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;
  bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
  bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
  int ya = y0, yb = y0 + 1, xa = x0, xb = x0 + 1;
  if (oob) {                                   // taps clamped to the border instead of contributing 0
    ya = min(max(ya, 0), H - 1); yb = min(max(yb, 0), H - 1);
    xa = min(max(xa, 0), W - 1); xb = min(max(xb, 0), W - 1);
    vy0 = vy1 = vx0 = vx1 = true;
  }
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 a00 = (vy0 && vx0) ? ld4(px + ((long)ya * W + xa) * C) : z;
  const float4 a01 = (vy0 && vx1) ? ld4(px + ((long)ya * W + xb) * C) : z;
  const float4 a10 = (vy1 && vx0) ? ld4(px + ((long)yb * W + xa) * C) : z;
  const float4 a11 = (vy1 && vx1) ? ld4(px + ((long)yb * W + xb) * C) : z;
#define MNC_BL(f) (w00 * a00.f + w01 * a01.f + w10 * a10.f + w11 * a11.f)
  return make_float4(MNC_BL(x), MNC_BL(y), MNC_BL(z), MNC_BL(w));
#undef MNC_BL
}

template <int POOL2, int SM>
__global__ __launch_bounds__(256) void roi_warp_conv_kernel(const float* __restrict__ feat_hwc, int C, int H, int W,
                                                            const float* __restrict__ rois, int R, int PH, int PW, float scale,
                                                            mnc_layer_conventions cv, float* __restrict__ out,
                                                            void* __restrict__ sm) {
  const unsigned C4 = (unsigned)C >> 2;
  const unsigned total = (unsigned)R * PH * PW * C4;
  const int GH = POOL2 ? 2 * PH : PH, GW = POOL2 ? 2 * PW : PW;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    unsigned t = idx / C4;
    const int pw = (int)(t % (unsigned)PW);
    t /= (unsigned)PW;
    const int ph = (int)(t % (unsigned)PH);
    const int r = (int)(t / (unsigned)PH);
    const float* roi = rois + (long)r * 5;
    float x1s = roi[1] * scale, y1s = roi[2] * scale, x2s = roi[3] * scale, y2s = roi[4] * scale;
    if (cv.warp_round_edges) { x1s = floorf(x1s + 0.5f); y1s = floorf(y1s + 0.5f); x2s = floorf(x2s + 0.5f); y2s = floorf(y2s + 0.5f); }
    const float extra = cv.warp_no_plus_one ? 0.0f : 1.0f;
    const float rw = fmaxf(x2s - x1s + extra, 1.0f), rh = fmaxf(y2s - y1s + extra, 1.0f);
    const float bw = rw / (float)GW, bh = rh / (float)GH;
    const float* px = feat_hwc + c4 * 4;
    float4 o;
    if (POOL2) {
      const float xa = warp_coord(x1s, bw, 2 * pw, cv.warp_sample), xb = warp_coord(x1s, bw, 2 * pw + 1, cv.warp_sample);
      const float ya = warp_coord(y1s, bh, 2 * ph, cv.warp_sample), yb = warp_coord(y1s, bh, 2 * ph + 1, cv.warp_sample);
      o = warp_sample_conv(px, H, W, C, xa, ya, cv.warp_oob);
      o = max4(o, warp_sample_conv(px, H, W, C, xb, ya, cv.warp_oob));
      o = max4(o, warp_sample_conv(px, H, W, C, xa, yb, cv.warp_oob));
      o = max4(o, warp_sample_conv(px, H, W, C, xb, yb, cv.warp_oob));
    } else {
      o = warp_sample_conv(px, H, W, C, warp_coord(x1s, bw, pw, cv.warp_sample), warp_coord(y1s, bh, ph, cv.warp_sample), cv.warp_oob);
    }
    *reinterpret_cast<float4*>(out + (long)idx * 4) = o;
    if (SM) sm_store4<SM>(sm, R, r, ((long)ph * PW + pw) * C + c4 * 4, o);
  }
}

// [R][PH][PW][C] -> [R][PH/2][PW/2][C], MAX 2x2/2 (PH, PW even on this path: 28->14, 14->7)
template <int SM>
__global__ __launch_bounds__(256) void maxpool2_rhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                            int PH, int PW, int C4, void* __restrict__ sm) {
  const int OH = PH / 2, OW = PW / 2;
  const unsigned total = (unsigned)R * OH * OW * C4;              // < 2^31 (checked by the launcher): 32-bit divisions
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % (unsigned)C4);
    unsigned t = idx / (unsigned)C4;
    const int ow = (int)(t % (unsigned)OW);
    t /= (unsigned)OW;
    const int oh = (int)(t % (unsigned)OH);
    const long r = t / (unsigned)OH;
    const float* p = in + (((r * PH + 2 * oh) * PW + 2 * ow) * C4 + c4) * 4;
    float4 m = max4(ld4(p), ld4(p + (long)C4 * 4));
    m = max4(m, ld4(p + (long)PW * C4 * 4));
    m = max4(m, ld4(p + (long)(PW + 1) * C4 * 4));
    *reinterpret_cast<float4*>(out + (long)idx * 4) = m;
    if (SM) sm_store4<SM>(sm, R, r, ((long)oh * OW + ow) * C4 * 4 + c4 * 4, m);
  }
}

// SPEC-CHOICE (SPEC.md 2; mode 0): ratio = in/out, source = dst*ratio (top-left aligned), floor + bilinear, nearest on the last
// source row/column -- the author's own convention in lib/nms/mv_kernel.cu:193-240.  mode 1 (half-pixel centres) / 2
// (align_corners): the alternatives of mnc_layer_conventions, as oracle/mnc_oracle.c:orc_mask_resize_ex evaluates them.
__global__ void mask_resize_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int IH, int IW, int OH,
                                   int OW, int mode) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * OH * OW) return;
  const int w = idx % OW, h = (idx / OW) % OH, r = idx / (OW * OH);
  const float rh = (float)IH / (float)OH, rw = (float)IW / (float)OW;
  const float* m = in + (long)r * IH * IW;
  float v;
  if (mode == 0) {
    const float ix = (float)w * rw, iy = (float)h * rh;
    const int sx = (int)floorf(ix), sy = (int)floorf(iy);
    if (sx == IW - 1 || sy == IH - 1) {
      v = m[sy * IW + sx];
    } else {
      const float fx = ix - (float)sx, fy = iy - (float)sy;
      v = (1.0f - fx) * (1.0f - fy) * m[sy * IW + sx] + fx * (1.0f - fy) * m[sy * IW + sx + 1] +
          (1.0f - fx) * fy * m[(sy + 1) * IW + sx] + fx * fy * m[(sy + 1) * IW + sx + 1];
    }
  } else {
    const float ah = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.0f, aw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.0f;
    float ix = mode == 1 ? ((float)w + 0.5f) * rw - 0.5f : (float)w * aw;
    float iy = mode == 1 ? ((float)h + 0.5f) * rh - 0.5f : (float)h * ah;
    ix = fminf(fmaxf(ix, 0.0f), (float)(IW - 1));
    iy = fminf(fmaxf(iy, 0.0f), (float)(IH - 1));
    const int sx = (int)floorf(ix), sy = (int)floorf(iy);
    const int tx = min(sx + 1, IW - 1), ty = min(sy + 1, IH - 1);
    const float fx = ix - (float)sx, fy = iy - (float)sy;
    v = (1.0f - fx) * (1.0f - fy) * m[sy * IW + sx] + fx * (1.0f - fy) * m[sy * IW + tx] +
        (1.0f - fx) * fy * m[ty * IW + sx] + fx * fy * m[ty * IW + tx];
  }
  out[idx] = v;
}

// SPEC-CHOICE (SPEC.md 3): feature * continuous mask, broadcast over channels; POOL2 fuses the MAX 2x2/2 that follows.
// bin_on (mnc_layer_conventions::maskpool_binary): the alternative -- the mask is binarised first, m >= bin_thr ? 1 : 0.
__device__ __forceinline__ float mask_value(float m, int bin_on, float bin_thr) { return bin_on ? (m >= bin_thr ? 1.0f : 0.0f) : m; }

template <int POOL2, int SM>
__global__ __launch_bounds__(256) void mask_pool_kernel(const float* __restrict__ feat, const float* __restrict__ mask,
                                                        float* __restrict__ out, int R, int PH, int PW, int C4,
                                                        void* __restrict__ sm, int bin_on, float bin_thr) {
  const int OH = POOL2 ? PH / 2 : PH, OW = POOL2 ? PW / 2 : PW;
  const unsigned total = (unsigned)R * OH * OW * C4;              // < 2^31 (checked by the launcher): 32-bit divisions
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % (unsigned)C4);
    unsigned t = idx / (unsigned)C4;
    const int ow = (int)(t % (unsigned)OW);
    t /= (unsigned)OW;
    const int oh = (int)(t % (unsigned)OH);
    const long r = t / (unsigned)OH;
    auto prod = [&](int h, int w) {
      const float mk = mask_value(mask[(r * PH + h) * PW + w], bin_on, bin_thr);
      const float4 f = ld4(feat + (((r * PH + h) * PW + w) * C4 + c4) * 4);
      return make_float4(f.x * mk, f.y * mk, f.z * mk, f.w * mk);
    };
    float4 v;
    if (POOL2) {
      v = max4(max4(prod(2 * oh, 2 * ow), prod(2 * oh, 2 * ow + 1)), max4(prod(2 * oh + 1, 2 * ow), prod(2 * oh + 1, 2 * ow + 1)));
    } else {
      v = prod(oh, ow);
    }
    *reinterpret_cast<float4*>(out + (long)idx * 4) = v;
    if (SM) sm_store4<SM>(sm, R, r, ((long)oh * OW + ow) * C4 * 4 + c4 * 4, v);
  }
}


// ---- ROIWarping, one wave per output position ------------------------------------------------------------------------------
// All channels of an output position (r, ph, pw) share its sample coordinates, bilinear weights and tap validity.  The kernels
// above recompute them in every thread (4 channels each): ~120 of their ~260 VALU instructions per output at POOL2, on top of
// three run-time integer divisions.  Here a wave takes whole positions: the position is wave-uniform (scalar unit), the sample
// set-up runs once per position, and the lanes stride over the channel groups (4 iterations at C = 1024, 2 at C = 512) doing
// only taps x weights -- in the same operation order as warp_sample, value for value.  Positions whose taps all lie inside the
// map (almost all) take straight-line unconditional loads; the others select zero taps as before.
struct WarpSample {
  int off;                           // ((y0 * W + x0) * C), clamped into the map for the unconditional loads
  int x0, y0;
  float w00, w01, w10, w11;
  bool v00, v01, v10, v11;
};

__device__ __forceinline__ WarpSample warp_setup(int H, int W, int C, float sx, float sy) {
  WarpSample s;
  const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
  s.x0 = x0; s.y0 = y0;
  const float ax = sx - (float)x0, ay = sy - (float)y0;
  s.w00 = (1.0f - ax) * (1.0f - ay); s.w01 = ax * (1.0f - ay); s.w10 = (1.0f - ax) * ay; s.w11 = ax * ay;
  const bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
  s.v00 = vy0 && vx0; s.v01 = vy0 && vx1; s.v10 = vy1 && vx0; s.v11 = vy1 && vx1;
  s.off = (min(max(y0, 0), H - 2) * W + min(max(x0, 0), W - 2)) * C;       // any in-map 2x2 cell when a tap is outside
  return s;
}

// every tap of the sample is inside the map (off is the true cell): unconditional loads
__device__ __forceinline__ float4 warp_taps(const float* __restrict__ px, int W, int C, const WarpSample& s) {
  const float* p00 = px + s.off;
  const float* p10 = p00 + (long)W * C;
  const float4 a00 = ld4(p00), a01 = ld4(p00 + C), a10 = ld4(p10), a11 = ld4(p10 + C);
#define MNC_BL(f) (s.w00 * a00.f + s.w01 * a01.f + s.w10 * a10.f + s.w11 * a11.f)
  return make_float4(MNC_BL(x), MNC_BL(y), MNC_BL(z), MNC_BL(w));
#undef MNC_BL
}

// POOL2, all taps inside the map, and the 2x2 samples of the pooling window at most one cell apart in x and in y (always, for a
// RoI narrower than 28 cells): the four samples' 16 taps are (2 + DX) x (2 + DY) = 4, 6 or 9 distinct pixels.  They are loaded
// once; each sample then blends its own four, in warp_sample's order -- same values, 16 -> 4..9 L1 reads per 4 channels.
template <int DX, int DY, int SM>
__device__ __forceinline__ void warp_pool2_shared_taps(const float* __restrict__ feat_hwc, int W, int C, int C4, int lane,
                                                       const WarpSample (&smp)[4], float* __restrict__ orow, void* __restrict__ sm,
                                                       long M, long r, long kpos) {
  for (int c4 = lane; c4 < C4; c4 += 64) {
    const float* base = feat_hwc + c4 * 4 + smp[0].off;
    float4 g[2 + DY][2 + DX];
#pragma unroll
    for (int y = 0; y < 2 + DY; ++y)
#pragma unroll
      for (int x = 0; x < 2 + DX; ++x) g[y][x] = ld4(base + ((long)y * W + x) * C);
    float4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ox = (i & 1) * DX, oy = (i >> 1) * DY;
      const WarpSample& q = smp[i];
      const float4 a00 = g[oy][ox], a01 = g[oy][ox + 1], a10 = g[oy + 1][ox], a11 = g[oy + 1][ox + 1];
#define MNC_BL(f) (q.w00 * a00.f + q.w01 * a01.f + q.w10 * a10.f + q.w11 * a11.f)
      const float4 v = make_float4(MNC_BL(x), MNC_BL(y), MNC_BL(z), MNC_BL(w));
#undef MNC_BL
      o = i == 0 ? v : max4(o, v);
    }
    if (!SM || orow) *reinterpret_cast<float4*>(orow + c4 * 4) = o;
    if (SM) sm_store4<SM>(sm, M, r, kpos + c4 * 4, o);
  }
}

template <int POOL2, int SM>
__global__ __launch_bounds__(256) void roi_warp_wave_kernel(const float* __restrict__ feat_hwc, int C, int H, int W,
                                                            const float* __restrict__ rois, int R, int PH, int PW, float scale,
                                                            float* __restrict__ out, void* __restrict__ sm) {
  constexpr int NS = POOL2 ? 4 : 1;
  const int C4 = C >> 2, lane = threadIdx.x & 63;
  const unsigned npos = (unsigned)R * PH * PW;
  const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
  const int GH = POOL2 ? 2 * PH : PH, GW = POOL2 ? 2 * PW : PW;
  for (unsigned p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; p < npos; p += nwaves) {
    const unsigned pos = __builtin_amdgcn_readfirstlane(p);        // wave-uniform: the index arithmetic runs on the scalar unit
    const int pw = (int)(pos % (unsigned)PW);
    const unsigned t = pos / (unsigned)PW;
    const int ph = (int)(t % (unsigned)PH), r = (int)(t / (unsigned)PH);
    const float* roi = rois + (long)r * 5;
    const float x1s = roi[1] * scale, y1s = roi[2] * scale, x2s = roi[3] * scale, y2s = roi[4] * scale;
    const float rw = fmaxf(x2s - x1s + 1.0f, 1.0f), rh = fmaxf(y2s - y1s + 1.0f, 1.0f);
    const float bw = rw / (float)GW, bh = rh / (float)GH;
    WarpSample smp[NS];
    bool safe = true;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int gx = POOL2 ? 2 * pw + (i & 1) : pw, gy = POOL2 ? 2 * ph + (i >> 1) : ph;
      smp[i] = warp_setup(H, W, C, x1s + (float)gx * bw, y1s + (float)gy * bh);
      safe = safe && smp[i].v00 && smp[i].v01 && smp[i].v10 && smp[i].v11;
    }
    float* orow = out ? out + (long)pos * C : nullptr;          // (null with a stage-major output only: round 6)
    const long kpos = ((long)ph * PW + pw) * C;
    if (POOL2 && safe && smp[NS - 1].x0 - smp[0].x0 <= 1 && smp[NS - 1].y0 - smp[0].y0 <= 1) {
      const int dx = smp[NS - 1].x0 - smp[0].x0, dy = smp[NS - 1].y0 - smp[0].y0;      // 0 or 1 each, wave-uniform
      const WarpSample(&q)[4] = reinterpret_cast<const WarpSample(&)[4]>(smp);
      if (dx == 0 && dy == 0) warp_pool2_shared_taps<0, 0, SM>(feat_hwc, W, C, C4, lane, q, orow, sm, R, r, kpos);
      else if (dy == 0) warp_pool2_shared_taps<1, 0, SM>(feat_hwc, W, C, C4, lane, q, orow, sm, R, r, kpos);
      else if (dx == 0) warp_pool2_shared_taps<0, 1, SM>(feat_hwc, W, C, C4, lane, q, orow, sm, R, r, kpos);
      else warp_pool2_shared_taps<1, 1, SM>(feat_hwc, W, C, C4, lane, q, orow, sm, R, r, kpos);
    } else if (safe) {
      for (int c4 = lane; c4 < C4; c4 += 64) {
        const float* px = feat_hwc + c4 * 4;
        float4 o = warp_taps(px, W, C, smp[0]);
#pragma unroll
        for (int i = 1; i < NS; ++i) o = max4(o, warp_taps(px, W, C, smp[i]));
        if (!SM || orow) *reinterpret_cast<float4*>(orow + c4 * 4) = o;
        if (SM) sm_store4<SM>(sm, R, r, ((long)ph * PW + pw) * C + c4 * 4, o);
      }
    } else {
      for (int c4 = lane; c4 < C4; c4 += 64) {
        const float* px = feat_hwc + c4 * 4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int gx = POOL2 ? 2 * pw + (i & 1) : pw, gy = POOL2 ? 2 * ph + (i >> 1) : ph;
          const float4 v = warp_sample(px, H, W, C, x1s + (float)gx * bw, y1s + (float)gy * bh);
          o = i == 0 ? v : max4(o, v);
        }
        if (!SM || orow) *reinterpret_cast<float4*>(orow + c4 * 4) = o;
        if (SM) sm_store4<SM>(sm, R, r, ((long)ph * PW + pw) * C + c4 * 4, o);
      }
    }
  }
}

// ---- ROIWarping, one wave per output ROW (ROI_WARP_VARIANT = 3; round 5) ---------------------------------------------------------
// The wave kernel above gives every output position its own taps: 4 .. 9 pixel reads of a kilobyte per position and 256 channels,
// ~6x the bytes it writes, all through L2 -- that, not the 120 MB it writes, is what its 60 us are.  Neighbouring positions of a
// row sample neighbouring (mostly the same) feature-map columns: a wave that walks a whole output row (r, ph, 256 channels) keeps
// the last columns it read in registers (3 columns x 2-3 rows for the fused 28x28 + pool, 2 x 2 for the plain warp) and reads only
// the columns the next position adds: 3 + 27 bin-widths columns per row instead of 14 x 2-3.  Per position: the same set-up
// (warp_setup), the same four taps with the same weights in the same order, the same maximum order -- the same bits.  A position
// that the window does not serve (a tap outside the map, samples more than one cell apart) takes the wave kernel's own code and
// the window starts again behind it.
template <int NC, int NR>
struct WarpWindow {
  float4 g[NC][NR];                  // [column cx + i][row y0 + j]
  int cx;                            // first column held; INT_MIN: nothing
};

template <int NC, int NR>
__device__ __forceinline__ void warp_window_seek(WarpWindow<NC, NR>& w, const float* __restrict__ px, int W, int C, int y0, int x0) {
  // px = feature map + the lane's channel offset.  Columns past the last one are clamped (read again, never used: a position
  // that needed them would have a tap outside the map and does not come here).
  auto col = [&](int i, int x) {
    const float* p = px + ((long)y0 * W + min(x, W - 1)) * C;
#pragma unroll
    for (int j = 0; j < NR; ++j) w.g[i][j] = ld4(p + (long)j * W * C);
  };
  const int s = w.cx == INT_MIN ? NC : x0 - w.cx;          // wave-uniform
  if (s == 0) return;
  if (NC == 3 && s == 1) {
#pragma unroll
    for (int j = 0; j < NR; ++j) { w.g[0][j] = w.g[1][j]; w.g[1][j] = w.g[2][j]; }
    col(2, x0 + 2);
  } else if (NC == 3 && s == 2) {
#pragma unroll
    for (int j = 0; j < NR; ++j) w.g[0][j] = w.g[2][j];
    col(1, x0 + 1);
    col(2, x0 + 2);
  } else if (NC == 2 && s == 1) {
#pragma unroll
    for (int j = 0; j < NR; ++j) w.g[0][j] = w.g[1][j];
    col(1, x0 + 1);
  } else {
#pragma unroll
    for (int i = 0; i < NC; ++i) col(i, x0 + i);
  }
  w.cx = x0;
}

// What the row kernel pays for is VALU work, not bytes (the first version, with warp_setup per sample and scalar blends, gained
// 4 % from reading 3.4 instead of 6 pixels per position): so the set-up is split into its x and y halves -- the y half once per
// row, the x half once per sample COLUMN, the same expressions as warp_setup, value for value -- and the blend runs on packed
// fp32 (v_pk_mul_f32 / v_pk_add_f32: two channels per instruction, each lane of a packed operation rounds like the scalar one).
typedef float warp_f2 __attribute__((ext_vector_type(2)));
struct WarpAxis {                    // one coordinate of a sample: cell, fraction, 1 - fraction, validity of the two taps
  int i0;
  float a, oma;
  bool v0, v1;
};
__device__ __forceinline__ WarpAxis warp_axis(float s, int n) {
  WarpAxis r;
  r.i0 = (int)floorf(s);
  r.a = s - (float)r.i0;
  r.oma = 1.0f - r.a;
  r.v0 = r.i0 >= 0 && r.i0 < n;
  r.v1 = r.i0 + 1 >= 0 && r.i0 + 1 < n;
  return r;
}
// w00 * a00 + w01 * a01 + w10 * a10 + w11 * a11, left to right, on the two halves of a float4
__device__ __forceinline__ float4 warp_blend_pk(float w00, float w01, float w10, float w11, float4 a00, float4 a01, float4 a10,
                                                float4 a11) {
  auto lo = [](float4 v) { return warp_f2{v.x, v.y}; };
  auto hi = [](float4 v) { return warp_f2{v.z, v.w}; };
  const warp_f2 rl = ((w00 * lo(a00) + w01 * lo(a01)) + w10 * lo(a10)) + w11 * lo(a11);
  const warp_f2 rh = ((w00 * hi(a00) + w01 * hi(a01)) + w10 * hi(a10)) + w11 * hi(a11);
  return make_float4(rl.x, rl.y, rh.x, rh.y);
}

template <int POOL2, int DY, int SM>
__device__ __forceinline__ void warp_row_walk(const float* __restrict__ feat_hwc, int C, int H, int W, int PW, int ph, float x1s,
                                              float y1s, float bw, float bh, int c4, float* __restrict__ orow0, int pw0, int pw1,
                                              void* __restrict__ sm, long M, long r) {
  constexpr int NSX = POOL2 ? 2 : 1, NSY = POOL2 ? 2 : 1, NC = POOL2 ? 3 : 2, NR = 2 + (POOL2 ? DY : 0);
  const float* px = feat_hwc + c4 * 4;
  WarpWindow<NC, NR> win;
  win.cx = INT_MIN;
  // the sample rows of this output row
  WarpAxis ya[NSY];
  bool ysafe = true;
#pragma unroll
  for (int j = 0; j < NSY; ++j) {
    ya[j] = warp_axis(y1s + (float)(POOL2 ? 2 * ph + j : ph) * bh, H);
    ysafe = ysafe && ya[j].v0 && ya[j].v1;
  }
  const bool rows_ok = ysafe && (!POOL2 || ya[NSY - 1].i0 - ya[0].i0 == DY);
  for (int pw = pw0; pw < pw1; ++pw) {
    WarpAxis xa[NSX];
    bool xsafe = true;
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
      xa[i] = warp_axis(x1s + (float)(POOL2 ? 2 * pw + i : pw) * bw, W);
      xsafe = xsafe && xa[i].v0 && xa[i].v1;
    }
    const int dx = xa[NSX - 1].i0 - xa[0].i0;
    float* orow = orow0 + (long)pw * C;
    float4 o;
    if (rows_ok && xsafe && dx >= 0 && dx <= 1) {
      warp_window_seek<NC, NR>(win, px, W, C, ya[0].i0, xa[0].i0);
      // (sample order as in the wave kernel: x fastest.)  Columns of a sample's taps inside the window: 0 / 1, or 1 / 2 for the
      // right-hand sample of a pair one cell apart (dx is wave-uniform: a branch, not 32 selects per position); rows j DY, j DY + 1
      auto blend_all = [&](auto right_) {
        constexpr int RIGHT = decltype(right_)::value;
#pragma unroll
        for (int j = 0; j < NSY; ++j)
#pragma unroll
          for (int i = 0; i < NSX; ++i) {
            constexpr int oy0 = 0;
            const int oy = oy0 + j * (POOL2 ? DY : 0);
            const int cl = (RIGHT && i == 1) ? 1 : 0;
            const float4 a00 = win.g[cl][oy], a01 = win.g[cl + 1][oy], a10 = win.g[cl][oy + 1], a11 = win.g[cl + 1][oy + 1];
            const float w00 = xa[i].oma * ya[j].oma, w01 = xa[i].a * ya[j].oma, w10 = xa[i].oma * ya[j].a, w11 = xa[i].a * ya[j].a;
            const float4 v = warp_blend_pk(w00, w01, w10, w11, a00, a01, a10, a11);
            o = (i == 0 && j == 0) ? v : max4(o, v);
          }
      };
      if (NC == 3 && dx == 1) blend_all(std::integral_constant<int, NC == 3 ? 1 : 0>());
      else blend_all(std::integral_constant<int, 0>());
    } else {
      win.cx = INT_MIN;
#pragma unroll
      for (int j = 0; j < NSY; ++j)
#pragma unroll
        for (int i = 0; i < NSX; ++i) {
          const float4 v = warp_sample(px, H, W, C, x1s + (float)(POOL2 ? 2 * pw + i : pw) * bw, y1s + (float)(POOL2 ? 2 * ph + j : ph) * bh);
          o = (i == 0 && j == 0) ? v : max4(o, v);
        }
    }
    if (!SM || orow0) *reinterpret_cast<float4*>(orow + c4 * 4) = o;           // (with a stage-major output the fp32 one is optional)
    if (SM) sm_store4<SM>(sm, M, r, ((long)ph * PW + pw) * C + c4 * 4, o);
  }
}

// nseg: a row is walked by nseg waves, PW / nseg positions each (a wave's positions are a chain: each waits for its own loads)
// (three workgroups per CU for the variants with a second, stage-major output: at four the 128-register budget spilled seven)
template <int POOL2, int SM>
__global__ __launch_bounds__(256, SM ? 3 : 4) void roi_warp_row_kernel(const float* __restrict__ feat_hwc, int C, int H, int W,
                                                              const float* __restrict__ rois, int R, int PH, int PW, float scale,
                                                              float* __restrict__ out, int nseg, void* __restrict__ sm) {
  const int C4 = C >> 2, lane = threadIdx.x & 63;
  const int citers = ((C4 + 63) >> 6) * nseg;                      // 256-channel pieces of a row x its segments
  const unsigned nitems = (unsigned)R * PH * citers;
  const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
  const int GH = POOL2 ? 2 * PH : PH, GW = POOL2 ? 2 * PW : PW;
  for (unsigned it = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; it < nitems; it += nwaves) {
    const unsigned item = __builtin_amdgcn_readfirstlane(it);
    const int cs = (int)(item % (unsigned)citers);
    const int seg = cs % nseg, ci = cs / nseg;
    const int pw0 = seg * PW / nseg, pw1 = (seg + 1) * PW / nseg;
    const unsigned t = item / (unsigned)citers;
    const int ph = (int)(t % (unsigned)PH), r = (int)(t / (unsigned)PH);
    const int c4 = ci * 64 + lane;
    if (c4 >= C4) continue;                                        // (a ragged last piece: the lanes past C just sit the row out)
    const float* roi = rois + (long)r * 5;
    const float x1s = roi[1] * scale, y1s = roi[2] * scale, x2s = roi[3] * scale, y2s = roi[4] * scale;
    const float rw = fmaxf(x2s - x1s + 1.0f, 1.0f), rh = fmaxf(y2s - y1s + 1.0f, 1.0f);
    const float bw = rw / (float)GW, bh = rh / (float)GH;
    float* orow0 = out ? out + ((long)r * PH + ph) * PW * C : nullptr;
    // the two sample rows of a pooled output row are the same for every position of the row: one cell apart or not
    int dy = 0;
    if (POOL2) dy = (int)floorf(y1s + (float)(2 * ph + 1) * bh) - (int)floorf(y1s + (float)(2 * ph) * bh);
    if (POOL2 && dy == 1) warp_row_walk<POOL2, 1, SM>(feat_hwc, C, H, W, PW, ph, x1s, y1s, bw, bh, c4, orow0, pw0, pw1, sm, R, r);
    else warp_row_walk<POOL2, 0, SM>(feat_hwc, C, H, W, PW, ph, x1s, y1s, bw, bh, c4, orow0, pw0, pw1, sm, R, r);
  }
}

// ---- 8 channels per thread: the variant used when a second output is written (except the fused 28x28 warp).  A thread then owns
// a whole 16-byte group of the stage-major tensor, and the per-thread index arithmetic of the second output is spent once per 8
// channels: measured at 1000 RoIs x 1024 channels with the fp16 second output: 14x14 warp 430 us (4 channels per thread + lane
// exchange: 520; no second output: 432), MAX pool 175 (190; 174), MaskPooling + pool 184 (192; 173).  Same fp32 arithmetic.
template <int SM>
__device__ __forceinline__ void sm_store8(void* __restrict__ sm, long M, long r, long k, const float4 a, const float4 b) {
  if (SM == 1 || SM == 3) {
    const uint2 lo = SM == 3 ? x3_bf16x4(a) : x3_f16x4(a), hi = SM == 3 ? x3_bf16x4(b) : x3_f16x4(b);
    reinterpret_cast<uint4*>(sm)[((k >> 6) * M + r) * 8 + ((k & 63) >> 3)] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  } else {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint4 hi, lo;
    x3_split8_rne(x, hi, lo);
    uint4* p = reinterpret_cast<uint4*>(sm) + (((k >> 5) * M + r) * 4 + ((k & 31) >> 3)) * 2;
    p[0] = hi;
    p[1] = lo;
  }
}

template <int POOL2, int SM>
__global__ __launch_bounds__(256) void roi_warp8_kernel(const float* __restrict__ feat_hwc, int C, int H, int W,
                                                        const float* __restrict__ rois, int R, int PH, int PW, float scale,
                                                        float* __restrict__ out, void* __restrict__ sm) {
  const unsigned C8 = (unsigned)C >> 3;
  const unsigned total = (unsigned)R * PH * PW * C8;
  const int GH = POOL2 ? 2 * PH : PH, GW = POOL2 ? 2 * PW : PW;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % C8);
    unsigned t = idx / C8;
    const int pw = (int)(t % (unsigned)PW);
    t /= (unsigned)PW;
    const int ph = (int)(t % (unsigned)PH);
    const int r = (int)(t / (unsigned)PH);
    const float* roi = rois + (long)r * 5;
    const float x1s = roi[1] * scale, y1s = roi[2] * scale, x2s = roi[3] * scale, y2s = roi[4] * scale;
    const float rw = fmaxf(x2s - x1s + 1.0f, 1.0f), rh = fmaxf(y2s - y1s + 1.0f, 1.0f);
    const float bw = rw / (float)GW, bh = rh / (float)GH;
    float4 o[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* px = feat_hwc + c8 * 8 + h * 4;
      if (POOL2) {
        o[h] = warp_sample(px, H, W, C, x1s + (float)(2 * pw) * bw, y1s + (float)(2 * ph) * bh);
        o[h] = max4(o[h], warp_sample(px, H, W, C, x1s + (float)(2 * pw + 1) * bw, y1s + (float)(2 * ph) * bh));
        o[h] = max4(o[h], warp_sample(px, H, W, C, x1s + (float)(2 * pw) * bw, y1s + (float)(2 * ph + 1) * bh));
        o[h] = max4(o[h], warp_sample(px, H, W, C, x1s + (float)(2 * pw + 1) * bw, y1s + (float)(2 * ph + 1) * bh));
      } else {
        o[h] = warp_sample(px, H, W, C, x1s + (float)pw * bw, y1s + (float)ph * bh);
      }
    }
    float4* dst = reinterpret_cast<float4*>(out + (long)idx * 8);      // == (((r*PH + ph)*PW + pw)*C + c8*8)
    dst[0] = o[0];
    dst[1] = o[1];
    sm_store8<SM>(sm, R, r, ((long)ph * PW + pw) * C + c8 * 8, o[0], o[1]);
  }
}

template <int SM>
__global__ __launch_bounds__(256) void maxpool2_rhwc8_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int PH,
                                                             int PW, int C8, void* __restrict__ sm) {
  const int OH = PH / 2, OW = PW / 2;
  const unsigned total = (unsigned)R * OH * OW * C8;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % (unsigned)C8);
    unsigned t = idx / (unsigned)C8;
    const int ow = (int)(t % (unsigned)OW);
    t /= (unsigned)OW;
    const int oh = (int)(t % (unsigned)OH);
    const long r = t / (unsigned)OH;
    float4 m[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* p = in + (((r * PH + 2 * oh) * PW + 2 * ow) * C8 + c8) * 8 + h * 4;
      m[h] = max4(ld4(p), ld4(p + (long)C8 * 8));
      m[h] = max4(m[h], ld4(p + (long)PW * C8 * 8));
      m[h] = max4(m[h], ld4(p + (long)(PW + 1) * C8 * 8));
    }
    float4* dst = reinterpret_cast<float4*>(out + (long)idx * 8);
    dst[0] = m[0];
    dst[1] = m[1];
    sm_store8<SM>(sm, R, r, ((long)oh * OW + ow) * C8 * 8 + c8 * 8, m[0], m[1]);
  }
}

template <int POOL2, int SM>
__global__ __launch_bounds__(256) void mask_pool8_kernel(const float* __restrict__ feat, const float* __restrict__ mask,
                                                         float* __restrict__ out, int R, int PH, int PW, int C8,
                                                         void* __restrict__ sm, int bin_on, float bin_thr) {
  const int OH = POOL2 ? PH / 2 : PH, OW = POOL2 ? PW / 2 : PW;
  const unsigned total = (unsigned)R * OH * OW * C8;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % (unsigned)C8);
    unsigned t = idx / (unsigned)C8;
    const int ow = (int)(t % (unsigned)OW);
    t /= (unsigned)OW;
    const int oh = (int)(t % (unsigned)OH);
    const long r = t / (unsigned)OH;
    float4 v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      auto prod = [&](int y, int x) {
        const float mk = mask_value(mask[(r * PH + y) * PW + x], bin_on, bin_thr);
        const float4 f = ld4(feat + (((r * PH + y) * PW + x) * C8 + c8) * 8 + h * 4);
        return make_float4(f.x * mk, f.y * mk, f.z * mk, f.w * mk);
      };
      if (POOL2)
        v[h] = max4(max4(prod(2 * oh, 2 * ow), prod(2 * oh, 2 * ow + 1)), max4(prod(2 * oh + 1, 2 * ow), prod(2 * oh + 1, 2 * ow + 1)));
      else
        v[h] = prod(oh, ow);
    }
    float4* dst = reinterpret_cast<float4*>(out + (long)idx * 8);
    dst[0] = v[0];
    dst[1] = v[1];
    sm_store8<SM>(sm, R, r, ((long)oh * OW + ow) * C8 * 8 + c8 * 8, v[0], v[1]);
  }
}

// ---- box-feature MAX pool and MaskPooling (+ its MAX pool) of the same 14x14 tensor in ONE pass (test.prototxt:571-582 and
// :631-650): both read every value of `feat` once -- 803 MB at 1000 RoIs x 1024 channels -- so running them as two kernels
// reads it twice.  out_box = the Pooling layer's output, out_mask = MaskPooling + Pooling; the arithmetic per output is that
// of maxpool2_rhwc_kernel / mask_pool_kernel<1> (max of the four values resp. of the four products, same order).  NV float4 per
// thread (2 when second outputs are written: a whole 16-byte group of the stage-major tensors, see above).
template <int SM, int NV>
__global__ __launch_bounds__(256) void box_mask_pool_kernel(const float* __restrict__ feat, const float* __restrict__ mask,
                                                            float* __restrict__ out_box, float* __restrict__ out_mask, int R,
                                                            int PH, int PW, int CV, void* __restrict__ sm_box,
                                                            void* __restrict__ sm_mask, int bin_on, float bin_thr) {
  const int OH = PH / 2, OW = PW / 2;
  const unsigned total = (unsigned)R * OH * OW * CV;              // CV = channel groups of 4*NV; < 2^31: checked by the launcher
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int cv = (int)(idx % (unsigned)CV);
    unsigned t = idx / (unsigned)CV;
    const int ow = (int)(t % (unsigned)OW);
    t /= (unsigned)OW;
    const int oh = (int)(t % (unsigned)OH);
    const long r = t / (unsigned)OH;
    const long m00 = (r * PH + 2 * oh) * PW + 2 * ow;               // window's first position in the [R][PH][PW] grids
    const float k00 = mask_value(mask[m00], bin_on, bin_thr), k01 = mask_value(mask[m00 + 1], bin_on, bin_thr),
                k10 = mask_value(mask[m00 + PW], bin_on, bin_thr), k11 = mask_value(mask[m00 + PW + 1], bin_on, bin_thr);
    float4 b[NV], v[NV];
#pragma unroll
    for (int h = 0; h < NV; ++h) {
      const float* p = feat + (m00 * CV + cv) * (4 * NV) + h * 4;
      const float4 f00 = ld4(p), f01 = ld4(p + (long)CV * 4 * NV), f10 = ld4(p + (long)PW * CV * 4 * NV),
                   f11 = ld4(p + (long)(PW + 1) * CV * 4 * NV);
      b[h] = max4(max4(max4(f00, f01), f10), f11);                  // maxpool2_rhwc_kernel's order
      auto mul = [](const float4 f, float k) { return make_float4(f.x * k, f.y * k, f.z * k, f.w * k); };
      v[h] = max4(max4(mul(f00, k00), mul(f01, k01)), max4(mul(f10, k10), mul(f11, k11)));      // mask_pool_kernel<1>'s order
    }
    if (!SM || out_box) {                            // (round 6: a caller whose InnerProducts read the stage-major forms only passes no fp32 outputs)
      float4* db = reinterpret_cast<float4*>(out_box + (long)idx * 4 * NV);
      float4* dm = reinterpret_cast<float4*>(out_mask + (long)idx * 4 * NV);
#pragma unroll
      for (int h = 0; h < NV; ++h) { db[h] = b[h]; dm[h] = v[h]; }
    }
    if (SM) {
      const long k = ((long)oh * OW + ow) * CV * 4 * NV + cv * 4 * NV;
      sm_store8<SM>(sm_box, R, r, k, b[0], b[NV - 1]);
      sm_store8<SM>(sm_mask, R, r, k, v[0], v[NV - 1]);
    }
  }
}

// The same pass reading the 14x14 tensor from its STAGE-MAJOR fp16 form (round 6: what the ROIWarping launch wrote for fc6_maskest,
// [K/64][R][64] halves, K = PH x PW x C) instead of the fp32 tensor -- 60 instead of 120 MB at 300 RoIs x 512 channels, and the warp
// need not write the fp32 tensor at all.  One thread = one 16-byte group (8 channels) of an output position.  The box pool is the
// same bits as before (rounding to fp16 is monotonic: the max of the rounded values is the rounded max); MaskPooling multiplies
// the ROUNDED features by the mask, so its outputs differ from the fp32-input pass in the last fp16 bit -- every executor of a graph
// reads the same form (pipeline.hip: run_stage; engine.py: Pooling with_mask), so they agree bit for bit.
// IN = 2: the split-bf16 stage-major form ([K/32][R][4][hi x8 | lo x8], the `mixed` mode's fc6_maskest input): a value is hi + lo, the
// fp32 value rounded to 16 significant bits, so here the box pool too can differ from the fp32-input pass in the last fp16 bit.
template <int SM, int IN>
__global__ __launch_bounds__(256) void box_mask_pool_h_kernel(const uint4* __restrict__ feat_sm, const float* __restrict__ mask,
                                                              float* __restrict__ out_box, float* __restrict__ out_mask, int R,
                                                              int PH, int PW, int C8, void* __restrict__ sm_box,
                                                              void* __restrict__ sm_mask, int bin_on, float bin_thr) {
  const int OH = PH / 2, OW = PW / 2;
  const unsigned total = (unsigned)R * OH * OW * C8;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % (unsigned)C8);
    unsigned t = idx / (unsigned)C8;
    const int ow = (int)(t % (unsigned)OW);
    t /= (unsigned)OW;
    const int oh = (int)(t % (unsigned)OH);
    const long r = t / (unsigned)OH;
    const long m00 = (r * PH + 2 * oh) * PW + 2 * ow;
    const float k00 = mask_value(mask[m00], bin_on, bin_thr), k01 = mask_value(mask[m00 + 1], bin_on, bin_thr),
                k10 = mask_value(mask[m00 + PW], bin_on, bin_thr), k11 = mask_value(mask[m00 + PW + 1], bin_on, bin_thr);
    auto load = [&](int y, int x, float4& lo, float4& hi) {
      const long k = ((long)(y * PW + x) * C8 + c8) * 8;           // K index of the group's first channel in the roi's row
      if (IN == 1) {
        const uint4 u = feat_sm[((k >> 6) * R + r) * 8 + ((k & 63) >> 3)];
        const f16x8 h = __builtin_bit_cast(f16x8, u);
        lo = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        hi = make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7]);
      } else if (IN == 3) {                          // bf16 in fp16's layout: element 2i in the low half of word i
        const uint4 u = feat_sm[((k >> 6) * R + r) * 8 + ((k & 63) >> 3)];
        auto b = [](unsigned w, int odd) { return __uint_as_float(odd ? (w & 0xFFFF0000u) : (w << 16)); };
        lo = make_float4(b(u.x, 0), b(u.x, 1), b(u.y, 0), b(u.y, 1));
        hi = make_float4(b(u.z, 0), b(u.z, 1), b(u.w, 0), b(u.w, 1));
      } else {
        const uint4* p = feat_sm + (((k >> 5) * R + r) * 4 + ((k & 31) >> 3)) * 2;
        const uint4 a = p[0], b = p[1];                              // hi x8, lo x8 (bf16 pairs: element 2i in the low half of word i)
        auto val = [](unsigned wh, unsigned wl, int odd) {
          return __uint_as_float(odd ? (wh & 0xFFFF0000u) : (wh << 16)) + __uint_as_float(odd ? (wl & 0xFFFF0000u) : (wl << 16));
        };
        lo = make_float4(val(a.x, b.x, 0), val(a.x, b.x, 1), val(a.y, b.y, 0), val(a.y, b.y, 1));
        hi = make_float4(val(a.z, b.z, 0), val(a.z, b.z, 1), val(a.w, b.w, 0), val(a.w, b.w, 1));
      }
    };
    float4 f[4][2];
    load(2 * oh, 2 * ow, f[0][0], f[0][1]);
    load(2 * oh, 2 * ow + 1, f[1][0], f[1][1]);
    load(2 * oh + 1, 2 * ow, f[2][0], f[2][1]);
    load(2 * oh + 1, 2 * ow + 1, f[3][0], f[3][1]);
    auto mul = [](const float4 a, float k) { return make_float4(a.x * k, a.y * k, a.z * k, a.w * k); };
    float4 b[2], v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      b[h] = max4(max4(max4(f[0][h], f[1][h]), f[2][h]), f[3][h]);                                           // maxpool2_rhwc_kernel's order
      v[h] = max4(max4(mul(f[0][h], k00), mul(f[1][h], k01)), max4(mul(f[2][h], k10), mul(f[3][h], k11)));   // mask_pool_kernel<1>'s
    }
    if (out_box) {
      float4* db = reinterpret_cast<float4*>(out_box + (long)idx * 8);
      float4* dm = reinterpret_cast<float4*>(out_mask + (long)idx * 8);
      db[0] = b[0]; db[1] = b[1]; dm[0] = v[0]; dm[1] = v[1];
    }
    const long k = ((long)oh * OW + ow) * C8 * 8 + c8 * 8;
    sm_store8<SM>(sm_box, R, r, k, b[0], b[1]);
    sm_store8<SM>(sm_mask, R, r, k, v[0], v[1]);
  }
}

// which variant of the two pooling kernels writes the second output: 8 channels per thread (MNC_ROI_SM_VARIANT=4 forces the other)
static bool sm_variant8(const mnc_ctx* ctx) { return tune(ctx, T_ROI_SM_VARIANT, 8) == 8; }

static bool sm_ok(int C, int fmt) { return (fmt == 1 || fmt == 3) ? C % 64 == 0 : fmt == 2 ? C % 32 == 0 : false; }

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

}  // extern "C"

namespace mnc {
// pixel-major copy [H][W][C] of a c8 feature map: what the warp kernels gather from (a pixel's channels contiguous).  The
// whole-image pipeline makes it ONCE per image for both head stages (pipeline.hip); mnc_roi_warp_sm makes its own per call.
int c8_to_hwc_launch(mnc_ctx* ctx, const float* d_feat, float* d_hwc, int C, int H, int W) {
  LaunchScope lt(ctx, "c8_to_hwc", 0.0, 8.0 * C * (double)H * W);
  hipLaunchKernelGGL(c8_to_hwc_kernel, dim3(grid_for((long)H * W * (C / 8) * 2)), dim3(256), 0, ctx->stream, d_feat, d_hwc, C / 8,
                     (long)H * W);
  return lt.finish("c8_to_hwc_kernel");
}
static int roi_warp_impl(mnc_ctx* ctx, const float* d_feat, const float* d_hwc_ready, int C, int H, int W, const float* d_rois, int R,
                         int PH, int PW, float scale, int pool2, float* d_out, void* d_sm, int sm_fmt);
// The fp32 output of a warp that also writes the stage-major form may be omitted on the kernels the SPEC's convention runs by
// default (roi_warp_row_kernel for the fused 28x28 + pool, roi_warp_kernel / roi_warp_wave_kernel for the plain warp below / from
// 1024 channels); not on the 8-channels-per-thread variant and the generic-convention kernel.
bool roi_warp_sm_only_ok(const mnc_ctx* ctx, int C, int pool2) {
  if (ctx->conv.warp_sample || ctx->conv.warp_round_edges || ctx->conv.warp_no_plus_one || ctx->conv.warp_oob) return false;
  const int vsel = tune(ctx, T_ROI_WARP_VARIANT, (pool2 || C >= 1024) ? 3 : 4);
  return vsel == 3 || vsel == 4 || vsel == 1;
}
int roi_warp_from_hwc(mnc_ctx* ctx, const float* d_hwc, int C, int H, int W, const float* d_rois, int R, int PH, int PW, float scale,
                      int pool2, float* d_out, void* d_sm, int sm_fmt) {
  return roi_warp_impl(ctx, nullptr, d_hwc, C, H, W, d_rois, R, PH, PW, scale, pool2, d_out, d_sm, sm_fmt);
}
}  // namespace mnc

extern "C" {

int mnc_roi_warp_sm_only_ok(mnc_ctx* ctx, int C, int pool2, int* ok) {
  MNC_REQUIRE(ctx && ok, "mnc_roi_warp_sm_only_ok: null pointer");
  *ok = mnc::roi_warp_sm_only_ok(ctx, C, pool2) ? 1 : 0;
  mnc::clear_error();
  return MNC_OK;
}

int mnc_roi_warp_sm(mnc_ctx* ctx, const float* d_feat, int C, int H, int W, const float* d_rois, int R, int PH, int PW,
                    float scale, int pool2, float* d_out, void* d_sm, int sm_fmt) {
  MNC_REQUIRE(d_feat, "mnc_roi_warp: null pointer");
  return mnc::roi_warp_impl(ctx, d_feat, nullptr, C, H, W, d_rois, R, PH, PW, scale, pool2, d_out, d_sm, sm_fmt);
}

}  // extern "C"

namespace mnc {
static int roi_warp_impl(mnc_ctx* ctx, const float* d_feat, const float* d_hwc_ready, int C, int H, int W, const float* d_rois, int R,
                         int PH, int PW, float scale, int pool2, float* d_out, void* d_sm, int sm_fmt) {
  MNC_REQUIRE(ctx && (d_feat || d_hwc_ready) && (d_out || (d_sm && sm_fmt)) && (R == 0 || d_rois), "mnc_roi_warp: null pointer");
  MNC_REQUIRE(d_out || roi_warp_sm_only_ok(ctx, C, pool2), "mnc_roi_warp: this variant / convention writes the fp32 tensor");
  MNC_REQUIRE(C > 0 && C % 8 == 0 && H > 0 && W > 0 && R >= 0 && PH > 0 && PW > 0, "mnc_roi_warp: bad shape");
  if (!d_sm) sm_fmt = 0;
  MNC_REQUIRE(sm_fmt == 0 || sm_ok(C, sm_fmt), "mnc_roi_warp_sm: format %d needs C %% %d == 0 (C = %d)", sm_fmt,
              (sm_fmt == 1 || sm_fmt == 3) ? 64 : 32, C);
  if (R == 0) return MNC_OK;
  const long total = (long)R * PH * PW * (C / 4);
  MNC_REQUIRE(total < (1L << 31), "mnc_roi_warp: %ld outputs exceed the kernel's 32-bit index range", total * 4);
  const double samples = pool2 ? 4.0 : 1.0;
  // pixel-major copy of the feature map: the caller's, or made here in the context's scratch arena (same stream: ordered after
  // any earlier user)
  int rc = MNC_OK;
  const float* d_hwc = d_hwc_ready;
  if (!d_hwc) {
    rc = ensure_scratch(ctx, (size_t)C * H * W * 4);
    if (rc) return rc;
    rc = c8_to_hwc_launch(ctx, d_feat, (float*)ctx->scratch, C, H, W);
    if (rc) return rc;
    d_hwc = (const float*)ctx->scratch;
  }
  LaunchScope ls(ctx, pool2 ? "roi_warp_pool2" : "roi_warp", 0.0,
                 4.0 * ((double)R * PH * PW * C * (1.0 + 4.0 * samples)) + ((sm_fmt == 1 || sm_fmt == 3) ? 2.0 : sm_fmt == 2 ? 4.0 : 0.0) * R * PH * PW * C);
  // One wave per output position for the fused 28x28 warp + pool (its set-up is four samples' worth and the window's taps are
  // shared), and for the plain warp from 1024 channels on (4+ channel iterations share a position's set-up).  Measured, fp16
  // second output, 1000 RoIs x 1024 channels: 28x28+pool 452 us against 629 / 680 for the 4- / 8-channels-per-thread kernels,
  // 14x14 407 against 520 / 430; fp32 only, 300 RoIs x 512 channels: 28x28+pool 61 against 68, 14x14 44 against 32 (so the
  // 4-channels-per-thread kernel keeps that case).  MNC_ROI_WARP_VARIANT = 1 (wave) / 3 (wave per output row) / 4 / 8 forces one.
  // Round 6: from 1024 channels on the plain warp runs on the row kernel too (ResNet-50 configuration, stage-major output only,
  // four images in flight: 291 -> 296 images/s against the wave kernel; same bits).
  if (ctx->conv.warp_sample || ctx->conv.warp_round_edges || ctx->conv.warp_no_plus_one || ctx->conv.warp_oob) {
    // a convention other than the SPEC's: the generic kernel (see roi_warp_conv_kernel)
#define MNC_WARPC(P2, SM)                                                                                                       \
  hipLaunchKernelGGL((roi_warp_conv_kernel<P2, SM>), dim3(grid_for(total)), dim3(256), 0, ctx->stream, d_hwc, C, H, W, d_rois, R, PH, \
                     PW, scale, ctx->conv, d_out, d_sm)
    if (pool2) { if (sm_fmt == 1) MNC_WARPC(1, 1); else if (sm_fmt == 2) MNC_WARPC(1, 2); else if (sm_fmt == 3) MNC_WARPC(1, 3); else MNC_WARPC(1, 0); }
    else { if (sm_fmt == 1) MNC_WARPC(0, 1); else if (sm_fmt == 2) MNC_WARPC(0, 2); else if (sm_fmt == 3) MNC_WARPC(0, 3); else MNC_WARPC(0, 0); }
#undef MNC_WARPC
    return ls.finish("roi_warp_conv_kernel");
  }
  // the fused 28x28 + pool: one wave per half output row (roi_warp_row_kernel; in the pipeline's serial trace,
  // 300 RoIs x 512 channels: 59.8 -> 46.2 us; the plain 14x14 warp gains nothing from it -- 33.0 against 34.1 -- and keeps its kernel).
  // profiles/r05_roi_warp_row.txt.  MNC_ROI_WARP_VARIANT = 3 forces the row kernel, MNC_ROI_ROW_SEGS the waves per row.
  // (with a stage-major output the row kernel for the plain warp below 1024 channels as well: f16 1056 -> 1069 images/s, mixed
  // 633 -> 637, two runs each; fp32 -- no second output -- 272.8 against 272.9: the 4-channels-per-thread kernel stays there)
  const int vsel = tune(ctx, T_ROI_WARP_VARIANT, (pool2 || C >= 1024 || sm_fmt != 0) ? 3 : 4);
  if (vsel == 3) {                                                 // one wave per (half) output row
    MNC_REQUIRE((double)H * W * C < 2.0e9, "mnc_roi_warp: feature map too large for 32-bit offsets");
    int nseg = tune(ctx, T_ROI_ROW_SEGS, 2);
    if (nseg < 1 || nseg > PW) nseg = 1;
    const int g = grid_for((long)R * PH * ((C / 4 + 63) / 64) * nseg * 64);
#define MNC_WARPR(P2, SM)                                                                                                   \
  hipLaunchKernelGGL((roi_warp_row_kernel<P2, SM>), dim3(g), dim3(256), 0, ctx->stream, d_hwc, C, H, W, d_rois, R, PH, PW, scale, \
                     d_out, nseg, d_sm)
    if (pool2) { if (sm_fmt == 1) MNC_WARPR(1, 1); else if (sm_fmt == 2) MNC_WARPR(1, 2); else if (sm_fmt == 3) MNC_WARPR(1, 3); else MNC_WARPR(1, 0); }
    else { if (sm_fmt == 1) MNC_WARPR(0, 1); else if (sm_fmt == 2) MNC_WARPR(0, 2); else if (sm_fmt == 3) MNC_WARPR(0, 3); else MNC_WARPR(0, 0); }
#undef MNC_WARPR
    return ls.finish("roi_warp_row_kernel");
  }
  if (vsel != 4 && vsel != 8) {
    MNC_REQUIRE((double)H * W * C < 2.0e9, "mnc_roi_warp: feature map too large for 32-bit offsets");
    const int g = grid_for((long)R * PH * PW * 64);
#define MNC_WARPW(P2, SM)                                                                                                    \
  hipLaunchKernelGGL((roi_warp_wave_kernel<P2, SM>), dim3(g), dim3(256), 0, ctx->stream, d_hwc, C, H, W, d_rois, R, PH, PW, scale, \
                     d_out, d_sm)
    if (pool2) { if (sm_fmt == 1) MNC_WARPW(1, 1); else if (sm_fmt == 2) MNC_WARPW(1, 2); else if (sm_fmt == 3) MNC_WARPW(1, 3); else MNC_WARPW(1, 0); }
    else { if (sm_fmt == 1) MNC_WARPW(0, 1); else if (sm_fmt == 2) MNC_WARPW(0, 2); else if (sm_fmt == 3) MNC_WARPW(0, 3); else MNC_WARPW(0, 0); }
#undef MNC_WARPW
    return ls.finish("roi_warp_wave_kernel");
  }
  if (sm_fmt && vsel == 8) {
#define MNC_WARP8(P2, SM)                                                                                                   \
  hipLaunchKernelGGL((roi_warp8_kernel<P2, SM>), dim3(grid_for(total / 2)), dim3(256), 0, ctx->stream, d_hwc, C, H, W, d_rois, R, \
                     PH, PW, scale, d_out, d_sm)
    if (pool2) { if (sm_fmt == 1) MNC_WARP8(1, 1); else if (sm_fmt == 3) MNC_WARP8(1, 3); else MNC_WARP8(1, 2); }
    else { if (sm_fmt == 1) MNC_WARP8(0, 1); else if (sm_fmt == 3) MNC_WARP8(0, 3); else MNC_WARP8(0, 2); }
#undef MNC_WARP8
    return ls.finish("roi_warp8_kernel");
  }
#define MNC_WARP(P2, SM)                                                                                                \
  hipLaunchKernelGGL((roi_warp_kernel<P2, SM>), dim3(grid_for(total)), dim3(256), 0, ctx->stream, d_hwc, C, H, W, d_rois, R, PH, \
                     PW, scale, d_out, d_sm)
  if (pool2) { if (sm_fmt == 1) MNC_WARP(1, 1); else if (sm_fmt == 2) MNC_WARP(1, 2); else if (sm_fmt == 3) MNC_WARP(1, 3); else MNC_WARP(1, 0); }
  else { if (sm_fmt == 1) MNC_WARP(0, 1); else if (sm_fmt == 2) MNC_WARP(0, 2); else if (sm_fmt == 3) MNC_WARP(0, 3); else MNC_WARP(0, 0); }
#undef MNC_WARP
  return ls.finish("roi_warp_kernel");
}
}  // namespace mnc

extern "C" {

int mnc_roi_warp(mnc_ctx* ctx, const float* d_feat, int C, int H, int W, const float* d_rois, int R, int PH, int PW,
                 float scale, int pool2, float* d_out) {
  return mnc_roi_warp_sm(ctx, d_feat, C, H, W, d_rois, R, PH, PW, scale, pool2, d_out, nullptr, 0);
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

int mnc_maxpool2_rhwc_sm(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int PH, int PW, int C, void* d_sm, int sm_fmt) {
  MNC_REQUIRE(ctx && d_in && d_out && R >= 0 && PH > 0 && PW > 0 && PH % 2 == 0 && PW % 2 == 0 && C > 0 && C % 4 == 0,
              "mnc_maxpool2_rhwc: bad argument (PH, PW must be even, C%%4==0)");
  if (!d_sm) sm_fmt = 0;
  MNC_REQUIRE(sm_fmt == 0 || sm_ok(C, sm_fmt), "mnc_maxpool2_rhwc_sm: format %d needs C %% %d == 0 (C = %d)", sm_fmt,
              (sm_fmt == 1 || sm_fmt == 3) ? 64 : 32, C);
  if (R == 0) return MNC_OK;
  MNC_REQUIRE((long)R * PH * PW * (C / 4) < (1L << 31), "mnc_maxpool2_rhwc: tensor exceeds the kernel's 32-bit index range");
  LaunchScope ls(ctx, "maxpool2_rhwc", 0.0, 4.0 * R * (double)C * PH * PW * 1.25);
  if (sm_fmt && sm_variant8(ctx)) {
    const int g8 = grid_for((long)R * (PH / 2) * (PW / 2) * (C / 8));
    if (sm_fmt == 1) hipLaunchKernelGGL(maxpool2_rhwc8_kernel<1>, dim3(g8), dim3(256), 0, ctx->stream, d_in, d_out, R, PH, PW, C / 8, d_sm);
    else if (sm_fmt == 3) hipLaunchKernelGGL(maxpool2_rhwc8_kernel<3>, dim3(g8), dim3(256), 0, ctx->stream, d_in, d_out, R, PH, PW, C / 8, d_sm);
    else hipLaunchKernelGGL(maxpool2_rhwc8_kernel<2>, dim3(g8), dim3(256), 0, ctx->stream, d_in, d_out, R, PH, PW, C / 8, d_sm);
    return ls.finish("maxpool2_rhwc8_kernel");
  }
#define MNC_POOLR(SM)                                                                                                  \
  hipLaunchKernelGGL(maxpool2_rhwc_kernel<SM>, dim3(grid_for((long)R * (PH / 2) * (PW / 2) * (C / 4))), dim3(256), 0, ctx->stream, \
                     d_in, d_out, R, PH, PW, C / 4, d_sm)
  if (sm_fmt == 1) MNC_POOLR(1); else if (sm_fmt == 2) MNC_POOLR(2); else if (sm_fmt == 3) MNC_POOLR(3); else MNC_POOLR(0);
#undef MNC_POOLR
  return ls.finish("maxpool2_rhwc_kernel");
}

int mnc_maxpool2_rhwc(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int PH, int PW, int C) {
  return mnc_maxpool2_rhwc_sm(ctx, d_in, d_out, R, PH, PW, C, nullptr, 0);
}

int mnc_mask_resize(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int IH, int IW, int OH, int OW) {
  MNC_REQUIRE(ctx && d_in && d_out && R >= 0 && IH > 1 && IW > 1 && OH > 0 && OW > 0, "mnc_mask_resize: bad argument");
  if (R == 0) return MNC_OK;
  LaunchScope ls(ctx, "mask_resize");
  hipLaunchKernelGGL(mask_resize_kernel, dim3(cdiv((long)R * OH * OW, 256)), dim3(256), 0, ctx->stream, d_in, d_out, R, IH,
                     IW, OH, OW, ctx->conv.resize_mode);
  return ls.finish("mask_resize_kernel");
}

int mnc_mask_pool_sm(mnc_ctx* ctx, const float* d_feat, const float* d_mask, float* d_out, int R, int PH, int PW, int C,
                     int pool2, void* d_sm, int sm_fmt) {
  MNC_REQUIRE(ctx && d_feat && d_mask && d_out && R >= 0 && PH > 0 && PW > 0 && C > 0 && C % 4 == 0,
              "mnc_mask_pool: bad argument");
  MNC_REQUIRE(!pool2 || (PH % 2 == 0 && PW % 2 == 0), "mnc_mask_pool: pool2 needs even PH, PW");
  if (!d_sm) sm_fmt = 0;
  MNC_REQUIRE(sm_fmt == 0 || sm_ok(C, sm_fmt), "mnc_mask_pool_sm: format %d needs C %% %d == 0 (C = %d)", sm_fmt,
              (sm_fmt == 1 || sm_fmt == 3) ? 64 : 32, C);
  if (R == 0) return MNC_OK;
  MNC_REQUIRE((long)R * PH * PW * (C / 4) < (1L << 31), "mnc_mask_pool: tensor exceeds the kernel's 32-bit index range");
  const int OH = pool2 ? PH / 2 : PH, OW = pool2 ? PW / 2 : PW;
  LaunchScope ls(ctx, pool2 ? "mask_pool_pool2" : "mask_pool", 0.0, 4.0 * R * (double)C * (PH * PW + OH * OW));
  const long total = (long)R * OH * OW * (C / 4);
  if (sm_fmt && sm_variant8(ctx)) {
#define MNC_MP8(P2, SM)                                                                                                     \
  hipLaunchKernelGGL((mask_pool8_kernel<P2, SM>), dim3(grid_for(total / 2)), dim3(256), 0, ctx->stream, d_feat, d_mask, d_out, R, \
                     PH, PW, C / 8, d_sm, ctx->conv.maskpool_binary, ctx->conv.maskpool_thresh)
    if (pool2) { if (sm_fmt == 1) MNC_MP8(1, 1); else if (sm_fmt == 3) MNC_MP8(1, 3); else MNC_MP8(1, 2); }
    else { if (sm_fmt == 1) MNC_MP8(0, 1); else if (sm_fmt == 3) MNC_MP8(0, 3); else MNC_MP8(0, 2); }
#undef MNC_MP8
    return ls.finish("mask_pool8_kernel");
  }
#define MNC_MP(P2, SM)                                                                                                  \
  hipLaunchKernelGGL((mask_pool_kernel<P2, SM>), dim3(grid_for(total)), dim3(256), 0, ctx->stream, d_feat, d_mask, d_out, R, PH, \
                     PW, C / 4, d_sm, ctx->conv.maskpool_binary, ctx->conv.maskpool_thresh)
  if (pool2) { if (sm_fmt == 1) MNC_MP(1, 1); else if (sm_fmt == 2) MNC_MP(1, 2); else if (sm_fmt == 3) MNC_MP(1, 3); else MNC_MP(1, 0); }
  else { if (sm_fmt == 1) MNC_MP(0, 1); else if (sm_fmt == 2) MNC_MP(0, 2); else if (sm_fmt == 3) MNC_MP(0, 3); else MNC_MP(0, 0); }
#undef MNC_MP
  return ls.finish("mask_pool_kernel");
}

int mnc_mask_pool(mnc_ctx* ctx, const float* d_feat, const float* d_mask, float* d_out, int R, int PH, int PW, int C,
                  int pool2) {
  return mnc_mask_pool_sm(ctx, d_feat, d_mask, d_out, R, PH, PW, C, pool2, nullptr, 0);
}

int mnc_box_mask_pool(mnc_ctx* ctx, const float* d_feat, const float* d_mask, float* d_box_out, float* d_mask_out, int R, int PH,
                      int PW, int C, void* d_box_sm, void* d_mask_sm, int sm_fmt) {
  return mnc_box_mask_pool_ex(ctx, d_feat, nullptr, 0, d_mask, d_box_out, d_mask_out, R, PH, PW, C, d_box_sm, d_mask_sm, sm_fmt);
}

int mnc_box_mask_pool_ex(mnc_ctx* ctx, const float* d_feat, const void* d_feat_sm, int feat_sm_fmt, const float* d_mask, float* d_box_out,
                         float* d_mask_out, int R, int PH, int PW, int C, void* d_box_sm, void* d_mask_sm, int sm_fmt) {
  if (feat_sm_fmt < 1 || feat_sm_fmt > 3 || !d_box_sm || !d_mask_sm || sm_fmt == 0) d_feat_sm = nullptr;      // (read only on the way to stage-major outputs)
  MNC_REQUIRE(ctx && (d_feat || d_feat_sm) && d_mask && (d_box_out != nullptr) == (d_mask_out != nullptr) && R >= 0 && PH > 0 && PW > 0 &&
                  PH % 2 == 0 && PW % 2 == 0 && C > 0 && C % 8 == 0,
              "mnc_box_mask_pool: bad argument (PH, PW even, C%%8==0)");
  if (!d_box_sm || !d_mask_sm) sm_fmt = 0;
  // the fp32 outputs may be omitted (both null) when the stage-major ones are written: 60 of 210 MB per call at 300 RoIs x 512 channels
  MNC_REQUIRE(d_box_out || sm_fmt != 0, "mnc_box_mask_pool: no output");
  MNC_REQUIRE(sm_fmt == 0 || sm_ok(C, sm_fmt), "mnc_box_mask_pool: format %d needs C %% %d == 0 (C = %d)", sm_fmt,
              (sm_fmt == 1 || sm_fmt == 3) ? 64 : 32, C);
  if (R == 0) return MNC_OK;
  MNC_REQUIRE((long)R * PH * PW * (C / 4) < (1L << 31), "mnc_box_mask_pool: tensor exceeds the kernel's 32-bit index range");
  const int OH = PH / 2, OW = PW / 2;
  LaunchScope ls(ctx, "box_mask_pool", 0.0, (d_feat_sm && feat_sm_fmt != 2 ? 2.0 : 4.0) * R * (double)C * PH * PW + 4.0 * R * (double)C * (d_box_out ? 2.0 : 0.0) * OH * OW +
                                                ((sm_fmt == 1 || sm_fmt == 3) ? 4.0 : sm_fmt == 2 ? 8.0 : 0.0) * R * (double)C * OH * OW);
  if (d_feat_sm) {
    MNC_REQUIRE(((long)PH * PW * C) % 64 == 0, "mnc_box_mask_pool: the stage-major input needs PH x PW x C %% 64 == 0");
#define MNC_BMPH(SM, IN)                                                                                                        \
  hipLaunchKernelGGL((box_mask_pool_h_kernel<SM, IN>), dim3(grid_for((long)R * OH * OW * (C / 8))), dim3(256), 0, ctx->stream,   \
                     (const uint4*)d_feat_sm, d_mask, d_box_out, d_mask_out, R, PH, PW, C / 8, d_box_sm, d_mask_sm,              \
                     ctx->conv.maskpool_binary, ctx->conv.maskpool_thresh)
    if (feat_sm_fmt == 3 || sm_fmt == 3) {           // plain bf16: both sides in that form (the mode has no other)
      MNC_REQUIRE(feat_sm_fmt == 3 && sm_fmt == 3, "mnc_box_mask_pool: the bf16 stage-major form pairs with itself only");
      MNC_BMPH(3, 3);
    }
    else if (feat_sm_fmt == 1) { if (sm_fmt == 1) MNC_BMPH(1, 1); else MNC_BMPH(2, 1); }
    else { if (sm_fmt == 1) MNC_BMPH(1, 2); else MNC_BMPH(2, 2); }
#undef MNC_BMPH
    return ls.finish("box_mask_pool_h_kernel");
  }
  if (sm_fmt == 1)
    hipLaunchKernelGGL((box_mask_pool_kernel<1, 2>), dim3(grid_for((long)R * OH * OW * (C / 8))), dim3(256), 0, ctx->stream, d_feat,
                       d_mask, d_box_out, d_mask_out, R, PH, PW, C / 8, d_box_sm, d_mask_sm, ctx->conv.maskpool_binary, ctx->conv.maskpool_thresh);
  else if (sm_fmt == 2)
    hipLaunchKernelGGL((box_mask_pool_kernel<2, 2>), dim3(grid_for((long)R * OH * OW * (C / 8))), dim3(256), 0, ctx->stream, d_feat,
                       d_mask, d_box_out, d_mask_out, R, PH, PW, C / 8, d_box_sm, d_mask_sm, ctx->conv.maskpool_binary, ctx->conv.maskpool_thresh);
  else if (sm_fmt == 3)
    hipLaunchKernelGGL((box_mask_pool_kernel<3, 2>), dim3(grid_for((long)R * OH * OW * (C / 8))), dim3(256), 0, ctx->stream, d_feat,
                       d_mask, d_box_out, d_mask_out, R, PH, PW, C / 8, d_box_sm, d_mask_sm, ctx->conv.maskpool_binary, ctx->conv.maskpool_thresh);
  else
    hipLaunchKernelGGL((box_mask_pool_kernel<0, 1>), dim3(grid_for((long)R * OH * OW * (C / 4))), dim3(256), 0, ctx->stream, d_feat,
                       d_mask, d_box_out, d_mask_out, R, PH, PW, C / 4, d_box_sm, d_mask_sm, ctx->conv.maskpool_binary, ctx->conv.maskpool_thresh);
  return ls.finish("box_mask_pool_kernel");
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
