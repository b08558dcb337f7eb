// General convolution / pooling / residual kernels for graphs beyond VGG-16 (a ResNet-50 trunk has
// 1x1, strided and 7x7 convolutions, 3x3/2 max pooling and residual adds; BASELINE.json configs[4]).  First correct path:
// the stride-1 3x3 layers keep using the tuned kernels of conv.hip / conv_sw.hip; everything else comes through here.
//
//   conv2d_c8_kernel   any KHxKW / stride / pad, c8 -> c8, fp32 MFMA implicit GEMM (v_mfma_f32_32x32x2_f32):
//                      M = output channels (A = weights), N = output pixels (B = gathered input pixels), K = taps x Cin walked
//                      one (tap, pair of 8-channel blocks) at a time.  WG = 4 waves = 64 channels x 128 pixels, wave = 32 x 64.
//                      There is no halo to reuse for 1x1 / strided taps, so each step stages exactly the 128 x 16 input values
//                      and 64 x 16 weights it multiplies -- one tap, two 8-channel blocks -- global -> registers -> LDS,
//                      double-buffered, one barrier per step, branch-free clamped + masked loads.  Each lane feeds its 4 of
//                      the 8 channels of a block with one ds_read_b128 (pitch 20 floats: conflict-free).  Epilogue: + bias
//                      (+ residual) (+ ReLU), 16-byte stores.
//   conv_stem_c3_kernel  KxK / stride / pad on the 3-channel NCHW input blob -> c8 (ResNet conv1 7x7/2): VALU, one output pixel
//                      per thread, 16 output channels at a time with the weights broadcast from LDS.  HBM/L2-bound.
//   maxpool_c8_kernel  Caffe MAX pooling with any kernel / stride / pad (ceil output size, windows clipped to the image).
//   add_kernel         out = a + b (+ ReLU): Eltwise SUM of two same-layout tensors.
#include <cfloat>
#include <cstdlib>

#include "mnc_internal.h"
#include "x3_split.h"

namespace mnc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGenCo = 64;      // output channels per workgroup
constexpr int kGenPx = 128;     // output pixels per workgroup
constexpr int kGenPitch = 20;   // floats per LDS row (2 x 8 data + 4 pad; 20 / 4 odd: conflict-free 16-byte fragment reads)

// [Cout][Cin][KH][KW] -> [KH*KW][Cin/8][Cout][8]
__global__ void pack_conv_gen_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KK) {
  const long total = (long)KK * Cin * Cout;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % 8);
    long r = idx / 8;
    const int co = (int)(r % Cout);
    r /= Cout;
    const int cb = (int)(r % (Cin / 8));
    const int t = (int)(r / (Cin / 8));
    out[idx] = w[((long)co * Cin + cb * 8 + c) * KK + t];
  }
}

template <int CT>
__global__ __launch_bounds__(256) void conv2d_c8_kernel(const float* __restrict__ in, const float* __restrict__ wpk,
                                                        const float* __restrict__ bias, const float* __restrict__ res,
                                                        float* __restrict__ out, int H, int W, int Cin, int Cout, int KH,
                                                        int KW, int stride, int pad, int OH, int OW, int relu) {
  __shared__ __attribute__((aligned(16))) float s_act[2][kGenPx * kGenPitch];
  constexpr int kCo = kGenCo * CT;              // output channels per workgroup: 64 (CT = 1) or 128 (CT = 2)
  __shared__ __attribute__((aligned(16))) float s_wt[2][kCo * kGenPitch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long P = (long)OH * OW;
  const long p0 = (long)blockIdx.x * kGenPx;
  const int co0 = blockIdx.y * kCo;
  const int CB = Cin >> 3;
  const int CP = (CB + 1) >> 1;                 // pairs of 8-channel blocks: one step multiplies 16 channels of one tap
  const int steps = KH * KW * CP;

  // staging roles: activations -- pixel tid/2, channel half tid%2 of BOTH blocks of the pair; weights -- channel tid/4,
  // float4 tid%4 of the 16 values (quarters 0,1 = first block, 2,3 = second block)
  const int a_px = tid >> 1, a_half = tid & 1;
  long ap = p0 + a_px;
  const bool a_live = ap < P;
  if (!a_live) ap = P - 1;
  const int a_oy = (int)(ap / OW), a_ox = (int)(ap % OW);
  const int a_iy0 = a_oy * stride - pad, a_ix0 = a_ox * stride - pad;
  const int w_co = tid >> 2, w_q = tid & 3;     // + 64 for the second weight item when CT = 2
  bool w_live[CT];
  int w_row[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    w_live[c] = co0 + w_co + 64 * c < Cout;
    w_row[c] = w_live[c] ? co0 + w_co + 64 * c : Cout - 1;
  }

  // (initialised: hipcc keeps registers that a lambda writes first as allocas -- 80 bytes of scratch at CT = 2)
  float4 ra0 = make_float4(0.f, 0.f, 0.f, 0.f), ra1 = ra0, rw[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) rw[c] = ra0;
  auto load = [&](int s) {
    const int t = s / CP, cp = s - t * CP;
    const int cb0 = cp * 2, cb1 = min(cb0 + 1, CB - 1);
    const bool second = cb0 + 1 < CB;                      // an odd block count leaves the last pair half empty
    const int ky = t / KW, kx = t - ky * KW;
    const int iy = a_iy0 + ky, ix = a_ix0 + kx;
    const bool ok = a_live && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
    const long pix = ((long)cy * W + cx) * 8 + a_half * 4;
    const float4 v0 = *reinterpret_cast<const float4*>(in + (long)cb0 * H * W * 8 + pix);
    const float4 v1 = *reinterpret_cast<const float4*>(in + (long)cb1 * H * W * 8 + pix);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    ra0 = ok ? v0 : zero;
    ra1 = (ok && second) ? v1 : zero;
    const int wb = (w_q >> 1) ? cb1 : cb0;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const float4 u = *reinterpret_cast<const float4*>(wpk + (((long)t * CB + wb) * Cout + w_row[c]) * 8 + (w_q & 1) * 4);
      rw[c] = (w_live[c] && (second || !(w_q >> 1))) ? u : zero;
    }
  };
  auto store = [&](int buf) {
    *reinterpret_cast<float4*>(&s_act[buf][a_px * kGenPitch + a_half * 4]) = ra0;
    *reinterpret_cast<float4*>(&s_act[buf][a_px * kGenPitch + 8 + a_half * 4]) = ra1;
#pragma unroll
    for (int c = 0; c < CT; ++c) *reinterpret_cast<float4*>(&s_wt[buf][(w_co + 64 * c) * kGenPitch + w_q * 4]) = rw[c];
  };

  // wave tile: CT x 32 channels by 64 pixels -> acc[c][j]
  f32x16 acc[CT][2];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][j][i] = 0.f;

  const int co_half = wave & 1, px_half = wave >> 1;
  const int frag_off = (lane >> 5) * 4;
  const float* a_ptr0 = &s_wt[0][(co_half * 32 * CT + (lane & 31)) * kGenPitch + frag_off];
  const float* b_ptr0 = &s_act[0][(px_half * 64 + (lane & 31)) * kGenPitch + frag_off];

  load(0);
  store(0);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    load(s + 1 < steps ? s + 1 : s);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                           // the two 8-channel blocks of the pair
      float4 a[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c)
        a[c] = *reinterpret_cast<const float4*>(a_ptr0 + buf * (kCo * kGenPitch) + c * 32 * kGenPitch + h * 8);
      const float4 b0 = *reinterpret_cast<const float4*>(b_ptr0 + buf * (kGenPx * kGenPitch) + h * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(b_ptr0 + buf * (kGenPx * kGenPitch) + 32 * kGenPitch + h * 8);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        acc[c][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].x, b0.x, acc[c][0], 0, 0, 0);
        acc[c][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].x, b1.x, acc[c][1], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        acc[c][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].y, b0.y, acc[c][0], 0, 0, 0);
        acc[c][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].y, b1.y, acc[c][1], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        acc[c][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].z, b0.z, acc[c][0], 0, 0, 0);
        acc[c][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].z, b1.z, acc[c][1], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        acc[c][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].w, b0.w, acc[c][0], 0, 0, 0);
        acc[c][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].w, b1.w, acc[c][1], 0, 0, 0);
      }
    }
    store(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds, per 32x32 block, 4 consecutive channels (regs 4g..4g+3) of one pixel for g = 0..3
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long px = p0 + px_half * 64 + j * 32 + (lane & 31);
      if (px >= P) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + (co_half * CT + c) * 32 + g * 8 + frag_off;
        if (co >= Cout) continue;
        const float4 bv = *reinterpret_cast<const float4*>(bias + co);
        float4 v = make_float4(acc[c][j][g * 4 + 0] + bv.x, acc[c][j][g * 4 + 1] + bv.y, acc[c][j][g * 4 + 2] + bv.z,
                               acc[c][j][g * 4 + 3] + bv.w);
        const long o = ((long)(co >> 3) * P + px) * 8 + (co & 7);
        if (res) {
          const float4 r = *reinterpret_cast<const float4*>(res + o);
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(out + o) = v;
      }
    }
}

// ---- "f16" math mode of the general convolution: one v_mfma_f32_32x32x16_f16 per 16 channels instead of eight fp32 MFMAs.
// Weights [KH*KW][ceil(Cin/32)][Cout][32 halves] (mnc_pack_conv_weights_f16, zero-padded channel groups); activations fp32 c8 in
// HBM, rounded to fp16 while they are staged.  Same 64-channel x 128-pixel workgroup tile; a step covers one tap and 32
// channels (four c8 blocks, two MFMA K-steps): LDS rows are 64 B + 16 B pad (pitch 20 dwords, conflict-free 16-byte reads).
typedef _Float16 gen_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gen_f16x4 __attribute__((ext_vector_type(4)));

__global__ void pack_conv_gen_f16_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int Cin,
                                         int KK) {
  const int G = (Cin + 31) / 32;
  const long total = (long)KK * G * Cout * 32;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % 32);
    long r = idx / 32;
    const int co = (int)(r % Cout);
    r /= Cout;
    const int g = (int)(r % G);
    const int t = (int)(r / G);
    const int ci = g * 32 + c;
    const _Float16 h = ci < Cin ? (_Float16)w[((long)co * Cin + ci) * KK + t] : (_Float16)0.f;
    out[idx] = __builtin_bit_cast(unsigned short, h);
  }
}

__global__ __launch_bounds__(256) void conv2d_c8_f16_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            float* __restrict__ out, int H, int W, int Cin, int Cout, int KH,
                                                            int KW, int stride, int pad, int OH, int OW, int relu) {
  constexpr int kPitch = 20;                    // dwords per LDS row: 32 halves + 4 pad
  __shared__ __attribute__((aligned(16))) unsigned s_act[2][kGenPx * kPitch];
  __shared__ __attribute__((aligned(16))) unsigned s_wt[2][kGenCo * kPitch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long P = (long)OH * OW;
  const long p0 = (long)blockIdx.x * kGenPx;
  const int co0 = blockIdx.y * kGenCo;
  const int CB = Cin >> 3;
  const int G = (CB + 3) >> 2;                  // groups of four 8-channel blocks
  const int steps = KH * KW * G;

  // staging roles: activations -- pixel tid/2, channel half tid%2 of the four blocks; weights -- channel tid/4, 16 bytes tid%4
  const int a_px = tid >> 1, a_half = tid & 1;
  long ap = p0 + a_px;
  const bool a_live = ap < P;
  if (!a_live) ap = P - 1;
  const int a_oy = (int)(ap / OW), a_ox = (int)(ap % OW);
  const int a_iy0 = a_oy * stride - pad, a_ix0 = a_ox * stride - pad;
  const int w_co = tid >> 2, w_q = tid & 3;
  const bool w_live = co0 + w_co < Cout;
  const int w_row = w_live ? co0 + w_co : Cout - 1;

  float4 ra[4];
  uint4 rw;
  auto load = [&](int s) {
    const int t = s / G, g = s - t * G;
    const int ky = t / KW, kx = t - ky * KW;
    const int iy = a_iy0 + ky, ix = a_ix0 + kx;
    const bool ok = a_live && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
    const long pix = ((long)cy * W + cx) * 8 + a_half * 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int cb = g * 4 + b;
      const float4 v = *reinterpret_cast<const float4*>(in + (long)min(cb, CB - 1) * H * W * 8 + pix);
      ra[b] = (ok && cb < CB) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const uint4 u = wpk[(((long)t * G + g) * Cout + w_row) * 4 + w_q];
    rw = w_live ? u : make_uint4(0, 0, 0, 0);
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const gen_f16x4 h = {(_Float16)ra[b].x, (_Float16)ra[b].y, (_Float16)ra[b].z, (_Float16)ra[b].w};
      *reinterpret_cast<uint2*>(&s_act[buf][a_px * kPitch + b * 4 + a_half * 2]) = __builtin_bit_cast(uint2, h);
    }
    *reinterpret_cast<uint4*>(&s_wt[buf][w_co * kPitch + w_q * 4]) = rw;
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  const int co_half = wave & 1, px_half = wave >> 1;
  const int kb = lane >> 5;                      // this lane's 8 of the 16 channels of an MFMA K-step = one c8 block
  const unsigned* a_ptr0 = &s_wt[0][(co_half * 32 + (lane & 31)) * kPitch + kb * 4];
  const unsigned* b_ptr0 = &s_act[0][(px_half * 64 + (lane & 31)) * kPitch + kb * 4];

  load(0);
  store(0);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    load(s + 1 < steps ? s + 1 : s);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                 // two MFMA K-steps of 16 channels
      const gen_f16x8 a = __builtin_bit_cast(gen_f16x8, *reinterpret_cast<const uint4*>(a_ptr0 + buf * (kGenCo * kPitch) + h * 8));
      const gen_f16x8 b0 = __builtin_bit_cast(gen_f16x8, *reinterpret_cast<const uint4*>(b_ptr0 + buf * (kGenPx * kPitch) + h * 8));
      const gen_f16x8 b1 =
          __builtin_bit_cast(gen_f16x8, *reinterpret_cast<const uint4*>(b_ptr0 + buf * (kGenPx * kPitch) + 32 * kPitch + h * 8));
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, acc[1], 0, 0, 0);
    }
    store(buf ^ 1);
    __syncthreads();
  }

  const int frag_off = kb * 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long px = p0 + px_half * 64 + j * 32 + (lane & 31);
    if (px >= P) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co0 + co_half * 32 + g * 8 + frag_off;
      if (co >= Cout) continue;
      const float4 bv = *reinterpret_cast<const float4*>(bias + co);
      float4 v = make_float4(acc[j][g * 4 + 0] + bv.x, acc[j][g * 4 + 1] + bv.y, acc[j][g * 4 + 2] + bv.z,
                             acc[j][g * 4 + 3] + bv.w);
      const long o = ((long)(co >> 3) * P + px) * 8 + (co & 7);
      if (res) {
        const float4 r = *reinterpret_cast<const float4*>(res + o);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4*>(out + o) = v;
    }
  }
}

// Stem: Cin = 3, NCHW input, weights [Cout][3][K][K] re-laid in LDS as [Cout/16][3*K*K][16].
__global__ __launch_bounds__(256) void conv_stem_c3_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ bias, void* __restrict__ out, int H,
                                                           int W, int Cout, int K, int stride, int pad, int OH, int OW,
                                                           int relu, int out_pk) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  const int taps = 3 * K * K, groups = Cout / 16;
  for (int i = threadIdx.x; i < Cout * taps; i += blockDim.x) {
    const int co = i / taps, t = i - co * taps;
    s_w[((co >> 4) * taps + t) * 16 + (co & 15)] = w[i];
  }
  __syncthreads();
  const long P = (long)OH * OW;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int oy = (int)(p / OW), ox = (int)(p % OW);
  const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
  for (int g = 0; g < groups; ++g) {
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = bias[g * 16 + i];
    const float* wg = s_w + (long)g * taps * 16;
    int t = 0;
    for (int c = 0; c < 3; ++c)
      for (int ky = 0; ky < K; ++ky) {
        const int iy = iy0 + ky;
        const bool rowok = iy >= 0 && iy < H;
        const float* row = in + ((long)c * H + min(max(iy, 0), H - 1)) * W;
        for (int kx = 0; kx < K; ++kx, ++t) {
          const int ix = ix0 + kx;
          const float v = (rowok && ix >= 0 && ix < W) ? row[ix] : 0.f;
          const float4* wp = reinterpret_cast<const float4*>(wg + t * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 ww = wp[q];
            acc[q * 4 + 0] = fmaf(v, ww.x, acc[q * 4 + 0]);
            acc[q * 4 + 1] = fmaf(v, ww.y, acc[q * 4 + 1]);
            acc[q * 4 + 2] = fmaf(v, ww.z, acc[q * 4 + 2]);
            acc[q * 4 + 3] = fmaf(v, ww.w, acc[q * 4 + 3]);
          }
        }
      }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 v0 = make_float4(acc[h * 8 + 0], acc[h * 8 + 1], acc[h * 8 + 2], acc[h * 8 + 3]);
      float4 v1 = make_float4(acc[h * 8 + 4], acc[h * 8 + 5], acc[h * 8 + 6], acc[h * 8 + 7]);
      if (relu) {
        v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
        v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
      }
      const long pix = (long)(g * 2 + h) * P + p;
      if (out_pk) x3_store8<1, true>(out, pix, v0, v1);         // packed fp16 c8 (nearest even): the f16 mode's activation tensor
      else x3_store8<1, false>(out, pix, v0, v1);
    }
  }
}

// MAX pooling on the packed fp16 c8 tensor: one thread per (channel block, output pixel), 16-byte loads and stores
__global__ __launch_bounds__(256) void maxpool_c8_f16_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int CB, int H,
                                                             int W, int K, int stride, int pad, int OH, int OW) {
  const long total = (long)CB * OH * OW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int cb = (int)(r / OH);
    const int y0 = max(oy * stride - pad, 0), y1 = min(oy * stride - pad + K, H);
    const int x0 = max(ox * stride - pad, 0), x1 = min(ox * stride - pad + K, W);
    f16x8 m = {(_Float16)-65504.f, (_Float16)-65504.f, (_Float16)-65504.f, (_Float16)-65504.f,
               (_Float16)-65504.f, (_Float16)-65504.f, (_Float16)-65504.f, (_Float16)-65504.f};
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        const f16x8 v = __builtin_bit_cast(f16x8, in[((long)cb * H + y) * W + x]);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
    out[idx] = __builtin_bit_cast(uint4, m);
  }
}

__global__ __launch_bounds__(256) void maxpool_c8_kernel(const float* __restrict__ in, float* __restrict__ out, int CB, int H,
                                                         int W, int K, int stride, int pad, int OH, int OW) {
  const long total = (long)CB * OH * OW * 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int half = (int)(idx & 1);
    long r = idx >> 1;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int cb = (int)(r / OH);
    const int y0 = max(oy * stride - pad, 0), y1 = min(oy * stride - pad + K, H);
    const int x0 = max(ox * stride - pad, 0), x1 = min(ox * stride - pad + K, W);
    float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        const float4 v = *reinterpret_cast<const float4*>(in + (((long)cb * H + y) * W + x) * 8 + half * 4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    *reinterpret_cast<float4*>(out + (((long)cb * OH + oy) * OW + ox) * 8 + half * 4) = m;
  }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n4, long n, int relu) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    float4 v = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    reinterpret_cast<float4*>(out)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float v = a[i] + b[i];
    out[i] = relu ? fmaxf(v, 0.f) : v;
  }
}

static int grid1d(long total, int cap = 256 * 64) {
  long g = (total + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace mnc

using namespace mnc;

// Caffe's output sizes: convolution floor((n + 2p - k)/s) + 1; pooling ceil((n + 2p - k)/s) + 1, minus one when the last
// window would start in the padding (pooling_layer.cpp).
static inline int conv_out(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }
static inline int pool_out(int n, int k, int s, int p) {
  int o = (n + 2 * p - k + s - 1) / s + 1;
  if (p > 0 && (o - 1) * s >= n + p) --o;
  return o;
}

int mnc_pack_conv_weights(mnc_ctx* ctx, const float* d_oihw, float* d_packed, int Cout, int Cin, int KH, int KW) {
  MNC_REQUIRE(ctx && d_oihw && d_packed, "mnc_pack_conv_weights: null pointer");
  MNC_REQUIRE(Cout > 0 && Cout % 8 == 0 && Cin > 0 && Cin % 8 == 0 && KH > 0 && KW > 0, "mnc_pack_conv_weights: bad shape");
  LaunchScope ls(ctx, "pack_conv_gen");
  hipLaunchKernelGGL(pack_conv_gen_kernel, dim3(grid1d((long)KH * KW * Cin * Cout)), dim3(256), 0, ctx->stream, d_oihw, d_packed,
                     Cout, Cin, KH * KW);
  return ls.finish("pack_conv_gen_kernel");
}

int mnc_conv2d(mnc_ctx* ctx, const float* d_in, const float* d_w, const float* d_bias, const float* d_residual, float* d_out,
               int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu) {
  MNC_REQUIRE(ctx && d_in && d_w && d_bias && d_out, "mnc_conv2d: null pointer");
  MNC_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 8 == 0 && KH > 0 && KW > 0 && stride > 0 &&
                  pad >= 0 && H + 2 * pad >= KH && W + 2 * pad >= KW,
              "mnc_conv2d: bad shape (Cin, Cout multiples of 8)");
  const int OH = conv_out(H, KH, stride, pad), OW = conv_out(W, KW, stride, pad);
  const long P = (long)OH * OW;
  const double flops = 2.0 * P * Cout * (double)Cin * KH * KW;
  LaunchScope ls(ctx, "conv2d_c8_mfma", flops, 4.0 * ((double)H * W * Cin + (double)P * Cout * (d_residual ? 2 : 1)));
  // MNC_CONV2D_WIDE=1 (tuning knob): 128-channel workgroup tiles (wave = 64 channels x 64 pixels: 8 B/clk/CU of staging
  // instead of 12, half the barriers per MFMA).  Measured on the ResNet-50 C4 trunk at 800x1333: 40 vs 44 TFLOP/s for the
  // 64-channel tile (fewer, fatter workgroups), so the narrow tile is the default.
  const bool wide = Cout >= 128 && tune(ctx, T_CONV2D_WIDE, 0) != 0;
  if (wide)
    hipLaunchKernelGGL(conv2d_c8_kernel<2>, dim3((unsigned)cdiv(P, kGenPx), (unsigned)cdiv(Cout, 128)), dim3(256), 0, ctx->stream,
                       d_in, d_w, d_bias, d_residual, d_out, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, relu);
  else
    hipLaunchKernelGGL(conv2d_c8_kernel<1>, dim3((unsigned)cdiv(P, kGenPx), (unsigned)cdiv(Cout, kGenCo)), dim3(256), 0,
                       ctx->stream, d_in, d_w, d_bias, d_residual, d_out, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, relu);
  return ls.finish("conv2d_c8_kernel");
}

int mnc_pack_conv_weights_f16(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin, int KH, int KW) {
  MNC_REQUIRE(ctx && d_oihw && d_packed, "mnc_pack_conv_weights_f16: null pointer");
  MNC_REQUIRE(Cout > 0 && Cout % 8 == 0 && Cin > 0 && Cin % 8 == 0 && KH > 0 && KW > 0, "mnc_pack_conv_weights_f16: bad shape");
  LaunchScope ls(ctx, "pack_conv_gen_f16");
  hipLaunchKernelGGL(pack_conv_gen_f16_kernel, dim3(grid1d((long)KH * KW * ((Cin + 31) / 32) * 32 * Cout)), dim3(256), 0,
                     ctx->stream, d_oihw, (unsigned short*)d_packed, Cout, Cin, KH * KW);
  return ls.finish("pack_conv_gen_f16_kernel");
}

int mnc_conv2d_f16(mnc_ctx* ctx, const float* d_in, const void* d_w, const float* d_bias, const float* d_residual, float* d_out,
                   int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu) {
  MNC_REQUIRE(ctx && d_in && d_w && d_bias && d_out, "mnc_conv2d_f16: null pointer");
  MNC_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 8 == 0 && KH > 0 && KW > 0 && stride > 0 &&
                  pad >= 0 && H + 2 * pad >= KH && W + 2 * pad >= KW,
              "mnc_conv2d_f16: bad shape (Cin, Cout multiples of 8)");
  const int OH = conv_out(H, KH, stride, pad), OW = conv_out(W, KW, stride, pad);
  const long P = (long)OH * OW;
  const double flops = 2.0 * P * Cout * (double)Cin * KH * KW;
  LaunchScope ls(ctx, "conv2d_c8_f16", flops, 4.0 * ((double)H * W * Cin + (double)P * Cout * (d_residual ? 2 : 1)));
  hipLaunchKernelGGL(conv2d_c8_f16_kernel, dim3((unsigned)cdiv(P, kGenPx), (unsigned)cdiv(Cout, kGenCo)), dim3(256), 0, ctx->stream,
                     d_in, (const uint4*)d_w, d_bias, d_residual, d_out, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, relu);
  return ls.finish("conv2d_c8_f16_kernel");
}

int mnc_conv_stem_c3_fmt(mnc_ctx* ctx, const float* d_in_nchw, const float* d_w_oihw, const float* d_bias, void* d_out, int H,
                         int W, int Cout, int K, int stride, int pad, int relu, int out_packed) {
  MNC_REQUIRE(ctx && d_in_nchw && d_w_oihw && d_bias && d_out, "mnc_conv_stem_c3: null pointer");
  MNC_REQUIRE(H > 0 && W > 0 && Cout > 0 && Cout % 16 == 0 && K > 0 && stride > 0 && pad >= 0 && H + 2 * pad >= K &&
                  W + 2 * pad >= K && (size_t)Cout * 3 * K * K * 4 <= 64 * 1024,
              "mnc_conv_stem_c3: bad shape (Cout multiple of 16, weights <= 64 KB)");
  const int OH = conv_out(H, K, stride, pad), OW = conv_out(W, K, stride, pad);
  const long P = (long)OH * OW;
  LaunchScope ls(ctx, "conv_stem_c3", 2.0 * P * Cout * 3.0 * K * K, 4.0 * 3.0 * H * W + (out_packed ? 2.0 : 4.0) * (double)P * Cout);
  hipLaunchKernelGGL(conv_stem_c3_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), (size_t)Cout * 3 * K * K * 4, ctx->stream,
                     d_in_nchw, d_w_oihw, d_bias, d_out, H, W, Cout, K, stride, pad, OH, OW, relu, out_packed ? 1 : 0);
  return ls.finish("conv_stem_c3_kernel");
}

int mnc_conv_stem_c3(mnc_ctx* ctx, const float* d_in_nchw, const float* d_w_oihw, const float* d_bias, float* d_out_c8, int H,
                     int W, int Cout, int K, int stride, int pad, int relu) {
  return mnc_conv_stem_c3_fmt(ctx, d_in_nchw, d_w_oihw, d_bias, d_out_c8, H, W, Cout, K, stride, pad, relu, 0);
}

int mnc_maxpool_c8_f16(mnc_ctx* ctx, const void* d_in, void* d_out, int C, int H, int W, int K, int stride, int pad) {
  MNC_REQUIRE(ctx && d_in && d_out && C > 0 && C % 8 == 0 && H > 0 && W > 0 && K > 0 && stride > 0 && pad >= 0 && pad < K,
              "mnc_maxpool_c8_f16: bad argument");
  const int OH = pool_out(H, K, stride, pad), OW = pool_out(W, K, stride, pad);
  LaunchScope ls(ctx, "maxpool_c8_f16", 0.0, 2.0 * C * ((double)H * W + (double)OH * OW));
  hipLaunchKernelGGL(maxpool_c8_f16_kernel, dim3(grid1d((long)(C / 8) * OH * OW)), dim3(256), 0, ctx->stream, (const uint4*)d_in,
                     (uint4*)d_out, C / 8, H, W, K, stride, pad, OH, OW);
  return ls.finish("maxpool_c8_f16_kernel");
}

int mnc_maxpool_c8(mnc_ctx* ctx, const float* d_in, float* d_out, int C, int H, int W, int K, int stride, int pad) {
  MNC_REQUIRE(ctx && d_in && d_out && C > 0 && C % 8 == 0 && H > 0 && W > 0 && K > 0 && stride > 0 && pad >= 0 && pad < K,
              "mnc_maxpool_c8: bad argument");
  const int OH = pool_out(H, K, stride, pad), OW = pool_out(W, K, stride, pad);
  LaunchScope ls(ctx, "maxpool_c8", 0.0, 4.0 * C * ((double)H * W + (double)OH * OW));
  hipLaunchKernelGGL(maxpool_c8_kernel, dim3(grid1d((long)(C / 8) * OH * OW * 2)), dim3(256), 0, ctx->stream, d_in, d_out, C / 8,
                     H, W, K, stride, pad, OH, OW);
  return ls.finish("maxpool_c8_kernel");
}

int mnc_add(mnc_ctx* ctx, const float* d_a, const float* d_b, float* d_out, size_t n, int relu) {
  MNC_REQUIRE(ctx && d_a && d_b && d_out, "mnc_add: null pointer");
  if (n == 0) return MNC_OK;
  LaunchScope ls(ctx, "add", 0.0, 12.0 * (double)n);
  hipLaunchKernelGGL(add_kernel, dim3(grid1d((long)(n / 4) + 1)), dim3(256), 0, ctx->stream, d_a, d_b, d_out, (long)(n / 4),
                     (long)n, relu);
  return ls.finish("add_kernel");
}

// ---- stem convolution on the fp16 matrix pipe ("f16" math mode; ResNet conv1 7x7/2 pad 3 on the 3-channel input blob) ---------
// The VALU stem above runs at 19 TFLOP/s (0.26 ms at 800x1333: one LDS weight read per 4 FMAs, the input re-read per 16-channel
// group).  As a GEMM: M = output channels, N = pixels, K = (input channel, kernel row) x 8 kernel columns -- the kernel row is
// padded from K to 8 taps with zero weights, so that a lane's 8 K-values are 8 CONSECUTIVE input pixels of one row: the B
// fragment of pixel (oy, ox), K-step ks, lane half kb is in[c][oy * stride - pad + ky][ox * stride - pad .. + 7] with
// (c, ky) = divmod(2 ks + kb, K), straight from the NCHW fp32 blob (rounded to fp16 in registers), no im2col buffer, no LDS.
// The weights (3 K rows x 8 taps = 168 values for 7x7: 11 K-steps) are packed once in A-fragment order and live in REGISTERS
// for the whole kernel (CT x KS x 4 VGPRs); a wave walks 32-pixel row segments with a grid stride.  Row segments whose windows
// lie inside the image take unconditional loads; border segments select zeros per element.
template <int CT, int KS>
__global__ __launch_bounds__(256) void conv_stem_f16_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk,
                                                            const float* __restrict__ bias, void* __restrict__ out, int H, int W,
                                                            int stride, int pad, int OH, int OW, int relu, int out_pk) {
  constexpr int K = (2 * KS) / 3;                         // KS = ceil(3 K / 2): 11 -> 7, 8 -> 5, 5 -> 3
  struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };     // a dword-aligned 16-byte global load
  const int lane = threadIdx.x & 63, j = lane & 31, kb = lane >> 5;
  const int co_group = blockIdx.y;                        // CT x 32 output channels
  uint4 a[KS][CT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int c = 0; c < CT; ++c) a[ks][c] = wpk[((long)ks * gridDim.y * CT + co_group * CT + c) * 64 + lane];
  const int segs = (OW + 31) >> 5;
  const unsigned ntiles = (unsigned)OH * segs;
  const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
  const long P = (long)OH * OW;
  for (unsigned t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; t < ntiles; t += nwaves) {
    const unsigned tile = __builtin_amdgcn_readfirstlane(t);
    const int oy = (int)(tile / (unsigned)segs), ox0 = (int)(tile % (unsigned)segs) * 32;
    const int ox = ox0 + j;
    const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
    // the segment's windows: rows iy0 .. iy0 + K - 1, columns (ox0 * stride - pad) .. ((ox0 + 31) * stride - pad + 7)
    const bool inside = iy0 >= 0 && iy0 + K <= H && ox0 * stride - pad >= 0 && (ox0 + 31) * stride - pad + 8 <= W;
    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int q = 2 * ks + kb;                          // (input channel, kernel row); q >= 3 K: the zero-weight padding of K
      const int c = min(q / K, 2), ky = q - (q / K) * K;
      const int iy = iy0 + ky;
      float v[8];
      if (inside) {
        const F4* p = reinterpret_cast<const F4*>(in + ((long)c * H + iy) * W + ix0);
        const F4 lo = p[0], hi = p[1];
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
      } else {
        const bool rowok = iy >= 0 && iy < H && q < 3 * K;
        const float* p = in + ((long)c * H + min(max(iy, 0), H - 1)) * W;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ix = ix0 + e;
          const float x = p[min(max(ix, 0), W - 1)];
          v[e] = (rowok && ix >= 0 && ix < W) ? x : 0.f;
        }
      }
      const gen_f16x8 b = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3],
                           (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
#pragma unroll
      for (int cc = 0; cc < CT; ++cc)
        acc[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gen_f16x8, a[ks][cc]), b, acc[cc], 0, 0, 0);
    }
    if (ox < OW) {
      const long p = (long)oy * OW + ox;
#pragma unroll
      for (int cc = 0; cc < CT; ++cc)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = (co_group * CT + cc) * 32 + g * 8 + kb * 4;
          const float4 bv = *reinterpret_cast<const float4*>(bias + co);
          float4 o = make_float4(acc[cc][g * 4 + 0] + bv.x, acc[cc][g * 4 + 1] + bv.y, acc[cc][g * 4 + 2] + bv.z,
                                 acc[cc][g * 4 + 3] + bv.w);
          if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
          const long idx = ((long)(co >> 3) * P + p) * 2 + kb;
          if (out_pk) reinterpret_cast<uint2*>(out)[idx] = x3_f16x4(o);
          else reinterpret_cast<float4*>(out)[idx] = o;
        }
    }
  }
}

// [Cout][3][K][K] fp32 -> A fragments [KS][Cout/32][64 lanes] x 8 halves: lane (i, kb) of (ks, ct) holds, for (c, ky) =
// divmod(2 ks + kb, K), W[ct*32 + i][c][ky][0..7] (taps >= K and rows >= 3 K are zero).
__global__ void pack_conv_stem_f16_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cout, int K, int KS) {
  const int CoT = Cout >> 5;
  const int total = KS * CoT * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, t = idx >> 6;
    const int ct = t % CoT, ks = t / CoT;
    const int co = ct * 32 + (lane & 31), q = 2 * ks + (lane >> 5);
    gen_f16x8 h = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    if (q < 3 * K) {
      const float* p = w + ((long)co * 3 * K + q) * K;          // [co][c][ky][.] with q = c * K + ky
      for (int e = 0; e < K; ++e) h[e] = (_Float16)p[e];
    }
    out[idx] = __builtin_bit_cast(uint4, h);
  }
}

extern "C" {

int mnc_pack_conv_stem_f16(mnc_ctx* ctx, const float* d_w_oihw, void* d_packed, int Cout, int K) {
  MNC_REQUIRE(ctx && d_w_oihw && d_packed && Cout > 0 && Cout % 32 == 0 && K > 0 && K <= 8,
              "mnc_pack_conv_stem_f16: bad argument (Cout%%32==0, K <= 8)");
  const int KS = (3 * K + 1) / 2;
  LaunchScope ls(ctx, "pack_conv_stem_f16");
  hipLaunchKernelGGL(pack_conv_stem_f16_kernel, dim3(cdiv((long)KS * (Cout / 32) * 64, 256)), dim3(256), 0, ctx->stream, d_w_oihw,
                     (uint4*)d_packed, Cout, K, KS);
  return ls.finish("pack_conv_stem_f16_kernel");
}

int mnc_conv_stem_f16(mnc_ctx* ctx, const float* d_in_nchw, const void* d_w_packed, const float* d_bias, void* d_out, int H, int W,
                      int Cout, int K, int stride, int pad, int relu, int out_packed) {
  MNC_REQUIRE(ctx && d_in_nchw && d_w_packed && d_bias && d_out, "mnc_conv_stem_f16: null pointer");
  MNC_REQUIRE(H > 0 && W > 0 && Cout > 0 && Cout % 32 == 0 && (K == 7 || K == 3 || K == 5) && stride > 0 && pad >= 0 &&
                  H + 2 * pad >= K && W + 2 * pad >= K,
              "mnc_conv_stem_f16: bad shape (Cout multiple of 32; K = 3, 5 or 7)");
  const int OH = conv_out(H, K, stride, pad), OW = conv_out(W, K, stride, pad);
  const long P = (long)OH * OW;
  const int CoT = Cout / 32;
  const int ct = CoT % 2 == 0 ? 2 : 1;
  const long tiles = (long)OH * cdiv(OW, 32);
  long gx = (tiles + 3) / 4;                               // 4 waves per workgroup, one row segment per wave and iteration
  if (gx > 1024) gx = 1024;                                // ~4 waves per SIMD; the rest by the grid-stride loop
  LaunchScope ls(ctx, "conv_stem_f16", 2.0 * P * Cout * 3.0 * K * K, 4.0 * 3.0 * H * W + (out_packed ? 2.0 : 4.0) * (double)P * Cout);
  dim3 grid((unsigned)gx, (unsigned)(CoT / ct));
#define MNC_STEM(CT, KS)                                                                                                    \
  hipLaunchKernelGGL((conv_stem_f16_kernel<CT, KS>), grid, dim3(256), 0, ctx->stream, d_in_nchw, (const uint4*)d_w_packed, d_bias, \
                     d_out, H, W, stride, pad, OH, OW, relu, out_packed ? 1 : 0)
  if (K == 7) { if (ct == 2) MNC_STEM(2, 11); else MNC_STEM(1, 11); }
  else if (K == 5) { if (ct == 2) MNC_STEM(2, 8); else MNC_STEM(1, 8); }
  else { if (ct == 2) MNC_STEM(2, 5); else MNC_STEM(1, 5); }
#undef MNC_STEM
  return ls.finish("conv_stem_f16_kernel");
}

}  // extern "C"
