// VGG-16 trunk + RPN head kernels for gfx950 (models/VGG16/mnc_5stage/test.prototxt:19-462).
//
// Feature maps live in the "c8" layout  float [C/8][H][W][8]  (batch is 1, proposal_layer.py:65):
//   * a halo row of one 8-channel block is (TW+2)*32 contiguous bytes -> coalesced staging loads;
//   * the MFMA 32x32 accumulator of a wave holds, per lane, 4 consecutive output channels of ONE pixel in regs
//     4g..4g+3, so the epilogue is one 16-byte store per register group and a wave writes 32 pixels x 32 B
//     contiguous per 8-channel block.
//
// conv3x3 (3x3, pad 1, stride 1, + bias, + ReLU) is an implicit GEMM on the fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: exact fp32, 64 cycles, 157 TFLOP/s peak):
//   M = output channels (MFMA A operand = weights), N = pixels (B operand), K = 9 taps x Cin.
//   A workgroup (4 waves) owns a 4-row x 32-column pixel tile and 32*CO_T output channels; wave w owns pixel row w.
//   K is walked in 8-channel blocks.  Per block the (4+2)x(32+2) input halo (all 9 taps reuse it from LDS) and the
//   32*CO_T x 72 weight panel are staged global -> registers -> LDS, double-buffered with one barrier per block:
//   the loads for block c+1 are issued before the 36*CO_T MFMAs of block c and written to LDS after them.
//   Each lane fetches its 4 k-values of a fragment with ONE ds_read_b128; pixel pitch 12 floats and weight row pitch
//   76 floats make those reads bank-conflict-free (12*i mod 64 is a distinct multiple of 4 for 16 consecutive i).
#include <cstdlib>

#include "mnc_internal.h"
#include "x3_split.h"

namespace mnc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileCols = 32;
constexpr int kHaloCols = kTileCols + 2;
constexpr int kPixPitch = 12;            // floats per halo pixel in LDS (8 data + 4 pad)
constexpr int kWPitch = 76;              // floats per weight row in LDS and in the packed global layout (72 + 4 pad)

// ROWS = pixel rows (= waves) per workgroup, CO_T = 32-channel tiles per wave.
//
// Staging is branch-free and two blocks deep: every thread always loads (addresses clamped into the image / the panel,
// out-of-image pixels masked to zero, surplus threads repeat the last item) and there are two register sets -- the loads for
// block c+2 are issued while block c is multiplied and block c+1 (requested a whole iteration earlier) is written to LDS.
// Conditional loads make hipcc wait with s_waitcnt vmcnt(0) at every join, and with a single register set the LDS writes
// form a serial phase after the MFMAs; this way every wait is for data that has long landed and the scheduler is free to
// spread the LDS writes among the MFMAs.
// ABL != 0: ablation builds for tuning (MNC_CONV_ABL, 4-row tiles only): 1 = no global loads / LDS stores in the loop,
// 2 = additionally no barrier, 3 = additionally no LDS fragment reads (MFMAs on constant registers).
// ksplit > 1 (small maps): blockIdx.z = split * (Cout / NCO) + channel tile; split s walks channel blocks
// [s * Cin/8/ksplit, (s+1) * Cin/8/ksplit) and writes its raw accumulators to part[s] (c8 planes); conv_splitk_reduce_kernel
// sums the splits in order and applies bias + ReLU.  conv5_x / rpn_conv at 600x1000 are 320 workgroups on 256 CUs: a quarter
// of the SIMDs carry two waves and the rest one, so the layer takes two waves' worth of MFMAs (pure-MFMA ablation: 145 us
// against 80 us balanced); four K splits make it 1280 equal workgroups, five per CU.
template <int CO_T, int ROWS = 4, int ABL = 0>
__global__ __launch_bounds__(64 * ROWS) void conv3x3_c8_kernel(const float* __restrict__ in, const float* __restrict__ wpk,
                                                         const float* __restrict__ bias, float* __restrict__ out, int H,
                                                         int W, int Cin, int Cout, int relu, int ksplit,
                                                         float* __restrict__ part) {
  constexpr int NT = 64 * ROWS;
  constexpr int kHaloRows = ROWS + 2;
  constexpr int kHaloFloats = kHaloRows * kHaloCols * kPixPitch;
  constexpr int kHaloVec = kHaloRows * kHaloCols * 2;        // float4 items per halo
  constexpr int kHPerThread = (kHaloVec + NT - 1) / NT;
  constexpr int NCO = 32 * CO_T;
  constexpr int kWVec = NCO * (kWPitch / 4);                 // float4 items per weight panel (incl. pad)
  constexpr int kWPerThread = (kWVec + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float s_halo[2][kHaloFloats];
  __shared__ __attribute__((aligned(16))) float s_w[2][NCO * kWPitch];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, kk = lane >> 5;
  const int ncot = Cout / NCO;
  const int split = blockIdx.z / ncot;
  const int w0 = blockIdx.x * kTileCols, h0 = blockIdx.y * ROWS, co0 = (blockIdx.z - split * ncot) * NCO;
  const int nchunks = (Cin >> 3) / ksplit;                   // channel blocks of this split (the launcher makes it divide)
  const int chunk0 = split * nchunks;

  // ---- staging assignment (fixed per thread) ----
  // halo: item q -> pixel q>>1 (row-major in the (ROWS+2) x 34 halo), half q&1
  int h_off[kHPerThread];
  long h_src[kHPerThread];
  unsigned h_keep[kHPerThread];
#pragma unroll
  for (int u = 0; u < kHPerThread; ++u) {
    const int q = min(tid + u * NT, kHaloVec - 1);
    const int pix = q >> 1, half = q & 1;
    const int r = pix / kHaloCols, c = pix - r * kHaloCols;
    const int gh = h0 - 1 + r, gw = w0 - 1 + c;
    h_off[u] = pix * kPixPitch + half * 4;
    h_src[u] = ((long)min(max(gh, 0), H - 1) * W + min(max(gw, 0), W - 1)) * 8 + half * 4;
    h_keep[u] = (gh >= 0 && gh < H && gw >= 0 && gw < W) ? 0xFFFFFFFFu : 0u;
  }
  int w_idx[kWPerThread];
#pragma unroll
  for (int u = 0; u < kWPerThread; ++u) w_idx[u] = min(tid + u * NT, kWVec - 1);
  const long plane = (long)H * W * 8;

  // NB: staging registers must be initialised, otherwise hipcc keeps the arrays as allocas -> scratch
  struct Regs {
    float4 h[kHPerThread];
    float4 w[kWPerThread];
  };
  Regs R0, R1;
#pragma unroll
  for (int u = 0; u < kHPerThread; ++u) R0.h[u] = R1.h[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < kWPerThread; ++u) R0.w[u] = R1.w[u] = make_float4(0.f, 0.f, 0.f, 0.f);

  auto load_chunk = [&](int c, Regs& G) {
    c = chunk0 + min(c, nchunks - 1);
    const float* src = in + (long)c * plane;
#pragma unroll
    for (int u = 0; u < kHPerThread; ++u) G.h[u] = *reinterpret_cast<const float4*>(src + h_src[u]);
    const float4* wsrc = reinterpret_cast<const float4*>(wpk + ((long)c * Cout + co0) * kWPitch);
#pragma unroll
    for (int u = 0; u < kWPerThread; ++u) G.w[u] = wsrc[w_idx[u]];
  };
  // live == false: the phantom block behind an odd block count -- its halo is stored as zeros, so it multiplies to nothing
  auto store_chunk = [&](int buf, const Regs& G, bool live) {
#pragma unroll
    for (int u = 0; u < kHPerThread; ++u) {
      const unsigned keep = live ? h_keep[u] : 0u;
      float4 v = G.h[u];
      v.x = __uint_as_float(__float_as_uint(v.x) & keep);
      v.y = __uint_as_float(__float_as_uint(v.y) & keep);
      v.z = __uint_as_float(__float_as_uint(v.z) & keep);
      v.w = __uint_as_float(__float_as_uint(v.w) & keep);
      *reinterpret_cast<float4*>(&s_halo[buf][h_off[u]]) = v;
    }
    float4* wdst = reinterpret_cast<float4*>(&s_w[buf][0]);
#pragma unroll
    for (int u = 0; u < kWPerThread; ++u) wdst[w_idx[u]] = G.w[u];
  };

  f32x16 acc[CO_T];
  f32x16 acc2;                 // second accumulator of the CO_T == 1 variant (see the inner loop)
#pragma unroll
  for (int t = 0; t < CO_T; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc2[e] = 0.f;

  const int p_base = (wave * kHaloCols + j) * kPixPitch + kk * 4;   // + (kh*34 + kw)*12 per tap
  const int w_base = j * kWPitch + kk * 4;                          // + ct*32*76 + tap*8

  auto multiply = [&](int buf) {
    const float* sh = s_halo[buf];
    const float* sw = s_w[buf];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      float4 p;
      float4 a[CO_T];
      if (ABL == 3) {
        p = make_float4(1.f, 2.f, 3.f, 4.f);
        asm volatile("" : "+v"(p.x), "+v"(p.y), "+v"(p.z), "+v"(p.w));
#pragma unroll
        for (int t = 0; t < CO_T; ++t) a[t] = p;
      } else {
        p = *reinterpret_cast<const float4*>(sh + p_base + (kh * kHaloCols + kw) * kPixPitch);
#pragma unroll
        for (int t = 0; t < CO_T; ++t) a[t] = *reinterpret_cast<const float4*>(sw + w_base + t * 32 * kWPitch + tap * 8);
      }
      // k-step outermost: consecutive MFMAs go to DIFFERENT accumulators (never two dependent MFMAs back to back);
      // with a single channel tile the k-steps alternate between two accumulators that are summed in the epilogue
      if (CO_T == 1) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].x, p.x, acc[0], 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].y, p.y, acc2, 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].z, p.z, acc[0], 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0].w, p.w, acc2, 0, 0, 0);
      } else {
#pragma unroll
        for (int t = 0; t < CO_T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, p.x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < CO_T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, p.y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < CO_T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, p.z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < CO_T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, p.w, acc[t], 0, 0, 0);
      }
    }
  };
  // block c sits in LDS[buf], block c+1 in `cur`, block c+2 is requested into `nxt`
  auto step = [&](int c, int buf, Regs& cur, Regs& nxt) {
    if (ABL == 0) load_chunk(c + 2, nxt);
    multiply(buf);
    if (ABL == 0) store_chunk(buf ^ 1, cur, c + 1 < nchunks);
    if (ABL < 2) __syncthreads();
  };

  load_chunk(0, R0);
  store_chunk(0, R0, true);
  load_chunk(1, R0);
  __syncthreads();
  for (int c = 0; c < nchunks; c += 2) {
    step(c, 0, R0, R1);
    step(c + 1, 1, R1, R0);        // for an odd block count the last call multiplies the zero-filled phantom block
  }

  // ---- epilogue: D[row = cout (reg&3)+8*(reg>>2)+4*kk][col = pixel j] ----
  if (CO_T == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][e] += acc2[e];
  }
  const int oh = h0 + wave, ow = w0 + j;
  if (ksplit > 1) {
    if (oh < H && ow < W) {
      float* dst = part + (long)split * Cout * H * W;
#pragma unroll
      for (int t = 0; t < CO_T; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = co0 + t * 32 + g * 8 + kk * 4;
          *reinterpret_cast<float4*>(dst + (((long)(co >> 3) * H + oh) * W + ow) * 8 + kk * 4) =
              make_float4(acc[t][4 * g + 0], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
        }
    }
    return;
  }
  if (oh < H && ow < W) {
#pragma unroll
    for (int t = 0; t < CO_T; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + t * 32 + g * 8 + kk * 4;
        const float4 b = *reinterpret_cast<const float4*>(bias + co);
        float4 v = make_float4(acc[t][4 * g + 0] + b.x, acc[t][4 * g + 1] + b.y, acc[t][4 * g + 2] + b.z,
                               acc[t][4 * g + 3] + b.w);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(out + (((long)(co >> 3) * H + oh) * W + ow) * 8 + kk * 4) = v;
      }
    }
  }
}

// out = act(sum_s part[s] + bias), c8 planes; one thread per float4 (4 channels of one pixel)
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                 float* __restrict__ out, long hw, int Cout, int ksplit,
                                                                 int relu) {
  const long total4 = (long)Cout * hw / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4;                                   // element index in [C/8][HW][8]
    const int co = (int)(e / (hw * 8)) * 8 + (int)(e & 7);
    float4 v = *reinterpret_cast<const float4*>(part + e);
    for (int s = 1; s < ksplit; ++s) {
      const float4 p = *reinterpret_cast<const float4*>(part + (long)s * Cout * hw + e);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    const float4 b = *reinterpret_cast<const float4*>(bias + co);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(out + e) = v;
  }
}

// ---- conv1_1: Cin = 3 (K = 27): HBM-bound (154 MB written at 600x1000), VALU.  One thread = one pixel: its 27 inputs are
// loaded once into registers and all Cout channels are produced from them, 8 at a time, with the weights read from LDS at
// wave-uniform addresses (broadcast reads).  A wave writes 64 pixels x 32 B contiguous per channel block.
// OUT: 0 = fp32 c8, 1 = packed bf16x3 (hi x8 | lo x8), 2 = packed fp16, 3 = packed bf16 -- the activation formats of conv_sw.hip
template <int OUT>
__global__ __launch_bounds__(256, 4) void conv3x3_c3_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ bias, void* __restrict__ out, int H,
                                                         int W, int Cout, int relu) {
  extern __shared__ __attribute__((aligned(16))) float s_wt[];   // [Cout/8][27][8] then bias[Cout]
  for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) {
    const int co = i / 27, k = i - co * 27;                       // OIHW: w[co][ci][kh][kw], k = ci*9 + kh*3 + kw
    s_wt[((co >> 3) * 27 + k) * 8 + (co & 7)] = w[i];
  }
  for (int i = threadIdx.x; i < Cout; i += blockDim.x) s_wt[27 * Cout + i] = bias[i];
  __syncthreads();
  const int nblk = Cout >> 3;
  const long hw = (long)H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < hw; pix += (long)gridDim.x * blockDim.x) {
    const int h = (int)(pix / W), x = (int)(pix - (long)h * W);
    float v[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ih = h + kh - 1, iw = x + kw - 1;
          // clamped address + select: no load under a branch
          const float t = in[((long)ci * H + min(max(ih, 0), H - 1)) * W + min(max(iw, 0), W - 1)];
          v[ci * 9 + kh * 3 + kw] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? t : 0.f;
        }
#pragma unroll 1
    for (int cb = 0; cb < nblk; ++cb) {
      const float4* wr = reinterpret_cast<const float4*>(s_wt + (long)cb * 27 * 8);
      const float4 b0 = *reinterpret_cast<const float4*>(s_wt + 27 * Cout + cb * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(s_wt + 27 * Cout + cb * 8 + 4);
      float acc[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int k = 0; k < 27; ++k) {                             // same k order as before: ci, kh, kw
        const float4 w0 = wr[2 * k], w1 = wr[2 * k + 1];
        acc[0] = fmaf(v[k], w0.x, acc[0]); acc[1] = fmaf(v[k], w0.y, acc[1]);
        acc[2] = fmaf(v[k], w0.z, acc[2]); acc[3] = fmaf(v[k], w0.w, acc[3]);
        acc[4] = fmaf(v[k], w1.x, acc[4]); acc[5] = fmaf(v[k], w1.y, acc[5]);
        acc[6] = fmaf(v[k], w1.z, acc[6]); acc[7] = fmaf(v[k], w1.w, acc[7]);
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.f);
      }
      x3_store8<(OUT == 2 ? 1 : OUT == 3 ? 2 : 0), OUT != 0>(out, (long)cb * hw + pix, make_float4(acc[0], acc[1], acc[2], acc[3]),
                                    make_float4(acc[4], acc[5], acc[6], acc[7]));
    }
  }
}

// conv1_1 on the fp32 matrix pipe (round 5; the kernel above is LDS bound: 54 broadcast weight reads per pixel and channel block,
// 60 us = 2.6 TB/s on a layer whose floor is writing 153.6 MB).  GEMM view: M = Cout (<= 64: four 16-row tiles), N = 16 pixels
// per wave step, K = 27 taps padded to 28 = seven v_mfma_f32_16x16x4_f32 steps.  The weights (A: lane (m = l % 16, g = l / 16)
// holds w[16 t + m][4 ks + g]) sit in 28 registers for the whole kernel; the B operand is im2col on the fly -- lane (n, g) loads
// tap 4 ks + g of pixel n straight from the NCHW input (16 consecutive pixels per lane group: 64-byte runs, L2 resident), seven
// 4-byte loads per step, the next step's requested before this step's 28 MFMAs.  D: lane (n, g) holds channels 16 t + 4 g .. + 3
// of pixel n = one half of a c8 pixel (block 2 t + g / 2, half g & 1): every store instruction writes two 512-byte runs.
// Accumulation order: the k-ordered fma chain of the instruction from zero, bias added last (the VALU kernel starts from the
// bias): the two differ in the last bit; a layer uses ONE of them for every output format (the launcher decides by Cout).
// (NT = Cout / 16 is a template parameter: with a run-time tile count hipcc branched around every MFMA and shuttled the
// accumulators through AGPRs -- 124 us, twice the VALU kernel)
template <int OUT, int NT>
__global__ __launch_bounds__(256) void conv3x3_c3_mfma_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ bias, void* __restrict__ out, int H, int W,
                                                              int relu, int ntiles) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  const long hw = (long)H * W;
  float a[NT][7];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
      const int k = 4 * ks + g;
      const float wv = w[(long)(16 * t + n) * 27 + min(k, 26)];
      a[t][ks] = k < 27 ? wv : 0.f;
    }
  f32x4 bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const f32x4*>(bias + 16 * t + 4 * g);
  // The lane's seven taps as lane constants: (dh, dw) and the element offset ci * H * W + dh * W + dw from the pixel; tap 27
  // (g = 3, ks = 6) is the zero pad: a dh no row satisfies.  The first build of this kernel rebuilt clamped 64-bit addresses and a
  // five-comparison mask per tap and divided by W per tile: ~350 VALU instructions beside 28 MFMAs (which the fp32 MFMAs do not
  // overlap with) -- 62 us, exactly the VALU kernel's time.  Now: two unsigned range checks per tap, the loads through a buffer
  // descriptor (an out-of-range offset answers zero: no select on the value, no 64-bit arithmetic), and (h, x) carried from tile
  // to tile.
  int toff[7], tdh[7], tdw[7];
#pragma unroll
  for (int ks = 0; ks < 7; ++ks) {
    const int k = 4 * ks + g, kk = min(k, 26), ci = kk / 9, r = kk - ci * 9;
    tdh[ks] = k < 27 ? r / 3 - 1 : (1 << 29);
    tdw[ks] = r - (r / 3) * 3 - 1;
    toff[ks] = ci * H * W + (r / 3 - 1) * W + tdw[ks];
  }
  const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 3 * H * W * 4, 0x00020000);
  const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int step = nwaves * 16, step_h = step / W, step_x = step - step_h * W;      // (wave-uniform: scalar division, once)
  int pix_l = wave0 * 16 + n, h_l = pix_l / W, x_l = pix_l - h_l * W;               // the tile being LOADED (one division per lane, once)
  auto load_taps = [&](float (&v)[7]) {
    const int hh = pix_l < (int)hw ? h_l : (1 << 28);              // (a tile past the map: no row passes the check)
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
      // (& not &&: a select, not a branch around the load -- hipcc waits for vmcnt(0) where branches around loads join)
      const bool ok = ((unsigned)(hh + tdh[ks]) < (unsigned)H) & ((unsigned)(x_l + tdw[ks]) < (unsigned)W);
      v[ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irs, ok ? (pix_l + toff[ks]) * 4 : 0x7FFFFFF0, 0, 0));
    }
    pix_l += step; h_l += step_h; x_l += step_x;
    if (x_l >= W) { x_l -= W; h_l += 1; }
  };
  // output: fp32 c8 / split-bf16 pixels are 32 bytes per 8-channel block (the lane's 4 channels: 16 bytes at half g & 1; split form:
  // 8 bytes of hi at g & 1, 8 of lo 16 bytes behind), fp16 pixels 16 bytes (8 at g & 1); channels 16 t + 4 g .. : block 2 t + g / 2
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  constexpr int kPB = OUT >= 2 ? 16 : 32;
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out, 0, 2 * NT * (int)hw * kPB, 0x00020000);
  const int lane_plane = (g >> 1) * (int)hw * kPB + (g & 1) * (OUT == 0 ? 16 : 8);
  float cur[7], nxt[7];
  load_taps(cur);
  for (int tile = wave0; tile < ntiles; tile += nwaves) {
    load_taps(nxt);                                                 // (past the last tile: every lane out of range, zeros)
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 7; ++ks)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][ks], cur[ks], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]));   // (architectural registers: no AGPR form of the MFMAs)
    // stores through a buffer descriptor too: the lane's part of the offset is one multiply-add per tile, the channel tile's plane
    // pair sits in the scalar offset, pixels past the map get an out-of-range offset (dropped) -- x3_store4's bytes, without its
    // 64-bit address arithmetic per store
    const int pix = tile * 16 + n;
    const int voff = pix < (int)hw ? pix * kPB + lane_plane : 0x7FFFFFF0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float4 o = make_float4(acc[t][0] + bv[t][0], acc[t][1] + bv[t][1], acc[t][2] + bv[t][2], acc[t][3] + bv[t][3]);
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      const int soff = t * 2 * (int)hw * kPB;
      if (OUT == 0) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mnc_u32x4, o), ors, voff, soff, 0);
      } else if (OUT == 2) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x3_f16x4(o)), ors, voff, soff, 0);
      } else if (OUT == 3) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x3_bf16x4(o)), ors, voff, soff, 0);
      } else {
        uint2 hi, lo;
        x3_split4(o, hi, lo);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), ors, voff, soff, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), ors, voff + 16, soff, 0);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) cur[ks] = nxt[ks];
  }
}

// ---- Pooling MAX 2x2/2, Caffe ceil mode, c8 ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_c8_kernel(const float* __restrict__ in, float* __restrict__ out, int CB,
                                                          int H, int W, int OH, int OW) {
  const long total = (long)CB * OH * OW * 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int half = (int)(idx & 1);
    long p = idx >> 1;
    const int ow = (int)(p % OW);
    p /= OW;
    const int oh = (int)(p % OH);
    const int cb = (int)(p / OH);
    const int h0 = oh * 2, x0 = ow * 2;
    const float* base = in + (long)cb * H * W * 8 + half * 4;
    float4 m = *reinterpret_cast<const float4*>(base + ((long)h0 * W + x0) * 8);
    auto upd = [&](int hh, int ww) {
      if (hh < H && ww < W) {
        const float4 v = *reinterpret_cast<const float4*>(base + ((long)hh * W + ww) * 8);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    };
    upd(h0, x0 + 1); upd(h0 + 1, x0); upd(h0 + 1, x0 + 1);
    *reinterpret_cast<float4*>(out + (((long)cb * OH + oh) * OW + ow) * 8 + half * 4) = m;
  }
}

// ---- 1x1 conv c8 -> NCHW (rpn_cls_score, rpn_bbox_pred): 0.13 GFLOP, VALU ------------------------------------------
__global__ __launch_bounds__(256) void conv1x1_to_nchw_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int HW, int Cin, int Cout) {
  const long total = (long)HW * Cout;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int p = (int)(idx % HW), o = (int)(idx / HW);
    const float* wr = w + (long)o * Cin;
    float acc = bias[o];
    for (int cb = 0; cb < (Cin >> 3); ++cb) {
      const float4 a0 = *reinterpret_cast<const float4*>(in + ((long)cb * HW + p) * 8);
      const float4 a1 = *reinterpret_cast<const float4*>(in + ((long)cb * HW + p) * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(wr + cb * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(wr + cb * 8 + 4);
      acc = fmaf(a0.x, b0.x, acc); acc = fmaf(a0.y, b0.y, acc); acc = fmaf(a0.z, b0.z, acc); acc = fmaf(a0.w, b0.w, acc);
      acc = fmaf(a1.x, b1.x, acc); acc = fmaf(a1.y, b1.y, acc); acc = fmaf(a1.z, b1.z, acc); acc = fmaf(a1.w, b1.w, acc);
    }
    out[idx] = acc;
  }
}

// ---- the 1x1 heads on the fp32 matrix pipe (round 5) ----------------------------------------------------------------
// rpn_cls_score / rpn_bbox_pred (test.prototxt:413-439): 54 x 512 weights on 38 x 63 pixels -- 0.13 GFLOP that the one-thread-per-
// output kernel above spends 22 us on (a 512-long dependent fma chain per thread).  Here a wave owns 16 pixels x two 16-row tiles
// of output channels on v_mfma_f32_16x16x4_f32, no LDS, no barrier: per 16 input channels (two c8 blocks) a lane loads ONE 16-byte
// piece of the input (lane (n = l % 16, g = l / 16): pixel n, channels 16 j + 4 g .. + 3 -- the c8 layout is already the B
// operand's order) and one piece of each tile's weight row (row l % 16, the same channels), then MFMA q multiplies element q:
// k-index g <-> channel 16 j + 4 g + q.  The next chunk's pieces are requested before the current chunk's MFMAs.
// Every output is the same k-ordered fma chain whatever tile row it sits in, so the plain form (rows in order) and the fused
// form below produce the same bits.
// SOFTMAX: the (bg, fg) softmax of the RPN (test.prototxt:440-462; rpn_softmax_kernel's arithmetic) in the epilogue.  Rows are
// arranged so that a lane holds both scores of an anchor: tile 0 = [bg 0 .. A-1 | the first 16 - A bbox rows], tile 1 = [fg 0 ..
// A-1 | the next 16 - A bbox rows], tiles 2.. = the remaining bbox rows: bg a and fg a are row a of tiles 0 and 1 = the same lane
// and register.
__device__ __forceinline__ int rpn_row(int tile, int r, int A, int Cout, bool softmax) {
  if (!softmax) { const int o = tile * 16 + r; return o < Cout ? o : -1; }
  const int spare = 16 - A;                               // bbox rows that ride along in tiles 0 and 1
  if (tile < 2) {
    if (r < A) return tile * A + r;
    const int o = 2 * A + tile * spare + (r - A);
    return o < Cout ? o : -1;
  }
  const int o = 2 * A + 2 * spare + (tile - 2) * 16 + r;
  return o < Cout ? o : -1;
}

template <bool SOFTMAX>
__global__ __launch_bounds__(256) void conv1x1_mfma_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           float* __restrict__ prob, int HW, int Cin, int Cout, int A, int pairs) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ptile = wid / pairs, pair = wid - ptile * pairs;
  if (ptile * 16 >= HW) return;
  const int n = lane & 15, g = lane >> 4;
  const int p = min(ptile * 16 + n, HW - 1);                      // (clamped: the last tile's dead pixels repeat a live one)
  int row[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) row[t] = rpn_row(2 * pair + t, n, A, Cout, SOFTMAX);
  const float* wp[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) wp[t] = w + (long)max(row[t], 0) * Cin + 4 * g;
  const float* ip = in + ((long)(g >> 1) * HW + p) * 8 + 4 * (g & 1);
  const long istep = 2L * HW * 8;                                 // two c8 blocks per chunk
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int chunks = Cin >> 4;
  // four chunks per trip, the next trip's twelve pieces requested before this trip's 32 MFMAs: only ~300 waves exist (38 x 63
  // pixels), so a wave's own loads in flight are all the latency hiding there is (one chunk ahead: 20 us, the VALU kernel's 22)
  constexpr int kD = 4;
  f32x4 b[kD], a0[kD], a1[kD];
  auto fetch = [&](int j0, f32x4 (&bb)[kD], f32x4 (&x0)[kD], f32x4 (&x1)[kD]) {
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      const int j = min(j0 + d, chunks - 1);                      // (clamped: a trip past the end repeats the last chunk, unused)
      bb[d] = *reinterpret_cast<const f32x4*>(ip + j * istep);
      x0[d] = *reinterpret_cast<const f32x4*>(wp[0] + j * 16);
      x1[d] = *reinterpret_cast<const f32x4*>(wp[1] + j * 16);
    }
  };
  fetch(0, b, a0, a1);
  for (int j0 = 0; j0 < chunks; j0 += kD) {
    f32x4 bn[kD], a0n[kD], a1n[kD];
    fetch(j0 + kD, bn, a0n, a1n);
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      if (j0 + d < chunks) {                                      // (wave-uniform; chunks % 4 != 0 only for reduced widths)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[d][q], b[d][q], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[d][q], b[d][q], acc[1], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int d = 0; d < kD; ++d) { b[d] = bn[d]; a0[d] = a0n[d]; a1[d] = a1n[d]; }
  }
  // D: lane (pixel n, g) holds rows 4 g + e of each tile
  const bool live = ptile * 16 + n < HW;
  float v[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int o = rpn_row(2 * pair + t, 4 * g + e, A, Cout, SOFTMAX);
      v[t][e] = acc[t][e] + bias[max(o, 0)];
      if (live && o >= 0) out[(long)o * HW + p] = v[t][e];
    }
  if (SOFTMAX && pair == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int a = 4 * g + e;
      if (live && a < A) {
        const float s0 = v[0][e], s1 = v[1][e];
        const float m = fmaxf(s0, s1);
        const float e0 = expf(s0 - m), e1 = expf(s1 - m);
        const float sum = e0 + e1;
        prob[(long)a * HW + p] = e0 / sum;
        prob[(long)(A + a) * HW + p] = e1 / sum;
      }
    }
  }
}

// Softmax over the (bg, fg) pair of every anchor: channel a against channel A+a (test.prototxt:440-462).
__global__ void rpn_softmax_kernel(const float* __restrict__ s, float* __restrict__ p, int A, int HW) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= A * HW) return;
  const float s0 = s[idx], s1 = s[idx + A * HW];
  const float m = fmaxf(s0, s1);
  const float e0 = expf(s0 - m), e1 = expf(s1 - m);
  const float sum = e0 + e1;
  p[idx] = e0 / sum;
  p[idx + A * HW] = e1 / sum;
}

// ---- layout conversion / weight packing ----------------------------------------------------------------------------
__global__ void nchw_to_c8_kernel(const float* __restrict__ in, float* __restrict__ out, int C, long HW) {
  const long total = (long)C * HW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int e = (int)(idx & 7);
    const long p = (idx >> 3) % HW;
    const int cb = (int)((idx >> 3) / HW);
    out[idx] = in[(long)(cb * 8 + e) * HW + p];
  }
}
__global__ void c8_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C, long HW) {
  const long total = (long)C * HW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long p = idx % HW;
    const int c = (int)(idx / HW);
    out[idx] = in[((long)(c >> 3) * HW + p) * 8 + (c & 7)];
  }
}
// OIHW [Cout][Cin][3][3] -> [Cin/8][Cout][76]: element (cb, co, tap*8 + e) = w[co][cb*8+e][tap]; pad = 0
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin) {
  const long total = (long)(Cin >> 3) * Cout * kWPitch;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % kWPitch);
    const long r = idx / kWPitch;
    const int co = (int)(r % Cout), cb = (int)(r / Cout);
    float v = 0.f;
    if (k < 72) {
      const int tap = k >> 3, e = k & 7;
      v = w[((long)co * Cin + cb * 8 + e) * 9 + tap];
    }
    out[idx] = v;
  }
}

static int grid_for(long total, int block = 256, int cap = 256 * 32) {
  long g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace mnc

namespace mnc {
void conv_splitk_reduce_launch(hipStream_t stream, const float* d_part, const float* d_bias, float* d_out, int H, int W,
                               int Cout, int ksplit, int relu) {
  hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(grid_for((long)Cout * H * W / 4)), dim3(256), 0, stream, d_part, d_bias,
                     d_out, (long)H * W, Cout, ksplit, relu);
}
}  // namespace mnc

using namespace mnc;

extern "C" {

int mnc_conv3x3(mnc_ctx* ctx, const float* d_in, const float* d_wpk, const float* d_bias, float* d_out, int H, int W,
                int Cin, int Cout, int relu) {
  MNC_REQUIRE(ctx && d_in && d_wpk && d_bias && d_out, "mnc_conv3x3: null pointer");
  MNC_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0,
              "mnc_conv3x3: unsupported shape H=%d W=%d Cin=%d Cout=%d (need Cin%%8==0, Cout%%32==0)", H, W, Cin, Cout);
  // (rows per workgroup, channel tiles per wave).  Measured on MI355X (tools/kernel_bench.py, round 1): every shape with
  // >= ~1000 workgroups plateaus at 95-105 TF/s regardless of the tile (4 or 8 rows, 32/64/128 channels); 5-row
  // workgroups lose 30 % (5 waves do not spread evenly over 4 SIMDs).  So: 4 rows, and the widest channel tile that still
  // leaves >= 4 workgroups per CU; narrow maps fall back to 32 channels.  MNC_CONV_ROWS=4|8 / MNC_CONV_COT=1|2|4 override.
  int best_rows = 4, best_cot = 4;
  while (best_cot > 1 && (Cout % (32 * best_cot) != 0 ||
                          (long)cdiv(W, kTileCols) * cdiv(H, best_rows) * (Cout / (32 * best_cot)) < 1024))
    best_cot >>= 1;
  if (tune_set(ctx, T_CONV_COT)) {
    const int v = tune(ctx, T_CONV_COT, 0);
    if ((v == 1 || v == 2 || v == 4) && Cout % (32 * v) == 0) best_cot = v;
  }
  if (tune_set(ctx, T_CONV_ROWS)) {
    const int v = tune(ctx, T_CONV_ROWS, 0);
    if (v == 2 || v == 4 || v == 8) best_rows = v;
  }
  const int rows = best_rows, co_t = best_cot;
  // K splits for the small maps (see the kernel): fewer than two workgroups per CU -> split the channel blocks 4 (or 2) ways
  int ksplit = 1;
  {
    const long wgs = (long)cdiv(W, kTileCols) * cdiv(H, rows) * (Cout / (32 * co_t));
    const int blocks = Cin / 8;
    if (wgs < 512) ksplit = blocks % 4 == 0 && blocks >= 16 ? 4 : (blocks % 2 == 0 && blocks >= 8 ? 2 : 1);
    if (tune_set(ctx, T_CONV_KSPLIT)) {
      const int v = tune(ctx, T_CONV_KSPLIT, 0);
      if (v >= 1 && v <= 8 && blocks % v == 0) ksplit = v;
    }
  }
  float* part = nullptr;
  if (ksplit > 1) {
    int rc = ensure_scratch(ctx, (size_t)ksplit * Cout * H * W * 4);
    if (rc) return rc;
    part = (float*)ctx->scratch;
  }
  const int tx = cdiv(W, kTileCols), ty = cdiv(H, rows);
  const double flops = 2.0 * H * W * 9.0 * Cin * Cout;
  const double bytes = 4.0 * ((double)H * W * (Cin + Cout) + 9.0 * Cin * Cout);
  LaunchScope ls(ctx, "conv3x3_c8_mfma", flops, bytes);
  dim3 grid(tx, ty, Cout / (32 * co_t) * ksplit);
#ifdef MNC_TUNING
  if (tune_set(ctx, T_CONV_ABL)) {
    const int a = tune(ctx, T_CONV_ABL, 0);
#define MNC_ABL_CASE(T, A) if (rows == 4 && co_t == T && a == A) { hipLaunchKernelGGL((conv3x3_c8_kernel<T, 4, A>), grid, dim3(256), 0, ctx->stream, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part); goto launched; }
    MNC_ABL_CASE(1, 1) MNC_ABL_CASE(1, 2) MNC_ABL_CASE(1, 3) MNC_ABL_CASE(2, 1) MNC_ABL_CASE(2, 2) MNC_ABL_CASE(2, 3)
#undef MNC_ABL_CASE
  }
#endif
#define MNC_CONV_CASE(R, T) if (rows == R && co_t == T) hipLaunchKernelGGL((conv3x3_c8_kernel<T, R>), grid, dim3(64 * R), 0, ctx->stream, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part);
  MNC_CONV_CASE(2, 1) MNC_CONV_CASE(2, 2) MNC_CONV_CASE(2, 4)
  MNC_CONV_CASE(4, 1) MNC_CONV_CASE(4, 2) MNC_CONV_CASE(4, 4)
  MNC_CONV_CASE(8, 1) MNC_CONV_CASE(8, 2) MNC_CONV_CASE(8, 4)
#undef MNC_CONV_CASE
#ifdef MNC_TUNING
launched:
#endif
  if (ksplit > 1) conv_splitk_reduce_launch(ctx->stream, part, d_bias, d_out, H, W, Cout, ksplit, relu);   // same profiling scope
  return ls.finish("conv3x3_c8_kernel");
}

int mnc_conv3x3_c3_fmt(mnc_ctx* ctx, const float* d_in, const float* d_w, const float* d_bias, void* d_out, int H, int W,
                       int Cout, int relu, int out_fmt) {
  MNC_REQUIRE(ctx && d_in && d_w && d_bias && d_out, "mnc_conv3x3_c3: null pointer");
  MNC_REQUIRE(H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0 && Cout <= 512, "mnc_conv3x3_c3: unsupported shape");
  MNC_REQUIRE(out_fmt >= 0 && out_fmt <= 3, "mnc_conv3x3_c3: out_fmt must be 0 (fp32), 1 (bf16x3 packed), 2 (fp16 packed) or 3 (bf16 packed)");
  const double flops = 2.0 * H * W * 27.0 * Cout, bytes = 4.0 * H * W * (3.0 + (out_fmt >= 2 ? 0.5 : 1.0) * Cout);
  LaunchScope ls(ctx, "conv3x3_c3", flops, bytes);
  // Cout a multiple of 16 up to 64 (VGG-16: 64): the matrix-pipe kernel; other widths: the VALU kernel.  CONV_COT=-1 forces the latter.
  if (Cout % 16 == 0 && Cout <= 64 && (long)H * W * Cout * 4 < 0x7FFFFFF0L && tune(ctx, T_CONV_COT, 0) != -1) {
    const int ntiles = cdiv((long)H * W, 16);
    typedef void (*c3_fn)(const float*, const float*, const float*, void*, int, int, int, int);
#define MNC_C3(F) {conv3x3_c3_mfma_kernel<F, 1>, conv3x3_c3_mfma_kernel<F, 2>, conv3x3_c3_mfma_kernel<F, 3>, conv3x3_c3_mfma_kernel<F, 4>}
    static const c3_fn table[4][4] = {MNC_C3(0), MNC_C3(1), MNC_C3(2), MNC_C3(3)};
#undef MNC_C3
    const c3_fn kern = table[out_fmt][Cout / 16 - 1];
    // three blocks per CU, each wave walking ~12 tiles: a wave's 28 weight gathers are paid once (grid cap 512 / 768 / 1024 / 2048 /
    // 4096: 38.6 / 38.0 / 39.7 / 42.0 / 52.8 us at 600x1000, profiles/r05_conv1_1.txt; CONV_ROWS > 8 overrides the cap)
    const int cap = tune(ctx, T_CONV_ROWS, 0) > 8 ? tune(ctx, T_CONV_ROWS, 0) : 768;
    const int blocks = ntiles / 4 < cap ? (ntiles + 3) / 4 : cap;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, ctx->stream, d_in, d_w, d_bias, d_out, H, W, relu, ntiles);
    return ls.finish("conv3x3_c3_mfma_kernel");
  }
  auto kern = out_fmt == 0 ? conv3x3_c3_kernel<0> : out_fmt == 1 ? conv3x3_c3_kernel<1> : out_fmt == 2 ? conv3x3_c3_kernel<2> : conv3x3_c3_kernel<3>;
  hipLaunchKernelGGL(kern, dim3(grid_for((long)H * W)), dim3(256), (size_t)(28 * Cout) * 4, ctx->stream, d_in, d_w, d_bias, d_out,
                     H, W, Cout, relu);
  return ls.finish("conv3x3_c3_kernel");
}

int mnc_conv3x3_c3(mnc_ctx* ctx, const float* d_in, const float* d_w, const float* d_bias, float* d_out, int H, int W,
                   int Cout, int relu) {
  return mnc_conv3x3_c3_fmt(ctx, d_in, d_w, d_bias, d_out, H, W, Cout, relu, 0);
}

int mnc_maxpool2_c8(mnc_ctx* ctx, const float* d_in, float* d_out, int C, int H, int W) {
  MNC_REQUIRE(ctx && d_in && d_out && C > 0 && C % 8 == 0 && H >= 2 && W >= 2, "mnc_maxpool2_c8: bad argument");
  const int OH = (H - 2 + 1) / 2 + 1, OW = (W - 2 + 1) / 2 + 1;
  LaunchScope ls(ctx, "maxpool2_c8", 0.0, 4.0 * C * ((double)H * W + (double)OH * OW));
  const long total = (long)(C / 8) * OH * OW * 2;
  hipLaunchKernelGGL(maxpool2_c8_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, d_in, d_out, C / 8, H, W, OH, OW);
  return ls.finish("maxpool2_c8_kernel");
}

int mnc_conv1x1_to_nchw(mnc_ctx* ctx, const float* d_in, const float* d_w, const float* d_bias, float* d_out, int H,
                        int W, int Cin, int Cout) {
  MNC_REQUIRE(ctx && d_in && d_w && d_bias && d_out && H > 0 && W > 0 && Cin % 8 == 0 && Cout > 0,
              "mnc_conv1x1_to_nchw: bad argument");
  LaunchScope ls(ctx, "conv1x1_to_nchw", 2.0 * H * W * Cin * Cout, 4.0 * H * W * (Cin + Cout));
  if (Cin % 16 == 0 && (reinterpret_cast<uintptr_t>(d_w) & 15) == 0) {        // the matrix-pipe form (16 channels per step)
    const int pairs = cdiv(cdiv(Cout, 16), 2);
    const long waves = (long)cdiv((long)H * W, 16) * pairs;
    hipLaunchKernelGGL(conv1x1_mfma_kernel<false>, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, ctx->stream, d_in, d_w, d_bias,
                       d_out, (float*)nullptr, H * W, Cin, Cout, 0, pairs);
    return ls.finish("conv1x1_mfma_kernel");
  }
  const long total = (long)H * W * Cout;
  hipLaunchKernelGGL(conv1x1_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, d_in, d_w, d_bias, d_out,
                     H * W, Cin, Cout);
  return ls.finish("conv1x1_to_nchw_kernel");
}

// rpn_cls_score + rpn_bbox_pred as ONE 1x1 convolution over the concatenated weights ([2A cls rows | 4A bbox rows]) with the
// RPN's 2-way softmax in its epilogue: d_score = the 6A score planes (NCHW), d_prob = the 2A probability planes -- the bits of
// mnc_conv1x1_to_nchw followed by mnc_rpn_softmax(A), one launch (csrc/pipeline.hip).
int mnc_rpn_heads(mnc_ctx* ctx, const float* d_in, const float* d_w, const float* d_bias, float* d_score, float* d_prob, int H,
                  int W, int Cin, int A) {
  MNC_REQUIRE(ctx && d_in && d_w && d_bias && d_score && d_prob && H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && A > 0,
              "mnc_rpn_heads: bad argument");
  const int Cout = 6 * A;
  if (Cin % 16 != 0 || A > 16 || (reinterpret_cast<uintptr_t>(d_w) & 15) != 0) {
    int rc = mnc_conv1x1_to_nchw(ctx, d_in, d_w, d_bias, d_score, H, W, Cin, Cout);
    if (rc) return rc;
    return mnc_rpn_softmax(ctx, d_score, d_prob, A, H, W);
  }
  LaunchScope ls(ctx, "rpn_heads", 2.0 * H * W * Cin * Cout, 4.0 * H * W * (Cin + Cout + 2 * A));
  const int rest = 4 * A - 2 * (16 - A);                          // bbox rows behind tiles 0 and 1
  const int tiles = 2 + (rest > 0 ? cdiv(rest, 16) : 0), pairs = cdiv(tiles, 2);
  const long waves = (long)cdiv((long)H * W, 16) * pairs;
  hipLaunchKernelGGL(conv1x1_mfma_kernel<true>, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, ctx->stream, d_in, d_w, d_bias,
                     d_score, d_prob, H * W, Cin, Cout, A, pairs);
  return ls.finish("conv1x1_mfma_kernel");
}

int mnc_rpn_softmax(mnc_ctx* ctx, const float* d_score, float* d_prob, int A, int H, int W) {
  MNC_REQUIRE(ctx && d_score && d_prob && A > 0 && H > 0 && W > 0, "mnc_rpn_softmax: bad argument");
  LaunchScope ls(ctx, "rpn_softmax");
  hipLaunchKernelGGL(rpn_softmax_kernel, dim3(cdiv((long)A * H * W, 256)), dim3(256), 0, ctx->stream, d_score, d_prob, A,
                     H * W);
  return ls.finish("rpn_softmax_kernel");
}

int mnc_nchw_to_c8(mnc_ctx* ctx, const float* d_nchw, float* d_c8, int C, int H, int W) {
  MNC_REQUIRE(ctx && d_nchw && d_c8 && C > 0 && C % 8 == 0 && H > 0 && W > 0, "mnc_nchw_to_c8: bad argument");
  LaunchScope ls(ctx, "nchw_to_c8");
  hipLaunchKernelGGL(nchw_to_c8_kernel, dim3(grid_for((long)C * H * W)), dim3(256), 0, ctx->stream, d_nchw, d_c8, C,
                     (long)H * W);
  return ls.finish("nchw_to_c8_kernel");
}

int mnc_c8_to_nchw(mnc_ctx* ctx, const float* d_c8, float* d_nchw, int C, int H, int W) {
  MNC_REQUIRE(ctx && d_nchw && d_c8 && C > 0 && C % 8 == 0 && H > 0 && W > 0, "mnc_c8_to_nchw: bad argument");
  LaunchScope ls(ctx, "c8_to_nchw");
  hipLaunchKernelGGL(c8_to_nchw_kernel, dim3(grid_for((long)C * H * W)), dim3(256), 0, ctx->stream, d_c8, d_nchw, C,
                     (long)H * W);
  return ls.finish("c8_to_nchw_kernel");
}

int mnc_pack_conv3x3_weights(mnc_ctx* ctx, const float* d_oihw, float* d_packed, int Cout, int Cin) {
  MNC_REQUIRE(ctx && d_oihw && d_packed && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0,
              "mnc_pack_conv3x3_weights: bad argument");
  LaunchScope ls(ctx, "pack_conv3x3");
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(grid_for((long)(Cin / 8) * Cout * kWPitch)), dim3(256), 0, ctx->stream,
                     d_oihw, d_packed, Cout, Cin);
  return ls.finish("pack_conv3x3_kernel");
}

}  // extern "C"
