// 1x1 convolution (stride 1 or 2, no padding) as a plain GEMM for gfx950 -- the bottleneck layers of a ResNet trunk
// (BASELINE.json configs[4]); two thirds of the general convolutions of ResNet-50 C4 are 1x1.
//
// A 1x1 convolution has no halo and no taps: out[co][p] = sum_ci W[co][ci] * in[ci][p] is a GEMM whose B operand (the
// activations) is, in the c8 layout, ALREADY in MFMA fragment order: lane (j = pixel, kb = half of the K-step) of
//   v_mfma_f32_32x32x16_f16  needs 8 consecutive channels of one pixel  = one 16-byte c8 pixel of the packed fp16 tensor,
//   v_mfma_f32_32x32x2_f32   needs (4 MFMAs at a time) 4 consecutive channels = half a 32-byte c8 pixel of the fp32 tensor,
// and 32 lanes read 32 consecutive pixels: 512 contiguous bytes per half wave.  So nothing is staged: both operands go
// global -> registers -> matrix pipe.  No LDS, no barrier, no format conversion in the loop.  The weights are packed once at
// load in the A operand's fragment order ([K-step][32-channel tile][lane] x 16 bytes: one contiguous kilobyte per wave load);
// they are small (<= 1 MB per layer) and stay in L2.
//
// Wave tile = CT x 32 output channels by PT x 32 pixels (chosen per call, see conv1x1_launch), four waves of a workgroup take
// four consecutive pixel ranges of the same channel group (their weight loads hit L1).  K loop: four fragment sets in flight (loads for K-step k+3 are issued while
// k is multiplied), every load unconditional with a clamped K index (a load under a branch makes hipcc drain vmcnt), phantom
// K-steps past the end are multiplied by zeroed B fragments.  Workgroups are numbered so that all channel groups of one pixel
// range run on the same XCD (the activations are fetched into that XCD's L2 once).
//
// f16 variant: activations are the packed fp16 c8 tensor ([C/8][H][W][8] halves, mnc_hip.h "2-byte activation tensors"), the
// output is packed fp16 (round to nearest even of the fp32 result) or fp32 c8, the residual either; fp32 accumulate.  This layer
// type is HBM-bound in fp16 (AI ~100 FLOP/B against a ridge of 312): what the kernel has to do is move the bytes once.
// fp32 variant: fp32 c8 in / out / residual on the fp32 matrix pipe (MFMA-bound: 12 B/clk/CU of operand loads).
#include "mnc_internal.h"
#include "x3_split.h"

namespace mnc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// [Cout][Cin] fp32 -> [KS][CoT][64 lanes] x 16 bytes; lane (i = lane % 32, kb = lane / 32) of (ks, ct) holds
//   f16 : W[ct*32 + i][16 ks + 8 kb + 0..7] as halves (nearest even)       KS = Cin / 16
//   fp32: W[ct*32 + i][ 8 ks + 4 kb + 0..3]                                 KS = Cin / 8
// rows >= Cout are zero.
template <int F16>
__global__ void pack_conv1x1_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cout, int Cin, int CoT) {
  const int KS = Cin / (F16 ? 16 : 8);
  const long total = (long)KS * CoT * 64;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const long t = idx >> 6;
    const int ct = (int)(t % CoT), ks = (int)(t / CoT);
    const int co = ct * 32 + (lane & 31), kb = lane >> 5;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (co < Cout) {
      if (F16) {
        const float* p = w + (long)co * Cin + ks * 16 + kb * 8;
        const f16x8 h = {(_Float16)p[0], (_Float16)p[1], (_Float16)p[2], (_Float16)p[3],
                         (_Float16)p[4], (_Float16)p[5], (_Float16)p[6], (_Float16)p[7]};
        v = __builtin_bit_cast(uint4, h);
      } else {
        const float* p = w + (long)co * Cin + ks * 8 + kb * 4;
        v = make_uint4(__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]), __float_as_uint(p[3]));
      }
    }
    out[idx] = v;
  }
}

template <int CT, int PT, int F16>
__global__ __launch_bounds__(256) void conv1x1_direct_kernel(const uint4* __restrict__ in, const uint4* __restrict__ wpk,
                                                             const float* __restrict__ bias, const void* __restrict__ res,
                                                             void* __restrict__ out, int H, int W, int OW, int P, int KS,
                                                             int CoT, int Cout, int stride, int relu, int res_pk, int out_pk,
                                                             int npg, int ncg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, kb = lane >> 5;
  // XCD-aware numbering (the dispatcher places block b on XCD b % 8): each XCD gets a contiguous range of the logical order
  // (channel group fastest, then pixel range), so the workgroups that read the same pixels share one L2
  const int total = npg * ncg;
  const int q = total >> 3, r = total & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int cg = logical % ncg, pg = logical / ncg;
  const int pbase = (pg * 4 + wave) * (32 * PT);
  if (pbase >= P) return;                       // no barrier anywhere in this kernel
  const long HW = (long)H * W;

  const uint4* bp[PT];
#pragma unroll
  for (int jj = 0; jj < PT; ++jj) {
    const int p = min(pbase + jj * 32 + j, P - 1);
    const int oy = p / OW, ox = p - oy * OW;
    const long ipix = (long)(oy * stride) * W + (long)ox * stride;
    bp[jj] = in + (F16 ? (long)kb * HW + ipix : ipix * 2 + kb);
  }
  const long b_step = 2 * HW;                   // uint4 per K-step: f16 two 16-byte planes, fp32 one 32-byte plane
  const uint4* ap[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) ap[c] = wpk + (long)min(cg * CT + c, CoT - 1) * 64 + lane;
  const long a_step = (long)CoT * 64;

  f32x16 acc[CT][PT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int jj = 0; jj < PT; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[c][jj][e] = 0.f;

  struct Frag { uint4 a[CT]; uint4 b[PT]; };
  auto load = [&](int ks, Frag& f) {
    const long k = min(ks, KS - 1);
#pragma unroll
    for (int c = 0; c < CT; ++c) f.a[c] = ap[c][k * a_step];
#pragma unroll
    for (int jj = 0; jj < PT; ++jj) f.b[jj] = bp[jj][k * b_step];
  };
  auto mm = [&](const Frag& f, bool live) {
    const unsigned keep = live ? 0xFFFFFFFFu : 0u;
    uint4 b[PT];
#pragma unroll
    for (int jj = 0; jj < PT; ++jj) b[jj] = make_uint4(f.b[jj].x & keep, f.b[jj].y & keep, f.b[jj].z & keep, f.b[jj].w & keep);
    if (F16) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int jj = 0; jj < PT; ++jj)
          acc[c][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x3_as_f16x8(f.a[c]), x3_as_f16x8(b[jj]), acc[c][jj], 0, 0, 0);
    } else {
#define MNC_C11_MFMA(E)                                                                                            \
  _Pragma("unroll") for (int c = 0; c < CT; ++c) _Pragma("unroll") for (int jj = 0; jj < PT; ++jj) acc[c][jj] = \
      __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(f.a[c].E), __uint_as_float(b[jj].E), acc[c][jj], 0, 0, 0);
      MNC_C11_MFMA(x)
      MNC_C11_MFMA(y)
      MNC_C11_MFMA(z)
      MNC_C11_MFMA(w)
#undef MNC_C11_MFMA
    }
  };

  // program order is pinned with sched_barrier: left alone, the machine scheduler sinks every load to just above its first use
  // (to save registers) and the loop waits with vmcnt(0) on loads it issued two instructions earlier
#define MNC_C11_STEP(L, FL, M, FM, LIVE)   \
  load(L, FL);                             \
  __builtin_amdgcn_sched_barrier(0);       \
  mm(FM, LIVE);                            \
  __builtin_amdgcn_sched_barrier(0);
  Frag F0, F1, F2, F3;
  load(0, F0);
  load(1, F1);
  load(2, F2);
  __builtin_amdgcn_sched_barrier(0);
  for (int ks = 0; ks < KS; ks += 4) {
    MNC_C11_STEP(ks + 3, F3, ks, F0, true)
    MNC_C11_STEP(ks + 4, F0, ks + 1, F1, ks + 1 < KS)
    MNC_C11_STEP(ks + 5, F1, ks + 2, F2, ks + 2 < KS)
    MNC_C11_STEP(ks + 6, F2, ks + 3, F3, ks + 3 < KS)
  }
#undef MNC_C11_STEP

  // epilogue: per 32x32 tile a lane holds channels 8g + 4kb + 0..3 (registers 4g..4g+3) of pixel j: half a c8 pixel, i.e.
  // one float4 of the fp32 tensor or one 8-byte group of the packed fp16 tensor, both at index (block * P + pixel) * 2 + kb.
  // Loads (bias, residual) use clamped addresses and are issued in batches ahead of their use -- a load under a per-lane branch
  // would be waited for one at a time; only the stores are masked.
  int pj[PT];
  bool pok[PT];
#pragma unroll
  for (int jj = 0; jj < PT; ++jj) {
    pok[jj] = pbase + jj * 32 + j < P;
    pj[jj] = min(pbase + jj * 32 + j, P - 1);
  }
  auto chan = [&](int c, int g) { return (cg * CT + c) * 32 + g * 8 + kb * 4; };
  auto offs = [&](int c, int jj, int g) { return (unsigned)(((min(chan(c, g), Cout - 4) >> 3) * P + pj[jj]) * 2 + kb); };   // < 2^31: checked by the launcher
  auto finish = [&](int c, int jj, int g, const float4 bv, const float4 rv) {
    float4 v = make_float4(acc[c][jj][g * 4 + 0] + bv.x, acc[c][jj][g * 4 + 1] + bv.y, acc[c][jj][g * 4 + 2] + bv.z,
                           acc[c][jj][g * 4 + 3] + bv.w);
    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (pok[jj] && chan(c, g) < Cout) {
      const unsigned o = offs(c, jj, g);
      if (out_pk) reinterpret_cast<uint2*>(out)[o] = x3_f16x4(v);
      else reinterpret_cast<float4*>(out)[o] = v;
    }
  };
  // one 32-channel tile at a time: its bias and residual loads go out together, then the tile is finished and stored
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    float4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4*>(bias + min(chan(c, g), Cout - 4));
    if (F16 && res && res_pk) {
      uint2 rr[PT][4];
#pragma unroll
      for (int jj = 0; jj < PT; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) rr[jj][g] = reinterpret_cast<const uint2*>(res)[offs(c, jj, g)];
#pragma unroll
      for (int jj = 0; jj < PT; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
          const f16x4 h = __builtin_bit_cast(f16x4, rr[jj][g]);
          finish(c, jj, g, bv[g], make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]));
        }
    } else if (res) {
      float4 rr[PT][4];
#pragma unroll
      for (int jj = 0; jj < PT; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) rr[jj][g] = reinterpret_cast<const float4*>(res)[offs(c, jj, g)];
#pragma unroll
      for (int jj = 0; jj < PT; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) finish(c, jj, g, bv[g], rr[jj][g]);
    } else {
#pragma unroll
      for (int jj = 0; jj < PT; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) finish(c, jj, g, bv[g], zero4);
    }
    __builtin_amdgcn_sched_barrier(0);             // keeps the next tile's accumulator reads (AGPR -> VGPR copies) from being hoisted
  }
}

template <int F16>
static int conv1x1_launch(mnc_ctx* ctx, const char* scope, const void* d_in, const void* d_w, const float* d_bias,
                          const void* d_res, void* d_out, int H, int W, int Cin, int Cout, int stride, int relu, int res_pk,
                          int out_pk) {
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int P = OH * OW;
  const int KS = Cin / (F16 ? 16 : 8), CoT = cdiv(Cout, 32);
  MNC_REQUIRE((double)(Cout / 8) * P * 2.0 < 2.0e9 && (double)H * W * (Cin / 8) * 2.0 < 2.0e9, "%s: tensor too large for 32-bit offsets", scope);
  // Tile per wave.  Measured on the ResNet-50 C4 shapes at 800x1333 (tools/kernel_bench.py conv1x1, MNC_CONV1X1_TILE sweep):
  // f16 is bound by HBM and by the length of the K chain, not by the matrix pipe -- 32 channels x 64 pixels (3 waves per SIMD)
  // is best or within 5 % of the best on every layer (0.41 ms for the 29 layers; 128 x 64: 0.57 ms).  fp32 is bound by the
  // matrix pipe and by how evenly the waves fill the 1024 SIMDs: 64 x 64 while that gives >= 512 workgroups, thinner after.
  int ct, pt = 2;
  if (F16) {
    ct = 1;
  } else {
    ct = CoT >= 2 ? 2 : 1;
    auto blocks = [&]() { return (long)cdiv(P, 128 * pt) * cdiv(CoT, ct); };
    if (blocks() < 512) pt = 1;
    while (blocks() < 512 && ct > 1) ct >>= 1;
  }
  if (tune_set(ctx, T_CONV1X1_TILE)) {                        // override "ct,pt" (mnc_ctx_set_tuning: ct * 1000 + pt)
    const int a = tune(ctx, T_CONV1X1_TILE, 0) / 1000, b = tune(ctx, T_CONV1X1_TILE, 0) % 1000;
    if ((a == 1 || a == 2 || a == 4) && (b == 1 || b == 2)) { ct = a; pt = b; }
  }
  const int npg = cdiv(P, 128 * pt), ncg = cdiv(CoT, ct);
  const double act_b = F16 ? 2.0 : 4.0;
  const double bytes = act_b * (double)P * Cin + (out_pk ? 2.0 : 4.0) * (double)P * Cout +
                       (d_res ? (res_pk ? 2.0 : 4.0) * (double)P * Cout : 0.0) + act_b * (double)Cin * Cout;
  LaunchScope ls(ctx, scope, 2.0 * (double)P * Cout * Cin, bytes);
#define MNC_C11_LAUNCH(CT, PT)                                                                                              \
  hipLaunchKernelGGL((conv1x1_direct_kernel<CT, PT, F16>), dim3((unsigned)(npg * ncg)), dim3(256), 0, ctx->stream,          \
                     (const uint4*)d_in, (const uint4*)d_w, d_bias, d_res, d_out, H, W, OW, P, KS, CoT, Cout, stride, relu, \
                     res_pk, out_pk, npg, ncg)
  if (ct == 4 && pt == 2) MNC_C11_LAUNCH(4, 2);
  else if (ct == 4) MNC_C11_LAUNCH(4, 1);
  else if (ct == 2 && pt == 2) MNC_C11_LAUNCH(2, 2);
  else if (ct == 2) MNC_C11_LAUNCH(2, 1);
  else if (pt == 2) MNC_C11_LAUNCH(1, 2);
  else MNC_C11_LAUNCH(1, 1);
#undef MNC_C11_LAUNCH
  return ls.finish("conv1x1_direct_kernel");
}

}  // namespace mnc

using namespace mnc;

extern "C" {

int mnc_pack_conv1x1(mnc_ctx* ctx, const float* d_w, void* d_packed, int Cout, int Cin, int f16) {
  MNC_REQUIRE(ctx && d_w && d_packed, "mnc_pack_conv1x1: null pointer");
  MNC_REQUIRE(Cout > 0 && Cout % 8 == 0 && Cin > 0 && Cin % (f16 ? 16 : 8) == 0,
              "mnc_pack_conv1x1: bad shape (Cout%%8==0, Cin%%8==0, f16: Cin%%16==0)");
  const int CoT = cdiv(Cout, 32);
  const long total = (long)(Cin / (f16 ? 16 : 8)) * CoT * 64;
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  LaunchScope ls(ctx, "pack_conv1x1");
  if (f16) hipLaunchKernelGGL(pack_conv1x1_kernel<1>, dim3((int)g), dim3(256), 0, ctx->stream, d_w, (uint4*)d_packed, Cout, Cin, CoT);
  else hipLaunchKernelGGL(pack_conv1x1_kernel<0>, dim3((int)g), dim3(256), 0, ctx->stream, d_w, (uint4*)d_packed, Cout, Cin, CoT);
  return ls.finish("pack_conv1x1_kernel");
}

int mnc_conv1x1(mnc_ctx* ctx, const float* d_in, const void* d_w, const float* d_bias, const float* d_residual, float* d_out,
                int H, int W, int Cin, int Cout, int stride, int relu) {
  MNC_REQUIRE(ctx && d_in && d_w && d_bias && d_out, "mnc_conv1x1: null pointer");
  MNC_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 8 == 0 && (stride == 1 || stride == 2),
              "mnc_conv1x1: bad shape (Cin, Cout multiples of 8; stride 1 or 2)");
  return conv1x1_launch<0>(ctx, "conv1x1_mfma", d_in, d_w, d_bias, d_residual, d_out, H, W, Cin, Cout, stride, relu, 0, 0);
}

int mnc_conv1x1_f16_pk(mnc_ctx* ctx, const void* d_in, const void* d_w, const float* d_bias, const void* d_residual, void* d_out,
                       int H, int W, int Cin, int Cout, int stride, int relu, int res_packed, int out_packed) {
  MNC_REQUIRE(ctx && d_in && d_w && d_bias && d_out, "mnc_conv1x1_f16_pk: null pointer");
  MNC_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cout > 0 && Cout % 8 == 0 && (stride == 1 || stride == 2),
              "mnc_conv1x1_f16_pk: bad shape (Cin multiple of 16, Cout of 8; stride 1 or 2)");
  return conv1x1_launch<1>(ctx, "conv1x1_f16", d_in, d_w, d_bias, d_residual, d_out, H, W, Cin, Cout, stride, relu,
                           res_packed ? 1 : 0, out_packed ? 1 : 0);
}

}  // extern "C"
