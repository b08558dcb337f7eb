// 3x3 convolution on the bf16 matrix pipe with fp32-class accuracy ("bf16x3" split precision, x3_split.h) for gfx950
// (models/VGG16/mnc_5stage/test.prototxt:41-412; BASELINE.json configs[2] "bf16 convs via MFMA").
//
// Same implicit GEMM as conv.hip (M = output channels = MFMA A operand, N = pixels = B operand, c8 feature maps in and
// out, fp32 in HBM), but the contraction runs on v_mfma_f32_32x32x16_bf16: three MFMAs (a_lo*b_hi, a_hi*b_lo, a_hi*b_hi)
// cover 16 K-values in 3 x 32 cycles where the fp32 pipe needs 8 x 64.
//
//   * K is walked in 8-channel blocks, as in conv.hip.  One MFMA K-step (16 values) = 8 channels x TWO taps: lanes 0-31
//     (k 0-7) take tap 2s, lanes 32-63 (k 8-15) take tap 2s+1, s = 0..4.  The tenth tap slot has zero weights
//     (1/10 of the MFMAs is padding; the matrix pipe is not what bounds this kernel, LDS bandwidth is).
//   * weights are split ONCE (mnc_pack_conv3x3_bf16x3): [Cin/8][Cout][84 dwords] = 10 tap slots x (hi x8 | lo x8) bf16
//     + 16 B pad -- a workgroup's panel per channel block is one linear copy into LDS, row pitch 84 dwords
//     (84/4 odd -> conflict-free ds_read_b128);
//   * activations stay fp32 in HBM and are split while the halo is staged into LDS: pixel = hi x8 (16 B) | lo x8 (16 B)
//     | 16 B pad, pitch 12 dwords as in conv.hip;
//   * a wave owns PR pixel rows x 32 columns x 32*CT output channels: per K-step it reads 2*(PR + CT) 16-byte fragments
//     for 3*PR*CT MFMAs -- the register tile is what keeps LDS traffic under the matrix pipe's appetite;
//   * staging loads are unconditional (clamped addresses, out-of-image pixels masked to zero) and issued for block c+1
//     before the MFMAs of block c, stored to the other LDS buffer after them: one barrier per block.
#include <atomic>
#include <cstdlib>

#include "mnc_internal.h"
#include "x3_split.h"

namespace mnc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kX3Cols = 32;
constexpr int kX3HaloCols = kX3Cols + 2;
constexpr int kX3PixPitch = 12;        // dwords per halo pixel in LDS: hi x8 | lo x8 | pad
constexpr int kX3WPitch = 84;          // dwords per weight row: 10 slots x 8 + 4 pad (global packed layout and LDS)

// F16 != 0: the "f16" math mode (BASELINE configs[4]) -- one product per term on v_mfma_f32_32x32x16_f16: activations are
// rounded to fp16 (nearest even) while the halo is staged and occupy the "hi" half of a pixel's LDS slot, the weights come
// from mnc_pack_conv3x3_f16 (fp16 in the hi halves of the same packed layout, lo halves zero); only the hi fragments are read
// and one MFMA per tile and K-step is issued instead of three.
template <int CT, int PR, int ROWS, int F16 = 0>
__global__ __launch_bounds__(64 * ROWS) void conv3x3_x3_kernel(const float* __restrict__ in, const uint4* __restrict__ wpk,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int H, int W, int Cin, int Cout, int relu, int ksplit,
                                                               float* __restrict__ part) {
  constexpr int NT = 64 * ROWS;
  constexpr int R = ROWS * PR;                               // pixel rows per workgroup
  constexpr int kHaloPix = (R + 2) * kX3HaloCols;
  constexpr int kHaloItems = kHaloPix * 2;                   // one item = 4 channels of one pixel (one float4)
  constexpr int kHPer = (kHaloItems + NT - 1) / NT;
  constexpr int NCO = 32 * CT;
  constexpr int kWVec = NCO * (kX3WPitch / 4);               // uint4 items per weight panel
  constexpr int kWPer = (kWVec + NT - 1) / NT;
  constexpr int kHaloDw = kHaloPix * kX3PixPitch;
  constexpr int kWDw = NCO * kX3WPitch;
  extern __shared__ __attribute__((aligned(16))) unsigned s_mem[];   // halo[2] then weights[2]
  unsigned* const s_halo = s_mem;
  unsigned* const s_w = s_mem + 2 * kHaloDw;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, kb = lane >> 5;
  // ksplit > 1 (small maps, see conv.hip): blockIdx.z = split * (Cout / NCO) + channel tile; raw partial sums go to part[split]
  const int ncot = Cout / NCO;
  const int split = blockIdx.z / ncot;
  const int w0 = blockIdx.x * kX3Cols, h0 = blockIdx.y * R, co0 = (blockIdx.z - split * ncot) * NCO;
  const int nchunks = (Cin >> 3) / ksplit;
  const int chunk0 = split * nchunks;

  // ---- staging assignment (fixed per thread): every thread always loads and stores; surplus threads repeat the last item
  int h_dst[kHPer];
  long h_src[kHPer];
  unsigned h_keep[kHPer];
#pragma unroll
  for (int u = 0; u < kHPer; ++u) {
    const int q = min(tid + u * NT, kHaloItems - 1);
    const int pix = q >> 1, half = q & 1;
    const int r = pix / kX3HaloCols, c = pix - r * kX3HaloCols;
    const int gh = h0 - 1 + r, gw = w0 - 1 + c;
    const bool inside = gh >= 0 && gh < H && gw >= 0 && gw < W;
    h_dst[u] = pix * kX3PixPitch + half * 2;
    h_src[u] = ((long)min(max(gh, 0), H - 1) * W + min(max(gw, 0), W - 1)) * 8 + half * 4;
    h_keep[u] = inside ? 0xFFFFFFFFu : 0u;
  }
  int w_idx[kWPer];
#pragma unroll
  for (int u = 0; u < kWPer; ++u) w_idx[u] = min(tid + u * NT, kWVec - 1);
  const long plane = (long)H * W * 8;

  // Two register sets: the loads for block c+2 are issued while block c is multiplied and block c+1 (requested one
  // iteration earlier, so certainly landed) is split and written to LDS.  Every s_waitcnt is then for data requested a full
  // iteration ago, wherever the scheduler places the loads, the splits and the LDS writes among the MFMAs.
  // (initialised: hipcc keeps arrays that a lambda writes first as allocas -> scratch otherwise)
  struct Regs {
    float4 h[kHPer];
    uint4 w[kWPer];
  };
  Regs R0, R1;
#pragma unroll
  for (int u = 0; u < kHPer; ++u) R0.h[u] = R1.h[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < kWPer; ++u) R0.w[u] = R1.w[u] = make_uint4(0, 0, 0, 0);
  auto load_chunk = [&](int c, Regs& G) {
    c = chunk0 + min(c, nchunks - 1);
    const float* src = in + (long)c * plane;
#pragma unroll
    for (int u = 0; u < kHPer; ++u) G.h[u] = *reinterpret_cast<const float4*>(src + h_src[u]);
    const uint4* wsrc = wpk + ((long)c * Cout + co0) * (kX3WPitch / 4);
#pragma unroll
    for (int u = 0; u < kWPer; ++u) G.w[u] = wsrc[w_idx[u]];
  };
  // live == false: a phantom block behind an odd block count -- its halo is stored as zeros, so it multiplies to nothing
  auto store_chunk = [&](int buf, const Regs& G, bool live) {
    unsigned* hdst = s_halo + buf * kHaloDw;
#pragma unroll
    for (int u = 0; u < kHPer; ++u) {
      const unsigned keep = live ? h_keep[u] : 0u;
      if (F16) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const f16x4 hv = {(_Float16)G.h[u].x, (_Float16)G.h[u].y, (_Float16)G.h[u].z, (_Float16)G.h[u].w};
        uint2 hi = __builtin_bit_cast(uint2, hv);
        hi.x &= keep; hi.y &= keep;
        *reinterpret_cast<uint2*>(hdst + h_dst[u]) = hi;
        continue;
      }
      uint2 hi, lo;
      x3_split4(G.h[u], hi, lo);
      hi.x &= keep; hi.y &= keep;
      lo.x &= keep; lo.y &= keep;
      *reinterpret_cast<uint2*>(hdst + h_dst[u]) = hi;
      *reinterpret_cast<uint2*>(hdst + h_dst[u] + 4) = lo;
    }
    uint4* wdst = reinterpret_cast<uint4*>(s_w + buf * kWDw);
#pragma unroll
    for (int u = 0; u < kWPer; ++u) wdst[w_idx[u]] = G.w[u];
  };

  f32x16 acc[PR][CT];
#pragma unroll
  for (int r = 0; r < PR; ++r)
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][t][e] = 0.f;

  // per-lane fragment addresses: K-step s reads tap 2s + kb (slot 9 = zero weights; its pixel read repeats tap 8)
  int p_off[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = min(2 * s + kb, 8);
    p_off[s] = ((wave * PR + tap / 3) * kX3HaloCols + j + tap % 3) * kX3PixPitch;
  }
  const int w_base = j * kX3WPitch + kb * 8;                 // + t*32*84 + s*16 (+4 for lo)

  auto multiply = [&](int buf) {
    const unsigned* sh = s_halo + buf * kHaloDw;
    const unsigned* sw = s_w + buf * kWDw;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      if (F16) {
        f16x8 b[PR], a[CT];
#pragma unroll
        for (int r = 0; r < PR; ++r)
          b[r] = x3_as_f16x8(*reinterpret_cast<const uint4*>(sh + p_off[s] + r * kX3HaloCols * kX3PixPitch));
#pragma unroll
        for (int t = 0; t < CT; ++t)
          a[t] = x3_as_f16x8(*reinterpret_cast<const uint4*>(sw + w_base + t * 32 * kX3WPitch + s * 16));
#pragma unroll
        for (int r = 0; r < PR; ++r)
#pragma unroll
          for (int t = 0; t < CT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], b[r], acc[r][t], 0, 0, 0);
        continue;
      }
      bf16x8 bh[PR], bl[PR], ah[CT], al[CT];
#pragma unroll
      for (int r = 0; r < PR; ++r) {
        const unsigned* p = sh + p_off[s] + r * kX3HaloCols * kX3PixPitch;
        bh[r] = x3_as_bf16x8(*reinterpret_cast<const uint4*>(p));
        bl[r] = x3_as_bf16x8(*reinterpret_cast<const uint4*>(p + 4));
      }
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const unsigned* p = sw + w_base + t * 32 * kX3WPitch + s * 16;
        ah[t] = x3_as_bf16x8(*reinterpret_cast<const uint4*>(p));
        al[t] = x3_as_bf16x8(*reinterpret_cast<const uint4*>(p + 4));
      }
      // term outermost: consecutive MFMAs never share an accumulator
#pragma unroll
      for (int r = 0; r < PR; ++r)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh[r], acc[r][t], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < PR; ++r)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl[r], acc[r][t], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < PR; ++r)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh[r], acc[r][t], 0, 0, 0);
    }
  };
  // block c sits in LDS[buf], block c+1 in `cur`, block c+2 is requested into `nxt`
  auto step = [&](int c, int buf, Regs& cur, Regs& nxt) {
    load_chunk(c + 2, nxt);
    multiply(buf);
    store_chunk(buf ^ 1, cur, c + 1 < nchunks);
    __syncthreads();
  };

  load_chunk(0, R0);
  store_chunk(0, R0, true);
  load_chunk(1, R0);
  __syncthreads();
  for (int c = 0; c < nchunks; c += 2) {
    step(c, 0, R0, R1);
    step(c + 1, 1, R1, R0);        // for an odd block count the last call multiplies the zero-filled phantom block
  }

  // ---- epilogue: D[row = cout (reg&3)+8*(reg>>2)+4*kb][col = pixel j] ----
  const int ow = w0 + j;
#pragma unroll
  for (int r = 0; r < PR; ++r) {
    const int oh = h0 + wave * PR + r;
    if (oh < H && ow < W) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = co0 + t * 32 + g * 8 + kb * 4;
          float4 v = make_float4(acc[r][t][4 * g + 0], acc[r][t][4 * g + 1], acc[r][t][4 * g + 2], acc[r][t][4 * g + 3]);
          float* dst = out;
          if (ksplit > 1) {
            dst = part + (long)split * Cout * H * W;
          } else {
            const float4 b = *reinterpret_cast<const float4*>(bias + co);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          }
          *reinterpret_cast<float4*>(dst + (((long)(co >> 3) * H + oh) * W + ow) * 8 + kb * 4) = v;
        }
      }
    }
  }
}

// OIHW fp32 -> [Cin/8][Cout][21 uint4]: slot t < 9: (hi x8 | lo x8) of w[co][cb*8 .. +8][tap t]; slot 9 and the pad: zero
// f16 != 0: hi = the 8 values rounded to fp16, lo = zero (conv3x3_x3_kernel<.., F16 = 1>)
__global__ void pack_conv_x3_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cout, int Cin, int f16) {
  const long total = (long)(Cin >> 3) * Cout * 11;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int slot = (int)(i % 11);
    const long row = i / 11;
    const int co = (int)(row % Cout), cb = (int)(row / Cout);
    uint4* dst = out + row * (kX3WPitch / 4);
    if (slot == 10) {
      dst[20] = make_uint4(0, 0, 0, 0);
      continue;
    }
    uint4 hi = make_uint4(0, 0, 0, 0), lo = hi;
    if (slot < 9) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = w[((long)co * Cin + cb * 8 + e) * 9 + slot];
      if (f16) {
        const f16x8 hv = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3],
                          (_Float16)x[4], (_Float16)x[5], (_Float16)x[6], (_Float16)x[7]};
        hi = __builtin_bit_cast(uint4, hv);
      } else {
        x3_split8_rne(x, hi, lo);
      }
    }
    dst[slot * 2] = hi;
    dst[slot * 2 + 1] = lo;
  }
}

static int x3_grid_for(long total) {
  const long g = (total + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

template <int CT, int PR, int ROWS, int F16>
static int launch_x3(mnc_ctx* ctx, const float* d_in, const void* d_wpk, const float* d_bias, float* d_out, int H, int W,
                     int Cin, int Cout, int relu, int ksplit, float* part) {
  constexpr int R = ROWS * PR;
  constexpr size_t lds = 2 * 4 * ((size_t)(R + 2) * kX3HaloCols * kX3PixPitch + (size_t)32 * CT * kX3WPitch);
  static_assert(lds <= 160 * 1024, "conv3x3_x3: LDS budget");
  auto kern = conv3x3_x3_kernel<CT, PR, ROWS, F16>;
  static std::atomic<unsigned long long> attr_set{0};          // one bit per device: function attributes are per device
  const unsigned long long bit = 1ull << (ctx->device & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  dim3 grid(cdiv(W, kX3Cols), cdiv(H, R), Cout / (32 * CT) * ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(64 * ROWS), lds, ctx->stream, d_in, (const uint4*)d_wpk, d_bias, d_out, H, W, Cin,
                     Cout, relu, ksplit, part);
  return MNC_OK;
}

}  // namespace mnc

using namespace mnc;

static int pack_conv_lowp(mnc_ctx* ctx, const char* name, const float* d_oihw, void* d_packed, int Cout, int Cin, int f16) {
  MNC_REQUIRE(ctx && d_oihw && d_packed && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0, "%s: bad argument", name);
  LaunchScope ls(ctx, name);
  hipLaunchKernelGGL(pack_conv_x3_kernel, dim3(x3_grid_for((long)(Cin / 8) * Cout * 11)), dim3(256), 0, ctx->stream, d_oihw,
                     (uint4*)d_packed, Cout, Cin, f16);
  return ls.finish("pack_conv_x3_kernel");
}

template <int F16>
static int conv3x3_lowp(mnc_ctx* ctx, const char* name, const float* d_in, const void* d_wpk, const float* d_bias, float* d_out,
                        int H, int W, int Cin, int Cout, int relu) {
  MNC_REQUIRE(ctx && d_in && d_wpk && d_bias && d_out, "%s: null pointer", name);
  MNC_REQUIRE(H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0,
              "%s: unsupported shape H=%d W=%d Cin=%d Cout=%d (need Cin%%8==0, Cout%%32==0)", name, H, W, Cin, Cout);
  // (channel tiles CT, pixel rows PR) per wave.  Measured on MI355X (tools/kernel_bench.py convx3, round 1): the 2x2
  // register tile (8 LDS fragment reads per 12 MFMAs, 2 workgroups per CU) wins wherever it still yields >= 512 workgroups
  // (conv1_2 .. conv3_3: 280-335 TF/s fp32-equivalent); on the small maps (conv4_x, conv5_x, rpn_conv) the 1x1 tile with
  // 4x the workgroups and 5 waves per SIMD is faster (175-285 vs 120-250).  4x2 (1 workgroup per CU) never wins.
  int ct = 1, pr = 1;
  if (Cout % 64 == 0 && (long)cdiv(W, kX3Cols) * cdiv(H, 8) * (Cout / 64) >= 512) ct = 2, pr = 2;
  if (const char* e = getenv("MNC_CONVX3_TILE")) {          // "CT,PR" tuning override
    int a = 0, b = 0;
    if (sscanf(e, "%d,%d", &a, &b) == 2 && (a == 1 || a == 2 || a == 4) && (b == 1 || b == 2) && Cout % (32 * a) == 0 &&
        !(a == 4 && b == 1) && !(a == 1 && b == 2)) {
      ct = a;
      pr = b;
    }
  }
  // K splits for the small maps (fewer than two workgroups per CU), as in mnc_conv3x3
  int ksplit = 1;
  {
    const long wgs = (long)cdiv(W, kX3Cols) * cdiv(H, 4 * pr) * (Cout / (32 * ct));
    const int blocks = Cin / 8;
    if (wgs < 512) ksplit = blocks % 4 == 0 && blocks >= 16 ? 4 : (blocks % 2 == 0 && blocks >= 8 ? 2 : 1);
    if (const char* e = getenv("MNC_CONV_KSPLIT")) {
      const int v = atoi(e);
      if (v >= 1 && v <= 8 && blocks % v == 0) ksplit = v;
    }
  }
  float* part = nullptr;
  if (ksplit > 1) {
    int rc = ensure_scratch(ctx, (size_t)ksplit * Cout * H * W * 4);
    if (rc) return rc;
    part = (float*)ctx->scratch;
  }
  const double flops = 2.0 * H * W * 9.0 * Cin * Cout;
  const double bytes = 4.0 * ((double)H * W * (Cin + Cout) + 9.0 * Cin * Cout);
  LaunchScope ls(ctx, name, flops, bytes);
  int rc = MNC_OK;
  if (ct == 4 && pr == 2) rc = launch_x3<4, 2, 4, F16>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part);
  else if (ct == 2 && pr == 2) rc = launch_x3<2, 2, 4, F16>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part);
  else if (ct == 2 && pr == 1) rc = launch_x3<2, 1, 4, F16>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part);
  else rc = launch_x3<1, 1, 4, F16>(ctx, d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu, ksplit, part);
  if (rc) return rc;
  if (ksplit > 1) conv_splitk_reduce_launch(ctx->stream, part, d_bias, d_out, H, W, Cout, ksplit, relu);
  return ls.finish("conv3x3_x3_kernel");
}

extern "C" {

int mnc_pack_conv3x3_bf16x3(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin) {
  return pack_conv_lowp(ctx, "pack_conv3x3_bf16x3", d_oihw, d_packed, Cout, Cin, 0);
}

int mnc_pack_conv3x3_f16(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin) {
  return pack_conv_lowp(ctx, "pack_conv3x3_f16", d_oihw, d_packed, Cout, Cin, 1);
}

int mnc_conv3x3_bf16x3(mnc_ctx* ctx, const float* d_in, const void* d_wpk, const float* d_bias, float* d_out, int H, int W,
                       int Cin, int Cout, int relu) {
  return conv3x3_lowp<0>(ctx, "conv3x3_bf16x3", d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu);
}

int mnc_conv3x3_f16(mnc_ctx* ctx, const float* d_in, const void* d_wpk, const float* d_bias, float* d_out, int H, int W, int Cin,
                    int Cout, int relu) {
  return conv3x3_lowp<1>(ctx, "conv3x3_f16", d_in, d_wpk, d_bias, d_out, H, W, Cin, Cout, relu);
}

}  // extern "C"
