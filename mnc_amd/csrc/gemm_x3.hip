// InnerProduct on the bf16 matrix pipe with fp32-class accuracy ("bf16x3" split precision) for gfx950.
//
// Every fp32 operand is split into two bf16 terms, x = hi + lo (x3_split.h); a product is evaluated as
// a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  with three v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are exact in
// the fp32 accumulator).  The dropped terms are O(2^-16) relative, i.e. ~1e-5 per product and less on a dot product --
// two orders of magnitude inside the 1e-3 parity bar (tests/test_gpu_ops.py::test_fc_bf16x3), while the matrix pipe
// runs the three bf16 MFMAs 5.3x faster than the eight fp32 MFMAs they replace (16 K-values in 3 x 32 cycles instead of
// 8 x 64).  At M = 300 that moves the FC layers from MFMA-bound to weight-streaming-bound.
//
// Same tiling as gemm.hip (320 rows x 128 columns x one K split per workgroup, 32-deep stages, one barrier per stage):
//   * weights are split ONCE at load (mnc_pack_fc_bf16x3): [N][K/8][hi x8 | lo x8] bf16 -- 32 B per 8 values, the same
//     bytes as fp32, so a stage's weight panel is still a linear 16 KB copy;
//   * activations stay fp32 in HBM and are split while they are staged into LDS (v_cvt_pk_bf16_f32 + one subtract per
//     value, hidden behind the MFMAs);
//   * LDS rows are 4 groups x 32 B + 16 B pad (pitch 36 dwords, conflict-free ds_read_b128); lane (row, kb) of K-step ks
//     reads group 2*ks + kb: hi and lo are two adjacent 16-byte fragments.
#include "mnc_internal.h"
#include "x3_split.h"

namespace mnc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kXBN = 128, kXBK = 32;
constexpr int kXPitch = 36;            // dwords per LDS row: 4 groups x 8 dwords + 4 pad
constexpr int kXBVec = kXBN * 8;       // uint4 items of the weight panel per stage (128 rows x 128 B)
constexpr int kXBPer = kXBVec / 256;   // 4

__device__ __forceinline__ float x3_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return 1.0f / (1.0f + expf(-v));
  return v;
}

template <int kMT>
__global__ __launch_bounds__(256) void fc_x3_kernel(const float* __restrict__ A, const uint4* __restrict__ Wx,
                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                    float* __restrict__ part, int M, int N, int K, int ldc, int kper,
                                                    int act, int fused, int tn_, int splits_, int tm_) {
  constexpr int kBM = 32 * kMT;
  constexpr int kAPer = (kBM * 4 + 255) / 256;          // 8-value groups of the A panel per thread
  __shared__ __attribute__((aligned(16))) unsigned sA[2][kBM * kXPitch];
  __shared__ __attribute__((aligned(16))) unsigned sB[2][kXBN * kXPitch];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, kb = lane >> 5;
  int bn, split, bmz;
  xcd_decode(blockIdx.x, tn_, splits_, tm_, bn, split, bmz);
  const int n0 = bn * kXBN, m0 = bmz * kBM;
  const int kbeg = split * kper, kend = min(K, kbeg + kper);
  const int nstages = (kend - kbeg) / kXBK;
  const int mrows = min(M - m0, kBM);
  const int mtiles = (mrows + 31) >> 5;
  const int groups_per_row = K >> 3;                    // uint4 pairs per weight row

  // staging map.  A: item q -> row q>>2, group q&3 (8 floats = 2 float4).  B: item q -> row q>>3, uint4 q&7.
  // Every staging load/store below is UNCONDITIONAL (ragged items are clamped onto the last row and simply rewrite it):
  // a load under a branch makes hipcc lose its vmcnt bookkeeping and drain the whole prefetch pipeline with
  // s_waitcnt vmcnt(0) every stage.
  const float* a_src[kAPer];
  int a_dst[kAPer];
#pragma unroll
  for (int u = 0; u < kAPer; ++u) {
    const int q = tid + u * 256, r = min(q >> 2, kBM - 1), g = q & 3;
    const int gr = m0 + min(r, mrows - 1);
    a_src[u] = A + (long)gr * K + kbeg + g * 8;
    a_dst[u] = r * kXPitch + g * 8;
  }
  const uint4* b_src[kXBPer];
  int b_dst[kXBPer];
#pragma unroll
  for (int u = 0; u < kXBPer; ++u) {
    const int q = tid + u * 256, r = q >> 3, c = q & 7;
    const int gr = min(n0 + r, N - 1);
    b_src[u] = Wx + ((long)gr * groups_per_row + (kbeg >> 3)) * 2 + c;
    b_dst[u] = r * kXPitch + c * 4;
  }
  // Two register sets (R0/R1): the loads of stage s+2 are in flight while stage s is multiplied and stage s+1 -- already
  // in registers -- is split into bf16 hi/lo and written to the free LDS buffer.  One barrier per stage; the global-load
  // latency gets a whole stage to hide and the VALU split / ds_write work overlaps the MFMAs instead of preceding them.
  struct Regs { float4 a[kAPer][2]; uint4 b[kXBPer]; };
  Regs R0, R1;
#pragma unroll
  for (int u = 0; u < kAPer; ++u) R0.a[u][0] = R0.a[u][1] = R1.a[u][0] = R1.a[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < kXBPer; ++u) R0.b[u] = R1.b[u] = make_uint4(0, 0, 0, 0);

  // load_stage / store_stage are branch-free: the stage index is clamped to the last real stage and a phantom stage
  // (index >= nstages, needed when the stage count is odd) is stored as zeros, so it multiplies to nothing.  A uniform
  // branch around the loads would leave the compiler unsure how many loads are outstanding at the join and it then
  // over-waits (s_waitcnt vmcnt(0..12) on the loads it has just issued).
  auto load_stage = [&](int s, Regs& R) {
    const long off = (long)min(s, nstages - 1) * kXBK;
#pragma unroll
    for (int u = 0; u < kAPer; ++u) {
      const float4* p = reinterpret_cast<const float4*>(a_src[u] + off);
      R.a[u][0] = p[0];
      R.a[u][1] = p[1];
    }
#pragma unroll
    for (int u = 0; u < kXBPer; ++u) R.b[u] = b_src[u][off >> 2];       // 32 values = 4 groups = 8 uint4 per row per stage
  };
  auto store_stage = [&](int buf, const Regs& R, bool live) {
    const unsigned keep = live ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int u = 0; u < kAPer; ++u) {
      uint4 hi, lo;
      x3_split8(R.a[u][0], R.a[u][1], hi, lo);
      hi.x &= keep; hi.y &= keep; hi.z &= keep; hi.w &= keep;
      lo.x &= keep; lo.y &= keep; lo.z &= keep; lo.w &= keep;
      *reinterpret_cast<uint4*>(&sA[buf][a_dst[u]]) = hi;
      *reinterpret_cast<uint4*>(&sA[buf][a_dst[u] + 4]) = lo;
    }
#pragma unroll
    for (int u = 0; u < kXBPer; ++u) *reinterpret_cast<uint4*>(&sB[buf][b_dst[u]]) = R.b[u];
  };

  f32x16 acc[kMT];
#pragma unroll
  for (int t = 0; t < kMT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const int a_base = j * kXPitch + kb * 8;                       // + t*32*pitch + ks*16
  const int b_base = (wave * 32 + j) * kXPitch + kb * 8;
  auto kstep = [&](int buf, int ks) {
    const unsigned* pa = sA[buf];
    const unsigned* pb = sB[buf];
    union { uint4 u; bf16x8 v; } bh, bl;
    bh.u = *reinterpret_cast<const uint4*>(pb + b_base + ks * 16);
    bl.u = *reinterpret_cast<const uint4*>(pb + b_base + ks * 16 + 4);
    // NO per-tile branch here: all kMT row tiles are always multiplied (rows past M hold clamped copies and are never
    // stored).  A branch per tile splits the loop into basic blocks of 2 ds_reads + 3 MFMAs and exposes the full LDS
    // latency 20 times per stage (measured: 7500 cycles per stage instead of ~2500).
#pragma unroll
    for (int t = 0; t < kMT; ++t) {
      union { uint4 u; bf16x8 v; } ah, al;
      ah.u = *reinterpret_cast<const uint4*>(pa + a_base + t * 32 * kXPitch + ks * 16);
      al.u = *reinterpret_cast<const uint4*>(pa + a_base + t * 32 * kXPitch + ks * 16 + 4);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bh.v, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bl.v, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bh.v, acc[t], 0, 0, 0);
    }
  };
  // one pipeline step: stage s sits in LDS[buf], stage s+1 in `cur`, stage s+2 is requested into `nxt`
  auto step = [&](int s, int buf, Regs& cur, Regs& nxt) {
    load_stage(s + 2, nxt);
    kstep(buf, 0);
    store_stage(buf ^ 1, cur, s + 1 < nstages);
    kstep(buf, 1);
    __syncthreads();
  };

  if (nstages > 0) {
    load_stage(0, R0);
    store_stage(0, R0, true);
    load_stage(1, R0);
    __syncthreads();
    for (int s = 0; s < nstages; s += 2) {
      step(s, 0, R0, R1);
      step(s + 1, 1, R1, R0);       // for an odd stage count the last call multiplies the zero-filled phantom stage
    }
  }

  const int n = n0 + wave * 32 + j;
  if (n < N) {
    const float bv = fused ? bias[n] : 0.f;
#pragma unroll
    for (int t = 0; t < kMT; ++t) {
      if (t < mtiles) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * kb;
          if (m < M) {
            if (fused) out[(long)m * ldc + n] = x3_act(acc[t][e] + bv, act);
            else part[((long)split * M + m) * N + n] = acc[t][e];
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void fc_x3_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                           float* __restrict__ out, int M, int N, int ldc, int splits,
                                                           int act) {
  const long total = (long)M * N;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int n = (int)(idx % N);
    const long m = idx / N;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += part[(long)s * total + idx];
    out[m * ldc + n] = x3_act(v + bias[n], act);
  }
}

// fp32 [N][K] -> [N][K/8][hi x8 | lo x8]
__global__ void pack_x3_kernel(const float* __restrict__ in, uint4* __restrict__ out, long groups) {
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (long)gridDim.x * blockDim.x) {
    const float4* p = reinterpret_cast<const float4*>(in + g * 8);
    const float4 a = p[0], b = p[1];
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint4 hi, lo;
    x3_split8_rne(x, hi, lo);
    out[g * 2] = hi;
    out[g * 2 + 1] = lo;
  }
}

}  // namespace mnc

using namespace mnc;

extern "C" {

int mnc_pack_fc_bf16x3(mnc_ctx* ctx, const float* d_w, void* d_packed, int N, int K) {
  MNC_REQUIRE(ctx && d_w && d_packed && N > 0 && K > 0 && K % 8 == 0, "mnc_pack_fc_bf16x3: bad argument (K%%8==0)");
  LaunchScope ls(ctx, "pack_fc_bf16x3");
  const long groups = (long)N * (K / 8);
  long g = (groups + 255) / 256;
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(pack_x3_kernel, dim3((int)g), dim3(256), 0, ctx->stream, d_w, (uint4*)d_packed, groups);
  return ls.finish("pack_x3_kernel");
}

int mnc_fc_bf16x3(mnc_ctx* ctx, const float* d_a, const void* d_w_packed, const float* d_bias, float* d_out, int M, int N,
                  int K, int ldc, int act) {
  MNC_REQUIRE(ctx && d_a && d_w_packed && d_bias && d_out, "mnc_fc_bf16x3: null pointer");
  MNC_REQUIRE(M >= 0 && N > 0 && K > 0 && K % kXBK == 0 && ldc >= N && act >= 0 && act <= 2,
              "mnc_fc_bf16x3: unsupported shape M=%d N=%d K=%d ldc=%d act=%d (need K%%32==0)", M, N, K, ldc, act);
  if (M == 0) return MNC_OK;
  // row tiles per workgroup: 2 (64 rows) for the small GEMMs, else the smallest of {5, 10} that covers M in one block
  const bool small = 2.0 * M * (double)N * K < 2.0e9;
  const int mt = small ? 2 : (M <= 160 ? 5 : 10);
  const int bm = 32 * mt;
  const int tn = cdiv(N, kXBN), tm = cdiv(M, bm), stages = K / kXBK;
  int splits = cdiv(small ? 512 : 256, tn * tm);
  const int min_stages = small ? 2 : 8;
  if (splits > stages / min_stages) splits = stages / min_stages;
  if (splits < 1) splits = 1;
  const int kper = cdiv(stages, splits) * kXBK;
  splits = cdiv(K, kper);
  float* part = nullptr;
  if (splits > 1) {
    int rc = ensure_scratch(ctx, (size_t)splits * M * N * 4);
    if (rc) return rc;
    part = (float*)ctx->scratch;
  }
  const double flops = 2.0 * M * (double)N * K, bytes = 4.0 * ((double)N * K + (double)M * K + (double)M * N);
  {
    LaunchScope ls(ctx, small ? "fc_bf16x3_small" : "fc_bf16x3", flops, bytes);
    if (mt == 2)
      hipLaunchKernelGGL(fc_x3_kernel<2>, dim3(tn * splits * tm), dim3(256), 0, ctx->stream, d_a, (const uint4*)d_w_packed,
                         d_bias, d_out, part, M, N, K, ldc, kper, act, splits == 1 ? 1 : 0, tn, splits, tm);
    else if (mt == 5)
      hipLaunchKernelGGL(fc_x3_kernel<5>, dim3(tn * splits * tm), dim3(256), 0, ctx->stream, d_a, (const uint4*)d_w_packed,
                         d_bias, d_out, part, M, N, K, ldc, kper, act, splits == 1 ? 1 : 0, tn, splits, tm);
    else
      hipLaunchKernelGGL(fc_x3_kernel<10>, dim3(tn * splits * tm), dim3(256), 0, ctx->stream, d_a, (const uint4*)d_w_packed,
                         d_bias, d_out, part, M, N, K, ldc, kper, act, splits == 1 ? 1 : 0, tn, splits, tm);
    int rc = ls.finish("fc_x3_kernel");
    if (rc) return rc;
  }
  if (splits > 1) {
    LaunchScope ls(ctx, "fc_reduce", 0.0, 4.0 * ((double)splits + 1.0) * M * N);
    long total = (long)M * N;
    int g = (int)((total + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(fc_x3_reduce_kernel, dim3(g), dim3(256), 0, ctx->stream, part, d_bias, d_out, M, N, ldc, splits, act);
    return ls.finish("fc_x3_reduce_kernel");
  }
  return MNC_OK;
}

}  // extern "C"
