// Split-precision helpers shared by the bf16x3 kernels (gemm_x3.hip, conv_sw.hip).
//
// x = hi + lo with hi, lo bf16.  A product a*b is evaluated as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on
// v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are exact in the fp32 accumulator); the dropped terms are a_lo*b_lo
// (2^-16 relative) and the representation error of lo (2^-17 relative), i.e. ~1e-5 per product and less on a dot product.
#ifndef MNC_X3_SPLIT_H_
#define MNC_X3_SPLIT_H_

#include <hip/hip_runtime.h>

namespace mnc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned x3_pack_hi16(unsigned x0, unsigned x1) {      // {x1[31:16], x0[31:16]}
  return __builtin_amdgcn_perm(x1, x0, 0x07060302u);
}

// Activations, split while they are staged into LDS: hi = x with the low 16 bits cleared (one v_and), lo = x - hi (exact
// in fp32) rounded half-up to bf16 (one v_add); two values are packed per dword with one v_perm_b32: 3.5 VALU per value
// (the compiler's `(__bf16)x` round-to-nearest-even sequence costs ~10).
__device__ __forceinline__ void x3_split(float x, unsigned& h, unsigned& l) {
  h = __float_as_uint(x) & 0xFFFF0000u;
  l = __float_as_uint(x - __uint_as_float(h)) + 0x8000u;
}
__device__ __forceinline__ void x3_split8(const float4 a, const float4 b, uint4& hi, uint4& lo) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x3_split(x[e], h[e], l[e]);
  hi = make_uint4(x3_pack_hi16(h[0], h[1]), x3_pack_hi16(h[2], h[3]), x3_pack_hi16(h[4], h[5]), x3_pack_hi16(h[6], h[7]));
  lo = make_uint4(x3_pack_hi16(l[0], l[1]), x3_pack_hi16(l[2], l[3]), x3_pack_hi16(l[4], l[5]), x3_pack_hi16(l[6], l[7]));
}
__device__ __forceinline__ void x3_split4(const float4 a, uint2& hi, uint2& lo) {
  unsigned h[4], l[4];
  x3_split(a.x, h[0], l[0]);
  x3_split(a.y, h[1], l[1]);
  x3_split(a.z, h[2], l[2]);
  x3_split(a.w, h[3], l[3]);
  hi = make_uint2(x3_pack_hi16(h[0], h[1]), x3_pack_hi16(h[2], h[3]));
  lo = make_uint2(x3_pack_hi16(l[0], l[1]), x3_pack_hi16(l[2], l[3]));
}

// Weights, split once when they are packed: both terms rounded to nearest even.
__device__ __forceinline__ unsigned x3_rne(float x) {
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}
__device__ __forceinline__ void x3_split8_rne(const float* x, uint4& hi, uint4& lo) {
  unsigned h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = x3_rne(x[e]);
    l[e] = x3_rne(x[e] - __uint_as_float(h[e]));
  }
  hi = make_uint4(x3_pack_hi16(h[0], h[1]), x3_pack_hi16(h[2], h[3]), x3_pack_hi16(h[4], h[5]), x3_pack_hi16(h[6], h[7]));
  lo = make_uint4(x3_pack_hi16(l[0], l[1]), x3_pack_hi16(l[2], l[3]), x3_pack_hi16(l[4], l[5]), x3_pack_hi16(l[6], l[7]));
}

// Plain bf16 ("bf16" math mode, round 4: BASELINE configs[2] as written -- ONE bf16 product per term): four fp32 values rounded to
// nearest-even bf16 and packed (v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ uint2 x3_bf16x4(const float4 v) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  const bf16x4 hv = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  return __builtin_bit_cast(uint2, hv);
}

// 4 consecutive channels (channel-block half kb) of pixel `pix` of a c8 plane set, in the activation format of the math mode:
// fp32 [..][8] float | packed bf16x3 [..][hi x8 | lo x8] (the x3_split of the value) | packed f16 [..][8] fp16 (nearest even) |
// F16 == 2: packed bf16 [..][8] bf16 (nearest even)
template <int F16, bool PACKED>
__device__ __forceinline__ void x3_store4(void* out, long pix, int kb, const float4 v) {
  if (!PACKED) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + pix * 8 + kb * 4) = v;
  } else if (F16 == 2) {
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned*>(out) + pix * 4 + kb * 2) = x3_bf16x4(v);
  } else if (F16) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const f16x4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned*>(out) + pix * 4 + kb * 2) = __builtin_bit_cast(uint2, hv);
  } else {
    uint2 hi, lo;
    x3_split4(v, hi, lo);
    unsigned* p = reinterpret_cast<unsigned*>(out) + pix * 8 + kb * 2;
    *reinterpret_cast<uint2*>(p) = hi;
    *reinterpret_cast<uint2*>(p + 4) = lo;
  }
}

__device__ __forceinline__ uint2 x3_f16x4(const float4 v) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const f16x4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  return __builtin_bit_cast(uint2, hv);
}

// all 8 channels of pixel `pix` (a = channels 0-3, b = 4-7): 16-byte stores
template <int F16, bool PACKED>
__device__ __forceinline__ void x3_store8(void* out, long pix, const float4 a, const float4 b) {
  if (!PACKED) {
    float4* p = reinterpret_cast<float4*>(out) + pix * 2;
    p[0] = a;
    p[1] = b;
  } else if (F16) {
    const uint2 lo = F16 == 2 ? x3_bf16x4(a) : x3_f16x4(a), hi = F16 == 2 ? x3_bf16x4(b) : x3_f16x4(b);
    reinterpret_cast<uint4*>(out)[pix] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  } else {
    uint2 ah, al, bh, bl;
    x3_split4(a, ah, al);
    x3_split4(b, bh, bl);
    uint4* p = reinterpret_cast<uint4*>(out) + pix * 2;
    p[0] = make_uint4(ah.x, ah.y, bh.x, bh.y);
    p[1] = make_uint4(al.x, al.y, bl.x, bl.y);
  }
}

__device__ __forceinline__ f16x8 x3_as_f16x8(const uint4 v) { return __builtin_bit_cast(f16x8, v); }

__device__ __forceinline__ bf16x8 x3_as_bf16x8(const uint4 v) {
  union { uint4 u; bf16x8 b; } c;
  c.u = v;
  return c.b;
}

// Second output of a producer of InnerProduct activations (the per-RoI kernels of roi.hip, the split-K reduction of gemm.hip): the tensor in the stage-major 2-byte form the reduced-precision InnerProducts multiply
// from (mnc_hip.h: mnc_fc_{f16,bf16x3}_pre), written by the threads that hold the fp32 values -- the FC's own conversion pass
// (read M x K fp32, write M x K halves; 0.2 ms per image at 300 RoIs, 0.75 ms at 1000 RoIs x 1024 channels) disappears.
// SM: 0 none, 1 = fp16 [K/64][M][64], 2 = split bf16 [K/32][M][4][hi x8 | lo x8], 3 = bf16 [K/64][M][64]; k = position * C + channel.
// A thread owns 4 consecutive channels (k a multiple of 4), its neighbour lane (lane ^ 1) the other half of the same 8-channel
// group: the two exchange halves so that every store is a full 16-byte group (8-byte stores from every lane measured 15-20 %
// slower on these kernels: 521 vs 430 us for the 14x14 warp of 1000 RoIs x 1024 channels).  All 64 lanes must call it together
// (the callers' element counts are multiples of 64 per wave: C % 8 == 0 and whole positions).
template <int SM>
__device__ __forceinline__ void sm_store4(void* __restrict__ sm, long M, long r, long k, const float4 v) {
  const bool odd = (k >> 2) & 1;
  const long k8 = k & ~7L;
  if (SM == 1 || SM == 3) {                          // 3 (round 6): fp16's layout, the values rounded to bf16 (the plain bf16 mode)
    const uint2 mine = SM == 3 ? x3_bf16x4(v) : x3_f16x4(v);
    const unsigned ox = __shfl_xor(mine.x, 1), oy = __shfl_xor(mine.y, 1);
    if (!odd) reinterpret_cast<uint4*>(sm)[((k8 >> 6) * M + r) * 8 + ((k8 & 63) >> 3)] = make_uint4(mine.x, mine.y, ox, oy);
  } else if (SM == 2) {
    unsigned h[4], l[4];
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = x3_rne(x[e]);
      l[e] = x3_rne(x[e] - __uint_as_float(h[e]));
    }
    const uint2 hi = make_uint2(x3_pack_hi16(h[0], h[1]), x3_pack_hi16(h[2], h[3]));
    const uint2 lo = make_uint2(x3_pack_hi16(l[0], l[1]), x3_pack_hi16(l[2], l[3]));
    // the even lane writes the group's hi x8 (its own half, then the neighbour's), the odd lane the lo x8 (neighbour's, then own)
    const uint2 give = odd ? hi : lo;
    const uint2 got = make_uint2(__shfl_xor(give.x, 1), __shfl_xor(give.y, 1));
    uint4* p = reinterpret_cast<uint4*>(sm) + (((k8 >> 5) * M + r) * 4 + ((k8 & 31) >> 3)) * 2;
    if (!odd) p[0] = make_uint4(hi.x, hi.y, got.x, got.y);
    else p[1] = make_uint4(got.x, got.y, lo.x, lo.y);
  }
}

}  // namespace mnc
#endif
