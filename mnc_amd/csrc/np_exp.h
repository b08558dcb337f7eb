// float32 exp with the bits of numpy's `np.exp` on float32 data.
//
// The reference decodes boxes with `np.exp(dw) * widths` on float32 arrays (lib/transform/bbox_transform.py:88-89, called
// from lib/pylayer/proposal_layer.py:121 and lib/pylayer/stage_bridge_layer.py:241), so "the reference's float32 bits" of
// every proposal and every stage-2 box are whatever numpy's float32 exp loop returns.  On x86 builds with AVX2+FMA3 or
// AVX512F (every numpy >= 1.17 wheel; dispatched at run time) that loop is NOT libm's expf: it is numpy's own vector
// routine (numpy/core/src/umath/loops_exponent_log.dispatch.c.src, `simd_exp_FLOAT`, numpy 1.17 .. 2.2), up to 2 ulp from the
// correctly rounded value and different from it on ~39 % of inputs.  This header restates that published algorithm
// operation by operation:
//     q = rint(x * log2(e))                       (round-to-nearest through the 1.5 * 2^23 magic constant)
//     r = fma(q, -ln2_lo, fma(q, -ln2_hi, x))     (Cody-Waite, two constants)
//     exp(r) ~ P5(r) / Q2(r)                      (Horner with fma, IEEE division)
//     result = ldexp(P/Q, q);   x >= 88.7228... -> +inf,  x <= -103.972... -> 0,  NaN -> NaN
// tests/test_np_exp.py compiles it for the host and checks it bit for bit against np.exp on 2^26 float32 bit patterns spread
// evenly over the whole encoding space (contiguous and strided views); the device build uses the same source with the
// correctly rounded device intrinsics.
#pragma once

#if defined(__HIPCC__)
#define MNC_NPEXP_FN __host__ __device__ __forceinline__
#else
#include <math.h>
#define MNC_NPEXP_FN static inline
#endif

namespace mnc {

MNC_NPEXP_FN float np_exp_f32(float x0) {
  const float xmax = 88.72283935546875f, xmin = -103.97208404541015625f;
  const float ln2_hi = -6.93145752e-1f, ln2_lo = -1.42860677e-6f;       // NPY_CODY_WAITE_LOGE_2_{HIGH,LOW}f
  const float p0 = 9.999999999980870924916e-01f, p1 = 7.257664613233124478488e-01f, p2 = 2.473615434895520810817e-01f,
              p3 = 5.114512081637298353406e-02f, p4 = 6.757896990527504603057e-03f, p5 = 5.082762527590693718096e-04f;
  const float q0 = 1.0f, q1 = -2.742335390411667452936e-01f, q2 = 2.159509375685829852307e-02f;
  const float magic = 12582912.0f;                                       // 0x1.8p23: adding and subtracting it rounds to an integer
  const float log2e = 1.442695040888963407359924681001892137f;
  if (x0 != x0) return x0;
  if (x0 >= xmax) return __builtin_inff();
  if (x0 <= xmin) return 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
  const float quadrant = __fsub_rn(__fadd_rn(__fmul_rn(x0, log2e), magic), magic);
  float r = __fmaf_rn(quadrant, ln2_hi, x0);
  r = __fmaf_rn(quadrant, ln2_lo, r);
  float num = __fmaf_rn(p5, r, p4);
  num = __fmaf_rn(num, r, p3);
  num = __fmaf_rn(num, r, p2);
  num = __fmaf_rn(num, r, p1);
  num = __fmaf_rn(num, r, p0);
  float den = __fmaf_rn(q2, r, q1);
  den = __fmaf_rn(den, r, q0);
  return ldexpf(__fdiv_rn(num, den), (int)quadrant);
#else
  volatile float t = x0 * log2e;          // volatile: no re-association / contraction of the rounding trick on the host
  volatile float u = t + magic;
  const float quadrant = u - magic;
  float r = fmaf(quadrant, ln2_hi, x0);
  r = fmaf(quadrant, ln2_lo, r);
  float num = fmaf(p5, r, p4);
  num = fmaf(num, r, p3);
  num = fmaf(num, r, p2);
  num = fmaf(num, r, p1);
  num = fmaf(num, r, p0);
  float den = fmaf(q2, r, q1);
  den = fmaf(den, r, q0);
  volatile float ratio = num / den;
  return ldexpf(ratio, (int)quadrant);
#endif
}

}  // namespace mnc
