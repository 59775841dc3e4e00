/*
 * coxgraph_hip_math.h -- asin / atan2 in plain IEEE float operations, shared by the HIP engine and its CPU checker.
 *
 * The projective integrator (voxblox ProjectiveTsdfIntegrator, selected by `method: "projective"` in
 * coxgraph/config/tsdf_server_default.yaml:6 and tsdf_server_carla.yaml:6) turns bearings into range-image pixels with
 * std::asin / std::atan2.  Those are libm calls: their last bit differs between glibc versions and between glibc and the
 * GPU's device library, and a last bit decides which pixel a bearing on a pixel border falls into.  So that the engine's
 * result is a FUNCTION of its inputs (and equal to the checker's, bit for bit), both sides evaluate the angles with the
 * polynomials below -- only + - * / and sqrt, compiled without contraction, the same sequence of roundings everywhere.
 * Accuracy: within 4 ulp of the exact value over the whole domain (tests/test_oracle_projective.py checks
 * against numpy's float64 results), a last-bit-class difference from libm, like the one between two libm versions.
 *
 * The approximations are the classical single-precision Cephes ones (asinf: odd polynomial in x on |x| <= 1/2, the
 * half-angle identity above; atanf: reduction with tan(3 pi / 8) and tan(pi / 8), odd polynomial on the reduced argument).
 */
#ifndef COXGRAPH_HIP_MATH_H_
#define COXGRAPH_HIP_MATH_H_

#include <math.h>

#if defined(__HIPCC__)
#define COX_MATH_FN static __host__ __device__ __forceinline__
#else
#define COX_MATH_FN static inline
#endif

COX_MATH_FN float cox_asinf(float xx) {
  float x = xx;
  int neg = 0;
  if (x < 0.0f) {
    x = -x;
    neg = 1;
  }
  if (x > 1.0f) return NAN;
  float z;
  int flag = 0;
  if (x > 0.5f) {
    z = 0.5f * (1.0f - x);
    x = sqrtf(z);
    flag = 1;
  } else {
    if (x < 1.0e-4f) return xx;
    z = x * x;
  }
  float p = 4.2163199048e-2f;
  p = p * z + 2.4181311049e-2f;
  p = p * z + 4.5470025998e-2f;
  p = p * z + 7.4953002686e-2f;
  p = p * z + 1.6666752422e-1f;
  z = (p * z) * x + x;
  if (flag) {
    z = z + z;
    z = 1.5707963267948966f - z;
  }
  return neg ? -z : z;
}

/* arc tangent of a non-negative argument, result in [0, pi / 2] */
COX_MATH_FN float cox_atanf_pos(float x) {
  float y;
  if (x > 2.414213562373095f) { /* tan(3 pi / 8) */
    y = 1.5707963267948966f;
    x = -(1.0f / x);
  } else if (x > 0.4142135623730950f) { /* tan(pi / 8) */
    y = 0.7853981633974483f;
    x = (x - 1.0f) / (x + 1.0f);
  } else {
    y = 0.0f;
  }
  const float z = x * x;
  float p = 8.05374449538e-2f;
  p = p * z - 1.38776856032e-1f;
  p = p * z + 1.99777106478e-1f;
  p = p * z - 3.33329491539e-1f;
  y = y + ((p * z) * x + x);
  return y;
}

/* std::atan2(y, x) for finite arguments: result in [-pi, pi], atan2(0, 0) = 0, atan2(+-0, x < 0) = +-pi */
COX_MATH_FN float cox_atan2f(float y, float x) {
  const float kPi = 3.14159265358979323846f, kHalfPi = 1.5707963267948966f;
  if (x != x || y != y) return NAN;
  if (y == 0.0f) {
    if (x > 0.0f || (x == 0.0f && !signbit(x))) return y; /* +-0 */
    return signbit(y) ? -kPi : kPi;
  }
  if (x == 0.0f) return y > 0.0f ? kHalfPi : -kHalfPi;
  const float ax = fabsf(x), ay = fabsf(y);
  float a = cox_atanf_pos(ay / ax); /* first-quadrant angle */
  if (x < 0.0f) a = kPi - a;
  return y < 0.0f ? -a : a;
}

#endif /* COXGRAPH_HIP_MATH_H_ */
