// fast_math.h -- arithmetic helpers of the fast-math kernels (step_fast.hip, step_coop.hip):
// polynomial exp2, reciprocal-based division, single-instruction clamps.  Internal header.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sipnet {
namespace {

constexpr double kTiny = 0.000001;
constexpr double kEps = 1e-8;
constexpr double kCWeight = 12.0, kTen9 = 1000000000.0, kSecPerDay = 86400.0;
constexpr double kLog2e = 1.4426950408889634074;

// ---- math ---------------------------------------------------------------------
__device__ __forceinline__ double ffma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// 2^x: n = rint(x), 2^(x-n) by a polynomial on [-0.5, 0.5] -- the interpolant of (2^f - 1)/f
// through the Chebyshev nodes, coefficients correctly rounded from a 50-digit computation
// (tools/fit_exp2.py prints them and the measured error) -- scaled with v_ldexp_f64.
//   degree 11: |rel err| <= 1.7e-16   degree 9: 3.7e-14 (the shipped build)   degree 8: 2.1e-12
// Degree 9 spends some of the tolerance (the bar is |dNEE| < 1e-6, the tests hold 1e-9): c10k's
// worst |dNEE| against the reference goes from 2.5e-16 to 1.4e-14 and the light wave's chain
// lai -> potential photosynthesis gets 14 instructions shorter (c10k -3 %, c4 -6 %); degree 8
// measured no faster than 9.
// The coefficients live in SGPR pairs for the whole time loop (a VOP3 fma takes one scalar
// operand): left as literals, hipcc re-materialises them with v_mov_b64 at each of the 13
// call sites of a step, which costs as much as the polynomial itself.
#ifndef SIPNET_EXP2_DEGREE
#define SIPNET_EXP2_DEGREE 9
#endif
struct Exp2Coef {
  double c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11;
};
__device__ __forceinline__ Exp2Coef loadExp2Coef() {
#if SIPNET_EXP2_DEGREE == 11
  Exp2Coef k = {0.6931471805599453,    0.24022650695910097,   0.0555041086648216,
                0.009618129107606888,  0.0013333558146416936, 0.0001540353044173605,
                1.525273382983612e-05, 1.321544258792169e-06, 1.0178062445845774e-07,
                7.072585949269223e-09, 4.4549605981865186e-10};
#elif SIPNET_EXP2_DEGREE == 9
  Exp2Coef k = {0.6931471805599453,    0.2402265069581299,     0.055504108664760424,
                0.009618129159402558,  0.0013333558179043112,  0.00015403455852453838,
                1.525268684626772e-05, 1.3255224878668817e-06, 1.0203121063391729e-07, 0.0, 0.0};
#elif SIPNET_EXP2_DEGREE == 8
  Exp2Coef k = {0.6931471805568324,     0.24022650695888503,   0.055504109063258665,
                0.00961812913523614,    0.0013333478473685416, 0.00015403475186530786,
                1.5303700711365693e-05, 1.325080551750225e-06, 0.0, 0.0, 0.0};
#else
#error "SIPNET_EXP2_DEGREE must be 8, 9 or 11 (tools/fit_exp2.py)"
#endif
  // opaque to constant propagation, pinned to scalar registers
  asm volatile("" : "+s"(k.c1), "+s"(k.c2), "+s"(k.c3), "+s"(k.c4), "+s"(k.c5), "+s"(k.c6));
  asm volatile("" : "+s"(k.c7), "+s"(k.c8));
#if SIPNET_EXP2_DEGREE >= 9
  asm volatile("" : "+s"(k.c9));
#endif
#if SIPNET_EXP2_DEGREE >= 11
  asm volatile("" : "+s"(k.c10), "+s"(k.c11));
#endif
  return k;
}
__device__ __forceinline__ double fexp2(double x, const Exp2Coef& k) {
  const double n = __builtin_rint(x);
  const double f = x - n;
#if SIPNET_EXP2_DEGREE == 11
  double p = k.c11;
  p = ffma(p, f, k.c10);
  p = ffma(p, f, k.c9);
  p = ffma(p, f, k.c8);
#elif SIPNET_EXP2_DEGREE == 9
  double p = k.c9;
  p = ffma(p, f, k.c8);
#else
  double p = k.c8;
#endif
  p = ffma(p, f, k.c7);
  p = ffma(p, f, k.c6);
  p = ffma(p, f, k.c5);
  p = ffma(p, f, k.c4);
  p = ffma(p, f, k.c3);
  p = ffma(p, f, k.c2);
  p = ffma(p, f, k.c1);
  p = ffma(p, f, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}
__device__ __forceinline__ float fexp2(float x, const Exp2Coef&) { return __builtin_amdgcn_exp2f(x); }

// a / b with b > 0 finite and well scaled: v_rcp + Newton steps + one residual step
#ifndef SIPNET_FDIV_NEWTON
#define SIPNET_FDIV_NEWTON 2
#endif
__device__ __forceinline__ double fdiv(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = ffma(ffma(-b, r, 1.0), r, r);
#if SIPNET_FDIV_NEWTON >= 2
  r = ffma(ffma(-b, r, 1.0), r, r);
#endif
  const double q = a * r;
  return ffma(ffma(-b, q, a), r, q);
}
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

__device__ __forceinline__ double flog2(double x) { return log2(x); }
__device__ __forceinline__ float flog2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ double fpow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float fpow(float x, float y) { return powf(x, y); }
// one v_max / v_min each (operands are never NaN here)
__device__ __forceinline__ double rminv(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float rminv(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double rmax0(double x) { return __builtin_fmax(x, 0.0); }
__device__ __forceinline__ float rmax0(float x) { return __builtin_fmaxf(x, 0.0f); }
__device__ __forceinline__ double clip01(double x) { return __builtin_fmin(__builtin_fmax(x, 0.0), 1.0); }
__device__ __forceinline__ float clip01(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }

__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// A NARROW field of the site record (plan.h, FastRec: climate values the step only ever uses at the
// precision of the flux arithmetic).  An fp64 batch holds a double there; the plan of an fp32-mixed
// batch holds the correctly rounded float in the slot's low word (its high word is a quiet-NaN tag, so
// that a reader that forgets this fails loudly): the conversion happens once per site on the host
// instead of in every wavefront on every step (7-14 v_cvt_f32_f64 of wave-uniform values per step).
template <class R>
__device__ __forceinline__ R recR(double slot);
template <>
__device__ __forceinline__ double recR<double>(double slot) { return slot; }
template <>
__device__ __forceinline__ float recR<float>(double slot) { return __int_as_float(__double2loint(slot)); }

}  // namespace
}  // namespace sipnet
