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

// 2^x, |rel err| <= 1.8e-16: n = rint(x), 2^(x-n) by a degree-11 polynomial on
// [-0.5, 0.5] (Chebyshev-node interpolant, tools/fit_exp2.py), scaled with v_ldexp_f64.
// The coefficients live in SGPR pairs for the whole time loop (a VOP3 fma takes one scalar
// operand): left as literals, hipcc re-materialises them with v_mov_b64 at each of the 13
// call sites of a step, which costs as much as the polynomial itself.
struct Exp2Coef {
  double c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11;
};
__device__ __forceinline__ Exp2Coef loadExp2Coef() {
  Exp2Coef k = {0.6931471805599453,     0.2402265069591016,     0.05550410866482163,
                0.009618129107587223,   0.0013333558146405434,  0.0001540353046375614,
                1.5252733842758916e-05, 1.3215432520547035e-06, 1.0178056472371986e-07,
                7.074197066047615e-09,  4.455930741563682e-10};
  // opaque to constant propagation, pinned to scalar registers
  asm volatile("" : "+s"(k.c1), "+s"(k.c2), "+s"(k.c3), "+s"(k.c4), "+s"(k.c5), "+s"(k.c6));
  asm volatile("" : "+s"(k.c7), "+s"(k.c8), "+s"(k.c9), "+s"(k.c10), "+s"(k.c11));
  return k;
}
__device__ __forceinline__ double fexp2(double x, const Exp2Coef& k) {
  const double n = __builtin_rint(x);
  const double f = x - n;
  double p = k.c11;
  p = ffma(p, f, k.c10);
  p = ffma(p, f, k.c9);
  p = ffma(p, f, k.c8);
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

// a / b with b > 0 finite and well scaled: v_rcp + two Newton steps + one residual step
__device__ __forceinline__ double fdiv(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = ffma(ffma(-b, r, 1.0), r, r);
  r = ffma(ffma(-b, r, 1.0), r, r);
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

}  // namespace
}  // namespace sipnet
