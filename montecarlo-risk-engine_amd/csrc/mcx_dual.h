// mcx_dual.h — forward-mode dual numbers (value + P tangents) for the tangent kernels (kt_tangent.hip, kt_book.hip).
// Subgradient conventions follow torch exactly (they decide the reference's numbers):
//   torch.clamp(x, min, max): gradient passes iff min <= x <= max;  torch.maximum(a, b): a > b -> a, tie -> 1/2 each;
//   (x > 0).float(): zero gradient;  degree_of_truth (fuzzy): clamp((x+eps)/(2eps), 0, 1).
#pragma once
#include "mcx_device.h"

template <int P>
struct Dual {
    double v;
    double d[P];
};

template <int P> __device__ __forceinline__ Dual<P> dconst(double c) { Dual<P> r; r.v = c; for (int j = 0; j < P; ++j) r.d[j] = 0.0; return r; }
template <int P> __device__ __forceinline__ Dual<P> dseed(double c, int k) { Dual<P> r = dconst<P>(c); r.d[k] = 1.0; return r; }
template <int P> __device__ __forceinline__ Dual<P> operator+(const Dual<P>& a, const Dual<P>& b) { Dual<P> r; r.v = a.v + b.v; for (int j = 0; j < P; ++j) r.d[j] = a.d[j] + b.d[j]; return r; }
template <int P> __device__ __forceinline__ Dual<P> operator-(const Dual<P>& a, const Dual<P>& b) { Dual<P> r; r.v = a.v - b.v; for (int j = 0; j < P; ++j) r.d[j] = a.d[j] - b.d[j]; return r; }
template <int P> __device__ __forceinline__ Dual<P> operator*(const Dual<P>& a, const Dual<P>& b) { Dual<P> r; r.v = a.v * b.v; for (int j = 0; j < P; ++j) r.d[j] = a.d[j] * b.v + a.v * b.d[j]; return r; }
template <int P> __device__ __forceinline__ Dual<P> operator/(const Dual<P>& a, const Dual<P>& b) { Dual<P> r; const double ib = mcx_rcp(b.v); r.v = a.v * ib; for (int j = 0; j < P; ++j) r.d[j] = (a.d[j] - r.v * b.d[j]) * ib; return r; }
template <int P> __device__ __forceinline__ Dual<P> operator+(const Dual<P>& a, double c) { Dual<P> r = a; r.v += c; return r; }
template <int P> __device__ __forceinline__ Dual<P> operator-(const Dual<P>& a, double c) { Dual<P> r = a; r.v -= c; return r; }
template <int P> __device__ __forceinline__ Dual<P> operator*(const Dual<P>& a, double c) { Dual<P> r; r.v = a.v * c; for (int j = 0; j < P; ++j) r.d[j] = a.d[j] * c; return r; }
template <int P> __device__ __forceinline__ Dual<P> operator*(double c, const Dual<P>& a) { return a * c; }
template <int P> __device__ __forceinline__ Dual<P> operator+(double c, const Dual<P>& a) { return a + c; }
template <int P> __device__ __forceinline__ Dual<P> operator-(double c, const Dual<P>& a) { Dual<P> r; r.v = c - a.v; for (int j = 0; j < P; ++j) r.d[j] = -a.d[j]; return r; }
template <int P> __device__ __forceinline__ Dual<P> operator/(double c, const Dual<P>& a) { return dconst<P>(c) / a; }
template <int P> __device__ __forceinline__ Dual<P> dexp(const Dual<P>& a) { Dual<P> r; r.v = mcx_exp(a.v); for (int j = 0; j < P; ++j) r.d[j] = r.v * a.d[j]; return r; }
template <int P> __device__ __forceinline__ Dual<P> dlog(const Dual<P>& a) { Dual<P> r; r.v = mcx_log(a.v); const double ia = mcx_rcp(a.v); for (int j = 0; j < P; ++j) r.d[j] = a.d[j] * ia; return r; }
// sqrt: a zero tangent stays zero even where 1/(2 sqrt(x)) is infinite (x clamped to 0).  Reverse mode gets the same result
// because torch.clamp's backward is a `where(mask, grad, 0)` that discards the inf/NaN produced by sqrt's backward.
template <int P> __device__ __forceinline__ Dual<P> dsqrt(const Dual<P>& a) { Dual<P> r; double h; r.v = mcx_sqrt_h(a.v, h); for (int j = 0; j < P; ++j) r.d[j] = a.d[j] == 0.0 ? 0.0 : a.d[j] * h; return r; }
// a / b with ib = 1 / b.v already known (reciprocals of several denominators from one v_rcp_f64 of their product)
template <int P> __device__ __forceinline__ Dual<P> ddiv_r(const Dual<P>& a, const Dual<P>& b, double ib) { Dual<P> r; r.v = a.v * ib; for (int j = 0; j < P; ++j) r.d[j] = (a.d[j] - r.v * b.d[j]) * ib; return r; }
template <int P> __device__ __forceinline__ Dual<P> drcp_r(const Dual<P>& b, double ib) { Dual<P> r; r.v = ib; const double m = -ib * ib; for (int j = 0; j < P; ++j) r.d[j] = b.d[j] * m; return r; }
// torch.clamp(x, min=lo): gradient mask x >= lo
template <int P> __device__ __forceinline__ Dual<P> dclamp_min(const Dual<P>& a, double lo) { Dual<P> r; const bool pass = a.v >= lo; r.v = a.v < lo ? lo : a.v;   /* (a NaN value stays NaN: torch.clamp) */ for (int j = 0; j < P; ++j) r.d[j] = pass ? a.d[j] : 0.0; return r; }
template <int P> __device__ __forceinline__ Dual<P> dclamp(const Dual<P>& a, double lo, double hi) { Dual<P> r; const bool pass = a.v >= lo && a.v <= hi; r.v = fmin(fmax(a.v, lo), hi); for (int j = 0; j < P; ++j) r.d[j] = pass ? a.d[j] : 0.0; return r; }
template <int P> __device__ __forceinline__ Dual<P> ddegree(const Dual<P>& x, bool fuzzy, double eps)
{
    if (!fuzzy) return dconst<P>(x.v > 0.0 ? 1.0 : 0.0);
    return dclamp((x + eps) * (1.0 / (2.0 * eps)), 0.0, 1.0);
}

