// hipadj_dual.hpp — forward-mode dual numbers for runtime-registered models WITHOUT hand-written VJPs.
//
// The reference builds (df/du)^T lam and (df/dp)^T lam by AD when the user gives only `f`
// (`autojacvec = true`: "the Jacobian is constructed via ForwardDiff.jl", src/sensitivity_algorithms.jl:629-660;
// `_vecjacobian!` for the ForwardDiff path, src/derivative_wrappers.jl:256-480).  The device analogue: the registered body
// of `f` is compiled a second time with `real = Dual<K>` (K seeded directions: the n states, or the np parameters), one
// evaluation yields all partial derivatives, and the VJP is the lam-weighted column sum.  Everything stays in registers and
// is inlined by hiprtc like the hand-written models.
#pragma once

#include "hipadj_models.hpp"

namespace hipadj {

template <int K> struct Dual {
    double v;
    double d[K];
    HIPADJ_HD Dual() : v(0.0) {
#pragma unroll
        for (int i = 0; i < K; ++i) d[i] = 0.0; }
    HIPADJ_HD Dual(double x) : v(x) {
#pragma unroll
        for (int i = 0; i < K; ++i) d[i] = 0.0; }
    HIPADJ_HD Dual(int x) : v((double)x) {
#pragma unroll
        for (int i = 0; i < K; ++i) d[i] = 0.0; }
    HIPADJ_HD static Dual seed(double x, int k) { Dual r(x); r.d[k] = 1.0; return r; }
};

// y = g(x.v) with derivative gp: chain rule helper
template <int K> HIPADJ_HD Dual<K> dual_chain(const Dual<K>& x, double g, double gp) {
    Dual<K> r; r.v = g;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = gp * x.d[i];
    return r;
}

template <int K> HIPADJ_HD Dual<K> operator+(const Dual<K>& a, const Dual<K>& b) { Dual<K> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] + b.d[i];
    return r; }
template <int K> HIPADJ_HD Dual<K> operator-(const Dual<K>& a, const Dual<K>& b) { Dual<K> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] - b.d[i];
    return r; }
template <int K> HIPADJ_HD Dual<K> operator*(const Dual<K>& a, const Dual<K>& b) { Dual<K> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r; }
template <int K> HIPADJ_HD Dual<K> operator/(const Dual<K>& a, const Dual<K>& b) { Dual<K> r; const double ib = 1.0 / b.v; r.v = a.v * ib;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r; }
template <int K> HIPADJ_HD Dual<K> operator-(const Dual<K>& a) { Dual<K> r; r.v = -a.v;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = -a.d[i];
    return r; }
template <int K> HIPADJ_HD Dual<K> operator+(const Dual<K>& a) { return a; }

// mixed with plain numbers (literals in the model text are doubles / ints)
template <int K> HIPADJ_HD Dual<K> operator+(const Dual<K>& a, double b) { Dual<K> r = a; r.v += b; return r; }
template <int K> HIPADJ_HD Dual<K> operator+(double a, const Dual<K>& b) { Dual<K> r = b; r.v += a; return r; }
template <int K> HIPADJ_HD Dual<K> operator-(const Dual<K>& a, double b) { Dual<K> r = a; r.v -= b; return r; }
template <int K> HIPADJ_HD Dual<K> operator-(double a, const Dual<K>& b) { Dual<K> r = -b; r.v += a; return r; }
template <int K> HIPADJ_HD Dual<K> operator*(const Dual<K>& a, double b) { Dual<K> r; r.v = a.v * b;
#pragma unroll
    for (int i = 0; i < K; ++i) r.d[i] = a.d[i] * b;
    return r; }
template <int K> HIPADJ_HD Dual<K> operator*(double a, const Dual<K>& b) { return b * a; }
template <int K> HIPADJ_HD Dual<K> operator/(const Dual<K>& a, double b) { return a * (1.0 / b); }
template <int K> HIPADJ_HD Dual<K> operator/(double a, const Dual<K>& b) { return Dual<K>(a) / b; }
template <int K> HIPADJ_HD Dual<K>& operator+=(Dual<K>& a, const Dual<K>& b) { a = a + b; return a; }
template <int K> HIPADJ_HD Dual<K>& operator-=(Dual<K>& a, const Dual<K>& b) { a = a - b; return a; }
template <int K> HIPADJ_HD Dual<K>& operator*=(Dual<K>& a, const Dual<K>& b) { a = a * b; return a; }
template <int K> HIPADJ_HD Dual<K>& operator/=(Dual<K>& a, const Dual<K>& b) { a = a / b; return a; }
template <int K> HIPADJ_HD Dual<K>& operator+=(Dual<K>& a, double b) { a.v += b; return a; }
template <int K> HIPADJ_HD Dual<K>& operator-=(Dual<K>& a, double b) { a.v -= b; return a; }
template <int K> HIPADJ_HD Dual<K>& operator*=(Dual<K>& a, double b) { a = a * b; return a; }
template <int K> HIPADJ_HD Dual<K>& operator/=(Dual<K>& a, double b) { a = a / b; return a; }

// comparisons act on the value (piecewise models differentiate the active branch, as ForwardDiff does)
template <int K> HIPADJ_HD bool operator<(const Dual<K>& a, const Dual<K>& b) { return a.v < b.v; }
template <int K> HIPADJ_HD bool operator>(const Dual<K>& a, const Dual<K>& b) { return a.v > b.v; }
template <int K> HIPADJ_HD bool operator<=(const Dual<K>& a, const Dual<K>& b) { return a.v <= b.v; }
template <int K> HIPADJ_HD bool operator>=(const Dual<K>& a, const Dual<K>& b) { return a.v >= b.v; }
template <int K> HIPADJ_HD bool operator<(const Dual<K>& a, double b) { return a.v < b; }
template <int K> HIPADJ_HD bool operator>(const Dual<K>& a, double b) { return a.v > b; }
template <int K> HIPADJ_HD bool operator<=(const Dual<K>& a, double b) { return a.v <= b; }
template <int K> HIPADJ_HD bool operator>=(const Dual<K>& a, double b) { return a.v >= b; }

// elementary functions (the plain-double ones stay visible next to the overloads: the model text is compiled in this namespace)
using ::sin; using ::cos; using ::tan; using ::exp; using ::log; using ::sqrt; using ::tanh; using ::sinh; using ::cosh; using ::atan; using ::fabs; using ::pow;
template <int K> HIPADJ_HD Dual<K> sin(const Dual<K>& x) { return dual_chain(x, ::sin(x.v), ::cos(x.v)); }
template <int K> HIPADJ_HD Dual<K> cos(const Dual<K>& x) { return dual_chain(x, ::cos(x.v), -::sin(x.v)); }
template <int K> HIPADJ_HD Dual<K> tan(const Dual<K>& x) { const double t = ::tan(x.v); return dual_chain(x, t, 1.0 + t * t); }
template <int K> HIPADJ_HD Dual<K> exp(const Dual<K>& x) { const double e = ::exp(x.v); return dual_chain(x, e, e); }
template <int K> HIPADJ_HD Dual<K> log(const Dual<K>& x) { return dual_chain(x, ::log(x.v), 1.0 / x.v); }
template <int K> HIPADJ_HD Dual<K> sqrt(const Dual<K>& x) { const double s = ::sqrt(x.v); return dual_chain(x, s, 0.5 / s); }
template <int K> HIPADJ_HD Dual<K> tanh(const Dual<K>& x) { const double t = ::tanh(x.v); return dual_chain(x, t, 1.0 - t * t); }
template <int K> HIPADJ_HD Dual<K> sinh(const Dual<K>& x) { return dual_chain(x, ::sinh(x.v), ::cosh(x.v)); }
template <int K> HIPADJ_HD Dual<K> cosh(const Dual<K>& x) { return dual_chain(x, ::cosh(x.v), ::sinh(x.v)); }
template <int K> HIPADJ_HD Dual<K> atan(const Dual<K>& x) { return dual_chain(x, ::atan(x.v), 1.0 / (1.0 + x.v * x.v)); }
template <int K> HIPADJ_HD Dual<K> fabs(const Dual<K>& x) { return dual_chain(x, ::fabs(x.v), x.v < 0.0 ? -1.0 : 1.0); }
// value and derivative are formed separately: pow(x, e-1) * x is inf * 0 = NaN at x == 0 for e < 1 where pow(0, e) is finite;
// e == 0 has derivative 0 everywhere (also at x == 0, where e * pow(0, -1) would be 0 * inf)
template <int K> HIPADJ_HD Dual<K> pow(const Dual<K>& x, double e) { return dual_chain(x, ::pow(x.v, e), e == 0.0 ? 0.0 : e * ::pow(x.v, e - 1.0)); }
template <int K> HIPADJ_HD Dual<K> pow(const Dual<K>& x, int e) { return pow(x, (double)e); }
template <int K> HIPADJ_HD Dual<K> pow(const Dual<K>& x, const Dual<K>& e) { return exp(e * log(x)); }

// `real`-generic spellings for T = double, so that one model text serves both instantiations
HIPADJ_HD double dual_value(double x) { return x; }
template <int K> HIPADJ_HD double dual_value(const Dual<K>& x) { return x.v; }

}  // namespace hipadj
