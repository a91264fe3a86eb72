// Device-side building blocks: span search, Cox-de Boor recursion, window contraction.
#pragma once
#include "bsk_common.hpp"

namespace bsk {

// Span search.  The reference does searchsorted(knots, u, 'right') clamped to
// [order, nCoef] (bspy/_spline_evaluation.py:6-8).  Searching the clamped range
// directly gives the same index: the first ix in [order, nCoef) with knots[ix] > u,
// else nCoef.  `steps` = ceil(log2(nCoef - order + 1)) is the same for every lane, so
// the loop is divergence free; finished lanes idle on the (lo < hi) guard.
// NaN sorts to the end in NumPy, hence ix = nCoef.
template <typename T, typename KP>
__device__ __forceinline__ int find_span(KP knots, int order, int ncoef, int steps, T u)
{
    int lo = order, hi = ncoef;
    for (int s = 0; s < steps; ++s) {
        const int mid = (lo + hi) >> 1;
        const T km = knots[mid];
        const bool open = lo < hi;
        const bool right = open && (km <= u);
        const bool left = open && !right;
        lo = right ? mid + 1 : lo;
        hi = left ? mid : hi;
    }
    return (u != u) ? ncoef : lo;
}

// Cox-de Boor recursion for a compile-time order O (reference
// bspy/_spline_evaluation.py:11-26).  `tab` points at the axis table of this variable
// (knots, then reciprocal rows; see Desc).  b[k] multiplies coefficient ix - O + k.
// `wrt` is wave-uniform (a kernel argument), so the value/derivative level choice is a
// scalar branch.
template <typename T, int O, typename KP>
__device__ __forceinline__ void basis_fixed(KP tab, int nk, int ix, T u, int wrt, T (&b)[O])
{
#pragma unroll
    for (int k = 0; k < O; ++k) b[k] = T(0);
    if (wrt >= O) return;                                  // :9-10
    b[O - 1] = T(1);                                       // :11
#pragma unroll
    for (int degree = 1; degree < O; ++degree) {
        if (degree < O - wrt) {                            // value levels, :12-18
#pragma unroll
            for (int j = 0; j < degree; ++j) {
                const int i = ix - degree + j;
                const int bi = O - degree + j;
                const T alpha = (u - tab[i]) * tab[degree * nk + i];
                b[bi - 1] += (T(1) - alpha) * b[bi];
                b[bi] *= alpha;
            }
        } else {                                           // derivative levels, :19-26
#pragma unroll
            for (int j = 0; j < degree; ++j) {
                const int i = ix - degree + j;
                const int bi = O - degree + j;
                const T alpha = T(degree) * tab[degree * nk + i];
                b[bi - 1] -= alpha * b[bi];
                b[bi] *= alpha;
            }
        }
    }
}

// Value basis and first-derivative basis from ONE recursion: both share the order O-1
// basis and differ only in the last level (the reference's jacobian recomputes the
// whole recursion nInd times, bspy/_spline_evaluation.py:205-213).
template <typename T, int O, typename KP>
__device__ __forceinline__ void basis_value_and_d1(KP tab, int nk, int ix, T u, T (&b)[O], T (&db)[O])
{
#pragma unroll
    for (int k = 0; k < O; ++k) { b[k] = T(0); db[k] = T(0); }
    b[O - 1] = T(1);
    if (O == 1) return;
#pragma unroll
    for (int degree = 1; degree < O - 1; ++degree) {
#pragma unroll
        for (int j = 0; j < degree; ++j) {
            const int i = ix - degree + j;
            const int bi = O - degree + j;
            const T alpha = (u - tab[i]) * tab[degree * nk + i];
            b[bi - 1] += (T(1) - alpha) * b[bi];
            b[bi] *= alpha;
        }
    }
#pragma unroll
    for (int k = 0; k < O; ++k) db[k] = b[k];
    constexpr int degree = O - 1;
#pragma unroll
    for (int j = 0; j < degree; ++j) {
        const int i = ix - degree + j;
        const int bi = O - degree + j;
        const T r = tab[degree * nk + i];
        const T alpha = (u - tab[i]) * r;
        const T dalpha = T(degree) * r;
        b[bi - 1] += (T(1) - alpha) * b[bi];
        b[bi] *= alpha;
        db[bi - 1] -= dalpha * db[bi];
        db[bi] *= dalpha;
    }
}

// Run-time order <= OMAX with statically indexed registers (mixed-order splines, eval_mixed): the
// recursion of basis_fixed on an array of OMAX entries, RIGHT aligned - r[OMAX - order + k]
// multiplies coefficient ix - order + k, i.e. r[m] multiplies coefficient ix - OMAX + m, and
// r[m] = 0 for m < OMAX - order.  Levels beyond the variable's own order are skipped by a
// wave-uniform test (order and wrt are kernel arguments).
template <typename T, int OMAX, typename KP>
__device__ __forceinline__ void basis_bounded(KP tab, int nk, int order, int ix, T u, int wrt, T (&r)[OMAX])
{
#pragma unroll
    for (int k = 0; k < OMAX; ++k) r[k] = T(0);
    if (wrt >= order) return;
    r[OMAX - 1] = T(1);
#pragma unroll
    for (int degree = 1; degree < OMAX; ++degree) {
        if (degree < order) {
            if (degree < order - wrt) {
#pragma unroll
                for (int j = 0; j < degree; ++j) {
                    const int i = ix - degree + j;
                    const int bi = OMAX - degree + j;
                    const T alpha = (u - tab[i]) * tab[degree * nk + i];
                    r[bi - 1] += (T(1) - alpha) * r[bi];
                    r[bi] *= alpha;
                }
            } else {
#pragma unroll
                for (int j = 0; j < degree; ++j) {
                    const int i = ix - degree + j;
                    const int bi = OMAX - degree + j;
                    const T alpha = T(degree) * tab[degree * nk + i];
                    r[bi - 1] -= alpha * r[bi];
                    r[bi] *= alpha;
                }
            }
        }
    }
}

// Same recursion for a run-time order (generic fallback, batched bspline_values).
// b lives in private memory: this path trades speed for generality.
template <typename T>
__device__ inline void basis_runtime(const T *tab, int nk, int O, int ix, T u, int wrt, bool taylor, T *b)
{
    for (int k = 0; k < O; ++k) b[k] = T(0);
    if (wrt >= O) return;
    b[O - 1] = T(1);
    for (int degree = 1; degree < O - wrt; ++degree) {
        int bi = O - degree;
        for (int i = ix - degree; i < ix; ++i, ++bi) {
            const T alpha = (u - tab[i]) * tab[degree * nk + i];
            b[bi - 1] += (T(1) - alpha) * b[bi];
            b[bi] *= alpha;
        }
    }
    for (int degree = (O - wrt > 1 ? O - wrt : 1); degree < O; ++degree) {
        int bi = O - degree;
        // reference :21 forms degree / (order - degree) in double before meeting the knots' dtype
        const T adj = taylor ? T(double(degree) / double(O - degree)) : T(degree);
        for (int i = ix - degree; i < ix; ++i, ++bi) {
            const T alpha = adj * tab[degree * nk + i];
            b[bi - 1] -= alpha * b[bi];
            b[bi] *= alpha;
        }
    }
}

// Window contraction, last variable first, exactly the reference's
// `for iv in range(nInd-1, -1, -1): myCoefs = myCoefs @ bValues[iv]`
// (bspy/_spline_evaluation.py:162-163).  c points at the window's first coefficient
// of one dependent variable.
template <typename T, int O, typename CP>
__device__ __forceinline__ T contract1(CP c, const T (&b0)[O])
{
    T acc = T(0);
#pragma unroll
    for (int a = 0; a < O; ++a) acc += c[a] * b0[a];
    return acc;
}

template <typename T, int O, typename CP>
__device__ __forceinline__ T contract2(CP c, int s0, const T (&b0)[O], const T (&b1)[O])
{
    T acc = T(0);
#pragma unroll
    for (int a = 0; a < O; ++a) {
        T t = T(0);
#pragma unroll
        for (int k = 0; k < O; ++k) t += c[a * s0 + k] * b1[k];
        acc += t * b0[a];
    }
    return acc;
}

template <typename T, int O, typename CP>
__device__ __forceinline__ T contract3(CP c, int s0, int s1, const T (&b0)[O], const T (&b1)[O], const T (&b2)[O])
{
    T acc = T(0);
#pragma unroll
    for (int a = 0; a < O; ++a) {
        T ta = T(0);
#pragma unroll
        for (int k = 0; k < O; ++k) {
            T t = T(0);
#pragma unroll
            for (int m = 0; m < O; ++m) t += c[a * s0 + k * s1 + m] * b2[m];
            ta += t * b1[k];
        }
        acc += ta * b0[a];
    }
    return acc;
}

// Results are written once and never read back by the kernels: non-temporal stores keep them
// from displacing the tables in L2 (measured 6 % on the cfg2 kernel).
template <typename T>
__device__ __forceinline__ void nt_store(T *p, T v)
{
    __builtin_nontemporal_store(v, p);
}

// Record the smallest out-of-domain point index (rare path).
__device__ __forceinline__ void record_bad(unsigned long long *bad, long long n)
{
    atomicMin(bad, (unsigned long long)n);
}

}  // namespace bsk
