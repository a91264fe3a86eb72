// Shared host/device declarations of libbspy_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BSK_INTERNAL 1      // the library defines its measurement hooks too
#include "../../include/bspy_amd.h"

namespace bsk {

constexpr int MAXI = BSK_MAX_NIND;
constexpr int MAXO = BSK_MAX_ORDER;
constexpr int WAVE = 64;

// Per-spline descriptor passed by value to every kernel.
//
// Axis table ("tab", one buffer for all axes, element type T):
//   tab[off[iv] + i]               knots of variable iv, i in [0, nk)
//   tab[off[iv] + d * nk + i]      1 / (knots[i + d] - knots[i]) for d in [1, order)
//                                  (0 where the difference is not positive)
// The reciprocal rows turn the de Boor divisions (reference
// bspy/_spline_evaluation.py:15 and :23) into multiplications.
//
// Coefficients: C-contiguous (nDep, nCoef[0], ..., nCoef[nInd-1]); cstride[0] is the
// stride of the dependent variable, cstride[iv + 1] the stride of variable iv.
template <typename T>
struct Desc {
    int nInd, nDep;
    int order[MAXI], ncoef[MAXI], nk[MAXI], off[MAXI], steps[MAXI];
    int cstride[MAXI + 1];
    T lo[MAXI], hi[MAXI];
    int tab_len;
    int coef_len;
};

template <typename T>
struct Params {
    const T *p[MAXI];
};

struct Wrt {
    int w[MAXI];
};

// "No offender yet" value of the out-of-domain record (unsigned min is taken).
constexpr unsigned long long NO_BAD = ~0ull;

}  // namespace bsk
