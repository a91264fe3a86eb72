/*
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE.
 *
 * Type-generic body of the CPU oracle; included twice by bspline_oracle.c with
 * REAL = double / float and SUF = f64 / f32.  Each function restates, in plain
 * scalar C and in the reference's own operation order, one function of the
 * reference's evaluation path (all citations relative to /root/reference):
 *
 *   orc_bspline_values_*   bspy/_spline_evaluation.py:4-27
 *   orc_evaluate_*         bspy/_spline_evaluation.py:109-133 (derivative) and
 *                          :140-164 (evaluate; wrt == all zeros), batched the way
 *                          bspy/spline.py:757-770 / :936-949 loop over points
 *   orc_jacobian_*         bspy/_spline_evaluation.py:205-213
 *
 * Compiled with -ffp-contract=off so no FMA is formed: the reference computes
 * with separately rounded NumPy multiplies and adds.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* _spline_evaluation.py:6-8: searchsorted(knots, u, side='right') clamped to
 * [order, len(knots) - order].  NaN sorts to the end, as in NumPy. */
static int FN(orc_span)(const REAL *knots, int nknots, int order, REAL u)
{
    int lo = 0, hi = nknots; /* first index with knots[idx] > u */
    if (u != u) {
        lo = nknots;
    } else {
        while (lo < hi) {
            int mid = (lo + hi) / 2;
            if (knots[mid] <= u) lo = mid + 1; else hi = mid;
        }
    }
    if (lo < order) lo = order;
    if (lo > nknots - order) lo = nknots - order;
    return lo;
}

/* _spline_evaluation.py:4-27.  knot < 0 means "search" (the reference's None). */
int FN(orc_bspline_values)(int knot, const REAL *knots, int nknots, int order, REAL u,
                           int deriv, int taylor, REAL *basis)
{
    for (int k = 0; k < order; ++k) basis[k] = (REAL)0;            /* :5 */
    if (knot < 0) knot = FN(orc_span)(knots, nknots, order, u);    /* :6-8 */
    if (deriv >= order) return knot;                               /* :9-10 */
    basis[order - 1] = (REAL)1;                                    /* :11 */
    for (int degree = 1; degree < order - deriv; ++degree) {       /* :12-18 */
        int b = order - degree;
        for (int i = knot - degree; i < knot; ++i) {
            REAL alpha = (u - knots[i]) / (knots[i + degree] - knots[i]);
            basis[b - 1] += ((REAL)1 - alpha) * basis[b];
            basis[b] *= alpha;
            ++b;
        }
    }
    for (int degree = order - deriv; degree < order; ++degree) {   /* :19-26 */
        int b = order - degree;
        /* :21 is evaluated in double by Python before it meets the knots' dtype */
        double adj = taylor ? (double)degree / (double)(order - degree) : (double)degree;
        for (int i = knot - degree; i < knot; ++i) {
            REAL alpha = (REAL)adj / (knots[i + degree] - knots[i]);
            basis[b - 1] += -alpha * basis[b];
            basis[b] *= alpha;
            ++b;
        }
    }
    return knot;
}

/* One point of _spline_evaluation.py:109-133 / :140-164 after the domain check.
 * work must hold nDep * prod(order) REALs. */
static void FN(orc_point)(int nInd, int nDep, const int *order, const int *nCoef,
                          const REAL *const *knots, const REAL *coefs, const int *wrt,
                          const REAL *uvw, REAL *work, REAL *out)
{
    int ix[ORC_MAX_NIND];
    REAL bval[ORC_MAX_NIND][ORC_MAX_ORDER];
    long stride[ORC_MAX_NIND + 1];
    for (int iv = 0; iv < nInd; ++iv)                               /* :124-129 / :157-160 */
        ix[iv] = FN(orc_bspline_values)(-1, knots[iv], order[iv] + nCoef[iv], order[iv], uvw[iv],
                                        wrt ? wrt[iv] : 0, 0, bval[iv]);
    stride[nInd] = 1;
    for (int iv = nInd - 1; iv >= 0; --iv) stride[iv] = stride[iv + 1] * nCoef[iv];
    /* myCoefs = coefs[:, ix0-o0:ix0, ...] (:130 / :161), gathered densely */
    long win = 1;
    for (int iv = 0; iv < nInd; ++iv) win *= order[iv];
    int idx[ORC_MAX_NIND];
    for (int d = 0; d < nDep; ++d) {
        for (int iv = 0; iv < nInd; ++iv) idx[iv] = 0;
        for (long w = 0; w < win; ++w) {
            long off = (long)d * stride[0];
            for (int iv = 0; iv < nInd; ++iv) off += (long)(ix[iv] - order[iv] + idx[iv]) * stride[iv + 1];
            work[d * win + w] = coefs[off];
            for (int iv = nInd - 1; iv >= 0; --iv) { if (++idx[iv] < order[iv]) break; idx[iv] = 0; }
        }
    }
    /* for iv in range(nInd-1, -1, -1): myCoefs = myCoefs @ bValues[iv]  (:131-132 / :162-163) */
    long rows = (long)nDep * win;
    for (int iv = nInd - 1; iv >= 0; --iv) {
        int o = order[iv];
        rows /= o;
        for (long r = 0; r < rows; ++r) {
            REAL acc = (REAL)0;
            for (int k = 0; k < o; ++k) acc += work[r * o + k] * bval[iv][k];
            work[r] = acc;
        }
    }
    for (int d = 0; d < nDep; ++d) out[d] = work[d];
}

/* Batched evaluate/derivative: points in SoA (uvw[iv][n]), output SoA out[d*N + n].
 * Returns -1, or the index of the first point outside the (inclusive) domain
 * (:117-121 / :148-152; NaN passes, as every comparison with it is false). */
long FN(orc_evaluate)(int nInd, int nDep, const int *order, const int *nCoef,
                      const REAL *const *knots, const REAL *coefs, const int *wrt,
                      const REAL *const *uvw, long N, REAL *out)
{
    long win = nDep;
    for (int iv = 0; iv < nInd; ++iv) win *= order[iv];
    REAL *work = (REAL *)malloc(sizeof(REAL) * (size_t)(win > 0 ? win : 1));
    REAL res[ORC_MAX_NDEP];
    REAL p[ORC_MAX_NIND];
    long bad = -1;
    for (long n = 0; n < N && bad < 0; ++n) {
        for (int iv = 0; iv < nInd; ++iv) {
            p[iv] = uvw[iv][n];
            REAL lo = knots[iv][order[iv] - 1], hi = knots[iv][nCoef[iv]];   /* :135-138 */
            if (p[iv] < lo || p[iv] > hi) bad = n;
        }
        if (bad >= 0) break;
        FN(orc_point)(nInd, nDep, order, nCoef, knots, coefs, wrt, p, work, res);
        for (int d = 0; d < nDep; ++d) out[(long)d * N + n] = res[d];
    }
    free(work);
    return bad;
}

/* Batched jacobian: out[(d*nInd + j)*N + n]; nInd derivative calls per point (:205-213). */
long FN(orc_jacobian)(int nInd, int nDep, const int *order, const int *nCoef,
                      const REAL *const *knots, const REAL *coefs,
                      const REAL *const *uvw, long N, REAL *out)
{
    long win = nDep;
    for (int iv = 0; iv < nInd; ++iv) win *= order[iv];
    REAL *work = (REAL *)malloc(sizeof(REAL) * (size_t)(win > 0 ? win : 1));
    REAL res[ORC_MAX_NDEP];
    REAL p[ORC_MAX_NIND];
    int wrt[ORC_MAX_NIND];
    long bad = -1;
    for (long n = 0; n < N; ++n) {
        for (int iv = 0; iv < nInd; ++iv) {
            p[iv] = uvw[iv][n];
            REAL lo = knots[iv][order[iv] - 1], hi = knots[iv][nCoef[iv]];
            if (p[iv] < lo || p[iv] > hi) bad = n;
        }
        if (bad >= 0) break;
        for (int j = 0; j < nInd; ++j) {
            for (int iv = 0; iv < nInd; ++iv) wrt[iv] = (iv == j);
            FN(orc_point)(nInd, nDep, order, nCoef, knots, coefs, wrt, p, work, res);
            for (int d = 0; d < nDep; ++d) out[((long)d * nInd + j) * N + n] = res[d];
        }
    }
    free(work);
    return bad;
}

#undef FN
#undef CAT
#undef CAT_
