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

/* Determinant of the m x m matrix a (row major, m <= 3); m == 0 gives 1 (np.linalg.det of a
 * 0 x 0 matrix). */
static REAL FN(orc_det)(const REAL *a, int m)
{
    if (m == 0) return (REAL)1;
    if (m == 1) return a[0];
    if (m == 2) return a[0] * a[3] - a[1] * a[2];
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

/* Batched normal (bspy/_spline_evaluation.py:215-246): |nInd - nDep| == 1, max(nInd, nDep) <= 4.
 * out[i * N + n], i < max(nInd, nDep).  negate = metadata["negateNormal"]. */
long FN(orc_normal)(int nInd, int nDep, const int *order, const int *nCoef, const REAL *const *knots,
                    const REAL *coefs, const REAL *const *uvw, long N, int normalize, int negate, REAL *out)
{
    const int big = nInd > nDep ? nInd : nDep, small = big - 1;
    REAL *jac = (REAL *)malloc(sizeof(REAL) * (size_t)nDep * nInd);
    const REAL *one_pt[ORC_MAX_NIND];
    REAL p[ORC_MAX_NIND];
    long bad = -1;
    for (long n = 0; n < N; ++n) {
        for (int iv = 0; iv < nInd; ++iv) { p[iv] = uvw[iv][n]; one_pt[iv] = &p[iv]; }
        if (FN(orc_jacobian)(nInd, nDep, order, nCoef, knots, coefs, one_pt, 1, jac) >= 0) { bad = n; break; }   /* :220 */
        /* tangent space with the larger dimension first (:223-228): T[r][c], r < big, c < small */
        REAL T[4][3];
        for (int r = 0; r < big; ++r)
            for (int c = 0; c < small; ++c)
                T[r][c] = nInd > nDep ? jac[c * nInd + r] : jac[r * nInd + c];     /* jac is (nDep, nInd) */
        REAL nrm[4];
        REAL sumsq = (REAL)0;
        for (int i = 0; i < big; ++i) {                                            /* :239-240 */
            REAL sub[9];
            int q = 0;
            for (int r = 0; r < big; ++r) {
                if (r == i) continue;
                for (int c = 0; c < small; ++c) sub[q++] = T[r][c];
            }
            REAL v = FN(orc_det)(sub, small);
            if (i & 1) v = -v;
            if (negate) v = -v;
            nrm[i] = v;
            sumsq += v * v;
        }
        if (normalize) {                                                           /* :243-244 */
            const double len = sqrt((double)sumsq);
            for (int i = 0; i < big; ++i) nrm[i] = (REAL)((double)nrm[i] / len);
        }
        for (int i = 0; i < big; ++i) out[(long)i * N + n] = nrm[i];
    }
    free(jac);
    return bad;
}

/* Batched curvature (bspy/_spline_evaluation.py:80-107): curves (nInd 1, nDep >= 2: signed in
 * 2-D, unsigned otherwise) and surfaces in 3-D (nInd 2, nDep 3: Gaussian curvature).  out[n]. */
long FN(orc_curvature)(int nInd, int nDep, const int *order, const int *nCoef, const REAL *const *knots,
                       const REAL *coefs, const REAL *const *uvw, long N, REAL *out)
{
    long win = nDep;
    for (int iv = 0; iv < nInd; ++iv) win *= order[iv];
    REAL *work = (REAL *)malloc(sizeof(REAL) * (size_t)(win > 0 ? win : 1));
    REAL p[ORC_MAX_NIND];
    long bad = -1;
    for (long n = 0; n < N; ++n) {
        for (int iv = 0; iv < nInd; ++iv) {
            p[iv] = uvw[iv][n];
            if (p[iv] < knots[iv][order[iv] - 1] || p[iv] > knots[iv][nCoef[iv]]) bad = n;
        }
        if (bad >= 0) break;
        if (nInd == 1) {                                                            /* :83-93 */
            REAL fp[ORC_MAX_NDEP], fpp[ORC_MAX_NDEP];
            int w1[1] = {1}, w2[1] = {2};
            FN(orc_point)(1, nDep, order, nCoef, knots, coefs, w1, p, work, fp);
            FN(orc_point)(1, nDep, order, nCoef, knots, coefs, w2, p, work, fpp);
            REAL fpfp = 0, fpfpp = 0, fppfpp = 0;
            for (int d = 0; d < nDep; ++d) { fpfp += fp[d] * fp[d]; fpfpp += fp[d] * fpp[d]; fppfpp += fpp[d] * fpp[d]; }
            const REAL denom = (REAL)pow((double)fpfp, 1.5);
            REAL num;
            if (nDep == 2) num = fp[0] * fpp[1] - fp[1] * fpp[0];
            else num = (REAL)sqrt((double)(fppfpp * fpfp - fpfpp * fpfpp));
            out[n] = num / denom;
        } else {                                                                    /* :94-107 */
            REAL su[3], sv[3], suu[3], suv[3], svv[3], nrm[3];
            int w10[2] = {1, 0}, w01[2] = {0, 1}, w20[2] = {2, 0}, w11[2] = {1, 1}, w02[2] = {0, 2};
            FN(orc_point)(2, 3, order, nCoef, knots, coefs, w10, p, work, su);
            FN(orc_point)(2, 3, order, nCoef, knots, coefs, w01, p, work, sv);
            FN(orc_point)(2, 3, order, nCoef, knots, coefs, w20, p, work, suu);
            FN(orc_point)(2, 3, order, nCoef, knots, coefs, w11, p, work, suv);
            FN(orc_point)(2, 3, order, nCoef, knots, coefs, w02, p, work, svv);
            const REAL *pt[2] = {&p[0], &p[1]};
            FN(orc_normal)(2, 3, order, nCoef, knots, coefs, pt, 1, 1, 0, nrm);
            REAL E = 0, F = 0, G = 0, L = 0, M = 0, Nn = 0;
            for (int d = 0; d < 3; ++d) {
                E += su[d] * su[d]; F += su[d] * sv[d]; G += sv[d] * sv[d];
                L += suu[d] * nrm[d]; M += suv[d] * nrm[d]; Nn += svv[d] * nrm[d];
            }
            out[n] = (L * Nn - M * M) / (E * G - F * F);
        }
    }
    free(work);
    return bad;
}

#undef FN
#undef CAT
#undef CAT_
