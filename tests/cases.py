"""
Deterministic input generator shared by the golden-vector script, the oracle
tests and the GPU parity tests.

Nothing in here touches the reference: it only builds *inputs* (spline
definitions and parameter points) from fixed seeds, following the generator
spec of SURVEY.md section 8(c)-2, so the GPU box can regenerate exactly the
inputs whose reference outputs are stored in ``tests/golden/*.npz``.
"""
import itertools
import numpy as np


def clamped_uniform_knots(order, ncoef, dtype=np.float64, lo=0.0, hi=1.0):
    """concat(lo*order, interior linspace, hi*order): the BASELINE.md generator."""
    interior = np.linspace(lo, hi, ncoef - order + 2)[1:-1]
    return np.concatenate((np.full(order, lo), interior, np.full(order, hi))).astype(dtype)


def nonuniform_knots(rng, order, ncoef, dtype=np.float64, lo=-2.0, hi=3.0):
    """Clamped knots on [lo, hi] with sorted random interior knots, some of
    them repeated (multiplicity 2 .. order-1), total interior count ncoef-order."""
    n_int = ncoef - order
    vals = []
    while len(vals) < n_int:
        x = lo + (hi - lo) * (0.02 + 0.96 * rng.random())
        mult = 1
        if order > 2 and rng.random() < 0.25:
            mult = int(rng.integers(2, order))
        mult = min(mult, n_int - len(vals))
        vals.extend([x] * mult)
    interior = np.sort(np.array(vals, dtype=np.float64))
    return np.concatenate((np.full(order, lo), interior, np.full(order, hi))).astype(dtype)


def all_wrt(nind, max_total):
    """Every derivative multi-index with total order <= max_total."""
    return [w for w in itertools.product(range(max_total + 1), repeat=nind) if sum(w) <= max_total]


class Case:
    """One spline + one batch of points + the derivative multi-indices to pin."""

    def __init__(self, name, seed, nind, ndep, order, ncoef, n, kdtype=np.float64, cdtype=np.float64,
                 pdtype=None, knots="uniform", wrts=None, jacobian=True, edge_points=False):
        self.name = name
        self.nInd, self.nDep = nind, ndep
        self.order, self.nCoef = tuple(order), tuple(ncoef)
        rng = np.random.default_rng(seed)
        if knots == "uniform":
            self.knots = [clamped_uniform_knots(o, c, kdtype) for o, c in zip(order, ncoef)]
        else:
            self.knots = [nonuniform_knots(rng, o, c, kdtype) for o, c in zip(order, ncoef)]
        self.coefs = rng.standard_normal((ndep, *ncoef)).astype(cdtype)
        pdtype = pdtype or (np.float32 if (kdtype == np.float32 and cdtype == np.float32) else np.float64)
        pts = []
        for iv in range(nind):
            k = self.knots[iv]
            lo, hi = float(k[order[iv] - 1]), float(k[ncoef[iv]])
            p = (lo + (hi - lo) * rng.random(n)).astype(pdtype)
            if edge_points:
                # parameters exactly on every distinct knot, on both domain ends and
                # one ulp either side of each interior knot (clipped to the domain)
                d = np.unique(k.astype(pdtype))
                d = d[(d >= pdtype(lo)) & (d <= pdtype(hi))]
                e = np.concatenate((d, np.nextafter(d, pdtype(-np.inf)), np.nextafter(d, pdtype(np.inf))))
                e = e[(e >= pdtype(lo)) & (e <= pdtype(hi))].astype(pdtype)
                e = e[rng.permutation(len(e))]
                m = min(len(e), n)
                p[:m] = e[:m]
            p = np.clip(p, pdtype(lo), pdtype(hi))
            pts.append(p)
        self.points = pts
        self.wrts = wrts if wrts is not None else [tuple([0] * nind)]
        self.jacobian = jacobian

    @property
    def n(self):
        return len(self.points[0]) if self.points else 0


def parity_cases():
    """The randomised parity sets plus the edge set (8c-3)."""
    f32, f64 = np.float32, np.float64
    cs = []
    # cfg1: 1-D cubic, nCoef 32
    cs.append(Case("cfg1_curve", 101, 1, 1, (4,), (32,), 2048, wrts=all_wrt(1, 4), edge_points=True))
    # cfg2 / cfg3: bicubic 64x64x3 fp64 (uniform and non-uniform knots)
    cs.append(Case("cfg2_bicubic", 102, 2, 3, (4, 4), (64, 64), 1024, wrts=all_wrt(2, 3) + [(4, 0), (0, 5), (2, 2)]))
    cs.append(Case("cfg2_bicubic_nonuniform", 103, 2, 3, (4, 4), (64, 64), 1024, knots="nonuniform",
                   wrts=all_wrt(2, 3), edge_points=True))
    # cfg5: trivariate order 5, 40^3, nDep 4, fp32 (coefs regenerated from the seed: 1 MB)
    cs.append(Case("cfg5_trivariate_f32", 105, 3, 4, (5, 5, 5), (40, 40, 40), 1024, f32, f32,
                   wrts=[(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1), (2, 0, 1)]))
    cs.append(Case("cfg5_trivariate_f64", 106, 3, 4, (5, 5, 5), (12, 11, 10), 512, knots="nonuniform",
                   wrts=[(0, 0, 0), (1, 0, 0), (0, 2, 1), (4, 0, 0), (5, 0, 0)], edge_points=True))
    # teapot-like patch: single Bezier span, fp32
    cs.append(Case("bezier_patch_f32", 107, 2, 3, (4, 4), (4, 4), 512, f32, f32, wrts=all_wrt(2, 2), edge_points=True))
    # orders 1..9 in one variable, nDep 1..6
    for o in range(1, 10):
        cs.append(Case(f"curve_order{o}", 200 + o, 1, 1 + (o % 6), (o,), (o + 7,), 128, knots="nonuniform",
                       wrts=[(d,) for d in range(0, o + 2)], edge_points=True))
    # mixed orders, 2 and 3 variables, various nDep
    cs.append(Case("surface_o3x4", 301, 2, 3, (3, 4), (4, 5), 512, knots="nonuniform", wrts=all_wrt(2, 3), edge_points=True))
    cs.append(Case("surface_o2x6_d1", 302, 2, 1, (2, 6), (9, 8), 512, knots="nonuniform", wrts=all_wrt(2, 2), edge_points=True))
    cs.append(Case("surface_o7x3_d6", 303, 2, 6, (7, 3), (11, 20), 512, wrts=all_wrt(2, 2)))
    cs.append(Case("surface_o1x4_d2", 304, 2, 2, (1, 4), (5, 6), 512, knots="nonuniform", wrts=all_wrt(2, 1), edge_points=True))
    cs.append(Case("volume_o3x4x2", 305, 3, 3, (3, 4, 2), (6, 7, 5), 512, knots="nonuniform", wrts=all_wrt(3, 2), edge_points=True))
    cs.append(Case("volume_o4_d1", 306, 3, 1, (4, 4, 4), (8, 9, 10), 512, wrts=all_wrt(3, 1)))
    cs.append(Case("four_var", 307, 4, 2, (3, 2, 4, 3), (4, 5, 6, 4), 256, knots="nonuniform",
                   wrts=[(0, 0, 0, 0), (1, 0, 0, 0), (0, 0, 1, 1), (0, 2, 0, 0)]))
    cs.append(Case("five_var", 308, 5, 1, (2, 2, 3, 2, 2), (3, 4, 4, 3, 3), 256,
                   wrts=[(0, 0, 0, 0, 0), (0, 1, 0, 0, 0), (1, 0, 0, 0, 1)]))
    # dtypes: fp32 knots + fp32 coefs; mixed fp32 knots / fp64 coefs and the converse
    cs.append(Case("curve_f32", 401, 1, 2, (4,), (12,), 512, f32, f32, wrts=all_wrt(1, 2), edge_points=True))
    cs.append(Case("surface_f32knots_f64coefs", 402, 2, 3, (4, 3), (9, 8), 512, f32, f64, pdtype=f64, wrts=all_wrt(2, 1)))
    cs.append(Case("surface_f64knots_f32coefs", 403, 2, 3, (4, 3), (9, 8), 512, f64, f32, pdtype=f64, wrts=all_wrt(2, 1)))
    # larger orders / wide tables
    cs.append(Case("curve_order12", 501, 1, 2, (12,), (30,), 512, knots="nonuniform", wrts=[(0,), (1,), (3,)], edge_points=True))
    cs.append(Case("curve_many_knots", 502, 1, 3, (4,), (900,), 1024, knots="nonuniform", wrts=[(0,), (1,), (2,)], edge_points=True))
    cs.append(Case("surface_wide", 503, 2, 3, (4, 5), (300, 11), 1024, knots="nonuniform", wrts=all_wrt(2, 1)))
    return cs


class BlockCase:
    """A block of splines (reference bspy/spline_block.py): rows of (map, spline definition) over
    the block's independent variables, plus a batch of points.  Spline definitions are plain
    tuples (nInd, nDep, order, nCoef, knots, coefs) so both the reference and this package can
    build their own Spline objects from them."""

    def __init__(self, name, seed, domains, rows, n, wrts):
        rng = np.random.default_rng(seed)
        self.name = name
        self.nInd = len(domains)
        self.rows = []
        for row in rows:
            new_row = []
            for (imap, ndep, order, ncoef, cdtype) in row:
                knots = []
                for o, c, iv in zip(order, ncoef, imap):
                    lo, hi = domains[iv]
                    interior = np.sort(lo + (hi - lo) * (0.05 + 0.9 * rng.random(c - o)))
                    knots.append(np.concatenate((np.full(o, lo), interior, np.full(o, hi))).astype(np.float64))
                coefs = rng.standard_normal((ndep, *ncoef)).astype(cdtype)
                new_row.append((list(imap), (len(imap), ndep, tuple(order), tuple(ncoef), knots, coefs)))
            self.rows.append(new_row)
        self.nDep = sum(r[0][1][1] for r in self.rows)
        self.points = [lo + (hi - lo) * rng.random(n) for (lo, hi) in domains]
        self.wrts = wrts


def block_cases():
    """SplineBlock evaluation goldens: default maps, mapped
    variables, single-spline rows and a row summing three splines."""
    f64, f32 = np.float64, np.float32
    cs = []
    # [[F(u,v,w), G(u)], [h(u,v)]] with default maps (consecutive variables per row)
    cs.append(BlockCase("block_default_maps", 901, [(0.0, 1.0), (-1.0, 2.0), (0.5, 1.5), (0.0, 1.0)],
                        [[((0, 1, 2), 2, (4, 3, 2), (6, 5, 4), f64), ((3,), 2, (3,), (5,), f64)],
                         [((0, 1), 1, (4, 3), (6, 5), f64)]], 257,
                        [(0, 0, 0, 0), (1, 0, 0, 0), (0, 1, 1, 0), (0, 0, 0, 2)]))
    # mapped variables over (s,t,u,v,w): F(u,v,w) + G(t,s) = 0, h(u,t,w,s) = 0.  (The splines of
    # a row must map to disjoint variables: the reference constructor rejects anything else,
    # spline_block.py:88-91, although its docstring example shares a variable.)
    cs.append(BlockCase("block_mapped", 902, [(0.0, 1.0), (0.0, 2.0), (-1.0, 1.0), (0.0, 1.0), (1.0, 3.0)],
                        [[((2, 3, 4), 2, (3, 4, 3), (5, 6, 4), f64), ((1, 0), 2, (4, 2), (7, 3), f64)],
                         [((2, 1, 4, 0), 1, (2, 3, 3, 2), (3, 4, 5, 3), f64)]], 193,
                        [(0, 0, 0, 0, 0), (0, 0, 1, 0, 0), (1, 0, 0, 1, 0), (0, 1, 0, 0, 0)]))
    # bicubic surfaces in the fast path: S1(a,b) + S2(d,c) + C(e), and a second row S3(b,a)
    cs.append(BlockCase("block_surfaces_sum", 903, [(0.0, 1.0), (0.0, 1.0), (0.0, 1.0), (-1.0, 1.0), (0.0, 2.0)],
                        [[((0, 1), 3, (4, 4), (16, 12), f64), ((3, 2), 3, (4, 4), (9, 10), f64), ((4,), 3, (4,), (7,), f64)],
                         [((1, 0), 2, (4, 4), (8, 8), f64)]], 1025,
                        [(0, 0, 0, 0, 0), (1, 0, 0, 0, 0), (0, 1, 1, 0, 0), (0, 0, 0, 0, 2), (0, 0, 0, 1, 0)]))
    # float32 coefficients decide the block's result dtype (first spline of the first row)
    cs.append(BlockCase("block_f32_coefs", 904, [(0.0, 1.0), (0.0, 1.0)],
                        [[((0, 1), 2, (3, 3), (6, 6), f32)], [((1,), 1, (4,), (9,), f32), ((0,), 1, (3,), (5,), f32)]], 129,
                        [(0, 0), (0, 1)]))
    return cs


def teapot_patches(tables, which=None, dtype=np.float32):
    """Control nets of the Utah teapot's bicubic Bezier patches as (order, nCoef, knots, coefs)
    from the reference's data tables (tests/golden/reference_tables.npz: examples/teapot.py:4-346),
    arranged like the reference example does (x, z, y -> nDep 0, 1, 2; examples/teapot.py:349-358)."""
    V, P = tables["teapot_vertices"], tables["teapot_patch_index"]
    knots = np.array((0, 0, 0, 0, 1, 1, 1, 1), dtype)
    out = []
    for pi in (range(len(P)) if which is None else which):
        c = np.empty((3, 4, 4), dtype)
        for i in range(4):
            for j in range(4):
                v = V[P[pi][4 * i + j] - 1]
                c[0, i, j], c[1, i, j], c[2, i, j] = v[0], v[2], v[1]
        out.append(((4, 4), (4, 4), [knots, knots], c))
    return out


def tess_cases():
    """Tessellation batches: name -> (list of (order, nCoef, knots, coefs), u, v).
    The teapot batch is built by the caller from reference_tables.npz (teapot_patches)."""
    rng = np.random.default_rng(4242)
    out = {}
    # five order-3 x 3 patches on shared non-uniform knots, fp64, odd grid sizes
    ku = nonuniform_knots(rng, 3, 6, np.float64, 0.0, 2.0)
    kv = nonuniform_knots(rng, 3, 7, np.float64, -1.0, 1.0)
    patches = [((3, 3), (6, 7), [ku, kv], rng.standard_normal((3, 6, 7))) for _ in range(5)]
    out["o3_f64"] = (patches, np.linspace(0.0, 2.0, 11), np.linspace(-1.0, 1.0, 9))
    # two order-5 patches, fp64, grid sizes divisible by the vector width
    ku = clamped_uniform_knots(5, 9)
    kv = clamped_uniform_knots(5, 8)
    patches = [((5, 5), (9, 8), [ku, kv], rng.standard_normal((3, 9, 8))) for _ in range(2)]
    out["o5_f64"] = (patches, np.linspace(0.0, 1.0, 6), np.linspace(0.0, 1.0, 8))
    # three patches of order (2, 5) - a ruled-surface shape - and grid sizes that take the vector path
    ku = nonuniform_knots(rng, 2, 6, np.float64, 0.0, 1.0)
    kv = nonuniform_knots(rng, 5, 9, np.float64, 0.0, 3.0)
    patches = [((2, 5), (6, 9), [ku, kv], rng.standard_normal((3, 6, 9))) for _ in range(3)]
    out["o2x5_f64"] = (patches, np.linspace(0.0, 1.0, 7), np.linspace(0.0, 3.0, 10))
    return out


def basis_cases():
    """Direct bspline_values goldens: tuples
    (knots, order, u, derivativeOrder, taylorCoefs, explicit_knot_or_None)."""
    rng = np.random.default_rng(777)
    out = []
    for order in range(1, 10):
        for dtype in (np.float64, np.float32):
            ncoef = order + int(rng.integers(0, 8))
            knots = nonuniform_knots(rng, order, ncoef, dtype, -1.0, 2.5)
            lo, hi = knots[order - 1], knots[ncoef]
            us = list(lo + (hi - lo) * rng.random(6)) + [lo, hi] + list(np.unique(knots))
            for u in us:
                u = dtype(u)
                for deriv in range(0, order + 2):
                    for taylor in (False, True):
                        out.append((knots, order, u, deriv, taylor, None))
            # explicit knot argument (no search): evaluate a span's polynomial outside its span
            for _ in range(4):
                ix = int(rng.integers(order, ncoef + 1))
                if knots[ix] - knots[ix - 1] <= 0:
                    continue
                u = dtype(lo + (hi - lo) * rng.random())
                out.append((knots, order, u, int(rng.integers(0, order)), bool(rng.integers(0, 2)), ix))
    return out


def bench_spline(cfg, seed=0):
    """The BASELINE.json configs' splines: returns
    (nInd, nDep, order, nCoef, knots, coefs, dtype)."""
    rng = np.random.default_rng(seed)
    if cfg == 1:
        nind, ndep, order, ncoef, dt = 1, 1, (4,), (32,), np.float64
    elif cfg in (2, 3):
        nind, ndep, order, ncoef, dt = 2, 3, (4, 4), (64, 64), np.float64
    elif cfg == 5:
        nind, ndep, order, ncoef, dt = 3, 4, (5, 5, 5), (40, 40, 40), np.float32
    else:
        raise ValueError(cfg)
    knots = [clamped_uniform_knots(o, c, dt) for o, c in zip(order, ncoef)]
    coefs = rng.standard_normal((ndep, *ncoef)).astype(dt)
    return nind, ndep, order, ncoef, knots, coefs, dt
