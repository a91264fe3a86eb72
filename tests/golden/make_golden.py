"""
Golden-vector generator.  Runs ONLY in the build container, where the read-only
reference checkout lives at /root/reference; its outputs (``*.npz`` / ``*.json``
in this directory) are committed and are what travels to the GPU box.

It imports the reference package (with dummy GUI modules),
feeds it the deterministic inputs of ``tests/cases.py`` and stores what the
reference returns.  No reference source text is copied: fixtures hold inputs
and expected outputs only.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402


def load_reference():
    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, n):
            return _Dummy()

        def __call__(self, *a, **k):
            return _Dummy()

    def stub(name):
        m = types.ModuleType(name)
        m.__getattr__ = lambda n: type(n, (_Dummy,), {})
        m.__all__ = []
        m.__path__ = []
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, m)

    for n in ["OpenGL", "OpenGL.GL", "OpenGL.GLU", "OpenGL.GL.shaders", "pyopengltk",
              "tkinter", "tkinter.ttk", "tkinter.colorchooser", "tkinter.filedialog"]:
        stub(n)
    sys.path.insert(0, "/root/reference")
    sys.dont_write_bytecode = True
    import bspy
    return bspy


def reference_tables(bspy):
    """The reference's own golden tables (tests/bspy_test.py:15-564) and the
    Utah teapot data tables (examples/teapot.py:4-346), captured as data."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_bspy_test", "/root/reference/tests/bspy_test.py")
    t = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(t)
    out = {
        "truthCurve": np.array(t.truthCurve, np.float64),
        "truthSurface": np.array(t.truthSurface, np.float64),
        "curve_knots0": np.array(t.myCurve.knots[0], np.float64),
        "curve_coefs": np.ascontiguousarray(t.myCurve.coefs, np.float64),
        "surface_knots0": np.array(t.mySurface.knots[0], np.float64),
        "surface_knots1": np.array(t.mySurface.knots[1], np.float64),
        "surface_coefs": np.ascontiguousarray(t.mySurface.coefs, np.float64),
    }
    # test_derivative's cross-check target: differentiate().evaluate(u) (tests/bspy_test.py:702-709)
    d = t.myCurve.differentiate()
    out["curve_differentiate_eval"] = np.array([d.evaluate(u) for u in out["truthCurve"][:, 0]])
    # test_curvature pin: gaussian curvature of mySurface at (0.25, 0.5) (tests/bspy_test.py:699-700)
    out["surface_jacobian_025_05"] = t.mySurface.jacobian([0.25, 0.5])
    spec = importlib.util.spec_from_file_location("ref_teapot", "/root/reference/examples/teapot.py")
    tp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tp)
    out["teapot_patch_index"] = np.array([p[1:] for p in tp.teapotPatches], np.int32)
    out["teapot_vertices"] = np.array(tp.teapotVertices, np.float64)
    names = [p[0] for p in tp.teapotPatches]
    # oracle output of every teapot patch on a 16x16 broadcast grid (fp32, cfg4 shape)
    knots = np.array((0, 0, 0, 0, 1, 1, 1, 1), np.float32)
    g = np.linspace(0, 1, 16, dtype=np.float32)
    grids = []
    for p in tp.teapotPatches:
        c = np.empty((3, 4, 4), np.float32)
        for i in range(4):
            for j in range(4):
                v = p[4 * i + j + 1] - 1
                c[0, i, j] = tp.teapotVertices[v][0]
                c[1, i, j] = tp.teapotVertices[v][2]
                c[2, i, j] = tp.teapotVertices[v][1]
        s = bspy.Spline(2, 3, (4, 4), (4, 4), (knots, knots), c)
        grids.append(np.stack(s(g[:, None], g[None, :])))
    out["teapot_grid16"] = np.stack(grids).astype(np.float32)
    return out, names


def run_case(bspy, c):
    s = bspy.Spline(c.nInd, c.nDep, c.order, c.nCoef, c.knots, c.coefs)
    out = {}
    for w in c.wrts:
        r = s.derivative(list(w), *c.points) if any(w) else s.evaluate(*c.points)
        r = np.stack(r) if isinstance(r, tuple) else np.asarray(r)[None, :]
        out["wrt_" + "_".join(map(str, w))] = r
    if c.jacobian:
        jac = np.empty((c.n, c.nDep, c.nInd), c.coefs.dtype)
        for i in range(c.n):
            jac[i] = s.jacobian([p[i] for p in c.points])
        out["jac"] = jac
    if abs(c.nInd - c.nDep) == 1 and max(c.nInd, c.nDep) <= 4:
        # Spline.normal (bspy/_spline_evaluation.py:215-246): unit and area-scaled, first 128 points
        m = min(c.n, 128)
        big = max(c.nInd, c.nDep)
        nu = np.empty((m, big), c.coefs.dtype)
        na = np.empty((m, big), c.coefs.dtype)
        flipped = bspy.Spline(c.nInd, c.nDep, c.order, c.nCoef, c.knots, c.coefs, {"negateNormal": True})
        nf = np.empty((m, big), c.coefs.dtype)
        for i in range(m):
            uvw = [p[i] for p in c.points]
            nu[i] = s.normal(uvw)
            na[i] = s.normal(uvw, False)
            nf[i] = flipped.normal(uvw, False)
        out["normal_unit"], out["normal_area"], out["normal_area_negated"] = nu, na, nf
    if (c.nInd == 1 and c.nDep >= 2) or (c.nInd == 2 and c.nDep == 3):
        # Spline.curvature (bspy/_spline_evaluation.py:80-107), first 128 points
        m = min(c.n, 128)
        with np.errstate(all="ignore"):
            out["curvature"] = np.array([s.curvature([p[i] for p in c.points]) for i in range(m)], np.float64)
    return out


def api_semantics(bspy):
    """Shapes, dtypes, return types and error strings of the public wrappers
    (spline.py:46-76, :720-770, :904-949; _spline_evaluation.py:109-164)."""
    rec = {}
    k = [0, 0, 0, 0, .3, .3, .7, 1, 1, 1, 1]
    s = bspy.Spline(1, 2, [4], [7], [k], np.arange(14.0).reshape(2, 7))
    rec["span_at_knots"] = [int(bspy.Spline.bspline_values(None, np.array(k), 4, u)[0]) for u in (0.0, 0.3, 0.7, 1.0)]

    def err(f):
        try:
            f()
        except Exception as e:  # noqa: BLE001
            return [type(e).__name__, str(e)]
        return None

    rec["err_outside_scalar"] = err(lambda: s.evaluate([1.5]))
    rec["err_outside_batch"] = err(lambda: s(np.array([0.1, 0.2, -0.25, 3.0])))
    rec["err_arity"] = err(lambda: s.evaluate([0.1, 0.2]))
    rec["err_outside_deriv"] = err(lambda: s.derivative([1], [-0.5]))
    s2 = bspy.Spline(2, 3, [3, 4], [4, 5], [[0, 0, 0, .5, 1, 1, 1], [0, 0, 0, 0, .5, 1, 1, 1, 1]],
                     np.arange(60.0).reshape(3, 4, 5))
    rec["err_outside_batch2"] = err(lambda: s2(np.array([0.1, 0.2, 0.3]), np.array([0.5, 1.25, 7.0])))
    rec["err_arity2"] = err(lambda: s2.evaluate([0.1]))
    rec["err_ctor"] = {
        "nInd": err(lambda: bspy.Spline(-1, 1, [], [], [], [])),
        "order": err(lambda: bspy.Spline(1, 1, [4, 4], [4], [k], [[0, 1, 2, 3]])),
        "nCoef": err(lambda: bspy.Spline(1, 1, [4], [4, 4], [k], [[0, 1, 2, 3]])),
        "nknots": err(lambda: bspy.Spline(1, 1, [4], [4], [k], [[0.0, 1, 2, 3]])),
        "knots_len": err(lambda: bspy.Spline(1, 1, [4], [4], [k, k], [[0.0, 1, 2, 3]])),
        "knot_order": err(lambda: bspy.Spline(1, 1, [4], [4], [[0, 0, 0.5, 0.2, 1, 1, 1, 1]], [[0.0, 1, 2, 3]])),
        "knot_mult": err(lambda: bspy.Spline(1, 1, [4], [5], [[0, 0, 0, 0, 0, 1, 1, 1, 1.0]], [[0.0, 1, 2, 3, 4]])),
        "coefs_len": err(lambda: bspy.Spline(1, 2, [4], [4], [[0, 0, 0, 0, 1, 1, 1, 1.0]], [[0.0, 1, 2]])),
    }
    # return types / shapes
    r = s.evaluate([0.25])
    rec["scalar_list_shape"] = [list(r.shape), str(r.dtype), [float(x) for x in r]]
    r = s.evaluate(0.25)
    rec["scalar_shape"] = [list(r.shape), str(r.dtype), [float(x) for x in r]]
    r = s2(0.25, 0.5)
    rec["scalar2_shape"] = [list(r.shape), str(r.dtype), [float(x) for x in r]]
    r = s2([0.25, 0.5])
    rec["scalar2_list_shape"] = [list(r.shape), str(r.dtype), [float(x) for x in r]]
    r = s2(np.array([0.25, 0.5]))
    rec["scalar2_ndarray_shape"] = [list(r.shape), str(r.dtype), [float(x) for x in r]]
    r = s(np.array([0.25, 0.5, 0.75]))
    rec["batch_type"] = [type(r).__name__, len(r), list(r[0].shape), str(r[0].dtype)]
    s1 = bspy.Spline(1, 1, [4], [7], [k], np.arange(7.0))
    r = s1(np.array([0.25, 0.5, 0.75]))
    rec["batch_ndep1_type"] = [type(r).__name__, list(r.shape), str(r.dtype), [float(x) for x in r]]
    u = np.linspace(0, 1, 5)
    r = s2(u[:, None], u[None, :])
    rec["grid_type"] = [type(r).__name__, len(r), list(r[0].shape)]
    rec["grid_values"] = [x.tolist() for x in r]
    r = s2(u, 0.5)
    rec["array_scalar_values"] = [x.tolist() for x in r]
    r = s2.derivative([1, 0], u, 0.5)
    rec["array_scalar_deriv_values"] = [x.tolist() for x in r]
    rec["jacobian"] = s2.jacobian([0.25, 0.5]).tolist()
    rec["tangent_space"] = s2.tangent_space([0.25, 0.5]).tolist()
    rec["domain"] = s2.domain().tolist()
    rec["nan_scalar"] = [float(x) for x in s.evaluate([float("nan")])]
    # flat "list of points" coefficient form (spline.py:72-73)
    flat = np.arange(60.0).reshape(20, 3)
    s3 = bspy.Spline(2, 3, [3, 4], [4, 5], s2.knots, flat)
    rec["flat_coefs"] = np.ascontiguousarray(s3.coefs).tolist()
    # per-dependent-variable list form with transposed blocks (spline.py:75)
    per = [np.arange(20.0).reshape(5, 4) + 100 * d for d in range(3)]
    s4 = bspy.Spline(2, 3, [3, 4], [4, 5], s2.knots, per)
    rec["perdep_coefs"] = np.ascontiguousarray(s4.coefs).tolist()
    # nInd == 0
    s0 = bspy.Spline(0, 2, [], [], [], [1.5, 2.5])
    rec["nind0_eval"] = np.asarray(s0.evaluate()).tolist()
    rec["nind0_deriv"] = np.asarray(s0.derivative([])).tolist()
    # derivativeOrder >= order gives exact zeros (tests/bspy_test.py:759-761)
    rec["zero_derivative"] = s.derivative([4], [0.5]).tolist()
    return rec


def block_golden(bspy):
    """SplineBlock evaluation path (reference bspy/spline_block.py:179-247), one point at a time -
    the reference has no batched block call - on the first 48 points of every block case."""
    out = {}
    for c in cases.block_cases():
        rows = [[(imap, bspy.Spline(*d)) for (imap, d) in row] for row in c.rows]
        blk = bspy.spline_block.SplineBlock(rows)
        m = 48
        pts = np.array([p[:m] for p in c.points])                       # (nInd, m)
        out[f"{c.name}/nInd_nDep"] = np.array([blk.nInd, blk.nDep])
        out[f"{c.name}/domain"] = np.asarray(blk.domain(), np.float64)
        out[f"{c.name}/evaluate"] = np.array([blk.evaluate(pts[:, i]) for i in range(m)], np.float64).T
        out[f"{c.name}/jacobian"] = np.moveaxis(np.array([blk.jacobian(pts[:, i]) for i in range(m)], np.float64), 0, -1)
        for w in c.wrts:
            out[f"{c.name}/wrt_" + "_".join(map(str, w))] = np.array([blk.derivative(w, pts[:, i]) for i in range(m)], np.float64).T
        out[f"{c.name}/dtype"] = np.array(str(blk.evaluate(pts[:, 0]).dtype))
        print("block", c.name, "done", flush=True)
    return out


def tess_golden(bspy):
    """Positions and unit normals on a grid, per patch, from the reference: the broadcast call
    s(u[:, None], v[None, :]) and Spline.normal at every grid point."""
    out = {}
    tables = np.load(os.path.join(HERE, "reference_tables.npz"))
    batches = dict(cases.tess_cases())
    batches["teapot_f32"] = (cases.teapot_patches(tables, which=(0, 5, 13, 31)), np.linspace(0, 1, 8, dtype=np.float32),
                             np.linspace(0, 1, 12, dtype=np.float32))
    for name, (patches, u, v) in batches.items():
        pos, nrm = [], []
        for (order, ncoef, knots, coefs) in patches:
            s = bspy.Spline(2, 3, order, ncoef, knots, coefs)
            pos.append(np.array(s(u[:, None], v[None, :]), np.float64))
            nrm.append(np.array([[s.normal((ui, vj)) for vj in v] for ui in u], np.float64).transpose(2, 0, 1))
        out[f"{name}/positions"] = np.array(pos)
        out[f"{name}/normals"] = np.array(nrm)
        print("tess", name, "done", flush=True)
    return out


def extras_golden(bspy):
    """Round-2 fixtures recorded from the reference (inputs regenerated from seeds by the tests):
      * Spline.normal with index subsets, normalised and not (bspy/_spline_evaluation.py:215-246: the
        normalisation divides by the norm of the SELECTED cofactors),
      * Spline.curvature of scalar-valued splines (nDep == 1: the reference evaluates the curvature of the
        graph of the function, bspy/_spline_evaluation.py:81-82 -> graph(), bspy/_spline_operations.py:281-288),
      * the collocation matrix A that the reference's least_squares assembles (bspy/_spline_fitting.py:736-751),
        captured from its own call of numpy.linalg.lstsq (repeated parameter values = derivative rows).
      * evaluation with ufunc keyword arguments where= / out= (bspy/spline.py:943-947)."""
    out = {}
    rng = np.random.default_rng(2024)
    # --- normal with indices
    ku = np.concatenate([np.zeros(4), np.linspace(0, 1, 6)[1:-1], np.ones(4)])
    kv = np.concatenate([np.zeros(3), np.array([0.3, 0.55]), np.ones(3)])
    surf = bspy.Spline(2, 3, [4, 3], [8, 5], [ku, kv], rng.standard_normal((3, 8, 5)))
    pts = rng.random((40, 2))
    out["normal_surf_knots0"], out["normal_surf_knots1"], out["normal_surf_coefs"] = ku, kv, surf.coefs
    out["normal_pts"] = pts
    for name, idx in (("0_2", [0, 2]), ("2", [2]), ("1_0", [1, 0])):
        for nz in (True, False):
            out[f"normal_idx_{name}_{int(nz)}"] = np.array([surf.normal(p, nz, idx) for p in pts])
    curve2 = bspy.Spline(1, 2, [4], [8], [ku], rng.standard_normal((2, 8)))
    out["normal_curve_coefs"] = curve2.coefs
    out["normal_curve_idx_1"] = np.array([curve2.normal([u], True, [1]) for u in pts[:, 0]])
    # --- curvature of graphs (nDep == 1)
    f1 = bspy.Spline(1, 1, [4], [8], [ku], rng.standard_normal((1, 8)))
    out["graph_curve_coefs"] = f1.coefs
    out["graph_curve_curvature"] = np.array([f1.curvature(u) for u in pts[:, 0]])
    f2 = bspy.Spline(2, 1, [4, 3], [8, 5], [ku, kv], rng.standard_normal((1, 8, 5)))
    out["graph_surf_coefs"] = f2.coefs
    out["graph_surf_curvature"] = np.array([f2.curvature(p) for p in pts])
    # --- collocation matrix captured from least_squares
    captured = []
    real = np.linalg.lstsq

    def spy(A, b, rcond=None):
        captured.append(np.array(A))
        return real(A, b, rcond=rcond)

    u = np.sort(rng.random(30))
    u = np.concatenate([u[:10], [u[10], u[10], u[10]], u[11:20], [u[20], u[20]], u[21:]])      # Hermite rows
    kn = np.concatenate([np.zeros(5), np.sort(rng.random(7)), np.ones(5)])
    np.linalg.lstsq = spy
    try:
        bspy.Spline.least_squares([u], rng.standard_normal((2, len(u))), order=[5], knots=[kn], fixEnds=False)
    finally:
        np.linalg.lstsq = real
    out["colloc_u"], out["colloc_knots"], out["colloc_A"] = u, kn, captured[0]
    # --- ufunc keyword arguments
    uu, vv = rng.random(12), rng.random(12)
    mask = rng.random(12) > 0.4
    res = surf.evaluate(uu, vv, where=mask)
    out["where_u"], out["where_v"], out["where_mask"] = uu, vv, mask
    out["where_result"] = np.array([[r[i] if mask[i] else np.nan for i in range(12)] for r in res], np.float64)
    return out


def main():
    bspy = load_reference()
    if "--only-extras" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "extras.npz"), **extras_golden(bspy))
        return
    if "--only-tess" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "tess.npz"), **tess_golden(bspy))
        return
    if "--only-block" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "block.npz"), **block_golden(bspy))
        return
    tables, names = reference_tables(bspy)
    np.savez_compressed(os.path.join(HERE, "reference_tables.npz"), **tables)
    with open(os.path.join(HERE, "teapot_names.json"), "w") as f:
        json.dump(names, f)

    parity = {}
    for c in cases.parity_cases():
        for k, v in run_case(bspy, c).items():
            parity[f"{c.name}/{k}"] = v
        print("case", c.name, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "parity.npz"), **parity)

    ix, basis = [], []
    for (knots, order, u, deriv, taylor, knot) in cases.basis_cases():
        i, b = bspy.Spline.bspline_values(knot, knots, order, u, deriv, taylor)
        ix.append(int(i))
        row = np.zeros(9, np.float64)
        row[:order] = b
        basis.append(row)
    np.savez_compressed(os.path.join(HERE, "basis.npz"), ix=np.array(ix, np.int32), basis=np.array(basis))

    with open(os.path.join(HERE, "api_semantics.json"), "w") as f:
        json.dump(api_semantics(bspy), f, indent=1)
    np.savez_compressed(os.path.join(HERE, "block.npz"), **block_golden(bspy))
    np.savez_compressed(os.path.join(HERE, "tess.npz"), **tess_golden(bspy))
    np.savez_compressed(os.path.join(HERE, "extras.npz"), **extras_golden(bspy))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
