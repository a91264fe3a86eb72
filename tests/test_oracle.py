"""
Pins the CPU oracle (oracle/) to the reference: the reference's own golden
tables (tests/bspy_test.py:15-564, :702-709, :743-761 in the reference) and
outputs of the reference itself captured by tests/golden/make_golden.py.
CPU only.
"""
import numpy as np
import pytest

import cases
import oracle

EPS = np.finfo(float).eps
CASES = {c.name: c for c in cases.parity_cases()}


def _tol(case):
    """All-fp32 cases: the C oracle reproduces the reference's fp32 operation
    sequence (few ulp of the largest term).  Mixed-precision cases are computed
    in fp64 by the oracle (and by the product) while the reference rounds its
    basis to fp32, so they are only comparable at fp32 resolution."""
    kd, cd = case.knots[0].dtype, case.coefs.dtype
    if kd == np.float32 and cd == np.float32:
        return 2e-5
    if kd == np.float32 or cd == np.float32:
        return 2e-5
    return 1e-12


def _scale(ref):
    return max(1.0, float(np.max(np.abs(ref))))


def test_reference_truth_curve(golden_tables):
    """reference test_evaluate, curve half: error <= eps at all 101 rows."""
    t = golden_tables
    knots, coefs = [t["curve_knots0"]], t["curve_coefs"]
    tc = t["truthCurve"]
    out, bad = oracle.c_evaluate([4], [5], knots, coefs, [0], [tc[:, 0]])
    assert bad == -1
    err = np.sqrt((out[0] - tc[:, 1]) ** 2 + (out[1] - tc[:, 2]) ** 2)
    assert err.max() <= EPS
    for u, x, y in tc[::10]:
        xt, yt = oracle.py_evaluate([4], [5], knots, coefs, [u])
        assert np.hypot(xt - x, yt - y) <= EPS


def test_reference_truth_surface(golden_tables):
    """reference test_evaluate, surface half: 21x21 grid, v outer / u inner, error <= 2.5 eps."""
    t = golden_tables
    knots, coefs = [t["surface_knots0"], t["surface_knots1"]], t["surface_coefs"]
    g = np.linspace(0, 1, 21)
    v, u = np.meshgrid(g, g, indexing="ij")
    out, bad = oracle.c_evaluate([3, 4], [4, 5], knots, coefs, [0, 0], [u.ravel(), v.ravel()])
    assert bad == -1
    d = out.T - t["truthSurface"]
    assert np.sqrt((d * d).sum(axis=1)).max() <= 2.5 * EPS
    for i in range(0, 441, 37):
        r = oracle.py_evaluate([3, 4], [4, 5], knots, coefs, [u.ravel()[i], v.ravel()[i]])
        assert np.linalg.norm(r - t["truthSurface"][i]) <= 2.5 * EPS


def test_reference_derivative_crosscheck(golden_tables):
    """reference test_derivative: derivative([1], u) against differentiate().evaluate(u)."""
    t = golden_tables
    tc = t["truthCurve"]
    out, bad = oracle.c_evaluate([4], [5], [t["curve_knots0"]], t["curve_coefs"], [1], [tc[:, 0]])
    assert bad == -1
    d = out.T - t["curve_differentiate_eval"]
    assert (d * d).sum(axis=1).max() <= EPS
    jac, _ = oracle.c_jacobian([3, 4], [4, 5], [t["surface_knots0"], t["surface_knots1"]], t["surface_coefs"],
                               [np.array([0.25]), np.array([0.5])])
    assert np.abs(jac[:, :, 0] - t["surface_jacobian_025_05"]).max() <= 4 * EPS


@pytest.mark.parametrize("name", sorted(CASES))
def test_c_oracle_against_reference_outputs(name, golden_parity):
    c = CASES[name]
    tol = _tol(c)
    for w in c.wrts:
        ref = golden_parity[f"{name}/wrt_" + "_".join(map(str, w))]
        out, bad = oracle.c_evaluate(c.order, c.nCoef, c.knots, c.coefs, list(w), c.points)
        assert bad == -1
        assert out.shape == ref.shape
        assert np.abs(out - ref).max() <= tol * _scale(ref), (name, w)
    if c.jacobian:
        ref = golden_parity[f"{name}/jac"]          # (N, nDep, nInd)
        out, bad = oracle.c_jacobian(c.order, c.nCoef, c.knots, c.coefs, c.points)
        assert bad == -1
        assert np.abs(out.transpose(2, 0, 1) - ref).max() <= tol * _scale(ref)


@pytest.mark.parametrize("name", ["cfg2_bicubic", "surface_o3x4", "curve_f32", "volume_o3x4x2", "curve_order5"])
def test_py_oracle_against_reference_outputs(name, golden_parity):
    c = CASES[name]
    n = min(c.n, 64)
    pts = [p[:n] for p in c.points]
    for w in c.wrts[:4]:
        ref = golden_parity[f"{name}/wrt_" + "_".join(map(str, w))][:, :n]
        out = oracle.py_batch(c.order, c.nCoef, c.knots, c.coefs, list(w), pts)
        assert np.abs(out - ref).max() <= (1e-6 if c.coefs.dtype == np.float32 else 4 * EPS) * _scale(ref)
    ref = golden_parity[f"{name}/jac"][:8]
    for i in range(8):
        j = oracle.py_jacobian(c.order, c.nCoef, c.knots, c.coefs, [float(p[i]) for p in c.points])
        assert np.abs(j - ref[i]).max() <= (1e-5 if c.coefs.dtype == np.float32 else 1e-13) * _scale(ref)


def test_bspline_values_goldens(golden_basis):
    ix_ref, basis_ref = golden_basis["ix"], golden_basis["basis"]
    for k, (knots, order, u, deriv, taylor, knot) in enumerate(cases.basis_cases()):
        ix, b = oracle.c_bspline_values(knot, knots, order, u, deriv, taylor)
        assert ix == ix_ref[k]
        ref = basis_ref[k, :order]
        tol = (4e-6 if knots.dtype == np.float32 else 1e-13) * max(1.0, np.abs(ref).max())
        assert np.abs(b - ref).max() <= tol, (k, order, deriv, taylor)
        if k % 17 == 0:
            ix2, b2 = oracle.py_bspline_values(knot, knots, order, u, deriv, taylor)
            assert ix2 == ix_ref[k]
            assert np.abs(b2 - ref).max() <= tol


def test_span_and_errors(golden_api):
    k = np.array([0, 0, 0, 0, .3, .3, .7, 1, 1, 1, 1])
    assert [oracle.c_bspline_values(None, k, 4, u)[0] for u in (0.0, 0.3, 0.7, 1.0)] == golden_api["span_at_knots"]
    coefs = np.arange(14.0).reshape(2, 7)
    out, bad = oracle.c_evaluate([4], [7], [k], coefs, [0], [np.array([0.1, 0.2, -0.25, 3.0])])
    assert bad == 2                                     # first offender, as in the reference's message
    with pytest.raises(ValueError, match=r"Spline evaluation outside domain: \[1.5\]"):
        oracle.py_evaluate([4], [7], [k], coefs, [1.5])
    with pytest.raises(ValueError, match="Incorrect number of parameter values: 2"):
        oracle.py_evaluate([4], [7], [k], coefs, [0.1, 0.2])
    out, bad = oracle.c_evaluate([4], [7], [k], coefs, [0], [np.array([np.nan])])
    assert bad == -1 and np.isnan(out).all()             # NaN passes the domain check
    out, bad = oracle.c_evaluate([4], [7], [k], coefs, [4], [np.array([0.5])])
    assert (out == 0).all()                             # derivativeOrder >= order -> exact zeros


NORMAL_CASES = [n for n, c in CASES.items() if abs(c.nInd - c.nDep) == 1 and max(c.nInd, c.nDep) <= 4]


@pytest.mark.parametrize("name", sorted(NORMAL_CASES))
def test_normal_oracle_against_reference(name, golden_parity):
    """Oracle normal (next row 8f-1) against Spline.normal of the reference: unit, area-scaled
    and with metadata negateNormal."""
    c = CASES[name]
    m = golden_parity[f"{name}/normal_unit"].shape[0]
    pts = [p[:m] for p in c.points]
    tol = 5e-5 if (c.knots[0].dtype == np.float32 or c.coefs.dtype == np.float32) else 1e-12
    for key, normalize, negate in (("normal_unit", True, False), ("normal_area", False, False),
                                   ("normal_area_negated", False, True)):
        ref = golden_parity[f"{name}/{key}"].T                     # (big, m)
        out, bad = oracle.c_normal(c.order, c.nCoef, c.knots, c.coefs, pts, normalize, negate)
        assert bad == -1
        # an order-1 curve has zero tangent: the unit normal is 0/0 = NaN in the reference too
        assert np.array_equal(np.isnan(out), np.isnan(ref)), (name, key)
        if not np.isnan(ref).all():
            assert np.nanmax(np.abs(out - ref)) <= tol * max(1.0, float(np.nanmax(np.abs(ref)))), (name, key)
    if c.coefs.dtype == np.float64 and c.order[0] > 1:
        r = oracle.py_normal(c.order, c.nCoef, c.knots, c.coefs, [float(p[3]) for p in c.points])
        assert np.abs(r - golden_parity[f"{name}/normal_unit"][3]).max() <= 1e-12


CURVATURE_CASES = [n for n, c in CASES.items() if (c.nInd == 1 and c.nDep >= 2) or (c.nInd == 2 and c.nDep == 3)]


def _curv_tol(c):
    if c.knots[0].dtype == np.float32 or c.coefs.dtype == np.float32:
        return 2e-3          # second derivatives in fp32
    return 1e-9


@pytest.mark.parametrize("name", sorted(CURVATURE_CASES))
def test_curvature_oracle_against_reference(name, golden_parity):
    """Oracle curvature (next row 8f-1) against Spline.curvature of the reference."""
    c = CASES[name]
    ref = golden_parity[f"{name}/curvature"]
    pts = [p[:len(ref)] for p in c.points]
    with np.errstate(all="ignore"):
        out, bad = oracle.c_curvature(c.order, c.nCoef, c.knots, c.coefs, pts)
    assert bad == -1
    ok = np.isfinite(ref)
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    if ok.any():
        # relative to each value's own size: curvatures span many orders of magnitude
        err = np.abs(out[ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok]))
        assert err.max() <= _curv_tol(c), (name, err.max())
