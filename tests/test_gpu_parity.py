"""
GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(bspy_amd.DeviceSpline -> ctypes -> libbspy_amd.so), against
  * outputs of the reference itself (tests/golden/parity.npz, basis.npz, reference_tables.npz),
  * the CPU oracle (oracle/) on the same seeded inputs,
  * size-independent properties at the BASELINE size (10 M points).
Tolerances: fp64 1e-10 (BASELINE.json north_star), asserted far tighter (1e-12 of the
result scale) where the arithmetic allows; fp32 cases 2e-5 of the result scale.
"""
import os

import numpy as np
import pytest

import cases
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import bspy_amd
from bspy_amd import DeviceSpline, Spline
from conftest import observe

pytestmark = pytest.mark.gpu

CASES = {c.name: c for c in cases.parity_cases()}
EPS = np.finfo(float).eps


def _is_f32(c):
    return c.knots[0].dtype == np.float32 and c.coefs.dtype == np.float32


def _tol(c):
    if _is_f32(c):
        return 2e-5
    if c.knots[0].dtype == np.float32 or c.coefs.dtype == np.float32:
        return 2e-5          # the reference rounds its basis / result to fp32 in these cases
    return 1e-12


def _scale(ref):
    return max(1.0, float(np.max(np.abs(ref))))


def _tables(c):
    dt = np.float32 if _is_f32(c) else np.float64
    return DeviceSpline(c.order, c.nCoef, c.knots, c.coefs, dt)


def test_library_is_native():
    assert bspy_amd._native.lib().bsk_version() == 1
    assert bspy_amd._native.device_count() >= 1


@pytest.mark.parametrize("name", sorted(CASES))
def test_evaluate_derivative_against_reference(name, golden_parity):
    c = CASES[name]
    t = _tables(c)
    tol = _tol(c)
    for w in c.wrts:
        ref = golden_parity[f"{name}/wrt_" + "_".join(map(str, w))]
        out = t.evaluate(c.points, list(w))
        assert out.shape == ref.shape
        err = np.abs(out - ref).max()
        assert err <= tol * _scale(ref), (name, w, err)
        orc, bad = oracle.c_evaluate(c.order, c.nCoef, c.knots, c.coefs, list(w), c.points)
        assert bad == -1
        assert np.abs(out - orc).max() <= tol * _scale(orc), (name, w)
        if not _is_f32(c) and tol > 1e-12:
            # mixed fp32 / fp64 inputs: the reference rounds its basis to fp32, the product computes in fp64
            # (documented deviation) - so it agrees with the reference to fp32 rounding (above) and with the
            # SAME algorithm carried out in fp64 on the promoted inputs to the fp64 bar
            k64 = [np.asarray(k, np.float64) for k in c.knots]
            p64 = [np.asarray(p, np.float64) for p in c.points]
            o64, _ = oracle.c_evaluate(c.order, c.nCoef, k64, np.asarray(c.coefs, np.float64), list(w), p64)
            assert np.abs(out - o64).max() <= 1e-12 * _scale(o64), (name, w, "fp64 restatement")
    # plain evaluation with wrt == None is the same kernel path as all-zero wrt
    zero = tuple([0] * c.nInd)
    if zero in c.wrts:
        assert np.array_equal(t.evaluate(c.points), t.evaluate(c.points, list(zero)))


@pytest.mark.parametrize("name", sorted(CASES))
def test_jacobian_against_reference(name, golden_parity):
    c = CASES[name]
    t = _tables(c)
    ref = golden_parity[f"{name}/jac"].transpose(1, 2, 0)      # (nDep, nInd, N)
    out = t.jacobian(c.points)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= _tol(c) * _scale(ref)
    orc, bad = oracle.c_jacobian(c.order, c.nCoef, c.knots, c.coefs, c.points)
    assert np.abs(out - orc).max() <= _tol(c) * _scale(orc)


def test_reference_truth_tables(golden_tables):
    """The reference's own test_evaluate / test_derivative tables through the Spline API."""
    t = golden_tables
    curve = Spline(1, 2, [4], [5], [t["curve_knots0"]], t["curve_coefs"])
    tc = t["truthCurve"]
    x, y = curve(tc[:, 0])
    assert np.sqrt((x - tc[:, 1]) ** 2 + (y - tc[:, 2]) ** 2).max() <= 8 * EPS
    for u, xr, yr in tc[::20]:
        xt, yt = curve.evaluate([u])
        assert np.hypot(xt - xr, yt - yr) <= 8 * EPS
    surf = Spline(2, 3, [3, 4], [4, 5], [t["surface_knots0"], t["surface_knots1"]], t["surface_coefs"])
    g = np.linspace(0, 1, 21)
    v, u = np.meshgrid(g, g, indexing="ij")
    xyz = np.stack(surf(u.ravel(), v.ravel())).T
    d = xyz - t["truthSurface"]
    assert np.sqrt((d * d).sum(axis=1)).max() <= 16 * EPS
    dx, dy = curve.derivative([1], tc[:, 0])
    dd = np.stack([dx, dy]).T - t["curve_differentiate_eval"]
    assert (dd * dd).sum(axis=1).max() <= 1e-26
    assert np.abs(surf.jacobian([0.25, 0.5]) - t["surface_jacobian_025_05"]).max() <= 64 * EPS


def test_bspline_values_goldens(golden_basis):
    ix_ref, basis_ref = golden_basis["ix"], golden_basis["basis"]
    bc = cases.basis_cases()
    # group by (knots identity, order, deriv, taylor, explicit) so each group is one batched call
    groups = {}
    for k, (knots, order, u, deriv, taylor, knot) in enumerate(bc):
        groups.setdefault((id(knots), order, deriv, taylor, knot is not None), []).append(k)
    for (_, order, deriv, taylor, explicit), idxs in groups.items():
        knots = bc[idxs[0]][0]
        us = np.array([bc[k][2] for k in idxs], knots.dtype)
        kin = np.array([bc[k][5] for k in idxs], np.int32) if explicit else None
        ix, basis = bspy_amd.bspline_values_batch(knots, order, us, deriv, taylor, kin)
        assert np.array_equal(ix, ix_ref[idxs])
        ref = basis_ref[idxs, :order]
        tol = (4e-6 if knots.dtype == np.float32 else 1e-12) * max(1.0, np.abs(ref).max())
        assert np.abs(basis - ref).max() <= tol
    # scalar form of the static method
    knots, order, u, deriv, taylor, knot = bc[5]
    ix, b = Spline.bspline_values(knot, knots, order, u, deriv, taylor)
    assert isinstance(ix, int) and ix == ix_ref[5] and b.shape == (order,) and b.dtype == knots.dtype


def test_api_semantics(golden_api):
    a = golden_api
    k = [0, 0, 0, 0, .3, .3, .7, 1, 1, 1, 1]
    s = Spline(1, 2, [4], [7], [k], np.arange(14.0).reshape(2, 7))
    assert [int(Spline.bspline_values(None, np.array(k), 4, u)[0]) for u in (0.0, 0.3, 0.7, 1.0)] == a["span_at_knots"]

    def err(f):
        try:
            f()
        except Exception as e:  # noqa: BLE001
            return [type(e).__name__, str(e)]
        return None

    assert err(lambda: s.evaluate([1.5])) == a["err_outside_scalar"]
    assert err(lambda: s(np.array([0.1, 0.2, -0.25, 3.0]))) == a["err_outside_batch"]
    assert err(lambda: s.evaluate([0.1, 0.2])) == a["err_arity"]
    assert err(lambda: s.derivative([1], [-0.5])) == a["err_outside_deriv"]
    s2 = Spline(2, 3, [3, 4], [4, 5], [[0, 0, 0, .5, 1, 1, 1], [0, 0, 0, 0, .5, 1, 1, 1, 1]],
                np.arange(60.0).reshape(3, 4, 5))
    assert err(lambda: s2(np.array([0.1, 0.2, 0.3]), np.array([0.5, 1.25, 7.0]))) == a["err_outside_batch2"]
    assert err(lambda: s2.evaluate([0.1])) == a["err_arity2"]

    def chk(r, rec):
        assert list(r.shape) == rec[0] and str(r.dtype) == rec[1]
        assert np.allclose(r, rec[2], rtol=0, atol=1e-13)

    chk(s.evaluate([0.25]), a["scalar_list_shape"])
    chk(s.evaluate(0.25), a["scalar_shape"])
    chk(s2(0.25, 0.5), a["scalar2_shape"])
    chk(s2([0.25, 0.5]), a["scalar2_list_shape"])
    chk(s2(np.array([0.25, 0.5])), a["scalar2_ndarray_shape"])
    r = s(np.array([0.25, 0.5, 0.75]))
    assert [type(r).__name__, len(r), list(r[0].shape), str(r[0].dtype)] == a["batch_type"]
    s1 = Spline(1, 1, [4], [7], [k], np.arange(7.0))
    r = s1(np.array([0.25, 0.5, 0.75]))
    assert [type(r).__name__, list(r.shape), str(r.dtype)] == a["batch_ndep1_type"][:3]
    assert np.allclose(r, a["batch_ndep1_type"][3], rtol=0, atol=1e-13)
    u = np.linspace(0, 1, 5)
    r = s2(u[:, None], u[None, :])
    assert [type(r).__name__, len(r), list(r[0].shape)] == a["grid_type"]
    assert np.allclose(np.stack(r), np.array(a["grid_values"]), rtol=0, atol=1e-12)
    assert np.allclose(np.stack(s2(u, 0.5)), np.array(a["array_scalar_values"]), rtol=0, atol=1e-12)
    assert np.allclose(np.stack(s2.derivative([1, 0], u, 0.5)), np.array(a["array_scalar_deriv_values"]), rtol=0, atol=1e-12)
    assert np.allclose(s2.jacobian([0.25, 0.5]), np.array(a["jacobian"]), rtol=0, atol=1e-12)
    assert np.allclose(s2.tangent_space([0.25, 0.5]), np.array(a["tangent_space"]), rtol=0, atol=1e-12)
    assert np.isnan(s.evaluate([float("nan")])).all()
    assert s.derivative([4], [0.5]).tolist() == a["zero_derivative"]


def test_grid_broadcast_matches_flat():
    c = CASES["surface_o3x4"]
    s = Spline(c.nInd, c.nDep, c.order, c.nCoef, c.knots, c.coefs)
    dom = s.domain()
    u = np.linspace(dom[0][0], dom[0][1], 97)
    v = np.linspace(dom[1][0], dom[1][1], 131)
    for w in ([0, 0], [1, 0], [1, 2]):
        g = np.stack(s.derivative(w, u[:, None], v[None, :]))                     # grid path (> 4096 points)
        uu, vv = np.meshgrid(u, v, indexing="ij")
        f = np.stack(s.derivative(w, uu.ravel(), vv.ravel())).reshape(g.shape)     # flat path
        assert np.abs(g - f).max() <= 1e-12 * _scale(f)
    # transposed broadcast: variable 0 along the last axis
    g = np.stack(s(u[None, :], v[:, None]))
    uu, vv = np.meshgrid(u, v, indexing="xy")
    f = np.stack(s(uu.ravel(), vv.ravel())).reshape(g.shape)
    assert g.shape == (3, 131, 97) and np.abs(g - f).max() <= 1e-12 * _scale(f)
    # same-order surface uses the surface grid kernel; three variables the generic one
    for name in ("cfg2_bicubic", "volume_o3x4x2"):
        c = CASES[name]
        s = Spline(c.nInd, c.nDep, c.order, c.nCoef, c.knots, c.coefs)
        dom = s.domain()
        axes = [np.linspace(dom[i][0], dom[i][1], 19 + 4 * i) for i in range(c.nInd)]
        t = s.device_tables()
        g = t.evaluate_grid(axes)
        mesh = np.meshgrid(*axes, indexing="ij")
        f = t.evaluate([m.ravel() for m in mesh]).reshape(g.shape)
        assert np.abs(g - f).max() <= 1e-12 * _scale(f)
    # mixed orders on the row-factored grid kernel (vector and scalar store paths), order 7 on the generic one
    for name in ("surface_o2x6_d1", "surface_o1x4_d2", "surface_o3x4", "surface_f64knots_f32coefs", "surface_o7x3_d6"):
        c = CASES[name]
        t = _tables(c)
        dom = [(k[o - 1], k[nc]) for k, o, nc in zip(c.knots, c.order, c.nCoef)]
        for n1 in (128, 77):
            axes = [np.linspace(dom[0][0], dom[0][1], 48), np.linspace(dom[1][0], dom[1][1], n1)]
            for w in ([0, 0], [1, 1]):
                g = t.evaluate_grid(axes, w)
                mesh = np.meshgrid(*axes, indexing="ij")
                f = t.evaluate([m.ravel() for m in mesh], w).reshape(g.shape)
                assert np.abs(g - f).max() <= 1e-12 * _scale(f), (name, n1, w)
    # out-of-domain value on a grid: first offender in broadcast order
    u2 = u.copy()
    u2[5] = 99.0
    c = CASES["surface_o3x4"]
    s = Spline(c.nInd, c.nDep, c.order, c.nCoef, c.knots, c.coefs)
    with pytest.raises(ValueError, match="Spline evaluation outside domain"):
        s(u2[:, None], v[None, :])


def test_teapot_grid(golden_tables):
    """cfg4 shape: the 32 Utah-teapot patches (examples/teapot.py:4-358) on a broadcast grid, fp32."""
    t = golden_tables
    knots = np.array((0, 0, 0, 0, 1, 1, 1, 1), np.float32)
    g = np.linspace(0, 1, 16, dtype=np.float32)
    g2 = np.linspace(0, 1, 96, dtype=np.float32)
    V = t["teapot_vertices"]
    for pi, patch in enumerate(t["teapot_patch_index"]):
        c = np.empty((3, 4, 4), np.float32)
        for i in range(4):
            for j in range(4):
                v = V[patch[4 * i + j] - 1]
                c[0, i, j], c[1, i, j], c[2, i, j] = v[0], v[2], v[1]
        s = Spline(2, 3, (4, 4), (4, 4), (knots, knots), c)
        out = np.stack(s(g[:, None], g[None, :]))
        assert out.dtype == np.float32
        assert np.abs(out - t["teapot_grid16"][pi]).max() <= 2e-5 * _scale(t["teapot_grid16"][pi])
        # grid kernel (96 x 96 > threshold) agrees with the flat kernel on the same points
        big = np.stack(s(g2[:, None], g2[None, :]))
        uu, vv = np.meshgrid(g2, g2, indexing="ij")
        flat = np.stack(s(uu.ravel(), vv.ravel())).reshape(big.shape)
        assert np.abs(big - flat).max() <= 2e-5 * _scale(flat)


def test_constructor_forms_and_mutation(golden_api):
    k2 = [[0, 0, 0, .5, 1, 1, 1], [0, 0, 0, 0, .5, 1, 1, 1, 1]]
    flat = np.arange(60.0).reshape(20, 3)
    s3 = Spline(2, 3, [3, 4], [4, 5], k2, flat)                      # flat list-of-points form
    assert np.array_equal(np.ascontiguousarray(s3.coefs), np.array(golden_api["flat_coefs"]))
    ref = Spline(2, 3, [3, 4], [4, 5], k2, np.ascontiguousarray(s3.coefs))
    assert np.array_equal(s3([0.3, 0.6]), ref([0.3, 0.6]))           # non-contiguous coefs view is handled
    # in-place mutation of coefs must invalidate the device tables
    before = ref([0.3, 0.6]).copy()
    ref.coefs[:, 1, 2] += 10.0
    after = ref([0.3, 0.6])
    assert np.abs(after - before).max() > 1e-3
    fresh = Spline(2, 3, [3, 4], [4, 5], k2, ref.coefs.copy())
    assert np.array_equal(after, fresh([0.3, 0.6]))
    # swaps of round values in place (a checksum that is linear mod 2^64 misses them: 0.0 <-> 2.0, 1.0 <-> 2.0)
    sw = Spline(1, 2, [2], [4], [[0, 0, 1 / 3, 2 / 3, 1, 1]], np.array([[0.0, 2.0, 2.0, 0.0], [0.0, 0.0, 2.0, 1.0]]))
    a0 = sw(0.1).copy()
    sw.coefs[0, 0], sw.coefs[0, 1] = 2.0, 0.0
    a1 = sw(0.1).copy()
    assert np.abs(a1 - a0).max() > 0.5
    sw.coefs[1, 2], sw.coefs[1, 3] = 1.0, 2.0
    assert np.array_equal(sw(0.9), Spline(1, 2, [2], [4], sw.knots, sw.coefs.copy())(0.9)) and abs(sw(0.9)[1] - 1.7) < 1e-12
    sw.knots[0][2], sw.knots[0][3] = 0.25, 0.75                      # knots are mutable too
    assert np.array_equal(sw(0.5), Spline(1, 2, [2], [4], [sw.knots[0].copy()], sw.coefs.copy())(0.5))
    # integer inputs are promoted to float64 (documented deviation)
    si = Spline(1, 1, [4], [4], [[0, 0, 0, 0, 1, 1, 1, 1]], [[0, 1, 2, 3]])
    assert abs(si(0.5)[0] - 1.5) < 1e-14


def test_torch_device_path_matches_host_path():
    torch = pytest.importorskip("torch")
    c = CASES["cfg2_bicubic"]
    s = Spline(c.nInd, c.nDep, c.order, c.nCoef, c.knots, c.coefs)
    host = np.stack(s(*c.points))
    dev = s(*[torch.as_tensor(p, device="cuda") for p in c.points])
    assert isinstance(dev, tuple) and dev[0].is_cuda
    assert np.array_equal(torch.stack(dev).cpu().numpy(), host)      # same kernel, bitwise
    jh = s.jacobian(c.points)
    jd = s.jacobian([torch.as_tensor(p, device="cuda") for p in c.points])
    assert np.array_equal(jd.cpu().numpy(), jh)
    bad = [torch.as_tensor(p.copy(), device="cuda") for p in c.points]
    bad[1][17] = 5.0
    with pytest.raises(ValueError, match="Spline evaluation outside domain"):
        s(*bad)
    s(*[torch.as_tensor(p, device="cuda") for p in c.points])         # the record was reset


def test_full_size_properties():
    """BASELINE cfg2 size: 10 M random points on the bicubic 64x64x3 fp64 surface."""
    nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
    n = 10_000_000
    rng = np.random.default_rng(0)
    uv = rng.random((2, n))
    t = DeviceSpline(order, ncoef, knots, coefs, dt)
    out = t.evaluate([uv[0], uv[1]])
    # (a) oracle on a 200 k sample spread over the batch
    idx = rng.choice(n, 200_000, replace=False)
    orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0], [uv[0][idx], uv[1][idx]])
    assert bad == -1
    assert np.abs(out[:, idx] - orc).max() <= 1e-12 * _scale(orc)
    # (b) determinism: bitwise identical on a second run
    assert np.array_equal(out, t.evaluate([uv[0], uv[1]]))
    # (b') host batches of this size go through the pipelined staging path: the first offender is
    # still reported by its batch index, and the record is clean afterwards
    ub = uv[0].copy()
    ub[7_654_321] = 1.25
    ub[9_000_000] = -0.5
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate([ub, uv[1]])
    assert e.value.index == 7_654_321
    with pytest.raises(bspy_amd.DomainError) as e:
        t.jacobian([ub[5_000_000:], uv[1][5_000_000:]])
    assert e.value.index == 2_654_321
    assert np.array_equal(out, t.evaluate([uv[0], uv[1]]))
    reuse = np.empty_like(out)
    assert t.evaluate([uv[0], uv[1]], out=reuse) is reuse and np.array_equal(reuse, out)
    with pytest.raises(ValueError):
        t.evaluate([uv[0], uv[1]], out=np.empty((3, n), np.float32))
    # (b'') the batched normal of the whole batch (pipelined host path) against the oracle on the sample
    nrm = t.normal([uv[0], uv[1]])
    onrm, _ = oracle.c_normal(order, ncoef, knots, coefs, [uv[0][idx], uv[1][idx]], True, False)
    assert np.abs(nrm[:, idx] - onrm).max() <= 1e-10
    # (c) permutation equivariance, bitwise (no dependence on a point's position in the batch)
    perm = rng.permutation(n)
    assert np.array_equal(out[:, perm], t.evaluate([uv[0][perm], uv[1][perm]]))
    # (d) partition of unity: all coefficients 1 -> value 1, first derivatives 0
    ones = DeviceSpline(order, ncoef, knots, np.ones_like(coefs), dt)
    assert np.abs(ones.evaluate([uv[0], uv[1]]) - 1.0).max() <= 64 * EPS
    jac = ones.jacobian([uv[0][:1_000_000], uv[1][:1_000_000]])
    assert np.abs(jac).max() <= 1e-10
    # (e) linearity in the coefficients
    c2 = np.random.default_rng(1).standard_normal(coefs.shape)
    both = DeviceSpline(order, ncoef, knots, coefs + 2.0 * c2, dt).evaluate([uv[0], uv[1]])
    second = DeviceSpline(order, ncoef, knots, c2, dt).evaluate([uv[0], uv[1]])
    assert np.abs(both - (out + 2.0 * second)).max() <= 1e-12 * _scale(both)
    # (f) jacobian agrees with the two derivative calls at full size (cfg3)
    m = 2_000_000
    jac = t.jacobian([uv[0][:m], uv[1][:m]])
    d0 = t.evaluate([uv[0][:m], uv[1][:m]], [1, 0])
    d1 = t.evaluate([uv[0][:m], uv[1][:m]], [0, 1])
    assert np.abs(jac[:, 0] - d0).max() <= 1e-11 * _scale(d0)
    assert np.abs(jac[:, 1] - d1).max() <= 1e-11 * _scale(d1)


@pytest.mark.parametrize("n", [50_000_000, 6_250_000, 70_000_000])
def test_cfg5_full_size_properties(n):
    """BASELINE configs[4] at its real size: 50 M points on the trivariate order-5 40^3 x 4 fp32 spline, the 6.25 M-point
    shard one of 8 GPUs gets, and 70 M points - beyond the 2^26 destinations a record's tag holds, so the cell-order
    pipeline runs two pieces of 35 M.  Device-resident I/O.  Oracle on a 200 k sample spread over the batch (the end of it included), bitwise
    determinism, bitwise permutation equivariance, partition of unity, and the first offender near the END of the batch
    (32-bit slot / chunk arithmetic)."""
    torch = pytest.importorskip("torch")
    nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(5)
    t = DeviceSpline(order, ncoef, knots, coefs, dt)
    g = torch.Generator(device="cuda").manual_seed(n % 1000)
    p = [torch.rand(n, dtype=torch.float32, device="cuda", generator=g) for _ in range(3)]
    p[0][-3:] = 1.0                                              # the domain's right end, at the very end of the batch
    p[2][-1] = 0.0
    out = t.evaluate_device(p)
    assert "eval_cellsort" in t.last_kernel()
    rng = np.random.default_rng(3)
    idx = np.unique(np.concatenate([rng.integers(0, n, 200_000), np.arange(n - 4096, n), np.arange(4096)]))
    tidx = torch.as_tensor(idx, device="cuda")
    host = [x[tidx].cpu().numpy() for x in p]
    orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0, 0], host)
    assert bad == -1
    err = float(np.abs(out[:, tidx].cpu().numpy() - orc).max() / _scale(orc))
    print(f"cfg5 {n} points: worst error on {len(idx)} sampled points {err:.2e} of the scale")
    assert err <= 2e-5
    assert torch.equal(out, t.evaluate_device(p))                # determinism (the SLOT of a record is not deterministic, its result is)
    perm = torch.randperm(n, device="cuda", generator=g)
    outp = t.evaluate_device([x[perm] for x in p])
    assert torch.equal(outp, out[:, perm])                       # no dependence on a point's position in the batch
    del outp, perm
    d1 = t.evaluate_device(p, [0, 1, 0])                         # a derivative pass at full size, on the same sample
    od, _ = oracle.c_evaluate(order, ncoef, knots, coefs, [0, 1, 0], host)
    assert float(np.abs(d1[:, tidx].cpu().numpy() - od).max() / _scale(od)) <= 2e-5
    del d1
    ones = DeviceSpline(order, ncoef, knots, np.ones_like(coefs), dt)
    one = ones.evaluate_device(p)
    assert float((one - 1.0).abs().max()) <= 1e-5                # partition of unity
    del one
    p[1][n - 7] = 1.5                                            # first offender near the end; a later one behind it
    p[0][n - 2] = -0.25
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate_device(p)
    assert e.value.index == n - 7
    p[1][n - 7] = 0.5
    p[0][n - 2] = 0.5
    p[2][n // 2 + 11] = 1.0 + 1e-6                               # the third variable (tested by the scatter kernel), second piece
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate_device(p)
    assert e.value.index == n // 2 + 11


def test_empty_and_ragged():
    c = CASES["cfg2_bicubic"]
    t = _tables(c)
    assert t.evaluate([np.empty(0), np.empty(0)]).shape == (3, 0)
    assert t.jacobian([np.empty(0), np.empty(0)]).shape == (3, 2, 0)
    with pytest.raises(ValueError):
        t.evaluate([c.points[0], c.points[1][:-1]])
    with pytest.raises(ValueError, match="Incorrect number of parameter values: 1"):
        t.evaluate([c.points[0]])
    # every batch size around the workgroup/wave boundaries
    for n in (1, 63, 64, 65, 255, 256, 257, 1023, 1024):
        out = t.evaluate([c.points[0][:n], c.points[1][:n]])
        orc, _ = oracle.c_evaluate(c.order, c.nCoef, c.knots, c.coefs, [0, 0], [c.points[0][:n], c.points[1][:n]])
        assert np.abs(out - orc).max() <= 1e-12 * _scale(orc)


def test_pair_kernels_odd_sizes_views_and_domain_index():
    """The rowrot kernels give a lane two consecutive points and use 16-byte accesses: odd batch
    sizes (last lane holds one point), 8-byte-misaligned device views, a jacobian / normal of the
    same batches, and the first-offender index in either slot of a pair."""
    torch = pytest.importorskip("torch")
    c = CASES["cfg2_bicubic"]
    t = _tables(c)
    u, v = c.points[0], c.points[1]
    for n in (1, 2, 3, 5, 127, 128, 129, 2049):
        pts = [u[:n], v[:n]]
        for w in ([0, 0], [1, 0], [0, 2]):
            orc, _ = oracle.c_evaluate(c.order, c.nCoef, c.knots, c.coefs, w, pts)
            assert np.abs(t.evaluate(pts, w) - orc).max() <= 1e-11 * _scale(orc), (n, w)
        jo, _ = oracle.c_jacobian(c.order, c.nCoef, c.knots, c.coefs, pts)
        assert np.abs(t.jacobian(pts) - jo).max() <= 1e-11 * _scale(jo), n
        no, _ = oracle.c_normal(c.order, c.nCoef, c.knots, c.coefs, pts, True, False)
        assert np.abs(t.normal(pts) - no).max() <= 1e-10, n
    # device views that start 8 bytes into a 16-byte line, and an output view likewise
    du = torch.as_tensor(np.concatenate([[0.5], u]), device="cuda")[1:]
    dv = torch.as_tensor(np.concatenate([[0.5], v, [0.5]]), device="cuda")[1:-1]
    ref = t.evaluate([u, v])
    got = t.evaluate_device([du, dv])
    assert np.array_equal(got.cpu().numpy(), ref)
    buf = torch.empty(3 * u.size + 1, dtype=torch.float64, device="cuda")
    got = t.evaluate_device([du, dv], out=buf[1:].view(3, u.size))
    assert np.array_equal(got.cpu().numpy(), ref)
    assert np.array_equal(t.jacobian_device([du, dv]).cpu().numpy(), t.jacobian([u, v]))
    # the first offender is reported whichever slot of its pair it sits in
    for badi in (10, 11, u.size - 1):
        ub = u.copy()
        ub[badi] = 1.5
        ub[min(badi + 7, u.size - 1)] = -3.0
        with pytest.raises(bspy_amd.DomainError) as e:
            t.evaluate([ub, v])
        assert e.value.index == badi
        with pytest.raises(bspy_amd.DomainError) as e:
            t.jacobian([ub[:-1], v[:-1]] if badi < u.size - 1 else [ub, v])
        assert e.value.index == badi


def test_launch_and_host_chunk_loops():
    """The rowrot kernels index points with 32 bits and the launcher cuts batches at 2^28 points;
    BSK_RR_CHUNK lowers the cut so the chunk loop (pointer offsets, row strides, first-offender
    index across chunks) runs on a small batch.  Separate process: the limit is read once."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import cases, bspy_amd
        c = {x.name: x for x in cases.parity_cases()}["cfg2_bicubic"]
        t = bspy_amd.DeviceSpline(c.order, c.nCoef, c.knots, c.coefs)
        rng = np.random.default_rng(5)
        n = 10_007
        u, v = rng.random(n), rng.random(n)
        np.save(sys.argv[1], np.concatenate([t.evaluate([u, v]).ravel(), t.evaluate([u, v], [1, 1]).ravel(),
                                             t.jacobian([u, v]).ravel(), t.normal([u, v]).ravel()]))
        ub = u.copy(); ub[7_777] = 4.0
        try:
            t.evaluate([ub, v]); print("no error")
        except bspy_amd.DomainError as e:
            print("bad", e.index)
        # staged host path (more points than the zero-copy path takes): chunk loop of BSK_HOST_CHUNK
        m = 150_001
        U, V = rng.random(m), rng.random(m)
        c4 = {x.name: x for x in cases.parity_cases()}["surface_o3x4"]
        t4 = bspy_amd.DeviceSpline(c4.order, c4.nCoef, c4.knots, c4.coefs)
        dom = [(k[o - 1], k[nc]) for k, o, nc in zip(c4.knots, c4.order, c4.nCoef)]
        P = [lo + (hi - lo) * rng.random(m) for lo, hi in dom]
        np.save(sys.argv[1] + ".host.npy", np.concatenate([t.evaluate([U, V]).ravel(), t.jacobian([U, V]).ravel(), t.normal([U, V]).ravel(),
                                                            t.curvature([U[:70_000], V[:70_000]]).ravel()]))
        np.save(sys.argv[1] + ".host4.npy", t4.evaluate(P, [1, 0]))
        Ub = U.copy(); Ub[123_456] = -1.0
        try:
            t.jacobian([Ub, V]); print("no error")
        except bspy_amd.DomainError as e:
            print("hostbad", e.index)
    """) % (ROOT, os.path.join(ROOT, "tests"))
    outs = []
    for env_extra in ({}, {"BSK_RR_CHUNK": "4096", "BSK_HOST_CHUNK": "40000", "BSK_SMALL_POINTS": "1000"}):
        f = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rr_chunk_{len(outs)}.npy")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env_extra), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "bad 7777" in r.stdout and "hostbad 123456" in r.stdout, r.stdout
        outs.append((np.load(f), np.load(f + ".host.npy"), np.load(f + ".host4.npy")))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1], equal_nan=True)
    # the mixed-order surface (batches of >= 2^20 points would run eval_slab2 in one slab; these are on eval_mixed either way)
    observe("mixed-order surface, chunked vs whole host batch", np.abs(outs[0][2] - outs[1][2]).max() / _scale(outs[0][2]), 1e-13)


@pytest.mark.parametrize("variant", ["1", "4", "9"])
def test_kernel_variants(variant, golden_parity, monkeypatch):
    """BSK_VARIANT pins the kernel family (1 eval_fixed, 4 eval_stream, 9 eval_rowrot where it
    applies); every family must meet the same parity bar on its own."""
    monkeypatch.setenv("BSK_VARIANT", variant)
    for name in ("cfg1_curve", "cfg2_bicubic", "cfg2_bicubic_nonuniform", "bezier_patch_f32", "curve_order5",
                 "volume_o4_d1", "surface_o7x3_d6", "curve_f32", "curve_many_knots"):
        c = CASES[name]
        t = _tables(c)
        for w in c.wrts:
            ref = golden_parity[f"{name}/wrt_" + "_".join(map(str, w))]
            out = t.evaluate(c.points, list(w))
            assert np.abs(out - ref).max() <= _tol(c) * _scale(ref), (variant, name, w)
    # a multi-tile batch with a ragged tail, against the oracle
    c = CASES["cfg2_bicubic_nonuniform"]
    rng = np.random.default_rng(5)
    n = 7 * 1024 + 333
    dom = [(k[o - 1], k[nc]) for k, o, nc in zip(c.knots, c.order, c.nCoef)]
    pts = [lo + (hi - lo) * rng.random(n) for lo, hi in dom]
    t = _tables(c)
    for w in ([0, 0], [1, 2]):
        out = t.evaluate(pts, w)
        orc, _ = oracle.c_evaluate(c.order, c.nCoef, c.knots, c.coefs, w, pts)
        assert np.abs(out - orc).max() <= 1e-12 * _scale(orc), (variant, w)
    bad = [p.copy() for p in pts]
    bad[0][5000] = dom[0][1] + 1.0
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate(bad)
    assert e.value.index == 5000


@pytest.mark.parametrize("shape", [
    (2, 2, (4, 4), (200, 150), np.float64),       # 480 KB surface table, 29 k cells -> coarsened to <= 8192
    (2, 4, (3, 3), (300, 300), np.float32),       # nDep 4 fp32: 16-byte records and results
    (3, 1, (3, 3, 3), (50, 40, 30), np.float64),  # fp64 volume: 32-byte records
    (3, 4, (5, 5, 5), (40, 40, 40), np.float32),  # cfg5 shape
    (3, 3, (4, 4, 4), (36, 30, 44), np.float64),  # nDep 3 fp64
    (3, 2, (3, 3, 3), (60, 60, 60), np.float32),  # nDep 2 fp32: two of the four MFMA rows are zero
    (2, 3, (4, 5), (900, 11), np.float64),        # mixed orders: the shape of the reference's examples/TomsNasty.json
    (3, 2, (3, 5, 2), (40, 50, 30), np.float64),  # mixed orders, three variables
    (2, 4, (2, 6), (400, 300), np.float32),       # orders 2 and 6
])
def test_large_batches_in_cell_order(shape, monkeypatch):
    """Batches of >= 2^18 points on L2-resident tables are counting-sorted by the cell of the first
    two variables and evaluated in cell order (bsk_binned.hpp).  Same arithmetic as the gather
    kernel: results are bitwise those of BSK_VARIANT=7 (cell order off), for every derivative, and
    match the oracle; the first out-of-domain point is still reported by its batch index."""
    nind, ndep, order, ncoef, dt = shape
    rng = np.random.default_rng(77)
    knots = [cases.nonuniform_knots(rng, o, c, dt, -1.0, 2.0) for o, c in zip(order, ncoef)]
    coefs = rng.standard_normal((ndep, *ncoef)).astype(dt)
    n = 600_007 if nind == 2 else 300_007                         # (surfaces: above eval_slab2's 2^19-point threshold)
    pts = [(-1.0 + 3.0 * rng.random(n)).astype(dt) for _ in range(nind)]
    # clustered points too: one cell takes most of the batch
    pts[0][: n // 2] = dt(0.123)
    pts[1][: n // 3] = dt(1.5)
    tol = 2e-5 if dt == np.float32 else 1e-12
    t = DeviceSpline(order, ncoef, knots, coefs, dt)
    monkeypatch.setenv("BSK_VARIANT", "7")
    plain = DeviceSpline(order, ncoef, knots, coefs, dt)
    sample = rng.choice(n, 20_000, replace=False)
    # three variables of one order are grouped a second time inside the evaluation workgroup and contracted
    # with a shared coefficient operand (eval_cellsort: MFMA for fp32): another summation order, so those
    # shapes agree with the gather kernel to rounding; the others bitwise
    regrouped = nind == 3 and len(set(order)) == 1
    for w in ([0] * nind, [1] + [0] * (nind - 1), [0] * (nind - 1) + [2]):
        out = t.evaluate(pts, w)
        assert ("eval_cellsort" in t.last_kernel()) == regrouped, t.last_kernel()
        ref_plain = plain.evaluate(pts, w)
        kind = "fp32" if dt == np.float32 else "fp64"
        if regrouped:
            observe(f"cell order vs gather kernel, regrouped, {kind}", np.abs(out - ref_plain).max() / _scale(ref_plain), tol)
        elif t.last_kernel() == "eval_slab2":
            # surfaces streamed through LDS (bsk_slab.hpp): the gather kernel's operations in the same order, but its
            # table reads are explicit and hipcc contracts multiply-adds around them differently: a few ulp
            observe(f"eval_slab2 vs gather kernel, {kind}", np.abs(out - ref_plain).max() / _scale(ref_plain), 1e-5 if dt == np.float32 else 1e-13)
        else:
            assert np.array_equal(out, ref_plain), (shape, w)
        orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, w, [p[sample] for p in pts])
        assert bad == -1
        observe(f"cell order vs oracle, evaluate / derivative, {kind}", np.abs(out[:, sample] - orc).max() / _scale(orc), tol)
    assert np.array_equal(out, t.evaluate(pts, w))                       # second run: same bits
    # the jacobian of such a batch = nInd derivative passes through the same pipeline
    jac = t.jacobian(pts)
    ojac, _ = oracle.c_jacobian(order, ncoef, knots, coefs, [p[sample] for p in pts])
    observe(f"cell order vs oracle, jacobian, {kind}", np.abs(jac[:, :, sample] - ojac).max() / _scale(ojac), tol)
    bad = [p.copy() for p in pts]
    bad[1][123_456] = dt(7.0)
    bad[0][250_000] = dt(-9.0)
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate(bad)
    assert e.value.index == 123_456
    # the last variable is tested by the scatter kernel (the count kernel reads only the two that make the bin)
    # NaN parameters (the reference's searchsorted puts them past the last span and returns NaN, no error): NaN at those
    # points only - their bin / span keys are clamped inside the sort kernels and nobody else's slot moves
    nan = [p.copy() for p in pts]
    where = [(iv, 1_000 + 77_777 * iv) for iv in range(nind)]
    for iv, i in where:
        nan[iv][i] = np.nan
    clean = t.evaluate(pts)
    out = t.evaluate(nan)
    hit = np.zeros(n, bool)
    hit[[i for _, i in where]] = True
    assert np.isnan(out[:, hit]).all()
    assert np.array_equal(out[:, ~hit], clean[:, ~hit])
    # points on every knot of every variable and one ulp either side, both ends of the domain included
    edge = [p.copy() for p in pts]
    for iv, k in enumerate(knots):
        lo, hi = k[order[iv] - 1], k[ncoef[iv]]
        d = np.unique(k)
        e = np.concatenate((d, np.nextafter(d, dt(-np.inf)), np.nextafter(d, dt(np.inf)))).astype(dt)
        e = e[(e >= lo) & (e <= hi)]
        edge[iv][10_000 * (iv + 1): 10_000 * (iv + 1) + len(e)] = e
    out = t.evaluate(edge)
    orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, [0] * nind, [p[:40_000] for p in edge])
    assert bad == -1
    observe(f"cell order vs oracle, points on knots, {kind}", np.abs(out[:, :40_000] - orc).max() / _scale(orc), tol)
    bad = [p.copy() for p in pts]
    bad[-1][77_777] = np.nextafter(dt(knots[-1][ncoef[-1]]), dt(np.inf))
    bad[0][250_000] = dt(-9.0)
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate(bad)
    assert e.value.index == 77_777


@pytest.mark.parametrize("variant,kernel", [("0", "eval_cellsort, MFMA"), ("12", "eval_cellsort, VALU"), ("13", "eval_binned_lds"),
                                            ("14", "eval_binned_lds"), ("7", "eval_gather")])
def test_cell_order_pipeline_variants(variant, kernel, monkeypatch):
    """Every form of the large-table path on the cfg5 shape (three variables, order 5, 40^3 x 4, fp32) against the
    oracle: MFMA and VALU contraction of eval_cellsort (round-3 sort: bin_totals / bin_scatter_tag / bin_unpermute_stream),
    round 1's eval_binned_lds behind the round-2 sort (chunk histograms + scans; 13), the same with the direct instead of
    the write-combining scatter / un-permute (14), and the gather kernel in batch order (7)."""
    monkeypatch.setenv("BSK_VARIANT", variant)
    nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(5)
    t = DeviceSpline(order, ncoef, knots, coefs, dt)
    rng = np.random.default_rng(int(variant) + 1)
    n = 400_003
    pts = [rng.random(n).astype(dt) for _ in range(3)]
    pts[2][:1000] = dt(1.0)                                  # right end of the domain, and a crowded cell
    pts[0][1000:2000] = dt(0.0)
    sample = rng.choice(n, 30_000, replace=False)
    sample[:50] = np.arange(50)
    for w in ([0, 0, 0], [1, 0, 2]):
        out = t.evaluate(pts, w)
        assert kernel in t.last_kernel(), t.last_kernel()
        orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, w, [p[sample] for p in pts])
        assert bad == -1
        assert np.abs(out[:, sample] - orc).max() <= 2e-5 * _scale(orc), (variant, w)


@pytest.mark.parametrize("shape", [
    (1, 3, (4,), (50_000,), np.float64),          # 1.2 MB curve table
    (2, 2, (4, 4), (200, 150), np.float64),       # 480 KB surface table
    (2, 4, (3, 3), (300, 300), np.float32),       # 1.4 MB, nDep 4 (one 16-byte load per control point)
    (3, 1, (3, 3, 3), (50, 40, 30), np.float64),  # 480 KB volume, nDep 1
    (3, 4, (5, 5, 5), (40, 40, 40), np.float32),  # cfg5 shape
    (2, 3, (4, 5), (900, 11), np.float64),        # mixed orders (TomsNasty shape): right-aligned windows at the largest order
    (3, 2, (3, 5, 2), (40, 50, 30), np.float64),
    (2, 1, (1, 4), (5000, 40), np.float64),       # order 1 beside order 4
])
def test_large_tables_gathered_from_l2(shape):
    """Tables that do not fit in LDS run on the control-point-major gather kernel."""
    nind, ndep, order, ncoef, dt = shape
    rng = np.random.default_rng(42)
    knots = [cases.nonuniform_knots(rng, o, c, dt, -1.0, 2.0) for o, c in zip(order, ncoef)]
    coefs = rng.standard_normal((ndep, *ncoef)).astype(dt)
    n = 20_000
    pts = [(-1.0 + 3.0 * rng.random(n)).astype(dt) for _ in range(nind)]
    t = DeviceSpline(order, ncoef, knots, coefs, dt)
    tol = 2e-5 if dt == np.float32 else 1e-12
    for w in ([0] * nind, [1] + [0] * (nind - 1), [0] * (nind - 1) + [2]):
        out = t.evaluate(pts, w)
        orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, w, pts)
        assert bad == -1
        assert np.abs(out - orc).max() <= tol * _scale(orc), (shape, w)
    # update() must refresh the control-point-major copy too
    coefs2 = coefs * 0.5
    t.update(knots, coefs2)
    out = t.evaluate(pts)
    orc, _ = oracle.c_evaluate(order, ncoef, knots, coefs2, [0] * nind, pts)
    assert np.abs(out - orc).max() <= tol * _scale(orc)


def test_reference_json_files_evaluate():
    """Splines from the reference's own JSON fixtures (order 7 curves with 112 / 140
    coefficients, a five-curve file) evaluate on the GPU like the oracle."""
    import os
    ref_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_json")
    for name in ("reverse-thing.json", "trim-issue.json", "offset-issue.json"):
        for s in Spline.load(os.path.join(ref_dir, name)):
            dom = s.domain()
            u = np.linspace(dom[0][0], dom[0][1], 2001)
            knots = [np.asarray(k, np.float64) for k in s.knots]
            coefs = np.ascontiguousarray(s.coefs, np.float64)
            for w in ([0], [1], [2]):
                got = s.derivative(w, u) if w[0] else s(u)
                got = np.stack(got) if isinstance(got, tuple) else np.asarray(got)[None, :]
                orc, bad = oracle.c_evaluate(s.order, s.nCoef, knots, coefs, w, [u])
                assert bad == -1
                if w[0] < 2:
                    assert np.abs(got - orc).max() <= 1e-11 * _scale(orc), (name, w)
                    continue
                # Second derivatives of these order-7 curves (knot spacing down to 6e-5) are ill-conditioned: fp64
                # rounding alone moves them by ~1e-8 of their scale.  Measured, not assumed: the reference's
                # algorithm restated in extended precision (np.longdouble) is the yardstick, and the GPU result
                # must be as close to it as the fp64 oracle (= the reference's arithmetic) is, within a factor 4.
                ext = _curve_derivative_longdouble(s.order[0], knots[0], coefs, w[0], u)
                scale = _scale(orc)
                d_orc = float(np.abs(orc - ext).max()) / scale
                d_gpu = float(np.abs(got - ext).max()) / scale
                assert d_orc <= 1e-7, (name, d_orc)                       # the yardstick itself is sane
                assert d_gpu <= max(4.0 * d_orc, 1e-11), (name, w, d_gpu, d_orc)


def _curve_derivative_longdouble(order, knots, coefs, deriv, us):
    """Reference bspy/_spline_evaluation.py:4-27 + :109-133 for a curve, every operation in np.longdouble."""
    L = np.longdouble
    k = knots.astype(L)
    c = coefs.astype(L)
    out = np.zeros((c.shape[0], len(us)), L)
    ncoef = len(k) - order
    for n, uf in enumerate(us):
        u = L(uf)
        ix = int(np.searchsorted(knots, uf, side="right"))
        ix = min(max(ix, order), ncoef)
        b = np.zeros(order, L)
        if deriv < order:
            b[-1] = 1
            for degree in range(1, order - deriv):
                bi = order - degree
                for i in range(ix - degree, ix):
                    alpha = (u - k[i]) / (k[i + degree] - k[i])
                    b[bi - 1] += (1 - alpha) * b[bi]
                    b[bi] *= alpha
                    bi += 1
            for degree in range(order - deriv, order):
                bi = order - degree
                for i in range(ix - degree, ix):
                    alpha = L(degree) / (k[i + degree] - k[i])
                    b[bi - 1] += -alpha * b[bi]
                    b[bi] *= alpha
                    bi += 1
        out[:, n] = c[:, ix - order:ix] @ b
    return out


NORMAL_CASES = [n for n, c in CASES.items() if abs(c.nInd - c.nDep) == 1 and max(c.nInd, c.nDep) <= 4]


@pytest.mark.parametrize("name", sorted(NORMAL_CASES))
def test_normal_against_reference(name, golden_parity):
    """Batched Spline.normal against the reference's single-point normal."""
    c = CASES[name]
    m = golden_parity[f"{name}/normal_unit"].shape[0]
    pts = [p[:m] for p in c.points]
    tol = 5e-5 if (c.knots[0].dtype == np.float32 or c.coefs.dtype == np.float32) else 1e-11
    for key, normalize, meta in (("normal_unit", True, {}), ("normal_area", False, {}),
                                 ("normal_area_negated", False, {"negateNormal": True})):
        s = Spline(c.nInd, c.nDep, c.order, c.nCoef, c.knots, c.coefs, meta)
        ref = golden_parity[f"{name}/{key}"].T
        out = s.normal(pts, normalize)
        assert out.shape == ref.shape
        assert np.array_equal(np.isnan(out), np.isnan(ref)), (name, key)
        if not np.isnan(ref).all():
            observe(f"normal vs reference, {'fp32' if tol > 1e-9 else 'fp64'}", np.nanmax(np.abs(out - ref)) / max(1.0, float(np.nanmax(np.abs(ref)))), tol)
        orc, bad = oracle.c_normal(c.order, c.nCoef, c.knots, c.coefs, pts, normalize, bool(meta))
        if not np.isnan(orc).all():
            observe(f"normal vs oracle, {'fp32' if tol > 1e-9 else 'fp64'}", np.nanmax(np.abs(out - orc)) / max(1.0, float(np.nanmax(np.abs(orc)))), tol)
    if c.order[0] > 1:
        s = Spline(c.nInd, c.nDep, c.order, c.nCoef, c.knots, c.coefs)
        one = s.normal([float(p[3]) for p in c.points])               # single point, reference call style
        assert one.shape == (max(c.nInd, c.nDep),)
        assert np.abs(one - golden_parity[f"{name}/normal_unit"][3]).max() <= tol
        area = s.normal([float(p[3]) for p in c.points], False)
        assert np.array_equal(s.normal([float(p[3]) for p in c.points], False, indices=(1, 0)), area[[1, 0]])
    with pytest.raises(ValueError, match="one different"):
        Spline(2, 2, [2, 2], [2, 2], [[0, 0, 1, 1.0]] * 2, np.zeros((2, 2, 2))).normal([0.5, 0.5])


def test_collocation_matrix():
    """Batched collocation matrix against the reference's row loop
    (bspy/_spline_fitting.py:736-751) restated with the oracle, Hermite rows included."""
    rng = np.random.default_rng(9)
    for order, ncoef, dt in ((4, 12, np.float64), (6, 30, np.float64), (3, 9, np.float32)):
        knots = cases.nonuniform_knots(rng, order, ncoef, dt, 0.0, 1.0)
        u = np.sort(rng.random(40).astype(dt))
        u = np.concatenate(([knots[order - 1]] * 2, u, u[5:7], u[5:7], [knots[ncoef]] * 3)).astype(dt)
        u = np.sort(u, kind="stable")
        A = bspy_amd.collocation_matrix(knots, order, u)
        ref = np.zeros((len(u), ncoef), dt)
        prev, deriv = None, 0
        for i, x in enumerate(u):
            deriv = deriv + 1 if (prev is not None and x == prev) else 0
            prev = x
            ix, row = oracle.c_bspline_values(None, knots, order, x, deriv)
            ref[i, ix - order:ix] = row
        assert A.shape == ref.shape and A.dtype == ref.dtype
        assert np.abs(A - ref).max() <= (1e-4 if dt == np.float32 else 1e-11) * max(1.0, np.abs(ref).max())
        first, rows = bspy_amd.collocation_matrix(knots, order, u, dense=False)
        assert np.array_equal(A[np.arange(len(u))[:, None], first[:, None] + np.arange(order)], rows)


def test_hip_graph_capture_of_device_calls():
    """BSK_DEVICE calls only enqueue kernels on the given stream (no allocation, no sync after
    the first call), so they can be captured into a HIP graph and replayed."""
    torch = pytest.importorskip("torch")
    c = CASES["cfg2_bicubic"]
    s = Spline(c.nInd, c.nDep, c.order, c.nCoef, c.knots, c.coefs)
    t = s.device_tables()
    u = torch.as_tensor(c.points[0], device="cuda")
    v = torch.as_tensor(c.points[1], device="cuda")
    out = torch.empty((3, u.numel()), dtype=torch.float64, device="cuda")
    jac = torch.empty((3, 2, u.numel()), dtype=torch.float64, device="cuda")
    t.evaluate_device([u, v], out=out, check=False)            # warm-up outside the capture
    t.jacobian_device([u, v], out=jac, check=False)
    torch.cuda.synchronize()
    expect, expect_j = out.clone(), jac.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        t.evaluate_device([u, v], out=out, check=False)
        t.jacobian_device([u, v], out=jac, check=False)
    out.zero_(); jac.zero_()
    u.copy_(torch.as_tensor(c.points[0][::-1].copy(), device="cuda"))      # new inputs, same buffers
    v.copy_(torch.as_tensor(c.points[1][::-1].copy(), device="cuda"))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, expect.flip(1)) and torch.equal(jac, expect_j.flip(2))
    t.domain_status()
    # normal and curvature: fused kernels (surface in 3-D on the LDS image) and the workspace forms (a curve's
    # curvature = two derivative passes + epilogue in a handle workspace that the warm-up call sized)
    nrm = torch.empty((3, u.numel()), dtype=torch.float64, device="cuda")
    crv = torch.empty((u.numel(),), dtype=torch.float64, device="cuda")
    cc = next(x for _, x in sorted(CASES.items()) if x.nInd == 1 and x.nDep >= 2 and x.order[0] >= 3 and x.knots[0].dtype == np.float64
              and x.coefs.dtype == np.float64)
    tc = Spline(cc.nInd, cc.nDep, cc.order, cc.nCoef, cc.knots, cc.coefs).device_tables()
    w = torch.as_tensor(np.asarray(cc.points[0], np.float64), device="cuda")
    ccrv = torch.empty((w.numel(),), dtype=torch.float64, device="cuda")
    t.normal_device([u, v], out=nrm, check=False)
    t.curvature_device([u, v], out=crv, check=False)
    tc.curvature_device([w], out=ccrv, check=False)
    torch.cuda.synchronize()
    e_n, e_c, e_cc = nrm.clone(), crv.clone(), ccrv.clone()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        t.normal_device([u, v], out=nrm, check=False)
        t.curvature_device([u, v], out=crv, check=False)
        tc.curvature_device([w], out=ccrv, check=False)
    nrm.zero_(); crv.zero_(); ccrv.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(nrm, e_n) and torch.equal(crv, e_c) and torch.isfinite(e_cc).any()
    assert torch.allclose(ccrv, e_cc, rtol=0, atol=0, equal_nan=True)
    # a large-table spline: the cell-order pipeline sizes a workspace per batch and therefore declines under
    # capture; the call falls back to the gather kernel and stays capturable
    rng = np.random.default_rng(3)
    kl = [cases.clamped_uniform_knots(3, 300), cases.clamped_uniform_knots(3, 300)]
    big = DeviceSpline((3, 3), (300, 300), kl, rng.standard_normal((2, 300, 300)))
    n = 1 << 18
    bu, bv = torch.rand(n, dtype=torch.float64, device="cuda"), torch.rand(n, dtype=torch.float64, device="cuda")
    bo = torch.empty((2, n), dtype=torch.float64, device="cuda")
    big.evaluate_device([bu, bv], out=bo, check=False)
    torch.cuda.synchronize()
    e_b = bo.clone()
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        big.evaluate_device([bu, bv], out=bo, check=False)
    bo.zero_()
    g3.replay()
    torch.cuda.synchronize()
    assert torch.equal(bo, e_b)
    # a workspace that would have to grow inside a capture is refused with a clear message (last: the failed
    # capture is abandoned)
    fresh = DeviceSpline(cc.order, cc.nCoef, cc.knots, cc.coefs)
    g0 = torch.cuda.CUDAGraph()
    with pytest.raises(bspy_amd.BskError, match="outside the capture"):
        with torch.cuda.graph(g0):
            fresh.curvature_device([w], out=ccrv, check=False)
    torch.cuda.synchronize()


CURVATURE_CASES = [n for n, c in CASES.items() if (c.nInd == 1 and c.nDep >= 2) or (c.nInd == 2 and c.nDep == 3)]


@pytest.mark.parametrize("name", sorted(CURVATURE_CASES))
def test_curvature_against_reference(name, golden_parity):
    """Batched Spline.curvature against the reference's single-point curvature."""
    c = CASES[name]
    ref = golden_parity[f"{name}/curvature"]
    pts = [p[:len(ref)] for p in c.points]
    s = Spline(c.nInd, c.nDep, c.order, c.nCoef, c.knots, c.coefs)
    out = np.asarray(s.curvature(pts), np.float64)
    assert out.shape == ref.shape
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    ok = np.isfinite(ref)
    f32 = c.knots[0].dtype == np.float32 or c.coefs.dtype == np.float32
    if ok.any():
        kind = "fp32" if f32 else "fp64"
        err = np.abs(out[ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok]))
        observe(f"curvature vs reference, {kind} [{name}]", err.max(), 5e-4 if f32 else 1e-11)
        orc, _ = oracle.c_curvature(c.order, c.nCoef, c.knots, c.coefs, pts)
        err = np.abs(out[ok] - orc[ok]) / np.maximum(1.0, np.abs(orc[ok]))
        observe(f"curvature vs oracle, {kind} [{name}]", err.max(), 1e-4 if f32 else 1e-11)
        one = s.curvature([float(p[5]) for p in c.points])            # single point
        observe(f"curvature, single point vs reference, {kind}", abs(one - ref[5]) / max(1.0, abs(ref[5])), 2e-5 if f32 else 1e-11)


def test_reference_curvature_pin(golden_tables):
    """reference test_curvature: Gaussian curvature of mySurface at (0.25, 0.5) is 1.024
    (tests/bspy_test.py:699-700)."""
    t = golden_tables
    surf = Spline(2, 3, [3, 4], [4, 5], [t["surface_knots0"], t["surface_knots1"]], t["surface_coefs"])
    assert abs(surf.curvature([0.25, 0.5]) - 1.024) < 1e-13


# ---------------------------------------------------------------------------------------------
# SplineBlock evaluation path (reference bspy/spline_block.py:179-247)
# ---------------------------------------------------------------------------------------------
def _block_oracle(c, kind, wrt=None, m=None):
    pts = [p[:m] for p in c.points]
    n = len(pts[0])
    out = np.zeros((c.nDep, c.nInd, n) if kind == "jacobian" else (c.nDep, n))
    r0 = 0
    for row in c.rows:
        k = row[0][1][1]
        for imap, (nind, ndep, order, ncoef, knots, coefs) in row:
            sub = [pts[i] for i in imap]
            if kind == "jacobian":
                j, _ = oracle.c_jacobian(order, ncoef, knots, coefs.astype(np.float64), sub)
                out[r0:r0 + k, imap] += j
            else:
                w = [0] * nind if wrt is None else [wrt[i] for i in imap]
                v, _ = oracle.c_evaluate(order, ncoef, knots, coefs.astype(np.float64), w, sub)
                out[r0:r0 + k] += v
        r0 += k
    return out


@pytest.mark.parametrize("case", cases.block_cases(), ids=lambda c: c.name)
def test_spline_block_against_reference(case):
    torch = pytest.importorskip("torch")
    c = case
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "block.npz"))
    blk = bspy_amd.SplineBlock([[(imap, Spline(*d)) for (imap, d) in row] for row in c.rows])
    f32 = "f32" in c.name
    tol = 2e-5 if f32 else 1e-12
    m = g[f"{c.name}/evaluate"].shape[1]
    pts = np.array([p[:m] for p in c.points])
    # the reference's single-point API, on a few points (every call is a launch per spline)
    for i in (0, 1, m - 1):
        v = blk.evaluate(pts[:, i])
        assert v.shape == (c.nDep,) and str(v.dtype) == str(g[f"{c.name}/dtype"])
        ref = g[f"{c.name}/evaluate"][:, i]
        assert np.abs(v - ref).max() <= tol * max(1.0, np.abs(ref).max())
        assert np.array_equal(blk(pts[:, i]), v)
        j = blk.jacobian(pts[:, i])
        ref = g[f"{c.name}/jacobian"][:, :, i]
        assert j.shape == (c.nDep, c.nInd) and np.abs(j - ref).max() <= 10 * tol * max(1.0, np.abs(ref).max())
        w = c.wrts[-1]
        ref = g[f"{c.name}/wrt_" + "_".join(map(str, w))][:, i]
        assert np.abs(blk.derivative(w, pts[:, i]) - ref).max() <= 100 * tol * max(1.0, np.abs(ref).max())
    # batched: goldens on the first m points, the oracle composition on the whole batch
    full = blk.evaluate(c.points)
    assert full.shape == (c.nDep, len(c.points[0]))
    ref = g[f"{c.name}/evaluate"]
    assert np.abs(full[:, :m] - ref).max() <= tol * max(1.0, np.abs(ref).max())
    orc = _block_oracle(c, "evaluate")
    kind = "fp32" if f32 else "fp64"
    observe(f"SplineBlock evaluate vs oracle, {kind}", np.abs(full - orc).max() / max(1.0, np.abs(orc).max()), tol)
    for w in c.wrts:
        got = blk.derivative(w, c.points)
        ref = g[f"{c.name}/wrt_" + "_".join(map(str, w))]
        observe(f"SplineBlock derivative vs reference, {kind}", np.abs(got[:, :m] - ref).max() / max(1.0, np.abs(ref).max()), 100 * tol)
        orc = _block_oracle(c, "evaluate", w)
        observe(f"SplineBlock derivative vs oracle, {kind}", np.abs(got - orc).max() / max(1.0, np.abs(orc).max()), 100 * tol)
    jac = blk.jacobian(c.points)
    ref = g[f"{c.name}/jacobian"]
    assert jac.shape == (c.nDep, c.nInd, len(c.points[0]))
    observe(f"SplineBlock jacobian vs reference, {kind}", np.abs(jac[:, :, :m] - ref).max() / max(1.0, np.abs(ref).max()), 10 * tol)
    orc = _block_oracle(c, "jacobian")
    observe(f"SplineBlock jacobian vs oracle, {kind}", np.abs(jac - orc).max() / max(1.0, np.abs(orc).max()), 10 * tol)
    # CUDA tensors in -> tensors out, same numbers; broadcasting shapes are kept
    dev = [torch.as_tensor(p, device="cuda") for p in c.points]
    td = blk.evaluate(dev)
    assert td.is_cuda and np.array_equal(td.cpu().numpy(), full)
    assert np.array_equal(blk.jacobian(dev).cpu().numpy(), jac)
    two = blk.evaluate([p[:6].reshape(2, 3) for p in c.points])
    assert two.shape == (c.nDep, 2, 3) and np.array_equal(two.reshape(c.nDep, 6), full[:, :6])
    # out-of-domain point: the reference's message names the mapped parameters of the failing spline
    bad = [p.copy() for p in c.points]
    bad[0][5] = 1e3
    with pytest.raises(ValueError, match="Spline evaluation outside domain"):
        blk.evaluate(bad)


# ---------------------------------------------------------------------------------------------
# batched tessellation: positions + normals of many patches from one launch
# ---------------------------------------------------------------------------------------------
def _tess_batches(golden_tables):
    b = dict(cases.tess_cases())
    b["teapot_f32"] = (cases.teapot_patches(golden_tables, which=(0, 5, 13, 31)), np.linspace(0, 1, 8, dtype=np.float32),
                       np.linspace(0, 1, 12, dtype=np.float32))
    return b


def test_tessellate_against_reference(golden_tables):
    torch = pytest.importorskip("torch")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tess.npz"))
    for name, (patches, u, v) in _tess_batches(golden_tables).items():
        f32 = "f32" in name
        dt = np.float32 if f32 else np.float64
        tabs = [DeviceSpline(o, c, k, cf, dt) for (o, c, k, cf) in patches]
        pos, nrm = bspy_amd.tessellate_tables(tabs, (u, v))
        assert pos.shape == nrm.shape == (len(patches), 3, u.size, v.size) and pos.dtype == dt
        ref = g[f"{name}/positions"]
        assert np.abs(pos - ref).max() <= (2e-5 if f32 else 1e-12) * max(1.0, np.abs(ref).max()), name
        ref = g[f"{name}/normals"]
        ok = np.isfinite(ref) & np.isfinite(nrm)
        assert ok.mean() > 0.8
        observe(f"tessellation unit normals vs reference, {'fp32' if f32 else 'fp64'} [{name}]", np.abs(nrm[ok] - ref[ok]).max(), 2e-5 if f32 else 1e-12)
        # degenerate points (zero-length cross product): NaN in the reference and here
        assert np.array_equal(np.isnan(ref).any(axis=1), np.isnan(nrm).any(axis=1)) or f32
        # positions only; area-scaled and negated normals against the oracle
        only = bspy_amd.tessellate_tables(tabs, (u, v), normals=False)
        assert np.abs(only - pos).max() <= (2e-6 if f32 else 0.0)        # two kernel instantiations: fp32 may contract differently
        _, raw = bspy_amd.tessellate_tables(tabs, (u, v), normalize=False, negate=True)
        uu, vv = [a.reshape(-1).astype(np.float64) for a in np.meshgrid(u, v, indexing="ij")]
        o, c, k, cf = patches[1]
        orc, _ = oracle.c_normal(o, c, k, cf, [uu.astype(dt), vv.astype(dt)], False, True)
        observe(f"tessellation area normals vs oracle, {'fp32' if f32 else 'fp64'}", np.abs(raw[1].reshape(3, -1) - orc).max() / max(1.0, np.abs(orc).max()), 2e-5 if f32 else 1e-12)
        # device tensors in -> tensors out, same numbers; the single-patch grid call agrees bitwise
        dp, dn = bspy_amd.tessellate_tables(tabs, (torch.as_tensor(u, device="cuda"), torch.as_tensor(v, device="cuda")))
        assert dp.is_cuda and np.array_equal(dp.cpu().numpy(), pos) and np.array_equal(dn.cpu().numpy(), nrm, equal_nan=True)
        assert np.abs(tabs[0].evaluate_grid([u, v]) - pos[0]).max() <= (1e-6 if f32 else 1e-14)
    # the Spline-level entry point and its checks
    patches, u, v = cases.tess_cases()["o3_f64"]
    sp = [Spline(2, 3, o, c, k, cf) for (o, c, k, cf) in patches]
    p2, n2 = bspy_amd.tessellate(sp, u, v)
    t2 = [DeviceSpline(o, c, k, cf) for (o, c, k, cf) in patches]
    assert np.array_equal(p2, bspy_amd.tessellate_tables(t2, (u, v))[0])
    other = Spline(2, 3, (3, 3), (6, 7), [patches[0][2][0] * 1.0, patches[0][2][1] + 0.0], patches[0][3])
    other.knots[0][3] += 1e-3
    with pytest.raises(bspy_amd.BskError, match="must share"):
        bspy_amd.tessellate([sp[0], other], u, v)
    with pytest.raises(bspy_amd.DomainError) as e:
        bspy_amd.tessellate_tables(t2, (np.concatenate([u[:4], [9.0], u[5:]]), v))
    assert e.value.index == 4 * v.size


def test_tessellate_full_teapot(golden_tables):
    """All 32 patches of the teapot on a 256 x 256 grid in one call: every patch equals its own
    single-patch grid evaluation, normals are unit length where defined and orthogonal to finite
    differences of the positions."""
    patches = cases.teapot_patches(golden_tables)
    tabs = [DeviceSpline(o, c, k, cf, np.float32) for (o, c, k, cf) in patches]
    u = np.linspace(0, 1, 256, dtype=np.float32)
    pos, nrm = bspy_amd.tessellate_tables(tabs, (u, u))
    assert pos.shape == (32, 3, 256, 256)
    for p in (0, 7, 19, 31):
        assert np.abs(tabs[p].evaluate_grid([u, u]) - pos[p]).max() <= 2e-6
    ok = np.isfinite(nrm).all(axis=1)
    assert ok.mean() > 0.99
    ln = np.sqrt((nrm.astype(np.float64) ** 2).sum(axis=1))
    assert np.abs(ln[ok] - 1.0).max() <= 1e-5
    du = (pos[:, :, 2:, 1:-1] - pos[:, :, :-2, 1:-1]).astype(np.float64)
    dv = (pos[:, :, 1:-1, 2:] - pos[:, :, 1:-1, :-2]).astype(np.float64)
    n = nrm[:, :, 1:-1, 1:-1].astype(np.float64)
    inner = ok[:, 1:-1, 1:-1]
    for d in (du, dv):
        cosang = np.abs((d * n).sum(axis=1)) / np.maximum(np.sqrt((d ** 2).sum(axis=1)), 1e-12)
        assert np.nanmax(np.where(inner, cosang, 0.0)) <= 0.05


def test_cfg4_full_size(golden_tables):
    """BASELINE configs[3] at its real size: the 32 teapot patches on a dense 2048 x 2048 grid each, positions and unit
    normals from ONE bsk_tessellate call, device resident.  Every patch equals its own single-patch grid call; the
    oracle on a 33 x 33 sub-grid of every patch; the four corner points of every patch are the corner points of the
    reference's 16 x 16 grid (tests/golden: teapot_grid16 - linspace(0, 1, 2048) and linspace(0, 1, 16) share no
    other point); determinism."""
    torch = pytest.importorskip("torch")
    patches = cases.teapot_patches(golden_tables)
    tabs = [DeviceSpline(o, c, k, cf, np.float32) for (o, c, k, cf) in patches]
    side = 2048
    g = torch.linspace(0, 1, side, dtype=torch.float32, device="cuda")
    pos, nrm = bspy_amd.tessellate_tables(tabs, (g, g))
    assert tuple(pos.shape) == (32, 3, side, side) and pos.dtype == torch.float32
    for p in (0, 11, 31):
        one = tabs[p].evaluate_grid_device([g, g])
        observe("cfg4 full size: batch vs single-patch grid call, fp32", float((one - pos[p]).abs().max()), 2e-6)
        del one
    sub = torch.arange(0, side, 64, device="cuda").tolist() + [side - 1]
    gs = g[sub].cpu().numpy()
    uu, vv = [a.reshape(-1) for a in np.meshgrid(gs, gs, indexing="ij")]
    worst = 0.0
    for p, (o, c, k, cf) in enumerate(patches):
        orc, bad = oracle.c_evaluate(o, c, k, cf, [0, 0], [uu, vv])
        assert bad == -1
        got = pos[p][:, sub][:, :, sub].reshape(3, -1).cpu().numpy()
        worst = max(worst, float(np.abs(got - orc).max() / _scale(orc)))
    observe("cfg4 full size: 33 x 33 sub-grid of every patch vs oracle, fp32", worst, 2e-5)
    ref16 = golden_tables["teapot_grid16"]
    corners = pos[:, :, [0, 0, side - 1, side - 1], [0, side - 1, 0, side - 1]].cpu().numpy()
    ref_c = ref16[:, :, [0, 0, 15, 15], [0, 15, 0, 15]]
    observe("cfg4 full size: patch corners vs the reference's 16 x 16 grid, fp32", float(np.abs(corners - ref_c).max() / _scale(ref_c)), 2e-6)
    ok = torch.isfinite(nrm).all(dim=1)
    assert float(ok.float().mean()) > 0.99
    ln = (nrm.double() ** 2).sum(dim=1).sqrt()
    assert float((ln[ok] - 1.0).abs().max()) <= 1e-5
    pos2 = bspy_amd.tessellate_tables(tabs, (g, g), normals=False)
    observe("cfg4 full size: positions-only call vs positions + normals, fp32", float((pos2 - pos).abs().max()), 2e-6)
    assert torch.equal(pos2, bspy_amd.tessellate_tables(tabs, (g, g), normals=False))


def test_cell_order_pipeline_in_pieces(monkeypatch):
    """A batch beyond 2^dest_bits points (67 M, or 16.7 M when the third variable has more than 64 spans) runs the pipeline
    piece by piece.  BSK_CS_PIECE shortens the pieces so that a 300 k batch takes four of them (ragged last one): same
    bits as the one-piece run for evaluate, a derivative and the fused jacobian; first offender in the last piece."""
    rng = np.random.default_rng(8)
    order, ncoef, ndep = (3, 3, 3), (40, 38, 90), 3               # 88 spans in the third variable: 24 destination bits
    knots = [cases.nonuniform_knots(rng, o, c, np.float32, 0.0, 1.0) for o, c in zip(order, ncoef)]
    coefs = rng.standard_normal((ndep, *ncoef)).astype(np.float32)
    n = 300_011
    pts = [rng.random(n).astype(np.float32) for _ in range(3)]
    whole = DeviceSpline(order, ncoef, knots, coefs, np.float32)
    ref = [whole.evaluate(pts), whole.evaluate(pts, [0, 1, 1]), whole.jacobian(pts)]
    assert "fused jacobian" in whole.last_kernel()
    monkeypatch.setenv("BSK_CS_PIECE", "90000")                  # (read by the library at every call)
    t = DeviceSpline(order, ncoef, knots, coefs, np.float32)
    got = [t.evaluate(pts), t.evaluate(pts, [0, 1, 1]), t.jacobian(pts)]
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    bad = [p.copy() for p in pts]
    bad[1][299_999] = np.float32(-0.5)
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate(bad)
    assert e.value.index == 299_999


@pytest.mark.parametrize("order,ncoef,ndep,dt", [((3, 4), (20, 20), 3, np.float64), ((2, 6), (30, 9), 1, np.float64),
                                                  ((5, 3), (12, 40), 4, np.float32), ((4, 6), (25, 14), 2, np.float32),
                                                  ((1, 4), (16, 16), 3, np.float64), ((6, 2), (9, 33), 2, np.float64)])
def test_mixed_order_surfaces_in_one_slab(order, ncoef, ndep, dt, monkeypatch):
    """LDS-resident surfaces of mixed orders: batches of >= 2^20 points run eval_slab2 with the whole table as its one slab
    (no ordering phase, batch-order results) instead of the compiler-managed eval_mixed.  Every derivative multi-index up
    to total order 2 (+ one beyond an order), points on every knot +- 1 ulp, against the oracle and eval_mixed
    (BSK_VARIANT=7); ragged last chunk; NaN; first offender."""
    rng = np.random.default_rng(sum(order) * 10 + ndep)
    knots = [cases.nonuniform_knots(rng, o, c, dt, -1.0, 1.0) for o, c in zip(order, ncoef)]
    coefs = rng.standard_normal((ndep, *ncoef)).astype(dt)
    n = 1_100_003
    pts = []
    for k, o, c in zip(knots, order, ncoef):
        lo, hi = k[o - 1], k[c]
        p = (lo + (hi - lo) * rng.random(n)).astype(dt)
        dd = np.unique(k)
        e = np.concatenate((dd, np.nextafter(dd, dt(-np.inf)), np.nextafter(dd, dt(np.inf)))).astype(dt)
        e = e[(e >= lo) & (e <= hi)]
        p[1_000: 1_000 + len(e)] = rng.permutation(e)
        pts.append(p)
    t = DeviceSpline(order, ncoef, knots, coefs, dt)
    monkeypatch.setenv("BSK_VARIANT", "7")
    plain = DeviceSpline(order, ncoef, knots, coefs, dt)
    kind = "fp32" if dt == np.float32 else "fp64"
    tol = 2e-5 if dt == np.float32 else 1e-12
    sample = np.unique(np.concatenate((np.arange(0, 3_000), rng.integers(0, n, 40_000), np.arange(n - 3_000, n))))   # the knot points, a random part, the ragged end
    for w in cases.all_wrt(2, 2) + [(max(order), 0), (0, min(order))]:
        out = t.evaluate(pts, list(w))
        assert t.last_kernel() == "eval_slab2", t.last_kernel()
        orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, list(w), [p[sample] for p in pts])
        assert bad == -1
        observe(f"eval_slab2 (one slab) vs oracle, {kind}", np.abs(out[:, sample] - orc).max() / _scale(orc), tol)
        ref = plain.evaluate(pts, list(w))
        assert plain.last_kernel() == "eval_mixed", plain.last_kernel()
        observe(f"eval_slab2 (one slab) vs eval_mixed, {kind}", np.abs(out - ref).max() / _scale(ref), 1e-5 if dt == np.float32 else 1e-13)
    nan = [p.copy() for p in pts]
    nan[0][n - 4] = np.nan
    nan[1][5] = np.nan
    out = t.evaluate(nan)
    orc, _ = oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0], [p[sample] for p in nan])      # (a variable of order 1 does not propagate its NaN)
    assert np.array_equal(np.isnan(out[:, sample]), np.isnan(orc))
    assert np.isnan(out[:, 5]).all() or min(order) == 1
    ok = ~np.isnan(orc)
    assert np.abs(out[:, sample][ok] - orc[ok]).max() <= tol * _scale(orc[ok])
    bad = [p.copy() for p in pts]
    bad[0][n - 1] = dt(9.0)
    bad[1][1_066_000] = dt(-9.0)
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate(bad)
    assert e.value.index == 1_066_000


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_slab_kernel_rounds(dt, monkeypatch):
    """eval_slab2 orders up to 32 chunks of a workgroup per ROUND and starts another round after them - only batches beyond
    67 M points reach a second round on 256 CUs.  BSK_SLAB_GRID = 1 makes one workgroup take all 65 chunks of a 530 k batch
    (three rounds, the last with one ragged chunk), 3 gives 22 / 22 / 21: same bits as the full grid, for evaluate and a
    derivative; offender index in the third round."""
    rng = np.random.default_rng(9)
    order, ncoef, ndep = (4, 5), (900 if dt == np.float64 else 1800, 11), 3      # the TomsNasty shape (238 KB: three passes); twice the rows in fp32
    knots = [cases.nonuniform_knots(rng, o, c, dt, 0.0, 1.0) for o, c in zip(order, ncoef)]
    coefs = rng.standard_normal((ndep, *ncoef)).astype(dt)
    n = 530_003
    pts = [rng.random(n).astype(dt) for _ in range(2)]
    t = DeviceSpline(order, ncoef, knots, coefs, dt)
    ref = [t.evaluate(pts), t.evaluate(pts, [1, 2])]
    assert t.last_kernel() == "eval_slab2", t.last_kernel()
    orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, [1, 2], [p[:20_000] for p in pts])
    assert bad == -1
    kind = "fp32" if dt == np.float32 else "fp64"
    observe(f"eval_slab2 vs oracle, derivative (1, 2), {kind}", np.abs(ref[1][:, :20_000] - orc).max() / _scale(orc), 2e-5 if dt == np.float32 else 1e-12)
    for grid in ("1", "3"):
        monkeypatch.setenv("BSK_SLAB_GRID", grid)
        assert np.array_equal(t.evaluate(pts), ref[0]), grid
        assert np.array_equal(t.evaluate(pts, [1, 2]), ref[1]), grid
    bad = [p.copy() for p in pts]
    bad[0][529_000] = dt(2.0)                                     # chunk 64: the third round of the only workgroup
    monkeypatch.setenv("BSK_SLAB_GRID", "1")
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate(bad)
    assert e.value.index == 529_000


def _lut_steps(knots, order, ncoef):
    """Bisection steps the library's bucket table of this knot vector needs (bsk_api.hip build_lut: 4 x spans buckets,
    brackets widened by 1 % of a bucket; the smaller table is only taken when one step suffices)."""
    k = np.asarray(knots, np.float64)
    lo, hi = k[order - 1], k[ncoef]
    m = 16
    while m < 4 * (ncoef - order + 1) and m < 4096:
        m <<= 1
    inv = (hi - lo) / m
    span = lambda x: order + int(np.searchsorted(k[order:ncoef], x, side="right"))
    widest = 0
    for b in range(m):
        l = span(lo + (b - 0.01) * inv)
        h = ncoef if b == m - 1 else span(lo + (b + 1.01) * inv)
        widest = max(widest, h - l)
    return int(np.ceil(np.log2(widest + 1)))


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("style,steps_class", [("uniform", 1), ("ratio20", 2), ("graded", 0)])
def test_sort_kernels_span_search_forms(style, steps_class, dt):
    """bin_totals / bin_scatter_tag search spans with a compile-time number of bisection steps (1 or 2) or, for knot vectors
    whose buckets hold more spans, the run-time loop: one knot style per form, fp32 (MFMA evaluation, 8 points per lane in
    the sort kernels) and fp64 (4 per lane), against the oracle and bitwise-deterministic; out-of-domain index."""
    rng = np.random.default_rng(5)
    order, ncoef, ndep = (3, 3, 3), (42, 40, 44), 2
    knots = []
    for o, c in zip(order, ncoef):
        s = c - o + 1
        if style == "uniform":
            inner = np.linspace(0.0, 1.0, s + 1)
        elif style == "ratio20":                                 # spans of width 1 and 20 alternating in blocks of three
            wdt = np.where((np.arange(s) // 3) % 2 == 0, 1.0, 20.0)
            inner = np.concatenate(([0.0], np.cumsum(wdt))) / wdt.sum()
        else:                                                    # spans shrinking towards the left end: a dozen in one bucket
            inner = (np.arange(s + 1) / s) ** 4
        knots.append(np.concatenate(([inner[0]] * (o - 1), inner, [inner[-1]] * (o - 1))).astype(dt))
    steps = max(_lut_steps(k, o, c) for k, o, c in zip(knots, order, ncoef))
    assert (steps if steps <= 2 else 0) == steps_class, steps
    coefs = rng.standard_normal((ndep, *ncoef)).astype(dt)
    n = 270_001
    pts = [rng.random(n).astype(dt) for _ in range(3)]
    if style == "graded":
        pts[0][: n // 2] = (rng.random(n // 2) ** 4).astype(dt)   # half the batch in the crowded end
    t = DeviceSpline(order, ncoef, knots, coefs, dt)
    out = t.evaluate(pts)
    assert "eval_cellsort" in t.last_kernel(), t.last_kernel()
    sample = rng.choice(n, 30_000, replace=False)
    orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0, 0], [p[sample] for p in pts])
    assert bad == -1
    kind = "fp32" if dt == np.float32 else "fp64"
    observe(f"cell order, span search forms, vs oracle, {kind}", np.abs(out[:, sample] - orc).max() / _scale(orc), 2e-5 if dt == np.float32 else 1e-12)
    assert np.array_equal(out, t.evaluate(pts))
    bad = [p.copy() for p in pts]
    bad[2][200_001] = dt(1.5)
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate(bad)
    assert e.value.index == 200_001


@pytest.mark.parametrize("order,ndep", [(1, 2), (2, 1), (2, 4), (3, 3), (4, 1), (4, 4), (5, 2), (5, 4), (6, 3)])
def test_fused_jacobian_in_cell_order(order, ndep):
    """eval_cellsort<..., JAC> (fp32, three variables of one order): value and derivative bases from one recursion, every
    window row read once for three accumulator sets.  Every order and several nDep against the oracle and against the three
    derivative passes of the same pipeline; clustered points, points on knots, determinism."""
    rng = np.random.default_rng(100 * order + ndep)
    ncoef = (order + 35, order + 31, order + 37)                # tables beyond LDS: the cell-order pipeline
    knots = [cases.nonuniform_knots(rng, order, c, np.float32, 0.0, 1.0) for c in ncoef]
    coefs = rng.standard_normal((ndep, *ncoef)).astype(np.float32)
    n = 280_003
    pts = [rng.random(n).astype(np.float32) for _ in range(3)]
    pts[2][: n // 4] = np.float32(0.4321)
    for iv, k in enumerate(knots):
        d = np.unique(k)
        e = np.concatenate((d, np.nextafter(d, np.float32(-1)), np.nextafter(d, np.float32(2)))).astype(np.float32)
        e = e[(e >= 0) & (e <= 1)]
        pts[iv][100_000 + 1_000 * iv: 100_000 + 1_000 * iv + len(e)] = e
    t = DeviceSpline((order,) * 3, ncoef, knots, coefs, np.float32)
    jac = t.jacobian(pts)
    assert "fused jacobian" in t.last_kernel(), t.last_kernel()
    assert jac.shape == (ndep, 3, n)
    sample = np.concatenate((rng.choice(n, 15_000, replace=False), np.arange(100_000, 103_000)))
    ojac, bad = oracle.c_jacobian((order,) * 3, ncoef, knots, coefs, [p[sample] for p in pts])
    assert bad == -1
    observe("fused jacobian (cell order) vs fp32 oracle", np.abs(jac[:, :, sample] - ojac).max() / _scale(ojac), 2e-5)
    for j in range(3):
        w = [0, 0, 0]
        w[j] = 1
        dj = t.evaluate(pts, w)
        assert "eval_cellsort" in t.last_kernel()
        observe("fused jacobian vs derivative passes (cell order)", np.abs(jac[:, j] - dj).max() / _scale(dj), 2e-5)
    assert np.array_equal(jac, t.jacobian(pts))


@pytest.mark.parametrize("ncoef,ndep,knots_kind", [((64, 64), 3, "uniform"), ((4, 4), 3, "bezier"), ((9, 23), 1, "nonuniform"),
                                                    ((37, 16), 4, "nonuniform"), ((12, 12), 2, "uniform")])
def test_fp32_bicubic_records(ncoef, ndep, knots_kind):
    """eval_rec32 (bsk_rec32.hpp): all-fp32 bicubics read per-span records and padded control points with 16-byte LDS
    reads.  Every derivative multi-index up to total order 3, points on every knot and one ulp either side, odd batch
    sizes, against the fp32 oracle and the general kernel (BSK_VARIANT=9 keeps eval_rowrot); out-of-domain index."""
    rng = np.random.default_rng(hash((ncoef, ndep)) % 1000)
    order = (4, 4)
    if knots_kind == "uniform":
        knots = [cases.clamped_uniform_knots(4, c, np.float32) for c in ncoef]
    elif knots_kind == "bezier":
        knots = [np.array((0, 0, 0, 0, 1, 1, 1, 1), np.float32)] * 2
    else:
        knots = [cases.nonuniform_knots(rng, 4, c, np.float32, -2.0, 3.0) for c in ncoef]
    coefs = rng.standard_normal((ndep, *ncoef)).astype(np.float32)
    t = DeviceSpline(order, ncoef, knots, coefs, np.float32)
    os.environ["BSK_VARIANT"] = "9"
    try:
        general = DeviceSpline(order, ncoef, knots, coefs, np.float32)
    finally:
        del os.environ["BSK_VARIANT"]
    for n in (1, 67, 20_011):
        pts = []
        for k, c in zip(knots, ncoef):
            lo, hi = np.float32(k[3]), np.float32(k[c])
            p = (lo + (hi - lo) * rng.random(n)).astype(np.float32)
            d = np.unique(k)
            e = np.concatenate((d, np.nextafter(d, np.float32(-np.inf)), np.nextafter(d, np.float32(np.inf)))).astype(np.float32)
            e = e[(e >= lo) & (e <= hi)]
            m = min(len(e), n)
            p[:m] = rng.permutation(e)[:m]
            pts.append(p)
        for w in cases.all_wrt(2, 3) + [(4, 0), (0, 5)]:
            out = t.evaluate(pts, list(w))
            assert t.last_kernel() == "eval_rec32", t.last_kernel()
            orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, list(w), pts)
            assert bad == -1
            observe("eval_rec32 vs fp32 oracle", np.abs(out - orc).max() / _scale(orc), 2e-5)
            ref = general.evaluate(pts, list(w))
            assert general.last_kernel() == "eval_rowrot"
            observe("eval_rec32 vs eval_rowrot (fp32)", np.abs(out - ref).max() / _scale(ref), 2e-5)
    bad = [p.copy() for p in pts]
    bad[1][4_321] = np.nextafter(np.float32(knots[1][ncoef[1]]), np.float32(np.inf))
    bad[0][9_999] = np.float32(-100.0)
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate(bad)
    assert e.value.index == 4_321
    nan = [p.copy() for p in pts]
    nan[0][5] = np.nan
    out = t.evaluate(nan)
    assert np.isnan(out[:, 5]).all() and np.isfinite(out[:, 6]).all()


def test_fused_curvature_orders_and_sizes():
    """curv_rowrot (fused Gaussian curvature on the LDS image) for both template orders against the
    oracle, on odd batch sizes; the piecewise bilinear case has S_uu = S_vv = 0."""
    rng = np.random.default_rng(31)
    for order, ncoef in (((2, 2), (9, 7)), ((4, 4), (12, 10))):
        knots = [cases.nonuniform_knots(rng, o, c, np.float64, 0.0, 1.0) for o, c in zip(order, ncoef)]
        coefs = rng.standard_normal((3, *ncoef))
        t = DeviceSpline(order, ncoef, knots, coefs)
        for n in (1, 65, 5003):
            pts = [rng.random(n), rng.random(n)]
            orc, bad = oracle.c_curvature(order, ncoef, knots, coefs, pts)
            got = t.curvature(pts)
            ok = np.isfinite(orc)
            assert ok.mean() > 0.9 and np.array_equal(ok, np.isfinite(got))
            observe(f"fused curvature vs oracle, order {order[0]}", np.abs(got[ok] - orc[ok]).max() / max(1.0, np.abs(orc[ok]).max()), 1e-11)
        bad = [rng.random(100), rng.random(100)]
        bad[1][42] = 1.5
        with pytest.raises(bspy_amd.DomainError) as e:
            t.curvature(bad)
        assert e.value.index == 42


def test_random_shapes_against_oracle():
    """Differential sweep over the dispatcher: random nInd (1..5), orders (1..9, equal or different per
    variable), nDep (1..6), knot styles, dtypes, table sizes (LDS-resident and L2-resident) and
    derivative orders; evaluate, derivative and jacobian against the C oracle, on a batch that takes
    the zero-copy host path and one that takes the staged path."""
    trials = int(os.environ.get("BSK_SOAK", "70"))              # BSK_SOAK=1000 for a longer soak with another seed
    rng = np.random.default_rng(20260 if trials == 70 else 777)
    checked = 0
    worst = {np.float32: 0.0, np.float64: 0.0}
    for trial in range(trials):
        nind = int(rng.choice([1, 2, 2, 2, 3, 3, 4, 5]))
        omax_allowed = {1: 9, 2: 8, 3: 6, 4: 4, 5: 3}[nind]
        if rng.random() < 0.4:
            order = tuple([int(rng.integers(1, omax_allowed + 1))] * nind)
        else:
            order = tuple(int(rng.integers(1, omax_allowed + 1)) for _ in range(nind))
        big = rng.random() < 0.25 and nind <= 3
        lim = {1: 20000, 2: 260, 3: 42}[nind] if big else {1: 40, 2: 14, 3: 8, 4: 5, 5: 4}[nind]
        ncoef = tuple(int(o + rng.integers(0, max(1, lim - o + 1))) for o in order)
        if big:
            ncoef = tuple(max(c, lim - 5) for c in ncoef)
        ndep = int(rng.integers(1, 7))
        dt = np.float32 if rng.random() < 0.3 else np.float64
        if rng.random() < 0.5:
            knots = [cases.clamped_uniform_knots(o, c, dt) for o, c in zip(order, ncoef)]
        else:
            knots = [cases.nonuniform_knots(rng, o, c, dt, -1.0, 2.0) for o, c in zip(order, ncoef)]
        coefs = rng.standard_normal((ndep, *ncoef)).astype(dt)
        t = DeviceSpline(order, ncoef, knots, coefs, dt)
        dom = [(float(k[o - 1]), float(k[c])) for k, o, c in zip(knots, order, ncoef)]
        # of the result scale.  fp64: the north_star bound is 1e-10; observed worst over the sweep 1.3e-15
        # (fp32 8.6e-7), printed below
        tol = 2e-5 if dt == np.float32 else 1e-12
        for n in (257, 70_001):
            pts = [(lo + (hi - lo) * rng.random(n)).astype(dt) for lo, hi in dom]
            pts = [np.clip(p, dt(lo), dt(hi)) for p, (lo, hi) in zip(pts, dom)]
            wrts = [[0] * nind, [int(rng.integers(0, 3)) for _ in range(nind)]]
            for w in wrts:
                got = t.evaluate(pts, w)
                orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, w, pts)
                assert bad == -1
                err = float(np.abs(got - orc).max()) / _scale(orc)
                worst[dt] = max(worst[dt], err)
                assert err <= tol, (trial, order, ncoef, ndep, dt, n, w, err)
            if n == 257 or nind <= 3:
                got = t.jacobian(pts)
                orc, _ = oracle.c_jacobian(order, ncoef, knots, coefs, pts)
                err = float(np.abs(got - orc).max()) / _scale(orc)
                worst[dt] = max(worst[dt], err)
                assert err <= tol, (trial, order, ncoef, ndep, dt, n, "jac", err)
            checked += 1
        t.close()
    assert checked == 2 * trials
    print(f"random shapes: worst error of the result scale fp64 {worst[np.float64]:.2e}, fp32 {worst[np.float32]:.2e}")


def test_tessellate_more_patches_than_one_launch_takes():
    """bsk_tessellate passes at most 64 coefficient pointers per launch: 70 patches take two."""
    rng = np.random.default_rng(8)
    ku, kv = cases.clamped_uniform_knots(3, 5), cases.clamped_uniform_knots(4, 6)
    tabs = [DeviceSpline((3, 4), (5, 6), [ku, kv], rng.standard_normal((3, 5, 6))) for _ in range(70)]
    u, v = np.linspace(0, 1, 9), np.linspace(0, 1, 12)
    pos, nrm = bspy_amd.tessellate_tables(tabs, (u, v))
    assert pos.shape == (70, 3, 9, 12)
    for p in (0, 63, 64, 69):
        assert np.abs(tabs[p].evaluate_grid([u, v]) - pos[p]).max() <= 1e-13
        uu, vv = [a.ravel() for a in np.meshgrid(u, v, indexing="ij")]
        assert np.abs(tabs[p].normal([uu, vv]).reshape(3, 9, 12) - nrm[p]).max() <= 1e-10


def test_random_large_tables_against_oracle():
    """Random L2-resident tables (nInd 1..3, equal or different orders <= 6, nDep 1..4, both dtypes):
    the gather kernel on a small batch, the cell-order pipeline on a batch >= 2^18 points and, for
    surfaces, eval_slab2 on a batch >= 2^19 points against the C oracle."""
    trials = int(os.environ.get("BSK_SOAK_LARGE", "14"))        # BSK_SOAK_LARGE=80 for a longer sweep with another seed
    rng = np.random.default_rng(99 if trials == 14 else 4242)
    for trial in range(trials):
        nind = int(rng.choice([1, 2, 2, 3, 3]))
        if rng.random() < 0.5:
            order = tuple([int(rng.integers(1, 7))] * nind)
        else:
            order = tuple(int(rng.integers(1, 7)) for _ in range(nind))
        ndep = int(rng.integers(1, 5))
        dt = np.float32 if rng.random() < 0.4 else np.float64
        target = {1: 60_000, 2: 330, 3: 48}[nind]                 # enough coefficients to overflow the LDS
        ncoef = tuple(int(max(o, target + rng.integers(-6, 7))) for o in order)
        knots = [cases.nonuniform_knots(rng, o, c, dt, 0.0, 1.0) if rng.random() < 0.5 else cases.clamped_uniform_knots(o, c, dt)
                 for o, c in zip(order, ncoef)]
        coefs = rng.standard_normal((ndep, *ncoef)).astype(dt)
        assert coefs.nbytes > 170_000
        t = DeviceSpline(order, ncoef, knots, coefs, dt)
        tol = 2e-5 if dt == np.float32 else 1e-12
        for n in (4_097, 270_001) + ((530_003,) if nind == 2 else ()):       # surfaces: eval_slab2 from 2^19 points on
            pts = [rng.random(n).astype(dt) for _ in range(nind)]
            w = [int(rng.integers(0, 2)) for _ in range(nind)]
            idx = rng.choice(n, 4_000, replace=False)
            for ww in ([0] * nind, w):
                got = t.evaluate(pts, ww)
                orc, bad = oracle.c_evaluate(order, ncoef, knots, coefs, ww, [p[idx] for p in pts])
                assert bad == -1
                observe(f"random large tables vs oracle, {'fp32' if dt == np.float32 else 'fp64'}", np.abs(got[:, idx] - orc).max() / _scale(orc), tol)
        t.close()


def test_handle_lifecycle():
    """Creating, using (every host path: zero-copy, staged, pipelined, cell-order workspace) and destroying
    many handles leaves the device memory where it was."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(1)
    c = CASES["cfg2_bicubic"]
    u, v = rng.random(300_000), rng.random(300_000)
    big_order, big_ncoef = (3, 3), (300, 300)
    bk = [cases.clamped_uniform_knots(o, n_) for o, n_ in zip(big_order, big_ncoef)]
    bc = rng.standard_normal((2, *big_ncoef))
    def cycle(i):
        t = DeviceSpline(c.order, c.nCoef, c.knots, c.coefs)
        t.evaluate([u[:10], v[:10]])                      # zero-copy path
        t.evaluate([u, v])                                # staged path
        if i % 8 == 0:
            t.jacobian([np.tile(u, 8), np.tile(v, 8)])    # pipelined path (2.4 M points)
        t.close()
        b = DeviceSpline(big_order, big_ncoef, bk, bc)
        b.evaluate([u, v])                                # cell-order workspace
        b.close()

    for i in range(3):                                    # first use: code objects, runtime pools
        cycle(0)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(40):
        cycle(i)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 32 << 20, (free0, free1)


def _uniform_knots(order, ncoef, lo, hi, clamp_lo=True, clamp_hi=True):
    ns = ncoef - order + 1
    dom = np.linspace(lo, hi, ns + 1)
    h = (hi - lo) / ns
    left = np.full(order - 1, lo) if clamp_lo else lo - h * np.arange(order - 1, 0, -1)
    right = np.full(order - 1, hi) if clamp_hi else hi + h * np.arange(1, order)
    return np.concatenate([left, dom, right])


@pytest.mark.parametrize("order,ncoef,dom,clamps,nDep,dt", [
    (4, (64, 64), ((0.0, 1.0), (0.0, 1.0)), (True, True, True, True), 3, np.float64),        # cfg2
    (4, (9, 23), ((-2.0, 3.0), (10.0, 10.5)), (True, True, True, True), 3, np.float64),      # shifted / scaled domains, few spans
    (4, (8, 8), ((0.0, 1.0), (0.0, 1.0)), (True, False, False, True), 3, np.float64),        # uniform continuation instead of a clamp
    (2, (17, 40), ((0.0, 1.0), (-1.0, 1.0)), (True, True, True, True), 3, np.float64),       # bilinear
    (4, (30, 21), ((0.0, 2.0), (0.0, 1.0)), (True, True, True, True), 5, np.float64),        # nDep > 3: dependent-variable-major image
    (4, (12, 40), ((0.0, 1.0), (0.0, 1.0)), (True, True, True, True), 1, np.float64),        # nDep 1
    (2, (33, 20), ((0.0, 1.0), (0.0, 4.0)), (True, True, True, True), 2, np.float32),        # fp32 runs this path at order 2 only
])
def test_uniform_knot_path(order, ncoef, dom, clamps, nDep, dt, monkeypatch):
    """Equally spaced knots run the table-free kernels (eval_uni / jac_uni) on an unclamped LDS image;
    they must meet the same bar as the general kernels on the same spline (BSK_VARIANT=9): every derivative,
    the fused jacobian and normal, points on every knot and one ulp either side of it (the span rule is
    the reference's searchsorted(..., 'right'): a third derivative tells a wrong side at once)."""
    rng = np.random.default_rng(77)
    tol = 1e-12 if dt == np.float64 else 2e-5
    knots = [_uniform_knots(order, nc, lo, hi, clamps[2 * i], clamps[2 * i + 1]).astype(dt) for i, (nc, (lo, hi)) in enumerate(zip(ncoef, dom))]
    coefs = rng.standard_normal((nDep, *ncoef)).astype(dt)
    t = DeviceSpline((order, order), ncoef, knots, coefs, dt)
    n = 20000
    dom = [(float(k[order - 1]), float(k[nc])) for k, nc in zip(knots, ncoef)]
    pts = [(lo + (hi - lo) * rng.random(n)).astype(dt).clip(dt(lo), dt(hi)) for lo, hi in dom]
    for i, (k, nc) in enumerate(zip(knots, ncoef)):
        kk = np.unique(k[order - 1:nc + 1])
        edge = np.concatenate([kk, np.nextafter(kk[1:], dt(-np.inf)), np.nextafter(kk[:-1], dt(np.inf))]).astype(dt)
        pts[i][:edge.size] = edge
        pts[1 - i][:edge.size] = (dom[1 - i][0] + (dom[1 - i][1] - dom[1 - i][0]) * rng.random(edge.size)).astype(dt).clip(dt(dom[1 - i][0]), dt(dom[1 - i][1]))
    pts[0][-1], pts[1][-1] = dom[0][1], dom[1][1]
    pts[0][-2], pts[1][-2] = dom[0][0], dom[1][0]
    wrts = [(a, b) for a in range(order + 1) for b in range(order + 1) if a + b <= order + 1]
    for w in wrts:
        out = t.evaluate(pts, list(w))
        assert t.last_kernel() == "eval_uni"
        orc, bad = oracle.c_evaluate((order, order), ncoef, knots, coefs, list(w), pts)
        assert bad == -1
        assert np.abs(out - orc).max() <= tol * _scale(orc), (w, np.abs(out - orc).max())
    jac = t.jacobian(pts)
    assert t.last_kernel() == "jac_uni"
    orj, _ = oracle.c_jacobian((order, order), ncoef, knots, coefs, pts)
    assert np.abs(jac - orj).max() <= tol * _scale(orj)
    has_normal = abs(2 - nDep) == 1
    if has_normal:
        nrm = t.normal(pts)
        assert t.last_kernel() == "jac_uni"
    # the general kernels on the same spline
    monkeypatch.setenv("BSK_VARIANT", "9")
    g = DeviceSpline((order, order), ncoef, knots, coefs, dt)
    out_g = g.evaluate(pts)
    assert g.last_kernel() == "eval_rowrot"
    assert np.abs(t.evaluate(pts) - out_g).max() <= tol * _scale(out_g)
    if has_normal:
        gn = g.normal(pts)
        ok = np.isfinite(gn).all(axis=0)
        assert np.abs(nrm - gn)[:, ok].max() <= (1e-10 if dt == np.float64 else 1e-3)
    # NaN parameters propagate, out-of-domain points are reported by index
    bad_pts = [p.copy() for p in pts]
    bad_pts[0][123] = np.nan
    assert np.isnan(t.evaluate(bad_pts)[:, 123]).all()
    d3 = t.evaluate(bad_pts, [order - 1, 0])[:, 123]
    o3, _ = oracle.c_evaluate((order, order), ncoef, knots, coefs, [order - 1, 0], [p[123:124] for p in bad_pts])
    assert np.allclose(d3, o3[:, 0], rtol=100 * tol, atol=100 * tol * _scale(o3), equal_nan=True)
    bad_pts[0][123] = dt(0.5 * (dom[0][0] + dom[0][1]))
    bad_pts[1][4567] = np.nextafter(dt(dom[1][1]), dt(np.inf))
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate(bad_pts)
    assert e.value.index == 4567
    # in-place mutation + update re-builds the unclamped image
    coefs2 = coefs.copy()
    coefs2[:, 0, :] += 1.0
    t.update(knots, coefs2)
    orc2, _ = oracle.c_evaluate((order, order), ncoef, knots, coefs2, [0, 0], pts)
    assert np.abs(t.evaluate(pts) - orc2).max() <= tol * _scale(orc2)


@pytest.fixture(scope="module")
def golden_extras():
    return np.load(os.path.join(ROOT, "tests", "golden", "extras.npz"))


def test_normal_index_subsets_against_reference(golden_extras):
    """Spline.normal(uvw, normalize, indices): the reference builds only the selected cofactors and divides
    by the norm of THAT vector (bspy/_spline_evaluation.py:234-244): indices=[2] of a unit 'normal' is +-1."""
    g = golden_extras
    surf = Spline(2, 3, [4, 3], [8, 5], [g["normal_surf_knots0"], g["normal_surf_knots1"]], g["normal_surf_coefs"])
    pts = g["normal_pts"]
    for name, idx in (("0_2", [0, 2]), ("2", [2]), ("1_0", [1, 0])):
        for nz in (True, False):
            ref = g[f"normal_idx_{name}_{int(nz)}"]
            one = np.array([surf.normal(p, nz, idx) for p in pts[:5]])               # the reference's single-point call
            assert one.shape == ref[:5].shape and np.abs(one - ref[:5]).max() <= 1e-12 * max(1.0, np.abs(ref).max())
            many = surf.normal([pts[:, 0], pts[:, 1]], nz, idx)                      # batched extension
            assert many.shape == ref.T.shape and np.abs(many - ref.T).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    assert np.abs(np.abs(surf.normal([pts[:, 0], pts[:, 1]], True, [2])) - 1.0).max() <= 1e-14
    curve = Spline(1, 2, [4], [8], [g["normal_surf_knots0"]], g["normal_curve_coefs"])
    got = curve.normal([pts[:, 0]], True, [1])
    assert np.abs(got - g["normal_curve_idx_1"].T).max() <= 1e-13
    import torch
    tu, tv = torch.as_tensor(pts[:, 0], device="cuda"), torch.as_tensor(pts[:, 1], device="cuda")
    t = surf.normal([tu, tv], True, [0, 2])
    assert t.is_cuda and np.abs(t.cpu().numpy() - g["normal_idx_0_2_1"].T).max() <= 1e-12


def test_curvature_of_scalar_valued_splines(golden_extras):
    """nDep == 1: the reference evaluates the curvature of the GRAPH of the function
    (bspy/_spline_evaluation.py:81-82 -> graph(), Greville abscissae as the new coordinates)."""
    g = golden_extras
    pts = g["normal_pts"]
    f1 = Spline(1, 1, [4], [8], [g["normal_surf_knots0"]], g["graph_curve_coefs"])
    ref = g["graph_curve_curvature"]
    assert np.abs(f1.curvature([pts[:, 0]]) - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    assert abs(f1.curvature(float(pts[3, 0])) - ref[3]) <= 1e-11 * max(1.0, abs(ref[3]))
    f2 = Spline(2, 1, [4, 3], [8, 5], [g["normal_surf_knots0"], g["normal_surf_knots1"]], g["graph_surf_coefs"])
    ref = g["graph_surf_curvature"]
    got = f2.curvature([pts[:, 0], pts[:, 1]])
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    with pytest.raises(NotImplementedError):
        Spline(1, 1, [3], [4], [[-1.0, 0, 0, 0.5, 1, 1, 2.0]], np.ones((1, 4))).curvature(0.25)    # not clamped


def test_collocation_matrix_against_reference(golden_extras):
    """The matrix the reference's least_squares assembles (bspy/_spline_fitting.py:736-751), captured from
    its own numpy.linalg.lstsq call: Hermite rows for repeated parameter values included."""
    g = golden_extras
    A = bspy_amd.collocation_matrix(g["colloc_knots"], 5, g["colloc_u"])
    ref = g["colloc_A"]
    assert A.shape == ref.shape
    assert np.abs(A - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.array_equal(A != 0, ref != 0) or np.abs(A - ref)[(A != 0) != (ref != 0)].max() <= 1e-300


def test_ufunc_keyword_arguments(golden_extras):
    """where= / out= of the batched wrappers (bspy/spline.py:943-947: np.frompyfunc honours them): only the
    selected points are evaluated, the rest is NaN or keeps what out held; out arrays receive the results."""
    g = golden_extras
    surf = Spline(2, 3, [4, 3], [8, 5], [g["normal_surf_knots0"], g["normal_surf_knots1"]], g["normal_surf_coefs"])
    u, v, mask, ref = g["where_u"], g["where_v"], g["where_mask"], g["where_result"]
    res = surf.evaluate(u, v, where=mask)
    assert isinstance(res, tuple) and len(res) == 3
    got = np.stack(res)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.nanmax(np.abs(got - ref)) <= 1e-13
    bad_u = u.copy()
    bad_u[~mask] = 7.0                                   # outside the domain, but never evaluated
    assert np.array_equal(np.stack(surf(bad_u, v, where=mask)), got, equal_nan=True)
    with pytest.raises(ValueError, match="outside domain"):
        surf(bad_u, v)
    outs = tuple(np.full(12, 7.0, dtype=object) for _ in range(3))       # the reference needs object arrays here
    res = surf.evaluate(u, v, where=mask, out=outs)
    full = np.stack(surf(u, v))
    for d in range(3):
        assert np.array_equal(res[d][~mask], np.full((~mask).sum(), 7.0)) and np.array_equal(res[d][mask], full[d][mask])
        assert all(float(outs[d][i]) == res[d][i] for i in range(12))
    fouts = tuple(np.zeros(12) for _ in range(3))                         # float arrays are accepted too (extension)
    surf.derivative([1, 0], u, v, out=fouts)
    assert np.array_equal(np.stack(fouts), np.stack(surf.derivative([1, 0], u, v)))
    with pytest.raises(TypeError):
        surf(u, v, casting="unsafe")


def test_uniform_path_declines_knots_far_from_the_origin():
    """Equally spaced knots whose rounding is large against their spacing (lo = 1e6, h = 1e-3: the stored knots are
    1e-7 spans away from lo + j h) must not take the table-free kernels: the reference works from the STORED knots."""
    rng = np.random.default_rng(5)
    order, ncoef = 4, (24, 24)
    knots = [_uniform_knots(order, nc, 1.0e6, 1.0e6 + 0.021) for nc in ncoef]
    coefs = rng.standard_normal((3, *ncoef))
    t = DeviceSpline((order, order), ncoef, knots, coefs)
    pts = [rng.uniform(k[order - 1], k[nc], 5000) for k, nc in zip(knots, ncoef)]
    for w in ([0, 0], [2, 1]):
        out = t.evaluate(pts, w)
        assert t.last_kernel() == "eval_rowrot"
        orc, bad = oracle.c_evaluate((order, order), ncoef, knots, coefs, w, pts)
        assert bad == -1 and np.abs(out - orc).max() <= 1e-12 * _scale(orc)
    near = DeviceSpline((order, order), ncoef, [_uniform_knots(order, nc, 10.0, 10.5) for nc in ncoef], coefs)
    near.evaluate([rng.uniform(10.0, 10.5, 100), rng.uniform(10.0, 10.5, 100)])
    assert near.last_kernel() == "eval_uni"


def test_uniform_path_declines_perturbed_knots():
    """Knots moved by a few hundred ulp of the span are the user's data, not rounding: the table-free kernels (which
    form every alpha from the nominal span width) must decline them, and the general kernels meet the parity bar."""
    rng = np.random.default_rng(6)
    order, ncoef = 4, (64, 64)
    knots = [_uniform_knots(order, nc, 0.0, 1.0) for nc in ncoef]
    h = 1.0 / (ncoef[0] - order + 1)
    for k in knots:
        k[order:ncoef[0]] += rng.uniform(-500, 500, ncoef[0] - order) * np.finfo(np.float64).eps * h
    coefs = rng.standard_normal((3, *ncoef))
    t = DeviceSpline((order, order), ncoef, knots, coefs)
    pts = [rng.random(20000), rng.random(20000)]
    worst = 0.0
    for w in ([0, 0], [1, 2]):
        out = t.evaluate(pts, w)
        assert t.last_kernel() == "eval_rowrot"
        orc, bad = oracle.c_evaluate((order, order), ncoef, knots, coefs, w, pts)
        worst = max(worst, float(np.abs(out - orc).max() / _scale(orc)))
    print(f"perturbed knots on the general kernels: worst error {worst:.2e} of the scale")
    assert worst <= 1e-13


@pytest.mark.parametrize("order,ncoef,dom,clamps,nDep,dt,expect", [
    (4, (40,), ((0.0, 1.0),), (True, True), 3, np.float64, True),                                  # cubic curve
    (5, (33,), ((-1.0, 2.0),), (True, True), 1, np.float64, True),
    (3, (12,), ((0.0, 1.0),), (False, True), 2, np.float64, True),                                 # continued, not clamped, at one end
    (1, (7,), ((0.0, 1.0),), (True, True), 2, np.float64, True),                                   # piecewise constant
    (2, (19,), ((0.0, 3.0),), (True, True), 2, np.float32, True),                                  # fp32: order <= 2 only
    (4, (25,), ((0.0, 1.0),), (True, True), 2, np.float32, False),
    (3, (20, 20), ((0.0, 1.0), (0.0, 2.0)), (True,) * 4, 3, np.float64, True),                      # surfaces of order 3 / 5
    (5, (30, 30), ((0.0, 1.0), (0.0, 1.0)), (True,) * 4, 3, np.float64, True),
    (5, (11, 14), ((2.0, 3.0), (-1.0, 0.0)), (True, False, True, True), 1, np.float64, True),
    (4, (8, 9, 10), ((0.0, 1.0),) * 3, (True,) * 6, 1, np.float64, True),                           # volumes
    (3, (12, 12, 12), ((0.0, 1.0), (0.0, 2.0), (1.0, 2.0)), (True,) * 6, 3, np.float64, True),
    (2, (6, 7, 8), ((0.0, 1.0),) * 3, (True,) * 6, 2, np.float64, True),
    (5, (10, 10, 10), ((0.0, 1.0),) * 3, (True,) * 6, 1, np.float64, False),                        # three clamped variables of order 5: declined
])
def test_uniform_knot_path_other_shapes(order, ncoef, dom, clamps, nDep, dt, expect):
    """Curves, surfaces of order 1 / 3 / 5 and volumes with equally spaced knots: eval_stream_uni / jac_stream_uni
    (table-free front end on the unclamped image) against the oracle - every derivative multi-index up to order + 1
    in total, the fused jacobian, points on every knot and one ulp either side, NaN, out-of-domain index, update."""
    rng = np.random.default_rng(123)
    nInd = len(ncoef)
    tol = 1e-12 if dt == np.float64 else 2e-5
    knots = [_uniform_knots(order, nc, lo, hi, clamps[2 * i], clamps[2 * i + 1]).astype(dt) for i, (nc, (lo, hi)) in enumerate(zip(ncoef, dom))]
    coefs = rng.standard_normal((nDep, *ncoef)).astype(dt)
    orders = (order,) * nInd
    t = DeviceSpline(orders, ncoef, knots, coefs, dt)
    n = 12000
    dm = [(float(k[order - 1]), float(k[nc])) for k, nc in zip(knots, ncoef)]
    pts = [(lo + (hi - lo) * rng.random(n)).astype(dt).clip(dt(lo), dt(hi)) for lo, hi in dm]
    at = 0
    for i, (k, nc) in enumerate(zip(knots, ncoef)):
        kk = np.unique(k[order - 1:nc + 1])
        edge = np.concatenate([kk, np.nextafter(kk[1:], dt(-np.inf)), np.nextafter(kk[:-1], dt(np.inf))]).astype(dt)
        pts[i][at:at + edge.size] = edge
        at += edge.size
    for i in range(nInd):
        pts[i][-1], pts[i][-2] = dm[i][1], dm[i][0]
    import itertools
    wrts = [w for w in itertools.product(range(order + 1), repeat=nInd) if sum(w) <= order + 1]
    ek, jk = ("eval_stream_uni", "jac_stream_uni") if expect else ("eval_stream", "jac_stream")
    for w in wrts:
        out = t.evaluate(pts, list(w))
        assert t.last_kernel() == ek, (t.last_kernel(), w)
        orc, bad = oracle.c_evaluate(orders, ncoef, knots, coefs, list(w), pts)
        assert bad == -1
        assert np.abs(out - orc).max() <= tol * _scale(orc), (w, np.abs(out - orc).max(), _scale(orc))
    jac = t.jacobian(pts)
    assert t.last_kernel() == jk or (not expect and t.last_kernel() == "jac_fixed")      # volumes of order 5
    orj, _ = oracle.c_jacobian(orders, ncoef, knots, coefs, pts)
    assert np.abs(jac - orj).max() <= tol * _scale(orj)
    if not expect:
        return
    # NaN parameters propagate (derivative levels do not multiply by u: the span rule decides), out-of-domain by index
    bad_pts = [p.copy() for p in pts]
    bad_pts[0][123] = np.nan
    if order > 1:
        assert np.isnan(t.evaluate(bad_pts)[:, 123]).all()
    else:       # piecewise constant: NaN sorts to the end, the last coefficient comes back (reference and oracle)
        o1, _ = oracle.c_evaluate(orders, ncoef, knots, coefs, [0] * nInd, [p[123:124] for p in bad_pts])
        assert np.array_equal(t.evaluate(bad_pts)[:, 123], o1[:, 0])
    if order > 1:
        w = [order - 1] + [0] * (nInd - 1)
        d3 = t.evaluate(bad_pts, w)[:, 123]
        o3, _ = oracle.c_evaluate(orders, ncoef, knots, coefs, w, [p[123:124] for p in bad_pts])
        assert np.allclose(d3, o3[:, 0], rtol=100 * tol, atol=100 * tol * _scale(o3), equal_nan=True)
    bad_pts[0][123] = dt(0.5 * (dm[0][0] + dm[0][1]))
    bad_pts[-1][4567] = np.nextafter(dt(dm[-1][1]), dt(np.inf))
    with pytest.raises(bspy_amd.DomainError) as e:
        t.evaluate(bad_pts)
    assert e.value.index == 4567
    coefs2 = coefs.copy()
    coefs2[:, 0] += 1.0
    t.update(knots, coefs2)
    out2 = t.evaluate(pts)
    assert t.last_kernel() == ek
    orc2, _ = oracle.c_evaluate(orders, ncoef, knots, coefs2, [0] * nInd, pts)
    assert np.abs(out2 - orc2).max() <= tol * _scale(orc2)



