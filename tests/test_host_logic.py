"""
CPU-only tests of the host logic: constructor semantics and error strings against the
reference's recorded behaviour, dispatch helpers, the C-ABI library's exported symbols,
the shard plan, and the multi-rank path under gloo (world_size 2) with the CPU oracle
injected as the per-rank evaluator.  No GPU compute is attempted here.
"""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import cases
import oracle
import bspy_amd
from bspy_amd import Spline
from bspy_amd import _native, _spline_evaluation as ev
from bspy_amd.sharding import shard_bounds, shard_chunk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "bspy_amd.h")).read()
    internal_block = header[header.index("#ifdef BSK_INTERNAL"):]
    internal_block = internal_block[:internal_block.index("#endif")]
    internal = set(re.findall(r"\b(bsk_[a-z_]+)\s*\(", internal_block))
    declared = set(re.findall(r"\b(bsk_[a-z_]+)\s*\(", header))
    declared -= {"bsk_status"}
    assert internal == set(_native.INTERNAL_SYMBOLS)                      # measurement hooks: behind BSK_INTERNAL only
    assert declared - internal == set(_native.PRODUCT_SYMBOLS)
    assert declared == set(_native.SYMBOLS)
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _native.lib().bsk_version() == 1
    # constants mirrored in Python match the header
    assert int(re.search(r"#define BSK_MAX_NIND (\d+)", header).group(1)) == _native.BSK_MAX_NIND
    assert int(re.search(r"#define BSK_MAX_ORDER (\d+)", header).group(1)) == _native.BSK_MAX_ORDER


def test_no_cpu_fallback_without_gpu():
    """Without a GPU (this container) every compute entry point raises: nothing is computed
    on the host behind the user's back."""
    if _native.device_count() > 0:
        pytest.skip("a GPU is present")
    s = Spline(1, 1, [4], [4], [[0, 0, 0, 0, 1, 1, 1, 1.0]], [[0.0, 1, 2, 3]])
    with pytest.raises(bspy_amd.BskError):
        s(0.5)
    with pytest.raises(bspy_amd.BskError):
        s(np.linspace(0, 1, 10))
    with pytest.raises(bspy_amd.BskError):
        s.jacobian([0.5])
    with pytest.raises(bspy_amd.BskError):
        Spline.bspline_values(None, np.array([0, 0, 1, 1.0]), 2, 0.5)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "libbspy_amd.so"))
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        _native.lib()


def test_product_does_not_import_the_oracle():
    out = subprocess.run(["grep", "-rIl", "-E", r"^\s*(import|from)\s+oracle", os.path.join(ROOT, "bspy_amd")],
                         capture_output=True, text=True).stdout.strip()
    assert out == "", out


def test_constructor_matches_reference(golden_api):
    a = golden_api
    k = [0, 0, 0, 0, .3, .3, .7, 1, 1, 1, 1]

    def err(f):
        try:
            f()
        except Exception as e:  # noqa: BLE001
            return [type(e).__name__, str(e)]
        return None

    got = {
        "nInd": err(lambda: Spline(-1, 1, [], [], [], [])),
        "order": err(lambda: Spline(1, 1, [4, 4], [4], [k], [[0, 1, 2, 3]])),
        "nCoef": err(lambda: Spline(1, 1, [4], [4, 4], [k], [[0, 1, 2, 3]])),
        "nknots": err(lambda: Spline(1, 1, [4], [4], [k], [[0.0, 1, 2, 3]])),
        "knots_len": err(lambda: Spline(1, 1, [4], [4], [k, k], [[0.0, 1, 2, 3]])),
        "knot_order": err(lambda: Spline(1, 1, [4], [4], [[0, 0, 0.5, 0.2, 1, 1, 1, 1]], [[0.0, 1, 2, 3]])),
        "knot_mult": err(lambda: Spline(1, 1, [4], [5], [[0, 0, 0, 0, 0, 1, 1, 1, 1.0]], [[0.0, 1, 2, 3, 4]])),
        "coefs_len": err(lambda: Spline(1, 2, [4], [4], [[0, 0, 0, 0, 1, 1, 1, 1.0]], [[0.0, 1, 2]])),
    }
    assert got == a["err_ctor"]
    k2 = [[0, 0, 0, .5, 1, 1, 1], [0, 0, 0, 0, .5, 1, 1, 1, 1]]
    s3 = Spline(2, 3, [3, 4], [4, 5], k2, np.arange(60.0).reshape(20, 3))
    assert s3.coefs.shape == (3, 4, 5)
    assert np.array_equal(np.ascontiguousarray(s3.coefs), np.array(a["flat_coefs"]))
    per = [np.arange(20.0).reshape(5, 4) + 100 * d for d in range(3)]
    s4 = Spline(2, 3, [3, 4], [4, 5], k2, per)
    assert np.array_equal(np.ascontiguousarray(s4.coefs), np.array(a["perdep_coefs"]))
    assert np.array_equal(s4.domain(), np.array(a["domain"]))
    s0 = Spline(0, 2, [], [], [], [1.5, 2.5])
    assert np.asarray(s0.evaluate()).tolist() == a["nind0_eval"]
    assert np.asarray(s0.derivative([])).tolist() == a["nind0_deriv"]
    # dtype rules: float32 stays float32, integers are promoted (documented deviation)
    f = Spline(1, 1, [2], [2], [np.array([0, 0, 1, 1], np.float32)], np.array([[0, 1]], np.float32))
    assert f.knots[0].dtype == np.float32 and f.coefs.dtype == np.float32 and ev.compute_dtype(f) == np.float32
    i = Spline(1, 1, [2], [2], [[0, 0, 1, 1]], [[0, 1]])
    assert i.knots[0].dtype == np.float64 and i.coefs.dtype == np.float64
    m = Spline(1, 1, [2], [2], [np.array([0, 0, 1, 1], np.float32)], np.array([[0, 1]], np.float64))
    assert ev.compute_dtype(m) == np.float64
    assert isinstance(s4.metadata, dict) and Spline(1, 1, [2], [2], [[0, 0, 1, 1.0]], [[0, 1.0]], {"Name": "x"}).metadata == {"Name": "x"}


def test_arity_and_kwargs_errors():
    s2 = Spline(2, 3, [3, 4], [4, 5], [[0, 0, 0, .5, 1, 1, 1], [0, 0, 0, 0, .5, 1, 1, 1, 1]],
                np.arange(60.0).reshape(3, 4, 5))
    with pytest.raises(ValueError, match="Incorrect number of parameter values: 1"):
        s2.evaluate([0.1])
    with pytest.raises(ValueError, match="Incorrect number of parameter values: 3"):
        s2.jacobian([0.1, 0.2, 0.3])
    with pytest.raises(TypeError, match="ufunc keyword"):
        s2(np.zeros(4), np.zeros(4), casting="unsafe")


def test_grid_detection():
    u, v, w = np.zeros(5), np.zeros(7), np.zeros(3)
    assert ev._grid_axes([u[:, None], v[None, :]], (5, 7)) == [0, 1]
    assert ev._grid_axes([u[None, :], v[:, None]], (7, 5)) == [1, 0]
    assert ev._grid_axes([u[:, None, None], v[None, :, None], w[None, None, :]], (5, 7, 3)) == [0, 1, 2]
    assert ev._grid_axes([u, np.zeros(5)], (5,)) is None                      # same axis twice: not a grid
    assert ev._grid_axes([np.zeros((5, 7)), v[None, :]], (5, 7)) is None      # 2-D parameter array
    assert ev._grid_axes([u[:, None], np.float64(0.5)], (5, 1)) == [0, None]  # scalar second variable
    assert ev._grid_axes([u[:, None], np.zeros((1, 1))], (5, 7)) is None      # axis 1 of the result driven by nobody


def test_shard_plan():
    for n in (0, 1, 7, 8, 9, 1000, 10_000_000):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                a, b = shard_bounds(n, world, r)
                assert 0 <= a <= b <= n and b - a <= shard_chunk(n, world)
                covered.extend(range(a, b) if n < 2000 else [])
                if r:
                    assert a == shard_bounds(n, world, r - 1)[1]
            assert shard_bounds(n, world, world - 1)[1] == n
            if n < 2000:
                assert covered == list(range(n))


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import cases, oracle
from bspy_amd import Spline
from bspy_amd.sharding import ShardedEvaluator, shard_bounds
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=int(sys.argv[4]))
rank, world = dist.get_rank(), dist.get_world_size()
c = {x.name: x for x in cases.parity_cases()}["cfg2_bicubic"]
s = Spline(c.nInd, c.nDep, c.order, c.nCoef, c.knots, c.coefs)

def cpu_checker(op, pts, wrt):      # tests may use the oracle; the product path never does
    if op == "jacobian":
        out, bad = oracle.c_jacobian(c.order, c.nCoef, c.knots, c.coefs, pts)
        return (None, bad) if bad >= 0 else (out.reshape(-1, len(pts[0])), -1)
    out, bad = oracle.c_evaluate(c.order, c.nCoef, c.knots, c.coefs, wrt or [0, 0], pts)
    return (None, bad) if bad >= 0 else (out, -1)

sh = ShardedEvaluator(s, local_eval=cpu_checker)
n = 1001                                            # not divisible by the world size: ragged tail shard
pts = [p[:n] for p in c.points]
full, _ = oracle.c_evaluate(c.order, c.nCoef, c.knots, c.coefs, [0, 0], pts)
got = sh.evaluate(pts)
assert got.shape == full.shape and np.array_equal(got, full), "gathered evaluate"
a, b = shard_bounds(n, world, rank)
loc = sh.evaluate(pts, gather=False)
assert np.array_equal(loc, full[:, a:b]), "sharded evaluate"
d = sh.derivative([1, 0], [p[a:b] for p in pts], sharded_input=True, total=n)
fd, _ = oracle.c_evaluate(c.order, c.nCoef, c.knots, c.coefs, [1, 0], pts)
assert np.array_equal(d, fd), "pre-sharded derivative"
j = sh.jacobian(pts)
fj, _ = oracle.c_jacobian(c.order, c.nCoef, c.knots, c.coefs, pts)
assert np.array_equal(j, fj.reshape(-1, n)), "gathered jacobian"
bad = [p.copy() for p in pts]
bad[1][900] = 7.0                                   # lives in the last rank's shard
bad[0][950] = -3.0
try:
    sh.evaluate(bad)
    raise SystemExit("no error raised")
except ValueError as e:
    assert "outside domain" in str(e) and "flat index 900" in str(e), str(e)   # every rank reports the first offender
# check=False: no agreement step.  The offending rank (the last one: index 900) must still take part in the
# all-gather - it raises afterwards - and the other rank returns with NaN rows for that shard (no hang).
owner = [r for r in range(world) if shard_bounds(n, world, r)[0] <= 900 < shard_bounds(n, world, r)[1]][0]
try:
    got = sh.evaluate(bad, check=False)
    assert rank != owner, "the offending rank must raise"
    oa, ob = shard_bounds(n, world, owner)
    assert np.isnan(got[:, oa:ob]).all() and np.array_equal(got[:, :oa], full[:, :oa]), "NaN rows of the offending shard"
except ValueError as e:
    assert rank == owner and "flat index 900" in str(e), str(e)
try:
    loc = sh.jacobian(bad, gather=False, check=False)        # no collective at all: the offender raises at once
    assert rank != owner
except ValueError as e:
    assert rank == owner
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_evaluator_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + os.getpid() % 2000
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r), "2"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out}"
        assert f"rank {r} ok" in out


def test_result_pool_recycles_only_dead_results():
    """Large host results reuse the memory of results the caller has DROPPED (no first-touch page faults), and never
    memory that any array still refers to: the rows of `x, y, z = s(u, v)` outlive the 2-D result they are views of."""
    import gc
    from bspy_amd.result_pool import ResultPool, MIN_BYTES
    pool = ResultPool()
    small = pool.empty((3, 10), np.float64)
    assert small.base is None and pool.allocated == 0                      # small results: plain arrays
    shape = (3, MIN_BYTES // 8)
    a = pool.empty(shape, np.float64)
    assert a.shape == shape and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable and a.flags.aligned
    addr = a.ctypes.data
    a[:] = 7.0
    x, y, z = a[0], a[1], a[2]                                             # the reference returns a tuple of rows
    del a
    gc.collect()
    b = pool.empty(shape, np.float64)
    assert b.ctypes.data != addr and pool.recycled == 0                    # rows alive: their memory is NOT handed out
    b[:] = 1.0
    assert float(x[0]) == 7.0 and float(z[-1]) == 7.0
    del x, y
    gc.collect()
    c = pool.empty(shape, np.float32)                                      # still one row alive
    assert c.ctypes.data != addr
    del z
    gc.collect()
    d = pool.empty(shape, np.float64)
    assert d.ctypes.data == addr and pool.recycled == 1                    # the last view is gone: recycled
    e = pool.empty((2, 5, MIN_BYTES // 8), np.float64)                     # larger than anything free: new memory
    assert e.shape == (2, 5, MIN_BYTES // 8) and pool.recycled == 1
    del b, c, d, e
    gc.collect()
    assert len(pool._free) <= 4


def test_json_round_trip_and_reference_files(tmp_path):
    """The reference's JSON format as an input format: files from the reference's
    own test suite load, and save -> load reproduces the spline."""
    ref_dir = os.path.join(ROOT, "tests", "golden", "reference_json")
    loaded = Spline.load(os.path.join(ref_dir, "offset-issue.json"))
    assert len(loaded) == 5 and all(s.nInd == 1 and s.nDep == 2 and s.order == (4,) for s in loaded)
    s = Spline.load(os.path.join(ref_dir, "reverse-thing.json"))[0]
    assert (s.nInd, s.nDep, s.order, s.nCoef) == (1, 2, (7,), (112,)) and s.coefs.shape == (2, 112)
    path = tmp_path / "out.json"
    k2 = [[0, 0, 0, .5, 1, 1, 1], [0, 0, 0, 0, .5, 1, 1, 1, 1]]
    a = Spline(2, 3, [3, 4], [4, 5], k2, np.arange(60.0).reshape(3, 4, 5), {"Name": "a", "flipNormal": True})
    a.save(str(path), s)
    back = Spline.load(str(path))
    assert len(back) == 2
    assert back[0].order == a.order and back[0].nCoef == a.nCoef
    assert np.array_equal(back[0].coefs, a.coefs) and all(np.array_equal(x, y) for x, y in zip(back[0].knots, a.knots))
    assert back[0].metadata == {"Name": "a", "negateNormal": True}          # legacy key translated on load
    assert np.array_equal(back[1].coefs, s.coefs)
    d = a.to_dict()
    assert d["type"] == "Spline" and Spline.from_dict(d).nDep == 3


def test_collocation_derivative_orders():
    from bspy_amd.collocation import derivative_orders
    assert derivative_orders([0, 0, 0.1, 0.2, 0.2, 0.2, 0.3]).tolist() == [0, 1, 0, 0, 1, 2, 0]
    assert derivative_orders([]).tolist() == []


# ---------------------------------------------------------------------------------------------
# build-time guard of the inline-asm LDS kernels (bspy_amd/csrc/check_lds_hazards.py)
# ---------------------------------------------------------------------------------------------
def _hazards(body):
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "check_lds_hazards", os.path.join(ROOT, "bspy_amd", "csrc", "check_lds_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = list(enumerate(("_ZN3bsk11eval_rowrotEv:\n" + body + "\ts_endpgm\n").splitlines(True), 1))
    return mod.check_kernel("k", lines[1:])


def test_lds_hazard_checker_accepts_counted_waits():
    ok = """
\tds_read_b64 v[10:11], v2 offset:0
\tds_read_b64 v[12:13], v2 offset:8
\tv_add_u32_e32 v3, 1, v3
\ts_waitcnt lgkmcnt(1)
\tv_mul_f64 v[20:21], v[10:11], v[4:5]
\ts_waitcnt lgkmcnt(0)
\tv_mul_f64 v[22:23], v[12:13], v[4:5]
"""
    assert _hazards(ok) == []


def test_lds_hazard_checker_flags_use_before_wait():
    copy_before_wait = """
\tds_read_b64 v[10:11], v2 offset:0
\tds_read_b64 v[12:13], v2 offset:8
\ts_waitcnt lgkmcnt(1)
\tv_mov_b32_e32 v30, v12
\ts_waitcnt lgkmcnt(0)
"""
    errs = _hazards(copy_before_wait)
    assert len(errs) == 1 and "v12" in errs[0][1]
    overwritten = """
\tds_read_b32 v7, v2
\tv_mov_b32_e32 v7, 0
\ts_waitcnt lgkmcnt(0)
"""
    assert len(_hazards(overwritten)) == 1
    # packed fp32 math names a pair but reads only the selected halves
    packed = """
\tds_read_b32 v11, v2
\tv_pk_mul_f32 v[32:33], v[4:5], v[10:11] op_sel_hi:[1,0]
\ts_waitcnt lgkmcnt(0)
"""
    assert _hazards(packed) == []
    assert len(_hazards(packed.replace(" op_sel_hi:[1,0]", ""))) == 1


def test_lds_hazard_checker_follows_control_flow():
    # the read is awaited on one path only: the use after the join is a hazard
    one_sided = """
\tds_read_b64 v[10:11], v2 offset:0
\ts_cbranch_vccnz .LBB0_2
\ts_waitcnt lgkmcnt(0)
.LBB0_2:
\tv_add_f64 v[20:21], v[10:11], v[4:5]
"""
    assert len(_hazards(one_sided)) == 1
    # a read left in flight around a loop piles up
    loop = """
.LBB0_1:
\tds_read_b64 v[10:11], v2 offset:0
\ts_cbranch_vccnz .LBB0_1
\ts_waitcnt lgkmcnt(0)
"""
    assert len(_hazards(loop)) >= 1
    # drained before the back edge: fine
    drained = """
.LBB0_1:
\tds_read_b64 v[10:11], v2 offset:0
\ts_waitcnt lgkmcnt(0)
\tv_add_f64 v[20:21], v[10:11], v[4:5]
\ts_cbranch_vccnz .LBB0_1
"""
    assert _hazards(drained) == []


# ---------------------------------------------------------------------------------------------
# SplineBlock (reference bspy/spline_block.py): constructor semantics and the oracle restatement of
# its evaluation path against goldens recorded from the reference (tests/golden/block.npz)
# ---------------------------------------------------------------------------------------------
def _block_oracle(c, kind, wrt=None, m=None):
    """rows of the block = sums of the oracle's batched evaluations on the mapped variables"""
    import oracle
    pts = [p[:m] for p in c.points]
    n = len(pts[0])
    out = np.zeros((c.nDep, c.nInd, n) if kind == "jacobian" else (c.nDep, n))
    r0 = 0
    for row in c.rows:
        k = row[0][1][1]
        for imap, (nind, ndep, order, ncoef, knots, coefs) in row:
            sub = [pts[i] for i in imap]
            if kind == "jacobian":
                j, bad = oracle.c_jacobian(order, ncoef, knots, coefs.astype(np.float64), sub)
                out[r0:r0 + k, imap] += j
            else:
                w = [0] * nind if wrt is None else [wrt[i] for i in imap]
                v, bad = oracle.c_evaluate(order, ncoef, knots, coefs.astype(np.float64), w, sub)
                out[r0:r0 + k] += v
            assert bad == -1
        r0 += k
    return out


def test_spline_block_oracle_matches_reference_goldens():
    g = np.load(os.path.join(ROOT, "tests", "golden", "block.npz"))
    for c in cases.block_cases():
        m = g[f"{c.name}/evaluate"].shape[1]
        assert tuple(g[f"{c.name}/nInd_nDep"]) == (c.nInd, c.nDep)
        ref = g[f"{c.name}/evaluate"]
        tol = 1e-5 if "f32" in c.name else 1e-12
        assert np.abs(_block_oracle(c, "evaluate", m=m) - ref).max() <= tol * max(1.0, np.abs(ref).max())
        ref = g[f"{c.name}/jacobian"]
        assert np.abs(_block_oracle(c, "jacobian", m=m) - ref).max() <= tol * 10 * max(1.0, np.abs(ref).max())
        for w in c.wrts:
            ref = g[f"{c.name}/wrt_" + "_".join(map(str, w))]
            assert np.abs(_block_oracle(c, "evaluate", w, m=m) - ref).max() <= tol * 100 * max(1.0, np.abs(ref).max())


def test_spline_block_constructor_semantics():
    from bspy_amd import Spline, SplineBlock
    c = cases.block_cases()[1]
    rows = [[(imap, Spline(*d)) for (imap, d) in row] for row in c.rows]
    g = np.load(os.path.join(ROOT, "tests", "golden", "block.npz"))
    b = SplineBlock(rows)
    assert (b.nInd, b.nDep, b.size) == (c.nInd, c.nDep, 3)
    assert np.array_equal(b.domain(), g[f"{c.name}/domain"])
    # a bare spline and a bare row are promoted (reference spline_block.py:55-58)
    s = rows[0][0][1]
    assert SplineBlock(s).nInd == s.nInd and SplineBlock([s]).nDep == s.nDep
    with pytest.raises(ValueError, match="All splines in the same row must have the same nDep"):
        SplineBlock([[rows[0][0], ([0, 1], rows[1][0][1].__class__(2, 1, (2, 2), (3, 3), [np.array([0., 0, .5, 1, 1]), np.array([0., 0, 1, 2, 2])], np.zeros((1, 3, 3))))]])
    with pytest.raises(ValueError, match="Multiple splines in the same row map to independent variable 3"):
        SplineBlock([[rows[0][0], ([3, 0], rows[0][1][1])]])
    with pytest.raises(ValueError, match="Domains of independent variables must match"):
        SplineBlock([[rows[0][0]], [([3, 2], rows[0][1][1])]])
    with pytest.raises(ValueError, match="Block is missing independent variable 0"):
        SplineBlock([[([1, 2], rows[0][1][1])]])


# ---------------------------------------------------------------------------------------------
# tessellation goldens (tests/golden/tess.npz, recorded from the reference): the oracle's
# evaluate / normal on the grid points must reproduce them
# ---------------------------------------------------------------------------------------------
def _tess_batches():
    tables = np.load(os.path.join(ROOT, "tests", "golden", "reference_tables.npz"))
    b = dict(cases.tess_cases())
    b["teapot_f32"] = (cases.teapot_patches(tables, which=(0, 5, 13, 31)), np.linspace(0, 1, 8, dtype=np.float32),
                       np.linspace(0, 1, 12, dtype=np.float32))
    return b


def test_tessellation_oracle_matches_reference_goldens():
    import oracle
    g = np.load(os.path.join(ROOT, "tests", "golden", "tess.npz"))
    for name, (patches, u, v) in _tess_batches().items():
        uu, vv = [a.reshape(-1) for a in np.meshgrid(u, v, indexing="ij")]
        tol = 2e-5 if "f32" in name else 1e-12
        for p, (order, ncoef, knots, coefs) in enumerate(patches):
            pos, _ = oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0], [uu, vv])
            ref = g[f"{name}/positions"][p].reshape(3, -1)
            assert np.abs(pos - ref).max() <= tol * max(1.0, np.abs(ref).max()), (name, p)
            nrm, _ = oracle.c_normal(order, ncoef, knots, coefs, [uu, vv], True, False)
            ref = g[f"{name}/normals"][p].reshape(3, -1)
            ok = np.isfinite(ref).all(axis=0) & np.isfinite(nrm).all(axis=0)
            assert ok.sum() >= 0.8 * ok.size
            assert np.abs(nrm[:, ok] - ref[:, ok]).max() <= (2e-3 if "f32" in name else 1e-9), (name, p)


def test_host_code_under_address_and_thread_sanitizer(tmp_path):
    """The host half of bsk_api.hip (handle life cycle, workspaces, the small-call path's pinned buffer, the
    pipelined BSK_HOST path: copy-thread pool + double-buffered pinned staging + three streams) compiled
    host-only with AddressSanitizer / ThreadSanitizer against a host-memory stand-in for the HIP runtime
    (tests/hipstub/) and driven from two threads on two handles.  GPU sanitizer runs are not available on the
    GPU pool; this is the CPU-side hardening SURVEY.md section 5 asks for."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    stub = os.path.join(ROOT, "tests", "hipstub")
    procs = {k: subprocess.Popen(["bash", os.path.join(stub, "build.sh"), k, str(tmp_path / k)], stdout=subprocess.PIPE,
                                 stderr=subprocess.STDOUT, text=True) for k in ("asan", "tsan")}
    for k, p in procs.items():
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, f"{k} build failed:\n{out[-3000:]}"
    for k, big in (("asan", "4500000"), ("tsan", "2200000")):
        # three fake devices: the multi-device entry points run the REAL single-device host code on each of them
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=1",
                   HIPSTUB_DEVICES="3", LD_LIBRARY_PATH=str(tmp_path / k) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([str(tmp_path / k / "driver"), big], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and "hipstub driver ok" in r.stdout, f"{k}:\n{r.stdout[-2000:]}\n{r.stderr[-6000:]}"
        assert "Sanitizer" not in r.stderr, r.stderr[-6000:]


def test_multi_device_entry_points_on_fake_devices(tmp_path):
    """bsk_multi_evaluate / bsk_multi_jacobian (csrc/bsk_multi.hip, compiled host-only and unchanged) on 1, 2, 3 and 8
    fake devices with a stub librccl (tests/hipstub/rccl_stub.cpp: grouped all-gather as memcpy) and a recognisable
    stand-in for the single-device layer (tests/hipstub/multi_driver.cpp): every value of every shard, staging block
    and gathered buffer is checked - ragged tail, empty tail shards, n = 0, first offender on the last device, a
    failing device in the middle (all devices drained, current device restored) - under ASAN and TSAN."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    stub = os.path.join(ROOT, "tests", "hipstub")
    procs = {k: subprocess.Popen(["bash", os.path.join(stub, "build_multi.sh"), k, str(tmp_path / k)], stdout=subprocess.PIPE,
                                 stderr=subprocess.STDOUT, text=True) for k in ("asan", "tsan")}
    for k, p in procs.items():
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, f"{k} build failed:\n{out[-3000:]}"
    for k in ("asan", "tsan"):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=1",
                   HIPSTUB_DEVICES="8", LD_LIBRARY_PATH=str(tmp_path / k) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([str(tmp_path / k / "multi_driver")], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and "multi driver ok: 8 devices, 48 cases" in r.stdout, f"{k}:\n{r.stdout[-2000:]}\n{r.stderr[-6000:]}"
        assert "Sanitizer" not in r.stderr, r.stderr[-6000:]
