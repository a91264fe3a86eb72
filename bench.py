"""
bench.py - BASELINE metric: M evals/s on the 10 M-point bicubic (order 4x4, nCoef 64x64,
nDep 3, fp64) surface (BASELINE.json configs[1]) at 1/2/4/8 GPUs, with the CPU path beside it.

    python bench.py --gpus 1 --steps 200 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (Spline.evaluate: span search + Cox-de Boor recursion +
tensor-product contraction, fused in one HIP kernel) over the 10 M synthetic (u, v) points of the
metric, inputs and outputs resident in HBM.  Multi-GPU: STRONG scaling - the same 10 M points are
sharded contiguously over the ranks (one process per GPU, no data-path collective in the timed
region; results stay sharded), value = 10 M / max-over-ranks time.  Beside the headline, N > 1 lines
carry "weak" (10 M points per rank) and "with_allgather" (the RCCL all-gather of results included).

Prints ONE JSON line (rank 0).  roofline.achieved uses the ALGORITHMIC bytes of SURVEY.md 8(d):
40 B per eval (2 x 8 B in + 3 x 8 B out) against 8 TB/s HBM3E, over the average launch duration
measured with HIP events on the launch stream inside the timed region.

`--config {2,3,4,5}` selects the BASELINE config the job runs (default 2 = the metric's; the others are the
parity-test configs, runnable sharded so that an 8-GPU node can drive each of them):
    3  same bicubic: derivative([1,1]) + fused jacobian per step (104 B per point), all-gather of (3,N) and (3,2,N)
    4  Utah teapot, 32 bicubic fp32 patches on a 2048 x 2048 grid each, sharded BY PATCH (12 B per grid point)
    5  trivariate order 5, 40^3 x 4, fp32, 50 M points (28 B per point): the cell-order pipeline, with its per-kernel
       times (sort_ms / eval_ms), and the fp32-ALU / MFMA fractions next to the HBM one
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
N_POINTS = 10_000_000
BYTES_PER_EVAL = {"evaluate": 40, "derivative": 40, "jacobian": 64, "normal": 40}
FP32_PEAK_TFLOPS = 157.3       # MI355X fp32 vector peak = fp32 MFMA peak (MI355X_MICROARCH.md)
CFG5_FLOP_PER_POINT = 1450     # algorithmic fp32 flop per cfg5 point
CFG5_MFMA_FLOP_PER_POINT = 1000  # the part of them the 4x4x1 MFMAs carry: 125 control points x 4 dependents x 2
PROFILE_DIR = "profiles/r03_eval_uni"    # rocprofv3 summaries of this command (kernel trace + PMC passes)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(order, ncoef, knots, coefs, sample):
    """The C oracle (a port of the reference's algorithm) on a bounded sample of the same workload:
    one thread (the reference is single-threaded), every host core (threads over contiguous slices;
    ctypes releases the GIL), and the reference-structured per-point Python loop on a smaller sample."""
    import concurrent.futures as cf
    import oracle
    rng = np.random.default_rng(12345)
    uv = rng.random((2, sample))
    oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0], [uv[0][:1000], uv[1][:1000]])   # load + warm
    t0 = time.perf_counter()
    _, bad = oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0], [uv[0], uv[1]])
    t1 = time.perf_counter()
    assert bad == -1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    bounds = np.linspace(0, sample, cores + 1).astype(np.int64)

    def part(i):
        a, b = int(bounds[i]), int(bounds[i + 1])
        return oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0], [uv[0][a:b], uv[1][a:b]])[1]

    with cf.ThreadPoolExecutor(cores) as ex:
        list(ex.map(part, range(cores)))                 # warm the pool
        t2 = time.perf_counter()
        bads = list(ex.map(part, range(cores)))
        t3 = time.perf_counter()
    assert all(b == -1 for b in bads)
    npy = 100_000
    t4 = time.perf_counter()
    oracle.py_batch(order, ncoef, knots, coefs, [0, 0], [uv[0][:npy], uv[1][:npy]])
    t5 = time.perf_counter()
    return {
        "value": round(sample / (t1 - t0) / 1e6, 4), "unit": "M evals/s", "cores": 1, "kind": "port",
        "sample": f"{sample} of the 10M random (u,v) points, C oracle (oracle/bspline_oracle.c), one thread",
        "all_cores": {"value": round(sample / (t3 - t2) / 1e6, 2), "unit": "M evals/s", "cores": cores,
                      "cpu_model": cpu_model(),
                      "sample": f"the same {sample} points, C oracle, {cores} threads over contiguous slices"},
        "python_reference_structured": {"value": round(npy / (t5 - t4) / 1e6, 5), "unit": "M evals/s", "cores": 1,
                                        "sample": f"{npy} points, per-point Python/NumPy loop (oracle.py_batch)"},
        "host_cpus": os.cpu_count(),
    }


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (<PROFILE_DIR>/pmc_per_launch.json: FETCH_SIZE and WRITE_SIZE in KB, separate passes; FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950 wide coalesced reads).  Returned only when the
    profile holds the kernel family that ran in THIS process; else None."""
    path = os.path.join(ROOT, PROFILE_DIR, "pmc_per_launch.json")
    try:
        with open(path) as f:
            prof = json.load(f)
        k = next(v for name, v in prof.items() if ("bsk::" + kernel + "<") in name and "double" in name)
        return int((2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024), os.path.join(PROFILE_DIR, "pmc_per_launch.json")
    except (OSError, StopIteration, KeyError, ValueError):
        return None, None


def timed(torch, f, steps, spin_s=0.06):
    t_end = time.perf_counter() + spin_s      # clock spin-up (CPU legs before this leave the GPU idle)
    while time.perf_counter() < t_end:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e-3


def row(sec, evals, bytes_per_eval):
    return {"ms": round(sec * 1e3, 4), "Mevals_s": round(evals / sec / 1e6, 1),
            "hbm_frac": round(bytes_per_eval * evals / sec / 1e9 / HBM_PEAK_GBS, 4)}


def stream_floor(torch, tables, u, v, n):
    """Memory-side floor of the evaluation kernels' launch geometry, measured in this process: the same
    2 x 8 B in + 3 x 8 B out per point streamed with trivial arithmetic (bsk_debug_probe)."""
    from bspy_amd import _native as nv
    out = torch.empty((3, n), dtype=torch.float64, device=u.device)
    st = ctypes.c_void_p(torch.cuda.current_stream(u.device).cuda_stream)

    def f():
        nv.check(nv.lib().bsk_debug_probe(tables._handle, 0, 1, 1024, 110000, u.data_ptr(), v.data_ptr(), n, out.data_ptr(), st))
    sec = timed(torch, f, 50)
    return round(40.0 * n / sec / 1e9, 1)


def other_configs(torch, tables, u, v, n):
    """The other BASELINE configs on this GPU, timed after the headline (diagnostic; device-resident
    I/O, steady state): the same cfg2 workload on the GENERAL-knot kernels (BSK_VARIANT=9: what a bicubic
    with arbitrary knots gets), cfg3 derivative / fused jacobian, cfg4 = the 32 teapot patches on a
    2048 x 2048 grid from one bsk_tessellate call, cfg5 = trivariate order 5, 40^3 x 4 fp32 on 10 M points."""
    import cases
    import bspy_amd
    res = {}
    try:
        out = torch.empty((3, n), dtype=torch.float64, device=u.device)
        jout = torch.empty((3, 2, n), dtype=torch.float64, device=u.device)
        nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
        old = os.environ.get("BSK_VARIANT")
        os.environ["BSK_VARIANT"] = "9"
        try:
            gen = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt, device=u.device.index)
        finally:
            if old is None:
                del os.environ["BSK_VARIANT"]
            else:
                os.environ["BSK_VARIANT"] = old
        r = row(timed(torch, lambda: gen.evaluate_device([u, v], out=out, check=False), 50), n, 40)
        r["kernel"] = gen.last_kernel()
        res["cfg2_general_knot_kernels"] = r
        gen.domain_status()
        r = row(timed(torch, lambda: tables.evaluate_device([u, v], [1, 1], out=out, check=False), 30), n, 40)
        r["kernel"] = tables.last_kernel()
        res["cfg3_derivative_1_1"] = r
        r = row(timed(torch, lambda: tables.jacobian_device([u, v], out=jout, check=False), 30), n, 64)
        r["kernel"] = tables.last_kernel()
        res["cfg3_jacobian_fused"] = r
        tables.domain_status()
        del out, jout
        g = np.load(os.path.join(ROOT, "tests", "golden", "reference_tables.npz"))
        patches = [bspy_amd.DeviceSpline(o, c, k, cf, np.float32) for (o, c, k, cf) in cases.teapot_patches(g)]
        gg = torch.linspace(0, 1, 2048, dtype=torch.float32, device=u.device)
        pos = torch.empty((32, 3, 2048, 2048), dtype=torch.float32, device=u.device)
        sec = timed(torch, lambda: bspy_amd.tessellate_tables(patches, (gg, gg), normals=False, out=(pos, None), check=False), 10)
        res["cfg4_teapot_32_patches_grid2048"] = row(sec, 32 * 2048 * 2048, 12)
        patches[0].domain_status()
        del pos
        nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(5)
        t5 = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
        p5 = [torch.rand(n, dtype=torch.float32, device=u.device) for _ in range(3)]
        o5 = torch.empty((4, n), dtype=torch.float32, device=u.device)
        f5 = lambda: t5.evaluate_device(p5, out=o5, check=False)
        r = row(timed(torch, f5, 10), n, 28)
        r["kernel"] = t5.last_kernel()
        # the bounds that apply to cfg5 (not HBM): fp32 ALU and the matrix pipe, and where the time goes
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from cfg5_stages import stage_times
        st_ms = stage_times(t5, f5, reps=10)
        ev = sum(v for k, v in st_ms.items() if k.startswith("eval"))
        r.update({"points": n, "stages_ms": {k: round(v, 4) for k, v in st_ms.items()},
                  "sort_ms": round(sum(v for k, v in st_ms.items() if not k.startswith("eval")), 4), "eval_ms": round(ev, 4),
                  "alu_frac": round(CFG5_FLOP_PER_POINT * n / (r["ms"] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                  "alu_frac_of_eval_kernel": round(CFG5_FLOP_PER_POINT * n / (ev * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4) if ev else None,
                  "mfma_frac_of_eval_kernel": round(CFG5_MFMA_FLOP_PER_POINT * n / (ev * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4) if ev else None,
                  "note": "alu_frac: 1450 algorithmic fp32 flop per point against the 157.3 TFLOP/s vector peak; mfma_frac: the 1000 flop per "
                          "point carried by v_mfma_f32_4x4x1 against the fp32 matrix peak (also 157.3), over the evaluation kernel's time; "
                          "stages: HIP events between the pipeline's kernels (bsk_debug_stage_times)"})
        res["cfg5_trivariate_f32"] = r
        t5.domain_status()
        del p5, o5
        # two shapes VERDICT r2 named (not BASELINE configs): the reference's examples/TomsNasty.json shape and an all-fp32 bicubic
        rng = np.random.default_rng(3)
        for key, order2, ncoef2, dt2, bpe in (("surface_o4x5_900x11_f64_TomsNasty_shape", (4, 5), (900, 11), np.float64, 40),
                                              ("surface_o3x4_20x20_f64_mixed_orders", (3, 4), (20, 20), np.float64, 40),
                                              ("bicubic_64x64_f32", (4, 4), (64, 64), np.float32, 20)):
            k2 = [cases.clamped_uniform_knots(o, c, dt2) for o, c in zip(order2, ncoef2)]
            t2 = bspy_amd.DeviceSpline(order2, ncoef2, k2, rng.standard_normal((3, *ncoef2)).astype(dt2), dt2)
            tdt2 = torch.float64 if dt2 == np.float64 else torch.float32
            p2 = [torch.rand(n, dtype=tdt2, device=u.device) for _ in range(2)]
            o2 = torch.empty((3, n), dtype=tdt2, device=u.device)
            r = row(timed(torch, lambda: t2.evaluate_device(p2, out=o2, check=False), 10), n, bpe)
            r["kernel"] = t2.last_kernel()
            res[key] = r
            t2.domain_status()
            del p2, o2
    except Exception as e:  # diagnostic only: never fail the headline line
        res["error"] = repr(e)
    return res


def host_end_to_end(tables, uv, n):
    """What a NumPy caller gets: host arrays in, host array out (PCIe and page faults included).
    first_call: the process's first result of this size (its pages are first-touched); fresh_result_array: every
    later call that returns a NEW array - its memory is recycled from results the caller has dropped
    (bspy_amd/result_pool.py), as in a loop `x, y, z = s(u, v)`; reused: the caller passes `out=`."""
    ps = [np.ascontiguousarray(uv[0][:n]), np.ascontiguousarray(uv[1][:n])]
    t0 = time.perf_counter()
    out = tables.evaluate(ps)
    t1 = time.perf_counter()
    fresh = []
    for _ in range(3):
        del out                                  # the caller is done with the previous result
        ta = time.perf_counter()
        out = tables.evaluate(ps)
        fresh.append(time.perf_counter() - ta)
    t2 = time.perf_counter()
    tables.evaluate(ps, out=out)
    t3 = time.perf_counter()
    tables.evaluate(ps, out=out)
    t4 = time.perf_counter()
    return {"first_call_ms": round((t1 - t0) * 1e3, 2), "fresh_result_array_ms": round(min(fresh) * 1e3, 2),
            "reused_result_array_ms": round(min(t3 - t2, t4 - t3) * 1e3, 2),
            "note": "Spline.evaluate on NumPy arrays: H2D + kernel + D2H over PCIe, never `value`"}


def run_other_config(args, torch, dist, dev, world, rank):
    """BASELINE configs[2], [3], [4] (--config 3 / 4 / 5) with the same protocol as the headline: device-resident
    synthetic inputs, W warm-up steps, K timed steps between barrier + synchronize, MAX over ranks, one JSON line.
    Strong scaling: the config's whole job is sharded over the ranks (points contiguously, cfg4 by patch), no
    collective in the timed region; `with_allgather` (N > 1) adds the RCCL all-gather of the results."""
    import cases
    import bspy_amd
    from bspy_amd import _native as nv
    from bspy_amd.sharding import shard_bounds
    lib = nv.lib()
    stream_ptr = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    cfg = args.config
    info = {}

    if cfg in (3, 5):
        nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(cfg)
        tdt = torch.float64 if dt == np.float64 else torch.float32
        n_total = args.points if args.points != N_POINTS or cfg == 3 else 50_000_000
        start, stop = shard_bounds(n_total, world, rank)
        n_local = stop - start
        tables = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt, device=dev.index)
        g = torch.Generator(device=dev).manual_seed(1000 + rank)
        pts = [torch.rand(max(n_local, 1), dtype=tdt, device=dev, generator=g)[:n_local] for _ in range(nind)]
        ptrs = nv.ptr_array([p.data_ptr() for p in pts])
        if cfg == 3:
            outs = [torch.empty((ndep, n_local), dtype=tdt, device=dev), torch.empty((ndep * nind, n_local), dtype=tdt, device=dev)]
            wrt_arr = nv.int_array([1, 1])
            o0, o1 = ctypes.c_void_p(outs[0].data_ptr()), ctypes.c_void_p(outs[1].data_ptr())

            def step():
                if n_local == 0:
                    return
                st = lib.bsk_evaluate(tables._handle, wrt_arr, ptrs, n_local, nv.BSK_DEVICE, o0, stream_ptr, None)
                if st == 0:
                    st = lib.bsk_jacobian(tables._handle, ptrs, n_local, nv.BSK_DEVICE, o1, stream_ptr, None)
                if st != 0:
                    nv.check(st)
            bpe = 104
            workload = ("BASELINE configs[2]: bicubic surface order 4x4, nCoef 64x64, nDep 3, fp64: Spline.derivative([1,1]) + fused "
                        f"jacobian (two C-ABI calls per step) on {n_total} uniform-random (u,v)")
            metric = "M evals/sec (derivative + jacobian pairs) on 10M-point bicubic (p=3x3) surface"
        else:
            outs = [torch.empty((ndep, n_local), dtype=tdt, device=dev)]
            o0 = ctypes.c_void_p(outs[0].data_ptr())

            def step():
                if n_local == 0:
                    return
                st = lib.bsk_evaluate(tables._handle, None, ptrs, n_local, nv.BSK_DEVICE, o0, stream_ptr, None)
                if st != 0:
                    nv.check(st)
            bpe = 28
            workload = ("BASELINE configs[4]: trivariate manifold nInd 3, order 5, nCoef 40^3, nDep 4, fp32, clamped uniform knots, "
                        f"Spline.evaluate on {n_total} uniform-random (u,v,w)")
            metric = "M evals/sec on 50M-point trivariate (order 5, 40^3 x 4, fp32) manifold"
        units_local, units_total = n_local, n_total
        unit_name = "points"
        gather_rows = [o.shape[0] for o in outs]
        gather_chunk = -(-n_total // world)

        def gather_blocks():
            return [(o, o.shape[0], gather_chunk, n_local) for o in outs]
        status = tables
    elif cfg == 4:
        tbl = np.load(os.path.join(ROOT, "tests", "golden", "reference_tables.npz"))
        allp = cases.teapot_patches(tbl)
        start, stop = shard_bounds(len(allp), world, rank)
        patches = [bspy_amd.DeviceSpline(o, c, k, cf, np.float32, device=dev.index) for (o, c, k, cf) in allp[start:stop]]
        side = 2048
        gg = torch.linspace(0, 1, side, dtype=torch.float32, device=dev)
        npl = len(patches)
        pos = torch.empty((max(npl, 1), 3, side, side), dtype=torch.float32, device=dev)[:npl]
        tdt = torch.float32

        def step():
            if npl:
                bspy_amd.tessellate_tables(patches, (gg, gg), normals=False, out=(pos, None), check=False)
        bpe = 12
        units_local, units_total = npl * side * side, len(allp) * side * side
        unit_name = "grid points"
        workload = (f"BASELINE configs[3]: Utah teapot, {len(allp)} bicubic fp32 Bezier patches on a dense {side} x {side} grid each "
                    "(positions), one bsk_tessellate call per step, sharded by patch")
        metric = "M evals/sec on the 32-patch teapot 2048x2048 tessellation"
        per_rank = -(-len(allp) // world)

        def gather_blocks():
            return [(pos.reshape(npl, 3 * side * side), per_rank, 3 * side * side, None)]
        status = patches[0] if patches else None
        tables = patches[0] if patches else None
    else:
        raise SystemExit(f"unknown --config {cfg}")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(f):
        for _ in range(args.warmup):
            f()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            f()
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        elapsed, kernel_ms = t1 - t0, e0.elapsed_time(e1) / args.steps
        if dist is not None:
            tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed, kernel_ms = float(tt[0]), float(tt[1])
        return elapsed, kernel_ms

    barrier()
    t_end = time.perf_counter() + 0.15                   # clock spin-up on the measured step itself
    while time.perf_counter() < t_end:
        step()
    torch.cuda.synchronize()
    elapsed, kernel_ms = measure(step)
    if status is not None:
        status.domain_status()
    kernel = tables.last_kernel() if tables is not None else ""

    extra = {}
    if cfg == 5 and units_local > 0:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from cfg5_stages import stage_times
        st_ms = stage_times(tables, step, reps=max(3, min(args.steps, 10)))
        sort_ms = sum(v for k, v in st_ms.items() if not k.startswith("eval"))
        eval_ms = sum(v for k, v in st_ms.items() if k.startswith("eval"))
        sec = kernel_ms * 1e-3
        extra["pipeline"] = {
            "stages_ms": {k: round(v, 4) for k, v in st_ms.items()}, "sort_ms": round(sort_ms, 4), "eval_ms": round(eval_ms, 4),
            "alu_frac": round(CFG5_FLOP_PER_POINT * units_local / sec / 1e12 / FP32_PEAK_TFLOPS, 4),
            "alu_frac_of_eval_kernel": round(CFG5_FLOP_PER_POINT * units_local / (eval_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4) if eval_ms else None,
            "mfma_frac_of_eval_kernel": round(CFG5_MFMA_FLOP_PER_POINT * units_local / (eval_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4) if eval_ms else None,
            "note": ("HIP events between the pipeline's kernels (bsk_debug_stage_times); alu_frac = 1450 algorithmic fp32 flop per "
                     "point against the 157.3 TFLOP/s vector peak (SURVEY 8d: the bound that applies - the table gather is served "
                     "from LDS); mfma_frac = the 1000 flop per point the v_mfma_f32_4x4x1 instructions carry against the fp32 "
                     "matrix peak (also 157.3), over the evaluation kernel's own time"),
        }
    if dist is not None:
        blocks = gather_blocks()
        bufs = []
        for (t, rows, width, valid) in blocks:
            send = torch.zeros((rows, width), dtype=tdt, device=dev)
            full = torch.empty((world * rows, width), dtype=tdt, device=dev)
            bufs.append((t, send, full, valid))

        def gstep():
            step()
            for (t, send, full, valid) in bufs:
                if valid is None:
                    send[:t.shape[0]].copy_(t)           # cfg4: whole patches, the tail rank may hold fewer
                else:
                    send[:, :valid].copy_(t)             # point shards: the tail rank's may be short
                dist.all_gather_into_tensor(full, send)
        gel, _ = measure(gstep)
        gsec = gel / args.steps
        nbytes = sum(send.numel() * send.element_size() for (_, send, _, _) in bufs)
        extra["with_allgather"] = {"value": round(units_total / gsec / 1e6, 1), "unit": "M evals/s", "ms_per_step": round(gsec * 1e3, 4),
                                   "allgather_bytes_per_rank_in": (world - 1) * nbytes,
                                   "xgmi_in_GBs_per_gpu": round((world - 1) * nbytes / gsec / 1e9, 1)}
        del bufs

    if rank == 0:
        sec_step = elapsed / args.steps
        achieved = bpe * units_local / (kernel_ms * 1e-3) / 1e9
        res = {
            "metric": metric, "value": round(units_total / sec_step / 1e6, 1), "unit": "M evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec_step * 1e3, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64" if tdt == torch.float64 else "f32",
            "data": "synthetic",
            "config": {"workload": workload, "baseline_config": cfg, f"{unit_name.replace(' ', '_')}_total": units_total,
                       f"{unit_name.replace(' ', '_')}_per_gpu": units_local,
                       "sharding": ("patches" if cfg == 4 else "contiguous shards of ceil(N / ranks) points") + ", no collective in the timed region"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": kernel, "kernel_ms": round(kernel_ms, 5),
                         "algorithmic_bytes_per_eval": bpe, "algorithmic_bytes_per_launch": bpe * units_local},
        }
        res.update(extra)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--op", choices=["evaluate", "derivative", "jacobian", "normal"], default="evaluate")
    ap.add_argument("--config", type=int, choices=[2, 3, 4, 5], default=2,
                    help="BASELINE config of the job: 2 = the metric's (default), 3 = derivative + jacobian, 4 = teapot tessellation, 5 = trivariate fp32")
    ap.add_argument("--points", type=int, default=N_POINTS, help="points of the whole job (sharded over the ranks; config 5 defaults to 50 M)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline and host end-to-end legs")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configs (reported beside the headline, N=1 only)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000)
    ap.add_argument("--spinup", type=int, default=300, help="untimed launches before warm-up (clock ramp)")
    args = ap.parse_args()

    import torch
    import cases
    import bspy_amd
    from bspy_amd import _native as nv
    from bspy_amd.sharding import shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    dist = None
    torch.cuda.set_device(local_rank)
    bspy_amd.set_device(local_rank)
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("BSPY_AMD_BENCH_DIST") == "1":
        # launched by torch.distributed.run: one process per GPU, RCCL ("nccl" on ROCm) over xGMI
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    dev = torch.device("cuda", local_rank)
    if args.config != 2:
        return run_other_config(args, torch, dist, dev, world, rank)
    nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
    n_total = args.points
    lib = nv.lib()
    tables = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt, device=local_rank)
    handle = tables._handle
    rows = ndep * nind if args.op == "jacobian" else ndep
    wrt = [1, 1] if args.op == "derivative" else None
    wrt_arr = nv.int_array(wrt) if wrt is not None else None
    stream_ptr = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def make_workload(n_local, seed):
        """n_local points of this rank, device resident, and the one-foreign-call step on them."""
        rng = np.random.default_rng(seed)
        uv = rng.random((2, max(n_local, 1)))[:, :n_local]
        u = torch.as_tensor(np.ascontiguousarray(uv[0]), device=dev)
        v = torch.as_tensor(np.ascontiguousarray(uv[1]), device=dev)
        out = torch.empty((rows, n_local), dtype=torch.float64, device=dev)
        ptrs = nv.ptr_array([u.data_ptr(), v.data_ptr()])
        out_ptr = ctypes.c_void_p(out.data_ptr())

        # One step = one C-ABI call on the current stream.  The ctypes arguments are built once so the host
        # side of a step is a single foreign call (the Python wrappers' tensor checks cost more than
        # the kernel's launch gap and would make the loop host-bound).
        def step():
            if n_local == 0:
                return
            if args.op == "jacobian":
                st = lib.bsk_jacobian(handle, ptrs, n_local, nv.BSK_DEVICE, out_ptr, stream_ptr, None)
            elif args.op == "normal":
                st = lib.bsk_normal(handle, ptrs, n_local, nv.BSK_DEVICE, 1, 0, out_ptr, stream_ptr, None)
            else:
                st = lib.bsk_evaluate(handle, wrt_arr, ptrs, n_local, nv.BSK_DEVICE, out_ptr, stream_ptr, None)
            if st != 0:
                nv.check(st)
        return uv, u, v, out, ptrs, step

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def spin(ptrs, n_local):
        # Clock spin-up (untimed, before the W warm-up steps): an idle MI355X needs tens of milliseconds of load
        # before its shader clock settles; the first ~100 launches of a fresh process run ~20 % slower
        # (tools/launch_gaps.py).  It runs a DIFFERENT kernel of the same family on the same batch (the fused
        # jacobian, or the evaluation when the jacobian is measured): same load, but the measured kernel's
        # rocprofv3 --kernel-trace statistics then hold steady-state launches only and agree with the HIP events.
        if n_local == 0:
            return
        spin_out = torch.empty((ndep * nind, n_local), dtype=torch.float64, device=dev)
        sp = ctypes.c_void_p(spin_out.data_ptr())
        for _ in range(args.spinup):
            if args.op != "jacobian":
                st = lib.bsk_jacobian(handle, ptrs, n_local, nv.BSK_DEVICE, sp, stream_ptr, None)
            else:
                st = lib.bsk_evaluate(handle, None, ptrs, n_local, nv.BSK_DEVICE, sp, stream_ptr, None)
            if st != 0:
                nv.check(st)
        torch.cuda.synchronize()

    def measure(step, per_launch):
        """W warm-up steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks.
        Returns (wall seconds for K steps, mean launch ms from HIP events, median launch ms or None).
        The timed region holds the K launches and ONE event at either end, on the stream the kernel is launched on.
        (Round 2 recorded an event after every launch inside it, for the median: each record costs the device ~3 us
        between two launches - 0.0899-0.0911 ms per step against 0.0871-0.0875 without, same box, same build.  The
        median now comes from a second, untimed pass.)"""
        for _ in range(args.warmup):
            step()
        barrier()
        k = args.steps
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        t0 = time.perf_counter()
        evs[0].record()
        for _ in range(k):
            step()
        evs[1].record()
        torch.cuda.synchronize()            # this rank's K steps are done: its clock stops here ...
        t1 = time.perf_counter()
        barrier()                           # ... every rank's are; the MAX over ranks below is the job's time (the
        t2 = time.perf_counter()            # closing barrier's own ~50 us of RCCL latency is not part of the K steps;
        elapsed = t1 - t0                   # the figure that includes it is reported beside it for N > 1)
        measure.incl_barrier = t2 - t0
        kernel_ms = evs[0].elapsed_time(evs[1]) / k
        median_ms = None
        if per_launch:                      # untimed diagnostic pass: an event after every launch
            pe = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
            pe[0].record()
            for i in range(k):
                step()
                pe[i + 1].record()
            torch.cuda.synchronize()
            median_ms = statistics.median(pe[i].elapsed_time(pe[i + 1]) for i in range(k))
        if dist is not None:
            tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed, kernel_ms = float(tt[0]), float(tt[1])
        return elapsed, kernel_ms, median_ms

    # ---- headline: the 10 M points of the metric, sharded over the ranks (N = 1: all of them)
    start, stop = shard_bounds(n_total, world, rank)
    n_local = stop - start
    uv, u, v, out, ptrs, step = make_workload(n_local, 1000 + rank)
    barrier()                       # first use of the communicator (lazy RCCL init) happens here, untimed
    spin(ptrs, n_local)
    elapsed, kernel_ms, median_ms = measure(step, per_launch=(world == 1 and n_local >= 5_000_000))
    tables.domain_status()                                # the in-kernel domain check found nothing
    kernel = tables.last_kernel()
    incl_barrier = measure.incl_barrier

    extra = {}
    if dist is not None:
        # results replicated on every rank: one all-gather of every rank's (rows, chunk) block per step
        chunk = -(-n_total // world)
        full = torch.empty((world * rows, chunk), dtype=torch.float64, device=dev)
        send = torch.zeros((rows, chunk), dtype=torch.float64, device=dev)

        def gstep():
            step()
            send[:, :n_local].copy_(out)
            dist.all_gather_into_tensor(full, send)
        gel, _, _ = measure(gstep, per_launch=False)
        gsec = gel / args.steps
        extra["with_allgather"] = {"value": round(n_total / gsec / 1e6, 1), "unit": "M evals/s", "ms_per_step": round(gsec * 1e3, 4),
                                   "allgather_bytes_per_rank_in": (world - 1) * rows * chunk * 8,
                                   "xgmi_in_GBs_per_gpu": round((world - 1) * rows * chunk * 8 / gsec / 1e9, 1)}
        del full, send
        # weak scaling: 10 M points on every rank
        wuv, wu, wv, wout, wptrs, wstep = make_workload(n_total, 2000 + rank)
        wel, wk, _ = measure(wstep, per_launch=False)
        extra["weak"] = {"value": round(world * n_total / (wel / args.steps) / 1e6, 1), "unit": "M evals/s",
                         "points_per_gpu": n_total, "ms_per_step": round(wel / args.steps * 1e3, 5)}
        tables.domain_status()
        del wu, wv, wout

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total / (elapsed / args.steps) / 1e6
        bpe = BYTES_PER_EVAL[args.op]
        achieved = bpe * n_local / (kernel_ms * 1e-3) / 1e9     # this rank's kernel: its shard's bytes over its launch time
        traffic, traffic_src = pmc_traffic(kernel) if (n_local == N_POINTS and args.op in ("evaluate", "jacobian")) else (None, None)
        res = {
            "metric": "M evals/sec on 10M-point bicubic (p=3x3) surface", "value": round(value, 1), "unit": "M evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: bicubic surface order 4x4, nCoef 64x64, nDep 3, fp64, clamped uniform "
                                   f"knots, Spline.{args.op} on {n_total} uniform-random (u,v)",
                       "points_total": n_total, "points_per_gpu": n_local, "op": args.op,
                       "sharding": "contiguous shards of ceil(N / ranks) points, no collective in the timed region"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel, "kernel_ms": round(kernel_ms, 5),
                         "kernel_ms_median": round(median_ms, 5) if median_ms is not None else None,
                         "algorithmic_bytes_per_eval": bpe, "algorithmic_bytes_per_launch": bpe * n_local},
        }
        if dist is not None:
            res["ms_per_step_incl_closing_barrier"] = round(incl_barrier / args.steps * 1e3, 5)
        res.update(extra)
        if world == 1:
            res["roofline"]["measured_stream_floor_GBs"] = stream_floor(torch, tables, u, v, n_local)
            if not args.no_extra and args.op == "evaluate":
                res["other_configs"] = other_configs(torch, tables, u, v, n_local)
            if not args.no_cpu:
                res["end_to_end_host"] = host_end_to_end(tables, uv, n_local)
                res["cpu_baseline"] = cpu_baseline(order, ncoef, knots, coefs, args.cpu_sample)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
