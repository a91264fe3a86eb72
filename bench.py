"""
bench.py - BASELINE metric: M evals/s on the 10 M-point bicubic (order 4x4, nCoef 64x64,
nDep 3, fp64) surface (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (Spline.evaluate: span search + Cox-de Boor recursion
+ tensor-product contraction, fused in one HIP kernel) over one batch of 10 M synthetic
(u, v) points per GPU, inputs and outputs resident in HBM.  Multi-GPU: the point batch is
sharded, one process per GPU, no data-path collective in the timed region (weak scaling:
10 M points per rank); the all-gather-inclusive figure is measured separately and reported
under "with_allgather".

Prints ONE JSON line (rank 0).  roofline.achieved uses the ALGORITHMIC bytes of
SURVEY.md 8(d): 40 B per eval (2 x 8 B in + 3 x 8 B out) against 8 TB/s HBM3E.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
N_POINTS = 10_000_000
BYTES_PER_EVAL = {"evaluate": 40, "derivative": 40, "jacobian": 64, "normal": 40}


def cpu_baseline(order, ncoef, knots, coefs, sample):
    """The C oracle (a port of the reference's algorithm, 1 core) on a bounded sample of
    the same workload, plus the reference-structured Python loop on a smaller one."""
    import oracle
    rng = np.random.default_rng(12345)
    uv = rng.random((2, sample))
    oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0], [uv[0][:1000], uv[1][:1000]])   # load + warm
    t0 = time.perf_counter()
    _, bad = oracle.c_evaluate(order, ncoef, knots, coefs, [0, 0], [uv[0], uv[1]])
    t1 = time.perf_counter()
    assert bad == -1
    npy = 100_000
    t2 = time.perf_counter()
    oracle.py_batch(order, ncoef, knots, coefs, [0, 0], [uv[0][:npy], uv[1][:npy]])
    t3 = time.perf_counter()
    return {
        "value": round(sample / (t1 - t0) / 1e6, 4), "unit": "M evals/s", "cores": 1, "kind": "port",
        "sample": f"{sample} of the 10M random (u,v) points, C oracle (oracle/bspline_oracle.c), one thread",
        "python_reference_structured": {"value": round(npy / (t3 - t2) / 1e6, 5), "unit": "M evals/s", "cores": 1,
                                        "sample": f"{npy} points, per-point Python/NumPy loop (oracle.py_batch)"},
        "host_cpus": os.cpu_count(),
    }


def pmc_traffic(op, n):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/<round>/pmc_per_launch.json: FETCH_SIZE and WRITE_SIZE in KB, collected in
    separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 wide
    coalesced reads).  None when no matching profile is committed."""
    if op not in ("evaluate", "jacobian") or n != N_POINTS:
        return None
    path = os.path.join(ROOT, "profiles", "r01_final_eval_rowrot", "pmc_per_launch.json")
    try:
        with open(path) as f:
            prof = json.load(f)
        want = "jac_rowrot<double, 4, false" if op == "jacobian" else "eval_rowrot<double, 4, false"
        k = next(v for name, v in prof.items() if want in name)
        return int((2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024)
    except (OSError, StopIteration, KeyError, ValueError):
        return None


def other_configs(torch, tables, u, v, n):
    """The other BASELINE configs on this GPU, timed after the headline (diagnostic; device-resident
    I/O, steady state): cfg3 derivative / fused jacobian on the same batch, cfg4 = the 32 teapot
    patches on a 2048 x 2048 grid from one bsk_tessellate call, cfg5 = trivariate order 5, 40^3 x 4
    fp32 on 10 M points (cell-order pipeline)."""
    import cases
    import bspy_amd

    def timed(f, steps, spin_s=0.06):
        t_end = time.perf_counter() + spin_s      # clock spin-up (the CPU legs before this leave the GPU idle)
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

    res = {}
    try:
        out = torch.empty((3, n), dtype=torch.float64, device=u.device)
        jout = torch.empty((3, 2, n), dtype=torch.float64, device=u.device)
        res["cfg3_derivative_1_1"] = row(timed(lambda: tables.evaluate_device([u, v], [1, 1], out=out, check=False), 30), n, 40)
        res["cfg3_jacobian_fused"] = row(timed(lambda: tables.jacobian_device([u, v], out=jout, check=False), 30), n, 64)
        tables.domain_status()
        del out, jout
        g = np.load(os.path.join(ROOT, "tests", "golden", "reference_tables.npz"))
        patches = [bspy_amd.DeviceSpline(o, c, k, cf, np.float32) for (o, c, k, cf) in cases.teapot_patches(g)]
        gg = torch.linspace(0, 1, 2048, dtype=torch.float32, device=u.device)
        pos = torch.empty((32, 3, 2048, 2048), dtype=torch.float32, device=u.device)
        sec = timed(lambda: bspy_amd.tessellate_tables(patches, (gg, gg), normals=False, out=(pos, None), check=False), 10)
        res["cfg4_teapot_32_patches_grid2048"] = row(sec, 32 * 2048 * 2048, 12)
        patches[0].domain_status()
        del pos
        nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(5)
        t5 = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt)
        p5 = [torch.rand(n, dtype=torch.float32, device=u.device) for _ in range(3)]
        o5 = torch.empty((4, n), dtype=torch.float32, device=u.device)
        res["cfg5_trivariate_f32"] = row(timed(lambda: t5.evaluate_device(p5, out=o5, check=False), 10), n, 28)
        t5.domain_status()
    except Exception as e:  # diagnostic only: never fail the headline line
        res["error"] = repr(e)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--op", choices=["evaluate", "derivative", "jacobian", "normal"], default="evaluate")
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configs (reported beside the headline, N=1 only)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000)
    ap.add_argument("--spinup", type=int, default=300, help="untimed launches before warm-up (clock ramp)")
    args = ap.parse_args()

    import torch
    import cases
    import bspy_amd

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

    nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
    n = args.points
    rng = np.random.default_rng(1000 + rank)            # every rank owns a different shard
    uv = rng.random((2, n))
    dev = torch.device("cuda", local_rank)
    u = torch.as_tensor(uv[0], device=dev)
    v = torch.as_tensor(uv[1], device=dev)
    tables = bspy_amd.DeviceSpline(order, ncoef, knots, coefs, dt, device=local_rank)
    rows = ndep * nind if args.op == "jacobian" else ndep
    out = torch.empty((ndep, nind, n) if args.op == "jacobian" else (ndep, n), dtype=torch.float64, device=dev)
    wrt = [1, 1] if args.op == "derivative" else None

    # One step = one C-ABI call on the current stream.  The ctypes arguments are built once so
    # the host side of a step is a single foreign call (the Python wrappers' tensor checks
    # cost more than the kernel's launch gap and would make the loop host-bound).
    import ctypes
    from bspy_amd import _native as nv
    lib = nv.lib()
    uvw_ptrs = nv.ptr_array([u.data_ptr(), v.data_ptr()])
    wrt_arr = nv.int_array(wrt) if wrt is not None else None
    stream_ptr = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out_ptr = ctypes.c_void_p(out.data_ptr())
    handle = tables._handle

    def step():
        if args.op == "jacobian":
            st = lib.bsk_jacobian(handle, uvw_ptrs, n, nv.BSK_DEVICE, out_ptr, stream_ptr, None)
        elif args.op == "normal":
            st = lib.bsk_normal(handle, uvw_ptrs, n, nv.BSK_DEVICE, 1, 0, out_ptr, stream_ptr, None)
        else:
            st = lib.bsk_evaluate(handle, wrt_arr, uvw_ptrs, n, nv.BSK_DEVICE, out_ptr, stream_ptr, None)
        if st != 0:
            nv.check(st)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Clock spin-up (untimed, before the W warm-up steps): an idle MI355X needs tens of
    # milliseconds of load before its shader clock settles; the first ~100 launches of a fresh
    # process run ~20 % slower than the steady state (tools/launch_gaps.py).
    barrier()                       # first use of the communicator (lazy RCCL init) happens here, untimed
    # The spin-up runs a DIFFERENT kernel of the same family on the same batch (the fused jacobian, or
    # the evaluation when the jacobian is the one measured): same load on the GPU, but the measured
    # kernel's rocprofv3 --kernel-trace statistics then hold steady-state launches only and agree
    # with the live HIP-event average below.
    spin_out = torch.empty((ndep, nind, n) if args.op != "jacobian" else (ndep, n), dtype=torch.float64, device=dev)
    spin_ptr = ctypes.c_void_p(spin_out.data_ptr())
    for _ in range(args.spinup):
        if args.op != "jacobian":
            st = lib.bsk_jacobian(handle, uvw_ptrs, n, nv.BSK_DEVICE, spin_ptr, stream_ptr, None)
        else:
            st = lib.bsk_evaluate(handle, None, uvw_ptrs, n, nv.BSK_DEVICE, spin_ptr, stream_ptr, None)
        if st != 0:
            nv.check(st)
    torch.cuda.synchronize()
    del spin_out
    for _ in range(args.warmup):
        step()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    t1 = time.perf_counter()
    tables.domain_status()                                # the in-kernel domain check found nothing
    elapsed = t1 - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps        # average launch duration, HIP events on the launch stream
    if dist is not None:
        tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(tt[0]), float(tt[1])

    # gather-inclusive variant (results replicated on every rank), outside the headline timing
    gathered = None
    if dist is not None:
        full = torch.empty((world, rows, n), dtype=torch.float64, device=dev)
        for _ in range(2):
            step()
            dist.all_gather_into_tensor(full, out.view(rows, n))
        barrier()
        g0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            dist.all_gather_into_tensor(full, out.view(rows, n))
        barrier()
        g1 = time.perf_counter()
        tt = torch.tensor([g1 - g0], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        gsec = float(tt[0]) / args.steps
        gathered = {"value": round(world * n / gsec / 1e6, 1), "unit": "M evals/s", "ms_per_step": round(gsec * 1e3, 4),
                    "allgather_bytes_per_rank_in": (world - 1) * rows * n * 8,
                    "xgmi_in_GBs_per_gpu": round((world - 1) * rows * n * 8 / gsec / 1e9, 1)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n / (elapsed / args.steps) / 1e6
        bpe = BYTES_PER_EVAL[args.op]
        achieved = bpe * n / (kernel_ms * 1e-3) / 1e9
        res = {
            "metric": "M evals/sec on 10M-point bicubic (p=3x3) surface",
            "value": round(value, 1), "unit": "M evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: bicubic surface order 4x4, nCoef 64x64, nDep 3, fp64, "
                                   f"Spline.{args.op} on {n} uniform-random (u,v) per GPU",
                       "points_per_gpu": n, "op": args.op, "sharding": "point batch sharded per rank, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args.op, n),
                         "kernel": {"jacobian": "jac_rowrot<double,4,false>", "normal": "jac_rowrot<double,4,true>"}.get(args.op, "eval_rowrot<double,4>"),
                         "kernel_ms": round(kernel_ms, 5), "algorithmic_bytes_per_eval": bpe,
                         "algorithmic_bytes_per_launch": bpe * n,
                         "measured_stream_floor_GBs": 5960.0},
        }
        if gathered is not None:
            res["with_allgather"] = gathered
        if world == 1 and not args.no_extra and args.op == "evaluate":
            res["other_configs"] = other_configs(torch, tables, u, v, n)
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(order, ncoef, knots, coefs, args.cpu_sample)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
