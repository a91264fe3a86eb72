"""The multi-GPU path on a GPU (one rank): ShardedEvaluator over RCCL ("nccl") with its collectives
forced on in a one-rank group - communicator initialisation, the MIN all-reduce of the first
offender, the all-gather into the final (rows, N) layout - for device tensors and for NumPy inputs;
results equal the unsharded calls bitwise.  (The world-size-2 logic runs under gloo in
tests/test_host_logic.py; the driver measures N > 1 on a multi-GPU node.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import cases, bspy_amd
from bspy_amd.sharding import ShardedEvaluator
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
s = bspy_amd.Spline(nind, ndep, order, ncoef, knots, coefs)
rng = np.random.default_rng(3)
n = 100_003
uv = rng.random((2, n))
sh = ShardedEvaluator(s, collectives_at_world1=True)
assert sh.coll and dist.get_backend() == "nccl"
tables = bspy_amd._spline_evaluation.device_tables(s)
ref = tables.evaluate([uv[0], uv[1]])
refj = tables.jacobian([uv[0], uv[1]]).reshape(-1, n)
refd = tables.evaluate([uv[0], uv[1]], [1, 1])
# NumPy inputs: staged on the GPU for RCCL, returned as NumPy
got = sh.evaluate([uv[0], uv[1]])
assert isinstance(got, np.ndarray) and got.shape == (3, n) and np.array_equal(got, ref)
assert np.array_equal(sh.jacobian([uv[0], uv[1]]), refj)
assert np.array_equal(sh.derivative([1, 1], [uv[0], uv[1]]), refd)
# device tensors: gathered on the device
tu, tv = torch.as_tensor(uv[0], device="cuda"), torch.as_tensor(uv[1], device="cuda")
gt = sh.evaluate([tu, tv])
assert gt.is_cuda and gt.shape == (3, n) and np.array_equal(gt.cpu().numpy(), ref)
gj = sh.jacobian([tu, tv], sharded_input=True, total=n)
assert np.array_equal(gj.cpu().numpy(), refj)
assert np.array_equal(sh.evaluate([tu, tv], gather=False, check=False).cpu().numpy(), ref)
# first offender agreed through the all-reduce
bad = uv.copy(); bad[1, 777] = 1.5; bad[0, 90000] = -0.1
for pts in ([bad[0], bad[1]], [torch.as_tensor(bad[0], device="cuda"), torch.as_tensor(bad[1], device="cuda")]):
    try:
        sh.evaluate(pts)
        raise SystemExit("no error raised")
    except ValueError as e:
        assert "flat index 777" in str(e), str(e)
dist.destroy_process_group()
print("sharded-nccl-ok")
'''


def test_sharded_evaluator_rccl_one_rank(tmp_path):
    f = tmp_path / "worker.py"
    f.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(f), ROOT, "29653"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sharded-nccl-ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_multi_device_c_abi_one_device():
    """bsk_multi_* (single process, shard plan + optional RCCL all-gather) with the devices this box has:
    host path, sharded device path, gathered device path, jacobian, first offender - all against the
    single-device calls, bitwise."""
    import numpy as np
    import torch
    import cases
    import bspy_amd
    from bspy_amd import DeviceSpline, MultiDeviceSpline
    nind, ndep, order, ncoef, knots, coefs, dt = cases.bench_spline(2)
    ndev = min(bspy_amd._native.device_count(), 2)
    devs = list(range(ndev))
    m = MultiDeviceSpline(order, ncoef, knots, coefs, devices=devs)
    one = DeviceSpline(order, ncoef, knots, coefs, device=0)
    rng = np.random.default_rng(11)
    n = 250_007
    uv = rng.random((2, n))
    plan = m.shard_plan(n)
    assert plan[0] == 0 and plan[-1] == n and all(b >= a for a, b in zip(plan, plan[1:]))
    ref = one.evaluate([uv[0], uv[1]])
    refd = one.evaluate([uv[0], uv[1]], [0, 2])
    refj = one.jacobian([uv[0], uv[1]])
    assert np.array_equal(m.evaluate([uv[0], uv[1]]), ref)
    assert np.array_equal(m.evaluate([uv[0], uv[1]], [0, 2]), refd)
    assert np.array_equal(m.jacobian([uv[0], uv[1]]), refj)
    shards = [[torch.as_tensor(uv[i][plan[d]:plan[d + 1]].copy(), device=f"cuda:{devs[d]}") for i in range(2)] for d in range(ndev)]
    outs = m.evaluate_device(shards, n)                               # sharded results, no collective
    assert np.array_equal(np.concatenate([o.cpu().numpy() for o in outs], axis=1), ref)
    full = m.evaluate_device(shards, n, gather=True)                  # one grouped RCCL all-gather
    for o in full:
        assert o.shape[0] == 3 and np.array_equal(o.cpu().numpy()[:, :n], ref)
    fullj = m.evaluate_device(shards, n, gather=True, jacobian=True)
    for o in fullj:
        assert np.array_equal(o.cpu().numpy()[:, :n], refj.reshape(6, n))
    bad = uv.copy()
    bad[0, n - 5] = 2.0
    with pytest.raises(bspy_amd.DomainError) as e:
        m.evaluate([bad[0], bad[1]])
    assert e.value.index == n - 5
    assert np.array_equal(m.evaluate([uv[0], uv[1]]), ref)            # the record was reset
    m.close()
