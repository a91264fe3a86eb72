import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_parity():
    return np.load(os.path.join(GOLDEN, "parity.npz"))


@pytest.fixture(scope="session")
def golden_tables():
    return np.load(os.path.join(GOLDEN, "reference_tables.npz"))


@pytest.fixture(scope="session")
def golden_basis():
    return np.load(os.path.join(GOLDEN, "basis.npz"))


@pytest.fixture(scope="session")
def golden_api():
    import json
    with open(os.path.join(GOLDEN, "api_semantics.json")) as f:
        return json.load(f)


# ---------------------------------------------------------------------------------------------
# Observed errors.  Parity tests assert against the contract's bar (fp64: 1e-10 of the result scale, BASELINE.json;
# asserted tighter where the arithmetic allows) AND record what they observed, so that a bar is never looser than
# the evidence: the table is printed at the end of the session and written to gpurun_out/observed_errors.json.
# ---------------------------------------------------------------------------------------------
OBSERVED = {}


def observe(label, err, bar):
    """Record the observed error under `label` (worst over the session), then assert it is within `bar`."""
    err = float(err)
    cur = OBSERVED.get(label)
    if cur is None or not (err <= cur[0]):
        OBSERVED[label] = (err, float(bar))
    assert err <= bar, f"{label}: observed {err:.3e} > bar {bar:.1e}"


def pytest_sessionfinish(session, exitstatus):
    if not OBSERVED:
        return
    import json
    lines = [f"  {k:<58s} observed {v[0]:9.2e}   bar {v[1]:7.1e}" for k, v in sorted(OBSERVED.items())]
    print("\nobserved errors (worst per label, relative to the result scale unless the label says otherwise):\n" + "\n".join(lines))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "observed_errors.json"), "w") as f:
            json.dump({k: {"observed": v[0], "bar": v[1]} for k, v in sorted(OBSERVED.items())}, f, indent=1)
