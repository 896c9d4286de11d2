import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The CPU checker (oracle/libpp_oracle.so) is built on demand; it is test infrastructure only."""
    so = os.path.join(ROOT, "oracle", "libpp_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("pp_oracle.cpp", "pp_oracle_c.cpp", "pp_oracle.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    yield


@pytest.fixture(autouse=True)
def _prepasses_on_small_launches(monkeypatch):
    """The two optional prepasses of a costing launch (pp_k_plan_skips, pp_k_approach_events) normally run only on launches of
    8 192 edges or more; the parity tests mostly cost a few thousand edges, so they lower the threshold to 0 and every record they
    compare with the oracle went through both.  tests/test_gpu_parity.py::test_small_launches_without_prepasses checks the other
    setting gives the same bytes.  (Read by ppgpu_create, also in the plan_cli subprocesses.)"""
    monkeypatch.setenv("PPGPU_PREPASS_MIN_EDGES", "0")
    yield


@pytest.fixture(params=["prepasses_always", "production_route"])
def prepass_route(request, monkeypatch, _prepasses_on_small_launches):
    """The plan-level oracle tests run twice: with the prepasses forced onto every launch (above), and with the production
    setting — PPGPU_PREPASS_MIN_EDGES unset, so the 40-edge launches of the reference-shaped expand() path
    (SamplingBasedPlanner.cpp:52-151) skip pp_k_plan_skips / pp_k_approach_events exactly as they do outside the tests."""
    if request.param == "production_route":
        monkeypatch.delenv("PPGPU_PREPASS_MIN_EDGES", raising=False)
    return request.param
