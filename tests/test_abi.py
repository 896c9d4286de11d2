"""The C-ABI library loads and exports every symbol include/ppgpu.h declares (no compute calls: this runs
without a GPU); the product refuses to pretend when there is no device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|int64_t|uint64_t|double|char\s*\*|const char\s*\*)\s+\*?\s*([a-z_0-9]+)\s*\(", txt, flags=re.M)
    return [n for n in names if n.startswith(("ppgpu_", "dubins_")) and n != "ppgpu_edge_pack"]


def test_libppgpu_exports_everything_the_header_declares():
    from path_planner_amd import api     # raises if the HIP library was not built
    declared = _declared("ppgpu.h")
    assert len(declared) >= 24, declared
    lib = C.CDLL(api.LIB_PATH)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(set(api.EXPORTS)) == sorted(set(declared))


def test_libdubins_exports_the_four_functions():
    so = os.path.join(ROOT, "path_planner_amd", "csrc", "libdubins.so")
    assert os.path.exists(so), "run __graft_entry__.build()"
    lib = C.CDLL(so)
    for n in _declared("dubins.h"):
        assert hasattr(lib, n), n
    assert sorted(_declared("dubins.h")) == ["dubins_extract_subpath", "dubins_path_length", "dubins_path_sample", "dubins_shortest_path"]


def test_struct_layouts_match_the_header():
    from path_planner_amd.types import PpgpuConfig, RESULT_DTYPE, VERTEX_DTYPE
    assert C.sizeof(PpgpuConfig) == 12 * 8 + 4 * 4
    assert VERTEX_DTYPE.itemsize == 64 and RESULT_DTYPE.itemsize == 128
    assert RESULT_DTYPE.fields["true_cost"][1] == 8 and RESULT_DTYPE.fields["param"][1] == 104


def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from path_planner_amd import api
    with pytest.raises(api.PpgpuError):
        api.Context(0)


def test_wire_formats_roundtrip():
    """path_planner_amd/host Messages.h (the five path_planner_common messages without ROS): plan -> Plan message -> ROS 1
    bytes -> plan, byte layouts, truncated input.  Host-only C++ (no GPU call)."""
    import subprocess
    host = os.path.join(ROOT, "path_planner_amd", "host")
    exe = os.path.join(host, "msg_roundtrip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", host, "msg_roundtrip"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
