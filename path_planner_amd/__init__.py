"""path_planner_amd — MI355X (gfx950) implementation of the ASV planner's sampling /
Dubins-edge / edge-cost hot path behind a C ABI (include/ppgpu.h).

Python here is glue only: a ctypes binding of libppgpu.so for tests and bench.py, and
deterministic synthetic workloads (SURVEY.md section 8 d).  The product is
path_planner_amd/csrc (HIP kernels + C ABI) and path_planner_amd/host (C++ mirror of the
reference's Planner interface).  There is no CPU fallback: without the HIP library,
importing `path_planner_amd.api` raises.
"""
from .types import CONFIG_DEFAULTS, VERTEX_DTYPE, RESULT_DTYPE, PpgpuConfig, make_config  # noqa: F401
