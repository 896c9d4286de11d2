"""ctypes / numpy mirrors of the structs in include/ppgpu.h."""
import ctypes as C

import numpy as np


class PpgpuConfig(C.Structure):
    """struct ppgpu_config (include/ppgpu.h)."""
    _fields_ = [
        ("max_speed", C.c_double), ("slow_speed", C.c_double), ("turning_radius", C.c_double),
        ("coverage_turning_radius", C.c_double), ("time_horizon", C.c_double), ("time_minimum", C.c_double),
        ("collision_checking_increment", C.c_double), ("start_state_time", C.c_double),
        ("ribbon_width", C.c_double), ("collision_penalty_factor", C.c_double),
        ("time_penalty_factor", C.c_double), ("heuristic_turning_radius", C.c_double),
        ("heuristic", C.c_int32), ("tsp_k", C.c_int32), ("branching_factor", C.c_int32), ("reserved", C.c_int32),
    ]


# Reference defaults: PlannerConfig.h:179-189, Ribbon.cpp:4, Edge.h:151-152, executive.cpp:391
CONFIG_DEFAULTS = dict(
    max_speed=2.5, slow_speed=0.5, turning_radius=8.0, coverage_turning_radius=16.0, time_horizon=30.0,
    time_minimum=5.0, collision_checking_increment=0.05, start_state_time=0.0, ribbon_width=1.5,
    collision_penalty_factor=600.0, time_penalty_factor=1.0, heuristic_turning_radius=8.0,
    heuristic=2, tsp_k=2, branching_factor=9, reserved=0,
)

H_MAX_DISTANCE, H_TSP_POINT_ALL, H_TSP_POINT_K, H_TSP_DUBINS_ALL, H_TSP_DUBINS_K = range(5)
OBST_NONE, OBST_BINARY = 0, 1
EDGE_COVERAGE, EDGE_SLOW = 1, 2
F_INFEASIBLE, F_THROWS, F_RIBBON_OVF, F_DUBINS_ERR, F_GOAL, F_DONE = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20
F_RIBBON_LOST = 0x40   # the sweep ran out of its 64 ribbons per vertex and dropped pieces (always with F_RIBBON_OVF)


def make_config(**kw):
    d = dict(CONFIG_DEFAULTS)
    for k, v in kw.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return PpgpuConfig(**d)


# struct ppgpu_vertex, 64 bytes
VERTEX_DTYPE = np.dtype([
    ("x", "<f8"), ("y", "<f8"), ("heading", "<f8"), ("speed", "<f8"), ("time", "<f8"), ("g", "<f8"),
    ("coverage_completed_time", "<f8"), ("ribbon_offset", "<i4"), ("ribbon_count", "<i4"),
])
assert VERTEX_DTYPE.itemsize == 64

# struct ppgpu_edge_result, 128 bytes
RESULT_DTYPE = np.dtype([
    ("flags", "<u4"), ("info", "<u4"), ("true_cost", "<f8"), ("collision_penalty", "<f8"), ("approx_cost", "<f8"),
    ("end_x", "<f8"), ("end_y", "<f8"), ("end_heading", "<f8"), ("end_speed", "<f8"), ("end_time", "<f8"),
    ("g", "<f8"), ("h", "<f8"), ("f", "<f8"), ("coverage_completed_time", "<f8"), ("param", "<f8", (3,)),
])
assert RESULT_DTYPE.itemsize == 128

# struct ppgpu_wrapper_edge, 96 bytes
WRAPPER_EDGE_DTYPE = np.dtype([
    ("vertex", "<i4"), ("coverage_allowed", "<i4"), ("qi", "<f8", (3,)), ("param", "<f8", (3,)), ("rho", "<f8"),
    ("type", "<i4"), ("reserved", "<i4"), ("speed", "<f8"), ("start_time", "<f8"), ("end_time", "<f8"),
])
assert WRAPPER_EDGE_DTYPE.itemsize == 96


def edge_pack(vertex, target, cfg):
    """ppgpu_edge_pack()."""
    vertex = np.asarray(vertex, dtype=np.uint64)
    target = np.asarray(target, dtype=np.uint64)
    cfg = np.asarray(cfg, dtype=np.uint64)
    return ((cfg & np.uint64(0xFF)) << np.uint64(56)) | ((vertex & np.uint64(0xFFFFFF)) << np.uint64(32)) | target
