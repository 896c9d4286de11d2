"""ctypes binding of the C entry points of path_planner_amd/host/libpp_hostcore.so (src/host_c_api.cpp): the host-side classes
(State, Ribbon, RibbonManager, GridWorldMap, obstacle managers, DubinsWrapper) as the CPU tests see them.  No GPU needed:
the library carries no device code and does not load the HIP runtime."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SO = os.path.join(ROOT, "path_planner_amd", "host", "libpp_hostcore.so")

vp, dbl, i32, i64, u32 = C.c_void_p, C.c_double, C.c_int, C.c_long, C.c_uint

H = C.CDLL(HOST_SO)

for _n, _r, _a in [
    ("pph_state_yaw", dbl, [dbl]),
    ("pph_state_heading_to", dbl, [dbl, dbl, dbl, dbl]),
    ("pph_state_distance_to", dbl, [vp, dbl, dbl]),
    ("pph_state_heading_difference", dbl, [vp, dbl]),
    ("pph_state_move", None, [vp, dbl]),
    ("pph_state_push", None, [vp, dbl, vp]),
    ("pph_state_interpolate", None, [vp, vp, dbl, vp]),
    ("pph_state_to_string", i32, [vp, i32, C.c_char_p, i32]),
    ("pph_set_ribbon_width", None, [dbl]),
    ("pph_get_ribbon_width", dbl, []),
    ("pph_ribbon_projection", None, [vp, dbl, dbl, vp]),
    ("pph_ribbon_contains", i32, [vp, dbl, dbl, i32]),
    ("pph_ribbon_contains_projection", i32, [vp, dbl, dbl]),
    ("pph_ribbon_distance", dbl, [vp, dbl, dbl]),
    ("pph_ribbon_length", dbl, [vp]),
    ("pph_ribbon_covered", i32, [vp, i32]),
    ("pph_ribbon_split", None, [vp, dbl, dbl, i32, vp]),
    ("pph_ribbon_end_states", None, [vp, vp, vp]),
    ("pph_ribbons_add", i32, [vp, i32, i32, dbl, dbl, dbl, dbl]),
    ("pph_ribbons_cover", i32, [vp, i32, i32, dbl, dbl, i32]),
    ("pph_ribbons_cover_between", i32, [vp, i32, i32, dbl, dbl, dbl, dbl, i32]),
    ("pph_ribbons_min_distance", dbl, [vp, i32, dbl, dbl]),
    ("pph_ribbons_heuristic", dbl, [vp, i32, i32, i32, dbl, dbl, dbl, dbl]),
    ("pph_ribbons_nearest_endpoint", i32, [vp, i32, vp, vp]),
    ("pph_ribbons_project", None, [vp, i32, vp]),
    ("pph_ribbons_near_states", i32, [vp, i32, vp, dbl, vp, i32]),
    ("pph_ribbons_total_uncovered_length", dbl, [vp, i32]),
    ("pph_ribbons_dump", i32, [vp, i32, C.c_char_p, i32]),
    ("pph_grid_load_text", vp, [C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(dbl)]),
    ("pph_grid_free", None, [vp]),
    ("pph_grid_extremes", None, [vp, vp]),
    ("pph_grid_is_blocked_many", None, [vp, i64, vp, vp, vp]),
    ("pph_grid_cells", None, [vp, vp]),
    ("pph_base_map_is_blocked", i32, [dbl, dbl]),
    ("pph_base_map_extremes", None, [vp]),
    ("pph_obst_create", vp, [i32]),
    ("pph_obst_free", None, [vp]),
    ("pph_obst_update", None, [vp, u32, dbl, dbl, dbl, dbl, dbl, dbl, dbl]),
    ("pph_obst_update_gaussian", None, [vp, u32, dbl, dbl, dbl, dbl, dbl, vp]),
    ("pph_obst_forget", None, [vp, u32]),
    ("pph_obst_collision_exists", dbl, [vp, dbl, dbl, dbl, i32]),
    ("pph_obst_device_rows", i32, [vp, vp, i32]),
    ("pph_wrapper_sample", i32, [vp, dbl, dbl, dbl, dbl, vp]),
    ("pph_wrapper_solve", dbl, [vp, vp, dbl, vp]),
]:
    f = getattr(H, _n)
    f.restype, f.argtypes = _r, _a


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _rows(ribbons4, cap):
    r = np.zeros((cap, 4))
    n = len(ribbons4)
    if n:
        r[:n] = f64(ribbons4).reshape(-1, 4)
    return r, n


def ribbons_cover(ribbons4, x, y, strict, cap=64):
    r, n = _rows(ribbons4, cap)
    m = H.pph_ribbons_cover(r.ctypes.data, n, cap, x, y, 1 if strict else 0)
    return r[:m].copy()


def ribbons_cover_between(ribbons4, x1, y1, x2, y2, strict, cap=64):
    r, n = _rows(ribbons4, cap)
    m = H.pph_ribbons_cover_between(r.ctypes.data, n, cap, x1, y1, x2, y2, 1 if strict else 0)
    return r[:m].copy()


def ribbons_add(ribbons4, x1, y1, x2, y2, cap=64):
    r, n = _rows(ribbons4, cap)
    m = H.pph_ribbons_add(r.ctypes.data, n, cap, x1, y1, x2, y2)
    return r[:m].copy()


def ribbons_heuristic(ribbons4, heuristic, K, x, y, yaw=0.0, turning_radius=8.0):
    r = f64(ribbons4).reshape(-1, 4)
    return H.pph_ribbons_heuristic(r.ctypes.data if r.shape[0] else None, r.shape[0], heuristic, K, turning_radius, x, y, yaw)


def ribbons_min_distance(ribbons4, x, y):
    r = f64(ribbons4).reshape(-1, 4)
    return H.pph_ribbons_min_distance(r.ctypes.data if r.shape[0] else None, r.shape[0], x, y)


def ribbons_nearest_endpoint(ribbons4, s5):
    r = f64(ribbons4).reshape(-1, 4)
    s = f64(s5)
    out = np.zeros(5)
    rc = H.pph_ribbons_nearest_endpoint(r.ctypes.data, r.shape[0], s.ctypes.data, out.ctypes.data)
    return rc, out


def ribbons_near_states(ribbons4, start5, radius, cap=64):
    r = f64(ribbons4).reshape(-1, 4)
    s = f64(start5)
    out = np.zeros((cap, 5))
    n = H.pph_ribbons_near_states(r.ctypes.data if r.shape[0] else None, r.shape[0], s.ctypes.data, radius, out.ctypes.data, cap)
    return out[:n].copy()


class Grid:
    def __init__(self, text):
        rows, cols, res = i32(), i32(), dbl()
        self.h = H.pph_grid_load_text(text.encode(), C.byref(rows), C.byref(cols), C.byref(res))
        self.rows, self.cols, self.res = rows.value, cols.value, res.value

    def __del__(self):
        if getattr(self, "h", None):
            H.pph_grid_free(self.h)

    def extremes(self):
        e = np.zeros(4)
        H.pph_grid_extremes(self.h, e.ctypes.data)
        return e

    def is_blocked(self, x, y):
        x, y = f64(x), f64(y)
        out = np.zeros(x.shape[0], dtype=np.uint8)
        H.pph_grid_is_blocked_many(self.h, x.shape[0], x.ctypes.data, y.ctypes.data, out.ctypes.data)
        return out

    def cells(self):
        out = np.zeros((self.rows, self.cols), dtype=np.uint8)
        H.pph_grid_cells(self.h, out.ctypes.data)
        return out


class Obstacles:
    def __init__(self, gaussian=False):
        self.h = H.pph_obst_create(1 if gaussian else 0)

    def __del__(self):
        if getattr(self, "h", None):
            H.pph_obst_free(self.h)

    def update(self, mmsi, x, y, heading, speed, time, width=0.0, length=0.0):
        H.pph_obst_update(self.h, mmsi, x, y, heading, speed, time, width, length)

    def update_gaussian(self, mmsi, x, y, heading, speed, time, cov4):
        c = f64(cov4)
        H.pph_obst_update_gaussian(self.h, mmsi, x, y, heading, speed, time, c.ctypes.data)

    def forget(self, mmsi):
        H.pph_obst_forget(self.h, mmsi)

    def collision_exists(self, x, y, t, strict=True):
        return H.pph_obst_collision_exists(self.h, x, y, t, 1 if strict else 0)

    def device_rows(self, width):
        out = np.zeros(4096)
        n = H.pph_obst_device_rows(self.h, out.ctypes.data, out.shape[0])
        return out[:n].reshape(-1, width).copy()
