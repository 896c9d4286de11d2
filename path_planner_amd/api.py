"""ctypes binding of libppgpu.so (include/ppgpu.h).  Glue for tests/ and bench.py.

Importing this module REQUIRES the in-tree HIP library; there is no fallback.
"""
import ctypes as C
import os

import numpy as np

from .types import PpgpuConfig, RESULT_DTYPE, VERTEX_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPGPU_LIB_OVERRIDE") or os.path.join(_HERE, "csrc", "libppgpu.so")   # override: tools/ablate.py only


class PpgpuError(RuntimeError):
    pass


def _load():
    # The PyTorch wheel carries its own ROCm runtime; whichever libamdhip64 a process loads first serves every later user.  Loaded
    # after this library's, torch reports "No HIP GPUs are available".  tests/ and bench.py pair the two, so torch goes first
    # whenever it is installed (importing it does not touch the device).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). path_planner_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
    sig = {
        "ppgpu_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "ppgpu_destroy": (C.c_int, [vp]),
        "ppgpu_last_error": (C.c_char_p, []),
        "ppgpu_set_stream": (C.c_int, [vp, vp]),
        "ppgpu_synchronize": (C.c_int, [vp]),
        "ppgpu_reserve_samples": (C.c_int, [vp, i64, i32]),
        "ppgpu_growth_stats": (C.c_int, [vp, C.POINTER(u64), C.POINTER(dbl)]),
        "ppgpu_device_alloc": (C.c_int, [vp, u64, C.POINTER(vp)]),
        "ppgpu_device_free": (C.c_int, [vp, vp]),
        "ppgpu_device_read": (C.c_int, [vp, vp, vp, u64]),
        "ppgpu_copy_engine_read": (C.c_int, [vp, vp, vp, u64]),
        "ppgpu_copy_engine_wait": (C.c_int, [vp]),
        "ppgpu_heuristic_host": (C.c_int, [vp, i32, vp, vp, vp, vp, vp]),
        "ppgpu_expand_capacity": (C.c_int64, [i32, i32]),
        "ppgpu_expand_host": (C.c_int, [vp, i32, vp, i32, vp, vp, i32, C.POINTER(C.c_int64), vp, vp, vp, i32]),
        "ppgpu_enable_timing": (C.c_int, [vp, i32]),
        "ppgpu_last_timing": (C.c_int, [vp, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl)]),
        "ppgpu_past_timing": (C.c_int, [vp, i32, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl)]),
        "ppgpu_last_cover_edges": (C.c_int, [vp, C.POINTER(i64)]),
        "ppgpu_set_config": (C.c_int, [vp, C.POINTER(PpgpuConfig)]),
        "ppgpu_set_grid": (C.c_int, [vp, vp, i32, i32, dbl]),
        "ppgpu_set_obstacles": (C.c_int, [vp, i32, i32, vp]),
        "ppgpu_set_gaussian_obstacles": (C.c_int, [vp, i32, vp, i32]),
        "ppgpu_set_vertices": (C.c_int, [vp, i32, vp, i32, vp]),
        "ppgpu_sampler_init": (C.c_int, [vp, vp, u64, i32, vp]),
        "ppgpu_sampler_add": (C.c_int, [vp, i64, C.POINTER(i64)]),
        "ppgpu_sampler_skip": (C.c_int, [vp, i64]),
        "ppgpu_set_samples": (C.c_int, [vp, i64, vp, vp, vp]),
        "ppgpu_set_extra_targets": (C.c_int, [vp, i32, vp, vp, vp, C.POINTER(i64)]),
        "ppgpu_get_samples": (C.c_int, [vp, i64, i64, vp]),
        "ppgpu_num_samples": (i64, [vp]),
        "ppgpu_dubins_lengths": (C.c_int, [vp, i32, i32, vp]),
        "ppgpu_select_nearest": (C.c_int, [vp, i32, i32, i32, vp, vp]),
        "ppgpu_expand_order": (C.c_int, [vp, i32, i32, i32, vp, C.POINTER(u32)]),
        "ppgpu_order_fallbacks": (u64, [vp]),
        "ppgpu_cost_edges_dense": (C.c_int, [vp, i32, i32, i64, i64, u32, vp, vp, i32]),
        "ppgpu_cost_edges_list": (C.c_int, [vp, i64, vp, vp, vp, i32]),
        "ppgpu_cost_edges_host": (C.c_int, [vp, i64, vp, vp, vp, i32]),
        "ppgpu_cost_wrapper_edges_host": (C.c_int, [vp, i64, vp, vp, vp, i32]),
        "ppgpu_dense_edge_count": (i64, [i32, i64, u32]),
        "ppgpu_best_edge": (C.c_int, [vp, i64, vp, i32, u64, vp]),
        "ppgpu_key_min": (C.c_int, [vp, i32, vp, vp]),
        "ppgpu_allreduce_best": (C.c_int, [vp, vp, vp]),
        "ppgpu_comm_unique_id": (C.c_int, [vp]),
        "ppgpu_comm_init_rank": (C.c_int, [vp, i32, i32, vp]),
        "ppgpu_comm_init_all": (C.c_int, [C.POINTER(vp), i32]),
        "ppgpu_comm_info": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "ppgpu_comm_destroy": (C.c_int, [vp]),
        "ppgpu_comm_abort": (C.c_int, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    return lib, list(sig)


LIB, EXPORTS = _load()


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    return int(a)   # raw device pointer (e.g. torch.Tensor.data_ptr())


class Context:
    """One ppgpu_ctx.  Methods mirror the C entry points one to one."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = LIB.ppgpu_create(device, C.byref(h))
        if rc != 0:
            raise PpgpuError(f"ppgpu_create: {LIB.ppgpu_last_error().decode()}")
        self._h = h

    def _ck(self, rc, what):
        if rc != 0:
            raise PpgpuError(f"{what} failed ({rc}): {LIB.ppgpu_last_error().decode()}")

    def close(self):
        if self._h:
            LIB.ppgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        self._ck(LIB.ppgpu_set_stream(self._h, stream_ptr), "ppgpu_set_stream")

    def enable_timing(self, on=True):
        self._ck(LIB.ppgpu_enable_timing(self._h, 1 if on else 0), "ppgpu_enable_timing")

    def last_timing(self):
        """(solve, pose sweep, cover sweep, heuristic) kernel durations in ms of the last costing launch."""
        t = [C.c_double() for _ in range(4)]
        self._ck(LIB.ppgpu_last_timing(self._h, *[C.byref(x) for x in t]), "ppgpu_last_timing")
        return tuple(x.value for x in t)

    def past_timing(self, back):
        """The same for the launch `back` launches ago (0 = the last one, at most 7)."""
        t = [C.c_double() for _ in range(4)]
        self._ck(LIB.ppgpu_past_timing(self._h, back, *[C.byref(x) for x in t]), "ppgpu_past_timing")
        return tuple(x.value for x in t)

    def last_cover_edges(self):
        """Edges of the last costing launch that pp_k_cover_sweep visited (the others were finished by the approach prepass)."""
        n = C.c_int64()
        self._ck(LIB.ppgpu_last_cover_edges(self._h, C.byref(n)), "ppgpu_last_cover_edges")
        return n.value

    def reserve_samples(self, max_samples, max_vertices=16):
        self._ck(LIB.ppgpu_reserve_samples(self._h, max_samples, max_vertices), "ppgpu_reserve_samples")

    def synchronize(self):
        self._ck(LIB.ppgpu_synchronize(self._h), "ppgpu_synchronize")

    def set_config(self, cfg):
        self.cfg = cfg
        self._ck(LIB.ppgpu_set_config(self._h, C.byref(cfg)), "ppgpu_set_config")

    def set_grid(self, cells, resolution):
        if cells is None:
            self._ck(LIB.ppgpu_set_grid(self._h, None, 0, 0, 0.0), "ppgpu_set_grid")
            return
        cells = np.ascontiguousarray(cells, dtype=np.uint8)
        self._ck(LIB.ppgpu_set_grid(self._h, _ptr(cells), cells.shape[0], cells.shape[1], float(resolution)), "ppgpu_set_grid")

    def set_obstacles(self, obst7, model=1):
        if obst7 is None or len(obst7) == 0:
            self._ck(LIB.ppgpu_set_obstacles(self._h, 0, 0, None), "ppgpu_set_obstacles")
            return
        o = np.ascontiguousarray(obst7, dtype=np.float64).reshape(-1, 7)
        self._ck(LIB.ppgpu_set_obstacles(self._h, model, o.shape[0], _ptr(o)), "ppgpu_set_obstacles")

    def set_gaussian_obstacles(self, rows):
        """rows: n x 5 {x, y, heading, speed, time} (default covariance) or n x 9 (+ row-major 2x2 covariance)."""
        o = np.ascontiguousarray(rows, dtype=np.float64)
        if o.size == 0:
            self._ck(LIB.ppgpu_set_gaussian_obstacles(self._h, 0, None, 0), "ppgpu_set_gaussian_obstacles")
            return
        assert o.ndim == 2 and o.shape[1] in (5, 9)
        self._ck(LIB.ppgpu_set_gaussian_obstacles(self._h, o.shape[0], _ptr(o), 1 if o.shape[1] == 9 else 0), "ppgpu_set_gaussian_obstacles")

    def set_vertices(self, vertices, ribbons4):
        v = np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE)
        r = np.ascontiguousarray(ribbons4, dtype=np.float64).reshape(-1, 4)
        self._ck(LIB.ppgpu_set_vertices(self._h, v.shape[0], _ptr(v), r.shape[0], _ptr(r) if r.shape[0] else None),
                 "ppgpu_set_vertices")

    def sampler_init(self, bounds6, seed, ribbons4=None):
        b = np.ascontiguousarray(bounds6, dtype=np.float64)
        if ribbons4 is None:
            self._ck(LIB.ppgpu_sampler_init(self._h, _ptr(b), int(seed), -1, None), "ppgpu_sampler_init")
        else:
            r = np.ascontiguousarray(ribbons4, dtype=np.float64).reshape(-1, 4)
            self._ck(LIB.ppgpu_sampler_init(self._h, _ptr(b), int(seed), r.shape[0], _ptr(r) if r.shape[0] else None),
                     "ppgpu_sampler_init")

    def sampler_add(self, n_attempts):
        tot = C.c_int64()
        self._ck(LIB.ppgpu_sampler_add(self._h, int(n_attempts), C.byref(tot)), "ppgpu_sampler_add")
        return tot.value

    def copy_engine_read(self, h_pinned_ptr, d_ptr, nbytes):
        self._ck(LIB.ppgpu_copy_engine_read(self._h, int(h_pinned_ptr), int(d_ptr), int(nbytes)), "ppgpu_copy_engine_read")

    def copy_engine_wait(self):
        self._ck(LIB.ppgpu_copy_engine_wait(self._h), "ppgpu_copy_engine_wait")

    def growth_stats(self):
        n, sec = C.c_uint64(), C.c_double()
        self._ck(LIB.ppgpu_growth_stats(self._h, C.byref(n), C.byref(sec)), "ppgpu_growth_stats")
        return int(n.value), float(sec.value)

    def sampler_skip(self, n_attempts):
        self._ck(LIB.ppgpu_sampler_skip(self._h, int(n_attempts)), "ppgpu_sampler_skip")

    def set_samples(self, x, y, heading):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        h = np.ascontiguousarray(heading, dtype=np.float64)
        self._ck(LIB.ppgpu_set_samples(self._h, x.shape[0], _ptr(x), _ptr(y), _ptr(h)), "ppgpu_set_samples")

    def num_samples(self):
        return LIB.ppgpu_num_samples(self._h)

    def get_samples(self, first=0, n=None):
        if n is None:
            n = self.num_samples() - first
        out = np.zeros((n, 5), dtype=np.float64)
        self._ck(LIB.ppgpu_get_samples(self._h, first, n, _ptr(out)), "ppgpu_get_samples")
        return out

    def dubins_lengths(self, v0, nv, d_lengths):
        self._ck(LIB.ppgpu_dubins_lengths(self._h, v0, nv, _ptr(d_lengths)), "ppgpu_dubins_lengths")

    def set_extra_targets(self, x, y, heading):
        """Explicit (non-sample) targets; returns the index of the first one for edge descriptors."""
        x, y, heading = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, heading)]
        first = C.c_int64(0)
        self._ck(LIB.ppgpu_set_extra_targets(self._h, x.shape[0], _ptr(x) if x.shape[0] else None, _ptr(y) if x.shape[0] else None,
                                             _ptr(heading) if x.shape[0] else None, C.byref(first)), "ppgpu_set_extra_targets")
        return int(first.value)

    def select_nearest(self, v0, nv, k):
        idx = np.zeros((nv, 2, k), dtype=np.int32)
        ln = np.zeros((nv, 2, k), dtype=np.float64)
        self._ck(LIB.ppgpu_select_nearest(self._h, v0, nv, k, _ptr(idx), _ptr(ln)), "ppgpu_select_nearest")
        return idx, ln

    def expand_order(self, nv, k):
        """The k winners per (vertex, radius) in the order expand() pushes them (the reference's heap array); (idx, fallbacks)."""
        idx = np.zeros((nv, 2, k), dtype=np.int32)
        fb = C.c_uint32(0)
        self._ck(LIB.ppgpu_expand_order(self._h, 0, nv, k, _ptr(idx), C.byref(fb)), "ppgpu_expand_order")
        return idx, int(fb.value)

    def order_fallbacks(self):
        return int(LIB.ppgpu_order_fallbacks(self._h))

    def cost_edges_dense(self, v0, nv, s0, ns, cfg_mask, d_results, d_child=None, stride=0):
        self._ck(LIB.ppgpu_cost_edges_dense(self._h, v0, nv, s0, ns, cfg_mask, _ptr(d_results), _ptr(d_child), stride),
                 "ppgpu_cost_edges_dense")

    def cost_edges_list(self, n, d_edges, d_results, d_child=None, stride=0):
        self._ck(LIB.ppgpu_cost_edges_list(self._h, n, _ptr(d_edges), _ptr(d_results), _ptr(d_child), stride),
                 "ppgpu_cost_edges_list")

    def heuristic_host(self, poses3, ribbon_lists):
        """h of each pose {x, y, heading} with its own ribbon list (Vertex::computeApproxToGo); returns (h, flags)."""
        ps = np.ascontiguousarray(poses3, dtype=np.float64).reshape(-1, 3)
        counts = np.array([len(r) for r in ribbon_lists], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1, 4) for r in ribbon_lists]) if counts.sum() else np.zeros((0, 4)))
        out = np.zeros(len(ps), dtype=np.float64)
        fl = np.zeros(len(ps), dtype=np.uint32)
        self._ck(LIB.ppgpu_heuristic_host(self._h, len(ps), _ptr(ps), _ptr(counts), _ptr(flat) if len(flat) else None, _ptr(out), _ptr(fl)), "ppgpu_heuristic_host")
        return out, fl

    def expand_host(self, vertices, ribbons4, nearest3, k, stride=0):
        """SamplingBasedPlanner::expand for several vertices in one round trip: (descriptors, records, child ribbons)."""
        v = np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE)
        r = np.ascontiguousarray(ribbons4, dtype=np.float64).reshape(-1, 4)
        nz = np.ascontiguousarray(nearest3, dtype=np.float64).reshape(-1, 3)
        cap = int(LIB.ppgpu_expand_capacity(v.shape[0], k))
        e = np.zeros(cap, dtype=np.uint64)
        res = np.zeros(cap, dtype=RESULT_DTYPE)
        child = np.zeros((cap, stride, 4), dtype=np.float64) if stride > 0 else None
        n = C.c_int64(0)
        self._ck(LIB.ppgpu_expand_host(self._h, v.shape[0], _ptr(v), r.shape[0], _ptr(r) if r.shape[0] else None, _ptr(nz), k, C.byref(n),
                                       _ptr(e), _ptr(res), _ptr(child), stride), "ppgpu_expand_host")
        m = n.value
        return e[:m], res[:m], (child[:m] if child is not None else None)

    def cost_edges_host(self, edges, stride=0):
        e = np.ascontiguousarray(edges, dtype=np.uint64)
        res = np.zeros(e.shape[0], dtype=RESULT_DTYPE)
        child = np.zeros((e.shape[0], stride, 4), dtype=np.float64) if stride > 0 else None
        self._ck(LIB.ppgpu_cost_edges_host(self._h, e.shape[0], _ptr(e), _ptr(res), _ptr(child), stride), "ppgpu_cost_edges_host")
        return (res, child) if stride > 0 else res

    def cost_wrapper_edges_host(self, wedges, stride=0):
        from .types import WRAPPER_EDGE_DTYPE
        e = np.ascontiguousarray(wedges, dtype=WRAPPER_EDGE_DTYPE)
        res = np.zeros(e.shape[0], dtype=RESULT_DTYPE)
        child = np.zeros((e.shape[0], stride, 4), dtype=np.float64) if stride > 0 else None
        self._ck(LIB.ppgpu_cost_wrapper_edges_host(self._h, e.shape[0], _ptr(e), _ptr(res), _ptr(child), stride), "ppgpu_cost_wrapper_edges_host")
        return (res, child) if stride > 0 else res

    @staticmethod
    def dense_edge_count(nv, ns, cfg_mask):
        return LIB.ppgpu_dense_edge_count(nv, ns, cfg_mask)

    def best_edge(self, n, d_results, d_key2, goal_only=False, base=0):
        self._ck(LIB.ppgpu_best_edge(self._h, n, _ptr(d_results), 1 if goal_only else 0, base, _ptr(d_key2)), "ppgpu_best_edge")

    def key_min(self, n, d_keys, d_key2):
        self._ck(LIB.ppgpu_key_min(self._h, n, _ptr(d_keys), _ptr(d_key2)), "ppgpu_key_min")

    # ---- the communicator of a sharded iteration (RCCL, loaded by the library at the first call)
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from ncclGetUniqueId: rank 0 makes them, every rank passes them to comm_init_rank."""
        buf = (C.c_uint8 * 128)()
        rc = LIB.ppgpu_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise PpgpuError(f"ppgpu_comm_unique_id failed ({rc}): {LIB.ppgpu_last_error().decode()}")
        return bytes(buf)

    def comm_init_rank(self, world, rank, id128):
        assert len(id128) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(id128)
        self._ck(LIB.ppgpu_comm_init_rank(self._h, world, rank, C.cast(buf, C.c_void_p)), "ppgpu_comm_init_rank")

    def comm_info(self):
        """(ranks, this handle's rank) as RCCL reports them (ncclCommCount, ncclCommUserRank)."""
        w, r = C.c_int32(), C.c_int32()
        self._ck(LIB.ppgpu_comm_info(self._h, C.byref(w), C.byref(r)), "ppgpu_comm_info")
        return w.value, r.value

    def comm_destroy(self):
        self._ck(LIB.ppgpu_comm_destroy(self._h), "ppgpu_comm_destroy")

    def allreduce_best(self, d_key2, comm=None):
        """One collective per iteration: the global incumbent, in place on d_key2 (the handle's own communicator by default)."""
        self._ck(LIB.ppgpu_allreduce_best(self._h, comm, _ptr(d_key2)), "ppgpu_allreduce_best")
