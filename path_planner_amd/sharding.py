"""How one planner iteration's batch is split over the GPUs of a node (SURVEY.md 8 e), as plain
functions so that bench.py and the CPU (gloo) tests exercise the same logic.

* The sample batch of an iteration (`attempts` draws from the StateGenerator stream) is cut into
  `world` contiguous slices; rank r skips the r lower slices of the stream (ppgpu_sampler_skip) and
  generates its own — the union over ranks is exactly the unsharded stream.
* Every rank costs the edges to ITS kept samples; there is no data-path exchange.
* The only exchange is the incumbent: each rank's best key {bit pattern of f, global edge index} is
  all-gathered (16 bytes per rank) and reduced lexicographically.  RCCL has no MINLOC, and a
  64-bit all-reduce(min) cannot carry a full-precision f and an index, hence gather + local min.
"""
import numpy as np

NO_KEY = np.array([0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)


def shard_attempts(total_attempts, rank, world):
    """[lo, hi) of the iteration's attempts that belong to `rank` (contiguous, sizes differ by at most 1)."""
    base, extra = divmod(total_attempts, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def edge_index_base(rank, max_edges_per_rank):
    """Global edge ids: rank r owns [r * max_edges_per_rank, (r + 1) * max_edges_per_rank)."""
    return rank * max_edges_per_rank


def f_bits(f):
    """Monotone integer image of a non-negative double (its IEEE bit pattern)."""
    return np.asarray(f, dtype=np.float64).view(np.uint64)


def local_best_key(f, ok, base=0):
    """What ppgpu_best_edge computes: lexicographic min of (bits(f), base + index) over edges with ok[i]."""
    f = np.asarray(f, dtype=np.float64)
    idx = np.nonzero(np.asarray(ok, dtype=bool))[0]
    if idx.size == 0:
        return NO_KEY.copy()
    fb = f_bits(f[idx])
    m = fb.min()
    first = idx[fb == m].min()
    return np.array([m, base + first], dtype=np.uint64)


def combine_keys(keys):
    """What ppgpu_key_min computes on the gathered [world, 2] keys."""
    keys = np.asarray(keys, dtype=np.uint64).reshape(-1, 2)
    m = keys[:, 0].min()
    cand = keys[keys[:, 0] == m]
    return np.array([m, cand[:, 1].min()], dtype=np.uint64)
