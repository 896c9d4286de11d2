"""N > 1 path on CPU: two gloo ranks shard one iteration's batch exactly as bench.py does on RCCL, each costing
its shard with the CPU oracle; the gathered-and-reduced incumbent must equal the single-process one."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _world_and_inputs():
    import oracle as orc
    from path_planner_amd import workloads
    w = workloads.config2(n_samples=256)
    world = orc.World(w.cfg, w.grid, w.res, w.obst)
    return orc, w, world


def _cost(world, w, samples):
    from path_planner_amd.types import edge_pack
    n = len(samples)
    e = edge_pack(np.zeros(4 * n, dtype=np.uint64), np.repeat(np.arange(n), 4), np.tile(np.arange(4), n))
    return world.cost_edges(w.root(), w.ribbons4, samples[:, 0], samples[:, 1], samples[:, 2], e)


def _rank_main(rank, world_size, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from path_planner_amd import sharding
    from path_planner_amd.types import F_INFEASIBLE
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    orc, w, world = _world_and_inputs()
    lo, hi = sharding.shard_attempts(total, rank, world_size)
    samples = world.add_samples(w.bounds6, w.seed, w.ribbons4, lo, hi - lo)      # skip lo attempts, draw hi-lo
    res = _cost(world, w, samples)
    max_edges = 4 * ((total + world_size - 1) // world_size)
    key = sharding.local_best_key(res["f"], (res["flags"] & F_INFEASIBLE) == 0, sharding.edge_index_base(rank, max_edges))
    t = torch.from_numpy(key.view(np.int64).copy())
    gathered = [torch.zeros(2, dtype=torch.int64) for _ in range(world_size)]
    dist.all_gather(gathered, t)                                                  # the one collective per iteration
    allk = np.stack([g.numpy().view(np.uint64) for g in gathered])
    best = sharding.combine_keys(allk)
    q.put((rank, samples, res["f"].copy(), res["flags"].copy(), best))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_iteration_matches_single_process():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from path_planner_amd import sharding
    from path_planner_amd.types import F_INFEASIBLE
    total, world_size = 301, 2               # odd on purpose: shard sizes differ
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world_size, port, total, q)) for r in range(world_size)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in range(world_size)], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    orc, w, world = _world_and_inputs()
    full = world.add_samples(w.bounds6, w.seed, w.ribbons4, 0, total)
    assert np.array_equal(np.concatenate([o[1] for o in outs]), full), "the shards must tile the unsharded stream"
    res = _cost(world, w, full)
    # translate the single-process best edge into the sharded global id space
    ok = (res["flags"] & F_INFEASIBLE) == 0
    single = sharding.local_best_key(res["f"], ok)
    max_edges = 4 * ((total + world_size - 1) // world_size)
    n0 = len(outs[0][1])
    e = int(single[1])
    expect_idx = e if e < 4 * n0 else sharding.edge_index_base(1, max_edges) + (e - 4 * n0)
    assert outs[0][4].tolist() == outs[1][4].tolist()                 # every rank ends with the same incumbent
    assert int(outs[0][4][0]) == int(single[0])                       # same f, bit for bit
    assert int(outs[0][4][1]) == expect_idx


def test_shard_attempts_tile_and_balance():
    from path_planner_amd import sharding
    for total in (0, 1, 7, 65536, 262144, 100003):
        for world in (1, 2, 3, 8):
            ends = [sharding.shard_attempts(total, r, world) for r in range(world)]
            assert ends[0][0] == 0 and ends[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(ends[:-1], ends[1:]))
            sizes = [b - a for a, b in ends]
            assert max(sizes) - min(sizes) <= 1


def test_key_order_matches_float_order_and_ties_take_lowest_index():
    from path_planner_amd import sharding
    f = np.array([3.5, 1.25, 1.25, 7.0, 0.0, 1e-300])
    assert np.array_equal(np.argsort(sharding.f_bits(f), kind="stable"), np.argsort(f, kind="stable"))
    k = sharding.local_best_key([5.0, 2.0, 2.0, 9.0], [True, True, True, True], base=100)
    assert k[1] == 101
    assert np.array_equal(sharding.local_best_key([1.0], [False]), sharding.NO_KEY)
    g = sharding.combine_keys([[sharding.f_bits(2.0), 900], [sharding.f_bits(2.0), 17], sharding.NO_KEY])
    assert g[1] == 17


def test_bench_launches_its_own_ranks_and_fails_at_the_device_not_at_the_launcher():
    """`python bench.py --gpus 2` with no launcher around it starts the two ranks itself (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set per rank, before any GPU call).  On a box without a GPU each rank must die in ppgpu_create — the product has no
    CPU fallback — and the launcher must stop the others and report failure promptly instead of hanging."""
    import subprocess
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU suite runs the launcher for real (test_gpu_bench.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert "ppgpu_create" in out.stderr and "WORLD_SIZE" not in out.stderr, out.stderr[-2000:]
    assert "stopping the other ranks" in out.stderr
    assert out.stdout.strip() == ""
    assert time.time() - t0 < 120


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=4" in out.stderr
