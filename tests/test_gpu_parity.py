"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (run through gpurun)"
    return torch


def _dense(torch, ctx, nv, n, mask, stride=8):
    from path_planner_amd import api
    from path_planner_amd.types import RESULT_DTYPE
    ne = api.Context.dense_edge_count(nv, n, mask)
    d_res = torch.zeros(ne * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0")
    d_child = torch.zeros(ne * stride * 4, dtype=torch.float64, device="cuda:0")
    ctx.cost_edges_dense(0, nv, 0, n, mask, d_res.data_ptr(), d_child.data_ptr(), stride)
    ctx.synchronize()
    return d_res.cpu().numpy().view(RESULT_DTYPE), d_child.cpu().numpy().reshape(ne, stride, 4)


def _setup(w, n_samples):
    from path_planner_amd import api
    import oracle as orc
    ctx = api.Context(0)
    ctx.set_config(w.cfg)
    ctx.set_grid(w.grid, w.res)
    ctx.set_obstacles(w.obst)
    ctx.set_vertices(w.root(), w.ribbons4)
    ctx.sampler_init(w.bounds6, w.seed, w.ribbons4)
    n = ctx.sampler_add(n_samples)
    world = orc.World(w.cfg, w.grid, w.res, w.obst)
    cs = world.add_samples(w.bounds6, w.seed, w.ribbons4, 0, n_samples)
    return ctx, world, n, cs


@pytest.mark.parametrize("cfgname,n_samples", [("cfg1", 64), ("cfg2", 1024), ("cfg3", 1024)])
def test_sampler_and_dense_costing_match_oracle(torch_cuda, cfgname, n_samples):
    from path_planner_amd import workloads
    from path_planner_amd.types import edge_pack
    from parity import compare_results
    w = workloads.by_name(cfgname)
    ctx, world, n, cs = _setup(w, n_samples)
    gs = ctx.get_samples()
    assert n == cs.shape[0]
    assert np.array_equal(gs[:, :3], cs[:, :3]), "sampler stream must be bit-identical"
    gpu, gchild = _dense(torch_cuda, ctx, 1, n, 0xF)
    ne = len(gpu)
    e = edge_pack(np.zeros(ne, dtype=np.uint64), np.repeat(np.arange(n), 4), np.tile(np.arange(4), n))
    cpu, cchild = world.cost_edges(w.root(), w.ribbons4, cs[:, 0], cs[:, 1], cs[:, 2], e, stride=8)
    rep = compare_results(gpu, cpu, gchild, cchild)
    print(cfgname, rep)
    assert rep["ok"], rep
