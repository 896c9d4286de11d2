"""-m gpu: bench.py's contract on the one-GPU box — the JSON line and its roofline / baseline objects, the self-launcher
(`python bench.py --gpus N` with no torchrun around it), and the product's own RCCL communicator on a 1-rank world."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PPGPU_PREPASS_MIN_EDGES")}


def test_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, BENCH, "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=_env())
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["unit"] == "edges/s" and line["dtype"] == "f64"
    assert line["config"]["workload"].startswith("cfg3_2048")
    r = line["roofline"]
    assert r["bound"] == "fp64_valu" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["hbm"]["bound"] == "hbm" and r["hbm"]["peak"] == 8000.0
    assert line["ms_per_step_median"] > 0 and line["ms_per_step_p99"] >= line["ms_per_step_median"]
    assert line["e2e_ms_per_step"] > line["ms_per_step_median"]           # the records' D2H copy is in it
    assert line["value"] > 1e7                                            # BASELINE's target on one MI355X


def test_self_launch_with_more_ranks_than_devices_fails_at_the_device():
    """Two ranks on a one-GPU box: rank 1 has no device, dies in ppgpu_create, and the launcher stops rank 0 (which would otherwise
    wait in the rendezvous) instead of hanging."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has a second GPU")
    t0 = time.time()
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=_env())
    assert out.returncode != 0
    assert "ppgpu_create" in out.stderr and "device index out of range" in out.stderr, out.stderr[-3000:]
    assert "stopping the other ranks" in out.stderr
    assert time.time() - t0 < 300


def test_self_launch_rehearsal_two_ranks_on_one_device():
    """PP_BENCH_REHEARSAL=1: the launcher, the sharding of the batch and the aggregation over ranks with both ranks on device 0
    (gloo + host combine, because RCCL takes one rank per device): n_gpus = 2, twice the edges of one rank, one JSON line."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=dict(_env(), PP_BENCH_REHEARSAL="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and "rehearsal" in line and line["launched_by"] == "bench.py itself"
    assert line["config"]["samples_per_iter_total"] == 2 * 65536
    per_rank = line["config"]["edges_per_iter_per_gpu"]
    total = line["value"] * line["ms_per_step"] * 1e-3
    assert 1.8 * per_rank < total < 2.2 * per_rank


def test_handle_owned_communicator_one_rank():
    """ppgpu_comm_unique_id / ppgpu_comm_init_rank / ppgpu_comm_info / ppgpu_allreduce_best(comm = NULL) / ppgpu_comm_destroy on
    hardware with the only world a one-GPU box can form."""
    import torch
    from path_planner_amd import api
    ctx = api.Context(0)
    uid = api.Context.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.comm_init_rank(1, 0, uid)
    assert ctx.comm_info() == (1, 0)
    with pytest.raises(api.PpgpuError):
        ctx.comm_init_rank(1, 0, uid)                       # one communicator per handle
    key = torch.tensor([0x4059000000000000, 777], dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    ctx.allreduce_best(key.data_ptr())
    ctx.synchronize()
    assert key.cpu().tolist() == [0x4059000000000000, 777]
    ctx.comm_destroy()
    with pytest.raises(api.PpgpuError):
        ctx.allreduce_best(key.data_ptr())                  # no communicator any more
    ctx.close()


def test_the_multi_rank_path_with_a_world_of_one():
    """PP_BENCH_FORCE_COMM=1: everything `--gpus N` does beyond one rank — torch's NCCL process group, the unique id made by rank 0 and
    broadcast through it, ppgpu_comm_init_rank, one ppgpu_allreduce_best per step on the step's stream, the max-over-ranks of
    the timing, the teardown — with the only world a one-GPU box can form.  rccl_ranks comes back from ncclCommCount."""
    out = subprocess.run([sys.executable, BENCH, "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                         env=dict(_env(), PP_BENCH_FORCE_COMM="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1
    assert line["value"] > 1e7 and line["workload_stats"]["best_edge"] >= 0
