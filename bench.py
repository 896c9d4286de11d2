#!/usr/bin/env python3
"""bench.py — edges costed per second on the BASELINE.json workload, one process per GPU.

One "step" = one planner iteration's hot path over one batch of synthetic input, all of it on
the device: draw the batch of states from the StateGenerator stream (this rank's shard), drop the
ones on blocked cells, generate the Dubins edge from the open vertex to every kept sample under
the four (radius, speed) configurations, cost every edge (collision sweep, dynamic-obstacle
penalty, ribbon coverage, heuristic), min-reduce the best (f, edge).  With N > 1 the batch is
sharded by sample index and one RCCL collective per step combines the per-rank incumbents.

Workload (config.workload): SURVEY.md 8(d) config 3 — 65 536 sample attempts per GPU per
iteration, 2048x2048 occupancy grid at 0.1 m with 10 % blocked, 16 moving obstacles, 5 ribbons,
TSP(K=2) heuristic; edges = root x kept samples x {rho 8, 16} x {2.5, 0.5 m/s}.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X datasheet, non-matrix fp64 (not in the microarch guide)
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="sample attempts per GPU per step")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: SURVEY config 4 — ONE iteration batch of --total attempts split over the ranks "
                         "(262 144 over 8 GPUs = 32 768 each) instead of --batch attempts per GPU")
    ap.add_argument("--total", type=int, default=262144, help="sample attempts per step over all GPUs with --strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--open-vertex-run", action="store_true",
                    help="(default since round 4) also time the 64-open-vertex x 4096-sample launch of SURVEY config 3; "
                         "PP_BENCH_PROFILED=1 (set by the rocprofv3 tools) leaves it and the plan()-level legs out")
    ap.add_argument("--no-plan-level", action="store_true", help="skip the 10 Hz replan loop (config 5) and the plan()-level CPU baseline")
    ap.add_argument("--replan-cycles", type=int, default=100)
    ap.add_argument("--as-rank", type=int, default=None, metavar="R",
                    help="one GPU, no communicator: time the step exactly as rank R of --of N would run it (its slice of the batch, "
                         "the sampler skip over the lower ranks' slices included).  A projection of the worst rank's step, not a scaling curve")
    ap.add_argument("--of", type=int, default=8, metavar="N")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launched ranks are stopped after this many seconds")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N rank processes here, BEFORE this process has touched
    the GPU (it never does: no torch import, no HIP call), each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would set them, and pass rank 0's JSON line through.  A rank that fails stops the others (by PID)."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "PP_BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr, start_new_session=True))
    deadline = time.time() + args.launch_timeout
    rc = 0
    live = set(range(args.gpus))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
        if live and (rc != 0 or time.time() > deadline):
            if rc == 0:
                rc = 124
                print("bench.py: --launch-timeout reached; stopping the ranks", file=sys.stderr, flush=True)
            for r in live:                       # exactly the processes started above
                try:
                    os.killpg(procs[r].pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
            t_end = time.time() + 10
            for r in live:
                try:
                    procs[r].wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    try:
                        os.killpg(procs[r].pid, signal.SIGKILL)
                    except ProcessLookupError:
                        pass
                    procs[r].wait()
            live.clear()
        if live:
            time.sleep(0.05)
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    run_rank(args)


def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")

    import numpy as np
    # torch BEFORE the HIP library: the wheel carries its own ROCm runtime, and whichever libamdhip64 is loaded first serves both
    # (loaded the other way round torch finds "No HIP GPUs").  Importing torch does not touch the device.
    import torch
    import torch.distributed as dist
    from path_planner_amd import api, sharding, workloads
    from path_planner_amd.types import RESULT_DTYPE, F_INFEASIBLE

    # PP_BENCH_REHEARSAL=1 (developer aid for a one-GPU box, never set by the driver): every rank uses device 0, torch's group is
    # gloo and the incumbents are combined through the host — RCCL refuses two ranks on one device, so this rehearses the
    # launcher, the sharding and the aggregation only; the line it prints is marked and is not a measurement
    rehearsal = os.environ.get("PP_BENCH_REHEARSAL") == "1" and world > 1
    if rehearsal:
        local = 0
    ctx = api.Context(local)                 # first device call of the process: no gfx950 device = PpgpuError here, nothing else runs
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    rccl_ranks = None
    # PP_BENCH_FORCE_COMM=1 (developer aid): take the N > 1 path — torch's NCCL group, the unique-id broadcast, the library's own
    # communicator and one ppgpu_allreduce_best per step — with a world of one, which is all a one-GPU box can form
    use_comm = (world > 1 or os.environ.get("PP_BENCH_FORCE_COMM") == "1") and not rehearsal
    if use_comm and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if rehearsal:
        dist.init_process_group("gloo")
    elif use_comm:
        # torch's process group carries the barrier and the max-over-ranks of the timing contract; the data path's one
        # collective per step goes through the product's own communicator (ppgpu_comm_* / ppgpu_allreduce_best)
        dist.init_process_group("nccl", device_id=dev)
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(api.Context.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, src=0)
        torch.cuda.synchronize(dev)
        ctx.comm_init_rank(world, rank, bytes(uid.cpu().numpy().tobytes()))
        rccl_ranks, my = ctx.comm_info()     # read back from RCCL: ncclCommCount / ncclCommUserRank
        assert rccl_ranks == world and my == rank, (rccl_ranks, my)

    w = workloads.config3()
    B = args.batch                                  # attempts per GPU per step (weak scaling: the default)
    # --as-rank R --of N: this one process plays rank R of an N-rank job (no communicator, no collective)
    emu_rank, emu_world = (args.as_rank, args.of) if args.as_rank is not None else (rank, world)
    if args.as_rank is not None and (world != 1 or not (0 <= emu_rank < emu_world)):
        raise SystemExit("bench.py: --as-rank R --of N needs --gpus 1 and 0 <= R < N")
    total_attempts = args.total if args.strong else B * emu_world
    if args.strong:
        B = -(-total_attempts // emu_world)         # the largest shard
    stream = torch.cuda.Stream(dev)          # a real (non-null) stream shared by the kernels, torch events and RCCL
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    ctx.set_config(w.cfg)
    ctx.set_grid(w.grid, w.res)
    ctx.set_obstacles(w.obst)
    ctx.set_vertices(w.root(), w.ribbons4)
    ctx.enable_timing(True)                  # HIP events between the kernels of a costing launch, on this stream

    max_edges = 4 * B
    d_res = torch.zeros(max_edges * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    d_key2 = torch.zeros(2, dtype=torch.int64, device=dev)
    kernel_events = []

    def step_sample():
        ctx.sampler_init(w.bounds6, w.seed, w.ribbons4)
        lo, hi = sharding.shard_attempts(total_attempts, emu_rank, emu_world)   # this rank's slice of the iteration's batch
        if lo:
            ctx.sampler_skip(lo)
        return ctx.sampler_add(hi - lo)          # (the step's one host wait: the kept-sample count)

    def step_cost(n, out=None):
        out = d_res if out is None else out
        ne = 4 * n
        ctx.cost_edges_dense(0, 1, 0, n, 0xF, out.data_ptr())
        ctx.best_edge(ne, out.data_ptr(), d_key2.data_ptr(), goal_only=False, base=sharding.edge_index_base(emu_rank, max_edges))
        if rehearsal:
            mine = d_key2.cpu()
            allk = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(allk, mine)
            d_all = torch.cat(allk).to(dev)
            ctx.key_min(world, d_all.data_ptr(), d_key2.data_ptr())
            ctx.synchronize()
        elif use_comm:
            ctx.allreduce_best(d_key2.data_ptr())     # the one collective of the iteration: 16 B per rank over xGMI + local min
        return ne

    def step(out=None):
        return step_cost(step_sample(), out)

    def fence():
        if world > 1 or use_comm:
            dist.barrier()
        torch.cuda.synchronize(dev)

    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    import gc
    gc.collect()
    gc.disable()                     # (the timed steps are driven from Python: no collector pause inside the timed region)
    for _ in range(args.warmup):     # (straight into the timed region: nothing between the warm-up and the fence lets the device go idle)
        step()
    fence()
    t0 = time.perf_counter()
    edges = 0
    marks[0].record(stream)
    # the library's HIP events between its kernels (six per costing launch: what roofline.kernels_ms is read from) are recorded in
    # the LAST eight timed steps only — each record is ~7 us of idle device between two kernels; PP_BENCH_EVENTS=all records them in
    # every step, =0 in none
    ev_mode = os.environ.get("PP_BENCH_EVENTS", "last8")
    timed_from = 0 if ev_mode == "all" else (args.steps if ev_mode == "0" else max(0, args.steps - 8))
    ctx.enable_timing(timed_from == 0)
    for k in range(args.steps):
        if k == timed_from and k > 0:
            ctx.enable_timing(True)
        edges += step()
        marks[k + 1].record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    step_ms = np.array([marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)])
    # (solve, pose, cover, heuristic) ms of the last timed steps' launches, from the HIP events the library recorded between its
    # kernels on this stream inside the timed region (a ring of 8 sets: read back here, so no step waited for its own events)
    for back in range(min(args.steps - timed_from, 8)):
        kernel_events.append(ctx.past_timing(back))
    ctx.enable_timing(True)
    if not kernel_events:                                   # (PP_BENCH_EVENTS=0: one launch after the clock has stopped, for the breakdown)
        step(); ctx.synchronize()
        kernel_events.append(ctx.past_timing(0))

    tot = torch.tensor([float(edges), elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1 or use_comm:
        e_sum = tot[0:1].clone()
        t_max = tot[1:2].clone()
        dist.all_reduce(e_sum, op=dist.ReduceOp.SUM)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        total_edges, t = float(e_sum.item()), float(t_max.item())
    else:
        total_edges, t = float(edges), elapsed

    # end to end: the same step followed by the D2H copy of its records into pinned host memory (the host A* needs
    # all of them).  Its own loop after the timed region; never part of `value`.
    e2e_ms = None
    if rank == 0:
        h_res = torch.empty(max_edges * RESULT_DTYPE.itemsize, dtype=torch.uint8, pin_memory=True)
    fence()
    n_e2e = max(1, min(args.steps, 10))
    t0 = time.perf_counter()
    for _ in range(n_e2e):
        ne = step()
        if rank == 0:
            h_res[: ne * RESULT_DTYPE.itemsize].copy_(d_res[: ne * RESULT_DTYPE.itemsize], non_blocking=True)
            stream.synchronize()
    fence()
    if rank == 0:
        e2e_ms = 1e3 * (time.perf_counter() - t0) / n_e2e
    # the same with the copy of step i overlapped with step i + 1 ("overlap D2H with the next batch"): two record
    # buffers, the copies on a second stream
    e2e_pipe_ms = None
    fence()
    if rank == 0:
        d_res2 = torch.zeros_like(d_res)
        h_res2 = torch.empty_like(h_res).pin_memory()
        copy_stream = torch.cuda.Stream(dev)
        bufs, hbufs = [d_res, d_res2], [h_res, h_res2]
        copied = [None, None]
    t0 = time.perf_counter()
    for i in range(n_e2e):
        if rank == 0:
            b = i & 1
            if copied[b] is not None:
                stream.wait_event(copied[b])                 # the copy that read this buffer two steps ago
            ne = step(bufs[b])
            done = torch.cuda.Event(); done.record(stream)
            copy_stream.wait_event(done)
            with torch.cuda.stream(copy_stream):
                hbufs[b][: ne * RESULT_DTYPE.itemsize].copy_(bufs[b][: ne * RESULT_DTYPE.itemsize], non_blocking=True)
                copied[b] = torch.cuda.Event(); copied[b].record(copy_stream)
        else:
            step()
    if rank == 0:
        copy_stream.synchronize()
    fence()
    if rank == 0:
        e2e_pipe_ms = 1e3 * (time.perf_counter() - t0) / n_e2e
    # ... and with the copy of step i on an SDMA engine (ppgpu_copy_engine_read: hsa_amd_memory_async_copy) while step i + 1 runs.
    # The hipMemcpyAsync of the two loops above runs as a blit KERNEL (__amd_rocclr_copyBuffer) whatever the SDMA settings say:
    # tools/copy_engine_ab.sh, profiles/r04_copy_engine.txt.
    e2e_sdma_ms, e2e_sdma_err = None, None
    fence()
    t0 = time.perf_counter()
    for i in range(n_e2e):
        if rank == 0 and e2e_sdma_err is None:
            try:
                b = i & 1
                n_kept = step_sample()                       # (its host wait also means: the records of step i - 1 are complete)
                # the bulk copy of step i - 1 is issued AFTER this step's small count read-back, not before: issued first it kept that
                # read-back waiting behind 30 MB on the engine, and the step with it (2.53 ms per step that way)
                if i > 0:
                    ctx.copy_engine_read(hbufs[1 - b].data_ptr(), bufs[1 - b].data_ptr(), ne * RESULT_DTYPE.itemsize)
                ne = step_cost(n_kept, bufs[b])
            except Exception as e:                           # the bench line must not depend on this leg
                e2e_sdma_err = repr(e)
        else:
            step()
    if rank == 0 and e2e_sdma_err is None:
        try:
            ctx.synchronize()
            ctx.copy_engine_read(hbufs[(n_e2e - 1) & 1].data_ptr(), bufs[(n_e2e - 1) & 1].data_ptr(), ne * RESULT_DTYPE.itemsize)   # the last step's records
            ctx.copy_engine_wait()
        except Exception as e:
            e2e_sdma_err = repr(e)
    fence()
    copy_alone_ms = None
    if rank == 0 and e2e_sdma_err is None:
        e2e_sdma_ms = 1e3 * (time.perf_counter() - t0) / n_e2e
        e2e_sdma_ok = bool(torch.equal(hbufs[(n_e2e - 1) & 1][: ne * RESULT_DTYPE.itemsize], bufs[(n_e2e - 1) & 1][: ne * RESULT_DTYPE.itemsize].cpu()))   # (after the clock)
        # the two copies on their own (nothing else on the device): the blit kernel and the SDMA engine
        nb = ne * RESULT_DTYPE.itemsize
        tt = []
        for _ in range(3):
            t1 = time.perf_counter(); hbufs[0][:nb].copy_(bufs[0][:nb], non_blocking=True); stream.synchronize(); tt.append(time.perf_counter() - t1)
        ts = []
        for _ in range(3):
            t1 = time.perf_counter(); ctx.copy_engine_read(hbufs[0].data_ptr(), bufs[0].data_ptr(), nb); ctx.copy_engine_wait(); ts.append(time.perf_counter() - t1)
        copy_alone_ms = {"bytes": nb, "hipMemcpyAsync_blit_kernel_ms": 1e3 * min(tt), "copy_engine_sdma_ms": 1e3 * min(ts),
                         "GBps": {"blit": nb / min(tt) / 1e9, "sdma": nb / min(ts) / 1e9}}

    if rank == 0:
        solve_ms, pose_ms, cover_ms, heur_ms = [float(x) for x in np.mean(np.array(kernel_events), axis=0)]
        kern_ms = cover_ms                                               # pp_k_cover_sweep (+ pp_k_cover_finish, its lane-per-edge second half), the dominant kernel
        launch_ms = solve_ms + pose_ms + cover_ms + heur_ms
        res = d_res.cpu().numpy().view(RESULT_DTYPE)[: edges // args.steps]
        n_edges_launch = len(res)
        steps_mean = float((res["info"] >> 16).mean())
        feas_frac = float(((res["flags"] & F_INFEASIBLE) == 0).mean())
        M = 0 if w.obst is None else len(w.obst)
        R = len(w.ribbons4)
        # algorithmic bytes / flops per edge with the MEASURED mean step count
        bytes_per_edge = 88 + 32 + 32 * R + 56 * M / 4.0 + 150 + steps_mean / 8.0
        flops_per_edge = 1000 + steps_mean * (70 + 21 * M) + 4e4        # solve + sweeps + heuristic
        cover_edges = ctx.last_cover_edges()
        # "roofline.achieved must be F x edges/s / fp64-vector peak": F x the edges of one costing launch over the
        # launch's kernel time (HIP events, live).  F is the survey's model of the reference's arithmetic, not a counter.
        ach_tf = flops_per_edge * n_edges_launch / (launch_ms * 1e-3) / 1e12
        hbm_model_gbs = bytes_per_edge * n_edges_launch / (launch_ms * 1e-3) / 1e9
        cover_gbs = bytes_per_edge * cover_edges / (kern_ms * 1e-3) / 1e9
        traffic = None
        per_kernel = None
        sweep_hbm = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")   # written from separate rocprofv3 --pmc passes (tools/traffic.sh)
        # the PMC passes ran the default per-GPU workload (config 3: 65 536 attempts per step): their per-launch figures say nothing
        # about a --strong slice or another batch size
        counters_apply = (not args.strong) and args.batch == 65536
        if counters_apply and os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                # the counters describe THIS run only if the kernel sources are still the ones they were measured on
                if tj.get("kernel_sources_sha256") != kernel_sources_sha256():
                    raise ValueError("profiles/traffic.json was measured on other kernel sources")
                traffic = tj.get("hbm_bytes_per_costing_launch", tj.get("hbm_bytes_per_launch"))
                per_kernel = {k: v.get("fetch_size_bytes", 0.0) + v.get("write_size_bytes", 0.0) for k, v in tj["kernels"].items()}
                # "achieved HBM GB/s on the collision sweep against the chip's peak" (BASELINE north_star): counter bytes of the
                # pose sweep (collision checks) over its live kernel time.  Low is good here: the sweep is ALU-bound.
                nbytes = sum(per_kernel.get(k, 0.0) for k in ("pp_k_plan_skips", "pp_k_pose_sweep"))
                sweep_hbm = {"kernel": "pp_k_plan_skips+pp_k_pose_sweep", "pmc_bytes_per_launch": nbytes, "GBps": nbytes / (pose_ms * 1e-3) / 1e9,
                             "frac_of_peak": nbytes / (pose_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            except Exception:
                traffic = per_kernel = sweep_hbm = None
        # counter-anchored utilisation: profiles/valu.json from its own rocprofv3 --pmc pass (tools/valu.sh),
        # stamped with the kernel sources it was measured on
        valu = None
        vp = os.path.join(ROOT, "profiles", "valu.json")
        if counters_apply and os.path.exists(vp):
            try:
                vj = json.load(open(vp))
                if vj.get("kernel_sources_sha256") != kernel_sources_sha256():
                    raise ValueError("profiles/valu.json was measured on other kernel sources")
                valu = vj
            except Exception:
                valu = None
        key = d_key2.cpu().numpy().view(np.uint64)
        alg_bytes_launch = bytes_per_edge * n_edges_launch
        out = {
            "metric": "Dubins edges costed/sec on 2048x2048 grid",
            "value": total_edges / t,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * t / args.steps,
            "ms_per_step_median": float(np.median(step_ms)),
            "ms_per_step_p99": float(np.percentile(step_ms, 99)),
            "ms_per_step_slowest": {"ms": float(step_ms.max()), "index": int(step_ms.argmax())},     # (which timed step, by stream events)
            "e2e_ms_per_step": e2e_ms,
            "e2e_pipelined_ms_per_step": e2e_pipe_ms,
            "e2e_copy_engine_ms_per_step": e2e_sdma_ms,
            "records_copy_alone": copy_alone_ms,
            **({"e2e_copy_engine_error": e2e_sdma_err} if e2e_sdma_err else {"e2e_copy_engine_bytes_verified": e2e_sdma_ok}),
            "e2e_note": f"rank 0, {n_e2e} further steps after the timed region, each followed by the D2H copy of its records "
                        f"({n_edges_launch * RESULT_DTYPE.itemsize / 1e6:.1f} MB) into pinned host memory; pipelined: the copy of step i on a second stream "
                        f"while step i + 1 runs (two record buffers) - hipMemcpyAsync runs as a blit kernel on the CUs here; copy_engine: the same "
                        f"overlap with the copy on an SDMA engine (ppgpu_copy_engine_read); never part of value",
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "rccl_ranks": rccl_ranks,
            **({"rehearsal": "PP_BENCH_REHEARSAL=1: all ranks on device 0, gloo, host combine - NOT a measurement"} if rehearsal else {}),
            "launched_by": ("bench.py itself" if os.environ.get("PP_BENCH_SELF_LAUNCHED") else ("an external launcher" if world > 1 else "single process")),
            "config": {"workload": (f"cfg4_2048_10pct_{total_attempts}_obst{M}_over_{world}gpus" if args.strong else w.name),
                       "grid": "2048x2048 @0.1m, 10% blocked", "samples_per_iter_per_gpu": B, "samples_per_iter_total": total_attempts,
                       "dynamic_obstacles": M, "obstacle_placement": "uniform in the map (SURVEY 8d), seed 3",
                       "ribbons": R, "heuristic": "TspPointRobotNoSplitKRibbons K=2",
                       "edges_per_iter_per_gpu": n_edges_launch,
                       "sharding": "sample batch split by rank; one RCCL collective per iteration (ppgpu_allreduce_best: all-gather of 16 B per rank + local min)"},
            "roofline": {"bound": "fp64_valu", "achieved": ach_tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / FP64_VECTOR_PEAK_TFLOPS,
                         "achieved_is": "SURVEY 8(d): algorithmic flops per edge F (model of the reference's arithmetic with the measured mean step count) "
                                        "x edges of one costing launch / the launch's kernel time, measured live with HIP events between the kernels.  "
                                        "F counts what the REFERENCE does per edge; the kernels skip chunks and cull obstacles exactly, so frac is not "
                                        "hardware utilisation and can exceed 1 - valu_issue_frac (counters) is the utilisation figure",
                         "valu_issue_frac": (valu or {}).get("valu_issue_frac"),
                         "executed_to_model_flops": ((valu["executed_lane_ops_per_launch"] / (flops_per_edge * n_edges_launch)) if valu else None),
                         "valu_is": ("profiles/valu.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU and, in its own pass, GRBM_GUI_ACTIVE on these kernel sources, "
                                     "tools/valu.sh): valu_issue_frac = sum over the costing kernels of SQ_ACTIVE_INST_VALU x 4 (the counter counts quad-cycles) / "
                                     "(1 024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); executed = SQ_INSTS_VALU x 64 lanes, an upper bound of the lanes doing arithmetic"
                                     if valu else ("null: no VALU counter pass on these kernel sources (tools/valu.sh)" if counters_apply else
                                                   "null: the PMC passes measured the default per-GPU workload (config 3), not this one")),
                         "valu_per_kernel": (valu or {}).get("kernels"),
                         "algorithmic_flops_per_edge": flops_per_edge,
                         "kernel": "the costing launch (pp_k_solve_edges .. pp_k_heuristic_lanes); dominant kernel pp_k_cover_sweep (kernel_ms includes pp_k_cover_finish, which ends its edges one lane each)",
                         "launch_ms": launch_ms, "kernel_ms": kern_ms, "edges_in_kernel": cover_edges,
                         "kernels_ms": {"pp_k_solve_edges": solve_ms, "pp_k_plan_skips+pp_k_pose_sweep": pose_ms, "pp_k_cover_sweep+pp_k_cover_finish": cover_ms,
                                        "others": heur_ms},
                         "kernels_note": "HIP events between the kernels of each costing launch of the timed region's last 8 steps (recorded in those steps only: a record is ~7 us of idle device between two kernels); "
                                         "'others' = pp_k_approach_events (finishes the edges whose coverage state machine has nothing to do: "
                                         "edges_in_kernel is what is left for the cover sweep) + pp_k_deferred_list + pp_k_heuristic_lanes (beside it, on a second stream, "
                                         "pp_k_heuristic_listed) + pp_k_heuristic_big; the four add up to the launch",
                         "traffic": traffic,
                         "traffic_is": ("PMC bytes per costing launch from profiles/traffic.json, measured on these kernel sources"
                                        if traffic is not None else ("null: no PMC pass on these kernel sources (tools/traffic.sh)" if counters_apply else
                                                                    "null: the PMC passes measured the default per-GPU workload (config 3), not this one")),
                         "hbm": {"bound": "hbm", "achieved": hbm_model_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_model_gbs / HBM_PEAK_GBS,
                                 "achieved_is": "algorithmic bytes per edge (model, SURVEY 8d) x edges of the launch / launch_ms",
                                 "algorithmic_bytes_per_edge": bytes_per_edge, "algorithmic_bytes_per_launch": alg_bytes_launch,
                                 "counter_bytes_per_launch": traffic,
                                 "counter_to_algorithmic": (traffic / alg_bytes_launch if traffic else None),
                                 "counter_GBps": (traffic / (launch_ms * 1e-3) / 1e9 if traffic else None),
                                 "counter_bytes_per_kernel": per_kernel,
                                 "cover_sweep": {"achieved": cover_gbs, "frac": cover_gbs / HBM_PEAK_GBS,
                                                 "achieved_is": "algorithmic bytes of the edges pp_k_cover_sweep visited / its own time"},
                                 "collision_sweep": sweep_hbm,
                                 "note": "reported because BASELINE's north_star asks for it; the path is fp64-VALU bound (SURVEY 8d: 630 flop/B "
                                         "against a machine balance of 10), so for HBM lower is better"}},
            "workload_stats": {"mean_sweep_steps_per_edge": steps_mean, "feasible_fraction": feas_frac,
                               "kernel_edges_per_s": n_edges_launch / (launch_ms * 1e-3),
                               "best_f": float(np.array([key[0]], dtype=np.uint64).view(np.float64)[0]), "best_edge": int(key[1])},
        }
        profiled = os.environ.get("PP_BENCH_PROFILED") == "1"     # under rocprofv3 (tools/*.sh): only the headline launches
        if args.as_rank is not None:
            out["as_rank"] = {"rank": emu_rank, "of": emu_world, "attempts_skipped_per_step": sharding.shard_attempts(total_attempts, emu_rank, emu_world)[0],
                              "note": "ONE GPU playing rank R of N: its slice of the batch and the sampler skip over the lower ranks' slices, no "
                                      "communicator, no collective.  A projection of that rank's step time, not a measured scaling curve"}
        if world == 1 and not args.no_cpu_baseline and args.as_rank is None:
            out.update(cpu_baseline_and_parity(ctx, w, res))
            out.update(first_goal_check())
        if world == 1 and not profiled and args.as_rank is None:
            out.update(open_vertex_run(ctx, w, torch, dev))
        if world == 1 and not profiled and not args.no_plan_level and args.as_rank is None:
            out.update(plan_level(args.replan_cycles, cpu_baseline=not args.no_cpu_baseline))
        print(json.dumps(out), flush=True)
    if world > 1 or use_comm:
        dist.barrier()
        torch.cuda.synchronize(dev)
        if use_comm:
            ctx.comm_destroy()
        dist.destroy_process_group()


# the device sources (path_planner_amd/csrc): what a PMC measurement is stamped with
KERNEL_SOURCES = ("ppgpu.hip", "pp_kernels.h", "pp_k_common.h", "pp_k_solve.h", "pp_k_sweep.h", "pp_k_cover.h", "pp_k_heuristic.h", "pp_k_expand.h",
                  "pp_k_incumbent.h", "pp_device.h", "pp_sampler.h", "pp_cr.h", "pp_cr_tables.h")


def kernel_sources_sha256():
    """Identity of the kernel sources a PMC measurement belongs to (profiles/traffic.json and profiles/valu.json carry the same stamp)."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "path_planner_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def open_vertex_run(ctx, w, torch, dev):
    """SURVEY 8(d) config 3, second half: 64 open vertices (the root and 63 of its first-generation children, each with its own
    time grid, ribbon list and coverage state) x 4 096 samples x 4 configurations in one launch.  Reported beside the headline
    figure, never part of it."""
    import numpy as np
    from path_planner_amd.types import RESULT_DTYPE, VERTEX_DTYPE, F_INFEASIBLE, F_GOAL, edge_pack
    ns, stride = 4096, 12
    res, child = ctx.cost_edges_host(edge_pack(np.zeros(4096, dtype=np.uint64), np.arange(4096) // 4, np.arange(4096) % 4), stride=stride)
    ok = np.nonzero(((res["flags"] & (F_INFEASIBLE | F_GOAL)) == 0) & (((res["info"] >> 8) & 0xFF) <= stride))[0][:63]
    v = np.zeros(len(ok) + 1, dtype=VERTEX_DTYPE)
    pool = [np.asarray(w.ribbons4, dtype=np.float64).reshape(-1, 4)]
    v[0] = w.root()[0]
    off = len(pool[0])
    for k, e in enumerate(ok):
        r, nr = res[e], int((res[e]["info"] >> 8) & 0xFF)
        v[k + 1] = (r["end_x"], r["end_y"], r["end_heading"], r["end_speed"], r["end_time"], r["g"], r["coverage_completed_time"], off, nr)
        pool.append(child[e, :nr]); off += nr
    ctx.set_vertices(v, np.concatenate(pool))
    ne = len(v) * ns * 4
    d = torch.zeros(ne * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ctx.cost_edges_dense(0, len(v), 0, ns, 0xF, d.data_ptr()); b.record()
        torch.cuda.synchronize(dev)
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts[1:]))
    r = d.cpu().numpy().view(RESULT_DTYPE)
    ctx.set_vertices(w.root(), w.ribbons4)
    return {"open_vertex_run": {"open_vertices": int(len(v)), "samples": ns, "edges": int(ne), "ms": ms, "edges_per_s": ne / (ms * 1e-3),
                                "mean_sweep_steps_per_edge": float((r["info"] >> 16).mean()),
                                "feasible_fraction": float(((r["flags"] & F_INFEASIBLE) == 0).mean())}}


def cpu_baseline_and_parity(ctx, w, gpu_res):
    """Time the CPU oracle (restatement of the reference path, kind "port") on a bounded sample of the same
    edge list, on this box's host cores, and gate the run on parity for those edges."""
    import numpy as np
    import oracle as orc
    from parity import compare_results
    from path_planner_amd.types import edge_pack

    samples = ctx.get_samples()
    world = orc.World(w.cfg, w.grid, w.res, w.obst)
    # the GPU box gives one GPU a 16-core CPU share; never oversubscribe it
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    n_multi = min(len(samples), 16384)      # 65 536 edges over all cores: ~1.5 s wall, ~22 s of CPU work
    n_single = min(len(samples), 2048)      # 8 192 edges on one core: ~2.5 s
    def edges_for(n):
        ne = 4 * n
        return edge_pack(np.zeros(ne, dtype=np.uint64), np.repeat(np.arange(n), 4), np.tile(np.arange(4), n))
    t0 = time.perf_counter()
    cpu1 = world.cost_edges(w.root(), w.ribbons4, samples[:, 0], samples[:, 1], samples[:, 2], edges_for(n_single), threads=1)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    cpu = world.cost_edges(w.root(), w.ribbons4, samples[:, 0], samples[:, 1], samples[:, 2], edges_for(n_multi), threads=cores)
    tm = time.perf_counter() - t0
    rep = compare_results(gpu_res[: 4 * n_multi], cpu)
    # the reference's catkin build sets no optimisation level (pp/CMakeLists.txt:4-6): the same port built -O0, one thread
    o0 = None
    try:
        import subprocess
        lib0 = os.path.join(ROOT, "oracle", "libpp_oracle_O0.so")
        if not os.path.exists(lib0):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "O0"])
        n0 = min(len(samples), 512)
        code = ("import sys,time,os,json;sys.path.insert(0,%r);sys.path.insert(0,%r);os.environ['PP_ORACLE_SO']=%r;"
                "import numpy as np,oracle as orc;from path_planner_amd import workloads;from path_planner_amd.types import edge_pack;"
                "w=workloads.config3();world=orc.World(w.cfg,w.grid,w.res,w.obst);s=np.load(sys.argv[1]);n=%d;ne=4*n;"
                "e=edge_pack(np.zeros(ne,dtype=np.uint64),np.repeat(np.arange(n),4),np.tile(np.arange(4),n));t=time.perf_counter();"
                "world.cost_edges(w.root(),w.ribbons4,s[:,0],s[:,1],s[:,2],e,threads=1);print(json.dumps(ne/(time.perf_counter()-t)))"
                % (ROOT, os.path.join(ROOT, "tests"), lib0, n0))
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".npy") as f:
            np.save(f.name, samples)
            o0 = float(subprocess.check_output([sys.executable, "-c", code, f.name], timeout=120).decode().strip().splitlines()[-1])
    except Exception:
        o0 = None
    return {
        "cpu_baseline": {"value": 4 * n_multi / tm, "unit": "edges/s", "cores": cores, "kind": "port",
                         "sample": f"first {4 * n_multi} edges of the same launch, static chunks over {cores} threads; "
                                   f"single thread on the first {4 * n_single} edges: {4 * n_single / t1:.1f} edges/s",
                         "single_thread_value": 4 * n_single / t1,
                         "single_thread_value_O0": o0,
                         "note": "kind 'port': the oracle's restatement built -O2 (the figures above); single_thread_value_O0 is the same code "
                                 "built -O0, the reference's default build type, on the first 2048 edges"},
        "parity": {"ok": rep["ok"], "edges_checked": rep["n"], "flags_equal": rep["flags_equal"], "worst_rel": rep["worst_rel"]},
    }


def plan_level(cycles, cpu_baseline=True):
    """SURVEY 8(d) config 5 and the plan()-level view of the path: the 10 Hz anytime replan loop through
    the C++ host planner (path_planner_amd/host/plan_cli: GpuAStarPlanner::plan behind the reference's Planner::plan seam, every edge
    costed on the GPU) — config-3 grid, 32 moving obstacles uniform in the map, 100 ms budget per cycle, 8 192 initial samples doubling,
    the start advanced 0.1 s along the returned plan, the plan handed back as previousPlan — and, as the CPU baseline of the SAME seam,
    the oracle's restatement of AStarPlanner::plan given the same wall budget on one host core (the reference plans on one thread).
    After the timed region, never part of `value`."""
    import subprocess
    import tempfile
    try:
        from path_planner_amd import workloads
        from test_gpu_host_planner import CLI, _scenario, _write_map
        if not os.path.exists(CLI):
            return {"replan": None, "plan_level": None}
        w = workloads.config3()
        w.obst = workloads.obstacles(32, 3, 204.8, time=float(w.start5[4]))
        budget_s, init = 0.1, 8192
        with tempfile.TemporaryDirectory() as d:
            mp = os.path.join(d, "grid.map")
            _write_map(w.grid, w.res, mp)
            sc = os.path.join(d, "s.txt")
            _scenario(w, sc, mp, float(w.start5[4]), 1e-3, 1, init, devices=[0, 0])     # two contexts = two HIP streams on the one GPU
            with open(sc, "a") as f:
                f.write(f"time_remaining {budget_s!r}\nreplan {cycles} 0.1\n")
            run = subprocess.run([CLI, sc], capture_output=True, text=True, timeout=600)
        if run.returncode != 0:
            raise RuntimeError(run.stdout[-400:] + run.stderr[-400:])
        r = json.loads(run.stdout.strip().splitlines()[-1])
        replan = {"workload": "cfg5: cfg3 grid (2048x2048 @0.1 m, 10 % blocked), 5 ribbons, 32 moving obstacles uniform in the map, moving start, "
                              "previous plan handed back; ONE GPU, two device contexts (streams) so that two round trips are in flight "
                              "(BASELINE's 8-GPU form is not the builder's to run)",
                  "budget_ms": 1e3 * budget_s, "initial_samples": init, "cycles": cycles,
                  "plan_latency_ms": {"p50": r["wall_ms_p50"], "p99": r["wall_ms_p99"], "max": r["wall_ms_max"], "first_cycle": r["first_cycle_ms"]},
                  "late_cycles": r["late_cycles"], "iterations_per_cycle": r["mean_iterations"],
                  "first_goal_iteration": {"median": r["first_goal_iteration_median"], "max": r["first_goal_iteration_max"], "mean": r["first_goal_iteration_mean"],
                                           "cycles_with_a_goal": r["cycles_with_a_goal"]},
                  "expansions_per_cycle": r["mean_expanded"], "edges_per_cycle": r["mean_edges"], "round_trips_per_cycle": r["mean_round_trips"],
                  "samples_reached": r["mean_samples"], "deadline_stops": r["deadline_stops"], "failed_plans": r["failed_plans"],
                  "failed_plans_with_start_in_collision": r["failed_plans_with_start_in_collision"], "grid_uploads": r["grid_uploads"],
                  "node_regrowths": r["node_regrowths"], "device_growths": r["device_growths"], "worst_cycle": r["worst_cycle"]}
        plan = {"seam": "Planner::plan (pp/src/planner/Planner.h:50-51): GpuAStarPlanner, one GPU",
                "expansions_per_s": r["expansions_per_s_inside_plan"], "edges_per_s": r["edges_per_s_inside_plan"],
                "note": "inside plan() the search is sequential: each round trip costs the <= 40 edges of each of up to 64 open vertices (two round trips in flight), so the plan-level "
                        "edge rate is bound by round-trip latency, two orders of magnitude below the batch rate (`value`)"}
        if cpu_baseline:
            import numpy as np
            import oracle as orc
            orc.O.ppo_set_ribbon_width(w.cfg.ribbon_width)
            world = orc.World(w.cfg, w.grid, w.res, w.obst)
            exp, edges, its, fg = [], [], [], []
            n_cpu = 10
            for _ in range(n_cpu):                 # clock_dt = -1: the wall clock, the same budget as the product's cycles
                rc, st, _plan, _itf, _ = world.plan(w.ribbons4, w.start5, budget_s, float(w.start5[4]), -1.0, initial_samples=init, dump_edges=1)
                exp.append(int(st.expanded)); edges.append(int(world.last_plan_edges)); its.append(int(st.iterations)); fg.append(int(st.first_goal_iteration))
            plan["cpu_baseline"] = {"kind": "port", "cores": 1, "budget_ms": 1e3 * budget_s,
                                    "sample": f"{n_cpu} plan() calls of the oracle's AStarPlanner restatement (-O2) from the loop's first start state, no previous plan, wall clock",
                                    "expansions_per_s": float(np.mean(exp)) / budget_s, "edges_per_s": float(np.mean(edges)) / budget_s,
                                    "iterations_per_call": float(np.mean(its)), "first_goal_iteration_median": int(np.median(fg))}
        return {"replan": replan, "plan_level": plan}
    except Exception as e:       # the throughput line must not depend on this leg
        return {"replan": {"error": repr(e)}, "plan_level": None}


def first_goal_check():
    """BASELINE.json's second metric, iterations-to-first-goal with a fixed seed: one whole plan() of the C++ host planner
    (path_planner_amd/host/plan_cli, every edge costed on the GPU) and of the CPU oracle's planner on config 2, both driven by
    the same injected clock.  Not timed; reported next to the throughput line.  Skipped (null) if the CLI is not built."""
    import tempfile
    import oracle as orc
    from path_planner_amd import workloads
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        from test_gpu_host_planner import CLI, _run_cli, _scenario, _write_map
        if not os.path.exists(CLI):
            return {"first_goal": None}
        # config 2's world with fine clutter (tools/find_late_goal.py): the first goal only appears after several doublings of the
        # sample set, so the index is not trivially 0
        from test_gpu_host_planner import LATE_GOALS, _late_goal_workload
        name, frac, gseed, blob, init, _ = LATE_GOALS[0]
        w2 = _late_goal_workload(name, frac, gseed, blob)
        orc.O.ppo_set_ribbon_width(w2.cfg.ribbon_width)
        world = orc.World(w2.cfg, w2.grid, w2.res, w2.obst)
        t0, dt, calls = 1000.0, 1e-3, 400
        with tempfile.TemporaryDirectory() as d:
            mp = os.path.join(d, "grid.map")
            _write_map(w2.grid, w2.res, mp)
            sc = os.path.join(d, "s.txt")
            _scenario(w2, sc, mp, t0, dt, calls, init)
            host = _run_cli(sc)
        rc, st, plan, _, _ = world.plan(w2.ribbons4, w2.start5, calls * dt, t0, dt, initial_samples=init)
        return {"first_goal": {"workload": w2.name + f" with {int(100 * frac)} % clutter (grid seed {gseed}, {blob}-cell blobs), {init} initial samples",
                               "seed": "fixed by the injected clock", "iteration_gpu": host["first_goal_iteration"],
                               "iteration_cpu_oracle": int(st.first_goal_iteration), "plan_f_gpu": host["plan_f"], "plan_f_cpu_oracle": float(st.plan_f),
                               "identical_index": host["first_goal_iteration"] == int(st.first_goal_iteration),
                               "plan_f_rel_diff": abs(host["plan_f"] - float(st.plan_f)) / max(1.0, abs(float(st.plan_f)))}}
    except Exception as e:   # the throughput line must not depend on this leg
        return {"first_goal": {"error": repr(e)}}


if __name__ == "__main__":
    main()
