#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched truss FEM environment step on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1: either the driver launches one rank per GPU with torch.distributed.run, or -- when WORLD_SIZE is not
  set -- this script starts the N rank processes itself (children, before anything touches the GPU) and rank 0
  prints the line.  RCCL carries only the barrier and the max-over-ranks of the elapsed time: the env batch
  shards with no data-path collective.

A "step" is ONE pass of the hot path over the whole resident batch: one `_game_modify`-equivalent per env
(action decode -> design update -> FP64 assembly + solve -> member stresses -> point).  Workload = BASELINE.json's
metric config: synthetic random-geometry trusses, 32 nodes / 80 elements / 60 DOF, FEM-only (no agent).  Inputs
(design state, per-env constants, a pool of pre-drawn actions) are resident in HBM before the timed region starts.

`value` is the path an AGENT CAN DRIVE: one launch of truss_step_kernel per step (`truss_step`; the K launches of a
timed block are issued back to back by the C entry `truss_rollout` in its one-launch-per-step mode, exactly what a
replayed hipGraph of K `truss_step` calls does) -- the reference never chains steps with actions known up front
(master_DDPG_truss2D_MO.py:249-260).  The persistent K-steps-in-one-launch kernel (`truss_rollout_kernel`) is reported
next to it as `persistent_rollout`, the state-emitting step (`TRUSS_F_EMIT_OBS`, what configs[2]-[4] execute) as
`state_emitting_step`, and a time-boxed `configs` object carries BASELINE configs[2]-[4] (bench_configs.py).

  --scaling weak    (default) 4096 envs PER GPU ("at 4096 envs" is the kernel's design point: one wave per SIMD)
  --scaling strong  --global-envs (default 4096; BASELINE configs[3] is 8192) split contiguously over the ranks
With N > 1 the weak line also carries a "strong" object (4096 and 8192 global envs) measured by the same ranks.

The K-step block is timed BLOCKS times (each bracketed by barrier + synchronize); `value` / `ms_per_step` are the
median block, min / max are reported next to them.  Rank 0 prints ONE JSON line; DESIGN.md "Measurement".
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mop-truss-marl_amd"))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
NUM_X = 16          # 32 nodes
N_EXTRA = 4         # 76 reference elements + 4 long braces = 80
N_ACTION_SETS = 8
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BLOCKS = 5


def algorithmic_bytes_per_env_step(N, E):
    """SURVEY.md §8d B_core: 4-byte words. reads y,x (2N) + sec (E) + actions (5N) + coin (1);
    writes y' (N) + sec' (E) + d (2N) + q0 (E) + sr (E) + point (4)  = 4*(10N + 4E + 5)."""
    return 4 * (10 * N + 4 * E + 5)


def obs_bytes_reference(N, E):
    """SURVEY.md §8d B_obs - B_core: the reference's 13N + 5N^2 + 12N + 21E floats per env."""
    return 4 * (13 * N + 5 * N * N + 12 * N + 21 * E)


def obs_bytes_written(N, E):
    """what this build writes per env: A_n and mask are topology-static and are not re-emitted (3 N^2, not 5 N^2)."""
    return 4 * (13 * N + 3 * N * N + 12 * N + 21 * E)


# ---- CPU baseline (oracle, test infrastructure: the checker / baseline, never the measured path) --------------
def _cpu_worker(job):
    """One process = one shard of envs stepped by the numpy oracle; returns (env-steps, seconds)."""
    seed, B, n_steps = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import truss_oracle as O
    import numpy as np
    import parity_common as pc
    from truss_mi355 import synthetic
    topo = synthetic.bench_topology(NUM_X, N_EXTRA)
    ot = pc.oracle_topology(topo)
    batch = synthetic.random_batch(topo, B, seed)
    load = pc.oracle_load(ot, batch)
    int_obj = O.initial_objectives(ot, batch["x"], batch["y"], batch["sec"], batch["target"])
    ag, at = synthetic.random_actions(2, B, topo.N, seed + 1)
    y, sec = batch["y"], batch["sec"]
    t0 = time.perf_counter()
    for n in range(n_steps):
        o = O.env_step(ot, batch["x"], y, sec, None, None, ag[n % 2], at[n % 2], np.zeros(B), batch["target"], load,
                       batch["y_max"], batch["d_min"], batch["max_def"], batch["is_roof"], int_obj)
        y, sec = o["y"], o["sec"]
    return B * n_steps, time.perf_counter() - t0


def cpu_baseline():
    """The numpy oracle on this box's host cores, B = 4096 envs sharded over a process pool (one process per core
    of the box's CPU share), plus the single-env latency.  Runs BEFORE this process touches the GPU (fork)."""
    import multiprocessing as mp
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))      # a 1-GPU box shares its host: 16 cores are ours
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(k, "1")   # one core per worker: the pool is the parallelism
    shard = 4096 // cores
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(100 + i, 8, 1) for i in range(cores)])           # imports, first touch
        n_steps = 64       # ~1 s of wall time on 16 cores = ~15 s of CPU work
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, [(1234 + i, shard, n_steps) for i in range(cores)])
        wall = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    n1, t1 = _cpu_worker((7, 1, 50))
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": done / wall, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_steps} steps x {shard * cores} envs (32 nodes / 80 elements, same generator) on {cores} processes x 1 thread, "
                      f"numpy oracle; wall time of the pool incl. its slowest worker",
            "single_env_ms_per_step": t1 / n1 * 1e3, "cpu_model": model, "host_cores_visible": avail,
            "reference_note": reference_note()}


def reference_note():
    """The reference's own CPU path, timed in the build container by tools/time_reference.py (it does not travel to the GPU box)."""
    try:
        r = json.load(open(os.path.join(ROOT, "profiles", "r3", "reference_cpu.json")))
        c = r["cases"]["large_bridge"]
        return {"source": "profiles/r3/reference_cpu.json (tools/time_reference.py, build container, 1 core, median of 5 x 100 calls)",
                "cpu_model": r["cpu_model"], "truss": f"{c['nodes']} nodes / {c['elements']} elements (the reference's own large truss)",
                "_game_modify_ms": c["_game_modify"]["ms_per_call_median"], "gen_all_ms": c["gen_all"]["ms_per_call_median"],
                "env_steps_per_s_per_core": c["_game_modify"]["calls_per_s_per_core"]}
    except (OSError, KeyError, ValueError):
        return "the reference's own _game_modify: BASELINE.md (build container, 1 core)"


# ---- launching ---------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """Start the N rank processes as children of a parent that never touches the GPU; returns the worst exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def measure(env, G, T, steps, warmup, dist, distributed):
    """BLOCKS timed K-step blocks -> (median, min, max wall seconds (each max-over-ranks), device ms of the median block)."""
    walls, devs = [], []
    for b in range(BLOCKS):
        el, dev_ms = distributed.timed_rollout(env, G, T, steps, warmup if b == 0 else 0, dist)
        walls.append(el)
        devs.append(dev_ms)
    order = sorted(range(BLOCKS), key=lambda i: walls[i])
    med = order[BLOCKS // 2]
    return walls[med], min(walls), max(walls), devs[med]


def committed_traffic(root, envs, N, E):
    """HBM bytes per launch / per step from the committed rocprofv3 PMC summary (profiles/r3/pmc_traffic.json), if it was taken on
    this kernel source: {"step": .., "rollout": .., "fused": ..} or {}."""
    p = os.path.join(root, "profiles", "r3", "pmc_traffic.json")
    try:
        rec = json.load(open(p))
        src = open(os.path.join(root, "mop-truss-marl_amd", "csrc", "truss_body.h"), "rb").read()
        if rec.get("envs") == envs and rec.get("nodes") == N and rec.get("elements") == E and \
                rec.get("truss_body_sha16") == hashlib.sha256(src).hexdigest()[:16]:
            return {"step": rec.get("step_kernel_bytes_per_launch"), "rollout": rec.get("rollout_kernel_bytes_per_step"),
                    "fused": rec.get("fused_step_kernel_bytes_per_launch")}
    except (OSError, ValueError):
        pass
    return {}


def run_configs(args, tm, lib, dev, dist, rank, world, budget_s):
    """BASELINE configs[2]-[4] (bench_configs.py), time-boxed: an entry that would start after the budget is spent is recorded as
    skipped; an entry that raises is recorded with its error -- the headline line is printed in every case.  The single-GPU
    entries run when there is one rank; large_bridge runs on the ranks present (its env batch shards, the MADDPG gradients are
    all-reduced over RCCL)."""
    import bench_configs as BC
    out, t0 = {}, time.perf_counter()

    def entry(name, fn, collective=False):
        left = budget_s - (time.perf_counter() - t0)
        go = left > 0
        if collective and dist is not None:      # all ranks take the same decision
            import torch
            f = torch.tensor([1 if go else 0], device=dev)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            go = bool(f.item())
        if not go:
            out[name] = {"skipped": f"time box of {budget_s:.0f} s spent"}
            return
        try:
            r = fn()
            out[name] = {k: v for k, v in r.items() if k not in ("profile_s", "config")}
            out[name]["workload"] = r.get("config")
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}

    if world == 1:
        entry("marl_small_roof_4096", lambda: BC.marl_small_roof(4096, 4, True, dev=dev))
        entry("marl_small_roof_4096_no_training", lambda: BC.marl_small_roof(4096, 4, False, dev=dev))
        entry("mixed_pool_step", lambda: BC.mixed_pool_step(lib, dev))
        entry("mixed_pool_marl_no_training", lambda: BC.mixed_marl(train=False, dev=dev))
        entry("mixed_pool_marl", lambda: BC.mixed_marl(train=True, dev=dev))
    entry("large_bridge_8192", lambda: BC.large_bridge(8192, 3, True, dev=dev, dist=dist, rank=rank, world=world), collective=True)
    out["seconds"] = time.perf_counter() - t0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU, help="envs per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--global-envs", type=int, default=4096, help="total envs, split over the ranks (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the time-boxed BASELINE configs[2]-[4] object")
    ap.add_argument("--configs-budget", type=float, default=90.0, help="seconds the configs object may take")
    ap.add_argument("--extras", action="store_true", help="also time the E=76 / small / large topologies (bench_extras.py)")
    ap.add_argument("--lib", default=None, help="diagnostic: another HIP build of the same ABI (default: the product library)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))      # the parent has not imported torch: nothing here touched a GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline()                                 # forks: before the GPU is initialised

    import torch
    backend = os.environ.get("TRUSS_BENCH_BACKEND", "nccl")  # "gloo" + the lane emulator: CPU rehearsal in the test-suite
    on_gpu = backend == "nccl"
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs the MI355X"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    dist = None
    if world > 1 or os.environ.get("TRUSS_BENCH_FORCE_DIST"):     # the env switch rehearses the RCCL plumbing on one GPU
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if on_gpu:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import truss_mi355 as tm
    from truss_mi355 import synthetic, distributed
    lib = tm.load(args.lib)              # raises if the HIP extension is missing
    assert lib.backend == ("hip" if on_gpu else "emu"), "bench.py measures the HIP path only"
    topo = synthetic.bench_topology(NUM_X, N_EXTRA)

    def shard_of(global_envs):
        lo, hi = distributed.shard_bounds(global_envs, world, rank)
        return hi - lo

    if args.scaling == "weak":
        B, global_envs = args.envs, args.envs * world
    else:
        global_envs = args.global_envs
        B = shard_of(global_envs)
    env, G, T, _ = distributed.make_rank_env(topo, B, rank, device=dev, lib=lib, seed=1234, n_action_sets=N_ACTION_SETS)
    # headline: ONE LAUNCH PER STEP (the path an agent can drive); every rank, barrier + max over ranks
    os.environ["TRUSS_ROLLOUT_LAUNCHES"] = "1"
    elapsed, el_min, el_max, dev_ms = measure(env, G, T, args.steps, args.warmup, dist, distributed)
    del os.environ["TRUSS_ROLLOUT_LAUNCHES"]
    st = int(env.status.sum().item())
    # the same K chained steps as ONE persistent launch (actions known up front: a bench / population-optimiser path)
    persistent = None
    if bool(env.persistent_rollout):
        w, wmin, wmax, d_ms = measure(env, G, T, args.steps, args.warmup, dist, distributed)
        persistent = {"env_steps_per_s": global_envs * args.steps / w, "ms_per_step": w / args.steps * 1e3,
                      "ms_per_step_min": wmin / args.steps * 1e3, "ms_per_step_max": wmax / args.steps * 1e3,
                      "kernel_us": (d_ms * 1e3 / args.steps) if d_ms is not None else None, "steps_per_launch": args.steps,
                      "kernel": "truss_rollout_kernel: the K chained steps of a block as ONE launch (actions known up front; "
                                "no caller in the agent loop)"}

    strong = None
    if world > 1 and args.scaling == "weak":
        strong = {}
        for ge in [int(v) for v in os.environ.get("TRUSS_BENCH_STRONG", "4096,8192").split(",")]:   # (the CPU rehearsal shrinks them)
            b = shard_of(ge)
            e2, G2, T2, _ = distributed.make_rank_env(topo, b, rank, device=dev, lib=lib, seed=4321, n_action_sets=N_ACTION_SETS)
            w, wmin, wmax, _ = measure(e2, G2, T2, args.steps, args.warmup, dist, distributed)
            strong[f"global_envs_{ge}"] = {"env_steps_per_s": ge * args.steps / w, "ms_per_step": w / args.steps * 1e3,
                                           "ms_per_step_min": wmin / args.steps * 1e3, "ms_per_step_max": wmax / args.steps * 1e3,
                                           "envs_per_gpu": b}
            del e2, G2, T2

    # the state-emitting step (BASELINE configs 3-5 need the observation tensors every step): TRUSS_F_EMIT_OBS
    state = None
    if rank == 0 and on_gpu:
        ag0, at0 = G[0].contiguous(), T[0].contiguous()
        ob = env.obs_buffers()
        for _ in range(20):
            env.step(ag0, at0, obs=ob)
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = []
        for _ in range(BLOCKS):
            a0.record()
            for _ in range(100):
                env.step(ag0, at0, obs=ob)
            a1.record()
            torch.cuda.synchronize()
            reps.append(a0.elapsed_time(a1) * 1e3 / 100)
        reps.sort()
        so_us = reps[BLOCKS // 2]
        b_core = algorithmic_bytes_per_env_step(topo.N, topo.E)
        b_ref, b_wr = b_core + obs_bytes_reference(topo.N, topo.E), b_core + obs_bytes_written(topo.N, topo.E)
        state = {"us_per_step": so_us, "us_min": reps[0], "us_max": reps[-1], "one_launch": bool(env.fused_obs),
                 "env_steps_per_s": B / (so_us * 1e-6), "bytes_per_env_B_obs": b_ref, "bytes_per_env_written": b_wr,
                 "frac_of_hbm_peak_B_obs": B * b_ref / (so_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                 "frac_of_hbm_peak_bytes_moved": B * b_wr / (so_us * 1e-6) / 1e9 / HBM_PEAK_GBS}

    extras = {}
    if args.extras and rank == 0 and on_gpu:
        import bench_extras
        extras = bench_extras.run(tm, synthetic, lib, dev, topo, env, G, T, args, B)

    if rank == 0:
        per_step_bytes = algorithmic_bytes_per_env_step(topo.N, topo.E) * B
        kern_s = (dev_ms * 1e-3 if dev_ms is not None else elapsed) / args.steps
        achieved = per_step_bytes / kern_s / 1e9
        info = topo.solver_info(lib)
        traffic = committed_traffic(ROOT, B, topo.N, topo.E)
        if persistent is not None and persistent["kernel_us"]:
            ks = persistent["kernel_us"] * 1e-6
            persistent["roofline"] = {
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                # bytes the kernel really moves per step (PMC; it keeps the design state in LDS and never re-reads it, so this
                # is LESS than the algorithmic B_core) -- null without a committed profile of this kernel source
                "traffic": traffic.get("rollout"),
                "achieved_counter_bytes": (traffic["rollout"] / ks / 1e9) if traffic.get("rollout") else None,
                "frac_counter_bytes": (traffic["rollout"] / ks / 1e9 / HBM_PEAK_GBS) if traffic.get("rollout") else None,
                "achieved_B_core": per_step_bytes / ks / 1e9, "frac_B_core": per_step_bytes / ks / 1e9 / HBM_PEAK_GBS}
        if state is not None:
            state["traffic"] = traffic.get("fused")
        out = {
            "metric": "env steps/sec (batched FEM solves) at 4096 envs",
            "value": global_envs * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_min": el_min / args.steps * 1e3,
            "ms_per_step_max": el_max / args.steps * 1e3,
            "timed_blocks": BLOCKS,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic random-geometry 2-row trusses, 32 nodes / 80 elements / 60 DOF, "
                            "FEM-only env.step (action decode + FP64 assembly/solve + stresses + point), no agent",
                "path": "step: one launch of truss_step_kernel per env step (truss_step), K launches per timed block",
                "envs_per_gpu": B, "global_envs": global_envs, "nodes": topo.N, "elements": topo.E,
                "ndof": int(env.ndof), "half_bandwidth": info["half_bandwidth"],
                "lanes_per_env": info["lanes_per_env"], "rows_per_lane": info["rows_per_lane"],
                "parallelism": f"env-batch sharded over {world} GPU(s), no data-path collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # HBM bytes per launch: rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes) of
                # THIS kernel source, from profiles/r3/pmc_traffic.json; null when the committed profile is of another build
                "traffic": traffic.get("step"),
                "kernel": "truss_step_kernel", "kernel_us": kern_s * 1e6, "steps_per_launch": 1, "launch_us": kern_s * 1e6,
                "bytes_per_launch": per_step_bytes,
                # the other two ceilings SURVEY §8d asks for (algorithmic banded flop count, 1e4 per env-step)
                "fp64_gflops": 1.0e4 * B / kern_s / 1e9, "fp64_frac_of_78.6_TFLOPs": 1.0e4 * B / kern_s / 78.6e12,
            },
            "persistent_rollout": persistent,
            "state_emitting_step": state,
            "nonpositive_pivots": st,
        }
        if strong:
            out["strong"] = strong
        if extras:
            out["extras"] = extras
        if cpu is not None:
            out["cpu_baseline"] = cpu
    else:
        out = None

    # BASELINE configs[2]-[4], after the headline is complete: the line is printed in every case.  A watchdog on every rank ends the
    # process cleanly if the block does not come back (a collective that never completes cannot be interrupted from Python): rank 0
    # then prints the headline with the reason in `configs`.
    if on_gpu and not args.no_configs:
        import threading

        def give_up():
            if rank == 0:
                out["configs"] = {"error": f"did not finish within {args.configs_budget + 150:.0f} s; headline measurements above are complete"}
                print(json.dumps(out), flush=True)
            os._exit(0)

        guard = threading.Timer(args.configs_budget + 150 + (0 if rank == 0 else 5), give_up)
        guard.daemon = True
        guard.start()
        configs = run_configs(args, tm, lib, dev, dist, rank, world, args.configs_budget)
        guard.cancel()
        if rank == 0:
            out["configs"] = configs
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
