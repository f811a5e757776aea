#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched truss FEM environment step on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py
  (one rank per GPU, RCCL only for the barrier and the max-over-ranks of the elapsed time: the env
  batch shards with no data-path collective).

A "step" is ONE pass of the hot path over the whole resident batch: one `_game_modify`-equivalent
per env (action decode -> design update -> FP64 assembly + solve -> member stresses -> point),
i.e. one launch of truss_step_kernel over ENVS_PER_GPU envs.  Workload = BASELINE.json's metric
config: synthetic random-geometry trusses, 32 nodes / 80 elements / 60 DOF, 4096 envs per GPU,
FEM-only (no agent, no observation tensors).  Inputs (design state, per-env constants, a pool of
pre-drawn actions) are resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how each field is derived.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mop-truss-marl_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

ENVS_PER_GPU = 4096
NUM_X = 16          # 32 nodes
N_EXTRA = 4         # 76 reference elements + 4 long braces = 80
N_ACTION_SETS = 8
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_env_step(N, E):
    """SURVEY.md §8d B_core: 4-byte words. reads y,x (2N) + sec (E) + actions (5N) + coin (1);
    writes y' (N) + sec' (E) + d (2N) + q0 (E) + sr (E) + point (4)  = 4*(10N + 4E + 5)."""
    return 4 * (10 * N + 4 * E + 5)


def cpu_baseline(topo, seed, budget_s=20.0):
    """The oracle (numpy restatement of the reference algorithm, oracle/truss_oracle.py) timed on this
    box's host cores on a bounded sample of the same workload.  Test infrastructure used as the
    checker/baseline only -- never on the measured GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import truss_oracle as O
    import parity_common as pc
    from truss_mi355 import synthetic
    B = 1024
    ot = pc.oracle_topology(topo)
    batch = synthetic.random_batch(topo, B, seed)
    load = pc.oracle_load(ot, batch)
    int_obj = O.initial_objectives(ot, batch["x"], batch["y"], batch["sec"], batch["target"])
    ag, at = synthetic.random_actions(2, B, topo.N, seed + 1)
    y, sec = batch["y"], batch["sec"]
    n, t0 = 0, time.perf_counter()
    while True:
        o = O.env_step(ot, batch["x"], y, sec, None, None, ag[n % 2], at[n % 2], np.zeros(B), batch["target"], load,
                       batch["y_max"], batch["d_min"], batch["max_def"], batch["is_roof"], int_obj)
        y, sec = o["y"], o["sec"]
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 64:
            break
    try:
        threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or 1
    except ValueError:
        threads = 1
    return {"value": B * n / el, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} steps x {B} envs of the same 32-node/80-element workload, numpy oracle, "
                      f"single process ({os.cpu_count()} host cores visible)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also time the E=76 topology and the obs kernel")
    ap.add_argument("--lib", default=None, help="diagnostic: another HIP build of the same ABI (default: the product library)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs the MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("TRUSS_BENCH_FORCE_DIST"):     # the env switch rehearses the RCCL plumbing on one GPU
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import truss_mi355 as tm
    from truss_mi355 import synthetic, distributed
    lib = tm.load(args.lib)              # raises if the HIP extension is missing
    assert lib.backend == "hip", "bench.py measures the HIP path only"
    topo = synthetic.bench_topology(NUM_X, N_EXTRA)
    B = args.envs
    env, G, T, _ = distributed.make_rank_env(topo, B, rank, device=dev, lib=lib, seed=1234,
                                             n_action_sets=N_ACTION_SETS)
    elapsed, dev_ms = distributed.timed_rollout(env, G, T, args.steps, args.warmup, dist)
    st = int(env.status.sum().item())

    extras = {}
    if args.extras and rank == 0:
        # reference-exact topology (E = 76, half-bandwidth 7 -> 8-lane kernel)
        t76 = tm.TrussTopology.grid(NUM_X)
        b76 = synthetic.random_batch(t76, B, seed=99)
        e76 = tm.BatchedTruss(t76, B, device=dev, lib=lib)
        e76.set_constants(b76["x"], b76["target"], b76["y_max"], b76["d_min"], b76["max_def"], b76["load_x"],
                          b76["load_y"], b76["is_roof"])
        e76.set_design(b76["y"], b76["sec"])
        e76.analyze(set_normalisers=True)
        e76.rollout(G, T, args.warmup)
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(); e76.rollout(G, T, args.steps); a1.record(); torch.cuda.synchronize()
        extras["e76_env_steps_per_s"] = B * args.steps / (a0.elapsed_time(a1) * 1e-3)
        # observation kernel on top of the step (state-emitting configs)
        env.observe(); torch.cuda.synchronize()
        a0.record()
        for _ in range(50):
            env.observe()
        a1.record(); torch.cuda.synchronize()
        obs_us = a0.elapsed_time(a1) * 1e3 / 50
        obs_bytes = 4 * (13 * topo.N + 3 * topo.N ** 2 + 12 * topo.N + 21 * topo.E)
        extras["obs_kernel_us"] = obs_us
        extras["obs_GBps"] = B * obs_bytes / (obs_us * 1e-6) / 1e9
        # state-emitting step (configs 3-5): env.step + env.observe per step, against SURVEY §8d B_obs
        # (the reference's 5 N^2 matrices; A_n and mask are topology-static here and are not re-written)
        ag0, at0 = G[0].contiguous(), T[0].contiguous()
        env.step(ag0, at0); env.observe(); torch.cuda.synchronize()
        a0.record()
        for _ in range(100):
            env.step(ag0, at0)
            env.observe()
        a1.record(); torch.cuda.synchronize()
        so_us = a0.elapsed_time(a1) * 1e3 / 100
        b_obs = algorithmic_bytes_per_env_step(topo.N, topo.E) + 4 * (13 * topo.N + 5 * topo.N ** 2 + 12 * topo.N + 21 * topo.E)
        extras["step_then_obs_two_launches_us"] = so_us
        # the same through TRUSS_F_EMIT_OBS: the step's own launch writes the observation tensors
        ob = env.obs_buffers()
        env.step(ag0, at0, obs=ob); torch.cuda.synchronize()
        a0.record()
        for _ in range(200):
            env.step(ag0, at0, obs=ob)
        a1.record(); torch.cuda.synchronize()
        so_us = a0.elapsed_time(a1) * 1e3 / 200
        moved = algorithmic_bytes_per_env_step(topo.N, topo.E) + obs_bytes
        extras["fused_obs_one_launch"] = bool(env.fused_obs)
        extras["step_plus_obs_us"] = so_us
        extras["step_plus_obs_env_steps_per_s"] = B / (so_us * 1e-6)
        extras["step_plus_obs_frac_of_hbm_peak_B_obs"] = B * b_obs / (so_us * 1e-6) / 1e9 / HBM_PEAK_GBS
        extras["step_plus_obs_frac_of_hbm_peak_bytes_moved"] = B * moved / (so_us * 1e-6) / 1e9 / HBM_PEAK_GBS
        # BASELINE configs[1] topology: small bridge, 16 nodes / 36 elements (reference-exact grid)
        t36 = tm.TrussTopology.grid(8)
        b36 = synthetic.random_batch(t36, B, seed=7)
        e36 = tm.BatchedTruss(t36, B, device=dev, lib=lib)
        e36.set_constants(b36["x"], b36["target"], b36["y_max"], b36["d_min"], b36["max_def"], b36["load_x"],
                          b36["load_y"], b36["is_roof"])
        e36.set_design(b36["y"], b36["sec"])
        e36.analyze(set_normalisers=True)
        g36, a36 = synthetic.random_actions(N_ACTION_SETS, B, t36.N, 5)
        g36, a36 = torch.tensor(g36, device=dev), torch.tensor(a36, device=dev)
        e36.rollout(g36, a36, args.warmup); torch.cuda.synchronize()
        a0.record(); e36.rollout(g36, a36, args.steps); a1.record(); torch.cuda.synchronize()
        extras["small_bridge_16n36e_env_steps_per_s"] = B * args.steps / (a0.elapsed_time(a1) * 1e-3)
        extras["small_bridge_lanes_per_env"] = t36.solver_info(lib)["lanes_per_env"]
        # BASELINE configs[4] sizes: 128- and 256-node trusses (316 / 636 elements) on the 32- / 64-lane kernels
        for nx_big, b_big in ((64, 2048), (128, 1024)):
            tb = tm.TrussTopology.grid(nx_big)
            bb = synthetic.random_batch(tb, b_big, seed=nx_big)
            eb = tm.BatchedTruss(tb, b_big, device=dev, lib=lib)
            eb.set_constants(bb["x"], bb["target"], bb["y_max"], bb["d_min"], bb["max_def"], bb["load_x"], bb["load_y"], bb["is_roof"])
            eb.set_design(bb["y"], bb["sec"])
            eb.analyze(set_normalisers=True)
            gb, ab = synthetic.random_actions(2, b_big, tb.N, 5)
            gb, ab = torch.tensor(gb, device=dev), torch.tensor(ab, device=dev)
            eb.rollout(gb, ab, 10); torch.cuda.synchronize()
            a0.record(); eb.rollout(gb, ab, 50); a1.record(); torch.cuda.synchronize()
            us = a0.elapsed_time(a1) * 1e3 / 50
            extras[f"large_{tb.N}n_{tb.E}e_{b_big}envs"] = {"us_per_step": us, "env_steps_per_s": b_big / (us * 1e-6),
                                                               "lanes_per_env": tb.solver_info(lib)["lanes_per_env"],
                                                               "nonpositive_pivots": int(eb.status.sum().item())}

    if rank == 0:
        per_step_bytes = algorithmic_bytes_per_env_step(topo.N, topo.E) * B
        kern_s = dev_ms * 1e-3 / args.steps
        achieved = per_step_bytes / kern_s / 1e9
        info = topo.solver_info(lib)
        out = {
            "metric": "env steps/sec (batched FEM solves) at 4096 envs",
            "value": B * world * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic random-geometry 2-row trusses, 32 nodes / 80 elements / 60 DOF, "
                            "FEM-only env.step (action decode + FP64 assembly/solve + stresses + point), no agent",
                "envs_per_gpu": B, "global_envs": B * world, "nodes": topo.N, "elements": topo.E,
                "ndof": int(env.ndof), "half_bandwidth": info["half_bandwidth"],
                "lanes_per_env": info["lanes_per_env"], "rows_per_lane": info["rows_per_lane"],
                "parallelism": f"env-batch sharded over {world} GPU(s), no data-path collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
                # separate passes): profiles/r1/07_pmc_counters.json (valid for envs=4096, 32n/80e)
                "traffic": 13468928 if (B == 4096 and topo.N == 32 and topo.E == 80) else None,
                "kernel": "truss_step_kernel", "kernel_us": kern_s * 1e6,
                "bytes_per_launch": per_step_bytes,
                # the other two ceilings SURVEY §8d asks for (algorithmic banded flop count, 1e4 per env-step)
                "fp64_gflops": 1.0e4 * B / kern_s / 1e9, "fp64_frac_of_78.6_TFLOPs": 1.0e4 * B / kern_s / 78.6e12,
            },
            "nonpositive_pivots": st,
        }
        if extras:
            out["extras"] = extras
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(topo, seed=1234)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
