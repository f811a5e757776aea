"""Workloads of BASELINE.json configs[2]-[4] as functions (used by bench.py's default `configs` object and by the CLIs under
tools/): the batched MADDPG rollout on small_roof, the mixed 32-256-node pool (FEM step and Pareto sweep with one set of
agents), and large_bridge with its env batch sharded over the ranks present.  One env-step = one agent's modification of
one design (one `_game_modify`, like the FEM-only metric); a game step of the batched loop
(master_DDPG_truss2D_MO.run() :198-681 for B trusses at once, truss_mi355/marl.py) makes 3 x live archive members of them
per env."""
import contextlib
import io
import os
import time

import numpy as np
import torch


def _maddpg(dev, dist=None):
    import master_DDPG_truss2D_MO as M
    import truss2D_RL as RL
    return RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, M.a_nn, M.c_nn, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma,
                     device=dev, dist=dist)


def _play(eng, steps, train, warm, dist=None, device=None):
    quiet = contextlib.redirect_stdout(io.StringIO())
    with quiet:
        for _ in range(warm):
            eng.game_step_all(train=train)                    # lazy layers, first launches, GEMM selection, graph capture
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    e0, t0 = eng.env_steps, time.perf_counter()
    with quiet:
        for _ in range(steps):
            st = eng.game_step_all(train=train)
    torch.cuda.synchronize()
    return eng.env_steps - e0, time.perf_counter() - t0, st


def marl_small_roof(B=4096, steps=4, train=True, nx=8, dev="cuda", tune=True, profile=False):
    """configs[2]: roof truss of 2 nx nodes (nx = 8: test/01_small_roof, 16 nodes / 36 elements), B envs, MADDPG GCN
    actors / critics in the loop, one GPU.  nx = 16 / 32 / 64 / 128: the size classes of configs[4]."""
    import truss_mi355 as tm
    from truss_mi355 import marl
    topo = tm.TrussTopology.grid(nx)
    eng = marl.BatchedMARL(topo, B, _maddpg(dev), max_front=20, device=dev, replay_capacity=32768, batch_size=32, tune_update_gemms=tune)
    x = np.tile(np.arange(nx) * 5.0, 2)
    tar = np.concatenate([np.zeros(nx), 2.0 + 2.0 * np.abs(np.linspace(-1, 1, nx))])
    y0 = np.concatenate([np.zeros(nx), np.full(nx, 8.0)]).astype(np.float32)
    eng.reset(x[None].repeat(B, 0), tar[None].repeat(B, 0), 8.0, 0.3, 0.001 * 5.0 * (nx - 1), 0.0, -120000.0 * 8 / nx, 1.0,
              y0[None].repeat(B, 0), np.full((B, topo.E), 4, np.int32))
    if profile:
        eng.profile = {}
    n, dt, st = _play(eng, steps, train, 2 if tune else 1)
    return {"config": f"roof truss {topo.N}n/{topo.E}e, MADDPG GCN agents in the loop", "envs": B, "game_steps": steps, "train": train,
            "env_steps": n, "seconds": dt, "env_steps_per_s": n / dt, "mean_front": float(st["n_front"].float().mean()),
            "mean_hv": float(st["hv"].mean()), "replay_size": st["replay_size"], "profile_s": eng.profile}


def mixed_marl(envs=(1024, 512, 256, 128), num_xs=(16, 32, 64, 128), steps=3, train=False, dev="cuda"):
    """configs[4]: the multi-objective Pareto sweep over a MIX of truss sizes (32 / 64 / 128 / 256 nodes) with one set of
    MADDPG agents (marl.MixedMARL over pool.grid_classes), one GPU."""
    from truss_mi355 import marl, pool, synthetic
    classes = pool.grid_classes(list(num_xs), list(envs))
    eng = marl.MixedMARL(classes, _maddpg(dev), max_front=20, device=dev, replay_capacity=4096, batch_size=32)
    eng.reset([synthetic.random_batch(e.topo, e.B, seed=11 + k) for k, e in enumerate(eng.engines)])
    # (with training every size class captures its own update graph the first time its turn comes: all of them inside the warm-up)
    if train:
        steps = max(steps, len(num_xs))                       # one update per game step, the classes in turn: every class once
    n, dt, st = _play(eng, steps, train, len(num_xs) + 1 if train else 2)
    return {"config": "mixed Pareto sweep, grid trusses of " + " / ".join(str(2 * n_) for n_ in num_xs) + " nodes, one MADDPG",
            "envs_per_class": list(envs), "game_steps": steps, "train": train, "env_steps": n, "seconds": dt,
            "env_steps_per_s": n / dt, "mean_front": float(st["n_front"].float().mean())}


def mixed_pool_step(lib, dev, envs=(2048, 1024, 512, 256), num_xs=(16, 32, 64, 128)):
    """configs[4], environment only: one FEM step (and one state-emitting step) of every env of the mixed pool."""
    from truss_mi355 import pool, synthetic
    p = pool.MixedTrussPool(pool.grid_classes(list(num_xs), list(envs)), bucket_envs=64, device=dev, lib=lib, streams=False)
    batches, acts = [], []
    for k, e in enumerate(p.envs):
        b = synthetic.random_batch(e.topo, e.B, seed=30 + k)
        batches.append(b)
        ag, at = synthetic.random_actions(1, e.B, e.N, 60 + k)
        acts.append((torch.tensor(ag[0], device=dev), torch.tensor(at[0], device=dev)))
    p.set_constants(batches)
    p.set_design(batches)
    p.analyze(set_normalisers=True)
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {"envs": list(p.sizes)}
    for tag, kw, reps in (("us_per_pool_step", {}, 50), ("us_per_pool_step_with_obs", {"obs": True}, 30)):
        for _ in range(5):
            p.step(acts, **kw)
        torch.cuda.synchronize()
        a0.record()
        for _ in range(reps):
            p.step(acts, **kw)
        a1.record()
        torch.cuda.synchronize()
        out[tag] = a0.elapsed_time(a1) * 1e3 / reps
    out["env_steps_per_s"] = p.n_envs / (out["us_per_pool_step"] * 1e-6)
    out["env_steps_per_s_with_obs"] = p.n_envs / (out["us_per_pool_step_with_obs"] * 1e-6)
    out["nonpositive_pivots"] = int((p.status & 1).sum().item())
    return out


def large_bridge(global_envs=8192, steps=3, train=True, nx=16, dev="cuda", dist=None, rank=0, world=1):
    """configs[3]: large_bridge (32 nodes / 76 elements), `global_envs` envs split contiguously over the ranks present, one
    MADDPG per rank whose gradients are all-reduced over RCCL (the only collective; with no process group: one GPU, the whole
    batch on it).  Every rank returns the job's figures (sum of env-steps over ranks / slowest rank's time)."""
    import truss_mi355 as tm
    from truss_mi355 import distributed, marl
    lo, hi = distributed.shard_bounds(global_envs, world, rank)
    B = hi - lo
    topo = tm.TrussTopology.grid(nx)
    torch.manual_seed(7)                                      # same initial weights everywhere (and broadcast once more by the engine)
    eng = marl.BatchedMARL(topo, B, _maddpg(dev, dist), max_front=20, device=dev, replay_capacity=32768, batch_size=32, seed=rank)
    if world > 1 and os.environ.get("TRUSS_DP_GRAPH", "0") != "1":
        # The update with its RCCL all-reduces captured into a hipGraph is tested with a one-rank group (tests/test_rccl_single_gpu.py);
        # with several ranks it has never run on hardware, so the default there is the eager update (collectives outside any graph,
        # the path the world-size-2 gloo tests cover).  TRUSS_DP_GRAPH=1 opts in.
        eng.use_train_graph = False
    x = np.tile(np.arange(nx) * 5.0, 2)                       # test/02_large_bridge: 15 bays of 5 m, span_y 6, targets 3.0 ... 2.0 ... 3.0
    tar = np.concatenate([np.zeros(nx), 2.0 + np.abs(np.linspace(-1, 1, nx))])
    y0 = np.concatenate([np.zeros(nx), np.full(nx, 6.0)]).astype(np.float32)
    eng.reset(x[None].repeat(B, 0), tar[None].repeat(B, 0), 6.0, 0.3, 0.001 * 5.0 * (nx - 1), 0.0, -7500.0, 0.0, y0[None].repeat(B, 0),
              np.full((B, topo.E), 4, np.int32))
    n, dt, _ = _play(eng, steps, train, 2, dist=dist)
    tot = torch.tensor([float(n), dt], dtype=torch.float64, device=dev)
    if dist is not None:
        s = tot[:1].clone()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(tot[1:], op=dist.ReduceOp.MAX)
        tot[0] = s[0]
    graph = eng._tg is not None
    return {"config": f"large_bridge {topo.N}n/{topo.E}e, {global_envs} envs over {world} GPU(s), MADDPG GCN agents, "
                      f"{'one update per game step' if train else 'no training'}",
            "n_gpus": world, "envs_per_gpu": B, "game_steps": steps, "train": train, "env_steps": tot[0].item(), "seconds": tot[1].item(),
            "env_steps_per_s": tot[0].item() / tot[1].item(),
            "update": ("hipGraph replay" if graph else "eager") + (" with its RCCL all-reduces" if dist is not None else "") if train else None}
