#!/usr/bin/env python3
"""BASELINE configs[3]: large_bridge (32 nodes / 76 elements), 8192 envs sharded over the GPUs of one node, one MADDPG whose
gradients are all-reduced over RCCL (the only collective: the env batch shards with no data-path exchange).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/marl_dist_bench.py [--global-envs 8192]
  python tools/marl_dist_bench.py                      # one GPU: the whole batch on it, no process group

Every rank plays its contiguous shard of the envs (own archives, own replay), draws its own minibatch, and takes part in the
collective update (`MADDPG.train_on_batch` all-reduces the flat gradient buffer of each network; weights are broadcast from rank 0
once the lazy layers exist).  Rank 0 prints one JSON line: env-steps/s of the whole job (sum over ranks / slowest rank's time)."""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mop-truss-marl_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--global-envs", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--train", type=int, default=1)
    ap.add_argument("--nx", type=int, default=16, help="bays + 1: 16 = 32 nodes / 76 elements (large_bridge)")
    args = ap.parse_args()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    import numpy as np
    import torch
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)       # "nccl" = RCCL on ROCm
    import truss_mi355 as tm
    from truss_mi355 import distributed, marl
    import master_DDPG_truss2D_MO as M
    import truss2D_RL as RL
    lo, hi = distributed.shard_bounds(args.global_envs, world, rank)
    B, nx = hi - lo, args.nx
    topo = tm.TrussTopology.grid(nx)
    torch.manual_seed(7)                                      # same initial weights everywhere (and broadcast once more by the engine)
    rl = RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, M.a_nn, M.c_nn, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma, device=dev, dist=dist)
    eng = marl.BatchedMARL(topo, B, rl, max_front=20, device=dev, replay_capacity=32768, batch_size=32, seed=rank)
    x = np.tile(np.arange(nx) * 5.0, 2)                       # test/02_large_bridge: 15 bays of 5 m, span_y 6, targets 3.0 ... 2.0 ... 3.0
    tar = np.concatenate([np.zeros(nx), 2.0 + np.abs(np.linspace(-1, 1, nx))])
    y0 = np.concatenate([np.zeros(nx), np.full(nx, 6.0)]).astype(np.float32)
    eng.reset(x[None].repeat(B, 0), tar[None].repeat(B, 0), 6.0, 0.3, 0.001 * 5.0 * (nx - 1), 0.0, -7500.0, 0.0, y0[None].repeat(B, 0),
              np.full((B, topo.E), 4, np.int32))
    quiet = contextlib.redirect_stdout(io.StringIO())
    with quiet:
        for _ in range(2):
            eng.game_step_all(train=bool(args.train))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    e0, t0 = eng.env_steps, time.perf_counter()
    with quiet:
        for _ in range(args.steps):
            eng.game_step_all(train=bool(args.train))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = torch.tensor([float(eng.env_steps - e0), dt], dtype=torch.float64, device=dev)
    if dist is not None:
        steps_all = tot[:1].clone()
        dist.all_reduce(steps_all, op=dist.ReduceOp.SUM)
        dist.all_reduce(tot[1:], op=dist.ReduceOp.MAX)
        tot[0] = steps_all[0]
    if rank == 0:
        print(json.dumps({"config": f"large_bridge {topo.N}n/{topo.E}e, {args.global_envs} envs over {world} GPU(s), MADDPG GCN agents, "
                                    f"{'one collective update per game step' if args.train else 'no training'}",
                          "n_gpus": world, "envs_per_gpu": B, "game_steps": args.steps, "env_steps": tot[0].item(), "seconds": tot[1].item(),
                          "env_steps_per_s": tot[0].item() / tot[1].item(), "update": "hipGraph replay" if world == 1 else "eager + RCCL all-reduce"}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
