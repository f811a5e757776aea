"""Multi-GPU driver pieces of the batched rollout (one process per GPU).

The env batch shards with NO data-path collective (every truss is independent): rank r
owns `envs_per_rank` envs generated from seed `seed + 7919*r`.  `torch.distributed` (backend "nccl" =
RCCL on the GPUs, "gloo" in the CPU tests) is used only for the start barrier and the MAX-reduce of
the elapsed time, which is what bench.py reports.
"""
from __future__ import annotations

import os
import time

import torch

from . import synthetic
from .batched import BatchedTruss


def rank_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(global_envs, world, rank):
    """Contiguous split of a global env batch over the ranks: [lo, hi) of this rank; the first
    `global_envs % world` ranks take one env more."""
    q, r = divmod(int(global_envs), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def make_rank_env(topo, envs_per_rank, rank, device=None, lib=None, seed=1234, n_action_sets=8, debug_f64=False):
    """Resident state of this rank's shard: env batch + a pool of pre-drawn action sets."""
    batch = synthetic.random_batch(topo, envs_per_rank, seed=seed + 7919 * rank)
    env = BatchedTruss(topo, envs_per_rank, device=device, lib=lib, debug_f64=debug_f64)
    env.set_constants(batch["x"], batch["target"], batch["y_max"], batch["d_min"], batch["max_def"], batch["load_x"],
                      batch["load_y"], batch["is_roof"])
    env.set_design(batch["y"], batch["sec"])
    env.analyze(set_normalisers=True)
    ag, at = synthetic.random_actions(n_action_sets, envs_per_rank, topo.N, seed=seed + 3087 + rank)
    G = torch.tensor(ag, device=env.device)
    T = torch.tensor(at, device=env.device)
    return env, G, T, batch


def _sync(env, dist):
    if env.device.type == "cuda":
        torch.cuda.synchronize(env.device)
    if dist is not None:
        dist.barrier()
        if env.device.type == "cuda":
            torch.cuda.synchronize(env.device)


def timed_rollout(env, G, T, steps, warmup, dist=None):
    """W untimed + K timed steps bracketed by barrier + synchronize; returns (max-over-ranks wall
    seconds, device milliseconds of this rank's K launches or None on CPU)."""
    if warmup > 0:
        env.rollout(G, T, warmup)
    _sync(env, dist)
    cuda = env.device.type == "cuda"
    if cuda:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()            # on the stream the kernels are launched on (torch's current stream)
    t0 = time.perf_counter()
    env.rollout(G, T, steps)
    if cuda:
        ev1.record()
        torch.cuda.synchronize(env.device)
    elapsed = time.perf_counter() - t0
    _sync(env, dist)
    dev_ms = ev0.elapsed_time(ev1) if cuda else None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=env.device if cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, dev_ms


def global_checksum(env, dist=None):
    """Order-independent digest of every rank's design state (sum of heights and sections), summed over
    ranks: lets a multi-process run be checked against single-process runs of the same shards."""
    s = torch.stack([env.y.double().sum(), env.sec.double().sum(), env.point.double().sum()]).cpu()
    if dist is not None:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return s.numpy()
