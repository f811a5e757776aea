#!/usr/bin/env python3
"""BASELINE configs[3]: large_bridge (32 nodes / 76 elements), 8192 envs sharded over the GPUs of one node, one MADDPG whose
gradients are all-reduced over RCCL (bench_configs.large_bridge; the only collective: the env batch shards with no data-path
exchange).  bench.py's default line carries the same measurement (`configs.large_bridge_8192`).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/marl_dist_bench.py [--global-envs 8192]
  python tools/marl_dist_bench.py                      # one GPU: the whole batch on it, no process group"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--global-envs", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--train", type=int, default=1)
    ap.add_argument("--nx", type=int, default=16, help="bays + 1: 16 = 32 nodes / 76 elements (large_bridge)")
    args = ap.parse_args()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    import torch
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)       # "nccl" = RCCL on ROCm
    import bench_configs
    r = bench_configs.large_bridge(args.global_envs, args.steps, bool(args.train), args.nx, dev, dist, rank, world)
    if rank == 0:
        print(json.dumps(r))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
