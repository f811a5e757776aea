#!/usr/bin/env python3
"""BASELINE configs[4]: the multi-objective Pareto sweep over a MIX of truss sizes (32 / 64 / 128 / 256 nodes) with one set of
MADDPG agents, one MI355X (bench_configs.mixed_marl).    tools/marl_mixed_bench.py [train 0/1]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import bench_configs


def run(envs=(1024, 512, 256, 128), num_xs=(16, 32, 64, 128), steps=3, train=False):
    return bench_configs.mixed_marl(envs, num_xs, steps, train)


if __name__ == "__main__":
    print(json.dumps(run(train=len(sys.argv) > 1 and sys.argv[1] != "0")))
