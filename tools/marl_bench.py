#!/usr/bin/env python3
"""BASELINE configs[2]: small_roof, 4096 envs, MADDPG GNN actors/critics in the loop, one MI355X (bench_configs.marl_small_roof:
whole game steps of truss_mi355.marl.BatchedMARL -- FEM steps + observations + actor inference + rewards + archive update +
replay + one MADDPG update per game step -- in env-steps/s).    tools/marl_bench.py [envs] [game steps] [train 0/1] [nx]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import bench_configs


def run(B=4096, steps=6, train=True, nx=8, profile=False, tune=None):
    if tune is None:
        tune = os.environ.get("MARL_TUNE", "1") != "0"
    return bench_configs.marl_small_roof(B, steps, train, nx, tune=tune, profile=profile)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    train = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
    nx = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    print(json.dumps(run(B, steps, train, nx, bool(os.environ.get("MARL_PROFILE")))))


if __name__ == "__main__":
    main()
