#!/usr/bin/env python3
"""BASELINE configs[2]: small_roof, 4096 envs, MADDPG GNN actors/critics in the loop, one MI355X.
Times whole game steps of truss_mi355.marl.BatchedMARL (FEM steps + observations + actor inference + rewards
+ archive update + replay + one MADDPG update per game step) and reports env-steps/s (one env-step = one
agent's modification of one design, as in the FEM-only metric)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mop-truss-marl_amd"))
import contextlib
import io
import numpy as np
import torch
import truss_mi355 as tm
from truss_mi355 import marl
import master_DDPG_truss2D_MO as M
import truss2D_RL as RL


def run(B=4096, steps=6, train=True, nx=8, profile=False, tune=None):
    """nx = bays + 1; 8 = test/01_small_roof (16 nodes, 36 elements); 16 / 32 / 64 / 128 = the size classes of BASELINE configs[4]"""
    dev = "cuda"
    if tune is None:
        tune = os.environ.get("MARL_TUNE", "1") != "0"
    topo = tm.TrussTopology.grid(nx)
    rl = RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, M.a_nn, M.c_nn, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma, device=dev)
    eng = marl.BatchedMARL(topo, B, rl, max_front=20, device=dev, replay_capacity=32768, batch_size=32, tune_update_gemms=tune)
    x = np.tile(np.arange(nx) * 5.0, 2)
    tar = np.concatenate([np.zeros(nx), 2.0 + 2.0 * np.abs(np.linspace(-1, 1, nx))])
    y0 = np.concatenate([np.zeros(nx), np.full(nx, 8.0)]).astype(np.float32)
    eng.reset(x[None].repeat(B, 0), tar[None].repeat(B, 0), 8.0, 0.3, 0.001 * 5.0 * (nx - 1), 0.0, -120000.0 * 8 / nx, 1.0, y0[None].repeat(B, 0),
              np.full((B, topo.E), 4, np.int32))
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(2 if tune else 1):
            eng.game_step_all(train=train)                # warm-up (lazy layers, first launches, GEMM selection)
    torch.cuda.synchronize()
    e0 = eng.env_steps
    if profile:
        eng.profile = {}
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(steps):
            st = eng.game_step_all(train=train)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"config": f"roof truss {topo.N}n/{topo.E}e, MADDPG GCN agents in the loop", "envs": B, "game_steps": steps, "train": train,
            "env_steps": eng.env_steps - e0, "seconds": dt, "env_steps_per_s": (eng.env_steps - e0) / dt,
            "mean_front": float(st["n_front"].float().mean()), "mean_hv": float(st["hv"].mean()),
            "replay_size": st["replay_size"], "profile_s": eng.profile}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    train = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
    nx = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    print(json.dumps(run(B, steps, train, nx, bool(os.environ.get("MARL_PROFILE")))))


if __name__ == "__main__":
    main()
