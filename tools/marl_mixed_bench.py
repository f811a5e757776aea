#!/usr/bin/env python3
"""BASELINE configs[4]: the multi-objective Pareto sweep over a MIX of truss sizes (32 / 64 / 128 / 256 nodes) with one set of
MADDPG agents (truss_mi355.marl.MixedMARL over pool.grid_classes), one MI355X.  Reports env-steps/s of whole game steps
(FEM + observations + actors + rewards + archive + replay [+ one MADDPG update per game step])."""
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mop-truss-marl_amd"))
import numpy as np
import torch
import truss_mi355 as tm
from truss_mi355 import marl, pool, synthetic
import master_DDPG_truss2D_MO as M
import truss2D_RL as RL


def run(envs=(1024, 512, 256, 128), num_xs=(16, 32, 64, 128), steps=3, train=False):
    dev = "cuda"
    rl = RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, M.a_nn, M.c_nn, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma, device=dev)
    classes = pool.grid_classes(list(num_xs), list(envs))
    eng = marl.MixedMARL(classes, rl, max_front=20, device=dev, replay_capacity=4096, batch_size=32)
    per_class = []
    for k, e in enumerate(eng.engines):
        b = synthetic.random_batch(e.topo, e.B, seed=11 + k)
        per_class.append(b)
    eng.reset(per_class)
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(2):
            eng.game_step_all(train=train)
    torch.cuda.synchronize()
    e0 = eng.env_steps
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(steps):
            st = eng.game_step_all(train=train)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"config": "mixed Pareto sweep, grid trusses of " + " / ".join(str(2 * n) for n in num_xs) + " nodes, one MADDPG",
            "envs_per_class": list(envs), "game_steps": steps, "train": train, "env_steps": eng.env_steps - e0, "seconds": dt,
            "env_steps_per_s": (eng.env_steps - e0) / dt, "mean_front": float(st["n_front"].float().mean())}


if __name__ == "__main__":
    train = len(sys.argv) > 1 and sys.argv[1] != "0"
    print(json.dumps(run(train=train)))
