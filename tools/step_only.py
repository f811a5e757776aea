#!/usr/bin/env python3
"""Diagnostic driver: N plain (FEM-only) steps of the metric topology at 4096 envs, one launch per step, nothing else (for PMC passes
over diagnostic builds, e.g. tools/lds_by_phase.sh):  step_only.py <lib.so> [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
os.environ["TRUSS_ROLLOUT_LAUNCHES"] = "1"
import torch
import truss_mi355 as tm
from truss_mi355 import synthetic, distributed
lib = tm.load(os.path.abspath(sys.argv[1]))
topo = synthetic.bench_topology(16, 4)
env, G, T, _ = distributed.make_rank_env(topo, 4096, 0, device=torch.device("cuda", 0), lib=lib, seed=1234, n_action_sets=4)
env.rollout(G, T, int(sys.argv[2]) if len(sys.argv) > 2 else 20)
torch.cuda.synchronize()
