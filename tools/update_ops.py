"""Diagnostic: which aten operators (and how many device kernels) one eager MADDPG update consists of (torch.profiler)."""
import os, sys, contextlib, io, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
import truss_mi355 as tm
from truss_mi355 import marl, synthetic
import master_DDPG_truss2D_MO as M
import truss2D_RL as RL

topo = tm.TrussTopology.grid(8)
B = 512
rl = RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, M.a_nn, M.c_nn, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma, device="cuda")
eng = marl.BatchedMARL(topo, B, rl, max_front=20, device="cuda", replay_capacity=8192, batch_size=32, tune_update_gemms=False)
eng.use_train_graph = False
b = synthetic.random_batch(topo, B, 3)
eng.reset(b["x"], b["target"], b["y_max"], b["d_min"], b["max_def"], b["load_x"], b["load_y"], b["is_roof"], b["y"], b["sec"])
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(3):
        eng.game_step_all(train=True)
S, NS, ag, at, R = eng.replay.sample(32, eng.gen)
A = [(ag[:, a].contiguous(), at[:, a].contiguous()) for a in range(3)]
st, nst = eng._net_state(S), [eng._net_state(ns) for ns in NS]
rl.train_on_batch(st, nst, A, R)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    rl.train_on_batch(st, nst, A, R)
    torch.cuda.synchronize()
ev = prof.key_averages()
kern = [(e.count, e.key) for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
print("device kernels:", sum(c for c, _ in kern))
for c, k in sorted(kern, reverse=True)[:25]:
    print("  K", c, k[:110])
ops = [(e.count, e.key) for e in ev if e.device_type == torch.autograd.DeviceType.CPU and e.key.startswith("aten::")]
for c, k in sorted(ops, reverse=True)[:40]:
    print("  op", c, k)

for key in ("aten::zero_", "aten::fill_", "aten::copy_", "aten::cat", "aten::sum", "aten::add", "aten::mul"):
    evs = [e for e in prof.key_averages(group_by_stack_n=6) if e.key == key]
    for e in sorted(evs, key=lambda e: -e.count)[:8]:
        print(key, e.count, [fr for fr in e.stack if "site-packages" not in fr and "dist-packages" not in fr][:4] or e.stack[:6])
