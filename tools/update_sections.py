"""Diagnostic: device kernels per section of the MADDPG update (grouped passes, backward, clip, optimiser steps), torch.profiler."""
import os, sys, contextlib, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
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
st = eng._net_state(S)
acts = [A[0][0], A[0][1], A[1][0], A[1][1], A[2][0], A[2][1]]
a0 = rl.agents[0]
cs = {}
ain = rl._actor_in(st)


def count(tag, fn, detail=False):
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    kern = [(e.count, e.key) for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
    print(f"{tag:44s} {sum(c for c, _ in kern):5d} kernels")
    if detail:
        for c, k in sorted(kern, reverse=True)[:14]:
            print("      ", c, k[:100])


def critics_fb():
    cps = [p for ag_ in rl.agents for p in ag_.critic_model.parameters()]
    qs = RL.run_networks([RL._critic_steps(ag_.critic_model, st + acts) for ag_ in rl.agents], cs)
    torch.autograd.grad(torch.cat(qs, 1).pow(2).mean(0).sum(), cps)


def actor_fb():
    ap = list(a0.actor_model.parameters())
    g, t = RL.run_networks([RL._actor_steps(a0.actor_model, ain)], cs)[0]
    q = RL.run_networks([RL._critic_steps(a0.critic_model, st + [g, t] + acts[2:], frozen=True)], cs)[0]
    grads = torch.autograd.grad(-q.mean(), ap, allow_unused=True)
    for p, g_ in zip(ap, grads):
        p.grad = g_


with torch.no_grad():
    count("actor forward (no grad)", lambda: RL.run_networks([RL._actor_steps(a0.actor_model, ain)], cs), True)
    count("3 actors forward together (no grad)", lambda: RL.run_networks([RL._actor_steps(ag_.actor_model, ain) for ag_ in rl.agents], cs))
    count("critic forward (no grad)", lambda: RL.run_networks([RL._critic_steps(a0.critic_model, st + acts)], cs), True)
    count("3 critics forward together (no grad)", lambda: RL.run_networks([RL._critic_steps(ag_.critic_model, st + acts) for ag_ in rl.agents], cs))
count("3 critics forward + backward together", critics_fb, True)
count("actor + critic forward, backward to the actor", actor_fb, True)
cp = [p for ag_ in rl.agents for p in ag_.critic_model.parameters()]
flat = RL._flat_grads(cp)
count("critics: flat gradient + clip", lambda: RL._clip_flat(RL._flat_grads(cp), cp))
count("critics: Adam step", lambda: rl.critics_opt.step(flat))
ap = list(a0.actor_model.parameters())
count("actor: flat gradient + clip + fresh-Adam step", lambda: RL._fresh_adam_step(ap, 1e-4, 1e-7, RL._clip_flat(RL._flat_grads(ap), ap)))
A3 = [(ag[:, a].contiguous(), at[:, a].contiguous()) for a in range(3)]
nst = [eng._net_state(ns) for ns in NS]
count("whole update", lambda: rl.train_on_batch(st, nst, A3, R), True)
