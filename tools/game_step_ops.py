#!/usr/bin/env python3
"""Diagnostic: operators and device kernels of ONE batched game step (small_roof, 4096 envs) by section, torch.profiler.
   tools/game_step_ops.py [train 0/1]"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
import bench_configs as BC
import truss_mi355 as tm
from truss_mi355 import marl

train = len(sys.argv) > 1 and sys.argv[1] != "0"
nx, B = 8, 4096
topo = tm.TrussTopology.grid(nx)
eng = marl.BatchedMARL(topo, B, BC._maddpg("cuda"), max_front=20, device="cuda", replay_capacity=32768, batch_size=32, tune_update_gemms=False)
x = np.tile(np.arange(nx) * 5.0, 2)
tar = np.concatenate([np.zeros(nx), 2.0 + 2.0 * np.abs(np.linspace(-1, 1, nx))])
y0 = np.concatenate([np.zeros(nx), np.full(nx, 8.0)]).astype(np.float32)
eng.reset(x[None].repeat(B, 0), tar[None].repeat(B, 0), 8.0, 0.3, 0.001 * 5.0 * (nx - 1), 0.0, -120000.0 * 8 / nx, 1.0, y0[None].repeat(B, 0),
          np.full((B, topo.E), 4, np.int32))
q = contextlib.redirect_stdout(io.StringIO())
with q:
    for _ in range(3):
        eng.game_step_all(train=train)
torch.cuda.synchronize()
t0 = time.perf_counter()
with q:
    for _ in range(4):
        eng.game_step_all(train=train)
torch.cuda.synchronize()
print(f"game step: {(time.perf_counter() - t0) / 4 * 1e3:.2f} ms wall")
sections = {}
orig_tick = eng._tick


def tick(name, tk):
    torch.cuda.synchronize()
    return orig_tick(name, tk)


with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with q:
        eng.game_step_all(train=train)
    torch.cuda.synchronize()
ev = prof.key_averages()
kern = [(e.count, e.device_time_total / 1e3, e.key) for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
print("device kernels:", sum(c for c, _, _ in kern), " device time %.2f ms" % sum(t for _, t, _ in kern))
for c, t, k in sorted(kern, key=lambda r: -r[1])[:24]:
    print(f"  {c:5d}  {t:7.3f} ms  {k[:100]}")
ops = [(e.count, e.self_cpu_time_total / 1e3, e.key) for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
print("host: total self CPU time of operators %.2f ms" % sum(t for _, t, _ in ops))
for c, t, k in sorted(ops, key=lambda r: -r[1])[:12]:
    print(f"  {c:5d}  {t:7.2f} ms  {k[:90]}")
print("operators that launch, by count:")
skip = ("aten::as_strided", "aten::view", "aten::reshape", "aten::slice", "aten::select", "aten::expand", "aten::unsqueeze", "aten::empty", "aten::empty_strided",
        "aten::empty_like", "aten::permute", "aten::transpose", "aten::t", "aten::_unsafe_view", "aten::resize_", "aten::squeeze", "aten::alias", "aten::detach",
        "aten::to", "aten::result_type", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::view_as", "aten::narrow", "aten::contiguous")
for c, t, k in sorted((o for o in ops if o[2].startswith("aten::") and o[2] not in skip), key=lambda r: -r[0])[:30]:
    print(f"  {c:5d}  {k}")
