#!/usr/bin/env python3
"""Diagnostic: cycles per phase of one workgroup of the FUSED step (TRUSS_F_EMIT_OBS), stamped build
(make -C mop-truss-marl_amd/csrc diag).  Shares only -- the stamped build's run time is not a benchmark number."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mop-truss-marl_amd"))
import numpy as np
import torch
import truss_mi355 as tm
from truss_mi355 import synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
path = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "mop-truss-marl_amd", "csrc", "libtruss_mi355_diag.so")
lib = tm.load(path)
lib.dll.truss_debug_stamps24.argtypes = [ctypes.c_void_p]
topo = synthetic.bench_topology(16, 4)
batch = synthetic.random_batch(topo, B, 1)
env = tm.BatchedTruss(topo, B, lib=lib)
env.set_constants(batch["x"], batch["target"], batch["y_max"], batch["d_min"], batch["max_def"], batch["load_x"], batch["load_y"], batch["is_roof"])
env.set_design(batch["y"], batch["sec"])
env.analyze(set_normalisers=True)
ag, at = synthetic.random_actions(2, B, topo.N, 2)
G, T = torch.tensor(ag[0], device=env.device), torch.tensor(at[0], device=env.device)
ob = env.obs_buffers()
for _ in range(20):
    env.step(G, T, obs=ob)
torch.cuda.synchronize()
# order of the stamps along the schedule
ORDER = [0, 1, 2, 3, 10, 11, 4, 13, 5, 14, 6, 7, 8, 12, 9, 15, 16, 17, 18]
NAMES = ["stage", "decode", "sizing", "elements", "assemble_nodes", "scratch_init", "factor clean", "factor merge+check",
         "backsub handover", "backsub clean", "post_elements", "post_nodes", "finish", "store rows", "(barrier)",
         "obs nodes raw", "obs bank fill", "obs emit"]
acc = np.zeros(len(ORDER) - 1)
for _ in range(10):
    env.step(G, T, obs=ob)
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 24)()
    lib.dll.truss_debug_stamps24(st)
    s = np.array([st[i] for i in ORDER], dtype=np.float64)
    acc += np.diff(s)
acc /= 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    env.step(G, T, obs=ob)
e1.record(); torch.cuda.synchronize()
print(f"fused={env.fused_obs} B={B} total {acc.sum():.0f} cycles; stamped launch {e0.elapsed_time(e1) * 10:.2f} us")
for n, c in zip(NAMES, acc):
    print(f"   {n:22s} {c:9.0f} cyc  {100 * c / acc.sum():5.1f} %")
