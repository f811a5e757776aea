#!/usr/bin/env python3
"""Diagnostic: timeline of one workgroup of the FUSED step (TRUSS_F_EMIT_OBS): compute wave phases and progress
publications, streaming wave segments.  Stamped build (make -C mop-truss-marl_amd/csrc diag); the stamped build's
run time is not a benchmark number."""
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
lib.dll.truss_debug_stamps32.argtypes = [ctypes.c_void_p]
topo = synthetic.bench_topology(16, 4)
batch = synthetic.random_batch(topo, B, 1)
env = tm.BatchedTruss(topo, B, lib=lib)
env.set_constants(batch["x"], batch["target"], batch["y_max"], batch["d_min"], batch["max_def"], batch["load_x"], batch["load_y"], batch["is_roof"])
env.set_design(batch["y"], batch["sec"])
env.analyze(set_normalisers=True)
ag, at = synthetic.random_actions(2, B, topo.N, 2)
G, T = torch.tensor(ag[0], device=env.device), torch.tensor(at[0], device=env.device)
full = env.obs_buffers()
SUB = {"all": full, "none": None, "rows_only": {k: full[k] for k in ("x_n", "nN_x_n", "nN_x_e")},
       "matrices_only": {k: full[k] for k in ("A_s", "A_n_ts", "A_n_cs")}, "A_s_only": {"A_s": full["A_s"]},
       "no_A_s": {k: v for k, v in full.items() if k != "A_s"}}
which = sys.argv[3] if len(sys.argv) > 3 else "all"
ob = SUB[which]
for _ in range(20):
    env.step(G, T, obs=ob)
torch.cuda.synchronize()
EVENTS = [(0, "C start"), (1, "C staged"), (19, "C publish 1 (sections final)"), (4, "C solver starts"), (6, "C solver done"),
          (18, "C publish 2 (element bank)"), (8, "C post_nodes done"), (15, "C own stores issued"), (16, "C raw node features done"),
          (17, "C publish 3 (node bank)"), (9, "C end"),
          (20, "S seg1 start (A_s)"), (21, "S seg1 issued"), (22, "S seg2 start (A_n_ts/cs)"), (23, "S seg2 issued"),
          (24, "S seg3 start (rows)"), (25, "S seg3 issued"), (26, "S all stores retired")]
acc = np.zeros(len(EVENTS))
N = 10
for _ in range(N):
    env.step(G, T, obs=ob)
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 32)()
    lib.dll.truss_debug_stamps32(st)
    acc += np.array([float(st[i]) - float(st[0]) for i, _ in EVENTS])
acc /= N
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    env.step(G, T, obs=ob)
e1.record(); torch.cuda.synchronize()
print(f"subset={which} fused={env.fused_obs} B={B}; stamped launch {e0.elapsed_time(e1) * 10:.2f} us; cycles since the compute wave's start (mid-grid workgroup)")
for (i, n), c in sorted(zip(EVENTS, acc), key=lambda t: t[1]):
    print(f"   {c:9.0f}  {n}")

# spread of the workgroups over the launch (wall clock, 100 MHz): first start -> last end of either wave
nb = B * topo.solver_info(lib)["lanes_per_env"] // 64
env.step(G, T, obs=ob); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (6 * nb))()
lib.dll.truss_debug_span.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.dll.truss_debug_span(buf, nb) == 0
a = np.array(list(buf), dtype=np.float64).reshape(nb, 6)
t0 = a[:, 1].min()
cs, ce, se = (a[:, 1] - t0) * 10, (a[:, 3] - t0) * 10, (a[:, 5] - t0) * 10
print(f"workgroups {nb}: compute-wave start  median {np.median(cs):.0f} ns, 90% {np.percentile(cs, 90):.0f}, max {cs.max():.0f}")
print(f"   compute-wave end    median {np.median(ce):.0f} ns, max {ce.max():.0f};   duration median {np.median(ce - cs):.0f} ns")
if ob is not None:
    print(f"   streaming-wave end  median {np.median(se):.0f} ns, max {se.max():.0f}")
