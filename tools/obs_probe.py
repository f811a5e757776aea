"""Diagnostic: the stand-alone observation kernel (truss_obs) per size class: us per launch and GB/s of tensor bytes written."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import torch
import truss_mi355 as tm
from truss_mi355 import synthetic

lib = tm.load()
dev = torch.device("cuda", 0)
a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}
for nx, B in ((8, 4096), (16, 4096), (16, 2048), (32, 1024), (64, 512), (128, 256)):
    topo = tm.TrussTopology.grid(nx)
    b = synthetic.random_batch(topo, B, seed=nx)
    e = tm.BatchedTruss(topo, B, device=dev, lib=lib)
    e.set_constants(b["x"], b["target"], b["y_max"], b["d_min"], b["max_def"], b["load_x"], b["load_y"], b["is_roof"])
    e.set_design(b["y"], b["sec"])
    e.analyze(set_normalisers=True)
    for _ in range(5):
        e.observe()
    torch.cuda.synchronize()
    a0.record()
    for _ in range(30):
        e.observe()
    a1.record(); torch.cuda.synchronize()
    us = a0.elapsed_time(a1) * 1e3 / 30
    N, E = topo.N, topo.E
    nbytes = 4 * (13 * N + 3 * N * N + 12 * N + 21 * E) * B
    out[f"{N}n_{E}e_{B}envs"] = {"us": round(us, 1), "MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / us / 1e3, 0)}
print(json.dumps(out))
