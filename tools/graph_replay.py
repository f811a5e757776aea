#!/usr/bin/env python3
"""Does a hipGraph replay shrink the gap between the chained step launches?  (diagnostic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mop-truss-marl_amd"))
import torch
import truss_mi355 as tm
from truss_mi355 import synthetic

B = 4096
lib = tm.load()
topo = synthetic.bench_topology(16, 4)
batch = synthetic.random_batch(topo, B, 1)
env = tm.BatchedTruss(topo, B, lib=lib)
env.set_constants(batch["x"], batch["target"], batch["y_max"], batch["d_min"], batch["max_def"], batch["load_x"],
                  batch["load_y"], batch["is_roof"])
env.set_design(batch["y"], batch["sec"])
env.analyze(set_normalisers=True)
ag, at = synthetic.random_actions(8, B, topo.N, 2)
G, T = torch.tensor(ag, device=env.device), torch.tensor(at, device=env.device)
env.rollout(G, T, 40)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); env.rollout(G, T, 400); e1.record(); torch.cuda.synchronize()
print(f"direct launches : {e0.elapsed_time(e1) * 1e3 / 400:.2f} us per step")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    env.rollout(G, T, 8)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    env.rollout(G, T, 40)     # even count: the ping-pong design buffers end where they started
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(10):
    g.replay()
e1.record(); torch.cuda.synchronize()
print(f"hipGraph replay : {e0.elapsed_time(e1) * 1e3 / 400:.2f} us per step (10 replays of a 40-step graph)")
