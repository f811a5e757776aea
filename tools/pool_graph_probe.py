"""Diagnostic: MixedTrussPool.step as eager launches (one stream / one stream per class) against a hipGraph whose
branches are the size classes (the per-class stream hand-shakes become graph edges)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import torch
import truss_mi355 as tm
from truss_mi355 import synthetic, pool

dev = torch.device("cuda", 0)
lib = tm.load()
classes = pool.grid_classes([16, 32, 64, 128], [2048, 1024, 512, 256])
a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a0.record()
    for _ in range(n):
        fn()
    a1.record(); torch.cuda.synchronize()
    return round(a0.elapsed_time(a1) * 1e3 / n, 1)


for streams in (False, True):
    p = pool.MixedTrussPool(classes, bucket_envs=64, device=dev, lib=lib, streams=streams)
    batches, acts = [], []
    for k, e in enumerate(p.envs):
        b = synthetic.random_batch(e.topo, e.B, seed=30 + k)
        batches.append(b)
        ag, at = synthetic.random_actions(1, e.B, e.N, 60 + k)
        acts.append((torch.tensor(ag[0], device=dev), torch.tensor(at[0], device=dev)))
    p.set_constants(batches); p.set_design(batches); p.analyze(set_normalisers=True)
    tag = "streams" if streams else "one_stream"
    out[tag + "_eager_us"] = timed(lambda: p.step(acts))
    out[tag + "_eager_obs_us"] = timed(lambda: p.step(acts, obs=True), 30)
    for obs in (None, True):
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                p.step(acts, obs=obs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            p.step(acts, obs=obs)
            p.step(acts, obs=obs)          # two pool steps per replay: the design buffers are back where they started
        out[tag + ("_graph_obs_us" if obs else "_graph_us")] = timed(g.replay, 30) / 2
    for k, e in enumerate(p.envs):
        out.setdefault("per_class_us", {})[f"{e.N}n_{e.B}envs"] = timed(lambda: e.step(*acts[k]))
    del p
print(json.dumps(out))
