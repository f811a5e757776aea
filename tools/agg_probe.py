"""Diagnostic: act(A @ H + b) per graph -- the fused kernel (truss_gcn_aggregate) against torch.matmul + add + relu, by graph size."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import torch
import truss_mi355 as tm
from truss_mi355 import marl

lib = tm.load()
dev = torch.device("cuda", 0)
a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a0.record()
    for _ in range(n):
        fn()
    a1.record(); torch.cuda.synchronize()
    return round(a0.elapsed_time(a1) * 1e3 / n, 1)


out = {}
for N, B in ((16, 10000), (20, 9000), (24, 6000), (32, 4096), (48, 2048), (64, 1024), (128, 512), (256, 256)):
    C = 224
    h = torch.rand(B, N, C, device=dev)
    bias = torch.rand(C, device=dev)
    for tag, adj in (("shared", torch.softmax(torch.randn(N, N, device=dev), -1)), ("per_graph", torch.softmax(torch.randn(B, N, N, device=dev), -1))):
        r = {"torch_us": timed(lambda: torch.relu(torch.matmul(adj, h) + bias))}
        if N <= 64:
            import truss_mi355.ops as ops
            o = torch.empty_like(h)
            r["fused_us"] = timed(lambda: ops.call(ops.namespace().gcn_aggregate, ops.bind(lib), ops.stream_of(dev), adj.contiguous(), h, bias, o, 1))
        if tag == "shared" or True:
            if N % 4:
                r["MB"] = round(2 * h.numel() * 4 / 1e6, 1); out[f"N{N}_B{B}_{tag}"] = r; continue
            topo = tm.TrussTopology.grid(N // 2)
            nbr = torch.tensor(topo.neighbor_table(), device=dev)
            import truss_mi355.ops as ops
            o2 = torch.empty_like(h)
            r["sparse_us"] = timed(lambda: ops.call(ops.namespace().gcn_aggregate_sparse, ops.bind(lib), ops.stream_of(dev), adj.contiguous(), nbr, h, bias, o2, 1))
        r["MB"] = round(2 * h.numel() * 4 / 1e6, 1)
        out[f"N{N}_B{B}_{tag}"] = r
print(json.dumps(out))
