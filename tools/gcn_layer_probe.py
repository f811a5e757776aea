#!/usr/bin/env python3
"""Diagnostic: the fused GCN layer kernel (truss_gcn_layer, MFMA) against what it replaces -- library GEMM (torch F.linear, padded to
224 output channels as round 2's actor_infer did) + truss_gcn_aggregate -- and against plain PyTorch, per truss size class at ~10^5
node rows, for the actors' layer shapes.  Prints us per layer, TFLOP/s of the 2 M K C product, and the max relative deviation."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import numpy as np
import torch
import truss_mi355 as tm
from truss_mi355 import marl

lib = tm.load(sys.argv[1] if len(sys.argv) > 1 else None)
dev = "cuda"
rows = int(os.environ.get("PROBE_ROWS", "98304"))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(reps):
        fn()
    a1.record()
    torch.cuda.synchronize()
    return a0.elapsed_time(a1) * 1e3 / reps


res = []
for nx in (8, 16, 32, 64, 128):
    topo = tm.TrussTopology.grid(nx)
    N = topo.N
    B = rows // N
    nbr = torch.tensor(topo.neighbor_table(), device=dev)
    pat = np.zeros((N, N), bool)
    tab = topo.neighbor_table()
    for i in range(N):
        pat[i, tab[i][tab[i] >= 0]] = True
    adj = torch.rand(B, N, N, device=dev) * torch.tensor(pat, device=dev)
    for K, C, act in ((200, 200, "relu"), (13, 200, "relu"), (200, 3, "sigmoid")):
        x = torch.randn(B, N, K, device=dev)
        w = torch.randn(C, K, device=dev) / K ** 0.5
        bias = torch.randn(C, device=dev)
        CP = 224 if C > 16 else C
        wp = torch.zeros(CP, K, device=dev); wp[:C] = w
        bp = torch.zeros(CP, device=dev); bp[:C] = bias
        out = torch.empty(B, N, C, device=dev)
        ws = marl.split_weights(lib, w) if (C > 32 and K % 4 == 0) else None
        t_f32 = timeit(lambda: marl.gcn_layer(lib, x, adj, w, bias, act, nbr, out, precision="f32"))
        t_new = timeit(lambda: marl.gcn_layer(lib, x, adj, w, bias, act, nbr, out, w_split=ws))
        t_old = timeit(lambda: marl.gcn_aggregate(lib, adj, torch.nn.functional.linear(x, wp).contiguous(), bp, act, nbr if N > 32 else None))
        f = torch.relu if act == "relu" else torch.sigmoid
        t_torch = timeit(lambda: f(torch.matmul(adj, torch.nn.functional.linear(x, w)) + bias), reps=5)
        ref = f(torch.matmul(adj, torch.nn.functional.linear(x, w)) + bias)
        err = float((out - ref).abs().max() / ref.abs().max())
        flop = 2.0 * B * N * K * C
        res.append({"nodes": N, "graphs": B, "K": K, "C": C, "fused_mfma_us": round(t_new, 1), "fused_f32_mfma_us": round(t_f32, 1), "gemm_plus_aggregate_us": round(t_old, 1),
                    "torch_us": round(t_torch, 1), "fused_TFLOPs": round(flop / t_new / 1e6, 1), "max_rel_dev": err})
        print(json.dumps(res[-1]), flush=True)
