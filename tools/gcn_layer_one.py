#!/usr/bin/env python3
"""Diagnostic: N launches of the fused GCN layer kernel at one shape (for rocprofv3 --pmc / --kernel-trace):  nodes K C [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import numpy as np, torch
import truss_mi355 as tm
from truss_mi355 import marl
nx, K, C = int(sys.argv[1]) // 2, int(sys.argv[2]), int(sys.argv[3])
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 98304
lib = tm.load()
topo = tm.TrussTopology.grid(nx)
N, B = topo.N, rows // topo.N
tab = topo.neighbor_table()
pat = np.zeros((N, N), bool)
for i in range(N):
    pat[i, tab[i][tab[i] >= 0]] = True
adj = torch.rand(B, N, N, device="cuda") * torch.tensor(pat, device="cuda")
x, w, bias = torch.randn(B, N, K, device="cuda"), torch.randn(C, K, device="cuda") / K ** 0.5, torch.randn(C, device="cuda")
out = torch.empty(B, N, C, device="cuda")
nbr = torch.tensor(tab, device="cuda")
for _ in range(20):
    marl.gcn_layer(lib, x, adj, w, bias, "relu", nbr, out)
torch.cuda.synchronize()
