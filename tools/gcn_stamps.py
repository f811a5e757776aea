#!/usr/bin/env python3
"""Diagnostic: cycles of the sections of the fused GCN layer kernel's slab loop (block 0, wave 0) from a -DTRUSS_GCN_STAMPS build:
   tools/abbuild.sh gcnst -DTRUSS_GCN_STAMPS ; tools/gcn_stamps.py mop-truss-marl_amd/csrc/abl/libtruss_gcnst.so [rows]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import numpy as np, torch
import truss_mi355 as tm
from truss_mi355 import marl
lib = tm.load(os.path.abspath(sys.argv[1]))
for rows in ([int(sys.argv[2])] if len(sys.argv) > 2 else [6144, 98304]):
    topo = tm.TrussTopology.grid(16)
    N, B = topo.N, rows // topo.N
    tab = topo.neighbor_table()
    pat = np.zeros((N, N), bool)
    for i in range(N):
        pat[i, tab[i][tab[i] >= 0]] = True
    adj = torch.rand(B, N, N, device="cuda") * torch.tensor(pat, device="cuda")
    x, w, bias = torch.randn(B, N, 200, device="cuda"), torch.randn(200, 200, device="cuda") / 14, torch.randn(200, device="cuda")
    out, nbr = torch.empty(B, N, 200, device="cuda"), torch.tensor(tab, device="cuda")
    prec = os.environ.get("GCN_PRECISION", "bf16x3")
    for _ in range(5):
        marl.gcn_layer(lib, x, adj, w, bias, "relu", nbr, out, precision=prec)
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 8)()
    lib.dll.truss_debug_gcn_stamps.argtypes = [ctypes.c_void_p]
    lib.dll.truss_debug_gcn_stamps(st)
    n = max(1, st[5])
    print(f"{prec} rows {rows}: per slab (cycles): barrier-in {st[0] / n:.0f}  body(fetch+operands+mfma+gather) {st[1] / n:.0f}  barrier-out {st[2] / n:.0f}  "
          f"stash(+drain) {st[3] / n:.0f};  whole loop {st[4]} = {st[4] / n:.0f} per slab, {n} slabs")
