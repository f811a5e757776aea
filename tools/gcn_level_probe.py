#!/usr/bin/env python3
"""Diagnostic: the fused level kernel (truss_gcn_level) alone -- time per launch (HIP events, back-to-back launches) for the level
shapes of the MADDPG update, and, from a -DTRUSS_GCN_STAMPS build (tools/abbuild.sh lvst -DTRUSS_GCN_STAMPS), the shader-clock
stamps of workgroup (0, 0, 0):   tools/gcn_level_probe.py [mop-truss-marl_amd/csrc/abl/libtruss_lvst.so]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import torch
import truss_mi355 as tm
from truss_mi355 import marl, ops

lib = tm.load(os.path.abspath(sys.argv[1])) if len(sys.argv) > 1 else tm.load()
dev = "cuda"
r = lambda *s: torch.rand(*s, device=dev)


def level(n_layers, B, N, K, C, want_x, tag):
    X = [r(B, N, K) for _ in range(n_layers)]
    A = [torch.softmax(torch.randn(B, N, N, device=dev), -1) for _ in range(n_layers)]
    W = [torch.randn(C, K, device=dev) / 14 for _ in range(n_layers)]
    Bs = [r(C) for _ in range(n_layers)]
    O = [torch.empty(B, N, C, device=dev) for _ in range(n_layers)]
    XA = [torch.empty(B * N, K, device=dev) for _ in range(n_layers)] if want_x else []
    act = [1] * n_layers
    call = lambda: ops.call(ops.namespace().gcn_level, ops.bind(lib), ops.stream_of(torch.device(dev)), X, A, [], W, Bs, O, XA, act)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    ref = torch.relu(torch.matmul(A[-1], X[-1] @ W[-1].t()) + Bs[-1])
    err = float((O[-1] - ref).abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        call()
    e1.record()
    torch.cuda.synchronize()
    line = f"{tag:44s} {e0.elapsed_time(e1) * 10:7.1f} us per launch   max err {err:.1e}"
    if hasattr(lib.dll, "truss_debug_level_stamps"):
        st = (ctypes.c_ulonglong * 8)()
        lib.dll.truss_debug_level_stamps.argtypes = [ctypes.c_void_p]
        lib.dll.truss_debug_level_stamps(st)
        d = [st[i + 1] - st[i] for i in range(6)]
        line += f"   cycles: setup {d[0]}  first fetch+stash {d[1]}  K loop {d[2]}  H to LDS {d[3]}  gather {d[4]}  act+store {d[5]}"
    print(line)


level(1, 32, 16, 200, 200, False, "1 layer, 512 rows, K 200, C 200")
level(1, 32, 16, 16, 200, False, "1 layer, 512 rows, K 16, C 200")
level(1, 32, 16, 13, 200, False, "1 layer, 512 rows, K 13, C 200")
level(11, 32, 16, 200, 200, False, "11 layers (critic level 2)")
level(11, 32, 16, 200, 200, True, "11 layers (critic level 2) + X'")
level(24, 32, 16, 200, 200, True, "24 layers + X'")
level(24, 96, 16, 200, 200, False, "24 layers, 1536 rows (target critics)")
level(2, 32, 16, 200, 3, True, "2 heads (C 3) + X'")
