#!/usr/bin/env python3
"""Diagnostic: how are the workgroups of one truss_step launch spread over time?  Uses the stamped
build (make -C mop-truss-marl_amd/csrc diag).  Prints per-workgroup duration statistics (shader
cycles), the first-start -> last-end span in wall time, and the effective shader clock."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mop-truss-marl_amd"))
import numpy as np
import torch
import truss_mi355 as tm
from truss_mi355 import synthetic


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    path = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "mop-truss-marl_amd", "csrc", "libtruss_mi355_diag.so")
    lib = tm.load(path)
    topo = synthetic.bench_topology(16, 4)
    batch = synthetic.random_batch(topo, B, 1)
    env = tm.BatchedTruss(topo, B, lib=lib)
    env.set_constants(batch["x"], batch["target"], batch["y_max"], batch["d_min"], batch["max_def"], batch["load_x"],
                      batch["load_y"], batch["is_roof"])
    env.set_design(batch["y"], batch["sec"])
    env.analyze(set_normalisers=True)
    ag, at = synthetic.random_actions(2, B, topo.N, 2)
    G, T = torch.tensor(ag, device=env.device), torch.tensor(at, device=env.device)
    env.rollout(G, T, 50)
    torch.cuda.synchronize()
    nb = B * topo.solver_info(lib)["lanes_per_env"] // 64
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.rollout(G, T, 200); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    buf = (ctypes.c_ulonglong * (6 * nb))()
    lib.dll.truss_debug_span.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.dll.truss_debug_span(buf, nb) == 0
    a = np.array(list(buf), dtype=np.float64).reshape(nb, 6)
    cyc = a[:, 2] - a[:, 0]
    wall = (a[:, 3] - a[:, 1]) * 10.0   # ns (100 MHz)
    span_ns = (a[:, 3].max() - a[:, 1].min()) * 10.0
    start_ns = (a[:, 1] - a[:, 1].min()) * 10.0
    print(f"{os.path.basename(path)}: B={B} workgroups={nb}  step (events, 200 chained launches) {us:.2f} us")
    print(f"  workgroup duration: cycles min/median/max {cyc.min():.0f} / {np.median(cyc):.0f} / {cyc.max():.0f}; "
          f"wall ns min/median/max {wall.min():.0f} / {np.median(wall):.0f} / {wall.max():.0f}")
    print(f"  effective shader clock (median cycles / median wall): {np.median(cyc) / np.median(wall):.2f} GHz")
    print(f"  first start -> last end: {span_ns / 1e3:.2f} us; start spread: median {np.median(start_ns):.0f} ns, max {start_ns.max():.0f} ns")
    for x in range(8):
        sel = np.arange(nb) % 8 == x
        print(f"  workgroups with id%8=={x}: median {np.median(cyc[sel]):.0f} cyc, {np.median(wall[sel]):.0f} ns, last end +{(a[sel, 3].max() - a[:, 1].min()) * 10 / 1e3:.2f} us")


if __name__ == "__main__":
    main()
