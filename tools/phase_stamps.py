#!/usr/bin/env python3
"""Diagnostic: where does one workgroup of truss_step_kernel spend its cycles?
Uses the stamped build (make -C mop-truss-marl_amd/csrc diag).  Shares only -- the stamped build's
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

NAMES = ["stage (HBM->LDS)", "decode", "sizing", "elements+assembly", "factorisation", "back-substitution",
         "post_elements", "post_nodes", "finish"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    path = os.path.join(ROOT, "mop-truss-marl_amd", "csrc", "libtruss_mi355_diag.so")
    if len(sys.argv) > 2:
        path = os.path.abspath(sys.argv[2])   # e.g. an ablated build from tools/ablate.sh
    lib = tm.load(path)
    lib.dll.truss_debug_stamps.argtypes = [ctypes.c_void_p]
    nx = int(os.environ.get("STAMP_NX", "0"))                   # e.g. STAMP_NX=128 with libtruss_mi355_diag64.so: the 256-node grid
    cases = (("32n/80e", synthetic.bench_topology(16, 4)),) if nx == 0 else ((f"grid {2 * nx}n", tm.TrussTopology.grid(nx)),)
    for label, topo in cases:
        batch = synthetic.random_batch(topo, B, 1)
        env = tm.BatchedTruss(topo, B, lib=lib)
        env.set_constants(batch["x"], batch["target"], batch["y_max"], batch["d_min"], batch["max_def"],
                          batch["load_x"], batch["load_y"], batch["is_roof"])
        env.set_design(batch["y"], batch["sec"])
        env.analyze(set_normalisers=True)
        ag, at = synthetic.random_actions(2, B, topo.N, 2)
        G, T = torch.tensor(ag, device=env.device), torch.tensor(at, device=env.device)
        env.rollout(G, T, 20)
        torch.cuda.synchronize()
        acc = np.zeros(9)
        extra = np.zeros(9)
        chained = int(os.environ.get("STAMP_STEPS", "1"))     # > 1: the LAST step of a persistent rollout (steady state)
        for _ in range(10):
            env.rollout(G, T, chained)
            torch.cuda.synchronize()
            st = (ctypes.c_ulonglong * 16)()
            lib.dll.truss_debug_stamps(st)
            s = np.array(list(st)[:10], dtype=np.float64)
            acc += np.diff(s)
            sub = np.array(list(st), dtype=np.float64)
            extra = extra + np.array([sub[10] - sub[3], sub[11] - sub[10], sub[4] - sub[11], sub[12] - sub[8], sub[9] - sub[12],
                                      sub[13] - sub[4], sub[5] - sub[13], sub[14] - sub[5], sub[6] - sub[14]])
        acc /= 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(G, T, 200); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        if len(sys.argv) > 2:
            print(f"{os.path.basename(path):28s} total {acc.sum():7.0f} cyc, launch {us:6.2f} us -> {acc.sum() / us / 1e3:4.2f} GHz if the "
                  f"workgroup spanned the launch | " + " ".join(f"{c:6.0f}" for c in acc))
            continue
        print(f"== {label} B={B}  {topo.solver_info(lib)['lanes_per_env']} lanes/env, total {acc.sum():.0f} cycles; stamped launch {us:.2f} us")
        for n, c in zip(NAMES, acc):
            print(f"   {n:22s} {c:9.0f} cyc  {100 * c / acc.sum():5.1f} %")
        for n, c in zip(["elements (pass 1+2)", "assemble_nodes", "scratch_init", "finish (point/obj/status)", "store rows",
                          "factor: clean blocks", "factor: merge blocks + check", "backsub: hand-over blocks", "backsub: clean blocks"], extra / 10):
            print(f"      - {n:20s} {c:9.0f} cyc")


if __name__ == "__main__":
    main()
