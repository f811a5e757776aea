"""One-off robustness sweep on the GPU (not part of the test-suite): random irregular topologies, many seeds, native vs oracle
for chained steps and for the observation tensors (fused where the topology allows).  Prints the seeds that fail.
  python tools/stress_parity.py [first_seed] [n_seeds]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), os.path.join(ROOT, "tests"), ROOT]
import truss_mi355 as tm
import parity_common as pc

lib = tm.load()
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad, kinds = [], {}
for seed in range(s0, s0 + n):
    try:
        topo = pc.irregular_topology(seed)
    except Exception as e:  # noqa: BLE001
        bad.append((seed, "topology", repr(e)[:120]))
        continue
    try:
        info = topo.solver_info(lib)
        kinds[(info["lanes_per_env"], info["half_bandwidth"])] = kinds.get((info["lanes_per_env"], info["half_bandwidth"]), 0) + 1
        # 1e-8: random designs include nearly singular ones (seed 1179, env 37: displacements of 270 m) where an LU (oracle) and an
        # LDL^T (kernel, emulator alike) differ by cond(K) x 1e-16 ~ 1e-9
        pc.run_random_rollout(lib, 0, 0, 48, 3, seed=seed, topo=topo, tight=1e-8)
        pc.run_obs_random(lib, 0, 0, 33, seed=seed, fused=True, topo=topo)
    except tm.TrussError as e:
        if "UNSUPPORTED" in str(e).upper() or "unsupported" in str(e):
            kinds["refused"] = kinds.get("refused", 0) + 1
        else:
            bad.append((seed, "TrussError", str(e)[:160]))
    except Exception as e:  # noqa: BLE001
        bad.append((seed, type(e).__name__, (str(e) or traceback.format_exc())[:200]))
    if (seed - s0) % 50 == 49:
        print("...", seed, "failures so far:", len(bad), flush=True)
print("variants (lanes, half-bandwidth):", kinds)
print("FAILED:", bad if bad else "none")

# second sweep: the reference's grid family at every size 3..40 bays, with 0..3 braces removed (every N mod 4 and E mod 4:
# staging fast / generic path, fused writer with 16- / 8- / 4-byte nN_x_e chunks, persistent rollout or chained launches)
import numpy as np
import torch
from truss_mi355 import synthetic
bad2, paths = [], {}
for nx in range(3, 41):
    for k in range(0, min(4, nx)):           # at most the nx - 1 '/' braces can go (every bay keeps its '\\' brace)
        try:
            topo = pc.pruned_grid(nx, k)
            env = pc.run_obs_random(lib, 0, 0, 19, seed=1000 + 4 * nx + k, fused=True, topo=topo)
            key = (bool(env.fused_obs), bool(env.persistent_rollout))
            paths[key] = paths.get(key, 0) + 1
            # chained rollout (one launch where persistent) == the same steps one by one, bit for bit
            b = synthetic.random_batch(topo, 21, 7)
            e1, e2 = pc.make_env(lib, topo, b), pc.make_env(lib, topo, b)
            for e in (e1, e2):
                e.analyze(set_normalisers=True)
            ag, at = synthetic.random_actions(3, 21, topo.N, 9)
            G, T = torch.tensor(ag, device=e1.device), torch.tensor(at, device=e1.device)
            e1.rollout(G, T, 5)
            for s in range(5):
                e2.step(G[s % 3], T[s % 3])
            for name in ("y", "sec", "point", "disp", "q0", "sr"):
                assert torch.equal(getattr(e1, name), getattr(e2, name)), name
        except Exception as e:  # noqa: BLE001
            bad2.append((nx, k, type(e).__name__, (str(e) or traceback.format_exc())[:200]))
print("(fused writer, persistent rollout) -> topologies:", paths)
print("FAILED (grid sweep):", bad2 if bad2 else "none")
