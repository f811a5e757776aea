#!/usr/bin/env python3
"""Step-kernel time per launch for every (lanes per env, envs per GPU) pair: what a strong-scaling shard sees
(512 envs per GPU at 4096 global envs / 8 GPUs, 1024 at 8192) and what large batches see.
One process per lanes-per-env setting (the variant is chosen at topology creation from TRUSS_LANES)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
    import torch
    import truss_mi355 as tm
    from truss_mi355 import synthetic, distributed
    lib = tm.load()
    topo = synthetic.bench_topology(16, 4)
    res = {}
    for B in [int(b) for b in sys.argv[2].split(",")]:
        env, G, T, _ = distributed.make_rank_env(topo, B, 0, device=torch.device("cuda", 0), lib=lib, seed=1234, n_action_sets=8)
        env.rollout(G, T, 40)
        torch.cuda.synchronize()
        reps = []
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            a0.record(); env.rollout(G, T, 200); a1.record(); torch.cuda.synchronize()
            reps.append(a0.elapsed_time(a1) * 1e3 / 200)
        reps.sort()
        res[B] = {"us_per_step": round(reps[2], 2), "env_steps_per_s": round(B / (reps[2] * 1e-6))}
        del env, G, T
    info = topo.solver_info(lib)
    print(json.dumps({"lanes_per_env": info["lanes_per_env"], "results": res}))
    sys.exit(0)

sizes = "512,1024,2048,4096,8192,16384,65536"
out = {}
for lanes in (8, 16, 32):
    env = dict(os.environ, TRUSS_LANES=str(lanes), TRUSS_WLANES="8", TRUSS_RPL="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", sizes], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    out[f"lanes_{lanes}"] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
print(json.dumps({"workload": "32 nodes / 80 elements, FEM-only step, one MI355X", "by_lanes_per_env": out}, indent=1))
