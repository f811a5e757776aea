"""Diagnostic: time the fused step (TRUSS_F_EMIT_OBS) with subsets of the observation tensors."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import torch
import truss_mi355 as tm
from truss_mi355 import synthetic, distributed

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = tm.load(sys.argv[2] if len(sys.argv) > 2 else None)
topo = synthetic.bench_topology(16, 4)
env, G, T, _ = distributed.make_rank_env(topo, B, 0, device=torch.device("cuda", 0), lib=lib, seed=1234, n_action_sets=8)
ag0, at0 = G[0].contiguous(), T[0].contiguous()
full = env.obs_buffers()
sets = {"none": None, "all": full, "rows_only": {k: full[k] for k in ("x_n", "nN_x_n", "nN_x_e")},
        "matrices_only": {k: full[k] for k in ("A_s", "A_n_ts", "A_n_cs")}, "nxe_only": {"nN_x_e": full["nN_x_e"]},
        "A_s_only": {"A_s": full["A_s"]}}
res = {}
a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, ob in sets.items():
    for _ in range(20):
        env.step(ag0, at0, obs=ob)
    torch.cuda.synchronize()
    a0.record()
    for _ in range(200):
        env.step(ag0, at0, obs=ob)
    a1.record(); torch.cuda.synchronize()
    res[name] = round(a0.elapsed_time(a1) * 1e3 / 200, 2)
print(json.dumps({"envs": B, "fused": bool(env.fused_obs), "us_per_step": res}))
