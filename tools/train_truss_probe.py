"""Diagnostic: step / fused step / persistent rollout times on the reference's training truss family
(test/ trusses and master_DDPG_truss2D_MO.py:807-825: 12 nodes / 26 elements -- E % 4 != 0) beside the bench truss."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]
import torch
import truss_mi355 as tm
from truss_mi355 import synthetic, distributed

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = tm.load()
out = {}
a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a0.record()
    for _ in range(n):
        fn()
    a1.record()
    torch.cuda.synchronize()
    return a0.elapsed_time(a1) * 1e3 / n


for name, topo in (("train_12n_26e", tm.TrussTopology.grid(6)), ("grid_10n_21e", tm.TrussTopology.grid(5)), ("roof_16n_36e", tm.TrussTopology.grid(8)),
                   ("bench_32n_80e", synthetic.bench_topology(16, 4))):
    env, G, T, _ = distributed.make_rank_env(topo, B, 0, device=torch.device("cuda", 0), lib=lib, seed=1234, n_action_sets=8)
    ag0, at0 = G[0].contiguous(), T[0].contiguous()
    full = env.obs_buffers()
    r = {"fused_obs": bool(env.fused_obs), "persistent_rollout": bool(env.persistent_rollout)}
    r["step_us"] = round(timed(lambda: env.step(ag0, at0), 200), 2)
    r["step_obs_us"] = round(timed(lambda: env.step(ag0, at0, obs=full), 200), 2)
    r["rollout_us_per_step"] = round(timed(lambda: env.rollout(G, T, 64), 10) / 64, 2)
    out[name] = r
print(json.dumps({"envs": B, "topologies": out}))
