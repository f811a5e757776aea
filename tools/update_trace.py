#!/usr/bin/env python3
"""The MADDPG update as the batched rollout runs it (one hipGraph replay per update, batch 32, small_roof), for rocprofv3:
   python3 tools/update_trace.py run R          -- warm up (captures the graph), then R timed replays; prints one JSON line
   python3 tools/update_trace.py reduce DIR     -- DIR holds two --kernel-trace runs (run_a: R = 20, run_b: R = 120) made by
                                                   tools/update_trace.sh; launches / GPU time PER UPDATE = difference / 100
The difference of two runs removes everything that is not a replay (engine set-up, warm-up game steps, the capture itself)."""
import contextlib
import glob
import io
import json
import os
import sqlite3
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "mop-truss-marl_amd"), ROOT]


def run(R):
    import torch
    import truss_mi355 as tm
    from truss_mi355 import marl, synthetic
    import master_DDPG_truss2D_MO as M
    import truss2D_RL as RL
    topo = tm.TrussTopology.grid(8)
    torch.manual_seed(0)
    rl = RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, M.a_nn, M.c_nn, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma, device="cuda")
    eng = marl.BatchedMARL(topo, 512, rl, max_front=20, device="cuda", replay_capacity=8192, batch_size=32)
    b = synthetic.random_batch(topo, 512, 3)
    eng.reset(b["x"], b["target"], b["y_max"], b["d_min"], b["max_def"], b["load_x"], b["load_y"], b["is_roof"], b["y"], b["sec"])
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(3):
            eng.game_step_all(train=True)
    assert eng._tg is not None, "the update was not captured"
    S, NS, a_geo, a_topo, Rw = eng.replay.sample(32, eng.gen)
    args = (eng._net_state(S), [eng._net_state(ns) for ns in NS], [(a_geo[:, k].contiguous(), a_topo[:, k].contiguous()) for k in range(3)], Rw)
    eng._train(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        eng._train(*args)
    torch.cuda.synchronize()
    print(json.dumps({"replays": R, "ms_per_update": (time.perf_counter() - t0) / R * 1e3}))


def kernels(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        c = sqlite3.connect(f)
        for name, n, tot in c.execute("select name, count(*), sum(duration) from kernels group by name"):
            k = out.setdefault(name, [0, 0.0])
            k[0] += n
            k[1] += tot / 1e3
    return out


def reduce(d):
    a, b = kernels(os.path.join(d, "run_a")), kernels(os.path.join(d, "run_b"))
    ja, jb = (json.loads([l for l in open(os.path.join(d, f"{r}.json")) if l.startswith("{")][-1]) for r in ("run_a", "run_b"))
    dr = jb["replays"] - ja["replays"]
    per = {k: ((b[k][0] - a.get(k, [0, 0])[0]) / dr, (b[k][1] - a.get(k, [0, 0])[1]) / dr) for k in b}
    per = {k: v for k, v in per.items() if v[0] > 0}
    top = sorted(per.items(), key=lambda kv: -kv[1][1])[:25]
    res = {"what": "MADDPG update (batch 32, small_roof) as one hipGraph replay, rocprofv3 --kernel-trace; per update = "
                   f"(run of {jb['replays']} replays - run of {ja['replays']} replays) / {dr}",
           "kernel_launches_per_update": sum(v[0] for v in per.values()),
           "gpu_kernel_time_us_per_update": sum(v[1] for v in per.values()),
           "host_timed_ms_per_update_under_trace": jb["ms_per_update"],
           "distinct_kernels": len(per),
           "top_kernels_by_time": [{"name": k[:160], "launches_per_update": round(v[0], 2), "us_per_update": round(v[1], 1)} for k, v in top]}
    shapes = {}
    for f in glob.glob(os.path.join(d, "run_b", "**", "*.db"), recursive=True):
        c = sqlite3.connect(f)
        for gx, gy, gz, wx, dur in c.execute("select grid_x, grid_y, grid_z, workgroup_x, duration from kernels where name like '%gcn_level%'"):
            shapes.setdefault(f"{gx // wx} x {gy} x {gz} workgroups", []).append(dur / 1e3)
    res["level_kernel_by_grid"] = {k: {"launches": len(v), "median_us": sorted(v)[len(v) // 2], "min_us": min(v)} for k, v in sorted(shapes.items())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        reduce(sys.argv[2])
