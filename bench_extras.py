"""bench.py --extras: secondary measurements on one GPU (rank 0): the reference-exact E = 76 topology, the
stand-alone observation kernel and the two-launch state-emitting step (what TRUSS_F_EMIT_OBS replaced), BASELINE
configs[1]'s small bridge, configs[4]'s large size classes and mixed sweep, and the batched MADDPG rollout of configs[2]."""
import torch

NUM_X = 16
N_ACTION_SETS = 8
HBM_PEAK_GBS = 8000.0


def _env(tm, synthetic, topo, B, dev, lib, seed):
    b = synthetic.random_batch(topo, B, seed=seed)
    e = tm.BatchedTruss(topo, B, device=dev, lib=lib)
    e.set_constants(b["x"], b["target"], b["y_max"], b["d_min"], b["max_def"], b["load_x"], b["load_y"], b["is_roof"])
    e.set_design(b["y"], b["sec"])
    e.analyze(set_normalisers=True)
    return e


def run(tm, synthetic, lib, dev, topo, env, G, T, args, B):
    extras = {}
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # reference-exact topology (E = 76)
    e76 = _env(tm, synthetic, tm.TrussTopology.grid(NUM_X), B, dev, lib, 99)
    e76.rollout(G, T, args.warmup)
    torch.cuda.synchronize()
    a0.record(); e76.rollout(G, T, args.steps); a1.record(); torch.cuda.synchronize()
    extras["e76_env_steps_per_s"] = B * args.steps / (a0.elapsed_time(a1) * 1e-3)
    # stand-alone observation kernel (reset path / topologies outside the fused writer)
    env.observe(); torch.cuda.synchronize()
    a0.record()
    for _ in range(50):
        env.observe()
    a1.record(); torch.cuda.synchronize()
    obs_us = a0.elapsed_time(a1) * 1e3 / 50
    obs_bytes = 4 * (13 * topo.N + 3 * topo.N ** 2 + 12 * topo.N + 21 * topo.E)
    extras["obs_kernel_us"] = obs_us
    extras["obs_GBps"] = B * obs_bytes / (obs_us * 1e-6) / 1e9
    # what the fused step replaced: env.step + env.observe, two launches and a host hop per step
    ag0, at0 = G[0].contiguous(), T[0].contiguous()
    env.step(ag0, at0); env.observe(); torch.cuda.synchronize()
    a0.record()
    for _ in range(100):
        env.step(ag0, at0)
        env.observe()
    a1.record(); torch.cuda.synchronize()
    extras["step_then_obs_two_launches_us"] = a0.elapsed_time(a1) * 1e3 / 100
    # BASELINE configs[1] topology: small bridge, 16 nodes / 36 elements (reference-exact grid)
    t36 = tm.TrussTopology.grid(8)
    e36 = _env(tm, synthetic, t36, B, dev, lib, 7)
    g36, a36 = synthetic.random_actions(N_ACTION_SETS, B, t36.N, 5)
    g36, a36 = torch.tensor(g36, device=dev), torch.tensor(a36, device=dev)
    e36.rollout(g36, a36, args.warmup); torch.cuda.synchronize()
    a0.record(); e36.rollout(g36, a36, args.steps); a1.record(); torch.cuda.synchronize()
    extras["small_bridge_16n36e_env_steps_per_s"] = B * args.steps / (a0.elapsed_time(a1) * 1e-3)
    extras["small_bridge_lanes_per_env"] = t36.solver_info(lib)["lanes_per_env"]
    ob36 = e36.obs_buffers()
    e36.step(g36[0], a36[0], obs=ob36); torch.cuda.synchronize()
    a0.record()
    for _ in range(100):
        e36.step(g36[0], a36[0], obs=ob36)
    a1.record(); torch.cuda.synchronize()
    extras["small_bridge_step_plus_obs_us"] = a0.elapsed_time(a1) * 1e3 / 100
    extras["small_bridge_fused_obs_one_launch"] = bool(e36.fused_obs)
    # BASELINE configs[4] sizes: 128- and 256-node trusses (316 / 636 elements) on the 32- / 64-lane kernels
    for nx_big, b_big in ((64, 2048), (128, 1024)):
        tb = tm.TrussTopology.grid(nx_big)
        eb = _env(tm, synthetic, tb, b_big, dev, lib, nx_big)
        gb, ab = synthetic.random_actions(2, b_big, tb.N, 5)
        gb, ab = torch.tensor(gb, device=dev), torch.tensor(ab, device=dev)
        eb.rollout(gb, ab, 10); torch.cuda.synchronize()
        a0.record(); eb.rollout(gb, ab, 50); a1.record(); torch.cuda.synchronize()
        us = a0.elapsed_time(a1) * 1e3 / 50
        extras[f"large_{tb.N}n_{tb.E}e_{b_big}envs"] = {"us_per_step": us, "env_steps_per_s": b_big / (us * 1e-6),
                                                           "lanes_per_env": tb.solver_info(lib)["lanes_per_env"],
                                                           "nonpositive_pivots": int(eb.status.sum().item())}
    # BASELINE configs[4]: the mixed 32-256-node pool on one GPU (one BatchedTruss per size class, one stream per class)
    from truss_mi355 import pool
    classes = pool.grid_classes([16, 32, 64, 128], [2048, 1024, 512, 256])
    for tag, streams in (("mixed_pool_32_64_128_256_nodes", True), ("mixed_pool_same_one_stream", False)):
        p = pool.MixedTrussPool(classes, bucket_envs=64, device=dev, lib=lib, streams=streams)
        batches, acts = [], []
        for k, e in enumerate(p.envs):
            b = synthetic.random_batch(e.topo, e.B, seed=30 + k)
            batches.append(b)
            ag, at = synthetic.random_actions(1, e.B, e.N, 60 + k)
            acts.append((torch.tensor(ag[0], device=dev), torch.tensor(at[0], device=dev)))
        p.set_constants(batches); p.set_design(batches); p.analyze(set_normalisers=True)
        for _ in range(5):
            p.step(acts)
        torch.cuda.synchronize()
        a0.record()
        for _ in range(50):
            p.step(acts)
        a1.record(); torch.cuda.synchronize()
        us = a0.elapsed_time(a1) * 1e3 / 50
        extras[tag] = {"envs": p.sizes, "us_per_pool_step": us, "env_steps_per_s": p.n_envs / (us * 1e-6), "nonpositive_pivots": int(p.status.sum().item())}
        for _ in range(5):
            p.step(acts, obs=True)
        torch.cuda.synchronize()
        a0.record()
        for _ in range(30):
            p.step(acts, obs=True)
        a1.record(); torch.cuda.synchronize()
        extras[tag]["us_per_pool_step_with_obs"] = a0.elapsed_time(a1) * 1e3 / 30
        del p
    # (BASELINE configs[2]-[4] -- the batched MADDPG rollout, the mixed sweep, large_bridge -- are in bench.py's default `configs` object)
    return extras
