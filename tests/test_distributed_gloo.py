"""N>1 path on CPU: two gloo ranks, each stepping its own shard through the lane emulator; checks that
(a) the shards are what single-process runs of the same seeds produce, (b) the barrier / MAX-reduce
plumbing bench.py relies on works, (c) no rank depends on another rank's data."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import parity_common as pc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, lib_path, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import truss_mi355 as tm
    from truss_mi355 import synthetic, distributed
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = tm.load(lib_path)
    topo = synthetic.bench_topology(16, 4)
    env, G, T, _ = distributed.make_rank_env(topo, 24, rank, lib=lib, seed=1234, n_action_sets=3)
    elapsed, dev_ms = distributed.timed_rollout(env, G, T, steps=3, warmup=1, dist=dist)
    total = distributed.global_checksum(env, dist)
    mine = distributed.global_checksum(env, None)
    gathered = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.tensor(mine))
    if rank == 0:
        out.put((elapsed, dev_ms, total.tolist(), [g.tolist() for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_rollout():
    lib_path = pc.build_emu()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lib_path, out)) for r in range(2)]
    for p in procs:
        p.start()
    elapsed, dev_ms, total, per_rank = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert elapsed > 0 and dev_ms is None
    # single-process reference of each shard
    import truss_mi355 as tm
    from truss_mi355 import synthetic, distributed
    lib = tm.load(lib_path)
    topo = synthetic.bench_topology(16, 4)
    want = []
    for r in range(2):
        env, G, T, _ = distributed.make_rank_env(topo, 24, r, lib=lib, seed=1234, n_action_sets=3)
        env.rollout(G, T, 1)
        env.rollout(G, T, 3)
        want.append(distributed.global_checksum(env, None))
    np.testing.assert_array_equal(np.array(per_rank), np.array(want))
    np.testing.assert_allclose(np.array(total), np.array(want).sum(axis=0), rtol=1e-15)
    assert not np.array_equal(want[0], want[1])      # the shards really are different envs


def _marl_worker(rank, world, port, lib_path, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import contextlib
    import io
    import truss_mi355 as tm
    from truss_mi355 import synthetic, marl
    import master_DDPG_truss2D_MO as M
    import truss2D_RL as RL
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = tm.load(lib_path)
    topo = tm.TrussTopology.grid(4)
    torch.manual_seed(100 + rank)                          # DIFFERENT initial weights per rank: the broadcast must fix that
    rl = RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, 16, 8, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma, device="cpu", dist=dist)
    eng = marl.BatchedMARL(topo, 5, rl, max_front=6, lib=lib, device="cpu", replay_capacity=128, batch_size=4, seed=rank)
    b = synthetic.random_batch(topo, 5, 50 + rank)         # each rank plays its own envs
    eng.reset(b["x"], b["target"], b["y_max"], b["d_min"], b["max_def"], b["load_x"], b["load_y"], b["is_roof"], b["y"], b["sec"])
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(3):
            eng.game_step_all(train=True, explore=True)
    sig = torch.stack([torch.cat([p.detach().double().reshape(-1) for p in net.parameters()]).sum()
                       for ag in rl.agents for net in (ag.actor_model, ag.critic_model)])
    got = [torch.zeros_like(sig) for _ in range(world)]
    dist.all_gather(got, sig)
    if rank == 0:
        out.put(([g.tolist() for g in got], eng.replay.size, eng.env_steps))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_marl():
    """BASELINE config 4 in miniature: two ranks, each with its own envs / archive / replay, one MADDPG whose
    gradients are all-reduced (fused flat buffer) -> identical weights on both ranks after training."""
    lib_path = pc.build_emu()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_marl_worker, args=(r, 2, port, lib_path, out)) for r in range(2)]
    for p in procs:
        p.start()
    sigs, replay, steps = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert replay >= 4 and steps >= 15
    np.testing.assert_allclose(sigs[0], sigs[1], rtol=1e-12)


def _bench_line(extra_args, extra_env=None):
    """bench.py started the way the driver starts it for one GPU (`python bench.py --gpus N ...`, no launcher): it has
    to spawn its own ranks.  CPU rehearsal: gloo + the lane emulator."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(TRUSS_BENCH_BACKEND="gloo", TRUSS_BENCH_STRONG="12,21", **(extra_env or {}))
    r = subprocess.run([sys.executable, os.path.join(pc.ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--lib", pc.build_emu()] + extra_args, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]       # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks_weak_and_strong():
    o = _bench_line(["--gpus", "2", "--envs", "8"])
    assert o["n_gpus"] == 2 and o["scaling"] == "weak" and o["timed_blocks"] >= 5
    assert o["config"]["envs_per_gpu"] == 8 and o["config"]["global_envs"] == 16
    assert o["ms_per_step_min"] <= o["ms_per_step"] <= o["ms_per_step_max"]
    assert abs(o["value"] - 16 / (o["ms_per_step"] * 1e-3)) < 1e-6 * o["value"]
    # the strong-scaling legs ride along: contiguous split of a fixed global batch
    assert o["strong"]["global_envs_12"]["envs_per_gpu"] == 6 and o["strong"]["global_envs_21"]["envs_per_gpu"] == 11
    s = _bench_line(["--gpus", "2", "--scaling", "strong", "--global-envs", "11"])
    assert s["scaling"] == "strong" and s["config"]["global_envs"] == 11 and s["config"]["envs_per_gpu"] == 6
    assert abs(s["value"] - 11 / (s["ms_per_step"] * 1e-3)) < 1e-6 * s["value"]


def test_shard_bounds_cover_the_batch():
    from truss_mi355 import distributed
    for ge, world in ((4096, 8), (8192, 8), (11, 2), (5, 8), (4097, 3)):
        b = [distributed.shard_bounds(ge, world, r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == ge and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def _pool_setup(lib, rank, world):
    """a rank's share of a 3-class pool, stepped once with the class's global actions restricted to the rank's envs"""
    from truss_mi355 import pool, synthetic
    classes = pool.grid_classes([4, 6, 8], [12, 8, 4])
    p = pool.MixedTrussPool(classes, bucket_envs=2, rank=rank, world=world, lib=lib)
    batches, acts = [], []
    for k, e in enumerate(p.envs):
        c, ids = p.class_ids[k], p.global_ids(k)
        full = synthetic.random_batch(e.topo, classes[c][1], 40 + c)
        batches.append({key: v[ids] for key, v in full.items()})
        ag, at = synthetic.random_actions(1, classes[c][1], e.N, 50 + c)
        acts.append((torch.tensor(ag[0][ids]), torch.tensor(at[0][ids])))
    p.set_constants(batches)
    p.set_design(batches)
    p.analyze(set_normalisers=True)
    p.step(acts)
    return p


def _pool_worker(rank, world, port, lib_path, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import truss_mi355 as tm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = _pool_setup(tm.load(lib_path), rank, world)
    mix = torch.tensor([p.sizes[p.class_ids.index(c)] if c in p.class_ids else 0 for c in range(3)], dtype=torch.int64)
    mixes = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(mixes, mix)
    s = p.point.double().sum(dim=0)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)                 # metrics only: no env data crosses ranks
    if rank == 0:
        out.put(([m.tolist() for m in mixes], s.tolist(), p.class_mix().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_mixed_pool_has_the_same_class_mix_on_every_rank():
    lib_path = pc.build_emu()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pool_worker, args=(r, 2, port, lib_path, out)) for r in range(2)]
    for p in procs:
        p.start()
    mixes, total, planned = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert mixes == planned == [[6, 4, 2], [6, 4, 2]]        # round-robin buckets: every rank plays 6 + 4 + 2 envs of the 3 classes
    import truss_mi355 as tm
    whole = _pool_setup(tm.load(lib_path), 0, 1)              # the same envs in one process
    np.testing.assert_allclose(total, whole.point.double().sum(dim=0).tolist(), rtol=1e-12)
