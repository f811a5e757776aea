"""MixedTrussPool / MixedMARL (BASELINE configs[4], SURVEY.md §8e "bucket by (N, E) class, round-robin buckets across
GPUs"): the deal, the fan-out over classes against single-class runs and the oracle, the shared-agent rollout."""
import contextlib
import io

import numpy as np
import pytest
import torch

import truss_mi355 as tm
from truss_mi355 import marl, pool, synthetic
from oracle import truss_oracle as O
import parity_common as pc


def test_deal_is_a_balanced_partition():
    for class_envs, be, world in (([4096, 2048, 1024, 512], 64, 8), ([300, 70, 10], 32, 3), ([64], 64, 2), ([1000, 1000], 100, 7)):
        share = pool.deal_buckets(class_envs, be, world)
        for c, n in enumerate(class_envs):
            got = sorted(r for rk in share for r in rk[c])
            assert got[0][0] == 0 and got[-1][1] == n and all(a[1] == b[0] for a, b in zip(got, got[1:]))    # a partition
            per_rank = [sum(hi - lo for lo, hi in rk[c]) for rk in share]
            assert max(per_rank) - min(per_rank) <= be                                # the same class mix on every rank
    mix = pool.MixedTrussPool.class_mix.__get__(type("P", (), {"share": pool.deal_buckets([4096, 2048, 1024, 512], 64, 8),
                                                                  "classes": [0] * 4})())()
    assert mix.tolist() == [[512, 256, 128, 64]] * 8                                   # configs[4] on 8 GPUs: identical shares


def _class_batches(p, seed):
    out = []
    for k, e in enumerate(p.envs):
        c = p.class_ids[k]
        full = synthetic.random_batch(e.topo, p.classes[c][1], seed + c)               # the class's GLOBAL batch ...
        ids = p.global_ids(k)
        out.append({key: v[ids] for key, v in full.items()})                           # ... and this rank's rows of it
    return out


def _check_pool(lib, device, num_xs, envs, world, seed, bucket):
    classes = pool.grid_classes(num_xs, envs)
    pools = [pool.MixedTrussPool(classes, bucket_envs=bucket, rank=r, world=world, device=device, lib=lib) for r in range(world)]
    seen = [np.zeros(n, int) for n in envs]
    for p in pools:
        batches = _class_batches(p, seed)
        p.set_constants(batches)
        p.set_design(batches)
        p.analyze(set_normalisers=True)
        acts, np_acts = [], []
        for k, e in enumerate(p.envs):
            c = p.class_ids[k]
            ag, at = synthetic.random_actions(1, p.classes[c][1], e.N, seed + 100 + c)
            ids = p.global_ids(k)
            np_acts.append((ag[0][ids], at[0][ids]))
            acts.append((torch.tensor(ag[0][ids], device=p.device), torch.tensor(at[0][ids], device=p.device)))
        obs = p.step(acts, obs=True)
        idx = p.index()
        assert p.point.shape == (p.n_envs, 4) and idx.shape == (p.n_envs, 2) and int(p.status.sum()) == 0
        for k, e in enumerate(p.envs):
            c, b = p.class_ids[k], batches[k]
            seen[c][p.global_ids(k)] += 1
            ot = pc.oracle_topology(e.topo)
            int_obj = O.initial_objectives(ot, b["x"], b["y"], b["sec"], b["target"])
            o = O.env_step(ot, b["x"], b["y"], b["sec"], None, None, np_acts[k][0], np_acts[k][1], np.zeros(e.B), b["target"],
                           pc.oracle_load(ot, b), b["y_max"], b["d_min"], b["max_def"], b["is_roof"], int_obj, with_obs=True)
            r = e.results()
            assert np.array_equal(r["y"], o["y"]) and np.array_equal(r["sec"], o["sec"]) and np.array_equal(r["comp"], o["fem"]["comp"])
            np.testing.assert_allclose(r["point"], o["point"], rtol=3e-6, atol=1e-30)
            lo, hi = p.offsets[k], p.offsets[k + 1]
            assert torch.equal(p.point[lo:hi], e.point) and np.all(idx[lo:hi, 0] == c)
            pc.compare_obs(e, o, obs[k])
    assert all(np.all(s == 1) for s in seen)          # every env of every class lives on exactly one rank


def test_mixed_pool_emulated():
    _check_pool(pc.emu_lib(), "cpu", [4, 6, 8, 16], [9, 7, 6, 5], world=2, seed=3, bucket=2)


@pytest.mark.gpu
def test_mixed_pool_hip_32_to_256_nodes():
    """the size classes of BASELINE configs[4] on one GPU (streams per class), every class against the oracle"""
    _check_pool(tm.load(), "cuda", [16, 32, 64, 128], [96, 48, 24, 12], world=1, seed=5, bucket=8)
    _check_pool(tm.load(), "cuda", [16, 32, 64, 128], [64, 32, 16, 8], world=2, seed=6, bucket=4)


def _mixed_marl(lib, device):
    import truss2D_RL as RL
    import master_DDPG_truss2D_MO as M
    torch.manual_seed(4)
    rl = RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, 16, 8, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma, device=device)
    classes = pool.grid_classes([4, 6], [6, 5])
    eng = marl.MixedMARL(classes, rl, bucket_envs=3, max_front=6, lib=lib, device=device, replay_capacity=128, batch_size=4, seed=2)
    per_class = []
    for k, e in enumerate(eng.engines):
        full = synthetic.random_batch(e.topo, classes[eng.class_ids[k]][1], 9 + k)
        per_class.append({key: v[eng.global_ids(k)] for key, v in full.items()})
    eng.reset(per_class)
    upd = 0
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(3):
            out = eng.game_step_all(train=True, explore=True, train_iters=2)
            upd += out["updates"]
    assert len(out["per_class"]) == 2 and out["hv"].shape[0] == 11 and eng.env_steps > 0
    assert upd >= 2                                    # both classes' replays fed the ONE set of agents
    assert all(e.replay.size > 0 for e in eng.engines)
    for e in eng.engines:                              # archives stay consistent per class (points of archived designs)
        assert int(e.n.min()) >= 1 and int(e.n.max()) <= 6


def test_mixed_marl_emulated():
    _mixed_marl(pc.emu_lib(), "cpu")


@pytest.mark.gpu
def test_mixed_marl_hip():
    _mixed_marl(tm.load(), "cuda")
