"""GPU parity tests: the hand-written HIP path (through the C ABI) against the oracle and against
the reference's golden vectors.  Run on the MI355X box with `pytest -m gpu`."""
import numpy as np
import pytest
import torch

import truss_mi355 as tm
from truss_mi355 import synthetic
from oracle import truss_oracle as O
from conftest import SCENARIOS
import parity_common as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    lib = tm.load()          # the in-tree HIP .so; raises if it is missing
    assert lib.backend == "hip"
    return lib


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_golden_transitions(lib, name):
    pc.run_golden_transitions(lib, name)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_observation_tensors_golden(lib, name):
    pc.run_obs_golden(lib, name)


@pytest.mark.parametrize("num_x,n_extra,B,steps", [(16, 4, 257, 4), (16, 0, 100, 3), (8, 0, 64, 3), (6, 0, 33, 3)])
def test_random_rollout(lib, num_x, n_extra, B, steps):
    pc.run_random_rollout(lib, num_x, n_extra, B, steps, seed=100 + num_x)


def test_symmetric_variants_random(lib):
    pc.run_random_rollout(lib, 8, 0, 50, 3, seed=5, symmetry="small")
    pc.run_random_rollout(lib, 16, 0, 50, 3, seed=6, symmetry="large")


def test_observation_tensors_random(lib):
    pc.run_obs_random(lib, 16, 4, 129, seed=21)
    pc.run_obs_random(lib, 6, 0, 17, seed=22)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_fused_observation_golden(lib, name):
    """TRUSS_F_EMIT_OBS: the observation tensors written by the step's own launch vs the reference's arrays."""
    pc.run_obs_golden(lib, name, fused=True)


def test_fused_observation_random(lib):
    pc.run_obs_random(lib, 16, 4, 129, seed=21, fused=True, expect_one_launch=True)   # 32 / 80, ragged last workgroup
    pc.run_obs_random(lib, 16, 0, 64, seed=23, fused=True, expect_one_launch=True)    # 32 / 76
    pc.run_obs_random(lib, 8, 0, 65, seed=24, fused=True, expect_one_launch=True)     # 16 / 36
    pc.run_obs_random(lib, 6, 0, 17, seed=22, fused=True, expect_one_launch=True)     # 12 / 26: nN_x_e in 8-byte chunks
    pc.run_obs_random(lib, 5, 0, 17, seed=25, fused=True, expect_one_launch=False)    # 10 / 21 -> step + observation kernel
    for k in range(4):                                                                 # every E mod 4: 16 nodes, 36 - k elements
        pc.run_obs_random(lib, 0, 0, 70, seed=30 + k, fused=True, expect_one_launch=True, topo=pc.pruned_grid(8, k))
    pc.run_obs_random(lib, 64, 0, 6, seed=64, fused=True, expect_one_launch=False)    # 128 nodes -> two launches


def test_fused_observation_full_size(lib):
    """4096 envs, 32 nodes / 80 elements: the fused launch writes what step + observation kernel write, and leaves the
    step's own results untouched; a 128-env sample agrees with the oracle; the reset path (NO_DECODE) emits too."""
    topo = synthetic.bench_topology(16, 4)
    B = 4096
    batch = synthetic.random_batch(topo, B, 77)
    ag, at = synthetic.random_actions(1, B, topo.N, 78)
    e1, e2 = pc.make_env(lib, topo, batch), pc.make_env(lib, topo, batch)
    assert e1.fused_obs
    for e in (e1, e2):
        e.analyze(set_normalisers=True)
    g0, t0 = torch.tensor(ag[0], device=e1.device), torch.tensor(at[0], device=e1.device)
    got = {k: torch.full_like(v, float("nan")) for k, v in e1.obs_buffers().items()}
    e1.step(g0, t0, obs=got)
    e2.step(g0, t0)
    ref = e2.observe()
    r1, r2 = e1.results(), e2.results()
    for k in ("y", "sec", "point", "disp", "q0", "sr", "comp", "max_up", "max_down", "status"):
        assert np.array_equal(r1[k], r2[k]), k
    for k, v in got.items():
        a, b = v.cpu().numpy(), ref[k].cpu().numpy()
        assert np.isfinite(a).all(), k
        np.testing.assert_allclose(a, b, rtol=4e-7, atol=1e-7, err_msg=k)
    idx = np.random.default_rng(1).choice(B, 128, replace=False)
    ot = pc.oracle_topology(topo)
    sub = {k: v[idx] for k, v in batch.items()}
    int_obj = O.initial_objectives(ot, sub["x"], sub["y"], sub["sec"], sub["target"])
    o = O.env_step(ot, sub["x"], sub["y"], sub["sec"], None, None, ag[0][idx], at[0][idx], np.zeros(128), sub["target"],
                   pc.oracle_load(ot, sub), sub["y_max"], sub["d_min"], sub["max_def"], sub["is_roof"], int_obj, with_obs=True)
    for k, v in got.items():
        a, b = v.cpu().numpy()[idx], o[k]
        scale = np.maximum(np.abs(b).reshape(-1, b.shape[-1]).max(axis=0), 1.0)
        assert float((np.abs(a - b) / scale).max()) < 2e-6, k
    # reset path: analysis + observation of the current design in one call
    got2 = e1.analyze(obs=True)
    ref2 = e1.observe({k: torch.empty_like(v) for k, v in got2.items()})
    for k in got2:
        np.testing.assert_allclose(got2[k].cpu().numpy(), ref2[k].cpu().numpy(), rtol=4e-7, atol=1e-7, err_msg=k)


def test_observation_timeout_is_reported(lib):
    """The bounded wait of the streaming wavefront, for real: a fault-injection build of the same kernel (`make all` builds
    libtruss_mi355_faultinj.so: the compute wave never announces progress 3, the wait is 20 000 polls) must drain, write the
    step's own results, leave the tensors of the missing segment untouched and raise TRUSS_STATUS_OBS_TIMEOUT for every env --
    through the streaming wave's own atomic OR (the compute wave has stored status before the wait ends)."""
    import os
    path = os.path.join(os.path.dirname(tm._lib.DEFAULT_LIB), "libtruss_mi355_faultinj.so")
    assert os.path.exists(path), "make -C mop-truss-marl_amd/csrc all"
    flib = tm.load(path)
    assert flib.backend == "hip"
    topo = synthetic.bench_topology(16, 4)
    B = 258                                            # ragged last workgroup
    batch = synthetic.random_batch(topo, B, 5)
    ag, at = synthetic.random_actions(1, B, topo.N, 6)
    e_ok, e_to = pc.make_env(lib, topo, batch), pc.make_env(flib, topo, batch)
    assert e_to.fused_obs
    for e in (e_ok, e_to):
        e.analyze(set_normalisers=True)
    g0, t0 = torch.tensor(ag[0], device=e_ok.device), torch.tensor(at[0], device=e_ok.device)
    ref = e_ok.step(g0.clone(), t0.clone(), obs=True)
    e_ok.check()
    got = {k: torch.full_like(v, float("nan")) for k, v in e_to.obs_buffers().items()}
    e_to.step(g0.clone(), t0.clone(), obs=got)
    torch.cuda.synchronize()
    st = e_to.status.cpu().numpy()
    assert np.all(st & tm._lib.STATUS_OBS_TIMEOUT), st
    assert not np.any(st & tm._lib.STATUS_NOT_SPD)
    r_ok, r_to = e_ok.results(), e_to.results()
    for k in ("y", "sec", "point", "q0", "sr", "disp", "comp"):
        assert np.array_equal(r_to[k], r_ok[k]), k
    for k in ("A_s", "A_n_ts", "A_n_cs"):                                            # segments 1 and 2 ran
        assert torch.equal(got[k], ref[k]), k
    for k in ("x_n", "nN_x_n", "nN_x_e"):                                            # segment 3 gave up
        assert torch.isnan(got[k]).all(), k
    with pytest.raises(tm.TrussError, match="timed out"):
        e_to.check()


def test_fused_observation_soak(lib):
    """60 chained state-emitting steps at 4096 envs on two copies of the same batch: every step's observation tensors and results are
    bitwise equal between the copies (the two-wavefront launch -- compute wave + streaming wave over LDS progress words -- is
    deterministic and leaves no stale byte), finite, and the last step's equal what the stand-alone kernel writes."""
    topo = synthetic.bench_topology(16, 4)
    B = 4096
    batch = synthetic.random_batch(topo, B, 91)
    ag, at = synthetic.random_actions(4, B, topo.N, 92)
    e1, e2 = pc.make_env(lib, topo, batch), pc.make_env(lib, topo, batch)
    assert e1.fused_obs
    for e in (e1, e2):
        e.analyze(set_normalisers=True)
    G = [torch.tensor(a, device=e1.device) for a in ag]
    T = [torch.tensor(a, device=e1.device) for a in at]
    o1 = {k: torch.full_like(v, float("nan")) for k, v in e1.obs_buffers().items()}
    o2 = {k: torch.full_like(v, float("nan")) for k, v in e2.obs_buffers().items()}
    for s in range(60):
        e1.step(G[s % 4], T[s % 4], obs=o1)
        e2.step(G[s % 4], T[s % 4], obs=o2)
        if s % 10 == 9:
            for k in o1:
                assert torch.equal(o1[k], o2[k]), (s, k)
                assert bool(torch.isfinite(o1[k]).all()), (s, k)
            assert torch.equal(e1.point, e2.point) and torch.equal(e1.y, e2.y) and torch.equal(e1.sec, e2.sec)
    ref = e2.observe({k: torch.empty_like(v) for k, v in o2.items()})
    for k in o1:
        np.testing.assert_allclose(o1[k].cpu().numpy(), ref[k].cpu().numpy(), rtol=4e-7, atol=1e-7, err_msg=k)
    assert int(e1.status.sum()) == int(e2.status.sum())


def test_full_size_properties(lib):
    """BASELINE size (32 nodes / 80 elements / 4096 envs): size-independent properties.
    (1) a random 256-env sample agrees with the oracle; (2) equilibrium: reactions balance the applied
    load; (3) work-energy: U = 1/2 d.P; (4) determinism: two runs are bit-identical;
    (5) linearity: doubling the load doubles displacements and member forces."""
    topo = synthetic.bench_topology(16, 4)
    B = 4096
    batch = synthetic.random_batch(topo, B, 77)
    env = pc.make_env(lib, topo, batch)
    env.analyze(set_normalisers=True)
    ag, at = synthetic.random_actions(2, B, topo.N, 78)
    g0, t0 = torch.tensor(ag[0], device=env.device), torch.tensor(at[0], device=env.device)
    env.step(g0, t0)
    r1 = env.results()
    assert int(r1["status"].sum()) == 0
    # (1) oracle on a sample
    idx = np.random.default_rng(0).choice(B, 256, replace=False)
    ot = pc.oracle_topology(topo)
    sub = {k: v[idx] for k, v in batch.items()}
    int_obj = O.initial_objectives(ot, sub["x"], sub["y"], sub["sec"], sub["target"])
    o = O.env_step(ot, sub["x"], sub["y"], sub["sec"], None, None, ag[0][idx], at[0][idx], np.zeros(256),
                   sub["target"], pc.oracle_load(ot, sub), sub["y_max"], sub["d_min"], sub["max_def"], sub["is_roof"],
                   int_obj)
    pc.compare_step({k: v[idx] for k, v in r1.items()}, o, ot)
    # (2) equilibrium
    nloaded = np.where(batch["is_roof"] > 0, topo.load_mask[1].sum(), topo.load_mask[0].sum())
    total = batch["load_y"] * nloaded
    ry = r1["reactions"][:, 1] + r1["reactions"][:, 3]
    rx = r1["reactions"][:, 0] + r1["reactions"][:, 2]
    np.testing.assert_allclose(ry, -total, rtol=1e-8)
    assert np.abs(rx).max() < 1e-6 * np.abs(total).max()
    # (3) work = energy
    P = np.zeros((B, topo.N))
    for b in range(B):
        P[b, topo.load_mask[1 if batch["is_roof"][b] else 0].astype(bool)] = batch["load_y"][b]
    work = 0.5 * (P * r1["disp_f64"][:, :, 1]).sum(axis=1)
    np.testing.assert_allclose(r1["energy"], work, rtol=1e-9)
    # (4) determinism
    env2 = pc.make_env(lib, topo, batch)
    env2.analyze(set_normalisers=True)
    env2.step(g0, t0)
    r2 = env2.results()
    for k in ("y", "sec", "point", "disp", "q0", "sr", "comp"):
        assert np.array_equal(r1[k], r2[k]), k
    # (5) linearity in the load
    b2 = dict(batch)
    b2["load_y"] = batch["load_y"] * 2.0
    b2["y"], b2["sec"] = r1["y"], r1["sec"]
    b1 = dict(batch)
    b1["y"], b1["sec"] = r1["y"], r1["sec"]
    ea, eb = pc.make_env(lib, topo, b1), pc.make_env(lib, topo, b2)
    ea.analyze()
    eb.analyze()
    ra, rb = ea.results(), eb.results()
    np.testing.assert_allclose(rb["disp_f64"], 2.0 * ra["disp_f64"], rtol=1e-9, atol=1e-18)
    np.testing.assert_allclose(rb["q0_f64"], 2.0 * ra["q0_f64"], rtol=1e-8, atol=1e-6)


@pytest.mark.parametrize("case", ["bench_1000", "bench_8200", "bench_8_lanes", "large_symmetric", "small_bridge", "train_12n", "nodes_128"])
def test_rollout_matches_stepwise(lib, case, monkeypatch):
    """truss_rollout (one persistent launch where the topology allows it: design state resident in LDS, next actions
    prefetched) leaves exactly what the same steps leave one launch at a time -- and what chained launches leave."""
    if case == "bench_8_lanes":                  # one team of 8 lanes per env (no two-sided elimination), 10 elements per lane
        monkeypatch.setenv("TRUSS_LANES", "8")
    topo, B, sym = {"bench_1000": (synthetic.bench_topology(16, 4), 1000, False), "bench_8_lanes": (synthetic.bench_topology(16, 4), 333, False),
                    "nodes_128": (tm.TrussTopology.grid(64), 70, False), "bench_8200": (synthetic.bench_topology(16, 4), 8200, False),
                    "large_symmetric": (tm.TrussTopology.grid(16, "large"), 300, True), "small_bridge": (tm.TrussTopology.grid(8), 515, False),
                    "train_12n": (tm.TrussTopology.grid(6), 77, False)}[case]
    batch = synthetic.random_batch(topo, B, 3)
    ag, at = synthetic.random_actions(3, B, topo.N, 9)
    envs = [pc.make_env(lib, topo, batch) for _ in range(3)]
    for e in envs:
        e.analyze(set_normalisers=True)
    e1, e2, e3 = envs
    assert e2.persistent_rollout
    G, T = torch.tensor(ag, device=e1.device), torch.tensor(at, device=e1.device)
    coin = torch.tensor((np.random.default_rng(1).random(B) >= 0.5).astype(np.uint8), device=e1.device) if sym else None
    for s in range(7):
        e1.step(G[s % 3], T[s % 3], coin)
    e2.rollout(G, T, 7, coin)
    monkeypatch.setenv("TRUSS_ROLLOUT_LAUNCHES", "1")
    assert not e3.persistent_rollout
    e3.rollout(G, T, 7, coin)
    r1, r2, r3 = e1.results(), e2.results(), e3.results()
    assert int(r1["status"].sum()) == 0
    for k in ("y", "sec", "point", "q0", "sr", "disp", "comp", "max_up", "max_down", "obj", "status"):
        assert np.array_equal(r1[k], r2[k]), k
        assert np.array_equal(r1[k], r3[k]), k
    # the other buffer of the double-buffered design holds the design before the last step in all three
    assert torch.equal(e1.ybuf[e1.cur ^ 1], e2.ybuf[e2.cur ^ 1]) and torch.equal(e1.secbuf[e1.cur ^ 1], e2.secbuf[e2.cur ^ 1])


def test_every_kernel_variant(lib, monkeypatch):
    """Each compiled lanes-per-env x rows-per-lane instantiation gives the same answers."""
    for G, WL, RPL in [(8, 8, 1), (16, 8, 1), (16, 16, 1), (16, 8, 2), (4, 4, 2)]:
        monkeypatch.setenv("TRUSS_LANES", str(G))
        monkeypatch.setenv("TRUSS_WLANES", str(WL))
        monkeypatch.setenv("TRUSS_RPL", str(RPL))
        topo = tm.TrussTopology.grid(16)
        info = topo.solver_info(lib)
        assert (info["lanes_per_env"], info["rows_per_lane"]) == (G, RPL)
        batch = synthetic.random_batch(topo, 70, 31)
        env = pc.make_env(lib, topo, batch)
        env.analyze(set_normalisers=True)
        ot = pc.oracle_topology(topo)
        int_obj = O.initial_objectives(ot, batch["x"], batch["y"], batch["sec"], batch["target"])
        ag, at = synthetic.random_actions(1, 70, topo.N, 32)
        env.step(torch.tensor(ag[0], device=env.device), torch.tensor(at[0], device=env.device))
        o = O.env_step(ot, batch["x"], batch["y"], batch["sec"], None, None, ag[0], at[0], np.zeros(70),
                       batch["target"], pc.oracle_load(ot, batch), batch["y_max"], batch["d_min"], batch["max_def"],
                       batch["is_roof"], int_obj)
        pc.compare_step(env.results(), o, ot)
        topo.close()


def test_singular_design_is_flagged(lib):
    topo = tm.TrussTopology.grid(6)
    batch = synthetic.random_batch(topo, 2, 1)
    batch["y"][0, 6:] = 0.0
    env = pc.make_env(lib, topo, batch)
    env.analyze()
    st = env.results()["status"]
    assert st[0] == 1 and st[1] == 0


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_irregular_topologies(lib, seed):
    topo = pc.irregular_topology(seed)
    env = pc.run_random_rollout(lib, 0, 0, 40, 3, seed=seed, topo=topo)
    assert int(env.status.sum()) == 0


@pytest.mark.parametrize("num_x,tight", [(40, 1e-9), (64, 1e-9), (128, 1e-7)])
def test_large_trusses(lib, num_x, tight):
    topo = tm.TrussTopology.grid(num_x)
    env = pc.run_random_rollout(lib, 0, 0, 24, 2, seed=num_x, topo=topo, tight=tight)
    assert int(env.status.sum()) == 0


@pytest.mark.parametrize("num_x", [32, 64, 68, 128])
def test_observation_tensors_large(lib, num_x):
    pc.run_obs_random(lib, num_x, 0, 6, seed=num_x)
