"""Shared helpers of the parity tests: run the native library (HIP on the GPU box, or the CPU lane
emulator here) and the oracle on the same inputs and compare."""
import os
import subprocess

import numpy as np
import torch

from oracle import truss_oracle as O
import truss_mi355 as tm
from truss_mi355 import synthetic
from conftest import GOLDEN, ROOT, SCENARIOS

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libtruss_emu.so")


def build_emu():
    src = os.path.join(EMU_DIR, "truss_emu.cpp")
    deps = [src, os.path.join(ROOT, "mop-truss-marl_amd", "csrc", "truss_body.h"),
            os.path.join(ROOT, "mop-truss-marl_amd", "csrc", "truss_host.h"),
            os.path.join(ROOT, "include", "truss_mi355.h")]
    if (not os.path.exists(EMU_LIB)) or any(os.path.getmtime(d) > os.path.getmtime(EMU_LIB) for d in deps):
        # TRUSS_EMU_CXXFLAGS: e.g. -DTRUSS_PIPELINE=1 to emulate the alternative factorisation schedule
        extra = os.environ.get("TRUSS_EMU_CXXFLAGS", "").split()
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                               "-Wno-unknown-pragmas"] + extra + ["-o", EMU_LIB, src], cwd=EMU_DIR)
    return EMU_LIB


def emu_lib():
    return tm.load(build_emu())


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-300))


def oracle_topology(topo: tm.TrussTopology):
    return O.Topology(topo.conn, topo.res, topo.top, topo.pair if topo.pair is not None else np.arange(topo.N),
                      topo.sym_nodes, topo.sym_elems)


def oracle_load(otopo, batch):
    B = batch["y"].shape[0]
    load = np.zeros((B, otopo.N, 2))
    for b in range(B):
        m = O.load_mask(otopo, bool(batch["is_roof"][b]))
        load[b, m, 0] = batch["load_x"][b]
        load[b, m, 1] = batch["load_y"][b]
    return load


def make_env(lib, topo, batch, debug_f64=True):
    B = batch["y"].shape[0]
    env = tm.BatchedTruss(topo, B, lib=lib, debug_f64=debug_f64)
    env.set_constants(batch["x"], batch["target"], batch["y_max"], batch["d_min"], batch["max_def"], batch["load_x"],
                      batch["load_y"], batch["is_roof"])
    env.set_design(batch["y"], batch["sec"])
    return env


def compare_step(r, o, otopo, tight=1e-9, pt_tol=0.0, zero_force=0.0):
    """native results `r` (BatchedTruss.results()) vs oracle step `o`: integers/heights bit-exact,
    float64 solver outputs to `tight`, float32 stores to 1 ulp-ish.
    zero_force > 0 (irregular topologies): members / DOFs whose exact value is zero carry only rounding
    noise, of either sign and different in an LU (the oracle) and an LDL^T (the kernel): elementwise checks
    then use an absolute floor of zero_force x the env's largest value, and the tension/compression flag is
    compared only above that floor."""
    assert np.array_equal(r["y"], o["y"]), "heights differ"
    assert np.array_equal(r["sec"], o["sec"]), "sections differ"
    if zero_force > 0:
        q = o["fem"]["q0"]
        big = np.abs(q) > zero_force * np.abs(q).max(axis=1, keepdims=True)
        assert np.array_equal(r["comp"][big], o["fem"]["comp"][big]), "tension/compression flags differ"
        assert int(r["status"].sum()) == 0
        assert rel(r["disp_f64"], o["fem"]["dnode"]) < tight and rel(r["q0_f64"], o["fem"]["q0"]) < tight
        for k, ref in (("disp", o["fem"]["dnode"]), ("q0", q), ("sr", o["fem"]["sr"])):
            ref = np.asarray(ref, np.float64).reshape(ref.shape[0], -1)
            got = np.asarray(r[k], np.float64).reshape(ref.shape)
            floor = zero_force * np.abs(ref).max(axis=1, keepdims=True)
            assert np.all(np.abs(got - ref) <= 3e-7 * np.abs(ref) + floor), k
        np.testing.assert_allclose(r["point"], o["point"], rtol=max(pt_tol, 3e-7), atol=1e-30)
        return
    assert np.array_equal(r["comp"], o["fem"]["comp"]), "tension/compression flags differ"
    assert int(r["status"].sum()) == 0
    assert np.array_equal(r["max_up"], o["max_up"]) and np.array_equal(r["max_down"], o["max_down"])
    assert rel(r["disp_f64"], o["fem"]["dnode"]) < tight
    assert rel(r["q0_f64"], o["fem"]["q0"]) < tight
    np.testing.assert_allclose(r["disp"], o["fem"]["dnode"].astype(np.float32), rtol=2e-7, atol=1e-30)
    np.testing.assert_allclose(r["q0"], o["fem"]["q0"].astype(np.float32), rtol=2e-7, atol=1e-30)
    np.testing.assert_allclose(r["sr"], o["fem"]["sr"].astype(np.float32), rtol=2e-7, atol=1e-30)
    np.testing.assert_allclose(r["point"], o["point"], rtol=max(pt_tol, 3e-7), atol=1e-30)
    assert rel(r["energy"], o["fem"]["U"]) < max(tight, 1e-8)
    nr = 2 * otopo.N - otopo.ndof
    assert rel(r["reactions"][:, :nr], o["fem"]["r"][:, otopo.ndof:]) < max(tight, 1e-8)


def run_golden_transitions(lib, name, obs=None):
    """Replay the reference's recorded `_game_modify` transitions through the native step.
    obs=True: the step also writes the observation tensors (TRUSS_F_EMIT_OBS) into the env's buffers."""
    nx, var = SCENARIOS[name]
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    topo = tm.TrussTopology.grid(nx, var)
    nsc, tt, nd = topo.dofs(lib)
    assert np.array_equal(nsc, f["nsc"]) and np.array_equal(tt, f["ttnsc"]) and nd == int(f["ndof"])
    B = f["tr_in_node"].shape[0]
    batch = dict(x=np.tile(f["xcoord"], (B, 1)), y=f["tr_in_node"][:, :, 1], sec=f["tr_in_elem"][:, :, 0].astype(np.int32),
                 target=np.tile(f["target0"], (B, 1)), y_max=np.full(B, f["y_max"]), d_min=np.full(B, f["d_min"]),
                 max_def=np.full(B, f["max_deformation"]), load_x=np.zeros(B), load_y=np.full(B, f["loady"]),
                 is_roof=np.full(B, float(f["is_roof"])))
    env = make_env(lib, topo, batch)
    env.env_params[:, 5] = float(f["int_obj1"])
    env.env_params[:, 6] = float(f["int_obj2"])
    dev = env.device
    geo = torch.tensor(f["tr_in_geo"], device=dev)
    tac = torch.tensor(f["tr_in_topo"], device=dev)
    coin = torch.tensor((f["tr_coin"] >= 0.5).astype(np.uint8), device=dev)
    mu = torch.tensor(f["tr_stale_max_up"].astype(np.float32), device=dev)
    md = torch.tensor(f["tr_stale_max_down"].astype(np.float32), device=dev)
    env.step(geo, tac, coin, mu, md, clamp_inplace=True, obs=obs)
    r = env.results()
    # --- against the reference's own outputs ---
    assert np.array_equal(geo.cpu().numpy(), f["tr_clamped_geo"])
    assert np.array_equal(tac.cpu().numpy(), f["tr_clamped_topo"])
    assert np.array_equal(r["y"], f["tr_out_nN_x_n"][:, :, 1])
    assert np.array_equal(r["sec"], f["tr_fem_sec"])
    assert np.array_equal(r["comp"], f["tr_fem_comp"])
    for b in range(B):   # 1e-5 relative (north star); observed ~4e-7 (float32 geometry in the reference)
        assert rel(r["disp_f64"][b], f["tr_fem_dnode"][b]) < 1e-5
        assert rel(r["q0_f64"][b], f["tr_fem_q0"][b]) < 1e-5
        assert rel(r["sr"][b], f["tr_fem_sr"][b]) < 1e-5
    np.testing.assert_allclose(r["point"], f["tr_point"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r["max_up"], f["tr_fem_max_up"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(r["max_down"], f["tr_fem_max_down"], rtol=0, atol=5e-7)
    assert rel(r["energy"], f["tr_fem_U"]) < 1e-5
    # --- and, tightly, against the oracle on the same inputs ---
    ot = oracle_topology(topo)
    o = O.env_step(ot, batch["x"], batch["y"], batch["sec"], f["tr_stale_max_up"], f["tr_stale_max_down"],
                   f["tr_in_geo"], f["tr_in_topo"], f["tr_coin"], batch["target"], oracle_load(ot, batch),
                   batch["y_max"], batch["d_min"], batch["max_def"], batch["is_roof"],
                   np.tile(np.array([f["int_obj1"], f["int_obj2"]], np.float32), (B, 1)))
    compare_step(r, o, ot)
    return env


def irregular_topology(seed):
    """A two-row truss that is NOT one of the reference's grid families: random number of bays, some '/' braces
    removed, long braces over two or three bays added (half-bandwidth up to ~11 -> the W = 16 kernels), a pin and a
    roller (or two pins), sometimes an extra interior roller -> odd DOF parities for the band ordering."""
    rng = np.random.default_rng(seed)
    nx = int(rng.integers(5, 13))
    g = tm.TrussTopology.grid(nx)
    conn = [tuple(c) for c in g.conn.tolist()]
    nb = nx - 1
    slash = conn[-nb:]                                     # '/' braces: the '\' brace of every bay keeps it stable
    keep = [c for c in slash if rng.random() > 0.35]
    conn = conn[:-nb] + keep
    have = set(map(frozenset, conn))
    for _ in range(int(rng.integers(1, 4))):
        s_ = int(rng.integers(2, 4))
        i = int(rng.integers(0, max(1, nx - s_)))
        e = (i, nx + i + s_) if rng.random() < 0.5 else (nx + i, i + s_)
        if frozenset(e) not in have and max(e) < 2 * nx:
            conn.append(e)
            have.add(frozenset(e))
    res = np.zeros((2 * nx, 2), np.uint8)
    res[0] = [1, 1]
    res[nx - 1] = [1, 1] if rng.random() < 0.5 else [0, 1]
    if nx > 7 and rng.random() < 0.5:
        res[nx // 2] = [0, 1]
    return tm.TrussTopology(np.array(conn, np.int32), res, g.top, g.pair)


def pruned_grid(num_x, k):
    """the reference's grid truss without its last k '/' braces (every bay keeps its '\\' brace: still stable, same
    half-bandwidth): element counts of any residue mod 4 at a node count that is a multiple of 4"""
    g = tm.TrussTopology.grid(num_x)
    conn = g.conn[: g.E - k] if k else g.conn
    return tm.TrussTopology(conn, g.res, g.top, g.pair, node_order=g.node_order)


def run_random_rollout(lib, num_x, n_extra, B, n_steps, seed, symmetry=None, topo=None, tight=1e-9):
    """Synthetic random-geometry batch, `n_steps` chained steps, native vs oracle every step."""
    irregular = topo is not None
    if irregular:
        pass
    elif n_extra:
        topo = synthetic.bench_topology(num_x, n_extra)
    else:
        topo = tm.TrussTopology.grid(num_x, symmetry)
    batch = synthetic.random_batch(topo, B, seed)
    env = make_env(lib, topo, batch)
    env.analyze(set_normalisers=True)
    ot = oracle_topology(topo)
    load = oracle_load(ot, batch)
    int_obj = O.initial_objectives(ot, batch["x"], batch["y"], batch["sec"], batch["target"])
    np.testing.assert_allclose(env.env_params[:, 5:7].cpu().numpy(), int_obj, rtol=0, atol=0)
    ag, at = synthetic.random_actions(n_steps, B, topo.N, seed + 1)
    rng = np.random.default_rng(seed + 2)
    y, sec = batch["y"], batch["sec"]
    for s in range(n_steps):
        coin = (rng.random(B) >= 0.5).astype(np.uint8)
        env.step(torch.tensor(ag[s], device=env.device), torch.tensor(at[s], device=env.device),
                 torch.tensor(coin, device=env.device))
        o = O.env_step(ot, batch["x"], y, sec, None, None, ag[s], at[s], coin.astype(np.float64), batch["target"], load,
                       batch["y_max"], batch["d_min"], batch["max_def"], batch["is_roof"], int_obj)
        compare_step(env.results(), o, ot, tight=tight, zero_force=1e-9 if irregular else 0.0)
        y, sec = o["y"], o["sec"]
    return env


def compare_obs(env, o, obs=None, zero_force=0.0):
    """native observation tensors vs the oracle's (float32; 1e-6 relative to the column scale).
    obs: tensors a step(obs=...) call has written; None = run the observation kernel.
    zero_force > 0 (non-grid topologies, see compare_step): the tension / compression flags of nN_x_e (columns 3, 4) are
    not compared for members whose force is rounding noise."""
    obs = {k: v.cpu().numpy() for k, v in (env.observe() if obs is None else obs).items()}
    for k in ("x_n", "A_s", "A_n_ts", "A_n_cs", "nN_x_n", "nN_x_e"):
        a, b = obs[k], o[k]
        assert a.shape == b.shape, k
        if k == "nN_x_e" and zero_force > 0:
            q = o["fem"]["q0"]
            noise = np.abs(q) <= zero_force * np.abs(q).max(axis=1, keepdims=True)
            a = a.copy()
            a[..., 3:5] = np.where(noise[..., None], b[..., 3:5], a[..., 3:5])
        scale = np.maximum(np.abs(b).reshape(-1, b.shape[-1]).max(axis=0), 1.0)
        err = np.abs(a - b) / scale
        assert float(err.max()) < 2e-6, (k, float(err.max()))
    return obs


def run_obs_random(lib, num_x, n_extra, B, seed, fused=False, expect_one_launch=None, topo=None):
    """fused: the observation tensors come out of the step call itself (TRUSS_F_EMIT_OBS) and are compared with the
    oracle AND with what the stand-alone observation kernel writes for the same step."""
    custom = topo is not None
    if topo is None:
        topo = synthetic.bench_topology(num_x, n_extra) if n_extra else tm.TrussTopology.grid(num_x)
    batch = synthetic.random_batch(topo, B, seed)
    env = make_env(lib, topo, batch)
    env.analyze(set_normalisers=True)
    ot = oracle_topology(topo)
    load = oracle_load(ot, batch)
    int_obj = O.initial_objectives(ot, batch["x"], batch["y"], batch["sec"], batch["target"])
    ag, at = synthetic.random_actions(1, B, topo.N, seed + 1)
    got = None
    if fused:
        if expect_one_launch is not None:
            assert env.fused_obs == expect_one_launch
        got = {k: torch.full_like(v, float("nan")) for k, v in env.obs_buffers().items()}   # every element must be written
    env.step(torch.tensor(ag[0], device=env.device), torch.tensor(at[0], device=env.device), obs=got)
    o = O.env_step(ot, batch["x"], batch["y"], batch["sec"], None, None, ag[0], at[0], np.zeros(B), batch["target"],
                   load, batch["y_max"], batch["d_min"], batch["max_def"], batch["is_roof"], int_obj, with_obs=True)
    zf = 1e-9 if custom else 0.0
    if fused:
        a = compare_obs(env, o, got, zero_force=zf)
        b = compare_obs(env, o, zero_force=zf)
        for k in a:     # same staged rows; element length by rsqrt-Newton vs sqrt, normalisation by reciprocal vs division
            np.testing.assert_allclose(a[k], b[k], rtol=4e-7, atol=1e-7, err_msg=k)
    else:
        compare_obs(env, o, zero_force=zf)
    return env


def run_obs_golden(lib, name, fused=False):
    """observation tensors of the replayed golden transitions vs the reference's own arrays; fused: as written by
    the step call itself (TRUSS_F_EMIT_OBS) instead of the stand-alone observation kernel."""
    env = run_golden_transitions(lib, name, obs=True if fused else None)
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    obs = {k: v.cpu().numpy() for k, v in (env.obs_buffers() if fused else env.observe()).items()}
    for k in ("x_n", "A_s", "A_n_ts", "A_n_cs", "nN_x_n"):
        np.testing.assert_allclose(obs[k], f["tr_out_" + k], rtol=1e-5, atol=5e-6, err_msg=k)
    xe, ref = obs["nN_x_e"], f["tr_out_nN_x_e"]
    scale = np.maximum(np.abs(ref).max(axis=(0, 1), keepdims=True), 1.0)
    assert float((np.abs(xe - ref) / scale).max()) < 1e-5
    A_n, mask = env.topo.normalized_adjacency()
    np.testing.assert_allclose(A_n, f["reset_A_n"], rtol=0, atol=1e-7)
    assert np.array_equal(mask, f["reset_mask"]) and np.array_equal(env.topo.incidence(), f["reset_nC_e"])
