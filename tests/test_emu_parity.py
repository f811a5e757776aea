"""CPU tests of the kernel's lane program through the lane emulator (tests/emu), of the host logic
and of the C ABI surface.  No GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import truss_mi355 as tm
from truss_mi355 import synthetic
from oracle import truss_oracle as O
from conftest import ROOT, SCENARIOS
import parity_common as pc


@pytest.fixture(scope="module")
def lib():
    return pc.emu_lib()


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_golden_transitions_emulated(lib, name):
    pc.run_golden_transitions(lib, name)


@pytest.mark.parametrize("num_x,n_extra,B,steps", [(16, 4, 19, 3), (16, 0, 8, 2), (6, 0, 5, 3)])
def test_random_rollout_emulated(lib, num_x, n_extra, B, steps):
    pc.run_random_rollout(lib, num_x, n_extra, B, steps, seed=11 + num_x)


@pytest.mark.parametrize("name", ["train0", "small_roof", "large_bridge"])
def test_observation_tensors_golden_emulated(lib, name):
    pc.run_obs_golden(lib, name)


def test_observation_tensors_random_emulated(lib):
    pc.run_obs_random(lib, 16, 4, 7, seed=21)
    pc.run_obs_random(lib, 6, 0, 3, seed=22)


@pytest.mark.parametrize("name", ["train0", "train_eval", "small_roof", "large_bridge", "large_roof"])
def test_fused_observation_golden_emulated(lib, name):
    """TRUSS_F_EMIT_OBS: the step writes the observation tensors itself; against the reference's recorded arrays."""
    pc.run_obs_golden(lib, name, fused=True)


def test_fused_observation_random_emulated(lib):
    pc.run_obs_random(lib, 16, 4, 7, seed=21, fused=True, expect_one_launch=True)    # 32 / 80: the metric topology
    pc.run_obs_random(lib, 16, 0, 5, seed=23, fused=True, expect_one_launch=True)    # 32 / 76
    pc.run_obs_random(lib, 8, 0, 6, seed=24, fused=True, expect_one_launch=True)     # 16 / 36
    pc.run_obs_random(lib, 6, 0, 3, seed=22, fused=True, expect_one_launch=True)     # 12 / 26 (the reference's training trusses):
                                                                                      # E % 4 == 2 -> nN_x_e in 8-byte chunks
    pc.run_obs_random(lib, 5, 0, 3, seed=25, fused=True, expect_one_launch=False)    # 10 / 21: N % 4 != 0 -> two launches


def test_observation_timeout_status_plumbing_emulated(lib, monkeypatch):
    """TRUSS_STATUS_OBS_TIMEOUT end to end on the host side: when the observation stream of the fused step gives up (forced in
    the emulator; the GPU test test_observation_timeout_is_reported runs the real bounded wait on a fault-injection build) the
    step's own results are complete, status carries bit 1 for every env, the late tensors are untouched and check() raises."""
    topo = synthetic.bench_topology(16, 4)
    B = 6
    batch = synthetic.random_batch(topo, B, 5)
    ag, at = synthetic.random_actions(1, B, topo.N, 6)
    e_ok, e_to = pc.make_env(lib, topo, batch), pc.make_env(lib, topo, batch)
    for e in (e_ok, e_to):
        e.analyze(set_normalisers=True)
    g0, t0 = torch.tensor(ag[0]), torch.tensor(at[0])
    e_ok.step(g0.clone(), t0.clone(), obs=True)
    e_ok.check()
    monkeypatch.setenv("TRUSS_EMU_FORCE_OBS_TIMEOUT", "1")
    got = {k: torch.full_like(v, float("nan")) for k, v in e_to.obs_buffers().items()}
    e_to.step(g0.clone(), t0.clone(), obs=got)
    monkeypatch.delenv("TRUSS_EMU_FORCE_OBS_TIMEOUT")
    st = e_to.status.numpy()
    assert np.all(st & tm._lib.STATUS_OBS_TIMEOUT) and not np.any(st & tm._lib.STATUS_NOT_SPD)
    for k in ("y", "sec", "point", "q0", "sr", "disp"):
        assert np.array_equal(e_to.results()[k], e_ok.results()[k]), k
    assert torch.equal(got["A_s"], e_ok.obs_buffers()["A_s"])                       # segment 1 ran
    for k in ("A_n_ts", "A_n_cs", "x_n", "nN_x_n", "nN_x_e"):
        assert torch.isnan(got[k]).all(), k                                          # segments 2 and 3 did not
    with pytest.raises(tm.TrussError, match="timed out"):
        e_to.check()
    e_to.step(g0.clone(), t0.clone(), obs=got)                                       # the next call is clean again
    e_to.check()
    assert not any(torch.isnan(v).any() for v in got.values())


def test_fused_observation_any_element_count_emulated(lib):
    """element counts of every residue mod 4: nN_x_e rows (21 E floats per env) are 16-, 8- or 4-byte aligned per env"""
    for k in range(4):
        topo = pc.pruned_grid(8, k)                      # 16 nodes, 36 - k elements
        env = pc.run_obs_random(lib, 0, 0, 5, seed=30 + k, fused=True, expect_one_launch=True, topo=topo)
        assert env.persistent_rollout


def test_symmetric_variants_random(lib):
    pc.run_random_rollout(lib, 8, 0, 9, 2, seed=5, symmetry="small")
    pc.run_random_rollout(lib, 16, 0, 6, 2, seed=6, symmetry="large")


@pytest.mark.parametrize("G,WL,RPL", [(8, 8, 1), (16, 8, 1), (16, 16, 1), (16, 8, 2), (4, 4, 2)])
def test_every_kernel_variant_emulated(lib, monkeypatch, G, WL, RPL):
    monkeypatch.setenv("TRUSS_LANES", str(G))
    monkeypatch.setenv("TRUSS_WLANES", str(WL))
    monkeypatch.setenv("TRUSS_RPL", str(RPL))
    env = pc.run_random_rollout(lib, 16, 4, 7, 2, seed=40 + G + RPL)
    info = env.topo.solver_info(lib)
    assert (info["lanes_per_env"], info["rows_per_lane"]) == (G, RPL)
    env.topo.close()


def test_threebar_analysis_only(lib):
    f = np.load(os.path.join(pc.GOLDEN, "threebar.npz"))
    topo = tm.TrussTopology(f["conn"], f["res"], np.zeros(4, np.uint8), pair=None,
                            load_mask=np.array([[1, 0, 0, 0], [1, 0, 0, 0]]),
                            sections=np.array([[8.0, 1.0], [6.0, 1.0]]), e_mod=float(f["em"]), long_stress=1.0)
    nsc, tt, nd = topo.dofs(lib)
    assert np.array_equal(nsc, f["nsc"]) and np.array_equal(tt, f["ttnsc"]) and nd == 2
    env = tm.BatchedTruss(topo, 3, lib=lib, debug_f64=True)
    env.set_constants(f["coords"][:, 0], np.zeros(4), 1e9, 0.0, 1.0, 150.0, -300.0, 0.0)
    env.set_design(f["coords"][:, 1], np.array([0, 1, 0]))
    env.analyze()
    r = env.results()
    for b in range(3):
        assert pc.rel(r["q0_f64"][b], f["q0"]) < 1e-12
        assert pc.rel(r["disp_f64"][b][0], f["d"]) < 1e-12
        assert pc.rel(r["reactions"][b], f["r"][2:]) < 1e-12
        assert pc.rel(r["energy"][b], f["U"]) < 1e-12
    with pytest.raises(tm.TrussError):      # no pair table -> the action decode is refused, loudly
        env.step(torch.zeros(3, 4, 2), torch.zeros(3, 4, 3))


@pytest.mark.parametrize("case", ["bench", "bench_8_lanes", "large_symmetric", "small_bridge", "train_12n"])
def test_rollout_matches_stepwise(lib, case, monkeypatch):
    """the persistent rollout (all steps of a workgroup on one LDS image, truss_emu.cpp::emu_rollout) vs single steps"""
    if case == "bench_8_lanes":                  # one team of 8 lanes per env (no two-sided elimination), 10 elements per lane
        monkeypatch.setenv("TRUSS_LANES", "8")
    topo, B, sym = {"bench": (synthetic.bench_topology(16, 4), 10, False), "bench_8_lanes": (synthetic.bench_topology(16, 4), 11, False), "large_symmetric": (tm.TrussTopology.grid(16, "large"), 6, True),
                    "small_bridge": (tm.TrussTopology.grid(8), 9, False), "train_12n": (tm.TrussTopology.grid(6), 5, False)}[case]
    batch = synthetic.random_batch(topo, B, 3)
    ag, at = synthetic.random_actions(3, B, topo.N, 9)
    e1 = pc.make_env(lib, topo, batch)
    e1.analyze(set_normalisers=True)
    e2 = pc.make_env(lib, topo, batch)
    e2.analyze(set_normalisers=True)
    assert e2.persistent_rollout
    coin = torch.tensor((np.random.default_rng(1).random(B) >= 0.5).astype(np.uint8)) if sym else None
    for s in range(5):
        e1.step(torch.tensor(ag[s % 3]), torch.tensor(at[s % 3]), coin)
    e2.rollout(torch.tensor(ag), torch.tensor(at), 5, coin)
    r1, r2 = e1.results(), e2.results()
    for k in ("y", "sec", "point", "q0", "sr", "disp", "comp", "max_up", "max_down", "status"):
        assert np.array_equal(r1[k], r2[k]), k


def test_singular_design_is_flagged(lib):
    """A mechanism (zero-area is impossible, so: all nodes of a bay collapsed) must raise status."""
    topo = tm.TrussTopology.grid(6)
    batch = synthetic.random_batch(topo, 2, 1)
    batch["y"][0, 6:] = 0.0          # top chord on the bottom chord: K singular for env 0
    env = pc.make_env(lib, topo, batch)
    env.analyze()
    st = env.results()["status"]
    assert st[0] == 1 and st[1] == 0


def test_argument_errors(lib):
    topo = tm.TrussTopology.grid(6)
    env = tm.BatchedTruss(topo, 4, lib=lib)
    with pytest.raises(tm.TrussError, match="a_geo must be Float"):       # dtype / device / contiguity: the operator's checks
        env.step(torch.zeros(4, 12, 2, dtype=torch.float64), torch.zeros(4, 12, 3))
    with pytest.raises(ValueError):                                          # shapes: the host class
        env.step(torch.zeros(4, 12, 3), torch.zeros(4, 12, 3))
    with pytest.raises(ValueError):
        env.step(torch.zeros(4, 12, 2), torch.zeros(4, 12, 3), obs=dict(x_n=torch.zeros(4, 12, 12)))
    bad = tm.TrussTopology([[0, 1], [1, 2]], np.zeros((3, 2)), np.zeros(3), pair=[1, 0, 2])
    with pytest.raises(tm.TrussError):
        bad.native(lib)
    # half-bandwidth beyond the compiled windows -> refused, not silently wrong
    N = 40
    conn = [(i, i + 1) for i in range(N - 1)] + [(7, j) for j in range(9, N)]   # free hub: no narrow band exists
    res = np.zeros((N, 2)); res[0] = 1; res[3] = 1
    wide = tm.TrussTopology(conn, res, np.zeros(N))
    with pytest.raises(tm.TrussError):
        wide.native(lib)


def test_hip_library_exports_every_declared_symbol():
    """The product .so must load and export exactly what include/truss_mi355.h declares."""
    hdr = open(os.path.join(ROOT, "include", "truss_mi355.h")).read()
    names = set(re.findall(r"\b(truss_[a-z_]+)\s*\(", hdr))
    names = {n for n in names if not n.endswith("_t")}
    assert {"truss_step", "truss_rollout", "truss_topo_create", "truss_obs"} <= names
    path = tm._lib.DEFAULT_LIB
    assert os.path.exists(path), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    dll = ctypes.CDLL(path)
    for n in sorted(names):
        assert hasattr(dll, n), n
    dll.truss_backend.restype = ctypes.c_char_p
    assert dll.truss_backend() == b"hip"
    dll.truss_abi_version.restype = ctypes.c_int
    assert dll.truss_abi_version() == tm._lib.TRUSS_ABI_VERSION == 3


def test_product_refuses_non_hip_default(monkeypatch, tmp_path):
    missing = tmp_path / "nope.so"
    with pytest.raises(tm.TrussError):
        tm._lib.TrussLib(str(missing))


def test_hip_kernels_use_no_scratch():
    """A runtime-indexed register array silently moves the whole lane state to scratch memory
    (hidden HBM traffic + latency; it happened twice during development).  The resource report of
    the shipped build must show zero scratch and zero VGPR spills for every product kernel."""
    rep = os.path.join(os.path.dirname(tm._lib.DEFAULT_LIB), "libtruss_mi355.resources.txt")
    assert os.path.exists(rep), "build with `make -C mop-truss-marl_amd/csrc` (or __graft_entry__.build())"
    assert os.path.getmtime(rep) >= os.path.getmtime(tm._lib.DEFAULT_LIB) - 120
    txt = open(rep).read()
    names = re.findall(r"Function Name: (\S+)", txt)
    scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", txt)]
    spills = [int(v) for v in re.findall(r"VGPRs Spill: (\d+)", txt)]
    assert len(names) == len(scratch) == len(spills) and len(names) >= 8
    assert sum("truss_step_kernel" in n for n in names) >= 7 and any("truss_obs_kernel" in n for n in names)
    assert all(v == 0 for v in scratch), dict(zip(names, scratch))
    # the kernels of the metric configuration (16 lanes per env, 5 elements per lane) stay below the 256 architectural VGPRs: every
    # build that needed AGPR copies there ran the rollout ~5 % slower (tools/experiments/README.md)
    agprs = dict(zip(names, [int(v) for v in re.findall(r"AGPRs: (\d+)", txt)]))
    metric = [n for n in names if "ILi16ELi8ELi1ELi5E" in n]
    assert len(metric) >= 3 and all(agprs[n] == 0 for n in metric), {n: agprs[n] for n in metric}
    # one-row-per-lane kernels (every shipped configuration): no spills at all.  The two-rows-per-lane
    # fallbacks for wide bands may park a register or two in the AGPR file (no memory traffic: scratch is 0).
    assert all(v == 0 for n, v in zip(names, spills) if "ELi2ELi" not in n), dict(zip(names, spills))
    assert all(v <= 4 for v in spills), dict(zip(names, spills))


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_irregular_topologies_emulated(seed):
    """not the reference's grid families: random bays / braces / supports (see parity_common.irregular_topology)"""
    topo = pc.irregular_topology(seed)
    env = pc.run_random_rollout(pc.emu_lib(), 0, 0, 5, 2, seed=seed, topo=topo)
    assert int(env.status.sum()) == 0


@pytest.mark.parametrize("num_x,tight", [(40, 1e-9), (64, 1e-9), (128, 1e-7)])
def test_large_trusses_emulated(num_x, tight):
    """BASELINE config 5 sizes: 80 / 128 / 256 nodes (196 / 316 / 636 elements, up to 508 DOF) on the 32- and
    64-lane kernels.  The LU of the oracle and the LDL^T of the kernel drift apart with the condition number
    (~ bays^4): 4e-9 at 256 nodes, far inside the 1e-5 budget."""
    topo = tm.TrussTopology.grid(num_x)
    info = topo.solver_info(pc.emu_lib())
    assert info["lanes_per_env"] == (32 if num_x <= 64 else 64) and info["half_bandwidth"] == 7
    env = pc.run_random_rollout(pc.emu_lib(), 0, 0, 2, 2, seed=num_x, topo=topo, tight=tight)
    assert int(env.status.sum()) == 0


@pytest.mark.parametrize("num_x", [32, 64, 68, 128])
def test_observation_tensors_large_emulated(num_x):
    """64 / 128 / 136 / 256 nodes: the three N x N matrices are written in row tiles, one workgroup per (env, tile) and one for the
    per-node / per-element rows: 8 tiles of 8 rows, 16 of 8, 20 of 6 + one of 4, 64 of 4"""
    pc.run_obs_random(pc.emu_lib(), num_x, 0, 2, seed=num_x)
