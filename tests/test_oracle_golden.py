"""Pins the CPU oracle (oracle/truss_oracle.py) to golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only.

Tolerances: integers (connectivity, DOF numbering, sections, flags) and heights on the 0.01 grid are
bit-exact; float64 FEM results are within 1e-5 relative as BASELINE.json's north_star states
(observed ~4e-7: the reference under numpy 2.x evaluates element length/cos/sin in float32, the
oracle in float64 -- see oracle header)."""
import os

import numpy as np
import pytest

from oracle import truss_oracle as O
from conftest import GOLDEN, SCENARIOS

RTOL = 1e-5


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()
                 / max(float(np.abs(b).max()), 1e-300))


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _run_transitions(f, topo):
    B = f["tr_in_node"].shape[0]
    x = np.tile(f["xcoord"], (B, 1))
    y = f["tr_in_node"][:, :, 1]
    sec = f["tr_in_elem"][:, :, 0].astype(np.int32)
    lm = O.load_mask(topo, bool(f["is_roof"]))
    load = np.zeros((B, topo.N, 2))
    load[:, lm, 1] = f["loady"]
    ones = np.ones(B)
    tgt = np.tile(f["target0"], (B, 1))
    int_obj = np.tile(np.array([f["int_obj1"], f["int_obj2"]], np.float32), (B, 1))
    return O.env_step(topo, x, y, sec, f["tr_stale_max_up"], f["tr_stale_max_down"], f["tr_in_geo"],
                      f["tr_in_topo"], f["tr_coin"], tgt, load, ones * f["y_max"], ones * f["d_min"],
                      ones * f["max_deformation"], ones * f["is_roof"], int_obj, with_obs=True)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_topology_integers_bit_exact(name):
    f = _load(name)
    nx, var = SCENARIOS[name]
    t = O.grid_topology(nx, var)
    for k in ("conn", "res", "top", "pair", "nsc", "tnsc", "ttnsc"):
        assert np.array_equal(getattr(t, k), f[k]), k
    assert t.ndof == int(f["ndof"])
    assert np.array_equal(O.load_mask(t, bool(f["is_roof"])), f["has_loady"] != 0)
    assert np.array_equal(O.load_mask(t, bool(f["is_roof"])), f["nload"] != 0)
    assert np.array_equal(O.incidence(t), f["reset_nC_e"])
    A_n, mask = O.normalized_adjacency(t)
    assert np.array_equal(mask, f["reset_mask"])
    np.testing.assert_allclose(A_n, f["reset_A_n"], rtol=0, atol=1e-7)
    assert np.allclose(O.SECTIONS, f["sections"], rtol=0, atol=0)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_reset_state(name):
    f = _load(name)
    nx, var = SCENARIOS[name]
    t = O.grid_topology(nx, var)
    x = f["xcoord"][None, :]
    y = f["reset_ycoord"][None, :].astype(np.float32)
    sec = f["reset_sec"][None, :]
    load = np.zeros((1, t.N, 2))
    load[:, O.load_mask(t, bool(f["is_roof"])), 1] = f["loady"]
    fem = O.fem_solve(t, x, y, sec, load)
    assert _rel(fem["K"][0], f["reset_K"]) < 1e-12
    assert np.array_equal(fem["P"][0], f["reset_P"])
    assert _rel(fem["d"][0], f["reset_d"]) < 1e-9
    assert _rel(fem["q0"][0], f["reset_q0"]) < 1e-9
    assert np.array_equal(fem["comp"][0], f["reset_comp"])
    int_obj = O.initial_objectives(t, x, y, sec, f["target0"][None, :])
    assert _rel(int_obj[0, 0], f["int_obj1"]) < 3e-7
    assert _rel(int_obj[0, 1], f["int_obj2"]) < 3e-7
    mu, md = O.move_range(t, y, [f["y_max"]], [f["d_min"]], [f["is_roof"]])
    np.testing.assert_allclose(mu[0], f["reset_max_up"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(md[0], f["reset_max_down"], rtol=0, atol=2e-7)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_transitions(name):
    f = _load(name)
    nx, var = SCENARIOS[name]
    t = O.grid_topology(nx, var)
    out = _run_transitions(f, t)
    # bit-exact: clamped actions, heights, sections, tension/compression flags
    assert np.array_equal(out["geo"], f["tr_clamped_geo"])
    assert np.array_equal(out["topo"], f["tr_clamped_topo"])
    assert np.array_equal(out["y"], f["tr_out_nN_x_n"][:, :, 1])
    assert np.array_equal(out["sec"], f["tr_fem_sec"])
    assert np.array_equal(out["fem"]["comp"], f["tr_fem_comp"])
    # float64 FEM within the north-star tolerance
    for k in ("K", "d", "q0", "sr", "U", "dnode"):
        for b in range(out["y"].shape[0]):
            assert _rel(out["fem"][k][b], f["tr_fem_" + k][b]) < RTOL, (k, b)
    assert np.array_equal(out["fem"]["P"], f["tr_fem_P"])
    rr = f["tr_fem_r"]
    m = ~np.isnan(rr)
    assert np.array_equal(np.isnan(out["fem"]["r"]), ~m)
    assert _rel(out["fem"]["r"][m], rr[m]) < RTOL
    np.testing.assert_allclose(out["point"], f["tr_point"], rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(out["max_up"], f["tr_fem_max_up"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(out["max_down"], f["tr_fem_max_down"], rtol=0, atol=5e-7)
    # observation tensors (float32)
    for k in ("x_n", "A_s", "A_n_ts", "A_n_cs", "nN_x_n"):
        np.testing.assert_allclose(out[k], f["tr_out_" + k], rtol=RTOL, atol=5e-6, err_msg=k)
    xe, xe_ref = out["nN_x_e"], f["tr_out_nN_x_e"]
    scale = np.maximum(np.abs(xe_ref).max(axis=(0, 1), keepdims=True), 1.0)
    assert float((np.abs(xe - xe_ref) / scale).max()) < RTOL


def test_move_range_recompute_matches_parent_state():
    """Product mode feeds max_up/max_down recomputed from the parent heights; that must equal columns
    7,8 of the parent's archived node state (what set_moveRange produced when the parent was analysed)."""
    for name, (nx, var) in SCENARIOS.items():
        f = _load(name)
        t = O.grid_topology(nx, var)
        pn = f["tr_out_nN_x_n"]
        B = pn.shape[0]
        mu, md = O.move_range(t, pn[:, :, 1], np.full(B, f["y_max"]), np.full(B, f["d_min"]),
                              np.full(B, f["is_roof"]))
        np.testing.assert_allclose(mu, pn[:, :, 7], rtol=0, atol=5e-7)
        np.testing.assert_allclose(md, pn[:, :, 8], rtol=0, atol=5e-7)


def test_threebar_textbook():
    f = _load("threebar")
    N = f["coords"].shape[0]
    t = O.Topology(f["conn"], f["res"], np.zeros(N, np.int8), np.arange(N))
    assert np.array_equal(t.nsc, f["nsc"]) and np.array_equal(t.ttnsc, f["ttnsc"]) and t.ndof == int(f["ndof"])
    fem = O.fem_solve(t, f["coords"][None, :, 0], f["coords"][None, :, 1].astype(np.float32),
                      np.zeros((1, 3), np.int32), f["load"][None], e_mod=float(f["em"]), area=f["area"][None])
    assert _rel(fem["K"][0], f["K"]) < 1e-12
    assert _rel(fem["d"][0], f["d"]) < 1e-12
    assert _rel(fem["q0"][0], f["q0"]) < 1e-12
    m = ~np.isnan(f["r"])
    assert _rel(fem["r"][0][m], f["r"][m]) < 1e-12
    # d = [0.21551724, -0.13995257], q = [-16.7700, 126.8320, 233.2300]
    np.testing.assert_allclose(fem["d"][0], [0.21551724, -0.13995257], atol=1e-8)
    np.testing.assert_allclose(fem["q0"][0], [-16.7700, 126.8320, 233.2300], atol=1e-4)


def test_hypervolume_and_cull_golden():
    f = _load("reward")
    assert abs(O.hypervolume_2d(f["hv4_points"].tolist(), (1, 1)) - float(f["hv4_ref11"])) < 1e-12
    assert abs(O.hypervolume_2d(f["hv4_points"].tolist(), (1, 0.75)) - float(f["hv4_ref1_075"])) < 1e-12
    # moduleforhypervolume.py:94-98 known answer: union of the two boxes = 0.3125
    assert abs(O.hypervolume_2d([[0.5, 0.5], [0.25, 0.75]]) - 0.3125) < 1e-12
    assert abs(float(f["hv_twobox"]) - 0.3125) < 1e-12
    for P, ref, hv in zip(f["hv_in"], f["hv_refpt"], f["hv_out"]):
        pts = P[~np.isnan(P[:, 0])].tolist()
        assert abs(O.hypervolume_2d(pts, tuple(ref)) - hv) < 1e-9
    for P, F, sc in zip(f["cull_in"], f["cull_front"], f["cull_scalars"]):
        pts = P[~np.isnan(P[:, 0])].tolist()
        front = O.pareto_front(pts)
        ref = F[~np.isnan(F[:, 0])]
        assert len(front) == int(sc[0])
        assert np.allclose(np.array(front), ref, rtol=0, atol=0)
        got = O.front_metrics(front)
        np.testing.assert_allclose(np.array(got, dtype=np.float64), sc[1:], rtol=1e-12, atol=1e-12)
