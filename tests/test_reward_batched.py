"""Rows a23-a25 for a batch: `truss_front` (C ABI) and the batched difference reward against the per-env
host path (drop-in utils.py / master_DDPG_truss2D_MO.difference_reward, themselves pinned by fixture G5/G6
in tests/test_dropin.py / tests/test_master_rl.py)."""
import numpy as np
import pytest
import torch

import truss_mi355 as tm
from truss_mi355 import reward as RW
import parity_common as pc
import utils as U
import master_DDPG_truss2D_MO as M


def _random_sets(rng, B, P, grid=None):
    pts = rng.uniform(0.05, 1.15, size=(B, P, 4))
    pts[:, :, 2:] = rng.uniform(0.2, 1.08, size=(B, P, 2))          # some rows infeasible
    if grid:                                                          # ties in obj1/obj2, duplicates
        pts[:, :, :2] = np.round(pts[:, :, :2] * grid) / grid
    n = rng.integers(1, P + 1, size=B).astype(np.int32)
    for b in range(B):
        pts[b, 0, 2:] = 0.5                                           # at least one feasible row (the reference
    return pts, n                                                     # crashes on an all-infeasible list)


def _check_front(lib, device, grid, max_front, seed):
    rng = np.random.default_rng(seed)
    B, P = 96, 24
    pts, n = _random_sets(rng, B, P, grid)
    ref = rng.uniform(0.85, 1.0, size=(B, 2))
    out = RW.front_hv(torch.tensor(pts, device=device), torch.tensor(n, device=device), torch.tensor(ref, device=device),
                      max_front=0, lib=lib)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    old = U.MAX_FRONT
    try:
        for b in range(B):
            rows = [list(r) for r in pts[b, :n[b]]]
            fr, max_d, dis_d, p_cd, sum_d, std_cd = U.simple_cull_final(rows)
            got = [tuple(pts[b, k]) for k in out["front_idx"][b, :out["n_front"][b]]]
            assert sorted(got) == sorted(tuple(r) for r in fr), b          # same set of rows
            assert [g[0] for g in got] == sorted(g[0] for g in got)        # sorted by obj1
            # rows with EQUAL obj1 are ordered by Python set iteration in the reference (utils.py:53-60: a set of
            # tuples, then a sort on obj1 only) -- not a defined order; the distance metrics are compared
            # only where the order is defined
            if len({g[0] for g in got}) == len(got):
                np.testing.assert_allclose(out["metrics"][b], [max_d, dis_d, p_cd, sum_d, std_cd], rtol=1e-11, atol=1e-13)
            hv = U.union_rectangles_fastest(fr, +1, -1, ref_point=list(ref[b]))
            hva = U.union_rectangles_fastest(rows, +1, -1, ref_point=list(ref[b]))
            assert abs(out["hv_front"][b] - hv) < 1e-12 and abs(out["hv_all"][b] - hva) < 1e-12, b
    finally:
        U.MAX_FRONT = old


@pytest.mark.parametrize("grid", [None, 20])
def test_front_hv_emulated(grid):
    _check_front(pc.emu_lib(), "cpu", grid, 0, 11)


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [None, 20])
def test_front_hv_hip(grid):
    _check_front(tm.load(), "cuda", grid, 0, 12)


def _check_truncation(lib, device):
    """More than max_front non-dominated rows: both ends survive, max_front rows remain, sorted by obj1, and
    the interior rows kept are those of largest crowding distance."""
    rng = np.random.default_rng(3)
    B, P, MF = 16, 40, 12
    x = np.sort(rng.uniform(0.05, 0.95, size=(B, P)), axis=1)
    y = np.sort(rng.uniform(0.05, 0.95, size=(B, P)), axis=1)[:, ::-1]
    pts = np.stack([x, y, np.full_like(x, 0.5), np.full_like(x, 0.5)], axis=2)
    n = np.full(B, P, np.int32)
    out = RW.front_hv(torch.tensor(pts, device=device), torch.tensor(n, device=device), None, max_front=MF, lib=lib)
    idx, nf = out["front_idx"].cpu().numpy(), out["n_front"].cpu().numpy()
    for b in range(B):
        assert nf[b] == MF and idx[b, 0] == 0 and idx[b, MF - 1] == P - 1
        assert list(idx[b, :MF]) == sorted(idx[b, :MF])
        d = np.hypot(np.diff(x[b]), np.diff(y[b]))
        crowd = d[:-1] + d[1:]                                            # interior rows 1..P-2
        want = set(1 + np.argsort(-crowd, kind="stable")[:MF - 2])
        assert set(idx[b, 1:MF - 1]) == want


def test_truncation_emulated():
    _check_truncation(pc.emu_lib(), "cpu")


@pytest.mark.gpu
def test_truncation_hip():
    _check_truncation(tm.load(), "cuda")


def _check_reward(lib, device):
    rng = np.random.default_rng(21)
    B, P = 48, 20
    front = np.zeros((B, P, 4)); nfr = np.zeros(B, np.int32)
    pfhv = np.zeros((B, P, 4)); npf = np.zeros(B, np.int32)
    parent = np.zeros((B, 2)); points = np.zeros((B, 3, 4)); ref = np.zeros((B, 2))
    want = []
    for b in range(B):
        k = int(rng.integers(1, 9))
        raw = rng.uniform(0.2, 1.0, size=(k, 4)); raw[:, 2:] = rng.uniform(0.3, 0.99, size=(k, 2))
        fr = U.simple_cull_final([list(r) for r in raw])[0]                 # a genuine non-dominated archive
        nfr[b] = len(fr); front[b, :len(fr)] = np.array(fr)
        m = int(rng.integers(1, 9))
        hvrows = rng.uniform(0.2, 1.0, size=(m, 4)); hvrows[:, 2:] = 0.5
        npf[b] = m; pfhv[b, :m] = hvrows
        par = fr[int(rng.integers(0, len(fr)))]
        parent[b] = par[:2]
        pts = rng.uniform(0.15, 1.05, size=(3, 4)); pts[:, 2:] = rng.uniform(0.4, 1.06, size=(3, 2))
        points[b] = pts
        ref[b] = rng.uniform(0.9, 1.0, size=2)
        want.append(M.difference_reward([list(r) for r in fr], [list(r) for r in hvrows], tuple(par[:2]),
                                        [list(p) for p in pts], list(ref[b]), m))
    t = lambda a, dt=torch.float64: torch.tensor(a, dtype=dt, device=device)
    R, GU, xm, ym = RW.difference_reward(t(front), t(nfr, torch.int32), t(pfhv), t(npf, torch.int32), t(parent), t(points),
                                         t(ref), t(npf, torch.int32), max_front=20, lib=lib)
    R, GU, xm, ym = R.cpu().numpy(), GU.cpu().numpy(), xm.cpu().numpy(), ym.cpu().numpy()
    for b in range(B):
        r0, r1, r2, gu, xmax, ymax = want[b]
        np.testing.assert_allclose(R[b], [r0, r1, r2], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose([GU[b], xm[b], ym[b]], [gu, xmax, ymax], rtol=1e-9, atol=1e-11)


def test_difference_reward_emulated():
    _check_reward(pc.emu_lib(), "cpu")


@pytest.mark.gpu
def test_difference_reward_hip():
    _check_reward(tm.load(), "cuda")


def _check_edge_cases(lib, device):
    """empty sets, all-infeasible sets, a single (1, 1) point, the full 64 rows"""
    P = 64
    pts = np.zeros((4, P, 4))
    n = np.array([0, 3, 1, 64], np.int32)
    pts[1, :3] = [[0.5, 0.5, 1.2, 0.1], [0.4, 0.6, 0.2, 1.5], [0.3, 0.7, 2.0, 2.0]]       # nothing feasible
    pts[2, 0] = [1.0, 1.0, 0.5, 0.5]
    rng = np.random.default_rng(9)
    x = np.sort(rng.uniform(0.1, 0.9, 64)); y = np.sort(rng.uniform(0.1, 0.9, 64))[::-1]
    pts[3] = np.stack([x, y, np.full(64, 0.3), np.full(64, 0.3)], axis=1)
    out = RW.front_hv(torch.tensor(pts, device=device), torch.tensor(n, device=device), None, 0, lib)
    o = {k: v.cpu().numpy() for k, v in out.items()}
    assert list(o["n_front"]) == [0, 0, 1, 64]
    assert o["hv_front"][0] == 0 and o["hv_all"][0] == 0 and np.all(o["front_idx"][0] == -1)
    assert o["hv_front"][1] == 0 and np.all(o["front_idx"][1] == -1)
    assert abs(o["hv_all"][1] - U.union_rectangles_fastest([list(r) for r in pts[1, :3]], +1, -1)) < 1e-12
    assert o["hv_front"][2] == 0 and o["hv_all"][2] == 0            # the (1, 1) special case (utils.py:279-281)
    assert list(o["front_idx"][3]) == list(range(64))
    fr = U.simple_cull_final([list(r) for r in pts[3]])
    np.testing.assert_allclose(o["metrics"][3], fr[1:6], rtol=1e-11)
    assert abs(o["hv_front"][3] - U.union_rectangles_fastest(fr[0], +1, -1)) < 1e-12
    assert np.all(np.isfinite(o["metrics"]))


def test_front_edge_cases_emulated():
    _check_edge_cases(pc.emu_lib(), "cpu")


@pytest.mark.gpu
def test_front_edge_cases_hip():
    _check_edge_cases(tm.load(), "cuda")
