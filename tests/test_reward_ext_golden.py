"""reward_ext.npz (tests/golden/make_golden.py::_reward_ext, generated from the reference): the TEST copies' simple_cull /
simple_cull_final, fronts longer than MAX_FRONT (truncated by random.sample under a recorded seed), and
pareto_state_data + the zero-padded Pareto-graph blocks for 1..20 archive members -- rows a24 / a26 of SURVEY.md §8.

  * the drop-in utils.py / truss2D_ENV.py / master pad (host, per env) must reproduce them row for row;
  * the batched host path (marl.pareto_graph) likewise;
  * the batched kernel truss_front is deterministic where the reference draws with random.sample: for long fronts it
    must agree on everything the two truncations share (size, both ends kept, rows from the non-dominated set, obj1
    order), and on the front itself wherever no truncation happens.
"""
import os
import random

import numpy as np
import pytest
import torch

import truss_mi355 as tm
from truss_mi355 import marl, reward as RW
import parity_common as pc
from conftest import GOLDEN


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "reward_ext.npz"))


def _rows(P):
    return P[~np.isnan(P[:, 0])]


@pytest.fixture()
def utils_variant():
    """drop-in utils / truss2D_ENV switched to a copy of the reference for one test"""
    import truss2D_ENV as ENV

    def set_(variant):
        ENV.configure(variant)
        import utils
        return utils
    yield set_
    ENV.configure(None)


@pytest.mark.parametrize("tag,variant,fn", [("testcull", "small", "simple_cull"), ("testfinal", "small", "simple_cull_final"),
                                            ("testfinal_long", "small", "simple_cull_final"), ("trainlong", None, "simple_cull"),
                                            ("testlong", "small", "simple_cull")])
def test_dropin_cull_reproduces_the_reference_row_for_row(fx, utils_variant, tag, variant, fn):
    U = utils_variant(variant)
    seed = int(fx[tag + "_seed"])
    n_trunc = 0
    for P, F, sc in zip(fx[tag + "_in"], fx[tag + "_front"], fx[tag + "_scalars"]):
        rows = _rows(P)
        random.seed(seed)
        front, max_d, dis_d, p_cd, sum_d, std_cd = getattr(U, fn)([list(r) for r in rows.tolist()])
        ref = _rows(F)[:, :5]
        assert np.array_equal(np.array(front)[:, :5], ref)            # same rows in the same order, truncation draw included
        np.testing.assert_allclose([len(front), max_d, dis_d, p_cd, sum_d, float(std_cd)], sc, rtol=1e-12, atol=1e-12)
        n_trunc += len(front) == U.MAX_FRONT and len(rows) > U.MAX_FRONT
    if tag in ("trainlong", "testlong"):
        assert n_trunc == len(fx[tag + "_in"])                       # every case did exercise the random.sample branch


def test_pareto_state_data_and_padding_match_the_reference(fx):
    import truss2D_ENV as ENV
    import master_DDPG_truss2D_MO as M
    ENV.configure(None)
    assert str(fx["pg_x_dtype"]) == "float32"
    for rec, x_ref, A_ref in zip(fx["pg_in"], fx["pg_x"], fx["pg_A"]):
        n, index = int(rec[0]), int(rec[1])
        pf = [[a, b, 0, 0] for a, b in rec[2:].reshape(-1, 2)[:n]]
        x_pf, A_pf = ENV.pareto_state_data(pf, index=index)
        assert x_pf.dtype == np.float32 and A_pf.shape == (n, n)
        x_pad, A_pad = M.pad_pareto_graph(x_pf, A_pf)
        np.testing.assert_array_equal(np.asarray(x_pad, np.float64), x_ref)
        np.testing.assert_allclose(np.asarray(A_pad, np.float64), A_ref, rtol=0, atol=1e-7)


def test_batched_pareto_graph_matches_the_reference(fx):
    rec, x_ref, A_ref = fx["pg_in"], fx["pg_x"], fx["pg_A"]
    B, P = rec.shape[0], 20
    pts = np.zeros((B, P, 4))
    pts[:, :, :2] = np.nan_to_num(rec[:, 2:].reshape(B, P, 2))
    n, index = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    x_p, A_p = marl.pareto_graph(torch.tensor(pts), torch.tensor(n), torch.tensor(index), 20)
    assert x_p.dtype == torch.float32 and A_p.dtype == torch.float32
    np.testing.assert_allclose(x_p.numpy(), x_ref, rtol=0, atol=1e-7)
    np.testing.assert_allclose(A_p.numpy(), A_ref, rtol=0, atol=1e-6)


def _front_kernel(lib, device, fx, tag, max_front):
    P_in, F_ref, sc = fx[tag + "_in"], fx[tag + "_front"], fx[tag + "_scalars"]
    B, P = P_in.shape[0], P_in.shape[1]
    pts = np.nan_to_num(P_in[:, :, :4]).copy()
    n = (~np.isnan(P_in[:, :, 0])).sum(axis=1).astype(np.int32)
    out = RW.front_hv(torch.tensor(pts, device=device), torch.tensor(n, device=device), None, max_front=max_front, lib=lib)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    return pts, n, out, F_ref, sc


def _check_front_kernel_untruncated(lib, device, fx, tag):
    pts, n, out, F_ref, sc = _front_kernel(lib, device, fx, tag, 0)
    for b in range(len(n)):
        ref = _rows(F_ref[b])
        got = pts[b][out["front_idx"][b, :out["n_front"][b]]]
        assert out["n_front"][b] == len(ref) == int(sc[b, 0])
        assert sorted(map(tuple, got)) == sorted(map(tuple, ref[:, :4]))          # the same set of rows
        if len({r[0] for r in ref}) == len(ref):                                  # obj1 order defined -> same order, same metrics
            np.testing.assert_array_equal(got, ref[:, :4])
            np.testing.assert_allclose(out["metrics"][b], [sc[b, 1], sc[b, 2], sc[b, 3], sc[b, 4], sc[b, 5]], rtol=1e-11, atol=1e-13)


def _check_front_kernel_truncated(lib, device, fx, tag, max_front):
    pts, n, out, F_ref, sc = _front_kernel(lib, device, fx, tag, max_front)
    _, _, full, _, _ = _front_kernel(lib, device, fx, tag, 0)
    for b in range(len(n)):
        ref = _rows(F_ref[b])                                            # the reference's draw: ends + 18 / 48 sampled rows
        k = out["front_idx"][b, :out["n_front"][b]]
        nd = full["front_idx"][b, :full["n_front"][b]]                  # the whole non-dominated set, obj1 order
        assert full["n_front"][b] > max_front
        assert out["n_front"][b] == max_front == len(ref)                # same size
        assert k[0] == nd[0] and k[-1] == nd[-1]                         # both ends kept ...
        assert tuple(pts[b][k[0]]) == tuple(ref[0, :4]) and tuple(pts[b][k[-1]]) == tuple(ref[-1, :4])   # ... as in the reference
        assert set(k) <= set(nd) and len(set(k)) == len(k)               # rows of the non-dominated set, once each
        assert set(map(tuple, ref[:, :4])) <= set(map(tuple, pts[b][nd]))          # (so are the reference's)
        x = pts[b][k][:, 0]
        assert np.all(np.diff(x) >= 0)                                   # obj1 order (the reference keeps its sample's order)


def test_truss_front_vs_reference_fronts_emulated(fx):
    lib = pc.emu_lib()
    for tag in ("testcull", "testfinal", "testfinal_long"):
        _check_front_kernel_untruncated(lib, "cpu", fx, tag)
    _check_front_kernel_truncated(lib, "cpu", fx, "trainlong", 20)
    _check_front_kernel_truncated(lib, "cpu", fx, "testlong", 50)


@pytest.mark.gpu
def test_truss_front_vs_reference_fronts_hip(fx):
    lib = tm.load()
    for tag in ("testcull", "testfinal", "testfinal_long"):
        _check_front_kernel_untruncated(lib, "cuda", fx, tag)
    _check_front_kernel_truncated(lib, "cuda", fx, "trainlong", 20)
    _check_front_kernel_truncated(lib, "cuda", fx, "testlong", 50)
