"""The reference-named drop-in modules (FEM_2Dtruss, truss2D_GEN, truss2D_ENV, utils) used the way
master_DDPG_truss2D_MO.py uses the reference's: build gen_model -> Game_research04 -> ENV, take the
reset state, call _game_modify with the recorded actions, compare with the reference's recorded
outputs (tests/golden).  CPU run goes through the lane emulator; the `gpu` variant through the HIP
library."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, SCENARIOS
import parity_common as pc

# scenario -> constructor arguments of the reference's gen_model (tests/golden/make_golden.py)
ARGS = {
    "small_bridge": dict(span_x=[5] * 7, span_y=[8], tar_y=[4, 3, 2.5, 2, 2, 2.5, 3, 4], dmin=0.3, loady=-75 * 1000, ttype="bridge"),
    "small_roof": dict(span_x=[5] * 7, span_y=[8], tar_y=[4, 3, 2.5, 2, 2, 2.5, 3, 4], dmin=0.3, loady=-120 * 1000, ttype="roof"),
    "large_bridge": dict(span_x=[5] * 15, span_y=[6], tar_y=[3.00, 2.75, 2.50, 2.25, 2.25, 2.00, 2.00, 2.00, 2.00, 2.00, 2.00,
                                                               2.25, 2.25, 2.50, 2.75, 3.00], dmin=0.3, loady=-7500, ttype="bridge"),
    "large_roof": dict(span_x=[5] * 15, span_y=[6], tar_y=[3.00, 2.75, 2.50, 2.25, 2.25, 2.00, 2.00, 2.00, 2.00, 2.00, 2.00,
                                                             2.25, 2.25, 2.50, 2.75, 3.00], dmin=0.3, loady=-8000, ttype="roof"),
    "train0": dict(span_x=[4.0, 3.0, 5.0, 3.0, 5.0], span_y=[5], tar_y=[1.0, 1.5, 2.0, 2.0, 1.5, 1.0], dmin=0.2, loady=-100000, ttype="roof"),
    "train3": dict(span_x=[4.0, 3.0, 5.0, 3.0, 5.0], span_y=[5], tar_y=[1.0, 3.0, 2.0, 2.0, 3.0, 1.0], dmin=0.2, loady=-100000, ttype="bridge"),
    "train_eval": dict(span_x=[5.0] * 7, span_y=[8], tar_y=[4.0, 3.0, 2.5, 2.0, 2.0, 2.5, 3.0, 4.0], dmin=0.3, loady=-120000, ttype="roof"),
}


class _Coin:
    def __init__(self):
        self.value = 0.0

    def random(self):
        return self.value


def _replay(name, lib, n_trans):
    import FEM_2Dtruss
    import truss2D_GEN
    import truss2D_ENV
    FEM_2Dtruss._LIB = lib
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    a = ARGS[name]
    truss2D_ENV.configure(SCENARIOS[name][1])
    coin = _Coin()
    saved_random = truss2D_ENV.random
    truss2D_ENV.random = coin
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            gm = truss2D_GEN.gen_model(len(a["span_x"]) + 1, 2, a["span_x"], a["span_y"], a["tar_y"], a["dmin"], 0, a["loady"],
                                       a["ttype"], 1, None)
            game = truss2D_ENV.Game_research04(50, gm, 2)
            env = truss2D_ENV.ENV(game)
            env.reset()
        m = gm.model
        # reset-time integers and analysis
        assert m.nsc == f["nsc"].tolist() and m.tnsc == f["tnsc"].tolist() and m.ttnsc == f["ttnsc"].tolist()
        assert m.ndof == int(f["ndof"])
        assert pc.rel(np.array(m.d).reshape(-1), f["reset_d"]) < 1e-5
        assert pc.rel(np.array([e.e_q[0][0] for e in m.elements]), f["reset_q0"]) < 1e-5
        assert [int(e.iscompress) for e in m.elements] == f["reset_comp"].tolist()
        assert pc.rel(game.int_obj1, f["int_obj1"]) < 1e-6 and pc.rel(game.int_obj2, f["int_obj2"]) < 1e-6
        S0 = game._game_get_1_state()
        names = ["x_n", "A_n", "A_s", "A_n_ts", "A_n_cs", "mask", "x_pf", "A_pf", "nN_x_n", "nN_x_e", "nC_e"]
        assert len(S0) == 11
        for k, nm in enumerate(names):
            ref = f["reset_" + nm]
            assert S0[k].shape == ref.shape and S0[k].dtype == np.float32, nm
            scale = max(1.0, float(np.abs(ref).max()))
            assert float(np.abs(S0[k] - ref).max()) / scale < 1e-5, nm
        # transitions (G4): same call sequence as the generator, so the stale move ranges line up
        for t in range(n_trans):
            geo, tac = f["tr_in_geo"][t].copy(), f["tr_in_topo"][t].copy()
            coin.value = float(f["tr_coin"][t])
            stale = np.array([float(n.max_up) for n in m.nodes])
            np.testing.assert_allclose(stale, f["tr_stale_max_up"][t], rtol=0, atol=5e-7)
            point, St = game._game_modify(f["tr_in_node"][t].copy(), f["tr_in_elem"][t].copy(), f["reset_nC_e"].copy(),
                                          [geo, tac])
            assert np.array_equal(geo, f["tr_clamped_geo"][t]) and np.array_equal(tac, f["tr_clamped_topo"][t])
            assert len(point) == 4 and all(isinstance(p, np.float32) for p in point)
            np.testing.assert_allclose(np.array(point), f["tr_point"][t], rtol=1e-5, atol=1e-7)
            assert len(St) == 11 and St[6] is None and St[7] is None
            assert np.array_equal(St[8][:, 1], f["tr_out_nN_x_n"][t][:, 1])          # heights bit-exact
            assert np.array_equal(St[9][:, 0], f["tr_out_nN_x_e"][t][:, 0])          # sections bit-exact
            for k, nm in ((0, "x_n"), (1, "A_n"), (2, "A_s"), (3, "A_n_ts"), (4, "A_n_cs"), (5, "mask"), (8, "nN_x_n"),
                          (9, "nN_x_e"), (10, "nC_e")):
                ref = f["tr_out_" + nm][t]
                scale = np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1.0)
                assert float((np.abs(St[k] - ref) / scale).max()) < 1e-5, (nm, t)
            assert [int(e.section_no) for e in m.elements] == f["tr_fem_sec"][t].tolist()
        game.step()
        assert game.game_step == 2
        env.check_over()
        assert env.over == 0
        game.done_counter = 1
        env.check_over()
        assert env.over == 1
    finally:
        truss2D_ENV.random = saved_random
        truss2D_ENV.configure(None)
        FEM_2Dtruss._LIB = None


@pytest.mark.parametrize("name", ["train0", "train3", "small_roof", "large_bridge"])
def test_dropin_modules_emulated(name):
    _replay(name, pc.emu_lib(), 24)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ARGS))
def test_dropin_modules_gpu(name):
    import truss_mi355 as tm
    assert torch.cuda.is_available()
    _replay(name, tm.load(), 96)


def test_dropin_utils_match_reference_answers():
    """utils.simple_cull / union_rectangles_fastest against answers recorded from the reference."""
    import random
    import utils
    f = np.load(os.path.join(GOLDEN, "reward.npz"))
    OPEN, CLOSE = +1, -1
    assert abs(utils.union_rectangles_fastest(f["hv4_points"].tolist(), OPEN, CLOSE, ref_point=[1, 1]) - float(f["hv4_ref11"])) < 1e-12
    assert abs(utils.union_rectangles_fastest(f["hv4_points"].tolist(), OPEN, CLOSE, ref_point=[1, 0.75]) - float(f["hv4_ref1_075"])) < 1e-12
    assert abs(utils.union_rectangles_fastest([[0.5, 0.5], [0.25, 0.75]], OPEN, CLOSE) - 0.3125) < 1e-12
    for P, ref, hv in zip(f["hv_in"], f["hv_refpt"], f["hv_out"]):
        pts = P[~np.isnan(P[:, 0])].tolist()
        assert abs(utils.union_rectangles_fastest(pts, OPEN, CLOSE, ref_point=list(ref)) - hv) < 1e-9
    for P, F, sc in zip(f["cull_in"], f["cull_front"], f["cull_scalars"]):
        pts = P[~np.isnan(P[:, 0])].tolist()
        random.seed(5)
        front, max_d, dis_d, p_cd, sum_d, std_cd = utils.simple_cull([list(r) for r in pts])
        ref = F[~np.isnan(F[:, 0])]
        assert np.array_equal(np.array(front), ref)
        np.testing.assert_allclose([len(front), max_d, dis_d, p_cd, sum_d, float(std_cd)], sc, rtol=1e-12, atol=1e-12)
    out7 = utils.simple_cull([[0.5, 0.5, 0, 0, 0], [0.4, 0.6, 0, 0, 1]], True)
    assert len(out7) == 7 and out7[6] is out7[0]


def test_structure_dump_format_round_trip(tmp_path):
    """Row f-3: `gen_model.savetxt` writes, byte for byte, what the reference writes for the same design
    (fixture = a dump produced by the reference's own savetxt, tests/golden/make_golden_formats.py), and
    `read_src` (the reader of the reference's render scripts) loads it back."""
    import json
    import FEM_2Dtruss
    import truss2D_GEN
    FEM_2Dtruss._LIB = pc.emu_lib()
    try:
        want = json.load(open(os.path.join(GOLDEN, "structure_small_bridge.json")))
        ref_txt = open(os.path.join(GOLDEN, "structure_small_bridge.txt"), newline="").read()
        with contextlib.redirect_stdout(io.StringIO()):
            gm = truss2D_GEN.gen_model(*want["args"])
            fresh = truss2D_GEN.gen_model(*want["args"])
        # reader: the reference's dump -> model
        fresh.read_src(os.path.join(GOLDEN, "structure_small_bridge.txt"))
        assert [n.coord[1] for n in fresh.model.nodes] == want["y"]
        assert [e.section_no for e in fresh.model.elements] == want["sec"]
        # writer: the same design set by hand -> identical bytes (CRLF line ends included)
        for n, y in zip(gm.model.nodes, want["y"]):
            if n.top_node == 1:              # untouched coordinates keep the builder's (integer) literals
                n.coord[1] = y
        for e, s in zip(gm.model.elements, want["sec"]):
            e.section_no = s
            e.area = gm.truss[s][0] * 1e-4
            e.set_i(gm.truss[s][1] * 1e-8)
        out = tmp_path / "dump.txt"
        gm.savetxt(str(out))
        assert open(out, newline="").read() == ref_txt
    finally:
        FEM_2Dtruss._LIB = None
