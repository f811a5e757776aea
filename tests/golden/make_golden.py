#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE (kupc25648/MOP-truss-MARL, mounted read-only at
/root/reference) in this container and records inputs/outputs of its hot path as small .npz
fixtures under tests/golden/.  The reference never travels; only these data files do.

What is pinned:
  G1  threebar.npz      textbook 3-bar truss ("Example3.8 Pg107", FEM_2Dtruss.py:474-558, data only)
  G2  <scenario>.npz    reset-time topology/DOF integers + first FEM + reset observation
  G3/G4 <scenario>.npz  chains of Game_research04._game_modify transitions with seeded random actions
                        (inputs, hidden stale move ranges, coin, y', sec', point, 9 state arrays,
                        and the FEM internals K, P, d, q0, stress ratio, flags, U, reactions)
  G5/G6 reward.npz      utils.simple_cull / union_rectangles_fastest answers + reward formula inputs

Environment facts recorded in every fixture's `meta`: numpy version (the reference pins 1.23.5,
this container has 2.2.6 -> NEP-50 scalar promotion is what the fixtures embed), and the fact that
`spektral.utils.degree_power` (spektral is not installed) is replaced by the 6-line stand-in below,
restating its published behaviour: diag(rowsum(A)**k) with inf -> 0.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
This script is the only thing in the repo that touches /root/reference, and it is never imported
by tests, bench.py or the product.
"""
import os
import sys
import json
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

VARIANTS = {
    # variant -> (reference code dir, list of scenarios)
    "train": ("train/code", ["train0", "train3", "train_eval"]),
    "small": ("test/00_small_bridge/code", ["small_bridge", "small_roof"]),
    "large": ("test/02_large_bridge/code", ["large_bridge", "large_roof"]),
}

SCENARIOS = {
    # name: num_x, span_x, span_y, tar_y, dmin, loadx, loady, type, end_step
    "small_bridge": dict(span_x=[5] * 7, span_y=[8], tar_y=[4, 3, 2.5, 2, 2, 2.5, 3, 4], dmin=0.3,
                         loady=-75 * 1000, ttype="bridge"),
    "small_roof": dict(span_x=[5] * 7, span_y=[8], tar_y=[4, 3, 2.5, 2, 2, 2.5, 3, 4], dmin=0.3,
                       loady=-120 * 1000, ttype="roof"),
    "large_bridge": dict(span_x=[5] * 15, span_y=[6],
                         tar_y=[3.00, 2.75, 2.50, 2.25, 2.25, 2.00, 2.00, 2.00, 2.00, 2.00, 2.00, 2.25,
                                2.25, 2.50, 2.75, 3.00], dmin=0.3, loady=-7500, ttype="bridge"),
    "large_roof": dict(span_x=[5] * 15, span_y=[6],
                       tar_y=[3.00, 2.75, 2.50, 2.25, 2.25, 2.00, 2.00, 2.00, 2.00, 2.00, 2.00, 2.25,
                              2.25, 2.50, 2.75, 3.00], dmin=0.3, loady=-8000, ttype="roof"),
    "train0": dict(span_x=[4.0, 3.0, 5.0, 3.0, 5.0], span_y=[5], tar_y=[1.0, 1.5, 2.0, 2.0, 1.5, 1.0],
                   dmin=0.2, loady=-100000, ttype="roof"),
    "train3": dict(span_x=[4.0, 3.0, 5.0, 3.0, 5.0], span_y=[5], tar_y=[1.0, 3.0, 2.0, 2.0, 3.0, 1.0],
                   dmin=0.2, loady=-100000, ttype="bridge"),
    "train_eval": dict(span_x=[5.0] * 7, span_y=[8], tar_y=[4.0, 3.0, 2.5, 2.0, 2.0, 2.5, 3.0, 4.0],
                       dmin=0.3, loady=-120000, ttype="roof"),
}

N_TRANS = 96


def _install_spektral_standin():
    import types
    import numpy as np
    sp = types.ModuleType("spektral")
    spu = types.ModuleType("spektral.utils")

    def degree_power(A, k):
        with np.errstate(divide="ignore"):
            d = np.power(np.array(A.sum(1)), k).ravel()
        d[np.isinf(d)] = 0.0
        return np.diag(d)

    spu.degree_power = degree_power
    sp.utils = spu
    sys.modules["spektral"] = sp
    sys.modules["spektral.utils"] = spu


def _meta():
    import numpy as np
    return json.dumps({
        "numpy": np.__version__,
        "reference_numpy_pin": "1.23.5",
        "degree_power": "stand-in: diag(rowsum(A)**k), inf->0 (spektral 1.2.0 not installed)",
        "generator": "tests/golden/make_golden.py",
    })


def _tcode(v):
    """0 = python int, 1 = python float, 2 = np.float32, 3 = np.float64/other."""
    import numpy as np
    if isinstance(v, (bool, int)):
        return 0
    if isinstance(v, np.float32):
        return 2
    if isinstance(v, float) and not isinstance(v, np.floating):
        return 1
    return 3


def _fem_record(gm):
    import numpy as np
    m = gm.model
    rec = {}
    rec["K"] = np.array(m.ssm, dtype=np.float64)
    rec["P"] = np.array(m.jlv, dtype=np.float64).reshape(-1)
    rec["d"] = np.array(m.d, dtype=np.float64).reshape(-1)
    rec["q0"] = np.array([float(e.e_q[0][0]) for e in m.elements])
    rec["sr"] = np.array([float(e.prop_yeield) for e in m.elements])
    rec["comp"] = np.array([int(e.iscompress) for e in m.elements], dtype=np.int8)
    rec["length"] = np.array([float(e.length) for e in m.elements])
    rec["U"] = np.float64(np.asarray(m.U_full).reshape(-1)[0])
    rec["r"] = np.array([np.nan if v is None else float(v) for v in m.r])
    rec["dnode"] = np.array([[float(n.global_d[0][0]), float(n.global_d[1][0])] for n in m.nodes])
    rec["max_up"] = np.array([float(n.max_up) for n in m.nodes])
    rec["max_down"] = np.array([float(n.max_down) for n in m.nodes])
    rec["ycoord"] = np.array([float(n.coord[1]) for n in m.nodes])
    rec["ytype"] = np.array([_tcode(n.coord[1]) for n in m.nodes], dtype=np.int8)
    rec["mutype"] = np.array([_tcode(n.max_up) for n in m.nodes], dtype=np.int8)
    rec["mdtype"] = np.array([_tcode(n.max_down) for n in m.nodes], dtype=np.int8)
    rec["sec"] = np.array([int(e.section_no) for e in m.elements], dtype=np.int32)
    rec["area"] = np.array([float(e.area) for e in m.elements])
    return rec


class _Coin:
    """Replaces the `random` module inside truss2D_ENV so the symmetry coin is an explicit input."""

    def __init__(self):
        self.value = 0.0
        self.calls = 0

    def random(self):
        self.calls += 1
        return self.value

    def seed(self, *_a):
        pass


def _draw_actions(rng, N, mode):
    import numpy as np
    if mode == 0:      # plain uniform
        g = rng.random((N, 2))
        t = rng.random((N, 3))
    elif mode == 1:    # out-of-range values exercise the in-place clamp
        g = rng.random((N, 2)) * 2.0 - 0.5
        t = rng.random((N, 3)) * 2.0 - 0.5
    elif mode == 2:    # coarse grid -> many exact argmax ties
        g = rng.integers(0, 3, (N, 2)) / 2.0
        t = rng.integers(0, 3, (N, 3)) / 2.0
    elif mode == 3:    # big moves: push nodes into the repair rules
        g = np.where(rng.random((N, 2)) < 0.5, 1.0, rng.random((N, 2)))
        t = rng.random((N, 3))
    else:              # small moves
        g = rng.random((N, 2)) * 0.1
        t = rng.random((N, 3))
    return g.astype(np.float32), t.astype(np.float32)


def worker(variant, outdir):
    import io
    import contextlib
    import numpy as np
    os.environ.setdefault("MPLBACKEND", "Agg")
    codedir = os.path.join(REF, VARIANTS[variant][0])
    os.chdir(codedir)
    sys.path.insert(0, codedir)
    _install_spektral_standin()
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        import truss2D_GEN as GEN
        import truss2D_ENV as ENVM
    coin = _Coin()
    has_sym = hasattr(ENVM, "random")
    if has_sym:
        ENVM.random = coin

    for sname in VARIANTS[variant][1]:
        sc = SCENARIOS[sname]
        num_x = len(sc["span_x"]) + 1
        with contextlib.redirect_stdout(sink):
            gm = GEN.gen_model(num_x, 2, sc["span_x"], sc["span_y"], sc["tar_y"], sc["dmin"], 0,
                               sc["loady"], sc["ttype"], 1, None)
            game = ENVM.Game_research04(50, gm, 2)
            env = ENVM.ENV(game)
            env.reset()
        m = gm.model
        N = len(m.nodes)
        E = len(m.elements)
        out = {"meta": _meta(), "variant": variant, "scenario": sname}
        # ---- G2: reset-time integers ----
        name2idx = {n.name: i for i, n in enumerate(m.nodes)}
        out["conn"] = np.array([[name2idx[e.nodes[0].name], name2idx[e.nodes[1].name]] for e in m.elements],
                               dtype=np.int32)
        out["res"] = np.array([n.res for n in m.nodes], dtype=np.int8)
        out["top"] = np.array([n.top_node for n in m.nodes], dtype=np.int8)
        out["pair"] = np.array([name2idx[n.vertical_pair[0].name] for n in m.nodes], dtype=np.int32)
        out["npairs"] = np.array([len(n.vertical_pair) for n in m.nodes], dtype=np.int32)
        out["target0"] = np.array([float(n.target) for n in m.nodes])
        out["has_loady"] = np.array([float(n.has_loady) for n in m.nodes])
        out["nload"] = np.array([len(n.loads) for n in m.nodes], dtype=np.int32)
        out["xcoord"] = np.array([float(n.coord[0]) for n in m.nodes])
        out["nsc"] = np.array(m.nsc, dtype=np.int32)
        out["tnsc"] = np.array(m.tnsc, dtype=np.int32)
        out["ttnsc"] = np.array(m.ttnsc, dtype=np.int32)
        out["ndof"] = np.int32(m.ndof)
        out["sections"] = np.array(gm.truss, dtype=np.float64)
        out["E_mod"] = np.float64(gm.YoungM)
        out["y_max"] = np.float64(gm.y_max)
        out["y_min"] = np.float64(gm.y_min)
        out["d_min"] = np.float64(gm.d_min)
        out["max_deformation"] = np.float64(gm.max_deformation)
        out["loady"] = np.float64(sc["loady"])
        out["is_roof"] = np.int8(sc["ttype"] == "roof")
        out["int_obj1"] = np.float32(game.int_obj1)
        out["int_obj2"] = np.float32(game.int_obj2)
        # ---- G2: reset observation + FEM ----
        with contextlib.redirect_stdout(sink):
            S0 = game._game_get_1_state()
        for k, nm in enumerate(["x_n", "A_n", "A_s", "A_n_ts", "A_n_cs", "mask", "x_pf", "A_pf", "nN_x_n",
                                "nN_x_e", "nC_e"]):
            out["reset_" + nm] = np.asarray(S0[k])
        for k, v in _fem_record(gm).items():
            out["reset_" + k] = v

        # ---- G3/G4: transitions ----
        rng = np.random.default_rng(1000 + sum(map(ord, sname)))
        archive = [(np.array(S0[8]), np.array(S0[9]), np.array(S0[10]))]
        recs = {}

        def push(key, val):
            recs.setdefault(key, []).append(np.asarray(val))

        for t in range(N_TRANS):
            # parent: mostly the most recent child (long chains drift far from the reset design),
            # sometimes a random archived design (exercises the stale move-range quirk)
            if t == 0 or rng.random() < 0.7:
                pn, pe, pc = archive[-1]
            else:
                pn, pe, pc = archive[int(rng.integers(0, len(archive)))]
            a_geo, a_topo = _draw_actions(rng, N, int(rng.integers(0, 5)))
            coin.value = float(rng.random())
            if t % 17 == 5:
                coin.value = 0.5   # boundary of `random.random() >= 0.5`
            push("in_node", pn)
            push("in_elem", pe)
            push("in_geo", a_geo.copy())
            push("in_topo", a_topo.copy())
            push("coin", np.float64(coin.value))
            push("stale_max_up", np.array([float(n.max_up) for n in m.nodes]))
            push("stale_max_down", np.array([float(n.max_down) for n in m.nodes]))
            push("stale_mu_type", np.array([_tcode(n.max_up) for n in m.nodes], dtype=np.int8))
            push("stale_md_type", np.array([_tcode(n.max_down) for n in m.nodes], dtype=np.int8))
            g2, t2 = a_geo.copy(), a_topo.copy()
            with contextlib.redirect_stdout(sink):
                point, St = game._game_modify(pn.copy(), pe.copy(), pc.copy(), [g2, t2])
            push("clamped_geo", g2)
            push("clamped_topo", t2)
            push("point", np.array([np.float32(p) for p in point], dtype=np.float32))
            for k, nm in [(0, "x_n"), (1, "A_n"), (2, "A_s"), (3, "A_n_ts"), (4, "A_n_cs"), (5, "mask"),
                          (8, "nN_x_n"), (9, "nN_x_e"), (10, "nC_e")]:
                push("out_" + nm, St[k])
            for k, v in _fem_record(gm).items():
                push("fem_" + k, v)
            push("target", np.array([float(n.target) for n in m.nodes]))
            archive.append((np.array(St[8]), np.array(St[9]), np.array(St[10])))
        for k, v in recs.items():
            out["tr_" + k] = np.stack(v)
        out["has_sym"] = np.int8(has_sym)
        out["coin_calls"] = np.int32(coin.calls)
        np.savez_compressed(os.path.join(outdir, sname + ".npz"), **out)
        print("wrote", sname, "N", N, "E", E, "ndof", int(m.ndof), file=sys.stderr)

    if variant == "train":
        _threebar(outdir)
        _reward(outdir, codedir_test=os.path.join(REF, "test/00_small_bridge/code"))
        _reward_ext(outdir, codedir_test=os.path.join(REF, "test/00_small_bridge/code"), ENVM=ENVM)


def _threebar(outdir):
    """G1: textbook example, data from the commented block FEM_2Dtruss.py:474-558."""
    import numpy as np
    from FEM_2Dtruss import Load, Node, Element, Model
    l1 = Load(); l1.set_name(1); l1.set_size(0, -300)
    l2 = Load(); l2.set_name(2); l2.set_size(150, 0)
    coords = [(144, 192), (0, 0), (144, 0), (288, 0)]
    ress = [(0, 0), (1, 1), (1, 1), (1, 1)]
    nodes = []
    for i, (c, r) in enumerate(zip(coords, ress)):
        n = Node(); n.set_name(i + 1); n.set_coord(*c); n.set_res(*r); nodes.append(n)
    nodes[0].set_load(l1); nodes[0].set_load(l2)
    els = []
    for i, (a, b, area) in enumerate([(1, 0, 8), (2, 0, 6), (3, 0, 8)]):
        e = Element(); e.set_name(i + 1); e.set_nodes(nodes[a], nodes[b]); e.set_em(29000); e.set_i(1000)
        e.set_area(area); els.append(e)
    m = Model()
    m.add_load(l1); m.add_load(l2)
    for n in nodes:
        m.add_node(n)
    for e in els:
        m.add_element(e)
    m.gen_all()
    np.savez_compressed(
        os.path.join(outdir, "threebar.npz"), meta=_meta(),
        coords=np.array(coords, dtype=np.float64), res=np.array(ress, dtype=np.int8),
        conn=np.array([[1, 0], [2, 0], [3, 0]], dtype=np.int32), em=np.float64(29000),
        area=np.array([8.0, 6.0, 8.0]), load=np.array([[150.0, -300.0], [0, 0], [0, 0], [0, 0]]),
        nsc=np.array(m.nsc, dtype=np.int32), tnsc=np.array(m.tnsc, dtype=np.int32),
        ttnsc=np.array(m.ttnsc, dtype=np.int32), ndof=np.int32(m.ndof),
        K=np.array(m.ssm), P=np.array(m.jlv, dtype=np.float64).reshape(-1), d=np.array(m.d).reshape(-1),
        q0=np.array([float(e.e_q[0][0]) for e in els]),
        q=np.array([np.asarray(e.e_q, dtype=np.float64).reshape(-1) for e in els]),
        r=np.array([np.nan if v is None else float(v) for v in m.r]),
        sr=np.array([float(e.prop_yeield) for e in els]),
        comp=np.array([int(e.iscompress) for e in els], dtype=np.int8),
        U=np.float64(np.asarray(m.U_full).reshape(-1)[0]))
    print("wrote threebar", file=sys.stderr)


def _reward(outdir, codedir_test):
    """G5/G6: Pareto cull + hypervolume answers from utils.py (train copy; MAX_FRONT=20) and the
    difference-reward formula of master_DDPG_truss2D_MO.py:263-368 evaluated with the reference's own
    utils functions (the master itself needs tensorflow and cannot be imported)."""
    import random
    import numpy as np
    import utils as U
    rng = np.random.default_rng(77)
    OPEN, CLOSE = +1, -1
    out = {"meta": _meta()}
    # G6 known answers
    pts = [[0.5, 0.5, 0.8, 0.8], [1.0, 1.0, 0.8, 0.8], [0.25, 0.75, 0.8, 0.8], [0.75, 0.25, 0.8, 0.8]]
    out["hv4_points"] = np.array(pts)
    out["hv4_ref11"] = np.float64(U.union_rectangles_fastest(pts, OPEN, CLOSE, ref_point=[1, 1]))
    out["hv4_ref1_075"] = np.float64(U.union_rectangles_fastest(pts, OPEN, CLOSE, ref_point=[1, 0.75]))
    out["hv_twobox"] = np.float64(U.union_rectangles_fastest([[0.5, 0.5, 0, 0], [0.25, 0.75, 0, 0]], OPEN, CLOSE))
    # random hypervolume cases (incl. points > 1 which the function clips, and duplicates)
    hv_in, hv_ref, hv_out = [], [], []
    for t in range(200):
        n = int(rng.integers(1, 25))
        P = rng.random((n, 4))
        if t % 5 == 0:
            P[:, :2] *= 1.3
        if t % 7 == 0 and n > 2:
            P[1] = P[0]
        if t % 11 == 0:
            P = np.round(P, 1)
        ref = [1, 1] if t % 3 else [float(min(1, 0.2 + rng.random())), float(min(1, 0.2 + rng.random()))]
        Pl = np.minimum(P, 1.0).tolist() if t % 5 == 0 else P.tolist()
        hv = U.union_rectangles_fastest(Pl, OPEN, CLOSE, ref_point=ref)
        pad = np.full((25, 4), np.nan); pad[:n] = np.array(Pl)
        hv_in.append(pad); hv_ref.append(ref); hv_out.append(hv)
    out["hv_in"] = np.array(hv_in); out["hv_refpt"] = np.array(hv_ref, dtype=np.float64)
    out["hv_out"] = np.array(hv_out, dtype=np.float64)
    # simple_cull cases (n <= MAX_FRONT so the random.sample truncation is not reached)
    sc_in, sc_front, sc_scal = [], [], []
    for t in range(200):
        n = int(rng.integers(1, 19))
        P = rng.random((n, 4))
        P[:, 2:] *= 1.15            # some infeasible rows
        if t % 4 == 0:
            P = np.round(P, 1)       # ties in the objectives
        if (P[:, 2] <= 1).sum() == 0 or ((P[:, 2] <= 1) & (P[:, 3] <= 1)).sum() == 0:
            P[0, 2:] = 0.5
        idx = np.arange(n, dtype=np.float64)[:, None]
        rows = np.hstack([P, idx]).tolist()
        random.seed(5)
        front, max_d, dis_d, p_cd, sum_d, std_cd = U.simple_cull([list(r) for r in rows])
        pad = np.full((18, 5), np.nan); pad[:n] = np.array(rows)
        fpad = np.full((18, 5), np.nan); fpad[:len(front)] = np.array(front)
        sc_in.append(pad); sc_front.append(fpad)
        sc_scal.append([len(front), max_d, dis_d, p_cd, sum_d, float(std_cd)])
    out["cull_in"] = np.array(sc_in); out["cull_front"] = np.array(sc_front)
    out["cull_scalars"] = np.array(sc_scal, dtype=np.float64)
    # G5: the difference-reward block of run() (master_DDPG_truss2D_MO.py:263-368) evaluated with the
    # reference's own simple_cull / union_rectangles_fastest.  The master cannot be imported
    # (tensorflow), so its formulas are restated here next to the line they come from.
    rb_in, rb_out = [], []
    for t in range(120):
        nf = int(rng.integers(1, 7))
        front_no = (rng.random((nf, 2)) * 0.9 + 0.05)
        front_no = [[float(a), float(b), 0.0, 0.0] for a, b in front_no]
        Pf_HV = [list(r) for r in front_no] if t % 3 else [[1, 1, 0, 0]]
        n_pf = len(Pf_HV)
        parent = Pf_HV[int(rng.integers(0, n_pf))][:2]
        pts = rng.random((3, 4))
        pts[:, 2:] *= 1.3                       # some infeasible agents
        if t % 5 == 0:
            pts[:, :2] *= 1.2                   # objectives above 1
        points = [[np.float32(v) for v in row] for row in pts]
        ref_points = [float(min(1, 0.2 * (1 + t % 5))), float(min(1, 0.2 * (1 + (t // 5) % 5)))]
        feas = [(p[0] <= 1 and p[1] <= 1 and p[2] <= 1 and p[3] <= 1) for p in points]
        random.seed(11)
        ff = []
        for i in range(3):                      # :267-287 leave-one-out fronts
            f_i = [e for e in front_no]
            for j in range(3):
                if j != i and feas[j]:
                    f_i.append(points[j])
            ff.append(U.simple_cull(f_i))
        f_all = [e for e in front_no]           # :291-305
        for j in range(3):
            if feas[j]:
                f_all.append(points[j])
        front, max_d, dis_d, p_cd, sum_distance, std_cd = U.simple_cull(f_all)
        hv_i = [U.union_rectangles_fastest(ff[i][0], OPEN, CLOSE, ref_point=ref_points) for i in range(3)]   # :311-313
        hyperV = U.union_rectangles_fastest(front, OPEN, CLOSE, ref_point=ref_points)                        # :314
        compareV = U.union_rectangles_fastest(Pf_HV, OPEN, CLOSE, ref_point=ref_points)                      # :316
        Real_compareV = U.union_rectangles_fastest(Pf_HV, OPEN, CLOSE, ref_point=[1, 1])                     # :317
        hv_i = [max([0, h - compareV]) for h in hv_i]                                                        # :324-332
        hyperV = max([0, hyperV - compareV])                                                                 # :336
        w = [0, 0, 0]
        if feas[0]:
            w[0] = (1) * max([0, (parent[0] - points[0][0])]) + (0) * max([0, (parent[1] - points[0][0])])   # :348
        if feas[1]:
            w[1] = (1 / 2) * max([0, (parent[0] - points[1][0])]) + (1 / 2) * max([0, (parent[1] - points[1][0])])  # :350
        if feas[2]:
            w[2] = (0) * max([0, (parent[0] - points[2][0])]) + (1) * max([0, (parent[1] - points[2][0])])   # :352
        R = []
        for i in range(3):                                                                                   # :365-367
            R.append(0.25 * w[i] / (max([0.25, Real_compareV]) * n_pf) + 0.25 * (hyperV - hv_i[i]) / (max([0.25, Real_compareV]) * n_pf)
                     + 10 * (Real_compareV / n_pf) - 0.05 * max([0, min([1, std_cd])]) / n_pf
                     + 0.05 * sum_distance / (2 * (max([0.25, Real_compareV]) ** 0.5) * n_pf))
        G_U = (20 * Real_compareV / n_pf) - (1 * std_cd) / n_pf + 1 * sum_distance / n_pf                    # :368
        pad = np.full((6, 4), np.nan); pad[:nf] = np.array(front_no)
        padh = np.full((6, 4), np.nan); padh[:n_pf] = np.array(Pf_HV, dtype=np.float64)
        rb_in.append(np.concatenate([pad.ravel(), padh.ravel(), np.array(parent, dtype=np.float64),
                                     np.array(points, dtype=np.float64).ravel(), np.array(ref_points)]))
        rb_out.append([float(R[0]), float(R[1]), float(R[2]), float(G_U)])
    out["rb_in"] = np.array(rb_in)
    out["rb_out"] = np.array(rb_out, dtype=np.float64)
    np.savez_compressed(os.path.join(outdir, "reward.npz"), **out)
    print("wrote reward", file=sys.stderr)


def _reward_ext(outdir, codedir_test, ENVM):
    """reward_ext.npz -- what reward.npz does not reach (round-2 additions):
      * the TEST copies' utils.simple_cull / simple_cull_final (test/00_small_bridge/code/utils.py:11-403, MAX_FRONT 50,
        wider edge lists), imported under another module name next to the train copy;
      * fronts LONGER than MAX_FRONT: the reference truncates with random.sample (train utils.py:118-123) -- recorded
        under a fixed random.seed, which a port that consumes the Python RNG the same way reproduces row for row;
      * truss2D_ENV.pareto_state_data(pf, index) (train copy :19-38) for 1..20 archive members, every index, and the
        zero-padded [MAX_PARETO_SIZE] blocks of master_DDPG_truss2D_MO.py:475-593 (np.block, restated here: the master
        needs tensorflow and cannot be imported)."""
    import contextlib
    import importlib.util
    import io
    import random
    import numpy as np
    import utils as U                                   # train copy (cwd / sys.path[0] = train/code)
    spec = importlib.util.spec_from_file_location("utils_testcopy", os.path.join(codedir_test, "utils.py"))
    UT = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):     # the file runs a demo at import (utils.py:533-568)
        spec.loader.exec_module(UT)
    rng = np.random.default_rng(2024)
    out = {"meta": _meta()}

    def cases(n_cases, n_lo, n_hi, convex):
        res = []
        for t in range(n_cases):
            n = int(rng.integers(n_lo, n_hi))
            if convex:      # many mutually non-dominated rows: a noisy convex curve obj2 ~ (1 - sqrt(obj1))^2
                a = np.sort(rng.random(n)) * 0.9 + 0.05
                P = np.stack([a, (1 - np.sqrt(a)) ** 2 + rng.random(n) * 0.002, rng.random(n) * 0.9, rng.random(n) * 0.9], axis=1)
                P = P[rng.permutation(n)]
            else:
                P = rng.random((n, 4))
                P[:, 2:] *= 1.15
                if t % 4 == 0:
                    P = np.round(P, 1)
                if ((P[:, 2] <= 1) & (P[:, 3] <= 1)).sum() == 0:
                    P[0, 2:] = 0.5
            res.append(np.hstack([P, np.arange(n, dtype=np.float64)[:, None]]))
        return res

    def record(tag, fn, rows_list, seed, width):
        ins, fronts, scal = [], [], []
        for rows in rows_list:
            random.seed(seed)
            front, max_d, dis_d, p_cd, sum_d, std_cd = fn([list(r) for r in rows.tolist()])
            pad = np.full((width, 5), np.nan); pad[:len(rows)] = rows
            f = np.array([list(r[:6]) + [np.nan] * (6 - len(r[:6])) for r in front], dtype=np.float64)   # [obj1, obj2, c1, c2, id, (distance)]
            fpad = np.full((width, 6), np.nan); fpad[:len(f)] = f
            ins.append(pad); fronts.append(fpad)
            scal.append([len(front), max_d, dis_d, p_cd, sum_d, float(std_cd)])
        out[tag + "_in"] = np.array(ins); out[tag + "_front"] = np.array(fronts)
        out[tag + "_scalars"] = np.array(scal, dtype=np.float64)
        out[tag + "_seed"] = np.int64(seed)

    record("testcull", UT.simple_cull, cases(120, 1, 30, False), 5, 32)                 # test copy, n <= MAX_FRONT 50
    record("testfinal", UT.simple_cull_final, cases(120, 1, 30, False), 5, 32)          # test copy, no truncation
    record("testfinal_long", UT.simple_cull_final, cases(12, 55, 64, True), 5, 64)      # ... with fronts beyond 50
    record("trainlong", U.simple_cull, cases(40, 24, 64, True), 3, 64)                   # train copy, fronts beyond MAX_FRONT 20
    record("testlong", UT.simple_cull, cases(20, 55, 64, True), 3, 64)                   # test copy, fronts beyond MAX_FRONT 50

    # pareto_state_data + padding
    MAXP = 20                                             # MAX_PARETO_SIZE, master…:475
    assert ENVM.MAX_FRONT == 20
    xs, As, meta_n = [], [], []
    for n in range(1, 21):
        a = np.sort(rng.random(n)) * 0.9 + 0.05
        pf = [[float(x), float((1 - np.sqrt(x)) ** 2), 0, 0] for x in a]
        for index in range(n):
            x_pf, A_pf = ENVM.pareto_state_data(pf, index=index)
            add = MAXP - x_pf.shape[0]
            if add > 0:                                   # master…:488-500 (np.block)
                x_pad = np.block([[x_pf], [np.zeros((add, 4))]])
                A_pad = np.block([[A_pf, np.zeros((x_pf.shape[0], add))], [np.zeros((add, x_pf.shape[0])), np.zeros((add, add))]])
            else:
                x_pad, A_pad = x_pf[:MAXP, :], A_pf[:MAXP, :MAXP]
            rec = np.full((MAXP, 2), np.nan); rec[:n] = np.array(pf)[:, :2]
            xs.append(x_pad); As.append(A_pad); meta_n.append(np.concatenate([[n, index], rec.ravel()]))
    out["pg_in"] = np.array(meta_n, dtype=np.float64)     # [n, index, obj pairs ...]
    out["pg_x"] = np.array(xs, dtype=np.float64)
    out["pg_A"] = np.array(As, dtype=np.float64)
    out["pg_x_dtype"] = str(ENVM.pareto_state_data([[0.5, 0.5, 0, 0]])[0].dtype)
    np.savez_compressed(os.path.join(outdir, "reward_ext.npz"), **out)
    print("wrote reward_ext", file=sys.stderr)


def main():
    outdir = HERE
    if len(sys.argv) >= 3 and sys.argv[1] == "--worker":
        worker(sys.argv[2], outdir)
        return
    for variant in VARIANTS:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", variant],
                              env=dict(os.environ, MPLBACKEND="Agg"))


if __name__ == "__main__":
    main()
