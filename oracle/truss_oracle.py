"""CPU ORACLE for the batched 2-D truss FEM environment step.  *** TEST INFRASTRUCTURE ONLY ***

This file is a plain-numpy restatement of the reference's algorithm for the hot path named in
BASELINE.json / SURVEY.md §8.  It exists to CHECK the HIP path; nothing under
`mop-truss-marl_amd/` may import it.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` use it.

Parity pin: every function below is checked against golden vectors produced by running the
reference itself in the build container (tests/golden/make_golden.py -> tests/golden/*.npz), see
tests/test_oracle_golden.py.  The fixtures embed numpy 2.2.6 scalar-promotion behaviour (the
reference pins numpy 1.23.5, SURVEY.md §7 "NumPy version drift").

Arithmetic contract restated from the reference (citations are relative to /root/reference/):
  * heights `y` are float32 values on a 0.01 grid; every operation of the action decode is a
    float32 operation (train/code/truss2D_ENV.py:358-490, heights arrive as np.float32 through
    `_set_model` :362 and stay float32 under NEP-50 promotion).  Python-typed constants that the
    reference assigns (d_min, y_max, y_max-d_min) are stored as their float32 roundings.
  * the FEM (train/code/FEM_2Dtruss.py:284-431) is float64.  Deliberate, documented deviation:
    the reference (under numpy 2.x) evaluates element length/cos/sin in float32 whenever an end
    height is np.float32; this oracle (and the HIP kernel) evaluate them in float64 from the same
    float32 heights.  The two differ by O(1e-7) relative, inside the 1e-5 parity budget.
  * `point`, observation tensors: float32 stores of float64 values exactly where the reference
    stores into float32 arrays (truss2D_ENV.py:40-193, 503-525).

All functions are batched over a leading env axis B and loop over nodes/elements only where the
reference's sequential semantics require it.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F64 = np.float64

# section catalogue, section_data/01_brace_rod2.csv (area [cm^2], inertia [cm^4]); truss2D_GEN.py:29-36
SECTIONS = np.array([[9.085, 59.5], [20.41, 300.0], [38.89, 830.0], [81.23, 4230.0], [164.6, 18700.0]],
                    dtype=F64)
E_MOD = 2 * 1e11                  # truss2D_GEN.py:59
YIELD_STRESS = 235 * 1e6          # FEM_2Dtruss.py:93
LONG_STRESS = YIELD_STRESS / 1.5  # FEM_2Dtruss.py:94
MOVE_FACTOR = F32(0.25)           # truss2D_ENV.py:395


def section_area(sec):
    """area [m^2] of catalogue entry `sec`: truss[sec][0]*1e-4 (truss2D_GEN.py:291, ENV:367)."""
    return SECTIONS[np.asarray(sec), 0] * 1e-4


def section_inertia(sec):
    """truss[sec][1]*1e-8 (truss2D_GEN.py:292)."""
    return SECTIONS[np.asarray(sec), 1] * 1e-8


# ----------------------------------------------------------------------------------------------
# topology (reset time, integers bit-exact)
# ----------------------------------------------------------------------------------------------
class Topology:
    """Static description of one truss topology shared by a batch of envs."""

    def __init__(self, conn, res, top, pair, sym_nodes=None, sym_elems=None):
        self.conn = np.asarray(conn, dtype=np.int32).reshape(-1, 2)
        self.res = np.asarray(res, dtype=np.int8).reshape(-1, 2)
        self.top = np.asarray(top, dtype=np.int8).reshape(-1)
        self.pair = np.asarray(pair, dtype=np.int32).reshape(-1)
        self.N = self.res.shape[0]
        self.E = self.conn.shape[0]
        # symmetry tables of the test variants (truss2D_ENV.py test copies :459-553 / :459-675):
        # rows (dst, src) applied when coin >= 0.5; when coin < 0.5 the roles are swapped.
        self.sym_nodes = np.zeros((0, 2), np.int32) if sym_nodes is None else np.asarray(sym_nodes, np.int32)
        self.sym_elems = np.zeros((0, 2), np.int32) if sym_elems is None else np.asarray(sym_elems, np.int32)
        self.nsc, self.tnsc, self.ndof = dof_numbering(self.res)
        self.ttnsc = element_dofs(self.conn, self.tnsc)


def dof_numbering(res):
    """Structure coordinate numbers.  FEM_2Dtruss.py:227-261 (gen_nsc, gen_tnsc, gen_ndof):
    scan nodes in order, x before y; free DOFs get 1..ndof in scan order, restrained DOFs get
    ndof+1..2N in scan order."""
    res = np.asarray(res).reshape(-1, 2)
    flat = res.reshape(-1)
    nsc = np.zeros(flat.shape[0], dtype=np.int32)
    c = 1
    for i in range(flat.shape[0]):
        if flat[i] == 0:
            nsc[i] = c
            c += 1
    for i in range(flat.shape[0]):
        if flat[i] == 1:
            nsc[i] = c
            c += 1
    ndof = int(flat.shape[0] - int((flat == 1).sum()))
    return nsc, nsc.reshape(-1, 2).copy(), ndof


def element_dofs(conn, tnsc):
    """ttnsc[e] = DOF ids of (n0x, n0y, n1x, n1y).  FEM_2Dtruss.py:310-317."""
    conn = np.asarray(conn)
    return np.concatenate([tnsc[conn[:, 0]], tnsc[conn[:, 1]]], axis=1).astype(np.int32)


def grid_topology(num_x, variant=None):
    """Two-row grid truss of truss2D_GEN.py:181-190, 241-434 (num_y == 2, support_case 1):
    nodes row-major (bottom row then top row); elements: beams row0, beams row1 (:280-295),
    columns (:299-321), braces top-left->bottom-right (:323-337), braces bottom-left->top-right
    (:339-353); pin supports at both bottom corners (:401-418); the upper node of every column is
    the `top_node` and columns define `vertical_pair` (:307-313).

    variant: None (train copy, no symmetry), "small" or "large" = the hard-coded mirror tables of the
    two test copies of truss2D_ENV.py (:459-553 / :459-675), generated here from the mirror rule and
    checked against the reference's behaviour by the golden transitions."""
    nx = int(num_x)
    N = 2 * nx
    conn = []
    for row in range(2):
        for i in range(nx - 1):
            conn.append((row * nx + i, row * nx + i + 1))
    for i in range(nx):
        conn.append((i, nx + i))
    for i in range(nx - 1):
        conn.append((nx + i, i + 1))
    for i in range(nx - 1):
        conn.append((i, nx + i + 1))
    res = np.zeros((N, 2), np.int8)
    res[0] = 1
    res[nx - 1] = 1
    top = np.zeros(N, np.int8)
    top[nx:] = 1
    pair = np.concatenate([np.arange(nx, 2 * nx), np.arange(0, nx)]).astype(np.int32)
    sym_nodes = sym_elems = None
    if variant is not None:
        half = nx // 2
        if variant == "small":
            # coin>=0.5: left copies from right; supports (0, nx-1) are not mirrored (test small :459-502)
            bot = [(i, nx - 1 - i) for i in range(1, half)]
            tp = [(nx + i, 2 * nx - 1 - i) for i in range(half)]
            sym_nodes = np.array(bot + tp, np.int32)            # (dst, src) when coin >= 0.5
        elif variant == "large":
            # coin>=0.5: right copies from left, supports included (test large :459-560)
            bot = [(nx - 1 - i, i) for i in range(half)]
            tp = [(2 * nx - 1 - i, nx + i) for i in range(half)]
            sym_nodes = np.array(bot + tp, np.int32)
        else:
            raise ValueError(variant)
        nb = nx - 1
        pairs = []
        for row in range(2):                                     # beams mirror inside their row
            for i in range(nb // 2):
                pairs.append((row * nb + i, row * nb + nb - 1 - i))
        c0 = 2 * nb
        for i in range(nx // 2):                                 # columns
            pairs.append((c0 + i, c0 + nx - 1 - i))
        b0 = c0 + nx
        for i in range(nb):                                      # "\" brace i  <->  "/" brace nb-1-i
            pairs.append((b0 + i, b0 + nb + nb - 1 - i))
        sym_elems = np.array(pairs, np.int32)
    return Topology(np.array(conn, np.int32), res, top, pair, sym_nodes, sym_elems)


def grid_coordinates(span_x, span_y):
    """x of every node and the initial heights (truss2D_GEN.py:181-190): rows at y=0 and y=span_y[0]."""
    xs = [sum(span_x[:i]) for i in range(len(span_x) + 1)]
    x = np.array(xs + xs, dtype=F64)
    y0 = np.array([0.0] * len(xs) + [float(span_y[0])] * len(xs), dtype=F64)
    return x, y0


def grid_targets(topo, tar_y):
    """targets go to top nodes in node order (truss2D_GEN.py:369-374); others 0."""
    t = np.zeros(topo.N, F64)
    t[topo.top == 1] = np.asarray(tar_y, F64)
    return t


def load_mask(topo, is_roof):
    """bridge: unsupported bottom nodes; roof: top nodes (truss2D_GEN.py:421-430).  'bottom' is
    `coord[1] == min(y)` at generation time == the non-top row."""
    if is_roof:
        return (topo.top == 1)
    return (topo.top == 0) & (topo.res[:, 1] == 0)


# ----------------------------------------------------------------------------------------------
# move ranges
# ----------------------------------------------------------------------------------------------
def move_range(topo, y, y_max, d_min, is_roof):
    """gen_model.set_moveRange, truss2D_GEN.py:118-133, in float32 (heights are np.float32)."""
    y = np.asarray(y, F32)
    B = y.shape[0]
    ymax32 = np.asarray(y_max, F64).astype(F32).reshape(B, 1)
    dmin32 = np.asarray(d_min, F64).astype(F32).reshape(B, 1)
    roof = np.asarray(is_roof).astype(bool).reshape(B, 1)
    yp = y[:, topo.pair]
    topm = (topo.top == 1)[None, :]
    up_top = np.abs(ymax32 - y)
    dn_top = np.abs((y - yp) - dmin32)
    up_bot = np.where(roof, np.abs((yp - y) - dmin32), F32(0))
    dn_bot = np.where(roof, np.abs(y), F32(0))
    max_up = np.where(topm, up_top, up_bot).astype(F32)
    max_down = np.where(topm, dn_top, dn_bot).astype(F32)
    return max_up, max_down


# ----------------------------------------------------------------------------------------------
# action decode  (Game_research04._game_modify prologue)
# ----------------------------------------------------------------------------------------------
def round2_f32(v):
    """Python round(np.float32, 2) == numpy's rint(v*100)/100 evaluated in float32
    (truss2D_ENV.py:413)."""
    v = np.asarray(v, F32)
    return (np.rint(v * F32(100.0)) / F32(100.0)).astype(F32)


def decode_actions(topo, y, sec, max_up, max_down, geo, topo_act, coin, y_max, d_min):
    """truss2D_ENV.py:370-490 (+ symmetry blocks of the test copies).  Returns
    (y', sec', clamped_geo, clamped_topo).  `max_up/max_down` are the move ranges the reference has
    on its model object when the call starts (stale from the previous analysis, :402/:407)."""
    y = np.array(y, dtype=F32, copy=True)
    sec = np.array(sec, dtype=np.int32, copy=True)
    B, N = y.shape
    geo = np.array(geo, dtype=F32, copy=True)
    tac = np.array(topo_act, dtype=F32, copy=True)
    # in-place clamp to [0,1]  (:376-388); NaN passes through like in the reference
    geo = np.where(geo > 1, F32(1), np.where(geo < 0, F32(0), geo)).astype(F32)
    tac = np.where(tac > 1, F32(1), np.where(tac < 0, F32(0), tac)).astype(F32)
    mu = np.asarray(max_up, F32)
    md = np.asarray(max_down, F32)
    ymax32 = np.asarray(y_max, F64).astype(F32).reshape(B)
    dmin32 = np.asarray(d_min, F64).astype(F32).reshape(B)
    ymax_minus_dmin32 = (np.asarray(y_max, F64).reshape(B) - np.asarray(d_min, F64).reshape(B)).astype(F32)
    # geometry move (:398-408): first-max argmax over the two channels
    down = geo[:, :, 1] > geo[:, :, 0]
    a = np.where(down, geo[:, :, 1], geo[:, :, 0])
    rng_ = np.where(down, md, mu)
    step = ((a * rng_).astype(F32) * MOVE_FACTOR).astype(F32)
    y = np.where(down, (y - step).astype(F32), (y + step).astype(F32)).astype(F32)
    # supports -> 0, everything rounded to 2 decimals (:410-413)
    y = round2_f32(y)
    y[:, topo.res[:, 1] == 1] = F32(0)
    # sizing (:421-466): (nC_e @ topo)[e] = topo[n0] + topo[n1] in float32; first-max argmax of 3
    pv = (tac[:, topo.conn[:, 0], :] + tac[:, topo.conn[:, 1], :]).astype(F32)
    am = np.zeros(pv.shape[:2], np.int32)
    best = pv[:, :, 0].copy()
    m1 = pv[:, :, 1] > best
    am[m1] = 1
    best = np.where(m1, pv[:, :, 1], best)
    m2 = pv[:, :, 2] > best
    am[m2] = 2
    sec = np.where(am == 0, np.maximum(0, sec - 1), np.where(am == 1, np.minimum(len(SECTIONS) - 1, sec + 1), sec))
    sec = sec.astype(np.int32)
    # sequential height repairs (:469-490); node i may write its vertical partner
    for i in range(N):
        c = y[:, i] < F32(0)
        if topo.top[i] == 1:
            y[:, i] = np.where(c, dmin32, y[:, i])
            y[:, topo.pair[i]] = np.where(c, F32(0), y[:, topo.pair[i]])
        else:
            y[:, i] = np.where(c, F32(0), y[:, i])
    for i in range(N):
        c = y[:, i] > ymax32
        if topo.top[i] == 1:
            y[:, i] = np.where(c, ymax32, y[:, i])
        else:
            y[:, i] = np.where(c, ymax_minus_dmin32, y[:, i])
            y[:, topo.pair[i]] = np.where(c, ymax32, y[:, topo.pair[i]])
    for i in range(N):
        if topo.top[i] == 1:
            p = topo.pair[i]
            c = np.abs((y[:, i] - y[:, p]).astype(F32)) < dmin32
            y[:, i] = np.where(c, (y[:, p] + dmin32).astype(F32), y[:, i])
    # symmetry of the test variants
    if topo.sym_nodes.shape[0] or topo.sym_elems.shape[0]:
        heads = np.asarray(coin, F64).reshape(B) >= 0.5
        for dst, src in topo.sym_nodes:
            ydst = y[:, dst].copy()
            ysrc = y[:, src].copy()
            y[:, dst] = np.where(heads, ysrc, ydst)
            y[:, src] = np.where(heads, ysrc, ydst)
        for a_, b_ in topo.sym_elems:
            mn = np.minimum(sec[:, a_], sec[:, b_])
            sec[:, a_] = mn
            sec[:, b_] = mn
    return y.astype(F32), sec, geo, tac


# ----------------------------------------------------------------------------------------------
# FEM  (Model.gen_all, FEM_2Dtruss.py:434-459)
# ----------------------------------------------------------------------------------------------
def element_geometry(topo, x, y):
    """length, cos, sin per element in float64 from float32-valued heights
    (FEM_2Dtruss.py:99-105, 290-298)."""
    x = np.asarray(x, F64)
    y = np.asarray(y, F32).astype(F64)
    dx = x[:, topo.conn[:, 1]] - x[:, topo.conn[:, 0]]
    dy = y[:, topo.conn[:, 1]] - y[:, topo.conn[:, 0]]
    L = np.sqrt(dx * dx + dy * dy)
    return L, dx / L, dy / L


def load_vector(topo, load):
    """P over free DOFs in DOF order (FEM_2Dtruss.py:207-223, 264-280). load: [B,N,2]."""
    load = np.asarray(load, F64)
    B = load.shape[0]
    P = np.zeros((B, topo.ndof), F64)
    flat = load.reshape(B, -1)
    free = topo.nsc <= topo.ndof
    P[:, topo.nsc[free] - 1] = flat[:, free]
    return P


def fem_solve(topo, x, y, sec, load, e_mod=E_MOD, area=None):
    """One direct-stiffness analysis per env.  Returns a dict with K, P, d (reference DOF order),
    dnode [B,N,2], q0, sr, comp, L, U, r.  FEM_2Dtruss.py:284-431."""
    L, c, s = element_geometry(topo, x, y)
    B = L.shape[0]
    A = section_area(sec) if area is None else np.asarray(area, F64)
    k = e_mod * A / L
    # global element matrix = k * [[cc,cs,-cc,-cs],[cs,ss,-cs,-ss],[-cc,-cs,cc,cs],[-cs,-ss,cs,ss]]
    v = np.stack([c, s, -c, -s], axis=-1)                       # [B,E,4]
    kg = k[:, :, None, None] * v[:, :, :, None] * v[:, :, None, :]
    nd = topo.ndof
    K = np.zeros((B, nd, nd), F64)
    tt = topo.ttnsc
    for e in range(topo.E):                                      # gen_ssm order (FEM:320-324)
        for j in range(4):
            if tt[e, j] > nd:
                continue
            for kk in range(4):
                if tt[e, kk] > nd:
                    continue
                K[:, tt[e, j] - 1, tt[e, kk] - 1] += kg[:, e, j, kk]
    P = load_vector(topo, load)
    d = np.linalg.solve(K, P[:, :, None])[:, :, 0]               # gen_d (FEM:337)
    dfull = np.concatenate([d, np.zeros((B, 2 * topo.N - nd), F64)], axis=1)
    dnode = dfull[:, topo.nsc - 1].reshape(B, topo.N, 2)          # gen_v (FEM:341-352)
    v0 = dnode[:, topo.conn[:, 0], :]
    v1 = dnode[:, topo.conn[:, 1], :]
    u0 = c * v0[:, :, 0] + s * v0[:, :, 1]                        # gen_u (FEM:356-357)
    u2 = c * v1[:, :, 0] + s * v1[:, :, 1]
    q0 = k * u0 + (-k) * u2                                       # gen_q (FEM:383-386)
    sr = np.abs(q0 / A) / LONG_STRESS                             # gen_yield (FEM:414-431)
    comp = (q0 > 0).astype(np.int8)
    U = 0.5 * np.einsum("bi,bij,bj->b", d, K, d)                  # gen_U_full (FEM:374-379)
    # reactions (FEM:393-411): f = T^T q = q0*[c, s, -c, -s]
    f = q0[:, :, None] * np.stack([c, s, -c, -s], axis=-1)
    r = np.zeros((B, 2 * topo.N), F64)
    for e in range(topo.E):
        for j in range(4):
            if tt[e, j] > nd:
                r[:, tt[e, j] - 1] += f[:, e, j]
    r[:, :nd] = np.nan
    return dict(K=K, P=P, d=d, dnode=dnode, q0=q0, sr=sr, comp=comp, L=L, U=U, r=r, area=A, c=c, s=s, k=k)


# ----------------------------------------------------------------------------------------------
# objectives / constraints  (truss2D_ENV.py:503-525, 264-274)
# ----------------------------------------------------------------------------------------------
def _sum_f32(a):
    """sum of float32 terms: accumulated in float64 then rounded once (the reference's np.sum on a
    float32 array is a float32 pairwise sum; the two agree to ~1e-7 relative)."""
    return np.asarray(a, F32).astype(F64).sum(axis=-1).astype(F32)


def objectives(topo, y, target, fem, max_def):
    y = np.asarray(y, F32)
    B = y.shape[0]
    all_v = (fem["area"] * fem["L"]).astype(F32)
    all_s = fem["sr"].astype(F32)
    topm = topo.top == 1
    all_dt = np.where(topm[None, :], np.abs(np.asarray(target, F64).astype(F32) - y), F32(0)).astype(F32)
    all_d = np.where(topm[None, :], F32(0),
                     (fem["dnode"][:, :, 1] / np.asarray(max_def, F64).reshape(B, 1)).astype(F32)).astype(F32)
    obj1 = _sum_f32(all_v)
    obj2 = _sum_f32(all_dt)
    con1 = np.abs(all_s).max(axis=1)
    con2 = np.abs(all_d).max(axis=1)
    return obj1, obj2, con1.astype(F32), con2.astype(F32)


def make_point(obj1, obj2, con1, con2, int_obj):
    int_obj = np.asarray(int_obj, F32)
    return np.stack([(obj1 / int_obj[:, 0]).astype(F32), (obj2 / int_obj[:, 1]).astype(F32), con1, con2],
                    axis=1).astype(F32)


# ----------------------------------------------------------------------------------------------
# observation tensors  (truss2D_ENV.py:40-193)
# ----------------------------------------------------------------------------------------------
def normalized_adjacency(topo):
    """A_n = D^-1/2 (A+I) D^-1/2 in float32 with spektral.utils.degree_power semantics
    (truss2D_ENV.py:83-84, 104-107).  Topology-static."""
    N = topo.N
    A = np.zeros((N, N), F32)
    A[topo.conn[:, 0], topo.conn[:, 1]] = 1
    A[topo.conn[:, 1], topo.conn[:, 0]] = 1
    mask = A.copy()
    A = A + np.eye(N, dtype=F32)
    with np.errstate(divide="ignore"):
        dg = np.power(np.array(A.sum(1)), -1 / 2).ravel()
    dg[np.isinf(dg)] = 0.0
    D = np.diag(dg)
    A_n = np.matmul(D, np.matmul(A, D))
    return A_n.astype(F32), mask


def incidence(topo):
    c_e = np.zeros((topo.E, topo.N), F32)
    c_e[np.arange(topo.E), topo.conn[:, 0]] = 1
    c_e[np.arange(topo.E), topo.conn[:, 1]] = 1
    return c_e


def _node_features(topo, x, y, has_load, max_up, max_down, target, fem, max_def, ncol):
    B, N = np.asarray(y).shape
    xn = np.zeros((B, N, ncol), F32)
    y32 = np.asarray(y, F32)
    md32 = np.asarray(max_def, F64).astype(F32).reshape(B, 1)
    xn[:, :, 0] = np.asarray(x, F64).astype(F32)
    xn[:, :, 1] = y32
    xn[:, :, 2] = topo.res[:, 0][None, :]
    xn[:, :, 3] = topo.res[:, 1][None, :]
    xn[:, :, 4] = np.abs(np.asarray(has_load, F64)).astype(F32)
    xn[:, :, 5] = topo.top[None, :]
    xn[:, :, 6] = np.abs(topo.top.astype(np.int32) - 1)[None, :]
    xn[:, :, 7] = np.asarray(max_up, F32)
    xn[:, :, 8] = np.asarray(max_down, F32)
    t32 = np.asarray(target, F64).astype(F32)
    xn[:, :, 9] = np.where((topo.top == 1)[None, :], (t32 / (y32 + F32(1e-6)).astype(F32)).astype(F32), F32(0))
    xn[:, :, 10] = np.abs(fem["dnode"][:, :, 1]).astype(F32)
    ratio = (xn[:, :, 10] / md32).astype(F32)
    return xn, ratio


def state_data(topo, x, y, sec, has_load, max_up, max_down, target, fem, max_def):
    """truss2D_ENV.py:40-109 -> x_n[B,N,13], A_s, A_n_ts, A_n_cs [B,N,N] (A_n and mask are
    topology-static: `normalized_adjacency`)."""
    xn, ratio = _node_features(topo, x, y, has_load, max_up, max_down, target, fem, max_def, 13)
    B, N = ratio.shape
    viol = ratio > 1
    xn[:, :, 11] = (np.minimum(ratio, F32(1)) * np.where(viol, F32(1), F32(0.5))).astype(F32)
    xn[:, :, 12] = viol
    mn = xn.min(axis=1, keepdims=True)
    mx = xn.max(axis=1, keepdims=True)
    xn = ((xn - mn) / (mx - mn + F32(1e-6))).astype(F32)
    A_s = np.zeros((B, N, N), F32)
    A_ts = np.zeros((B, N, N), F32)
    A_cs = np.zeros((B, N, N), F32)
    a, b = topo.conn[:, 0], topo.conn[:, 1]
    vs = (fem["area"] / (SECTIONS[-1, 0] * 1e-4)).astype(F32)
    A_s[:, a, b] = vs
    A_s[:, b, a] = vs
    sr = fem["sr"]
    val = (np.minimum(sr, 1.0) * np.where(sr > 1, 1.0, 0.5)).astype(F32)
    tens = fem["comp"] == 0
    A_ts[:, a, b] = np.where(tens, val, F32(0))
    A_ts[:, b, a] = np.where(tens, val, F32(0))
    A_cs[:, a, b] = np.where(tens, F32(0), val)
    A_cs[:, b, a] = np.where(tens, F32(0), val)
    return xn, A_s, A_ts, A_cs


def state_data_not_norm(topo, x, y, sec, has_load, max_up, max_down, target, fem, max_def):
    """truss2D_ENV.py:112-193 -> nN_x_n[B,N,12], nN_x_e[B,E,21]."""
    xn, ratio = _node_features(topo, x, y, has_load, max_up, max_down, target, fem, max_def, 12)
    xn[:, :, 11] = ratio >= 1
    B = xn.shape[0]
    xe = np.zeros((B, topo.E, 21), F32)
    xe[:, :, 0] = sec
    xe[:, :, 1] = fem["area"].astype(F32)
    xe[:, :, 2] = fem["L"].astype(F32)
    xe[:, :, 3] = np.abs(fem["comp"].astype(np.int32) - 1)
    xe[:, :, 4] = fem["comp"]
    xe[:, :, 5] = fem["q0"].astype(F32)
    xe[:, :, 6] = fem["sr"] > 1
    for off, col in ((7, 0), (14, 1)):
        n = topo.conn[:, col]
        xe[:, :, off + 0] = xn[:, n, 0]
        xe[:, :, off + 1] = xn[:, n, 1]
        xe[:, :, off + 2] = xn[:, n, 2]
        xe[:, :, off + 3] = xn[:, n, 3]
        xe[:, :, off + 4] = xn[:, n, 4]
        xe[:, :, off + 5] = xn[:, n, 10]
        xe[:, :, off + 6] = xn[:, n, 11]
    return xn, xe


# ----------------------------------------------------------------------------------------------
# the whole transition (== Game_research04._game_modify, truss2D_ENV.py:370-525)
# ----------------------------------------------------------------------------------------------
def env_step(topo, x, y, sec, max_up, max_down, geo, topo_act, coin, target, load, y_max, d_min,
             max_def, is_roof, int_obj, with_obs=False):
    """One `_game_modify`-equivalent per env.  `max_up/max_down` may be None = recompute from the
    parent heights (what the reference would hold had it last analysed that parent)."""
    if max_up is None:
        max_up, max_down = move_range(topo, y, y_max, d_min, is_roof)
    y2, sec2, geo_c, topo_c = decode_actions(topo, y, sec, max_up, max_down, geo, topo_act, coin, y_max, d_min)
    mu2, md2 = move_range(topo, y2, y_max, d_min, is_roof)
    fem = fem_solve(topo, x, y2, sec2, load)
    obj1, obj2, con1, con2 = objectives(topo, y2, target, fem, max_def)
    out = dict(y=y2, sec=sec2, geo=geo_c, topo=topo_c, max_up=mu2, max_down=md2, fem=fem,
               point=make_point(obj1, obj2, con1, con2, int_obj), obj=np.stack([obj1, obj2], 1))
    if with_obs:
        has_load = (np.asarray(load)[:, :, 1] != 0) | (np.asarray(load)[:, :, 0] != 0)
        xn, A_s, A_ts, A_cs = state_data(topo, x, y2, sec2, has_load, mu2, md2, target, fem, max_def)
        nxn, nxe = state_data_not_norm(topo, x, y2, sec2, has_load, mu2, md2, target, fem, max_def)
        out.update(x_n=xn, A_s=A_s, A_n_ts=A_ts, A_n_cs=A_cs, nN_x_n=nxn, nN_x_e=nxe)
    return out


def initial_objectives(topo, x, y, sec, target):
    """int_obj1/int_obj2 of Game_research04.__init__ (truss2D_ENV.py:264-274)."""
    L, _, _ = element_geometry(topo, x, y)
    all_v = (section_area(sec) * L).astype(F32)
    y32 = np.asarray(y, F32)
    all_dt = np.where((topo.top == 1)[None, :], np.abs(np.asarray(target, F64).astype(F32) - y32), F32(0))
    return np.stack([_sum_f32(all_v), _sum_f32(all_dt.astype(F32))], axis=1).astype(F32)


# ----------------------------------------------------------------------------------------------
# reward half: Pareto cull + 2-D hypervolume (utils.py)
# ----------------------------------------------------------------------------------------------
def dominates(a, b):
    """utils.py:8-9: strict dominance on the two objectives."""
    return a[0] < b[0] and a[1] < b[1]


def pareto_front(points):
    """Non-dominated feasible subset, sorted by obj1 (utils.py:11-54).  `points` rows are
    [obj1, obj2, con1, con2, ...]; rows with con1>1 or con2>1 are dropped (:16-23); duplicates
    collapse (the reference collects tuples in a set, :25,:48)."""
    feas = [tuple(p) for p in points if not (p[2] > 1 or p[3] > 1)]
    front = set()
    for p in feas:
        if not any(dominates(q, p) for q in feas):
            front.add(p)
    return sorted([list(p) for p in front], key=lambda r: r[0])


def front_metrics(front):
    """max/std of neighbour distances, sum, crowding-distance std and p-norm (utils.py:151-199)
    for a front that is not longer than MAX_FRONT."""
    n = len(front)
    dist = [((front[i][0] - front[i + 1][0]) ** 2 + (front[i][1] - front[i + 1][1]) ** 2) ** 0.5
            for i in range(n - 1)]
    if n >= 2:
        max_d = max(dist)
        dis_d = (sum((v - max_d / len(dist)) ** 2 for v in dist) / len(dist)) ** 0.5
        sum_d = sum(dist)
    else:
        max_d, dis_d, sum_d = 0, 1, 0
    p = 10
    if n > 3:
        cd = [abs(front[i - 1][0] - front[i + 1][0]) + abs(front[i - 1][1] - front[i + 1][1])
              for i in range(1, n - 1)]
        if np.sum(np.array(cd)) == 0:
            std_cd, p_cd = 1, 0
        else:
            cdn = np.array(cd) / np.max(np.array(cd))
            std_cd = np.std(cdn)
            p_cd = sum(abs(v) ** p for v in cdn) ** (1 / p)
    else:
        std_cd, p_cd = 1, 0
    return max_d, dis_d, p_cd, sum_d, std_cd


def hypervolume_2d(points, ref_point=(1, 1)):
    """utils.union_rectangles_fastest (utils.py:275-342): area of the union of the rectangles
    [min(x,1), 1] x [0, 1-min(y,1)] minus the reference-point correction term (:340).  Restated as a
    sort-and-sweep staircase instead of the segment tree; same value."""
    pts = [(p[0], p[1]) for p in points]
    if len(pts) == 0:
        return 0
    if len(pts) == 1 and pts[0][0] == 1 and pts[0][1] == 1:
        return 0
    cl = sorted((min(px, 1), min(py, 1)) for px, py in pts)
    area = 0.0
    best_h = 0.0          # tallest rectangle seen so far among points with smaller x
    xs = [c[0] for c in cl] + [1.0]
    for i, (cx, cy) in enumerate(cl):
        best_h = max(best_h, 1 - cy)
        area += (xs[i + 1] - cx) * best_h
    min_x = min(p[0] for p in pts)
    min_y = min(p[1] for p in pts)
    rx, ry = ref_point
    remove = (1 - rx) * (1 - min_x) + (1 - ry) * (1 - min_y) - (1 - rx) * (1 - ry)
    return area - remove
