"""Drop-in replacement for the reference's truss2D_ENV.py: ENV, Game_research04 and the observation
builders, on top of the MI355X batched step (truss_mi355.BatchedTruss, B = 1 here; the same kernels
step 4096 envs at once in the batched rollouts).

What the master script touches and where it lands:
    ENV(game) / .reset() / .check_over() / .over                truss2D_ENV.py:204-224
    Game_research04(end_step, model, num_agents) / .re_game     :229-334  (int_obj1/int_obj2)
    ._game_get_1_state()  -> 11-tuple                           :336-351  one analysis launch + obs launch
    ._set_model(set_node, set_element)                          :358-368
    ._game_modify(set_node, set_element, nC_e, actions)         :370-525  ONE step launch + obs launch
    .reset() / .step()                                          :527-540

Reference behaviours kept on purpose:
  * `_game_modify` clamps the caller's action arrays in place (:376-388);
  * the geometry move uses the move ranges left on the model by the PREVIOUS analysis (:402,:407),
    not those of the design being restored -- they are kept as hidden state exactly like the
    reference's Node.max_up / max_down;
  * the test copies' mirror symmetry with its `random.random() >= 0.5` coin (:459 of the test copies):
    choose it with configure("small") / configure("large"); the coin is drawn from this module's
    `random` exactly once per call, as in the reference.
"""
import random

import numpy as np
import torch

import truss_mi355 as _tm
import FEM_2Dtruss as _fem
from set_seed_global import seedThis
import utils as _utils
from utils import dominates, simple_cull, simple_cull_final, CoverQuery, union_rectangles_fastest  # noqa: F401

np.random.seed(seedThis)
random.seed(seedThis)

MAX_MEM_NO = 5
MAX_FRONT = 20
SYMMETRY = None      # None (train copy) | "small" (test/00,01) | "large" (test/02,03)


def configure(variant=None):
    """Select which copy of the reference's truss2D_ENV.py / utils.py this module stands in for."""
    global MAX_FRONT, SYMMETRY
    if variant not in (None, "train", "small", "large"):
        raise ValueError(variant)
    SYMMETRY = None if variant in (None, "train") else variant
    MAX_FRONT = 20 if SYMMETRY is None else 50
    _utils.MAX_FRONT = MAX_FRONT
    _utils.EDGE_KEEP = 3 if SYMMETRY is None else 10
    _utils.EDGE_ADD = 1 if SYMMETRY is None else 3


def _degree_power(A, k):
    """spektral.utils.degree_power (spektral 1.2.0): diag(rowsum(A)**k) with inf -> 0."""
    with np.errstate(divide="ignore"):
        d = np.power(np.array(A.sum(1)), k).ravel()
    d[np.isinf(d)] = 0.0
    return np.diag(d)


def pareto_state_data(pf, index=0):
    """truss2D_ENV.py:19-38."""
    x_pf = np.zeros((len(pf), 4), dtype=np.float32)
    for i in range(len(pf)):
        x_pf[i][0] = pf[i][0]
        x_pf[i][1] = pf[i][1]
        if i == index:
            x_pf[i][2] = 1
        x_pf[i][3] = len(pf) / MAX_FRONT
    A_pf = np.eye(len(pf), dtype=np.float32)
    for i in range(len(pf) - 1):
        A_pf[i][i + 1] = 1
        A_pf[i + 1][i] = 1
    D_pf = _degree_power(A_pf, -1 / 2)
    A_pf = np.matmul(D_pf, np.matmul(A_pf, D_pf))
    return x_pf, A_pf


class ENV:
    def __init__(self, game):
        self.name = 'FRAME_ENV'
        self.game = game
        self.num_agents = game.num_agents
        self.over = 0
        self.output = []

    def check_over(self):
        if self.game.done_counter == 1:
            self.over = 1

    def reset(self):
        self.over = 0
        self.game.reset()
        self.output = []


class Game_research04:
    def __init__(self, end_step, model, num_agents=2):
        self._bt = None
        self.re_game(end_step, model, num_agents)

    # ------------------------------------------------------------------ native backend (B = 1)
    def _backend(self):
        gm = self.gen_model
        key = (gm.num_x, SYMMETRY, _fem._library().path)
        if self._bt is None or self._bt_key != key:
            topo = _tm.TrussTopology.grid(gm.num_x, SYMMETRY)
            self._bt = _tm.BatchedTruss(topo, 1, lib=_fem._library())
            self._bt_key = key
            self._A_n, self._mask = topo.normalized_adjacency()
            self._nC_e = topo.incidence()
        return self._bt

    def _push_constants(self):
        gm, bt = self.gen_model, self._backend()
        nodes = gm.model.nodes
        x = np.array([float(n.coord[0]) for n in nodes])
        target = np.array([float(n.target) if n.top_node == 1 else 0.0 for n in nodes])
        bt.set_constants(x, target, float(gm.y_max), float(gm.d_min), float(gm.max_deformation), float(gm.loadx),
                         float(gm.loady), 1.0 if gm.truss_type == 'roof' else 0.0)
        bt.env_params[:, _tm._lib.P_INTOBJ1] = float(self.int_obj1)
        bt.env_params[:, _tm._lib.P_INTOBJ2] = float(self.int_obj2)

    def _pull(self, with_design=True):
        """mirror the device results onto the reference-shaped object graph (host lists, B = 1)."""
        gm, bt = self.gen_model, self._bt
        r = bt.results()
        m = gm.model
        y, sec = r["y"][0], r["sec"][0]
        for i, n in enumerate(m.nodes):
            if with_design:
                n.coord[1] = np.float32(y[i])
            n.max_up = np.float32(r["max_up"][0, i])
            n.max_down = np.float32(r["max_down"][0, i])
            n.global_d = [[float(r["disp"][0, i, 0])], [float(r["disp"][0, i, 1])]]
            n.set_target()
        for k, e in enumerate(m.elements):
            if with_design:
                e.section_no = int(sec[k])
                e.area = gm.truss[e.section_no][0] * 1e-4
                e.set_i(gm.truss[e.section_no][1] * 1e-8)
            e.gen_length()
            q0 = float(r["q0"][0, k])
            e.e_q = np.array([[q0], [0.0], [-q0], [0.0]])
            e.prop_yeield = float(r["sr"][0, k])
            e.iscompress = int(r["comp"][0, k])
        m.U_full = float(r["energy"][0])
        if int(r["status"][0]) != 0:
            raise np.linalg.LinAlgError("Singular matrix")
        return r

    def _state(self, obs=None):
        """observation arrays of env 0: `obs` = what the analysis / step call has just written (TRUSS_F_EMIT_OBS: the
        reference builds them inside the same _game_modify call, :497-500); None = run the observation kernel"""
        o = {k: v.cpu().numpy()[0] for k, v in (self._bt.observe() if obs is None else obs).items()}
        return o

    # ------------------------------------------------------------------ reference surface
    def re_game(self, end_step, model, num_agents=2):
        self.name = 'Game_research04'
        self.description = 'There are 2 type of agent \n Agent_s adjust node up and down\n Agent_t adjust element section'
        self.objective = 'min(Weigth),min(Diff_btw_targetShape_and_currentShape)'
        self.num_agents = num_agents
        self.gen_model = model
        self.num_x, self.num_y = model.num_x, model.num_y
        self.game_step = 1
        self.end_step = end_step
        self.height_change, self.topology_change = [], []
        self.max_y_val, self.min_y_val = model.y_max, model.y_min
        self.reward_counter = [0, 0]
        self.done_counter = 0
        self.current_hv = 0
        self.ref_point = [1, 1]
        self.front_max_distance = 0
        self.front_dis_distance = 0
        # int_obj1 / int_obj2 (truss2D_ENV.py:264-274): float32 sums of float32 terms
        els, nodes = model.model.elements, model.model.nodes
        all_v = np.array([e.area * e.length for e in els], dtype=np.float32)
        all_dt = np.array([abs(n.target - n.coord[1]) if n.top_node == 1 else 0.0 for n in nodes], dtype=np.float32)
        self.int_obj1 = np.float32(all_v.astype(np.float64).sum())
        self.int_obj2 = np.float32(all_dt.astype(np.float64).sum())
        print('-------------------------------------------------------')
        print(self.description)
        print(self.objective)
        print('GAME WILL BE ENDED AFTER {} STEP'.format(self.end_step))
        print('-------------------------------------------------------')

    def _design_arrays(self):
        m = self.gen_model.model
        y = np.array([float(n.coord[1]) for n in m.nodes], np.float32)
        sec = np.array([int(e.section_no) for e in m.elements], np.int32)
        return y, sec

    def _game_get_1_state(self, index=0):
        bt = self._backend()
        self._push_constants()
        y, sec = self._design_arrays()
        bt.set_design(y, sec)
        ob = bt.analyze(obs=True)
        self._pull(with_design=False)
        o = self._state(ob)
        x_pf = np.zeros((1, 4), dtype=np.float32)
        x_pf[0][0] = 1
        x_pf[0][1] = 1
        x_pf[0][2] = 1
        x_pf[0][3] = 1 / MAX_FRONT
        A_pf = np.eye(1, dtype=np.float32)
        return (o["x_n"], self._A_n.copy(), o["A_s"], o["A_n_ts"], o["A_n_cs"], self._mask.copy(), x_pf, A_pf,
                o["nN_x_n"], o["nN_x_e"], self._nC_e.copy())

    def _set_model(self, set_node, set_element):
        gm = self.gen_model
        for i, n in enumerate(gm.model.nodes):
            n.coord[1] = set_node[i][1]
        for i, e in enumerate(gm.model.elements):
            e.section_no = int(set_element[i][0])
            e.area = gm.truss[e.section_no][0] * 1e-4
            e.set_i(gm.truss[e.section_no][1] * 1e-8)

    def _game_modify(self, set_node, set_element, nC_e, actions):
        bt = self._backend()
        self._push_constants()
        dev = bt.device
        N = len(self.gen_model.model.nodes)
        a_geo = np.ascontiguousarray(actions[0], dtype=np.float32).reshape(1, N, 2)
        a_topo = np.ascontiguousarray(actions[1], dtype=np.float32).reshape(1, N, 3)
        g_t, t_t = torch.from_numpy(a_geo.copy()).to(dev), torch.from_numpy(a_topo.copy()).to(dev)
        # stale move ranges of whatever was analysed last (hidden state, see module docstring)
        nodes = self.gen_model.model.nodes
        mu = torch.tensor([[float(n.max_up) for n in nodes]], dtype=torch.float32, device=dev)
        md = torch.tensor([[float(n.max_down) for n in nodes]], dtype=torch.float32, device=dev)
        y = np.asarray(set_node, np.float32)[:, 1]
        sec = np.asarray(set_element)[:, 0].astype(np.int32)
        bt.set_design(y, sec)
        coin = None
        if SYMMETRY is not None:
            coin = torch.tensor([1 if random.random() >= 0.5 else 0], dtype=torch.uint8, device=dev)
        ob = bt.step(g_t, t_t, coin, mu, md, clamp_inplace=True, obs=True)   # step + observation tensors: one native call
        # the reference mutates the caller's arrays (truss2D_ENV.py:376-388)
        np.copyto(np.asarray(actions[0]), g_t.cpu().numpy()[0].reshape(np.asarray(actions[0]).shape))
        np.copyto(np.asarray(actions[1]), t_t.cpu().numpy()[0].reshape(np.asarray(actions[1]).shape))
        r = self._pull(with_design=True)
        o = self._state(ob)
        St_S = [o["x_n"], self._A_n.copy(), o["A_s"], o["A_n_ts"], o["A_n_cs"], self._mask.copy(), None, None,
                o["nN_x_n"], o["nN_x_e"], self._nC_e.copy()]
        p = r["point"][0]
        point = [np.float32(p[0]), np.float32(p[1]), np.float32(p[2]), np.float32(p[3])]
        return point, St_S

    def reset(self):
        self.height_change, self.topology_change = [], []
        self.reward_counter = [0, 0]
        self.done_counter = 0
        self.current_hv = 0
        self.ref_point = [1, 1]
        self.front_max_distance = 0
        self.front_dis_distance = 0

    def step(self):
        self.game_step += 1


def state_data(generated):
    """truss2D_ENV.py:40-109 for the design currently on `generated` (a gen_model): runs one analysis +
    observation launch and returns x_n, A_n, A_s, A_n_ts, A_n_cs, mask."""
    g = Game_research04.__new__(Game_research04)
    g._bt = None
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        g.re_game(1, generated, 2)
    s = g._game_get_1_state()
    return s[0], s[1], s[2], s[3], s[4], s[5]


def state_data_not_norm(generated):
    """truss2D_ENV.py:112-193: nN_x_n, A_n, nN_x_e, nC_e."""
    g = Game_research04.__new__(Game_research04)
    g._bt = None
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        g.re_game(1, generated, 2)
    s = g._game_get_1_state()
    return s[8], s[1], s[9], s[10]
