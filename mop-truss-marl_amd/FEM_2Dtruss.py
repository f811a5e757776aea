"""Drop-in replacement for the reference's FEM_2Dtruss.py (Load / Node / Element / Model).

Same public surface (attribute names, method names, call order: add_load / add_node / add_element,
restore(), gen_all()), but `Model.gen_all()` does not interpret Python object graphs: it packs the
model into struct-of-arrays tensors and runs ONE launch of the MI355X step kernel in analysis mode
(truss_step with TRUSS_F_NO_DECODE, include/truss_mi355.h) through `truss_mi355.BatchedTruss`, then
mirrors the results back onto the objects:

    Model.nsc / tnsc / ttnsc / ndof      FEM_2Dtruss.py:227-261, 310-317   (bit-exact integers)
    Model.jlv, Model.d                   :264-280, :327-337
    Node.global_d, Model.v               :341-352
    Element.e_q, Model.q                 :383-386  (axial end forces; e_q[0][0] = k (u0 - u2))
    Element.prop_yeield / iscompress     :414-431
    Element.length                       :99-105
    Model.U_full, Model.r                :374-379, :393-411
    Node.target of non-top nodes         :52-56, :458-459

Not materialised (they never leave LDS on the GPU): Model.ssm, local_k, global_k, T_matrix.
There is no CPU fallback: without the HIP library `gen_all()` raises truss_mi355.TrussError.
"""
import numpy as np

import truss_mi355 as _tm

_LIB = None          # tests may inject a library object (CPU lane emulator); the product uses the HIP .so


def _library():
    return _LIB if _LIB is not None else _tm.load()


class Load():
    def __init__(self):
        self.name = 1
        self.size = [0, 0]

    def set_name(self, name):
        self.name = name

    def set_size(self, x, y):
        self.size[0] = x
        self.size[1] = y

    def __repr__(self):
        return "{0}, {1}".format(self.name, self.size)


class Node:
    def __init__(self):
        self.name = 1
        self.coord = [0, 0]
        self.res = [0, 0]
        self.loads = []
        self.global_d = []
        self.adj_ele = []
        self.connected = 0
        self.top_node = 0
        self.vertical_pair = []
        self.int_y = 0
        self.max_up = 0
        self.max_down = 0
        self.target = 0
        self.has_loady = 0

    def set_target(self):
        if self.top_node == 0:
            self.target = self.coord[1]

    def set_name(self, name):
        self.name = name

    def set_coord(self, xval, yval):
        self.coord[0] = xval
        self.coord[1] = yval
        self.int_y = yval

    def set_res(self, xres, yres):
        self.res[0] = xres
        self.res[1] = yres

    def set_load(self, load):
        self.loads.append([load])
        self.has_loady = load.size[1]

    def __repr__(self):
        return "{0}, {1}, {2}, {3}".format(self.name, self.coord, self.res, self.loads)


class Element(Node):
    def __init__(self):
        self.name = 1
        self.nodes = []
        self.em = 0
        self.area = 0
        self.dia = 0
        self.length = None
        self.e_q = []
        self.i = [[0]]
        self.section_no = 0
        self.has_changed = 0
        self.yield_stress = 235 * 1e6
        self.long_stress = self.yield_stress / 1.5
        self.iscompress = None
        self.prop_yeield = 0

    def gen_length(self):
        dx = self.nodes[1].coord[0] - self.nodes[0].coord[0]
        dy = float(self.nodes[1].coord[1]) - float(self.nodes[0].coord[1])
        self.length = (dx ** 2 + dy ** 2) ** 0.5
        return self.length

    def set_name(self, name):
        self.name = name

    def set_nodes(self, startnode, endnode):
        self.nodes.append(startnode)
        self.nodes.append(endnode)
        startnode.adj_ele.append(self.name)
        endnode.adj_ele.append(self.name)
        startnode.connected += 1
        endnode.connected += 1

    def set_em(self, emval):
        self.em = emval

    def set_area(self, area):
        self.area = area

    def set_i(self, xval):
        self.i[0][0] = xval

    def __repr__(self):
        return "{0}, {1}, {2}".format(self.nodes, self.em, self.area)


class Model():
    def __init__(self):
        self.nodes = []
        self.elements = []
        self.loads = []
        self._native = None      # (signature, TrussTopology, BatchedTruss)
        self._areas = []
        self.restore()

    def restore(self):
        self.jp, self.pj = [], []
        self.nsc, self.tnsc, self.ttnsc = [], [], []
        self.ndof = 0
        self.jlv = []
        self.d, self.v, self.u, self.q, self.f, self.r = [], [], [], [], [], []
        self.U_full = 0

    def add_load(self, load):
        self.loads.append(load)

    def add_node(self, node):
        self.nodes.append(node)

    def add_element(self, element):
        self.elements.append(element)

    def reset(self):
        self.nodes = []
        self.elements = []
        self._native = None
        self._areas = []

    # ---- packing -------------------------------------------------------------------------------
    def _pack(self):
        idx = {id(n): i for i, n in enumerate(self.nodes)}
        conn = np.array([[idx[id(e.nodes[0])], idx[id(e.nodes[1])]] for e in self.elements], np.int32)
        res = np.array([n.res for n in self.nodes], np.uint8)
        top = np.array([n.top_node for n in self.nodes], np.uint8)
        pair = None
        if all(len(n.vertical_pair) == 1 for n in self.nodes):
            pair = np.array([idx[id(n.vertical_pair[0])] for n in self.nodes], np.int32)
        # append-only table of the cross-section areas this model has used (the kernel looks areas up
        # by index); it only grows when a new area shows up, so the native topology is rarely rebuilt
        table = self._areas
        for e in self.elements:
            if float(e.area) not in table:
                table.append(float(e.area))
        areas = list(table)
        sec_of = {a: i for i, a in enumerate(areas)}
        sec = np.array([sec_of[float(e.area)] for e in self.elements], np.int32)
        em = {float(e.em) for e in self.elements}
        ls = {float(e.long_stress) for e in self.elements}
        if len(em) != 1 or len(ls) != 1:
            raise ValueError("the batched kernel takes one Young's modulus / allowable stress per model")
        # summed nodal loads (gen_pj): every loaded node must carry the same (Fx, Fy) for the
        # per-env scalar load of the C ABI; the mask says which nodes carry it
        sums = []
        for n in self.nodes:
            sx = sum(l[0].size[0] for l in n.loads)
            sy = sum(l[0].size[1] for l in n.loads)
            sums.append((float(sx), float(sy)))
        loaded = [s for s, n in zip(sums, self.nodes) if len(n.loads) != 0]
        if len(set(loaded)) > 1:
            raise ValueError("the batched kernel takes one (Fx, Fy) per model applied to a node mask")
        fx, fy = loaded[0] if loaded else (0.0, 0.0)
        mask = np.array([len(n.loads) != 0 for n in self.nodes], np.uint8)
        return conn, res, top, pair, np.array(areas), sec, em.pop(), ls.pop(), fx, fy, mask

    def gen_all(self):
        conn, res, top, pair, areas, sec, em, ls, fx, fy, mask = self._pack()
        lib = _library()
        sig = (conn.tobytes(), res.tobytes(), top.tobytes(), None if pair is None else pair.tobytes(),
               areas.tobytes(), em, ls, mask.tobytes(), lib.path)
        if self._native is None or self._native[0] != sig:
            order = np.argsort(np.array([float(n.coord[0]) for n in self.nodes]), kind="stable").astype(np.int32)
            topo = _tm.TrussTopology(conn, res, top, pair, load_mask=np.stack([mask, mask]), node_order=order,
                                     sections=np.stack([areas, np.zeros_like(areas)], axis=1), e_mod=em, long_stress=ls)
            self._native = (sig, topo, _tm.BatchedTruss(topo, 1, lib=lib, debug_f64=True))
        _, topo, env = self._native
        N, E = len(self.nodes), len(self.elements)
        x = np.array([float(n.coord[0]) for n in self.nodes])
        y = np.array([float(n.coord[1]) for n in self.nodes])
        env.set_constants(x, np.zeros(N), 0.0, 0.0, 1.0, fx, fy, 0.0)
        env.set_design(y, sec)
        env.analyze()
        r = env.results()
        if int(r["status"][0]) != 0:
            raise np.linalg.LinAlgError("Singular matrix")      # what np.linalg.solve raises at FEM:337
        nsc, ttnsc, ndof = topo.dofs(lib)
        # ---- mirror the results onto the object graph ----
        self.jp = [n.name for n in self.nodes if len(n.loads) != 0]
        self.pj = [[sum(l[0].size[0] for l in n.loads), sum(l[0].size[1] for l in n.loads)]
                   for n in self.nodes if len(n.loads) != 0]
        self.nsc = [int(v) for v in nsc]
        self.tnsc = [[int(nsc[2 * i]), int(nsc[2 * i + 1])] for i in range(N)]
        self.ttnsc = [[int(v) for v in row] for row in ttnsc]
        self.ndof = int(ndof)
        flat = np.zeros(2 * N)
        flat[np.flatnonzero(mask) * 2] = fx
        flat[np.flatnonzero(mask) * 2 + 1] = fy
        self.jlv = [[float(flat[i])] for i in range(2 * N) if nsc[i] <= ndof]
        dn = r["disp_f64"][0]
        dvec = np.zeros(ndof)
        for i in range(2 * N):
            if nsc[i] <= ndof:
                dvec[nsc[i] - 1] = dn.reshape(-1)[i]
        self.d = dvec.reshape(-1, 1)
        for i, n in enumerate(self.nodes):
            n.global_d = [[float(dn[i, 0])], [float(dn[i, 1])]]
        q0 = r["q0_f64"][0]
        self.v, self.q = [], []
        for k, e in enumerate(self.elements):
            a, b = conn[k]
            self.v.append([[float(dn[a, 0])], [float(dn[a, 1])], [float(dn[b, 0])], [float(dn[b, 1])]])
            e.gen_length()
            e.e_q = np.array([[q0[k]], [0.0], [-q0[k]], [0.0]])
            self.q.append(e.e_q)
            e.prop_yeield = abs(q0[k] / e.area) / e.long_stress
            e.iscompress = 1 if q0[k] > 0 else 0
        self.U_full = float(r["energy"][0])
        rr = [None] * (2 * N)
        for j in range(2 * N - ndof):
            rr[ndof + j] = float(r["reactions"][0, j])
        self.r = rr
        for n in self.nodes:
            n.set_target()
