"""Reset-time description of a truss topology (host side, numpy).

Mirrors the static output of the reference's structure builder `gen_model.generate`
(truss2D_GEN.py:241-434): node/element order, supports, top nodes, vertical pairs, load placement,
plus the hard-coded mirror tables of the two test copies of truss2D_ENV.py (:459-553 / :459-675)
expressed as data.  The DOF numbering itself (FEM_2Dtruss.py:227-261, 310-317) is computed by the
native library (truss_topo_create) and read back with `dofs()`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

# section_data/01_brace_rod2.csv: area [cm^2], inertia [cm^4]  (truss2D_GEN.py:29-36, 60-64)
SECTION_TABLE_CM = np.array([[9.085, 59.5], [20.41, 300.0], [38.89, 830.0], [81.23, 4230.0], [164.6, 18700.0]],
                            dtype=np.float64)
YOUNG_MODULUS = 2 * 1e11            # truss2D_GEN.py:59
YIELD_STRESS = 235 * 1e6            # FEM_2Dtruss.py:93
LONG_STRESS = YIELD_STRESS / 1.5    # FEM_2Dtruss.py:94


def sections_si(table_cm=SECTION_TABLE_CM):
    """[S,2] area m^2, inertia m^4 (truss[i][0]*1e-4, truss[i][1]*1e-8; truss2D_GEN.py:291-292)."""
    t = np.asarray(table_cm, np.float64)
    return np.stack([t[:, 0] * 1e-4, t[:, 1] * 1e-8], axis=1)


class TrussTopology:
    def __init__(self, conn, res, top, pair=None, sym_nodes=None, sym_elems=None, load_mask=None, node_order=None,
                 sections=None, e_mod=YOUNG_MODULUS, long_stress=LONG_STRESS, num_x=None):
        self.conn = np.ascontiguousarray(conn, np.int32).reshape(-1, 2)
        self.res = np.ascontiguousarray(res, np.uint8).reshape(-1, 2)
        self.top = np.ascontiguousarray(top, np.uint8).reshape(-1)
        self.pair = None if pair is None else np.ascontiguousarray(pair, np.int32).reshape(-1)
        self.N, self.E = self.res.shape[0], self.conn.shape[0]
        self.sym_nodes = np.zeros((0, 2), np.int32) if sym_nodes is None else np.ascontiguousarray(sym_nodes, np.int32)
        self.sym_elems = np.zeros((0, 2), np.int32) if sym_elems is None else np.ascontiguousarray(sym_elems, np.int32)
        if load_mask is None:   # truss2D_GEN.py:421-430
            bridge = (self.top == 0) & (self.res[:, 1] == 0)
            roof = self.top == 1
            load_mask = np.stack([bridge, roof])
        self.load_mask = np.ascontiguousarray(load_mask, np.uint8).reshape(2, self.N)
        self.node_order = None if node_order is None else np.ascontiguousarray(node_order, np.int32)
        self.sections = np.ascontiguousarray(sections_si() if sections is None else sections, np.float64)
        self.e_mod, self.long_stress = float(e_mod), float(long_stress)
        self.num_x = num_x
        self._native = {}

    # ---- the reference's parametric 2-row grid truss ----
    @classmethod
    def grid(cls, num_x, symmetry=None, extra_elements=None):
        """Nodes row-major (bottom row, then top row); elements: beams row 0, beams row 1, columns,
        '\\' braces, '/' braces (truss2D_GEN.py:280-353); pins at both bottom corners (:401-418).
        symmetry: None | 'small' | 'large' selects the mirror tables of test/00,01 resp. test/02,03.
        extra_elements: optional [[n0,n1],...] appended after the reference families (synthetic
        benchmark topologies only)."""
        nx = int(num_x)
        N = 2 * nx
        conn = [(r * nx + i, r * nx + i + 1) for r in range(2) for i in range(nx - 1)]
        conn += [(i, nx + i) for i in range(nx)]
        conn += [(nx + i, i + 1) for i in range(nx - 1)]
        conn += [(i, nx + i + 1) for i in range(nx - 1)]
        if extra_elements is not None:
            conn += [tuple(e) for e in extra_elements]
        res = np.zeros((N, 2), np.uint8)
        res[0] = 1
        res[nx - 1] = 1
        top = np.zeros(N, np.uint8)
        top[nx:] = 1
        pair = np.concatenate([np.arange(nx, N), np.arange(0, nx)])
        sym_nodes = sym_elems = None
        if symmetry is not None:
            half, nb = nx // 2, nx - 1
            if symmetry == "small":      # coin: left <- right, supports excluded
                sym_nodes = [(i, nx - 1 - i) for i in range(1, half)] + [(nx + i, N - 1 - i) for i in range(half)]
            elif symmetry == "large":    # coin: right <- left, supports included
                sym_nodes = [(nx - 1 - i, i) for i in range(half)] + [(N - 1 - i, nx + i) for i in range(half)]
            else:
                raise ValueError(f"unknown symmetry variant {symmetry!r}")
            se = [(r * nb + i, r * nb + nb - 1 - i) for r in range(2) for i in range(nb // 2)]
            se += [(2 * nb + i, 2 * nb + nx - 1 - i) for i in range(nx // 2)]
            b0 = 2 * nb + nx
            se += [(b0 + i, b0 + 2 * nb - 1 - i) for i in range(nb)]
            sym_elems = se
        order = np.stack([np.arange(nx), nx + np.arange(nx)], axis=1).reshape(-1)   # column by column
        return cls(conn, res, top, pair, sym_nodes, sym_elems, node_order=order, num_x=nx)

    # ---- native handle ----
    def native(self, lib: "_lib.TrussLib", device_index=None):
        """Native handle (host tables + device blob).  The blob is allocated on the HIP device that is current
        at creation: handles are cached per (library, device index); `device_index` None = the current device
        (BatchedTruss passes its own and makes that device current around the call)."""
        if device_index is None and lib.backend == "hip":
            import torch
            device_index = torch.cuda.current_device()
        key = (lib.path, device_index)
        if key not in self._native:
            h = C.c_void_p()

            def ptr(a):
                return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)

            rc = lib.dll.truss_topo_create(
                C.byref(h), self.N, self.E, ptr(self.conn), ptr(self.res), ptr(self.top), ptr(self.pair),
                ptr(self.load_mask), self.sym_nodes.shape[0], ptr(self.sym_nodes), self.sym_elems.shape[0],
                ptr(self.sym_elems), self.sections.shape[0], ptr(self.sections), self.e_mod, self.long_stress,
                ptr(self.node_order))
            lib.check(rc, "truss_topo_create")
            self._native[key] = (lib, h)
        return self._native[key][1]

    def dofs(self, lib):
        """(nsc[2N], ttnsc[E,4], ndof) exactly as Model.gen_nsc/gen_ssm number them."""
        nsc = np.zeros(2 * self.N, np.int32)
        tt = np.zeros((self.E, 4), np.int32)
        nd = lib.check(lib.dll.truss_topo_dofs(self.native(lib), nsc.ctypes.data_as(C.c_void_p),
                                               tt.ctypes.data_as(C.c_void_p)), "truss_topo_dofs")
        return nsc, tt, nd

    def solver_info(self, lib):
        _, _, nd = self.dofs(lib)
        perm = np.zeros(nd, np.int32)
        bw, g, rpl = C.c_int32(), C.c_int32(), C.c_int32()
        lib.check(lib.dll.truss_topo_solver_info(self.native(lib), perm.ctypes.data_as(C.c_void_p), C.byref(bw),
                                                 C.byref(g), C.byref(rpl)), "truss_topo_solver_info")
        return dict(perm=perm, half_bandwidth=bw.value, lanes_per_env=g.value, rows_per_lane=rpl.value)

    def close(self):
        for lib, h in self._native.values():
            lib.dll.truss_topo_destroy(h)
        self._native = {}

    # ---- topology-static observation pieces (truss2D_ENV.py:83-84, 89-90, 104-107, 178-179) ----
    def normalized_adjacency(self):
        A = np.zeros((self.N, self.N), np.float32)
        A[self.conn[:, 0], self.conn[:, 1]] = 1
        A[self.conn[:, 1], self.conn[:, 0]] = 1
        mask = A.copy()
        A = A + np.eye(self.N, dtype=np.float32)
        with np.errstate(divide="ignore"):
            d = np.power(np.array(A.sum(1)), -1 / 2).ravel()     # spektral.utils.degree_power(A, -1/2)
        d[np.isinf(d)] = 0.0
        D = np.diag(d)
        return np.matmul(D, np.matmul(A, D)).astype(np.float32), mask

    def neighbor_table(self):
        """int16 [N, K]: per node the columns in which a node-graph adjacency of this truss (A_n, A_s, A_n_ts, A_n_cs,
        truss2D_ENV.py:40-193) can be non-zero -- the node itself and the far ends of its members -- ascending, padded with -1.
        The sparsity pattern for `truss_gcn_aggregate_sparse`."""
        nb = [{n} for n in range(self.N)]
        for a, b in self.conn.tolist():
            nb[a].add(b)
            nb[b].add(a)
        K = max(len(s) for s in nb)
        tab = np.full((self.N, K), -1, np.int16)
        for n, s in enumerate(nb):
            tab[n, :len(s)] = sorted(s)
        return tab

    def incidence(self):
        c = np.zeros((self.E, self.N), np.float32)
        c[np.arange(self.E), self.conn[:, 0]] = 1
        c[np.arange(self.E), self.conn[:, 1]] = 1
        return c
