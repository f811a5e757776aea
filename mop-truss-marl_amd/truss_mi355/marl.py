"""Batched multi-agent rollout + training loop (BASELINE configs 3-5: thousands of envs with the MADDPG
agents in the loop).  The reference's `run()` (master_DDPG_truss2D_MO.py:164-705) plays ONE truss: per game
step it lets the three agents modify every member of the Pareto archive, scores the moves with the
hypervolume-difference reward, culls the archive and trains.  Here the same game step runs for B trusses at
once, every piece on the device:

    parents' analysis + observation   BatchedTruss.analyze / observe     (truss_step, truss_obs kernels)
    three agents' actions             truss2D_RL actors on a [B, N, .] batch (rocBLAS)
    the 3 B candidate designs         one truss_step launch over 3 B envs (+ truss_obs for the next states)
    rewards                           reward.difference_reward            (truss_front kernel)
    archive update                    reward.front_hv + gathers
    replay + MADDPG update            device tensors, MADDPG.train_on_batch

One game step is ONE flat pass over all live (env, archive member) pairs (member-major list, in chunks of at most
`pair_capacity` pairs): one parent analysis, one actor pass per agent, one candidate launch, one reward evaluation for
all of them -- not one pass per member index, whose later, thin iterations were bound by launch overhead.

Differences from the per-env loop, all forced by batching and documented here:
  D1  the archive is culled once per chunk of pairs -- for an archive of up to 14 members (MAX_FRONT 20: the front
      kernel takes 64 rows = 20 archive rows + the 3 x 14 candidates of 14 members) that is once per game step, like the
      reference (:430); larger archives are folded in two rounds (the Pareto front of a union is the front of the
      partial fronts, so the set is the same unless the MAX_FRONT truncation intervenes); rewards use the front at
      the START of the game step, like the reference (front_no / Pf_HV, :263-368);
  D2  a transition enters the replay when its candidate is in the archive after that cull (the reference remembers the
      candidates that survive the end-of-step cull, :595-621);
  D3  truncation to MAX_FRONT is deterministic (largest crowding distance), see include/truss_mi355.h;
  D4  when an agent's move is infeasible its next state in the replay is the first feasible agent's
      (the reference draws a random survivor, :380-407); the Pareto-graph inputs (x_p, A_p) of the next
      states are those of the step-start front;
  D5  exploration noise is drawn on the device (same law as truss2D_RL.OUNoise: theta (mu - a) dt + sigma N(0,1)).
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import reward as RW
from .batched import BatchedTruss
from .topology import TrussTopology


_TUNE_UPDATE_GEMMS = False


def enable_gemm_tuning(max_ms_per_shape: int = 30, filename: str | None = None):
    """Let PyTorch's TunableOp choose the library GEMM for the shapes of the MADDPG UPDATE (fixed: batch 32), timed once
    during the eager warm-up updates that precede the hipGraph capture (BatchedMARL._train).  The weight gradients of the
    GCN layers (200 x 200 outputs, K = 32 x 16 = 512) otherwise get a single 256 x 224 tile from the library's heuristic:
    118 us each, half of an update.  The inference GEMMs are left alone: their row count (live (env, member) pairs x nodes)
    changes every game step and every new shape would be tuned again (11 s for four game steps when tried).
    No-op on the CPU backend."""
    global _TUNE_UPDATE_GEMMS
    if not torch.cuda.is_available():
        return False
    t = torch.cuda.tunable
    t.enable(True)                 # use tuned solutions where a shape has one
    t.tuning_enable(False)         # ... but time new shapes only inside the update's warm-up
    t.set_max_tuning_duration(int(max_ms_per_shape))
    t.set_max_tuning_iterations(20)
    filename = filename or os.environ.get("TRUSS_GEMM_TUNE_FILE")   # keep / reuse the choices (a profiled run reads them: no tuning)
    if filename:
        t.set_filename(filename)
        if hasattr(t, "write_file_on_exit"):
            t.write_file_on_exit(True)
    elif hasattr(t, "write_file_on_exit"):
        t.write_file_on_exit(False)    # no tunableop_results*.csv in the caller's working directory
    _TUNE_UPDATE_GEMMS = True
    return True


def pareto_graph(pts, n, index, max_front):
    """Batched truss2D_ENV.pareto_state_data (:19-38) + zero padding to P nodes (master…:488-593).
    pts [B,P,4] (rows beyond n ignored), n [B], index [B] -> x_p [B,P,4], A_p [B,P,P] float32."""
    B, P, _ = pts.shape
    dev = pts.device
    ar = torch.arange(P, device=dev)
    vf = (ar[None, :] < n[:, None]).float()                                             # [B, P] 1 on the front's members
    x = torch.stack([pts[:, :, 0].float(), pts[:, :, 1].float(), (ar[None, :] == index[:, None]).float(),
                     (n.float() / max_front)[:, None].expand(B, P)], dim=2) * vf[:, :, None]
    # a path over the members with self loops, symmetrically normalised: deg_i = 1 + (left neighbour) + (right neighbour) on members
    z = torch.zeros((B, 1), dtype=torch.float32, device=dev)
    deg = vf * (1.0 + torch.cat([z, vf[:, :-1]], dim=1) + torch.cat([vf[:, 1:], z], dim=1))
    d = torch.where(deg > 0, deg.pow(-0.5), 0.0)
    return x, _tridiagonal(P, dev) * d[:, :, None] * d[:, None, :]


_TRI: dict = {}


def _tridiagonal(P, dev):
    """[P, P] ones on the three central diagonals (the path graph with self loops), cached per size and device"""
    key = (P, str(dev))
    if key not in _TRI:
        i = torch.arange(P, device=dev)
        _TRI[key] = ((i[:, None] - i[None, :]).abs() <= 1).float()
    return _TRI[key]


def gcn_aggregate(lib, adj, h, bias, act, nbr=None):
    """act(adj @ h + bias) through the fused HIP kernels `truss_gcn_aggregate` / `truss_gcn_aggregate_sparse` (inference only, float32).
    adj [N,N] (shared) or [B,N,N]; h [B,N,C] contiguous; act in {None,'relu','sigmoid'}.
    nbr: int16 [N,K] device table of the columns that can be non-zero in each row of `adj` (TrussTopology.neighbor_table());
    with it, graphs above 32 nodes sum over those K columns only (tools/agg_probe.py: 64 nodes 39 us against 74 us for the
    library's batched GEMM + bias + activation, 256 nodes 42 against 104; at 32 nodes the dense channel-quad kernel is as fast, at 16 faster)."""
    B, N, Cc = h.shape
    from . import ops
    code = {None: 0, "relu": 1, "sigmoid": 2}[act]
    if nbr is not None and N > 32 and Cc % 4 == 0 and nbr.shape[1] <= 16:
        adj = adj.contiguous()
        if adj.dim() == 3 and adj.shape[0] == 1:
            adj = adj[0]
        out = torch.empty_like(h)
        ops.call(ops.namespace().gcn_aggregate_sparse, ops.bind(lib), ops.stream_of(h.device), adj, nbr, h, bias, out, code)
        return out
    if N > 64:     # larger graphs without a pattern: a batched N x N x C GEMM (rocBLAS); the fused dense kernels cover N <= 64
        out = torch.matmul(adj, h) + bias
        return torch.relu(out) if act == "relu" else torch.sigmoid(out) if act == "sigmoid" else out
    adj = adj.contiguous()
    if adj.dim() == 3 and adj.shape[0] == 1:
        adj = adj[0]
    out = torch.empty_like(h)
    ops.call(ops.namespace().gcn_aggregate, ops.bind(lib), ops.stream_of(h.device), adj, h, bias, out, code)
    return out


def split_weights(lib, w):
    """w [C <= 224, K] float32 -> int16 [3, 224, KP] (bfloat16 bit patterns, zero padded): the exact three-term split the bf16x3 path
    of the fused layer kernel reads (`truss_gcn_split_w`; one small launch).  `layer_split_weights` caches it per GCN layer."""
    from . import ops
    C, K = w.shape
    out = torch.empty((3, 224, (K + 15) // 16 * 16), dtype=torch.int16, device=w.device)
    ops.call(ops.namespace().gcn_split_w, ops.bind(lib), ops.stream_of(w.device), w, out)
    return out


def layer_split_weights(lib, layer, k_pad=0):
    """(w, split_weights(w)) of a truss2D_RL.GCNConv, kept on the module and redone when its kernel has changed (training updates the
    weights in place: tensor version counter; load_state_dict / re-materialisation: data pointer).  k_pad > k_in: the kernel's input
    columns are zero-padded to k_pad first (the 13-feature input layers run at 16 so that they take the 16-byte loaders)."""
    w = layer.lin.weight
    tag = (w._version, w.data_ptr(), tuple(w.shape), k_pad)
    hit = getattr(layer, "_truss_split", None)
    if hit is None or hit[0] != tag:
        wd = w.detach()
        if k_pad > wd.shape[1]:
            wd = torch.nn.functional.pad(wd, (0, k_pad - wd.shape[1])).contiguous()
        hit = (tag, wd, split_weights(lib, wd) if (wd.shape[0] > 32 and wd.shape[1] % 4 == 0) else None)
        layer._truss_split = hit
    return hit[1], hit[2]


def gcn_layer(lib, x, adj, w, bias, act, nbr=None, out=None, accumulate=False, precision="bf16x3", w_split=None):
    """One whole GCN layer through the hand-written MFMA kernel (`truss_gcn_layer`, csrc/truss_gcn.h):
    out = act(adj @ (x @ w.T) + bias), or out += ... with `accumulate`; float32 in and out, inference (no autograd).
    x [B,N,K] contiguous; adj [N,N] (shared) or [B,N,N]; w [C,K] = nn.Linear.weight, C <= 224; nbr: int16 [N,Kn] sparsity pattern of
    adj (TrussTopology.neighbor_table(), Kn <= 16) or None for a dense adjacency of at most 64 nodes; N <= 256.
    precision "bf16x3" (default): where the shape allows (the hidden layers: C > 32, K % 4 == 0, <= 9 terms per row) the product runs
    on the bf16 matrix cores as six partial products of exactly split operands with float32 accumulation -- float32 accuracy, 2.7 x
    fewer matrix-core cycles; "f32": always the float32 matrix cores.  w_split: split_weights(lib, w) if the caller keeps it."""
    from . import ops
    B, N, K = x.shape
    C = w.shape[0]
    if out is None:
        assert not accumulate
        out = torch.empty((B, N, C), dtype=torch.float32, device=x.device)
    adj = adj.contiguous()
    if adj.dim() == 3 and adj.shape[0] == 1:
        adj = adj[0]
    code = {None: 0, "relu": 1, "sigmoid": 2}[act]
    ws = None
    if precision == "bf16x3" and C > 32 and K % 4 == 0 and x.data_ptr() % 16 == 0 and (nbr.shape[1] if nbr is not None else N) <= 9:
        ws = w_split if w_split is not None else split_weights(lib, w)
    ops.call(ops.namespace().gcn_layer, ops.bind(lib), ops.stream_of(x.device), x, adj, nbr, w, bias, out, code, bool(accumulate), ws)
    return out


def level_forward(lib):
    """The fused forward of a whole level of GCN layers (`truss_gcn_level`, csrc/truss_gcn_level.h: one launch for every layer of
    every group) in the shape truss2D_RL._GcnLevel asks for: callable(groups, xs, ws, bs, want_grad) -> ([out_g [n, B, N, C]],
    [X'_g [n, B N, K] or None]), or None when a shape is outside the kernel's envelope (the caller then evaluates the level with
    batched library GEMMs).  Installed by BatchedMARL on the GPU (`truss2D_RL.set_level_forward`)."""
    from . import ops
    code = {None: 0, "relu": 1, "sigmoid": 2}

    def run(groups, xs, ws, bs, want_grad):
        X, ADJ, W, BIAS, OUT, XAGG, ACT, outs, xaggs = [], [], [], [], [], [], [], [], []
        for g in groups:
            n, (B, N, K), C = len(g.idx), g.shape, ws[g.idx[0]].shape[0]
            if N > 64 or C > 224 or K > 256 or xs[g.idx[0]].dtype != torch.float32:      # outside the kernel's envelope
                return None
            o = torch.empty((n, B, N, C), dtype=torch.float32, device=xs[g.idx[0]].device)
            xa = torch.empty((n, B * N, K), dtype=torch.float32, device=o.device) if want_grad else None
            for j, i in enumerate(g.idx):
                a = g.adjs[j]
                if a.dim() == 3 and (a.shape[0] == 1 or a.stride(0) == 0):
                    a = a[0]
                X.append(xs[i].detach().contiguous())
                ADJ.append(a.contiguous())
                W.append(ws[i].detach())
                BIAS.append(bs[i].detach())
                OUT.append(o[j])
                if want_grad:
                    XAGG.append(xa[j])
                ACT.append(code[g.act])
            outs.append(o)
            xaggs.append(xa)
        ops.call(ops.namespace().gcn_level, ops.bind(lib), ops.stream_of(X[0].device), X, ADJ, [], W, BIAS, OUT, XAGG, ACT)
        return outs, xaggs
    return run


def gcn_layer_supported(n_nodes, c_out, nbr):
    """shapes the fused layer kernel takes (include/truss_mi355.h); anything else goes through library GEMM + aggregation kernels"""
    return c_out <= 224 and n_nodes <= 256 and ((nbr is not None and nbr.shape[1] <= 16) or (nbr is None and n_nodes <= 64))


def path_graph_table(n_nodes):
    """int16 [P, 3] neighbour table of the Pareto graph: `pareto_graph` / truss2D_ENV.pareto_state_data build a PATH over the front's
    members (self loops + consecutive members), so row i of A_p is zero outside columns i - 1, i, i + 1"""
    t = np.full((n_nodes, 3), -1, np.int16)
    for i in range(n_nodes):
        t[i, 0] = i - 1 if i > 0 else -1
        t[i, 1] = i
        t[i, 2] = i + 1 if i + 1 < n_nodes else -1
    return t


def actor_infer(lib, actor, ins, nbr=None, nbr_p=None):
    """truss2D_RL.multimodes_actor.forward (truss2D_RL.py:49-120) for inference, every GCN layer ONE launch of the fused MFMA
    kernel (`gcn_layer`: neighbourhood sum on the input rows, product with W^T on the matrix cores, bias + activation in the
    epilogue; H = X W never exists in HBM); the five second-level layers accumulate their sum x3 in place.  Same values as the
    module up to float32 summation order ((A X) W instead of A (X W)).
    nbr: the truss's neighbour table on the device (TrussTopology.neighbor_table()) for the layers over the node graph; nbr_p: the
    same for the Pareto graph (path_graph_table(P) when A_p comes from `pareto_graph`; None = dense, at most 64 members).  Shapes outside the kernel's envelope (a node graph without pattern above 64 nodes)
    fall back to library GEMM + `gcn_aggregate`."""
    x_n, A_n, A_s, A_ts, A_cs, x_p, A_p = ins

    def g(layer, x, a, act="relu", out=None, accumulate=False):
        if isinstance(layer.lin.weight, torch.nn.parameter.UninitializedParameter):   # lazy layers: let the module materialise itself once
            with torch.no_grad():
                layer(x[:1], a[:1] if a.dim() == 3 else a)
        w, bvec = layer.lin.weight, layer.bias
        pat = nbr_p if a is A_p else nbr
        x = x.contiguous()
        if gcn_layer_supported(x.shape[1], w.shape[0], pat):
            wd, ws = layer_split_weights(lib, layer, x.shape[2])     # (x_n arrives zero-padded to 16 features)
            return gcn_layer(lib, x, a, wd, bvec.detach(), act, pat, out, accumulate, w_split=ws)
        h = gcn_aggregate(lib, a, torch.nn.functional.linear(x, w).contiguous(), bvec, act, pat)
        if out is None:
            return h
        return out.add_(h) if accumulate else out.copy_(h)

    a = actor
    if x_n.shape[2] % 4 and gcn_layer_supported(x_n.shape[1], 200, nbr):
        # 13 node features -> 16 (zero columns, matched by zero columns of the three input kernels): 16-byte loads, bf16x3 product
        for layer in (a.gcn_l1_1, a.gcn_l1_2, a.gcn_l1_3):
            if isinstance(layer.lin.weight, torch.nn.parameter.UninitializedParameter):
                with torch.no_grad():
                    layer(x_n[:1], A_n[:1] if A_n.dim() == 3 else A_n)
        x_n = torch.nn.functional.pad(x_n, (0, 4 - x_n.shape[2] % 4))
    x11, x12, x13 = g(a.gcn_l1_1, x_n, A_n), g(a.gcn_l1_2, x_n, A_n), g(a.gcn_l1_3, x_n, A_n)
    x14 = g(a.gcn_l1_4, x_p, A_p).sum(dim=1)                                             # GlobalSumPool over the Pareto graph
    B, H = x14.shape
    x14 = x14.unsqueeze(-1).expand(B, H, x11.shape[1]).reshape(B, x11.shape[1], H)      # _tile_pool (:87-93)
    x3 = g(a.gcn_l2_1, x11, A_n)
    g(a.gcn_l2_2, x12, A_ts, out=x3, accumulate=True)
    g(a.gcn_l2_3, x12, A_cs, out=x3, accumulate=True)
    g(a.gcn_l2_4, x13, A_s, out=x3, accumulate=True)
    g(a.gcn_l2_5, x14.contiguous(), A_n, out=x3, accumulate=True)
    x31, x32 = g(a.gcn_l3_1, x3, A_n), g(a.gcn_l3_2, x3, A_s)
    return g(a.gcn_l4_1, x31, A_n, "sigmoid"), g(a.gcn_l4_2, x32, A_n, "sigmoid")


class DeviceReplay:
    """Ring buffer of transitions in device memory (state / three next states as the eight observation
    tensors the networks take minus the topology-static A_n and mask, three agents' actions, rewards)."""

    KEYS = ("x_n", "A_s", "A_n_ts", "A_n_cs", "x_p", "A_p")

    def __init__(self, capacity, N, P, device):
        f = lambda *s: torch.zeros((capacity,) + s, dtype=torch.float32, device=device)
        shapes = dict(x_n=(N, 13), A_s=(N, N), A_n_ts=(N, N), A_n_cs=(N, N), x_p=(P, 4), A_p=(P, P))
        self.S = {k: f(*shapes[k]) for k in self.KEYS}
        self.NS = [{k: f(*shapes[k]) for k in self.KEYS} for _ in range(3)]
        self.a_geo, self.a_topo, self.R = f(3, N, 2), f(3, N, 3), f(3)
        self.capacity, self.size, self.head = capacity, 0, 0

    def add(self, sel, S, NS, a_geo, a_topo, R, src=None):
        """append the transitions of the rows where sel[K] is True.  NS: three dicts (one per agent's next state), or -- with src
        [K, 3] -- ONE dict of tensors [3, K, ...] from which agent a's next state of row r is taken at [src[r, a], r]"""
        idx = torch.nonzero(sel, as_tuple=False).flatten()
        k = int(idx.numel())
        if k == 0:
            return 0
        if k > self.capacity:
            idx, k = idx[: self.capacity], self.capacity
        pos = (self.head + torch.arange(k, device=idx.device)) % self.capacity
        pick = None if src is None else src[idx]                               # [k, 3]
        for key in self.KEYS:
            self.S[key][pos] = S[key][idx]
            for a in range(3):
                self.NS[a][key][pos] = NS[a][key][idx] if src is None else NS[key][pick[:, a], idx]
        self.a_geo[pos], self.a_topo[pos], self.R[pos] = a_geo[idx], a_topo[idx], R[idx]
        self.head = (self.head + k) % self.capacity
        self.size = min(self.capacity, self.size + k)
        return k

    def sample(self, batch, generator=None):
        i = torch.randint(0, self.size, (batch,), device=self.R.device, generator=generator)
        pick = lambda d: {k: v[i] for k, v in d.items()}
        return pick(self.S), [pick(ns) for ns in self.NS], self.a_geo[i], self.a_topo[i], self.R[i]


class BatchedMARL:
    def __init__(self, topo: TrussTopology, n_envs: int, maddpg, *, max_front: int = 20, lib=None, device=None,
                 replay_capacity: int = 32768, batch_size: int = 32, hv_margin: float = 0.2, seed: int = 0,
                 pair_capacity: int | None = None, tune_update_gemms: bool = True):
        self.topo, self.B, self.P = topo, int(n_envs), int(max_front)
        self.rl = maddpg
        if tune_update_gemms and (device is None or torch.device(device).type == "cuda"):
            enable_gemm_tuning()      # library GEMM per (fixed) shape of the update, chosen during its warm-up (23 -> 14 ms per update)
        # members whose candidates fit one cull: P archive rows + 3 candidates per member <= the front kernel's 64 rows
        self.Gm = max(1, min(self.P, (64 - self.P) // 3))
        # (env, member) pairs per pass: the env objects below hold that many designs (3 x as many candidates)
        self.cap = int(pair_capacity) if pair_capacity else min(self.B * self.Gm, 4 * self.B)
        self.cap = max(self.cap, self.B)
        self.envP = BatchedTruss(topo, self.cap, device=device, lib=lib)          # archive members under study
        self.envC = BatchedTruss(topo, 3 * self.cap, device=device, lib=lib)      # their candidates, agent-major
        self.lib, self.device = self.envP.lib, self.envP.device
        if self.device.type == "cuda" and os.environ.get("TRUSS_LEVEL_FORWARD", "1") != "0":
            import truss2D_RL
            truss2D_RL.set_level_forward(level_forward(self.lib), "cuda")   # the update's forward passes: one launch per level
        dev, B, P, N, E = self.device, self.B, self.P, topo.N, topo.E
        A_n, mask = topo.normalized_adjacency()
        self.A_n = torch.tensor(A_n, device=dev)[None]
        self.mask = torch.tensor(mask, device=dev)[None]
        self.nbr = torch.tensor(topo.neighbor_table(), device=dev)     # sparsity pattern of every node-graph adjacency (actor inference)
        self.nbr_p = torch.tensor(path_graph_table(P), device=dev)     # ... and of the Pareto graph (a path over the front's members)
        self.pts = torch.zeros((B, P, 4), dtype=torch.float64, device=dev)
        self.arch_y = torch.zeros((B, P, N), dtype=torch.float32, device=dev)
        self.arch_sec = torch.zeros((B, P, E), dtype=torch.int32, device=dev)
        self.n = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.ref_points = torch.ones((B, 2), dtype=torch.float64, device=dev)
        self.replay = DeviceReplay(replay_capacity, N, P, dev)
        self.batch_size, self.hv_margin = batch_size, hv_margin
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.game_step = 1
        self._steps_dev = torch.zeros((), dtype=torch.int64, device=self.device)
        self._synced = False
        self._tg, self.use_train_graph = None, True
        self.profile = None            # set to {} to accumulate synchronised wall time per segment (diagnostic)

    def _tick(self, name, t0):
        if self.profile is None:
            return t0
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t1 = time.perf_counter()
        self.profile[name] = self.profile.get(name, 0.0) + (t1 - t0)
        return t1

    @property
    def env_steps(self):
        """agent modifications evaluated so far (3 per live archive member per game step)"""
        return int(self._steps_dev.item())

    # ---- episode set-up (Game_research04.__init__ / ENV.reset / run() prologue, :164-196) ----
    def reset(self, x, target, y_max, d_min, max_def, load_x, load_y, is_roof, y0, sec0):
        def tiled(a, rep):                                   # per-env arrays [B] / [B, N] -> [rep B, ...]; scalars as they are
            a = np.asarray(a, np.float64)
            if a.ndim == 0:
                return a
            a = np.broadcast_to(a, (self.B,) + a.shape[1:]) if a.shape[0] != self.B else a
            return np.tile(a, (rep,) + (1,) * (a.ndim - 1))
        # the reset analysis of the B designs (normalisers int_obj1 / int_obj2, :264-274) on a B-env object
        e0 = BatchedTruss(self.topo, self.B, device=self.device, lib=self.lib)
        e0.set_constants(*[tiled(a, 1) for a in (x, target, y_max, d_min, max_def, load_x, load_y, is_roof)])
        e0.set_design(y0, sec0)
        e0.analyze(set_normalisers=True)
        # per-env constants of the whole batch: the pair-sized env objects are re-filled with the constants of the LIVE
        # (env, member) pairs of every pass, compacted to the front (BatchedTruss n_active)
        self.c_x, self.c_target, self.c_params = e0.x.clone(), e0.target.clone(), e0.env_params.clone()
        self._y_reset, self._sec_reset = e0.y.clone(), e0.sec.clone()
        self.pts.zero_(); self.arch_y.zero_(); self.arch_sec.zero_()
        self.pts[:, 0, 0:2] = 1.0                                    # Pf = [[1, 1, 0, 0, S0, ...]] (:168)
        self.arch_y[:, 0] = self._y_reset
        self.arch_sec[:, 0] = self._sec_reset
        self.n.fill_(1)
        self.ref_points.fill_(1.0)
        self.game_step = 1

    # ---- observation tensors in the networks' order ----
    def _obs(self, env, pts0, n0, index, rep=1, k=None, o=None, graph=None):
        """o: observation tensors the analysis / step call has already written (TRUSS_F_EMIT_OBS); None = observation kernel.
        graph: (x_p, A_p) of the same (pts0, n0, index) if the caller has them already (the candidates see their parent's front)"""
        k = env.B if k is None else k
        if o is None:
            o = env.observe(n_active=k)
        x_p, A_p = graph if graph is not None else pareto_graph(pts0, n0, index, self.P)
        if rep > 1:
            x_p, A_p = x_p.repeat(rep, 1, 1), A_p.repeat(rep, 1, 1)
        # views of the env's own observation buffers: the caller is done with them (actor inference, replay rows) before the same env
        # object analyses / steps again
        c = lambda key: o[key][:k]
        return dict(x_n=c("x_n"), A_s=c("A_s"), A_n_ts=c("A_n_ts"), A_n_cs=c("A_n_cs"), x_p=x_p, A_p=A_p)

    def _net_state(self, S):
        b = S["x_n"].shape[0]
        return [S["x_n"], self.A_n.expand(b, -1, -1), S["A_s"], S["A_n_ts"], S["A_n_cs"], self.mask.expand(b, -1, -1), S["x_p"], S["A_p"]]

    def _noise_vectors(self, noises):
        """(theta dt, mu, sigma) of a list of OUNoise objects as per-column device vectors (cached on the list's first object)"""
        hit = getattr(noises[0], "_dev_vectors", None)
        if hit is None or hit[0].device != self.device:
            f = lambda vals: torch.tensor(vals, dtype=torch.float32, device=self.device)
            hit = (f([nz.theta * nz.dt for nz in noises]), f([nz.mu for nz in noises]), f([nz.sigma for nz in noises]))
            noises[0]._dev_vectors = hit
        return hit

    def _act(self, S, explore):
        ins = self._net_state(S)
        actor_in = [ins[0], ins[1], ins[2], ins[3], ins[4], ins[6], ins[7]]
        geo, topo = [], []
        with torch.no_grad():
            for ag in self.rl.agents:
                g, t = actor_infer(self.lib, ag.actor_model, [actor_in[0], self.A_n[0], actor_in[2], actor_in[3], actor_in[4],
                                                              actor_in[5], actor_in[6]], nbr=self.nbr, nbr_p=self.nbr_p)
                if explore:                                           # truss2D_RL.OUNoise.gen_noise per scalar (:41-48), all columns at once
                    for out, noises in ((g, ag.noise_geo), (t, ag.noise_topo)):
                        th_dt, mu, sg = self._noise_vectors(noises)
                        out.addcmul_(th_dt, mu - out).addcmul_(sg, torch.randn(out.shape, device=out.device, generator=self.gen))
                geo.append(g.float().contiguous())
                topo.append(t.float().contiguous())
        return geo, topo

    # ---- the MADDPG update: eager, or (one GPU) replayed as a hipGraph ----
    def _train(self, S, NS, A, R):
        """The update is thousands of small kernels at batch 32 -- launch-bound.  On the GPU it is captured once into a hipGraph
        (static input buffers, capturable Adam) and replayed -- data-parallel too: the gradient all-reduces (RCCL; one flat
        buffer for the three critics, one per actor) are captured WITH the update, every rank replaying its own copy of the same
        graph.  On the CPU backend (gloo tests) it runs eagerly."""
        if self.device.type != "cuda" or not self.use_train_graph:
            return self.rl.train_on_batch(S, NS, A, R)
        flat_in = list(S) + [t for ns in NS for t in ns] + [t for a in A for t in a] + [R]
        if self._tg is None:
            try:
                # static input buffers of the graph.  Inputs that are broadcast constants (the shared adjacency A_n, the mask: stride 0
                # over the batch) are captured as they are -- nothing to copy per update, and the passes see ONE shared adjacency
                bufs = [t if (t.dim() and t.stride(0) == 0) else t.clone() for t in flat_in]
                nS = len(S)

                def unpack(b):
                    s_ = b[:nS]
                    ns_ = [b[nS * (1 + k): nS * (2 + k)] for k in range(3)]
                    a_ = b[4 * nS: 4 * nS + 6]
                    return s_, ns_, [(a_[0], a_[1]), (a_[2], a_[3]), (a_[4], a_[5])], b[4 * nS + 6]

                # the warm-up runs real updates: weights and optimiser moments are put back afterwards (in place,
                # the graph holds their addresses), so that training is what it would have been without capture
                nets = [n for ag_ in self.rl.agents for n in (ag_.actor_model, ag_.critic_model, ag_.target_actor_model,
                                                                ag_.target_critic_model)]
                self.rl._ensure_ready(S, [A[0][0], A[0][1], A[1][0], A[1][1], A[2][0], A[2][1]])
                snap = [[p.detach().clone() for p in n.parameters()] for n in nets]
                osnap = [t.clone() for t in self.rl.critics_opt.state_tensors()]    # [] = no step taken yet (one optimiser for the three critics)
                n_loss = [len(ag_.c_loss) for ag_ in self.rl.agents]
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    if _TUNE_UPDATE_GEMMS:
                        torch.cuda.tunable.tuning_enable(True)           # the update's GEMM shapes are fixed: pick their kernels now
                    try:
                        for _ in range(3):                               # warm-up on the capture stream
                            self.rl.train_on_batch(*unpack(bufs))
                    finally:
                        if _TUNE_UPDATE_GEMMS:
                            torch.cuda.tunable.tuning_enable(False)
                torch.cuda.current_stream(self.device).wait_stream(side)
                torch.cuda.synchronize(self.device)
                d_ = getattr(self.rl, "dist", None)
                if d_ is not None and d_.is_initialized():
                    # The process group's watchdog thread polls the events of every eagerly enqueued collective until it has seen
                    # it complete (about every 100 ms).  A poll that lands inside the capture below, on an event of the stream the
                    # captured collectives run on, aborts the process (hipErrorCapturedEvent; this PyTorch build does not hold a
                    # capture back for pending polls).  The warm-up's collectives have completed (synchronize above): wait until
                    # the watchdog has retired them.
                    time.sleep(0.5)

                def restore():      # weights, Adam moments and loss log back to the state before the warm-up
                    with torch.no_grad():
                        for n, sp in zip(nets, snap):
                            for p, v in zip(n.parameters(), sp):
                                p.copy_(v)
                        for k, t in enumerate(self.rl.critics_opt.state_tensors()):
                            t.copy_(osnap[k]) if osnap else t.zero_()
                    for ag_, nl in zip(self.rl.agents, n_loss):
                        del ag_.c_loss[nl:]

                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g, stream=side):
                        self.rl.train_on_batch(*unpack(bufs))
                finally:
                    restore()       # captured or not: training continues from the pre-warm-up state
                self._tg = (g, bufs)
            except RuntimeError as e:                                    # capture is an optimisation, never a requirement
                print(f"[marl] hipGraph capture of the MADDPG update failed ({type(e).__name__}: {e}); running eagerly")
                self.use_train_graph = False
                torch.cuda.synchronize(self.device)
                return self.rl.train_on_batch(S, NS, A, R)
        g, bufs = self._tg
        pairs = [(b, t) for b, t in zip(bufs, flat_in) if b is not t and b.data_ptr() != t.data_ptr()]
        torch._foreach_copy_([b for b, _ in pairs], [t for _, t in pairs])    # multi-tensor launches for the ~30 input tensors
        g.replay()

    # ---- one game step of every env (run() :198-705) ----
    def game_step_all(self, train: bool = True, explore: bool = True, train_iters: int = 1, update: bool = True):
        """train: push accepted transitions to the replay and (with `update`) run `train_iters` MADDPG updates from it;
        update=False leaves the updates to the caller (MixedMARL: one set of agents over several size classes)."""
        B, P, Gm = self.B, self.P, self.Gm
        pts0, n0 = self.pts.clone(), self.n.clone()                   # front_no / Pf_HV of this step (:203-209)
        y0, sec0 = self.arch_y.clone(), self.arch_sec.clone()
        added = 0
        rsum = torch.zeros((B, 3), dtype=torch.float64, device=self.device)
        tk = time.perf_counter()
        dev = self.device
        N, E = self.topo.N, self.topo.E
        arP = self._const("arP", lambda: torch.arange(P, device=dev))
        n_max = int(n0.max().item())
        for g0 in range(0, n_max, Gm):                                # member groups whose candidates fit one cull
            # live (env, member) pairs of this group, member-major: pair k = (env pb[k], member pm[k])
            livemask = (arP[None, g0:g0 + Gm] < n0[:, None])          # [B, Gm]
            pm_all, pb_all = torch.nonzero(livemask.t(), as_tuple=True)
            pm_all = pm_all + g0
            for c0 in range(0, int(pb_all.numel()), self.cap):
                tk = self._tick("other", tk)
                idx, ms = pb_all[c0:c0 + self.cap], pm_all[c0:c0 + self.cap]
                K = int(idx.numel())
                ark = torch.arange(K, device=dev)
                p0, nn0 = pts0[idx].contiguous(), n0[idx].contiguous()
                py, ps = y0[idx, ms], sec0[idx, ms]
                cx, ct, cp = self.c_x[idx], self.c_target[idx], self.c_params[idx]
                # parents: the members of the step-start archive (:211-221)
                eP, eC = self.envP, self.envC
                eP.x[:K], eP.target[:K], eP.env_params[:K] = cx, ct, cp
                eP.y[:K], eP.sec[:K] = py, ps
                S = self._obs(eP, p0, nn0, ms, k=K, o=eP.analyze(n_active=K, obs=True))   # analysis + observations: one launch
                tk = self._tick("parent analysis + obs", tk)
                geo, topo = self._act(S, explore)
                tk = self._tick("actors", tk)
                # the three agents modify the SAME parent (:249-260): one launch over 3 K envs, agent-major
                eC.x[:3 * K], eC.target[:3 * K], eC.env_params[:3 * K] = cx.repeat(3, 1), ct.repeat(3, 1), cp.repeat(3, 1)
                eC.y[:3 * K], eC.sec[:3 * K] = py.repeat(3, 1), ps.repeat(3, 1)
                a_geo, a_topo = torch.cat(geo, 0).contiguous(), torch.cat(topo, 0).contiguous()
                oC = eC.step(a_geo, a_topo, clamp_inplace=True, n_active=3 * K, obs=True)   # clamped actions go to the replay (:375);
                self._steps_dev += 3 * K                                                     # step + next-state observations: one launch
                NSall = self._obs(eC, p0, nn0, ms, rep=3, k=3 * K, o=oC, graph=(S["x_p"], S["A_p"]))
                tk = self._tick("candidate step + obs", tk)
                points = eC.point[:3 * K].view(3, K, 4).permute(1, 0, 2).double().contiguous()
                cand_y = eC.y[:3 * K].view(3, K, -1).permute(1, 0, 2)
                cand_sec = eC.sec[:3 * K].view(3, K, -1).permute(1, 0, 2)
                R, _, _, _ = RW.difference_reward(p0, nn0, p0, nn0, p0[ark, ms, :2].contiguous(), points, self.ref_points[idx].contiguous(),
                                                  nn0, max_front=P, lib=self.lib)
                rsum.index_add_(0, idx, R)
                tk = self._tick("reward", tk)
                ok = (points[:, :, 2:4] <= 1).all(dim=2)                                      # archive candidates (:372)
                # ---- archive update (D1): front of (working archive + the feasible candidates of this chunk's members) ----
                C3 = 3 * Gm
                slot = ((ms - g0) * 3)[:, None] + self._const("ar3", lambda: torch.arange(3, device=dev)[None, :])   # [K, 3] candidate slot of (pair, agent)
                candP = self._const("candP", lambda: torch.tensor([0.0, 0.0, 2.0, 0.0], dtype=torch.float64, device=dev).expand(B, C3, 4)).clone()
                #                                                                               (empty slot = infeasible row: [0, 0, 2, 0])
                candY = torch.zeros((B, C3, N), dtype=torch.float32, device=dev)
                candS = torch.zeros((B, C3, E), dtype=torch.int32, device=dev)
                pmark = points.clone()
                pmark[:, :, 2] = torch.where(ok, pmark[:, :, 2], 2.0)
                candP[idx[:, None], slot] = pmark
                candY[idx[:, None], slot] = cand_y
                candS[idx[:, None], slot] = cand_sec
                wp, wy, ws, wn = self.pts, self.arch_y, self.arch_sec, self.n
                origp = torch.cat([wp, candP], dim=1)
                allp = origp.clone()
                dead = arP[None, :] >= wn[:, None]
                allp[:, :P, 2] = torch.where(dead, 2.0, allp[:, :P, 2])                          # infeasible marker
                fr = RW.front_hv(allp, self._const("full_front", lambda: torch.full((B,), P + C3, dtype=torch.int32, device=dev)), None,
                                 max_front=P, lib=self.lib)
                fidx = fr["front_idx"][:, :P].long()
                take = fidx.clamp(min=0)
                ally = torch.cat([wy, candY], dim=1)
                alls = torch.cat([ws, candS], dim=1)
                rows = self._const("rows", lambda: torch.arange(B, device=dev)[:, None])
                live = (fidx >= 0)[:, :, None]
                newp = torch.where(live, origp[rows, take], 0.0)
                newp[:, :, 0:2].clamp_(max=1.0)                                                   # :434-436
                self.pts = newp
                self.arch_y = torch.where(live, ally[rows, take], 0.0)
                self.arch_sec = torch.where(live, alls[rows, take], 0)
                self.n = fr["n_front"].clamp(max=P)                                           # (int32 already)
                tk = self._tick("archive update", tk)
                # ---- replay (D2): one row per pair with an accepted candidate ----
                if train:
                    infront = torch.zeros((B, P + C3), dtype=torch.bool, device=dev)
                    infront.scatter_(1, take, live[:, :, 0])
                    accepted = infront[idx[:, None], P + slot] & ok                          # [K, 3]
                    first_ok = torch.argmax(ok.int(), dim=1)                                   # D4
                    # next state of agent a = its own candidate where that is feasible, else the first feasible one (D4): the rows are
                    # picked from the agent-major candidate tensors inside `add`, for the accepted pairs only
                    src = torch.where(ok, torch.arange(3, device=dev)[None, :], first_ok[:, None])   # [K, 3]
                    NSv = {k: NSall[k].view(3, K, *NSall[k].shape[1:]) for k in DeviceReplay.KEYS}
                    ag = a_geo.view(3, K, -1, 2).permute(1, 0, 2, 3)
                    at = a_topo.view(3, K, -1, 3).permute(1, 0, 2, 3)
                    added += self.replay.add(accepted.any(dim=1), S, NSv, ag, at, R.float(), src=src)
                tk = self._tick("replay", tk)
        # ---- end of the game step (:430-473, 642) ----
        hv = RW.front_hv(self.pts.contiguous(), self.n, None, 0, self.lib)
        self.ref_points = torch.clamp(self.ref_points + self.hv_margin, max=1.0)
        self.game_step += 1
        if update:
            self.train_from_replay(train_iters if train else 0)
        tk = self._tick("train", tk)
        return dict(hv=hv["hv_front"], n_front=self.n.clone(), sum_distance=hv["metrics"][:, 3], reward=rsum, replay_added=added,
                    replay_size=self.replay.size)

    def _const(self, name, make):
        """small constant device tensors of the game step (index ranges, fill patterns), made once"""
        c = self.__dict__.setdefault("_consts", {})
        if name not in c:
            c[name] = make()
        return c[name]

    def train_from_replay(self, train_iters: int = 1):
        """`train_iters` MADDPG updates on batches sampled from this engine's replay (when it holds a batch; collective
        decision under data parallelism).  Returns the number of updates run."""
        train = train_iters > 0
        ready = train and self.replay.size >= self.batch_size
        d = getattr(self.rl, "dist", None)
        if train and d is not None and d.is_initialized() and d.get_world_size() > 1:
            # data parallel: every rank plays its own envs and replay; the update is collective
            # (fused gradient all-reduce inside train_on_batch), so all ranks must agree to run it
            flag = torch.tensor([1 if ready else 0], device=self.device)
            d.all_reduce(flag, op=d.ReduceOp.MIN)
            ready = bool(flag.item())
        if ready:
            if not self._synced:
                with torch.no_grad():                                     # materialise lazy layers, then one broadcast
                    S, NS, ag, at, R = self.replay.sample(2, self.gen)
                    A = [(ag[:, a].contiguous(), at[:, a].contiguous()) for a in range(3)]
                    st = self._net_state(S)
                    self.rl._ensure_ready(st, [A[0][0], A[0][1], A[1][0], A[1][1], A[2][0], A[2][1]])
                self.rl.sync_parameters()
                self._synced = True
            for _ in range(train_iters):
                S, NS, ag, at, R = self.replay.sample(self.batch_size, self.gen)
                A = [(ag[:, a].contiguous(), at[:, a].contiguous()) for a in range(3)]
                self._train(self._net_state(S), [self._net_state(ns) for ns in NS], A, R)
            return train_iters
        return 0


class MixedMARL:
    """The batched rollout over a MIX of truss sizes with ONE set of agents (BASELINE configs[4]; the reference trains one
    MADDPG on a different truss every episode, master…:807-825; its GCN layers do not depend on the node count).

    One `BatchedMARL` engine per size class -- archives, env objects and replay have the class's shapes -- all sharing
    `maddpg`.  Envs are dealt to ranks in buckets like `MixedTrussPool` (pool.deal_buckets): every rank plays the same
    class mix.  A game step plays every class; the updates of the step then draw their batches from the classes'
    replays in turn (a batch is of one class: its tensors have that class's shapes)."""

    def __init__(self, classes, maddpg, *, bucket_envs=64, rank=0, world=1, **engine_kw):
        from .pool import deal_buckets
        self.classes = [(t, int(n)) for t, n in classes]
        self.share = deal_buckets([n for _, n in self.classes], bucket_envs, world)
        self.class_ids, self.ranges, self.engines = [], [], []
        for c, (topo, _) in enumerate(self.classes):
            n_local = sum(hi - lo for lo, hi in self.share[rank][c])
            if n_local:
                self.class_ids.append(c)
                self.ranges.append(self.share[rank][c])
                self.engines.append(BatchedMARL(topo, n_local, maddpg, **engine_kw))
        self.rl = maddpg
        self._turn = 0

    def global_ids(self, k):
        return np.concatenate([np.arange(lo, hi) for lo, hi in self.ranges[k]])

    @property
    def env_steps(self):
        return sum(e.env_steps for e in self.engines)

    def reset(self, per_class):
        """per_class[k]: dict(x, target, y_max, d_min, max_def, load_x, load_y, is_roof, y, sec) of this rank's k-th class"""
        for e, b in zip(self.engines, per_class):
            e.reset(b["x"], b["target"], b["y_max"], b["d_min"], b["max_def"], b["load_x"], b["load_y"], b["is_roof"], b["y"], b["sec"])

    def game_step_all(self, train: bool = True, explore: bool = True, train_iters: int = 1):
        outs = [e.game_step_all(train=train, explore=explore, update=False) for e in self.engines]
        done = 0
        if train:
            for _ in range(train_iters):                       # one class per update, in turn
                for _try in range(len(self.engines)):
                    e = self.engines[self._turn % len(self.engines)]
                    self._turn += 1
                    if e.train_from_replay(1):
                        done += 1
                        break
        return dict(per_class=outs, updates=done, hv=torch.cat([o["hv"] for o in outs]), n_front=torch.cat([o["n_front"] for o in outs]))
