"""Drop-in replacement for the reference's truss2D_RL.py (MADDPG with GCN actors/critics), restated in
PyTorch-ROCm (the reference is TensorFlow 2.11 / Keras / Spektral 1.2; neither is available here, so
parity is STRUCTURAL: same layers, shapes, data flow, update rules and quirks -- not bit-level).

    OUNoise                         truss2D_RL.py:41-48
    multimodes_actor                :49-127   13 GCNConv (Spektral GCNConv = A @ (X W) + b, no activation)
    multimodes_critic               :130-266  21 GCNConv + Dense(c_nn) x2 + Dense(1)
    multimodals_OneAgent            :269-404
    MADDPG                          :407-704  .agents[i].act(...), .remember(...), .train(), .update()

Quirks of the reference that are kept on purpose:
  * the Pareto-graph embedding is tiled over the nodes with stack(axis=-1) followed by a RESHAPE
    (not a transpose) to [batch, N, hidden] (:87-93);
  * `done` is unpacked from sample slot 4 (an action array) and compared with `is 1`, so the terminal
    branch of the TD target never fires (:605-611); the target averages the three next-state Qs;
  * the actor is stepped by a FRESH Adam(lr*0.1, clipnorm=1) every call (:629);
  * `OneAgent.update()` acts only when update_num == 1000, `MADDPG.update()` calls it only when
    agents[0].update_num % 300 == 0 -> the targets keep their initial hard copy (:392-400, :692-699).

Multi-GPU: pass `dist=torch.distributed` (backend "nccl" = RCCL on MI355X): every optimiser step is
preceded by ONE all-reduce of the flat gradient buffer; the env batch itself needs
no collective.
"""
import random
from collections import deque

import numpy as np
import torch
import torch.nn as nn

from set_seed_global import seedThis

random.seed(seedThis)
np.random.seed(seedThis)
torch.manual_seed(seedThis)


class OUNoise():
    def __init__(self, mu, theta, sigma):
        self.mu, self.theta, self.sigma = mu, theta, sigma
        self.dt = 0.0001

    def gen_noise(self, x):
        return self.theta * (self.mu - x) * self.dt + self.sigma * np.random.randn(1)


class GCNConv(nn.Module):
    """Spektral 1.2 GCNConv in batch mode with a dense, already normalised adjacency:
    out = A @ (X @ W) + b; kernel GlorotNormal, bias zeros, no activation."""

    def __init__(self, channels):
        super().__init__()
        self.lin = nn.LazyLinear(channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(channels))

    def forward(self, x, a, act=None):
        """act in {None, 'relu', 'sigmoid'}: the layer's activation (Spektral's `activation=`)."""
        # GlorotNormal exactly once: when this call materialises the lazy kernel.  A kernel that came in through
        # load_state_dict / the TensorFlow reader is already materialised and is left alone (a flag on the module is
        # not part of the state_dict: restored kernels were re-initialised on their first forward).
        fresh = isinstance(self.lin.weight, nn.parameter.UninitializedParameter)
        h = self.lin(x)
        if fresh:
            nn.init.xavier_normal_(self.lin.weight)
            h = self.lin(x)
        out = torch.matmul(a, h) + self.bias
        return torch.relu(out) if act == "relu" else torch.sigmoid(out) if act == "sigmoid" else out


def _tile_pool(x_pool, n_nodes):
    """[B,H] -> stack N copies on the last axis -> reshape (not transpose) to [B,N,H] (:87-93)."""
    B, H = x_pool.shape
    return x_pool.unsqueeze(-1).expand(B, H, n_nodes).reshape(B, n_nodes, H)


class multimodes_actor(nn.Module):
    def __init__(self, n_hidden, n_action1, n_action2):
        super().__init__()
        g = lambda c: GCNConv(c)
        self.gcn_l1_1, self.gcn_l1_2, self.gcn_l1_3, self.gcn_l1_4 = g(n_hidden), g(n_hidden), g(n_hidden), g(n_hidden)
        self.gcn_l2_1, self.gcn_l2_2, self.gcn_l2_3 = g(n_hidden), g(n_hidden), g(n_hidden)
        self.gcn_l2_4, self.gcn_l2_5 = g(n_hidden), g(n_hidden)
        self.gcn_l3_1, self.gcn_l3_2 = g(n_hidden), g(n_hidden)
        self.gcn_l4_1, self.gcn_l4_2 = g(n_action1), g(n_action2)

    def forward(self, inputs):
        x_n, A_n, A_s, A_n_ts, A_n_cs, x_p, A_p = inputs
        x_1_1 = self.gcn_l1_1(x_n, A_n, "relu")
        x_1_2 = self.gcn_l1_2(x_n, A_n, "relu")
        x_1_3 = self.gcn_l1_3(x_n, A_n, "relu")
        x_1_4 = self.gcn_l1_4(x_p, A_p, "relu").sum(dim=1)        # GlobalSumPool
        x_1_4 = _tile_pool(x_1_4, x_1_1.shape[1])
        x_2_1 = self.gcn_l2_1(x_1_1, A_n, "relu")
        x_2_2 = self.gcn_l2_2(x_1_2, A_n_ts, "relu")
        x_2_3 = self.gcn_l2_3(x_1_2, A_n_cs, "relu")
        x_2_4 = self.gcn_l2_4(x_1_3, A_s, "relu")
        x_2_5 = self.gcn_l2_5(x_1_4, A_n, "relu")
        x_3 = x_2_1 + x_2_2 + x_2_3 + x_2_4 + x_2_5
        x_3_1 = self.gcn_l3_1(x_3, A_n, "relu")
        x_3_2 = self.gcn_l3_2(x_3, A_s, "relu")
        out_1 = self.gcn_l4_1(x_3_1, A_n, "sigmoid")
        out_2 = self.gcn_l4_2(x_3_2, A_n, "sigmoid")
        return out_1, out_2


class multimodes_critic(nn.Module):
    def __init__(self, n_hidden, n_q):
        super().__init__()
        self.l1 = nn.ModuleList([GCNConv(n_hidden) for _ in range(10)])
        self.l2 = nn.ModuleList([GCNConv(n_hidden) for _ in range(11)])
        self.dense_1 = nn.LazyLinear(n_q)
        self.dense_2 = nn.Linear(n_q, n_q)
        self.dense_out = nn.Linear(n_q, 1)
        nn.init.xavier_normal_(self.dense_2.weight)

    def forward(self, inputs):
        x_n, A_n, A_s, A_n_ts, A_n_cs, mask, x_p, A_p, self_g, self_t, other_g1, other_t1, other_g2, other_t2 = inputs
        relu = torch.relu
        x_1_1 = self.l1[0](x_n, A_n, "relu")
        x_1_2 = self.l1[1](x_n, A_n, "relu")
        x_1_3 = self.l1[2](x_n, A_n, "relu")
        x_1_4 = _tile_pool(self.l1[3](x_p, A_p, "relu").sum(dim=1), x_1_1.shape[1])
        acts = [self_g, self_t, other_g1, other_t1, other_g2, other_t2]
        x_1_a = [self.l1[4 + i](a, A_n, "relu") for i, a in enumerate(acts)]
        x2 = [self.l2[0](x_1_1, A_n, "relu"), self.l2[1](x_1_2, A_n_ts, "relu"), self.l2[2](x_1_2, A_n_cs, "relu"),
              self.l2[3](x_1_3, A_s, "relu")]
        x2 += [self.l2[4 + i](h, A_n, "relu") for i, h in enumerate(x_1_a)]
        x2.append(self.l2[10](x_1_4, A_n, "relu"))
        # 11 x GlobalSumPool -> Concatenate, as one reduction: [B, N, 11, C] summed over the nodes = the 11 pooled vectors side by side
        q = torch.stack(x2, dim=2).sum(dim=1).flatten(1)
        q = relu(self.dense_1(q))
        q = relu(self.dense_2(q))
        return self.dense_out(q)


# ---- level-wise evaluation of the networks for the MADDPG update ------------------------------------------------------------------
# At batch 32 a GCN layer is four small launches forward (X W, A H, + b, activation) and about seven backward, a critic has 21 of
# them although it is only two levels deep, and an update passes through nineteen networks: it is bound by its kernel count (2 312
# launches, 9.5 ms as a replayed hipGraph, when evaluated layer by layer).  The layers of one level do not depend on each other --
# nor do the same levels of DIFFERENT networks (the three target actors, the three critics ...) -- so a level is evaluated as ONE
# operation over all of them (`gcn_level`, an autograd Function with a hand-written backward):
#   forward   X'_i = A_i X_i, out_i = act(X'_i W_i^T + b_i)      one fused launch for the whole level on the GPU (truss_gcn_level,
#             csrc/truss_gcn.h: MFMA, every layer of the level a slice of the grid) -- or, per group of equally shaped layers, two
#             batched GEMMs over stacked operands (CPU, and the reference for the kernel);
#   backward  dZ = dOut * act'(out);  db = sum dZ;  dW = dZ^T X'  (one batched GEMM per group: X' was kept);
#             dX = A^T (dZ W)  (two batched GEMMs, only where an input asks for it).
# The networks are written as generators that yield one level's requests at a time (`_actor_steps`, `_critic_steps`);
# `run_networks` steps several of them in lockstep and hands each level of all of them to `gcn_level` together.  Same parameters
# (the modules' own tensors: state_dict, checkpoints and the per-layer `forward` are untouched), same mathematics in a different
# summation order (tests/test_master_rl.py::test_grouped_forward_matches_layerwise: outputs <= 1e-5, gradients <= 1e-4 of their
# scale).

_LEVEL_FORWARD = {}        # device type -> callable(groups, xs, ws, bs, want_grad) -> ([out_g], [X'_g]) or None (truss_mi355.marl.level_forward)


def set_level_forward(fn, device_type="cuda"):
    """Install (or, with None, remove) the fused level-forward for tensors of `device_type`; see `_GcnLevel.forward` for its contract."""
    if fn is None:
        _LEVEL_FORWARD.pop(device_type, None)
    else:
        _LEVEL_FORWARD[device_type] = fn


class _Group:
    """equally shaped layers of a level: requests `idx`, activation, stacked adjacencies [n B, N, N]"""
    __slots__ = ("idx", "act", "a", "adjs", "shape")

    def __init__(self, idx, act, a, adjs, shape):
        self.idx, self.act, self.a, self.adjs, self.shape = idx, act, a, adjs, shape


def _adj_stack(adjs, B, N, cache=None):
    """the adjacencies of a group, stacked for the batched GEMMs: [n * B, N, N] (data only: cached for the minibatch)"""
    key = (B,) + tuple(id(a_) for a_ in adjs)
    hit = None if cache is None else cache.get(key)
    if hit is not None:
        return hit[0]
    a = torch.stack([a_.expand(B, N, N) for a_ in adjs]).reshape(len(adjs) * B, N, N)
    if cache is not None:
        cache[key] = (a, tuple(adjs))                          # (the tensors are kept: their ids stay unique while the cache lives)
    return a


class _GcnLevel(torch.autograd.Function):
    """apply((groups, want_grad), *xs, *ws, *bs) -> one tensor [n_g, B, N, C] per group (see the section comment)."""

    @staticmethod
    def forward(ctx, plan, *tensors):
        groups, want_grad = plan           # want_grad: decided by the caller (inside forward the grad mode is always off)
        L = len(tensors) // 3
        xs, ws, bs = tensors[:L], tensors[L:2 * L], tensors[2 * L:]
        outs = xaggs = None
        fused = _LEVEL_FORWARD.get(xs[0].device.type)
        if fused is not None:
            res = fused(groups, xs, ws, bs, want_grad)
            if res is not None:
                outs, xaggs = res
        if outs is None:
            outs, xaggs = [], []
            for g in groups:
                n, (B, N, K), C = len(g.idx), g.shape, ws[g.idx[0]].shape[0]
                x0 = xs[g.idx[0]]
                xst = x0.expand(n, B, N, K) if all(xs[i] is x0 for i in g.idx) else torch.stack([xs[i] for i in g.idx])
                xa = torch.bmm(g.a, xst.reshape(n * B, N, K)).view(n, B * N, K)                    # X' = A X
                wst = torch.stack([ws[i] for i in g.idx])                                          # [n, C, K]
                bst = torch.stack([bs[i] for i in g.idx])[:, None, :]                              # [n, 1, C]
                o = torch.baddbmm(bst, xa, wst.transpose(1, 2))
                o = torch.relu_(o) if g.act == "relu" else torch.sigmoid_(o) if g.act == "sigmoid" else o
                outs.append(o.view(n, B, N, C))
                xaggs.append(xa)
        ctx.groups, ctx.L = groups, L
        ctx.set_materialize_grads(False)       # a group nothing flows back into arrives as None in backward, not as a tensor of zeros
        if want_grad:
            ctx.save_for_backward(*outs, *xaggs, *ws)
            # a group none of whose inputs asks for a gradient (frozen networks evaluated alongside, layers on replayed inputs whose
            # kernels are not trained in this pass): nothing flows back through it, nothing downstream stacks gradients for it
            need = ctx.needs_input_grad
            frozen = [o for g, o in zip(groups, outs) if not any(need[1 + i] or need[1 + L + i] or need[1 + 2 * L + i] for i in g.idx)]
            if frozen:
                ctx.mark_non_differentiable(*frozen)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        groups, L = ctx.groups, ctx.L
        G = len(groups)
        saved = ctx.saved_tensors
        outs, xaggs, ws = saved[:G], saved[G:2 * G], saved[2 * G:]
        need = ctx.needs_input_grad
        dx, dw, db = [None] * L, [None] * L, [None] * L
        for gi, g in enumerate(groups):
            need_w = any(need[1 + L + i] or need[1 + 2 * L + i] for i in g.idx)
            need_x = any(need[1 + i] for i in g.idx)
            if douts[gi] is None or not (need_w or need_x):
                continue
            n, (B, N, K) = len(g.idx), g.shape
            o = outs[gi]
            C = o.shape[-1]
            d = douts[gi].contiguous()
            dz = (torch.ops.aten.threshold_backward(d, o, 0.0) if g.act == "relu" else
                  torch.ops.aten.sigmoid_backward(d, o) if g.act == "sigmoid" else d).view(n, B * N, C)
            if need_w:
                for i, piece in zip(g.idx, torch.bmm(dz.transpose(1, 2), xaggs[gi]).unbind(0)):     # dW = dZ^T X'   [n, C, K]
                    dw[i] = piece
                for i, piece in zip(g.idx, dz.sum(dim=1).unbind(0)):
                    db[i] = piece
            if need_x:
                gw = torch.bmm(dz, torch.stack([ws[i] for i in g.idx]))                             # dZ W           [n, B N, K]
                gx = torch.bmm(g.a.transpose(1, 2), gw.view(n * B, N, K)).view(n, B, N, K)         # A^T (dZ W)
                for i, piece in zip(g.idx, gx.unbind(0)):
                    dx[i] = piece                          # (one tensor passed for several layers: autograd adds per argument)
        return (None, *dx, *dw, *db)


def gcn_level(reqs, cache=None):
    """reqs: [(GCNConv layer, x [B,N,K], adjacency [B,N,N] or [1,N,N], activation)] -> [act(adj @ x @ W^T + b)] in request order.
    Layers of equal shapes and activation form a group (one set of batched GEMMs forward and backward)."""
    by_key = {}
    live = torch.is_grad_enabled()
    for i, (layer, x, adj, act) in enumerate(reqs):
        trains = live and (x.requires_grad or layer.lin.weight.requires_grad)      # frozen passes evaluated alongside: groups of their own
        by_key.setdefault((tuple(x.shape), layer.lin.out_features, act, trains), []).append(i)
    groups = []
    for (shape, _, act, _), idx in by_key.items():
        adjs = [reqs[i][2] for i in idx]
        groups.append(_Group(idx, act, _adj_stack(adjs, shape[0], shape[1], cache), adjs, shape))
    tensors = [r[1] for r in reqs] + [r[0].lin.weight for r in reqs] + [r[0].bias for r in reqs]
    want_grad = torch.is_grad_enabled() and any(t.requires_grad for t in tensors)
    outs = _GcnLevel.apply((groups, want_grad), *tensors)
    res = [None] * len(reqs)
    for g, o in zip(groups, outs):
        # unbind, not o[i]: its backward is ONE stack of the pieces' gradients; every select's backward is a fill, a copy and an add
        # (a single layer: a view, whose backward is a view)
        for i, piece in zip(g.idx, (o.squeeze(0),) if len(g.idx) == 1 else o.unbind(0)):
            res[i] = piece
    return res


class _Frozen:
    """a GCNConv seen through detached parameters: a network that is evaluated WITHOUT gradients in the same `run_networks` call as
    one that is trained (its layers form groups of their own, which the level operation marks non-differentiable)"""

    class _Lin:
        __slots__ = ("weight", "out_features")

    __slots__ = ("lin", "bias")

    def __init__(self, layer):
        self.lin = _Frozen._Lin()
        self.lin.weight, self.lin.out_features = layer.lin.weight.detach(), layer.lin.out_features
        self.bias = layer.bias.detach()


def _ready(layers):
    return not any(isinstance(L.lin.weight, nn.parameter.UninitializedParameter) for L in layers)


def _actor_steps(a, inputs, frozen=False):
    """multimodes_actor.forward as a generator: yields the requests of one level, receives that level's outputs.
    frozen: evaluate through detached parameters (no gradient is wanted from this pass)"""
    x_n, A_n, A_s, A_n_ts, A_n_cs, x_p, A_p = inputs
    f = _Frozen if frozen else (lambda layer: layer)
    o = yield [(f(a.gcn_l1_1), x_n, A_n, "relu"), (f(a.gcn_l1_2), x_n, A_n, "relu"), (f(a.gcn_l1_3), x_n, A_n, "relu"),
               (f(a.gcn_l1_4), x_p, A_p, "relu")]
    x_1_4 = _tile_pool(o[3].sum(dim=1), x_n.shape[1])                                                # GlobalSumPool, tiled
    o = yield [(f(a.gcn_l2_1), o[0], A_n, "relu"), (f(a.gcn_l2_2), o[1], A_n_ts, "relu"), (f(a.gcn_l2_3), o[1], A_n_cs, "relu"),
               (f(a.gcn_l2_4), o[2], A_s, "relu"), (f(a.gcn_l2_5), x_1_4, A_n, "relu")]
    x_3 = torch.stack(o).sum(dim=0)
    o = yield [(f(a.gcn_l3_1), x_3, A_n, "relu"), (f(a.gcn_l3_2), x_3, A_s, "relu")]
    o = yield [(f(a.gcn_l4_1), o[0], A_n, "sigmoid"), (f(a.gcn_l4_2), o[1], A_n, "sigmoid")]
    return o[0], o[1]


def _critic_steps(c, inputs, frozen=False):
    """multimodes_critic.forward as a generator (see _actor_steps).  frozen: the critic's own parameters are not trained by this pass
    (the actor update differentiates THROUGH the critic, with respect to the action inputs only): its layers are seen through
    detached parameters, so that only the layers on the path from those inputs take part in the backward pass."""
    x_n, A_n, A_s, A_n_ts, A_n_cs, mask, x_p, A_p, self_g, self_t, other_g1, other_t1, other_g2, other_t2 = inputs
    f = _Frozen if frozen else (lambda layer: layer)
    acts = [self_g, self_t, other_g1, other_t1, other_g2, other_t2]
    o = yield ([(f(c.l1[i]), x_n, A_n, "relu") for i in range(3)] + [(f(c.l1[3]), x_p, A_p, "relu")] +
               [(f(c.l1[4 + i]), a_, A_n, "relu") for i, a_ in enumerate(acts)])
    x_1_4 = _tile_pool(o[3].sum(dim=1), x_n.shape[1])
    xs = [o[0], o[1], o[1], o[2]] + list(o[4:10]) + [x_1_4]
    adjs = [A_n, A_n_ts, A_n_cs, A_s] + [A_n] * 7
    o = yield [(f(c.l2[i]), xs[i], adjs[i], "relu") for i in range(11)]
    # 11 x GlobalSumPool -> Concatenate, as one reduction: [B, 11, N, C] summed over the nodes = the 11 pooled vectors side by side
    q = torch.stack(o, dim=1).sum(dim=2).flatten(1)
    dense = (lambda m, t: nn.functional.linear(t, m.weight.detach(), m.bias.detach())) if frozen else (lambda m, t: m(t))
    q = torch.relu(dense(c.dense_1, q))
    q = torch.relu(dense(c.dense_2, q))
    return dense(c.dense_out, q)


def run_networks(gens, cache=None):
    """Step the network generators in lockstep: level k of all of them is ONE `gcn_level`.  Returns their return values."""
    reqs = [next(g) for g in gens]
    results = [None] * len(gens)
    live = list(range(len(gens)))
    while live:
        outs = gcn_level([r for k in live for r in reqs[k]], cache)
        pos, still = 0, []
        for k in live:
            o = outs[pos:pos + len(reqs[k])]
            pos += len(reqs[k])
            try:
                reqs[k] = gens[k].send(o)
                still.append(k)
            except StopIteration as done:
                results[k] = done.value
        live = still
    return results


def _actor_layers(a):
    return [a.gcn_l1_1, a.gcn_l1_2, a.gcn_l1_3, a.gcn_l1_4, a.gcn_l2_1, a.gcn_l2_2, a.gcn_l2_3, a.gcn_l2_4, a.gcn_l2_5, a.gcn_l3_1,
            a.gcn_l3_2, a.gcn_l4_1, a.gcn_l4_2]


def _critic_is_ready(c):
    return _ready(list(c.l1) + list(c.l2)) and not isinstance(c.dense_1.weight, nn.parameter.UninitializedParameter)


def actor_forward_grouped(actor, inputs, cache=None):
    """multimodes_actor.forward, level by level (see above)."""
    if not _ready(_actor_layers(actor)):
        return actor(inputs)                               # first call: let the lazy kernels materialise (Glorot) layer by layer
    return run_networks([_actor_steps(actor, inputs)], cache)[0]


def critic_forward_grouped(critic, inputs, cache=None):
    """multimodes_critic.forward, level by level (see above)."""
    if not _critic_is_ready(critic):
        return critic(inputs)
    return run_networks([_critic_steps(critic, inputs)], cache)[0]


_seg_cache: dict = {}


def _clip_each(params, max_norm=1.0):
    """Keras `clipnorm`: every gradient tensor is clipped to L2 norm <= clipnorm on its own (in place, on p.grad; the update itself
    clips the flat gradient vector: `_clip_flat`)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    # branch-free (no host synchronisation per tensor): the factor is exactly 1.0 where the norm is within bounds.
    # A handful of launches whatever the number of tensors: the norms by a multi-tensor kernel, the per-tensor factors as ONE
    # vector, applied to a flat copy of the gradients through a (cached) element -> tensor index and copied back.  Multiplying a
    # tensor list by a list of 0-dim factors falls off PyTorch's multi-tensor path: one launch per parameter tensor.
    fac = torch.stack(torch._foreach_norm(grads)).add_(1e-12).reciprocal_()
    if max_norm != 1.0:
        fac.mul_(max_norm)
    fac.clamp_(max=1.0)
    key = (tuple(g.numel() for g in grads), grads[0].device)
    seg = _seg_cache.get(key)
    if seg is None:
        seg = torch.repeat_interleave(torch.arange(len(grads), device=grads[0].device),
                                      torch.tensor([g.numel() for g in grads], device=grads[0].device))
        _seg_cache[key] = seg
    flat = torch.cat([g.reshape(-1) for g in grads])
    flat.mul_(fac[seg])
    torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])


def _flat_grads(params):
    """the gradients of `params` as ONE contiguous vector (the all-reduce buffer, and what the flat clip / Adam below work on)"""
    return torch.cat([p.grad.reshape(-1) for p in params])


def _clip_flat(flat, params, max_norm=1.0):
    """`_clip_each` on the flat gradient vector of `params` (in place): every tensor's slice scaled to L2 norm <= max_norm."""
    sizes = [p.numel() for p in params]
    fac = torch.stack(torch._foreach_norm(list(flat.split(sizes)))).add_(1e-12).reciprocal_()
    if max_norm != 1.0:
        fac.mul_(max_norm)
    fac.clamp_(max=1.0)
    key = (tuple(sizes), flat.device)
    seg = _seg_cache.get(key)
    if seg is None:
        seg = torch.repeat_interleave(torch.arange(len(sizes), device=flat.device), torch.tensor(sizes, device=flat.device))
        _seg_cache[key] = seg
    return flat.mul_(fac[seg])


class SharedStepAdam:
    """`torch.optim.Adam(params, lr, betas, eps)` for a parameter list that always steps together (the critics of a MADDPG): ONE step
    counter on the device and the moments as two FLAT vectors, so that a step is a dozen element-wise launches whatever the number
    of parameter tensors (144 for three critics), usable inside a hipGraph.  PyTorch's capturable Adam keeps a step tensor per
    parameter and runs multi-tensor kernels of at most a few dozen tensors each: 35 launches for the same step.  Same arithmetic:
    m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)."""

    def __init__(self, params, lr, eps=1e-8, betas=(0.9, 0.999)):
        self.params = list(params)
        self.lr, self.eps, (self.b1, self.b2) = float(lr), float(eps), betas
        self.step_t, self.exp_avg, self.exp_avg_sq = None, None, None      # 0-dim float64; flat float32 vectors

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def state_tensors(self):
        """the optimiser's state as a flat list (empty before the first step): step counter, first moments, second moments"""
        return [] if self.step_t is None else [self.step_t, self.exp_avg, self.exp_avg_sq]

    @torch.no_grad()
    def step(self, flat_grad=None):
        """flat_grad: the (clipped) gradient of all parameters as one vector (`_flat_grads`); default: built from p.grad"""
        ps = self.params
        if flat_grad is None:
            if any(p.grad is None for p in ps):
                raise RuntimeError("SharedStepAdam: every parameter must have a gradient (the list steps together)")
            flat_grad = _flat_grads(ps)
        if self.step_t is None:
            self.step_t = torch.zeros((), dtype=torch.float64, device=ps[0].device)   # float64: 1 - 0.999^t cancels badly in float32
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(flat_grad), torch.zeros_like(flat_grad)
        m, v, g = self.exp_avg, self.exp_avg_sq, flat_grad
        self.step_t += 1
        bc1 = 1 - torch.pow(self.b1, self.step_t)
        bc2_sqrt = (1 - torch.pow(self.b2, self.step_t)).sqrt_().float()
        m.lerp_(g, 1 - self.b1)
        v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
        upd = v.sqrt().div_(bc2_sqrt).add_(self.eps).reciprocal_().mul_(m).mul_((-self.lr / bc1).float())
        torch._foreach_add_(ps, [u.view_as(p) for u, p in zip(upd.split([p.numel() for p in ps]), ps)])


def _fresh_adam_step(params, lr, eps, flat_grad=None):
    """`Adam(params, lr, eps=eps).step()` of an optimiser that is created for this one call (the reference builds a new Adam for
    every actor update, truss2D_RL.py:629/658/688): with zero moments and step = 1 the bias-corrected moments are g and g^2, so
    the step is p -= lr * g / (|g| + eps).  A few launches instead of the state allocation (three fills per parameter tensor) and
    the general update.  flat_grad: the (clipped) gradients of `params` as one vector; default: p.grad of every parameter."""
    with torch.no_grad():
        if flat_grad is not None:
            upd = flat_grad / flat_grad.abs().add_(eps)
            torch._foreach_add_(list(params), [u.view_as(p) for u, p in zip(upd.split([p.numel() for p in params]), params)], alpha=-lr)
            return
        ps = [p for p in params if p.grad is not None]
        if not ps:
            return
        grads = [p.grad for p in ps]
        den = torch._foreach_abs(grads)
        torch._foreach_add_(den, eps)
        upd = torch._foreach_div(grads, den)
        torch._foreach_add_(ps, upd, alpha=-lr)


def _allreduce_flat(flat, dist):
    """all-reduce (mean over ranks) of an already flat gradient vector, in place"""
    if dist is None or not dist.is_initialized():
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    world = dist.get_world_size()
    return flat.mul_(1.0 / world) if world > 1 else flat


class multimodals_OneAgent:
    def __init__(self, lr, ep, epd, gamma, a_nn, c_nn, num_action1, num_action2, mu_s, theta_s, sigma_s, mu_t, theta_t,
                 sigma_t, all_agent, batch, device="cpu"):
        self.number = 1
        self.lr, self.epmin, self.gamma = lr, 0.05, gamma
        self.a_nn, self.c_nn = a_nn, c_nn
        self.num_action1, self.num_action2 = num_action1, num_action2
        self.tau = 0.005
        self.noise_geo = [OUNoise(mu_s[i], theta_s[i], sigma_s[i]) for i in range(num_action1)]
        self.noise_topo = [OUNoise(mu_t[i], theta_t[i], sigma_t[i]) for i in range(num_action2)]
        self.update_num = 0
        self.update_lr = 0
        self.c_loss = []
        self.device = torch.device(device)
        self.actor_model = multimodes_actor(a_nn, num_action1, num_action2).to(self.device)
        self.target_actor_model = multimodes_actor(a_nn, num_action1, num_action2).to(self.device)
        self.critic_model = multimodes_critic(a_nn, c_nn).to(self.device)
        self.target_critic_model = multimodes_critic(a_nn, c_nn).to(self.device)
        self.critic_opt = None       # created once the lazy layers are materialised
        self.all_agent = all_agent
        self.batch_size = batch
        self._targets_ready = False

    def _t(self, a):
        return torch.as_tensor(np.asarray(a), dtype=torch.float32, device=self.device)

    def act(self, x_n, A_n, A_s, A_n_ts, A_n_cs, x_p, A_p):
        ins = [self._t(v).unsqueeze(0) for v in (x_n, A_n, A_s, A_n_ts, A_n_cs, x_p, A_p)]
        with torch.no_grad():
            out_geo, out_topo = self.actor_model(ins)
        action_geo = out_geo[0].cpu().numpy().astype(np.float32)
        for i in range(len(action_geo)):
            for j in range(len(action_geo[i])):
                action_geo[i][j] += self.noise_geo[j].gen_noise(action_geo[i][j])[0]
        action_topo = out_topo[0].cpu().numpy().astype(np.float32)
        for i in range(len(action_topo)):
            for j in range(len(action_topo[i])):
                action_topo[i][j] += self.noise_topo[j].gen_noise(action_topo[i][j])[0]
        self.update_num += 1
        return action_geo, action_topo

    @staticmethod
    def _copy(src, dst, tau=None):
        with torch.no_grad():
            for ps, pd in zip(src.parameters(), dst.parameters()):
                if isinstance(ps, nn.parameter.UninitializedParameter) or isinstance(pd, nn.parameter.UninitializedParameter):
                    return False
                pd.copy_(ps if tau is None else ps * tau + pd * (1 - tau))
        return True

    def _update_actor_target(self, init=None):
        return self._copy(self.actor_model, self.target_actor_model, None if init == 1 else self.tau)

    def _update_critic_target(self, init=None):
        return self._copy(self.critic_model, self.target_critic_model, None if init == 1 else self.tau)

    def update(self):
        if self.update_num == 1000:
            self._update_actor_target()
            self._update_critic_target()
            self.update_num = 0
            print('update target')

    def update_init(self):
        a = self._update_actor_target(1)
        c = self._update_critic_target(1)
        self._targets_ready = bool(a and c)


class MADDPG:
    def __init__(self, lr, ep, epd, gamma, a_nn, c_nn, max_mem, num_agents, num_action, mu, theta, sigma,
                 max_poss_n_num=1, device="cpu", dist=None):
        self.num_agents = num_agents
        self.lr, self.epint, self.ep, self.epd, self.epmin, self.gamma = lr, ep, ep, epd, 0.05, gamma
        self.a_nn, self.c_nn = a_nn, c_nn
        self.mu, self.theta, self.sigma = mu, theta, sigma
        self.temprp = deque(maxlen=max_mem)
        for _ in range(max_poss_n_num):
            self.temprp.append(deque(maxlen=max_mem))
        self.agents, self.update_counter = [], []
        self.num_state = [0, 0]
        self.num_action = num_action
        self.batch_size = 32
        self.max_poss_n_num = max_poss_n_num
        self.device = torch.device(device)
        self.dist = dist
        self.critics_opt = None      # one SharedStepAdam over the three critics, created once the lazy layers are materialised
        self.gen_agents()

    def gen_agents(self):
        for i in range(3):       # the reference builds exactly three agents (:438-452)
            ag = multimodals_OneAgent(self.lr, self.ep, self.epd, self.gamma, self.a_nn, self.c_nn, self.num_action[0],
                                      self.num_action[1], self.mu[0], self.theta[0], self.sigma[0], self.mu[1],
                                      self.theta[1], self.sigma[1], self.num_agents, self.batch_size, device=self.device)
            ag.number = i + 1
            self.agents.append(ag)
            self.update_counter.append(0)

    def remember(self, state, a0_g, a0_t, a1_g, a1_t, a2_g, a2_t, reward, next_state1, next_state2, next_state3, done, n_node):
        self.temprp[0].append([state, a0_g, a0_t, a1_g, a1_t, a2_g, a2_t, reward, next_state1, next_state2, next_state3, done])

    def _stack(self, states, idx):
        return torch.as_tensor(np.array([s[idx] for s in states]), dtype=torch.float32, device=self.device)

    def _state_tensors(self, states):
        return [self._stack(states, i) for i in range(8)]      # x_n, A_n, A_s, A_n_ts, A_n_cs, mask, x_p, A_p

    @staticmethod
    def _actor_in(s):
        return [s[0], s[1], s[2], s[3], s[4], s[6], s[7]]

    def _ensure_ready(self, S, acts):
        """materialise lazy layers, take the initial hard copy of the targets (update_init, :402-404)"""
        for i, ag in enumerate(self.agents):
            if not ag._targets_ready:
                with torch.no_grad():
                    ag.actor_model(self._actor_in(S))
                    ag.target_actor_model(self._actor_in(S))
                    ag.critic_model(S + acts)
                    ag.target_critic_model(S + acts)
                ag.update_init()
        if self.critics_opt is None:
            # ONE optimiser for the three critics: they are created together and always step together
            assert len({ag.lr for ag in self.agents}) == 1
            self.critics_opt = SharedStepAdam([p for ag in self.agents for p in ag.critic_model.parameters()], lr=self.agents[0].lr, eps=1e-7)
            for ag in self.agents:
                ag.critic_opt = self.critics_opt

    def train(self):
        batch_size = self.batch_size
        if len(self.temprp[0]) < batch_size:
            return
        samples = random.sample(self.temprp[0], batch_size)
        t = lambda k: torch.as_tensor(np.array([v[k] for v in samples]), dtype=torch.float32, device=self.device)
        S = self._state_tensors([v[0] for v in samples])
        NS = [self._state_tensors([v[8 + f] for v in samples]) for f in range(3)]
        A = [(t(1), t(2)), (t(3), t(4)), (t(5), t(6))]
        R = torch.as_tensor(np.array([v[7] for v in samples]), dtype=torch.float32, device=self.device)   # [B,3]
        self.train_on_batch(S, NS, A, R)

    def train_on_batch(self, S, NS, A, R):
        """One MADDPG update (truss2D_RL.py:536-689) on a prepared minibatch of device tensors:
        S = [x_n, A_n, A_s, A_n_ts, A_n_cs, mask, x_p, A_p] (batch first), NS = the same per agent's next state
        (three lists), A = [(a_geo, a_topo)] x 3, R [batch, 3].  `train()` feeds it from the host replay of
        the reference loop; the batched rollout (truss_mi355/marl.py) from its device replay."""
        flat = lambda order: [A[order[0]][0], A[order[0]][1], A[order[1]][0], A[order[1]][1], A[order[2]][0], A[order[2]][1]]
        orders = [(0, 1, 2), (1, 0, 2), (2, 0, 1)]          # (self, other1, other2) per agent (:561-563)
        self._ensure_ready(S, flat(orders[0]))
        # Passes that do not depend on each other go through `run_networks` TOGETHER: one operation per level for all of them (on the
        # GPU one launch), including passes WITHOUT gradients next to one that is trained (`frozen`).
        agents = self.agents
        pair = lambda preds, o: [preds[o[0]][0], preds[o[0]][1], preds[o[1]][0], preds[o[1]][1], preds[o[2]][0], preds[o[2]][1]]
        cn, cs = {}, {}                  # per-minibatch caches of the stacked adjacencies (data only)
        with torch.no_grad():
            # the three next states (one per agent's move, :561-600) go through the target networks as ONE batch of 3 x batch
            # samples, and the three target actors -- then the three target critics -- level by level together
            nb = NS[0][0].shape[0]
            def three(k):         # a broadcast constant (the shared adjacency A_n, the mask) stays one; the rest is concatenated
                t = NS[0][k]
                if t.stride(0) == 0 and all(NS[f][k].stride(0) == 0 and NS[f][k].data_ptr() == t.data_ptr() for f in range(3)):
                    return t[:1].expand(3 * nb, *t.shape[1:])
                return torch.cat([NS[f][k] for f in range(3)], dim=0)
            NSc = [three(k) for k in range(len(NS[0]))]
            na = run_networks([_actor_steps(ag.target_actor_model, self._actor_in(NSc)) for ag in agents], cn)
            tq = run_networks([_critic_steps(ag.target_critic_model, NSc + pair(na, orders[i])) for i, ag in enumerate(agents)], cn)
            # TD targets; `done` never fires in the reference (it compares an action array with `is 1`)
            tq3 = torch.stack(tq, dim=1).view(3, nb, 3)                                   # [next state f, sample, agent]
            y = R + self.gamma * tq3.sum(dim=0) / 3                                        # [sample, agent]
        # The three critic updates depend on nothing another network's update changes (replayed actions, target networks): they
        # come first -- exactly the reference's results, since critic i is not touched between its own step and agent i's actor
        # update (:603-629) -- together, level by level; data-parallel, their gradients travel as ONE flat buffer (one RCCL
        # all-reduce for the three).
        qs = run_networks([_critic_steps(ag.critic_model, S + flat(orders[i])) for i, ag in enumerate(agents)], cs)
        losses = ((torch.cat(qs, dim=1) - y) ** 2).mean(dim=0)                            # [agent]
        cps = [list(ag.critic_model.parameters()) for ag in agents]
        # autograd.grad, not backward(): the kernels' gradients are slices of their level's stacked gradient, and backward() would
        # clone every one of them into .grad
        grads = torch.autograd.grad(losses.sum(), [p for cp in cps for p in cp])
        k = 0
        for cp in cps:
            for p in cp:
                p.grad = grads[k]
                k += 1
        for i, ag in enumerate(agents):
            ag.c_loss.append(losses[i].detach())     # stays on the device: no synchronisation inside the update
        allp = [p for cp in cps for p in cp]
        flat = _allreduce_flat(_flat_grads(allp), self.dist)         # (p.grad keeps the local, unclipped gradient)
        self.critics_opt.step(_clip_flat(flat, allp))
        # The actor updates stay one after the other: agent i's loss re-evaluates ALL three actors (:617-629), i.e. it sees the
        # weights agents < i have just stepped to -- one collective per actor.  (The other two actors' passes carry no gradient
        # that is asked for -- autograd.grad is taken with respect to agent i's parameters only -- so they run without a graph.)
        frozen = {}          # no-gradient evaluations that are still valid: actor j's weights only change in iteration j
        for i, ag in enumerate(agents):
            o = orders[i]
            todo = [j for j in range(len(agents)) if j != i and j not in frozen]
            res = run_networks([_actor_steps(ag.actor_model, self._actor_in(S))] +
                               [_actor_steps(agents[j].actor_model, self._actor_in(S), frozen=True) for j in todo], cs)
            frozen.update(zip(todo, res[1:]))
            preds = [res[0] if j == i else frozen[j] for j in range(len(agents))]
            q = run_networks([_critic_steps(ag.critic_model, S + pair(preds, o), frozen=True)], cs)[0]
            actor_loss = -q.mean()
            ap = list(ag.actor_model.parameters())
            for p in ap:
                p.grad = None
            grads = torch.autograd.grad(actor_loss, ap)
            for p, g in zip(ap, grads):
                p.grad = g
            flat = _clip_flat(_allreduce_flat(_flat_grads(ap), self.dist), ap)
            _fresh_adam_step(ap, ag.lr * 0.1, 1e-7, flat)                                # a fresh optimiser every call (:629)
            # agent i's weights have just changed: an evaluation of actor i taken before this step must not be reused (the others'
            # stay valid: agents > i have not stepped yet, evaluations of agents < i were taken after their steps)
            frozen = {j: v for j, v in frozen.items() if j != i}

    def sync_parameters(self, src=0):
        """Data-parallel start: every rank takes rank `src`'s actor / critic / target weights (one broadcast
        of a flat buffer per network).  Call once the lazy layers exist (after the first forward)."""
        d = self.dist
        if d is None or not d.is_initialized():
            return
        for ag in self.agents:
            for net in (ag.actor_model, ag.target_actor_model, ag.critic_model, ag.target_critic_model):
                ps = [p for p in net.parameters() if not isinstance(p, nn.parameter.UninitializedParameter)]
                if not ps:
                    continue
                flat = torch.cat([p.detach().reshape(-1) for p in ps])
                d.broadcast(flat, src=src)
                off = 0
                with torch.no_grad():
                    for p in ps:
                        p.copy_(flat[off:off + p.numel()].view_as(p))
                        off += p.numel()

    def update(self):
        interval = len(self.agents) * 100
        if self.agents[0].update_num % interval == 0:
            for i in range(len(self.agents)):
                self.agents[i].update()
                print('update target agent{}'.format(i + 1))

    def update_init(self):
        for ag in self.agents:
            ag.update_init()

    # checkpoints (master_DDPG_truss2D_MO.py:710-733, 885-906 use Keras save/load_weights)
    def save_weights(self, prefix):
        for i, ag in enumerate(self.agents):
            torch.save({"actor": ag.actor_model.state_dict(), "critic": ag.critic_model.state_dict()},
                       "{}Agent{}_pickle.pt".format(prefix, i + 1))

    def load_tf_actors(self, prefix):
        """Restore the actors (and target actors) from the reference's published TensorFlow checkpoints
        `<prefix>Agent{i}_Actor_pickle.{index,data-00000-of-00001}` (master…:721-729; model/2000pickle_base).
        Critic shards are not published; critics keep their initialisation.  Returns parameters copied."""
        import tf_checkpoint
        n = 0
        for i, ag in enumerate(self.agents):
            path = "{}Agent{}_Actor_pickle".format(prefix, i + 1)
            n += tf_checkpoint.load_gcn_actor(ag.actor_model, path)
            tf_checkpoint.load_gcn_actor(ag.target_actor_model, path)
        return n

    def load_weights(self, prefix):
        import os
        if not os.path.exists("{}Agent1_pickle.pt".format(prefix)) and os.path.exists("{}Agent1_Actor_pickle.index".format(prefix)):
            self.load_tf_actors(prefix)          # a directory written by the reference (TensorFlow format)
            return
        for i, ag in enumerate(self.agents):
            sd = torch.load("{}Agent{}_pickle.pt".format(prefix, i + 1), weights_only=True, map_location=self.device)
            ag.actor_model.load_state_dict(sd["actor"])
            ag.target_actor_model.load_state_dict(sd["actor"])
            ag.critic_model.load_state_dict(sd["critic"])
            ag.target_critic_model.load_state_dict(sd["critic"])
            ag._targets_ready = True
