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


# ---- level-wise ("grouped") evaluation of the networks for the MADDPG update -----------------------------------------------
# At batch 32 a GCN layer is four small launches forward (X W, A H, + b, activation) and about seven backward, and a critic has 21 of
# them although it is only two levels deep: the update is bound by its kernel count (2 312 launches, 9.5 ms as a replayed hipGraph).
# The layers of one level do not depend on each other, so they are evaluated TOGETHER: the products X_i W_i of a level as one batched
# GEMM over the stacked operands, the neighbourhood sums as one batched GEMM with bias (baddbmm) over the stacked adjacencies -- or,
# where the layers share one adjacency, over the layers' outputs laid side by side along the channels -- and one activation.  Same
# parameters (the modules' own tensors: state_dict, checkpoints and the per-layer `forward` are untouched), same mathematics; the
# GEMMs sum in a different order (tests/test_master_rl.py::test_grouped_forward_matches_layerwise: outputs <= 1e-5, gradients <= 1e-4
# of their scale).

class _LevelMM(torch.autograd.Function):
    """H[i] = X[i] @ W[i]^T for the stacked inputs X [n, M, K] and the stacked kernels W [n, C, K] (nn.Linear layout).  torch.bmm
    with the transposed stack does the same forward; its backward hands the kernels' gradient back as a TRANSPOSED view, the
    per-layer slices are then not contiguous, and every multi-tensor step that follows (clip, Adam, the flat all-reduce buffer)
    falls back to one launch per parameter tensor.  Here the gradient is produced in the kernels' own layout."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return torch.bmm(x, w.transpose(1, 2))

    @staticmethod
    def backward(ctx, dh):
        x, w = ctx.saved_tensors
        dx = torch.bmm(dh, w) if ctx.needs_input_grad[0] else None
        dw = torch.bmm(dh.transpose(1, 2), x) if ctx.needs_input_grad[1] else None       # [n, C, K], contiguous
        return dx, dw


def _ready(layers):
    return not any(isinstance(L.lin.weight, nn.parameter.UninitializedParameter) for L in layers)


def _level_shared_adj(layers, xs, adj, act="relu"):
    """layers[i](xs[i], adj) for all i, one adjacency [B,N,N]: outputs side by side, [B, N, n, C]."""
    n, C = len(layers), layers[0].lin.out_features
    if all(x is xs[0] for x in xs):                       # one input too: a single wide GEMM
        h = nn.functional.linear(xs[0], torch.cat([L.lin.weight for L in layers], dim=0))            # [B,N,n*C]
    elif len({x.shape[-1] for x in xs}) == 1:
        B, N, K = xs[0].shape
        h = _LevelMM.apply(torch.stack(xs).reshape(n, B * N, K), torch.stack([L.lin.weight for L in layers]))
        h = h.reshape(n, B, N, C).permute(1, 2, 0, 3).reshape(B, N, n * C)
    else:
        h = torch.cat([nn.functional.linear(x, L.lin.weight) for L, x in zip(layers, xs)], dim=-1)
    B, N = h.shape[0], h.shape[1]
    bias = torch.cat([L.bias for L in layers])
    out = torch.baddbmm(bias.expand(B, N, n * C), adj.expand(B, N, N), h)
    out = torch.relu(out) if act == "relu" else torch.sigmoid(out) if act == "sigmoid" else out
    return out.reshape(B, N, n, C)


def _adj_stack(adjs, B, N, cache=None):
    """the adjacencies of a level, stacked for the batched GEMM: [n * B, N, N]"""
    key = (B,) + tuple(id(a_) for a_ in adjs)
    a = None if cache is None else cache.get(key)
    if a is None:
        a = torch.stack([a_.expand(B, N, N) for a_ in adjs]).reshape(len(adjs) * B, N, N)
        if cache is not None:
            cache[key] = a
            cache.setdefault("_keep", []).extend(adjs)        # the ids in the key stay unique while the cache lives
    return a


def _level_adjacencies(S):
    """The three adjacency stacks the grouped passes over minibatch S = [x_n, A_n, A_s, A_n_ts, A_n_cs, ...] need (actor levels 2 and
    3, critic level 2), built once, on the CALLER's stream: the passes of one update run on parallel streams and only read them."""
    x_n, A_n, A_s, A_n_ts, A_n_cs = S[:5]
    B, N = x_n.shape[0], x_n.shape[1]
    cache = {}
    for adjs in ([A_n, A_n_ts, A_n_cs, A_s, A_n], [A_n, A_s], [A_n, A_n_ts, A_n_cs, A_s] + [A_n] * 7):
        _adj_stack(adjs, B, N, cache)
    return cache


def _level_own_adj(layers, xs, adjs, act="relu", cache=None):
    """layers[i](xs[i], adjs[i]) for all i (same shapes): [n, B, N, C].  `cache`: a dict that lives for one update -- the stacked
    adjacencies are data only, the same for every network evaluated on the same minibatch."""
    n, C = len(layers), layers[0].lin.out_features
    B, N, K = xs[0].shape
    h = _LevelMM.apply(torch.stack(xs).reshape(n, B * N, K), torch.stack([L.lin.weight for L in layers]))              # [n, B*N, C]
    a = _adj_stack(adjs, B, N, cache)
    bias = torch.stack([L.bias for L in layers])[:, None, None, :].expand(n, B, N, C).reshape(n * B, N, C)
    out = torch.baddbmm(bias, a, h.reshape(n * B, N, C))
    out = torch.relu(out) if act == "relu" else torch.sigmoid(out) if act == "sigmoid" else out
    return out.reshape(n, B, N, C)


def actor_forward_grouped(actor, inputs, cache=None):
    """multimodes_actor.forward, level by level (see above)."""
    a = actor
    x_n, A_n, A_s, A_n_ts, A_n_cs, x_p, A_p = inputs
    layers = [a.gcn_l1_1, a.gcn_l1_2, a.gcn_l1_3, a.gcn_l1_4, a.gcn_l2_1, a.gcn_l2_2, a.gcn_l2_3, a.gcn_l2_4, a.gcn_l2_5, a.gcn_l3_1,
              a.gcn_l3_2, a.gcn_l4_1, a.gcn_l4_2]
    if not _ready(layers):
        return a(inputs)                                   # first call: let the lazy kernels materialise (Glorot) layer by layer
    l1 = _level_shared_adj([a.gcn_l1_1, a.gcn_l1_2, a.gcn_l1_3], [x_n, x_n, x_n], A_n)              # [B,N,3,H]
    # unbind, not l1[:, :, i]: its backward is ONE stack of the pieces' gradients; every select's backward is a fill, a copy and an
    # add into the running sum
    x_1_1, x_1_2, x_1_3 = l1.unbind(2)
    x_1_4 = _tile_pool(a.gcn_l1_4(x_p, A_p, "relu").sum(dim=1), x_n.shape[1])
    l2 = _level_own_adj([a.gcn_l2_1, a.gcn_l2_2, a.gcn_l2_3, a.gcn_l2_4, a.gcn_l2_5], [x_1_1, x_1_2, x_1_2, x_1_3, x_1_4],
                        [A_n, A_n_ts, A_n_cs, A_s, A_n], cache=cache)
    x_3 = l2.sum(dim=0)
    x_3_1, x_3_2 = _level_own_adj([a.gcn_l3_1, a.gcn_l3_2], [x_3, x_3], [A_n, A_s], cache=cache).unbind(0)
    return a.gcn_l4_1(x_3_1, A_n, "sigmoid"), a.gcn_l4_2(x_3_2, A_n, "sigmoid")


def critic_forward_grouped(critic, inputs, cache=None):
    """multimodes_critic.forward, level by level (see above)."""
    c = critic
    x_n, A_n, A_s, A_n_ts, A_n_cs, mask, x_p, A_p, self_g, self_t, other_g1, other_t1, other_g2, other_t2 = inputs
    if not _ready(list(c.l1) + list(c.l2)) or isinstance(c.dense_1.weight, nn.parameter.UninitializedParameter):
        return c(inputs)
    acts = [self_g, self_t, other_g1, other_t1, other_g2, other_t2]
    # level 1 over the node graph: the three layers on x_n and the six on the actions share A_n -> nine outputs side by side
    h_n = nn.functional.linear(x_n, torch.cat([c.l1[0].lin.weight, c.l1[1].lin.weight, c.l1[2].lin.weight], dim=0))
    h_a = [nn.functional.linear(a_, c.l1[4 + i].lin.weight) for i, a_ in enumerate(acts)]
    h = torch.cat([h_n] + h_a, dim=-1)                                                               # [B,N,9*H]
    B, N = h.shape[0], h.shape[1]
    H = c.l1[0].lin.out_features
    bias = torch.cat([c.l1[i].bias for i in (0, 1, 2, 4, 5, 6, 7, 8, 9)])
    l1 = torch.relu(torch.baddbmm(bias.expand(B, N, 9 * H), A_n.expand(B, N, N), h)).reshape(B, N, 9, H)
    parts = l1.unbind(2)                                   # (one stack in the backward pass instead of nine fill + copy + add)
    x_1_1, x_1_2, x_1_3 = parts[:3]
    x_1_a = list(parts[3:])
    x_1_4 = _tile_pool(c.l1[3](x_p, A_p, "relu").sum(dim=1), N)
    # level 2: eleven layers, each with its own input; adjacencies A_n, A_n_ts, A_n_cs, A_s, A_n x 6, A_n
    xs = [x_1_1, x_1_2, x_1_2, x_1_3] + x_1_a + [x_1_4]
    adjs = [A_n, A_n_ts, A_n_cs, A_s] + [A_n] * 7
    l2 = _level_own_adj(list(c.l2), xs, adjs, cache=cache)                                            # [11,B,N,H]
    # 11 x GlobalSumPool -> Concatenate: [B, 11*H]
    q = l2.sum(dim=2).permute(1, 0, 2).reshape(B, 11 * H)
    q = torch.relu(c.dense_1(q))
    q = torch.relu(c.dense_2(q))
    return c.dense_out(q)


_seg_cache: dict = {}


def _clip_each(params, max_norm=1.0):
    """Keras `clipnorm`: every gradient tensor is clipped to L2 norm <= clipnorm on its own."""
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


class SharedStepAdam:
    """`torch.optim.Adam(params, lr, betas, eps)` for a parameter list that always steps together (a critic): ONE step counter on the
    device and multi-tensor launches throughout, usable inside a hipGraph.  PyTorch's capturable Adam keeps a step tensor per
    parameter, and dividing a tensor list by a list of 0-dim tensors falls off the multi-tensor path: two launches per parameter
    tensor and step, 96 per critic update.  Same arithmetic: m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g^2;
    p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)."""

    def __init__(self, params, lr, eps=1e-8, betas=(0.9, 0.999)):
        self.params = list(params)
        self.lr, self.eps, (self.b1, self.b2) = float(lr), float(eps), betas
        self.step_t, self.exp_avg, self.exp_avg_sq = None, None, None

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def state_tensors(self):
        """the optimiser's state as a flat list (empty before the first step): step counter, first moments, second moments"""
        return [] if self.step_t is None else [self.step_t] + self.exp_avg + self.exp_avg_sq

    @torch.no_grad()
    def step(self):
        ps = self.params
        if self.step_t is None:
            self.step_t = torch.zeros((), dtype=torch.float64, device=ps[0].device)   # float64: 1 - 0.999^t cancels badly in float32
            self.exp_avg = [torch.zeros_like(p) for p in ps]
            self.exp_avg_sq = [torch.zeros_like(p) for p in ps]
        grads = [p.grad for p in ps]
        if any(g is None for g in grads):
            raise RuntimeError("SharedStepAdam: every parameter must have a gradient (the list steps together)")
        m, v = self.exp_avg, self.exp_avg_sq
        self.step_t += 1
        bc1 = 1 - torch.pow(self.b1, self.step_t)
        bc2_sqrt = (1 - torch.pow(self.b2, self.step_t)).sqrt_().float()
        torch._foreach_lerp_(m, grads, 1 - self.b1)
        torch._foreach_mul_(v, self.b2)
        torch._foreach_addcmul_(v, grads, grads, value=1 - self.b2)
        den = torch._foreach_sqrt(v)
        torch._foreach_div_(den, bc2_sqrt)
        torch._foreach_add_(den, self.eps)
        upd = torch._foreach_div(m, den)
        torch._foreach_mul_(upd, (-self.lr / bc1).float())
        torch._foreach_add_(ps, upd)


def _fresh_adam_step(params, lr, eps):
    """`Adam(params, lr, eps=eps).step()` of an optimiser that is created for this one call (the reference builds a new Adam for
    every actor update, truss2D_RL.py:629/658/688): with zero moments and step = 1 the bias-corrected moments are g and g^2, so
    the step is p -= lr * g / (|g| + eps).  Four multi-tensor launches instead of the state allocation (three fills per
    parameter tensor) and the general update."""
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    grads = [p.grad for p in ps]
    den = torch._foreach_abs(grads)
    torch._foreach_add_(den, eps)
    upd = torch._foreach_div(grads, den)
    with torch.no_grad():
        torch._foreach_add_(ps, upd, alpha=-lr)


def _allreduce_grads(param_lists, dist):
    """ONE all-reduce (mean over ranks) of the flat gradient buffer of every network in `param_lists` (a list of parameter
    lists).  The buffer is built and scattered back by multi-tensor launches; inside a hipGraph capture the collective is
    captured with the update (RCCL kernels on the capture stream)."""
    if dist is None or not dist.is_initialized():
        return
    grads = [p.grad for ps in param_lists for p in ps if p.grad is not None]
    if not grads:
        return
    world = dist.get_world_size()
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if world > 1:
        flat.mul_(1.0 / world)
    torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])


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
            if ag.critic_opt is None:
                ag.critic_opt = SharedStepAdam(ag.critic_model.parameters(), lr=ag.lr, eps=1e-7)   # multi-tensor, usable inside a hipGraph

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
        # Independent network passes run side by side: inside a hipGraph capture (truss_mi355/marl.py) each goes to its own stream,
        # forked from and joined back to the capturing stream, so that the replayed graph has parallel branches -- at batch 32 a
        # network pass is a chain of ~100 kernels that each fill a few percent of the chip.  Eager runs (warm-up, CPU tests, the
        # per-env reference loop) keep one stream: the same operations in the same order per network, hence the same numbers.
        par = self._branches()
        with torch.no_grad():
            # the three next states (one per agent's move, :561-600) go through the target networks as ONE batch of
            # 3 x batch samples: 6 network passes instead of 18 (the update is bound by its kernel count)
            nb = NS[0][0].shape[0]
            NSc = [torch.cat([NS[f][k] for f in range(3)], dim=0) for k in range(len(NS[0]))]
            cn, cs = _level_adjacencies(NSc), _level_adjacencies(S)      # data only: shared by every pass over that minibatch
            na = par([lambda ag=ag: actor_forward_grouped(ag.target_actor_model, self._actor_in(NSc), cn) for ag in self.agents])
            tq = par([lambda ag=ag, o=orders[i]: critic_forward_grouped(
                ag.target_critic_model, NSc + [na[o[0]][0], na[o[0]][1], na[o[1]][0], na[o[1]][1], na[o[2]][0], na[o[2]][1]], cn)
                for i, ag in enumerate(self.agents)])
            q_next = [[tq[i][f * nb:(f + 1) * nb] for i in range(3)] for f in range(3)]
        # The three critic updates depend on nothing another network's update changes (replayed actions, target networks): they
        # come first -- exactly the reference's results, since critic i is not touched between its own step and agent i's actor
        # update (:603-629) -- and, data-parallel, their gradients travel as ONE flat buffer (one RCCL all-reduce for the three).

        def critic_grads(i, ag):
            o = orders[i]
            # TD target; `done` never fires in the reference (it compares an action array with `is 1`)
            y = R[:, i:i + 1] + self.gamma * (q_next[0][i] + q_next[1][i] + q_next[2][i]) / 3
            ag.critic_opt.zero_grad(set_to_none=True)
            loss = torch.mean((critic_forward_grouped(ag.critic_model, S + flat(o), cs) - y) ** 2)
            loss.backward()      # (autograd.grad + `p.grad = g` would save backward()'s clone per parameter, but the gradients are then
            #                       slices of the levels' stacked gradients and PyTorch's multi-tensor Adam falls back to one launch
            #                       per tensor: 1 646 -> 1 991 launches when tried)
            return loss.detach()                     # stays on the device: no synchronisation inside the update

        for ag, loss in zip(self.agents, par([lambda i=i, ag=ag: critic_grads(i, ag) for i, ag in enumerate(self.agents)])):
            ag.c_loss.append(loss)
        cps = [list(ag.critic_model.parameters()) for ag in self.agents]
        _allreduce_grads(cps, self.dist)
        for ag, cp in zip(self.agents, cps):
            _clip_each(cp)
            ag.critic_opt.step()
        # The actor updates stay one after the other: agent i's loss re-evaluates ALL three actors (:617-629), i.e. it sees the
        # weights agents < i have just stepped to -- one collective per actor.  (The other two actors' passes carry no gradient
        # that is asked for -- autograd.grad is taken with respect to agent i's parameters only -- so they run without a graph.)
        def no_grad_eval(a2):
            with torch.no_grad():
                return actor_forward_grouped(a2.actor_model, self._actor_in(S), cs)

        frozen = {}          # no-gradient evaluations that are still valid: actor j's weights only change in iteration j
        for i, ag in enumerate(self.agents):
            o = orders[i]
            todo = [j for j in range(len(self.agents)) if j != i and j not in frozen]
            res = par([(lambda: actor_forward_grouped(ag.actor_model, self._actor_in(S), cs))] + [lambda a2=self.agents[j]: no_grad_eval(a2) for j in todo])
            for j, r_ in zip(todo, res[1:]):
                frozen[j] = r_
            preds = [res[0] if j == i else frozen[j] for j in range(len(self.agents))]
            q = critic_forward_grouped(ag.critic_model, S + [preds[o[0]][0], preds[o[0]][1], preds[o[1]][0], preds[o[1]][1], preds[o[2]][0],
                                                             preds[o[2]][1]], cs)
            actor_loss = -q.mean()
            ap = list(ag.actor_model.parameters())
            for p in ap:
                p.grad = None
            grads = torch.autograd.grad(actor_loss, ap, allow_unused=True)
            for p, g in zip(ap, grads):
                p.grad = g
            _allreduce_grads([ap], self.dist)
            _clip_each(ap)
            _fresh_adam_step(ap, ag.lr * 0.1, 1e-7)                                      # a fresh optimiser every call (:629)
            # agent i's weights have just changed: an evaluation of actor i taken before this step must not be reused (the others'
            # stay valid: agents > i have not stepped yet, evaluations of agents < i were taken after their steps)
            frozen = {j: v for j, v in frozen.items() if j != i}

    def _branches(self):
        """callable(list of thunks) -> list of results; on side streams when the current CUDA stream is being captured"""
        dev = self.device
        if dev.type != "cuda" or not torch.cuda.is_current_stream_capturing():
            return lambda fns: [f() for f in fns]
        if getattr(self, "_side_streams", None) is None:
            self._side_streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
        streams = self._side_streams

        def run(fns):
            cur = torch.cuda.current_stream(dev)
            outs = []
            for st, f in zip(streams, fns):
                st.wait_stream(cur)                  # fork: the branch sees everything enqueued so far
                with torch.cuda.stream(st):
                    outs.append(f())
            for st in streams[:len(fns)]:
                cur.wait_stream(st)                  # join
            return outs
        return run

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
