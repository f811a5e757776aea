"""Batched Pareto front / hypervolume and the difference reward of the MADDPG loop (SURVEY.md §8 rows
a23-a25) for B envs at once.

`front_hv` is the host mirror of the C entry `truss_front` (include/truss_mi355.h): utils.simple_cull /
simple_cull_final + utils.union_rectangles_fastest for a batch of small point sets, one wave per env.
`difference_reward` restates the reward block of master_DDPG_truss2D_MO.run() (:263-368) on top of it:
three launches (the three leave-one-agent-out fronts and the full front as one batch of 4 B point sets, and the archive's own
hypervolume with and without the reference point) plus a few elementwise float64 torch ops.

Deviations from the per-env host path (master_DDPG_truss2D_MO.difference_reward), both documented in the
header: (1) fronts longer than MAX_FRONT are truncated deterministically, not with random.sample;
(2) the final sum is float64 throughout, the reference's is float32 where an agent is feasible (NEP-50),
so R agrees to ~1e-6 relative.
"""
from __future__ import annotations

import torch

from . import _lib


def front_hv(points, n_points, ref_points=None, max_front=0, lib=None, stream=None):
    """points [B,P,4] float64, n_points [B] int32 (device of the library's backend).
    Returns dict(front_idx [B,P] int32, n_front [B] int32, hv_front [B], hv_all [B], metrics [B,5])."""
    lib = lib or _lib.load()
    B, P, four = points.shape
    assert four == 4 and points.dtype == torch.float64 and points.is_contiguous()
    assert n_points.dtype == torch.int32 and n_points.shape == (B,)
    if P > _lib.FRONT_MAXP:
        raise ValueError(f"at most {_lib.FRONT_MAXP} points per env")
    dev = points.device
    out = dict(front_idx=torch.empty((B, P), dtype=torch.int32, device=dev), n_front=torch.empty((B,), dtype=torch.int32, device=dev),
               hv_front=torch.empty((B,), dtype=torch.float64, device=dev), hv_all=torch.empty((B,), dtype=torch.float64, device=dev),
               metrics=torch.empty((B, 5), dtype=torch.float64, device=dev))
    if ref_points is not None:
        assert ref_points.dtype == torch.float64 and ref_points.shape == (B, 2) and ref_points.is_contiguous()
    from . import ops
    ns, stream_i = ops.namespace(), (ops.stream_of(dev) if stream is None else int(getattr(stream, "value", stream) or 0))
    ops.call(ns.front, ops.bind(lib), stream_i, int(max_front), _lib.F_FRONT_TRUNCATE if max_front else 0, points, n_points, ref_points,
             out["front_idx"], out["n_front"], out["hv_front"], out["hv_all"], out["metrics"])
    return out


def _append(base, n_base, extra, use):
    """rows of `extra` [B,K,4] with use[B,K] appended (in order) behind the first n_base rows of base [B,P,4]."""
    B, P, _ = base.shape
    K = extra.shape[1]
    out = torch.zeros((B, P + K, 4), dtype=torch.float64, device=base.device)
    out[:, :P] = base
    n = n_base.to(torch.int64)
    c = torch.cumsum(use.to(torch.int64), dim=1)
    # row k of `extra` goes to slot n + (used rows before it); unused rows are sent to the last slot with the value it already holds
    # (zero: it is the target of a used row only when all K rows are used, and then nothing is unused)
    idx = torch.where(use, n[:, None] + c - 1, P + K - 1)
    rows = torch.arange(B, device=base.device)[:, None]
    out[rows, idx] = torch.where(use[:, :, None], extra, 0.0)
    return out, (n + c[:, -1]).to(torch.int32)


def difference_reward(front_no, n_front_no, pf_hv, n_pf_hv, parent, points, ref_points, n_pf, max_front=20, lib=None):
    """Batched master_DDPG_truss2D_MO.difference_reward (:263-368).

    front_no [B,P,4] / n_front_no [B]   current non-dominated archive rows
    pf_hv    [B,P,4] / n_pf_hv    [B]   archive rows the step started from
    parent   [B,2]                      (obj1, obj2) of the solution the agents acted on
    points   [B,3,4]                    the three agents' new points
    ref_points [B,2], n_pf [B]          reference point schedule, len(Pf)
    returns R [B,3], G_U [B], xmax [B], ymax [B]   (float64 device tensors)"""
    f64 = torch.float64
    points = points.to(f64)
    feas = (points <= 1.0).all(dim=2)                                   # _feasible: all four entries <= 1
    # four point sets per env -- the archive plus the feasible new points with agent 0 / 1 / 2 left out, and with all three --
    # as ONE batch of 4 B sets: one append pass and one launch instead of four of each
    B = points.shape[0]
    use4 = feas[None].repeat(4, 1, 1)                                   # [4, B, 3]
    for i in range(3):
        use4[i, :, i] = False                                           # set i: leave agent i out; set 3: everyone
    rep = lambda t: t[None].expand(4, *t.shape).reshape(4 * B, *t.shape[1:])
    pts, n = _append(rep(front_no), rep(n_front_no), rep(points), use4.reshape(4 * B, 3))
    four = front_hv(pts.contiguous(), n, rep(ref_points).contiguous(), max_front, lib)
    hv4 = four["hv_front"].view(4, B)
    hyperV = hv4[3]
    metrics = four["metrics"].view(4, B, -1)[3]
    sum_distance, std_cd = metrics[:, 3], metrics[:, 4]
    compareV = front_hv(pf_hv, n_pf_hv, ref_points, 0, lib)["hv_all"]
    real_compareV = front_hv(pf_hv, n_pf_hv, None, 0, lib)["hv_all"]
    # the three agents at once ([B, 3]); per element the same operations in the same order as the per-agent form
    hv = (hv4[:3].t() - compareV[:, None]).clamp_(min=0)                               # [B, 3]
    hyperV = (hyperV - compareV).clamp_(min=0)
    npf = n_pf.to(f64)
    m = torch.clamp(real_compareV, min=0.25)
    c0 = torch.tensor((1.0, 0.5, 0.0), dtype=f64, device=points.device)              # coef[i][0]; coef[i][1] = 1 - coef[i][0]
    p0 = points[:, :, 0]
    w = c0 * (parent[:, :1] - p0).clamp_(min=0) + (1.0 - c0) * (parent[:, 1:2] - p0).clamp_(min=0)   # both terms use point[0] (master…:348-352)
    w = torch.where(feas, w, 0.0)
    mn = (m * npf)[:, None]
    R = (0.25 * w / mn + 0.25 * (hyperV[:, None] - hv) / mn + (10 * (real_compareV / npf))[:, None]
         - (0.05 * torch.clamp(std_cd, 0, 1) / npf)[:, None] + (0.05 * sum_distance / (2 * m.sqrt() * npf))[:, None])
    G_U = 20 * real_compareV / npf - std_cd / npf + sum_distance / npf
    ninf = -float("inf")
    xmax = torch.maximum(parent[:, 0], torch.where(feas, p0, ninf).max(dim=1).values)
    ymax = torch.maximum(parent[:, 1], torch.where(feas, points[:, :, 1], ninf).max(dim=1).values)
    return R, G_U, xmax, ymax
