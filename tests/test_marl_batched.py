"""Batched MARL game steps (truss_mi355/marl.py, BASELINE configs 3-5) on the CPU backend: archive
invariants, consistency of rewards / archive update with the per-env host path, replay and training."""
import contextlib
import io

import numpy as np
import pytest
import torch

import truss_mi355 as tm
from truss_mi355 import marl, reward as RW, synthetic
import parity_common as pc
import utils as U
import master_DDPG_truss2D_MO as M
import truss2D_RL as RL


def _engine(lib, device, B=6, num_x=4, seed=3):
    topo = tm.TrussTopology.grid(num_x)
    torch.manual_seed(seed)
    rl = RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, 16, 8, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma, device=device)
    eng = marl.BatchedMARL(topo, B, rl, max_front=6, lib=lib, device=device, replay_capacity=256, batch_size=8, seed=seed)
    b = synthetic.random_batch(topo, B, seed)
    eng.reset(b["x"], b["target"], b["y_max"], b["d_min"], b["max_def"], b["load_x"], b["load_y"], b["is_roof"], b["y"], b["sec"])
    return eng


def _run(lib, device, **engine_kw):
    eng = _engine(lib, device, **engine_kw)
    B, P = eng.B, eng.P
    assert int(eng.n.min()) == 1 and torch.all(eng.pts[:, 0, :2] == 1.0)
    calls = []
    orig = RW.difference_reward

    def spy(*a, **k):
        out = orig(*a, **k)
        calls.append(([t.clone() if torch.is_tensor(t) else t for t in a], [o.clone() for o in out]))
        return out

    RW.difference_reward = spy
    marl.RW.difference_reward = spy
    try:
        w0 = [p.detach().clone() for p in eng.rl.agents[0].actor_model.parameters() if not isinstance(p, torch.nn.parameter.UninitializedParameter)]
        stats = [eng.game_step_all(train=True, explore=True, train_iters=2) for _ in range(3)]
    finally:
        RW.difference_reward = orig
        marl.RW.difference_reward = orig
    # archive invariants: 1 <= n <= P, rows sorted by obj1, mutually non-dominated, feasible, clipped to <= 1
    pts, n = eng.pts.cpu().numpy(), eng.n.cpu().numpy()
    for b in range(B):
        assert 1 <= n[b] <= P
        rows = pts[b, :n[b]]
        assert np.all(rows[:, :2] <= 1.0) and np.all(rows[:, 2:] <= 1.0)
        assert list(rows[:, 0]) == sorted(rows[:, 0])
        for i in range(n[b]):
            assert not any(rows[j, 0] < rows[i, 0] and rows[j, 1] < rows[i, 1] for j in range(n[b]))
    # every reward call agrees with the per-env host reward block on the same inputs
    args, outs = calls[0]
    front, nf, pf, npf, parent, points, ref, n_pf = [a.cpu().numpy() if torch.is_tensor(a) else a for a in args[:8]]
    for b in range(B):
        want = M.difference_reward([list(r) for r in front[b, :nf[b]]], [list(r) for r in pf[b, :npf[b]]], tuple(parent[b]),
                                   [list(p) for p in points[b]], list(ref[b]), int(n_pf[b]))
        np.testing.assert_allclose(outs[0][b].cpu().numpy(), want[:3], rtol=1e-9, atol=1e-11)
    # the stored designs reproduce their archived points (analysis of the archive = its points)
    chk = tm.BatchedTruss(eng.topo, B, device=device, lib=lib)
    chk.x.copy_(eng.c_x); chk.target.copy_(eng.c_target); chk.env_params.copy_(eng.c_params)
    for m in range(int(n.max())):
        chk.y.copy_(eng.arch_y[:, m]); chk.sec.copy_(eng.arch_sec[:, m])
        chk.analyze()
        got = chk.point.cpu().numpy().astype(np.float64)
        for b in range(B):
            if m < n[b] and not (pts[b, m, 2] == 0.0 and pts[b, m, 3] == 0.0):   # the initial member is archived as [1, 1, 0, 0] (master…:168)
                np.testing.assert_allclose(np.minimum(got[b, :2], 1.0), pts[b, m, :2], rtol=1e-6)
                np.testing.assert_allclose(got[b, 2:], pts[b, m, 2:], rtol=1e-6, atol=1e-9)
    assert stats[-1]["replay_size"] >= 1 and eng.env_steps >= 3 * B
    w1 = [p.detach() for p in eng.rl.agents[0].actor_model.parameters()]
    assert eng.replay.size < eng.batch_size or any(not torch.equal(a, b) for a, b in zip(w0, w1[: len(w0)])) or len(w0) == 0
    assert torch.isfinite(stats[-1]["hv"]).all()


def test_batched_marl_emulated():
    with contextlib.redirect_stdout(io.StringIO()):
        _run(pc.emu_lib(), "cpu")


@pytest.mark.gpu
def test_batched_marl_hip():
    with contextlib.redirect_stdout(io.StringIO()):
        _run(tm.load(), "cuda")


@pytest.mark.gpu
def test_batched_marl_hip_small_roof_256_envs():
    """BASELINE configs[2]'s truss (test/01_small_roof: 16 nodes / 36 elements) at 256 envs: the same invariants"""
    with contextlib.redirect_stdout(io.StringIO()):
        _run(tm.load(), "cuda", B=256, num_x=8, seed=7)


def _check_actor_infer(lib, device):
    """fused GCN aggregation (truss_gcn_aggregate) vs the plain PyTorch float32 actor"""
    torch.manual_seed(0)
    B, N, P = 37, 16, 20
    actor = RL.multimodes_actor(200, 2, 3).to(device)
    r = lambda *s: torch.rand(*s, device=device)
    A = lambda n: torch.softmax(torch.randn(B, n, n, device=device), dim=-1)
    ins = [r(B, N, 13), A(N)[0], A(N), A(N), A(N), r(B, P, 4), A(P)]
    with torch.no_grad():
        ref = actor([ins[0], ins[1][None].expand(B, -1, -1)] + ins[2:])
        got = marl.actor_infer(lib, actor, ins)
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)
    # the way the batched rollout calls it: the Pareto graph from `pareto_graph` (a path over the front's members) with its
    # neighbour table, the node-graph adjacencies on the truss's pattern with theirs -- every layer takes the sparse / bf16x3 path
    topo = tm.TrussTopology.grid(8)
    tab = topo.neighbor_table()
    pat = np.zeros((N, N), bool)
    for i in range(N):
        pat[i, tab[i][tab[i] >= 0]] = True
    patt = torch.tensor(pat, device=device)
    pts = torch.rand(B, P, 4, dtype=torch.float64, device=device)
    nfr = torch.randint(1, P + 1, (B,), device=device)
    x_p, A_p = marl.pareto_graph(pts, nfr, torch.zeros(B, dtype=torch.long, device=device), P)
    A_n = torch.tensor(topo.normalized_adjacency()[0], device=device)
    ins2 = [r(B, N, 13), A_n, r(B, N, N) * patt, r(B, N, N) * patt, r(B, N, N) * patt, x_p, A_p]
    with torch.no_grad():
        ref = actor([ins2[0], A_n[None].expand(B, -1, -1)] + ins2[2:])
        got = marl.actor_infer(lib, actor, ins2, nbr=torch.tensor(tab, device=device), nbr_p=torch.tensor(marl.path_graph_table(P), device=device))
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)
    h = r(B, 24, 70)                                   # odd sizes, per-env adjacency, every activation
    adj = A(24)
    bias = r(70)
    for act, f in ((None, lambda t: t), ("relu", torch.relu), ("sigmoid", torch.sigmoid)):
        torch.testing.assert_close(marl.gcn_aggregate(lib, adj, h, bias, act), f(torch.matmul(adj, h) + bias), rtol=2e-5, atol=2e-6)
    for n, c in ((32, 224), (48, 36), (64, 8), (17, 4)):       # 17..64 nodes, channel counts that are multiples of 4: the slab kernel
        h, adj, bias = r(B, n, c), A(n), r(c)
        for a_ in (adj, adj[0]):
            torch.testing.assert_close(marl.gcn_aggregate(lib, a_, h, bias, "relu"), torch.relu(torch.matmul(a_, h) + bias), rtol=2e-5, atol=2e-6)


def _check_sparse_aggregate(lib, device):
    """truss_gcn_aggregate_sparse: the sum over a truss's neighbour table equals the dense A @ H + b wherever the adjacency is zero
    outside that pattern (every node-graph adjacency of the reference is); through actor_infer on a 64-node truss too."""
    torch.manual_seed(3)
    for nx, B, C in ((8, 7, 8), (32, 5, 224), (128, 2, 12)):
        topo = tm.TrussTopology.grid(nx)
        N = topo.N
        tab = topo.neighbor_table()
        assert tab.shape[0] == N and tab.shape[1] <= 9 and np.all(np.diff(np.where(tab < 0, 1 << 14, tab), axis=1) > -1)
        pat = np.zeros((N, N), bool)
        for i in range(N):
            pat[i, tab[i][tab[i] >= 0]] = True
        A_n, mask = topo.normalized_adjacency()
        assert np.all(pat[A_n != 0]) and np.all(pat[mask != 0]) and pat.sum() == 2 * topo.E + N
        nbr = torch.tensor(tab, device=device)
        patt = torch.tensor(pat, device=device)
        h, bias = torch.randn(B, N, C, device=device), torch.randn(C, device=device)
        for adj in (torch.tensor(A_n, device=device), torch.rand(B, N, N, device=device) * patt):
            for act, f in ((None, lambda t: t), ("relu", torch.relu), ("sigmoid", torch.sigmoid)):
                out = torch.empty_like(h)
                from truss_mi355 import ops
                ops.call(ops.namespace().gcn_aggregate_sparse, ops.bind(lib), ops.stream_of(h.device), adj, nbr, h, bias, out,
                         {None: 0, "relu": 1, "sigmoid": 2}[act])
                torch.testing.assert_close(out, f(torch.matmul(adj, h) + bias), rtol=2e-5, atol=2e-6)
    # through the actor: 64 nodes, adjacencies on the truss's pattern
    topo = tm.TrussTopology.grid(32)
    N, B, P = topo.N, 3, 20
    A_n, mask = topo.normalized_adjacency()
    m = torch.tensor(mask, device=device)
    actor = RL.multimodes_actor(40, 2, 3).to(device)
    r = lambda *s: torch.rand(*s, device=device)
    ins = [r(B, N, 13), torch.tensor(A_n, device=device), r(B, N, N) * m, r(B, N, N) * m, r(B, N, N) * m, r(B, P, 4),
           torch.softmax(torch.randn(B, P, P, device=device), -1)]
    with torch.no_grad():
        ref = actor([ins[0], ins[1][None].expand(B, -1, -1)] + ins[2:])
        got = marl.actor_infer(lib, actor, ins, nbr=torch.tensor(topo.neighbor_table(), device=device))
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)


def _check_gcn_layer(lib, device):
    """truss_gcn_layer (the fused MFMA layer kernel: neighbourhood sum on the way in, X' W^T on the matrix cores, bias / activation /
    accumulation in the epilogue) against the plain float32 PyTorch layer act(A @ (X @ W^T) + b), at every truss size class of
    BASELINE configs (12 ... 256 nodes, sparsity pattern of the truss), on the dense Pareto graph, for the actors' layer shapes
    (13 -> 200, 4 -> 200, 200 -> 200, 200 -> 2 / 3) and odd ones.  Tolerance 2e-5 relative to the layer's output scale: the kernel
    sums in a different order ((A X) W, K in slabs of 16, two k per MFMA step)."""
    torch.manual_seed(7)
    f = {None: lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid}

    def check(x, adj, w, bias, act, nbr, accumulate=False):
        ref0 = f[act](torch.matmul(adj, torch.matmul(x, w.t())) + (bias if bias is not None else 0.0))
        for precision in ("bf16x3", "f32"):          # the bf16 matrix cores with exactly split operands (hidden layers) / the fp32 ones
            out, ref = None, ref0
            if accumulate:
                out = torch.randn_like(ref0)
                ref = ref0 + out
            got = marl.gcn_layer(lib, x, adj, w, bias, act, nbr, out, accumulate, precision=precision)
            scale = max(1.0, float(ref.abs().max()))
            torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5 * scale)

    for nx, B in ((6, 21), (8, 17), (16, 9), (32, 5), (64, 3), (128, 2)):           # 12 / 16 / 32 / 64 / 128 / 256 nodes
        topo = tm.TrussTopology.grid(nx)
        N = topo.N
        tab = topo.neighbor_table()
        pat = np.zeros((N, N), bool)
        for i in range(N):
            pat[i, tab[i][tab[i] >= 0]] = True
        nbr, patt = torch.tensor(tab, device=device), torch.tensor(pat, device=device)
        A_n, _ = topo.normalized_adjacency()
        shared, per_env = torch.tensor(A_n, device=device), torch.rand(B, N, N, device=device) * patt
        for K, C, act in ((13, 200, "relu"), (200, 200, "relu"), (200, 2, "sigmoid"), (200, 3, "sigmoid"), (37, 70, None)):
            x, w, bias = torch.randn(B, N, K, device=device), torch.randn(C, K, device=device) / K ** 0.5, torch.randn(C, device=device)
            check(x, per_env, w, bias, act, nbr)
            check(x, shared, w, bias if K != 37 else None, act, nbr, accumulate=(K == 200 and C == 200))
    for P, B in ((20, 13), (50, 4), (64, 3), (7, 40)):                               # dense: the Pareto graph (20 train / 50 test copies)
        adj = torch.softmax(torch.randn(B, P, P, device=device), dim=-1)
        x, w, bias = torch.rand(B, P, 4, device=device), torch.randn(200, 4, device=device), torch.randn(200, device=device)
        check(x, adj, w, bias, "relu", None)
        check(torch.randn(B, P, 200, device=device), adj[0], torch.randn(200, 200, device=device) / 14.0, bias, None, None)
    # the split itself: three bfloat16 terms whose sum is the float32 weight to 2^-23 of its magnitude, zero padding to [3, 224, KP]
    w = torch.randn(200, 200, device=device) * torch.logspace(-6, 6, 200, device=device)[:, None]
    ws = marl.split_weights(lib, w)
    assert ws.shape == (3, 224, 208) and ws.dtype == torch.int16
    terms = (ws.to(torch.int32) << 16).view(torch.float32)
    back = terms[0].double() + terms[1].double() + terms[2].double()
    assert float((back[:200, :200] - w.double()).abs().max() / w.abs().max()) < 2.0 ** -23
    assert torch.all((back[:200, :200] - w.double()).abs() <= w.abs().double() * 2.0 ** -23 + 1e-45)
    assert int(ws[:, 200:, :].abs().max()) == 0 and int(ws[:, :, 200:].abs().max()) == 0
    with pytest.raises(tm.TrussError):                                               # outside the envelope: refused, never silently wrong
        marl.gcn_layer(lib, torch.zeros(1, 80, 8, device=device), torch.zeros(80, 80, device=device), torch.zeros(8, 8, device=device), None, None)
    with pytest.raises(tm.TrussError):
        marl.gcn_layer(lib, torch.zeros(1, 8, 8, device=device), torch.zeros(8, 8, device=device), torch.zeros(230, 8, device=device), None, None)


def test_gcn_layer_emulated():
    _check_gcn_layer(pc.emu_lib(), "cpu")


@pytest.mark.gpu
def test_gcn_layer_hip():
    _check_gcn_layer(tm.load(), "cuda")


def _check_gcn_level(lib, device):
    """the fused level forward (truss_gcn_level: every layer of a level, of several networks, in one launch) installed as the
    forward of truss2D_RL's level operation: outputs and gradients of three merged critics + three merged actors against the
    modules' own layer-by-layer float32 evaluation; the hook must really have run (every group, with and without gradients)."""
    torch.manual_seed(2)
    B, N, P, H = 32, 16, 20, 200
    r = lambda *s: torch.rand(*s, device=device)
    A = lambda n: torch.softmax(torch.randn(B, n, n, device=device), dim=-1)
    S = [r(B, N, 13), A(N)[:1].expand(B, -1, -1), A(N), A(N), A(N), torch.ones(B, N, N, device=device), r(B, P, 4), A(P)]
    ain = [S[0], S[1], S[2], S[3], S[4], S[6], S[7]]
    acts = [t.requires_grad_() for t in (r(B, N, 2), r(B, N, 3), r(B, N, 2), r(B, N, 3), r(B, N, 2), r(B, N, 3))]
    actors = [RL.multimodes_actor(H, 2, 3).to(device) for _ in range(3)]
    critics = [RL.multimodes_critic(H, 64).to(device) for _ in range(3)]
    with torch.no_grad():
        for a, c in zip(actors, critics):
            a(ain), c(S + acts)
    fused = marl.level_forward(lib)
    calls = []

    def hook(groups, xs, ws, bs, want_grad):
        res = fused(groups, xs, ws, bs, want_grad)
        calls.append((len(groups), want_grad, res is not None))
        return res

    def loss(outs, qs):
        return sum((k + 1.0) * (o[0].sum() + o[1].pow(2).sum()) for k, o in enumerate(outs)) + sum((k + 2.0) * q.pow(2).mean() for k, q in enumerate(qs))

    RL.set_level_forward(hook, device)
    try:
        with torch.no_grad():
            o_ng = RL.run_networks([RL._actor_steps(a, ain) for a in actors], {})
        outs = RL.run_networks([RL._actor_steps(a, ain) for a in actors], {})
        qs = RL.run_networks([RL._critic_steps(c, S + acts) for c in critics], {})
        wrt = [p for n in actors + critics for p in n.parameters()] + acts
        g_fused = torch.autograd.grad(loss(outs, qs), wrt)
    finally:
        RL.set_level_forward(None, device)
    assert len(calls) == 4 + 4 + 2 and all(ok for _, _, ok in calls) and [w for _, w, _ in calls] == [False] * 4 + [True] * 6
    ref_o, ref_q = [a(ain) for a in actors], [c(S + acts) for c in critics]
    for o, o2, ref in zip(outs, o_ng, ref_o):
        for x, x2, y in zip(o, o2, ref):
            torch.testing.assert_close(x, y, rtol=2e-5, atol=2e-6)
            torch.testing.assert_close(x2, y, rtol=2e-5, atol=2e-6)
    for q, y in zip(qs, ref_q):
        torch.testing.assert_close(q, y, rtol=2e-5, atol=2e-5)
    for a, b in zip(torch.autograd.grad(loss(ref_o, ref_q), wrt), g_fused):
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-9
    # shapes outside the kernel's envelope are handed back (None), not computed wrongly
    big = RL.GCNConv(8).to(device)
    big(r(1, 80, 8), torch.eye(80, device=device)[None])
    RL.set_level_forward(hook, device)
    try:
        with torch.no_grad():
            got = RL.gcn_level([(big, r(2, 80, 8), torch.eye(80, device=device)[None], "relu")])[0]
    finally:
        RL.set_level_forward(None, device)
    assert calls[-1][2] is False and got.shape == (2, 80, 8)


def _check_gcn_level_shapes(lib, device):
    """`truss_gcn_level` called directly over a mix of shapes in ONE launch, against the float32 PyTorch layer: dense graphs of 16 /
    32 / 12 nodes (register path, LDS tables, a size that does not divide the 128-row tile), a 24-node graph on a sparsity pattern,
    k_in 200 / 256 / 8 / 13 / 40 (four full slabs, unaligned input), c_out 224 / 200 / 70 / 3, every activation, a shared and
    per-graph adjacencies, an empty layer; with and without X' = A X; shapes outside the envelope are refused."""
    from truss_mi355 import ops
    torch.manual_seed(4)
    r = lambda *s: torch.rand(*s, device=device)
    topo = tm.TrussTopology.grid(12)                                       # 24 nodes
    tab = topo.neighbor_table()
    pat = np.zeros((24, 24), bool)
    for i in range(24):
        pat[i, tab[i][tab[i] >= 0]] = True
    cases = [  # B, N, K, C, act, adjacency kind
        (5, 32, 200, 200, 1, "dense"), (7, 12, 256, 224, 0, "dense"), (33, 16, 8, 3, 2, "dense"), (9, 16, 13, 200, 1, "shared"),
        (3, 24, 40, 70, 1, "pattern"), (0, 16, 200, 200, 1, "dense"), (64, 16, 200, 200, 1, "dense"), (2, 64, 20, 33, 2, "dense")]
    X, A, NBR, W, BIAS, ACT, REF = [], [], [], [], [], [], []
    for B, N, K, C, act, kind in cases:
        x, w, b = r(B, N, K) - 0.5, (r(C, K) - 0.5) / 4, r(C) - 0.5
        if kind == "shared":
            a = torch.softmax(torch.randn(N, N, device=device), -1)
        elif kind == "pattern":
            a = r(B, N, N) * torch.tensor(pat, device=device)
        else:
            a = torch.softmax(torch.randn(B, N, N, device=device), -1)
        z = torch.matmul(a, x @ w.t()) + b
        REF.append((torch.relu(z) if act == 1 else torch.sigmoid(z) if act == 2 else z, torch.matmul(a, x)))
        X.append(x), A.append(a), W.append(w), BIAS.append(b), ACT.append(act)
        NBR.append(torch.tensor(tab, device=device) if kind == "pattern" else None)
    for with_x in (False, True):
        OUT = [torch.full((x.shape[0], x.shape[1], w.shape[0]), float("nan"), device=device) for x, w in zip(X, W)]
        XA = [torch.full(x.shape, float("nan"), device=device) for x in X] if with_x else []
        ops.call(ops.namespace().gcn_level, ops.bind(lib), ops.stream_of(torch.device(device)), X, A, NBR, W, BIAS, OUT, XA, ACT)
        for k, (o, (ro, rx)) in enumerate(zip(OUT, REF)):
            torch.testing.assert_close(o, ro, rtol=2e-5, atol=2e-5, msg=lambda m, k=k: f"case {cases[k]}: {m}")
            if with_x:
                torch.testing.assert_close(XA[k], rx, rtol=2e-5, atol=2e-6, msg=lambda m, k=k: f"X' of case {cases[k]}: {m}")
    one = lambda x, a, w: ops.call(ops.namespace().gcn_level, ops.bind(lib), ops.stream_of(torch.device(device)), [x], [a], [], [w], [r(w.shape[0])],
                                   [torch.empty(x.shape[0], x.shape[1], w.shape[0], device=device)], [], [0])
    for x, a, w in ((r(1, 80, 8), r(80, 80), r(8, 8)),          # dense adjacency above 64 nodes
                    (r(1, 8, 8), r(8, 8), r(230, 8)),            # c_out above 224
                    (r(1, 8, 300), r(8, 8), r(8, 300))):         # k_in above 256
        with pytest.raises(tm.TrussError):
            one(x, a, w)


def test_gcn_level_shapes_emulated():
    _check_gcn_level_shapes(pc.emu_lib(), "cpu")


@pytest.mark.gpu
def test_gcn_level_shapes_hip():
    _check_gcn_level_shapes(tm.load(), "cuda")


def test_gcn_level_emulated():
    _check_gcn_level(pc.emu_lib(), "cpu")


@pytest.mark.gpu
def test_gcn_level_hip():
    _check_gcn_level(tm.load(), "cuda")


def test_sparse_aggregate_emulated():
    _check_sparse_aggregate(pc.emu_lib(), "cpu")


@pytest.mark.gpu
def test_sparse_aggregate_hip():
    _check_sparse_aggregate(tm.load(), "cuda")


def test_actor_infer_emulated():
    _check_actor_infer(pc.emu_lib(), "cpu")


@pytest.mark.gpu
def test_actor_infer_hip():
    _check_actor_infer(tm.load(), "cuda")


@pytest.mark.gpu
def test_train_graph_matches_eager_updates():
    """The hipGraph-captured MADDPG update (one GPU) trains like the eager one: same seeds, same batches -> the
    same weights up to the run-to-run noise of atomically accumulated gradients."""
    lib = tm.load()
    out = []
    params = lambda e: [p.detach().clone() for ag in e.rl.agents for p in list(ag.actor_model.parameters()) + list(ag.critic_model.parameters())]
    for use_graph in (True, False):
        with contextlib.redirect_stdout(io.StringIO()):
            eng = _engine(lib, "cuda", B=64, seed=5)
            eng.use_train_graph = use_graph
            eng.game_step_all(train=True, explore=True, train_iters=2)      # materialises every lazy layer
            first = params(eng)
            for _ in range(3):
                eng.game_step_all(train=True, explore=True, train_iters=2)
        assert (eng._tg is not None) == use_graph
        out.append(params(eng))
        # training moves the weights: six more Adam steps at the reference's lr = 1e-7 (master…:45) are ~1e-7 each
        assert max((a - f).abs().max().item() for a, f in zip(out[-1], first)) > 3 * M.lr
    for a, b in zip(*out):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-5)



def test_replay_add_picks_next_states_like_the_explicit_form():
    """DeviceReplay.add with `src` (agent a's next state of row r = candidate tensors[src[r, a], r], picked for the accepted rows only)
    stores exactly what the explicit form (three per-agent dicts over all rows) stores."""
    torch.manual_seed(1)
    K, N, P = 9, 4, 3
    shapes = dict(x_n=(N, 13), A_s=(N, N), A_n_ts=(N, N), A_n_cs=(N, N), x_p=(P, 4), A_p=(P, P))
    S = {k: torch.rand(K, *s) for k, s in shapes.items()}
    NSall = {k: torch.rand(3, K, *s) for k, s in shapes.items()}
    src = torch.randint(0, 3, (K, 3))
    sel = torch.rand(K) > 0.4
    ag, at, R = torch.rand(K, 3, N, 2), torch.rand(K, 3, N, 3), torch.rand(K, 3)
    a, b = marl.DeviceReplay(16, N, P, "cpu"), marl.DeviceReplay(16, N, P, "cpu")
    ark = torch.arange(K)
    n1 = a.add(sel, S, [{k: NSall[k][src[:, i], ark] for k in shapes} for i in range(3)], ag, at, R)
    n2 = b.add(sel, S, NSall, ag, at, R, src=src)
    assert n1 == n2 == int(sel.sum()) and a.size == b.size and a.head == b.head
    for k in shapes:
        assert torch.equal(a.S[k], b.S[k])
        for i in range(3):
            assert torch.equal(a.NS[i][k], b.NS[i][k])
    assert torch.equal(a.a_geo, b.a_geo) and torch.equal(a.R, b.R)
