"""master_DDPG_truss2D_MO (reward block, Pareto-graph padding, the run() loop) and the PyTorch MADDPG
(truss2D_RL) -- CPU tests; run() goes through the lane emulator."""
import contextlib
import io
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN
import parity_common as pc


def test_difference_reward_matches_reference_formulas():
    import master_DDPG_truss2D_MO as M
    f = np.load(os.path.join(GOLDEN, "reward.npz"))
    for row, want in zip(f["rb_in"], f["rb_out"]):
        fn = row[:24].reshape(6, 4)
        front_no = fn[~np.isnan(fn[:, 0])].tolist()
        ph = row[24:48].reshape(6, 4)
        Pf_HV = ph[~np.isnan(ph[:, 0])].tolist()
        parent = row[48:50].tolist()
        points = [[np.float32(v) for v in p] for p in row[50:62].reshape(3, 4)]
        ref_points = row[62:64].tolist()
        random.seed(11)
        R0, R1, R2, GU, _, _ = M.difference_reward(front_no, Pf_HV, parent, points, ref_points, len(Pf_HV))
        np.testing.assert_allclose([R0, R1, R2, GU], want, rtol=1e-12, atol=1e-12)


def test_pad_pareto_graph():
    import master_DDPG_truss2D_MO as M
    from truss2D_ENV import pareto_state_data
    pf = [[0.2, 0.8], [0.5, 0.5], [0.9, 0.1]]
    x, A = pareto_state_data(pf, index=1)
    assert x.shape == (3, 4) and x[1, 2] == 1 and abs(x[0, 3] - 3 / 20) < 1e-7
    np.testing.assert_allclose(A, A.T)
    xp, Ap = M.pad_pareto_graph(x, A)
    assert xp.shape == (20, 4) and Ap.shape == (20, 20)
    assert np.array_equal(xp[:3], x) and not xp[3:].any() and not Ap[3:].any() and not Ap[:, 3:].any()
    xb, Ab = M.pad_pareto_graph(np.ones((25, 4)), np.ones((25, 25)))
    assert xb.shape == (20, 4) and Ab.shape == (20, 20)


def _tiny_maddpg(M, device="cpu", dist=None):
    import truss2D_RL as RL
    return RL.MADDPG(1e-4, 1, 0.95, 0.99, 8, 8, 1000, 3, [2, 3], M.mu, M.theta, M.sigma, device=device, dist=dist)


def test_actor_critic_shapes_and_quirks():
    import truss2D_RL as RL
    torch.manual_seed(0)
    B, N, P = 4, 12, 20
    actor = RL.multimodes_actor(16, 2, 3)
    ins = [torch.rand(B, N, 13), torch.rand(B, N, N), torch.rand(B, N, N), torch.rand(B, N, N), torch.rand(B, N, N),
           torch.rand(B, P, 4), torch.rand(B, P, P)]
    g, t = actor(ins)
    assert g.shape == (B, N, 2) and t.shape == (B, N, 3) and g.min() >= 0 and g.max() <= 1
    critic = RL.multimodes_critic(16, 8)
    cin = ins[:5] + [torch.rand(B, N, N)] + ins[5:] + [g, t, g, t, g, t]
    assert critic(cin).shape == (B, 1)
    assert sum(1 for m in actor.modules() if isinstance(m, RL.GCNConv)) == 13
    assert sum(1 for m in critic.modules() if isinstance(m, RL.GCNConv)) == 21
    # the Pareto embedding is tiled with stack(axis=-1) + RESHAPE (not transpose), truss2D_RL.py:87-93
    x = torch.arange(6.0).reshape(2, 3)
    tiled = RL._tile_pool(x, 4)
    assert tiled.shape == (2, 4, 3)
    assert torch.equal(tiled[0].reshape(-1), x[0].unsqueeze(-1).expand(3, 4).reshape(-1))
    # GCNConv = A @ (X W) + b
    conv = RL.GCNConv(5)
    X, A = torch.rand(2, 7, 3), torch.rand(2, 7, 7)
    out = conv(X, A)
    np.testing.assert_allclose(out.detach().numpy(), (A @ (X @ conv.lin.weight.T) + conv.bias).detach().numpy(), rtol=1e-5, atol=1e-6)


def test_actor_critic_survive_a_state_dict_round_trip():
    """A restored kernel must not be re-initialised by the first forward (round-1 bug: the Glorot flag of GCNConv
    was a plain attribute, not part of the state_dict)."""
    import truss2D_RL as RL
    torch.manual_seed(1)
    B, N, P = 3, 12, 20
    ins = [torch.rand(B, N, 13), torch.rand(B, N, N), torch.rand(B, N, N), torch.rand(B, N, N), torch.rand(B, N, N),
           torch.rand(B, P, 4), torch.rand(B, P, P)]
    actor = RL.multimodes_actor(16, 2, 3)
    g, t = actor(ins)
    critic = RL.multimodes_critic(16, 8)
    cin = ins[:5] + [torch.rand(B, N, N)] + ins[5:] + [g, t, g, t, g, t]
    q = critic(cin)
    a2, c2 = RL.multimodes_actor(16, 2, 3), RL.multimodes_critic(16, 8)      # fresh: lazy kernels not yet materialised
    a2.load_state_dict(actor.state_dict())
    c2.load_state_dict(critic.state_dict())
    g2, t2 = a2(ins)
    assert torch.equal(g, g2) and torch.equal(t, t2) and torch.equal(q, c2(cin))
    for (k, v), (k2, v2) in zip(actor.state_dict().items(), a2.state_dict().items()):
        assert k == k2 and torch.equal(v, v2), k


def test_maddpg_save_load_weights_round_trip(tmp_path):
    """MADDPG.save_weights -> load_weights into a NEW trainer (master_DDPG_truss2D_MO.main() with base_num != 0,
    master…:710-733): actors, critics and both target nets give the saved model's outputs."""
    import truss2D_RL as RL
    torch.manual_seed(2)
    N, P = 12, 20

    import master_DDPG_truss2D_MO as M

    def make():
        return RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, 16, 16, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma, device="cpu")

    m1 = make()
    ins = [torch.rand(1, N, 13), torch.rand(1, N, N), torch.rand(1, N, N), torch.rand(1, N, N), torch.rand(1, N, N),
           torch.rand(1, P, 4), torch.rand(1, P, P)]
    outs1 = [ag.actor_model(ins) for ag in m1.agents]
    acts = [o for pair in outs1 for o in pair]
    cin = ins[:5] + [torch.rand(1, N, N)] + ins[5:] + acts
    q1 = [ag.critic_model(cin) for ag in m1.agents]
    for ag in m1.agents:                      # materialise the targets too (update_init hard copy, RL:402-404)
        ag.target_actor_model(ins)
        ag.target_critic_model(cin)
        ag.update_init()
    prefix = str(tmp_path) + "/"
    m1.save_weights(prefix)
    m2 = make()
    m2.load_weights(prefix)
    for a1, a2_, (g1, t1), qa in zip(m1.agents, m2.agents, outs1, q1):
        g2, t2 = a2_.actor_model(ins)
        assert torch.equal(g1, g2) and torch.equal(t1, t2)
        gt, tt = a2_.target_actor_model(ins)
        assert torch.equal(g1, gt) and torch.equal(t1, tt)
        assert torch.equal(qa, a2_.critic_model(cin)) and torch.equal(qa, a2_.target_critic_model(cin))


def test_run_two_game_steps_emulated(tmp_path, monkeypatch):
    """One short game through run(): three agents act on every archived design, rewards are finite,
    the archive is a non-dominated feasible set, transitions reach the replay buffer and train() runs."""
    import FEM_2Dtruss
    import truss2D_ENV
    import truss2D_GEN
    import master_DDPG_truss2D_MO as M
    monkeypatch.chdir(tmp_path)
    FEM_2Dtruss._LIB = pc.emu_lib()
    try:
        random.seed(3); np.random.seed(3); torch.manual_seed(3)
        M.reinforcement_learning = _tiny_maddpg(M)
        M.reinforcement_learning.batch_size = 2
        with contextlib.redirect_stdout(io.StringIO()) as log:
            c = M.trainChoice[0]
            gm = truss2D_GEN.gen_model(6, 2, c[0], c[1], c[2], 0.2, 0, -100000, 'roof', None, 1)
            game = truss2D_ENV.Game_research04(3, gm, 2)
            M.env1_test = truss2D_ENV.ENV(game)
            M.game_reward = [0, 0, 0, 0]
            M.hyperS, M.Utility, M.numHV = [], [], []
            n_fem = M.run(game, train_period=1, savedata=1)
        assert n_fem >= 9 and n_fem % 3 == 0
        assert M.env1_test.over == 1 and game.game_step == 4
        assert len(M.hyperS) == 3 and all(np.isfinite(M.hyperS)) and all(0 <= h <= 1 for h in M.hyperS)
        assert all(np.isfinite(M.game_reward))
        assert len(M.reinforcement_learning.temprp[0]) >= 1
        assert "Step 3 ||" in log.getvalue()
        d = tmp_path / "MADDPG_Model_data_txt_Game1"
        assert (d / "out.txt").exists() and any(p.name.startswith("Game1_Step") for p in d.iterdir())
        txt = (d / "Game1_Step1_Sol_0.txt").read_bytes()
        assert txt.startswith(b" 1, [0, -100000]\r\n")            # Load repr, then nodes, then elements (GEN:193-211)
        assert len(M.reinforcement_learning.agents[0].c_loss) >= 1   # train() ran
    finally:
        FEM_2Dtruss._LIB = None


def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import master_DDPG_truss2D_MO as M
    torch.manual_seed(0); random.seed(100 + rank); np.random.seed(100 + rank)
    rl = _tiny_maddpg(M, dist=dist)
    rl.batch_size = 4
    N, P = 12, 20
    rng = np.random.default_rng(rank)          # different experience on every rank
    mk = lambda: [rng.random((N, 13), dtype=np.float32)] + [rng.random((N, N), dtype=np.float32) for _ in range(5)] + \
        [rng.random((P, 4), dtype=np.float32), rng.random((P, P), dtype=np.float32)]
    for _ in range(6):
        a = [rng.random((N, 2), dtype=np.float32), rng.random((N, 3), dtype=np.float32)] * 3
        rl.remember(mk(), *a, [0.1, 0.2, 0.3], mk(), mk(), mk(), 0, 1)
    rl.train()
    rl.train()
    w = torch.cat([p.detach().reshape(-1) for ag in rl.agents for p in list(ag.actor_model.parameters()) + list(ag.critic_model.parameters())])
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    if rank == 0:
        out.put(float((gathered[0] - gathered[1]).abs().max()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_keeps_ranks_in_sync():
    """Data-parallel MADDPG over two gloo ranks: same initial weights, different replay samples, one fused
    gradient all-reduce per optimiser step -> identical weights afterwards."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    diff = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert diff < 1e-6


def test_update_helpers_match_their_per_tensor_definitions():
    import truss2D_RL as RL
    """The multi-tensor helpers of the MADDPG update: Keras `clipnorm` per gradient tensor, and the first (= only) step of an
    Adam optimiser that is created for one call (the reference's actor update, truss2D_RL.py:629)."""
    torch.manual_seed(0)
    mk = lambda: [torch.nn.Parameter(torch.randn(4, 5)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(3, 3))]
    ps, qs = mk(), None
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    for k, (p, q) in enumerate(zip(ps, qs)):
        g = torch.randn_like(p) * (1e-4, 1.0, 1e3)[k]            # norms below, around and above the clip threshold
        p.grad, q.grad = g, g.clone()
    RL._clip_each(ps)
    for p, q in zip(ps, qs):
        n = q.grad.norm()
        want = q.grad * torch.clamp(1.0 / (n + 1e-12), max=1.0)
        assert torch.equal(p.grad, want)
        assert p.grad.norm() <= 1.0 + 1e-6
        q.grad = p.grad.clone()
    torch.optim.Adam(ps, lr=1e-3, eps=1e-7).step()
    RL._fresh_adam_step(qs, 1e-3, 1e-7)
    for p, q in zip(ps, qs):
        torch.testing.assert_close(q, p, rtol=1e-6, atol=1e-9)
    # the critics' optimiser: one shared step counter, flat moments -- six steps against torch.optim.Adam, with the flat clip
    ps, qs = mk(), None
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ref, mine = torch.optim.Adam(ps, lr=1e-2, eps=1e-7), RL.SharedStepAdam(qs, lr=1e-2, eps=1e-7)
    assert mine.state_tensors() == []
    for it in range(6):
        for p, q in zip(ps, qs):
            g = torch.randn_like(p) * 10.0 ** (it - 3)
            p.grad, q.grad = g, g.clone()
        RL._clip_each(ps)
        ref.step()
        mine.step(RL._clip_flat(RL._flat_grads(qs), qs)) if it % 2 else (RL._clip_each(qs), mine.step())
        for p, q in zip(ps, qs):
            torch.testing.assert_close(q, p, rtol=2e-6, atol=1e-8)
    assert len(mine.state_tensors()) == 3 and float(mine.step_t) == 6.0 and mine.exp_avg.numel() == sum(q.numel() for q in qs)
    # ... and the flat form of the actor's one-step Adam
    ps, qs = mk(), None
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    for p, q in zip(ps, qs):
        g = torch.randn_like(p)
        p.grad, q.grad = g, g.clone()
    RL._fresh_adam_step(ps, 1e-3, 1e-7)
    RL._fresh_adam_step(qs, 1e-3, 1e-7, RL._flat_grads(qs))
    for p, q in zip(ps, qs):
        torch.testing.assert_close(q, p, rtol=1e-6, atol=1e-9)


def test_grouped_forward_matches_layerwise():
    """The level-wise evaluation the MADDPG update uses (truss2D_RL.actor_forward_grouped / critic_forward_grouped: the layers of a
    level as batched GEMMs over stacked operands) computes what the modules' own layer-by-layer forward computes: outputs to 1e-5,
    gradients with respect to every parameter to 1e-4 of their scale; lazy kernels are still materialised by the first call."""
    import truss2D_RL as RL
    torch.manual_seed(3)
    B, N, P, H = 6, 16, 20, 48
    r = lambda *s: torch.rand(*s)
    A = lambda n: torch.softmax(torch.randn(B, n, n), -1)
    S = [r(B, N, 13), A(N)[:1].expand(B, -1, -1), A(N), A(N), A(N), torch.ones(B, N, N), r(B, P, 4), A(P)]
    ain = [S[0], S[1], S[2], S[3], S[4], S[6], S[7]]
    acts = [r(B, N, 2), r(B, N, 3), r(B, N, 2), r(B, N, 3), r(B, N, 2), r(B, N, 3)]
    actor, critic = RL.multimodes_actor(H, 2, 3), RL.multimodes_critic(H, 24)
    first = RL.actor_forward_grouped(actor, ain)           # lazy layers: the grouped entry materialises them through the module
    assert not isinstance(actor.gcn_l2_3.lin.weight, torch.nn.parameter.UninitializedParameter)
    RL.critic_forward_grouped(critic, S + acts)
    for a, b in zip(RL.actor_forward_grouped(actor, ain), actor(ain)):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(first[0], actor(ain)[0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(RL.critic_forward_grouped(critic, S + acts), critic(S + acts), rtol=1e-5, atol=1e-6)
    loss_a = lambda o: o[0].sum() + o[1].pow(2).sum()
    for net, fwd_g, fwd, loss in ((actor, lambda: RL.actor_forward_grouped(actor, ain), lambda: actor(ain), loss_a),
                                  (critic, lambda: RL.critic_forward_grouped(critic, S + acts), lambda: critic(S + acts), lambda q: q.pow(2).mean())):
        ps = list(net.parameters())
        g1 = torch.autograd.grad(loss(fwd()), ps)
        g2 = torch.autograd.grad(loss(fwd_g()), ps)
        for a, b in zip(g1, g2):
            assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-9


def _update_batch(seed, B=5, N=12, P=20, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g) * scale
    A = lambda n: torch.softmax(torch.randn(B, n, n, generator=g), -1)
    state = lambda: [r(B, N, 13), A(N), A(N), A(N), A(N), torch.ones(B, N, N), r(B, P, 4), A(P)]
    return state(), [state() for _ in range(3)], [(r(B, N, 2), r(B, N, 3)) for _ in range(3)], r(B, 3)


def test_merged_network_passes_match_single_passes():
    """`run_networks` over several networks at once (one `gcn_level` per level for all of them, the hand-written backward over the
    stacked groups) against each network's own layer-by-layer forward: outputs, parameter gradients, and the gradient that flows
    back into the critics' action inputs (the actor update's path)."""
    import truss2D_RL as RL
    torch.manual_seed(11)
    S, _, A, _ = _update_batch(1, B=4)
    ain = [S[0], S[1], S[2], S[3], S[4], S[6], S[7]]
    actors = [RL.multimodes_actor(24, 2, 3) for _ in range(3)]
    critics = [RL.multimodes_critic(24, 16) for _ in range(3)]
    acts = [t.clone().requires_grad_() for a in A for t in a]
    for a, c in zip(actors, critics):
        a(ain), c(S + acts)                                   # materialise the lazy kernels
    outs = RL.run_networks([RL._actor_steps(a, ain) for a in actors], {})
    qs = RL.run_networks([RL._critic_steps(c, S + acts) for c in critics], {})
    loss_m = sum((k + 1.0) * (o[0].sum() + o[1].pow(2).sum()) for k, o in enumerate(outs)) + sum((k + 2.0) * q.pow(2).mean() for k, q in enumerate(qs))
    loss_s = sum((k + 1.0) * (o[0].sum() + o[1].pow(2).sum()) for k, o in enumerate(a(ain) for a in actors)) + \
        sum((k + 2.0) * c(S + acts).pow(2).mean() for k, c in enumerate(critics))
    for o, a in zip(outs, actors):
        for x, y in zip(o, a(ain)):
            torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-6)
    for q, c in zip(qs, critics):
        torch.testing.assert_close(q, c(S + acts), rtol=1e-5, atol=1e-6)
    wrt = [p for n in actors + critics for p in n.parameters()] + acts
    for a, b in zip(torch.autograd.grad(loss_s, wrt), torch.autograd.grad(loss_m, wrt)):
        assert a.shape == b.shape and b.is_contiguous()
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-9


def _reference_order_update(rl, S, NS, A, R, critic_opts):
    """One MADDPG update exactly as the reference orders it (train/code/truss2D_RL.py:561-689): agent after agent -- TD target from
    the target networks, critic step (Adam, clipnorm 1 per tensor), then the actor step through the just-updated critic with a
    fresh Adam -- every network evaluated by its own layer-by-layer forward, torch.optim.Adam for the arithmetic."""
    import truss2D_RL as RL
    ain = lambda s: [s[0], s[1], s[2], s[3], s[4], s[6], s[7]]
    orders = [(0, 1, 2), (1, 0, 2), (2, 0, 1)]
    pick = lambda acts, o: [acts[o[0]][0], acts[o[0]][1], acts[o[1]][0], acts[o[1]][1], acts[o[2]][0], acts[o[2]][1]]

    def clip(params):
        for p in params:
            if p.grad is not None:
                p.grad.mul_(torch.clamp(1.0 / (p.grad.norm() + 1e-12), max=1.0))

    with torch.no_grad():
        q_next = []
        for ns in NS:
            na = [ag.target_actor_model(ain(ns)) for ag in rl.agents]
            q_next.append([ag.target_critic_model(ns + pick(na, orders[i])) for i, ag in enumerate(rl.agents)])
    for i, ag in enumerate(rl.agents):
        y = R[:, i:i + 1] + rl.gamma * (q_next[0][i] + q_next[1][i] + q_next[2][i]) / 3
        critic_opts[i].zero_grad()
        torch.mean((ag.critic_model(S + pick(A, orders[i])) - y) ** 2).backward()
        clip(ag.critic_model.parameters())
        critic_opts[i].step()
        preds = [a2.actor_model(ain(S)) for a2 in rl.agents]
        loss = -ag.critic_model(S + pick(preds, orders[i])).mean()
        ap = list(ag.actor_model.parameters())
        for p, g in zip(ap, torch.autograd.grad(loss, ap)):
            p.grad = g
        clip(ap)
        torch.optim.Adam(ap, lr=ag.lr * 0.1, eps=1e-7).step()


def test_update_matches_the_reference_order_layer_by_layer():
    """`MADDPG.train_on_batch` (critics first and together, merged level-wise passes, multi-tensor clip / Adam) against the
    reference's own order of operations evaluated layer by layer: same weights after three updates on three minibatches."""
    import copy
    import master_DDPG_truss2D_MO as M
    torch.manual_seed(5)
    rl = _tiny_maddpg(M)
    batches = [_update_batch(20 + k, scale=1.0 + k) for k in range(3)]
    S, NS, A, R = batches[0]
    rl._ensure_ready(S, [t for a in A for t in a])
    ref = copy.deepcopy(rl)
    for ag in ref.agents:
        ag.critic_opt = None
    opts = [torch.optim.Adam(ag.critic_model.parameters(), lr=ag.lr, eps=1e-7) for ag in ref.agents]
    for b in batches:
        rl.train_on_batch(*b)
        _reference_order_update(ref, *b, opts)
        for ag, bg in zip(rl.agents, ref.agents):
            for net in ("actor_model", "critic_model"):
                for p, q in zip(getattr(ag, net).parameters(), getattr(bg, net).parameters()):
                    torch.testing.assert_close(p, q, rtol=1e-4, atol=2e-6)
    assert float(rl.agents[0].critic_opt.step_t) == 3.0 and len(rl.agents[2].c_loss) == 3
