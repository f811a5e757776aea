"""RCCL on the target: every code path that touches a process group -- bench.py's distributed plumbing, the data-parallel MADDPG
update with its gradient all-reduces (captured into the hipGraph with the update), `sync_parameters`, a batched game step --
run here with a ONE-rank `nccl` (= RCCL) group on cuda:0, which is what a one-GPU box can execute.  The world-size-2 semantics
(different shards, identical weights after the collective update) are covered on CPU by tests/test_distributed_gloo.py."""
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import time

import pytest
import torch

import truss_mi355 as tm
from truss_mi355 import marl, synthetic
import master_DDPG_truss2D_MO as M
import truss2D_RL as RL
import parity_common as pc

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl():
    """a one-rank RCCL process group on cuda:0 for the tests of this module"""
    import torch.distributed as dist
    assert torch.cuda.is_available(), "these tests need the MI355X"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_allreduce_and_broadcast_execute(nccl):
    t = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
    ref = t.clone()
    nccl.all_reduce(t, op=nccl.ReduceOp.SUM)
    nccl.broadcast(t, src=0)
    m = torch.tensor([3.5], dtype=torch.float64, device="cuda")
    nccl.all_reduce(m, op=nccl.ReduceOp.MAX)
    torch.cuda.synchronize()
    assert torch.equal(t, ref) and m.item() == 3.5
    assert nccl.get_backend() == "nccl"


def _engine(lib, dist, B=64, seed=5):
    topo = tm.TrussTopology.grid(8)
    torch.manual_seed(seed)
    rl = RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, M.a_nn, M.c_nn, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma,
                   device="cuda", dist=dist)
    eng = marl.BatchedMARL(topo, B, rl, max_front=20, lib=lib, device="cuda", replay_capacity=4096, batch_size=32, seed=seed)
    b = synthetic.random_batch(topo, B, seed)
    eng.reset(b["x"], b["target"], b["y_max"], b["d_min"], b["max_def"], b["load_x"], b["load_y"], b["is_roof"], b["y"], b["sec"])
    return eng


def _params(eng):
    return [p.detach().clone() for ag in eng.rl.agents for p in list(ag.actor_model.parameters()) + list(ag.critic_model.parameters())]


def test_data_parallel_update_is_captured_with_its_collectives(nccl):
    """MADDPG(dist=...) on the GPU: the update -- ONE flat all-reduce for the three critics' gradients, one per actor -- is
    captured into a hipGraph together with its RCCL calls and trains like the eager update and like the update without a
    process group (a one-rank mean is the identity); `sync_parameters` runs its broadcasts."""
    lib = tm.load()
    out = {}
    for tag, dist, use_graph in (("group+graph", nccl, True), ("group+eager", nccl, False), ("no group", None, True)):
        with contextlib.redirect_stdout(io.StringIO()):
            eng = _engine(lib, dist)
            eng.use_train_graph = use_graph
            eng.game_step_all(train=True, explore=True, train_iters=2)      # materialises every lazy layer
            eng.rl.sync_parameters(0)
            for _ in range(3):
                eng.game_step_all(train=True, explore=True, train_iters=2)
        assert (eng._tg is not None) == use_graph, tag
        out[tag] = _params(eng)
    for other in ("group+eager", "no group"):
        for a, b in zip(out["group+graph"], out[other]):
            torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-5)


def test_update_time_with_a_group_is_graph_replay_time(nccl):
    """the data-parallel update is no longer the slow (eager) variant: replaying the graph that holds the update AND its
    collectives costs about what the update without a process group costs (within 10 %; 25 % is asserted to
    leave room for box-to-box noise, the measured ratio is printed)"""
    lib = tm.load()
    times = {}
    for tag, dist in (("group", nccl), ("no group", None)):
        with contextlib.redirect_stdout(io.StringIO()):
            eng = _engine(lib, dist, B=256)
            for _ in range(3):
                eng.game_step_all(train=True, explore=True)
        assert eng._tg is not None
        S, NS, a_geo, a_topo, R = eng.replay.sample(32, eng.gen)
        args = (eng._net_state(S), [eng._net_state(ns) for ns in NS],
                [(a_geo[:, k].contiguous(), a_topo[:, k].contiguous()) for k in range(3)], R)
        for _ in range(3):
            eng._train(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng._train(*args)
        torch.cuda.synchronize()
        times[tag] = (time.perf_counter() - t0) / 10 * 1e3
    print(f"MADDPG update (hipGraph replay): {times['group']:.2f} ms with a one-rank RCCL group, {times['no group']:.2f} ms without")
    assert times["group"] <= 1.25 * times["no group"] + 0.5


def test_bench_distributed_path_on_one_gpu(nccl):
    """bench.py with TRUSS_BENCH_FORCE_DIST=1: init_process_group("nccl", device_id=...), barriers, the MAX-reduce of the elapsed
    time and the collective decisions of the `configs` object, as its own process (it owns its process group)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(TRUSS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(pc.ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline",
                        "--configs-budget", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    o = json.loads(lines[0])
    assert o["n_gpus"] == 1 and o["value"] > 1e7 and o["config"]["path"].startswith("step")
    assert o["persistent_rollout"]["env_steps_per_s"] > 1e7 and o["state_emitting_step"]["one_launch"]
    assert "large_bridge_8192" in o["configs"]
