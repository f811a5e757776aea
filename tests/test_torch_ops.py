"""The PyTorch custom-operator boundary (torch.ops.truss_mi355.*, csrc/truss_torch_ops.cpp): registration, schemas,
argument checks, Meta kernels (CPU, through the lane emulator), and -- on the GPU -- hipGraph capture of the fused
step."""
import numpy as np
import pytest
import torch

import truss_mi355 as tm
from truss_mi355 import ops, synthetic
import parity_common as pc


def test_operators_are_registered_with_mutation_schemas():
    ns = ops.namespace()
    for name in ("step", "rollout", "obs", "front", "gcn_aggregate", "gcn_aggregate_sparse"):
        assert hasattr(ns, name), name
    sch = str(ns.step.default._schema)
    assert sch.startswith("truss_mi355::step(int lib, int topo, int stream, int flags, int n_envs, int n_nodes, int n_elems")
    for out in ("Tensor(c!) y_out", "Tensor(k!) point", "Tensor(r!)? x_n", "Tensor(w!)? nN_x_e"):
        assert out in sch, out                      # outputs are declared as mutated: functionalisation / graphs see them
    assert "-> ()" in sch


def test_step_operator_matches_the_host_class_and_checks_its_tensors():
    lib = pc.emu_lib()
    topo = synthetic.bench_topology(16, 4)
    B = 5
    batch = synthetic.random_batch(topo, B, 3)
    env = pc.make_env(lib, topo, batch, debug_f64=False)
    env.analyze(set_normalisers=True)
    ag, at = synthetic.random_actions(1, B, topo.N, 4)
    g, t = torch.tensor(ag[0]), torch.tensor(at[0])
    # the raw operator, by hand, into fresh outputs
    ns, lid = ops.namespace(), ops.bind(lib)
    f32, i32 = torch.float32, torch.int32
    o = dict(y=torch.zeros(B, topo.N), sec=torch.zeros(B, topo.E, dtype=i32), mu=torch.zeros(B, topo.N), md=torch.zeros(B, topo.N),
             disp=torch.zeros(B, topo.N, 2), q0=torch.zeros(B, topo.E), sr=torch.zeros(B, topo.E),
             comp=torch.zeros(B, topo.E, dtype=torch.uint8), point=torch.zeros(B, 4),
             x_n=torch.zeros(B, topo.N, 13), nxe=torch.zeros(B, topo.E, 21))
    coin = torch.zeros(B, dtype=torch.uint8)

    def run(**over):
        a = dict(x=env.x, y_in=env.y, sec_in=env.sec, a_geo=g.clone(), a_topo=t.clone(), env_params=env.env_params)
        a.update(over)
        ns.step(lid, env.h.value, 0, tm.F_EMIT_OBS, B, topo.N, topo.E, a["x"], a["y_in"], a["sec_in"], None, None, a["a_geo"],
                a["a_topo"], coin, env.target, a["env_params"], o["y"], o["sec"], o["mu"], o["md"], o["disp"], o["q0"], o["sr"], o["comp"],
                o["point"], None, None, None, None, None, None, o["x_n"], None, None, None, None, o["nxe"])

    run()
    got = env.step(g.clone(), t.clone(), obs=True)
    r = env.results()
    assert np.array_equal(o["y"].numpy(), r["y"]) and np.array_equal(o["sec"].numpy(), r["sec"])
    assert np.array_equal(o["point"].numpy(), r["point"]) and np.array_equal(o["q0"].numpy(), r["q0"])
    assert np.array_equal(o["x_n"].numpy(), got["x_n"].numpy()) and np.array_equal(o["nxe"].numpy(), got["nN_x_e"].numpy())
    with pytest.raises(RuntimeError, match="a_geo must be Float"):
        run(a_geo=g.double())
    with pytest.raises(RuntimeError, match="x has 32 elements, needs 160"):
        run(x=env.x[:1])
    with pytest.raises(RuntimeError, match="env_params must be contiguous"):
        run(env_params=torch.zeros(8, B, dtype=torch.float64).t())
    with pytest.raises(RuntimeError, match="not bound"):
        ns.front(7, 0, 0, 0, torch.zeros(1, 4, 4, dtype=torch.float64), torch.zeros(1, dtype=i32), None, None, None, None, None, None)


def test_meta_kernels_make_the_operators_traceable():
    """On meta tensors every operator is a no-op that only 'mutates' its outputs: shape inference / tracing works
    without a device (and without touching a native library)."""
    ns = ops.namespace()
    B, N, E = 3, 32, 80
    m = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device="meta")
    ns.step(0, 0, 0, 0, B, N, E, m(B, N), m(B, N), m(B, E, dt=torch.int32), None, None, m(B, N, 2), m(B, N, 3), None, m(B, N),
            m(B, 8, dt=torch.float64), m(B, N), m(B, E, dt=torch.int32), m(B, N), m(B, N), m(B, N, 2), m(B, E), m(B, E),
            m(B, E, dt=torch.uint8), m(B, 4), None, None, None, None, None, None, m(B, N, 13), None, None, None, None, None)
    ns.gcn_aggregate(0, 0, m(N, N), m(B, N, 16), m(16), m(B, N, 16), 1)
    ns.front(0, 0, 20, 1, m(B, 8, 4, dt=torch.float64), m(B, dt=torch.int32), None, None, None, None, None, None)


@pytest.mark.gpu
def test_fused_step_is_capturable_in_a_hipgraph():
    """step(obs=...) through torch.ops inside torch.cuda.graph: the replay writes what eager steps write."""
    lib = tm.load()
    topo = synthetic.bench_topology(16, 4)
    B = 2048
    batch = synthetic.random_batch(topo, B, 11)
    ag, at = synthetic.random_actions(2, B, topo.N, 12)
    envs = [pc.make_env(lib, topo, batch, debug_f64=False) for _ in range(2)]
    for e in envs:
        e.analyze(set_normalisers=True)
    dev = envs[0].device
    G, T = torch.tensor(ag, device=dev), torch.tensor(at, device=dev)
    eager, cap = envs
    obs_e = [{k: torch.empty_like(v) for k, v in eager.obs_buffers().items()} for _ in range(2)]
    obs_c = [{k: torch.empty_like(v) for k, v in cap.obs_buffers().items()} for _ in range(2)]
    for s in range(2):
        eager.step(G[s], T[s], obs=obs_e[s])
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):     # two steps: the double-buffered design state is back where it started
        for s in range(2):
            cap.step(G[s], T[s], obs=obs_c[s])
    y_after_capture = cap.y.clone()                # capture runs nothing: the design has not moved yet
    graph.replay()
    torch.cuda.synchronize()
    assert not torch.equal(y_after_capture, cap.y) and torch.equal(cap.y, eager.y) and torch.equal(cap.sec, eager.sec)
    assert torch.equal(cap.point, eager.point)
    for s in range(2):
        for k in obs_e[s]:
            assert torch.equal(obs_c[s][k], obs_e[s][k]), (s, k)
