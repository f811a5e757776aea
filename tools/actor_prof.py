"""Micro-benchmark behind DESIGN.md section 4.2c: the fused GCN aggregation kernel vs torch.matmul + bias + relu,
and one full actor inference (4096 graphs of 16 nodes, 200 channels) through both paths.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, "mop-truss-marl_amd")
import torch, truss_mi355 as tm
from truss_mi355 import marl
import truss2D_RL as RL
lib = tm.load(); dev = "cuda"
B, N, P = 4096, 16, 20
actor = RL.multimodes_actor(200, 2, 3).to(dev)
r = lambda *s: torch.rand(*s, device=dev)
A = lambda n: torch.softmax(torch.randn(B, n, n, device=dev), dim=-1)
ins = [r(B, N, 13), A(N)[0], A(N), A(N), A(N), r(B, P, 4), A(P)]
def timeit(f, n=20):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    print("actor_infer ms", timeit(lambda: marl.actor_infer(lib, actor, ins)))
    print("actor torch ms", timeit(lambda: actor([ins[0], ins[1][None].expand(B, -1, -1)] + ins[2:])))
    x = r(B, N, 200); lin = actor.gcn_l2_1.lin
    print("lin 200->200 ms", timeit(lambda: lin(x)))
    x2 = x.reshape(B * N, 200)
    print("lin flat ms", timeit(lambda: lin(x2)))
    h = lin(x).contiguous()
    print("aggregate ms", timeit(lambda: marl.gcn_aggregate(lib, ins[2], h, actor.gcn_l2_1.bias, "relu")))
    print("bmm ms", timeit(lambda: torch.relu(torch.matmul(ins[2], h) + actor.gcn_l2_1.bias)))
    print("randn ms", timeit(lambda: torch.randn(B, N, device=dev)))
    W = lin.weight.t().contiguous()
    print("mm ms", timeit(lambda: torch.mm(x2, W)))
    xh = x2.half(); Wh = W.half()
    print("mm fp16 ms", timeit(lambda: torch.mm(xh, Wh)))
