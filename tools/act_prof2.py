import os, sys, time, contextlib, io
sys.path.insert(0, "mop-truss-marl_amd")
import numpy as np, torch, truss_mi355 as tm
from truss_mi355 import marl
import master_DDPG_truss2D_MO as M, truss2D_RL as RL
B = 4096; dev = "cuda"
topo = tm.TrussTopology.grid(8)
rl = RL.MADDPG(M.lr, M.ep, M.epd, M.gamma, M.a_nn, M.c_nn, 100, M.num_agents, M.num_action, M.mu, M.theta, M.sigma, device=dev)
eng = marl.BatchedMARL(topo, B, rl, max_front=20, device=dev)
nx = 8
x = np.tile(np.arange(nx) * 5.0, 2); tar = np.concatenate([np.zeros(nx), [4, 3, 2.5, 2, 2, 2.5, 3, 4]])
y0 = np.concatenate([np.zeros(nx), np.full(nx, 8.0)]).astype(np.float32)
eng.reset(x[None].repeat(B, 0), tar[None].repeat(B, 0), 8.0, 0.3, 0.035, 0.0, -120000.0, 1.0, y0[None].repeat(B, 0), np.full((B, topo.E), 4, np.int32))
with contextlib.redirect_stdout(io.StringIO()):
    eng.game_step_all(train=False)
idx = torch.zeros(B, dtype=torch.int64, device=dev)
S = eng._obs(eng.envP, eng.pts, eng.n, idx)
def timeit(f, n=10):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("_act explore ms", timeit(lambda: eng._act(S, True)))
print("_act greedy  ms", timeit(lambda: eng._act(S, False)))
ins = eng._net_state(S)
ai = [ins[0], eng.A_n[0], ins[2], ins[3], ins[4], ins[6], ins[7]]
with torch.no_grad():
    print("actor_infer ms", timeit(lambda: marl.actor_infer(eng.lib, rl.agents[0].actor_model, ai)))
print("obs ms", timeit(lambda: eng._obs(eng.envP, eng.pts, eng.n, idx)))
print("pareto_graph ms", timeit(lambda: marl.pareto_graph(eng.pts, eng.n, idx, 20)))
