"""Entry point `python -m master_DDPG_truss2D_MO` -- the MADDPG training / evaluation loop of the
reference (train/code/master_DDPG_truss2D_MO.py) on top of the MI355X truss environment.

What is kept from the reference
  * the module-level parameters (master…:37-100) and the train/test choices (:787-795);
  * `run(game, train_period, savedata)` (:164-705): for every archived Pareto solution three agents act
    on the same parent design (three `_game_modify` transitions), difference rewards from leave-one-out
    hypervolumes (:263-368), archive update by non-dominated cull (:430-473), Pareto-graph
    observations zero-padded to MAX_PARETO_SIZE (:475-593), replay push (:612-621), reference-point
    schedule (:642), train trigger (:647-649), termination (:678-681), solution dumps in the reference's
    text format (:212-220, :683-703);
  * the reward formula bit by bit, including `R_i_weight_reward` using `point[0]` in both terms
    (:348-352) and the discarded `sorted(Pf, ...)` (:201).
What is different
  * the environment transitions run on the GPU (truss2D_ENV drop-in), the agents are PyTorch
    (truss2D_RL drop-in); nothing executes at import time: `main()` runs under `__main__`;
  * plotting (`plotforgame`, :117-156) is replaced by a CSV row per game with the same quantities the
    PNG file names encode.
"""
import csv
import gc
import os
import random

import numpy as np

from set_seed_global import seedThis
from FEM_2Dtruss import *          # noqa: F401,F403  (the reference star-imports these four modules)
from truss2D_GEN import *          # noqa: F401,F403
from truss2D_RL import *           # noqa: F401,F403
from truss2D_ENV import *          # noqa: F401,F403
from truss2D_ENV import pareto_state_data
from utils import dominates, simple_cull, CoverQuery, union_rectangles_fastest  # noqa: F401

random.seed(seedThis)
np.random.seed(seedThis)

# ---------------------------------------------------------------- parameters (master…:37-100)
gen_load_x = 0
gen_load_y = -100000
gen_load_y_eval = -120000
gen_topo_code = None
env_end_step = 50
env_num_agents = 2
lr = 0.0000001
ep = 1
epd = 0.95
gamma = 0.99
a_nn = 200
c_nn = 200
max_mem = 100000
num_agents = 3
num_action = [2, 3]
theta = [[0.1] * num_action[0], [0.1] * num_action[1]]
mu = [[0.1] * num_action[0], [0.1] * num_action[1]]
sigma = [[0.1] * num_action[0], [0.1] * num_action[1]]
train_period = 1
num_episodes = 2001
base_num = 0
save_iter = 100
MAX_PARETO_SIZE = 20

trainChoice = [
    [[4.0, 3.0, 5.0, 3.0, 5.0], [5], [1.0, 1.5, 2.0, 2.0, 1.5, 1.0], 5, 0],
    [[4.0, 3.0, 5.0, 3.0, 5.0], [5], [1.0, 3.0, 3.0, 2.0, 1.5, 1.0], 5, 0],
    [[4.0, 3.0, 5.0, 3.0, 5.0], [5], [1.0, 1.5, 2.0, 3.0, 3.0, 1.0], 5, 0],
    [[4.0, 3.0, 5.0, 3.0, 5.0], [5], [1.0, 3.0, 2.0, 2.0, 3.0, 1.0], 5, 0],
    [[4.0, 3.0, 5.0, 3.0, 5.0], [5], [3.0, 2.0, 1.0, 1.0, 2.0, 3.0], 5, 0],
]
testChoice = [[[5.0] * 7, [8], [4.0, 3.0, 2.5, 2.0, 2.0, 2.5, 3.0, 4.0], 8, 0]]

# ---------------------------------------------------------------- globals the loop works on
game_reward = [0, 0, 0, 0]
Utility, hyperS, numHV = [], [], []
super_step = 0
counter = 0
env1_test = None
reinforcement_learning = None
OPENING, CLOSING = +1, -1


def _feasible(p):
    return p[0] <= 1 and p[1] <= 1 and p[2] <= 1 and p[3] <= 1


def difference_reward(front_no, Pf_HV, parent, points, ref_points, n_pf):
    """Reward block of run() (master…:263-368) for one archived solution.

    front_no   current non-dominated archive rows [obj1, obj2, c1, c2, ...]
    Pf_HV      rows [obj1, obj2, c1, c2] of the archive the step started from
    parent     (obj1, obj2) of the solution the three agents acted on
    points     [point_a0, point_a1, point_a2]
    returns    (R_0, R_1, R_2, G_U, max(for_xmax), max(for_ymax))"""
    feas = [_feasible(p) for p in points]
    fronts, scal = [], []
    for i in range(3):                                   # leave agent i out
        ff = [e for e in front_no]
        for j in range(3):
            if j != i and feas[j]:
                ff.append(points[j])
        fr, _, _, _, _, _ = simple_cull(ff)
        fronts.append(fr)
    for_front = [e for e in front_no]
    for_xmax, for_ymax = [parent[0]], [parent[1]]
    for j in range(3):
        if feas[j]:
            for_front.append(points[j])
            for_xmax.append(points[j][0])
            for_ymax.append(points[j][1])
    front, max_d, dis_d, p_cd, sum_distance, std_cd = simple_cull(for_front)
    hv = [union_rectangles_fastest(fronts[i], OPENING, CLOSING, ref_point=ref_points) for i in range(3)]
    hyperV = union_rectangles_fastest(front, OPENING, CLOSING, ref_point=ref_points)
    compareV = union_rectangles_fastest(Pf_HV, OPENING, CLOSING, ref_point=ref_points)
    Real_compareV = union_rectangles_fastest(Pf_HV, OPENING, CLOSING, ref_point=[1, 1])
    hv = [max([0, h - compareV]) for h in hv]
    hyperV = max([0, hyperV - compareV])
    w = [0, 0, 0]
    coef = [(1, 0), (1 / 2, 1 / 2), (0, 1)]
    for i in range(3):
        if feas[i]:                                      # both terms use point[0] (master…:348-352)
            w[i] = coef[i][0] * max([0, (parent[0] - points[i][0])]) + coef[i][1] * max([0, (parent[1] - points[i][0])])
    m = max([0.25, Real_compareV])
    # same left-to-right association as master…:365-367: w[i] is np.float32 when agent i is feasible, so the
    # whole sum is float32 arithmetic (NEP-50) and the order of the additions is observable
    R = [0.25 * w[i] / (m * n_pf) + 0.25 * (hyperV - hv[i]) / (m * n_pf) + 10 * (Real_compareV / n_pf)
         - 0.05 * max([0, min([1, std_cd])]) / n_pf + 0.05 * sum_distance / (2 * (m ** 0.5) * n_pf) for i in range(3)]
    G_U = (20 * Real_compareV / n_pf) - (1 * std_cd) / n_pf + 1 * sum_distance / n_pf
    return R[0], R[1], R[2], G_U, max(for_xmax), max(for_ymax)


def pad_pareto_graph(x_pf, A_pf, size=None):
    """zero-pad / truncate the Pareto-graph observation to `size` nodes (master…:488-593)."""
    size = MAX_PARETO_SIZE if size is None else size
    n = x_pf.shape[0]
    add = size - n
    if add > 0:
        x_pf = np.block([[x_pf], [np.zeros((add, 4))]])
        A_pf = np.block([[A_pf, np.zeros((n, add))], [np.zeros((add, n)), np.zeros((add, add))]])
    elif add < 0:
        x_pf = x_pf[:size, :]
        A_pf = A_pf[:size, :size]
    return x_pf, A_pf


def _dump(game, name):
    os.makedirs(os.path.dirname(name), exist_ok=True)
    game.gen_model.savetxt(name)


def run(game, train_period=1, savedata=1):
    """One game (episode); returns the number of FEM analyses (master…:164-705)."""
    global game_reward
    env1_test.reset()
    S0 = list(env1_test.game._game_get_1_state())
    Pf = [[1, 1, 0, 0, S0, None, None, None, None, None, None, 0, 0, 0, S0, S0, S0, S0]]
    Pf_HV = [[1, 1, 0, 0]]
    my_step = 0
    for_out = []
    ref_points = [1, 1]
    sizeHV_HVl = {0.0: 0, 0.1: 0, 0.2: 0, 0.3: 0, 0.4: 0, 0.5: 0, 0.6: 0, 0.7: 0, 0.8: 0, 0.9: 0, 10: 0}
    front_no = [[1, 1, 0, 0]]
    hv_margin = 0.2
    agents = reinforcement_learning.agents
    while env1_test.over != 1:
        sorted(Pf, key=lambda x: x[0])                    # result discarded, as in the reference (:201)
        Ppf = Pf
        Ppf_no = [Pf[i][:4] + [i] for i in range(len(Pf))]
        n_pf = len(Pf)
        for sol in range(n_pf):
            if savedata == 1:
                _dump(env1_test.game, os.path.join('MADDPG_Model_data_txt_Game{}'.format(counter + 1),
                                                   'Game{}_Step{}_Sol_{}.txt'.format(counter + 1, env1_test.game.game_step, sol)))
            state = Pf[sol][4]
            acts, points, nxt = [], [], []
            for i in range(3):                            # three agents, same parent design (:249-260)
                a_geo, a_topo = agents[i].act(state[0], state[1], state[2], state[3], state[4], state[6], state[7])
                point, St = env1_test.game._game_modify(state[-3], state[-2], state[-1], [a_geo, a_topo])
                acts.append((a_geo, a_topo))
                points.append(point)
                nxt.append(St)
            R_0, R_1, R_2, G_U, _, _ = difference_reward(front_no, Pf_HV, (Pf[sol][0], Pf[sol][1]), points, ref_points, n_pf)
            ok = [(p[2] <= 1) and (p[3] <= 1) for p in points]
            flat = [acts[0][0], acts[0][1], acts[1][0], acts[1][1], acts[2][0], acts[2][1], R_0, R_1, R_2]
            # archive candidates (master…:374-415); a missing agent's next state is a random survivor's
            if any(ok):
                surv = [i for i in range(3) if ok[i]]
                if len(surv) == 3:
                    ns = [nxt[0], nxt[1], nxt[2]]
                    rows = [(i, ns) for i in surv]
                elif len(surv) == 2:
                    rows = []
                    for i in surv:
                        ns = [nxt[j] if ok[j] else random.choice([nxt[surv[0]], nxt[surv[1]]]) for j in range(3)]
                        rows.append((i, ns))
                else:
                    rows = [(surv[0], [nxt[surv[0]]] * 3)]
                for i, ns in rows:
                    p = points[i]
                    Ppf.append([p[0], p[1], p[2], p[3], nxt[i]] + flat + [ns[0], ns[1], ns[2], state])
                    Ppf_no.append([p[0], p[1], p[2], p[3], len(Ppf_no)])
            game_reward[0] += R_0
            game_reward[1] += R_1
            game_reward[2] += R_2
            game_reward[3] += G_U
            my_step += 3
        # ---- archive update (master…:430-473)
        front_no, new_max_d, new_dis_d, _, new_sum_d, _, front_no_edges = simple_cull(Ppf_no, True)
        for row in front_no:
            row[0] = min(row[0], 1)
            row[1] = min(row[1], 1)
        env1_test.game.current_hv = union_rectangles_fastest(front_no, OPENING, CLOSING, ref_point=[1, 1])
        env1_test.game.front_max_distance = new_max_d
        env1_test.game.front_dis_distance = new_dis_d
        key = round(env1_test.game.current_hv, 1)
        if sizeHV_HVl.get(key, 0) < new_sum_d:
            sizeHV_HVl[key] = new_sum_d
        front = [Ppf[int(r[-1])] for r in front_no_edges]
        for r in front_no:
            for_out.append('{} {} {}'.format(env1_test.game.game_step, r[0], r[1]))
        # ---- Pareto-graph observations + replay (master…:475-621)
        for sol in range(len(front)):
            for slot in (-1, -2, -3, -4, 4):
                x_pf, A_pf = pareto_state_data(front, index=sol)
                front[sol][slot][-5], front[sol][slot][-4] = pad_pareto_graph(x_pf, A_pf)
            if train_period == 1 and front[sol][5] is not None:
                reinforcement_learning.remember(front[sol][-1], front[sol][5], front[sol][6], front[sol][7], front[sol][8],
                                                front[sol][9], front[sol][10], [front[sol][11], front[sol][12], front[sol][13]],
                                                front[sol][14], front[sol][15], front[sol][16],
                                                env1_test.game.done_counter, 1)
        ref_points = [min([1, ref_points[0] + hv_margin]), min([1, ref_points[1] + hv_margin])]
        if train_period == 1:
            reinforcement_learning.train()
            reinforcement_learning.update()
        Pf = front
        Pf_HV = [x[:4] for x in Pf]
        hyperS.append(env1_test.game.current_hv)
        Utility.append(game_reward[3])
        numHV.append(len(Pf))
        print('Step {} || Hypervolumes {} n {} || R0 {} R1 {} R2 {} Gr {}'.format(
            env1_test.game.game_step, round(hyperS[-1], 3), len(Pf), round(game_reward[0], 6), round(game_reward[1], 6),
            round(game_reward[2], 6), round(game_reward[3], 6)))
        if env1_test.game.game_step == env1_test.game.end_step:
            env1_test.game.done_counter = 1
        env1_test.check_over()
        env1_test.game.step()
    if savedata == 1:
        d = 'MADDPG_Model_data_txt_Game{}'.format(counter + 1)
        for sol in range(len(Pf)):
            state = Pf[sol][-1]
            env1_test.game._set_model(state[8], state[9])
            _dump(env1_test.game, os.path.join(d, 'Game{}_Step{}_Sol_{}.txt'.format(counter + 1, env1_test.game.game_step, sol)))
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'out.txt'), "w+") as f:
            for line in for_out:
                f.write(" {}\r\n".format(line))
    return my_step


def _log_game(path='00_result.csv'):
    new = not os.path.exists(path)
    with open(path, 'a', newline='') as f:
        w = csv.writer(f)
        if new:
            w.writerow(['game', 'final_hv', 'n_front', 'R0', 'R1', 'R2', 'Gr'])
        w.writerow([counter + 1, round(hyperS[-1], 3), numHV[-1]] + [round(v, 3) for v in game_reward])


def main(device=None, episodes=None):
    """The episode loop of the reference (master…:735-912): every 10th game is an evaluation on
    `testChoice`, the others train on a random `trainChoice` as roof or bridge."""
    global counter, env1_test, reinforcement_learning, super_step, game_reward, hyperS, Utility, numHV
    import torch
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    reinforcement_learning = MADDPG(lr, ep, epd, gamma, a_nn, c_nn, max_mem, num_agents, num_action, mu, theta, sigma,
                                    device=device)
    if base_num != 0:
        try:
            reinforcement_learning.load_weights('{}pickle_base/'.format(base_num))
        except Exception:
            print("No model file to restore")
    env_model = game = None
    n_ep = num_episodes if episodes is None else episodes
    while counter < n_ep:
        evaluate = counter % 10 == 0
        if evaluate:
            choice, dmin, ttype, load = testChoice[0], 0.3, 'roof', gen_load_y_eval
        else:
            choice, dmin, ttype, load = random.choice(trainChoice), 0.2, random.choice(['roof', 'bridge']), gen_load_y
        num_x, num_y = len(choice[0]) + 1, len(choice[1]) + 1
        args = (num_x, num_y, choice[0], choice[1], choice[2], dmin, gen_load_x, load, ttype, gen_topo_code, 1)
        if env_model is None:
            env_model = gen_model(*args)
        else:
            env_model.re_value(*args)
        if game is None:
            game = Game_research04(env_end_step, env_model, env_num_agents)
        else:
            game.re_game(env_end_step, env_model, env_num_agents)
        env1_test = ENV(game)
        print('Episode{}'.format(counter + 1))
        super_step += run(game, 0 if evaluate else train_period, savedata=1 if evaluate else 0)
        if evaluate:
            _log_game()
        game_reward = [0, 0, 0, 0]
        hyperS, Utility, numHV = [], [], []
        counter += 1
        gc.collect()
        if counter % save_iter == 0:
            os.makedirs('{}pickle_base'.format(counter), exist_ok=True)
            reinforcement_learning.save_weights('{}pickle_base/'.format(counter))
        print('TOTAL ANALYSIS {}'.format(super_step))


if __name__ == "__main__":
    main()
