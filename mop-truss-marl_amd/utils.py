"""Drop-in replacement for the reference's utils.py: Pareto cull + front metrics + 2-D hypervolume.

Host-side Python (the archives hold <= 20/50 points); the batched device version of the same
arithmetic is `truss_mi355.reward` (used by the 4096-env rollouts).  Semantics follow the reference
line by line where they are observable, including the parts that look accidental:
  * infeasible rows (con1 > 1 or con2 > 1) are dropped first                       utils.py:16-23
  * strict dominance on (obj1, obj2); the front is collected in a `set` of tuples  utils.py:25-53
  * the front is sorted by obj1 only (ties keep the set's iteration order)          utils.py:54
  * when a front is longer than MAX_FRONT it is cut with `random.sample`            utils.py:120-142
    (the module-level `random` stream advances exactly as in the reference, also for the
    front-plus-edges list that is computed and then discarded)
  * crowding-distance statistics                                                   utils.py:151-199
  * hypervolume w.r.t. (1,1) minus the reference-point correction term              utils.py:275-342
"""
import random

import numpy as np

from set_seed_global import seedThis

random.seed(seedThis)
np.random.seed(seedThis)

MAX_FRONT = 20   # train copy: 20 (utils.py:120); test copies: 50.  truss2D_ENV.configure() sets both.
EDGE_KEEP = 3    # train copy keeps the 3 worst rows per objective (utils.py:67,76); test copies 10
EDGE_ADD = 1     # ... and appends 1 of them (utils.py:88,93); test copies 3


def dominates(row, candidateRow):
    return sum([row[x] < candidateRow[x] for x in range(2)]) == 2


def _front_and_metrics(Allpoint, truncate):
    feasible = [p for p in Allpoint if not (p[2] > 1 or p[3] > 1)]
    pool = list(feasible)
    pareto = set()
    while True:                              # same elimination order as utils.py:28-51
        cand = pool.pop(0)
        keep = []
        non_dominated = True
        for row in pool:
            if dominates(cand, row):
                continue
            if dominates(row, cand):
                non_dominated = False
            keep.append(row)
        pool = keep
        if non_dominated:
            pareto.add(tuple(cand))
        if len(pool) == 0:
            break
    is_front = sorted([list(p) for p in pareto], key=lambda x: x[0])
    worst_x = sorted(sorted(feasible, key=lambda x: x[0])[-max([1, min(EDGE_KEEP, len(feasible))]):], key=lambda x: x[1])
    worst_y = sorted(sorted(feasible, key=lambda x: x[1])[-max([1, min(EDGE_KEEP, len(feasible))]):], key=lambda x: x[0])
    front_and_edges = [list(p) for p in pareto]
    for i in range(min([len(worst_x), EDGE_ADD])):
        front_and_edges.append(worst_x[-i - 1])
    for i in range(min([len(worst_y), EDGE_ADD])):
        front_and_edges.append(worst_y[-i - 1])
    front_and_edges = sorted(front_and_edges, key=lambda x: x[0])
    if len(is_front) > 1:
        dist = [((is_front[i][0] - is_front[i + 1][0]) ** 2 + (is_front[i][1] - is_front[i + 1][1]) ** 2) ** 0.5
                for i in range(len(is_front) - 1)]
        for i in range(len(is_front)):
            if i == 0:
                is_front[i].append(dist[0])
            elif i == len(is_front) - 1:
                is_front[i].append(dist[-1])
            else:
                is_front[i].append(dist[i - 1] + dist[i])
    else:
        is_front[0].append(0)
    if truncate:
        if len(is_front) > MAX_FRONT:
            mid = sorted(is_front[1:-1], key=lambda x: x[-1], reverse=True)
            mid = random.sample(mid, MAX_FRONT - 2)
            is_front = [is_front[0]] + mid + [is_front[-1]]
        if len(front_and_edges) > MAX_FRONT:
            mid = sorted(front_and_edges[1:-1], key=lambda x: x[-1], reverse=True)
            mid = random.sample(mid, MAX_FRONT - 2)          # result unused; keeps the RNG stream aligned
            front_and_edges = [front_and_edges[0]] + mid + [front_and_edges[-1]]
    is_front = [p[:-1] for p in is_front]
    dist = [((is_front[i][0] - is_front[i + 1][0]) ** 2 + (is_front[i][1] - is_front[i + 1][1]) ** 2) ** 0.5
            for i in range(len(is_front) - 1)]
    if len(is_front) >= 2:
        max_distance = max(dist)
        dis_distance = (sum([((x - max_distance / len(dist)) ** 2) for x in dist]) / len(dist)) ** 0.5
        sum_distance = sum(dist)
    else:
        dis_distance, max_distance, sum_distance = 1, 0, 0
    p_norm_val = 10
    if len(is_front) > 3:
        cd = [abs(is_front[i - 1][0] - is_front[i + 1][0]) + abs(is_front[i - 1][1] - is_front[i + 1][1])
              for i in range(1, len(is_front) - 1)]
        if np.sum(np.array(cd)) == 0:
            std_cd, p_norm_inv_cd = 1, 0
        else:
            cdn = np.array(cd) / np.max(np.array(cd))
            std_cd = np.std(cdn)
            p_norm_inv_cd = sum([abs(i) ** p_norm_val for i in cdn]) ** (1 / p_norm_val)
    else:
        std_cd, p_norm_inv_cd = 1, 0
    return is_front, max_distance, dis_distance, p_norm_inv_cd, sum_distance, std_cd


def simple_cull(Allpoint, optional=False):
    out = _front_and_metrics(Allpoint, truncate=True)
    return out + (out[0],) if optional == True else out   # noqa: E712  (7th item is is_front again, utils.py:215)


def simple_cull_final(Allpoint, optional=False):
    """test copies, utils.py:220-403: the same without the MAX_FRONT truncation."""
    out = _front_and_metrics(Allpoint, truncate=False)
    return out + (out[0],) if optional == True else out   # noqa: E712


class CoverQuery:
    """Union length of a multiset of index intervals over segments of given lengths (the reference
    imports this name from utils; utils.py:222-271).  Counter per segment: O(L) per update, which is
    plenty for fronts of <= 50 points."""

    def __init__(self, L):
        assert L != []
        self.w = list(L)
        self.c = [0] * len(L)

    def cover(self):
        return sum(w for w, c in zip(self.w, self.c) if c > 0)

    def change(self, i, k, offset):
        for j in range(i, k):
            self.c[j] += offset


def union_rectangles_fastest(R, OPENING, CLOSING, ref_point=[1, 1]):
    """Area of the union of the rectangles [min(x,1), 1] x [0, 1 - min(y,1)] minus the reference-point
    correction (utils.py:275-342)."""
    if R == []:
        return 0
    if len(R) == 1 and R[0][0] == 1 and R[0][1] == 1:
        return 0
    X = set()
    events = []
    all_x, all_y = [], []
    for p in R:
        x, y = min([p[0], 1]), min([p[1], 1])
        x1, y1, x2, y2 = x, 0, 1, 1 - y
        all_x.append(p[0])
        all_y.append(p[1])
        assert x1 <= x2 and y1 <= y2
        X.add(x1)
        X.add(x2)
        events.append((y1, OPENING, x1, x2))
        events.append((y2, CLOSING, x1, x2))
    i_to_x = list(sorted(X))
    x_to_i = {v: i for i, v in enumerate(i_to_x)}
    L = [i_to_x[i + 1] - i_to_x[i] for i in range(len(i_to_x) - 1)]
    if L == []:
        L = [0]
    C = CoverQuery(L)
    area = 0
    previous_y = 0
    for y, offset, x1, x2 in sorted(events):
        area += (y - previous_y) * C.cover()
        C.change(x_to_i[x1], x_to_i[x2], offset)
        previous_y = y
    remove_area = ((1 - ref_point[0]) * (1 - min(all_x)) + (1 - ref_point[1]) * (1 - min(all_y))
                   - (1 - ref_point[0]) * (1 - ref_point[1]))
    return area - remove_area
