"""MixedTrussPool: one env pool over several truss size classes (BASELINE configs[4]: "mixed 32-256-node trusses").

The reference picks a different truss per episode (master_DDPG_truss2D_MO.py:807-825): at any time its single env is of
ONE size, and over a run the sizes mix.  A batched pool holds the mix at once.  SURVEY.md §8e: "bucket by (N, E) class and
round-robin buckets across GPUs so each rank gets the same class mix":

  * a CLASS is a topology (N nodes, E elements) with a global number of envs;
  * every class is cut into BUCKETS of `bucket_envs` envs; the buckets of all classes, enumerated class-major, are dealt
    round-robin over the ranks -- every rank ends up with (almost) the same number of envs of every class, so no rank is
    left with only the expensive 256-node trusses;
  * on a rank, the buckets of one class are merged into one `BatchedTruss` (one launch per class and step);
  * classes are independent; `streams=True` puts each class's launches on its own HIP stream.  Measured on one MI355X
    (2048 / 1024 / 512 / 256 envs of 32 / 64 / 128 / 256 nodes): 190 us per pool step with streams against 163 us on one
    stream (the launches do not overlap enough to pay for the event hand-shakes), 252 against 265 us with the observation
    tensors -- hence off by default;
  * `step / analyze / observe` fan out over the classes; `point`, `status`, `obj` come back concatenated in POOL ORDER
    (class-major, bucket order), with `index()` mapping pool rows back to (class, local env).

No data crosses ranks (the env path has no collective); the deal is a pure function of
(classes, bucket_envs, world, rank), so every rank can compute every other rank's share.
"""
from __future__ import annotations

import numpy as np
import torch

from .batched import BatchedTruss
from .topology import TrussTopology


def deal_buckets(class_envs, bucket_envs, world):
    """-> per rank: list over classes of the env ranges [(lo, hi), ...] (global env ids inside the class) it owns.
    Buckets are enumerated class-major and dealt round-robin; the last bucket of a class may be short."""
    share = [[[] for _ in class_envs] for _ in range(world)]
    k = 0
    for c, n in enumerate(class_envs):
        for lo in range(0, int(n), int(bucket_envs)):
            share[k % world][c].append((lo, min(lo + int(bucket_envs), int(n))))
            k += 1
    return share


class MixedTrussPool:
    """classes: list of (TrussTopology, n_envs_global).  rank / world: this process's place in the job (one process
    per GPU).  The pool owns one BatchedTruss per class with envs on this rank."""

    def __init__(self, classes, *, bucket_envs=64, rank=0, world=1, device=None, lib=None, streams=False):
        self.classes = [(t, int(n)) for t, n in classes]
        self.rank, self.world, self.bucket_envs = int(rank), int(world), int(bucket_envs)
        self.share = deal_buckets([n for _, n in self.classes], bucket_envs, world)
        mine = self.share[self.rank]
        self.class_ids, self.ranges, self.envs = [], [], []
        for c, (topo, _) in enumerate(self.classes):
            n_local = sum(hi - lo for lo, hi in mine[c])
            if n_local == 0:
                continue
            self.class_ids.append(c)
            self.ranges.append(mine[c])
            self.envs.append(BatchedTruss(topo, n_local, device=device, lib=lib))
        if not self.envs:
            raise ValueError(f"rank {rank} of {world} owns no env: fewer buckets than ranks")
        self.device, self.lib = self.envs[0].device, self.envs[0].lib
        self.sizes = [e.B for e in self.envs]
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        self.n_envs = int(self.offsets[-1])
        self._streams = None
        if streams and self.device.type == "cuda" and len(self.envs) > 1:
            self._streams = [torch.cuda.Stream(device=self.device) for _ in self.envs]

    # ---- bookkeeping ----
    def global_ids(self, k):
        """global env ids (inside its class) of the local envs of this rank's k-th class, in local order"""
        return np.concatenate([np.arange(lo, hi) for lo, hi in self.ranges[k]])

    def index(self):
        """[n_envs, 2] int64: (class id, global env id inside the class) of every pool row"""
        rows = [np.stack([np.full(self.sizes[k], self.class_ids[k]), self.global_ids(k)], axis=1) for k in range(len(self.envs))]
        return np.concatenate(rows).astype(np.int64)

    def class_mix(self):
        """envs of every class on every rank ([world, n_classes]): what the round-robin deal balances"""
        return np.array([[sum(hi - lo for lo, hi in r[c]) for c in range(len(self.classes))] for r in self.share], dtype=np.int64)

    # ---- fan-out over the classes, one stream per class ----
    def _each(self, fn):
        if self._streams is None:
            return [fn(k, e) for k, e in enumerate(self.envs)]
        cur = torch.cuda.current_stream(self.device)
        out = []
        for k, (e, s) in enumerate(zip(self.envs, self._streams)):
            s.wait_stream(cur)                      # inputs produced on the caller's stream
            with torch.cuda.stream(s):
                out.append(fn(k, e))
        for s in self._streams:
            cur.wait_stream(s)                      # results visible to the caller's stream
        return out

    def set_constants(self, per_class):
        """per_class[k] = dict(x, target, y_max, d_min, max_def, load_x, load_y, is_roof) for this rank's k-th class, rows in
        local order (use global_ids(k) to slice a global per-class array)"""
        for e, c in zip(self.envs, per_class):
            e.set_constants(c["x"], c["target"], c["y_max"], c["d_min"], c["max_def"], c["load_x"], c["load_y"], c["is_roof"])

    def set_design(self, per_class):
        for e, c in zip(self.envs, per_class):
            e.set_design(c["y"], c["sec"])

    def analyze(self, set_normalisers=False, obs=None):
        """reset path of every class; obs=True -> list of observation dicts, one per class (shapes differ per class)"""
        return self._each(lambda k, e: e.analyze(set_normalisers=set_normalisers, obs=obs))

    def step(self, actions, coins=None, obs=None):
        """actions[k] = (a_geo [B_k, N_k, 2], a_topo [B_k, N_k, 3]) per class; one `_game_modify` per env of the pool.
        obs=True -> the observation tensors per class from the same launches (TRUSS_F_EMIT_OBS)."""
        if len(actions) != len(self.envs):
            raise ValueError(f"{len(actions)} action sets for {len(self.envs)} classes")
        return self._each(lambda k, e: e.step(actions[k][0], actions[k][1], None if coins is None else coins[k], obs=obs))

    def observe(self):
        return self._each(lambda k, e: e.observe())

    def _cat(self, name):
        return torch.cat([getattr(e, name) for e in self.envs], dim=0)

    @property
    def point(self):
        """[n_envs, 4] objectives / constraints of every env, pool order (truss2D_ENV.py:518-523)"""
        return self._cat("point")

    @property
    def obj(self):
        return self._cat("obj")

    @property
    def status(self):
        return self._cat("status")

    def env_steps_per_step(self):
        return self.n_envs


def grid_classes(num_xs, envs_per_class):
    """the reference's grid family at several sizes: [(TrussTopology.grid(num_x), n_envs), ...]"""
    if np.isscalar(envs_per_class):
        envs_per_class = [envs_per_class] * len(num_xs)
    return [(TrussTopology.grid(int(nx)), int(n)) for nx, n in zip(num_xs, envs_per_class)]
