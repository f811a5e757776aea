"""Synthetic random-geometry truss batches for benchmarks and parity tests.

Per env: integer bay widths U{1..5} m, y_max U{5..20} m, d_min 0.3 m, bridge/roof 50/50, load
-U(5e3, 1.2e5) N, targets U(0.2, y_max - d_min) on the 0.01 grid, initial top heights
U(2 d_min, y_max) on the 0.01 grid, sections U{0..4}; bottom chord at y = 0.
Actions: U(0,1) float32.  Everything is generated with numpy on the host (outside any timed
region) and is reproducible from the seed.
"""
from __future__ import annotations

import numpy as np

from .topology import TrussTopology


def bench_topology(num_x=16, n_extra=4):
    """Reference grid families (E = 5*num_x - 4) plus `n_extra` long braces bottom node i -> top
    node i+2 at evenly spaced bays (32 nodes / 80 elements when num_x=16)."""
    extra = None
    if n_extra:
        bays = np.linspace(1, num_x - 3, n_extra).round().astype(int)
        extra = [(int(i), int(num_x + i + 2)) for i in bays]
    return TrussTopology.grid(num_x, None, extra_elements=extra)


def random_batch(topo: TrussTopology, n_envs: int, seed: int = 1234):
    rng = np.random.default_rng(seed)
    B, N, E = n_envs, topo.N, topo.E
    nx = N // 2
    span = rng.integers(1, 6, size=(B, nx - 1)).astype(np.float64)
    xs = np.concatenate([np.zeros((B, 1)), np.cumsum(span, axis=1)], axis=1)
    x = np.concatenate([xs, xs], axis=1)
    y_max = rng.integers(5, 21, size=B).astype(np.float64)
    d_min = np.full(B, 0.3)
    is_roof = rng.integers(0, 2, size=B).astype(np.float64)
    load_y = -rng.uniform(5e3, 1.2e5, size=B)
    max_def = 0.001 * span.sum(axis=1)                       # truss2D_GEN.py:78
    tar = np.round(rng.uniform(0.2, 1.0, size=(B, nx)) * (y_max - d_min)[:, None], 2)
    target = np.concatenate([np.zeros((B, nx)), tar], axis=1)
    ytop = np.round(2 * d_min[:, None] + rng.uniform(0, 1, size=(B, nx)) * (y_max - 2 * d_min)[:, None], 2)
    y = np.concatenate([np.zeros((B, nx)), ytop], axis=1).astype(np.float32)
    sec = rng.integers(0, 5, size=(B, E)).astype(np.int32)
    return dict(x=x.astype(np.float32), y=y, sec=sec, target=target.astype(np.float32), y_max=y_max, d_min=d_min,
                max_def=max_def, load_x=np.zeros(B), load_y=load_y, is_roof=is_roof)


def random_actions(n_sets: int, n_envs: int, n_nodes: int, seed: int = 4321):
    rng = np.random.default_rng(seed)
    return (rng.random((n_sets, n_envs, n_nodes, 2), dtype=np.float32),
            rng.random((n_sets, n_envs, n_nodes, 3), dtype=np.float32))
