"""Gene -> design -> point evaluator for population-based optimisers.

The reference's MOEA/D benchmark copies (test/benchmarks/MOEAD/<variant>.zip) evaluate one individual at a
time with `gen_model.read_genes(genes, int_obj1, int_obj2)` (truss2D_GEN.py:117-230 of those copies): the
first N genes scale to nodal heights, the last E genes to section numbers, two repair loops and a
hard-coded mirror symmetry follow, then `Model.gen_all()` and the objective/constraint sums.  A population
of 900 individuals is a B = 900 batch for the same HIP analysis kernel the RL environment uses:
`decode_genes` is the (vectorised) host transcription of the design update, `evaluate_genes` runs the
batched analysis (`BatchedTruss.analyze`, no action decode) and returns `point[B, 4]`.

Deviation: the reference keeps gene heights as Python floats (float64); the batched design state is
float32 like everywhere else in this package, so `point` agrees to ~1e-7 relative, not bit for bit
(tests/test_genes.py pins both the decoded designs, exactly, and the points, to 1e-5).
"""
from __future__ import annotations

import numpy as np

from .topology import TrussTopology

# per reference copy: heights scale (read_genes:127) and whether the supports take part in the node mirror
GENE_VARIANTS = {
    "small": dict(max_height=8.0, mirror_supports=False),   # 00_small_bridge, 01_small_roof
    "large": dict(max_height=6.0, mirror_supports=True),    # 02_large_bridge, 03_large_roof
}


def mirror_tables(topo: TrussTopology, variant: str):
    """(dst, src) index pairs of the 'FORCE SYMMETRY' block (read_genes:169-194): nodes right <- left,
    elements left <- right, for the two-row grid families."""
    nx = topo.N // 2
    half, nb = nx // 2, nx - 1
    lo = 0 if GENE_VARIANTS[variant]["mirror_supports"] else 1
    nodes = [(nx - 1 - i, i) for i in range(lo, half)] + [(topo.N - 1 - i, nx + i) for i in range(half)]
    elems = [(r * nb + i, r * nb + nb - 1 - i) for r in range(2) for i in range(nb // 2)]
    elems += [(2 * nb + i, 2 * nb + nx - 1 - i) for i in range(nx // 2)]
    b0 = 2 * nb + nx
    elems += [(b0 + i, b0 + 2 * nb - 1 - i) for i in range(nb)]
    return np.asarray(nodes, np.int64), np.asarray(elems, np.int64)


def decode_genes(topo: TrussTopology, genes, *, variant: str, is_roof: bool, d_min: float, y_min: float = 0.0,
                 y_prev=None):
    """genes [B, N+E] in [0,1] -> (y [B,N] float64, sec [B,E] int32); read_genes:126-199.

    y_prev: heights the model held before the call (the reference's builder is stateful: bridge bottom
    nodes are not assigned, :144-149); default zeros = a fresh model."""
    g = np.atleast_2d(np.asarray(genes, np.float64))
    B, N, E = g.shape[0], topo.N, topo.E
    assert g.shape[1] == N + E, "genes = N nodal heights followed by E section genes"
    v = GENE_VARIANTS[variant]
    h = g[:, :N] * v["max_height"]
    y = np.zeros((B, N)) if y_prev is None else np.array(np.broadcast_to(y_prev, (B, N)), np.float64)
    top = topo.top.astype(bool)
    if is_roof:
        free_y = topo.res[:, 1] == 0
        y[:, free_y] = np.maximum(h[:, free_y], d_min)
        y[:, N - 1] = 0.0                      # the for-else of :138-142 zeroes the LAST node after the loop
    else:
        y[:, top] = np.maximum(h[:, top], d_min)
    # section genes: min(4, round(4 g)) with Python's round-half-even (:132)
    sec = np.minimum(4, np.rint(g[:, N:] * 4.0)).astype(np.int32)
    pair = topo.pair
    # FIX NODAL HEIGHT PAIR (:156-159): a top node closer than d_min to its partner pushes the partner down
    for i in np.nonzero(top)[0]:
        j = pair[i]
        m = y[:, i] - d_min < y[:, j]
        y[m, j] = y[m, i] - d_min
    # FIX NODAL HEIGHT LOWER THAN y_min (:163-167)
    for i in np.nonzero(~top)[0]:
        j = pair[i]
        m = y[:, i] < y_min
        y[m, i] = y_min
        y[m, j] = d_min
    sn, se = mirror_tables(topo, variant)
    for d, s in sn:                            # sequential like the reference (pairs are disjoint anyway)
        y[:, d] = y[:, s]
    for d, s in se:
        sec[:, d] = sec[:, s]
    return y, sec


def evaluate_genes(env, genes, *, variant: str, is_roof: bool, d_min: float, y_min: float = 0.0):
    """point[B,4] (device tensor, float32) of a population: one `truss_step` launch without action decode.
    `env` must hold the per-env constants and the normalisers (`analyze(set_normalisers=True)` on the
    initial design = MOEAD_master.py:50-60)."""
    y, sec = decode_genes(env.topo, genes, variant=variant, is_roof=is_roof, d_min=d_min, y_min=y_min)
    if y.shape[0] != env.B:
        raise ValueError(f"population of {y.shape[0]} individuals for an env batch of {env.B}")
    env.set_design(y.astype(np.float32), sec)
    env.analyze()
    return env.point
