#!/usr/bin/env python3
"""Golden vectors for the gene -> design -> point evaluator.

Runs the REFERENCE's population-based benchmark copy of the structure builder
(/root/reference/test/benchmarks/MOEAD/<variant>.zip : truss2D_GEN.gen_model.read_genes, :117-230, with
its own FEM_2Dtruss.py) straight from the zip archives (zipimport; nothing is extracted or copied) on
seeded random gene vectors, and records genes, resulting design and `point` as tests/golden/genes.npz.
pymoo (MOEAD_master.py's optimiser) is not installed and is not needed: only the evaluator is pinned.

The reference resolves './section_data/01_brace_rod2.csv' relative to the working directory; the run
uses the repository's copy of that data file (mop-truss-marl_amd/section_data).
Usage:  python tests/golden/make_golden_genes.py
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ZIPS = "/root/reference/test/benchmarks/MOEAD"
VARIANTS = {
    # name: span_x, span_y, tar_y, load_y, type   (MOEAD_master.py:33-47 of each copy)
    "00_small_bridge": ([5] * 7, [8.0], [4, 3, 2.5, 2, 2, 2.5, 3, 4], -75000, "bridge"),
    "01_small_roof": ([5] * 7, [8.0], [4, 3, 2.5, 2, 2, 2.5, 3, 4], -120000, "roof"),
    "02_large_bridge": ([5] * 15, [6.0], [3.00, 2.75, 2.50, 2.25, 2.25, 2.00, 2.00, 2.00, 2.00, 2.00, 2.00, 2.25, 2.25,
                                          2.50, 2.75, 3.00], -7500, "bridge"),
    "03_large_roof": ([5] * 15, [6.0], [3.00, 2.75, 2.50, 2.25, 2.25, 2.00, 2.00, 2.00, 2.00, 2.00, 2.00, 2.25, 2.25,
                                        2.50, 2.75, 3.00], -8000, "roof"),
}
N_GENES = 24

WORKER = r'''
import sys, json, os
os.environ["MPLBACKEND"] = "Agg"
name, zpath, spec = sys.argv[1], sys.argv[2], json.loads(sys.argv[3])
sys.path.insert(0, zpath + "/" + name)            # zipimport: modules are read from inside the archive
import numpy as np
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    import truss2D_GEN as G
    span_x, span_y, tar_y, load_y, ttype = spec
    m = G.gen_model(len(span_x) + 1, len(span_y) + 1, span_x, span_y, tar_y, 0.3, 0, load_y, ttype, 1, None)
nodes, elems = m.model.nodes, m.model.elements
all_v = np.zeros(len(elems), np.float32)
for i, e in enumerate(elems):
    all_v[i] = e.area * e.length
all_dt = np.zeros(len(nodes), np.float32)
for i, n in enumerate(nodes):
    if n.top_node == 1:
        all_dt[i] = abs(n.target - n.coord[1])
int_obj1, int_obj2 = np.sum(all_v), np.sum(all_dt)          # MOEAD_master.py:50-60
rng = np.random.default_rng(1000 + len(nodes) + (ttype == "roof"))
genes = rng.random((int(sys.argv[4]), len(nodes) + len(elems)))
genes[0, :] = 0.0
genes[1, :] = 1.0
genes[2, len(nodes):] = np.linspace(0, 1, len(elems))       # exercises round-half-even at k/8
pts, ys, secs = [], [], []
for g in genes:
    with contextlib.redirect_stdout(io.StringIO()):
        p = m.read_genes(list(g), int_obj1, int_obj2)
    pts.append([float(v) for v in p])
    ys.append([float(n.coord[1]) for n in nodes])
    secs.append([int(e.section_no) for e in elems])
out = dict(genes=genes.tolist(), point=pts, y=ys, sec=secs, int_obj=[float(int_obj1), float(int_obj2)],
           x=[float(n.coord[0]) for n in nodes], y0=None, target=[float(n.target) for n in nodes],
           max_def=float(m.max_deformation), numpy=np.__version__)
print("@@" + json.dumps(out))
'''


def main():
    import numpy as np
    data = {}
    for name, spec in VARIANTS.items():
        z = os.path.join(ZIPS, name + ".zip")
        r = subprocess.run([sys.executable, "-c", WORKER, name, z, json.dumps(spec), str(N_GENES)], capture_output=True, text=True,
                           cwd=os.path.join(ROOT, "mop-truss-marl_amd"))
        line = [l for l in r.stdout.splitlines() if l.startswith("@@")]
        if not line:
            sys.stderr.write(r.stdout[-2000:] + r.stderr[-4000:])
            raise SystemExit(f"{name}: reference run failed")
        o = json.loads(line[0][2:])
        key = name[3:]
        for k in ("genes", "point", "y", "sec", "int_obj", "x", "target"):
            data[f"{key}__{k}"] = np.asarray(o[k], np.int32 if k == "sec" else np.float64)
        data[f"{key}__max_def"] = np.float64(o["max_def"])
        data[f"{key}__load_y"] = np.float64(spec[3])
        data[f"{key}__is_roof"] = np.int32(spec[4] == "roof")
        print(name, "ok", np.asarray(o["point"]).shape, "numpy", o["numpy"])
    data["meta"] = np.array(json.dumps({"numpy": np.__version__, "source": "MOEAD zip copies, read_genes truss2D_GEN.py:117-230"}))
    np.savez_compressed(os.path.join(HERE, "genes.npz"), **data)


if __name__ == "__main__":
    main()
