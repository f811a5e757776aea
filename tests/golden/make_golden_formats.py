#!/usr/bin/env python3
"""Golden DATA for the on-disk structure format: a structure dump written by the
REFERENCE's own `gen_model.savetxt` (train/code/truss2D_GEN.py:193-211) for one modified design, plus the
design it encodes (written under numpy's 1.x scalar print mode, see below).  Output: tests/golden/structure_small_bridge.txt (the dump, byte for byte, CRLF line
ends) and tests/golden/structure_small_bridge.json (heights / sections it must read back to)."""
import contextlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
os.environ["MPLBACKEND"] = "Agg"
os.chdir(os.path.join(ROOT, "mop-truss-marl_amd"))           # ./section_data/01_brace_rod2.csv (data file)
sys.path.insert(0, "/root/reference/train/code")
with contextlib.redirect_stdout(io.StringIO()):
    import truss2D_GEN as G
    m = G.gen_model(8, 2, [5] * 7, [8], [4, 3, 2.5, 2, 2, 2.5, 3, 4], 0.3, 0, -75000, 'bridge', 1, None)
import numpy as np
# The reference pins numpy 1.23.5, whose scalars print as plain literals; numpy 2.x prints `np.float64(..)`,
# which the reference's own reader (ast.literal_eval) cannot parse.  The dump is written in the pinned
# version's print mode, the only form in which the format round-trips.
np.set_printoptions(legacy="1.25")
rng = np.random.default_rng(5)
ys, secs = [], []
for n in m.model.nodes:
    if n.top_node == 1:
        n.coord[1] = float(np.round(rng.uniform(0.6, 8.0), 2))
    ys.append(float(n.coord[1]))
for e in m.model.elements:
    e.section_no = int(rng.integers(0, 5))
    e.area = m.truss[e.section_no][0] * 1e-4
    e.set_i(m.truss[e.section_no][1] * 1e-8)
    secs.append(e.section_no)
out = os.path.join(HERE, "structure_small_bridge.txt")
m.savetxt(out)
json.dump({"y": ys, "sec": secs, "args": [8, 2, [5] * 7, [8], [4, 3, 2.5, 2, 2, 2.5, 3, 4], 0.3, 0, -75000, "bridge", 1, None]},
          open(os.path.join(HERE, "structure_small_bridge.json"), "w"))
print(open(out, newline="").read()[:300])
