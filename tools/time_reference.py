#!/usr/bin/env python3
"""Times the REFERENCE's own CPU path (kupc25648/MOP-truss-MARL, read-only at /root/reference) in the BUILD CONTAINER and
records the result in profiles/r3/reference_cpu.json (median of 5 x 100 calls).

Like tests/golden/make_golden.py this script imports the reference (with the 6-line stand-in for the one spektral symbol that
is not installed) and is never imported by tests, bench.py or the product; the reference does not travel to the GPU box, so
bench.py's `cpu_baseline.reference_note` quotes the committed JSON.

What is timed, per truss (one process, one core, one env -- the reference has no batch dimension):
  gen_all        Model.restore(); Model.gen_all()                 FEM only               FEM_2Dtruss.py:434-459
  _game_modify   Game_research04._game_modify(parent, actions)    one full env step      truss2D_ENV.py:370-525
on the reference's small (16 nodes / 36 elements) and large (32 nodes / 76 elements) test trusses.

Usage:  python tools/time_reference.py        (about a minute)
"""
import contextlib
import io
import json
import os
import platform
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG      # scenario definitions + the spektral stand-in; importing it runs nothing

REPS, CALLS = 5, 100
CASES = [("small", "small_bridge"), ("large", "large_bridge")]


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def time_case(variant, sname):
    import numpy as np
    codedir = os.path.join(MG.REF, MG.VARIANTS[variant][0])
    os.chdir(codedir)                      # the reference reads ./section_data/*.csv relative to its code dir
    sys.path.insert(0, codedir)
    for m in ("truss2D_GEN", "truss2D_ENV", "FEM_2Dtruss"):      # each test copy has its own modules
        sys.modules.pop(m, None)
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        import truss2D_GEN as GEN
        import truss2D_ENV as ENVM
        sc = MG.SCENARIOS[sname]
        num_x = len(sc["span_x"]) + 1
        gm = GEN.gen_model(num_x, 2, sc["span_x"], sc["span_y"], sc["tar_y"], sc["dmin"], 0, sc["loady"], sc["ttype"], 1, None)
        game = ENVM.Game_research04(50, gm, 2)
        env = ENVM.ENV(game)
        env.reset()
        S0 = game._game_get_1_state()
    model = gm.model
    N, E = len(model.nodes), len(model.elements)
    rng = np.random.default_rng(7)
    acts = [(rng.random((N, 2)).astype(np.float32), rng.random((N, 3)).astype(np.float32)) for _ in range(CALLS)]
    pn, pe, pc = np.array(S0[8]), np.array(S0[9]), np.array(S0[10])
    out = {"nodes": N, "elements": E, "ndof": int(model.ndof)}
    with contextlib.redirect_stdout(sink):
        for _ in range(10):                                       # warm-up
            model.restore(); model.gen_all()
            game._game_modify(pn.copy(), pe.copy(), pc.copy(), [acts[0][0].copy(), acts[0][1].copy()])
        fem, step = [], []
        for _ in range(REPS):
            t0 = time.perf_counter()
            for _ in range(CALLS):
                model.restore()
                model.gen_all()
            fem.append((time.perf_counter() - t0) / CALLS)
        for _ in range(REPS):
            t0 = time.perf_counter()
            for k in range(CALLS):
                game._game_modify(pn.copy(), pe.copy(), pc.copy(), [acts[k][0].copy(), acts[k][1].copy()])
            step.append((time.perf_counter() - t0) / CALLS)
    sys.path.remove(codedir)
    for name, v in (("gen_all", fem), ("_game_modify", step)):
        out[name] = {"ms_per_call_median": statistics.median(v) * 1e3, "ms_per_call_min": min(v) * 1e3, "ms_per_call_max": max(v) * 1e3,
                     "calls_per_s_per_core": 1.0 / statistics.median(v)}
    return out


def main():
    os.environ.setdefault("MPLBACKEND", "Agg")
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(k, "1")
    MG._install_spektral_standin()
    import numpy as np
    res = {"what": "the reference's own CPU path timed in the build container (never on the GPU box: the reference does not travel)",
           "method": f"median of {REPS} repetitions x {CALLS} calls, one process, one core, one env per call; time.perf_counter",
           "cpu_model": _cpu_model(), "python": platform.python_version(), "numpy": np.__version__,
           "reference_numpy_pin": "1.23.5", "spektral": "degree_power stand-in (tests/golden/make_golden.py)",
           "generated_by": "tools/time_reference.py", "cases": {}}
    cwd = os.getcwd()
    for variant, sname in CASES:
        res["cases"][sname] = time_case(variant, sname)
        c = res["cases"][sname]
        print(f"{sname}: {c['nodes']} nodes / {c['elements']} elements: gen_all {c['gen_all']['ms_per_call_median']:.3f} ms, "
              f"_game_modify {c['_game_modify']['ms_per_call_median']:.3f} ms", flush=True)
    os.chdir(cwd)
    outp = os.path.join(ROOT, "profiles", "r3", "reference_cpu.json")
    os.makedirs(os.path.dirname(outp), exist_ok=True)
    with open(outp, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", outp)


if __name__ == "__main__":
    main()
