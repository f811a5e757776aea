"""ctypes binding of the C ABI declared in include/truss_mi355.h.

The product library is `mop-truss-marl_amd/csrc/libtruss_mi355.so` (hand-written HIP for gfx950,
built by `__graft_entry__.build()` / `csrc/Makefile`).  There is NO CPU fallback: if the library is
missing, or it is not the HIP build, loading fails loudly.  (The build's own test-suite can point
`load(path)` at the CPU lane emulator under tests/emu to debug kernel indexing without a GPU; that
library identifies itself as backend "emu" and is never picked up implicitly.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(os.path.dirname(_HERE), "csrc", "libtruss_mi355.so")

TRUSS_ABI_VERSION = 3
F_NO_DECODE = 0x1
F_CLAMP_INPLACE = 0x2
F_EMIT_OBS = 0x4
STATUS_NOT_SPD = 1        # status[] bits (include/truss_mi355.h)
STATUS_OBS_TIMEOUT = 2
NPARAM = 8
P_YMAX, P_DMIN, P_MAXDEF, P_LOADX, P_LOADY, P_INTOBJ1, P_INTOBJ2, P_ISROOF = range(8)

_vp = C.c_void_p


class StepArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_size_t), ("n_envs", C.c_int32), ("flags", C.c_uint32),
        ("x", _vp), ("y_in", _vp), ("sec_in", _vp), ("max_up_in", _vp), ("max_down_in", _vp),
        ("a_geo", _vp), ("a_topo", _vp), ("coin", _vp), ("target", _vp), ("env_params", _vp),
        ("y_out", _vp), ("sec_out", _vp), ("max_up_out", _vp), ("max_down_out", _vp),
        ("disp", _vp), ("q0", _vp), ("sr", _vp), ("comp", _vp), ("point", _vp), ("obj", _vp),
        ("disp_f64", _vp), ("q0_f64", _vp), ("energy", _vp), ("reactions", _vp), ("status", _vp),
        ("x_n", _vp), ("A_s", _vp), ("A_n_ts", _vp), ("A_n_cs", _vp), ("nN_x_n", _vp), ("nN_x_e", _vp),
    ]


class ObsArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_size_t), ("n_envs", C.c_int32), ("flags", C.c_uint32),
        ("x", _vp), ("y", _vp), ("sec", _vp), ("max_up", _vp), ("max_down", _vp), ("target", _vp),
        ("disp", _vp), ("q0", _vp), ("sr", _vp), ("comp", _vp), ("env_params", _vp),
        ("x_n", _vp), ("A_s", _vp), ("A_n_ts", _vp), ("A_n_cs", _vp), ("nN_x_n", _vp), ("nN_x_e", _vp),
    ]


class FrontArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_size_t), ("n_envs", C.c_int32), ("max_points", C.c_int32), ("max_front", C.c_int32),
        ("flags", C.c_uint32), ("points", _vp), ("n_points", _vp), ("ref_points", _vp), ("front_idx", _vp),
        ("n_front", _vp), ("hv_front", _vp), ("hv_all", _vp), ("metrics", _vp),
    ]


F_FRONT_TRUNCATE = 0x1
FRONT_MAXP = 64


class TrussError(RuntimeError):
    pass


class TrussLib:
    """Loaded shared library + typed entry points."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise TrussError(
                f"HIP extension not found: {path}\n"
                "build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C mop-truss-marl_amd/csrc`). There is no CPU fallback.")
        self.path = path
        self.dll = C.CDLL(path)
        d = self.dll
        d.truss_abi_version.restype = C.c_int
        d.truss_last_error.restype = C.c_char_p
        d.truss_backend.restype = C.c_char_p
        d.truss_topo_create.restype = C.c_int
        d.truss_topo_create.argtypes = [C.POINTER(_vp), C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp,
                                        C.c_int32, _vp, C.c_int32, _vp, C.c_double, C.c_double, _vp]
        d.truss_topo_destroy.argtypes = [_vp]
        d.truss_topo_dofs.restype = C.c_int
        d.truss_topo_dofs.argtypes = [_vp, _vp, _vp]
        d.truss_topo_solver_info.restype = C.c_int
        d.truss_topo_solver_info.argtypes = [_vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        d.truss_topo_fused_obs.restype = C.c_int
        d.truss_topo_fused_obs.argtypes = [_vp]
        d.truss_topo_persistent_rollout.restype = C.c_int
        d.truss_topo_persistent_rollout.argtypes = [_vp]
        d.truss_step.restype = C.c_int
        d.truss_step.argtypes = [_vp, C.POINTER(StepArgs), _vp]
        d.truss_rollout.restype = C.c_int
        d.truss_rollout.argtypes = [_vp, C.POINTER(StepArgs), C.c_int32, C.c_int32, _vp]
        if hasattr(d, "truss_obs"):
            d.truss_obs.restype = C.c_int
            d.truss_obs.argtypes = [_vp, C.POINTER(ObsArgs), _vp]
        d.truss_gcn_aggregate.restype = C.c_int
        d.truss_gcn_aggregate.argtypes = [_vp, C.c_int64, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]
        d.truss_gcn_layer.restype = C.c_int
        d.truss_gcn_layer.argtypes = [_vp, _vp]
        d.truss_gcn_level.restype = C.c_int
        d.truss_gcn_level.argtypes = [_vp, C.c_int32, _vp, _vp]
        d.truss_gcn_split_w.restype = C.c_int
        d.truss_gcn_split_w.argtypes = [_vp, C.c_int32, C.c_int32, _vp, _vp]
        d.truss_gcn_aggregate_sparse.restype = C.c_int
        d.truss_gcn_aggregate_sparse.argtypes = [_vp, C.c_int64, _vp, C.c_int32, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]
        d.truss_front.restype = C.c_int
        d.truss_front.argtypes = [C.POINTER(FrontArgs), _vp]
        if d.truss_abi_version() != TRUSS_ABI_VERSION:
            raise TrussError(f"{path}: ABI version {d.truss_abi_version()} != {TRUSS_ABI_VERSION}")
        self.backend = d.truss_backend().decode()

    def check(self, rc: int, what: str):
        if rc < 0:
            raise TrussError(f"{what} failed ({rc}): {self.dll.truss_last_error().decode()}")
        return rc


_cache: dict = {}


def load(path: str | None = None) -> TrussLib:
    """Load the HIP library (default path) or an explicitly named build of the same ABI."""
    p = os.path.abspath(path or DEFAULT_LIB)
    if p not in _cache:
        lib = TrussLib(p)
        if path is None and lib.backend != "hip":
            raise TrussError(f"{p} is not the HIP build (backend={lib.backend})")
        _cache[p] = lib
    return _cache[p]
