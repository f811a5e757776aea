"""torch.ops.truss_mi355.* -- the env step as PyTorch custom operators.

`csrc/truss_torch_ops.cpp` registers `step`, `rollout`, `obs`, `front`, `gcn_aggregate`, `gcn_aggregate_sparse`, `gcn_layer` and `gcn_level` with the dispatcher
(CPU / CUDA(=HIP) / Meta keys): tensors in, outputs mutated in place, launched on the stream the caller names,
capturable in a hipGraph, traceable.  The operators do no arithmetic; they call the C ABI of the native library that
`bind()` registered under an index -- the HIP product library, or (test-suite only) the CPU lane emulator.

There is no fallback: without `libtruss_torch_ops.so` (built by `make -C mop-truss-marl_amd/csrc`) loading fails.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib

OPS_LIB = os.path.join(os.path.dirname(_lib.DEFAULT_LIB), "libtruss_torch_ops.so")
_dll = None
_ids: dict = {}


def _load():
    global _dll
    if _dll is None:
        if not os.path.exists(OPS_LIB):
            raise _lib.TrussError(f"PyTorch operator library not found: {OPS_LIB}\nbuild it with `python -c 'import __graft_entry__ as g; "
                                  "g.build()'` (or `make -C mop-truss-marl_amd/csrc`).")
        torch.ops.load_library(OPS_LIB)              # runs the TORCH_LIBRARY registrations
        _dll = C.CDLL(OPS_LIB)                       # the same image: for truss_torch_bind
        _dll.truss_torch_bind.restype = C.c_int
        _dll.truss_torch_bind.argtypes = [C.c_int] + [C.c_void_p] * 10 + [C.c_int]
    return _dll


def bind(lib: "_lib.TrussLib") -> int:
    """Index under which the operators reach this native library's entry points."""
    if lib.path not in _ids:
        dll, idx = _load(), len(_ids)
        d = lib.dll
        addr = lambda f: C.cast(f, C.c_void_p)
        rc = dll.truss_torch_bind(idx, addr(d.truss_step), addr(d.truss_rollout), addr(d.truss_obs), addr(d.truss_front),
                                  addr(d.truss_gcn_aggregate), addr(d.truss_gcn_aggregate_sparse), addr(d.truss_gcn_layer),
                                  addr(d.truss_gcn_split_w), addr(d.truss_gcn_level), addr(d.truss_last_error),
                                  1 if lib.backend == "hip" else 0)
        if rc != 0:
            raise _lib.TrussError("truss_torch_bind failed (more than 8 native libraries bound?)")
        _ids[lib.path] = idx
    return _ids[lib.path]


def stream_of(device) -> int:
    """hipStream_t of torch's current stream on `device` as an integer (0 = the default stream / CPU)."""
    if device.type == "cuda":
        return torch.cuda.current_stream(device).cuda_stream
    return 0


def namespace():
    _load()
    return torch.ops.truss_mi355


def call(op, *args):
    """Run an operator; a failure reported by the native library / the operator's argument checks surfaces as
    TrussError (what the ctypes binding raised), anything else is passed through."""
    try:
        return op(*args)
    except RuntimeError as e:
        msg = str(e)
        if "truss_" in msg:
            raise _lib.TrussError(msg.split("\nException raised from")[0]) from None
        raise
