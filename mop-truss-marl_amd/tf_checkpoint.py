"""Reader for TensorFlow object-based checkpoints (`<prefix>.index` + `<prefix>.data-00000-of-00001`),
SURVEY.md §8 row f-2: the reference publishes its trained actors in this format
(model/2000pickle_base/Agent{1,2,3}_Actor_pickle.*, written by `Model.save_weights`,
master_DDPG_truss2D_MO.py:710-733) and TensorFlow is not available here.

Nothing in the files is executed: the `.index` is a LevelDB-format sorted string table (uncompressed
blocks, prefix-compressed keys) whose values are `BundleEntryProto` messages (dtype, shape, shard, offset,
size, crc32c); the `.data` shard is the raw little-endian tensor bytes.  Both are parsed by hand (varints,
protobuf wire format) and every tensor is checked against its masked CRC-32C.
"""
from __future__ import annotations

import struct

import numpy as np

_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}
_TABLE_MAGIC = 0xDB4775248B80FB57


def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """protobuf wire format -> list of (field_number, wire_type, value)"""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        out.append((fn, wt, v))
    return out


def _block(data, offset, size):
    """entries of one table block: [(key, value)]"""
    blk = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError("compressed table blocks are not supported (TensorFlow writes checkpoints uncompressed)")
    n_restarts = struct.unpack("<I", blk[-4:])[0]
    end = len(blk) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        out.append((key, blk[pos:pos + vlen]))
        pos += vlen
    return out


def _handle(buf, pos):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def read_index(path):
    """{tensor name: dict(dtype, shape, shard_id, offset, size, crc32c)} of `<prefix>.index`."""
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != _TABLE_MAGIC:
        raise ValueError(f"{path}: not a TensorFlow checkpoint index (bad table magic)")
    footer = data[-48:]
    _, _, p = _handle(footer, 0)                 # metaindex
    ioff, isize, _ = _handle(footer, p)          # index block
    entries = {}
    for _, hv in _block(data, ioff, isize):
        boff, bsize, _ = _handle(hv, 0)
        for key, val in _block(data, boff, bsize):
            if key == b"":
                continue                          # BundleHeaderProto
            e = dict(dtype=None, shape=(), shard_id=0, offset=0, size=0, crc32c=None)
            for fn, wt, v in _fields(val):
                if fn == 1:
                    e["dtype"] = v
                elif fn == 2:
                    dims = []
                    for f2, _, v2 in _fields(v):
                        if f2 == 2:               # TensorShapeProto.Dim
                            sz = 0
                            for f3, _, v3 in _fields(v2):
                                if f3 == 1:
                                    sz = v3
                            dims.append(sz)
                    e["shape"] = tuple(dims)
                elif fn == 3:
                    e["shard_id"] = v
                elif fn == 4:
                    e["offset"] = v
                elif fn == 5:
                    e["size"] = v
                elif fn == 6:
                    e["crc32c"] = struct.unpack("<I", v)[0]
            entries[key.decode()] = e
    return entries


def _crc32c_table():
    tab = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        tab[i] = c
    return tab


_CRC_TAB = _crc32c_table()


def crc32c(data: bytes) -> int:
    tab = _CRC_TAB
    c = 0xFFFFFFFF
    for b in data:
        c = int(tab[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def load_variables(prefix, verify=True):
    """{name: ndarray} of every numeric variable of the checkpoint `<prefix>` (names without the
    '/.ATTRIBUTES/VARIABLE_VALUE' suffix).  The object-graph entry and string tensors are skipped."""
    idx = read_index(prefix + ".index")
    blob = open(prefix + ".data-00000-of-00001", "rb").read()
    out = {}
    for name, e in idx.items():
        if e["dtype"] not in _DTYPES or e["shard_id"] != 0 or not name.endswith("/.ATTRIBUTES/VARIABLE_VALUE"):
            continue
        raw = blob[e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise ValueError(f"{prefix}: data shard too short for {name} (a truncated / missing blob)")
        if verify and e["crc32c"] is not None and masked_crc32c(raw) != e["crc32c"]:
            raise ValueError(f"{prefix}: CRC mismatch in {name}")
        arr = np.frombuffer(raw, dtype=_DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
        out[name[: -len("/.ATTRIBUTES/VARIABLE_VALUE")]] = arr
    return out


def load_gcn_actor(actor, prefix, verify=True):
    """Copy a published TensorFlow actor checkpoint into a PyTorch `truss2D_RL.multimodes_actor`
    (`Model.load_weights(prefix).expect_partial()`, master_DDPG_truss2D_MO.py:721-729).  The TF variables are
    named after the same layer attributes (gcn_l1_1 ... gcn_l4_2); a TF kernel is [in, out], torch's
    Linear weight is [out, in].  Optimizer slots are ignored (expect_partial).  Returns the number of
    parameters copied."""
    import torch
    from torch import nn
    var = load_variables(prefix, verify=verify)
    names = sorted({k.split("/")[0] for k in var if k.startswith("gcn_")})
    n = 0
    for name in names:
        layer = getattr(actor, name)                      # AttributeError = not this architecture
        k, b = var[name + "/kernel"], var[name + "/bias"]
        lin = nn.Linear(k.shape[0], k.shape[1], bias=False)
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(np.ascontiguousarray(k.T)))
            dev = layer.bias.device
            layer.lin = lin.to(dev)
            layer.bias.copy_(torch.from_numpy(b).to(dev))
        n += k.size + b.size
    missing = [m for m, _ in actor.named_children() if m.startswith("gcn_") and m not in names]
    if missing:
        raise ValueError(f"{prefix}: no variables for layers {missing}")
    return n
