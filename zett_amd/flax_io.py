"""Read (and write) ``flax_model.msgpack`` hypernetwork checkpoints without jax / flax.

Hub checkpoints of the reference ship their canonical weights as Flax msgpack
(scripts/transfer.py:145-151 restores it; scripts/convert_to_pt.py:35-46 turns it into the PyTorch
layout).  This module does both steps with ``msgpack`` + numpy only:

* the wire format of ``flax.serialization`` — msgpack maps with ExtType 1 for ndarrays
  (``msgpack((shape, dtype name, C-order bytes))``), ExtType 3 for numpy scalars, and the
  ``__msgpack_chunked_array__`` wrapper flax uses for arrays above 2**30 bytes, whose ``shape`` and ``chunks``
  are tuples written as dicts ``{"0": ..., "1": ...}`` (flax's ``_tuple_to_dict``; a plain list is accepted too);
* the name map of convert_to_pt.py: ``layers_N`` -> ``N``; ``kernel`` [in,out] -> ``weight`` [out,in];
  ``scale`` / ``embedding`` -> ``weight``; ``model.embeddings.lang_embedding`` -> ``lang_embeddings``;
  the 1-row Flax ``word_embeddings`` table (never read by the forward) is skipped.

No Flax installation exists in the build image, so the reader is exercised on (a) files produced by the
writer below (round trip + name map against the PyTorch state dict) and (b) a checkpoint whose bytes
tests/flax_fixture.py assembles by hand in flax's layout, with the Flax-side parameter names spelled out
(tests/test_flax_io.py); it has not been run on a file written by flax itself.
"""
from __future__ import annotations

import re
from typing import Dict

import msgpack
import numpy as np

_CHUNK_KEY = "__msgpack_chunked_array__"
_MAX_CHUNK = 2 ** 30


def _bf16_to_f32(buf: bytes, shape) -> np.ndarray:
    u = np.frombuffer(buf, dtype=np.uint16).astype(np.uint32) << 16
    return u.view(np.float32).reshape(shape)


def _ext_hook(code: int, data: bytes):
    if code == 1:                                           # ndarray
        shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
        if dtype_name == "bfloat16":
            return _bf16_to_f32(buf, tuple(shape))
        return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(tuple(shape)).copy()
    if code == 3:                                           # numpy scalar
        dtype_name, buf = msgpack.unpackb(data, raw=False)
        return np.frombuffer(buf, dtype=np.dtype(dtype_name))[0]
    return msgpack.ExtType(code, data)


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get(_CHUNK_KEY):
            flat = np.concatenate([np.asarray(tree["chunks"][str(i)]).reshape(-1) for i in range(len(tree["chunks"]))])
            shape = tree["shape"]
            if isinstance(shape, dict):                      # flax: _tuple_to_dict(arr.shape)
                shape = [shape[str(i)] for i in range(len(shape))]
            return flat.reshape(tuple(int(d) for d in shape))
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def read_msgpack(path: str) -> dict:
    """The nested parameter dict of a flax msgpack file (numpy arrays at the leaves)."""
    with open(path, "rb") as f:
        tree = msgpack.unpackb(f.read(), ext_hook=_ext_hook, raw=False, strict_map_key=False)
    return _unchunk(tree)


def _pack_array(a: np.ndarray):
    a = np.ascontiguousarray(a)
    if a.nbytes > _MAX_CHUNK:
        flat = a.reshape(-1)
        step = _MAX_CHUNK // a.dtype.itemsize
        return {_CHUNK_KEY: True, "shape": {str(i): int(d) for i, d in enumerate(a.shape)},
                "chunks": {str(i): _pack_array(flat[o:o + step]) for i, o in enumerate(range(0, flat.size, step))}}
    return msgpack.ExtType(1, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes("C")), use_bin_type=True))


def write_msgpack(tree: dict, path: str) -> None:
    def conv(t):
        return {k: conv(v) for k, v in t.items()} if isinstance(t, dict) else _pack_array(np.asarray(t))
    with open(path, "wb") as f:
        f.write(msgpack.packb(conv(tree), use_bin_type=True))


def _flatten(tree: dict, prefix=()):
    for k, v in tree.items():
        if isinstance(v, dict):
            yield from _flatten(v, prefix + (k,))
        else:
            yield prefix + (k,), v


def flax_to_torch_state_dict(tree: dict) -> Dict[str, np.ndarray]:
    """convert_to_pt.py:35-46 — Flax parameter tree -> PyTorch ``state_dict`` names and layouts."""
    if set(tree.keys()) == {"params"}:
        tree = tree["params"]
    out: Dict[str, np.ndarray] = {}
    for key, value in _flatten(tree):
        key = tuple(re.sub(r"^layers_(\d+)$", r"\1", part) for part in key)
        value = np.asarray(value)
        leaf = key[-1]
        if key[:3] == ("model", "embeddings", "lang_embedding"):
            out["lang_embeddings.weight"] = value
            continue
        if key[:3] == ("model", "embeddings", "word_embeddings"):
            continue                                        # Flax allocates 1 row; the forward never reads it
        if leaf == "kernel":
            key, value = key[:-1] + ("weight",), value.T if value.ndim == 2 else value
        elif leaf in ("scale", "embedding"):
            key = key[:-1] + ("weight",)
        out[".".join(key)] = np.ascontiguousarray(value, dtype=np.float32)
    return out


def torch_state_dict_to_flax(state: Dict[str, np.ndarray]) -> dict:
    """Inverse of :func:`flax_to_torch_state_dict` (used by the tests and to export checkpoints)."""
    tree: dict = {}
    for name, value in state.items():
        value = np.asarray(value)
        parts = name.split(".")
        if name == "lang_embeddings.weight":
            parts = ["model", "embeddings", "lang_embedding", "embedding"]
        elif parts[-1] == "weight":
            parent = parts[-2]
            if "LayerNorm" in parent or parent == "ln":
                parts[-1] = "scale"
            elif "embeddings" in parent or parent == "fallback_embeddings":
                parts[-1] = "embedding"
            else:
                parts[-1] = "kernel"
                value = value.T
        # Sequential children are called layers_N in Flax
        parts = [f"layers_{p}" if (p.isdigit() and i > 0 and parts[i - 1] in ("input_projection", "output_projection", "output_projection_out")) else p
                 for i, p in enumerate(parts)]
        node = tree
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = np.ascontiguousarray(value)
    return tree


def load_flax_checkpoint(path: str) -> Dict[str, np.ndarray]:
    """``flax_model.msgpack`` -> dict ready for ``ZettHypernet.load_state_dict`` (strict=False for word_embeddings)."""
    return flax_to_torch_state_dict(read_msgpack(path))
