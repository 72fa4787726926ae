"""A ``flax_model.msgpack`` assembled BY HAND in flax's on-disk layout — independent of zett_amd/flax_io.py's writer.

What flax.serialization.msgpack_serialize puts on disk (restated here from its published format; flax itself is not
installed in this image):

  * the parameter pytree as nested msgpack maps with str keys;
  * every ndarray leaf as msgpack ExtType 1 whose payload is msgpack((shape tuple, dtype name, C-order bytes));
    bfloat16 leaves carry the dtype name "bfloat16" and two bytes per element;
  * an array above 2**30 bytes as {"__msgpack_chunked_array__": True, "shape": {"0": d0, "1": d1, ...},
    "chunks": {"0": <ExtType 1 of a flat piece>, "1": ...}}  (tuples go through _tuple_to_dict).

The parameter NAMES are the Flax side's (zett/model/__init__.py + transformers' FlaxRobertaModule, the tree
scripts/convert_to_pt.py:35-46 starts from), written out literally below: Dense -> kernel [in, out] / bias,
LayerNorm -> scale / bias, Embed -> embedding, nn.Sequential children -> layers_N, the language table under
model/embeddings/lang_embedding, Rescaler -> w / b.
"""
import os

import msgpack
import numpy as np


def _nd(a, dtype_name=None):
    a = np.ascontiguousarray(a)
    if dtype_name == "bfloat16":        # truncate-free: callers pass values that are exactly representable
        payload = (a.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16).tobytes()
        return msgpack.ExtType(1, msgpack.packb((tuple(a.shape), "bfloat16", payload), use_bin_type=True))
    return msgpack.ExtType(1, msgpack.packb((tuple(a.shape), a.dtype.name, a.tobytes("C")), use_bin_type=True))


def _chunked(a, pieces):
    flat = np.ascontiguousarray(a).reshape(-1)
    step = -(-flat.size // pieces)
    return {"__msgpack_chunked_array__": True,
            "shape": {str(i): int(d) for i, d in enumerate(a.shape)},
            "chunks": {str(i): _nd(flat[o:o + step]) for i, o in enumerate(range(0, flat.size, step))}}


def bf16_round(a):
    """fp32 values rounded (to nearest even) to what bfloat16 can hold, still as fp32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def write_flax_checkpoint(path, cfg, w):
    """`w`: PyTorch-layout state dict (numpy).  Returns the state dict a loader must reproduce (the two arrays stored
    in bfloat16 are rounded accordingly)."""
    def dense(prefix):      # torch Linear [out, in] -> Flax kernel [in, out]
        return {"kernel": _nd(w[prefix + ".weight"].T), "bias": _nd(w[prefix + ".bias"])}

    def ln(prefix):
        return {"scale": _nd(w[prefix + ".weight"]), "bias": _nd(w[prefix + ".bias"])}

    def projector(prefix):
        return {"dense1": dense(prefix + ".dense1"), "dense2": dense(prefix + ".dense2"), "ln": ln(prefix + ".ln")}

    expected = {k: np.array(v, dtype=np.float32) for k, v in w.items() if k != "model.embeddings.word_embeddings.weight"}
    pos = bf16_round(w["model.embeddings.position_embeddings.weight"])
    expected["model.embeddings.position_embeddings.weight"] = pos
    layers = {}
    for l in range(int(cfg.get("hn_n_layers", 3))):
        p = f"model.encoder.layer.{l}."
        layers[str(l)] = {
            "attention": {
                "self": {"query": dense(p + "attention.self.query"), "key": dense(p + "attention.self.key"),
                         "value": dense(p + "attention.self.value")},
                "output": {"dense": dense(p + "attention.output.dense"), "LayerNorm": ln(p + "attention.output.LayerNorm")},
            },
            "intermediate": {"dense": dense(p + "intermediate.dense")},
            "output": {"dense": dense(p + "output.dense"), "LayerNorm": ln(p + "output.LayerNorm")},
        }
    embeddings = {
        "word_embeddings": {"embedding": _nd(np.zeros((1, w["model.embeddings.LayerNorm.weight"].shape[0]), dtype=np.float32))},   # Flax allocates one row
        "position_embeddings": {"embedding": _nd(pos, "bfloat16")},
        "token_type_embeddings": {"embedding": _nd(w["model.embeddings.token_type_embeddings.weight"])},
        "LayerNorm": ln("model.embeddings.LayerNorm"),
    }
    if cfg.get("hn_embed_lang_id"):
        embeddings["lang_embedding"] = {"embedding": _nd(w["lang_embeddings.weight"])}
    tree = {
        "model": {"embeddings": embeddings, "encoder": {"layer": layers}},
        "fallback_embeddings": {"embedding": _chunked(w["fallback_embeddings.weight"], 3)},
        "input_projection": {"layers_0": dense("input_projection.0"), "layers_1": projector("input_projection.1")},
        "output_projection": {"layers_0": projector("output_projection.0"), "layers_1": dense("output_projection.1")},
    }
    if cfg.get("separate_out_embeddings") and not cfg.get("hn_single_head"):
        tree["output_projection_out"] = {"layers_0": projector("output_projection_out.0"), "layers_1": dense("output_projection_out.1")}
    if cfg.get("hn_rescale_embeddings"):
        tree["in_scaler"] = {"w": _nd(w["in_scaler.w"]), "b": _nd(w["in_scaler.b"])}
        tree["scaler"] = {"w": _nd(w["scaler.w"]), "b": _nd(w["scaler.b"])}
        if cfg.get("separate_out_embeddings"):
            tree["out_scaler"] = {"w": _nd(w["out_scaler.w"]), "b": _nd(w["out_scaler.b"])}
    if cfg.get("hn_predict_bias"):
        tree["bias_projection"] = dense("bias_projection")
    with open(os.path.join(path, "flax_model.msgpack"), "wb") as f:
        f.write(msgpack.packb(tree, use_bin_type=True, strict_types=False))
    return expected
