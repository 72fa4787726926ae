#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE.

Runs only in the build container (it imports /root/reference, which does not
exist on the GPU box).  Nothing of the reference is copied: the fixtures hold
inputs (or the seeds that regenerate them) and the reference's outputs.

    python tests/golden/make_golden.py [forward] [big] [retok]

forward  ->  fwd_*.npz   outputs of hf_hypernet.modeling_hypernet.ZettHypernet
             (reference hf_hypernet/modeling_hypernet.py:156-267) with the inner
             RobertaModel forced to eager attention, weights and
             inputs from zett_amd.synth seeds.
big      ->  fwd_big_*.npz  the same reference forward on 640 rows of the four real shapes; outputs kept for a
             32-row sample (the batch is what makes the GPU path take its 256x256 GEMM tiles).
retok    ->  retok_*.json outputs of zett.utils.get_surface_form_matrix
             (reference zett/utils.py:651-689) on synthetic hn tokenizers, with
             jax/flax/optax stubbed by MagicMock so the module imports.
"""
from __future__ import annotations

import itertools
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, REPO)

from zett_amd import synth  # noqa: E402


def _roberta_dir():
    d = tempfile.mkdtemp(prefix="rb_cfg_")
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump({"model_type": "roberta", "max_position_embeddings": 514, "type_vocab_size": 1,
                   "layer_norm_eps": 1e-5, "hidden_act": "gelu", "initializer_range": 0.02,
                   "vocab_size": 50265, "pad_token_id": 1, "bos_token_id": 0, "eos_token_id": 2}, f)
    return d


def _reference_forward(cfg, weights, ids, src, lang_index):
    import torch
    sys.path.insert(0, REFERENCE)
    from hf_hypernet.configuration_hypernet import ZettHypernetConfig
    from hf_hypernet.modeling_hypernet import ZettHypernet

    rcfg = ZettHypernetConfig(**dict(cfg, hn_model_name_or_path=_roberta_dir()))
    model = ZettHypernet(rcfg).eval()
    model.model.config._attn_implementation = "eager"
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    with torch.no_grad():
        out = model(torch.from_numpy(ids.astype(np.int64)),
                    source_embeddings=torch.from_numpy(src),
                    lang_index=None if lang_index is None else torch.tensor(lang_index))
    return [None if o is None else o.float().numpy() for o in out]


def _save(name, cfg, seed, ids, src_dtype, lang_index, out, extra=None):
    path = os.path.join(HERE, name + ".npz")
    payload = dict(cfg_json=np.array(json.dumps(cfg)), seed=np.int64(seed), ids=ids.astype(np.int32),
                   src_dtype=np.array(src_dtype), lang_index=np.int64(-1 if lang_index is None else lang_index),
                   pred_in=out[0], bias=out[2])
    if out[1] is not None:
        payload["pred_out"] = out[1]
    if extra:
        payload.update(extra)
    np.savez_compressed(path, **payload)
    print("wrote", os.path.relpath(path, REPO), {k: getattr(v, "shape", None) for k, v in payload.items()})


def tiny_ids(cfg, n, seq, seed):
    """Surface forms that hit every edge: all-pad rows, pad in the middle,
    fallback ids (>= V0), id == V0-1, full-length rows, position 0 == pad."""
    ids = synth.make_surface_forms(cfg, n, seed=seed, seq=seq)
    v0, pad = cfg["original_vocab_size"], cfg["pad_token_id"]
    x = max(cfg.get("hn_n_extra_tokens", 0), 1)
    ids[0, :] = pad                                     # all-pad row
    ids[1, :] = np.arange(3, 3 + seq)                   # full-length row
    ids[2, 0] = v0 - 1
    ids[3, 0] = v0                                      # first fallback id
    ids[4, :min(seq, 2)] = v0 + x - 1                   # last fallback id
    if seq >= 3:
        ids[5, :3] = (7, pad, 9)                        # pad in the middle
        ids[6, :3] = (pad, 11, 12)                      # position 0 is pad but row is not empty
    return ids


def gen_forward():
    base, _, _, _ = synth.workload("tiny")
    flags = ("separate_out_embeddings", "hn_embed_lang_id", "hn_rescale_embeddings",
             "hn_predict_bias", "hn_single_head")
    n = 24
    for bits in itertools.product((False, True), repeat=len(flags)):
        cfg = dict(base, **dict(zip(flags, bits)))
        tag = "".join("1" if b else "0" for b in bits)
        seed = int(tag, 2) + 100
        w = synth.make_weights(cfg, seed)
        src = synth.make_source_embeddings(cfg, seed)
        ids = tiny_ids(cfg, n, 7, seed)
        lang = 3 if cfg["hn_embed_lang_id"] else None
        out = _reference_forward(cfg, w, ids, src, lang)
        _save(f"fwd_tiny_{tag}", cfg, seed, ids, "float32", lang, out)

    # sequence-length variants (L = 1 identity warm-up, L = 15 long config, L = 24 generic)
    for seq in (1, 15, 24):
        cfg = dict(base, hn_surface_maxlen=seq)
        seed = 200 + seq
        w = synth.make_weights(cfg, seed)
        src = synth.make_source_embeddings(cfg, seed)
        ids = tiny_ids(cfg, n, seq, seed)
        out = _reference_forward(cfg, w, ids, src, 1)
        _save(f"fwd_tiny_L{seq}", cfg, seed, ids, "float32", 1, out)

    # fp16 / bf16 source embeddings (reference upcasts at the first fp32 op)
    for dt in ("float16",):
        cfg = dict(base)
        seed = 300
        w = synth.make_weights(cfg, seed)
        src = synth.make_source_embeddings(cfg, seed, dtype=dt)
        ids = tiny_ids(cfg, n, 7, seed)
        out = _reference_forward(cfg, w, ids, src, 0)
        _save(f"fwd_tiny_src_{dt}", cfg, seed, ids, dt, 0, out)

    # real shapes, a few sampled rows each
    for name, rows in (("xlmr_gpt2", 64), ("tinyllama_neox", 32), ("mistral_gpt2_32k", 16), ("llama3_256k", 16)):
        cfg, _, src_dtype, hist = synth.workload(name)
        seed = 7
        w = synth.make_weights(cfg, seed)
        src = synth.make_source_embeddings(cfg, seed, dtype=src_dtype)
        ids = synth.make_surface_forms(cfg, rows, seed=seed, hist=hist, n_special=1)
        lang = 5 if cfg.get("hn_embed_lang_id") else None
        out = _reference_forward(cfg, w, ids, src, lang)
        _save(f"fwd_real_{name}", cfg, seed, ids, src_dtype, lang, out)
        del w, src


def gen_forward_big():
    """Real shapes at a batch large enough that the GPU path takes its 256x256 GEMM tiles (M = packed positions >= 1 400):
    640 rows go through the REFERENCE, the fixture keeps the inputs of all of them and the reference's outputs for a
    32-row sample (`sample` holds the row indices) — the tiles that carry the benchmark meet reference outputs directly."""
    for name in ("xlmr_gpt2", "tinyllama_neox", "mistral_gpt2_32k", "llama3_256k"):
        cfg, _, src_dtype, hist = synth.workload(name)
        seed, rows = 11, 640
        w = synth.make_weights(cfg, seed)
        src = synth.make_source_embeddings(cfg, seed, dtype=src_dtype)
        ids = synth.make_surface_forms(cfg, rows, seed=seed, hist=hist, n_special=2)
        lang = 7 if cfg.get("hn_embed_lang_id") else None
        out = _reference_forward(cfg, w, ids, src, lang)
        sample = np.sort(np.concatenate([[0, 2, rows - 1], np.random.default_rng(seed).choice(np.arange(3, rows - 1), 29, replace=False)]))
        out = [None if o is None else o[sample] for o in out]
        _save(f"fwd_big_{name}", cfg, seed, ids, src_dtype, lang, out, extra=dict(sample=sample.astype(np.int32)))
        del w, src


if __name__ == "__main__":
    what = sys.argv[1:] or ["forward", "retok"]
    if "big" in what:
        gen_forward_big()
    if "forward" in what:
        gen_forward()
    if "retok" in what:
        from make_golden_retok import gen_retok  # noqa: E402
        gen_retok()
