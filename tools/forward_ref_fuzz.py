#!/usr/bin/env python
"""Differential campaign for the forward oracle against the REFERENCE ITSELF (hf_hypernet/modeling_hypernet.py:156-267, imported
from /root/reference in the build container, RobertaModel on eager attention; the 48 committed fwd_*.npz fixtures are cases
of this kind).  Random widths, head counts, 1-4 layers, 1-24 surface positions, every flag, rows with all-pad / pad-in-the-
middle / fallback-id patterns, fp32 and fp16 source tables: oracle/hypernet_ref.forward and .forward_levers (the algebra the
HIP path executes) against the reference's outputs.  With tools/forward_fuzz.py (HIP vs oracle, on the GPU) this ties the
device forward to the reference beyond the fixtures.

    python tools/forward_ref_fuzz.py --seeds 0 300
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs=2, default=[0, 100])
    args = ap.parse_args()
    from make_golden import _reference_forward, tiny_ids
    from oracle import hypernet_ref
    from tests import util
    from zett_amd import synth
    base = dict(synth.workload("tiny")[0])
    t0 = time.time()
    worst = {"forward": 0.0, "forward_levers": 0.0}
    failures = []
    n = 0
    for seed in range(*args.seeds):
        rng = np.random.default_rng(660000 + seed)
        h = int(rng.choice([64, 128, 192, 256]))
        heads = int(rng.choice([x for x in (1, 2, 3, 4, 6, 8, 12) if h % x == 0]))
        seq = int(rng.choice([1, 2, 3, 7, 8, 15, 24]))
        cfg = dict(base, n_embd=int(rng.choice([64, 128, 192])), hn_hidden_size=h, hn_intermediate_size=int(rng.choice([128, 192, 384])),
                   hn_num_attention_heads=heads, hn_n_layers=int(rng.choice([1, 2, 3, 4])), hn_surface_maxlen=seq,
                   separate_out_embeddings=bool(rng.integers(2)), hn_embed_lang_id=bool(rng.integers(2)), hn_rescale_embeddings=bool(rng.integers(2)),
                   hn_predict_bias=bool(rng.integers(2)), hn_single_head=bool(rng.integers(2)))
        rows = int(rng.choice([8, 24, 60]))
        src_dtype = str(rng.choice(["float32", "float16"]))
        w = synth.make_weights(cfg, seed)
        src = synth.make_source_embeddings(cfg, seed, dtype=src_dtype)
        ids = tiny_ids(cfg, rows, seq, seed) if rows >= 8 else synth.make_surface_forms(cfg, rows, seed=seed, seq=seq)
        lang = int(rng.integers(0, cfg["n_langs"])) if cfg["hn_embed_lang_id"] else None
        want = _reference_forward(cfg, w, ids, src.astype(np.float32) if src_dtype != "float32" else src, lang)
        keep = ~util.all_pad_rows(cfg, ids)
        n += 1
        for fn in ("forward", "forward_levers"):
            got = getattr(hypernet_ref, fn)(w, cfg, ids, src, lang)
            for g, t, name in zip(got, want, ("pred_in", "pred_out", "bias")):
                if t is None:
                    if g is not None:
                        failures.append({"seed": seed, "fn": fn, "what": f"{name}: the reference returns None"})
                    continue
                g, t = np.asarray(g)[keep], np.asarray(t)[keep]
                scale = np.maximum(1.0, np.abs(t).max(axis=-1, keepdims=True)) if t.ndim > 1 else np.maximum(1.0, np.abs(t))
                err = float((np.abs(g.astype(np.float64) - t.astype(np.float64)) / scale).max()) if t.size else 0.0
                worst[fn] = max(worst[fn], err)
                if not np.isfinite(g).all() or err > util.F32_ABS:
                    failures.append({"seed": seed, "fn": fn, "what": f"{name}: max scaled abs err {err:.3e}", "cfg": {k: cfg[k] for k in cfg if k.startswith("hn_") or k == "n_embd"}})
    print(json.dumps({"seeds": list(args.seeds), "cases": n, "tolerance": util.F32_ABS, "worst_scaled_abs_err": worst, "failures": failures[:10], "n_failures": len(failures),
                      "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
