#!/usr/bin/env python
"""Fuzz campaign for the training use of the path (N4): zett_amd/autograd.py on the GPU against float64 torch autograd of
tests/torch_port.py (the oracle's math in torch) on random widths / heads / layers / positions / flags / row counts, both
schedules (packed, dense) and the three contraction arithmetics.  Every parameter's gradient is compared relative to its own
norm (with a floor at 1e-6 of the largest gradient: some gradients are round-off around an exact zero, e.g. key biases).

    python tools/train_fuzz.py --seeds 0 200 [--budget-s 600]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from tests import torch_port  # noqa: E402
from zett_amd import synth  # noqa: E402

LIMITS = {"f32": 3e-4, "f16": 1.5e-2, "bf16": 8e-2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs=2, default=[0, 100])
    ap.add_argument("--budget-s", type=float, default=1e9)
    args = ap.parse_args()
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    base = dict(synth.workload("tiny")[0])
    dev = "cuda:0"
    t0 = time.time()
    done, failures, worst = 0, [], {"f32": 0.0, "f16": 0.0, "bf16": 0.0}
    seed = args.seeds[0]
    for seed in range(*args.seeds):
        if time.time() - t0 > args.budget_s:
            break
        rng = np.random.default_rng(330000 + seed)
        h = int(rng.choice([64, 128, 192, 256, 320]))
        heads = int(rng.choice([x for x in (1, 2, 4, 8) if h % x == 0 and (h // x) in (16, 32, 64, 128)] or [h // 64]))
        cfg = dict(base, n_embd=int(rng.choice([64, 128, 192])), hn_hidden_size=h, hn_intermediate_size=int(rng.choice([128, 192, 384])),
                   hn_num_attention_heads=heads, hn_n_layers=int(rng.choice([1, 2, 3])), hn_surface_maxlen=int(rng.choice([1, 2, 3, 7, 12])),
                   separate_out_embeddings=bool(rng.integers(2)), hn_embed_lang_id=bool(rng.integers(2)), hn_rescale_embeddings=bool(rng.integers(2)),
                   hn_predict_bias=bool(rng.integers(2)), hn_single_head=bool(rng.integers(2)))
        rows = int(rng.choice([1, 5, 64, 300, 1100]))
        packed = bool(rng.integers(2))
        precision = str(rng.choice(["f32", "f32", "f16", "bf16"]))
        try:
            w = synth.make_weights(cfg, seed)
            src_np = synth.make_source_embeddings(cfg, seed)
            ids_np = synth.make_surface_forms(cfg, rows, seed=seed, n_special=min(1, rows))
            lang = int(rng.integers(0, cfg["n_langs"])) if cfg["hn_embed_lang_id"] else None
            W64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in w.items()}
            ref = torch_port.forward(W64, cfg, torch.from_numpy(ids_np).long(), torch.from_numpy(src_np), lang)
            gen = torch.Generator().manual_seed(seed)
            cot = [None if r is None else torch.randn(r.shape, generator=gen, dtype=torch.float64) for r in ref]
            # rows whose every key is masked and that carry no language token are implementation-defined in the reference
            # (tests/util.py all_pad_rows): keep them out of the loss
            if not cfg["hn_embed_lang_id"]:
                dead = torch.from_numpy((ids_np == cfg["pad_token_id"]).all(1))
                for c in cot:
                    if c is not None:
                        c[dead] = 0
            sum((r * c).sum() for r, c in zip(ref, cot) if r is not None).backward()
            model = ZettHypernet(ZettHypernetConfig(**cfg))
            model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
            model = model.to(dev).requires_grad_(True).train()
            model.train_packed, model.train_precision = packed, precision
            out = model(torch.from_numpy(ids_np).to(dev), source_embeddings=torch.from_numpy(src_np).to(dev), lang_index=None if lang is None else torch.tensor(lang))
            sum((o.double() * c.to(dev)).sum() for o, c in zip(out, cot) if o is not None).backward()
            params = dict(model.named_parameters())
            gmax = max(float(p.grad.norm()) for p in W64.values() if p.grad is not None)
            if gmax == 0.0:                                   # (only implementation-defined rows: nothing to compare but "zero stays zero")
                assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in params.values()), "a gradient out of zero cotangents"
            for name, p64 in W64.items():
                if name not in params or p64.grad is None or gmax == 0.0:
                    continue
                g = params[name].grad
                assert g is not None, f"{name}: no gradient"
                rel = float((g.double().cpu() - p64.grad).norm() / (p64.grad.norm() + 1e-6 * gmax))
                worst[precision] = max(worst[precision], rel)
                assert np.isfinite(rel) and rel < LIMITS[precision], f"{name}: relative gradient error {rel:.3e}"
            del model
        except Exception as e:
            failures.append({"seed": seed, "precision": precision, "packed": packed, "rows": rows, "error": repr(e)[:300],
                             "cfg": {k: v for k, v in cfg.items() if k.startswith("hn_") or k in ("n_embd", "separate_out_embeddings")}})
            if len(failures) >= 5:
                break
            continue
        done += 1
    print(json.dumps({"seeds": [args.seeds[0], seed + 1], "cases": done, "limits": LIMITS, "worst_relative_gradient_error": worst, "failures": failures,
                      "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
