#!/usr/bin/env python
"""Training-time use of the path: one step = the differentiable forward of the whole vocabulary +
the backward to every hypernetwork parameter (what train.py:1007-1013 does once per training step on a sampled
vocabulary), through zett_amd/autograd.py.

    python tools/train_bench.py [--workload mistral_gpt2_32k] [--rows N] [--steps K] [--dense]

Prints one JSON line: rows/s, ms per step (forward / backward split by HIP events), the GEMM FLOPs of a step (counted at
the C-ABI calls: forward + dgrad + wgrad) and the rate they imply against the fp32-MFMA roof (157.3 TFLOP/s) — a LOWER
bound of the GEMM rate, since the row kernels (LayerNorm, GELU, attention, transposes, reductions) are inside the time.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from bench import device_weights  # noqa: E402
from zett_amd import autograd, synth  # noqa: E402
from zett_amd.config import ZettHypernetConfig  # noqa: E402
from zett_amd.hypernet import ZettHypernet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mistral_gpt2_32k", choices=sorted(synth.WORKLOADS))
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16", "f16"], help="arithmetic of the dense contractions (forward, dgrad, wgrad)")
    ap.add_argument("--dense", action="store_true", help="the reference's dense layout instead of the packed schedule")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg, rows, src_dtype, hist = synth.workload(args.workload)
    rows = args.rows or rows
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    model = model.to(dev)
    with torch.no_grad():
        for name, w in device_weights(cfg, dev, seed=0).items():
            dict(model.named_parameters())[name].copy_(w)
    model.requires_grad_(True).train()
    model.train_packed = not args.dense
    model.train_precision = args.precision
    peak = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}[args.precision]
    src = torch.from_numpy(synth.make_source_embeddings(cfg, seed=0, dtype=src_dtype)).to(dev)
    ids = torch.from_numpy(synth.make_surface_forms(cfg, rows, seed=0, hist=hist)).to(dev)
    lang = torch.tensor(3) if cfg.get("hn_embed_lang_id") else None
    flops = {"n": 0.0}
    orig = autograd.Ops.gemm

    def counted(self, x, w, *a, **k):
        flops["n"] += 2.0 * x.shape[0] * w.shape[0] * min(x.shape[1], w.shape[1])
        return orig(self, x, w, *a, **k)

    autograd.Ops.gemm = counted
    g = torch.Generator(device=dev).manual_seed(0)
    cot = None
    t_f = t_b = 0.0

    def step():
        nonlocal cot, t_f, t_b
        model.zero_grad(set_to_none=True)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        out = model(ids, source_embeddings=src, lang_index=lang)
        e1.record()
        if cot is None:
            cot = [None if o is None else torch.randn(o.shape, device=dev, generator=g) for o in out]
        loss = sum((o * c).sum() for o, c in zip(out, cot) if o is not None)
        loss.backward()
        e2.record()
        torch.cuda.synchronize()
        t_f += e0.elapsed_time(e1)
        t_b += e1.elapsed_time(e2)

    for _ in range(args.warmup):
        step()
    flops["n"] = 0.0
    t_f = t_b = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt = time.perf_counter() - t0
    per = flops["n"] / args.steps
    print(json.dumps({"metric": "training step of the embedding-prediction path (forward + backward to every parameter)",
                      "workload": args.workload, "rows": rows, "schedule": "dense" if args.dense else "packed (levers 1-3)", "dtype": args.precision,
                      "rows_per_s": rows * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                      "forward_ms": t_f / args.steps, "backward_ms": t_b / args.steps,
                      "gemm_tflop_per_step": per / 1e12, "gemm_tflops_lower_bound": per / (dt / args.steps) / 1e12, "mfma_peak_tflops": peak,
                      "frac_lower_bound": per / (dt / args.steps) / 1e12 / peak,
                      "peak_memory_gb": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    main()
