#!/usr/bin/env python
"""GPU box: what computing the hoisted table once per JOB is worth to the reference CLI's batched prediction (zett_amd.transfer.
predict_vocabulary with --batch_size 16384, scripts/transfer.py's default): the whole vocabulary of a BASELINE workload through the
kept Python API with ZETT_JOB_TABLE = 0 and 1, device-resident id matrix, f16 policy.  One JSON line per workload.

    python tools/job_table_bench.py [workload ...]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import device_weights  # noqa: E402
from zett_amd import synth  # noqa: E402
from zett_amd.config import ZettHypernetConfig  # noqa: E402
from zett_amd.hypernet import ZettHypernet  # noqa: E402
from zett_amd.transfer import Args, predict_vocabulary  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for name in sys.argv[1:] or ["mistral_gpt2_32k", "mistral_neox", "llama3_256k"]:
        cfg, rows, src_dtype, hist = synth.workload(name)
        with torch.device("meta"):
            pass
        model = ZettHypernet(ZettHypernetConfig(**cfg))
        model.load_state_dict({k: v.float().cpu() for k, v in device_weights(cfg, dev, seed=0).items()})
        model = model.to(dev).eval()
        g = torch.Generator(device=dev); g.manual_seed(1)
        src = (0.02 * torch.randn((cfg["original_vocab_size"], model.dims.n_in_embd), device=dev, generator=g)).to(getattr(torch, src_dtype))
        sfm = torch.from_numpy(synth.make_surface_forms(cfg, rows, seed=0, hist=hist)).to(dev)
        lang = torch.tensor(3) if model.dims.embed_lang else None
        res = {"workload": name, "rows": rows, "batch_size": 16384, "batches": -(-rows // 16384)}
        outs = {}
        for flag in ("0", "1"):
            os.environ["ZETT_JOB_TABLE"] = flag
            for _ in range(2):
                out = predict_vocabulary(model, sfm, src, lang, Args(output="", batch_size=16384), rng=__import__("numpy").random.default_rng(0))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                out = predict_vocabulary(model, sfm, src, lang, Args(output="", batch_size=16384), rng=__import__("numpy").random.default_rng(0))
            torch.cuda.synchronize()
            res["ms_job_table_" + flag] = (time.perf_counter() - t0) / 3 * 1e3
            outs[flag] = out
        res["identical"] = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(outs["0"], outs["1"]))
        print(json.dumps(res), flush=True)
        del model, src, sfm, outs, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
