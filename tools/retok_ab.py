#!/usr/bin/env python
"""GPU: the stage-2 retokenizer kernels side by side on the bench workloads' own surface forms (text resident in HBM, HIP
events around `reps` calls of zett_retokenize_async): Unigram models on the workgroup-per-64-tokens kernel (r6) vs the
lane-per-token kernel (zett_retok_set_option "unigram_workgroup" 1 / 0), whole vocabulary and a 4 096-token shard.

    python tools/retok_ab.py > gpurun_out/retok_ab.json
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zett_amd import synth  # noqa: E402
from zett_amd.surface_forms import DeviceRetokenizer, HnTokenizerSpec  # noqa: E402


def timed(rt, d_text, d_off, n, seq, reps=20):
    for _ in range(3):
        rt.run_async(d_text, d_off, n, seq)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = rt.run_async(d_text, d_off, n, seq)
    e1.record()
    torch.cuda.synchronize()
    rt.result()
    return e0.elapsed_time(e1) / reps * 1e3, out


def main():
    dev = torch.device("cuda", 0)
    res = []
    for name in (sys.argv[1:] or ("xlmr_gpt2", "mistral_gpt2_32k", "tinyllama_neox", "llama3_256k")):
        cfg, rows, _, hist = synth.workload(name)
        ids = synth.make_surface_forms(cfg, rows, seed=0, hist=hist)
        hn_model, piece_of_id = synth.make_hn_model(name, cfg)
        spec = HnTokenizerSpec.from_model_json(hn_model, ["<unk>", "<s>", "</s>"], [0, 1, 2], cfg["pad_token_id"])
        rt = DeviceRetokenizer(spec, dev)
        for n_rows in (rows, 4096):
            d_text, d_off, n = rt.encode(synth.tokens_for_surface_forms(cfg, ids[:n_rows], piece_of_id))
            line = {"workload": name, "model": hn_model["type"], "tokens": n, "text_bytes": int(d_text.numel())}
            if os.environ.get("ZETT_RETOK_STOP"):          # (tools/_dbg build only: the stage-2 kernel returns after that phase)
                line["stop_after_phase"] = int(os.environ["ZETT_RETOK_STOP"])
            if hn_model["type"] == "Unigram":
                rt.set_option("unigram_workgroup", 2)
                line["us_workgroup_kernel"], a = timed(rt, d_text, d_off, n, ids.shape[1])
                rt.set_option("unigram_workgroup", 0)
                line["us_lane_kernel"], b = timed(rt, d_text, d_off, n, ids.shape[1])
                line["identical"] = bool(torch.equal(a, b))
                rt.set_option("unigram_workgroup", 1)
            else:
                line["us"], a = timed(rt, d_text, d_off, n, ids.shape[1])
            line["equals_workload_ids"] = bool(torch.equal(a.cpu(), torch.from_numpy(ids[:n_rows])))
            res.append(line)
            print(json.dumps(line), flush=True)
        rt.close()


if __name__ == "__main__":
    main()
