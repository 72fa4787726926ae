#!/usr/bin/env python
"""Throughput of get_surface_form_matrix on the GPU vs the oracle (C) and the tokenizers library
loop the reference runs (zett/utils.py:670-687), on a 50k-token synthetic target vocabulary."""
import json
import os
import random
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import retok_ref  # noqa: E402
from zett_amd.surface_forms import HnTokenizerSpec, device_retokenizer  # noqa: E402


def main():
    out = {}
    for name in ("retok_mistral_like", "retok_unigram"):
        g = json.load(open(os.path.join(REPO, "tests", "golden", name + ".json")))
        rng = random.Random(5)
        base = [t for t in g["tokens"] if t not in g["special_tokens"] and t]
        tokens = list(g["tokens"])
        while len(tokens) < 50000:
            tokens.append((rng.choice(base) + rng.choice(base))[:rng.randint(1, 24)])
        spec = HnTokenizerSpec.from_model_json(g["model"], g["special_tokens"], g["special_ids"], g["pad_token_id"])
        rt = device_retokenizer(spec, "cuda:0")
        rt(tokens, 7)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m, ntr = rt(tokens, 7)
        torch.cuda.synchronize()
        gpu_s = (time.perf_counter() - t0) / 5
        # device-only part: inputs already resident
        enc = [t.encode() for t in tokens]
        om = retok_ref.model_from_tokenizer_json({"model": g["model"]}, g["special_tokens"], g["special_ids"])
        t0 = time.perf_counter()
        want, _ = retok_ref.surface_form_matrix_c(om, tokens, 7, g["pad_token_id"])
        c_s = time.perf_counter() - t0
        assert (m.cpu().numpy() == want).all()
        from tests.retok_random import build_tokenizers_model
        lib = build_tokenizers_model(dict(g["model"], unk_token=g["model"].get("unk_token"), fuse_unk=g["model"].get("fuse_unk", False),
                                          ignore_merges=g["model"].get("ignore_merges", False)) if g["model"]["type"] == "BPE" else g["model"])
        t0 = time.perf_counter()
        for t in tokens:
            [x.id for x in lib.tokenize(t)]
        lib_s = time.perf_counter() - t0
        nbytes = sum(map(len, enc))
        out[name] = {"tokens": len(tokens), "text_bytes": nbytes, "gpu_ms_incl_host_encode_and_h2d": gpu_s * 1e3,
                     "gpu_tokens_per_s": len(tokens) / gpu_s, "oracle_c_tokens_per_s": len(tokens) / c_s,
                     "tokenizers_python_loop_tokens_per_s": len(tokens) / lib_s}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
