#!/usr/bin/env python
"""Fuzz campaign for the retokenizer beyond the seeds tests/ pins (not part of the suite: a one-off hunt whose result is
recorded in profiles/).

    python tools/retok_fuzz.py --leg cpu --seeds 0 3000     # oracle (C and Python) against the installed `tokenizers` wheel
    python tools/retok_fuzz.py --leg gpu --seeds 0 1500     # zett_retokenize (HIP, through the C ABI) against the C oracle

Random BPE / Unigram models (tests/retok_random.py) with varied sizes, merge counts, unk / byte-fallback / ignore_merges
settings, token lengths up to 300 bytes and matrix widths 1..33.  Prints one JSON line: seeds run, cases, first mismatch."""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import retok_ref  # noqa: E402
from tests import retok_random as rr  # noqa: E402


def case(seed):
    rng = random.Random(77000 + seed)
    make = rng.choice([rr.random_bpe, rr.random_unigram, rr.random_wordpiece])          # (WordPiece: r4)
    size = rng.choice([3, 10, 40, 80, 200, 600])
    model = make(rng, size)
    tokens = rr.random_tokens(rng, rng.choice([1, 50, 400]), maxlen=rng.choice([1, 4, 12, 40])) + rr.random_tokens(rng, rng.choice([0, 10]), maxlen=300)
    width = rng.choice([1, 2, 7, 9, 16, 33])
    return model, tokens, width


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", choices=["cpu", "gpu"], required=True)
    ap.add_argument("--seeds", type=int, nargs=2, default=[0, 500])
    ap.add_argument("--budget-s", type=float, default=1e9)
    args = ap.parse_args()
    t0 = time.time()
    n_cases = n_tokens = 0
    bad = None
    if args.leg == "gpu":
        from zett_amd.surface_forms import HnTokenizerSpec, get_surface_form_matrix
    seed = args.seeds[0]
    for seed in range(*args.seeds):
        if time.time() - t0 > args.budget_s:
            break
        model, tokens, width = case(seed)
        has_unk = isinstance(model["vocab"], dict) and "<unk>" in model["vocab"]
        specials, special_ids = (["<unk>"], [model["vocab"]["<unk>"]]) if has_unk else ([], [])
        try:
            if args.leg == "cpu":
                lib = rr.build_tokenizers_model(model)
                om = retok_ref.model_from_tokenizer_json(model)
                want = [[t.id for t in lib.tokenize(tok)] for tok in tokens]
                got = [retok_ref.tokenize(om, retok_ref.token_to_bytes(tok)) for tok in tokens]
                mat, _ = retok_ref.surface_form_matrix_c(om, tokens, width, -7)
                ok = got == want and all(list(r[:len(w)]) == w[:width] and all(x == -7 for x in r[len(w):]) for r, w in zip(mat, want))
            else:
                om = retok_ref.model_from_tokenizer_json(model, specials, special_ids)
                want, want_tr = retok_ref.surface_form_matrix_c(om, tokens + specials, width, 77)
                spec = HnTokenizerSpec.from_model_json(model, specials, special_ids, 77)
                got, got_tr = get_surface_form_matrix(tokens + specials, width, spec)
                ok = np.array_equal(got, want) and got_tr == want_tr
        except Exception as e:                         # both sides must also agree on what they refuse
            ok, err = False, repr(e)
            if args.leg == "gpu":
                try:
                    retok_ref.surface_form_matrix_c(om, tokens + specials, width, 77)
                except Exception:
                    ok = True                           # the oracle refuses the same input (e.g. an unknown byte without unk)
            else:                                       # the library refused (a word that needs a missing [UNK] / unk id): so must the oracle
                try:
                    om = retok_ref.model_from_tokenizer_json(model)
                    [retok_ref.tokenize(om, retok_ref.token_to_bytes(tok)) for tok in tokens]
                except Exception:
                    ok = True
            if not ok:
                bad = bad or {"seed": seed, "error": err}
                break
        if not ok and bad is None:
            bad = {"seed": seed, "model_type": model["type"], "width": width, "n_tokens": len(tokens)}
            break
        n_cases += 1
        n_tokens += len(tokens)
    print(json.dumps({"leg": args.leg, "seeds": [args.seeds[0], seed + 1], "cases": n_cases, "tokens": n_tokens, "first_mismatch": bad,
                      "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
