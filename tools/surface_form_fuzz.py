#!/usr/bin/env python
"""Differential campaign for get_surface_form_matrix against the REFERENCE ITSELF
(zett/utils.py:651-701, imported from /root/reference in the build container; the five committed fixtures
tests/golden/retok_*.json are cases of this kind).  The side compared here is the oracle (oracle/retok_ref: C and Python),
which the HIP retokenizer is bit-exact against on 20 000 random models (profiles/r3h_fuzz.md) and on the fixtures: together
the two legs tie the device path to the reference's own outputs.

hn tokenizers: byte-level BPE (plain / ignore_merges), Metaspace Unigram and a Mistral-like byte-fallback BPE, each trained
here on a fresh corpus and converted by the reference's convert_to_byte_level; target vocabularies: byte-level BPEs trained
on other text (made whitespace-consistent, special tokens matched) + edge strings; maxlen 1..24.

    python tools/surface_form_fuzz.py [--rounds 4]
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    from make_golden_retok import _import_reference, corpus, train_bytelevel_bpe, train_metaspace_unigram, train_mistral_like, wrap
    from oracle import retok_ref
    from tokenizers import Tokenizer
    convert_to_byte_level, gsfm, _, _ = _import_reference()
    t0 = time.time()
    n_cases = n_tokens = 0
    failures = []
    extra = ["", "Ġ", "ĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠ", "a" * 40, "ĊĊĊ", "Ġthe", "ÿÿ", "ĠĠhelloĠworldĠĠ", "Ā", "ĉĉ", "!" * 70]
    for rnd in range(args.rounds):
        rng = random.Random(500 + rnd)
        la, lb = corpus(300 + rnd, 3000), corpus(400 + rnd, 3000)
        size = rng.choice([600, 1200, 2000, 3000])
        hns = {}
        src = wrap(train_bytelevel_bpe(la, size, ["<|endoftext|>"]), eos_token="<|endoftext|>")
        hn = convert_to_byte_level(wrap(train_bytelevel_bpe(la, size, ["<|endoftext|>"]), eos_token="<|endoftext|>"))[0]
        hn.pad_token = hn.pad_token or hn.eos_token
        hns["bytebpe"] = (hn, src)
        data = json.loads(hn._tokenizer.to_str())
        data["model"]["ignore_merges"] = True
        hns["bytebpe_ignore_merges"] = (wrap(Tokenizer.from_str(json.dumps(data)), eos_token="<|endoftext|>", pad_token="<|endoftext|>"), src)
        usrc = wrap(train_metaspace_unigram(la, max(size, 1500)), bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<pad>")
        hns["unigram"] = (convert_to_byte_level(wrap(train_metaspace_unigram(la, max(size, 1500)), bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<pad>"))[0], usrc)
        msrc = wrap(train_mistral_like(la, size + 300), bos_token="<s>", eos_token="</s>", unk_token="<unk>")
        mis = convert_to_byte_level(wrap(train_mistral_like(la, size + 300), bos_token="<s>", eos_token="</s>", unk_token="<unk>"))[0]
        mis.pad_token = mis.pad_token or mis.eos_token
        hns["mistral_like"] = (mis, msrc)
        for kind, (hn, match_to) in hns.items():
            tgt = wrap(train_bytelevel_bpe(lb, rng.choice([800, 2500]), ["<|endoftext|>"]), eos_token="<|endoftext|>")
            tgt = convert_to_byte_level(tgt, make_whitespace_consistent=True, match_special_tokens_to=match_to)[0]
            tokens = tgt.convert_ids_to_tokens(range(len(tgt))) + extra
            spec = json.loads(hn._tokenizer.to_str())
            specials = list(hn.all_special_tokens)
            model = retok_ref.model_from_tokenizer_json({"model": spec["model"]}, specials, [hn.convert_tokens_to_ids(s) for s in specials])
            for maxlen in (1, 2, 7, rng.choice([3, 5, 11, 15, 24])):
                want, want_tr = gsfm(tokens, maxlen=maxlen, tokenizer_to_use=hn)
                got, got_tr = retok_ref.surface_form_matrix_c(model, tokens, maxlen, hn.pad_token_id)
                n_cases += 1
                n_tokens += len(tokens)
                ok = np.array_equal(got, np.asarray(want, dtype=np.int32)) and int(got_tr) == int(want_tr)
                if ok and maxlen == 7:
                    got_py, _ = retok_ref.surface_form_matrix_py(model, tokens[:200], maxlen, hn.pad_token_id)
                    ok = np.array_equal(np.asarray(got_py), np.asarray(want)[:200])
                if not ok:
                    failures.append({"round": rnd, "hn": kind, "maxlen": maxlen, "rows_differing": int((np.asarray(got) != np.asarray(want)).any(1).sum())})
    print(json.dumps({"cases": n_cases, "tokens": n_tokens, "failures": failures[:10], "n_failures": len(failures), "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
