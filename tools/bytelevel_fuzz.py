#!/usr/bin/env python
"""Differential campaign for convert_to_byte_level against the REFERENCE ITSELF, imported from
/root/reference in the build container (it cannot run anywhere else; the committed fixtures tests/golden/bytelevel_cases.json
are 10 of these cases).  Every tokenizer kind x every match kind (or none) x every flag combination, on tokenizers trained
here on several corpora / vocabulary sizes; the comparison is tests/test_byte_level.py's.

    python tools/bytelevel_fuzz.py [--rounds 2]
"""
import argparse
import itertools
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))


def compare(case_kind, before, got_tok, n_added, want_tok, want_n_added):
    got, want = json.loads(got_tok._tokenizer.to_str()), json.loads(want_tok._tokenizer.to_str())
    assert n_added == want_n_added, ("n_added", n_added, want_n_added)
    assert got_tok.convert_ids_to_tokens(range(len(got_tok))) == want_tok.convert_ids_to_tokens(range(len(want_tok))), "tokens"
    for key in ("normalizer", "pre_tokenizer"):
        assert got[key] == want[key], key
    assert got.get("post_processor") == want.get("post_processor"), "post_processor"
    gm, wm = got["model"], want["model"]
    assert gm["type"] == wm["type"]
    if gm["type"] == "Unigram":
        assert [tuple(v) for v in gm["vocab"]] == [tuple(v) for v in wm["vocab"]] and gm.get("unk_id") == wm.get("unk_id"), "unigram vocab"
    else:
        assert gm["vocab"] == wm["vocab"], "vocab"
    if gm["type"] == "BPE":
        norm = lambda ms: [tuple(m.split(" ")) if isinstance(m, str) else tuple(m) for m in ms]
        g, w = norm(gm["merges"]), norm(wm["merges"])
        assert sorted(g) == sorted(w), "merges"                 # (the reference emits the surgery's extra merges in set order)
        if case_kind == "bytebpe":
            # the reference keeps the original merges in order up to the whitespace surgery and emits the surgery's merges — which
            # may include original ones it took out (a vocabulary that already merges runs of spaces) — in Python set order,
            # i.e. in an order that changes from process to process: the prefix must be identical, the tail the same set
            orig = set(norm(before["json"]["model"]["merges"]))
            k = next((i for i, m in enumerate(w) if m not in orig), len(w))
            assert g[:k] == w[:k], "the original merges in front of the whitespace surgery"
    for attr in ("bos_token", "eos_token", "unk_token", "pad_token", "sep_token", "cls_token", "mask_token"):
        assert getattr(got_tok, attr) == getattr(want_tok, attr), attr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    args = ap.parse_args()
    from make_golden_bytelevel import clone, describe, train_wordpiece
    from make_golden_retok import _import_reference, corpus, train_bytelevel_bpe, train_metaspace_unigram, train_mistral_like, wrap
    from zett_amd.byte_level import convert_to_byte_level as ours
    ref_convert, *_ = _import_reference()
    t0 = time.time()
    n = compared = both_refuse = 0
    failures = []
    for rnd in range(args.rounds):
        la, lb = corpus(100 + rnd, 1500 + 700 * rnd), corpus(200 + rnd, 1800)
        size = [500, 900, 1400][rnd % 3]
        makers = {
            "bytebpe": lambda lines: wrap(train_bytelevel_bpe(lines, size, ["<|endoftext|>"]), eos_token="<|endoftext|>"),
            "unigram": lambda lines: wrap(train_metaspace_unigram(lines, size + 600), bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<pad>"),
            "mistral": lambda lines: wrap(train_mistral_like(lines, size + 200), bos_token="<s>", eos_token="</s>", unk_token="<unk>"),
            "wordpiece": lambda lines: wrap(train_wordpiece(lines, size + 600), unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]", mask_token="[MASK]"),
        }
        src = {k: describe(m(la)) for k, m in makers.items()}
        match = {k: describe(m(lb)) for k, m in makers.items()}
        for kind, match_kind in itertools.product(makers, [None] + list(makers)):
            for bits in itertools.product([False, True], repeat=3):
                flags = {k: True for k, b in zip(("make_whitespace_consistent", "keep_normalizer", "keep_pretokenizer"), bits) if b}
                n += 1
                try:
                    want, want_n = ref_convert(clone(src[kind]), match_special_tokens_to=clone(match[match_kind]) if match_kind else None, **flags)
                except Exception as e:
                    want, want_n, ref_err = None, None, type(e).__name__
                try:
                    got, got_n = ours(clone(src[kind]), match_special_tokens_to=clone(match[match_kind]) if match_kind else None, **flags)
                    if want is None:
                        failures.append({"round": rnd, "kind": kind, "match": match_kind, "flags": flags, "what": f"the reference raises {ref_err}, the product does not"})
                        continue
                    compare(kind, src[kind], got, got_n, want, want_n)
                    compared += 1
                except Exception as e:
                    if want is None:
                        both_refuse += 1                       # both refuse the combination
                        continue
                    failures.append({"round": rnd, "kind": kind, "match": match_kind, "flags": flags, "what": repr(e)[:300]})
    print(json.dumps({"cases": n, "compared": compared, "both_refuse": both_refuse, "failures": failures[:10], "n_failures": len(failures), "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
