"""convert_to_byte_level against the reference's own outputs
(tests/golden/bytelevel_cases.json, made by tests/golden/make_golden_bytelevel.py).  CPU only."""
import json
import os

import pytest

from tests import util

CASES = json.load(open(os.path.join(util.GOLDEN, "bytelevel_cases.json")))


def _rebuild(desc):
    from tokenizers import Tokenizer
    from transformers import PreTrainedTokenizerFast
    special = {k: v for k, v in desc["special"].items() if v is not None}
    tok = PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_str(json.dumps(desc["json"])), **special)
    assert len(tok) == desc["len"]
    return tok


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['kind']}-{c['match_kind']}-{'+'.join(sorted(c['flags'])) or 'plain'}")
def test_matches_reference(case):
    from zett_amd.byte_level import convert_to_byte_level
    tok = _rebuild(case["before"])
    match = _rebuild(case["match"]) if case["match"] is not None else None
    out, n_added = convert_to_byte_level(tok, match_special_tokens_to=match, **case["flags"])
    assert n_added == case["n_added"]
    got = json.loads(out._tokenizer.to_str())
    want = case["after"]["json"]
    assert out.convert_ids_to_tokens(range(len(out))) == case["tokens_after"]
    assert got["normalizer"] == want["normalizer"]
    assert got["pre_tokenizer"] == want["pre_tokenizer"]
    assert got.get("post_processor") == want.get("post_processor")
    gm, wm = got["model"], want["model"]
    assert gm["type"] == wm["type"]
    if gm["type"] == "Unigram":
        assert [tuple(v) for v in gm["vocab"]] == [tuple(v) for v in wm["vocab"]]
        assert gm.get("unk_id") == wm.get("unk_id")
    else:
        assert gm["vocab"] == wm["vocab"]
    if gm["type"] == "BPE":
        norm = lambda ms: [tuple(m.split(" ")) if isinstance(m, str) else tuple(m) for m in ms]
        g, w = norm(gm["merges"]), norm(wm["merges"])
        # the reference emits the surgery's extra merges in Python-set order (hash dependent); this
        # implementation sorts.  Same merges, and the original merges keep their relative order.
        assert sorted(g) == sorted(w)
        original = [m for m in w if m in set(norm(case["before"]["json"]["model"]["merges"]))] if case["kind"] == "bytebpe" else None
        if original is not None:
            assert [m for m in g if m in set(original)] == original
    for attr, value in case["after"]["special"].items():
        assert getattr(out, attr) == value, attr


def test_byte_alphabet_order_matches_reference():
    """Fill bytes are appended in the order of the reference's CHARS_TO_BYTES literal."""
    from zett_amd.surface_forms import CHARS_TO_BYTES
    ref = json.load(open(os.path.join(util.GOLDEN, "byte_table.json")))["chars_to_bytes"]
    assert list(ref.items()) == list(CHARS_TO_BYTES.items())
