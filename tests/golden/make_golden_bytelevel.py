"""Fixtures for convert_to_byte_level: inputs and the reference's outputs.

Runs only in the build container (imports /root/reference with jax/flax/optax stubbed, see
make_golden_retok.py).  A fixture holds tokenizer JSON (data produced by `tokenizers` training on
local text) before and after the reference's zett.tokenizer_converters.convert_to_byte_level.
"""
from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_retok import (_import_reference, corpus, train_bytelevel_bpe, train_metaspace_unigram,  # noqa: E402
                               train_mistral_like, wrap)

SPECIAL_ATTRS = ("bos_token", "eos_token", "unk_token", "pad_token", "sep_token", "cls_token", "mask_token")


def train_wordpiece(lines, vocab_size):
    from tokenizers import Tokenizer, decoders, models, normalizers, pre_tokenizers, trainers
    tok = Tokenizer(models.WordPiece(unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.decoder = decoders.WordPiece()
    trainer = trainers.WordPieceTrainer(vocab_size=vocab_size, special_tokens=["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"],
                                        show_progress=False)
    tok.train_from_iterator(lines, trainer)
    return tok


def describe(tok):
    return {"json": json.loads(tok._tokenizer.to_str()), "special": {a: getattr(tok, a) for a in SPECIAL_ATTRS},
            "len": len(tok)}


def clone(desc):
    from tokenizers import Tokenizer
    special = {k: v for k, v in desc["special"].items() if v is not None}
    return wrap(Tokenizer.from_str(json.dumps(desc["json"])), **special)


def gen_bytelevel():
    convert_to_byte_level, *_ = _import_reference()
    la, lb = corpus(11, 2500), corpus(12, 2500)
    makers = {
        "bytebpe": lambda lines: wrap(train_bytelevel_bpe(lines, 700, ["<|endoftext|>"]), eos_token="<|endoftext|>"),
        "unigram": lambda lines: wrap(train_metaspace_unigram(lines, 1500), bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<pad>"),
        "mistral": lambda lines: wrap(train_mistral_like(lines, 900), bos_token="<s>", eos_token="</s>", unk_token="<unk>"),
        "wordpiece": lambda lines: wrap(train_wordpiece(lines, 1500), unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]", mask_token="[MASK]"),
    }
    cases = [
        ("bytebpe", None, {}),
        ("bytebpe", "mistral", {"make_whitespace_consistent": True}),
        ("unigram", None, {}),
        ("unigram", "bytebpe", {"make_whitespace_consistent": True}),
        ("mistral", None, {}),
        ("mistral", "unigram", {"make_whitespace_consistent": True}),
        ("wordpiece", None, {}),
        ("wordpiece", "mistral", {"make_whitespace_consistent": True}),
        ("unigram", None, {"keep_normalizer": True, "keep_pretokenizer": True}),
        ("bytebpe", None, {"keep_normalizer": True, "keep_pretokenizer": True, "make_whitespace_consistent": True}),
    ]
    out = []
    for kind, match_kind, flags in cases:
        src = makers[kind](la)
        before = describe(src)          # (Unigram training is not deterministic: convert THIS instance's twin)
        match = makers[match_kind](lb) if match_kind else None
        match_desc = describe(match) if match is not None else None
        converted, n_added = convert_to_byte_level(clone(before), match_special_tokens_to=match, **flags)
        after = describe(converted)
        out.append({"kind": kind, "match_kind": match_kind, "flags": flags, "before": before, "match": match_desc,
                    "after": after, "n_added": n_added,
                    "tokens_after": converted.convert_ids_to_tokens(range(len(converted)))})
        print(kind, match_kind, flags, "len", before["len"], "->", after["len"], "n_added", n_added)
    path = os.path.join(HERE, "bytelevel_cases.json")
    with open(path, "w") as f:
        json.dump(out, f, ensure_ascii=False, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    gen_bytelevel()
