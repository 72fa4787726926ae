"""Retokenizer fixtures: outputs of the reference's get_surface_form_matrix.

Runs only in the build container.  zett/utils.py imports jax/flax/optax at module
level (zett/utils.py:3,6-11,19-20); none is installed, none is needed by
get_surface_form_matrix / convert_to_byte_level, so those names are stubbed with
MagicMock before the import.  Nothing of the reference is copied: a fixture holds the
hn tokenizer's model (vocabulary / merges / scores produced by `tokenizers` training
on local text and by the reference's own convert_to_byte_level), the target token
list, and the matrix the reference computed.
"""
from __future__ import annotations

import glob
import json
import os
import random
import sys
import types
from unittest.mock import MagicMock

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def _import_reference():
    for name in ("jax", "jax.numpy", "jax.sharding", "jax.experimental", "jax.experimental.multihost_utils",
                 "flax", "flax.core", "flax.core.frozen_dict", "flax.traverse_util", "flax.serialization",
                 "flax.training", "flax.training.common_utils", "flax.training.train_state", "optax", "h5py",
                 "jax.experimental.pjit", "jax.tree_util", "flax.linen"):
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    linen = types.ModuleType("flax.linen")
    linen.Module = type("Module", (), {})
    linen.__getattr__ = lambda name: MagicMock()
    sys.modules["flax.linen"] = linen
    sys.path.insert(0, REFERENCE)
    from zett.tokenizer_converters import convert_to_byte_level
    from zett.utils import BYTES_TO_CHARS, CHARS_TO_BYTES, get_surface_form_matrix
    return convert_to_byte_level, get_surface_form_matrix, CHARS_TO_BYTES, BYTES_TO_CHARS


def corpus(seed: int, n_lines: int = 6000):
    """Local text: Python stdlib sources + synthetic multilingual words."""
    rng = random.Random(seed)
    files = sorted(glob.glob("/usr/lib/python3.10/*.py"))
    rng.shuffle(files)
    lines = []
    for f in files[:60]:
        try:
            lines += [ln.strip() for ln in open(f, encoding="utf-8", errors="ignore") if len(ln.strip()) > 20]
        except OSError:
            pass
    rng.shuffle(lines)
    lines = lines[:n_lines]
    blocks = [(0x00C0, 0x017F), (0x0400, 0x045F), (0x0370, 0x03FF), (0x4E00, 0x4E80), (0x0900, 0x0940)]
    for _ in range(n_lines // 4):
        lo, hi = rng.choice(blocks)
        words = ["".join(chr(rng.randint(lo, hi)) for _ in range(rng.randint(2, 7))) for _ in range(rng.randint(3, 9))]
        lines.append(" ".join(words))
    rng.shuffle(lines)
    return lines


def train_bytelevel_bpe(lines, vocab_size, specials):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=specials,
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(lines, trainer)
    return tok


def train_metaspace_unigram(lines, vocab_size):
    from tokenizers import Tokenizer, decoders, models, normalizers, pre_tokenizers, trainers
    tok = Tokenizer(models.Unigram())
    tok.normalizer = normalizers.NFKC()
    tok.pre_tokenizer = pre_tokenizers.Metaspace()
    tok.decoder = decoders.Metaspace()
    trainer = trainers.UnigramTrainer(vocab_size=vocab_size, special_tokens=["<s>", "<pad>", "</s>", "<unk>"],
                                      unk_token="<unk>", show_progress=False)
    tok.train_from_iterator(lines, trainer)
    return tok


def train_mistral_like(lines, vocab_size):
    """BPE + byte_fallback + <0xXX> tokens + Prepend/Replace normalizer (Llama/Mistral style)."""
    from tokenizers import Tokenizer, decoders, models, normalizers, trainers
    tok = Tokenizer(models.BPE(unk_token="<unk>", byte_fallback=True, fuse_unk=True))
    tok.normalizer = normalizers.Sequence([normalizers.Prepend("▁"), normalizers.Replace(" ", "▁")])
    tok.decoder = decoders.Sequence([decoders.Replace("▁", " "), decoders.ByteFallback(), decoders.Fuse()])
    specials = ["<unk>", "<s>", "</s>"] + [f"<0x{i:02X}>" for i in range(256)]
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=specials, show_progress=False)
    tok.train_from_iterator(lines, trainer)
    return tok


def train_bert_wordpiece(lines, vocab_size):
    """WordPiece + BertNormalizer + BertPreTokenizer + "##" continuing prefix (BERT style)."""
    from tokenizers import Tokenizer, decoders, models, normalizers, pre_tokenizers, trainers
    tok = Tokenizer(models.WordPiece(unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=False)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.decoder = decoders.WordPiece(prefix="##")
    trainer = trainers.WordPieceTrainer(vocab_size=vocab_size, special_tokens=["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"],
                                        continuing_subword_prefix="##", show_progress=False)
    tok.train_from_iterator(lines, trainer)
    return tok


def wrap(tok, **special):
    from transformers import PreTrainedTokenizerFast
    return PreTrainedTokenizerFast(tokenizer_object=tok, **special)


def dump_case(name, hn, tokens, maxlen, gsfm):
    matrix, n_truncated = gsfm(tokens, maxlen=maxlen, tokenizer_to_use=hn)
    data = json.loads(hn._tokenizer.to_str())
    specials = list(hn.all_special_tokens)
    out = {
        "model": data["model"],
        "special_tokens": specials,
        "special_ids": [hn.convert_tokens_to_ids(s) for s in specials],
        "pad_token_id": hn.pad_token_id,
        "maxlen": maxlen,
        "tokens": tokens,
        "expected": matrix.tolist(),
        "n_truncated": int(n_truncated),
    }
    path = os.path.join(HERE, name + ".json")
    with open(path, "w") as f:
        json.dump(out, f, ensure_ascii=False, separators=(",", ":"))
    lens = (matrix != hn.pad_token_id).sum(1)
    print("wrote", os.path.relpath(path, REPO), f"{len(tokens)} tokens, truncated {n_truncated}, mean len {lens.mean():.2f},",
          f"{os.path.getsize(path) / 1024:.0f} KiB")


def gen_retok():
    convert_to_byte_level, gsfm, c2b, b2c = _import_reference()

    # G0: the byte table itself (known-answer vector for A0)
    with open(os.path.join(HERE, "byte_table.json"), "w") as f:
        json.dump({"chars_to_bytes": c2b}, f, ensure_ascii=False)

    lines_a, lines_b = corpus(1), corpus(2)

    # target vocabulary: byte-level BPE trained on other text, made whitespace-consistent
    def target_tokens(match_to, size=2500):
        tgt = wrap(train_bytelevel_bpe(lines_b, size, ["<|endoftext|>"]), eos_token="<|endoftext|>")
        tgt = convert_to_byte_level(tgt, make_whitespace_consistent=True, match_special_tokens_to=match_to)[0]
        return tgt.convert_ids_to_tokens(range(len(tgt)))

    extra = ["", "Ġ", "ĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠ", "a" * 40, "ĊĊĊ", "Ġthe", "ÿÿ", "ĠĠhelloĠworldĠĠ"]

    # (1) byte-level BPE hn tokenizer (GPT-2 style), pad := eos like train.py:291-292
    hn_src = wrap(train_bytelevel_bpe(lines_a, 2000, ["<|endoftext|>"]), eos_token="<|endoftext|>")
    hn = convert_to_byte_level(wrap(train_bytelevel_bpe(lines_a, 2000, ["<|endoftext|>"]), eos_token="<|endoftext|>"))[0]
    if hn.pad_token is None:
        hn.pad_token = hn.eos_token
    toks = target_tokens(hn_src) + extra
    dump_case("retok_bytebpe", hn, toks, 7, gsfm)

    # (2) the same with ignore_merges
    data = json.loads(hn._tokenizer.to_str())
    data["model"]["ignore_merges"] = True
    from tokenizers import Tokenizer
    hn2 = wrap(Tokenizer.from_str(json.dumps(data)), eos_token="<|endoftext|>", pad_token="<|endoftext|>")
    dump_case("retok_bytebpe_ignore_merges", hn2, toks, 7, gsfm)

    # (3) XLM-R-like: Unigram + Metaspace + NFKC converted by the reference's convert_to_byte_level
    uni_src = wrap(train_metaspace_unigram(lines_a, 2000), bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<pad>")
    uni = convert_to_byte_level(wrap(train_metaspace_unigram(lines_a, 2000), bos_token="<s>", eos_token="</s>",
                                     unk_token="<unk>", pad_token="<pad>"))[0]
    toks_u = target_tokens(uni_src) + extra
    dump_case("retok_unigram", uni, toks_u, 7, gsfm)
    dump_case("retok_unigram_L15", uni, toks_u[:600], 15, gsfm)

    # (4) Mistral-like: BPE + byte_fallback converted by convert_to_byte_level, pad := eos
    mis_src = wrap(train_mistral_like(lines_a, 2200), bos_token="<s>", eos_token="</s>", unk_token="<unk>")
    mis = convert_to_byte_level(wrap(train_mistral_like(lines_a, 2200), bos_token="<s>", eos_token="</s>", unk_token="<unk>"))[0]
    if mis.pad_token is None:
        mis.pad_token = mis.eos_token
    toks_m = target_tokens(mis_src) + extra
    dump_case("retok_mistral_like", mis, toks_m, 7, gsfm)


def gen_wordpiece():
    """(5) WordPiece hn tokenizer (zett/utils.py:681 is model-agnostic; zett/tokenizer_converters.py:370-373 carries WordPiece
    through convert_to_byte_level, which empties the continuing prefix): BERT-style tokenizer converted by the reference,
    matrix computed by the reference's get_surface_form_matrix.  Separate entry point: the fixtures of gen_retok() stay as
    they were generated."""
    convert_to_byte_level, gsfm, c2b, b2c = _import_reference()
    lines_a, lines_b = corpus(1), corpus(2)
    kw = dict(unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]", mask_token="[MASK]")
    wp_src = wrap(train_bert_wordpiece(lines_a, 2000), **kw)
    wp = convert_to_byte_level(wrap(train_bert_wordpiece(lines_a, 2000), **kw))[0]
    tgt = wrap(train_bytelevel_bpe(lines_b, 2500, ["<|endoftext|>"]), eos_token="<|endoftext|>")
    tgt = convert_to_byte_level(tgt, make_whitespace_consistent=True, match_special_tokens_to=wp_src)[0]
    toks = tgt.convert_ids_to_tokens(range(len(tgt)))
    extra = ["", "Ġ", "ĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠĠ", "a" * 40, "b" * 101, "Ġ" + "c" * 100, "ĊĊĊ", "Ġthe", "ÿÿ", "ĠĠhelloĠworldĠĠ", "Ġimport", "ĠimportĠos"]
    dump_case("retok_wordpiece", wp, toks + extra, 7, gsfm)
    dump_case("retok_wordpiece_L15", wp, (toks + extra)[-700:], 15, gsfm)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wordpiece":
        gen_wordpiece()
    else:
        gen_retok()
