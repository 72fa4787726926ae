"""Random hn-tokenizer models for the retokenizer tests (BPE / Unigram / WordPiece in tokenizers.json form)."""
import random

from oracle.retok_ref import BYTES_TO_CHARS

ALPHABET = [ord(c) for c in "abcde"] + [0x20, 0x0A, 0x80, 0xC4, 0xE2, 0xFF]


def chars(raw: bytes) -> str:
    return "".join(BYTES_TO_CHARS[b] for b in raw)


def random_bpe(rng: random.Random, n_merges=40):
    singles = [b for b in ALPHABET if rng.random() < 0.8]
    vocab = {}
    for b in singles:
        vocab[chars(bytes([b]))] = len(vocab)
    toks = [bytes([b]) for b in singles]
    merges = []
    for _ in range(n_merges):
        a, b = rng.choice(toks), rng.choice(toks)
        if len(a) + len(b) > 8:
            continue
        new = a + b
        if chars(new) not in vocab:
            vocab[chars(new)] = len(vocab)
            toks.append(new)
        merges.append([chars(a), chars(b)])          # duplicates of a pair / of a product are allowed
    model = {"type": "BPE", "vocab": vocab, "merges": merges, "unk_token": None, "fuse_unk": False,
             "byte_fallback": False, "ignore_merges": rng.random() < 0.3, "dropout": None,
             "continuing_subword_prefix": None, "end_of_word_suffix": None}
    if rng.random() < 0.5:
        vocab["<unk>"] = len(vocab)
        model["unk_token"] = "<unk>"
        model["fuse_unk"] = rng.random() < 0.5
    if rng.random() < 0.4:
        model["byte_fallback"] = True
        for b in range(256):
            if rng.random() < 0.7:
                vocab[f"<0x{b:02X}>"] = len(vocab)
    return model


def random_unigram(rng: random.Random, n_pieces=60):
    vocab = [["<unk>", 0.0]]
    for b in ALPHABET:
        if rng.random() < 0.75:
            vocab.append([chars(bytes([b])), rng.choice([-1.0, -2.0, -3.0, -1.5])])
    for _ in range(n_pieces):
        n = rng.randint(2, 5)
        piece = bytes(rng.choice(ALPHABET) for _ in range(n))
        vocab.append([chars(piece), rng.choice([-2.0, -3.0, -4.0, -2.5, -6.0, -100000.0])])   # duplicates allowed
    model = {"type": "Unigram", "unk_id": 0, "vocab": vocab, "byte_fallback": False}
    if rng.random() < 0.3:
        model["byte_fallback"] = True
        for b in range(256):
            if rng.random() < 0.8:
                vocab.append([f"<0x{b:02X}>", -20.0])
    return model


def random_wordpiece(rng: random.Random, n_pieces=70):
    """WordPiece in tokenizers.json form: a random continuing prefix ("" as convert_to_byte_level leaves it, "##" as BERT
    ships it, or a prefix that is itself made of alphabet characters), some pieces with it, sometimes no [UNK] in the
    vocabulary, a small max_input_chars_per_word now and then."""
    prefix = rng.choice(["", "", "##", "a", "ab"])
    vocab = {}
    if rng.random() < 0.8:
        vocab["[UNK]"] = 0
    for b in ALPHABET:
        if rng.random() < 0.7:
            vocab.setdefault(chars(bytes([b])), len(vocab))
        if rng.random() < 0.6:
            vocab.setdefault(prefix + chars(bytes([b])), len(vocab))
    for _ in range(n_pieces):
        piece = bytes(rng.choice(ALPHABET) for _ in range(rng.randint(2, 5)))
        vocab.setdefault((prefix if rng.random() < 0.5 else "") + chars(piece), len(vocab))
    return {"type": "WordPiece", "unk_token": "[UNK]", "continuing_subword_prefix": prefix,
            "max_input_chars_per_word": rng.choice([100, 100, 100, 6, 9]), "vocab": vocab}


def random_tokens(rng: random.Random, n=200, maxlen=12):
    out = []
    for _ in range(n):
        k = rng.randint(0, maxlen)
        raw = bytes(rng.choice(ALPHABET + [ord("z")]) for _ in range(k))
        out.append(chars(raw))
    return out


def build_tokenizers_model(model: dict):
    """The same model as a `tokenizers` object (the third-party library the reference calls)."""
    from tokenizers import models
    if model["type"] == "BPE":
        kw = dict(unk_token=model["unk_token"], fuse_unk=model["fuse_unk"], byte_fallback=model["byte_fallback"],
                  ignore_merges=model["ignore_merges"])
        return models.BPE(dict(model["vocab"]), [tuple(m) for m in model["merges"]], **kw)
    if model["type"] == "WordPiece":
        return models.WordPiece(dict(model["vocab"]), unk_token=model["unk_token"], max_input_chars_per_word=model["max_input_chars_per_word"],
                                continuing_subword_prefix=model["continuing_subword_prefix"])
    return models.Unigram([tuple(v) for v in model["vocab"]], unk_id=model["unk_id"], byte_fallback=model["byte_fallback"])
