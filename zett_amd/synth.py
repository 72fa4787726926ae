"""Synthetic workloads at the reference's real shapes.

There is no network on either box: no hypernet checkpoints, no GPT-2 / GPT-NeoX /
Mistral tokenizers (the reference's ``artifacts/tokenizers/**/tokenizer.json`` are
git-LFS stubs).  Every BASELINE.json config is therefore realised as seeded
random weights of the exact architecture (shapes from the reference's
``configs/zeroshot/*.json``) plus synthetic surface forms drawn
from the length histogram measured on real-looking byte-BPE tokenizer pairs.

All generators use numpy's Philox bit generator keyed by (seed, stream) so the
build container and the GPU box produce identical tensors.
"""
from __future__ import annotations

import zlib
from typing import Dict, Optional

import numpy as np

from .dims import HypernetDims, weight_shapes

# P(len = 1..7) of a 20k byte-BPE target vocab retokenized by an 8k byte-BPE hn
# tokenizer (mean 1.81) and of the Mistral-like pair (mean 2.36) — SURVEY.md §8d.
LEN_HIST_BPE = (0.395, 0.470, 0.098, 0.023, 0.005, 0.002, 0.007)
LEN_HIST_MISTRAL = (0.18, 0.44, 0.26, 0.076, 0.019, 0.006, 0.010)

_COMMON = dict(
    hn_model_name_or_path="roberta-base",
    hn_surface_maxlen=7,
    hn_n_layers=3,
    hn_rescale_embeddings=True,
    hn_embed_using_source_embeddings=True,
    hn_predict_bias=True,
    hn_model_type="roberta",
)

# name -> (config dict, default row count, source-embedding dtype, length histogram)
WORKLOADS = {
    # tiny shape for unit tests (all dims multiples of 64)
    "tiny": (dict(_COMMON, n_embd=64, hn_hidden_size=128, hn_intermediate_size=256,
                  hn_num_attention_heads=2, separate_out_embeddings=True, hn_embed_lang_id=True,
                  n_langs=5, pad_token_id=1, original_vocab_size=300, hn_n_extra_tokens=5,
                  vocab_size=305), 96, "float32", LEN_HIST_MISTRAL),
    # C1/C2: xlm-roberta-base hypernet -> GPT-2 vocab
    "xlmr_gpt2": (dict(_COMMON, n_embd=768, hn_hidden_size=768, hn_intermediate_size=1536,
                       hn_num_attention_heads=12, separate_out_embeddings=False, hn_embed_lang_id=True,
                       n_langs=26, pad_token_id=1, original_vocab_size=250002, hn_n_extra_tokens=200,
                       vocab_size=250202), 50350, "float32", LEN_HIST_MISTRAL),
    # C3: TinyLlama-1.1B hypernet -> GPT-NeoX vocab
    "tinyllama_neox": (dict(_COMMON, n_embd=2048, hn_hidden_size=2048, hn_intermediate_size=4096,
                            hn_num_attention_heads=32, separate_out_embeddings=True, hn_embed_lang_id=False,
                            pad_token_id=2, original_vocab_size=32000, hn_n_extra_tokens=1,
                            vocab_size=32001), 50370, "float32", LEN_HIST_MISTRAL),
    # C4: Mistral-7B hypernet -> GPT-NeoX vocab
    "mistral_neox": (dict(_COMMON, n_embd=4096, hn_hidden_size=4096, hn_intermediate_size=8192,
                          hn_num_attention_heads=32, separate_out_embeddings=True, hn_embed_lang_id=False,
                          pad_token_id=2, original_vocab_size=32000, hn_n_extra_tokens=1,
                          vocab_size=32001), 50370, "float32", LEN_HIST_MISTRAL),
    # north-star headline: Mistral-7B hypernet, 32k-token GPT-2-style vocab
    "mistral_gpt2_32k": (dict(_COMMON, n_embd=4096, hn_hidden_size=4096, hn_intermediate_size=8192,
                              hn_num_attention_heads=32, separate_out_embeddings=True, hn_embed_lang_id=False,
                              pad_token_id=2, original_vocab_size=32000, hn_n_extra_tokens=1,
                              vocab_size=32001), 32768, "float32", LEN_HIST_MISTRAL),
    # C5: Llama-3-8B hypernet -> 256k synthetic multilingual Unigram vocab, fp16 source embeddings
    "llama3_256k": (dict(_COMMON, n_embd=4096, hn_hidden_size=4096, hn_intermediate_size=8192,
                         hn_num_attention_heads=32, separate_out_embeddings=True, hn_embed_lang_id=False,
                         pad_token_id=128001, original_vocab_size=128256, hn_n_extra_tokens=0,
                         vocab_size=128256), 262144, "float16", LEN_HIST_MISTRAL),
}


def workload(name: str):
    cfg, rows, src_dtype, hist = WORKLOADS[name]
    return dict(cfg), rows, src_dtype, hist


def _rng(seed: int, stream: str) -> np.random.Generator:
    key = (int(seed) << 32) ^ zlib.crc32(stream.encode())
    return np.random.Generator(np.random.Philox(key=key))


def _normal(rng, shape, std, mean=0.0):
    out = rng.standard_normal(size=shape, dtype=np.float32)
    out *= np.float32(std)
    if mean:
        out += np.float32(mean)
    return out


def make_weights(cfg, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random checkpoint (fp32) — recipe of SURVEY.md §8d."""
    out: Dict[str, np.ndarray] = {}
    for name, shape in weight_shapes(cfg).items():
        rng = _rng(seed, name)
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith("LayerNorm.weight") or name.endswith("ln.weight"):
            w = _normal(rng, shape, 0.05, 1.0)
        elif name.endswith("LayerNorm.bias") or name.endswith("ln.bias"):
            w = _normal(rng, shape, 0.02)
        elif name.endswith("scaler.w"):
            w = rng.uniform(0.5, 2.0, size=shape).astype(np.float32)
        elif name.endswith("scaler.b"):
            w = _normal(rng, shape, 0.01)
        elif leaf == "bias":
            w = _normal(rng, shape, 0.01)
        else:                                   # Linear / Embedding weights
            w = _normal(rng, shape, 0.02)
        out[name] = w
    return out


def make_source_embeddings(cfg, seed: int = 0, dtype: str = "float32", rows: Optional[int] = None) -> np.ndarray:
    """Frozen LM embedding matrix [V_src, E_in] ~ N(0, 0.02^2)."""
    d = HypernetDims.from_config(cfg)
    v = int(rows if rows is not None else d.original_vocab_size)
    return _normal(_rng(seed, "source_embeddings"), (v, d.n_in_embd), 0.02).astype(dtype)


def make_surface_forms(cfg, n_rows: int, seed: int = 0, hist=LEN_HIST_MISTRAL,
                       seq: Optional[int] = None, n_special: int = 0) -> np.ndarray:
    """int32 [n_rows, L] surface-form matrix shaped like get_surface_form_matrix output.

    Row lengths follow ``hist`` (renormalised / truncated to L); ids are uniform in
    [3, V0 + X) excluding the pad id; the first ``n_special`` rows carry only the
    pad id in column 0 (what a special token equal to the pad token produces,
    reference zett/utils.py:671-673) — the all-pad edge case.
    """
    d = HypernetDims.from_config(cfg)
    seq = int(seq if seq is not None else (cfg.get("hn_surface_maxlen", 7) if isinstance(cfg, dict)
                                           else getattr(cfg, "hn_surface_maxlen", 7)))
    rng = _rng(seed, "surface_forms")
    p = np.asarray(hist[:seq], dtype=np.float64)
    if len(p) < seq:
        p = np.concatenate([p, np.full(seq - len(p), 1e-3)])
    p /= p.sum()
    lengths = rng.choice(np.arange(1, seq + 1), size=n_rows, p=p)
    hi = d.original_vocab_size + (d.n_extra if (cfg.get("hn_n_extra_tokens", 0) if isinstance(cfg, dict)
                                                else getattr(cfg, "hn_n_extra_tokens", 0)) else 0)
    ids = rng.integers(3, hi, size=(n_rows, seq), dtype=np.int64)
    ids[ids == d.pad_token_id] = 3 if d.pad_token_id != 3 else 4
    col = np.arange(seq)[None, :]
    out = np.where(col < lengths[:, None], ids, d.pad_token_id).astype(np.int32)
    if n_special:
        out[:n_special] = d.pad_token_id
    return out


# ---------------------------------------------------------------------------------------------
# Synthetic hn tokenizer + target-token strings whose retokenization IS a given surface-form matrix,
# so that a benchmark can start from surface forms (bytes) and still run the exact id workload above.
# ---------------------------------------------------------------------------------------------
_PIECE_ALPHABET = "".join(chr(c) for c in range(33, 127) if chr(c) not in "<>/[]|_#'\"\\`{}")   # 80 byte-chars that are plain ASCII


def _piece(i: int) -> str:
    a = _PIECE_ALPHABET
    n = len(a)
    return a[i % n] + a[(i // n) % n] + a[(i // (n * n)) % n]


def make_hn_unigram_model(cfg) -> dict:
    """A tokenizers-format Unigram model JSON with one 3-byte piece per id in [0, V0 + X): every piece has the
    same score and no shorter piece exists, so the Viterbi segmentation of a concatenation of pieces is that
    concatenation.  Ids 0..2 are special-token strings (never produced by make_surface_forms)."""
    d = HypernetDims.from_config(cfg)
    n = d.original_vocab_size + d.n_extra
    assert n <= len(_PIECE_ALPHABET) ** 3
    vocab = [["<unk>", 0.0], ["<s>", 0.0], ["</s>", 0.0]] + [[_piece(i), -10.0] for i in range(3, n)]
    return {"type": "Unigram", "unk_id": 0, "vocab": vocab, "byte_fallback": False}


# Which bare model the SOURCE model's tokenizer is on each workload (what zett/utils.py:681 calls per target token):
# Mistral-7B / TinyLlama ship a sentencepiece-style BPE with byte fallback (32 000 entries, ~31.7 k merges), Llama-3 a
# byte-level BPE of 128 k entries with ignore_merges and no byte fallback, XLM-R a Unigram model of 250 k pieces.
HN_MODEL_KIND = {
    "tiny": "unigram", "xlmr_gpt2": "unigram",
    "tinyllama_neox": "bpe_byte_fallback", "mistral_neox": "bpe_byte_fallback", "mistral_gpt2_32k": "bpe_byte_fallback",
    "llama3_256k": "bpe_ignore_merges",
}

_BPE_FALLBACK_CHARS = "_#|{}[]"                       # ASCII characters the BPE vocabulary does NOT hold: they leave as "<0xXX>" ids
_BPE_HEADS = _PIECE_ALPHABET[:44]                     # a piece is ONE head character followed by tail characters only, and every
_BPE_TAILS = _PIECE_ALPHABET[44:]                     # merge is (piece, tail character): no merge can cross a piece boundary


def make_hn_bpe_model(cfg, byte_fallback: bool = True, ignore_merges: bool = False, seed: int = 0):
    """A tokenizers-format BPE model JSON shaped like the source models' own tokenizers — one merge per multi-character
    entry (31.9 k merges at the 32 000-entry Mistral / TinyLlama size, 128 k at Llama-3's), pieces of 1-4 characters (1-5 from
    40 k entries: mean 3.6, so a target token of the workload's 2.4 pieces is ~8 bytes, as GPT-2 / NeoX tokens are) grown as
    a random prefix-closed tree in rank order, optional byte fallback — whose segmentation of a concatenation of pieces
    is that concatenation.  Returns (model, piece_of_id): piece_of_id[i] is the text that retokenizes to id i, for every
    i in [3, V0 + X).

    Layout of the ids: 0..2 special-token strings; with byte fallback the next 7 are the "<0xXX>" entries of the ASCII
    characters in _BPE_FALLBACK_CHARS (the vocabulary does not hold those characters, so the library emits the byte
    token: the fallback branch of merge_word runs on ~0.02 % of the positions, as often as the uniform id sampler of
    make_surface_forms draws one of those 7 ids); then the 44 head characters; then one entry per merge, in rank order.
    Entries that exist only so that BPE can start from characters — the tail characters and the other 249 byte tokens —
    take ids from V0 + X upwards: no surface form ever ends in one, so the id matrix stays inside the embedding matrix."""
    d = HypernetDims.from_config(cfg)
    n = d.original_vocab_size + d.n_extra
    rng = _rng(seed, "hn_bpe_model")
    vocab: Dict[str, int] = {"<unk>": 0, "<s>": 1, "</s>": 2}
    piece_of_id: Dict[int, str] = {}
    nxt = 3
    if byte_fallback:
        for ch in _BPE_FALLBACK_CHARS:
            vocab["<0x%02X>" % ord(ch)] = nxt
            piece_of_id[nxt] = ch
            nxt += 1
    pieces = []
    for ch in _BPE_HEADS:
        vocab[ch] = nxt
        piece_of_id[nxt] = ch
        pieces.append(ch)
        nxt += 1
    merges = []
    growable = list(pieces)
    nt = len(_BPE_TAILS)
    max_piece = 4 if n <= 40000 else 5
    while nxt < n:
        k = int(rng.integers(0, len(growable)))
        parent = growable[k]
        t = _BPE_TAILS[int(rng.integers(0, nt))]
        child = parent + t
        if child in vocab:
            continue
        vocab[child] = nxt
        piece_of_id[nxt] = child
        merges.append([parent, t])
        if len(child) < max_piece:
            growable.append(child)
        nxt += 1
    for ch in _BPE_TAILS:                               # start symbols only: beyond the embedding matrix
        vocab[ch] = nxt
        nxt += 1
    if byte_fallback:
        for b in range(256):
            key = "<0x%02X>" % b
            if key not in vocab:
                vocab[key] = nxt
                nxt += 1
    model = {"type": "BPE", "vocab": vocab, "merges": merges, "unk_token": "<unk>", "fuse_unk": bool(byte_fallback),
             "byte_fallback": bool(byte_fallback), "ignore_merges": bool(ignore_merges), "dropout": None,
             "continuing_subword_prefix": None, "end_of_word_suffix": None}
    return model, piece_of_id


def make_hn_model(workload_name: str, cfg):
    """(model JSON, piece_of_id or None) of the workload's synthetic hn tokenizer: the kind of HN_MODEL_KIND."""
    kind = HN_MODEL_KIND.get(workload_name, "unigram")
    if kind == "unigram":
        return make_hn_unigram_model(cfg), None
    return make_hn_bpe_model(cfg, byte_fallback=kind == "bpe_byte_fallback", ignore_merges=kind == "bpe_ignore_merges")


def tokens_for_surface_forms(cfg, ids: np.ndarray, piece_of_id=None):
    """Target-token strings (byte-level alphabet) that retokenize to the rows of `ids` under the synthetic hn model
    (make_hn_unigram_model when piece_of_id is None, else the model piece_of_id came with): the pieces of the non-pad ids
    of the row, concatenated.  Rows must be pad-free up to their length (what make_surface_forms produces with
    n_special = 0)."""
    d = HypernetDims.from_config(cfg)
    ids = np.asarray(ids)
    lengths = (ids != d.pad_token_id).sum(axis=1)
    assert ((np.arange(ids.shape[1])[None, :] < lengths[:, None]) == (ids != d.pad_token_id)).all(), "pads must be trailing"
    assert (lengths > 0).all(), "all-pad rows stand for special tokens: not representable as a byte string"
    cache: Dict[int, str] = {}
    out = []
    for row, ln in zip(ids.tolist(), lengths.tolist()):
        parts = []
        for i in row[:ln]:
            p = cache.get(i)
            if p is None:
                p = cache[i] = _piece(i) if piece_of_id is None else piece_of_id[i]
            parts.append(p)
        out.append("".join(parts))
    return out
