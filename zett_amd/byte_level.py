"""``convert_to_byte_level`` — rewrite any fast tokenizer into a byte-level one.

The step BEFORE the embedding-prediction path: it produces the
byte-level token list of the target tokenizer and the hn tokenizer whose bare model
``get_surface_form_matrix`` retokenizes with.  Host-only tokenizer-JSON surgery, restated from
the behaviour of the reference's ``zett/tokenizer_converters.py:78-406`` (same signature, same
return value ``(tokenizer, n_added_tokens | None)``, the tokenizer object is rewritten in place):

  1. every vocabulary entry becomes its UTF-8 bytes spelled in the byte-level alphabet
     (meta space characters and continuing-subword prefixes resolved first);
  2. ``<0xXX>`` byte-fallback tokens turn into the byte character they stand for, missing byte
     characters are appended ("fill bytes");
  3. optionally whitespace is made consistent: exactly the tokens {Ġ,Ċ,ĉ}·{Ġ,Ċ,ĉ}^i (i ≤ 15) may
     hold more than one whitespace character, everything else that does is retired to
     ``<unused_whitespace__N>``;
  4. optionally the special tokens of another tokenizer are spliced in at that tokenizer's ids;
  5. the model is rebuilt: Unigram scores are re-keyed (fill bytes get -1e5), BPE merges are
     re-spelled and extended so that every new token is reachable, WordPiece keeps its vocab;
  6. normalizer / pre-tokenizer become Prepend(" ") and Split(SPLIT_REGEX) + ByteLevel.

Where the reference iterates over a Python ``set`` (the order in which extra BPE merges are
emitted) this implementation iterates in sorted order, which makes the result deterministic.
"""
from __future__ import annotations

import copy
import json
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .surface_forms import BYTES_TO_CHARS_LIST, CHARS_TO_BYTES

NEGATIVE_INF_FILL_VALUE = -100_000           # zett/utils.py:23
SPLIT_REGEX = (r"'s|'t|'re|'ve|'m|'ll|'d| ?[\p{L}\p{M}]+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+")   # zett/utils.py:29
WHITESPACE_CHARS = ("Ġ", "Ċ", "ĉ")            # space, newline, tab in the byte-level alphabet
MAX_WHITESPACE_RUN = 15


def _spell(raw: bytes) -> str:
    return "".join(BYTES_TO_CHARS_LIST[b] for b in raw)


def _n_whitespace(s: str) -> int:
    return sum(ch in WHITESPACE_CHARS for ch in s)


def _has_byte_level_pretokenizer(data: dict) -> bool:
    pre = data.get("pre_tokenizer") or {}
    if pre.get("type") == "ByteLevel":
        return True
    return pre.get("type") == "Sequence" and any(p.get("type") == "ByteLevel" for p in pre.get("pretokenizers", []))


def _byte_speller(tokenizer, data: dict) -> Tuple[Callable[[str], str], Optional[str]]:
    """token -> byte-level spelling, plus the continuing-subword prefix that was resolved (if any)."""
    if _has_byte_level_pretokenizer(data):
        assert len(data["model"].get("continuing_subword_prefix") or "") == 0
        return (lambda tok: tok), None

    backend = tokenizer._tokenizer
    probe = " test"
    if backend.normalizer is not None:
        probe = backend.normalizer.normalize_str(probe)
    if backend.pre_tokenizer is not None:
        probe = backend.pre_tokenizer.pre_tokenize_str(probe)[0][0]
    meta = probe[0] if (probe[0] != " " and probe != "test") else None
    prefix = data["model"].get("continuing_subword_prefix")

    def spell(tok: str) -> str:
        if meta is not None:
            tok = tok.replace(meta, " ")
        if prefix is not None:
            tok = tok[len(prefix):] if tok.startswith(prefix) else " " + tok
        return _spell(tok.encode("utf-8"))

    return spell, prefix


def _allowed_whitespace_tokens() -> List[str]:
    return [tail + head * i for head in WHITESPACE_CHARS for i in range(1, MAX_WHITESPACE_RUN + 1) for tail in WHITESPACE_CHARS]


def _rebuild_unigram(original_vocab, surface_forms, spell, whitespace_consistent):
    scores: Dict[str, float] = {}
    for piece, score in original_vocab:
        scores[spell(piece)] = score
    for ch in CHARS_TO_BYTES:
        scores.setdefault(ch, NEGATIVE_INF_FILL_VALUE)            # fill bytes must never be preferred
    if whitespace_consistent:
        for key in [k for k in scores if _n_whitespace(k) > 1]:
            del scores[key]
    return [(sf, scores.get(sf, 0.0)) for sf in surface_forms]


class _MergeSurgeon:
    """Re-spells BPE merges and derives the extra merges that keep every new token reachable."""

    def __init__(self, merges_json, spell, whitespace_consistent: bool):
        self.whitespace_consistent = whitespace_consistent
        self.producers: Dict[str, List[Tuple[str, str]]] = {}
        self.merges: List[str] = []
        for entry in merges_json:
            left, right = entry.split(" ") if isinstance(entry, str) else entry
            left, right = spell(left), spell(right)
            product = left + right
            if whitespace_consistent and _n_whitespace(product) > 1:
                continue
            self.producers.setdefault(product, []).append((left, right))
            self.merges.append(f"{left} {right}")

    def atoms(self, token: str) -> set:
        """Closure of `token` under un-merging: the pieces no merge produces."""
        parts = {token}
        while True:
            splittable = next((p for p in parts if p in self.producers), None)
            if splittable is None:
                return parts
            parts.discard(splittable)
            for left, right in self.producers[splittable]:
                parts.add(left)
                parts.add(right)

    @staticmethod
    def chain(token: str, known: set) -> Tuple[List[str], set]:
        """Merges that build `token` from single characters, left to right, and the new
        intermediate tokens they create."""
        merges: List[str] = []
        created = set()
        pieces = list(token)
        while len(pieces) > 1:
            snapshot = list(pieces)
            for left, right in zip(snapshot, snapshot[1:]):
                applied = False
                i = 0
                while i < len(pieces) - 1:
                    if pieces[i] == left and pieces[i + 1] == right:
                        pieces[i] = left + right
                        del pieces[i + 1]
                        applied = True
                    i += 1
                if applied:
                    merges.append(f"{left} {right}")
                    if left + right not in known:
                        created.add(left + right)
        return merges, created


def _fix_post_processor(post: dict, surface_forms: List[str]) -> None:
    kind = post.get("type")
    if kind == "TemplateProcessing":
        for entry in post["special_tokens"].values():
            entry["ids"] = [surface_forms.index(t) for t in entry["tokens"]]
    elif kind == "RobertaProcessing":
        post["sep"][1] = surface_forms.index(post["sep"][0])
        post["cls"][1] = surface_forms.index(post["cls"][0])
    elif kind == "Sequence":
        for inner in post["processors"]:
            _fix_post_processor(inner, surface_forms)


def convert_to_byte_level(tokenizer, keep_normalizer=False, keep_pretokenizer=False,
                          make_whitespace_consistent=False, match_special_tokens_to=None):
    from tokenizers import Tokenizer, decoders

    match_data = json.loads(match_special_tokens_to._tokenizer.to_str()) if match_special_tokens_to is not None else {}
    data = json.loads(tokenizer._tokenizer.to_str())
    data.pop("added_tokens", None)                      # they become ordinary vocabulary entries
    original = copy.deepcopy(data)
    original_length = len(tokenizer)
    indices_preserved = True

    spell, prefix = _byte_speller(tokenizer, data)
    already_byte_level = _has_byte_level_pretokenizer(data)
    if prefix is not None:
        data["model"]["continuing_subword_prefix"] = ""

    own_specials = set(tokenizer.all_special_tokens)
    surface_forms = [tok if tok in own_specials else spell(tok)
                     for tok in tokenizer.convert_ids_to_tokens(range(len(tokenizer)))]

    # <0xXX> byte-fallback tokens -> the byte character (the reference covers bytes 0..254)
    fallback_names: Dict[str, str] = {}
    if data["model"].get("byte_fallback"):
        fallback_names = {f"<0x{b:02X}>": BYTES_TO_CHARS_LIST[b] for b in range(255)}
        present = set(surface_forms)
        for i, sf in enumerate(surface_forms):
            ch = fallback_names.get(sf)
            if ch is not None and ch not in present:
                surface_forms[i] = ch

    seen = set(surface_forms)
    fill = [ch for ch in CHARS_TO_BYTES if ch not in seen]
    if fill:
        print(f"WARNING: {len(fill)} bytes not in surface forms.")
        surface_forms += fill

    if make_whitespace_consistent:
        allowed = _allowed_whitespace_tokens()
        for i, sf in enumerate(surface_forms):
            if sf in allowed:
                allowed.remove(sf)
            elif _n_whitespace(sf) > 1 or len(sf.strip()) == 0:
                surface_forms[i] = f"<unused_whitespace__{i}>"
        surface_forms += allowed

    if match_special_tokens_to is not None:
        other_tokens = list(match_special_tokens_to.all_special_tokens)
        other_ids = list(match_special_tokens_to.all_special_ids)
        drop = own_specials | set(other_tokens)
        surface_forms = [sf for sf in surface_forms if sf not in drop]
        for k in np.argsort(other_ids):
            surface_forms.insert(other_ids[k], other_tokens[k])
        special_tokens = other_tokens
        indices_preserved = False
    else:
        special_tokens = list(tokenizer.all_special_tokens)

    prepend_space = {"type": "Prepend", "prepend": " "}
    split_then_bytes = {"type": "Sequence", "pretokenizers": [
        {"type": "Split", "pattern": {"Regex": SPLIT_REGEX}, "behavior": "Removed", "invert": True},
        {"type": "ByteLevel", "add_prefix_space": False, "trim_offsets": True, "use_regex": False}]}
    if not keep_normalizer:
        data["normalizer"] = prepend_space
    else:
        previous = data.get("normalizer")
        data["normalizer"] = {"type": "Sequence", "normalizers": ([previous] if previous is not None else []) + [prepend_space]}
    if not keep_pretokenizer:
        data["pre_tokenizer"] = split_then_bytes
    elif not already_byte_level:
        previous = data.get("pre_tokenizer")
        split_then_bytes["use_regex"] = False
        data["pre_tokenizer"] = {"type": "Sequence",
                                 "pretokenizers": ([previous] if previous is not None else []) + [split_then_bytes]}

    model_type = type(tokenizer._tokenizer.model).__name__
    if model_type == "Unigram":
        data["model"]["vocab"] = _rebuild_unigram(original["model"]["vocab"], surface_forms, spell, make_whitespace_consistent)
    elif model_type == "BPE":
        surgeon = _MergeSurgeon(data["model"]["merges"], spell, make_whitespace_consistent)
        known = set(surface_forms)
        to_check = surface_forms[original_length:] if already_byte_level else surface_forms
        unreachable = set()
        for tok in to_check:
            if tok in special_tokens or tok in fallback_names or tok.startswith("<unused_whitespace__"):
                continue
            unreachable.update(part for part in surgeon.atoms(tok) if len(part) > 1)
        emitted, before, after, new_vocab = set(), [], [], set()
        for tok in sorted(unreachable):
            chain, created = surgeon.chain(tok, known)
            new_vocab |= created
            late = make_whitespace_consistent and _n_whitespace(tok) > 1
            for merge in chain:
                if merge not in emitted:
                    emitted.add(merge)
                    (after if late else before).append(merge)
        surface_forms += sorted(new_vocab)
        data["model"]["vocab"] = {sf: i for i, sf in enumerate(surface_forms)}
        data["model"]["merges"] = before + surgeon.merges + after
    elif model_type == "WordPiece":
        data["model"]["vocab"] = {sf: i for i, sf in enumerate(surface_forms)}
    else:
        raise ValueError(f"Unknown model type: {type(tokenizer._tokenizer.model)}")

    if match_special_tokens_to is not None and match_data.get("post_processor") is not None:
        _fix_post_processor(match_data["post_processor"], surface_forms)
        data["post_processor"] = match_data["post_processor"]

    tokenizer._tokenizer = Tokenizer.from_str(json.dumps(data))
    tokenizer._tokenizer.decoder = decoders.ByteLevel()
    if match_special_tokens_to is not None:
        for attr in ("eos_token", "pad_token", "sep_token", "unk_token", "bos_token", "cls_token", "mask_token"):
            setattr(tokenizer, attr, getattr(match_special_tokens_to, attr))
    return tokenizer, (len(tokenizer) - original_length if indices_preserved else None)
