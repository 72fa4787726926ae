"""Oracle of the retokenizer: byte table + BPE merge + Unigram Viterbi.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Never imported by zett_amd/.

Restates, in plain Python (and in C, retok_ref.c, for the larger fixtures):

* the byte <-> printable-character table the reference uses for byte-level tokens
  (zett/utils.py:351-609 ``CHARS_TO_BYTES``; it is the GPT-2 ``bytes_to_unicode`` map);
* ``get_surface_form_matrix`` (zett/utils.py:651-689);
* what ``tokenizer_to_use._tokenizer.model.tokenize(token)`` does (zett/utils.py:681):
  the HF ``tokenizers`` library's ``BPE::tokenize`` / ``Unigram::tokenize`` /
  ``WordPiece::tokenize`` on the bare model (zett/tokenizer_converters.py:370-373 carries
  WordPiece hn tokenizers through convert_to_byte_level).  That library (pinned by the image at 0.22.2; reference requirements pull
  0.20.x) is a third-party Rust dependency and is NOT under /root/reference; its
  published algorithm is restated here and pinned by differential tests against the
  installed wheel (tests/test_retok_oracle.py) and by the golden fixtures
  ``tests/golden/retok_*.json`` produced by the reference function itself.

Models are described by :class:`RetokModel` in RAW BYTE space: every vocabulary piece
is the byte string obtained by mapping its byte-level characters through the table
(pieces containing any other character can never match a byte-level token and are
dropped by ``model_from_tokenizer_json``).
"""
from __future__ import annotations

import ctypes as C
import heapq
import json
import os
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

BPE, UNIGRAM, WORDPIECE = 0, 1, 2
K_UNK_PENALTY = 10.0      # tokenizers: unigram/model.rs kUnkPenalty


def bytes_to_chars_table() -> List[str]:
    """byte -> character (GPT-2 bytes_to_unicode): printable Latin-1 bytes map to
    themselves, the other 68 bytes to U+0100.. in byte order."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table = [""] * 256
    extra = 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


BYTES_TO_CHARS = bytes_to_chars_table()
CHARS_TO_BYTES = {c: b for b, c in enumerate(BYTES_TO_CHARS)}


def token_to_bytes(token: str) -> bytes:
    """zett/utils.py:675 — KeyError on a character outside the table."""
    return bytes([CHARS_TO_BYTES[c] for c in token])


@dataclass
class RetokModel:
    kind: int
    pieces: List[bytes]                    # raw-byte pieces (duplicates allowed: last wins)
    piece_ids: List[int]
    piece_scores: Optional[List[float]] = None      # Unigram
    merges: List[Tuple[int, int, int]] = field(default_factory=list)   # (left id, right id, new id); rank = index
    unk_id: int = -1
    fuse_unk: bool = False
    byte_fallback: bool = False
    byte_fallback_ids: Optional[List[int]] = None   # 256 ids of "<0xXX>" (-1 = absent)
    ignore_merges: bool = False
    specials: List[bytes] = field(default_factory=list)     # hn all_special_tokens (raw bytes)
    special_ids: List[int] = field(default_factory=list)
    min_score: float = 0.0                 # Unigram: min over ALL vocabulary scores (tokenizers Unigram::from)
    # WordPiece: vocabulary entries that start with continuing_subword_prefix, prefix stripped — what a lookup at start > 0
    # can match (`pieces` holds every entry as listed: what a lookup at start == 0 can match)
    cont_pieces: List[bytes] = field(default_factory=list)
    cont_ids: List[int] = field(default_factory=list)
    max_chars: int = 100                   # max_input_chars_per_word


def _piece_bytes(piece: str) -> Optional[bytes]:
    try:
        return token_to_bytes(piece)
    except KeyError:
        return None


def model_from_tokenizer_json(data: dict, special_tokens: Sequence[str] = (), special_ids: Sequence[int] = ()) -> RetokModel:
    """Build the raw-byte model from a ``tokenizer.json`` dict (its "model" section)."""
    m = data["model"] if "model" in data else data
    kind = m.get("type")
    if kind != "WordPiece" and (m.get("continuing_subword_prefix") or m.get("end_of_word_suffix")):
        raise NotImplementedError("continuing_subword_prefix / end_of_word_suffix")
    if m.get("dropout"):
        raise NotImplementedError("BPE dropout")
    bf_ids = None
    specials = [(b, i) for b, i in ((_piece_bytes(s), i) for s, i in zip(special_tokens, special_ids)) if b is not None]
    if kind == "BPE" or (kind is None and "merges" in m):
        vocab: Dict[str, int] = m["vocab"]
        pieces, ids = [], []
        for tok, i in vocab.items():
            b = _piece_bytes(tok)
            if b is not None and len(b) > 0:
                pieces.append(b)
                ids.append(int(i))
        merges = []
        for mg in m["merges"]:
            a, b = mg.split(" ") if isinstance(mg, str) else mg
            # tokenizers BPE::new: pair ids and the id of the concatenation must exist
            merges.append((vocab[a], vocab[b], vocab[a + b]))
        unk = m.get("unk_token")
        if m.get("byte_fallback"):
            bf_ids = [int(vocab.get(f"<0x{b:02X}>", -1)) for b in range(256)]
        return RetokModel(kind=BPE, pieces=pieces, piece_ids=ids, merges=merges,
                          unk_id=int(vocab[unk]) if unk is not None else -1, fuse_unk=bool(m.get("fuse_unk", False)),
                          byte_fallback=bool(m.get("byte_fallback", False)), byte_fallback_ids=bf_ids,
                          ignore_merges=bool(m.get("ignore_merges", False)),
                          specials=[b for b, _ in specials], special_ids=[i for _, i in specials])
    if kind == "Unigram":
        vocab_list = m["vocab"]
        pieces, ids, scores = [], [], []
        str_to_id = {}
        for i, (tok, score) in enumerate(vocab_list):
            str_to_id[tok] = i
            b = _piece_bytes(tok)
            if b is not None and len(b) > 0:
                pieces.append(b)
                ids.append(i)
                scores.append(float(score))
        if m.get("byte_fallback"):
            bf_ids = [int(str_to_id.get(f"<0x{b:02X}>", -1)) for b in range(256)]
        unk_id = m.get("unk_id")
        return RetokModel(kind=UNIGRAM, pieces=pieces, piece_ids=ids, piece_scores=scores,
                          unk_id=-1 if unk_id is None else int(unk_id), fuse_unk=True,
                          byte_fallback=bool(m.get("byte_fallback", False)), byte_fallback_ids=bf_ids,
                          specials=[b for b, _ in specials], special_ids=[i for _, i in specials],
                          min_score=min(float(s) for _, s in vocab_list) if vocab_list else 0.0)
    if kind == "WordPiece":
        vocab = m["vocab"]
        prefix = m.get("continuing_subword_prefix") or ""
        pieces, ids, cont, cont_ids = [], [], [], []
        for tok, i in vocab.items():
            b = _piece_bytes(tok)
            if b is not None and len(b) > 0:
                pieces.append(b)
                ids.append(int(i))
            if tok.startswith(prefix):               # (prefix + substring) is looked up: the substring alone must be byte-level
                b = _piece_bytes(tok[len(prefix):])
                if b is not None and len(b) > 0:
                    cont.append(b)
                    cont_ids.append(int(i))
        unk = m.get("unk_token")
        return RetokModel(kind=WORDPIECE, pieces=pieces, piece_ids=ids, unk_id=int(vocab[unk]) if unk in vocab else -1,
                          specials=[b for b, _ in specials], special_ids=[i for _, i in specials],
                          cont_pieces=cont, cont_ids=cont_ids, max_chars=int(m.get("max_input_chars_per_word", 100)))
    raise NotImplementedError(f"hn tokenizer model type {kind!r}")


def model_from_hf_tokenizer(tok) -> RetokModel:
    """From a transformers fast tokenizer (what the reference passes as tokenizer_to_use)."""
    data = json.loads(tok._tokenizer.to_str())
    specials = list(tok.all_special_tokens)
    return model_from_tokenizer_json(data, specials, [tok.convert_tokens_to_ids(s) for s in specials])


# ---------------------------------------------------------------------------------------
# pure-Python twin (small cases; the readable statement of the algorithm)
# ---------------------------------------------------------------------------------------
def _fallback_ids(model: RetokModel, raw: bytes) -> Optional[List[int]]:
    """ids of the "<0xXX>" tokens of the UTF-8 bytes of the byte-level characters of
    `raw` (tokenizers formats the bytes of the *string*, i.e. of the printable chars)."""
    out = []
    for b in raw:
        for u in BYTES_TO_CHARS[b].encode("utf-8"):
            i = model.byte_fallback_ids[u] if model.byte_fallback_ids else -1
            if i < 0:
                return None
            out.append(i)
    return out


def bpe_tokenize(model: RetokModel, raw: bytes) -> List[int]:
    """tokenizers ``BPE::tokenize`` (no dropout): ``merge_word`` + ``Word::merge_all``."""
    if not raw:
        return []
    vocab = {}
    for p, i in zip(model.pieces, model.piece_ids):
        vocab[p] = i
    if model.ignore_merges and raw in vocab:
        return [vocab[raw]]
    merges = {}
    for rank, (a, b, n) in enumerate(model.merges):
        merges[(a, b)] = (rank, n)               # collected into a HashMap: the LAST occurrence of a pair wins
    sym: List[int] = []
    unk: Optional[int] = None                    # a pending unk symbol
    for b in raw:
        s = bytes([b])
        if s in vocab:
            if unk is not None:
                sym.append(unk)
                unk = None
            sym.append(vocab[s])
            continue
        if model.byte_fallback:
            fb = _fallback_ids(model, s)
            if fb is not None:
                sym.extend(fb)                   # NB: a pending unk is not flushed first (library behaviour)
                continue
        if model.unk_id >= 0:
            if unk is not None and not model.fuse_unk:
                sym.append(unk)
            unk = model.unk_id
        # no unk token: the character is dropped
    if unk is not None:
        sym.append(unk)

    n = len(sym)
    c = list(sym)
    prev = [i - 1 for i in range(n)]
    nxt = [i + 1 if i + 1 < n else -1 for i in range(n)]
    alive = [True] * n
    heap = []
    for i in range(n - 1):
        m = merges.get((c[i], c[i + 1]))
        if m is not None:
            heapq.heappush(heap, (m[0], i, m[1]))
    while heap:
        rank, pos, new_id = heapq.heappop(heap)
        if not alive[pos] or nxt[pos] == -1:
            continue
        r = nxt[pos]
        m = merges.get((c[pos], c[r]))
        if m is None or m[1] != new_id:          # expired entry (compared by new id only)
            continue
        c[pos] = new_id
        alive[r] = False
        nxt[pos] = nxt[r]
        if nxt[r] != -1:
            prev[nxt[r]] = pos
        if prev[pos] >= 0:
            m = merges.get((c[prev[pos]], c[pos]))
            if m is not None:
                heapq.heappush(heap, (m[0], prev[pos], m[1]))
        if nxt[pos] != -1:
            m = merges.get((c[pos], c[nxt[pos]]))
            if m is not None:
                heapq.heappush(heap, (m[0], pos, m[1]))
    return [c[i] for i in range(n) if alive[i]]


def unigram_tokenize(model: RetokModel, raw: bytes) -> List[int]:
    """tokenizers ``Unigram::tokenize``: Viterbi (``encode_optimized``) + piece -> id."""
    if not raw:
        return []
    table = {}
    for p, i, s in zip(model.pieces, model.piece_ids, model.piece_scores):
        table[p] = (i, s)                        # duplicates: the last entry wins
    n = len(raw)
    maxlen = max((len(p) for p in table), default=0)
    unk_score = model.min_score - K_UNK_PENALTY
    best_score = [0.0] * (n + 1)
    best_start = [-1] * (n + 1)
    best_id = [-1] * (n + 1)
    for s in range(n):
        base = best_score[s]
        has_single = False
        for e in range(s + 1, min(n, s + maxlen) + 1):
            hit = table.get(raw[s:e])
            if hit is None:
                continue
            cand = hit[1] + base
            if best_start[e] == -1 or cand > best_score[e]:
                best_score[e], best_start[e], best_id[e] = cand, s, hit[0]
            if e == s + 1:
                has_single = True
        if not has_single:
            if model.unk_id < 0:
                raise RuntimeError("Encountered an unknown token but `unk_id` is missing")
            e = s + 1
            cand = unk_score + base
            if best_start[e] == -1 or cand > best_score[e]:
                best_score[e], best_start[e], best_id[e] = cand, s, -2    # -2 = unknown
    # backtrack, fusing consecutive unknown pieces
    spans = []
    e = n
    while e > 0:
        s = best_start[e]
        spans.append((s, e, best_id[e]))
        e = s
    spans.reverse()
    out: List[int] = []
    i = 0
    while i < len(spans):
        s, e, pid = spans[i]
        if pid != -2:
            out.append(pid)
            i += 1
            continue
        j = i
        while j + 1 < len(spans) and spans[j + 1][2] == -2:
            j += 1
        fused = raw[s:spans[j][1]]
        fb = _fallback_ids(model, fused) if model.byte_fallback else None
        if fb is not None:
            out.extend(fb)
        else:
            out.append(model.unk_id)
        i = j + 1
    return out


def wordpiece_tokenize(model: RetokModel, raw: bytes) -> List[int]:
    """tokenizers ``WordPiece::tokenize``: a word of more than max_input_chars_per_word characters is [unk]; otherwise
    greedy longest match from each start (continuing pieces carry the prefix); a start without any match makes the WHOLE
    word [unk].  One byte-level character is one raw byte, so characters = bytes."""
    if not raw:
        return []

    def unk():
        if model.unk_id < 0:
            raise RuntimeError("WordPiece error: Missing [UNK] token from the vocabulary")
        return [model.unk_id]

    if len(raw) > model.max_chars:
        return unk()
    first = dict(zip(model.pieces, model.piece_ids))
    cont = dict(zip(model.cont_pieces, model.cont_ids))
    out, start = [], 0
    while start < len(raw):
        end = len(raw)
        hit = None
        while start < end:
            hit = (first if start == 0 else cont).get(raw[start:end])
            if hit is not None:
                break
            end -= 1
        if hit is None:
            return unk()
        out.append(hit)
        start = end
    return out


def tokenize(model: RetokModel, raw: bytes) -> List[int]:
    if model.kind == WORDPIECE:
        return wordpiece_tokenize(model, raw)
    return bpe_tokenize(model, raw) if model.kind == BPE else unigram_tokenize(model, raw)


def surface_form_matrix_py(model: RetokModel, tokens: Sequence[str], maxlen: int, pad_id: int, padding: int = 0):
    """get_surface_form_matrix (zett/utils.py:651-689) — pure Python."""
    out = np.full((len(tokens) + padding, maxlen), pad_id, dtype=np.int32)
    special = {b: i for b, i in zip(model.specials, model.special_ids)}
    n_truncated = 0
    for r, token in enumerate(tokens):
        raw_special = _piece_bytes(token)
        if raw_special is not None and raw_special in special:       # :671-673
            out[r, 0] = special[raw_special]
            continue
        raw = token_to_bytes(token)                                   # :675 (KeyError)
        ids = tokenize(model, raw)                                    # :681
        if len(ids) > maxlen:                                         # :683-685
            ids = ids[:maxlen]
            n_truncated += 1
        out[r, :len(ids)] = ids
    return out, n_truncated


# ---------------------------------------------------------------------------------------
# C twin (retok_ref.c) through ctypes
# ---------------------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libretok_ref.so")
_clib = None


def _load_c():
    global _clib
    if _clib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "retok_ref.c")):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        lib = C.CDLL(_SO)
        lib.retok_ref_new.restype = C.c_void_p
        lib.retok_ref_new.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.retok_ref_free.argtypes = [C.c_void_p]
        lib.retok_ref_set_wordpiece.restype = None
        lib.retok_ref_set_wordpiece.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.retok_ref_surface_forms.restype = C.c_int
        lib.retok_ref_surface_forms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                                C.c_void_p, C.POINTER(C.c_int64)]
        _clib = lib
    return _clib


def _blob(items: Sequence[bytes]):
    offs = np.zeros(len(items) + 1, dtype=np.int32)
    if items:
        offs[1:] = np.cumsum([len(b) for b in items])
    data = np.frombuffer(b"".join(items) or b"\0", dtype=np.uint8).copy()
    return data, offs


def model_arrays(model: RetokModel) -> dict:
    """Flat numpy arrays of a model (also what the product's host code must produce)."""
    pb, po = _blob(model.pieces)
    sb, so = _blob(model.specials)
    return dict(
        kind=model.kind, n_pieces=len(model.pieces), piece_bytes=pb, piece_offsets=po,
        piece_ids=np.asarray(model.piece_ids, dtype=np.int32).reshape(-1),
        piece_scores=None if model.piece_scores is None else np.asarray(model.piece_scores, dtype=np.float64),
        n_merges=len(model.merges), merges=np.asarray(model.merges, dtype=np.int32).reshape(-1, 3),
        unk_id=model.unk_id, fuse_unk=int(model.fuse_unk), byte_fallback=int(model.byte_fallback),
        byte_fallback_ids=None if model.byte_fallback_ids is None else np.asarray(model.byte_fallback_ids, dtype=np.int32),
        ignore_merges=int(model.ignore_merges), min_score=float(model.min_score),
        n_special=len(model.specials), special_bytes=sb, special_offsets=so,
        special_ids=np.asarray(model.special_ids, dtype=np.int32).reshape(-1))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def surface_form_matrix_c(model: RetokModel, tokens: Sequence[str], maxlen: int, pad_id: int, padding: int = 0):
    """get_surface_form_matrix through the C oracle (fast enough for 50k-token vocabularies)."""
    lib = _load_c()
    a = model_arrays(model)
    raws = [token_to_bytes(t) for t in tokens]
    data, offs = _blob(raws)
    h = lib.retok_ref_new(a["kind"], a["n_pieces"], _ptr(a["piece_bytes"]), _ptr(a["piece_offsets"]), _ptr(a["piece_ids"]),
                          _ptr(a["piece_scores"]), a["n_merges"], _ptr(a["merges"]), a["unk_id"], a["fuse_unk"],
                          a["byte_fallback"], _ptr(a["byte_fallback_ids"]), a["ignore_merges"], a["min_score"],
                          a["n_special"], _ptr(a["special_bytes"]), _ptr(a["special_offsets"]), _ptr(a["special_ids"]))
    try:
        if model.kind == WORDPIECE:
            cb, co = _blob(model.cont_pieces)
            ci = np.asarray(model.cont_ids, dtype=np.int32).reshape(-1)
            lib.retok_ref_set_wordpiece(h, len(model.cont_pieces), _ptr(cb), _ptr(co), _ptr(ci), int(model.max_chars))
        out = np.full((len(tokens) + padding, maxlen), pad_id, dtype=np.int32)
        ntr = C.c_int64(0)
        rc = lib.retok_ref_surface_forms(h, _ptr(data), _ptr(offs), len(tokens), maxlen, pad_id, _ptr(out), C.byref(ntr))
        if rc != 0:
            raise RuntimeError("Encountered an unknown token but `unk_id` is missing")
        return out, int(ntr.value)
    finally:
        lib.retok_ref_free(h)
