"""``get_surface_form_matrix`` on MI355X — drop-in for zett/utils.py:651-689.

    matrix, n_truncated = get_surface_form_matrix(tokens_or_tokenizer, maxlen, tokenizer_to_use)

expresses every (byte-level) target token as at most ``maxlen`` ids of the source
model's ("hn") tokenizer, pad-filled, exactly as the reference does — but the per-token
Python loop around ``tokenizer_to_use._tokenizer.model.tokenize`` is replaced by two
HIP kernels behind ``zett_retokenize`` (byte-table gather + scan, then BPE merge /
Unigram Viterbi / WordPiece longest match per token; zett_amd/csrc/retok.hip.h).  This module only flattens the
hn tokenizer's bare model (vocabulary, merges / scores, flags, special tokens) into
the arrays the C ABI takes.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib


# ---- the byte <-> character table (reference zett/utils.py:351-609) -----------------------
def _build_byte_table() -> Tuple[List[str], Dict[str, int]]:
    """Printable Latin-1 bytes stand for themselves; the remaining 68 bytes (controls,
    space, DEL..0xA0, soft hyphen) are numbered upwards from U+0100 in byte order."""
    b2c: List[str] = []
    shifted = 0
    for b in range(256):
        if 0x21 <= b <= 0x7E or 0xA1 <= b <= 0xAC or 0xAE <= b <= 0xFF:
            b2c.append(chr(b))
        else:
            b2c.append(chr(0x100 + shifted))
            shifted += 1
    return b2c, {c: b for b, c in enumerate(b2c)}


BYTES_TO_CHARS_LIST, CHARS_TO_BYTES = _build_byte_table()
BYTES_TO_CHARS = dict(enumerate(BYTES_TO_CHARS_LIST))
_TRANSLATE = {ord(c): b for c, b in CHARS_TO_BYTES.items()}


def _raw(piece: str) -> Optional[bytes]:
    """Byte string of a byte-level piece, or None if it holds a character outside the table."""
    try:
        return bytes(_TRANSLATE[ord(ch)] for ch in piece)
    except KeyError:
        return None


# ---- flattening the hn tokenizer ---------------------------------------------------------------
@dataclass
class HnTokenizerSpec:
    """The hn tokenizer's bare model in the layout of ``zett_retok_model`` (include/zett_hip.h)."""
    kind: int
    piece_bytes: np.ndarray          # uint8
    piece_offsets: np.ndarray        # int32 [n_pieces + 1]
    piece_ids: np.ndarray            # int32
    piece_scores: Optional[np.ndarray]
    unigram_min_score: float
    merges: np.ndarray               # int32 [n_merges, 3]
    unk_id: int
    fuse_unk: bool
    byte_fallback: bool
    byte_fallback_ids: Optional[np.ndarray]
    ignore_merges: bool
    special_bytes: np.ndarray
    special_offsets: np.ndarray
    special_ids: np.ndarray
    pad_token_id: int
    special_tokens: Tuple[str, ...] = ()
    # special tokens the device table cannot hold (a character outside the byte-level alphabet, e.g. a space or
    # '▁' inside '<|begin▁of▁sentence|>'): matched on the host BEFORE the byte lookup, as the reference does
    # (zett/utils.py:671-673 tests `token in all_special_tokens` on the character string)
    host_specials: Tuple[Tuple[str, int], ...] = ()
    piece_continuing: Optional[np.ndarray] = None      # uint8 per piece, WordPiece only (include/zett_hip.h)
    max_input_chars_per_word: int = 100

    @staticmethod
    def _pack(items: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
        offsets = np.zeros(len(items) + 1, dtype=np.int32)
        if len(items):
            np.cumsum(np.fromiter(map(len, items), dtype=np.int64, count=len(items)), out=offsets[1:])
        blob = np.frombuffer(b"".join(items) or b"\0", dtype=np.uint8).copy()
        return blob, offsets

    @classmethod
    def from_model_json(cls, model: dict, special_tokens: Sequence[str], special_ids: Sequence[int],
                        pad_token_id: int) -> "HnTokenizerSpec":
        kind_name = model.get("type") or ("BPE" if "merges" in model else None)
        if kind_name != "WordPiece" and (model.get("continuing_subword_prefix") or model.get("end_of_word_suffix")):
            raise NotImplementedError("BPE hn tokenizers with a continuing_subword_prefix / end_of_word_suffix")
        if model.get("dropout"):
            raise NotImplementedError("BPE dropout")
        pieces: List[bytes] = []
        ids: List[int] = []
        scores: Optional[List[float]] = None
        merges = np.zeros((0, 3), dtype=np.int32)
        bf_ids = None
        min_score = 0.0
        continuing: Optional[List[int]] = None
        if kind_name == "WordPiece":
            # tokenizers WordPiece::tokenize (zett/utils.py:681 calls whatever model the hn tokenizer has;
            # zett/tokenizer_converters.py:370-373 carries WordPiece through, with the continuing prefix emptied): a lookup at
            # the start of a word matches an entry as listed, a lookup further in matches prefix + substring — every entry is
            # listed as it is, and every entry that starts with the prefix once more with the prefix stripped (flag 1)
            vocab = model["vocab"]
            prefix = model.get("continuing_subword_prefix") or ""
            continuing = []
            for piece, i in vocab.items():
                raw = _raw(piece)
                if raw:
                    pieces.append(raw); ids.append(int(i)); continuing.append(0)
                if piece.startswith(prefix):
                    raw = _raw(piece[len(prefix):])
                    if raw:
                        pieces.append(raw); ids.append(int(i)); continuing.append(1)
            unk = model.get("unk_token")
            unk_id = int(vocab[unk]) if unk in vocab else -1
            fuse_unk = False
            kind = _lib.RETOK_WORDPIECE
        elif kind_name == "BPE":
            vocab: Dict[str, int] = model["vocab"]
            for piece, i in vocab.items():
                raw = _raw(piece)
                if raw:
                    pieces.append(raw)
                    ids.append(int(i))
            rows = []
            for entry in model.get("merges", []):
                left, right = entry.split(" ") if isinstance(entry, str) else entry
                rows.append((vocab[left], vocab[right], vocab[left + right]))
            if rows:
                merges = np.asarray(rows, dtype=np.int32)
            unk = model.get("unk_token")
            unk_id = int(vocab[unk]) if unk is not None else -1
            fuse_unk = bool(model.get("fuse_unk", False))
            if model.get("byte_fallback"):
                bf_ids = np.asarray([vocab.get("<0x%02X>" % b, -1) for b in range(256)], dtype=np.int32)
            kind = _lib.RETOK_BPE
        elif kind_name == "Unigram":
            scores = []
            lookup: Dict[str, int] = {}
            listing = model["vocab"]
            for i, (piece, score) in enumerate(listing):
                lookup[piece] = i
                raw = _raw(piece)
                if raw:
                    pieces.append(raw)
                    ids.append(i)
                    scores.append(float(score))
            min_score = min((float(s) for _, s in listing), default=0.0)
            unk_id = -1 if model.get("unk_id") is None else int(model["unk_id"])
            fuse_unk = True
            if model.get("byte_fallback"):
                bf_ids = np.asarray([lookup.get("<0x%02X>" % b, -1) for b in range(256)], dtype=np.int32)
            kind = _lib.RETOK_UNIGRAM
        else:
            raise NotImplementedError(f"hn tokenizer model type {kind_name!r} (tokenizers has BPE, Unigram, WordPiece and "
                                      "WordLevel; convert_to_byte_level itself refuses anything but the first three)")
        sp = [(r, int(i)) for r, i in ((_raw(s), i) for s, i in zip(special_tokens, special_ids)) if r]
        pb, po = cls._pack(pieces)
        sb, so = cls._pack([r for r, _ in sp])
        return cls(kind=kind, piece_bytes=pb, piece_offsets=po, piece_ids=np.asarray(ids, dtype=np.int32),
                   piece_scores=None if scores is None else np.asarray(scores, dtype=np.float64),
                   unigram_min_score=float(min_score), merges=np.ascontiguousarray(merges), unk_id=unk_id,
                   fuse_unk=fuse_unk, byte_fallback=bool(model.get("byte_fallback", False)), byte_fallback_ids=bf_ids,
                   ignore_merges=bool(model.get("ignore_merges", False)), special_bytes=sb, special_offsets=so,
                   special_ids=np.asarray([i for _, i in sp], dtype=np.int32), pad_token_id=int(pad_token_id),
                   special_tokens=tuple(special_tokens),
                   host_specials=tuple((s, int(i)) for s, i in zip(special_tokens, special_ids) if not _raw(s)),
                   piece_continuing=None if continuing is None else np.asarray(continuing, dtype=np.uint8),
                   max_input_chars_per_word=int(model.get("max_input_chars_per_word", 100)))

    @classmethod
    def from_tokenizer(cls, tokenizer) -> "HnTokenizerSpec":
        """From the object the reference passes as ``tokenizer_to_use`` (a transformers fast tokenizer)."""
        if type(tokenizer).__name__ == "ByT5Tokenizer":
            # zett/utils.py:677-678: ids = convert_tokens_to_ids([chr(b) for b in token_bytes]) — one id per BYTE (ord + the
            # tokenizer's offset), no merges.  On the device that is a BPE model of 256 single-byte pieces without merges.
            specials = list(tokenizer.all_special_tokens)
            byte_ids = tokenizer.convert_tokens_to_ids([chr(b) for b in range(256)])
            model = {"type": "BPE", "vocab": {BYTES_TO_CHARS_LIST[b]: int(i) for b, i in enumerate(byte_ids)}, "merges": []}
            return cls.from_model_json(model, specials, [tokenizer.convert_tokens_to_ids(s) for s in specials], tokenizer.pad_token_id)
        data = json.loads(tokenizer._tokenizer.to_str())
        specials = list(tokenizer.all_special_tokens)
        ids = [tokenizer.convert_tokens_to_ids(s) for s in specials]
        return cls.from_model_json(data["model"], specials, ids, tokenizer.pad_token_id)


class DeviceRetokenizer:
    """A ``zett_retok`` handle: the hn tokenizer's tables resident on one GPU."""

    def __init__(self, spec: HnTokenizerSpec, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("zett_amd computes on MI355X only: pass a cuda (ROCm) device; there is no CPU path")
        self.lib = _lib.load()
        self.spec = spec
        self.device = device

        def ptr(a):
            return None if a is None else a.ctypes.data_as(C.c_void_p)

        m = _lib.ZettRetokModel(
            kind=spec.kind, n_pieces=len(spec.piece_ids), piece_bytes=ptr(spec.piece_bytes),
            piece_offsets=ptr(spec.piece_offsets), piece_ids=ptr(spec.piece_ids), piece_scores=ptr(spec.piece_scores),
            unigram_min_score=spec.unigram_min_score, n_merges=len(spec.merges), merges=ptr(spec.merges),
            unk_id=spec.unk_id, fuse_unk=int(spec.fuse_unk), byte_fallback=int(spec.byte_fallback),
            byte_fallback_ids=ptr(spec.byte_fallback_ids), ignore_merges=int(spec.ignore_merges),
            n_special=len(spec.special_ids), special_bytes=ptr(spec.special_bytes),
            special_offsets=ptr(spec.special_offsets), special_ids=ptr(spec.special_ids),
            piece_continuing=ptr(spec.piece_continuing), max_input_chars_per_word=int(spec.max_input_chars_per_word))
        handle = C.c_void_p()
        index = device.index if device.index is not None else torch.cuda.current_device()
        _lib.check(self.lib.zett_retok_create(C.byref(m), index, C.byref(handle)), "zett_retok_create")
        self.handle = handle
        self._outstanding = []          # (text, offsets) tensors of the asynchronous calls since the last result()
        self._staging = {}              # element width -> pinned staging buffer + the event of its last transfer (_to_device)

    def set_option(self, key: str, value: int) -> None:
        """zett_retok_set_option: A/B switches of the handle ("unigram_workgroup": which Unigram kernel runs — 1 = by size, the
        default; 2 = the workgroup-per-64-tokens kernel; 0 = the lane-per-token kernel)."""
        _lib.check(self.lib.zett_retok_set_option(self.handle, key.encode(), int(value)), "zett_retok_set_option")

    @staticmethod
    def flatten_tokens(tokens: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
        """UTF-8 text of all tokens back to back (uint8) + int32 offsets [n + 1] — ONE join / encode for the whole list and
        numpy for the offsets instead of a Python-level ``.encode`` per token (50 k tokens: ~1 ms instead of ~9).  NUL is the
        separator: no byte-level token holds it (byte 0 is written U+0100, zett/utils.py:351-609); a list that does falls
        back to the per-token walk, as does a list with non-str entries (which then fails as the reference's loop would)."""
        n = len(tokens)
        if n == 0:
            return np.zeros(1, dtype=np.uint8), np.zeros(1, dtype=np.int32)
        try:
            blob = np.frombuffer("\0".join(tokens).encode("utf-8"), dtype=np.uint8)
            seps = np.flatnonzero(blob == 0)
        except TypeError:
            seps = None
        if seps is None or len(seps) != n - 1:
            encoded = [t.encode("utf-8") for t in tokens]
            offsets = np.zeros(n + 1, dtype=np.int32)
            np.cumsum(np.fromiter(map(len, encoded), dtype=np.int64, count=n), out=offsets[1:])
            return np.frombuffer(b"".join(encoded) or b"\0", dtype=np.uint8), offsets
        if len(blob) - (n - 1) >= 2 ** 31 - 1:
            raise ValueError("more than 2 GiB of token text in one call")
        offsets = np.empty(n + 1, dtype=np.int32)
        offsets[0] = 0
        offsets[1:n] = seps - np.arange(n - 1)
        offsets[n] = len(blob) - (n - 1)
        text = blob[blob != 0] if n > 1 else blob
        return (text if len(text) else np.zeros(1, dtype=np.uint8)), offsets

    def _to_device(self, host: np.ndarray) -> torch.Tensor:
        """Host array -> device through a pinned staging buffer (grow-only, one per element width): the copy is a DMA from
        page-locked memory on the current stream instead of a blocking pageable copy."""
        nbytes = host.nbytes
        slot = self._staging.setdefault(host.dtype.itemsize, {"pin": None, "event": None})
        if slot["pin"] is None or slot["pin"].numel() < nbytes:
            slot["pin"] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, pin_memory=True)
            slot["event"] = None
        if slot["event"] is not None:
            slot["event"].synchronize()          # the previous transfer out of this buffer has left
        slot["pin"][:nbytes].numpy()[:] = np.ascontiguousarray(host).view(np.uint8).reshape(-1)
        dev = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        dev.copy_(slot["pin"][:nbytes], non_blocking=True)
        slot["event"] = torch.cuda.Event()
        slot["event"].record(torch.cuda.current_stream(self.device))
        return dev.view(getattr(torch, str(host.dtype))).reshape(host.shape)

    def encode_joined(self, tokens: Sequence[str]):
        """Host side of __call__: ONE ``"\\0".join(tokens).encode()`` — the NUL-separated text zett_retokenize_async takes with
        offsets == NULL (ABI 6): the token boundaries are found on the GPU by the scan that numbers the characters, the host
        does no per-token work (50 k tokens: ~1.2 ms for the join itself, against ~2.8 ms with the offsets made in numpy and
        ~9 ms with a Python-level ``.encode`` per token).  Returns (d_text, None, n); falls back to encode() — text + offsets —
        when a token holds a NUL (the reference raises KeyError for it, which the offsets path reports) or is not a str."""
        n = len(tokens)
        if n == 0:
            return self.encode(tokens)
        try:
            blob = "\0".join(tokens).encode("utf-8")
        except TypeError:
            return self.encode(tokens)
        if blob.count(b"\0") != n - 1 or len(blob) >= 2 ** 31 - 1:
            return self.encode(tokens)
        with torch.cuda.device(self.device):
            d_text = self._to_device(np.frombuffer(blob or b"\0", dtype=np.uint8))
        d_text._zett_n_text = len(blob)
        return d_text, None, n

    def encode(self, tokens: Sequence[str]) -> Tuple[torch.Tensor, torch.Tensor, int]:
        """Host side of a call: UTF-8 text of the byte-level token strings + int32 offsets, copied to the device."""
        text, offsets = self.flatten_tokens(tokens)
        n = len(tokens)
        with torch.cuda.device(self.device):
            d_text = self._to_device(text)
            d_off = self._to_device(offsets)
        d_text._zett_n_text = int(offsets[-1])          # the text length, known here: run_async() needs it without a device round trip
        return d_text, d_off, n

    def run(self, d_text: torch.Tensor, d_off: torch.Tensor, n: int, maxlen: int, tokens: Optional[Sequence[str]] = None) -> Tuple[torch.Tensor, int]:
        """Device side: zett_retokenize on resident text/offsets -> int32 [n, maxlen] on the device + n_truncated."""
        out = torch.empty((n, maxlen), dtype=torch.int32, device=self.device)
        n_trunc = C.c_int64(0)
        bad = C.c_int64(-1)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self.lib.zett_retokenize(self.handle, C.c_void_p(d_text.data_ptr()), C.c_void_p(d_off.data_ptr()), n,
                                          int(maxlen), self.spec.pad_token_id, C.c_void_p(out.data_ptr()),
                                          C.byref(n_trunc), C.byref(bad), C.c_void_p(stream))
        if rc == _lib.E_KEY and bad.value >= 0 and tokens is not None:
            for ch in tokens[bad.value]:                 # the reference raises KeyError(<character>)
                if ch not in CHARS_TO_BYTES:
                    raise KeyError(ch)
        if rc == _lib.E_STATE:
            raise Exception(self.lib.zett_last_error().decode())      # tokenizers raises a bare Exception here
        _lib.check(rc, "zett_retokenize")
        return out, int(n_trunc.value)

    def run_async(self, d_text: torch.Tensor, d_off: torch.Tensor, n: int, maxlen: int) -> torch.Tensor:
        """zett_retokenize_async: as run(), without waiting — the id matrix is enqueued on the current stream, the count of
        truncated tokens and any error (KeyError, missing unk id) are collected by result(), once for all calls since the
        last result()."""
        out = torch.empty((n, maxlen), dtype=torch.int32, device=self.device)
        n_text = getattr(d_text, "_zett_n_text", None)
        if n_text is None:
            if d_off is None:
                raise ValueError("NUL-separated text (d_off = None) must come from encode_joined()")
            n_text = int(d_off[-1].item()) if n else 0          # (text not made by encode(): one device round trip)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self.lib.zett_retokenize_async(self.handle, C.c_void_p(d_text.data_ptr()), C.c_void_p(d_off.data_ptr() if d_off is not None else 0), n,
                                                n_text, int(maxlen), self.spec.pad_token_id,
                                                C.c_void_p(out.data_ptr()), C.c_void_p(stream))
        _lib.check(rc, "zett_retokenize_async")
        self._outstanding.append((d_text, d_off))       # zett_retok_result reads `offsets` back on a KeyError: alive until result()
        return out

    def result(self) -> int:
        """Waits for the asynchronous calls since the last result(): their number of truncated tokens; raises what run() raises."""
        n_trunc, bad_call, bad = C.c_int64(0), C.c_int64(-1), C.c_int64(-1)
        rc = self.lib.zett_retok_result(self.handle, C.byref(n_trunc), C.byref(bad_call), C.byref(bad))
        self._outstanding.clear()
        if rc == _lib.E_KEY:
            err = KeyError(f"call {bad_call.value}, token {bad.value}: {self.lib.zett_last_error().decode()}")
            err.zett_bad_call, err.zett_bad_token = int(bad_call.value), int(bad.value)
            raise err
        if rc == _lib.E_STATE:
            raise Exception(self.lib.zett_last_error().decode())
        _lib.check(rc, "zett_retok_result")
        return int(n_trunc.value)

    def __call__(self, tokens: Sequence[str], maxlen: int) -> Tuple[torch.Tensor, int]:
        """int32 [len(tokens), maxlen] on the device + number of truncated tokens."""
        patches = []
        if self.spec.host_specials:                # exact string match first (zett/utils.py:671-673): such a token never reaches the byte table
            tokens = list(tokens)
            for special, sid in self.spec.host_specials:     # (list.index scans at C speed: no Python-level pass over 50 k tokens)
                start = 0
                while True:
                    try:
                        i = tokens.index(special, start)
                    except ValueError:
                        break
                    patches.append((i, sid))
                    tokens[i] = ""                 # an empty token retokenizes to an all-pad row; column 0 is set below
                    start = i + 1
        d_text, d_off, n = self.encode_joined(tokens)
        if n == 0:
            return torch.empty((0, maxlen), dtype=torch.int32, device=self.device), 0
        # one host round trip (result()) instead of zett_retokenize's two: the text length is known here
        if self._outstanding:
            self.result()                          # (results of earlier asynchronous calls nobody asked for: dropped, as zett_retokenize does)
        out = self.run_async(d_text, d_off, n, maxlen)
        try:
            n_trunc = self.result()
        except KeyError as err:                    # the reference raises KeyError(<character>) (zett/utils.py:675)
            bad = getattr(err, "zett_bad_token", -1)
            if 0 <= bad < len(tokens):
                for ch in tokens[bad]:
                    if ch not in CHARS_TO_BYTES:
                        raise KeyError(ch) from None
            raise
        if patches:
            rows = torch.tensor([i for i, _ in patches], dtype=torch.long, device=out.device)
            out[rows, 0] = torch.tensor([sid for _, sid in patches], dtype=out.dtype, device=out.device)
        return out, n_trunc

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.zett_retok_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_CACHE: Dict[Tuple[int, str], DeviceRetokenizer] = {}


def device_retokenizer(tokenizer_to_use, device=None) -> DeviceRetokenizer:
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("zett_amd computes on MI355X only: no cuda (ROCm) device is visible; there is no CPU path")
        device = torch.device("cuda", torch.cuda.current_device())
    else:
        device = torch.device(device)
    if isinstance(tokenizer_to_use, HnTokenizerSpec):
        spec = tokenizer_to_use
    else:
        spec = getattr(tokenizer_to_use, "_zett_spec", None)
        if spec is None:
            spec = HnTokenizerSpec.from_tokenizer(tokenizer_to_use)
            try:
                tokenizer_to_use._zett_spec = spec
            except Exception:
                pass
    key = (id(spec), str(device))
    rt = _CACHE.get(key)
    if rt is None:
        rt = DeviceRetokenizer(spec, device)
        _CACHE[key] = rt
    return rt


def surface_form_matrix_device(tokens: Sequence[str], maxlen: int, tokenizer_to_use, device=None) -> Tuple[torch.Tensor, int]:
    """Like get_surface_form_matrix but leaves the int32 matrix on the GPU (feeds ZettHypernet directly)."""
    return device_retokenizer(tokenizer_to_use, device)(tokens, maxlen)


def get_surface_form_matrix(tokenizer_or_tokens, maxlen, tokenizer_to_use=None, padding=0, verbose=False, device=None):
    """Drop-in for zett.utils.get_surface_form_matrix (zett/utils.py:651-689).

    Returns ``(np.int32 [V + padding, maxlen], n_truncated)``; raises ``KeyError`` on a token
    holding a character outside the byte-level alphabet (zett/utils.py:675).
    """
    if isinstance(tokenizer_or_tokens, list):
        tokens = tokenizer_or_tokens
    else:
        tokens = tokenizer_or_tokens.convert_ids_to_tokens(range(len(tokenizer_or_tokens)))   # :659
    if tokenizer_to_use is None:
        raise ValueError("tokenizer_to_use (the hn tokenizer) is required")
    rt = device_retokenizer(tokenizer_to_use, device)
    matrix, n_truncated = rt(tokens, maxlen)
    if not padding:
        return matrix.cpu().numpy(), n_truncated
    out = np.full((len(tokens) + padding, maxlen), rt.spec.pad_token_id, dtype=np.int32)      # :662-666
    out[:len(tokens)] = matrix.cpu().numpy()
    return out, n_truncated
