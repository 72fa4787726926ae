"""``convert_to_byte_level`` — rewrite a fast tokenizer into a byte-level one.

The step BEFORE the embedding-prediction path (reference zett/tokenizer_converters.py:78-406;
SURVEY.md §8f row N2, marked "next"): host-only tokenizer-JSON surgery that produces the
byte-level token list and the hn tokenizer which ``get_surface_form_matrix`` consumes.
It is not part of the accelerated path and is not restated yet; tokenizers that are already
byte-level pass through unchanged when no surgery is requested.
"""
from __future__ import annotations


def _is_byte_level(tokenizer) -> bool:
    from tokenizers import pre_tokenizers
    import json
    pre = tokenizer._tokenizer.pre_tokenizer
    if isinstance(pre, pre_tokenizers.ByteLevel):
        return True
    data = json.loads(tokenizer._tokenizer.to_str()).get("pre_tokenizer") or {}
    return data.get("type") == "Sequence" and any(p.get("type") == "ByteLevel" for p in data.get("pretokenizers", []))


def convert_to_byte_level(tokenizer, keep_normalizer=False, keep_pretokenizer=False,
                          make_whitespace_consistent=False, match_special_tokens_to=None):
    """Same signature and return value ``(tokenizer, n_added_or_None)`` as the reference.

    Implemented so far: the identity case (an already byte-level tokenizer, no whitespace or
    special-token surgery requested).  Everything else raises NotImplementedError.
    """
    if _is_byte_level(tokenizer) and not make_whitespace_consistent and match_special_tokens_to is None:
        return tokenizer, 0
    raise NotImplementedError(
        "convert_to_byte_level surgery (byte-fallback remap, fill bytes, whitespace tokens, special-token "
        "matching, Unigram score remap, BPE merge surgery) is the 'next' row N2 of SURVEY.md §8f; pass "
        "tokenizers that are already byte-level, or token lists, to get_surface_form_matrix")
