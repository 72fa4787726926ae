"""Fixture for the ByT5 hn-tokenizer branch of the reference's get_surface_form_matrix (zett/utils.py:677-678).

Runs only in the build container (imports /root/reference with jax / flax / optax stubbed, as make_golden_retok.py does).
The fixture holds the token list, maxlen and the matrix the REFERENCE computed with transformers' ByT5Tokenizer() (which
needs no vocabulary file); nothing of the reference is copied."""
import json
import os

from make_golden_retok import HERE, _import_reference

if __name__ == "__main__":
    from transformers import ByT5Tokenizer
    _, gsfm, c2b, b2c = _import_reference()
    hn = ByT5Tokenizer()
    b2c_list = [b2c[i] for i in range(256)]
    words = ["hello", " world", "Größe", "日本語", "x" * 30, "", "\n\n", "a", "\x00\xff"]
    tokens = ["".join(b2c_list[b] for b in w.encode("latin-1" if w == "\x00\xff" else "utf-8")) for w in words]
    tokens += ["</s>", "<pad>", "<unk>", "<extra_id_7>"]                      # special tokens: matched by string first (:671-673)
    tokens += ["".join(b2c_list[(7 * i + j) % 256] for j in range(1 + i % 11)) for i in range(200)]
    out = {}
    for maxlen in (7, 16):
        matrix, n_truncated = gsfm(tokens, maxlen=maxlen, tokenizer_to_use=hn)
        out[str(maxlen)] = {"expected": matrix.tolist(), "n_truncated": int(n_truncated)}
    path = os.path.join(HERE, "byt5_case.json")
    json.dump({"tokens": tokens, "pad_token_id": hn.pad_token_id, "cases": out}, open(path, "w"), ensure_ascii=False, separators=(",", ":"))
    print("wrote", path, len(tokens), "tokens", {k: v["n_truncated"] for k, v in out.items()})
