"""GPU parity of the retokenizer (get_surface_form_matrix through the C ABI): bit-exact
against the reference's outputs (golden fixtures) and against the oracle on random models."""
import json
import os
import random

import numpy as np
import pytest

from oracle import retok_ref
from tests import retok_random as rr
from tests import util

pytestmark = pytest.mark.gpu

RETOK = sorted(p for p in os.listdir(util.GOLDEN) if p.startswith("retok_") and p.endswith(".json"))


def _spec(g):
    from zett_amd.surface_forms import HnTokenizerSpec
    return HnTokenizerSpec.from_model_json(g["model"], g["special_tokens"], g["special_ids"], g["pad_token_id"])


@pytest.mark.parametrize("name", RETOK)
def test_golden_matrix(name):
    from zett_amd.surface_forms import get_surface_form_matrix
    g = json.load(open(os.path.join(util.GOLDEN, name)))
    got, n_tr = get_surface_form_matrix(g["tokens"], g["maxlen"], _spec(g))
    assert got.dtype == np.int32
    np.testing.assert_array_equal(got, np.array(g["expected"], dtype=np.int32))
    assert n_tr == g["n_truncated"]


def test_padding_rows_and_tokenizer_object():
    """`padding` extra rows (zett/utils.py:662-666) and the transformers-tokenizer entry point."""
    from tokenizers import Tokenizer
    from transformers import PreTrainedTokenizerFast

    from zett_amd.surface_forms import get_surface_form_matrix
    g = json.load(open(os.path.join(util.GOLDEN, "retok_unigram.json")))
    tok = Tokenizer.from_str(json.dumps({"version": "1.0", "truncation": None, "padding": None, "added_tokens": [],
                                         "normalizer": None, "pre_tokenizer": None, "post_processor": None,
                                         "decoder": None, "model": g["model"]}))
    hf = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<pad>")
    assert hf.pad_token_id == g["pad_token_id"]
    got, n_tr = get_surface_form_matrix(g["tokens"], g["maxlen"], hf, padding=5)
    want = np.array(g["expected"], dtype=np.int32)
    assert got.shape == (len(want) + 5, g["maxlen"])
    np.testing.assert_array_equal(got[:len(want)], want)
    assert (got[len(want):] == g["pad_token_id"]).all()
    assert n_tr == g["n_truncated"]


def test_keyerror_names_the_character():
    from zett_amd.surface_forms import get_surface_form_matrix
    g = json.load(open(os.path.join(util.GOLDEN, "retok_bytebpe.json")))
    with pytest.raises(KeyError) as e:
        get_surface_form_matrix(["fine", "Ġalso", "not▁byte"], 7, _spec(g))
    assert e.value.args[0] == "▁"
    with pytest.raises(KeyError):
        get_surface_form_matrix(["a b"], 7, _spec(g))      # a raw space is not in the alphabet


def test_empty_inputs():
    from zett_amd.surface_forms import get_surface_form_matrix
    g = json.load(open(os.path.join(util.GOLDEN, "retok_bytebpe.json")))
    got, n_tr = get_surface_form_matrix([], 7, _spec(g))
    assert got.shape == (0, 7) and n_tr == 0
    got, n_tr = get_surface_form_matrix(["", "", ""], 7, _spec(g))
    assert (got == g["pad_token_id"]).all() and n_tr == 0


@pytest.mark.parametrize("seed", range(30))
def test_random_models_match_oracle(seed):
    from zett_amd.surface_forms import HnTokenizerSpec, get_surface_form_matrix
    rng = random.Random(1000 + seed)
    for make in (rr.random_bpe, rr.random_unigram):
        model_json = make(rng, 80)
        specials, special_ids = (["<unk>"], [model_json["vocab"]["<unk>"]]) if isinstance(model_json["vocab"], dict) and "<unk>" in model_json["vocab"] else ([], [])
        tokens = rr.random_tokens(rng, 300, maxlen=40) + rr.random_tokens(rng, 20, maxlen=300) + specials
        oracle_model = retok_ref.model_from_tokenizer_json(model_json, specials, special_ids)
        want, want_tr = retok_ref.surface_form_matrix_c(oracle_model, tokens, 9, 77)
        spec = HnTokenizerSpec.from_model_json(model_json, specials, special_ids, 77)
        got, got_tr = get_surface_form_matrix(tokens, 9, spec)
        np.testing.assert_array_equal(got, want)
        assert got_tr == want_tr


@pytest.mark.parametrize("seed", range(20))
def test_random_wordpiece_models_match_oracle(seed):
    """WordPiece hn tokenizers (zett/utils.py:681 is model-agnostic, zett/tokenizer_converters.py:370-373): random models with
    the prefixes "", "##", "a", "ab", small max_input_chars_per_word, [UNK] present or not (then only words that do not
    need it), against the C oracle (itself equal to the tokenizers wheel: tests/test_retok_oracle.py)."""
    from zett_amd.surface_forms import HnTokenizerSpec, get_surface_form_matrix
    rng = random.Random(3000 + seed)
    model_json = rr.random_wordpiece(rng, 90)
    has_unk = "[UNK]" in model_json["vocab"]
    specials, special_ids = (["[UNK]"], [model_json["vocab"]["[UNK]"]]) if has_unk else ([], [])
    tokens = rr.random_tokens(rng, 400, maxlen=14) + rr.random_tokens(rng, 20, maxlen=120) + specials
    oracle_model = retok_ref.model_from_tokenizer_json(model_json, specials, special_ids)
    if not has_unk:        # keep the words the library can tokenize; one that needs [UNK] must fail the call
        bad = []
        for t in tokens:
            try:
                retok_ref.tokenize(oracle_model, retok_ref.token_to_bytes(t))
            except RuntimeError:
                bad.append(t)
        tokens = [t for t in tokens if t not in set(bad)]
        spec = HnTokenizerSpec.from_model_json(model_json, specials, special_ids, 77)
        if bad:
            with pytest.raises(Exception, match="UNK"):
                get_surface_form_matrix(tokens + bad[:1], 9, spec)
    want, want_tr = retok_ref.surface_form_matrix_c(oracle_model, tokens, 9, 77)
    spec = HnTokenizerSpec.from_model_json(model_json, specials, special_ids, 77)
    got, got_tr = get_surface_form_matrix(tokens, 9, spec)
    np.testing.assert_array_equal(got, want)
    assert got_tr == want_tr


def test_wordpiece_tokenizer_object_entry_point():
    """The transformers-tokenizer entry point with a WordPiece model, "##" prefix kept (not converted): the library's own
    Model.tokenize is the expectation, token by token."""
    from tokenizers import Tokenizer
    from transformers import PreTrainedTokenizerFast

    from zett_amd.surface_forms import get_surface_form_matrix
    vocab = {"[PAD]": 0, "[UNK]": 1, "a": 2, "b": 3, "ab": 4, "##a": 5, "##b": 6, "##ab": 7, "##": 8, "#": 9, "abab": 10, "c": 11}
    tok = Tokenizer.from_str(json.dumps({"version": "1.0", "truncation": None, "padding": None, "added_tokens": [], "normalizer": None,
                                         "pre_tokenizer": None, "post_processor": None, "decoder": None,
                                         "model": {"type": "WordPiece", "unk_token": "[UNK]", "continuing_subword_prefix": "##",
                                                   "max_input_chars_per_word": 8, "vocab": vocab}}))
    hf = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="[UNK]", pad_token="[PAD]")
    tokens = ["ab", "aab", "abab", "ababab", "ba", "##", "##a", "a#", "cab", "abc", "ababababa", "abababab", "", "[UNK]", "[PAD]", "bbbb"]
    want = np.zeros((len(tokens), 5), dtype=np.int32)
    for r, t in enumerate(tokens):
        if t in ("[UNK]", "[PAD]"):
            want[r, 0] = vocab[t]
            continue
        ids = [x.id for x in tok.model.tokenize(t)][:5]
        want[r, :len(ids)] = ids
    got, _ = get_surface_form_matrix(tokens, 5, hf)
    np.testing.assert_array_equal(got, want)


def test_unigram_without_unk_raises():
    from zett_amd.surface_forms import HnTokenizerSpec, get_surface_form_matrix
    model = {"type": "Unigram", "unk_id": None, "byte_fallback": False, "vocab": [["a", -1.0], ["b", -1.0]]}
    spec = HnTokenizerSpec.from_model_json(model, [], [], 0)
    got, _ = get_surface_form_matrix(["ab", "ba"], 4, spec)
    np.testing.assert_array_equal(got, [[0, 1, 0, 0], [1, 0, 0, 0]])
    with pytest.raises(Exception, match="unk_id"):
        get_surface_form_matrix(["ab", "azb"], 4, spec)


def test_unigram_workgroup_kernel_equals_lane_kernel_and_oracle():
    """Unigram models run a workgroup per 64 tokens (r6: every (start, end) piece lookup of the 64 tokens in flight at once, then
    the Viterbi walk on LDS; zett_retok_set_option "unigram_workgroup"): same ids as the lane-per-token kernel and as the C oracle —
    on pieces of up to 24 bytes (tables of len x 24 slots), tokens long enough that 64 of them need several rounds of the 3 072-slot
    table, one that exceeds it alone, a workgroup whose text does not fit its LDS stage, ties between equal scores, byte
    fallback; and on the 50 k tokens / 250 k pieces of the XLM-R workload bench.py measures."""
    import torch

    from zett_amd import synth
    from zett_amd.surface_forms import DeviceRetokenizer, HnTokenizerSpec
    rng = random.Random(77)
    alphabet = "abcdeĠ"
    for byte_fallback in (False, True):
        vocab = [["<unk>", 0.0]] + [[c, rng.choice([-3.0, -2.5, -3.5])] for c in "abcdĠ"]          # 'e' has no piece: unknown runs
        seen = set()
        for _ in range(3000):
            piece = "".join(rng.choice(alphabet[:5] + "Ġ") for _ in range(rng.choice([2, 2, 3, 3, 4, 5, 6, 8, 12, 16, 24])))
            if piece not in seen:
                seen.add(piece)
                vocab.append([piece, rng.choice([-4.0, -5.0, -6.0, -7.5, -9.0, -12.0])])
        if byte_fallback:
            vocab += [[f"<0x{b:02X}>", -20.0] for b in range(256)]
        model = {"type": "Unigram", "unk_id": 0, "byte_fallback": byte_fallback, "vocab": vocab}
        tokens = (["".join(rng.choice(alphabet) for _ in range(rng.randint(1, 30))) for _ in range(3000)]
                  + ["".join(rng.choice(alphabet) for _ in range(rng.randint(40, 90))) for _ in range(200)]          # several rounds per workgroup
                  + ["".join(rng.choice(alphabet) for _ in range(200))]                                              # > 3 072 slots alone
                  + ["".join(rng.choice(alphabet) for _ in range(100)) for _ in range(64)]                           # > 4 KiB of text in one workgroup
                  + ["", "a", "<unk>", "e", "eee", "aeea"])
        spec = HnTokenizerSpec.from_model_json(model, ["<unk>"], [0], 1)
        rt = DeviceRetokenizer(spec, torch.device("cuda", 0))
        rt.set_option("unigram_workgroup", 2)
        got_wg, tr_wg = rt(tokens, 11)
        rt.set_option("unigram_workgroup", 0)
        got_lane, tr_lane = rt(tokens, 11)
        assert torch.equal(got_wg, got_lane) and tr_wg == tr_lane
        oracle_model = retok_ref.model_from_tokenizer_json(model, ["<unk>"], [0])
        want, want_tr = retok_ref.surface_form_matrix_c(oracle_model, tokens, 11, 1)
        np.testing.assert_array_equal(got_wg.cpu().numpy(), want)
        assert tr_wg == want_tr
        with pytest.raises(Exception, match="option"):
            rt.set_option("no_such_option", 1)
        rt.close()
    cfg, rows, _, hist = synth.workload("xlmr_gpt2")
    ids = synth.make_surface_forms(cfg, rows, seed=0, hist=hist)
    hn_model, piece_of_id = synth.make_hn_model("xlmr_gpt2", cfg)
    spec = HnTokenizerSpec.from_model_json(hn_model, ["<unk>", "<s>", "</s>"], [0, 1, 2], cfg["pad_token_id"])
    rt = DeviceRetokenizer(spec, torch.device("cuda", 0))
    d_text, d_off, n = rt.encode(synth.tokens_for_surface_forms(cfg, ids, piece_of_id))
    rt.set_option("unigram_workgroup", 2)
    a, tr_a = rt.run(d_text, d_off, n, 7)
    rt.set_option("unigram_workgroup", 0)
    b, tr_b = rt.run(d_text, d_off, n, 7)
    assert torch.equal(a, b) and tr_a == tr_b == 0 and torch.equal(a.cpu(), torch.from_numpy(ids))
    rt.close()


def test_full_size_vocab_matches_oracle():
    """50k-token target vocabulary against the C oracle (size of the GPT-2 / GPT-NeoX configs)."""
    from zett_amd.surface_forms import get_surface_form_matrix
    g = json.load(open(os.path.join(util.GOLDEN, "retok_mistral_like.json")))
    rng = random.Random(5)
    base = [t for t in g["tokens"] if t not in g["special_tokens"] and t]
    tokens = list(g["tokens"])
    while len(tokens) < 50000:
        a, b = rng.choice(base), rng.choice(base)
        tokens.append((a + b)[:rng.randint(1, 24)])
    oracle_model = retok_ref.model_from_tokenizer_json({"model": g["model"]}, g["special_tokens"], g["special_ids"])
    want, want_tr = retok_ref.surface_form_matrix_c(oracle_model, tokens, 7, g["pad_token_id"])
    got, got_tr = get_surface_form_matrix(tokens, 7, _spec(g))
    np.testing.assert_array_equal(got, want)
    assert got_tr == want_tr


def test_special_tokens_outside_the_byte_alphabet_match_by_string():
    """zett/utils.py:671-673 tests `token in all_special_tokens` on the character string BEFORE the byte lookup: a special
    token holding characters outside the byte-level alphabet ('▁', a space) is matched, not a KeyError — and an
    ordinary token with such a character still is one."""
    import torch

    from zett_amd.surface_forms import DeviceRetokenizer, HnTokenizerSpec
    model = {"type": "Unigram", "unk_id": 0, "byte_fallback": False,
             "vocab": [["<unk>", 0.0], ["<|begin▁of▁sentence|>", 0.0], ["<pad token>", 0.0], ["<s>", 0.0]] + [[c, -1.0] for c in "abcdef"] + [["ab", -0.5]]}
    specials = ["<|begin▁of▁sentence|>", "<pad token>", "<s>"]
    spec = HnTokenizerSpec.from_model_json(model, specials, [1, 2, 3], pad_token_id=2)
    assert dict(spec.host_specials) == {"<|begin▁of▁sentence|>": 1, "<pad token>": 2}
    rt = DeviceRetokenizer(spec, torch.device("cuda:0"))
    out, n_trunc = rt(["abc", "<|begin▁of▁sentence|>", "<s>", "<pad token>", "fe"], 4)
    want = [[10, 6, 2, 2], [1, 2, 2, 2], [3, 2, 2, 2], [2, 2, 2, 2], [9, 8, 2, 2]]
    assert out.cpu().tolist() == want and n_trunc == 0
    with pytest.raises(KeyError):
        rt(["a▁b"], 4)


def test_byt5_hn_tokenizer_branch():
    """zett/utils.py:677-678: with a ByT5 hn tokenizer every BYTE is one id (ord + the tokenizer's offset), no merges; special
    tokens are matched by string first.  The fixture is the reference's own output with transformers' ByT5Tokenizer()."""
    from transformers import ByT5Tokenizer

    from zett_amd.surface_forms import get_surface_form_matrix
    g = json.load(open(os.path.join(util.GOLDEN, "byt5_case.json")))
    hn = ByT5Tokenizer()
    for maxlen, case in g["cases"].items():
        got, n_tr = get_surface_form_matrix(g["tokens"], int(maxlen), hn)
        np.testing.assert_array_equal(got, np.array(case["expected"], dtype=np.int32))
        assert n_tr == case["n_truncated"]


def test_separator_mode_equals_offsets_mode():
    """ABI 6: zett_retokenize_async with offsets == NULL takes the tokens NUL-separated and finds their boundaries on the
    device (what __call__ / get_surface_form_matrix now send); the offsets mode (encode() + run(): what bench.py and
    zett_retokenize use) must give the same matrix — empty tokens at the head, in the middle and at the end, two-byte
    characters straddling the 16-byte spans of the byte-table kernel's threads and its 4 096-byte blocks, one token, and a
    list long enough for several blocks."""
    import torch
    from zett_amd.surface_forms import DeviceRetokenizer
    g = json.load(open(os.path.join(util.GOLDEN, "retok_bytebpe.json")))
    rt = DeviceRetokenizer(_spec(g), torch.device("cuda", 0))
    rng = random.Random(7)
    alphabet = "abcdefghijklmnopqrstuvwxyzĠĊčĉ0123456789"
    cases = [["a"], [""], ["", "a"], ["a", ""], ["", "", "Ġ", ""], ["Ġ"] * 5000,
             ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 23))) for _ in range(20000)]]
    for tokens in cases:
        d_text, d_off, n = rt.encode_joined(tokens)
        assert d_off is None
        sep = rt.run_async(d_text, d_off, n, 9)
        assert isinstance(rt.result(), int)
        off, n_tr = rt.run(*rt.encode(tokens), 9)
        assert torch.equal(sep, off), len(tokens)
        called, n_tr2 = rt(tokens, 9)
        assert torch.equal(called, off) and n_tr2 == n_tr
    # a bad character is reported with its token, in either mode (the reference raises KeyError(<character>))
    tokens = ["fine"] * 4500 + ["not▁byte"] + ["x"] * 10
    with pytest.raises(KeyError) as e:
        rt(tokens, 9)
    assert e.value.args[0] == "▁"
    # a token holding a NUL cannot be sent separator-joined: the offsets path takes it and reports the NUL itself
    with pytest.raises(KeyError) as e:
        rt(["a", "b\0c"], 9)
    assert e.value.args[0] == "\0"
    # the C ABI refuses a NUL-separated text with the wrong number of separators
    from zett_amd import _lib
    d_text, _, _ = rt.encode_joined(["a", "b", "c"])
    rt.run_async(d_text, None, 5, 9)
    with pytest.raises(Exception) as e:
        rt.result()
    assert "separators" in str(e.value)
    rt.close()
