"""The oracle pinned against outputs of the reference (tests/golden/*, made by
tests/golden/make_golden.py in the build container).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import hypernet_ref, retok_ref
from tests import util
from zett_amd import synth

SMALL = util.golden_cases("fwd_tiny_*.npz") + util.golden_cases("fwd_real_xlmr*.npz")
RETOK = sorted(p for p in os.listdir(util.GOLDEN) if p.startswith("retok_") and p.endswith(".json"))


@pytest.mark.parametrize("path", SMALL, ids=lambda p: os.path.basename(p)[:-4])
def test_forward_oracle_matches_reference(path):
    case = util.load_case(path)
    w = synth.make_weights(case["cfg"], case["seed"])
    src = synth.make_source_embeddings(case["cfg"], case["seed"], dtype=case["src_dtype"])
    out = hypernet_ref.forward(w, case["cfg"], case["ids"], src, case["lang"])
    # fp32 restatement vs fp32 reference: reassociation-level differences only
    np.testing.assert_allclose(out[0], case["pred_in"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out[2], case["bias"], rtol=0, atol=2e-5)
    if case["pred_out"] is None:
        assert out[1] is None
    else:
        np.testing.assert_allclose(out[1], case["pred_out"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("path", util.golden_cases("fwd_big_xlmr*.npz") + util.golden_cases("fwd_big_tinyllama*.npz"),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_forward_oracle_matches_reference_big_batch(path):
    """The 640-row fixtures (outputs of the reference for a 32-row sample): the oracle on the same rows."""
    case = util.load_case(path)
    w = synth.make_weights(case["cfg"], case["seed"])
    src = synth.make_source_embeddings(case["cfg"], case["seed"], dtype=case["src_dtype"])
    out = hypernet_ref.forward(w, case["cfg"], case["ids"][case["sample"]], src, case["lang"])
    np.testing.assert_allclose(out[0], case["pred_in"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out[2], case["bias"], rtol=0, atol=2e-5)
    if case["pred_out"] is not None:
        np.testing.assert_allclose(out[1], case["pred_out"], rtol=0, atol=2e-5)


def test_flops_per_row_matches_survey():
    for name, want in (("xlmr_gpt2", 0.2743e9), ("tinyllama_neox", 1.8467e9), ("mistral_neox", 7.3844e9)):
        cfg, *_ = synth.workload(name)
        assert abs(hypernet_ref.flops_per_row(cfg, 7) - want) / want < 1e-3
        from zett_amd.dims import HypernetDims, as_written_flops_per_row
        assert as_written_flops_per_row(HypernetDims.from_config(cfg), 7) == hypernet_ref.flops_per_row(cfg, 7)


def test_byte_table_matches_reference():
    tbl = json.load(open(os.path.join(util.GOLDEN, "byte_table.json")))["chars_to_bytes"]
    assert tbl == retok_ref.CHARS_TO_BYTES
    assert len(tbl) == 256 and sorted(tbl.values()) == list(range(256))


@pytest.mark.parametrize("name", RETOK)
def test_retok_oracle_matches_reference(name):
    g = json.load(open(os.path.join(util.GOLDEN, name)))
    model = retok_ref.model_from_tokenizer_json({"model": g["model"]}, g["special_tokens"], g["special_ids"])
    want = np.array(g["expected"], dtype=np.int32)
    got, n_tr = retok_ref.surface_form_matrix_c(model, g["tokens"], g["maxlen"], g["pad_token_id"])
    assert n_tr == g["n_truncated"]
    np.testing.assert_array_equal(got, want)
    got_py, _ = retok_ref.surface_form_matrix_py(model, g["tokens"][:300], g["maxlen"], g["pad_token_id"])
    np.testing.assert_array_equal(got_py, want[:300])


def test_retok_keyerror_on_non_byte_char():
    g = json.load(open(os.path.join(util.GOLDEN, "retok_bytebpe.json")))
    model = retok_ref.model_from_tokenizer_json({"model": g["model"]}, g["special_tokens"], g["special_ids"])
    with pytest.raises(KeyError):
        retok_ref.surface_form_matrix_c(model, ["ok", "not byte level: ▁"], 7, 0)


@pytest.mark.parametrize("path", util.golden_cases("fwd_tiny_*.npz") + util.golden_cases("fwd_real_xlmr*.npz"),
                         ids=lambda p: p.split("/")[-1][:-4])
def test_exact_levers_equal_as_written(path):
    """oracle.forward_levers (pad skipping, per-distinct-id input projection, CLS-only last layer —
    the algebra the HIP path executes) is the same function as the as-written forward:
    checked on every flag combination, L = 1/15/24, all-pad rows, fallback ids, language token."""
    from oracle import hypernet_ref
    case = util.load_case(path)
    w = synth.make_weights(case["cfg"], case["seed"])
    src = synth.make_source_embeddings(case["cfg"], case["seed"], dtype=case["src_dtype"])
    got = hypernet_ref.forward_levers(w, case["cfg"], case["ids"], src, case["lang"])
    for g, name in zip(got, ("pred_in", "pred_out", "bias")):
        want = case[name]
        if want is None:
            assert g is None
            continue
        if name == "bias" and not case["cfg"].get("hn_predict_bias"):
            assert (g == 0).all()
            continue
        np.testing.assert_allclose(g, want, rtol=0, atol=2e-5, err_msg=f"{case['name']} {name}")
