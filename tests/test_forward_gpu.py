"""GPU parity of the hypernet forward: HIP path (through the C ABI) vs the golden
fixtures taken from the reference, and vs the oracle on fresh seeded inputs."""
import numpy as np
import pytest

from tests import util
from zett_amd import synth

pytestmark = pytest.mark.gpu

TINY = util.golden_cases("fwd_tiny_*.npz")
REAL = util.golden_cases("fwd_real_*.npz")
BIG = util.golden_cases("fwd_big_*.npz")


def _check(case, out, precision):
    close = util.CLOSE[precision]
    close(out[0], case["pred_in"], f"{case['name']}[{precision}] pred_in")
    if case["pred_out"] is None:
        assert out[1] is None
    else:
        close(out[1], case["pred_out"], f"{case['name']}[{precision}] pred_out")
    if case["cfg"].get("hn_predict_bias"):
        close(out[2], case["bias"], f"{case['name']}[{precision}] bias")
    else:
        assert (out[2] == 0).all()


@pytest.mark.parametrize("path", TINY, ids=lambda p: p.split("/")[-1][:-4])
@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
def test_tiny_golden(path, precision):
    case = util.load_case(path)
    w = synth.make_weights(case["cfg"], case["seed"])
    src = synth.make_source_embeddings(case["cfg"], case["seed"], dtype=case["src_dtype"])
    model = util.hip_model(case["cfg"], w, precision)
    out = util.hip_forward(model, case["ids"], src, case["lang"])
    _check(case, out, precision)


# (the precisions of a case run back to back — the top-most parametrize varies fastest — so that util.seeded_inputs' one-entry cache hits)
@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("path", REAL, ids=lambda p: p.split("/")[-1][:-4])
def test_real_shape_golden(path, precision):
    case = util.load_case(path)
    w, src = util.seeded_inputs(case["cfg"], case["seed"], case["src_dtype"])
    model = util.hip_model(case["cfg"], w, precision)
    del w
    out = util.hip_forward(model, case["ids"], src, case["lang"])
    _check(case, out, precision)


@pytest.mark.parametrize("precision,residual_lo", [("f16", 0), ("bf16", 2)])
@pytest.mark.parametrize("path", REAL, ids=lambda p: p.split("/")[-1][:-4])
def test_real_shape_golden_other_residual_stream(path, precision, residual_lo):
    """The encoder's residual stream is 16-bit in f16 mode and fp32 in bf16 mode (zett_set_option "residual_lo", default 1).
    The other setting of each mode is an A/B option and must hold too: f16 on the fp32 stream inside the f16 tolerance; bf16
    on the 16-bit stream (8 significand bits per layer: why it is not the default) inside twice the bf16 rel-L2 tolerance."""
    import torch
    case = util.load_case(path)
    w, src = util.seeded_inputs(case["cfg"], case["seed"], case["src_dtype"])
    model = util.hip_model(case["cfg"], w, precision)
    model.engine(torch.device("cuda:0"), precision).set_option("residual_lo", residual_lo)
    out = util.hip_forward(model, case["ids"], src, case["lang"])
    if precision == "f16":
        _check(case, out, precision)
    else:
        keep = ~util.all_pad_rows(case["cfg"], case["ids"])
        for got, want in ((out[0], case["pred_in"]), (out[1], case["pred_out"])):
            if want is not None:
                rel = np.linalg.norm(got[keep] - want[keep]) / np.linalg.norm(want[keep])
                assert rel < 2e-2, rel


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("path", BIG, ids=lambda p: p.split("/")[-1][:-4])
def test_big_batch_golden(path, precision):
    """640 rows per real shape: ~1 500 packed positions, so every large GEMM runs on its 256x256 tile (gemm4d for
    K >= 2048, gemm8r below and in fp32 mode) and meets outputs of the REFERENCE directly; the fixture holds the
    reference's rows for a 32-row sample."""
    case = util.load_case(path)
    w, src = util.seeded_inputs(case["cfg"], case["seed"], case["src_dtype"])
    model = util.hip_model(case["cfg"], w, precision)
    del w
    out = util.hip_forward(model, case["ids"], src, case["lang"])
    _check(case, [None if o is None else o[case["sample"]] for o in out], precision)


_EDGE_ORACLE = {}


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("hidden,heads,rows,lang,layers", [(640, 10, 1237, True, 3), (640, 10, 333, False, 2), (1152, 18, 2051, True, 3)])
def test_fold_epilogues_on_edge_shapes_vs_oracle(precision, hidden, heads, rows, lang, layers):
    """The LayerNorm-fold producers (fp32 stream in bf16, 16-bit residual stream in f16: residual rows requested from the last
    K step, clamped at the matrix edge) on shapes none of the reference-shaped goldens has: a hidden width that fills only half
    of the last 256-column tile (640 = 2.5 tiles, 1152 = 4.5), row counts that end inside a 64-row pass, with / without the
    language token and the pair lever.  Checked against the numpy oracle at the tolerance of the arithmetic."""
    from oracle import hypernet_ref
    cfg, _, _, hist = synth.workload("xlmr_gpt2")
    cfg = dict(cfg, n_embd=256, hn_hidden_size=hidden, hn_intermediate_size=2 * hidden, hn_num_attention_heads=heads,
               hn_n_layers=layers, hn_embed_lang_id=lang, original_vocab_size=5000, hn_n_extra_tokens=7, vocab_size=5007,
               separate_out_embeddings=True)
    w = synth.make_weights(cfg, seed=9)
    src = synth.make_source_embeddings(cfg, seed=9)
    ids = synth.make_surface_forms(cfg, rows, seed=9, hist=hist, n_special=2)
    li = 3 if lang else None
    key = (hidden, heads, rows, lang, layers)
    if key not in _EDGE_ORACLE:           # (the oracle's CPU forward is most of this test's time: once per shape, not per precision)
        _EDGE_ORACLE[key] = hypernet_ref.forward(w, cfg, ids, src, lang_index=li)
    want = _EDGE_ORACLE[key]
    model = util.hip_model(cfg, w, precision)
    got = util.hip_forward(model, ids, src, li)
    keep = ~util.all_pad_rows(cfg, ids)
    for g, r, what in zip(got, want, ("pred_in", "pred_out", "bias")):
        util.CLOSE[precision](g[keep], r[keep], f"edge {hidden} {precision} {what}")
