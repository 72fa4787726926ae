"""Range guard of the 16-bit arithmetic (include/zett_hip.h zett_check_range).

f16 operands overflow above 65504.  LayerNorm'd activations and embedding-scale weights stay far inside, but with the
LayerNorm fold the operand copy of the RAW residual sum is rounded to half: a checkpoint with a massive activation in the
residual stream leaves the range.  Nothing may be silent: the device raises a bit where it happens, the predicted
embeddings are checked as the catch-all, zett_finalize refuses weights that do not fit, and the Python layer repeats the
call with bf16 operands (fp32's exponent range) with a warning.  The fallback's result is held to the oracle with the
tolerance of bf16 arithmetic, and that tolerance to what bf16 operand rounding itself costs (oracle emulation)."""
import warnings

import numpy as np
import pytest
import torch

from tests import util
from zett_amd import _lib, synth
from zett_amd.dims import HypernetDims

pytestmark = pytest.mark.gpu


def _cfg():
    cfg, *_ = synth.workload("tiny")
    # H = 512: the LayerNorm fold is on (H % 128 == 0, H >= 512), so the residual sum really is written as a 16-bit operand
    return dict(cfg, n_embd=256, hn_hidden_size=512, hn_intermediate_size=1024, hn_num_attention_heads=8)


def _engine(cfg, weights, precision):
    from zett_amd.hypernet import HipEngine
    eng = HipEngine(HypernetDims.from_config(cfg), 1e-5, torch.device("cuda:0"), precision)
    eng.load_weights({k: torch.from_numpy(v).cuda() for k, v in weights.items()})
    return eng


def _massive(weights, value=1.0e5):
    """One channel of the residual stream that no LayerNorm has touched yet carries `value`: layer 0's attention-output
    bias feeds the pre-LayerNorm sum directly."""
    w = dict(weights)
    b = w["model.encoder.layer.0.attention.output.dense.bias"].copy()
    b[7] = value
    w["model.encoder.layer.0.attention.output.dense.bias"] = b
    return w


def test_healthy_checkpoint_raises_nothing():
    cfg = _cfg()
    w = synth.make_weights(cfg, seed=21)
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 21)).cuda()
    ids = torch.from_numpy(synth.make_surface_forms(cfg, 700, seed=21, n_special=2)).cuda()
    for precision in ("f16", "bf16", "f32"):
        eng = _engine(cfg, w, precision)
        eng.forward(ids, src, 2)
        assert eng.range_flags() == 0, precision
        eng.forward(ids[:0], src, 2)                    # an empty call clears the word too
        assert eng.range_flags() == 0
        eng.close()


@pytest.mark.parametrize("rows", [64, 700])             # 128x128 tile / the large tiles with the fold's producer epilogue
def test_massive_activation_fires_the_guard_and_falls_back_to_bf16(rows):
    from oracle import hypernet_ref
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg = _cfg()
    w = _massive(synth.make_weights(cfg, seed=21))
    src_np = synth.make_source_embeddings(cfg, 21)
    ids_np = synth.make_surface_forms(cfg, rows, seed=21, n_special=1)
    src, ids = torch.from_numpy(src_np).cuda(), torch.from_numpy(ids_np).cuda()
    # the engine alone: asynchronous forward, the word says what happened; the same call in bf16 and f32 is clean
    eng = _engine(cfg, w, "f16")
    out16 = eng.forward(ids, src, 2)
    flags = eng.range_flags()
    assert flags & _lib.RANGE_ACTIVATION, flags
    assert flags & _lib.RANGE_OUTPUT and not bool(torch.isfinite(out16[0]).all())        # inf propagates to the outputs of its row
    with pytest.raises(_lib.RangeError):
        _lib.check(eng.lib.zett_check_range(eng.handle, None, None), "zett_check_range")
    eng.close()
    for precision in ("bf16", "f32"):
        e2 = _engine(cfg, w, precision)
        e2.forward(ids, src, 2)
        assert e2.range_flags() == 0, precision
        e2.close()
    # the drop-in class: f16 by default, one warning, the result of the bf16 repeat, and it stays on bf16
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to("cuda:0")
    assert model.precision == "f16"
    with pytest.warns(UserWarning, match="left the half range"):
        got = model(ids, source_embeddings=src, lang_index=torch.tensor(2))
    assert model.precision == "bf16"
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        again = model(ids, source_embeddings=src, lang_index=torch.tensor(2))
    assert all(torch.equal(a, b) for a, b in zip(got, again))
    # against the oracle: the tolerance of bf16 arithmetic, itself bounded by what rounding every GEMM
    # operand of the oracle to bf16 costs on this very checkpoint
    want = hypernet_ref.forward(w, cfg, ids_np, src_np, lang_index=2)
    hypernet_ref.set_operand_rounding("bf16")
    try:
        emulated = hypernet_ref.forward(w, cfg, ids_np, src_np, lang_index=2)
    finally:
        hypernet_ref.set_operand_rounding(None)
    keep = ~util.all_pad_rows(cfg, ids_np)
    for g, r, e, what in zip(got, want, emulated, ("pred_in", "pred_out", "bias")):
        g = g.cpu().numpy()[keep]
        util.assert_bf16_close(g, r[keep], f"bf16 fallback {what}")
        rel = np.linalg.norm(g - r[keep]) / np.linalg.norm(r[keep])
        rel_emulated = np.linalg.norm(e[keep] - r[keep]) / np.linalg.norm(r[keep])
        assert rel <= 2.0 * rel_emulated + 1e-4, (what, rel, rel_emulated)


def test_source_embeddings_beyond_the_half_range():
    cfg = _cfg()
    w = synth.make_weights(cfg, seed=22)
    src_np = synth.make_source_embeddings(cfg, 22)
    ids_np = synth.make_surface_forms(cfg, 300, seed=22)
    src_np[int(ids_np[5, 0]), 3] = 3.0e5                  # a referenced row
    src, ids = torch.from_numpy(src_np).cuda(), torch.from_numpy(ids_np).cuda()
    eng = _engine(cfg, w, "f16")
    eng.forward(ids, src, 1)
    assert eng.range_flags() & _lib.RANGE_SOURCE
    # an unreferenced row may hold anything: it is never read
    src2 = torch.from_numpy(synth.make_source_embeddings(cfg, 22)).cuda()
    unused = sorted(set(range(3, cfg["original_vocab_size"])) - set(ids_np.ravel().tolist()))[0]
    src2[unused] = float("inf")
    eng.forward(ids, src2, 1)
    assert eng.range_flags() == 0
    eng.close()
    # non-finite inputs are not a range problem of f16: bf16 reports non-finite outputs, as computed
    src3 = src2.clone()
    src3[unused] = 0.0
    src3[int(ids_np[7, 0]), 0] = float("nan")
    e2 = _engine(cfg, w, "bf16")
    out = e2.forward(ids, src3, 1)
    assert e2.range_flags() == _lib.RANGE_OUTPUT
    assert not bool(torch.isfinite(out[0][7]).all()) and bool(torch.isfinite(out[0][8]).all())
    e2.close()


def test_weight_beyond_the_half_range_is_refused_at_finalize():
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg = _cfg()
    w = synth.make_weights(cfg, seed=23)
    big = w["model.encoder.layer.1.intermediate.dense.weight"].copy()
    big[11, 13] = 7.0e4
    w["model.encoder.layer.1.intermediate.dense.weight"] = big
    with pytest.raises(_lib.RangeError, match="half range"):
        _engine(cfg, w, "f16")
    _engine(cfg, w, "bf16").close()
    # a LayerNorm gain that pushes W * gamma over the edge is caught where the folded operand is built
    w2 = synth.make_weights(cfg, seed=23)
    g = w2["model.encoder.layer.0.attention.output.LayerNorm.weight"].copy()
    g[5] = 5.0e6
    w2["model.encoder.layer.0.attention.output.LayerNorm.weight"] = g
    with pytest.raises(_lib.RangeError):
        _engine(cfg, w2, "f16")
    # the class falls back instead of failing
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to("cuda:0")
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 23)).cuda()
    ids = torch.from_numpy(synth.make_surface_forms(cfg, 64, seed=23)).cuda()
    with pytest.warns(UserWarning, match="bf16"):
        out = model(ids, source_embeddings=src, lang_index=torch.tensor(0))
    assert model.precision == "bf16" and bool(torch.isfinite(out[0]).all())
