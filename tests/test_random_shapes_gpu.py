"""Randomised shapes and flags: the HIP forward (fp32 mode, exact-parity tolerance; bf16 mode, its tolerance) against
the oracle on configurations the fixed fixtures do not cover — hidden sizes that are not multiples of 128 or 256
(N edges of the GEMM tiles), head dims 16..128, one to eight surface positions, forced large-tile GEMM variants on
small shapes.  (Every contraction width — E_in, H, I — must be a multiple of 64: zett_create rejects others.)"""
import numpy as np
import pytest
import torch

from tests import util
from zett_amd import synth

pytestmark = pytest.mark.gpu

BASE = dict(synth.workload("tiny")[0])


def _cases():
    rng = np.random.default_rng(20240917)
    out = []
    for k in range(10):
        h = int(rng.choice([64, 128, 192, 320, 448]))
        heads = int(rng.choice([x for x in (1, 2, 4, 8) if h % x == 0 and (h // x) in (16, 32, 64, 128)] or [h // 64]))
        cfg = dict(BASE, n_embd=int(rng.choice([64, 128, 192, 320])), hn_hidden_size=h, hn_intermediate_size=int(rng.choice([128, 192, 384])),
                   hn_num_attention_heads=heads, separate_out_embeddings=bool(rng.integers(2)), hn_embed_lang_id=bool(rng.integers(2)),
                   hn_rescale_embeddings=bool(rng.integers(2)), hn_predict_bias=bool(rng.integers(2)), hn_single_head=bool(rng.integers(2)),
                   hn_surface_maxlen=int(rng.choice([1, 3, 7, 8])))
        out.append((k, cfg, int(rng.integers(1, 700)), int(rng.choice([0, 2, 8])) if k % 3 else 7))
    return out


@pytest.mark.parametrize("k,cfg,rows,variant", _cases(), ids=lambda v: str(v) if isinstance(v, int) else None)
def test_random_shape(k, cfg, rows, variant):
    from oracle import hypernet_ref
    w = synth.make_weights(cfg, 100 + k)
    src = synth.make_source_embeddings(cfg, 100 + k)
    ids = synth.make_surface_forms(cfg, rows, seed=100 + k, n_special=min(2, rows))
    lang = 1 if cfg["hn_embed_lang_id"] else None
    want = hypernet_ref.forward(w, cfg, ids, src, lang)
    keep = ~util.all_pad_rows(cfg, ids)
    for precision, close in (("f32", util.assert_f32_close), ("bf16", util.assert_bf16_close)):
        model = util.hip_model(cfg, w, precision)
        model.engine(torch.device("cuda:0")).set_option("gemm_variant", variant)
        got = util.hip_forward(model, ids, src, lang)
        for g, t, name in zip(got, want, ("pred_in", "pred_out", "bias")):
            if t is None:
                assert g is None
                continue
            if name == "bias" and not cfg["hn_predict_bias"]:
                assert (g == 0).all()
                continue
            if keep.sum() >= (8 if precision == "bf16" and name == "bias" else 1):
                close(g[keep], t[keep], f"case {k} {precision} {name}")
        del model
