"""Parity at BASELINE.json's full sizes (SURVEY.md §8d T1, configs C2-C5 and the north-star headline).

Every configuration runs its whole target vocabulary through the C ABI on the GPU, in f16 (the default)
and bf16 arithmetic, and is checked three ways:

  * a fixed sample of rows against `oracle.hypernet_ref.forward` (the as-written reference math in
    numpy fp32, itself pinned to the reference by tests/golden) — this is where the 256x256 GEMM tiles
    that carry the benchmark (gemm4d at K >= 2048, gemm8r below) meet the oracle DIRECTLY: the launches
    have M = 20 000 .. 600 000 rows here, whereas the golden fixtures' 16-64 rows run on the 128x128 tile;
  * the 8-way row-sharded result reassembles to the single-GPU result bit for bit (what the all-gather
    of `predict_sharded` relies on), with the small per-shard launches taking other tile kernels;
  * the encoder chunking (`max_chunk_tokens`) changes no bit: the 262 144-row Llama-3 workload needs
    five chunks at the default limit, the others are additionally run with a limit that forces chunks.

Also here: the heavy-tailed checkpoint.  The synthetic weights are N(0, 0.02^2); real checkpoints have
outlier channels.  With Student-t (3 degrees of freedom) Linear weights, outlier LayerNorm gains and a
heavy-tailed source table, f16 arithmetic (the default) stays inside its tolerance with the same margin as on
normal weights; bf16 arithmetic lands at rel-L2 1.04e-2 at the 4096-wide shape — on the wrong side of the 1e-2
line it clears by 3 % on normal weights (the operand rounding error of a dot product is relative, so it barely
moves with the tails: it was on the edge before).  That measurement is why bf16 is not the default.  bf16 mode is
bounded there by what bf16 operands cost in the oracle itself (operand-rounding emulation, + 10 %): the error is the
arithmetic's, the arithmetic is the reference CLI's default, and the kernel adds none of its own.
"""
import numpy as np
import pytest
import torch

from tests import util
from zett_amd import synth
from zett_amd.dims import HypernetDims

pytestmark = pytest.mark.gpu

SAMPLE_ROWS = 96


def _engine(cfg, weights, precision):
    from zett_amd.hypernet import HipEngine
    dev = torch.device("cuda:0")
    eng = HipEngine(HypernetDims.from_config(cfg), 1e-5, dev, precision)
    eng.load_weights(weights)
    return eng


def _run(eng, ids, src, lang):
    out = eng.forward(torch.as_tensor(ids).to(eng.device), src, lang)
    torch.cuda.synchronize()
    return out


def _eq(a, b):
    return all((x is None and y is None) or torch.equal(x, y) for x, y in zip(a, b))


def _check_sample(full, want, sample, precision, what):
    close = util.CLOSE[precision]
    idx = torch.from_numpy(sample).cuda()
    for got, ref, name in zip(full, want, ("pred_in", "pred_out", "bias")):
        if ref is None:
            assert got is None
            continue
        close(got[idx].cpu().numpy(), ref, f"{what} {precision} {name}")


@pytest.mark.parametrize("name", ["xlmr_gpt2", "tinyllama_neox", "mistral_gpt2_32k", "mistral_neox", "llama3_256k"])
def test_full_size_parity(name):
    from bench import device_weights
    from oracle import hypernet_ref
    from zett_amd.sharding import shard_bounds

    cfg, rows, src_dtype, hist = synth.workload(name)
    dev = torch.device("cuda:0")
    weights = device_weights(cfg, dev, seed=0)
    src_np = synth.make_source_embeddings(cfg, 0, dtype=src_dtype)
    src = torch.from_numpy(src_np).cuda()
    ids = synth.make_surface_forms(cfg, rows, seed=0, hist=hist, n_special=2)
    lang = 3 if cfg.get("hn_embed_lang_id") else -1
    # oracle on a row sample that always holds the two all-pad rows' neighbours, the first and the last row
    rng = np.random.default_rng(0)
    sample = np.unique(np.concatenate([[2, 3, rows - 1], rng.choice(np.arange(2, rows), SAMPLE_ROWS - 3, replace=False)]))
    hypernet_ref.set_matmul_backend("torch")
    want = hypernet_ref.forward({k: v.float().cpu().numpy() for k, v in weights.items()}, cfg, ids[sample], src_np, None if lang < 0 else lang)
    # f32 — the mode whose tolerance north_star names — at full size on the headline workload and on the narrow one
    # (its launches run gemm8r's fp32 instantiation at M = 77 k / 169 k rows; 0.5 s per forward)
    precisions = ("f16", "bf16", "f32") if name in ("mistral_gpt2_32k", "xlmr_gpt2") else ("f16", "bf16")
    for precision in precisions:
        eng = _engine(cfg, weights, precision)
        full = _run(eng, ids, src, lang)
        assert all(t is None or bool(torch.isfinite(t).all()) for t in full)
        assert eng.range_flags() == 0                        # range guard: nothing left the operand range
        st = eng.stats()
        assert st["rows"] == rows and 0 < st["packed_tokens"] <= rows * 8 and 0 < st["distinct_ids"]
        if name == "llama3_256k":
            assert st["chunks"] >= 5, st["chunks"]          # 131 072 packed positions per chunk, ~620 k in the workload
        _check_sample(full, want, sample, precision, name)
        if precision == "f16":
            parts = [_run(eng, ids[slice(*shard_bounds(rows, 8, r))], src, lang) for r in range(8)]
            cat = [None if parts[0][k] is None else torch.cat([p[k] for p in parts]) for k in range(3)]
            assert _eq(cat, full), f"{name}: 8 row shards differ from the whole vocabulary"
            del parts, cat
            if name != "llama3_256k":                        # force several encoder chunks
                eng.set_option("max_chunk_tokens", 24576)
                chunked = _run(eng, ids, src, lang)
                assert eng.stats()["chunks"] >= 3
                assert _eq(chunked, full), f"{name}: encoder chunking changed the result"
                del chunked
        eng.close()
        del full


def _student_t(gen, shape, device, std):
    """Student-t with 3 degrees of freedom, scaled to the given standard deviation (variance of t3 is 3)."""
    z = torch.randn(shape, device=device, generator=gen)
    chi = torch.randn((3,) + tuple(shape), device=device, generator=gen).square_().sum(0).div_(3.0).sqrt_()
    return z.div_(chi).mul_(std / 3.0 ** 0.5)


@pytest.mark.parametrize("name,rows", [("tinyllama_neox", 4096), ("mistral_gpt2_32k", 2048)])
def test_heavy_tailed_checkpoint_stays_inside_tolerance(name, rows):
    from bench import device_weights
    from oracle import hypernet_ref

    cfg, _, src_dtype, hist = synth.workload(name)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    weights = device_weights(cfg, dev, seed=7)
    for k in list(weights):
        w = weights[k]
        if w.dim() == 2 and w.shape[0] > 1 and w.shape[1] > 1 and not k.endswith("embeddings.weight"):
            weights[k] = _student_t(gen, w.shape, dev, 0.02)                   # every Linear weight
        elif k.endswith("LayerNorm.weight") or k.endswith("ln.weight"):
            out = torch.rand(w.shape, device=dev, generator=gen) < 0.01        # 1 % outlier gains of 4..8
            weights[k] = torch.where(out, 4.0 + 4.0 * torch.rand(w.shape, device=dev, generator=gen), w)
    d = HypernetDims.from_config(cfg)
    src = _student_t(gen, (d.original_vocab_size, d.n_in_embd), dev, 0.02)
    ids = synth.make_surface_forms(cfg, rows, seed=7, hist=hist, n_special=1)
    sample = np.arange(1, 129)
    hypernet_ref.set_matmul_backend("torch")
    want = hypernet_ref.forward({k: v.float().cpu().numpy() for k, v in weights.items()}, cfg, ids[sample], src.cpu().numpy(), None)
    kurt = float(((src - src.mean()) ** 4).mean() / src.var() ** 2)
    assert kurt > 6.0, kurt                                  # the table really is heavy-tailed (normal: 3)
    # What bf16 operands cost on THIS checkpoint whatever the kernel: the oracle with both operands of every Linear rounded
    # to bf16 (fp32 accumulate).  That is (an upper bound on the accuracy of) the reference CLI's own default arithmetic
    # (scripts/transfer.py:41: bfloat16 parameters and compute), so bf16 mode is held to SURVEY 8d's 1e-2 OR to that
    # emulation + 10 %, whichever is larger: the HIP path may not add error of its own, and no bare relaxed number remains.
    hypernet_ref.set_operand_rounding("bf16")
    try:
        emulated = hypernet_ref.forward({k: v.float().cpu().numpy() for k, v in weights.items()}, cfg, ids[sample], src.cpu().numpy(), None)
    finally:
        hypernet_ref.set_operand_rounding(None)
    for precision in ("f16", "bf16"):
        eng = _engine(cfg, weights, precision)
        full = _run(eng, ids, src, -1)
        assert eng.range_flags() == 0                        # heavy tails, but nothing near the half range
        if precision != "bf16":
            _check_sample(full, want, sample, precision, f"{name} heavy-tailed")
        else:
            idx = torch.from_numpy(sample).cuda()
            for got, ref, emu, what in zip(full, want, emulated, ("pred_in", "pred_out", "bias")):
                rel_emu = float(np.linalg.norm(emu - ref) / np.linalg.norm(ref))
                lim = util.BF16_REL_L2 if ref.ndim > 1 else util.BF16_REL_L2_VECTOR
                util.assert_bf16_close(got[idx].cpu().numpy(), ref, f"{name} heavy-tailed bf16 {what} (emulated bf16 operands: {rel_emu:.2e})",
                                       rel_max=max(lim, 1.1 * rel_emu))
        eng.close()


@pytest.mark.parametrize("name", ["mistral_gpt2_32k", "xlmr_gpt2"])
def test_training_backward_against_the_inference_forward_full_size(name):
    """N4 at BASELINE's sizes, through a size-independent property.  The differentiable forward (zett_amd/autograd.py, packed
    schedule, exact fp32 MFMA) of the WHOLE vocabulary must reproduce the inference engine's fp32 outputs, and its backward must
    predict what the inference engine measures: with L = <outputs, cotangents> and g = dL/d(parameters) from the backward,
    moving every parameter by +-eps g changes L (evaluated by the inference path, an independent schedule with levers 1-4) by
    2 eps |g|^2 up to third order (measured at the Mistral shape: ratio 0.99989)."""
    from bench import device_weights
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet

    cfg, rows, src_dtype, hist = synth.workload(name)
    dev = torch.device("cuda:0")
    model = ZettHypernet(ZettHypernetConfig(**cfg)).to(dev)
    params = dict(model.named_parameters())
    with torch.no_grad():
        for pname, w in device_weights(cfg, dev, seed=0).items():
            params[pname].copy_(w)
    model.precision, model.train_precision, model.train_packed = "f32", "f32", True
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 0, dtype=src_dtype)).to(dev)
    ids = torch.from_numpy(synth.make_surface_forms(cfg, rows, seed=0, hist=hist, n_special=2)).to(dev)
    lang = torch.tensor(3) if cfg.get("hn_embed_lang_id") else None

    def infer():
        model.eval()
        model.refresh_weights()
        with torch.no_grad():
            out = model(ids, source_embeddings=src, lang_index=lang)
        torch.cuda.synchronize()
        return out

    base = infer()
    g = torch.Generator(device=dev).manual_seed(5)
    cot = [None if o is None else torch.randn(o.shape, device=dev, generator=g) for o in base]
    loss_of = lambda out: sum(float((o.double() * c.double()).sum()) for o, c in zip(out, cot) if o is not None)

    model.requires_grad_(True).train()
    out = model(ids, source_embeddings=src, lang_index=lang)
    for got, want, what in zip(out, base, ("pred_in", "pred_out", "bias")):
        if want is None:
            assert got is None
            continue
        rel = float((got.detach().double() - want.double()).norm() / want.double().norm())
        assert rel < 2e-5, f"{name} {what}: differentiable forward vs inference engine (both fp32) rel-L2 {rel:.2e}"
    sum((o * c).sum() for o, c in zip(out, cot) if o is not None).backward()
    grads = {n: p.grad for n, p in params.items() if p.grad is not None}
    assert len(grads) >= len(params) - 1                       # (only the unused word-embedding table may be without one)
    del out
    g2 = sum(float(v.double().pow(2).sum()) for v in grads.values())
    theta = sum(float(params[n].detach().double().pow(2).sum()) for n in grads) ** 0.5
    eps = 1e-5 * theta / g2 ** 0.5                             # a step of 1e-5 of the parameter norm along the gradient (1e-3: 47 % third-order
                                                               # error at the Mistral shape, 1e-4: 1 %, 1e-5: 1e-4, 1e-6: 3e-5 - tools measured)
    model.requires_grad_(False)
    losses = []
    for sign in (1.0, -2.0):                                   # theta + eps g, then theta - eps g
        with torch.no_grad():
            for n, v in grads.items():
                params[n].add_(v, alpha=sign * eps)
        losses.append(loss_of(infer()))
    measured, predicted = losses[0] - losses[1], 2.0 * eps * g2
    assert abs(measured - predicted) < 5e-3 * abs(predicted), f"{name}: dL measured {measured:.6e}, predicted by the backward {predicted:.6e}"
