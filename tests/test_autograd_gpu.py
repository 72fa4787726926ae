"""Training use of the forward (reference: train.py:1007-1013, 1191-1197): zett_amd/autograd.py on
the GPU against torch.autograd of a float64 torch restatement of the oracle (tests/torch_port.py, itself checked against
oracle/hypernet_ref.py here).  Primitives one by one, then the whole hypernetwork: outputs, the gradient of every
parameter, all flag combinations of the tiny shape, and the drop-in class with requires_grad parameters."""
import numpy as np
import pytest
import torch

from tests import torch_port, util
from zett_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _ops():
    from zett_amd.autograd import Ops
    return Ops(torch.device(DEV))


def test_gemm_and_linear_backward_match_torch():
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(0)
    for m, n, k in ((37, 64, 96), (300, 384, 128), (1000, 256, 256), (5, 128, 32)):
        x = torch.randn(m, k, device=DEV, generator=g)
        w = torch.randn(n, k, device=DEV, generator=g) * 0.1
        b = torch.randn(n, device=DEV, generator=g)
        r = torch.randn(m, n, device=DEV, generator=g)
        y = ops.gemm(x, w, b, residual=r)
        assert _rel(y, x.double() @ w.double().T + b.double() + r.double()) < 2e-6
        dy = torch.randn(m, n, device=DEV, generator=g)
        dx, dw, db = ops.linear_bwd(dy, x, w)
        assert _rel(dx, dy.double() @ w.double()) < 2e-6 and _rel(dw, dy.double().T @ x.double()) < 2e-6 and _rel(db, dy.double().sum(0)) < 2e-6
    with pytest.raises(ValueError):
        ops.gemm(torch.randn(4, 20, device=DEV), torch.randn(8, 20, device=DEV))         # contraction width not a multiple of 32
    # tall and narrow: the weight gradient is summed from row slices that run as concurrent launches
    x = torch.randn(40001, 384, device=DEV, generator=g)
    w = torch.randn(256, 384, device=DEV, generator=g) * 0.1
    dy = torch.randn(40001, 256, device=DEV, generator=g)
    dx, dw, db = ops.linear_bwd(dy, x, w)
    assert _rel(dw, dy.double().T @ x.double()) < 2e-6 and _rel(dx, dy.double() @ w.double()) < 2e-6
    dw2 = ops.linear_bwd(dy, x, w)[1]
    assert torch.equal(dw, dw2)                                                          # deterministic: fixed slices, fixed order of the sum


def test_layernorm_gelu_rowops_match_torch():
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(1)
    x = (torch.randn(77, 384, device=DEV, generator=g) * 3 + 0.5).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(384, device=DEV, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(384, device=DEV, generator=g)).requires_grad_(True)
    dy = torch.randn(77, 384, device=DEV, generator=g)
    y, st = ops.layernorm(x.detach(), gamma.detach(), beta.detach(), 1e-5)
    ref = torch.nn.functional.layer_norm(x.double(), (384,), gamma.double(), beta.double(), 1e-5)
    assert _rel(y, ref) < 1e-6
    ref.backward(dy.double())
    dx, dg, db = ops.layernorm_bwd(dy, x.detach(), st, gamma.detach())
    assert _rel(dx, x.grad) < 2e-5 and _rel(dg, gamma.grad) < 1e-5 and _rel(db, beta.grad) < 1e-5
    for kind, approx in ((1, "tanh"), (2, "none")):
        z = (torch.randn(5000, device=DEV, generator=g) * 2.5).requires_grad_(True)
        dh = torch.randn(5000, device=DEV, generator=g)
        h = ops.gelu(z.detach(), kind)
        ref = torch.nn.functional.gelu(z.double(), approximate=approx)
        assert float((h.double() - ref.detach()).abs().max()) < 1e-6
        ref.backward(dh.double())
        assert float((ops.gelu_bwd(z.detach(), dh, kind).double() - z.grad.double()).abs().max()) < 2e-6
    a = torch.randn(50, 64, device=DEV, generator=g)
    assert _rel(ops.colsum(a), a.double().sum(0)) < 1e-6
    t = ops.transpose(a)
    assert t.shape == (64, 64) and torch.equal(t[:, :50], a.T) and bool((t[:, 50:] == 0).all())
    w = torch.randn(64, device=DEV, generator=g)
    s = torch.randn(50, device=DEV, generator=g)
    assert _rel(ops.rowdot(a, w, s[:1]), a.double() @ w.double() + s[0].double()) < 1e-6
    assert _rel(ops.add_outer(a, s, w), a.double() + s.double()[:, None] * w.double()[None, :]) < 1e-6        # (one fused multiply-add on the device)
    assert torch.equal(ops.scale_rows(a, s), a * s[:, None])
    assert torch.equal(ops.affine_cols(a, w, None), a * w) and torch.equal(ops.affine_cols(a, None, w), a + w)


def test_layernorm_widths_residual_gradient_and_partials():
    """every register layout of the LayerNorm kernels (H <= 1024, 2048, 4096, 8192 incl. a partly filled last pass), more rows
    than workgroups (the partial sums of the parameter gradients), and the second gradient input (the residual branch)"""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(11)
    for rows, h in ((3, 64), (130, 768), (1500, 1024), (70, 2048), (33, 3072), (20, 4096), (9, 8192)):
        x = (torch.randn(rows, h, device=DEV, generator=g) * 2 - 0.3).requires_grad_(True)
        gamma = (1 + 0.1 * torch.randn(h, device=DEV, generator=g)).requires_grad_(True)
        beta = (0.1 * torch.randn(h, device=DEV, generator=g)).requires_grad_(True)
        dy, dy2 = torch.randn(rows, h, device=DEV, generator=g), torch.randn(rows, h, device=DEV, generator=g)
        y, st = ops.layernorm(x.detach(), gamma.detach(), beta.detach(), 1e-5)
        ref = torch.nn.functional.layer_norm(x.double(), (h,), gamma.double(), beta.double(), 1e-5)
        assert _rel(y, ref) < 1e-6, (rows, h)
        ref.backward(dy.double() + dy2.double())
        dx, dg, db = ops.layernorm_bwd(dy, x.detach(), st, gamma.detach(), dy2=dy2)
        assert _rel(dx, x.grad) < 2e-5 and _rel(dg, gamma.grad) < 1e-5 and _rel(db, beta.grad) < 1e-5, (rows, h)
        again = ops.layernorm_bwd(dy, x.detach(), st, gamma.detach(), dy2=dy2)
        assert all(torch.equal(a, b) for a, b in zip((dx, dg, db), again))                   # fixed grid, fixed order: deterministic
    with pytest.raises(Exception):
        ops.layernorm(torch.randn(4, 8200, device=DEV), torch.ones(8200, device=DEV), torch.zeros(8200, device=DEV), 1e-5)


def test_elementwise_and_gelu_wide_and_narrow_paths_agree():
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(12)
    for rows, cols in ((50, 64), (7, 30), (33, 1028)):                                      # 16-byte path, scalar path (cols % 4 != 0), wide again
        a, b = torch.randn(rows, cols, device=DEV, generator=g), torch.randn(rows, cols, device=DEV, generator=g)
        w, s = torch.randn(cols, device=DEV, generator=g), torch.randn(rows, device=DEV, generator=g)
        assert torch.equal(ops.add(a, b), a + b) and torch.equal(ops.mul(a, b), a * b)
        assert torch.equal(ops.scale_rows(a, s), a * s[:, None])
        assert torch.equal(ops.affine_cols(a, w, None), a * w) and torch.equal(ops.affine_cols(a, None, w), a + w)
        assert _rel(ops.add_outer(a, s, w), a.double() + s.double()[:, None] * w.double()[None, :]) < 1e-6
        for kind in (1, 2):
            z, dh = a.reshape(-1), b.reshape(-1)
            h1, h2 = ops.gelu(z, kind), ops.gelu(z[1:].clone(), kind)                        # (an odd length takes the scalar path)
            assert torch.equal(h1[1:], h2) or z.numel() % 4
            assert _rel(ops.gelu_bwd(z[1:].clone(), dh[1:].clone(), kind), ops.gelu_bwd(z, dh, kind)[1:]) < 1e-7


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_grad_operands_one_read_matches_the_separate_kernels(precision):
    """zett_op_grad_operands_lo: lo(dy), lo(dy)^T (zero-padded) and the column sums from one read of dy — bit-equal to the
    conversion and transposition kernels, the sums to float64"""
    from zett_amd.autograd import Ops
    ops = Ops(torch.device(DEV), precision)
    g = torch.Generator(device=DEV).manual_seed(13)
    for m, n in ((1, 64), (63, 128), (64, 64), (1000, 192), (4097, 320)):
        dy = torch.randn(m, n, device=DEV, generator=g) * 3
        lo, t, cs = ops.grad_operands(dy)
        assert torch.equal(lo, ops.to_lo(dy)) and torch.equal(t, ops.transpose(dy))
        assert t.shape[1] % 64 == 0 and bool((t[:, m:] == 0).all())
        assert _rel(cs, dy.double().sum(0)) < 1e-6
    x = torch.randn(5000, 256, device=DEV, generator=g)
    w = torch.randn(128, 256, device=DEV, generator=g) * 0.1
    dy = torch.randn(5000, 128, device=DEV, generator=g)
    dx, dw, db = ops.linear_bwd(dy, x, w)
    tol = 2e-2 if precision == "bf16" else 3e-3
    assert _rel(dx, dy.double() @ w.double()) < tol and _rel(dw, dy.double().T @ x.double()) < tol and _rel(db, dy.double().sum(0)) < 1e-6
    # the GELU's backward on the way: dy is the gradient of gelu(z), the three results are those of dy * gelu'(z)
    for kind in (1, 2):
        z = torch.randn(5000, 128, device=DEV, generator=g) * 2
        eff = ops.gelu_bwd(z, dy, kind)
        lo, t, cs = ops.grad_operands(dy, z, kind)
        assert torch.equal(lo, ops.to_lo(eff)) and torch.equal(t, ops.transpose(eff)) and _rel(cs, eff.double().sum(0)) < 1e-6
    # activations that are already 16-bit operands: the GELU written as an operand, LayerNorm's twin; both transposed as they are
    z = torch.randn(777, 192, device=DEV, generator=g)
    for kind in (1, 2):
        h = ops.gelu(z, kind, operand=True)
        assert h.dtype == ops.lo_dtype and torch.equal(h, ops.to_lo(ops.gelu(z, kind)))
        assert torch.equal(ops.transpose(h), ops.transpose(ops.gelu(z, kind)))
    y, st = ops.layernorm(z, torch.ones(192, device=DEV), torch.zeros(192, device=DEV), 1e-5)
    assert torch.equal(y._zett_lo, ops.to_lo(y.clone())) and torch.equal(ops.transpose(y), ops.transpose(y.clone()))
    w2 = torch.randn(64, 192, device=DEV, generator=g)
    assert torch.equal(ops.gemm(y, w2), ops.gemm(y.clone(), w2))           # (the twin is the operand the conversion would have made)
    # the attention context written as a 16-bit operand only
    n, L, heads, d = 11, 5, 2, 64
    qkv = torch.randn(n * L, 3 * heads * d, device=DEV, generator=g)
    mask = torch.ones(n * L, dtype=torch.uint8, device=DEV)
    H = heads * d
    c32, p32 = ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], mask, None, n, L, heads, H)
    c16, p16 = ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], mask, None, n, L, heads, H, operand=True)
    assert c16.dtype == ops.lo_dtype and torch.equal(c16, ops.to_lo(c32)) and torch.equal(p16, p32)


def test_attention_forward_backward_match_torch():
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(2)
    n, L, heads, d = 19, 8, 4, 32
    H = heads * d
    qkv = torch.randn(n * L, 3 * H, device=DEV, generator=g).requires_grad_(True)
    mask = (torch.rand(n, L, device=DEV, generator=g) < 0.6)
    mask[:, 0] = True
    mask[3] = False                                           # a row whose keys are all masked: uniform attention (eager semantics)
    dctx = torch.randn(n * L, H, device=DEV, generator=g)
    qd = qkv.detach()
    ctx, probs = ops.attention(qd[:, :H], qd[:, H:2 * H], qd[:, 2 * H:], mask.to(torch.uint8).view(-1), None, n, L, heads, H)
    q, k, v = [t.view(n, L, heads, d).transpose(1, 2) for t in qkv.double().split(H, dim=1)]
    bias = torch.where(mask, 0.0, torch.finfo(torch.float32).min).double()[:, None, None, :]
    p = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5 + bias, -1)
    ref = (p @ v).transpose(1, 2).reshape(n * L, H)
    assert _rel(ctx, ref) < 2e-6 and float((probs[3] - 1.0 / L).abs().max()) < 1e-6
    ref.backward(dctx.double())
    dqkv = torch.empty_like(qd)
    ops.attention_bwd(dctx, qd[:, :H], qd[:, H:2 * H], qd[:, 2 * H:], probs, None, n, L, heads, H, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:])
    assert _rel(dqkv, qkv.grad) < 2e-5
    # packed rows (each row keeps its own number of positions) and the position-0-only query of the last layer
    lens = torch.randint(1, L + 1, (n,), generator=torch.Generator().manual_seed(4))
    off = torch.zeros(n + 1, dtype=torch.int32)
    off[1:] = lens.cumsum(0)
    T = int(off[-1])
    pq = torch.randn(T, 3 * H, device=DEV, generator=g).requires_grad_(True)
    pmask = torch.rand(T, device=DEV, generator=g) < 0.7
    offd = off.to(DEV)
    pctx, pprobs = ops.attention(pq.detach()[:, :H], pq.detach()[:, H:2 * H], pq.detach()[:, 2 * H:], pmask.to(torch.uint8), offd, n, L, heads, H)
    refs = []
    for r in range(n):
        a, b = int(off[r]), int(off[r + 1])
        qq, kk, vv = [t.view(b - a, heads, d).transpose(0, 1) for t in pq[a:b].double().split(H, dim=1)]
        bb = torch.where(pmask[a:b], 0.0, torch.finfo(torch.float32).min).double()[None, None, :]
        refs.append((torch.softmax(qq @ kk.transpose(-1, -2) * d ** -0.5 + bb, -1) @ vv).transpose(0, 1).reshape(b - a, H))
    pref = torch.cat(refs)
    assert _rel(pctx, pref) < 2e-6
    dpc = torch.randn(T, H, device=DEV, generator=g)
    pref.backward(dpc.double())
    dpq = torch.empty(T, 3 * H, device=DEV)
    pd = pq.detach()
    ops.attention_bwd(dpc, pd[:, :H], pd[:, H:2 * H], pd[:, 2 * H:], pprobs, offd, n, L, heads, H, dpq[:, :H], dpq[:, H:2 * H], dpq[:, 2 * H:])
    assert _rel(dpq, pq.grad) < 2e-5
    q1 = torch.randn(n, H, device=DEV, generator=g).requires_grad_(True)          # one query per row (position 0)
    kv = torch.randn(T, 2 * H, device=DEV, generator=g).requires_grad_(True)
    cctx, cprobs = ops.attention(q1.detach(), kv.detach()[:, :H], kv.detach()[:, H:], pmask.to(torch.uint8), offd, n, L, heads, H, cls_only=True)
    refs = []
    for r in range(n):
        a, b = int(off[r]), int(off[r + 1])
        qq = q1[r].double().view(heads, 1, d)
        kk, vv = [t.view(b - a, heads, d).transpose(0, 1) for t in kv[a:b].double().split(H, dim=1)]
        bb = torch.where(pmask[a:b], 0.0, torch.finfo(torch.float32).min).double()[None, None, :]
        refs.append((torch.softmax(qq @ kk.transpose(-1, -2) * d ** -0.5 + bb, -1) @ vv).reshape(1, H))
    cref = torch.cat(refs)
    assert _rel(cctx, cref) < 2e-6
    dcc = torch.randn(n, H, device=DEV, generator=g)
    cref.backward(dcc.double())
    dq1, dkv = torch.empty(n, H, device=DEV), torch.empty(T, 2 * H, device=DEV)
    ops.attention_bwd(dcc, q1.detach(), kv.detach()[:, :H], kv.detach()[:, H:], cprobs, offd, n, L, heads, H, dq1, dkv[:, :H], dkv[:, H:], cls_only=True)
    assert _rel(dq1, q1.grad) < 2e-5 and _rel(dkv, kv.grad) < 2e-5
    # indexed rows
    src = torch.randn(40, 96, device=DEV, generator=g)
    idx = torch.randint(0, 40, (300,), device=DEV, generator=g, dtype=torch.int32)
    a = torch.randn(300, 96, device=DEV, generator=g)
    assert torch.equal(ops.gather_rows(src, idx), src[idx.long()]) and torch.equal(ops.gather_rows(src, idx, a), a + src[idx.long()])
    dst = torch.zeros(40, 96, device=DEV)
    ops.scatter_add_rows(dst, idx, a)
    assert _rel(dst, torch.zeros(40, 96, device=DEV, dtype=torch.float64).index_add_(0, idx.long(), a.double())) < 1e-6


@pytest.mark.parametrize("L,heads,d", [(1, 2, 64), (2, 3, 16), (3, 4, 128), (5, 1, 32), (8, 5, 128), (9, 2, 64), (16, 4, 128), (17, 2, 64), (32, 1, 128), (7, 2, 256), (16, 1, 192)])
def test_attention_register_layouts(L, heads, d):
    """every register layout of the attention kernels (positions rounded up to 2 / 4 / 8 / 16 / 32, one / two / four 64-column
    slices of the head dim, head counts that do not fill the four waves of a workgroup): dense rows with random masks"""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(100 + L)
    n, H = 13, heads * d
    qkv = torch.randn(n * L, 3 * H, device=DEV, generator=g).requires_grad_(True)
    mask = torch.rand(n, L, device=DEV, generator=g) < 0.7
    mask[:, 0] = True
    if L > 1:
        mask[5] = False
    dctx = torch.randn(n * L, H, device=DEV, generator=g)
    qd = qkv.detach()
    ctx, probs = ops.attention(qd[:, :H], qd[:, H:2 * H], qd[:, 2 * H:], mask.to(torch.uint8).view(-1), None, n, L, heads, H)
    q, k, v = [t.view(n, L, heads, d).transpose(1, 2) for t in qkv.double().split(H, dim=1)]
    bias = torch.where(mask, 0.0, torch.finfo(torch.float32).min).double()[:, None, None, :]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5 + bias, -1) @ v).transpose(1, 2).reshape(n * L, H)
    assert _rel(ctx, ref) < 2e-6
    ref.backward(dctx.double())
    dqkv = torch.empty_like(qd)
    ops.attention_bwd(dctx, qd[:, :H], qd[:, H:2 * H], qd[:, 2 * H:], probs, None, n, L, heads, H, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:])
    assert _rel(dqkv, qkv.grad) < 5e-6


def test_gather_forward_backward_match_torch():
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(3)
    v0, e_in, nfb = 50, 96, 3
    for dtype in (torch.float32, torch.float16):
        src = torch.randn(v0 + 5, e_in, device=DEV, generator=g).to(dtype)
        fb = torch.randn(nfb, e_in, device=DEV, generator=g).requires_grad_(True)
        sw = (1 + torch.rand(e_in, device=DEV, generator=g)).requires_grad_(True)
        sb = torch.randn(e_in, device=DEV, generator=g).requires_grad_(True)
        ids = torch.randint(0, v0 + nfb, (200,), device=DEV, generator=g, dtype=torch.int32)
        x = ops.gather(ids, src, v0, fb.detach(), sw.detach(), sb.detach())
        il = ids.long()
        ref = torch.where((il >= v0)[:, None], fb.double()[torch.clamp(il - v0, min=0)], sw.double() * src.double()[torch.clamp(il, max=v0 - 1)] + sb.double())
        assert _rel(x, ref) < 1e-6
        dx = torch.randn(200, e_in, device=DEV, generator=g)
        ref.backward(dx.double())
        dfb, dsw, dsb = ops.gather_bwd(ids, src, v0, dx, nfb)
        assert _rel(dfb, fb.grad) < 1e-5 and _rel(dsw, sw.grad) < 1e-5 and _rel(dsb, sb.grad) < 1e-5


def _case(flags, seed, rows=24):
    cfg, *_ = synth.workload("tiny")
    cfg = dict(cfg, **flags)
    w = synth.make_weights(cfg, seed=seed)
    src = synth.make_source_embeddings(cfg, seed)
    ids = synth.make_surface_forms(cfg, rows, seed=seed, n_special=1)
    ids[2, 1] = cfg["original_vocab_size"] + 2                       # a fallback id
    return cfg, w, src, ids


FLAGS = [dict(), dict(hn_embed_lang_id=False), dict(separate_out_embeddings=False), dict(hn_single_head=True),
         dict(hn_rescale_embeddings=False), dict(hn_predict_bias=False), dict(hn_single_head=True, separate_out_embeddings=False, hn_embed_lang_id=False)]


@pytest.mark.parametrize("packed", [True, False], ids=["packed", "dense"])
@pytest.mark.parametrize("flags", FLAGS, ids=lambda f: "+".join(f"{k[3:] if k.startswith('hn_') else k}={int(v)}" for k, v in f.items()) or "default")
def test_whole_hypernetwork_outputs_and_every_gradient(flags, packed):
    """packed: the inference path's schedule (pad skipping, input projection per distinct id, position-0-only last layer);
    dense: the reference's layout.  Both must reproduce float64 torch autograd of the as-written math."""
    from oracle import hypernet_ref
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, w, src_np, ids_np = _case(flags, seed=31)
    lang = 2 if cfg.get("hn_embed_lang_id") else None
    # the torch restatement is the oracle's math
    W64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in w.items()}
    ref = torch_port.forward(W64, cfg, torch.from_numpy(ids_np).long(), torch.from_numpy(src_np), lang)
    want = hypernet_ref.forward(w, cfg, ids_np, src_np, lang_index=lang)
    for a, b in zip(ref, want):
        assert (a is None) == (b is None) and (a is None or float(np.abs(a.detach().numpy() - b).max()) < 5e-6)
    # the drop-in class with trainable parameters takes the differentiable path
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to(DEV)
    src, ids = torch.from_numpy(src_np).to(DEV), torch.from_numpy(ids_np).to(DEV)
    model.precision = "f32"
    with torch.no_grad():
        inference = model(ids, source_embeddings=src, lang_index=None if lang is None else torch.tensor(lang))
    model.requires_grad_(True).eval()
    eval_out = model(ids, source_embeddings=src, lang_index=None if lang is None else torch.tensor(lang))
    assert not eval_out[0].requires_grad                   # eval mode (what from_pretrained returns): the inference path, no graph
    model.train()
    model.train_packed = packed
    out = model(ids, source_embeddings=src, lang_index=None if lang is None else torch.tensor(lang))
    keep = ~util.all_pad_rows(cfg, ids_np)
    for got, inf, r, what in zip(out, inference, want, ("pred_in", "pred_out", "bias")):
        if r is None:
            assert got is None
            continue
        assert got.requires_grad
        util.assert_f32_close(got.detach().cpu().numpy()[keep], r[keep], f"training forward {what}")
        util.assert_f32_close(got.detach().cpu().numpy()[keep], inf.cpu().numpy()[keep], f"training forward vs inference forward {what}")
    # one scalar loss over all three outputs, random cotangents
    gen = torch.Generator().manual_seed(5)
    cot = [None if r is None else torch.randn(r.shape, generator=gen, dtype=torch.float64) for r in ref]
    loss_ref = sum((r * c).sum() for r, c in zip(ref, cot) if r is not None)
    loss_ref.backward()
    loss = sum((o.double() * c.to(DEV)).sum() for o, c in zip(out, cot) if o is not None)
    loss.backward()
    params = dict(model.named_parameters())
    worst = {}
    for name, p64 in W64.items():
        if name not in params:
            continue                                           # model.embeddings.word_embeddings.weight etc.: never read
        g_ref = torch.zeros_like(p64) if p64.grad is None else p64.grad
        g = params[name].grad
        if g is None and p64.grad is None:
            continue                                           # model.embeddings.word_embeddings.weight: a parameter of the checkpoint contract that nothing reads
        assert g is not None, name
        denom = float(g_ref.norm())
        err = float((g.double().cpu() - g_ref).norm())
        if denom < 1e-12:
            assert err < 1e-6, (name, err)
        else:
            worst[name] = err / denom
    bad = {k: v for k, v in worst.items() if v > 2e-4}
    assert not bad, bad
    assert len(worst) >= 40


@pytest.mark.parametrize("precision,lim", [("bf16", 6e-2), ("f16", 1e-2)])
def test_sixteen_bit_contractions_in_training(precision, lim):
    """model.train_precision = "bf16" / "f16": every forward / dgrad / wgrad contraction on 16-bit MFMA operands (the tile kernels
    of the inference path), fp32 accumulation, everything else fp32.  Outputs meet the tolerance of that arithmetic against
    the oracle; every parameter's gradient stays within the operand rounding of float64 torch autograd."""
    from oracle import hypernet_ref
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, w, src_np, ids_np = _case({}, seed=33, rows=300)
    W64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in w.items()}
    ref = torch_port.forward(W64, cfg, torch.from_numpy(ids_np).long(), torch.from_numpy(src_np), 2)
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to(DEV).requires_grad_(True).train()
    model.train_precision = precision
    src, ids = torch.from_numpy(src_np).to(DEV), torch.from_numpy(ids_np).to(DEV)
    out = model(ids, source_embeddings=src, lang_index=torch.tensor(2))
    keep = ~util.all_pad_rows(cfg, ids_np)
    for got, r, what in zip(out, ref, ("pred_in", "pred_out", "bias")):
        util.CLOSE[precision](got.detach().cpu().numpy()[keep], r.detach().numpy()[keep].astype(np.float32), f"{precision} training forward {what}")
    gen = torch.Generator().manual_seed(5)
    cot = [torch.randn(r.shape, generator=gen, dtype=torch.float64) for r in ref]
    sum((r * c).sum() for r, c in zip(ref, cot)).backward()
    sum((o.double() * c.to(DEV)).sum() for o, c in zip(out, cot)).backward()
    params = dict(model.named_parameters())
    worst = {}
    for name, p64 in W64.items():
        if name not in params or params[name].grad is None or p64.grad is None:
            continue
        denom = float(p64.grad.norm())
        if denom > 1e-12:
            worst[name] = float((params[name].grad.double().cpu() - p64.grad).norm()) / denom
    bad = {k: v for k, v in worst.items() if v > lim}
    assert not bad and len(worst) >= 40, bad


def test_a_training_step_lowers_the_loss():
    """What train.py does with the path: predict embeddings, take a loss on them, step the hypernetwork's parameters."""
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, w, src_np, ids_np = _case({}, seed=41, rows=64)
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to(DEV).requires_grad_(True).train()
    src, ids = torch.from_numpy(src_np).to(DEV), torch.from_numpy(ids_np).to(DEV)
    target = torch.randn(64, cfg["n_embd"], device=DEV, generator=torch.Generator(device=DEV).manual_seed(0)) * 0.05
    losses = []
    for _ in range(5):
        model.zero_grad()
        pred_in, pred_out, bias = model(ids, source_embeddings=src, lang_index=torch.tensor(1))
        loss = ((pred_in - target) ** 2).mean() + ((pred_out - target) ** 2).mean() + (bias ** 2).mean()
        loss.backward()
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads))
        with torch.no_grad():                                   # a normalised gradient step of length 0.004 in parameter space
            for p in model.parameters():
                if p.grad is not None:
                    p -= (0.004 / float(norm)) * p.grad
        model.refresh_weights()
        losses.append(float(loss.detach()))
    assert losses[1] < losses[0] and losses[-1] < 0.9 * losses[0] and all(np.isfinite(losses)), losses
    # no gradient wanted: the inference path (pad skipping, hoisting, ...) runs as before
    with torch.no_grad():
        out = model(ids, source_embeddings=src, lang_index=torch.tensor(1))
    assert not out[0].requires_grad


def test_data_parallel_training_step_two_ranks():
    """train.py's data-parallel step: two ranks, each with its row shard, torch DistributedDataParallel around the drop-in class
    (its gradient hooks see the gradients zett_amd/autograd.py returns): the averaged gradients equal the single-process
    gradients of all rows / 2.  On a 1-GPU box both ranks share cuda:0 over gloo; with two GPUs it runs over nccl (RCCL)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = torch.cuda.device_count() < 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if one:
        env["ZETT_ONE_DEVICE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(repo, "tests", "ddp_worker.py")]
    out = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[:3000] + " ... " + out.stderr[-1500:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["world"] == 2 and d["compared"] >= 40 and d["worst_rel"] < 1e-4, d


def test_f16_training_overflow_is_an_error_not_a_nan():
    """f16 operands without loss scaling: cotangents of 1e9 overflow the half range in the gradient operands; the step raises
    (zett_amd._lib.RangeError) instead of returning NaN gradients, and the same step in bf16 is finite."""
    from zett_amd import _lib
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, w, src_np, ids_np = _case({}, seed=51, rows=64)
    src, ids = torch.from_numpy(src_np).to(DEV), torch.from_numpy(ids_np).to(DEV)
    for precision in ("f16", "bf16"):
        model = ZettHypernet(ZettHypernetConfig(**cfg))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        model = model.to(DEV).requires_grad_(True).train()
        model.train_precision = precision
        out = model(ids, source_embeddings=src, lang_index=torch.tensor(1))
        loss = sum((o * 1e9).sum() for o in out)
        if precision == "f16":
            with pytest.raises(_lib.RangeError, match="bf16"):
                loss.backward()
        else:
            loss.backward()
            assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)


def test_inference_after_an_optimizer_step_uses_the_new_weights():
    """train.py's eval_step pattern: train a step, then predict under no_grad / eval().  The inference engine holds a copy of
    the weights; it must notice the in-place update (parameter version counters) without refresh_weights()."""
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, w, src_np, ids_np = _case({}, seed=43, rows=48)
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to(DEV).requires_grad_(True).train()
    model.precision = "f32"
    src, ids = torch.from_numpy(src_np).to(DEV), torch.from_numpy(ids_np).to(DEV)
    lang = torch.tensor(1)
    with torch.no_grad():
        before = model(ids, source_embeddings=src, lang_index=lang)
        again = model(ids, source_embeddings=src, lang_index=lang)
    assert torch.equal(before[0], again[0])                   # (unchanged weights: the cached engine, same bits)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    out = model(ids, source_embeddings=src, lang_index=lang)
    (out[0] ** 2).mean().backward()
    opt.step()
    with torch.no_grad():
        after = model(ids, source_embeddings=src, lang_index=lang)
    fresh = ZettHypernet(ZettHypernetConfig(**cfg))
    fresh.load_state_dict(model.state_dict())
    fresh = fresh.to(DEV)
    fresh.precision = "f32"
    with torch.no_grad():
        want = fresh(ids, source_embeddings=src, lang_index=lang)
    assert not torch.equal(before[0], after[0])
    for a, b in zip(after, want):
        assert torch.equal(a, b)
    # p.data = ... (a new storage, version counter untouched) is seen too
    with torch.no_grad():
        p = model.get_parameter("bias_projection.bias") if cfg.get("hn_predict_bias", True) else None
    if p is not None:
        p.data = p.data + 1.0
        with torch.no_grad():
            shifted = model(ids, source_embeddings=src, lang_index=lang)
        assert float((shifted[2] - after[2]).mean()) == pytest.approx(1.0, abs=1e-4)
        # a REPLACED Parameter object (module.bias = nn.Parameter(t): parametrize / PEFT-style swaps) is seen as well
        model.bias_projection.bias = torch.nn.Parameter(p.detach().clone() + 2.0)
        with torch.no_grad():
            swapped = model(ids, source_embeddings=src, lang_index=lang)
        assert float((swapped[2] - shifted[2]).mean()) == pytest.approx(2.0, abs=1e-4)


def test_training_forward_validates_its_indices_like_the_inference_path():
    """A bad id / language index / over-long surface form raises IndexError before any kernel reads (or, in the backward,
    atomically writes) through it — the reference's F.embedding IndexError, the inference path's ZETT_E_INDEX."""
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, w, src_np, ids_np = _case({}, seed=44, rows=16)
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to(DEV).requires_grad_(True).train()
    src, ids = torch.from_numpy(src_np).to(DEV), torch.from_numpy(ids_np).to(DEV)
    top = model.dims.original_vocab_size + model.dims.n_extra
    for packed in (True, False):
        model.train_packed = packed
        bad = ids.clone(); bad[3, 1] = top
        with pytest.raises(IndexError, match="outside"):
            model(bad, source_embeddings=src, lang_index=torch.tensor(1))
        bad = ids.clone(); bad[0, 0] = -1
        with pytest.raises(IndexError):
            model(bad, source_embeddings=src, lang_index=torch.tensor(1))
        with pytest.raises(IndexError, match="lang_index"):
            model(ids, source_embeddings=src, lang_index=torch.tensor(model.dims.n_langs))
        with pytest.raises(IndexError, match="rows"):
            model(ids, source_embeddings=src[: model.dims.original_vocab_size - 1], lang_index=torch.tensor(1))
        long = torch.full((2, model.dims.max_positions), 5, dtype=ids.dtype, device=DEV)
        with pytest.raises(IndexError, match="position_embeddings"):
            model(long, source_embeddings=src, lang_index=torch.tensor(1))
    ok = ids.clone(); ok[3, 1] = top - 1                          # the last fallback row is a valid id
    out = model(ok, source_embeddings=src, lang_index=torch.tensor(1))
    assert bool(torch.isfinite(out[0]).all())
