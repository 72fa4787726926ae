"""numpy fp32 oracle of the ZeTT hypernetwork forward (as-written math).

TEST INFRASTRUCTURE — see oracle/__init__.py.  Never imported by zett_amd/.

Every padded position is computed, every (row, position) runs the input
projection, every position runs the last encoder layer: this is the
reference's arithmetic, not the optimised schedule the HIP path uses, so that a
disagreement between the two points at the HIP path.

Weights are a dict of numpy arrays under the reference checkpoint names
(reference scripts/convert_to_pt.py:35-49).  All Linear weights
are torch-layout ``[out, in]``.
"""
from __future__ import annotations

import math

import numpy as np

try:  # scipy is present on both boxes; keep a slow fallback anyway
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float32])

F32 = np.float32
ROBERTA_LN_EPS = 1e-5      # roberta-base layer_norm_eps, modeling_hypernet.py:67-69
PROJECTOR_LN_EPS = 1e-6    # modeling_hypernet.py:34


def _cfg(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


_MATMUL_BACKEND = "numpy"


def set_matmul_backend(name: str) -> None:
    """"numpy" (default; what the parity tests use) or "torch": the same fp32 GEMMs through
    torch's CPU BLAS, which threads far better on many-core hosts — used only by the timed
    cpu_baseline leg of bench.py."""
    global _MATMUL_BACKEND
    assert name in ("numpy", "torch")
    _MATMUL_BACKEND = name


_OPERAND_ROUNDING = None


def set_operand_rounding(kind) -> None:
    """None (default: the fp32 reference math) or "bf16" / "f16": round BOTH operands of every Linear to that type before
    an fp32-accumulated product — a CPU emulation of what 16-bit MFMA operands must cost, whatever the kernel.
    "bf16" is (an upper bound of the accuracy of) the reference CLI's own default arithmetic (scripts/transfer.py:41,
    145-151: bfloat16 parameters and compute).  Used by tests that bound the HIP path's 16-bit modes by the error the
    arithmetic itself implies instead of by a bare number."""
    global _OPERAND_ROUNDING
    assert kind in (None, "bf16", "f16")
    _OPERAND_ROUNDING = kind


def _round_operand(a):
    if _OPERAND_ROUNDING is None:
        return a
    a = np.ascontiguousarray(a, dtype=F32)
    if _OPERAND_ROUNDING == "f16":
        return a.astype(np.float16).astype(F32)
    u = a.view(np.uint32).astype(np.uint64)                  # bf16, round to nearest even
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(F32)


def linear(x, w, b):
    """torch.nn.Linear: x @ w.T + b (fp32)."""
    x, w = _round_operand(x), _round_operand(w)
    if _MATMUL_BACKEND == "torch":
        import torch
        x2 = np.ascontiguousarray(x, dtype=F32).reshape(-1, x.shape[-1])
        y = torch.addmm(torch.from_numpy(np.ascontiguousarray(b, dtype=F32)), torch.from_numpy(x2),
                        torch.from_numpy(np.ascontiguousarray(w, dtype=F32)).t()).numpy()
        return y.reshape(x.shape[:-1] + (w.shape[0],))
    return (x @ w.T + b).astype(F32, copy=False)


def gelu_tanh(x):
    """F.gelu(approximate="tanh") — modeling_hypernet.py:36-39."""
    x = x.astype(F32, copy=False)
    c = F32(math.sqrt(2.0 / math.pi))
    inner = c * (x + F32(0.044715) * x * x * x)
    return (F32(0.5) * x * (F32(1.0) + np.tanh(inner))).astype(F32, copy=False)


def gelu_erf(x):
    """hidden_act="gelu" of RobertaIntermediate (exact erf form)."""
    x = x.astype(F32, copy=False)
    return (x * F32(0.5) * (F32(1.0) + _erf(x / F32(math.sqrt(2.0))).astype(F32))).astype(F32, copy=False)


def layer_norm(x, w, b, eps):
    """torch.nn.LayerNorm over the last axis (biased variance)."""
    x = x.astype(F32, copy=False)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps)) * w + b).astype(F32, copy=False)


def projector_block(x, W, prefix):
    """ProjectorBlock.__call__ — modeling_hypernet.py:22-40."""
    h = gelu_tanh(linear(x, W[prefix + "dense1.weight"], W[prefix + "dense1.bias"]))
    h = gelu_tanh(linear(h, W[prefix + "dense2.weight"], W[prefix + "dense2.bias"]))
    return layer_norm(h + x, W[prefix + "ln.weight"], W[prefix + "ln.bias"], PROJECTOR_LN_EPS)


def embed_inputs(W, cfg, ids, source_embeddings):
    """A2 + A3: id split, gather, in_scaler, fallback select.

    modeling_hypernet.py:170-188.
    """
    v0 = int(_cfg(cfg, "original_vocab_size"))
    use_fallback = ids >= v0
    main_ids = np.minimum(ids, v0 - 1)
    fallback_ids = np.maximum(ids - v0, 0)
    # F.embedding on an fp16/bf16 table is upcast by the next fp32 op (SURVEY §3.2 :179)
    src = np.asarray(source_embeddings)[main_ids].astype(F32)
    if _cfg(cfg, "hn_rescale_embeddings", False):
        src = W["in_scaler.w"].reshape(-1) * src + W["in_scaler.b"].reshape(-1)
    fb = W["fallback_embeddings.weight"][fallback_ids]
    return np.where(use_fallback[..., None], fb, src).astype(F32)


def roberta_encoder(W, cfg, x, key_mask):
    """RobertaModel(inputs_embeds, attention_mask, position_ids=arange) in eval mode.

    Mirrors the installed transformers ``RobertaEmbeddings.forward`` /
    ``eager_attention_forward`` / ``RobertaSelfOutput`` / ``RobertaIntermediate`` /
    ``RobertaOutput`` as called at modeling_hypernet.py:220-229.  ``key_mask`` is
    ``ids != pad`` (plus the always-visible language token).  The additive mask is
    ``finfo(float32).min`` on masked keys, so a row whose keys are ALL masked gets
    uniform attention over every position (eager / Flax semantics).
    """
    n, lp, hdim = x.shape
    heads = int(_cfg(cfg, "hn_num_attention_heads") or hdim // 64)
    d = hdim // heads
    p = "model.embeddings."
    emb = x + W[p + "token_type_embeddings.weight"][0]
    emb = emb + W[p + "position_embeddings.weight"][np.arange(lp)]
    z = layer_norm(emb, W[p + "LayerNorm.weight"], W[p + "LayerNorm.bias"], ROBERTA_LN_EPS)

    bias = np.where(key_mask, F32(0.0), np.finfo(F32).min).astype(F32)[:, None, None, :]
    scaling = F32(d ** -0.5)
    for layer in range(int(_cfg(cfg, "hn_n_layers", 3))):
        lpfx = f"model.encoder.layer.{layer}."
        a = lpfx + "attention.self."
        q = linear(z, W[a + "query.weight"], W[a + "query.bias"]).reshape(n, lp, heads, d).transpose(0, 2, 1, 3)
        k = linear(z, W[a + "key.weight"], W[a + "key.bias"]).reshape(n, lp, heads, d).transpose(0, 2, 1, 3)
        v = linear(z, W[a + "value.weight"], W[a + "value.bias"]).reshape(n, lp, heads, d).transpose(0, 2, 1, 3)
        s = (q @ k.transpose(0, 1, 3, 2)).astype(F32) * scaling
        s = s + bias
        s = s - s.max(axis=-1, keepdims=True)
        e = np.exp(s).astype(F32)
        prob = e / e.sum(axis=-1, keepdims=True, dtype=F32)
        ctx = (prob @ v).astype(F32).transpose(0, 2, 1, 3).reshape(n, lp, hdim)
        o = lpfx + "attention.output."
        z = layer_norm(linear(ctx, W[o + "dense.weight"], W[o + "dense.bias"]) + z,
                       W[o + "LayerNorm.weight"], W[o + "LayerNorm.bias"], ROBERTA_LN_EPS)
        inter = gelu_erf(linear(z, W[lpfx + "intermediate.dense.weight"], W[lpfx + "intermediate.dense.bias"]))
        o = lpfx + "output."
        z = layer_norm(linear(inter, W[o + "dense.weight"], W[o + "dense.bias"]) + z,
                       W[o + "LayerNorm.weight"], W[o + "LayerNorm.bias"], ROBERTA_LN_EPS)
    return z


def forward(W, cfg, target_surface_forms, source_embeddings, lang_index=None):
    """ZettHypernet.__call__ — modeling_hypernet.py:156-267.

    Returns (pred_in [N,E], pred_out [N,E] | None, bias [N]) as fp32 numpy arrays.
    """
    if not _cfg(cfg, "hn_embed_using_source_embeddings", False):
        raise NotImplementedError()          # modeling_hypernet.py:167-168
    ids = np.asarray(target_surface_forms).astype(np.int64)
    n, seq = ids.shape
    e = int(_cfg(cfg, "n_embd"))
    separate = bool(_cfg(cfg, "separate_out_embeddings", False))
    pad = int(_cfg(cfg, "pad_token_id"))

    x = embed_inputs(W, cfg, ids, source_embeddings)                      # :170-188
    x = linear(x, W["input_projection.0.weight"], W["input_projection.0.bias"])
    x = projector_block(x, W, "input_projection.1.")                      # :189
    mask = ids != pad                                                     # :190

    if _cfg(cfg, "hn_embed_lang_id", False):                              # :192-218
        lang = W["lang_embeddings.weight"][int(lang_index)].astype(F32).copy()
        lang = lang - (W["model.embeddings.token_type_embeddings.weight"][0]
                       + W["model.embeddings.position_embeddings.weight"][seq])
        x = np.concatenate([x, np.broadcast_to(lang[None, None, :], (n, 1, x.shape[-1]))], axis=1)
        mask = np.concatenate([mask, np.ones((n, 1), dtype=bool)], axis=1)

    hidden = roberta_encoder(W, cfg, x.astype(F32), mask)                 # :220-229
    cls = hidden[:, 0]                                                    # :234

    pred = linear(projector_block(cls, W, "output_projection.0."),
                  W["output_projection.1.weight"], W["output_projection.1.bias"])
    if _cfg(cfg, "hn_single_head", False):                                # :238-246
        pred_in = pred[..., :e]
        pred_out = pred[..., e:] if separate else None
    else:
        pred_in = pred
        pred_out = None
        if separate:                                                      # :248-252
            pred_out = linear(projector_block(cls, W, "output_projection_out.0."),
                              W["output_projection_out.1.weight"], W["output_projection_out.1.bias"])
    if _cfg(cfg, "hn_rescale_embeddings", False):                         # :254-258
        pred_in = W["scaler.w"].reshape(-1) * pred_in + W["scaler.b"].reshape(-1)
        if pred_out is not None:
            pred_out = W["out_scaler.w"].reshape(-1) * pred_out + W["out_scaler.b"].reshape(-1)
    if _cfg(cfg, "hn_predict_bias", False):                               # :260-265
        bias = (cls @ W["bias_projection.weight"][0] + W["bias_projection.bias"][0]).astype(F32)
    else:
        bias = np.zeros((n,), dtype=F32)
    return (pred_in.astype(F32), None if pred_out is None else pred_out.astype(F32), bias)


def _heads(W, cfg, cls):
    """Output heads, rescalers and bias head on the CLS states (modeling_hypernet.py:231-265)."""
    e = int(_cfg(cfg, "n_embd"))
    separate = bool(_cfg(cfg, "separate_out_embeddings", False))
    pred = linear(projector_block(cls, W, "output_projection.0."),
                  W["output_projection.1.weight"], W["output_projection.1.bias"])
    if _cfg(cfg, "hn_single_head", False):
        pred_in = pred[..., :e]
        pred_out = pred[..., e:] if separate else None
    else:
        pred_in = pred
        pred_out = None
        if separate:
            pred_out = linear(projector_block(cls, W, "output_projection_out.0."),
                              W["output_projection_out.1.weight"], W["output_projection_out.1.bias"])
    if _cfg(cfg, "hn_rescale_embeddings", False):
        pred_in = W["scaler.w"].reshape(-1) * pred_in + W["scaler.b"].reshape(-1)
        if pred_out is not None:
            pred_out = W["out_scaler.w"].reshape(-1) * pred_out + W["out_scaler.b"].reshape(-1)
    if _cfg(cfg, "hn_predict_bias", False):
        bias = (cls @ W["bias_projection.weight"][0] + W["bias_projection.bias"][0]).astype(F32)
    else:
        bias = np.zeros((cls.shape[0],), dtype=F32)
    return (pred_in.astype(F32), None if pred_out is None else pred_out.astype(F32), bias)


def forward_levers(W, cfg, target_surface_forms, source_embeddings, lang_index=None):
    """The same function as forward(), computed with the four exact levers the HIP path uses
    (DESIGN.md §2): (L1) positions that are pad, are not position 0 and are not the language token
    never influence the CLS state of a row with at least one visible key, so they are dropped;
    (L2) the input projection depends on the source id only, so it is evaluated once per DISTINCT
    id; (L3) only position 0 of the last layer is read, so its attention output, FFN and
    LayerNorms are evaluated for position 0 only; (L4) the embeddings' output — hence layer 0's
    query / key / value — depends on the (source id, position) pair only, so with two or more layers they are
    evaluated once per DISTINCT pair.  Rows whose keys are all masked keep every
    position (uniform attention over all of them, the eager / Flax behaviour).

    Test infrastructure like the rest of this file: it is (a) a CPU proof that the levers are exact
    (tests compare it with forward()), (b) the "equally optimised CPU path" of SURVEY.md §8d that
    bench.py times beside the as-written one, so that the GPU speed-up is not credited with
    algorithmic gains.
    """
    if not _cfg(cfg, "hn_embed_using_source_embeddings", False):
        raise NotImplementedError()
    ids = np.asarray(target_surface_forms).astype(np.int64)
    n, seq = ids.shape
    pad = int(_cfg(cfg, "pad_token_id"))
    has_lang = bool(_cfg(cfg, "hn_embed_lang_id", False))
    hdim = int(_cfg(cfg, "hn_hidden_size"))
    layers = int(_cfg(cfg, "hn_n_layers", 3))
    heads = int(_cfg(cfg, "hn_num_attention_heads") or hdim // 64)
    d = hdim // heads
    scaling = F32(d ** -0.5)

    visible = ids != pad                                     # key mask of the surface positions
    all_masked = ~visible.any(axis=1) & (not has_lang)       # uniform rows: keep everything
    keep = visible.copy()
    keep[:, 0] = True                                        # the CLS query
    keep[all_masked] = True

    # L2: one input projection per distinct kept id
    uniq, inverse = np.unique(ids[keep], return_inverse=True)
    table = embed_inputs(W, cfg, uniq[None, :], source_embeddings)[0]
    table = linear(table, W["input_projection.0.weight"], W["input_projection.0.bias"])
    table = projector_block(table, W, "input_projection.1.")
    slot = np.full(ids.shape, -1, dtype=np.int64)
    slot[keep] = inverse

    p = "model.embeddings."
    tt = W[p + "token_type_embeddings.weight"][0]
    pos_emb = W[p + "position_embeddings.weight"]
    lang_vec = None
    if has_lang:
        lang_vec = W["lang_embeddings.weight"][int(lang_index)].astype(F32)   # the cancel trick nets out to lang + nothing

    # L4: layer 0's q / k / v once per distinct (table slot, position) pair (+ one row for the language token)
    qkv0 = pair_of = None
    if layers >= 2:
        pos_all = np.broadcast_to(np.arange(seq)[None, :], ids.shape)
        pkey = slot[keep] * seq + pos_all[keep]
        upair, pinv = np.unique(pkey, return_inverse=True)
        pair_of = np.full(ids.shape, -1, dtype=np.int64)
        pair_of[keep] = pinv
        xp = table[upair // seq] + tt + pos_emb[upair % seq]
        if has_lang:
            xp = np.concatenate([xp, lang_vec[None, :]], axis=0)          # last pair row = the language token
        zp = layer_norm(xp, W[p + "LayerNorm.weight"], W[p + "LayerNorm.bias"], ROBERTA_LN_EPS)
        a0 = "model.encoder.layer.0.attention.self."
        qkv0 = tuple(linear(zp, W[a0 + nm + ".weight"], W[a0 + nm + ".bias"]) for nm in ("query", "key", "value"))

    out_cls = np.zeros((n, hdim), dtype=F32)
    kept_len = keep.sum(axis=1)
    for k in np.unique(kept_len):                            # L1: dense batches of rows with k kept positions
        rows = np.nonzero(kept_len == k)[0]
        m = len(rows)
        pos = np.stack([np.nonzero(keep[r])[0] for r in rows])          # [m, k] original positions, ascending (0 first)
        x = table[slot[rows[:, None], pos]] + tt + pos_emb[pos]
        key_ok = visible[rows[:, None], pos] | all_masked[rows][:, None]
        if has_lang:                                         # extra token: embedding = lang (+type+pos of slot seq, cancelled)
            x = np.concatenate([x, np.broadcast_to(lang_vec[None, None, :], (m, 1, hdim))], axis=1)
            key_ok = np.concatenate([key_ok, np.ones((m, 1), dtype=bool)], axis=1)
        z = layer_norm(x, W[p + "LayerNorm.weight"], W[p + "LayerNorm.bias"], ROBERTA_LN_EPS)
        kk = z.shape[1]
        # a row with every key masked attends uniformly (finfo.min on all keys)
        bias = np.where(key_ok, F32(0.0), np.finfo(F32).min).astype(F32)[:, None, None, :]
        for layer in range(layers):
            last = layer == layers - 1
            lpfx = f"model.encoder.layer.{layer}."
            a = lpfx + "attention.self."
            zq = z[:, :1] if last else z                      # L3: only the CLS query in the last layer
            if layer == 0 and qkv0 is not None:               # L4: gathered from the per-pair projections
                pidx = pair_of[rows[:, None], pos]
                if has_lang:
                    pidx = np.concatenate([pidx, np.full((m, 1), len(qkv0[0]) - 1, dtype=np.int64)], axis=1)
                q, kx, v = (t[pidx].reshape(m, kk, heads, d).transpose(0, 2, 1, 3) for t in qkv0)
            else:
                q = linear(zq, W[a + "query.weight"], W[a + "query.bias"]).reshape(m, zq.shape[1], heads, d).transpose(0, 2, 1, 3)
                kx = linear(z, W[a + "key.weight"], W[a + "key.bias"]).reshape(m, kk, heads, d).transpose(0, 2, 1, 3)
                v = linear(z, W[a + "value.weight"], W[a + "value.bias"]).reshape(m, kk, heads, d).transpose(0, 2, 1, 3)
            sc = (q @ kx.transpose(0, 1, 3, 2)).astype(F32) * scaling + bias
            sc = sc - sc.max(axis=-1, keepdims=True)
            ex = np.exp(sc).astype(F32)
            prob = ex / ex.sum(axis=-1, keepdims=True, dtype=F32)
            ctx = (prob @ v).astype(F32).transpose(0, 2, 1, 3).reshape(m, zq.shape[1], hdim)
            o = lpfx + "attention.output."
            z1 = layer_norm(linear(ctx, W[o + "dense.weight"], W[o + "dense.bias"]) + zq,
                            W[o + "LayerNorm.weight"], W[o + "LayerNorm.bias"], ROBERTA_LN_EPS)
            inter = gelu_erf(linear(z1, W[lpfx + "intermediate.dense.weight"], W[lpfx + "intermediate.dense.bias"]))
            o = lpfx + "output."
            z = layer_norm(linear(inter, W[o + "dense.weight"], W[o + "dense.bias"]) + z1,
                           W[o + "LayerNorm.weight"], W[o + "LayerNorm.bias"], ROBERTA_LN_EPS)
        out_cls[rows] = z[:, 0]
    return _heads(W, cfg, out_cls)


def flops_per_row(cfg, seq):
    """As-written algorithmic FLOPs per target row (SURVEY.md §8d F_ref)."""
    e = int(_cfg(cfg, "n_embd"))
    e_in = 2 * e if _cfg(cfg, "separate_out_embeddings", False) else e
    h = int(_cfg(cfg, "hn_hidden_size"))
    i = int(_cfg(cfg, "hn_intermediate_size"))
    lp = seq + (1 if _cfg(cfg, "hn_embed_lang_id", False) else 0)
    layers = int(_cfg(cfg, "hn_n_layers", 3))
    heads_out = 1 if (_cfg(cfg, "hn_single_head", False) or not _cfg(cfg, "separate_out_embeddings", False)) else 2
    e_out = e_in if _cfg(cfg, "hn_single_head", False) else e
    return (seq * (2 * e_in * h + 4 * h * i)
            + layers * (lp * (8 * h * h + 4 * h * i) + 4 * lp * lp * h)
            + heads_out * (4 * h * i + 2 * h * e_out) + 2 * h)
