"""A torch restatement of oracle/hypernet_ref.forward (the as-written reference math, modeling_hypernet.py:156-267) that
torch.autograd can differentiate: the reference for the gradients of the training path (zett_amd/autograd.py).
TEST INFRASTRUCTURE, like oracle/: never imported by zett_amd/.  Checked against the numpy oracle in
tests/test_autograd_gpu.py before it is trusted for gradients."""
import math

import torch


def _gelu_tanh(x):
    return torch.nn.functional.gelu(x, approximate="tanh")


def _ln(x, w, b, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps)


def _pb(x, W, p):
    h = _gelu_tanh(x @ W[p + "dense1.weight"].T + W[p + "dense1.bias"])
    h = _gelu_tanh(h @ W[p + "dense2.weight"].T + W[p + "dense2.bias"])
    return _ln(h + x, W[p + "ln.weight"], W[p + "ln.bias"], 1e-6)


def forward(W, cfg, ids, src, lang=None, ln_eps=1e-5):
    """W: dict of torch tensors (any float dtype, requires_grad as wanted); ids int64 [N, L]; src [V, E_in]."""
    dt = next(iter(W.values())).dtype
    n, L = ids.shape
    v0, e = cfg["original_vocab_size"], cfg["n_embd"]
    fb = ids >= v0
    x = src.to(dt)[torch.clamp(ids, max=v0 - 1)]
    if cfg.get("hn_rescale_embeddings"):
        x = W["in_scaler.w"].reshape(-1) * x + W["in_scaler.b"].reshape(-1)
    x = torch.where(fb[..., None], W["fallback_embeddings.weight"][torch.clamp(ids - v0, min=0)], x)
    x = x @ W["input_projection.0.weight"].T + W["input_projection.0.bias"]
    x = _pb(x, W, "input_projection.1.")
    mask = ids != cfg["pad_token_id"]
    type0 = W["model.embeddings.token_type_embeddings.weight"][0]
    pos = W["model.embeddings.position_embeddings.weight"]
    if cfg.get("hn_embed_lang_id"):
        lv = W["lang_embeddings.weight"][int(lang)] - (type0 + pos[L])
        x = torch.cat([x, lv[None, None, :].expand(n, 1, -1)], 1)
        mask = torch.cat([mask, torch.ones((n, 1), dtype=torch.bool, device=mask.device)], 1)
    lp, h = x.shape[1], x.shape[2]
    heads = cfg.get("hn_num_attention_heads") or h // 64
    d = h // heads
    z = _ln(x + type0 + pos[:lp], W["model.embeddings.LayerNorm.weight"], W["model.embeddings.LayerNorm.bias"], ln_eps)
    bias = torch.where(mask, 0.0, torch.finfo(torch.float32).min).to(dt)[:, None, None, :]
    for l in range(cfg.get("hn_n_layers", 3)):
        p = f"model.encoder.layer.{l}."
        a = p + "attention.self."
        q = (z @ W[a + "query.weight"].T + W[a + "query.bias"]).view(n, lp, heads, d).transpose(1, 2)
        k = (z @ W[a + "key.weight"].T + W[a + "key.bias"]).view(n, lp, heads, d).transpose(1, 2)
        v = (z @ W[a + "value.weight"].T + W[a + "value.bias"]).view(n, lp, heads, d).transpose(1, 2)
        s = q @ k.transpose(-1, -2) * (d ** -0.5) + bias
        ctx = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(n, lp, h)
        o = p + "attention.output."
        z = _ln(ctx @ W[o + "dense.weight"].T + W[o + "dense.bias"] + z, W[o + "LayerNorm.weight"], W[o + "LayerNorm.bias"], ln_eps)
        u = torch.nn.functional.gelu(z @ W[p + "intermediate.dense.weight"].T + W[p + "intermediate.dense.bias"])
        o = p + "output."
        z = _ln(u @ W[o + "dense.weight"].T + W[o + "dense.bias"] + z, W[o + "LayerNorm.weight"], W[o + "LayerNorm.bias"], ln_eps)
    cls = z[:, 0]
    pred = _pb(cls, W, "output_projection.0.") @ W["output_projection.1.weight"].T + W["output_projection.1.bias"]
    separate = bool(cfg.get("separate_out_embeddings"))
    if cfg.get("hn_single_head"):
        pred_in, pred_out = pred[:, :e], (pred[:, e:] if separate else None)
    else:
        pred_in, pred_out = pred, None
        if separate:
            pred_out = _pb(cls, W, "output_projection_out.0.") @ W["output_projection_out.1.weight"].T + W["output_projection_out.1.bias"]
    if cfg.get("hn_rescale_embeddings"):
        pred_in = W["scaler.w"].reshape(-1) * pred_in + W["scaler.b"].reshape(-1)
        if pred_out is not None:
            pred_out = W["out_scaler.w"].reshape(-1) * pred_out + W["out_scaler.b"].reshape(-1)
    if cfg.get("hn_predict_bias"):
        b = cls @ W["bias_projection.weight"].reshape(-1) + W["bias_projection.bias"].reshape(-1)
    else:
        b = torch.zeros(n, dtype=dt, device=cls.device)
    return pred_in, pred_out, b
