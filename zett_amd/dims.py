"""Shape bookkeeping for the ZeTT hypernetwork.

Derives every tensor shape of the embedding-prediction path from a
``ZettHypernetConfig``-like object (attribute access) or a plain dict, following
the reference constructor (hf_hypernet/modeling_hypernet.py:46-154) and the
checkpoint contract written by scripts/convert_to_pt.py:35-49.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass

ROBERTA_MAX_POSITIONS = 514      # roberta-base max_position_embeddings (modeling_hypernet.py:67-69)
ROBERTA_LN_EPS = 1e-5            # roberta-base layer_norm_eps
PROJECTOR_LN_EPS = 1e-6          # modeling_hypernet.py:34


def cfg_get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


@dataclass(frozen=True)
class HypernetDims:
    n_embd: int            # E
    n_in_embd: int         # E_in (2E when the LM has untied output embeddings)
    n_out_embd: int        # width of the scaler / first head slice
    hidden: int            # H
    intermediate: int      # I
    heads: int
    head_dim: int
    layers: int
    n_extra: int           # rows of fallback_embeddings = max(hn_n_extra_tokens, 1)
    original_vocab_size: int
    pad_token_id: int
    separate_out: bool
    single_head: bool
    rescale: bool
    predict_bias: bool
    embed_lang: bool
    n_langs: int
    max_positions: int
    word_emb_rows: int     # unused RoBERTa word-embedding table (pad_id + 1 rows)

    @property
    def head_out_width(self) -> int:
        return self.n_in_embd if self.single_head else self.n_out_embd

    @property
    def two_heads(self) -> bool:
        return self.separate_out and not self.single_head

    @staticmethod
    def from_config(cfg, max_positions: int = ROBERTA_MAX_POSITIONS) -> "HypernetDims":
        e = int(cfg_get(cfg, "n_embd"))
        separate = bool(cfg_get(cfg, "separate_out_embeddings", False))
        h = int(cfg_get(cfg, "hn_hidden_size"))
        heads = cfg_get(cfg, "hn_num_attention_heads", None)
        if heads is None:
            heads = h // 64                       # modeling_hypernet.py:73-74
        heads = int(heads)
        pad = cfg_get(cfg, "pad_token_id", None)
        assert pad is not None                    # modeling_hypernet.py:92
        return HypernetDims(
            n_embd=e,
            n_in_embd=2 * e if separate else e,
            n_out_embd=e,
            hidden=h,
            intermediate=int(cfg_get(cfg, "hn_intermediate_size")),
            heads=heads,
            head_dim=h // heads,
            layers=int(cfg_get(cfg, "hn_n_layers", 3)),
            n_extra=max(int(cfg_get(cfg, "hn_n_extra_tokens", 0) or 0), 1),
            original_vocab_size=int(cfg_get(cfg, "original_vocab_size")),
            pad_token_id=int(pad),
            separate_out=separate,
            single_head=bool(cfg_get(cfg, "hn_single_head", False)),
            rescale=bool(cfg_get(cfg, "hn_rescale_embeddings", False)),
            predict_bias=bool(cfg_get(cfg, "hn_predict_bias", False)),
            embed_lang=bool(cfg_get(cfg, "hn_embed_lang_id", False)),
            n_langs=int(cfg_get(cfg, "n_langs", 0) or 0),
            max_positions=int(max_positions),
            word_emb_rows=int(pad) + 1,
        )


def _projector(prefix: str, h: int, i: int, out: OrderedDict) -> None:
    out[prefix + "dense1.weight"] = (i, h)
    out[prefix + "dense1.bias"] = (i,)
    out[prefix + "dense2.weight"] = (h, i)
    out[prefix + "dense2.bias"] = (h,)
    out[prefix + "ln.weight"] = (h,)
    out[prefix + "ln.bias"] = (h,)


def weight_shapes(cfg) -> "OrderedDict[str, tuple]":
    """name -> shape for every tensor of the PyTorch checkpoint layout."""
    d = cfg if isinstance(cfg, HypernetDims) else HypernetDims.from_config(cfg)
    h, i = d.hidden, d.intermediate
    s: "OrderedDict[str, tuple]" = OrderedDict()
    if d.embed_lang:
        s["lang_embeddings.weight"] = (d.n_langs, h)
    e = "model.embeddings."
    s[e + "word_embeddings.weight"] = (d.word_emb_rows, h)      # never read by the forward
    s[e + "token_type_embeddings.weight"] = (1, h)
    s[e + "LayerNorm.weight"] = (h,)
    s[e + "LayerNorm.bias"] = (h,)
    s[e + "position_embeddings.weight"] = (d.max_positions, h)
    for layer in range(d.layers):
        p = f"model.encoder.layer.{layer}."
        for proj in ("query", "key", "value"):
            s[p + f"attention.self.{proj}.weight"] = (h, h)
            s[p + f"attention.self.{proj}.bias"] = (h,)
        s[p + "attention.output.dense.weight"] = (h, h)
        s[p + "attention.output.dense.bias"] = (h,)
        s[p + "attention.output.LayerNorm.weight"] = (h,)
        s[p + "attention.output.LayerNorm.bias"] = (h,)
        s[p + "intermediate.dense.weight"] = (i, h)
        s[p + "intermediate.dense.bias"] = (i,)
        s[p + "output.dense.weight"] = (h, i)
        s[p + "output.dense.bias"] = (h,)
        s[p + "output.LayerNorm.weight"] = (h,)
        s[p + "output.LayerNorm.bias"] = (h,)
    s["fallback_embeddings.weight"] = (d.n_extra, d.n_in_embd)
    s["input_projection.0.weight"] = (h, d.n_in_embd)
    s["input_projection.0.bias"] = (h,)
    _projector("input_projection.1.", h, i, s)
    _projector("output_projection.0.", h, i, s)
    s["output_projection.1.weight"] = (d.head_out_width, h)
    s["output_projection.1.bias"] = (d.head_out_width,)
    if d.two_heads:
        _projector("output_projection_out.0.", h, i, s)
        s["output_projection_out.1.weight"] = (d.n_embd, h)
        s["output_projection_out.1.bias"] = (d.n_embd,)
    if d.rescale:
        s["in_scaler.w"] = (1, d.n_in_embd)
        s["in_scaler.b"] = (1, d.n_in_embd)
        s["scaler.w"] = (1, d.n_out_embd)
        s["scaler.b"] = (1, d.n_out_embd)
        if d.separate_out:
            s["out_scaler.w"] = (1, d.n_embd)
            s["out_scaler.b"] = (1, d.n_embd)
    if d.predict_bias:
        s["bias_projection.weight"] = (1, h)
        s["bias_projection.bias"] = (1,)
    return s


def as_written_flops_per_row(dims: "HypernetDims", seq: int) -> int:
    """Algorithmic FLOPs the reference spends per target row (SURVEY.md §8d F_ref): every (row, position) through the
    input projection and all encoder layers, pads included — what the throughput-derived "as written" TFLOP/s of
    bench.py is quoted in.  (oracle/hypernet_ref.py:flops_per_row is the same formula on the oracle's side;
    tests/test_oracle_golden.py pins both to the survey's three figures.)"""
    e, e_in, h, i = dims.n_embd, dims.n_in_embd, dims.hidden, dims.intermediate
    lp = seq + (1 if dims.embed_lang else 0)
    heads_out = 1 if (dims.single_head or not dims.separate_out) else 2
    e_out = e_in if dims.single_head else e
    return (seq * (2 * e_in * h + 4 * h * i)
            + dims.layers * (lp * (8 * h * h + 4 * h * i) + 4 * lp * lp * h)
            + heads_out * (4 * h * i + 2 * h * e_out) + 2 * h)
