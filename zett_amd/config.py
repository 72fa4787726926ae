"""``ZettHypernetConfig`` — configuration of a ZeTT hypernetwork checkpoint.

Keeps the field names (and defaults) of the reference config class so that a
``config.json`` written by the reference loads unchanged
(reference: hf_hypernet/configuration_hypernet.py:3-56).  The fields the
reference's training script injects without declaring them — ``pad_token_id``,
``original_vocab_size``, ``separate_out_embeddings``, ``hn_n_extra_tokens``,
``langs``, ``vocab_size`` (train.py:295,298-299,314,350,361; convert_to_pt.py:33)
— pass through ``**kwargs`` exactly as they do there.
"""
from __future__ import annotations

from transformers import PretrainedConfig

MODEL_TYPE = "zett_hypernetwork"

# (field, default) in the reference's declaration order
_FIELDS = (
    ("hn_model_name_or_path", "roberta-base"),
    ("hn_surface_maxlen", 16),
    ("hn_n_layers", 3),
    ("n_embd", 768),
    ("hn_hidden_size", None),
    ("hn_intermediate_size", None),
    ("hn_rescale_embeddings", False),
    ("use_unigram_bias", False),
    ("hn_embed_target_priors", False),
    ("hn_add_inter_token_attention", False),
    ("hn_inter_token_attention_bias_by_priors", False),
    ("hn_inter_token_attention_bias_scaler", 1.0),
    ("hn_n_inter_token_blocks", 16),
    ("hn_language_adapter_bottleneck_dim", 0),
    ("hn_embed_using_source_embeddings", False),
    ("hn_concat_last_hidden_state", False),
    ("hn_single_head", False),
    ("hn_predict_bias", True),
    ("hn_num_attention_heads", None),
    ("hn_embed_lang_id", False),
    ("hn_model_type", "roberta"),
    ("n_langs", None),
)


class ZettHypernetConfig(PretrainedConfig):
    model_type = MODEL_TYPE

    def __init__(self, **kwargs):
        own = {name: kwargs.pop(name, default) for name, default in _FIELDS}
        super().__init__(**kwargs)
        self.model_type = MODEL_TYPE
        for name, value in own.items():
            setattr(self, name, value)


def hypernet_fields():
    """Names of the declared hypernet fields (used by tests and the transfer CLI)."""
    return tuple(name for name, _ in _FIELDS)
