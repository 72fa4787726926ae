"""``ZettHypernet`` — the hypernetwork behind the reference's ``AutoModel`` API,
computing on MI355X through libzett_hip.so.

Drop-in for hf_hypernet/modeling_hypernet.py:43-267:

    hypernet = AutoModel.from_pretrained(path)          # after `import zett_amd`
    pred_in, pred_out, bias = hypernet(target_surface_forms,
                                       source_embeddings=source_embeddings,
                                       lang_index=lang_index)

Same config class fields, same ``state_dict`` names and shapes (so checkpoints
written by scripts/convert_to_pt.py load unchanged), same call signature, return
tuple and error behaviour.  The arithmetic runs in HIP kernels; this module only
holds the parameters as torch tensors and hands device pointers across the C ABI.
There is no CPU path: calling the model with CPU tensors raises.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, Optional, Tuple

import torch
from torch import nn
from transformers import PreTrainedModel

from . import _lib
from .config import ZettHypernetConfig
from .dims import PROJECTOR_LN_EPS, ROBERTA_LN_EPS, ROBERTA_MAX_POSITIONS, HypernetDims, weight_shapes

_TORCH_TO_ZETT = {torch.float32: _lib.DTYPE_F32, torch.float16: _lib.DTYPE_F16, torch.bfloat16: _lib.DTYPE_BF16}
DEFAULT_PRECISION = "f16"
_PRECISIONS = {"bf16": _lib.PREC_BF16, "bfloat16": _lib.PREC_BF16, "f32": _lib.PREC_F32,
               "fp32": _lib.PREC_F32, "float32": _lib.PREC_F32, "f16": _lib.PREC_F16, "fp16": _lib.PREC_F16,
               "float16": _lib.PREC_F16}


def _backbone_settings(name_or_path: str) -> Tuple[int, float]:
    """max_position_embeddings / layer_norm_eps of the RoBERTa backbone config.

    The reference reads them from ``RobertaConfig.from_pretrained(hn_model_name_or_path)``
    (modeling_hypernet.py:67-69).  A local directory is honoured; otherwise the
    roberta-base values apply (every shipped config uses roberta-base and there is no
    network to fetch anything else).
    """
    path = os.path.join(str(name_or_path), "config.json")
    if os.path.isfile(path):
        with open(path) as f:
            d = json.load(f)
        return int(d.get("max_position_embeddings", ROBERTA_MAX_POSITIONS)), float(d.get("layer_norm_eps", ROBERTA_LN_EPS))
    return ROBERTA_MAX_POSITIONS, ROBERTA_LN_EPS


class _Node(nn.Module):
    """Plain container: gives parameters the dotted names of the reference modules."""


def _attach(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    *path, leaf = dotted.split(".")
    node = root
    for part in path:
        child = node._modules.get(part)
        if child is None:
            child = _Node()
            node.add_module(part, child)
        node = child
    node.register_parameter(leaf, param)


class HipEngine:
    """One libzett_hip handle: device + arithmetic mode + uploaded weights."""

    def __init__(self, dims: HypernetDims, ln_eps_encoder: float, device: torch.device, precision: str):
        if device.type != "cuda":
            raise RuntimeError("zett_amd computes on MI355X only: tensors must live on a cuda (ROCm) device")
        self.lib = _lib.load()
        self.dims = dims
        self.device = device
        self.precision = precision
        cfg = _lib.ZettConfig(
            n_embd=dims.n_embd, n_in_embd=dims.n_in_embd, hidden=dims.hidden, intermediate=dims.intermediate,
            heads=dims.heads, layers=dims.layers, n_extra=dims.n_extra, original_vocab_size=dims.original_vocab_size,
            pad_token_id=dims.pad_token_id, separate_out=int(dims.separate_out), single_head=int(dims.single_head),
            rescale=int(dims.rescale), predict_bias=int(dims.predict_bias), embed_lang=int(dims.embed_lang),
            n_langs=dims.n_langs, max_positions=dims.max_positions, ln_eps_encoder=ln_eps_encoder,
            ln_eps_projector=PROJECTOR_LN_EPS)
        handle = C.c_void_p()
        index = device.index if device.index is not None else torch.cuda.current_device()
        _lib.check(self.lib.zett_create(C.byref(cfg), index, _PRECISIONS[precision], C.byref(handle)), "zett_create")
        self.handle = handle
        self.weights_stamp = None

    def load_weights(self, tensors: Dict[str, torch.Tensor]) -> None:
        for name, t in tensors.items():
            t = t.detach()
            if t.dtype not in _TORCH_TO_ZETT:
                t = t.float()
            t = t.contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(self.lib.zett_load_weight(self.handle, name.encode(), C.c_void_p(t.data_ptr()),
                                                 _TORCH_TO_ZETT[t.dtype], shape, t.dim()), f"zett_load_weight({name})")
        _lib.check(self.lib.zett_finalize(self.handle), "zett_finalize")

    def set_option(self, key: str, value: int) -> None:
        _lib.check(self.lib.zett_set_option(self.handle, key.encode(), int(value)), f"zett_set_option({key})")

    def workspace_bytes(self, n_rows: int, seq: int) -> int:
        """Upper bound of the device bytes a [n_rows, seq] forward reserves (zett_workspace_bytes)."""
        out = C.c_int64(0)
        _lib.check(self.lib.zett_workspace_bytes(self.handle, int(n_rows), int(seq), C.byref(out)), "zett_workspace_bytes")
        return out.value

    def range_flags(self) -> int:
        """Wait for the current stream and return the range word of the most recent forward (zett_check_range):
        0 = every 16-bit operand stayed inside its type's range and the predicted embeddings are finite; otherwise a
        combination of _lib.RANGE_SOURCE / RANGE_ACTIVATION / RANGE_OUTPUT."""
        flags = C.c_int32(0)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self.lib.zett_check_range(self.handle, C.c_void_p(stream), C.byref(flags))
        if rc not in (_lib.ZETT_OK, _lib.E_RANGE):
            _lib.check(rc, "zett_check_range")
        return int(flags.value)

    def stream_wait_output(self, which: str, stream: "torch.cuda.Stream") -> None:
        """Make `stream` wait until output `which` ("in" = pred_in, "bias") of the most recent forward is complete
        (zett_stream_wait_output): pred_in is final after the first head's last GEMM, while the second head still
        computes — the vocabulary-sharded path starts its all-gather there (zett_amd/sharding.py)."""
        code = {"in": _lib.OUT_IN, "bias": _lib.OUT_BIAS}[which]
        _lib.check(self.lib.zett_stream_wait_output(self.handle, code, C.c_void_p(stream.cuda_stream)), "zett_stream_wait_output")

    def gemm_log(self) -> list:
        """Per-launch records of the GEMMs of the most recent forward (zett_get_gemm_log): dicts with m, n, k, variant,
        epilogue (bit mask, include/zett_hip.h), ms (0 unless the "time_gemm" option is on), flops, bytes."""
        n = C.c_int64(0)
        _lib.check(self.lib.zett_get_gemm_log(self.handle, None, 0, C.byref(n)), "zett_get_gemm_log")
        buf = (_lib.ZettGemmRecord * max(1, n.value))()
        _lib.check(self.lib.zett_get_gemm_log(self.handle, buf, n.value, C.byref(n)), "zett_get_gemm_log")
        return [{f: getattr(buf[i], f) for f, _ in _lib.ZettGemmRecord._fields_} for i in range(n.value)]

    def stats(self) -> dict:
        s = _lib.ZettStats()
        _lib.check(self.lib.zett_get_stats(self.handle, C.byref(s)))
        return {n: getattr(s, n) for n, _ in s._fields_}

    def prepare(self, surface_forms: torch.Tensor, input_stream: Optional["torch.cuda.Stream"] = None) -> None:
        """zett_forward_prepare: enqueue the plan of the NEXT forward(surface_forms, ...) on the handle's own stream, behind the
        work `input_stream` (default: the current stream) holds now.  The forward that follows with the SAME tensor then waits
        on the host for that plan only, not for whatever its stream still holds (include/zett_hip.h).  The tensor must already
        be what the C ABI takes — int32, contiguous, on the engine's device — so that the forward sees the same pointer."""
        if surface_forms.dtype != torch.int32 or not surface_forms.is_contiguous() or surface_forms.device != self.device or surface_forms.dim() != 2:
            raise ValueError("prepare() takes the int32, contiguous [n_tokens, surface_maxlen] tensor on the engine's device that forward() will get")
        n, seq = surface_forms.shape
        with torch.cuda.device(self.device):
            stream = (input_stream or torch.cuda.current_stream(self.device)).cuda_stream
            _lib.check(self.lib.zett_forward_prepare(self.handle, C.c_void_p(surface_forms.data_ptr()), n, seq, C.c_void_p(stream)),
                       "zett_forward_prepare")
        self._prepared = surface_forms          # the plan reads it on a stream torch's allocator does not know: keep it alive until the next forward has waited for the plan

    def forward(self, surface_forms: torch.Tensor, source_embeddings: torch.Tensor, lang_index: int):
        d = self.dims
        if surface_forms.dim() != 2:
            raise ValueError("target_surface_forms must be [n_tokens, surface_maxlen]")
        if surface_forms.device != self.device or source_embeddings.device != self.device:
            raise RuntimeError(f"all tensors must be on {self.device}")
        ids = surface_forms.to(torch.int32).contiguous()
        src = source_embeddings
        if src.dtype not in _TORCH_TO_ZETT:
            src = src.float()
        src = src.contiguous()
        if src.dim() != 2 or src.shape[1] != d.n_in_embd:
            raise ValueError(f"source_embeddings must be [V, {d.n_in_embd}], got {tuple(src.shape)}")
        n, seq = ids.shape
        out_in = torch.empty((n, d.n_embd), dtype=torch.float32, device=self.device)
        out_out = torch.empty((n, d.n_embd), dtype=torch.float32, device=self.device) if d.separate_out else None
        out_bias = torch.empty((n,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self.lib.zett_forward(
                self.handle, C.c_void_p(ids.data_ptr()), n, seq, C.c_void_p(src.data_ptr()),
                _TORCH_TO_ZETT[src.dtype], src.shape[0], int(lang_index),
                C.c_void_p(out_in.data_ptr()), C.c_void_p(out_out.data_ptr() if out_out is not None else 0),
                C.c_void_p(out_bias.data_ptr()), C.c_void_p(stream))
        self._prepared = None                   # (zett_forward waited on the host for every plan that read a prepared tensor)
        _lib.check(rc, "zett_forward")
        return out_in, out_out, out_bias

    # ---- the hoisted table shared between ranks (ABI 8; zett_amd/sharding.py SharedTable) -----------------------------------
    def table_plan(self, surface_forms_all: torch.Tensor):
        """zett_table_plan: the distinct source ids the WHOLE vocabulary's surface-form matrix references, ascending ->
        (id_slot int32 [V + 1]: table slot of every id, id_list int32 [n_ids]: slot -> id, n_ids).  One host round trip."""
        ids = surface_forms_all.to(torch.int32).contiguous()
        if ids.dim() != 2 or ids.device != self.device:
            raise ValueError(f"table_plan takes the [n_tokens, surface_maxlen] matrix on {self.device}")
        v = self.dims.original_vocab_size + self.dims.n_extra
        id_slot = torch.empty((v + 1,), dtype=torch.int32, device=self.device)
        id_list = torch.empty((v,), dtype=torch.int32, device=self.device)
        n_ids = C.c_int64(0)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self.lib.zett_table_plan(self.handle, C.c_void_p(ids.data_ptr()), ids.shape[0], ids.shape[1], C.c_void_p(id_slot.data_ptr()),
                                                C.c_void_p(id_list.data_ptr()), C.byref(n_ids), C.c_void_p(stream)), "zett_table_plan")
        return id_slot, id_list[:n_ids.value], int(n_ids.value)

    def table_buffers(self, n_rows: int):
        """Empty buffers of a folded table of n_rows rows: (table [n_rows, H] in the engine's 16-bit type, stats float32 [n_rows, 2])."""
        lo = {"f16": torch.float16, "fp16": torch.float16, "float16": torch.float16, "bf16": torch.bfloat16}.get(self.precision)
        if lo is None:
            raise ValueError("the folded 16-bit table exists in the 16-bit modes only")
        return (torch.empty((n_rows, self.dims.hidden), dtype=lo, device=self.device), torch.empty((n_rows, 2), dtype=torch.float32, device=self.device))

    def table_rows(self, id_list: torch.Tensor, first: int, count: int, source_embeddings: torch.Tensor, table: torch.Tensor, stats: torch.Tensor) -> None:
        """zett_table_rows: rows [first, first + count) of the table of `id_list` into `table` / `stats` (whole-table buffers). Asynchronous."""
        src = source_embeddings
        if src.dtype not in _TORCH_TO_ZETT:
            src = src.float()
        src = src.contiguous()
        if src.dim() != 2 or src.shape[1] != self.dims.n_in_embd or src.device != self.device:
            raise ValueError(f"source_embeddings must be [V, {self.dims.n_in_embd}] on {self.device}, got {tuple(src.shape)}")
        if not (table.is_contiguous() and stats.is_contiguous() and table.shape[0] >= first + count and stats.shape[0] >= first + count and id_list.numel() >= first + count):
            raise ValueError("table / stats / id_list are smaller than the requested rows")
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self.lib.zett_table_rows(self.handle, C.c_void_p(id_list.data_ptr()), first, count, C.c_void_p(src.data_ptr()), _TORCH_TO_ZETT[src.dtype],
                                                src.shape[0], C.c_void_p(table.data_ptr()), C.c_void_p(stats.data_ptr()), C.c_void_p(stream)), "zett_table_rows")

    def forward_table(self, surface_forms: torch.Tensor, table: torch.Tensor, stats: torch.Tensor, id_slot: torch.Tensor, lang_index: int):
        """zett_forward_table: forward() on a complete shared table instead of source embeddings; same outputs, bit for bit."""
        d = self.dims
        if surface_forms.dim() != 2 or surface_forms.device != self.device:
            raise ValueError(f"target_surface_forms must be [n_tokens, surface_maxlen] on {self.device}")
        ids = surface_forms.to(torch.int32).contiguous()
        n, seq = ids.shape
        out_in = torch.empty((n, d.n_embd), dtype=torch.float32, device=self.device)
        out_out = torch.empty((n, d.n_embd), dtype=torch.float32, device=self.device) if d.separate_out else None
        out_bias = torch.empty((n,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self.lib.zett_forward_table(
                self.handle, C.c_void_p(ids.data_ptr()), n, seq, C.c_void_p(table.data_ptr()), C.c_void_p(stats.data_ptr()), C.c_void_p(id_slot.data_ptr()),
                int(lang_index), C.c_void_p(out_in.data_ptr()), C.c_void_p(out_out.data_ptr() if out_out is not None else 0),
                C.c_void_p(out_bias.data_ptr()), C.c_void_p(stream))
        self._prepared = None
        _lib.check(rc, "zett_forward_table")
        return out_in, out_out, out_bias

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.zett_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ZettHypernet(PreTrainedModel):
    config_class = ZettHypernetConfig
    base_model_prefix = ""
    _no_split_modules = []

    def __init__(self, config: ZettHypernetConfig):
        super().__init__(config)
        if getattr(config, "hn_model_type", "roberta") != "roberta":
            raise NotImplementedError()                          # modeling_hypernet.py:78-79
        if getattr(config, "hn_add_inter_token_attention", False) or getattr(config, "hn_embed_target_priors", False):
            raise NotImplementedError()                          # modeling_hypernet.py:85-89
        assert getattr(config, "pad_token_id", None) is not None  # modeling_hypernet.py:92
        if getattr(config, "hn_num_attention_heads", None) is None:
            config.hn_num_attention_heads = config.hn_hidden_size // 64   # modeling_hypernet.py:73-74
        self.has_separate_out_embeddings = getattr(config, "separate_out_embeddings", False)
        self.pad_token_id = config.pad_token_id
        max_pos, ln_eps = _backbone_settings(getattr(config, "hn_model_name_or_path", "roberta-base"))
        self._ln_eps_encoder = ln_eps
        self.dims = HypernetDims.from_config(config, max_positions=max_pos)
        # arithmetic of the dense contractions: "f16" / "bf16" (MFMA operands, fp32 accumulate) or "f32".
        # Default f16: 11-bit significands put the predicted embeddings at rel-L2 1.2e-3 of the fp32 reference at
        # the 4096-wide shapes; bf16 sits at 0.97e-2 on N(0, 0.02^2) weights and 1.04e-2 on a heavy-tailed
        # checkpoint, i.e. on the edge of the 1e-2 tolerance (tests/test_full_size_gpu.py), and is ~4 % faster
        # (same MFMA rate: the chip is power-limited and significand bits cost energy; rounding the f16
        # activations to 9 bits recovered 0.5-1.4 % of that by box at 3x the error and was dropped).
        self.precision = os.environ.get("ZETT_PRECISION", getattr(config, "zett_precision", None) or DEFAULT_PRECISION)
        # Range guard (include/zett_hip.h zett_check_range).  f16 operands overflow above 65504; with the LayerNorm fold the
        # operand copy of the raw residual sum is rounded to half, and a checkpoint with massive activations can leave the
        # range.  Policy of this layer, ONE policy for the class and the CLI: compute in f16, ask the device after every
        # forward (one 4-byte read behind a stream sync), and on a hit repeat the call with bf16 operands (fp32's
        # exponent range, the arithmetic of the reference CLI) with a warning; the model then stays on bf16.  Set
        # range_guard = False to take the asynchronous forward and call engine(...).range_flags() yourself.
        # The per-call check covers f16 ONLY: bf16 / f32 forwards have nothing to fall back to and stay asynchronous — a caller
        # that wants their non-finite-output warning calls check_outputs() (predict_vocabulary does, once, at the end).
        self.range_guard = True
        for name, shape in weight_shapes(self.dims).items():
            _attach(self, name, nn.Parameter(torch.empty(shape, dtype=torch.float32), requires_grad=False))
        self._engines: Dict[Tuple[str, str], HipEngine] = {}
        self._reset_parameters()
        self.post_init()

    @classmethod
    def from_flax_checkpoint(cls, checkpoint_path: str) -> "ZettHypernet":
        """Load ``config.json`` + ``flax_model.msgpack`` (the reference's canonical checkpoint format,
        scripts/transfer.py:131,145-151) without jax / flax."""
        from .flax_io import load_flax_checkpoint
        config = ZettHypernetConfig.from_pretrained(checkpoint_path)
        model = cls(config)
        state = {k: torch.from_numpy(v) for k, v in load_flax_checkpoint(os.path.join(checkpoint_path, "flax_model.msgpack")).items()}
        missing, unexpected = model.load_state_dict(state, strict=False)
        missing = [m for m in missing if m != "model.embeddings.word_embeddings.weight"]
        if missing or unexpected:
            raise ValueError(f"flax checkpoint does not match the config: missing {missing}, unexpected {list(unexpected)}")
        return model.eval()          # (as from_pretrained does)

    # ---- parameter initialisation (same distributions as the torch modules of the reference)
    def _reset_parameters(self) -> None:
        with torch.no_grad():
            for name, p in self.named_parameters():
                if p.device.type == "meta":
                    continue
                leaf = name.rsplit(".", 1)[-1]
                if name.endswith("LayerNorm.weight") or name.endswith("ln.weight") or "scaler." in name:
                    p.fill_(1.0)                                  # Rescaler w and b start at ones (:15-16)
                elif name.endswith("LayerNorm.bias") or name.endswith("ln.bias") or leaf == "bias":
                    p.zero_()
                else:
                    p.normal_(mean=0.0, std=0.02)

    def _init_weights(self, module) -> None:     # parameters are initialised in __init__
        return

    # ---- engine management ---------------------------------------------------------------
    def _drop_engines(self) -> None:
        for eng in self._engines.values():
            eng.close()
        self._engines.clear()
        self.__dict__.pop("_param_slots", None)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if hasattr(self, "_engines"):
            self._drop_engines()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._drop_engines()
        return out

    def refresh_weights(self) -> None:
        """Re-upload parameters at the next forward.  REQUIRED after writes that go through ``p.data`` (``p.data.copy_(...)``,
        ``p.data.add_(...)``, an EMA or a manual weight load written that way): those do not bump the tensor's version
        counter (torch semantics), so engine() cannot see them.  Writes through the parameter itself (``optimizer.step()``,
        ``p.copy_()`` / ``p.add_()`` under no_grad, ``load_state_dict``, ``.to()``) and storage replacement (``p.data = t``)
        are noticed without it."""
        self._drop_engines()

    def engine(self, device: torch.device, precision: Optional[str] = None) -> HipEngine:
        precision = precision or self.precision
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        key = (str(device), precision)
        # An engine holds the weight copy uploaded when it was built.  Parameters modified in place since then through the
        # parameter (optimizer.step(), p.copy_(...), p.add_(...): they bump the tensor's version counter) or given a new storage
        # (p.data = t) make it stale: rebuilt here, so that eval / no_grad prediction after a training step never runs on
        # pre-training weights (train.py's eval_step pattern).  NOT seen: writes through p.data (p.data.add_(...) leaves the
        # version counter alone) — those callers call refresh_weights().  Costs one pass over ~50 tensors per forward (~15 us).
        stamp = self._weights_stamp()
        eng = self._engines.get(key)
        if eng is not None and eng.weights_stamp != stamp:
            eng.close()
            eng = None
        if eng is None:
            eng = HipEngine(self.dims, self._ln_eps_encoder, device, precision)
            eng.load_weights({n: p.data for n, p in self.named_parameters()})
            eng.weights_stamp = stamp
            self._engines[key] = eng
        return eng

    def _weights_stamp(self):
        # What is cached is WHERE the parameters live — (owning module's _parameters dict, name) slots — not the Parameter objects:
        # the slot is read on every call, so a parameter REPLACED after the first forward (`module.weight = nn.Parameter(t)`,
        # register_parameter, parametrize / PEFT-style swaps) shows up as a new (id, version, pointer) and the engine is rebuilt.
        # (named_parameters() walks ~40 modules and was most of this function's cost; the module tree itself is fixed after __init__;
        #  _drop_engines forgets the slots anyway.)
        slots = self.__dict__.get("_param_slots")
        if slots is None:
            slots = self.__dict__["_param_slots"] = [(m._parameters, n) for m in self.modules() for n, p in m._parameters.items() if p is not None]
        try:
            params = [d[n] for d, n in slots]
            if any(p is None for p in params):
                raise KeyError
        except KeyError:          # a parameter was deleted or set to None: the slots are stale, walk the modules again
            slots = self.__dict__["_param_slots"] = [(m._parameters, n) for m in self.modules() for n, p in m._parameters.items() if p is not None]
            params = [d[n] for d, n in slots]
        try:
            return tuple([(id(p), p._version, p.data_ptr()) for p in params])
        except RuntimeError:
            # a model built under torch.inference_mode(): inference tensors have no version counter (and cannot be modified
            # in place outside inference mode either); the objects and storage pointers alone identify the upload
            return tuple([(id(p), 0, p.data_ptr()) for p in params])

    # ---- the forward ----------------------------------------------------------------------
    def forward(self, target_surface_forms, target_priors=None, source_embeddings=None, lang_index=None,
                deterministic: bool = True):
        if target_priors is not None:
            raise NotImplementedError()                          # modeling_hypernet.py:164-165
        if not getattr(self.config, "hn_embed_using_source_embeddings", False):
            raise NotImplementedError()                          # modeling_hypernet.py:167-168
        if getattr(self.config, "hn_concat_last_hidden_state", False):
            # modeling_hypernet.py:231-232 reshapes the hidden states to [N, L' * H], but the port builds its output heads with
            # in_features = hn_hidden_size (:112-144): the reference itself only runs for L' = 1, where the reshape IS hidden[:, 0]
            n_pos = int(torch.as_tensor(target_surface_forms).shape[1]) + (1 if self.dims.embed_lang else 0)
            if n_pos != 1:
                raise NotImplementedError("hn_concat_last_hidden_state with more than one position: the reference's own heads "
                                          "(in_features = hn_hidden_size) cannot take the concatenated states either")
        if source_embeddings is None:
            raise ValueError("source_embeddings is required")
        if not torch.is_tensor(target_surface_forms):
            target_surface_forms = torch.as_tensor(target_surface_forms)
        device = source_embeddings.device
        if device.type != "cuda":
            raise RuntimeError("zett_amd computes on MI355X only: move target_surface_forms and "
                               "source_embeddings to a cuda (ROCm) device; there is no CPU path")
        if target_surface_forms.device != device:
            target_surface_forms = target_surface_forms.to(device)
        if self.dims.embed_lang:
            if lang_index is None:
                raise ValueError("this hypernetwork embeds a language id: lang_index is required")
            lang = int(lang_index.item()) if torch.is_tensor(lang_index) else int(lang_index)
        else:
            lang = -1
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training use (train.py:1007-1013): the same forward, differentiable with respect to the parameters
            # (zett_amd/autograd.py: fp32, HIP primitives through the C ABI).  Only in train() mode: a model that
            # from_pretrained returned (eval mode, whatever its parameters' requires_grad flags say) predicts on the
            # inference path, which builds no graph
            from .autograd import differentiable_forward
            return differentiable_forward(self, target_surface_forms, source_embeddings, lang, packed=getattr(self, "train_packed", True),
                                          precision=getattr(self, "train_precision", "f32"))
        return self._guarded_forward(device, target_surface_forms, source_embeddings, lang)

    def _guarded_forward(self, device, surface_forms, source_embeddings, lang):
        import warnings
        if self.precision in ("f16", "fp16", "float16") and self.range_guard:
            try:
                eng = self.engine(device)
                self._last_engine = eng
                out = eng.forward(surface_forms, source_embeddings, lang)
                flags = eng.range_flags()
            except _lib.RangeError as err:           # zett_finalize: a weight does not fit the half type
                out, flags = None, _lib.RANGE_WEIGHT
                warnings.warn(f"zett_amd: {err}; continuing with bf16 operands")
            if not flags:
                return out
            if out is not None:
                where = [n for b, n in ((_lib.RANGE_SOURCE, "in_scaler(source_embeddings)"), (_lib.RANGE_ACTIVATION, "a 16-bit activation"),
                                        (_lib.RANGE_OUTPUT, "non-finite outputs")) if flags & b]
                warnings.warn("zett_amd: the f16 forward left the half range (" + ", ".join(where) + "); repeating the call with bf16 "
                              "operands (fp32 exponent range, rel-L2 ~1e-2 instead of ~1e-3 of the fp32 reference). The model stays on bf16.")
            self.precision = "bf16"
        # bf16 / f32: nothing to fall back to, so the call stays ASYNCHRONOUS (no stream sync, no device-to-host read: the
        # caller's next batch preparation and index_add overlap the forward, as batched_inference relies on).  A caller that
        # wants to know whether the outputs are finite asks once, when it chooses: model.check_outputs().
        eng = self.engine(device)
        self._last_engine = eng
        return eng.forward(surface_forms, source_embeddings, lang)

    def check_outputs(self) -> int:
        """Range word of the most recent bf16 / f32 forward (waits for the stream): 0, or _lib.RANGE_* bits; warns when the
        predicted embeddings are not finite (non-finite weights or source embeddings: the reference would return them too)."""
        import warnings
        eng = getattr(self, "_last_engine", None)
        if eng is None or eng.handle is None:
            return 0
        flags = eng.range_flags()
        if flags & _lib.RANGE_OUTPUT:
            warnings.warn(f"zett_amd: non-finite predicted embeddings in {self.precision} arithmetic (non-finite weights or source embeddings?); "
                          "returned as computed, as the reference would")
        return flags
