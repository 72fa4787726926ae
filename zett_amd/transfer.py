"""Zero-shot tokenizer transfer CLI — the surface of the reference's scripts/transfer.py
on MI355X, with PyTorch/safetensors weights instead of Flax msgpack.

    python scripts/transfer.py --output out/ --checkpoint_path <hypernet> \
        --tokenizer_name <target tokenizer> --target_model <LM> --model_class AutoModelForCausalLM

Same 17 flags and defaults as scripts/transfer.py:30-51.  The embedding-prediction
path (surface-form matrix -> hypernetwork -> [V, E] matrices) runs in HIP through
libzett_hip.so; this module holds only the host orchestration the reference keeps in
Python: ``batched_inference`` (scripts/transfer.py:54-124), the special-token
overwrite (scripts/transfer.py:274-300) and the splice into the language model.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import torch

NEGATIVE_INF_FILL_VALUE = -100000.0      # zett/utils.py (score used for filled-in unigram pieces)


@dataclass
class Args:
    output: str
    checkpoint_path: str = "output_gpt2_noise_std_1.0_inter_embed_bias_scratch_nlayers=6"
    tokenizer_name: str = "artifacts/gpt2_unigramify"
    model_class: str = "AutoModel"
    # args for target model and projection
    target_model: str = "gpt2"
    copy_inner_parameters_from: str = None
    dtype: str = "bfloat16"
    revision: str = None
    do_batching: bool = True
    batch_size: int = 16384
    sample_batches: bool = False
    min_k: int = 10
    n_samples: int = 100
    lang_path: str = None
    lang_code: str = None
    make_whitespace_consistent: bool = True
    save_pt: bool = False


Predict = Callable[[torch.Tensor], Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]]


def get_sample_indices(n: int, p: np.ndarray, batch_size: int, min_k: int, n_samples: int,
                       rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """Batch composition of ``--sample_batches`` (contract of zett/utils.py:612-648): the ``n_samples`` batches form
    ``min_k`` sweeps of ``n_samples // min_k`` batches; within a sweep a fresh shuffle of all ``n`` tokens is dealt over the
    sweep's batches (``n // per_sweep`` each, the sweep's last batch takes the remainder), so every token is visited at
    least ``min_k`` times; each batch is then filled up to ``batch_size`` with tokens drawn without replacement,
    proportionally to exp(prior), from those it does not hold yet.  The random stream is consumed in the reference's order
    (one permutation up front and one after each sweep, one weighted draw per batch)."""
    rng = rng or np.random.default_rng()
    per_sweep, rem = divmod(n_samples, min_k)
    assert rem == 0 and per_sweep > 0
    share = n // per_sweep
    # dealt[j] = the slice of a sweep's shuffle that batch j of the sweep holds
    bounds = [(j * share, n if j == per_sweep - 1 else (j + 1) * share) for j in range(per_sweep)]
    prior_weight = np.exp(np.where(p > NEGATIVE_INF_FILL_VALUE, p, -np.inf))
    out = np.empty((n_samples, batch_size), dtype=np.int32)
    shuffle = rng.permutation(n)
    for i in range(n_samples):
        lo, hi = bounds[i % per_sweep]
        dealt = shuffle[lo:hi]
        out[i, :len(dealt)] = dealt
        if (i + 1) % per_sweep == 0:
            shuffle = rng.permutation(n)          # (the reference draws the next sweep's shuffle before the batch's fill)
        w = prior_weight.copy()
        w[dealt] = 0
        out[i, len(dealt):] = rng.choice(n, size=batch_size - len(dealt), p=w / w.sum(), replace=False)
    return out


def batched_inference(predict: Predict, target_surface_form_matrix: torch.Tensor, n_embd: int, batch_size: int = 16384,
                      sample_batches: bool = False, target_priors: Optional[np.ndarray] = None, min_k: int = 10,
                      n_samples: int = 100, rng: Optional[np.random.Generator] = None):
    """scripts/transfer.py:54-124 with device-resident accumulation.

    Rows are visited in a random permutation, the last batch is padded with row 0, predictions
    are accumulated into fp32 [V, E] matrices (and averaged by visit count when batches are
    sampled).  Rows are independent, so the result does not depend on the batching.
    """
    rng = rng or np.random.default_rng()
    sfm = target_surface_form_matrix
    device = sfm.device
    n = sfm.shape[0]
    if sample_batches:
        assert target_priors is not None
        batches = list(get_sample_indices(n, np.asarray(target_priors), batch_size, min_k, n_samples, rng))
        empty_in_last = 0
    else:
        total = math.ceil(n / batch_size) * batch_size
        padded = np.pad(rng.permutation(n), (0, total - n))
        batches = np.array_split(padded, total // batch_size)
        empty_in_last = total - n
    acc_in = torch.zeros((n, n_embd), dtype=torch.float32, device=device)
    acc_out = None
    acc_bias = torch.zeros((n,), dtype=torch.float32, device=device)
    counts = torch.zeros((n,), dtype=torch.float32, device=device)
    for bi, idx in enumerate(batches):
        index = torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(device)
        p_in, p_out, p_bias = predict(sfm.index_select(0, index))
        if bi == len(batches) - 1 and empty_in_last > 0:
            keep = len(idx) - empty_in_last
            index, p_in, p_bias = index[:keep], p_in[:keep], p_bias[:keep]
            p_out = None if p_out is None else p_out[:keep]
        acc_in.index_add_(0, index, p_in.float())
        acc_bias.index_add_(0, index, p_bias.float())
        if p_out is not None:
            if acc_out is None:
                acc_out = torch.zeros((n, n_embd), dtype=torch.float32, device=device)
            acc_out.index_add_(0, index, p_out.float())
        counts.index_add_(0, index, torch.ones_like(index, dtype=torch.float32))
    if sample_batches:   # tokens seen several times are averaged (scripts/transfer.py:113-122)
        assert bool((counts > 0).all())
        acc_in /= counts[:, None]
        acc_bias /= counts
        if acc_out is not None:
            acc_out /= counts[:, None]
    return acc_in, acc_out, acc_bias


def overwrite_special_tokens(predicted: torch.Tensor, source_embeddings: torch.Tensor,
                             source_special_ids: Sequence[int], target_special_ids: Sequence[int]) -> torch.Tensor:
    """scripts/transfer.py:274-300: rows of the source model's special tokens are copied, not predicted."""
    src = torch.as_tensor(list(source_special_ids), dtype=torch.long, device=source_embeddings.device)
    dst = torch.as_tensor(list(target_special_ids), dtype=torch.long, device=predicted.device)
    predicted[dst] = source_embeddings.index_select(0, src).to(predicted.dtype).to(predicted.device)
    return predicted


def predict_vocabulary(hypernet, target_surface_form_matrix: torch.Tensor, source_embeddings: torch.Tensor,
                       lang_index=None, args: Optional[Args] = None, target_priors=None, rng=None):
    """The whole-vocabulary prediction of scripts/transfer.py:221-270 (batching flags honoured)."""
    args = args or Args(output="")

    def predict_local(rows):
        return hypernet(rows, source_embeddings=source_embeddings, lang_index=lang_index)

    # One process per GPU (torchrun): every batch is cut into row shards over the ranks and all-gathered, as the reference
    # shards every batch over its local devices (scripts/transfer.py:90-91, zett/utils.py:26).  All ranks then hold the
    # whole result; they must walk the SAME batches, so the batch order comes from one seed, broadcast from rank 0.
    predict = predict_local
    # The hoisted table once per JOB (r6, ABI 8): input_projection(in_scaler(source_embeddings[id])) depends on the id only, and a
    # vocabulary walked in several batches (the reference's default: --batch_size 16384) references most ids in every batch — the
    # table of the whole vocabulary's distinct ids is then computed once (1/P of it per rank, all-gathered: zett_amd.sharding.SharedTable)
    # and every batch's forward runs on it; the same rows bit for bit.  f16 arithmetic (the default policy) and a hypernet with the
    # folded table (H >= 512) only — anything else predicts as before.  ZETT_JOB_TABLE=0 switches it off; a ONE-batch job on several
    # ranks shares the table only with ZETT_SHARED_TABLE=1 (there the exchange is not amortised: DESIGN.md section 6).
    n_rows = int(target_surface_form_matrix.shape[0])
    n_batches = 1 if not args.do_batching else (int(args.n_samples) if args.sample_batches else -(-n_rows // max(int(args.batch_size), 1)))
    f16_names = ("f16", "fp16", "float16")

    def want_table(precision):
        if os.environ.get("ZETT_JOB_TABLE", "1") == "0" or precision not in f16_names or not source_embeddings.is_cuda:
            return False
        return n_batches >= 2 or (_world_size() > 1 and os.environ.get("ZETT_SHARED_TABLE") == "1")

    sharded = hasattr(hypernet, "engine") and (_world_size() > 1 or want_table(getattr(hypernet, "precision", None)))
    if _world_size() > 1 and not sharded:              # (a stand-in model without an engine: plain sharding)
        from zett_amd.sharding import predict_sharded

        def predict(rows):
            return predict_sharded(predict_local, rows)

        rng = _shared_rng(rng, target_surface_form_matrix.device)
    elif sharded:
        # One process per GPU: the forwards stay ASYNCHRONOUS (the class's per-call range check would synchronise the host
        # after every shard and take the early exchange of pred_in with it): the range word accumulates over the whole
        # vocabulary and all ranks ask once at the end — together, so that a fallback to bf16 is every rank's or none's.
        from zett_amd.sharding import predict_sharded
        device = source_embeddings.device
        if hypernet.dims.embed_lang and lang_index is None:
            raise ValueError("this hypernetwork embeds a language id: lang_index is required")
        lang = int(lang_index) if hypernet.dims.embed_lang else -1
        rng = _shared_rng(rng, target_surface_form_matrix.device) if _world_size() > 1 else (rng or np.random.default_rng())
        state = {"rng_state": rng.bit_generator.state}

        def run_with(precision):
            import torch.distributed as dist
            from zett_amd import _lib
            try:
                eng = hypernet.engine(device, precision)
            except _lib.RangeError as err:               # zett_finalize: a weight (or gamma-folded weight) does not fit the half type.
                import warnings                          # Every rank holds the same weights, so every rank lands here together.
                warnings.warn(f"zett_amd: {err}")
                return None, _lib.RANGE_WEIGHT
            eng.set_option("range_accumulate", 1)
            eng.range_flags()                           # (clears whatever an earlier call left)
            shared = None
            if want_table(precision):
                from zett_amd.sharding import SharedTable
                try:
                    shared = SharedTable(eng, target_surface_form_matrix.to(device=device, dtype=torch.int32).contiguous(), source_embeddings)
                except ValueError:                      # no folded table for this hypernet / mode (H < 512, LayerNorm fold off): predict as before
                    shared = None
            predict_vocabulary.last_job_table = shared is not None      # (what the last call did: tests, logs)
            fwd = (lambda r: eng.forward_table(r, shared.table, shared.stats, shared.id_slot, lang)) if shared is not None else \
                  (lambda r: eng.forward(r, source_embeddings, lang))

            def predict_rows(rows):
                r32 = rows.to(device=device, dtype=torch.int32).contiguous()        # what the C ABI takes: prepare() and forward() see one pointer
                return predict_sharded(fwd, r32, ready=eng.stream_wait_output, prepare=None if shared is not None else eng.prepare)

            if not args.do_batching:
                out = predict_rows(target_surface_form_matrix)
            else:
                gen = np.random.default_rng()
                gen.bit_generator.state = state["rng_state"]        # the same batch order on a repeat
                out = batched_inference(predict_rows, target_surface_form_matrix, hypernet.config.n_embd, args.batch_size,
                                        args.sample_batches, target_priors, args.min_k, args.n_samples, gen)
            flags = reduce_flag_word(eng.range_flags(), device) if _world_size() > 1 else eng.range_flags()
            eng.set_option("range_accumulate", 0)
            return out, flags

        import warnings
        out, flags = run_with(hypernet.precision)
        if flags and hypernet.precision in ("f16", "fp16", "float16") and getattr(hypernet, "range_guard", True):
            warnings.warn("zett_amd: the f16 forward left the half range on some rank; repeating the prediction with bf16 operands on every rank")
            hypernet.precision = "bf16"
            out, flags = run_with("bf16")
        if out is None:          # (a weight outside the half range with the guard off, or the caller forced f16)
            raise _lib_range_error("a GEMM weight does not fit the f16 operand type and the range guard is off: pass --dtype bfloat16")
        if flags:
            warnings.warn(f"zett_amd: non-finite predicted embeddings in {hypernet.precision} arithmetic; returned as computed")
        return out

    if not args.do_batching:   # scripts/transfer.py:243-262 pads to a multiple of 128 for XLA; no need here
        out = predict(target_surface_form_matrix)
    else:
        out = batched_inference(predict, target_surface_form_matrix, hypernet.config.n_embd, args.batch_size, args.sample_batches,
                                target_priors, args.min_k, args.n_samples, rng)
    # bf16 / f32 forwards are asynchronous and unguarded (ZettHypernet._guarded_forward): ask once, here, whether the last
    # forward's outputs were finite (warns; the f16 path has asked after every call)
    if hasattr(hypernet, "check_outputs"):
        hypernet.check_outputs()
    return out


def _lib_range_error(msg: str):
    from zett_amd import _lib
    return _lib.RangeError(msg)


def reduce_flag_word(word: int, device, group=None, bits: int = 8) -> int:
    """Bitwise OR of a small flag word over the ranks.  RCCL has no BOR ("Cannot use ReduceOp.BOR with NCCL"), and MAX of the
    words would lose bits (max(1, 2) = 2): one int32 per bit, reduced with MAX — works on nccl and gloo alike."""
    import torch.distributed as dist
    vec = torch.tensor([(int(word) >> b) & 1 for b in range(bits)], dtype=torch.int32, device=device)
    dist.all_reduce(vec, op=dist.ReduceOp.MAX, group=group)
    return sum(int(v) << b for b, v in enumerate(vec.tolist()))


def _world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _shared_rng(rng, device) -> np.random.Generator:
    """A generator every rank seeds identically: rank 0 draws the seed (from `rng` when the caller gave one)."""
    import torch.distributed as dist
    seed = int((rng or np.random.default_rng()).integers(0, 2 ** 62))
    t = torch.tensor([seed], dtype=torch.int64, device=device)
    dist.broadcast(t, src=0)
    return np.random.default_rng(int(t.item()))


def init_distributed(device_index_from_env: bool = True):
    """torchrun launch (WORLD_SIZE > 1 in the environment): bind this process to its GPU and join the RCCL group
    (backend "nccl" is RCCL on ROCm).  Returns the device; a plain `python scripts/transfer.py` stays single-GPU."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # ZETT_ONE_DEVICE_TEST=1 (test hook, as in bench.py): every rank uses cuda:0 and the collectives go through gloo, so
    # that the multi-process control flow runs on a 1-GPU box; never set otherwise.
    one_device = os.environ.get("ZETT_ONE_DEVICE_TEST") == "1"
    if world > 1 and device_index_from_env:
        torch.cuda.set_device(0 if one_device else int(os.environ.get("LOCAL_RANK", "0")))
    device = torch.device("cuda", torch.cuda.current_device())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (the collectives of a run are all over within seconds of each other; the long timeout only keeps a watchdog from
        #  firing while one rank is still paging a 7B-parameter model in)
        import datetime
        timeout = datetime.timedelta(hours=2)
        if one_device:
            dist.init_process_group("gloo", timeout=timeout)
        else:
            dist.init_process_group("nccl", device_id=device, timeout=timeout)
    return device


def dtype_given(argv) -> bool:
    """True when --dtype was passed on the command line (argv None = sys.argv)."""
    import sys
    args = sys.argv[1:] if argv is None else list(argv)
    return any(a == "--dtype" or a.startswith("--dtype=") for a in args)


def precision_for_dtype(dtype: str, explicit: bool) -> str:
    """Arithmetic of the dense contractions for the CLI's --dtype (scripts/transfer.py:41).

    The reference flag is the dtype its parameters are held and computed in; its default is bfloat16.  Here parameters are
    fp32 and the flag picks the MFMA operand type.  NOT passing the flag selects the library policy
    (zett_amd/hypernet.py DEFAULT_PRECISION: f16 operands with the range guard and its bf16 fallback — 8x closer to the
    fp32 reference than bf16 at the same MFMA rate, and inside SURVEY.md section 8d's tolerance where bf16 sits on its edge);
    passing --dtype bfloat16 / float16 / float32 explicitly selects exactly that arithmetic; anything else is an error
    (the reference would fail in getattr(jnp, dtype))."""
    from .hypernet import DEFAULT_PRECISION
    if not explicit and dtype == "bfloat16":
        return DEFAULT_PRECISION
    table = {"bfloat16": "bf16", "float32": "f32", "float16": "f16"}
    if dtype not in table:
        raise ValueError(f"--dtype {dtype!r}: expected one of {sorted(table)}")
    return table[dtype]


def target_priors_of(tokenizer) -> np.ndarray:
    """scripts/transfer.py:210-219: the Unigram scores of the target tokenizer where its model has them (padded with
    0.0 for added tokens), uniform otherwise.  Rows are independent, so the priors only steer which rows share a batch
    under --sample_batches."""
    model = tokenizer._tokenizer.model
    if hasattr(model, "get_scores"):
        priors = list(model.get_scores())
    else:
        print("WARNING: using uniform priors, get_scores() not available.")
        priors = [0.0] * len(tokenizer)
    priors += [0.0] * (len(tokenizer) - len(priors))          # for added special tokens
    return np.array(priors)


def main(argv=None):
    import transformers
    from transformers import AutoConfig, AutoModel, AutoTokenizer, HfArgumentParser

    import zett_amd  # noqa: F401  (registers the hypernetwork with AutoModel)
    from zett_amd.byte_level import convert_to_byte_level
    from zett_amd.surface_forms import surface_form_matrix_device

    (args,) = HfArgumentParser([Args]).parse_args_into_dataclasses(argv)
    zett_amd.configure_hw_queues()          # (torchrun: before the HIP runtime initialises)
    if not torch.cuda.is_available():
        raise SystemExit("scripts/transfer.py needs an MI355X: torch.cuda.is_available() is False")
    device = init_distributed()          # one process per GPU under torchrun; a single process otherwise

    tokenizer = AutoTokenizer.from_pretrained(args.tokenizer_name)
    config = AutoConfig.from_pretrained(args.checkpoint_path)
    lang_index = None
    if args.lang_code is not None:                                   # scripts/transfer.py:133-143
        langs = getattr(config, "langs", None)
        if langs is None:
            assert args.lang_path is not None
            langs = [x.strip() for x in open(args.lang_path).readlines()]
        lang_index = torch.tensor(langs.index(args.lang_code), dtype=torch.int32)

    # PyTorch weights if the checkpoint has them, else the reference's canonical flax_model.msgpack (scripts/transfer.py:
    # 145-151 restores exactly that file), read without jax / flax by zett_amd/flax_io.py
    has_pt = any(os.path.exists(os.path.join(args.checkpoint_path, f)) for f in
                 ("model.safetensors", "pytorch_model.bin", "model.safetensors.index.json", "pytorch_model.bin.index.json"))
    if not has_pt and os.path.exists(os.path.join(args.checkpoint_path, "flax_model.msgpack")):
        from zett_amd.hypernet import ZettHypernet
        hypernet = ZettHypernet.from_flax_checkpoint(args.checkpoint_path).to(device)
    else:
        hypernet = AutoModel.from_pretrained(args.checkpoint_path).to(device)
    hypernet.precision = precision_for_dtype(args.dtype, dtype_given(argv))

    source_tokenizer = AutoTokenizer.from_pretrained(args.target_model)
    hn_tokenizer = type(source_tokenizer).from_pretrained(args.checkpoint_path)
    hn_tokenizer = convert_to_byte_level(hn_tokenizer)[0]            # scripts/transfer.py:153-159
    if hn_tokenizer.pad_token is None:
        hn_tokenizer.pad_token = hn_tokenizer.eos_token

    model_cls = getattr(transformers, args.model_class)
    downstream = model_cls.from_pretrained(args.target_model, revision=args.revision, torch_dtype=torch.float32)
    emb_in = downstream.get_input_embeddings().weight.data
    out_layer = downstream.get_output_embeddings() if args.model_class != "AutoModel" else None
    tied = bool(getattr(downstream.config, "tie_word_embeddings", False)) or out_layer is None
    emb_out = None if tied else out_layer.weight.data
    source_embeddings = (emb_in if emb_out is None else torch.cat([emb_in, emb_out], dim=1)).to(device)   # :162-191

    if _rank() != 0:
        # only rank 0 writes the model: the other ranks needed the embedding matrices (now on their GPU) and nothing else
        downstream = emb_in = emb_out = out_layer = None
    elif args.copy_inner_parameters_from is not None:
        downstream = model_cls.from_pretrained(args.copy_inner_parameters_from, torch_dtype=torch.float32)

    tokenizer = convert_to_byte_level(tokenizer, make_whitespace_consistent=args.make_whitespace_consistent,
                                      match_special_tokens_to=source_tokenizer)[0]                       # :198-202
    tokens = tokenizer.convert_ids_to_tokens(range(len(tokenizer)))
    sfm, n_truncated = surface_form_matrix_device(tokens, config.hn_surface_maxlen, hn_tokenizer, device)  # :204-206
    print(f"Truncated {n_truncated} tokens.")

    target_priors = target_priors_of(tokenizer)                                                          # :210-219
    hypernet.eval()
    with torch.no_grad():          # inference: the pad-skipping / hoisting forward, no autograd graph
        pred_in, pred_out, pred_bias = predict_vocabulary(hypernet, sfm.long(), source_embeddings, lang_index, args, target_priors)

    # every rank holds the whole prediction; the last collective is over.  The group is torn down HERE, so that no rank
    # sits in a barrier (and no watchdog runs) while rank 0 spends minutes writing a multi-GB model.
    if _world_size() > 1:
        import torch.distributed as dist
        rank = _rank()
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
    special_src = list(source_tokenizer.all_special_ids)                                                  # :274-300
    special_dst = [tokenizer.get_vocab()[t] for t in source_tokenizer.all_special_tokens]
    os.makedirs(args.output, exist_ok=True)
    source_tokenizer.save_pretrained(args.output)
    tokenizer.save_pretrained(args.output)
    pred_in = overwrite_special_tokens(pred_in, emb_in, special_src, special_dst)
    downstream.resize_token_embeddings(len(tokenizer))
    downstream.get_input_embeddings().weight.data.copy_(pred_in.cpu())
    if emb_out is not None and pred_out is not None:
        pred_out = overwrite_special_tokens(pred_out, emb_out, special_src, special_dst)
        downstream.get_output_embeddings().weight.data.copy_(pred_out.cpu())
    bias_param = getattr(downstream.get_output_embeddings(), "bias", None) if out_layer is not None else None
    if bias_param is not None:
        bias_param.data.copy_(pred_bias.cpu())
    else:
        from safetensors.torch import save_file
        save_file({"bias": pred_bias.cpu()}, os.path.join(args.output, "bias.safetensors"))
    downstream.config.vocab_size = len(tokenizer)
    downstream.save_pretrained(args.output, max_shard_size="20GB")


if __name__ == "__main__":
    main()
