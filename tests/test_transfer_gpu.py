"""End-to-end scripts/transfer.py surface on synthetic local checkpoints (no network): a tiny GPT-2
language model, its byte-level BPE tokenizer as hn tokenizer, a second tokenizer as transfer target.
Checks the spliced embedding matrix against the oracle (forward + retokenizer) and the reference's
special-token rule (scripts/transfer.py:274-300)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import hypernet_ref, retok_ref
from tests import util
from zett_amd import synth

pytestmark = pytest.mark.gpu


def _train_bpe(lines, size):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tok.train_from_iterator(lines, trainers.BpeTrainer(vocab_size=size, special_tokens=["<|endoftext|>"],
                                                       initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False))
    return PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|endoftext|>")


def _lines(seed):
    import glob
    import random
    rng = random.Random(seed)
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.__file__), "*.py")))
    rng.shuffle(files)
    out = []
    for f in files[:25]:
        out += [ln.strip() for ln in open(f, encoding="utf-8", errors="ignore") if len(ln.strip()) > 20]
    rng.shuffle(out)
    return out[:3000]


def _make_checkpoints(tmp_path, fmt):
    """A tiny GPT-2 language model + tokenizer, a second tokenizer as transfer target, and a hypernetwork checkpoint in
    PyTorch or flax layout.  Returns (dirs, cfg, weights, lm, src_tok)."""
    from transformers import GPT2Config, GPT2LMHeadModel

    import zett_amd  # noqa: F401
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet

    src_dir, hn_dir, tgt_dir, out_dir = (str(tmp_path / d) for d in ("lm", "hypernet", "target_tok", "out"))
    src_tok = _train_bpe(_lines(1), 600)
    tgt_tok = _train_bpe(_lines(2), 900)
    tgt_tok.save_pretrained(tgt_dir)
    torch.manual_seed(0)
    lm = GPT2LMHeadModel(GPT2Config(vocab_size=len(src_tok), n_embd=64, n_layer=1, n_head=2, n_positions=32))
    lm.save_pretrained(src_dir)
    src_tok.save_pretrained(src_dir)

    cfg = dict(synth.workload("tiny")[0], n_embd=64, separate_out_embeddings=False, hn_embed_lang_id=False,
               original_vocab_size=len(src_tok), hn_n_extra_tokens=0, pad_token_id=src_tok.eos_token_id, vocab_size=len(src_tok))
    weights = synth.make_weights(cfg, 21)
    if fmt == "pt":
        hn = ZettHypernet(ZettHypernetConfig(**cfg))
        hn.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
        hn.save_pretrained(hn_dir)
    else:
        from tests.flax_fixture import write_flax_checkpoint
        os.makedirs(hn_dir, exist_ok=True)
        ZettHypernetConfig(**cfg).save_pretrained(hn_dir)
        weights = dict(weights, **write_flax_checkpoint(hn_dir, cfg, weights))      # (position embeddings come back bf16-rounded)
        assert not any(f.endswith((".safetensors", ".bin")) for f in os.listdir(hn_dir))
    src_tok.save_pretrained(hn_dir)                       # the checkpoint ships its hn tokenizer (transfer.py:153-157)
    return (src_dir, hn_dir, tgt_dir, out_dir), cfg, weights, lm, src_tok


def _cli_args(dirs, out_dir=None):
    src_dir, hn_dir, tgt_dir, out = dirs
    return ["--output", out_dir or out, "--checkpoint_path", hn_dir, "--tokenizer_name", tgt_dir, "--target_model", src_dir,
            "--model_class", "AutoModelForCausalLM", "--dtype", "float32", "--batch_size", "256"]


@pytest.mark.parametrize("fmt", ["pt", "flax"])
def test_transfer_cli_end_to_end(tmp_path, fmt):
    """fmt = "flax": the hypernet checkpoint directory holds config.json + flax_model.msgpack only (the reference's
    canonical format, scripts/transfer.py:145-151), written byte by byte in flax's layout by tests/flax_fixture.py."""
    from transformers import AutoModelForCausalLM, AutoTokenizer

    from zett_amd.byte_level import convert_to_byte_level
    from zett_amd.transfer import main

    dirs, cfg, weights, lm, src_tok = _make_checkpoints(tmp_path, fmt)
    src_dir, hn_dir, tgt_dir, out_dir = dirs

    main(_cli_args(dirs))

    new_tok = AutoTokenizer.from_pretrained(out_dir)
    new_lm = AutoModelForCausalLM.from_pretrained(out_dir)
    emb = new_lm.get_input_embeddings().weight.detach().numpy()
    assert emb.shape == (len(new_tok), 64) and new_lm.config.vocab_size == len(new_tok)
    assert np.array_equal(new_lm.get_output_embeddings().weight.detach().numpy(), emb)      # GPT-2 ties them

    # expected: oracle retokenizer + oracle forward on the same converted tokenizers
    hn_tok = convert_to_byte_level(AutoTokenizer.from_pretrained(hn_dir))[0]
    hn_tok.pad_token = hn_tok.eos_token
    tgt_conv = convert_to_byte_level(AutoTokenizer.from_pretrained(tgt_dir), make_whitespace_consistent=True,
                                     match_special_tokens_to=AutoTokenizer.from_pretrained(src_dir))[0]
    tokens = tgt_conv.convert_ids_to_tokens(range(len(tgt_conv)))
    assert len(tokens) == len(new_tok)
    model = retok_ref.model_from_hf_tokenizer(hn_tok)
    sfm, _ = retok_ref.surface_form_matrix_c(model, tokens, cfg["hn_surface_maxlen"], hn_tok.pad_token_id)
    source = lm.get_input_embeddings().weight.detach().numpy()
    want = hypernet_ref.forward(weights, cfg, sfm, source, None)[0]
    special_id = tgt_conv.get_vocab()["<|endoftext|>"]
    want[special_id] = source[src_tok.eos_token_id]
    util.assert_f32_close(emb, want, "spliced input embeddings")
    assert np.array_equal(emb[special_id], source[src_tok.eos_token_id])
    assert os.path.exists(os.path.join(out_dir, "bias.safetensors"))


def test_transfer_cli_two_processes_match_one(tmp_path):
    """`torchrun --nproc-per-node 2 scripts/transfer.py ...`: every batch is sharded over the ranks and all-gathered
    (scripts/transfer.py:90-91), rank 0 writes the model, and the written embeddings equal the single-process run bit for
    bit.  On a 1-GPU box both ranks share cuda:0 and the collectives go through gloo (ZETT_ONE_DEVICE_TEST); with two or
    more GPUs the same command runs on RCCL (tests/test_multi_gpu.py covers the nccl collectives)."""
    import subprocess
    import sys

    from safetensors.torch import load_file

    from zett_amd.transfer import main

    dirs, *_ = _make_checkpoints(tmp_path, "pt")
    one, two = str(tmp_path / "out1"), str(tmp_path / "out2")
    main(_cli_args(dirs, one))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < 2:
        env["ZETT_ONE_DEVICE_TEST"] = "1"
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(29620 + os.getpid() % 100), os.path.join(repo, "scripts", "transfer.py")] + _cli_args(dirs, two),
                         cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    a, b = load_file(os.path.join(one, "model.safetensors")), load_file(os.path.join(two, "model.safetensors"))
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(load_file(os.path.join(one, "bias.safetensors"))["bias"], load_file(os.path.join(two, "bias.safetensors"))["bias"])


def test_transfer_cli_tears_down_on_a_rank_failure(tmp_path):
    """A rank that dies in the middle of the sharded prediction must not leave the job hanging: under torchrun the CLI's other
    rank sits in a collective with a two-hour timeout (zett_amd/transfer.py init_distributed), so the launcher — which kills
    the surviving workers when one fails — is what ends it.  Rank 1 raises inside predict_vocabulary (injected here through a
    wrapper script, no hook in the product); the command must exit non-zero within the time a normal run takes, write no
    model, and leave no worker behind.  Reference: the device sharding of scripts/transfer.py:90-91 (one process there)."""
    import subprocess
    import sys
    import time

    dirs, *_ = _make_checkpoints(tmp_path, "pt")
    out = str(tmp_path / "out_fail")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    wrapper = tmp_path / "transfer_with_a_failing_rank.py"
    wrapper.write_text(f"""
import os, sys
sys.path.insert(0, {repo!r})
import zett_amd.transfer as T
real = T.predict_vocabulary
def failing(*a, **k):
    if int(os.environ.get("RANK", "0")) == 1:
        raise RuntimeError("injected failure on rank 1")
    return real(*a, **k)
T.predict_vocabulary = failing
T.main()
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < 2:
        env["ZETT_ONE_DEVICE_TEST"] = "1"
    t0 = time.time()
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(29720 + os.getpid() % 100), str(wrapper)] + _cli_args(dirs, out),
                         cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    elapsed = time.time() - t0
    assert res.returncode != 0, "a failed rank must fail the command"
    assert "injected failure on rank 1" in res.stderr
    assert elapsed < 300, f"the surviving rank kept the job alive for {elapsed:.0f} s"
    assert not os.path.exists(os.path.join(out, "model.safetensors"))


def test_batched_prediction_computes_the_hoisted_table_once_per_job(monkeypatch):
    """r6 (ABI 8): a vocabulary predicted in several batches (the reference CLI's default, scripts/transfer.py:243-262 with --batch_size)
    computes input_projection(in_scaler(source_embeddings[id])) ONCE for the distinct ids of the whole vocabulary (SharedTable) and runs
    every batch on that table, instead of once per batch for the batch's ids: the same predictions BIT FOR BIT as the single forward and as
    the batches without the table (ZETT_JOB_TABLE=0).  f16 policy and a hypernet with the folded table only: the tiny hypernet (H = 128) and
    an explicit float32 / bfloat16 run predict as before."""
    from zett_amd.transfer import Args, predict_vocabulary
    cfg, _, src_dtype, hist = synth.workload("xlmr_gpt2")
    w = synth.make_weights(cfg, 8)
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 8, dtype=src_dtype)).cuda()
    sfm = torch.from_numpy(synth.make_surface_forms(cfg, 2500, seed=8, hist=hist, n_special=2)).cuda()
    lang = torch.tensor(3)
    model = util.hip_model(cfg, w, "f16").eval()

    def same(a, b):
        return all((x is None and y is None) or torch.equal(x, y) for x, y in zip(a, b))

    monkeypatch.delenv("ZETT_JOB_TABLE", raising=False)
    predict_vocabulary.last_job_table = None
    one = predict_vocabulary(model, sfm, src, lang, Args(output="", do_batching=False))
    assert predict_vocabulary.last_job_table is None                  # (one forward: the class's own call)
    batched = predict_vocabulary(model, sfm, src, lang, Args(output="", batch_size=700))
    assert predict_vocabulary.last_job_table is True and same(batched, one)
    pri = -np.abs(np.random.default_rng(3).standard_normal(2500))
    sampled = predict_vocabulary(model, sfm, src, lang, Args(output="", batch_size=700, sample_batches=True, n_samples=12, min_k=2), target_priors=pri,
                                 rng=np.random.default_rng(5))
    assert predict_vocabulary.last_job_table is True
    for a, b in zip(sampled, one):          # (rows predicted several times are averaged: equal to rounding of the mean)
        assert a is None and b is None or torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    monkeypatch.setenv("ZETT_JOB_TABLE", "0")
    predict_vocabulary.last_job_table = None
    assert same(predict_vocabulary(model, sfm, src, lang, Args(output="", batch_size=700)), one) and predict_vocabulary.last_job_table is None
    monkeypatch.delenv("ZETT_JOB_TABLE")
    model.precision = "bf16"
    predict_vocabulary.last_job_table = None
    b1 = predict_vocabulary(model, sfm, src, lang, Args(output="", do_batching=False))
    assert same(predict_vocabulary(model, sfm, src, lang, Args(output="", batch_size=700)), b1) and predict_vocabulary.last_job_table is None
    # a hypernet without the folded table (H = 128): the job asks for the table, the library refuses, the batches predict as before
    tcfg = synth.workload("tiny")[0]
    tw = synth.make_weights(tcfg, 8)
    tsrc = torch.from_numpy(synth.make_source_embeddings(tcfg, 8)).cuda()
    tsfm = torch.from_numpy(synth.make_surface_forms(tcfg, 900, seed=8, n_special=2)).cuda()
    tiny = util.hip_model(tcfg, tw, "f16").eval()
    t1 = predict_vocabulary(tiny, tsfm, tsrc, torch.tensor(2), Args(output="", do_batching=False))
    t2 = predict_vocabulary(tiny, tsfm, tsrc, torch.tensor(2), Args(output="", batch_size=256))
    assert predict_vocabulary.last_job_table is False and same(t1, t2)


def test_two_ranks_share_the_job_table_in_predict_vocabulary(tmp_path):
    """predict_vocabulary under torchrun (two ranks; on a 1-GPU box both on cuda:0 over gloo): the job's hoisted table is computed half per
    rank and all-gathered once, every batch is sharded over the ranks and runs on it, and every rank ends with the single forward's
    predictions bit for bit — also for a ONE-batch job with ZETT_SHARED_TABLE=1.  XLM-R-shaped hypernet, 3 001 rows, f16."""
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = tmp_path / "job_table_worker.py"
    worker.write_text(f"""
import os, sys
sys.path.insert(0, {repo!r})
import numpy as np, torch, torch.distributed as dist
from tests import util
from zett_amd import synth
from zett_amd.transfer import Args, init_distributed, predict_vocabulary
device = init_distributed()
cfg, _, src_dtype, hist = synth.workload("xlmr_gpt2")
w = synth.make_weights(cfg, 8)
src = torch.from_numpy(synth.make_source_embeddings(cfg, 8, dtype=src_dtype)).to(device)
sfm = torch.from_numpy(synth.make_surface_forms(cfg, 3001, seed=8, hist=hist, n_special=2)).to(device)
model = util.hip_model(cfg, w, "f16").eval()
lang = torch.tensor(3)
eng = model.engine(device)
one = eng.forward(sfm, src, 3)                      # the plain local forward of the whole matrix
same = lambda a, b: all((x is None and y is None) or torch.equal(x, y) for x, y in zip(a, b))
predict_vocabulary.last_job_table = None
got = predict_vocabulary(model, sfm, src, lang, Args(output="", batch_size=1000))
assert predict_vocabulary.last_job_table is True and same(got, one), "batched job on the shared table"
os.environ["ZETT_SHARED_TABLE"] = "1"
got = predict_vocabulary(model, sfm, src, lang, Args(output="", do_batching=False))
assert predict_vocabulary.last_job_table is True and same(got, one), "one-batch job on the shared table"
del os.environ["ZETT_SHARED_TABLE"]
got = predict_vocabulary(model, sfm, src, lang, Args(output="", do_batching=False))
assert predict_vocabulary.last_job_table is False and same(got, one), "one-batch job, own tables"
dist.barrier(); dist.destroy_process_group()
print("rank", os.environ["RANK"], "ok")
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < 2:
        env["ZETT_ONE_DEVICE_TEST"] = "1"
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(29820 + os.getpid() % 100), str(worker)], cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "rank 0 ok" in res.stdout and "rank 1 ok" in res.stdout
