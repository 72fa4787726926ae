"""Host-side logic that needs no GPU: the C ABI surface, config / checkpoint contract,
batched_inference, special-token overwrite, row sharding (gloo, world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import util
from zett_amd import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C ABI ---------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    from zett_amd import _lib
    header = open(os.path.join(REPO, "include", "zett_hip.h")).read()
    declared = set(re.findall(r"\b(zett_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.zett_abi_version() == _lib.ABI_VERSION == 8


def test_cli_dtype_selects_one_precision_policy():
    """--dtype (scripts/transfer.py:41): not passed = the library policy (f16 operands + range guard); passed explicitly =
    exactly that arithmetic; unknown strings raise instead of silently becoming bf16 (r2 advisor finding)."""
    from zett_amd.hypernet import DEFAULT_PRECISION
    from zett_amd.transfer import Args, dtype_given, precision_for_dtype
    assert Args(output="x").dtype == "bfloat16"                      # the reference's default string is kept
    assert DEFAULT_PRECISION == "f16"
    assert not dtype_given(["--output", "o"]) and dtype_given(["--dtype", "bfloat16"]) and dtype_given(["--dtype=float32"])
    assert precision_for_dtype("bfloat16", explicit=False) == DEFAULT_PRECISION
    assert precision_for_dtype("bfloat16", explicit=True) == "bf16"
    assert precision_for_dtype("float16", explicit=True) == "f16"
    assert precision_for_dtype("float32", explicit=True) == "f32"
    with pytest.raises(ValueError):
        precision_for_dtype("float64", explicit=True)


def test_struct_layouts_match_header():
    from zett_amd import _lib
    assert ctypes.sizeof(_lib.ZettConfig) == 16 * 4 + 2 * 4
    assert ctypes.sizeof(_lib.ZettStats) == 9 * 8
    assert ctypes.sizeof(_lib.ZettGemmRecord) == 5 * 4 + 4 + 2 * 8
    # zett_retok_model: int,int,4 ptr,double,int,ptr,3 int,ptr,int,int,3 ptr  (natural alignment)
    assert _lib.ZettRetokModel.piece_scores.offset == 32
    assert _lib.ZettRetokModel.unigram_min_score.offset == 40
    # ABI 4: the WordPiece fields at the end (pointer, int32; the struct is padded to 8)
    assert _lib.ZettRetokModel.piece_continuing.offset == _lib.ZettRetokModel.special_ids.offset + 8
    assert _lib.ZettRetokModel.max_input_chars_per_word.offset == _lib.ZettRetokModel.piece_continuing.offset + 8
    assert ctypes.sizeof(_lib.ZettRetokModel) == _lib.ZettRetokModel.max_input_chars_per_word.offset + 8


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, *_ = synth.workload("tiny")
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    with pytest.raises(RuntimeError, match="MI355X"):
        model(torch.zeros(2, 7, dtype=torch.long), source_embeddings=torch.zeros(300, 128), lang_index=torch.tensor(0))


def test_concat_last_hidden_state_beyond_one_position_is_refused():
    """modeling_hypernet.py:231-232: the reference's own heads take hn_hidden_size inputs, so it raises (a shape error) for
    more than one position itself; here that is a NotImplementedError before anything is computed."""
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg = dict(synth.workload("tiny")[0], hn_concat_last_hidden_state=True)
    model = ZettHypernet(ZettHypernetConfig(**cfg)).eval()
    with pytest.raises(NotImplementedError, match="concat"):
        model(torch.zeros(2, 7, dtype=torch.long), source_embeddings=torch.zeros(300, 128), lang_index=torch.tensor(0))


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "zett_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


# ---- config / checkpoint contract --------------------------------------------------------------
def test_reference_error_behaviour():
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, *_ = synth.workload("tiny")
    with pytest.raises(NotImplementedError):
        ZettHypernet(ZettHypernetConfig(**dict(cfg, hn_model_type="t5")))
    with pytest.raises(NotImplementedError):
        ZettHypernet(ZettHypernetConfig(**dict(cfg, hn_add_inter_token_attention=True)))
    with pytest.raises(NotImplementedError):
        ZettHypernet(ZettHypernetConfig(**dict(cfg, hn_embed_target_priors=True)))
    with pytest.raises(AssertionError):
        ZettHypernet(ZettHypernetConfig(**{k: v for k, v in cfg.items() if k != "pad_token_id"}))
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    with pytest.raises(NotImplementedError):
        model(torch.zeros(1, 7, dtype=torch.long), target_priors=torch.zeros(1), source_embeddings=torch.zeros(300, 128))
    model2 = ZettHypernet(ZettHypernetConfig(**dict(cfg, hn_embed_using_source_embeddings=False)))
    with pytest.raises(NotImplementedError):
        model2(torch.zeros(1, 7, dtype=torch.long), source_embeddings=torch.zeros(300, 128))


def test_state_dict_names_and_automodel_roundtrip(tmp_path):
    from transformers import AutoConfig, AutoModel

    import zett_amd  # noqa: F401
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.dims import weight_shapes
    from zett_amd.hypernet import ZettHypernet
    cfg, *_ = synth.workload("tiny")
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    sd = model.state_dict()
    want = weight_shapes(cfg)
    assert list(sd.keys()) == list(want.keys())
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in want.items())
    w = synth.make_weights(cfg, 3)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model.save_pretrained(tmp_path)
    again = AutoModel.from_pretrained(tmp_path)
    assert type(again).__name__ == "ZettHypernet"
    for k, v in w.items():
        assert torch.equal(again.state_dict()[k], torch.from_numpy(v)), k
    c = AutoConfig.from_pretrained(tmp_path)
    assert c.model_type == "zett_hypernetwork" and c.hn_surface_maxlen == 7 and c.original_vocab_size == 300


@pytest.mark.skipif(not os.path.isdir("/root/reference/hf_hypernet"), reason="reference only exists in the build container")
def test_loads_checkpoint_written_by_reference(tmp_path):
    """A checkpoint saved by the reference's own torch class loads into ours unchanged."""
    code = f"""
import sys, json, torch, numpy as np
sys.path.insert(0, "/root/reference"); sys.path.insert(0, {REPO!r})
from hf_hypernet.configuration_hypernet import ZettHypernetConfig
from hf_hypernet.modeling_hypernet import ZettHypernet
from zett_amd import synth
import os
rb = os.path.join({str(tmp_path)!r}, "rb"); os.makedirs(rb)
json.dump({{"model_type": "roberta", "max_position_embeddings": 514, "type_vocab_size": 1, "layer_norm_eps": 1e-5,
           "hidden_act": "gelu", "vocab_size": 50265, "pad_token_id": 1}}, open(os.path.join(rb, "config.json"), "w"))
cfg, *_ = synth.workload("tiny")
m = ZettHypernet(ZettHypernetConfig(**dict(cfg, hn_model_name_or_path=rb)))
m.load_state_dict({{k: torch.from_numpy(v) for k, v in synth.make_weights(cfg, 5).items()}})
m.save_pretrained(os.path.join({str(tmp_path)!r}, "ckpt"))
"""
    subprocess.run([sys.executable, "-c", code], check=True, cwd="/tmp")
    # reference checkpoints carry no usable model_type (the class sets it on the instance only) and are
    # loaded by remote code upstream; here the class is addressed directly.
    from zett_amd.hypernet import ZettHypernet
    cfg, *_ = synth.workload("tiny")
    model = ZettHypernet.from_pretrained(os.path.join(tmp_path, "ckpt"))
    assert model.config.hn_surface_maxlen == 7 and model.dims.original_vocab_size == 300
    for k, v in synth.make_weights(cfg, 5).items():
        assert torch.equal(model.state_dict()[k], torch.from_numpy(v)), k


# ---- batched_inference / special tokens (scripts/transfer.py:54-124, 274-300) ---------------------
def _fake_predict(rows):
    x = rows.float()
    e = torch.arange(1, 9, dtype=torch.float32)
    return x.sum(1, keepdim=True) * e, x.max(1, keepdim=True).values * e, x[:, 0].float()


def test_batched_inference_is_batching_independent():
    from zett_amd.transfer import batched_inference
    sfm = torch.from_numpy(np.random.default_rng(0).integers(0, 50, size=(1001, 7)))
    want = _fake_predict(sfm)
    for bs in (64, 1001, 4096):
        got = batched_inference(_fake_predict, sfm, 8, batch_size=bs, rng=np.random.default_rng(bs))
        for g, w in zip(got, want):
            assert torch.equal(g, w)


def test_batched_inference_sampled_batches_average():
    from zett_amd.transfer import batched_inference, get_sample_indices
    n = 200
    pri = np.random.default_rng(1).normal(size=n)
    idx = get_sample_indices(n, pri, batch_size=64, min_k=2, n_samples=10, rng=np.random.default_rng(2))
    assert idx.shape == (10, 64) and set(np.unique(idx)) == set(range(n))
    assert all(len(set(row)) == 64 for row in idx)
    sfm = torch.from_numpy(np.random.default_rng(0).integers(0, 50, size=(n, 7)))
    got = batched_inference(_fake_predict, sfm, 8, batch_size=64, sample_batches=True, target_priors=pri, min_k=2,
                            n_samples=10, rng=np.random.default_rng(3))
    for g, w in zip(got, _fake_predict(sfm)):
        torch.testing.assert_close(g, w, rtol=1e-6, atol=1e-6)


def test_overwrite_special_tokens():
    from zett_amd.transfer import overwrite_special_tokens
    pred = torch.zeros(10, 4)
    src = torch.arange(40, dtype=torch.float32).reshape(10, 4)
    out = overwrite_special_tokens(pred, src, [0, 2, 9], [5, 1, 3])
    assert torch.equal(out[5], src[0]) and torch.equal(out[1], src[2]) and torch.equal(out[3], src[9])
    assert out[[0, 2, 4, 6, 7, 8]].abs().sum() == 0


def test_transfer_cli_flags_match_reference():
    import dataclasses

    from zett_amd.transfer import Args
    names = [f.name for f in dataclasses.fields(Args)]
    assert names == ["output", "checkpoint_path", "tokenizer_name", "model_class", "target_model",
                     "copy_inner_parameters_from", "dtype", "revision", "do_batching", "batch_size", "sample_batches",
                     "min_k", "n_samples", "lang_path", "lang_code", "make_whitespace_consistent", "save_pt"]
    d = Args(output="x")
    assert (d.dtype, d.do_batching, d.batch_size, d.make_whitespace_consistent, d.model_class) == \
        ("bfloat16", True, 16384, True, "AutoModel")


# ---- vocab-row sharding over ranks (gloo, world_size 2 and 3) --------------------------------------
def test_shard_bounds_cover_rows():
    from zett_amd.sharding import padded_rows, plan_blocks, shard_bounds
    for n in (0, 1, 7, 8, 1001, 32768):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # row blocks: every global row belongs to exactly one (block, rank); block k starts where the gathered shards of the
    # blocks before it end, so shards land in place; small vocabularies are not cut below 4 096 rows per shard
    for n in (1, 7, 1001, 4096, 32768, 50370, 262144):
        for world in (1, 2, 3, 8):
            for chunks in (1, 2, 4):
                per_rank = [plan_blocks(n, world, r, chunks) for r in range(world)]
                nb = len(per_rank[0])
                assert all(len(b) == nb for b in per_rank) and 1 <= nb <= chunks
                owner = [0] * n
                for r, blocks in enumerate(per_rank):
                    off = 0
                    for b in blocks:
                        assert b.start == off and (b.start, b.rows, b.per) == (per_rank[0][blocks.index(b)].start, per_rank[0][blocks.index(b)].rows, per_rank[0][blocks.index(b)].per)
                        assert b.lo == min(b.start + r * b.per, b.start + b.rows) and b.hi - b.lo <= b.per
                        for i in range(b.lo, b.hi):
                            owner[i] += 1
                        off += world * b.per
                    assert padded_rows(blocks, world) == off >= n
                assert all(c == 1 for c in owner)
                assert all(b.rows == world * b.per for b in per_rank[0][:-1])
                if nb > 1:
                    assert per_rank[0][0].per >= 4096


_WORKER = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
from oracle import hypernet_ref
from zett_amd import synth
from zett_amd.sharding import predict_sharded
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg, *_ = synth.workload("tiny")
w = synth.make_weights(cfg, 9); src = synth.make_source_embeddings(cfg, 9)
ids = synth.make_surface_forms(cfg, 37, seed=9, n_special=1)
def predict(rows):   # the oracle stands in for the HIP engine: this test covers the partition + all-gather
    o = hypernet_ref.forward(w, cfg, rows.numpy(), src, 2)
    return tuple(None if x is None else torch.from_numpy(x) for x in o)
full = predict_sharded(predict, torch.from_numpy(ids))
single = predict(torch.from_numpy(ids))
# numpy BLAS is not batch-size invariant, so the oracle stand-in is compared to rounding; the
# bit-exactness of the gather itself is checked with a row-wise deterministic predictor
ok = all(torch.allclose(a, b, rtol=0, atol=1e-5) for a, b in zip(full, single))
def fake(rows):
    x = rows.float(); e = torch.arange(1, 5, dtype=torch.float32)
    return x.sum(1, keepdim=True) * e, None, x[:, 0].clone()
f_full = predict_sharded(fake, torch.from_numpy(ids)); f_single = fake(torch.from_numpy(ids))
ok = ok and torch.equal(f_full[0], f_single[0]) and f_full[1] is None and torch.equal(f_full[2], f_single[2])
# several row blocks per rank (the all-gather of a block overlaps the next block's forward): 5 003 rows, 2 and 3 blocks
big = torch.arange(40003 * 7, dtype=torch.int64).reshape(40003, 7) % 1000
for chunks in (1, 2, 3):
    b_full = predict_sharded(fake, big, chunks=chunks); b_single = fake(big)
    ok = ok and torch.equal(b_full[0], b_single[0]) and b_full[1] is None and torch.equal(b_full[2], b_single[2])
from zett_amd.sharding import plan_blocks
blocks = plan_blocks(40003, world, rank, 3)
ok = ok and len(blocks) == 3 and all(b.rows == world * b.per for b in blocks[:-1]) and sum(b.rows for b in blocks) == 40003
# the direct fan-out transport (every rank sends its shard to all peers at once) puts the same bytes in the same places,
# with and without a second output and with rows that do not divide by the world size
def fake2(rows):
    x = rows.float(); e = torch.arange(1, 5, dtype=torch.float32)
    return x.sum(1, keepdim=True) * e, (x.sum(1, keepdim=True) - 3) * e, x[:, 0].clone()
for chunks in (1, 3):
    for fn in (fake, fake2):
        f_fan = predict_sharded(fn, big, chunks=chunks, mode="fanout"); f_one = fn(big)
        ok = ok and all((a is None and b is None) or torch.equal(a, b) for a, b in zip(f_fan, f_one))
f_fan = predict_sharded(fake2, torch.from_numpy(ids), mode="fanout"); f_one = fake2(torch.from_numpy(ids))
ok = ok and all(torch.equal(a, b) for a, b in zip(f_fan, f_one))
# the early-start hook is only taken for device tensors: on CPU tensors it must be ignored, not called
called = []
f_hook = predict_sharded(fake2, big, chunks=2, ready=lambda name, stream: called.append(name))
ok = ok and not called and all(torch.equal(a, b) for a, b in zip(f_hook, fake2(big)))
try:
    predict_sharded(fake, big, mode="ring"); ok = False
except ValueError:
    pass
# rows sharded in another ORDER (zett_amd.sharding.affinity_order on the GPU; any permutation here) and put back: same result
g = torch.Generator(); g.manual_seed(5)
for chunks in (1, 3):
    order = torch.randperm(big.shape[0], generator=g)
    for mode in ("allgather", "fanout"):
        f_perm = predict_sharded(fake2, big, chunks=chunks, mode=mode, order=order)
        ok = ok and all(torch.equal(a, b) for a, b in zip(f_perm, fake2(big)))
try:
    predict_sharded(fake, big, order=torch.arange(3)); ok = False
except ValueError:
    pass
# the hoisted table shared between ranks (zett_amd.sharding.SharedTable, ABI 8): every rank plans the same distinct-id list, computes ITS
# slice, the slices are all-gathered, and the sharded forward on the complete table equals the single-process one — here with a
# stand-in engine (table row i = a function of id_list[i]); the HIP engine's bit-identity is tests/test_invariants_gpu.py's
from zett_amd.sharding import SharedTable
class FakeEngine:
    V = 1000
    def table_plan(self, m):
        ids = torch.unique(m)
        flag = torch.zeros(self.V, dtype=torch.int32); flag[ids] = 1
        slot = torch.zeros(self.V + 1, dtype=torch.int32); slot[1:] = torch.cumsum(flag, 0)
        return slot, ids.to(torch.int32), int(ids.numel())
    def table_buffers(self, n):
        return torch.full((n, 8), float("nan"), dtype=torch.float16), torch.full((n, 2), float("nan"))
    def table_rows(self, id_list, first, count, src, table, stats):
        i = id_list[first:first + count].long()
        table[first:first + count] = src[i].to(torch.float16); stats[first:first + count, 0] = i.float(); stats[first:first + count, 1] = 1.0
    def forward_table(self, rows, table, stats, id_slot, lang):
        sl = id_slot[rows.long()].long()
        x = table[sl].float().sum(1) * stats[sl][..., 1].sum(1, keepdim=True) + stats[sl][..., 0].sum(1, keepdim=True)
        return x, None, x[:, 0].clone() + lang
g2 = torch.Generator(); g2.manual_seed(11)
src_t = torch.randn(1000, 8, generator=g2)
eng = FakeEngine()
shared = SharedTable(eng, big, src_t)
lo_, hi_ = shared.rows
ok = ok and shared.world == world and shared.rank == rank and shared.n_ids == 1000 and (lo_, hi_) == (min(rank * shared.per, 1000), min((rank + 1) * shared.per, 1000))
ok = ok and not torch.isnan(shared.table[:1000].float()).any() and not torch.isnan(shared.stats[:1000]).any()          # every peer's slice arrived
ok = ok and shared.bytes_received() == (world - 1) * shared.per * (8 * 2 + 8)
alone = SharedTable(eng, big, src_t, only_rank=0, world=1)
ok = ok and torch.equal(alone.table[:1000], shared.table[:1000]) and torch.equal(alone.stats[:1000], shared.stats[:1000])
for chunks in (1, 2):
    t_full = predict_sharded(shared.predict(3), big, chunks=chunks); t_one = alone.predict(3)(big)
    ok = ok and torch.equal(t_full[0], t_one[0]) and t_full[1] is None and torch.equal(t_full[2], t_one[2])
# a rank's share in three pieces, each all-gathered behind its computation (rows of one piece of all ranks are contiguous): the same table
piecewise = SharedTable(eng, big, src_t, pieces=3)
ok = ok and piecewise.pieces == 3 and len(piecewise.ranges) == 3 and piecewise.per == 3 * piecewise.piece_rows
ok = ok and all(lo == min((k * world + rank) * piecewise.piece_rows, 1000) for k, (lo, hi) in enumerate(piecewise.ranges))
ok = ok and torch.equal(piecewise.table[:1000], alone.table[:1000]) and torch.equal(piecewise.stats[:1000], alone.stats[:1000])
p_full = predict_sharded(piecewise.predict(3), big); p_one = alone.predict(3)(big)
ok = ok and torch.equal(p_full[0], p_one[0]) and torch.equal(p_full[2], p_one[2])
shapes = [tuple(t.shape) for t in full]
if rank == 0:
    json.dump({{"ok": ok, "shapes": shapes}}, open({out!r}, "w"))
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_predict_sharded_gloo(tmp_path, world):
    import json
    out = os.path.join(tmp_path, "res.json")
    script = os.path.join(tmp_path, "worker.py")
    open(script, "w").write(_WORKER.format(repo=REPO, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + world + (os.getpid() % 200)),
               OMP_NUM_THREADS="2")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                    "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], script],
                   check=True, env=env, timeout=600, cwd="/tmp")
    res = json.load(open(out))
    assert res["ok"], "all-gathered shards differ from the single-process result"
    assert res["shapes"] == [[37, 64], [37, 64], [37]]


_CLI_WORKER = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
from zett_amd.transfer import Args, predict_vocabulary
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
class Cfg: n_embd = 4
class Fake:                      # a row-wise deterministic stand-in for the hypernetwork: this test covers the CLI's sharding
    config = Cfg()
    calls = 0
    def __call__(self, rows, source_embeddings=None, lang_index=None):
        Fake.calls += rows.shape[0]
        x = rows.float(); e = torch.arange(1, 5, dtype=torch.float32)
        return x.sum(1, keepdim=True) * e, (x.sum(1, keepdim=True) + 1) * e, x[:, 0].clone()
sfm = (torch.arange(1003 * 7, dtype=torch.int64).reshape(1003, 7) * 7919) % 1000
want = Fake()(sfm); Fake.calls = 0
ok = True
# every rank draws a DIFFERENT local generator: the batch order must still agree (seed broadcast from rank 0)
got = predict_vocabulary(Fake(), sfm, None, None, Args(output="", batch_size=128), rng=np.random.default_rng(100 + rank))
ok = ok and all(torch.equal(a, b) for a, b in zip(got, want))
per_rank_rows = Fake.calls; Fake.calls = 0
got = predict_vocabulary(Fake(), sfm, None, None, Args(output="", do_batching=False))
ok = ok and all(torch.equal(a, b) for a, b in zip(got, want))
pri = -np.abs(np.random.default_rng(3).standard_normal(1003))
got = predict_vocabulary(Fake(), sfm, None, None, Args(output="", batch_size=128, sample_batches=True, n_samples=40, min_k=5), target_priors=pri)
ok = ok and all(torch.allclose(a, b, rtol=1e-6, atol=0) for a, b in zip(got, want))
# the range word of the sharded run is OR-ed over the ranks bit by bit (RCCL has no BOR, MAX of the words loses bits)
from zett_amd.transfer import reduce_flag_word
ok = ok and reduce_flag_word(1 if rank == 0 else 2, "cpu") == 3 and reduce_flag_word(0, "cpu") == 0
ok = ok and reduce_flag_word(8 if rank == 1 else 0, "cpu") == 8 and reduce_flag_word(5, "cpu") == 5
flag = torch.tensor([1 if ok else 0]); dist.all_reduce(flag, op=dist.ReduceOp.MIN)
rows = torch.tensor([per_rank_rows]); dist.all_reduce(rows)
if rank == 0:
    json.dump({{"ok": bool(flag.item()), "rows_all_ranks": int(rows.item())}}, open({out!r}, "w"))
dist.barrier(); dist.destroy_process_group()
"""


def test_cli_prediction_shards_batches_over_ranks_gloo(tmp_path):
    """zett_amd.transfer.predict_vocabulary under torchrun: every batch is cut over the ranks (scripts/transfer.py:90-91),
    all ranks end with the whole result, and the ranks together compute each batch once (not once per rank)."""
    import json
    out = os.path.join(tmp_path, "res.json")
    script = os.path.join(tmp_path, "worker.py")
    open(script, "w").write(_CLI_WORKER.format(repo=REPO, out=out))
    port = str(29700 + (os.getpid() % 200))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                    "--master-addr", "127.0.0.1", "--master-port", port, script],
                   check=True, env=env, timeout=600, cwd="/tmp")
    res = json.load(open(out))
    assert res["ok"], "sharded CLI prediction differs from the single-process result"
    assert res["rows_all_ranks"] == 8 * 128          # 1003 rows in 8 batches of 128 (the last one padded), each computed once


@pytest.mark.skipif(not os.path.isdir("/root/reference/hf_hypernet"), reason="reference only exists in the build container")
def test_install_remote_code_routes_reference_checkpoint(tmp_path):
    """AutoModel(..., trust_remote_code=True) on a reference-written checkpoint returns the zett_amd class."""
    code = f"""
import sys, json, torch, os
sys.path.insert(0, "/root/reference"); sys.path.insert(0, {REPO!r})
from hf_hypernet.configuration_hypernet import ZettHypernetConfig
from hf_hypernet.modeling_hypernet import ZettHypernet
from zett_amd import synth
rb = os.path.join({str(tmp_path)!r}, "rb"); os.makedirs(rb)
json.dump({{"model_type": "roberta", "max_position_embeddings": 514, "type_vocab_size": 1, "layer_norm_eps": 1e-5,
           "hidden_act": "gelu", "vocab_size": 50265, "pad_token_id": 1}}, open(os.path.join(rb, "config.json"), "w"))
cfg, *_ = synth.workload("tiny")
ZettHypernetConfig.register_for_auto_class()
ZettHypernet.register_for_auto_class("AutoModel")
m = ZettHypernet(ZettHypernetConfig(**dict(cfg, hn_model_name_or_path=rb)))
m.load_state_dict({{k: torch.from_numpy(v) for k, v in synth.make_weights(cfg, 8).items()}})
m.save_pretrained(os.path.join({str(tmp_path)!r}, "ckpt"))
"""
    subprocess.run([sys.executable, "-c", code], check=True, cwd="/tmp")
    ckpt = os.path.join(tmp_path, "ckpt")
    import zett_amd
    zett_amd.install_remote_code(ckpt)
    check = f"""
import sys; sys.path.insert(0, {REPO!r})
import torch
from transformers import AutoModel
from zett_amd import synth
m = AutoModel.from_pretrained({ckpt!r}, trust_remote_code=True)
assert type(m).__name__ == "ZettHypernet" and "zett_amd" in sys.modules, type(m)
from zett_amd.hypernet import ZettHypernet
assert isinstance(m, ZettHypernet) or type(m).__module__.endswith("modeling_hypernet")
cfg, *_ = synth.workload("tiny")
for k, v in synth.make_weights(cfg, 8).items():
    assert torch.equal(m.state_dict()[k], torch.from_numpy(v)), k
assert hasattr(m, "engine")
print("ok")
"""
    out = subprocess.run([sys.executable, "-c", check], check=True, cwd="/tmp", capture_output=True, text=True)
    assert "ok" in out.stdout


def test_no_product_kernel_uses_scratch(tmp_path):
    """Every kernel of libzett_hip.so must fit its registers: a spilled GEMM epilogue once returned wrong
    tiles on the first launches of a process (gemm384 with a residual epilogue), and scratch reloads
    sit behind vmcnt(0) waits inside store sequences.  hipcc's resource-usage remarks: the ones zett_amd.build kept
    beside each object when it compiled the translation unit, as long as no source changed since (r6: recompiling every
    unit here took two of the CPU suite's six minutes) - otherwise a device-only compile of that unit."""
    import re
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    from zett_amd.build import CSRC, SOURCES, fresh_remarks
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

    def remarks(name):
        kept = fresh_remarks(name)
        if kept is not None and "Function Name:" in kept:
            return kept
        out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only", os.path.join(CSRC, name),
                              "-o", str(tmp_path / (name + ".o")), "-Rpass-analysis=kernel-resource-usage"],
                             capture_output=True, text=True, timeout=1500)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stderr

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:      # one translation unit per kernel family
        text = "".join(pool.map(remarks, SOURCES))
    names = re.findall(r"Function Name: (\S+)", text)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", text)]
    assert len(names) == len(scratch) and len(names) > 50
    bad = [(n, s) for n, s in zip(names, scratch) if s]
    assert not bad, bad


def test_training_primitives_refuse_bad_arguments_before_touching_the_device():
    """Error behaviour of the zett_op_* entry points at the boundary: arguments are validated before any launch, the return code is
    ZETT_E_INVALID and zett_last_error says why (no GPU needed: nothing is launched)."""
    import ctypes as C
    from zett_amd import _lib
    lib = _lib.load()
    P = C.c_void_p
    buf = (C.c_float * 4096)()
    a = C.cast(buf, P)
    null = P(0)

    def refused(rc, *words):
        assert rc == _lib.E_INVALID, rc
        msg = lib.zett_last_error().decode()
        assert all(w in msg for w in words), msg

    refused(lib.zett_op_gemm_f32(a, 32, a, 32, 4, 4, 20, null, 0, null, 0, a, 4, null), "multiple of 32")
    refused(lib.zett_op_gemm_f32(null, 32, a, 32, 4, 4, 32, null, 0, null, 0, a, 4, null), "null")
    refused(lib.zett_op_gemm_lo(_lib.PREC_BF16, a, 64, a, 64, 4, 4, 96, null, 0, null, 0, a, 4, null), "multiple of 64")
    refused(lib.zett_op_gemm_lo(_lib.PREC_F32, a, 64, a, 64, 4, 4, 64, null, 0, null, 0, a, 4, null), "ZETT_PREC")
    refused(lib.zett_op_convert_lo(_lib.PREC_F16, a, 8, a, 4, 2, 8, 8, null), "conversion")          # ld_out < cols_padded
    refused(lib.zett_op_transpose_lo(_lib.PREC_F16, a, 8, a, 2, 4, 8, 4, null), "transpose")          # ld_out < rows_padded
    refused(lib.zett_op_transpose_lo16(7, a, 8, a, 64, 4, 8, 64, null), "ZETT_PREC")
    refused(lib.zett_op_grad_operands_lo(_lib.PREC_BF16, a, 8, null, 0, 0, 4, 8, 64, null, 8, a, 64, a, null), "null")
    refused(lib.zett_op_grad_operands_lo(_lib.PREC_BF16, a, 8, a, 8, 3, 4, 8, 64, a, 8, a, 64, a, null), "activation kind")
    refused(lib.zett_op_elementwise_f32(9, a, a, null, null, null, a, 16, 4, null), "elementwise")
    refused(lib.zett_op_layernorm_fwd_f32(a, 8, a, a, 1e-5, a, a, 2, 6, null, 0, null), "multiples of 4")
    refused(lib.zett_op_layernorm_fwd_f32(a, 8200, a, a, 1e-5, a, a, 2, 8200, null, 0, null), "8192")
    refused(lib.zett_op_layernorm_bwd_f32(a, null, a, 8, a, a, a, a, 0, 2, 8, null), "n_part")
    refused(lib.zett_op_gelu_fwd_f32(a, a, 16, 5, null), "gelu")
    refused(lib.zett_op_gelu_fwd_lo(_lib.PREC_F16, a, a, 16, 0, null), "gelu")
    m = C.cast((C.c_uint8 * 64)(), P)
    refused(lib.zett_op_attention_fwd_f32(a, 64, a, a, 64, m, null, 1, 33, 1, 64, 0, a, 64, a, null, 0, null), "32")
    refused(lib.zett_op_attention_fwd_f32(a, 64, a, a, 64, m, null, 1, 4, 1, 300, 0, a, 64, a, null, 0, null), "256")
    refused(lib.zett_op_attention_fwd_f32(a, 64, a, a, 64, m, null, 1, 4, 1, 64, 0, null, 64, a, a, _lib.PREC_F32, null), "16-bit context")
    refused(lib.zett_op_attention_bwd_f32(a, 64, a, 64, a, a, 64, a, null, 1, 20, 1, 192, 0, a, 64, a, a, 64, null), "16 positions")
    refused(lib.zett_op_gather_fwd_f32(a, 4, a, 9, 8, 10, a, null, null, a, null), "gather")
    # and nothing to do is not an error
    assert lib.zett_op_gemm_f32(a, 32, a, 32, 0, 4, 32, null, 0, null, 0, a, 4, null) == 0
    assert lib.zett_op_gelu_fwd_f32(a, a, 0, 1, null) == 0


# ---- round 5: host pieces of the NUL-separated retokenizer input and of the row partition --------------------------
def test_flatten_tokens_equals_the_per_token_walk():
    """DeviceRetokenizer.flatten_tokens (one join / encode + numpy offsets) == encoding every token by itself, including empty
    tokens, two-byte characters, a token that holds the separator itself (falls back to the walk) and the empty list."""
    import numpy as np

    from zett_amd.surface_forms import DeviceRetokenizer as D

    def walk(tokens):
        enc = [t.encode("utf-8") for t in tokens]
        off = np.zeros(len(enc) + 1, dtype=np.int32)
        if enc:
            np.cumsum([len(e) for e in enc], out=off[1:])
        return np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8), off

    for tokens in (["Ġhello", "", "wörld", "a", "", ""], ["a"], [""], [], ["", ""], ["x\0y", "z"], ["ĠĊč"] * 1000):
        text, off = D.flatten_tokens(tokens)
        want_text, want_off = walk(tokens)
        assert off.dtype == np.int32 and np.array_equal(off, want_off), tokens[:3]
        assert np.array_equal(text[:off[-1]], want_text[:want_off[-1]])


def test_partition_specification_properties():
    """oracle/partition_ref.py (the specification zett_partition_rows is held to on the GPU): a permutation with the requested
    group sizes, deterministic, fewer distinct ids per rank than contiguous shards and balanced packed positions — and
    rank_row_counts hands it exactly the rows plan_blocks gives each rank."""
    import numpy as np

    from oracle import partition_ref
    from zett_amd import synth
    from zett_amd.sharding import plan_blocks, rank_row_counts
    cfg, _, _, hist = synth.workload("mistral_gpt2_32k")
    rows = 12000
    ids = synth.make_surface_forms(cfg, rows, seed=3, hist=hist, n_special=2)
    pad, n_ids = cfg["pad_token_id"], cfg["original_vocab_size"] + cfg["hn_n_extra_tokens"]
    for world, chunks in ((8, 1), (4, 2), (3, 1)):
        caps = rank_row_counts(rows, world, chunks, min_rows_per_shard=512)
        assert sum(caps) == rows and caps == [sum(b.hi - b.lo for b in plan_blocks(rows, world, r, chunks, 512)) for r in range(world)]
        perm = partition_ref.partition_rows(ids, pad, n_ids, caps)
        assert sorted(perm.tolist()) == list(range(rows))
        assert np.array_equal(perm, partition_ref.partition_rows(ids, pad, n_ids, caps))
        off = np.concatenate([[0], np.cumsum(caps)])
        aff = partition_ref.shard_statistics(ids, pad, [perm[off[r]:off[r + 1]] for r in range(world)])
        con = partition_ref.shard_statistics(ids, pad, [np.arange(off[r], off[r + 1]) for r in range(world)])
        assert [s[0] for s in aff] == caps
        assert sum(s[2] for s in aff) < 0.93 * sum(s[2] for s in con)                 # fewer table rows in all
        assert max(s[1] for s in aff) < 1.05 * (sum(s[1] for s in aff) / world)       # positions near the mean


def test_host_classes_call_only_methods_they_define():
    """No GPU here, so the classes that only run on one are at least checked statically: every ``self.<name>(`` inside
    DeviceRetokenizer / HipEngine / RowGather names a method (or callable attribute) the class defines — an edit that drops a
    method (it happened: ``encode_joined``) fails here instead of on the GPU box."""
    import inspect
    import re

    from zett_amd import hypernet, sharding, surface_forms
    for cls, extra in ((surface_forms.DeviceRetokenizer, {"lib"}), (hypernet.HipEngine, {"lib"}), (sharding.RowGather, set())):
        src = inspect.getsource(cls)
        called = set(re.findall(r"self\.(\w+)\(", src))
        defined = {n for n, _ in inspect.getmembers(cls)} | extra
        for name in called - defined:
            # attributes set in __init__ that are called (streams, events, the ctypes library) do not count
            assert re.search(rf"self\.{name}\s*(:[^=]+)?=", src), f"{cls.__name__}.{name} is called but never defined"


def test_weights_stamp_sees_replaced_and_modified_parameters():
    """The engine cache key (ZettHypernet._weights_stamp) is computed from the parameters the modules hold NOW: an in-place write
    through the parameter, a new storage (p.data = t) and a REPLACED Parameter object (module.weight = nn.Parameter(t), the
    parametrize / PEFT pattern) all change it; nothing else does."""
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, _, _, _ = synth.workload("tiny")
    model = ZettHypernet(ZettHypernetConfig(**cfg))
    s0 = model._weights_stamp()
    assert model._weights_stamp() == s0
    with torch.no_grad():
        model.bias_projection.bias.add_(1.0)
    s1 = model._weights_stamp()
    assert s1 != s0 and model._weights_stamp() == s1
    model.bias_projection.bias.data = model.bias_projection.bias.data.clone()
    s2 = model._weights_stamp()
    assert s2 != s1
    old = model.scaler.w
    model.scaler.w = torch.nn.Parameter(old.detach().clone() * 2)          # a new Parameter OBJECT in the same slot
    s3 = model._weights_stamp()
    assert s3 != s2 and model._weights_stamp() == s3
    lin = model.get_submodule("output_projection.1")
    lin.register_parameter("weight", torch.nn.Parameter(lin.weight.detach().clone()))
    assert model._weights_stamp() != s3
