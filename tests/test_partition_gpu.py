"""zett_partition_rows / zett_amd.sharding.affinity_order on the GPU: the kernel's permutation equals its numpy specification
(oracle/partition_ref.py) bit for bit, is a permutation with the requested group sizes, lowers the distinct ids per rank, and
the forward of the shards in that order reassembles to the single forward bit for bit (rows are independent: the reference
itself permutes them, scripts/transfer.py:54-67)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import partition_ref
from zett_amd import synth

pytestmark = pytest.mark.gpu


def _partition(ids, pad, n_ids, caps):
    from zett_amd import _lib
    lib = _lib.load()
    d = torch.from_numpy(ids).cuda()
    n, seq = ids.shape
    ws_bytes = C.c_int64(0)
    _lib.check(lib.zett_partition_workspace_bytes(n, n_ids, C.byref(ws_bytes)))
    ws = torch.empty((ws_bytes.value,), dtype=torch.uint8, device="cuda")
    perm = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    _lib.check(lib.zett_partition_rows(C.c_void_p(d.data_ptr()), n, seq, pad, n_ids, len(caps), (C.c_int32 * len(caps))(*caps),
                                       C.c_void_p(perm.data_ptr()), C.c_void_p(ws.data_ptr()), ws_bytes.value, 0,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "zett_partition_rows")
    torch.cuda.synchronize()
    return perm.cpu().numpy()


@pytest.mark.parametrize("name,rows,world", [("mistral_gpt2_32k", 32768, 8), ("mistral_gpt2_32k", 32768, 2), ("mistral_gpt2_32k", 5000, 3),
                                             ("xlmr_gpt2", 20011, 8), ("tiny", 96, 4), ("tiny", 1, 1), ("tiny", 2049, 8)])
def test_kernel_equals_its_specification(name, rows, world):
    cfg, _, _, hist = synth.workload(name)
    ids = synth.make_surface_forms(cfg, rows, seed=4, hist=hist, n_special=min(3, rows - 1) if rows > 1 else 0)
    pad = cfg["pad_token_id"]
    n_ids = cfg["original_vocab_size"] + cfg["hn_n_extra_tokens"]
    per = -(-rows // world)
    caps = [max(0, min(per, rows - r * per)) for r in range(world)]
    got = _partition(ids, pad, n_ids, caps)
    again = _partition(ids, pad, n_ids, caps)
    np.testing.assert_array_equal(got, again)                                   # deterministic
    assert sorted(got.tolist()) == list(range(rows))                            # a permutation
    np.testing.assert_array_equal(got, partition_ref.partition_rows(ids, pad, n_ids, caps))
    # uneven capacities (what plan_blocks hands out when the rows do not divide)
    if rows >= 4 * world:
        caps2 = list(caps)
        caps2[0] -= 1
        caps2[-1] += 1
        np.testing.assert_array_equal(_partition(ids, pad, n_ids, caps2), partition_ref.partition_rows(ids, pad, n_ids, caps2))


@pytest.mark.parametrize("seq,n_ids", [(16, 3000), (16, 200000), (70, 3000), (70, 200000)], ids=["seq16-lds", "seq16-global", "seq70-lds", "seq70-global"])
def test_long_rows_equal_the_specification(seq, n_ids):
    """Rows wider than the eight ids the kernel keeps in registers (PART_REG_IDS): the tail is read from memory un-prefetched, only a
    row's first 63 usable ids count in its score and its packed-position term, every id is marked.  Both homes of the rank bytes
    (LDS: n_ids <= 150 KiB; global memory beyond), rows with more than 63 usable ids included (seq 70)."""
    rng = np.random.default_rng(seq * 7 + n_ids)
    rows, pad, world = 5000, 1, 8
    ids = rng.integers(0, min(n_ids, 1500), size=(rows, seq), dtype=np.int64).astype(np.int32)          # a narrow id range: rows really share ids
    if n_ids > 1500:
        far = rng.random((rows, seq)) < 0.2
        ids[far] = rng.integers(1500, n_ids, size=int(far.sum()), dtype=np.int64).astype(np.int32)
    length = rng.integers(1, seq + 1, size=rows)
    length[rng.random(rows) < 0.1] = seq                                          # full-width rows (> 63 usable ids at seq 70)
    ids[np.arange(seq)[None, :] >= length[:, None]] = pad
    ids[rng.random((rows, seq)) < 0.02] = pad                                     # pads in the middle of a row
    per = -(-rows // world)
    caps = [max(0, min(per, rows - r * per)) for r in range(world)]
    got = _partition(ids, pad, n_ids, caps)
    assert sorted(got.tolist()) == list(range(rows))
    np.testing.assert_array_equal(got, partition_ref.partition_rows(ids, pad, n_ids, caps))


def test_partition_lowers_distinct_ids_per_rank():
    cfg, rows, _, hist = synth.workload("mistral_gpt2_32k")
    ids = synth.make_surface_forms(cfg, rows, seed=0, hist=hist)
    pad, n_ids = cfg["pad_token_id"], cfg["original_vocab_size"] + cfg["hn_n_extra_tokens"]
    caps = [rows // 8] * 8
    perm = _partition(ids, pad, n_ids, caps)
    aff = partition_ref.shard_statistics(ids, pad, np.split(perm, 8))
    con = partition_ref.shard_statistics(ids, pad, np.split(np.arange(rows), 8))
    assert max(s[2] for s in aff) < 0.72 * min(s[2] for s in con), (aff, con)            # 8 370 -> ~5 790 distinct ids per rank
    assert max(s[3] / s[1] for s in aff) < 0.85 < min(s[3] / s[1] for s in con)           # the pair lever's threshold is met again
    assert max(s[1] for s in aff) < 1.03 * min(s[1] for s in aff)                        # packed positions within ~2 % of each other (the balance term)


def test_bad_arguments():
    from zett_amd import _lib
    cfg, _, _, hist = synth.workload("tiny")
    ids = synth.make_surface_forms(cfg, 64, seed=1, hist=hist)
    with pytest.raises(ValueError):
        _partition(ids, 1, 305, [32, 31])                # capacities do not sum to the rows
    with pytest.raises(ValueError):
        _partition(ids, 1, 305, [8] * 8 + [0])           # more than eight ranks


def test_forward_in_affinity_order_reassembles_bit_for_bit():
    """Eight ranks' shards in the affinity order (emulated one after the other on this GPU, as tests/test_full_size_gpu.py does for
    contiguous shards), un-permuted by the indexed copy predict_sharded uses: identical to the single forward; and the per-rank
    plan finds fewer distinct ids and takes the pair lever."""
    from tests.test_invariants_gpu import _engine, _eq, _run
    from zett_amd.sharding import affinity_order, plan_blocks
    cfg, _, src_dtype, hist = synth.workload("tinyllama_neox")
    rows = 16000
    eng = _engine(cfg, 6, "f16")
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 6, dtype=src_dtype)).cuda()
    ids = synth.make_surface_forms(cfg, rows, seed=6, hist=hist, n_special=2)
    full = _run(eng, ids, src, -1)
    d_ids = torch.from_numpy(ids).cuda()
    n_ids = cfg["original_vocab_size"] + cfg["hn_n_extra_tokens"]
    for chunks in (1, 2):
        order = affinity_order(d_ids, 8, cfg["pad_token_id"], n_ids, chunks=chunks, min_rows_per_shard=512)
        assert sorted(order.tolist()) == list(range(rows))
        permuted = d_ids.index_select(0, order)
        out = [None if t is None else torch.empty_like(t) for t in full]
        ids_seen, ids_contig = [], []
        for r in range(8):
            for b in plan_blocks(rows, 8, r, chunks, min_rows_per_shard=512):
                part = eng.forward(permuted[b.lo:b.hi], src, -1)
                ids_seen.append(eng.stats()["distinct_ids"])
                for o, t in zip(out, part):
                    if o is not None:
                        o.index_copy_(0, order[b.lo:b.hi], t)
                eng.forward(d_ids[b.lo:b.hi], src, -1)
                ids_contig.append(eng.stats()["distinct_ids"])
        torch.cuda.synchronize()
        assert _eq(out, full), f"chunks {chunks}"
        assert sum(ids_seen) < 0.9 * sum(ids_contig), (sum(ids_seen), sum(ids_contig))      # (0.79 with one block per rank, 0.88 with two)
