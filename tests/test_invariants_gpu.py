"""Size-independent properties of the HIP forward, checked bit-for-bit,
at small and at BASELINE.json's full sizes: rows are independent, so the result of a row may
not depend on its neighbours, on the shard / chunk it lands in, on the content behind pad
positions, or on the storage dtype of an fp16-representable source table."""
import numpy as np
import pytest
import torch

from tests import util
from zett_amd import synth
from zett_amd.dims import HypernetDims

pytestmark = pytest.mark.gpu


def _engine(cfg, seed, precision, device="cuda:0"):
    from bench import device_weights
    from zett_amd.hypernet import HipEngine
    dev = torch.device(device)
    eng = HipEngine(HypernetDims.from_config(cfg), 1e-5, dev, precision)
    eng.load_weights(device_weights(cfg, dev, seed=seed))
    return eng


def _run(eng, ids, src, lang=-1):
    out = eng.forward(torch.as_tensor(ids).to(eng.device), src, lang)
    torch.cuda.synchronize()
    return out


def _eq(a, b):
    return all((x is None and y is None) or torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("precision", ["bf16", "f16", "f32"])
def test_row_order_shard_and_chunk_independence_tiny(precision):
    cfg, *_ = synth.workload("tiny")
    eng = _engine(cfg, 1, precision)
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 1)).cuda()
    ids = synth.make_surface_forms(cfg, 3000, seed=1, n_special=3)
    full = _run(eng, ids, src, 2)
    perm = np.random.default_rng(0).permutation(len(ids))
    shuffled = _run(eng, ids[perm], src, 2)
    assert _eq([None if t is None else t[torch.from_numpy(np.argsort(perm)).cuda()] for t in shuffled], full)
    # vocab shards of 1, 2, 3, 8 ranks reassemble to the same bits (what the all-gather relies on)
    from zett_amd.sharding import shard_bounds
    for world in (2, 3, 8):
        parts = [_run(eng, ids[slice(*shard_bounds(len(ids), world, r))], src, 2) for r in range(world)]
        cat = [None if parts[0][k] is None else torch.cat([p[k] for p in parts]) for k in range(3)]
        assert _eq(cat, full)
    # encoder chunking
    eng.set_option("max_chunk_tokens", 1024)
    assert _eq(_run(eng, ids, src, 2), full)
    eng.set_option("max_chunk_tokens", 65536)
    # last layer computed for every position vs position 0 only
    eng.set_option("cls_only_last_layer", 0)
    ref = _run(eng, ids, src, 2)
    eng.set_option("cls_only_last_layer", 1)
    for a, b in zip(ref, full):
        if a is not None:
            torch.testing.assert_close(a, b, rtol=0, atol=0)
    # layer 0's Q/K/V once per distinct (source id, position) pair vs once per packed position
    assert _eq(_run(eng, ids, src, 2), full)
    st_on = eng.stats()
    assert st_on["distinct_positions"] < 0.85 * st_on["packed_tokens"], st_on          # (the lever is taken on this workload)
    eng.set_option("pair_dedupe", 0)
    ref = _run(eng, ids, src, 2)
    st_off = eng.stats()
    eng.set_option("pair_dedupe", 1)
    assert _eq(ref, full)
    assert st_off["distinct_positions"] == st_off["packed_tokens"] == st_on["packed_tokens"]
    assert st_off["executed_flops"] > st_on["executed_flops"]


@pytest.mark.parametrize("precision,name,rows", [("bf16", "tiny", 3000), ("f16", "tiny", 3000), ("f32", "tiny", 1500),
                                                  ("bf16", "tinyllama_neox", 6000), ("f16", "tinyllama_neox", 6000)])
def test_gemm_tile_variants_bit_identical(precision, name, rows):
    """The GEMM kernels (128x128, 256x256 register-staged eight-wave, 384x256 LDS-DMA, 256x256 four-wave
    direct-to-LDS with its streamlined epilogues (7) and with the generic epilogue drain (8)) share one K
    reduction order and one epilogue arithmetic: forcing any of them through zett_set_option("gemm_variant")
    must not change a single bit of the outputs."""
    cfg, _, src_dtype, hist = synth.workload(name)
    eng = _engine(cfg, 2, precision)
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 2, dtype=src_dtype)).cuda()
    ids = synth.make_surface_forms(cfg, rows, seed=2, hist=hist, n_special=2)
    lang = 1 if cfg.get("hn_embed_lang_id") else -1
    folded = _run(eng, ids, src, lang)
    # (the LayerNorm fold exists in the default tile only and rounds at other points than a LayerNorm launch does: the tile
    #  variants are compared with it off, the fold itself with the unfolded path to the tolerance of the arithmetic)
    eng.set_option("ln_fold", 0)
    auto = _run(eng, ids, src, lang)
    assert all(t is None or bool(torch.isfinite(t).all()) for t in auto)
    for variant in (1, 2, 3, 7, 8):
        eng.set_option("gemm_variant", variant)
        assert _eq(_run(eng, ids, src, lang), auto), f"gemm_variant {variant}"
    eng.set_option("gemm_variant", 0)
    eng.set_option("ln_fold", 1)
    lim = {"f32": 1e-5, "f16": 2e-3, "bf16": 1.2e-2}[precision]
    for a, b in zip(folded, auto):
        if a is not None:
            assert float((a - b).norm() / b.norm()) <= lim, precision
    for bad in (4, 5, 6, 9):
        with pytest.raises(ValueError):
            eng.set_option("gemm_variant", bad)


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("name,rows", [("tinyllama_neox", 1500), ("tinyllama_neox", 37), ("mistral_gpt2_32k", 700)])
def test_gemm_tail_split_bit_identical(precision, name, rows):
    """gemm4d's 128x256 HALF tile (r5: the partly filled last round of a launch, gemm4d.hip.h) runs the K order and the epilogue
    arithmetic of the 256x256 tile: with the split off (0), chosen per launch (1, default), forced in the middle of every launch
    (2) and with half tiles only (3) the outputs are the same bits — on the default path (LayerNorm fold, 16-bit residual
    stream in f16: the LN16 / LO_FOLD / LO_LN / F32_SCALE_FOLD epilogues) and with the fold off (F32 / F32_LN-free path: LO,
    BOTH, F32 with residual, F32_SCALE)."""
    cfg, _, src_dtype, hist = synth.workload(name)
    eng = _engine(cfg, 5, precision)
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 5, dtype=src_dtype)).cuda()
    ids = synth.make_surface_forms(cfg, rows, seed=5, hist=hist, n_special=2)
    lang = 1 if cfg.get("hn_embed_lang_id") else -1
    for fold in (1, 0):
        eng.set_option("ln_fold", fold)
        eng.set_option("gemm_tail_split", 0)
        base = _run(eng, ids, src, lang)
        assert all(t is None or bool(torch.isfinite(t).all()) for t in base)
        for mode in (1, 2, 3):
            eng.set_option("gemm_tail_split", mode)
            assert _eq(_run(eng, ids, src, lang), base), f"gemm_tail_split {mode}, ln_fold {fold}"
    # the pair lever on (layer 0's attention-output epilogue then fetches INDEXED residual rows: `res_index` in the HALF tile's LN16 /
    # F32_LN epilogues): ids folded into a small range so that (id, position) pairs repeat
    folded = np.where(ids == cfg["pad_token_id"], ids, 3 + ids % 257).astype(np.int32)
    assert cfg["pad_token_id"] < 3          # (3 + id % 257 never collides with the pad id of these workloads)
    eng.set_option("ln_fold", 1)
    eng.set_option("gemm_tail_split", 0)
    base = _run(eng, folded, src, lang)
    st = eng.stats()
    if rows >= 700:
        assert st["distinct_positions"] < st["packed_tokens"], st           # the lever is taken
    for mode in (2, 3):
        eng.set_option("gemm_tail_split", mode)
        assert _eq(_run(eng, folded, src, lang), base), f"pair lever, gemm_tail_split {mode}"
    with pytest.raises(ValueError):
        eng.set_option("gemm_tail_split", 4)


def test_repeated_launches_identical_bits_small_grids():
    """Race screen: few workgroups and long K (a 32-row batch of the TinyLlama-shape hypernet) leave
    the waves of a workgroup free to drift apart; a missing barrier in a K loop shows up as runs
    that differ from each other (it did, once: tools/determinism_check.py).  Every tile variant,
    12 launches each, all bits equal to the first launch and to the automatic choice."""
    cfg, _, src_dtype, hist = synth.workload("tinyllama_neox")
    eng = _engine(cfg, 3, "bf16")
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 3, dtype=src_dtype)).cuda()
    ids = synth.make_surface_forms(cfg, 32, seed=3, hist=hist, n_special=1)
    first = _run(eng, ids, src, -1)
    for it in range(12):         # the default path (LayerNorm fold on: gemm4d's producer / consumer epilogues)
        assert _eq(_run(eng, ids, src, -1), first), f"default path, launch {it}"
    eng.set_option("ln_fold", 0)
    auto = _run(eng, ids, src, -1)
    for variant in (0, 2, 3, 7, 8):
        eng.set_option("gemm_variant", variant)
        for it in range(12):
            assert _eq(_run(eng, ids, src, -1), auto), f"gemm_variant {variant}, launch {it}"
    eng.set_option("gemm_variant", 0)
    eng.set_option("ln_fold", 1)
    # the order in which workgroups take tiles is not allowed to matter either
    eng.set_option("gemm_tile_order", 1)
    big = synth.make_surface_forms(cfg, 3000, seed=4, hist=hist, n_special=1)
    alt = _run(eng, big, src, -1)
    eng.set_option("gemm_tile_order", 0)
    assert _eq(_run(eng, big, src, -1), alt)
    with pytest.raises(ValueError):
        eng.set_option("gemm_tile_order", 2)


def test_long_rows_pair_lever_bit_identical_and_vs_oracle():
    """Rows with more than 64 packed positions (hn_surface_maxlen >= 64 is legal: position_embeddings has 514 rows).
    The attention kernel hands a row's first 64 (source id, position) pair slots out by wave shuffle; slots past the
    64th must be read directly (r2 advisor finding: the shuffle index wrapped modulo 64).  Pair lever on == off bit
    for bit, both == the as-written oracle in f32, with an all-pad (uniform) 80-position row in the batch."""
    from oracle import hypernet_ref
    cfg, *_ = synth.workload("tiny")
    cfg = dict(cfg, hn_surface_maxlen=80)
    hist = tuple([1e-3] * 60 + [1.0] * 20)                       # lengths 61..80 (+ the language token: 62..81 positions)
    ids = synth.make_surface_forms(cfg, 400, seed=9, hist=hist, seq=80, n_special=2)
    assert (ids != cfg["pad_token_id"]).sum(1).max() > 64
    w = synth.make_weights(cfg, seed=9)
    src_np = synth.make_source_embeddings(cfg, 9)
    src = torch.from_numpy(src_np).cuda()
    from zett_amd.hypernet import HipEngine
    for precision in ("f32", "f16"):
        eng = HipEngine(HypernetDims.from_config(cfg), 1e-5, torch.device("cuda:0"), precision)
        eng.load_weights({k: torch.from_numpy(v).cuda() for k, v in w.items()})
        on = _run(eng, ids, src, 2)
        st = eng.stats()
        assert st["distinct_positions"] < st["packed_tokens"], st           # the lever is taken
        eng.set_option("pair_dedupe", 0)
        off = _run(eng, ids, src, 2)
        assert _eq(on, off), precision
        if precision == "f32":
            want = hypernet_ref.forward(w, cfg, ids[:48], src_np, lang_index=2)
            for g, r, what in zip(on, want, ("pred_in", "pred_out", "bias")):
                util.assert_f32_close(g[:48].cpu().numpy(), r, f"long rows {what}")
        eng.close()


def test_output_completion_events():
    """zett_stream_wait_output: pred_in is final after the first head's last GEMM and bias after the position-0 readout —
    a side stream that waits for those points (and for nothing else) reads the final values while the second head still
    runs.  What the vocabulary-sharded path starts its exchange behind (zett_amd/sharding.py RowGather)."""
    cfg, _, src_dtype, hist = synth.workload("tinyllama_neox")
    eng = _engine(cfg, 6, "f16")
    side = torch.cuda.Stream()
    with pytest.raises(RuntimeError):
        eng.stream_wait_output("in", side)                     # no forward has run on the handle yet
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 6, dtype=src_dtype)).cuda()
    ids = torch.from_numpy(synth.make_surface_forms(cfg, 9000, seed=6, hist=hist, n_special=1)).cuda()
    ref = _run(eng, ids, src)
    for it in range(3):
        out = eng.forward(ids, src, -1)                        # asynchronous: nothing has been waited for
        eng.stream_wait_output("bias", side)
        eng.stream_wait_output("in", side)
        with torch.cuda.stream(side):
            early_in, early_bias = out[0].clone(), out[2].clone()
        side.synchronize()
        assert torch.equal(early_in, ref[0]) and torch.equal(early_bias, ref[2]), it
        torch.cuda.synchronize()
        assert _eq(out, ref)
    # chunked forward: the points are those of the LAST chunk
    eng.set_option("max_chunk_tokens", 4096)
    out = eng.forward(ids, src, -1)
    eng.stream_wait_output("in", side)
    with torch.cuda.stream(side):
        early_in = out[0].clone()
    side.synchronize()
    assert eng.stats()["chunks"] >= 3 and torch.equal(early_in, ref[0])
    # an empty call completes at once
    eng.forward(ids[:0], src, -1)
    eng.stream_wait_output("in", side)
    side.synchronize()


def test_concat_last_hidden_state_with_one_position_is_the_cls_path():
    """hn_concat_last_hidden_state reshapes the hidden states to [N, L' * H]; for L' = 1 (the only case the reference's own
    heads can take: checked against it in the build container) that is hidden[:, 0]."""
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg = dict(synth.workload("tiny")[0], hn_embed_lang_id=False)
    w = synth.make_weights(cfg, seed=12)
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 12)).cuda()
    ids = torch.from_numpy(synth.make_surface_forms(cfg, 50, seed=12, seq=1)).cuda()
    outs = []
    for flag in (False, True):
        model = ZettHypernet(ZettHypernetConfig(**dict(cfg, hn_concat_last_hidden_state=flag)))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        model = model.to("cuda:0").eval()
        model.precision = "f32"
        outs.append(model(ids, source_embeddings=src))
    assert _eq(outs[0], outs[1])


def test_pad_content_independence():
    """Changing the pad token's source embedding must change nothing for rows with a visible key."""
    cfg, *_ = synth.workload("tiny")
    cfg = dict(cfg, hn_embed_lang_id=False)
    eng = _engine(cfg, 2, "f32")
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 2)).cuda()
    ids = synth.make_surface_forms(cfg, 512, seed=2)
    a = _run(eng, ids, src)
    src2 = src.clone()
    src2[cfg["pad_token_id"]] += 3.0
    b = _run(eng, ids, src2)
    uses_pad_as_token = (ids[:, 0] == cfg["pad_token_id"])
    keep = torch.from_numpy(~uses_pad_as_token).cuda()
    assert _eq([None if t is None else t[keep] for t in a], [None if t is None else t[keep] for t in b])


def test_fp16_source_equals_upcast():
    cfg, *_ = synth.workload("tiny")
    eng = _engine(cfg, 3, "f32")
    src16 = torch.from_numpy(synth.make_source_embeddings(cfg, 3, dtype="float16")).cuda()
    ids = synth.make_surface_forms(cfg, 256, seed=3)
    assert _eq(_run(eng, ids, src16, 1), _run(eng, ids, src16.float(), 1))
    assert _eq(_run(eng, ids, src16.float().bfloat16(), 1), _run(eng, ids, src16.float().bfloat16().float(), 1))


def test_index_errors_like_f_embedding():
    cfg, *_ = synth.workload("tiny")
    eng = _engine(cfg, 4, "bf16")
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 4)).cuda()
    ids = synth.make_surface_forms(cfg, 64, seed=4)
    bad = ids.copy()
    bad[10, 2] = cfg["original_vocab_size"] + max(cfg["hn_n_extra_tokens"], 1)     # one past the fallback rows
    with pytest.raises(IndexError):
        _run(eng, bad, src, 0)
    bad = ids.copy()
    bad[3, 0] = -1
    with pytest.raises(IndexError):
        _run(eng, bad, src, 0)
    assert _eq(_run(eng, ids, src, 0), _run(eng, ids, src, 0))     # the handle survives an error


def test_empty_and_single_row():
    cfg, *_ = synth.workload("tiny")
    eng = _engine(cfg, 5, "bf16")
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 5)).cuda()
    out = _run(eng, np.zeros((0, 7), dtype=np.int32), src, 0)
    assert out[0].shape == (0, 64) and out[2].shape == (0,)
    ids = synth.make_surface_forms(cfg, 9, seed=5)
    nine = _run(eng, ids, src, 0)
    one = _run(eng, ids[4:5], src, 0)
    assert _eq([None if t is None else t[4:5] for t in nine], one)


# (full-size shard / chunk / oracle-sample checks: tests/test_full_size_gpu.py)


def test_workspace_bytes_bounds_what_a_forward_reserves():
    """zett_workspace_bytes is an upper bound of the device memory a forward takes from HIP
    (measured as the drop of free HBM, minus the torch-owned outputs), grows with the batch
    and shrinks with max_chunk_tokens."""
    cfg, *_ = synth.workload("tiny")
    eng = _engine(cfg, 3, "bf16")
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 3)).cuda()
    ids = synth.make_surface_forms(cfg, 20000, seed=3)
    bound = eng.workspace_bytes(*ids.shape)
    assert bound > 0 and eng.workspace_bytes(2 * ids.shape[0], ids.shape[1]) > bound
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    reserved0 = torch.cuda.memory_reserved()
    out = _run(eng, ids, src, 2)
    free1, _ = torch.cuda.mem_get_info()
    torch_grew = torch.cuda.memory_reserved() - reserved0
    taken = (free0 - free1) - torch_grew
    assert taken <= bound + (8 << 20), (taken, bound)          # 8 MiB: allocator granularity of ~12 buffers
    eng.set_option("max_chunk_tokens", 4096)
    assert eng.workspace_bytes(*ids.shape) < bound
    with pytest.raises(Exception):
        eng.workspace_bytes(-1, 7)
    del out


@pytest.mark.parametrize("name,rows", [("tinyllama_neox", 3001), ("xlmr_gpt2", 2500)])
def test_folded_table_option(name, rows):
    """r6: with the 16-bit residual stream (f16) the hoisted input-projection table is kept as the LayerNorm-fold producer leaves it
    — the 16-bit pre-LayerNorm sum + (mean, rstd) per distinct id — and normalised by the embeddings' kernel as it reads it
    (zett_set_option "table_lo", default 1).  The fp32 table (0) is the A/B: both meet the oracle inside the f16 tolerance, they
    differ from each other by one 16-bit rounding of the table (rel-L2 well under the tolerance), shards and chunks of the folded
    table reproduce its single forward bit for bit (a table row's bits do not depend on which tile computed it), and bf16 mode —
    fp32 stream — ignores the option."""
    from oracle import hypernet_ref
    from zett_amd.sharding import shard_bounds
    cfg, _, src_dtype, hist = synth.workload(name)
    lang = 3 if cfg.get("hn_embed_lang_id") else -1
    src_np = synth.make_source_embeddings(cfg, 4, dtype=src_dtype)
    src = torch.from_numpy(src_np).cuda()
    ids = synth.make_surface_forms(cfg, rows, seed=4, hist=hist, n_special=2)
    eng = _engine(cfg, 4, "f16")
    folded = _run(eng, ids, src, lang)
    for world in (3, 8):
        parts = [_run(eng, ids[slice(*shard_bounds(len(ids), world, r))], src, lang) for r in range(world)]
        assert _eq([None if parts[0][k] is None else torch.cat([p[k] for p in parts]) for k in range(3)], folded)
    eng.set_option("max_chunk_tokens", 1024)
    assert _eq(_run(eng, ids, src, lang), folded)
    eng.set_option("max_chunk_tokens", 1 << 22)
    eng.set_option("table_lo", 0)
    plain = _run(eng, ids, src, lang)
    assert not _eq(plain, folded)
    for a, b in zip(plain, folded):
        if a is not None and a.dim() == 2:
            assert float((a - b).norm() / a.norm()) < 1e-3
    sample = np.arange(0, rows, 37)
    from bench import device_weights
    w = {k: v.float().cpu().numpy() for k, v in device_weights(cfg, torch.device("cuda:0"), seed=4).items()}
    hypernet_ref.set_matmul_backend("torch")
    want = hypernet_ref.forward(w, cfg, ids[sample], src_np, None if lang < 0 else lang)
    keep = ~util.all_pad_rows(cfg, ids[sample])
    for out in (folded, plain):
        for g, r in zip(out, want):
            if g is not None and r is not None:
                util.CLOSE["f16"](g[torch.from_numpy(sample).cuda()].cpu().numpy()[keep], r[keep], f"{name} table option")
    b16 = _engine(cfg, 4, "bf16")
    x = _run(b16, ids, src, lang)
    b16.set_option("table_lo", 0)
    assert _eq(_run(b16, ids, src, lang), x)


@pytest.mark.parametrize("hidden,heads,rows", [(768, 12, 2501), (128, 2, 1003), (256, 4, 515), (576, 9, 333), (2048, 32, 700), (320, 5, 257)])
def test_narrow_row_kernels_r6(hidden, heads, rows):
    """r6, the row kernels of the narrow hypernets.  (a) `attention_pack`: a last column group of 256 / 128 / 64 columns (H = 768,
    128, 256, 576) takes 2 / 4 / 8 rows per wave instead of idle lanes — the same arithmetic per element, so ON and OFF give the same
    bits, also with forced chunks (a chunk's last wave is partly filled) and in the position-0-only last layer; H = 2048 / 320 have
    no such group and the option changes nothing.  (b) `ln_rows8`: the 16-bit LayerNorm launches with eight columns per lane
    (H = 256·k <= 1024: 32 lanes per row; 512·k <= 2048: 64) add a row's values up in another order than the float4 kernel: both
    meet the oracle inside the f16 tolerance and differ from each other by rounding only; shards reproduce the single forward bit
    for bit with either."""
    from oracle import hypernet_ref
    from zett_amd.sharding import shard_bounds
    base = synth.workload("tiny")[0]
    cfg = dict(base, n_embd=128, hn_hidden_size=hidden, hn_intermediate_size=2 * hidden if hidden <= 1024 else 512, hn_num_attention_heads=heads, hn_n_layers=3)
    lang = 3 if cfg.get("hn_embed_lang_id") else -1
    src_np = synth.make_source_embeddings(cfg, 6)
    src = torch.from_numpy(src_np).cuda()
    ids = synth.make_surface_forms(cfg, rows, seed=6, n_special=2)
    for precision in ("f16", "bf16"):
        eng = _engine(cfg, 6, precision)
        ref = _run(eng, ids, src, lang)
        eng.set_option("attention_pack", 0)
        assert _eq(_run(eng, ids, src, lang), ref), "attention_pack changed bits"
        eng.set_option("attention_pack", 1)
        eng.set_option("max_chunk_tokens", 1024)
        assert _eq(_run(eng, ids, src, lang), ref)
        eng.set_option("max_chunk_tokens", 1 << 22)
        eng.set_option("attention_fast", 0)                  # the generic loop under the packed mapping
        slow = _run(eng, ids, src, lang)
        eng.set_option("attention_pack", 0)
        assert _eq(_run(eng, ids, src, lang), slow)
        eng.set_option("attention_pack", 1)
        eng.set_option("attention_fast", 1)
        parts = [_run(eng, ids[slice(*shard_bounds(len(ids), 3, r))], src, lang) for r in range(3)]
        assert _eq([None if parts[0][k] is None else torch.cat([p[k] for p in parts]) for k in range(3)], ref)
        eng.set_option("ln_rows8", 0)
        old = _run(eng, ids, src, lang)
        parts = [_run(eng, ids[slice(*shard_bounds(len(ids), 3, r))], src, lang) for r in range(3)]
        assert _eq([None if parts[0][k] is None else torch.cat([p[k] for p in parts]) for k in range(3)], old)
        for a, b in zip(old, ref):
            if a is not None and a.dim() == 2:
                assert float((a - b).norm() / a.norm()) < (2e-4 if precision == "f16" else 2e-3)
        if precision == "f16":
            from bench import device_weights
            w = {k: v.float().cpu().numpy() for k, v in device_weights(cfg, torch.device("cuda:0"), seed=6).items()}
            want = hypernet_ref.forward(w, cfg, ids, src_np, None if lang < 0 else lang)
            keep = ~util.all_pad_rows(cfg, ids)
            for out in (ref, old):
                for g, r in zip(out, want):
                    if g is not None and r is not None:
                        util.CLOSE["f16"](g.cpu().numpy()[keep], r[keep], f"H={hidden} narrow row kernels")


@pytest.mark.parametrize("name,rows", [("xlmr_gpt2", 2500), ("tinyllama_neox", 3001), ("mistral_gpt2_32k", 700)])
def test_shared_table_reproduces_the_forward(name, rows):
    """ABI 8 (SURVEY 8e's optional second exchange): the hoisted table computed in P slices of the GLOBAL distinct-id list
    (zett_table_plan / zett_table_rows — what P ranks would compute and exchange) and a forward on that table (zett_forward_table)
    give zett_forward's rows BIT FOR BIT: for the whole matrix and for row shards, whatever P, with forced chunks on both sides, and
    with the slices computed out of order.  The global list is the ascending list of referenced ids; the modes without the folded
    16-bit table (bf16, f32, table_lo 0) refuse."""
    from zett_amd.sharding import SharedTable, shard_bounds
    cfg, _, src_dtype, hist = synth.workload(name)
    lang = 3 if cfg.get("hn_embed_lang_id") else -1
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 5, dtype=src_dtype)).cuda()
    ids_np = synth.make_surface_forms(cfg, rows, seed=5, hist=hist, n_special=2)
    ids = torch.from_numpy(ids_np).cuda()
    eng = _engine(cfg, 5, "f16")
    ref = _run(eng, ids, src, lang)
    id_slot, id_list, n = eng.table_plan(ids)
    want_ids = np.unique(np.concatenate([ids_np[ids_np != cfg["pad_token_id"]], ids_np[:, 0]]))
    assert n == len(want_ids) and np.array_equal(id_list.cpu().numpy(), want_ids)
    assert np.array_equal(id_slot.cpu().numpy()[want_ids], np.arange(n)) and int(id_slot[-1]) == n
    for world in (1, 3, 8):
        table, stats = eng.table_buffers(n + 5)
        table.fill_(float("nan")); stats.fill_(float("nan"))
        for r in reversed(range(world)):
            lo, hi = shard_bounds(n, world, r)
            eng.table_rows(id_list, lo, hi - lo, src, table, stats)
        assert _eq(eng.forward_table(ids, table, stats, id_slot, lang), ref), f"world {world}"
        parts = [eng.forward_table(ids[slice(*shard_bounds(rows, world, r))], table, stats, id_slot, lang) for r in range(world)]
        assert _eq([None if parts[0][k] is None else torch.cat([p[k] for p in parts]) for k in range(3)], ref)
    eng.set_option("max_chunk_tokens", 1024)
    table2, stats2 = eng.table_buffers(n)
    eng.table_rows(id_list, 0, n, src, table2, stats2)
    assert torch.equal(table2.view(torch.int16), table[:n].view(torch.int16)) and torch.equal(stats2, stats[:n])
    assert _eq(eng.forward_table(ids, table2, stats2, id_slot, lang), ref)
    eng.set_option("max_chunk_tokens", 1 << 22)
    # the helper the sharded callers use, as rank 2 of 4 would build it (its own slice only) and as one rank builds all of it
    one = SharedTable(eng, ids, src)
    assert one.n_ids == n and _eq(one.predict(lang)(ids), ref)
    part = SharedTable(eng, ids, src, only_rank=2, world=4)
    lo, hi = part.rows
    assert (lo, hi) == (2 * part.per, min(3 * part.per, n)) and torch.equal(part.table[lo:hi].view(torch.int16), table[lo:hi].view(torch.int16))
    assert part.bytes_received() == 3 * part.per * (cfg["hn_hidden_size"] * 2 + 8)
    # a rank's share as two pieces (each all-gathered behind its computation in a real group): both ranks' pieces fill the same table
    pa = SharedTable(eng, ids, src, only_rank=0, world=2, pieces=2)
    pb = SharedTable(eng, ids, src, only_rank=1, world=2, pieces=2, buffers=(pa.table, pa.stats))
    cs = pa.piece_rows
    assert pa.ranges == [(min(0, n), min(cs, n)), (min(2 * cs, n), min(3 * cs, n))] and pb.ranges == [(min(cs, n), min(2 * cs, n)), (min(3 * cs, n), min(4 * cs, n))]
    assert torch.equal(pa.table[:n].view(torch.int16), table[:n].view(torch.int16)) and torch.equal(pa.stats[:n], stats[:n])
    assert _eq(pa.predict(lang)(ids), ref)
    # a plain forward behind a table forward is the plain forward again (the handle keeps nothing of the external table)
    assert _eq(_run(eng, ids, src, lang), ref)
    eng.set_option("table_lo", 0)
    with pytest.raises(ValueError):
        eng.table_rows(id_list, 0, n, src, table2, stats2)
    with pytest.raises(ValueError):
        eng.forward_table(ids, table2, stats2, id_slot, lang)
    b16 = _engine(cfg, 5, "bf16")
    with pytest.raises(ValueError):
        b16.table_rows(id_list, 0, n, src, *b16.table_buffers(n))
