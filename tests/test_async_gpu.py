"""How asynchronous the C ABI is (include/zett_hip.h, zett_forward_prepare / zett_retokenize_async).

zett_forward waits on the host for the PLAN of its call only.  With zett_forward_prepare the plan of the next forward is made
on the handle's own stream while the current forward runs, so a second zett_forward returns to the host while the kernels of
the first are still running — and produces the same bits as a forward issued alone.  The retokenizer's asynchronous entry point
returns without a device round trip; the truncation count and the errors arrive through zett_retok_result."""
import numpy as np
import pytest
import torch

from tests import util
from zett_amd import synth
from zett_amd.dims import HypernetDims

pytestmark = pytest.mark.gpu


def _engine(cfg, precision="f16"):
    from zett_amd.hypernet import HipEngine
    dev = torch.device("cuda:0")
    eng = HipEngine(HypernetDims.from_config(cfg), 1e-5, dev, precision)
    eng.load_weights({k: torch.from_numpy(v) for k, v in synth.make_weights(cfg, seed=5).items()})
    return eng


def _eq(a, b):
    return all((x is None and y is None) or torch.equal(x, y) for x, y in zip(a, b))


def test_second_forward_returns_while_the_first_still_runs():
    cfg, _, src_dtype, hist = synth.workload("tinyllama_neox")
    eng = _engine(cfg)
    dev = eng.device
    src = torch.from_numpy(synth.make_source_embeddings(cfg, seed=5, dtype=src_dtype)).to(dev)
    ids1 = torch.from_numpy(synth.make_surface_forms(cfg, 16384, seed=1, hist=hist)).to(dev)
    ids2 = torch.from_numpy(synth.make_surface_forms(cfg, 4096, seed=2, hist=hist)).to(dev)
    assert ids1.dtype == torch.int32 and ids2.dtype == torch.int32
    alone2 = eng.forward(ids2, src, -1)            # (also grows the workspace to its final size: no allocation below)
    eng.forward(ids1, src, -1)
    side = torch.cuda.Stream(device=dev)           # an idle stream: ids2 is complete, nothing to wait for
    eng.prepare(ids2, side)                        # (first use: the handle creates its plan stream and pinned buffers — that synchronises, once)
    eng.forward(ids2, src, -1)
    torch.cuda.synchronize()

    first_done = torch.cuda.Event()
    out1 = eng.forward(ids1, src, -1)              # ~8 ms of kernels; the host leaves after the plan
    first_done.record()
    eng.prepare(ids2, side)                        # the plan of the next forward runs NOW, beside forward 1
    out2 = eng.forward(ids2, src, -1)              # waits for that plan only
    still_running = not first_done.query()
    torch.cuda.synchronize()
    assert still_running, "zett_forward of the second call waited for the first forward's kernels"
    assert _eq(out2, alone2), "a forward behind a prepared plan must produce the bits of a forward issued alone"
    again1 = eng.forward(ids1, src, -1)
    torch.cuda.synchronize()
    assert _eq(out1, again1)


def test_unprepared_forward_waits_for_its_stream_and_a_stale_plan_is_discarded():
    cfg, _, src_dtype, hist = synth.workload("tiny")
    eng = _engine(cfg, "f32")
    dev = eng.device
    src = torch.from_numpy(synth.make_source_embeddings(cfg, seed=5, dtype=src_dtype)).to(dev)
    a = torch.from_numpy(synth.make_surface_forms(cfg, 200, seed=1, hist=hist)).to(dev)
    b = torch.from_numpy(synth.make_surface_forms(cfg, 300, seed=2, hist=hist)).to(dev)
    want_a, want_b = eng.forward(a, src, 2), eng.forward(b, src, 2)
    torch.cuda.synchronize()
    eng.prepare(a)                                 # prepared for a ...
    got_b = eng.forward(b, src, 2)                 # ... but b arrives: the prepared plan is dropped, b is planned on its own stream
    got_a = eng.forward(a, src, 2)
    eng.prepare(b); eng.prepare(a)                 # the second prepare replaces the first
    got_a2 = eng.forward(a, src, 2)
    torch.cuda.synchronize()
    assert _eq(got_b, want_b) and _eq(got_a, want_a) and _eq(got_a2, want_a)
    bad = a.clone(); bad[7, 0] = 10 ** 6           # an id outside the table: IndexError, synchronously, also behind a prepared plan
    eng.prepare(bad)
    with pytest.raises(IndexError):
        eng.forward(bad, src, 2)
    assert _eq(eng.forward(a, src, 2), want_a)


def test_retokenizer_async_matches_the_synchronous_call():
    from zett_amd.surface_forms import DeviceRetokenizer, HnTokenizerSpec
    cfg, _, _, hist = synth.workload("mistral_gpt2_32k")
    dev = torch.device("cuda:0")
    model, piece_of_id = synth.make_hn_model("mistral_gpt2_32k", cfg)
    spec = HnTokenizerSpec.from_model_json(model, ["<unk>", "<s>", "</s>"], [0, 1, 2], cfg["pad_token_id"])
    rt = DeviceRetokenizer(spec, dev)
    blocks = []
    for seed in (1, 2, 3):
        ids = synth.make_surface_forms(cfg, 3000 + 500 * seed, seed=seed, hist=hist)
        blocks.append((ids, rt.encode(synth.tokens_for_surface_forms(cfg, ids, piece_of_id))))
    outs = [rt.run_async(*enc, 7) for _, enc in blocks]          # three calls, no host wait in between
    assert rt.result() == 0
    for (ids, enc), out in zip(blocks, outs):
        assert torch.equal(out.cpu(), torch.from_numpy(ids))
        sync_out, n_tr = rt.run(*enc, 7)
        assert n_tr == 0 and torch.equal(sync_out, out)
    # truncation counts add up over the calls of one result(); maxlen 2 cuts every token of more than two pieces
    cut = [rt.run_async(*enc, 2) for _, enc in blocks]
    want = sum(int(((ids != cfg["pad_token_id"]).sum(1) > 2).sum()) for ids, _ in blocks)
    assert rt.result() == want
    for (ids, _), out in zip(blocks, cut):
        assert torch.equal(out.cpu(), torch.from_numpy(ids[:, :2]))
    # a character outside the byte table: KeyError at result(), naming the call
    d_text, d_off, n = rt.encode(["ab", "c d", "e"])             # the space is not a byte-level character
    rt.run_async(*blocks[0][1], 7)
    rt.run_async(d_text, d_off, n, 7)
    with pytest.raises(KeyError) as e:
        rt.result()
    assert "call 1, token 1" in str(e.value)
    assert rt.result() == 0                                      # nothing outstanding
