"""The RCCL code path on the hardware that exists: a ONE-rank `nccl` process group on cuda:0.

The build's GPU boxes have one device, so the world-2 tests of tests/test_multi_gpu.py skip there and every other N > 1
test runs over gloo.  A one-rank nccl group still loads librccl, creates a communicator and executes the very calls the
sharded path makes — `all_gather_into_tensor(async_op=True)` on RCCL's stream, the side-stream early start behind
zett_stream_wait_output, `record_stream`, `Work.wait()` — so a wrong argument, dtype, stream or lifetime shows up here
rather than on the first 8-GPU run.  (The direct fan-out has no peer at world 1: its grouped isend / irecv is covered by
the two-rank tests; what runs here is its local-copy branch.)  Reference: the row sharding of scripts/transfer.py:90-91.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, json
import torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
from bench import device_weights
from zett_amd import synth
from zett_amd.dims import HypernetDims
from zett_amd.hypernet import HipEngine
from zett_amd.sharding import RowGather, plan_blocks, predict_sharded
from zett_amd.transfer import reduce_flag_word
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
cfg, _, src_dtype, hist = synth.workload("tiny")
dims = HypernetDims.from_config(cfg)
eng = HipEngine(dims, 1e-5, dev, "f16")
eng.load_weights(device_weights(cfg, dev, seed=5))
src = torch.from_numpy(synth.make_source_embeddings(cfg, 5, dtype=src_dtype)).to(dev)
ids = torch.from_numpy(synth.make_surface_forms(cfg, 9001, seed=5, hist=hist, n_special=1)).to(dev)
predict = lambda rows: eng.forward(rows, src, 2)
single = predict(ids)
torch.cuda.synchronize()
ok, runs = True, 0
for chunks in (1, 2, 3):
    for mode in ("allgather", "fanout", "auto"):
        for ready in (None, eng.stream_wait_output):
            blocks = plan_blocks(ids.shape[0], 1, 0, chunks, min_rows_per_shard=1024)
            assert len(blocks) == chunks
            gather = RowGather(blocks, mode=mode)
            for b in blocks:
                gather.add(b, predict(ids[b.lo:b.hi]), ready)
            full = gather.finish(ids.shape[0], timed=True)
            torch.cuda.synchronize()
            ok = ok and gather.exposed_ms is not None and all(torch.equal(a, b) for a, b in zip(full, single))
            runs += 1
full = predict_sharded(predict, ids, chunks=2, ready=eng.stream_wait_output)
torch.cuda.synchronize()
ok = ok and all(torch.equal(a, b) for a, b in zip(full, single))
# the id-affinity order (zett_partition_rows) through the same exchange: every block scattered to its vocabulary rows behind
# its all-gather / fan-out, on the side streams (zett_scatter_rows) — early start or not, one block or three
from zett_amd.sharding import affinity_order
for chunks in (1, 3):
    order = affinity_order(ids, 1, cfg["pad_token_id"], cfg["original_vocab_size"] + cfg["hn_n_extra_tokens"], chunks=chunks, min_rows_per_shard=1024)
    order = order[torch.randperm(order.shape[0], device=dev)] if chunks == 3 else order          # (one rank: the partition is the identity; any order must work)
    for mode in ("allgather", "fanout"):
        for ready in (None, eng.stream_wait_output):
            full = predict_sharded(predict, ids, chunks=chunks, ready=ready, mode=mode, order=order, min_rows_per_shard=1024)
            torch.cuda.synchronize()
            ok = ok and all(torch.equal(a, b) for a, b in zip(full, single))
            runs += 1
    if chunks == 1:           # an order laid out for ANOTHER block plan is refused (the affinity would be lost silently otherwise)
        try:
            predict_sharded(predict, ids, chunks=2, order=order, min_rows_per_shard=1024); ok = False
        except ValueError:
            pass
# the flag-word reduction of the sharded CLI (MAX per bit: RCCL refuses ReduceOp.BOR)
ok = ok and reduce_flag_word(5, dev) == 5 and reduce_flag_word(0, dev) == 0
try:
    t = torch.tensor([1], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.BOR)
    bor = "accepted"
except Exception as e:
    bor = type(e).__name__
maps = open("/proc/self/maps").read()
rccl = sorted({{l.split()[-1] for l in maps.splitlines() if "rccl" in l.lower() or "nccl" in l.lower()}})
json.dump({{"ok": bool(ok), "runs": runs, "rccl": rccl, "bor": bor}}, open({out!r}, "w"))
dist.barrier(); dist.destroy_process_group()
"""


def test_row_gather_over_a_one_rank_rccl_group(tmp_path):
    out = os.path.join(tmp_path, "res.json")
    script = os.path.join(tmp_path, "worker.py")
    open(script, "w").write(_WORKER.format(repo=REPO, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 100), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, script], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.load(open(out))
    assert d["ok"] and d["runs"] == 26, d
    assert d["rccl"], "no RCCL library mapped into the process: the nccl backend did not run"


def test_bench_through_a_one_rank_rccl_group():
    """bench.py --gpus 1 --force-gather: the N > 1 step (row blocks, exchange on RCCL's stream, early start) with world = 1."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra in ([], ["--gather-mode", "fanout"], ["--serial-allgather", "--no-early-gather"]):
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--force-gather", "--steps", "3", "--warmup", "1",
               "--workload", "tiny", "--rows", "20001", "--no-cpu-baseline"] + extra
        out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 1 and d["exchange"]["backend"] == "nccl" and d["exchange"]["one_rank_group"]
        assert d["exchange"]["mode"] == ("fanout" if "fanout" in extra else "allgather")
        assert d["exchange_exposed_ms_per_step"] is not None and d["value"] > 0
