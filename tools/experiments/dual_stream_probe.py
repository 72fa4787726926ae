"""Probe (r2): would running the two row halves of a vocabulary on two HIP streams — so that one half's LayerNorm /
attention / gathers overlap the other half's GEMMs — beat one forward over all rows?  Two engines (own workspaces), two
host threads (zett_forward synchronises once on its plan), each on its own stream; compared with one engine over all rows.
    python tools/experiments/dual_stream_probe.py [rows] [precision]
"""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from bench import device_weights  # noqa: E402
from zett_amd import synth  # noqa: E402
from zett_amd.dims import HypernetDims  # noqa: E402
from zett_amd.hypernet import HipEngine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
precision = sys.argv[2] if len(sys.argv) > 2 else "f16"
dev = torch.device("cuda:0")
cfg, _, src_dtype, hist = synth.workload("mistral_gpt2_32k")
dims = HypernetDims.from_config(cfg)
w = device_weights(cfg, dev, seed=0)
engines = [HipEngine(dims, 1e-5, dev, precision) for _ in range(3)]
for e in engines:
    e.load_weights(w)
src = torch.from_numpy(synth.make_source_embeddings(cfg, 0, dtype=src_dtype)).to(dev)
ids = torch.from_numpy(synth.make_surface_forms(cfg, rows, seed=0, hist=hist)).to(dev)
half = rows // 2
parts = [ids[:half].contiguous(), ids[half:].contiguous()]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def single():
    return engines[0].forward(ids, src, -1)


def dual():
    outs = [None, None]

    def run(i):
        with torch.cuda.stream(streams[i]):
            outs[i] = engines[1 + i].forward(parts[i], src, -1)
    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return outs


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a = single()
b = dual()
torch.cuda.synchronize()
same = all(torch.equal(x, torch.cat([p[k], q[k]])) for k, x in enumerate(a) for p, q in [b] if x is not None)
for rep in range(3):
    print(f"rows {rows} {precision}: one forward {timeit(single):.2f} ms   two halves on two streams {timeit(dual):.2f} ms   identical bits: {same}", flush=True)
