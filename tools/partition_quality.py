#!/usr/bin/env python
"""CPU: what the id-affinity row partition (zett_partition_rows; specification: oracle/partition_ref.py) does to a rank's shard on
the BASELINE workloads — distinct source ids (rows of the hoisted input projection), distinct (id, position) pairs per packed
position (the pair lever is taken below 0.85) and packed positions per rank, against contiguous shards.

    python tools/partition_quality.py > profiles/r5_partition_quality.md
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import partition_ref  # noqa: E402
from zett_amd import synth  # noqa: E402

print("| workload | ranks | shards | distinct ids per rank (mean / max) | perfect split | pairs per position (mean) | packed positions per rank (min .. max) |")
print("|---|---|---|---|---|---|---|")
for name in ("mistral_gpt2_32k", "mistral_neox", "tinyllama_neox", "xlmr_gpt2", "llama3_256k"):
    cfg, rows, _, hist = synth.workload(name)
    ids = synth.make_surface_forms(cfg, rows, seed=0, hist=hist)
    pad, n_ids = cfg["pad_token_id"], cfg["original_vocab_size"] + cfg["hn_n_extra_tokens"]
    total = len(np.unique(ids[ids != pad]))
    for world in (2, 4, 8):
        per = -(-rows // world)
        caps = [max(0, min(per, rows - r * per)) for r in range(world)]
        off = np.concatenate([[0], np.cumsum(caps)])
        perm = partition_ref.partition_rows(ids, pad, n_ids, caps)
        for label, groups in (("contiguous", [np.arange(off[r], off[r + 1]) for r in range(world)]), ("affinity", [perm[off[r]:off[r + 1]] for r in range(world)])):
            st = partition_ref.shard_statistics(ids, pad, groups)
            print(f"| {name} ({rows}) | {world} | {label} | {np.mean([s[2] for s in st]):.0f} / {max(s[2] for s in st)} | {total // world} | "
                  f"{np.mean([s[3] / s[1] for s in st]):.3f} | {min(s[1] for s in st)} .. {max(s[1] for s in st)} |")
