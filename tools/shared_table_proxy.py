#!/usr/bin/env python
"""GPU box: rank 0's share of every BASELINE vocabulary at 8 GPUs (bench.side_config, exchange excluded) with and without the hoisted
table shared between the ranks (zett_amd.sharding.SharedTable, ABI 8) -> one JSON line per run (profiles/r6_shared_table.md).

    python tools/shared_table_proxy.py > gpurun_out/shared_table_proxies.jsonl
"""
import json, sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
out = []
for w, part, tx in [("mistral_gpt2_32k", "contiguous", False), ("mistral_gpt2_32k", "contiguous", True), ("mistral_gpt2_32k", "affinity", False), ("mistral_gpt2_32k", "affinity", True),
                    ("mistral_neox", "contiguous", False), ("mistral_neox", "contiguous", True), ("llama3_256k", "contiguous", False), ("llama3_256k", "contiguous", True),
                    ("tinyllama_neox", "contiguous", False), ("tinyllama_neox", "contiguous", True), ("xlmr_gpt2", "contiguous", False), ("xlmr_gpt2", "contiguous", True)]:
    r = bench.side_config(w, 0, "f16", dev, steps=5, warmup=2, shard_of=8, partition=part, table_exchange=tx)
    line = {k: r.get(k) for k in ("workload", "rows", "ms_per_step", "ms_per_step_uninstrumented", "distinct_source_ids", "distinct_id_position_pairs", "table_bytes_received", "unpermute_ms")}
    line["gemm_ms"] = r["roofline"]["gemm_ms_per_step"]; line["shared_table"] = tx
    print(json.dumps(line), flush=True)
