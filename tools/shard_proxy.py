#!/usr/bin/env python
"""GPU box: the single-GPU proxy of a P-GPU step (bench.side_config: rank 0's share of a workload's vocabulary, exchange excluded)
as a command of its own, for A/Bs and rocprofv3.

    python tools/shard_proxy.py --workload mistral_gpt2_32k --shard-of 8 --partition affinity [--steps 5 --warmup 2]
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mistral_gpt2_32k")
ap.add_argument("--shard-of", type=int, default=8)
ap.add_argument("--partition", default="contiguous", choices=["contiguous", "affinity"])
ap.add_argument("--precision", default="f16")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
a = ap.parse_args()
res = bench.side_config(a.workload, 0, a.precision, torch.device("cuda", 0), steps=a.steps, warmup=a.warmup, shard_of=a.shard_of, partition=a.partition)
r = res.pop("roofline")
res["gemm_ms_per_step"], res["non_gemm_ms_per_step"], res["frac"] = r["gemm_ms_per_step"], r["non_gemm_ms_per_step"], r["frac"]
print(json.dumps(res))
