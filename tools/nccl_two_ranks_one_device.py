#!/usr/bin/env python
"""Probe (GPU box, one MI355X): can TWO processes sharing cuda:0 form an RCCL ("nccl") group, so that the direct fan-out of
zett_amd/sharding.py (batch_isend_irecv with a real peer) runs once without a second GPU?

    python tools/nccl_two_ranks_one_device.py            # tries the variants below, prints one JSON line per variant

Variants: the plain environment; NCCL_IGNORE_DISABLED_P2P=1; HIP_VISIBLE_DEVICES=0,0 (device aliasing: rank r uses logical
device r).  Each variant is a pair of worker processes under a 120 s timeout; a worker joins the group, runs one
all_gather_into_tensor and one RowGather in fan-out mode over a tiny matrix, and reports.  RCCL normally refuses two ranks on
one device ("Duplicate GPU detected"): the outcome — whatever it is — is what profiles/ records.  Reference: the row sharding
of scripts/transfer.py:90-91 / zett/utils.py:26.
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json, traceback
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
out = {"rank": rank, "GPU_MAX_HW_QUEUES_env": os.environ.get("GPU_MAX_HW_QUEUES")}
try:
    sys.path.insert(0, %(repo)r)
    import zett_amd
    out["hw_queues_set_by_hook"] = zett_amd.configure_hw_queues(log=False)
    out["GPU_MAX_HW_QUEUES_after_hook"] = os.environ.get("GPU_MAX_HW_QUEUES")
    out["proc_environ_has_it"] = b"GPU_MAX_HW_QUEUES=8" in open("/proc/self/environ", "rb").read() or os.environ.get("GPU_MAX_HW_QUEUES") == "8"
    import torch, torch.distributed as dist
    n_dev = torch.cuda.device_count()
    out["device_count"] = n_dev
    idx = rank if (os.environ.get("ZETT_PROBE_ALIAS") == "1" and n_dev > rank) else 0
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    dist.init_process_group("nccl", device_id=dev)
    out["init"] = "ok"
    x = torch.full((4, 8), float(rank + 1), device=dev)
    full = torch.empty((8, 8), device=dev)
    dist.all_gather_into_tensor(full, x)
    torch.cuda.synchronize()
    out["all_gather"] = bool(full[:4].eq(1).all() and full[4:].eq(2).all())
    from zett_amd.sharding import RowGather, plan_blocks
    blocks = plan_blocks(10, world, rank, 1, min_rows_per_shard=1)
    g = RowGather(blocks, mode="fanout")
    b = blocks[0]
    rows = torch.arange(b.lo, b.hi, device=dev, dtype=torch.float32)[:, None].repeat(1, 8)
    g.add(b, (rows, rows + 100, rows[:, 0].clone()))
    fin = g.finish(10)
    torch.cuda.synchronize()
    want = torch.arange(10, device=dev, dtype=torch.float32)
    out["fanout"] = bool(fin[0][:, 0].eq(want).all() and fin[1][:, 0].eq(want + 100).all() and fin[2].eq(want).all())
    dist.destroy_process_group()
except Exception as e:
    out["error"] = f"{type(e).__name__}: {str(e)[:400]}"
print("PROBE " + json.dumps(out), flush=True)
"""


def run_variant(name, extra_env, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GPU_MAX_HW_QUEUES", None)
    env.update(extra_env)
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"repo": REPO}], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = {"variant": name, "env": extra_env, "ranks": []}
    for p in procs:
        try:
            so, se = p.communicate(timeout=120)
            line = next((l[6:] for l in so.splitlines() if l.startswith("PROBE ")), None)
            res["ranks"].append(json.loads(line) if line else {"rc": p.returncode, "stderr_tail": se[-400:]})
        except subprocess.TimeoutExpired:
            p.kill()
            res["ranks"].append({"error": "timeout after 120 s (killed)"})
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    run_variant("plain", {}, 29711)
    run_variant("NCCL_IGNORE_DISABLED_P2P", {"NCCL_IGNORE_DISABLED_P2P": "1"}, 29712)
    run_variant("HIP_VISIBLE_DEVICES=0,0", {"HIP_VISIBLE_DEVICES": "0,0", "ZETT_PROBE_ALIAS": "1"}, 29713)
