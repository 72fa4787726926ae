"""Data-parallel training use of the path (train.py's data-parallel step over a sampled vocabulary): each rank runs the
differentiable forward + backward on ITS row shard, torch's DistributedDataParallel averages the parameter gradients.
Launched by tests/test_autograd_gpu.py under torch.distributed.run; on a 1-GPU box both ranks share cuda:0 and the
all-reduce runs over gloo (ZETT_ONE_DEVICE=1), on a multi-GPU box it is nccl (= RCCL), one rank per GPU.
Rank 0 then repeats the step alone on all rows and prints the worst relative gradient difference as one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    from bench import device_weights
    from zett_amd import synth
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    from zett_amd.sharding import shard_bounds

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    one_device = os.environ.get("ZETT_ONE_DEVICE") == "1"
    dev = torch.device("cuda", 0 if one_device else int(os.environ.get("LOCAL_RANK", rank)))
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo" if one_device else "nccl")
    cfg, _, src_dtype, hist = synth.workload("tiny")
    rows = 3001
    model = ZettHypernet(ZettHypernetConfig(**cfg)).to(dev)
    params = dict(model.named_parameters())
    with torch.no_grad():
        for name, w in device_weights(cfg, dev, seed=0).items():
            params[name].copy_(w)
    model.requires_grad_(True).train()
    src = torch.from_numpy(synth.make_source_embeddings(cfg, 0, dtype=src_dtype)).to(dev)
    ids = torch.from_numpy(synth.make_surface_forms(cfg, rows, seed=0, hist=hist, n_special=2)).to(dev)
    lang = torch.tensor(3)
    g = torch.Generator(device="cpu").manual_seed(9)
    cot = [torch.randn(rows, cfg["n_embd"], generator=g).to(dev), torch.randn(rows, cfg["n_embd"], generator=g).to(dev), torch.randn(rows, generator=g).to(dev)]
    a, b = shard_bounds(rows, world, rank)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=None if one_device else [dev.index], find_unused_parameters=True)
    out = ddp(ids[a:b], source_embeddings=src, lang_index=lang)
    sum((o * c[a:b]).sum() for o, c in zip(out, cot)).backward()
    torch.cuda.synchronize()
    mean_of_shards = {n: p.grad.clone() for n, p in params.items() if p.grad is not None}
    dist.barrier()
    if rank == 0:
        model.zero_grad(set_to_none=True)
        out = model(ids, source_embeddings=src, lang_index=lang)
        sum((o * c).sum() for o, c in zip(out, cot)).backward()
        worst, where, n = 0.0, "", 0
        floor = 1e-6 * max(float(p.grad.double().norm()) for p in params.values() if p.grad is not None) / world
        for name, gsum in mean_of_shards.items():
            whole = params[name].grad.double() / world              # DDP averages over the ranks
            # (relative to the gradient's own size, with a floor: the key biases' gradients are round-off around an exact zero —
            # softmax does not see a shift common to a row's scores)
            rel = float((gsum.double() - whole).norm() / (whole.norm() + floor))
            worst, where = (rel, name) if rel > worst else (worst, where)
            n += 1
        print(json.dumps({"world": world, "backend": dist.get_backend(), "compared": n, "worst_rel": worst, "worst": where, "rows": rows}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
