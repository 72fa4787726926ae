#!/usr/bin/env python
"""Headline benchmark: predicted token-embeddings/s over a full target vocab.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--precision bf16|f32]

One "step" = one pass of the embedding-prediction hot path over the whole target
vocab.  At N = 1 that is one forward (retokenization -> plan -> hoisted input projection
-> packed encoder -> output heads) through libzett_hip.so.  At N > 1 (one process per
GPU) the vocab's rows are cut into row blocks, every block is sharded over the ranks,
and the RCCL all-gather that puts a block's rows on every GPU runs under the forward
of the next block (zett_amd/sharding.py) — all inside the step: nothing overlaps
across steps, every step ends with the full [V, E] matrices (+ bias) on every GPU.
Inputs (surface forms, source embeddings, weights) are resident in HBM before the
timed region.

Default workload = the north-star headline of BASELINE.json: Mistral-7B hypernetwork
shape (E 4096, E_in 8192, H 4096, I 8192, 32 heads, 2 output heads), 32 768-row
GPT-2-style target vocab, synthetic surface forms + seeded random weights (no network:
no real checkpoints or tokenizers exist on the box).

Rank 0 prints ONE JSON line.  `value` = vocab rows / wall time (whole job).
`roofline` prices the dominant kernel (the MFMA GEMM): algorithmic FLOPs of every
GEMM launch in the timed region (2*M*N*K, exactly what the launch computes) divided
by the launch durations measured with HIP events on the launch stream.
`cpu_baseline` times oracle/hypernet_ref.py (the as-written numpy port of the
reference forward) on this box's host cores on a bounded row sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from zett_amd import synth  # noqa: E402
from zett_amd.dims import HypernetDims, as_written_flops_per_row, weight_shapes  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md


def device_weights(cfg, device, seed=0):
    """Seeded random checkpoint generated directly in HBM (same recipe as synth.make_weights)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    for name, shape in weight_shapes(cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith("LayerNorm.weight") or name.endswith("ln.weight"):
            w = 1.0 + 0.05 * torch.randn(shape, device=device, generator=g)
        elif name.endswith("LayerNorm.bias") or name.endswith("ln.bias"):
            w = 0.02 * torch.randn(shape, device=device, generator=g)
        elif name.endswith("scaler.w"):
            w = 0.5 + 1.5 * torch.rand(shape, device=device, generator=g)
        elif name.endswith("scaler.b") or leaf == "bias":
            w = 0.01 * torch.randn(shape, device=device, generator=g)
        else:
            w = 0.02 * torch.randn(shape, device=device, generator=g)
        out[name] = w
    return out


def cpu_baseline(cfg, weights, ids, src, lang, budget_s=15.0):
    """Time the numpy oracle (as-written reference math) on a bounded row sample."""
    from oracle import hypernet_ref

    hypernet_ref.set_matmul_backend("torch")      # same as-written math, GEMMs through torch's CPU BLAS
    # pick the thread count that actually gives the best GEMM rate on this host (container CPU quotas
    # make os.cpu_count() a poor guide): short calibration on an [1024, 4096] x [4096, 4096] product
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    xa, xb = torch.randn(1024, 4096), torch.randn(4096, 4096)
    best_rate, threads = 0.0, 1
    for nt in sorted({t for t in (4, 8, 16, 32, 64, 128, avail) if t <= avail}):
        torch.set_num_threads(nt)
        torch.mm(xa, xb)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 0.4:
            torch.mm(xa, xb)
            reps += 1
        rate = reps / (time.perf_counter() - t0)
        if rate > best_rate * 1.05:
            best_rate, threads = rate, nt
    torch.set_num_threads(threads)
    w_np = {k: v.float().cpu().numpy() for k, v in weights.items()}
    src_np = src.cpu().numpy()
    def timed(fn, budget):
        probe = min(16, len(ids))
        fn(w_np, cfg, ids[:probe], src_np, lang)             # warm-up (BLAS thread pool, page faults)
        t0 = time.perf_counter()
        fn(w_np, cfg, ids[:4 * probe], src_np, lang)
        t_probe = (time.perf_counter() - t0) / 4
        rows = int(max(probe, min(len(ids), probe * budget / max(t_probe, 1e-6))))
        rows = max(probe, (rows // 16) * 16)
        t0 = time.perf_counter()
        out = fn(w_np, cfg, ids[:rows], src_np, lang)
        return rows, time.perf_counter() - t0, out

    # (i) the as-written reference math: `value`; (ii) the same CPU code with the exact algebraic levers the
    # HIP path uses (pad skipping, per-distinct-id input projection, CLS-only last layer): `levers_value`, so that
    # the GPU/CPU ratio is not credited with algorithmic gains
    rows, dt, out = timed(hypernet_ref.forward, 0.6 * budget_s)
    rows_l, dt_l, _ = timed(hypernet_ref.forward_levers, 0.4 * budget_s)
    return {"value": rows / dt, "unit": "token-embeddings/s", "cores": int(threads), "kind": "port",
            "sample": f"first {rows} rows of the workload, oracle/hypernet_ref.py (fp32, as-written reference math, "
                      f"GEMMs on torch CPU BLAS with {threads} threads), {dt:.1f} s",
            "levers_value": rows_l / dt_l,
            "levers_sample": f"first {rows_l} rows, oracle forward_levers (same code with the exact levers of DESIGN.md §2; "
                             f"on a sample the per-distinct-id table amortises less than on the whole vocab), {dt_l:.1f} s"}, out, rows


def measure_traffic_live(args, timeout_s=240):
    """HBM bytes per GEMM launch of THIS build on THIS box: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — separate
    passes, `--pmc` only with `--kernel-trace`, as MI355X_MICROARCH.md prescribes) over a child run of this script (one
    instrumented + one uninstrumented warm-up-free step of the same workload and precision), after the timed region.
    FETCH_SIZE is doubled (the gfx950 correction for wide coalesced reads), WRITE_SIZE taken as reported; both are KiB.
    Returns (bytes per launch, details) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    totals, launches = {}, {}
    work = tempfile.mkdtemp(prefix="zett_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(work, counter)
            cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "t", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--workload", args.workload, "--precision", args.precision,
                   "--no-cpu-baseline", "--no-alt-precision", "--no-live-traffic", "--no-side-configs"]
            # (the A/B switches of the parent run apply to the child too: the traffic is that of the kernels that were timed)
            for kv in args.option:
                cmd += ["--option", kv]
            for flag, val in (("--gemm-variant", args.gemm_variant), ("--gemm4d-min-k", args.gemm4d_min_k), ("--gemm-tile-order", args.gemm_tile_order),
                              ("--max-chunk-tokens", args.max_chunk_tokens)):
                if val:
                    cmd += [flag, str(val)]
            if args.no_pair_dedupe:
                cmd.append("--no-pair-dedupe")
            if args.no_ln_fold:
                cmd.append("--no-ln-fold")
            elif args.ln_fold != 1:
                cmd += ["--ln-fold", str(args.ln_fold)]
            env = dict(os.environ, TMPDIR=work)
            res = subprocess.run(cmd, cwd=work, env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if res.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {res.returncode}): {res.stderr[-300:]}"
            tot, n = 0.0, 0
            for path in files:
                for r in csv.DictReader(open(path)):
                    if r.get("Counter_Name") == counter and "gemm" in r.get("Kernel_Name", "") and "zett" in r.get("Kernel_Name", ""):
                        tot += float(r["Counter_Value"])
                        n += 1
            if n == 0:
                return None, f"no GEMM rows in the {counter} pass"
            totals[counter], launches[counter] = tot, n
    except Exception as e:          # a profiler that cannot run must not take the benchmark line with it
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(work, ignore_errors=True)
    per_launch = totals["FETCH_SIZE"] * 2 * 1024 / launches["FETCH_SIZE"] + totals["WRITE_SIZE"] * 1024 / launches["WRITE_SIZE"]
    return per_launch, {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over a child run of this bench.py after the timed region; "
                                  "FETCH_SIZE x2 (gfx950), KiB -> bytes; GEMM kernels, launch-weighted", "launches_counted": launches["FETCH_SIZE"],
                        "read_bytes_per_launch": totals["FETCH_SIZE"] * 2 * 1024 / launches["FETCH_SIZE"],
                        "write_bytes_per_launch": totals["WRITE_SIZE"] * 1024 / launches["WRITE_SIZE"]}


HBM_PEAK_TBS = 8.0            # HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a streaming copy achieves
HBM_ACHIEVABLE_TBS = 6.3
RIDGE_FLOP_PER_BYTE = 2500.0 / HBM_ACHIEVABLE_TBS      # ~400: below it a launch cannot reach the MFMA roof even at the achievable HBM rate


def launch_class(r):
    e = r["epilogue"]
    act = {0: "", 1: " + tanh-GELU", 2: " + erf-GELU"}[(e >> 8) & 3]
    if e & 16 and e & 64:
        return "LayerNorm-fold producer on the 16-bit residual stream (16-bit residual in, 16-bit out, row statistics)"
    if e & 16:
        return ("LayerNorm-fold producer (fp32 residual in, fp32 + 16-bit out, row statistics)" if e & 2 else
                "LayerNorm-fold producer of an output head (fp32 residual in, 16-bit out, row statistics)" + act)
    if e & 32:
        return "LayerNorm-fold consumer, 16-bit out" + act
    if e & 8:
        return "output head (Rescaler, fp32 out)"
    if e & 4:
        return "fp32 residual, fp32 out" + act
    if (e & 3) == 3:
        return "fp32 + 16-bit out" + act
    return ("16-bit out" if e & 1 else "fp32 out") + act


def class_table(classes, steps, peak):
    """by_class rows of the roofline object: per launch class, the FLOP / HIP-event-time ratio against the MFMA roof AND the
    algorithmic bytes / time ratio against HBM (launches of the narrow hypernets sit near the ridge, N*K/(N+K) ~ 400 FLOP/B:
    both fractions say what they are)."""
    out = []
    for k, v in sorted(classes.items(), key=lambda kv: -kv[1][1]):
        n, ms, fl, by = v
        tf = (fl / (ms * 1e-3) / 1e12) if ms > 0 else None
        tbs = (by / (ms * 1e-3) / 1e12) if ms > 0 else None
        out.append({"class": k, "launches_per_step": n / max(steps, 1), "ms_per_step": ms / max(steps, 1), "achieved": tf,
                    "frac": (tf / peak) if tf is not None else None,
                    "algorithmic_gb_per_launch": by / n / 1e9 if n else None,
                    "flop_per_byte": (fl / by) if by else None,
                    "hbm_tb_per_s": tbs, "hbm_frac": (tbs / HBM_PEAK_TBS) if tbs is not None else None})
    return out


def side_config(workload, rows, precision, device, steps=3, warmup=1, shard_of=0, partition="contiguous", table_exchange=False):
    """One more workload of BASELINE.json on this GPU, measured AFTER the timed region of the main line and reported under
    `configs` (never `value`): the same step — surface forms resident in HBM -> GPU retokenization -> hypernet forward — with
    per-launch HIP events, `steps` steps.  shard_of = P > 0: the rows are what rank 0 of P ranks computes of the workload's
    vocabulary (zett_amd.sharding.plan_blocks: the single-GPU proxy of the P-GPU step, exchange excluded).  partition =
    "affinity": the rank's rows are chosen by zett_amd.sharding.affinity_order instead of being a contiguous range — then every
    step retokenizes the WHOLE vocabulary (each rank needs every row's ids to compute the same order), runs the partition kernel,
    gathers its rows and runs the forward; the indexed copy that puts the gathered matrices back into vocabulary order is timed
    on full-size buffers and reported beside it (`unpermute_ms`: local HBM traffic, the same on 8 GPUs).  table_exchange (f16): the
    rank computes its 1/P slice of the hoisted table of the WHOLE vocabulary's distinct ids (zett_amd.sharding.SharedTable: whole-vocabulary
    retokenization, zett_table_plan, zett_table_rows) and runs its rows on the complete table (zett_forward_table; the peers' slices were
    computed once, untimed); the table's all-gather is NOT in the time, as no exchange is in any of these proxies — its size is reported."""
    from zett_amd.hypernet import HipEngine
    from zett_amd.sharding import affinity_order, plan_blocks
    from zett_amd.surface_forms import DeviceRetokenizer, HnTokenizerSpec

    affinity = bool(shard_of) and partition == "affinity"
    table_x = bool(shard_of) and table_exchange
    whole = affinity or table_x
    cfg, default_rows, src_dtype, hist = synth.workload(workload)
    vocab_rows = rows or default_rows
    dims = HypernetDims.from_config(cfg)
    lang = 3 if dims.embed_lang else -1
    peak = PEAK_TFLOPS[precision]
    engine = HipEngine(dims, 1e-5, device, precision)
    engine.load_weights(device_weights(cfg, device, seed=0))
    engine.set_option("time_gemm", 1)
    ids_all = synth.make_surface_forms(cfg, vocab_rows, seed=0, hist=hist)
    if shard_of:
        blocks = plan_blocks(vocab_rows, shard_of, 0, 1)
        ids_np = ids_all if whole else np.concatenate([ids_all[b.lo:b.hi] for b in blocks])
    else:
        ids_np = ids_all
    n = int(sum(b.hi - b.lo for b in blocks)) if shard_of else int(ids_np.shape[0])
    n_ids = dims.original_vocab_size + dims.n_extra
    seq_len = int(ids_np.shape[1])
    hn_model, piece_of_id = synth.make_hn_model(workload, cfg)
    spec = HnTokenizerSpec.from_model_json(hn_model, ["<unk>", "<s>", "</s>"], [0, 1, 2], dims.pad_token_id)
    retok = DeviceRetokenizer(spec, device)
    d_text, d_off, n_tok = retok.encode(synth.tokens_for_surface_forms(cfg, ids_np, piece_of_id))
    sfm0, n_trunc0 = retok.run(d_text, d_off, n_tok, seq_len)
    if n_trunc0 != 0 or not torch.equal(sfm0.cpu(), torch.from_numpy(ids_np)):
        raise SystemExit(f"{workload}: the retokenized surface forms differ from the workload's id matrix")
    # random source embeddings generated in HBM (the side lines have no CPU twin to agree with)
    g = torch.Generator(device=device)
    g.manual_seed(1)
    src = (0.02 * torch.randn((dims.original_vocab_size, dims.n_in_embd), device=device, generator=g)).to(getattr(torch, src_dtype))
    classes, acc = {}, {"gemm_ms": 0.0, "gemm_flops_timed": 0.0, "gemm_launches": 0}
    table_bufs, table_bytes = None, None
    if table_x:
        from zett_amd.sharding import SharedTable
        n_all, = engine.table_plan(sfm0)[2:]
        table_bufs = engine.table_buffers(n_all + shard_of)
        SharedTable(engine, sfm0, src, only_rank=0, world=1, buffers=table_bufs)          # the complete table, once, untimed: the peers' slices

    def forward_rows(sfm):
        """what rank 0 runs on the retokenized matrix `sfm` (the whole vocabulary's with an affinity order or a shared table)"""
        nonlocal table_bytes
        rows_sfm = sfm
        if affinity:          # the whole vocabulary's ids -> the order every rank computes -> this rank's rows
            order = affinity_order(sfm, shard_of, dims.pad_token_id, n_ids, chunks=1)
            rows_sfm = sfm.index_select(0, torch.cat([order[b.lo:b.hi] for b in blocks]))
        elif table_x:
            rows_sfm = sfm[blocks[0].lo:blocks[0].hi] if len(blocks) == 1 else torch.cat([sfm[b.lo:b.hi] for b in blocks])
        if table_x:           # this rank's slice of the shared table, then its rows on the complete table
            shared = SharedTable(engine, sfm, src, only_rank=0, world=shard_of, buffers=table_bufs)
            table_bytes = shared.bytes_received()
            return engine.forward_table(rows_sfm, shared.table, shared.stats, shared.id_slot, lang)
        return engine.forward(rows_sfm, src, lang)

    def one():
        out = forward_rows(retok.run_async(d_text, d_off, n_tok, seq_len))
        st_k = engine.stats()
        for key in acc:
            acc[key] += st_k[key]
        for r in engine.gemm_log():
            c = classes.setdefault(launch_class(r), [0, 0.0, 0.0, 0.0])
            c[0] += 1; c[1] += r["ms"]; c[2] += r["flops"]; c[3] += r["bytes"]
        return out

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    classes.clear()
    for key in acc:
        acc[key] = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if retok.result() != 0:
        raise SystemExit("a retokenized surface form was truncated")
    st = engine.stats()
    flags = engine.range_flags()
    # the same steps as a caller runs them: without the per-launch HIP events and the stream sync that reads them at the end of
    # every forward (on the narrow workloads the instrumentation is 2-3 % of a step)
    engine.set_option("time_gemm", 0)

    def plain():
        return forward_rows(retok.run_async(d_text, d_off, n_tok, seq_len))

    keep = None
    for _ in range(warmup):
        keep = plain()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(steps):
        keep = plain()
    torch.cuda.synchronize()
    ms_plain = (time.perf_counter() - t2) / steps * 1e3
    del keep
    retok.result()
    engine.set_option("time_gemm", 1)
    tf = acc["gemm_flops_timed"] / (acc["gemm_ms"] * 1e-3) / 1e12 if acc["gemm_ms"] > 0 else 0.0
    by = sum(v[3] for v in classes.values())
    tbs = by / (acc["gemm_ms"] * 1e-3) / 1e12 if acc["gemm_ms"] > 0 else 0.0
    unpermute_ms = None
    if affinity:
        # what follows the exchange: one indexed copy per output over the WHOLE vocabulary (local HBM traffic)
        order = affinity_order(retok.run_async(d_text, d_off, n_tok, seq_len), shard_of, dims.pad_token_id, n_ids, chunks=1)
        bufs = [torch.randn((vocab_rows, dims.n_embd), device=device) for _ in range(2 if dims.separate_out else 1)] + [torch.randn((vocab_rows,), device=device)]
        import ctypes as C
        from zett_amd import _lib
        lib = _lib.load()
        outs = [torch.empty_like(t) for t in bufs]

        def scatter_all():
            st = torch.cuda.current_stream(device).cuda_stream
            for t, o in zip(bufs, outs):
                rb = t[0].numel() * 4 if t.dim() > 1 else 4
                _lib.check(lib.zett_scatter_rows(C.c_void_p(t.data_ptr()), C.c_void_p(o.data_ptr()), C.c_void_p(order.data_ptr()), vocab_rows, rb,
                                                 device.index or 0, C.c_void_p(st)), "zett_scatter_rows")
        for _ in range(2):
            scatter_all()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            scatter_all()
        torch.cuda.synchronize()
        unpermute_ms = (time.perf_counter() - t1) / 5 * 1e3
        if not all(torch.equal(o.index_select(0, order), t) for t, o in zip(bufs, outs)):
            raise SystemExit("zett_scatter_rows: rows landed in the wrong place")
        retok.result()
        del bufs, outs
    res = {"workload": workload + (f" (rank 0 of {shard_of}: {n} of {vocab_rows} rows, {partition} shards)" if shard_of else ""), "rows": n, "dtype": precision,
           "partition": partition if shard_of else None, "unpermute_ms": unpermute_ms,
           "table_exchange": bool(table_x) if shard_of else None, "table_bytes_received": table_bytes,
           "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "ms_per_step_uninstrumented": ms_plain, "value": n * steps / dt, "unit": "token-embeddings/s",
           "packed_tokens": st["packed_tokens"], "distinct_source_ids": st["distinct_ids"], "distinct_id_position_pairs": st["distinct_positions"],
           "range_flags": flags,
           "roofline": {"bound": "mfma" if (acc["gemm_flops_timed"] / by if by else 1e9) >= RIDGE_FLOP_PER_BYTE else "mfma/hbm (launch intensity under the ridge)",
                        "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                        "flop_per_byte": (acc["gemm_flops_timed"] / by) if by else None, "ridge_flop_per_byte": RIDGE_FLOP_PER_BYTE,
                        "hbm_tb_per_s": tbs, "hbm_frac": tbs / HBM_PEAK_TBS, "hbm_frac_of_achievable": tbs / HBM_ACHIEVABLE_TBS,
                        "gemm_ms_per_step": acc["gemm_ms"] / steps, "non_gemm_ms_per_step": dt / steps * 1e3 - acc["gemm_ms"] / steps,
                        "by_class": class_table(classes, steps, peak)}}
    engine.close()
    retok.close()
    del src, engine
    torch.cuda.empty_cache()
    return res


def api_path(workload, precision, device, reps=3):
    """Throughput through the KEPT API (README.md:91-121 of the reference): a Python list[str] of byte-level target tokens ->
    zett_amd.get_surface_form_matrix (zett/utils.py:651-689: returns a numpy matrix) -> ZettHypernet.__call__(torch.from_numpy(m),
    source_embeddings=...) (hf_hypernet/modeling_hypernet.py:156-163), wall per vocabulary, host work, H2D / D2H copies and the
    f16 range check included.  Also the documented fast path that leaves the matrix on the GPU (surface_form_matrix_device).
    A side measurement after the timed region; never `value`."""
    import zett_amd
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    from zett_amd.surface_forms import HnTokenizerSpec, surface_form_matrix_device

    cfg, rows, src_dtype, hist = synth.workload(workload)
    dims = HypernetDims.from_config(cfg)
    with torch.device(device):
        model = ZettHypernet(ZettHypernetConfig(**cfg))
    model.load_state_dict(device_weights(cfg, device, seed=0))
    model = model.to(device).eval()
    model.precision = precision
    ids_np = synth.make_surface_forms(cfg, rows, seed=0, hist=hist)
    hn_model, piece_of_id = synth.make_hn_model(workload, cfg)
    spec = HnTokenizerSpec.from_model_json(hn_model, ["<unk>", "<s>", "</s>"], [0, 1, 2], dims.pad_token_id)
    tokens = synth.tokens_for_surface_forms(cfg, ids_np, piece_of_id)
    maxlen = int(ids_np.shape[1])
    g = torch.Generator(device=device)
    g.manual_seed(1)
    src = (0.02 * torch.randn((dims.original_vocab_size, dims.n_in_embd), device=device, generator=g)).to(getattr(torch, src_dtype))
    lang = torch.tensor(3) if dims.embed_lang else None

    def numpy_api():
        t0 = time.perf_counter()
        m, n_trunc = zett_amd.get_surface_form_matrix(tokens, maxlen, tokenizer_to_use=spec)
        t1 = time.perf_counter()
        with torch.no_grad():
            out = model(torch.from_numpy(m), source_embeddings=src, lang_index=lang)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, m, out

    def device_api():
        t0 = time.perf_counter()
        m, n_trunc = surface_form_matrix_device(tokens, maxlen, spec, device)
        with torch.no_grad():
            out = model(m, source_embeddings=src, lang_index=lang)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    numpy_api(); device_api()                                   # warm-up: engine build + weight upload, retokenizer tables
    runs = [numpy_api() for _ in range(reps)]
    if not np.array_equal(runs[-1][3], ids_np):
        raise SystemExit("api_path: get_surface_form_matrix differs from the workload's id matrix")
    dev = [device_api() for _ in range(reps)]
    best = min(runs, key=lambda r: r[0])
    res = {"workload": workload, "rows": rows, "dtype": model.precision, "reps": reps,
           "ms": best[0], "ms_get_surface_form_matrix": best[1], "ms_hypernet_call": best[2],
           "ms_all_reps": [r[0] for r in runs],
           "device_matrix_ms": min(dev), "device_matrix_ms_all_reps": dev,
           "what": "list[str] -> zett_amd.get_surface_form_matrix (numpy out) -> ZettHypernet.__call__(torch.from_numpy(m), source_embeddings=...) "
                   "-> torch.cuda.synchronize(); device_matrix_ms: surface_form_matrix_device (matrix stays in HBM) -> ZettHypernet.__call__; "
                   "best of `reps`, after one warm-up call of each"}
    model._drop_engines()
    del model, src
    torch.cuda.empty_cache()
    return res


def train_step_line(workload, rows, precision, device, steps=2, warmup=1):
    """Training-time use of the path (reference call sites train.py:1007-1013, 1191-1197): one step = the
    differentiable forward of `rows` rows + the backward to every hypernetwork parameter, through zett_amd/autograd.py (packed
    schedule, `precision` MFMA operands with fp32 accumulation, everything else fp32).  A side measurement after the timed region;
    never `value`.  tools/train_bench.py is the same measurement as a command."""
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet

    cfg, _, src_dtype, hist = synth.workload(workload)
    dims = HypernetDims.from_config(cfg)
    with torch.device(device):
        model = ZettHypernet(ZettHypernetConfig(**cfg))
    model.load_state_dict(device_weights(cfg, device, seed=0))
    model = model.to(device)
    model.requires_grad_(True).train()
    model.train_packed = True
    model.train_precision = precision
    g = torch.Generator(device=device)
    g.manual_seed(1)
    src = (0.02 * torch.randn((dims.original_vocab_size, dims.n_in_embd), device=device, generator=g)).to(getattr(torch, src_dtype))
    ids = torch.from_numpy(synth.make_surface_forms(cfg, rows, seed=0, hist=hist)).to(device)
    lang = torch.tensor(3) if dims.embed_lang else None
    cot = None
    t_f = t_b = 0.0

    def step():
        nonlocal cot, t_f, t_b
        model.zero_grad(set_to_none=True)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        out = model(ids, source_embeddings=src, lang_index=lang)
        e1.record()
        if cot is None:
            cot = [None if o is None else torch.randn(o.shape, device=device, generator=g) for o in out]
        sum((o * c).sum() for o, c in zip(out, cot) if o is not None).backward()
        e2.record()
        torch.cuda.synchronize()
        t_f += e0.elapsed_time(e1)
        t_b += e1.elapsed_time(e2)

    torch.cuda.reset_peak_memory_stats(device)
    for _ in range(warmup):
        step()
    t_f = t_b = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    finite = all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for n, p in model.named_parameters() if n != "model.embeddings.word_embeddings.weight")
    res = {"workload": workload, "rows": rows, "dtype": precision, "schedule": "packed (levers 1-3)", "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "forward_ms": t_f / steps, "backward_ms": t_b / steps, "rows_per_s": rows * steps / dt,
           "peak_memory_gb": torch.cuda.max_memory_allocated(device) / 1e9, "all_parameter_gradients_finite": finite,
           "what": "differentiable forward + backward to every hypernetwork parameter (zett_amd/autograd.py), HIP events around the two halves"}
    model._drop_engines()
    del model, src, ids, cot
    torch.cuda.empty_cache()
    return res


LINE_LIMIT = 6000          # bytes of the ONE stdout line (the driver keeps an ~8 KB tail of stdout; round 5's 22 KB line was cut)
SIDE_FILE = "bench_side.json"

CLASS_CODE = (("LayerNorm-fold producer on the 16-bit residual stream", "ln16_producer"), ("LayerNorm-fold producer of an output head", "head_ln_producer"),
              ("LayerNorm-fold producer", "ln_producer"), ("LayerNorm-fold consumer", "ln_consumer"), ("output head", "head_rescaler"),
              ("fp32 residual, fp32 out", "f32_residual"), ("fp32 + 16-bit out", "f32_and_16"), ("16-bit out", "out16"), ("fp32 out", "out32"))


def _short_class(name):
    code = next((c for prefix, c in CLASS_CODE if name.startswith(prefix)), name[:24])
    return code + ("+tanh" if "tanh-GELU" in name else "+erf" if "erf-GELU" in name else "")


def _r(x, sig=5):
    """Floats to `sig` significant digits (the line is a record, not an archive: the full floats are in bench_side.json)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{sig}g}")


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d}


def _cached_cpu_baselines():
    """Per-config CPU baselines (a >= 1 024-row slice per config) are measured once per round with
    tools/cpu_baselines.py on a GPU box's host and cached in profiles/cpu_baselines.json, labelled with box and cores;
    only the headline's is re-timed inside every bench run."""
    try:
        return json.load(open(os.path.join(REPO, "profiles", "cpu_baselines.json")))
    except Exception:
        return {}


def compact_line(result, limit=LINE_LIMIT):
    """The ONE stdout line from the full result object: every field the contract names, `roofline` and `cpu_baseline` whole in
    their contract fields, and a few numbers per side measurement.  Prose (`what`, `note`, methods), per-class tables of the side
    configs and full-precision floats go to bench_side.json only.  If the line is still over `limit` bytes, the optional parts are
    dropped, least important first; the contract fields are never dropped."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: _r(result.get(k)) for k in top}
    cfg = result.get("config") or {}
    out["config"] = {k: (v if not isinstance(v, str) else v[:200]) for k, v in cfg.items() if k in
                     ("workload", "rows", "rows_per_gpu", "partition", "table_exchange", "parallelism", "precision", "packed_tokens_rank0", "distinct_source_ids_rank0",
                      "distinct_id_position_pairs_rank0", "hn_tokenizer") and v is not None}
    if isinstance(out["config"].get("parallelism"), str):
        out["config"]["parallelism"] = out["config"]["parallelism"].split(" (")[0]
    rf = result.get("roofline") or {}
    out["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "algorithmic_bytes_per_launch",
                                  "launches_per_step", "gemm_ms_per_step", "executed_tflop_per_step"))
    out["roofline"]["kernel"] = "zett::gemm4d_tn_kernel (+gemm8r/384x256/128x128 tiles): all GEMM launches, FLOP-weighted"
    if "ONE untimed" in str(rf.get("source", "")):          # an N > 1 line: the timed steps carry no per-launch events
        out["roofline"]["source"] = "one untimed instrumented pass over rank 0's row blocks after the timed region"
    ts = rf.get("traffic_source") or {}
    out["roofline"]["traffic_live"] = bool(ts.get("live")) if ts else None
    if ts.get("stale"):
        out["roofline"]["traffic_stale"] = True
    out["roofline"]["by_class"] = [{"class": _short_class(c["class"]), "n": _r(c.get("launches_per_step")), "ms": _r(c.get("ms_per_step"), 4),
                                    "frac": _r(c.get("frac"), 3)} for c in rf.get("by_class") or []]
    cb = result.get("cpu_baseline")
    out["cpu_baseline"] = None if not cb else dict(_pick(cb, ("value", "unit", "cores", "kind", "levers_value")), sample=str(cb.get("sample", ""))[:160])
    for k in ("parity_vs_cpu_port_rel_l2", "ms_per_step_uninstrumented", "value_uninstrumented", "range_flags", "as_written_tflops", "as_written_gflop_per_row",
              "exchange_exposed_ms_per_step"):
        if result.get(k) is not None:
            out[k] = _r(result[k])
    ex = result.get("exchange")
    if ex:
        out["exchange"] = _pick(ex, ("mode", "backend", "one_rank_group", "early_start_of_pred_in_and_bias", "bytes_received_per_rank_per_step", "table_bytes_received_per_rank_per_step"))
        out["exchange"]["GPU_MAX_HW_QUEUES"] = os.environ.get("GPU_MAX_HW_QUEUES")
    if result.get("alt_precision"):
        out["alt_precision"] = _pick(result["alt_precision"], ("dtype", "value", "ms_per_step", "roofline_frac"))
    if result.get("f32_mode"):
        out["f32_mode"] = _pick(result["f32_mode"], ("dtype", "value", "ms_per_step", "roofline_frac", "peak"))
    cached = _cached_cpu_baselines()
    if result.get("configs") is not None:
        out["configs"] = []
        for c in result["configs"]:
            if "error" in c:
                out["configs"].append({"workload": c.get("workload"), "error": str(c["error"])[:120]})
                continue
            r = c.get("roofline") or {}
            name = str(c.get("workload", "")).split(" (")[0]
            e = {"workload": name, "rows": c.get("rows"), "ms_per_step": _r(c.get("ms_per_step"), 4), "frac": _r(r.get("frac"), 3), "hbm_frac": _r(r.get("hbm_frac"), 3),
                 "gemm_ms": _r(r.get("gemm_ms_per_step"), 4)}
            if c.get("ms_per_step_uninstrumented") is not None:          # (the same steps without the per-launch HIP events)
                e["ms_uninstrumented"] = _r(c["ms_per_step_uninstrumented"], 4)
            if c.get("partition"):
                e["shard"] = f"rank 0 of 8, {c['partition']}" + (", shared table" if c.get("table_exchange") else "")
                if c.get("table_bytes_received"):
                    e["table_mb_received"] = _r(c["table_bytes_received"] / 1e6, 4)
                if c.get("unpermute_ms") is not None:
                    e["unpermute_ms"] = _r(c["unpermute_ms"], 3)
            elif name in cached.get("configs", {}):
                cc = cached["configs"][name]
                e["cpu_baseline"] = {"value": _r(cc.get("value"), 4), "cores": cc.get("cores"), "kind": cc.get("kind"), "cached": cached.get("box", "profiles/cpu_baselines.json")[:60]}
            out["configs"].append(e)
    if result.get("api_path") is not None:
        out["api_path"] = [({"workload": a.get("workload"), "error": str(a["error"])[:120]} if "error" in a else
                            dict(_pick(a, ("workload", "ms", "ms_per_step_engine", "over_engine_step", "device_matrix_ms")))) for a in result["api_path"]]
    if result.get("train_step") is not None:
        t = result["train_step"]
        out["train_step"] = {"error": str(t["error"])[:120]} if "error" in t else _pick(t, ("rows", "dtype", "ms_per_step", "forward_ms", "backward_ms", "peak_memory_gb"))
    out["full"] = SIDE_FILE
    for drop in (None, ("roofline", "by_class"), ("train_step",), ("api_path",), ("f32_mode",), ("alt_precision",), ("configs",), ("config", "hn_tokenizer"),
                 ("config", "precision"), ("cpu_baseline", "sample")):
        if drop:
            node = out
            for k in drop[:-1]:
                node = node.get(k) or {}
            node.pop(drop[-1], None)
        line = json.dumps(out, separators=(",", ":"))
        if len(line.encode()) <= limit:
            break
    return line


def emit(result):
    """Full object -> bench_side.json (stderr too with ZETT_BENCH_ECHO_FULL=1), compact line -> stdout (the only stdout line)."""
    full = json.dumps(result)
    try:
        with open(os.path.join(os.environ.get("ZETT_BENCH_SIDE_DIR", REPO), SIDE_FILE), "w") as f:
            f.write(full + "\n")
    except OSError as e:
        print(f"bench.py: could not write {SIDE_FILE}: {e}", file=sys.stderr)
    if os.environ.get("ZETT_BENCH_ECHO_FULL") == "1":
        print(full, file=sys.stderr, flush=True)
    print(compact_line(result), flush=True)


def self_launch(gpus):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` (one rank per GPU; rank 0 prints the line)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("bench.py: --gpus %d without a launcher: re-executing under torch.distributed.run (port %d)" % (gpus, port), file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="mistral_gpt2_32k", choices=sorted(synth.WORKLOADS))
    ap.add_argument("--precision", default="f16", choices=["bf16", "f16", "f32"],
                    help="arithmetic of the dense contractions; f16 is the library default (zett_amd/hypernet.py DEFAULT_PRECISION)")
    ap.add_argument("--rows", type=int, default=0, help="override the vocab size of the workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--max-chunk-tokens", type=int, default=0, help="A/B only: zett_set_option max_chunk_tokens (0 = library default)")
    ap.add_argument("--gemm4d-min-k", type=int, default=0, help="A/B only: K threshold of the four-wave direct-to-LDS tile (zett_set_option gemm4d_min_k)")
    ap.add_argument("--gemm-tile-order", type=int, default=0, help="A/B only: zett_set_option gemm_tile_order (0 = default)")
    ap.add_argument("--gemm-variant", type=int, default=0, help="A/B only: force one GEMM tile variant (zett_set_option gemm_variant); 0 = per-launch choice")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the side measurement of the same steps in the other 16-bit arithmetic (N = 1; reported as alt_precision, never as value)")
    ap.add_argument("--no-pair-dedupe", action="store_true", help="A/B only: layer 0's Q/K/V per packed position instead of per distinct (source id, position) pair (zett_set_option pair_dedupe 0; same bits)")
    ap.add_argument("--ln-fold", type=int, default=1, help="A/B only: zett_set_option ln_fold (1 = encoder and output heads, default; 2 = encoder only; 0 = off)")
    ap.add_argument("--no-ln-fold", action="store_true", help="A/B only: the encoder's LayerNorms as launches instead of folded into the GEMMs around them (zett_set_option ln_fold 0)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 PMC passes that measure roofline.traffic after the timed region (N = 1, default workload sizes); "
                    "the figure then comes from profiles/pmc_traffic.json if that still matches the HIP sources, else null")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the side measurements of the other 1-GPU BASELINE configs, the 8-GPU shard proxy (`configs`) "
                    "and of the kept Python API (`api_path`) after the timed region")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE", help="A/B only: any zett_set_option key (repeatable), applied last")
    ap.add_argument("--no-retokenize", action="store_true", help="A/B only: start every step from the id matrix instead of the surface forms")
    ap.add_argument("--chunks", type=int, default=2, help="N > 1: row blocks per step (zett_amd/sharding.py: the all-gather of a block overlaps the next block's forward)")
    ap.add_argument("--serial-allgather", action="store_true", help="N > 1: one block per step, i.e. forward, then all-gather (A/B)")
    ap.add_argument("--force-gather", action="store_true", help="N = 1: join a ONE-rank RCCL group (backend nccl) and run every step through the row exchange "
                    "(RowGather: all_gather_into_tensor on RCCL's stream, side-stream early start) exactly as N > 1 does — the nccl code path on a 1-GPU box")
    ap.add_argument("--gather-mode", default="auto", choices=["auto", "allgather", "fanout"],
                    help="N > 1: transport of the row exchange (zett_amd/sharding.py RowGather): RCCL all-gather, or direct fan-out (every rank sends its shard to all peers at once, one xGMI link each)")
    ap.add_argument("--partition", default="contiguous", choices=["contiguous", "affinity"],
                    help="N > 1: which rows a rank computes — contiguous ranges of the vocabulary, or the id-affinity order of zett_amd.sharding.affinity_order "
                         "(zett_partition_rows: rows that share source ids share a rank; every step then retokenizes the whole vocabulary on every rank, runs the "
                         "partition kernel, and ends with the indexed copy that puts the gathered matrices back into vocabulary order)")
    ap.add_argument("--table-exchange", action="store_true",
                    help="N > 1 (f16): the hoisted input-projection table is computed ONCE across the ranks (zett_amd.sharding.SharedTable, zett_table_* of ABI 8: "
                         "every rank retokenizes the whole vocabulary, computes 1/N of the table of its distinct ids and all-gathers the slices) instead of "
                         "per rank for the ids of its own rows; same rows, same bits")
    ap.add_argument("--no-early-gather", action="store_true", help="N > 1, A/B: start the exchange of pred_in / bias behind the whole forward instead of behind their own completion point")
    args = ap.parse_args()

    import zett_amd
    zett_amd.configure_hw_queues()          # (N > 1: before the HIP runtime initialises)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            self_launch(args.gpus)          # (does not return)
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    # ZETT_BENCH_ONE_DEVICE=1 (test hook): every rank uses cuda:0 and the collectives go through gloo, so
    # that the N > 1 control flow can be exercised on a 1-GPU box; never set for a measurement.
    one_device = os.environ.get("ZETT_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    elif args.force_gather:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    exchange = world > 1 or args.force_gather          # steps end with the row exchange (RowGather)
    from zett_amd.sharding import resolve_gather_mode
    gather_mode = resolve_gather_mode(args.gather_mode, world)

    from zett_amd.hypernet import HipEngine
    from zett_amd.sharding import RowGather, plan_blocks

    cfg, rows, src_dtype, hist = synth.workload(args.workload)
    if args.rows:
        rows = args.rows
    dims = HypernetDims.from_config(cfg)
    lang = 3 if dims.embed_lang else None

    weights = device_weights(cfg, device, seed=0)
    engine = HipEngine(dims, 1e-5, device, args.precision)
    engine.load_weights(weights)
    if not exchange:
        engine.set_option("time_gemm", 1)      # (brackets every GEMM with HIP events and synchronises at the end of a forward: kept out of the N > 1 overlap)
    if args.gemm_variant:
        engine.set_option("gemm_variant", args.gemm_variant)
    if args.gemm4d_min_k:
        engine.set_option("gemm4d_min_k", args.gemm4d_min_k)
    if args.gemm_tile_order:
        engine.set_option("gemm_tile_order", args.gemm_tile_order)
    if args.max_chunk_tokens:
        engine.set_option("max_chunk_tokens", args.max_chunk_tokens)
    if args.no_pair_dedupe:
        engine.set_option("pair_dedupe", 0)
    if args.no_ln_fold:
        engine.set_option("ln_fold", 0)
    elif args.ln_fold != 1:
        engine.set_option("ln_fold", args.ln_fold)
    for kv in args.option:
        key, _, val = kv.partition("=")
        engine.set_option(key, int(val))
    # the same weights in the OTHER 16-bit arithmetic, for the side measurement after the timed region (N = 1 only)
    alt_precision = {"f16": "bf16", "bf16": "f16"}.get(args.precision) if (world == 1 and not exchange and not args.no_alt_precision) else None
    alt_engine = None
    if alt_precision:
        alt_engine = HipEngine(dims, 1e-5, device, alt_precision)
        alt_engine.load_weights(weights)
        alt_engine.set_option("time_gemm", 1)
        if args.no_pair_dedupe:
            alt_engine.set_option("pair_dedupe", 0)
        if args.no_ln_fold:
            alt_engine.set_option("ln_fold", 0)
        for kv in args.option:
            key, _, val = kv.partition("=")
            alt_engine.set_option(key, int(val))
    if rank != 0 or args.no_cpu_baseline or world > 1:
        weights_keep = None
    else:
        weights_keep = weights
    del weights

    ids_all = synth.make_surface_forms(cfg, rows, seed=0, hist=hist)
    # Row blocks of this rank (zett_amd/sharding.py): at N = 1 the whole vocabulary in one forward; at N > 1 the
    # vocabulary is cut into `--chunks` row blocks, each sharded over the ranks, and the all-gather of a block runs
    # on RCCL's stream under the forward of the next one — inside ONE step, which is what a caller with a single
    # vocabulary gets from predict_sharded.  Nothing is carried across steps.
    chunks = 1 if (not exchange or args.serial_allgather) else args.chunks
    blocks = plan_blocks(rows, world, rank, chunks)
    assert all(b.hi > b.lo for b in blocks), "fewer rows than ranks"
    ids_blocks = [torch.from_numpy(ids_all[b.lo:b.hi]).to(device) for b in blocks]
    # The step starts from SURFACE FORMS: the byte-level strings of this rank's target tokens and the tables of a
    # synthetic hn tokenizer of the SOURCE model's kind (synth.HN_MODEL_KIND: BPE with byte fallback and ~31.9 k merges for
    # Mistral / TinyLlama, BPE with ignore_merges and 128 k merges for Llama-3, Unigram for XLM-R) are resident on the
    # device; every step retokenizes them on the GPU (zett_retokenize: byte table, then merge queue / Viterbi) into the
    # [rows, L] id matrix the forward consumes.  The strings are built so that this matrix is exactly the workload's
    # (checked below, untimed).
    retok = None
    texts = []
    affinity = exchange and args.partition == "affinity"
    table_x = exchange and args.table_exchange
    if table_x and (args.no_retokenize or args.precision != "f16"):
        raise SystemExit("--table-exchange shares the folded 16-bit table of the f16 mode and starts from the whole vocabulary's retokenized ids")
    whole = affinity or table_x          # every rank retokenizes the WHOLE vocabulary
    if affinity and args.no_retokenize:
        raise SystemExit("--partition affinity orders the rows by the retokenized ids: it cannot be combined with --no-retokenize")
    seq_len = int(ids_all.shape[1])
    if not args.no_retokenize:
        from zett_amd.surface_forms import DeviceRetokenizer, HnTokenizerSpec
        hn_model, piece_of_id = synth.make_hn_model(args.workload, cfg)
        spec = HnTokenizerSpec.from_model_json(hn_model, ["<unk>", "<s>", "</s>"], [0, 1, 2], dims.pad_token_id)
        retok = DeviceRetokenizer(spec, device)
        # (affinity order: every rank retokenizes the WHOLE vocabulary — it needs every row's ids to compute the same order)
        for lo, hi in ([(0, rows)] if whole else [(b.lo, b.hi) for b in blocks]):
            d_text, d_off, n_tok = retok.encode(synth.tokens_for_surface_forms(cfg, ids_all[lo:hi], piece_of_id))
            sfm0, n_trunc0 = retok.run(d_text, d_off, n_tok, seq_len)
            if n_trunc0 != 0 or not torch.equal(sfm0.cpu(), torch.from_numpy(ids_all[lo:hi])):
                raise SystemExit("the retokenized surface forms differ from the workload's id matrix")
            texts.append((d_text, d_off, n_tok))
    src = torch.from_numpy(synth.make_source_embeddings(cfg, seed=0, dtype=src_dtype)).to(device)
    lang_arg = -1 if lang is None else lang

    acc = {"gemm_ms": 0.0, "gemm_flops_timed": 0.0, "gemm_launches": 0}
    exposed = []          # N > 1: per step, how long the compute stream waited for the row exchange after its last forward
    classes = {}          # launch class -> [launches, ms, flops, algorithmic bytes] over the timed steps (zett_get_gemm_log)

    def step():
        gather = None
        ready = None if args.no_early_gather else engine.stream_wait_output
        outs = None
        # the step's id matrices: every block retokenized at the head of the step, without a host round trip
        # (zett_retokenize_async; the truncation count of all steps is asked for once, after the timed region)
        order = None
        shared = None
        if whole:
            sfm_all = retok.run_async(*texts[0], seq_len)
            if affinity:
                from zett_amd.sharding import affinity_order
                order = affinity_order(sfm_all, world, dims.pad_token_id, dims.original_vocab_size + dims.n_extra, chunks=chunks)
                step.order = order
                sfms = [sfm_all.index_select(0, order[b.lo:b.hi]) for b in blocks]
            else:
                sfms = [sfm_all[b.lo:b.hi] for b in blocks]
            if table_x:          # plan of the whole vocabulary's ids, this rank's slice of the table, all-gather of the slices
                from zett_amd.sharding import SharedTable
                shared = SharedTable(engine, sfm_all, src)
                step.table_bytes = shared.bytes_received()
        else:
            sfms = ids_blocks if retok is None else [retok.run_async(*texts[k], seq_len) for k in range(len(blocks))]
        if exchange:
            gather = RowGather(blocks, mode=gather_mode, order=order)          # (affinity: every block is scattered to its vocabulary rows behind its exchange)
        if len(blocks) > 1:
            ahead.wait_stream(torch.cuda.current_stream(device))      # behind this step's retokenization
        for k, b in enumerate(blocks):
            outs = engine.forward(sfms[k], src, lang_arg) if shared is None else engine.forward_table(sfms[k], shared.table, shared.stats, shared.id_slot, lang_arg)
            if k + 1 < len(blocks) and shared is None:
                engine.prepare(sfms[k + 1], ahead)                    # the next block's plan runs under this block's forward (zett_forward_prepare)
            st_k = engine.stats()
            for key in acc:
                acc[key] += st_k[key]
            if world == 1:
                for r in engine.gemm_log():
                    c = classes.setdefault(launch_class(r), [0, 0.0, 0.0, 0.0])
                    c[0] += 1; c[1] += r["ms"]; c[2] += r["flops"]; c[3] += r["bytes"]
            if gather is not None:
                gather.add(b, outs, ready)     # async exchange of this block (pred_in / bias from their own completion point); the next block's forward runs meanwhile
        if gather is None:
            return outs
        full = gather.finish(rows, timed=True)
        exposed.append(gather.exposed_ms)
        return full

    gemm_ms = gemm_fl = 0.0
    launches = 0
    out = None
    ahead = torch.cuda.Stream(device=device) if len(blocks) > 1 else None
    for _ in range(args.warmup):
        out = step()        # the previous outputs stay referenced while the next step runs, as in the timed loop: the
                            # caching allocator gets both output sets it will alternate between before the clock starts
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for key in acc:
        acc[key] = 0
    classes.clear()
    exposed.clear()
    for _ in range(args.steps):
        out = step()
    gemm_ms, gemm_fl, launches = acc["gemm_ms"], acc["gemm_flops_timed"], acc["gemm_launches"]
    timed_classes = {k: list(v) for k, v in classes.items()}
    torch.cuda.synchronize()   # (every step ends with its own all-gathers complete on the compute stream)
    if retok is not None and retok.result() != 0:
        raise SystemExit("a retokenized surface form was truncated")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if exchange:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # untimed: the gathered matrix must hold this rank's rows bit for bit (rows are shard-independent)
        ids_dev = torch.from_numpy(ids_all).to(device) if affinity else None
        # the same untimed pass measures this rank's GEMM launches with per-launch HIP events (kept out of the timed steps: the sync that
        # reads them would serialise the overlap of exchange and forward): the roofline figure of an N > 1 line
        post = {"gemm_ms": 0.0, "gemm_flops_timed": 0.0, "gemm_launches": 0}
        post_classes = {}
        engine.set_option("time_gemm", 1)
        for b, ids_b in zip(blocks, ids_blocks):
            mine = step.order[b.lo:b.hi] if affinity else None
            chk = engine.forward(ids_dev.index_select(0, mine) if affinity else ids_b, src, lang_arg)          # `ids_b` == the retokenized matrix (checked above)
            st_k = engine.stats()
            for key in post:
                post[key] += st_k[key]
            for r in engine.gemm_log():
                c = post_classes.setdefault(launch_class(r), [0, 0.0, 0.0, 0.0])
                c[0] += 1; c[1] += r["ms"]; c[2] += r["flops"]; c[3] += r["bytes"]
            for full, loc in zip(out, chk):
                if full is not None and not torch.equal(full.index_select(0, mine) if affinity else full[b.lo:b.hi], loc):
                    raise SystemExit(f"rank {rank}: all-gathered rows [{b.lo}, {b.hi}) differ from the local forward")
        engine.set_option("time_gemm", 0)
    st = engine.stats()

    ms_per_step = dt / args.steps * 1e3
    value = rows * args.steps / dt
    roofline_steps = args.steps
    roofline_source = "per-launch HIP events of the timed steps"
    if exchange and gemm_ms <= 0 and post["gemm_ms"] > 0:
        gemm_ms, gemm_fl, launches = post["gemm_ms"], post["gemm_flops_timed"], post["gemm_launches"]
        timed_classes = {k: list(v) for k, v in post_classes.items()}
        roofline_steps = 1
        roofline_source = "per-launch HIP events of ONE untimed pass over this rank's row blocks after the timed region (rank 0; the timed steps run without events)"
    achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    peak = PEAK_TFLOPS[args.precision]

    # HBM traffic of the dominant kernel comes from PMC counters, which cannot be read from inside this process: the
    # figure is the one measured by the committed rocprofv3 passes (profiles/*_pmc.md, tools/pmc_summary.py --json) for
    # this same command.  It is only quoted for the workload and precision it was taken on AND while the HIP sources
    # still hash to what was profiled (zett_amd.build.source_hash): a stale file yields null, not an old number.
    traffic = None
    traffic_source = None
    try:
        from zett_amd.build import source_hash
        pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
        if (args.workload == "mistral_gpt2_32k" and args.precision == pmc.get("precision") and world == 1 and not args.rows
                and pmc.get("source_hash") == source_hash() and not pmc.get("stale")):
            traffic = pmc["hbm_bytes_per_launch"]
            traffic_source = {"profile_set": pmc.get("tag"), "commit": pmc.get("commit"), "source_hash": pmc.get("source_hash"), "method": pmc.get("method")}
        else:
            traffic_source = {"stale": True, "reason": "profiles/pmc_traffic.json was taken on other HIP sources, another workload or another precision",
                              "profile_set": pmc.get("tag"), "source_hash_profiled": pmc.get("source_hash"), "source_hash_now": source_hash()}
    except Exception:
        traffic = None
    alg_bytes = sum(v[3] for v in timed_classes.values())
    alg_launches = sum(v[0] for v in timed_classes.values())

    f_ref = as_written_flops_per_row(dims, int(ids_all.shape[1]))

    result = {
        "metric": "predicted token-embeddings/sec (full target vocab)",
        "value": value, "unit": "token-embeddings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic" + (" [TEST HOOK: all ranks on one device, gloo]" if one_device else "") + (f" [A/B: gemm_variant {args.gemm_variant} forced]" if args.gemm_variant else ""),
        "config": {"workload": f"{args.workload}: {rows}-row target vocab, hypernet E={dims.n_embd} E_in={dims.n_in_embd} "
                               f"H={dims.hidden} I={dims.intermediate} heads={dims.heads} layers={dims.layers} "
                               f"L={ids_all.shape[1]}, source_embeddings {src_dtype}",
                   "rows": rows, "rows_per_gpu": sum(b.hi - b.lo for b in blocks),
                   "partition": args.partition if exchange else None, "table_exchange": bool(table_x) if exchange else None,
                   "parallelism": f"vocab-row shards x{world} + RCCL all-gather" + ("" if world == 1 else (" (after the forward)" if chunks == 1 else f" ({len(blocks)} row blocks per step: the all-gather of a block runs on the RCCL stream under the next block's forward; nothing overlaps across steps)")),
                   "precision": f"{args.precision} MFMA operands, fp32 accumulate/LN/softmax/GELU/outputs" if args.precision != "f32" else "fp32 MFMA",
                   "packed_tokens_rank0": st["packed_tokens"], "distinct_source_ids_rank0": st["distinct_ids"],
                   "distinct_id_position_pairs_rank0": st["distinct_positions"],
                   "hn_tokenizer": (None if retok is None else
                                    f"synthetic {hn_model['type']}" + (f", {len(hn_model['merges'])} merges" if hn_model["type"] == "BPE" else f", {len(hn_model['vocab'])} pieces")
                                    + (", byte fallback" if hn_model.get("byte_fallback") else "") + (", ignore_merges" if hn_model.get("ignore_merges") else "")),
                   "step": ("surface forms (byte strings resident on the device) -> GPU retokenization -> hypernet forward" if retok is not None
                            else "id matrix -> hypernet forward [A/B: --no-retokenize]") + ("" if world == 1 else " -> all-gather")},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak if peak else None, "traffic": traffic,
                     "kernel": "zett::gemm4d_tn_kernel (256x256 four-wave direct-to-LDS MFMA GEMM on 16x16x32 MFMAs: every 16-bit launch with K >= 512; its epilogues carry bias / GELU / residual and, with the LayerNorm fold, the encoder's LayerNorms), zett::gemm8r_tn_kernel (256x256 register-staged: shorter K and fp32 mode), 384x256 / 128x128 tiles where wave quantisation / small shapes call for them: all GEMM launches, FLOP-weighted", "launches_per_step": launches / max(roofline_steps, 1), "source": roofline_source,
                     "note": ("the GEMM launches also carry the encoder's LayerNorms (LayerNorm fold, DESIGN.md section 4): same box with --no-ln-fold, "
                              "frac +0.01 and ms_per_step +0.9" if (args.precision != "f32" and not args.no_ln_fold) else "no LayerNorm fold in this run"),
                     "gemm_ms_per_step": gemm_ms / max(roofline_steps, 1),
                     "executed_tflop_per_step": gemm_fl / max(roofline_steps, 1) / 1e12,
                     "traffic_source": traffic_source,
                     # A and W read once, every output written once, residual rows read once, summed over the launches of the
                     # timed steps / their number: what `traffic` (a PMC measurement, when present) is to be compared with
                     "algorithmic_bytes_per_launch": alg_bytes / alg_launches if alg_launches else None,
                     "traffic_over_algorithmic": (traffic / (alg_bytes / alg_launches)) if (traffic and alg_launches) else None,
                     # the same FLOP / HIP-event-time ratio per launch class (zett_get_gemm_log), so that the fraction can be
                     # read class by class: K loops are alike, the epilogues differ
                     "by_class": class_table(timed_classes, roofline_steps, peak)},
        # N > 1: the part of a step the compute stream spent waiting for the row exchange (HIP events around the waits in
        # RowGather.finish, this rank); the rest of the exchange ran under forwards
        "exchange_exposed_ms_per_step": (sum(x for x in exposed if x is not None) / max(len(exposed), 1)) if exchange else None,
        "exchange": None if not exchange else {"mode": gather_mode, "backend": dist.get_backend(), "one_rank_group": world == 1, "early_start_of_pred_in_and_bias": not args.no_early_gather,
                                             "bytes_received_per_rank_per_step": int(rows * (world - 1) / world * (dims.n_embd * (2 if dims.separate_out else 1) + 1) * 4),
                                             "table_bytes_received_per_rank_per_step": getattr(step, "table_bytes", None) if table_x else None},
        "as_written_tflops": rows * f_ref * args.steps / dt / 1e12,
        "as_written_gflop_per_row": f_ref / 1e9,
    }
    if world == 1 and not exchange:
        # What a caller gets: the same steps with the per-launch HIP events (and the stream sync that reads them at the end of
        # every forward) switched off.  Measured after the timed region; `value` / `ms_per_step` stay the instrumented,
        # conservative figures the roofline is priced on.
        engine.set_option("time_gemm", 0)
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt_plain = time.perf_counter() - t1
        engine.set_option("time_gemm", 1)
        result["ms_per_step_uninstrumented"] = dt_plain / args.steps * 1e3
        result["value_uninstrumented"] = rows * args.steps / dt_plain
        result["range_flags"] = engine.range_flags()          # range guard of the 16-bit arithmetic: 0 = nothing left the operand range
    if alt_engine is not None:
        # Side measurement, outside the timed region and never `value`: the SAME steps in the other 16-bit arithmetic
        # (f16 is the default because bf16 sits on the edge of the parity tolerance; it costs ~4 %: both
        # numbers belong next to each other in the driver's record).
        main_engine, engine = engine, alt_engine
        for key in acc:
            acc[key] = 0
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        for key in acc:
            acc[key] = 0
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt_alt = time.perf_counter() - t1
        alt_tf = acc["gemm_flops_timed"] / (acc["gemm_ms"] * 1e-3) / 1e12 if acc["gemm_ms"] > 0 else 0.0
        result["alt_precision"] = {"dtype": alt_precision, "value": rows * args.steps / dt_alt, "unit": "token-embeddings/s",
                                   "ms_per_step": dt_alt / args.steps * 1e3, "roofline_frac": alt_tf / PEAK_TFLOPS[alt_precision],
                                   "gemm_tflops": alt_tf,
                                   "note": "same workload, steps and warm-up in the other 16-bit arithmetic, measured after the timed region; "
                                           "bf16: rel-L2 ~0.97e-2 of the fp32 reference at this shape (tolerance 1e-2), f16: ~0.12e-2"}
        engine = main_engine
        alt_engine.close()
    if world == 1 and not exchange and args.precision != "f32" and not args.no_alt_precision and not args.rows:
        # Side measurement, outside the timed region and never `value`: the same workload in exact fp32 MFMA arithmetic — the
        # mode whose tolerance north_star names (max |err| <= 1e-4 relative to the row maximum against the fp32 reference).
        engine.close()
        torch.cuda.empty_cache()
        f32_engine = HipEngine(dims, 1e-5, device, "f32")
        f32_engine.load_weights(device_weights(cfg, device, seed=0))
        f32_engine.set_option("time_gemm", 1)
        engine = f32_engine
        for key in acc:
            acc[key] = 0
        step()
        torch.cuda.synchronize()
        for key in acc:
            acc[key] = 0
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        dt_f32 = time.perf_counter() - t1
        f32_tf = acc["gemm_flops_timed"] / (acc["gemm_ms"] * 1e-3) / 1e12 if acc["gemm_ms"] > 0 else 0.0
        result["f32_mode"] = {"dtype": "f32", "value": rows * 2 / dt_f32, "unit": "token-embeddings/s", "ms_per_step": dt_f32 / 2 * 1e3, "steps": 2,
                              "gemm_tflops": f32_tf, "roofline_frac": f32_tf / PEAK_TFLOPS["f32"], "peak": PEAK_TFLOPS["f32"],
                              "note": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32, gemm8r tile); measured after the timed region"}
        f32_engine.close()
    if rank == 0 and world == 1 and not exchange and not args.no_side_configs and not args.rows and args.workload == "mistral_gpt2_32k":
        # Side measurements, outside the timed region and never `value`: the other 1-GPU configs of BASELINE.json (C2, C3) and the
        # single-GPU proxy of the 8-GPU step (rank 0's share of the headline vocabulary), so that they are in the driver's record;
        # then the same workloads through the kept Python API.
        engine.close()
        torch.cuda.empty_cache()
        result["configs"] = []
        for name, shard_of, part, tx in (("xlmr_gpt2", 0, "contiguous", False), ("tinyllama_neox", 0, "contiguous", False), ("mistral_gpt2_32k", 8, "contiguous", False),
                                         ("mistral_gpt2_32k", 8, "affinity", False), ("mistral_gpt2_32k", 8, "contiguous", True)):
            if tx and args.precision != "f16":
                continue          # (the shared table is the folded 16-bit table of the f16 mode)
            try:
                result["configs"].append(side_config(name, 0, args.precision, device, steps=5, warmup=2, shard_of=shard_of, partition=part, table_exchange=tx))
            except Exception as e:          # a side line that cannot run must not take the benchmark line with it
                result["configs"].append({"workload": name, "error": f"{type(e).__name__}: {e}"})
        result["api_path"] = []
        for name in ("xlmr_gpt2", "mistral_gpt2_32k"):
            try:
                r = api_path(name, args.precision, device)
                ref = ms_per_step if name == args.workload else next((c["ms_per_step"] for c in result["configs"] if c.get("workload") == name and "ms_per_step" in c), None)
                r["ms_per_step_engine"] = ref
                r["over_engine_step"] = (r["ms"] / ref) if ref else None
                result["api_path"].append(r)
            except Exception as e:
                result["api_path"].append({"workload": name, "error": f"{type(e).__name__}: {e}"})
        try:          # N4, the training use of the path: 16 384 rows of the headline shape, bf16 contractions (the arithmetic to train in)
            result["train_step"] = train_step_line(args.workload, 16384, "bf16", device)
        except Exception as e:
            result["train_step"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not exchange and not args.no_live_traffic and not args.rows:
        # roofline.traffic measured HERE: same build, same box, same workload, right after the timed region
        live, info = measure_traffic_live(args)
        if live is not None:
            result["roofline"]["traffic"] = live
            result["roofline"]["traffic_source"] = dict(info, live=True)
            if alg_launches:
                result["roofline"]["traffic_over_algorithmic"] = live / (alg_bytes / alg_launches)
        else:
            src_info = result["roofline"].get("traffic_source") or {}
            result["roofline"]["traffic_source"] = dict(src_info, live=False, live_failure=str(info))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, ref_out, n_ref = cpu_baseline(cfg, weights_keep, ids_all, src, lang, args.cpu_budget_s)
        result["cpu_baseline"] = cb
        got = out[0][:n_ref].float().cpu().numpy()
        num = np.linalg.norm(got - ref_out[0])
        result["parity_vs_cpu_port_rel_l2"] = float(num / (np.linalg.norm(ref_out[0]) + 1e-30))
    else:
        result["cpu_baseline"] = None
    if rank == 0:
        emit(result)
    if exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
