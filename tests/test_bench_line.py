"""bench.py's stdout line (CPU): the compact line assembled from a full result object stays under the byte limit with the
contract fields intact, and `python bench.py --gpus N` without a launcher becomes a torch.distributed.run command."""
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def _round5_full_line():
    """The 22 KB line of round 5 (the one the driver could not parse), kept under profiles/ — a real full result object."""
    return json.loads(open(os.path.join(REPO, "profiles", "r5z_bench.json")).read().strip().splitlines()[-1])


def test_compact_line_of_a_real_result():
    full = _round5_full_line()
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    assert len(line.encode()) < bench.LINE_LIMIT <= 6000 and "\n" not in line
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["value"] == pytest.approx(full["value"], rel=1e-4) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-4)
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4) and rf["traffic"] == pytest.approx(full["roofline"]["traffic"], rel=1e-4)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert [c["workload"] for c in d["configs"]] == ["xlmr_gpt2", "tinyllama_neox", "mistral_gpt2_32k", "mistral_gpt2_32k"]
    assert d["configs"][0]["ms_per_step"] == pytest.approx(full["configs"][0]["ms_per_step"], rel=1e-3)
    assert d["api_path"][0]["over_engine_step"] == pytest.approx(full["api_path"][0]["over_engine_step"], rel=1e-3)
    assert d["train_step"]["ms_per_step"] == pytest.approx(full["train_step"]["ms_per_step"], rel=1e-3)


def test_compact_line_sheds_optional_parts_never_the_contract():
    full = _round5_full_line()
    full["configs"] = full["configs"] * 40                       # a run that grew its side measurements again
    full["api_path"] = full["api_path"] * 40
    full["roofline"]["by_class"] = full["roofline"]["by_class"] * 30
    full["config"]["workload"] = full["config"]["workload"] * 10
    line = bench.compact_line(full)
    assert len(line.encode()) <= bench.LINE_LIMIT
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["roofline"]["frac"] and d["cpu_baseline"]["value"]


def test_compact_line_multi_gpu_fields():
    full = _round5_full_line()
    for k in ("configs", "api_path", "train_step", "alt_precision", "f32_mode"):
        full.pop(k)
    full.update(n_gpus=8, cpu_baseline=None, exchange_exposed_ms_per_step=0.31,
                exchange={"mode": "fanout", "backend": "nccl", "one_rank_group": False, "early_start_of_pred_in_and_bias": True, "bytes_received_per_rank_per_step": 1})
    d = json.loads(bench.compact_line(full))
    assert d["exchange"]["mode"] == "fanout" and "GPU_MAX_HW_QUEUES" in d["exchange"] and d["exchange_exposed_ms_per_step"] == 0.31
    assert d["cpu_baseline"] is None and d["n_gpus"] == 8


def test_self_launch_command(monkeypatch):
    seen = {}
    monkeypatch.setattr(os, "execv", lambda exe, argv: seen.update(exe=exe, argv=argv))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "3"])
    bench.self_launch(8)
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in a and "--nproc-per-node=8" in a and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(a[a.index("--master-port") + 1]) < 65536
    k = a.index(os.path.join(REPO, "bench.py"))
    assert a[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"]


def test_committed_pmc_traffic_matches_the_sources_or_says_stale():
    """profiles/pmc_traffic.json is the fallback of roofline.traffic when the live PMC passes cannot run: it must have been taken
    on the HIP sources of THIS tree (zett_amd.build.source_hash) or carry "stale": true — bench.py then reports null, never an
    old number (the live measurement of every default run does not depend on the file)."""
    from zett_amd.build import source_hash
    pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
    assert pmc.get("stale") is True or pmc["source_hash"] == source_hash(), (pmc.get("source_hash"), source_hash())


def test_cached_cpu_baselines_cover_the_side_configs():
    """profiles/cpu_baselines.json (tools/cpu_baselines.py on a GPU box's host): one entry per BASELINE.json config, each on a
    >= 1 024-row slice, labelled with the box; the compact line quotes C2 / C3 from it."""
    cached = json.load(open(os.path.join(REPO, "profiles", "cpu_baselines.json")))
    assert cached["box"] and len(cached["by_config"]) == 6
    for e in cached["by_config"]:
        assert e["kind"] == "port" and e["cores"] >= 1 and e["value"] > 0 and e["levers_value"] > 0
        rows = int(e["sample"].split("first ")[1].split(" rows")[0])
        assert rows >= min(1024, e["rows_in_config"]), e
    assert {"xlmr_gpt2", "tinyllama_neox", "mistral_neox", "llama3_256k", "mistral_gpt2_32k"} <= set(cached["configs"])
    full = _round5_full_line()
    d = json.loads(bench.compact_line(full))
    assert d["configs"][0]["cpu_baseline"]["value"] == pytest.approx(cached["configs"]["xlmr_gpt2"]["value"], rel=1e-3)
    assert d["configs"][1]["cpu_baseline"]["kind"] == "port" and "cpu_baseline" not in d["configs"][2]
