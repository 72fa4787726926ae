"""bench.py's N > 1 control flow (row shards, overlapped / serial all-gather, the gathered-rows check)
exercised on a 1-GPU box: both ranks share cuda:0 and the collectives run over gloo
(ZETT_BENCH_ONE_DEVICE=1, a test hook that labels its JSON line as such)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--serial-allgather"], ["--gather-mode", "fanout"], ["--serial-allgather", "--no-early-gather"],
                                   ["--partition", "affinity"], ["--partition", "affinity", "--serial-allgather", "--gather-mode", "fanout"]],
                         ids=["two-blocks-self-launched", "serial", "fanout", "serial-late", "affinity", "affinity-serial-fanout"])
def test_two_ranks_on_one_device(extra):
    env = dict(os.environ, ZETT_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533"]
    if not extra:
        # the form the driver uses for every N: `python bench.py --gpus N ...` with no launcher around it — bench.py re-executes
        # itself under torch.distributed.run (bench.self_launch)
        launcher = [sys.executable]
        env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    cmd = launcher + [os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                      "--workload", "tiny", "--rows", "20001", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]              # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["rows"] == 20001 and d["config"]["rows_per_gpu"] == 10001
    assert d["value"] > 0 and "TEST HOOK" in d["data"]
    # the exchange is accounted for: transport, early start of pred_in / bias, and what of it the compute stream waited for
    assert d["config"]["partition"] == ("affinity" if "affinity" in extra else "contiguous")
    assert d["exchange"]["mode"] == ("fanout" if "fanout" in extra else "allgather")
    assert d["exchange"]["early_start_of_pred_in_and_bias"] == ("--no-early-gather" not in extra)
    assert d["exchange_exposed_ms_per_step"] is not None and 0 <= d["exchange_exposed_ms_per_step"] < d["ms_per_step"]
    assert len(lines[0].encode()) < 6000
    # (r6) an N > 1 line has a roofline figure too: the timed steps run without per-launch events, the untimed bit-identity pass behind them with
    assert 0 < d["roofline"]["frac"] < 1 and "untimed" in d["roofline"]["source"] and d["roofline"]["gemm_ms_per_step"] > 0


@pytest.mark.parametrize("extra", [[], ["--partition", "affinity", "--gather-mode", "fanout"]], ids=["shared-table", "shared-table-affinity-fanout"])
def test_two_ranks_share_the_hoisted_table(extra):
    """--table-exchange (ABI 8, SURVEY 8e's optional second exchange): every rank retokenizes the whole vocabulary, computes half of the
    table of its distinct ids, the halves are all-gathered, and the forwards run on the complete table — bench.py's own untimed check then
    compares every gathered row with the PLAIN local forward bit for bit.  XLM-R shape (the folded 16-bit table needs H >= 512), 20 001 rows."""
    env = dict(os.environ, ZETT_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "xlmr_gpt2", "--rows", "20001",
           "--no-cpu-baseline", "--table-exchange"] + extra
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["table_exchange"] is True and d["config"]["rows"] == 20001
    assert d["exchange"]["table_bytes_received_per_rank_per_step"] > 0 and d["value"] > 0 and len(lines[0].encode()) < 6000


def test_default_line_is_short_and_complete(tmp_path):
    """The DEFAULT command (what the driver runs, with fewer steps): ONE `{` line on stdout, under 6 000 bytes (the driver keeps an
    ~8 KB tail of stdout: round 5's 22 KB line was cut and its record did not parse), carrying the contract fields, `roofline`
    and `cpu_baseline`, the compact side measurements — and the full objects in bench_side.json."""
    env = dict(os.environ, ZETT_BENCH_SIDE_DIR=str(tmp_path))
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-budget-s", "8"], cwd=REPO, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-3000:]
    assert len(lines[0].encode()) < 6000
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["steps"] == 2 and d["n_gpus"] == 1 and d["config"]["workload"].startswith("mistral_gpt2_32k")
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and 0.3 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "traffic" in rf and rf["gemm_ms_per_step"] <= d["ms_per_step"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["parity_vs_cpu_port_rel_l2"] < 2.5e-3
    assert [c["workload"] for c in d["configs"]] == ["xlmr_gpt2", "tinyllama_neox", "mistral_gpt2_32k", "mistral_gpt2_32k", "mistral_gpt2_32k"] and all("error" not in c for c in d["configs"])
    # (r6) the third shard proxy runs on the hoisted table shared between 8 ranks: fewer table rows than the contiguous shard computes
    assert "shared table" in d["configs"][4]["shard"] and d["configs"][4]["table_mb_received"] > 100 and d["configs"][4]["ms_per_step"] < d["configs"][2]["ms_per_step"]
    assert len(d["api_path"]) == 2 and all("error" not in a for a in d["api_path"]) and "error" not in d["train_step"]
    full = json.loads((tmp_path / "bench_side.json").read_text())
    assert abs(full["value"] / d["value"] - 1) < 1e-3 and "by_class" in full["configs"][0]["roofline"]
