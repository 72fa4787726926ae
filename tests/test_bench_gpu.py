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
                         ids=["two-blocks", "serial", "fanout", "serial-late", "affinity", "affinity-serial-fanout"])
def test_two_ranks_on_one_device(extra):
    env = dict(os.environ, ZETT_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
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
