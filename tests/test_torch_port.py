"""CPU: the torch restatement the gradient tests differentiate (tests/torch_port.py) IS the oracle's math, for every flag
combination the GPU gradient test uses, on golden fixtures taken from the reference; and the training path has no CPU
fallback either."""
import numpy as np
import pytest
import torch

from tests import torch_port, util
from zett_amd import synth


@pytest.mark.parametrize("path", util.golden_cases("fwd_tiny_*.npz")[::3], ids=lambda p: p.split("/")[-1][:-4])
def test_torch_port_reproduces_reference_outputs(path):
    case = util.load_case(path)
    w = synth.make_weights(case["cfg"], case["seed"])
    src = synth.make_source_embeddings(case["cfg"], case["seed"], dtype=case["src_dtype"])
    W = {k: torch.from_numpy(v).double() for k, v in w.items()}
    got = torch_port.forward(W, case["cfg"], torch.from_numpy(case["ids"]).long(), torch.from_numpy(src), case["lang"])
    keep = ~util.all_pad_rows(case["cfg"], case["ids"])
    for g, want in zip(got, (case["pred_in"], case["pred_out"], case["bias"])):
        if want is None:
            assert g is None
        else:
            assert float(np.abs(g.numpy()[keep] - want[keep]).max()) < 2e-5


def test_training_path_refuses_cpu_tensors():
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, *_ = synth.workload("tiny")
    model = ZettHypernet(ZettHypernetConfig(**cfg)).requires_grad_(True).train()
    with pytest.raises(RuntimeError, match="MI355X"):
        model(torch.zeros(2, 7, dtype=torch.long), source_embeddings=torch.zeros(300, 128), lang_index=torch.tensor(0))
