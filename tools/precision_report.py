"""Per-output rel-L2 of the HIP forward vs the committed real-shape goldens in bf16 and f16 mode
(prints the numbers quoted in tests/util.py and DESIGN.md).  Run on a GPU box: python tools/bias_err.py"""
import numpy as np, sys
sys.path.insert(0, "/root/repo")
from tests import util
from zett_amd import synth
import glob
for path in sorted(glob.glob("/root/repo/tests/golden/fwd_real_*.npz")):
    case = util.load_case(path)
    w = synth.make_weights(case["cfg"], case["seed"])
    src = synth.make_source_embeddings(case["cfg"], case["seed"], dtype=case["src_dtype"])
    for prec in ("bf16", "f16"):
        model = util.hip_model(case["cfg"], w, prec)
        out = util.hip_forward(model, case["ids"], src, case["lang"])
        def rel(g, w_): return np.linalg.norm(g - w_) / np.linalg.norm(w_)
        r = [rel(out[0], case["pred_in"])]
        if case["pred_out"] is not None: r.append(rel(out[1], case["pred_out"]))
        if case["cfg"].get("hn_predict_bias"): r.append(rel(out[2], case["bias"]))
        print(case["name"], prec, "rows", len(case["ids"]), "rel-L2 in/out/bias:", ["%.4f" % x for x in r])
        if prec == "bf16" and case["cfg"].get("hn_predict_bias"):
            print("   bias want", np.round(case["bias"][:8], 3), "err", np.round((out[2] - case["bias"])[:8], 4))
        del model
