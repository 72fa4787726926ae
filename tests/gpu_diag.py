"""Not a test: prints per-case error figures of the HIP forward vs the goldens."""
import sys, time
import numpy as np
from tests import util
from zett_amd import synth

def main(patterns):
    for pat in patterns:
        for path in util.golden_cases(pat):
            case = util.load_case(path)
            w = synth.make_weights(case["cfg"], case["seed"])
            src = synth.make_source_embeddings(case["cfg"], case["seed"], dtype=case["src_dtype"])
            for prec in ("f32", "bf16"):
                model = util.hip_model(case["cfg"], w, prec)
                t = time.time()
                out = util.hip_forward(model, case["ids"], src, case["lang"])
                dt = time.time() - t
                msg = []
                for got, want, nm in ((out[0], case["pred_in"], "in"), (out[1], case["pred_out"], "out"), (out[2], case["bias"], "bias")):
                    if want is None: continue
                    err = np.abs(got - want)
                    rel = np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30)
                    worst_row = int(err.reshape(len(err), -1).max(1).argmax())
                    msg.append(f"{nm}: max {err.max():.2e} relL2 {rel:.2e} row {worst_row} finite {np.isfinite(got).all()}")
                print(case["name"], prec, " | ".join(msg), f"{dt:.2f}s", model.engine(model.device if hasattr(model,'device') else None).stats() if False else "", flush=True)
                del model

if __name__ == "__main__":
    main(sys.argv[1:] or ["fwd_tiny_1111*.npz", "fwd_tiny_L*.npz", "fwd_real_x*.npz"])
