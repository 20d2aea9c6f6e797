import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import restatement as R
from oracle.torch_restatement import TorchModel
import pivp_amd as pivp
np.set_printoptions(precision=3, linewidth=200)
P = R.init_params(seed=2, dtype=np.float64, scale=1.0, num_masks=4)
imgs, acts, stas = R.synthetic_batch(2, 3)
tm = TorchModel(4, params=P, requires_grad=True); l = tm([imgs, acts, stas], 0); l.backward()
for key in ("current_state/W", "current_state/b", "enc3/W"):
    g = tm.p[key].grad.numpy()
    outs = []
    for r in range(3):
        m = pivp.Model(4, prefix="t", keep_activations=True); m.load_state_dict_reference(P)
        m([imgs, acts, stas], 0); m.cleargrads(); m.backward()
        outs.append(m.grads_reference()[key].astype(np.float64))
    print(key, "norm", np.linalg.norm(g), "rel errs", [np.linalg.norm(o - g) / np.linalg.norm(g) for o in outs], "run-to-run", np.linalg.norm(outs[0] - outs[1]) / np.linalg.norm(g))
    if key == "current_state/W":
        print("ref\n", g); print("err run0\n", outs[0] - g); print("err run1\n", outs[1] - g)
for ss in (0, 1):
    import os
    os.environ["X"] = "1"
