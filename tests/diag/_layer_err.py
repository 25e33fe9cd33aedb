import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from clair3_amd import synthetic as syn
from clair3_amd.model import Clair3_F
from oracle import oracle
from tests import util
meta = util.manifest()["fa_realistic"]
sd, x = util.case_inputs(meta); x = x[:6]
y_o, d = oracle.fa_forward(sd, x, True, debug=True)
for env in ({}, {"C3HIP_WINOGRAD_F16MASK": "0"}, {"C3HIP_CONV1_F16": "0"}):
    for k in ("C3HIP_WINOGRAD_F16MASK", "C3HIP_CONV1_F16"): os.environ.pop(k, None)
    os.environ.update(env)
    m = Clair3_F(add_indel_length=True, predict=True, input_channels=8); m.keep_activations(True); m.to("cuda:0"); m.load_state_dict(sd)
    y = m.predict_numpy(x)
    out = []
    for l in range(9):
        a = m.debug_fetch(f"act{l}", d[f"act{l}"].shape); r = d[f"act{l}"]
        e = np.abs(a - r); i = np.unravel_index(e.argmax(), e.shape)
        out.append(f"act{l}: {e.max()/max(1,np.abs(r).max()):.2e} (|ref|max {np.abs(r).max():.1f}, at ref {r[i]:.3g})")
    print(env, "\n  " + "\n  ".join(out))
