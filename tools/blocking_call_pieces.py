import os, sys, time, subprocess, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from clair3_amd import synthetic as syn
from clair3_amd.model import Clair3_F
sd = syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=0)
m = Clair3_F(add_indel_length=True, predict=True, input_channels=8).to("cuda:0"); m.eval(); m.load_state_dict(sd)
B = int(os.environ.get('PIECES_BATCH', '1000'))
x = syn.make_windows(syn.FULL_ALIGNMENT, B, seed=1, channels=8)
for _ in range(10): y = m.predict_numpy(x)
best = 0
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(60): y = m.predict_numpy(x)
    best = max(best, 60 * B / (time.perf_counter() - t0))
print("%%.0f" %% best)
''' % ROOT
cfgs = [("B=%s default" % b, {"PIECES_BATCH": b}) for b in ("600", "1000", "1500", "2000")] + \
       [("B=%s growing pieces (C3HIP_PREDICT_EQUAL=0)" % b, {"PIECES_BATCH": b, "C3HIP_PREDICT_EQUAL": "0"}) for b in ("600", "1000", "1500")]
for rnd in range(2):
    for name, env in cfgs:
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
        print(f"round {rnd}: {name:38s} {r.stdout.strip() or r.stderr[-200:]} windows/s", flush=True)
