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
x = syn.make_windows(syn.FULL_ALIGNMENT, 1000, seed=1, channels=8)
for _ in range(10): y = m.predict_numpy(x)
best = 0
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(60): y = m.predict_numpy(x)
    best = max(best, 60 * 1000 / (time.perf_counter() - t0))
print("%%.0f" %% best)
''' % ROOT
cfgs = [("default (250 + 750, lanes <= 512)", {}),
        ("one lane, no tail stream (round 5)", {"C3HIP_RING_LANES": "1", "C3HIP_TAIL_STREAM": "0"}),
        ("250,750 lanes <= 1000", {"C3HIP_PREDICT_PIECES": "250,750", "C3HIP_RING_LANES_MAX_BATCH": "1000"}),
        ("128,256,616 lanes <= 1000", {"C3HIP_PREDICT_PIECES": "128,256,616", "C3HIP_RING_LANES_MAX_BATCH": "1000"}),
        ("128,256,616 lanes <= 512", {"C3HIP_PREDICT_PIECES": "128,256,616"}),
        ("200,400,400", {"C3HIP_PREDICT_PIECES": "200,400,400"}),
        ("250,250,250,250", {"C3HIP_PREDICT_PIECES": "250"}),
        ("160,280,280,280", {"C3HIP_PREDICT_PIECES": "160,280"}),
        ("334,333,333", {"C3HIP_PREDICT_PIECES": "334,333,333"}),
        ("500,500", {"C3HIP_PREDICT_PIECES": "500"}),
        ("128,436,436", {"C3HIP_PREDICT_PIECES": "128,436"})]
for rnd in range(2):
    for name, env in cfgs:
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
        print(f"round {rnd}: {name:38s} {r.stdout.strip() or r.stderr[-200:]} windows/s", flush=True)
