"""The ring of one handle in a FRESH process (no torch streams, no other handle: the worker's situation): windows/s host to host.
python tools/ring_fresh.py full_alignment|pileup BATCH [STEPS]   (env: C3HIP_RING_LANES, C3HIP_RING_LANES_MAX_BATCH, ...)"""
import os
import sys

sys.path.insert(0, os.getcwd())
import bench
from clair3_amd import synthetic as syn

name, batch = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else max(25, 100 * 256 // batch if name == "full_alignment" else 100 * 1024 // batch)
kind, ch, indel = (syn.FULL_ALIGNMENT, 8, True) if name == "full_alignment" else (syn.PILEUP, 18, False)
model, _ = bench.build_model(kind, ch, indel, 0)
x = syn.make_windows(kind, batch, seed=1000, channels=ch)
slots = int(os.environ.get("RING_SLOTS", "3"))
el, y, els = bench.host_leg_median(model, x, steps, 5, slots=slots)
print(f"{name} B={batch} slots={slots} lanes={os.environ.get('C3HIP_RING_LANES', 'default')} max_batch={os.environ.get('C3HIP_RING_LANES_MAX_BATCH', 'default')}: "
      f"{batch * steps / el:,.0f} windows/s {[round(batch * steps / e) for e in els]}", flush=True)
