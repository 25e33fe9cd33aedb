"""Soak of the row printer (clair3_amd/vcf_rows.py) against the unpatched reference decoder: many seeds of story rows, flat
rows (softmax of small logits: close calls, long rejection chains) and adversarial alt_info.  Needs the reference checkout.
python tests/diag/decode_soak.py [rounds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("CLAIR3_REFERENCE", "/root/reference"))
from tests.decode_rows import consistent_rows  # noqa: E402
from tests.test_decode_dropin import alt_infos, config, widen  # noqa: E402
import clair3.CallVariants as cv  # noqa: E402
from clair3_amd import decode  # noqa: E402

unpatched = cv.batch_output
decode.install_decoder()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
total = taken = retried = back = 0
t_fast = t_ref = 0.0
for rnd in range(rounds):
    for indel in (True, False):
        cfg = config(cv, not indel, indel, is_show_reference=bool(rnd % 2), quality_score_for_pass=(None, 8)[rnd % 2])
        rng = np.random.default_rng(100 + rnd)
        sets = []
        for noise in (0.0, 0.1, 0.5):
            sets.append(consistent_rows(1200, seed=1000 * rnd + int(100 * noise) + indel, indel=indel, noise=noise,
                                        sharp=float(rng.uniform(2.0, 9.0)))[:3])
        # flat rows: four soft-maxes of N(0, s) logits, story or adversarial alt_info
        n = 1200
        widths = (21, 3, 33, 33) if indel else (21, 3)
        s = float(rng.uniform(0.3, 3.0))
        parts = []
        for w in widths:
            e = np.exp(rng.normal(0, s, size=(n, w)))
            parts.append(e / e.sum(axis=1, keepdims=True))
        flat = np.concatenate(parts, axis=1).astype(np.float32)
        p_a, a_a = alt_infos(n, seed=rnd)
        sets.append((p_a, a_a, flat))
        p_s, a_s, _ = consistent_rows(n, seed=rnd + 77, indel=indel, noise=0.2)[:3]
        sets.append((p_s, a_s, flat))
        for pos, alt, y in sets:
            yw = widen(y, indel)  # the decoder columns (oracle/decode_oracle.py here; the device appends them to the rows)
            t0 = time.perf_counter()
            want = unpatched(pos, alt, y, cfg, None)
            t1 = time.perf_counter()
            got = cv.batch_output(pos, alt, yw, cfg, None)
            t2 = time.perf_counter()
            t_ref += t1 - t0
            t_fast += t2 - t1
            pr = cv._c3hip_row_printers[(cfg, id(cv.param))]
            if got != want:
                for g, w in zip(got.splitlines(), want.splitlines()):
                    if g != w:
                        print("DIFF\n  got ", g, "\n  want", w)
                        break
                sys.exit(1)
            total += len(y)
    print(f"round {rnd}: {total} rows identical so far", flush=True)
for pr in cv._c3hip_row_printers.values():
    taken += pr.taken
    retried += pr.retried
    back += pr.handed_back
print(f"{total} rows: identical text; {taken} printed from the columns ({retried} after rejected candidates), {back} handed back to "
      f"output_with; reference {total / t_ref:,.0f} rows/s, with the columns {total / t_fast:,.0f} rows/s (widening excluded: the device does it)")
