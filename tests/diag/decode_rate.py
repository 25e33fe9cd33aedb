"""Rows per second of one decode core: the reference's batch_output as it is, on decoder columns through its own output_from
(round 2: list look-alikes), and with the row printer (clair3_amd/vcf_rows.py) -- best of five passes over the same rows.
Needs the reference checkout.  python tests/diag/decode_rate.py [rows]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("CLAIR3_REFERENCE", "/root/reference"))
from tests.decode_rows import consistent_rows  # noqa: E402
from tests.test_decode_dropin import config, widen  # noqa: E402
import clair3.CallVariants as cv  # noqa: E402
from clair3_amd import decode, vcf_rows  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
unpatched = cv.batch_output
decode.install_decoder()


def best(fn, reps=5):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return n / min(t)


print(f"{n} rows per case, one core, best of 5; columns appended beforehand (the device does that at 60 M rows/s)")
for indel, name in ((True, "full alignment, 90 columns"), (False, "pileup, 24 columns")):
    for noise in (0.0, 0.05, 0.3):
        pos, alt, y, _ = consistent_rows(n, seed=3, indel=indel, noise=noise)
        cfg = config(cv, not indel, indel)
        yw = widen(y, indel)
        r_ref = best(lambda: unpatched(pos, alt, y, cfg, None), reps=2 if indel else 5)
        r_new = best(lambda: cv.batch_output(pos, alt, yw, cfg, None))
        pr = cv._c3hip_row_printers[(cfg, id(cv.param))]
        usable, pr.usable = pr.usable, False  # round 2's path: the reference's output_with on the look-alike lists
        r_old = best(lambda: cv.batch_output(pos, alt, yw, cfg, None), reps=2 if indel else 5)
        pr.usable = usable
        print(f"  {name}, {int(100 * noise):2d} % of the rows lack the alleles of their best class: reference {r_ref:8,.0f}  "
              f"columns + output_from {r_old:8,.0f}  row printer {r_new:8,.0f} rows/s  (x{r_new / r_ref:.1f})")
