#!/usr/bin/env python3
"""How long the first model call after fork() takes (the reference's stage-B loop forks its decode pool right after its first model
call: profiles/r05_k_*), in a process that has / has not imported torch, and has / has not let torch touch the GPU.  Needs an MI355X.
    python tests/diag/fork_stall.py [none|import|cuda] [forks]"""
import os
import sys
import time

if os.environ.get("C3_FS_NO_THP"):  # no transparent huge pages for this process (prctl PR_SET_THP_DISABLE), before anything is allocated
    import ctypes as _ct
    print("PR_SET_THP_DISABLE ->", _ct.CDLL(None).prctl(41, 1, 0, 0, 0), open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), flush=True)
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
forks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if mode in ("import", "cuda"):
    import torch
    if mode == "cuda":
        torch.zeros(1).cuda()
        torch.cuda.synchronize()
from clair3_amd import synthetic as syn  # noqa: E402
from clair3_amd.model import Clair3_F  # noqa: E402

N = int(os.environ.get("C3_FS_N", "2000"))
if os.environ.get("C3_FS_KIND") == "pileup":
    from clair3_amd.model import Clair3_P
    sd = syn.make_state_dict(syn.PILEUP, 18, False, seed=1)
    m = Clair3_P(add_indel_length=False, predict=True, input_channels=18)
    x = syn.make_windows(syn.PILEUP, N, seed=2, channels=18)
else:
    sd = syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=1)
    m = Clair3_F(add_indel_length=True, predict=True, input_channels=8)
    x = syn.make_fa_windows(N, seed=2)
m.to("cuda:0")
m.eval()
m.load_state_dict(sd)
for _ in range(5):
    m.predict_numpy(x)


kind = os.environ.get("C3_FS_CALL", "host")
if kind == "device":
    xd = torch.from_numpy(x).cuda()


def call():
    t0 = time.perf_counter()
    if kind == "host":
        m.predict_numpy(x)
    elif kind == "small":
        m.predict_numpy(x[:8])
    elif kind == "device":
        m(xd)
        torch.cuda.synchronize()
    elif kind == "sync":
        m.synchronize()
    return 1e3 * (time.perf_counter() - t0)


steady = min(call() for _ in range(5))
kids = []
t0 = time.perf_counter()
sel = os.environ.get("C3_FS_DONTFORK")  # experiment: mark anonymous private writable mappings MADV_DONTFORK before the fork (the child
# will not survive it; only the parent's stall is of interest).  "all" | "heap" | "big" (>= 1 MB) | "small" | "lo-hi" index range
if sel:
    import ctypes
    libc_ = ctypes.CDLL(None, use_errno=True)
    libc_.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    cand = []
    for ln in open("/proc/self/maps"):
        f = ln.split()
        if f[1] != "rw-p" or (len(f) > 5 and f[5] not in ("[heap]",)) or f[4] != "0":
            continue
        lo, hi = (int(v, 16) for v in f[0].split("-"))
        cand.append((lo, hi, f[5] if len(f) > 5 else ""))
    picked = []
    for i, (lo, hi, nm) in enumerate(cand):
        mb = (hi - lo) / 2 ** 20
        take = sel == "all" or (sel == "heap" and nm == "[heap]") or (sel == "big" and mb >= 1 and nm != "[heap]") or (sel == "small" and mb < 1 and nm != "[heap]")
        if sel.startswith("mb:"):  # mappings whose size lies in [lo, hi) MB (the heap excluded)
            lo_, hi_ = (float(v) for v in sel[3:].split(":"))
            take = nm != "[heap]" and lo_ <= mb < hi_
        if "-" in sel and sel.replace("-", "").isdigit():
            a_, b_ = (int(v) for v in sel.split("-"))
            take = a_ <= i < b_
        if take and libc_.madvise(lo, hi - lo, 10) == 0:  # MADV_DONTFORK = 10
            picked.append((i, round(mb, 2), nm))
    from collections import Counter
    print(f"DONTFORK on {len(picked)} of {len(cand)} anonymous rw-p mappings ({sum(p[1] for p in picked):.0f} MB); sizes of all candidates (MB: count): "
          f"{sorted(Counter(round((hi - lo) / 2 ** 20, 1) for lo, hi, _ in cand).items(), reverse=True)[:14]}", flush=True)
how = os.environ.get("C3_FS_FORK", "os")
if how == "libc":
    import ctypes
    libc = ctypes.CDLL(None)
for _ in range(forks):
    if how == "os":
        pid = os.fork()
        if pid == 0:
            time.sleep(15)
            os._exit(0)
    elif how == "libc":  # the bare system call wrapper: none of the interpreter's before / after-fork work
        pid = libc.fork()
        if pid == 0:
            libc.sleep(15)
            libc._exit(0)
    elif how == "spawn":  # posix_spawn / vfork of an unrelated program: no copy of this address space
        import subprocess
        pid = subprocess.Popen(["sleep", "15"]).pid
    kids.append(pid)
t_fork = 1e3 * (time.perf_counter() - t0)
after = [call() for _ in range(6)]
print(f"fork={how} N={N} {os.environ.get('C3_FS_KIND', 'fa')} call: {kind:6s} torch: {mode:6s} (in sys.modules: {'torch' in sys.modules}); {forks} forks in {t_fork:.1f} ms; one call of 2000 windows: steady {steady:.1f} ms, "
      f"after the forks {' '.join('%.1f' % v for v in after)} ms", flush=True)
for pid in kids:
    os.kill(pid, 9)
    os.waitpid(pid, 0)
