#!/usr/bin/env python3
"""Where the wall time of the reference's stage-B loop goes (clair3/CallVariantsFromCffi.py:260-371: tensor generator -> model call ->
shared memory + ProcessPoolExecutor.submit -> batch_output in the decode processes -> VCF write) when it runs on libc3hip with the
decoder columns: the UNMODIFIED loop, in this process, with timers wrapped around the functions callvar.install() rebinds and around
the pool.  Needs an MI355X and the reference checkout.

    python tests/diag/loop_timeline.py [windows per file] [files] [cpu_threads] [pileup|full_alignment]
"""
import json
import os
import runpy
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "stubs"))
from clair3_amd import synthetic as syn  # noqa: E402
from tests import refloop  # noqa: E402

per_file = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
files = int(sys.argv[2]) if len(sys.argv) > 2 else 60
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
name = sys.argv[4] if len(sys.argv) > 4 else "full_alignment"
kind, channels, indel, pileup = (syn.PILEUP, 18, False, True) if name == "pileup" else (syn.FULL_ALIGNMENT, 8, True, False)

ref = refloop.reference_root()
d = tempfile.mkdtemp(prefix="c3_timeline_")
import atexit, shutil  # noqa: E401,E402
atexit.register(shutil.rmtree, d, ignore_errors=True)
lst = refloop.write_job(d, kind, [per_file] * files, channels=channels)
ck = os.path.join(d, "model.pt")
refloop.write_checkpoint(ck, kind, channels, indel)
LOG = os.path.join(d, "events")
fd = os.open(LOG, os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)


def ev(tag, t0, t1, n=0):
    os.write(fd, f"{os.getpid()} {tag} {t0:.6f} {t1:.6f} {n}\n".encode())


sys.path.insert(0, ref)
from clair3_amd import callvar  # noqa: E402
callvar.install(decoder=os.environ.get("C3_TL_DECODER", "1") == "1")
import clair3.CallVariantsFromCffi as w  # noqa: E402
import concurrent.futures as cf  # noqa: E402

MAIN = os.getpid()
_predict, _gen, _worker, _load = w._torch_predict, w.tensor_generator_for_chunk, w.batch_output_worker, w._load_torch_checkpoint


def predict(m, device, X):
    t0 = time.time()
    y = _predict(m, device, X)
    ev("predict", t0, time.time(), len(X))
    return y


def gen(gen_cls, args, batch_size=50):
    it = _gen(gen_cls, args, batch_size=batch_size)
    while True:
        t0 = time.time()
        try:
            b = next(it)
        except StopIteration:
            ev("generator_end", t0, time.time())
            return
        ev("generator", t0, time.time(), len(b[0]))
        yield b


def worker(*a, **k):
    t0 = time.time()
    r = _worker(*a, **k)
    ev("decode", t0, time.time(), len(a[0]))
    return r


MODELS = []


def load(m, path, device):
    MODELS.append(m)
    t0 = time.time()
    r = _load(m, path, device)
    ev("load", t0, time.time())
    return r


_pool_factory = w.ProcessPoolExecutor  # callvar's: hands out the pool it forked before the device was in use (C3HIP_PREFORK_POOL)


def Pool(*a, **k):
    t0 = time.time()
    ex = _pool_factory(*a, **k)
    ev("pool_init", t0, time.time(), len(getattr(ex, "_processes", {}) or {}))  # n = processes that exist already
    _submit, _exit = ex.submit, type(ex).__exit__

    def submit(*sa, **sk):
        t0 = time.time()
        f = _submit(*sa, **sk)
        ev("submit", t0, time.time())
        return f

    ex.submit = submit

    class Timed(type(ex)):
        def __exit__(self, *xa):
            t0 = time.time()
            r = _exit(self, *xa)
            ev("pool_exit", t0, time.time())
            return r

    ex.__class__ = Timed
    return ex


# the pool pickles the task function by name: the wrapper answers to the name it replaces (this script is not importable as __main__
# while runpy runs clair3.py)
worker.__module__, worker.__qualname__, worker.__name__ = w.__name__, "batch_output_worker", "batch_output_worker"
w._torch_predict, w.tensor_generator_for_chunk, w.batch_output_worker, w._load_torch_checkpoint = predict, gen, worker, load
w.ProcessPoolExecutor = Pool
_as_completed = w.as_completed


def as_completed(fs, *a, **k):  # the loop takes ONE future per call and breaks: the time to that first future is the wait
    t0 = time.time()
    for f in _as_completed(fs, *a, **k):
        ev("wait_one_done", t0, time.time(), len(fs))
        yield f
        t0 = time.time()


w.as_completed = as_completed
from multiprocessing import shared_memory as _shm  # noqa: E402
_SharedMemory = _shm.SharedMemory


class SharedMemory(_SharedMemory):
    def __init__(self, *a, **k):
        t0 = time.time()
        super().__init__(*a, **k)
        if os.getpid() == MAIN:
            ev("shm_open", t0, time.time())


_shm.SharedMemory = SharedMemory
_time = w.time
marks = []


def marked_time():  # the loop reads the clock twice: at "Calling variants ..." and for "Total time elapsed"
    t = _time()
    marks.append(time.time())
    return t


w.time = marked_time
vcf = os.path.join(d, "out.vcf")
argv = ["CallVariantsFromCffi", "--chkpnt_fn", ck, "--bam_fn", "unused.bam", "--call_fn", vcf, "--sampleName", "SAMPLE", "--platform",
        "ont", "--use_gpu", "True", "--cpu_threads", str(threads), "--threads", "4", "--output_tensor_can_fn_list", lst, "--gpu_id", "0"]
if pileup:
    argv.append("--pileup")
if indel:
    argv.append("--add_indel_length")
os.chdir(d)
sys.argv = [os.path.join(ref, "clair3.py")] + argv
t_start = time.time()
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit as e:
    assert not e.code, e.code
if os.getpid() != MAIN:
    os._exit(0)
rows = [ln.split() for ln in open(LOG)]
E = [(int(p), tag, float(a), float(b), int(n)) for p, tag, a, b, n in rows]
t0, t1 = marks[0], marks[-1]
main = [e for e in E if e[0] == MAIN]
dec = [e for e in E if e[1] == "decode"]
tot = lambda tag: sum(b - a for _, t, a, b, _ in main if t == tag)  # noqa: E731
first = lambda tag: min((a for _, t, a, _, _ in E if t == tag), default=t0) - t0  # noqa: E731
n = per_file * files
out = {
    "job": f"{name}: {n} windows in {files} files, {threads} decode processes, decoder columns {os.environ.get('C3_TL_DECODER', '1')}"
           + ", decode pool forked ahead by callvar: C3HIP_PREFORK_POOL=" + os.environ.get("C3HIP_PREFORK_POOL", "1"),
    "loop_seconds (the reference's 'Total time elapsed')": round(t1 - t0, 3),
    "windows_per_s_in_the_loop": round(n / (t1 - t0)),
    "main thread, seconds inside": {
        "generator (files -> batches, submit to the GPU ahead)": round(tot("generator") + tot("generator_end"), 3),
        "model call (wait for the batch's rows)": round(tot("predict"), 3),
        "ProcessPoolExecutor.submit (pickle + fork of a new decode process on demand)": round(tot("submit"), 3),
        "as_completed until one decode is done (pending full, or the generator is exhausted)": round(tot("wait_one_done"), 3),
        "SharedMemory(create=True)": round(tot("shm_open"), 3),
        "pool construction": round(tot("pool_init"), 3),
        "decode processes that existed when the loop asked for its pool": next((k for _, t, _, _, k in main if t == "pool_init"), None),
        "pool exit (join the decode processes)": round(tot("pool_exit"), 3),
    },
    "as_completed calls with fewer than 2 x cpu_threads pending (the tail: generator exhausted)": sum(1 for _, t, _, _, k in main if t == "wait_one_done" and k < 2 * threads),
    "slowest submits (s)": sorted((round(b - a, 3) for _, t, a, b, _ in main if t == "submit"), reverse=True)[:10],
    "first batch out of the generator at (s after loop start)": round(min(b for _, t, a, b, _ in main if t == "generator") - t0, 3),
    "first model call returned at": round(min(b for _, t, a, b, _ in main if t == "predict") - t0, 3),
    "first decode started at": round(first("decode"), 3),
    "last decode ended at": round(max(b for _, _, _, b, _ in dec) - t0, 3),
    "last batch left the generator at": round(max(b for _, t, a, b, _ in main if t == "generator") - t0, 3),
    "decode processes": len({e[0] for e in dec}),
    "decode busy seconds per process": sorted(round(sum(b - a for p, _, a, b, _ in dec if p == q), 3) for q in {e[0] for e in dec}),
    "decode rows/s per busy process": round(n / max(1e-9, sum(b - a for _, _, a, b, _ in dec))),
    "first decode call of each process (s)": sorted((round(min((a, b - a) for p, _, a, b, _ in dec if p == q)[1], 3) for q in {e[0] for e in dec}), reverse=True)[:3],
    "model load (before the loop)": round(sum(b - a for _, t, a, b, _ in main if t == "load"), 3),
}
try:
    out["handle"] = MODELS[0].describe()
except Exception as e:  # noqa: BLE001
    out["handle"] = repr(e)
print(json.dumps(out, indent=1))
