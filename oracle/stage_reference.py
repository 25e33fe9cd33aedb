#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not product code): stage the reference's own Python modules for the hot path into the
git-ignored ``oracle/_ref/`` so that they travel to the GPU box with the snapshot, the way the built ``.so`` files do.

The GPU box has no ``/root/reference``.  What the parity tests and ``bench.py``'s ``cpu_baseline`` leg need there is
the reference itself -- its worker loop (clair3/CallVariantsFromCffi.py:186-381), its decoder
(clair3/CallVariants.py:1069-1394), its modules (clair3/model.py) and the ``shared`` / ``preprocess`` modules those
import -- so that
  * the UNMODIFIED ``call_variants_from_cffi`` GPU branch can be run against the real ``libc3hip.so`` on an MI355X
    (tests/test_reference_loop_gpu.py), and
  * the reference modules themselves can be timed on the GPU node's host cores (``cpu_baseline.kind = "reference"``).

Nothing is copied into the repository's history: ``oracle/_ref/`` is listed in ``.gitignore`` (and NOT in
``.gpurunignore``).  Only ``tests/``, ``bench.py``'s cpu_baseline leg and ``__graft_entry__`` may read it.

    python oracle/stage_reference.py [--reference /root/reference] [--dest oracle/_ref]
"""
import argparse
import hashlib
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# what the worker loop imports, directly or through `import_module` in clair3.py; python sources only
WANTED = (
    ("clair3.py", None),
    ("clair3", (".py",)),
    ("shared", (".py",)),
    ("preprocess", (".py",)),
)


def stage(reference="/root/reference", dest=None, verbose=True):
    dest = dest or os.path.join(HERE, "_ref")
    if not os.path.isdir(os.path.join(reference, "clair3")):
        if verbose:
            print(f"[stage_reference] {reference} not present: keeping {dest} as it is "
                  f"({'present' if os.path.isdir(os.path.join(dest, 'clair3')) else 'absent'})")
        return os.path.isdir(os.path.join(dest, "clair3"))
    os.makedirs(dest, exist_ok=True)
    digest = hashlib.sha256()
    n_files = 0
    for entry, exts in WANTED:
        src = os.path.join(reference, entry)
        if exts is None:
            shutil.copyfile(src, os.path.join(dest, entry))
            digest.update(open(src, "rb").read())
            n_files += 1
            continue
        for root, dirs, files in os.walk(src):
            dirs[:] = sorted(d for d in dirs if d != "__pycache__")
            rel = os.path.relpath(root, reference)
            for f in sorted(files):
                if not f.endswith(exts):
                    continue
                os.makedirs(os.path.join(dest, rel), exist_ok=True)
                shutil.copyfile(os.path.join(root, f), os.path.join(dest, rel, f))
                digest.update(open(os.path.join(root, f), "rb").read())
                n_files += 1
    with open(os.path.join(dest, "STAGED_FROM"), "w") as fh:
        fh.write(f"{reference}\nfiles {n_files}\nsha256 {digest.hexdigest()}\n")
    if verbose:
        print(f"[stage_reference] {n_files} files of {reference} -> {dest} (sha256 {digest.hexdigest()[:12]})")
    return True


def reference_root():
    """Where tests / bench find the reference: $CLAIR3_REFERENCE, /root/reference (build container), oracle/_ref (GPU box)."""
    for cand in (os.environ.get("CLAIR3_REFERENCE"), "/root/reference", os.path.join(HERE, "_ref")):
        if cand and os.path.isdir(os.path.join(cand, "clair3")):
            return cand
    return None


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--dest", default=None)
    a = ap.parse_args()
    sys.exit(0 if stage(a.reference, a.dest) else 1)
