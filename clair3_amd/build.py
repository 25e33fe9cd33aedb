"""Build libc3hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m clair3_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libc3hip.so")
SOURCES = ["c3_model.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "c3hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
         "-fgpu-flush-denormals-to-zero" if False else "-fno-gpu-flush-denormals-to-zero"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_hash():
    """sha256 over every source and header the library is compiled from (+ the flags): c3_version() carries its first
    12 hex digits, so a binary that does not correspond to the tree is detectable on any box (tests/test_abi.py)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def built_hash():
    """the source hash embedded in the existing binary (None if there is none), read without loading the library"""
    if not os.path.exists(LIB):
        return None
    with open(LIB, "rb") as fh:
        blob = fh.read()
    i = blob.find(b"srchash:")
    return blob[i + 8:i + 20].decode("ascii", "replace") if i >= 0 else None


def stale():
    return built_hash() != source_hash()


def build(force=False, verbose=False, extra=()):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc()] + FLAGS + [f'-DC3HIP_SRC_HASH="{source_hash()}"'] + list(extra) + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
