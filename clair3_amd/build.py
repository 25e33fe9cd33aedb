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


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=()):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc()] + FLAGS + list(extra) + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
