"""In-tree build of libtagdust_hip.so (hand-written HIP for gfx950 + the C-ABI host layer).

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU box with the
repo snapshot.  -ffp-contract=off: the arithmetic contract is plain IEEE float32 adds/multiplies in the
reference's order (an FMA would change logsum's table index)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtagdust_hip.so")
SOURCES = ["td_kernels.hip", "td_api.hip"]
HEADERS = ["td_device.h", os.path.join("..", "..", "include", "tagdust_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
