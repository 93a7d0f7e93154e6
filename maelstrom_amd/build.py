"""In-tree build of libmaelsim.so for gfx950 (explicit hipcc; no JIT cache, so the .so travels with the repo).

    python -m maelstrom_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is the product; the oracle is built separately by
oracle/Makefile (see __graft_entry__.build)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmaelsim.so")
SOURCES = ["config.cpp", "engine.hip", "checker.hip", "lin_check.cpp", "txn_check.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "maelsim.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return OUT
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = ["hipcc"] + FLAGS + ["-o", OUT] + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
