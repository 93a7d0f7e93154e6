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
SOURCES = ["config.cpp", "engine.hip", "checker.hip", "lin_check.cpp", "txn_check.cpp", "pn_check.cpp", "edn.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


STAMP = OUT + ".stamp"


def _digest():
    """Content hash of everything the library is built from (mtimes do not survive every copy of the tree)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "maelsim.h")]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale():
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _digest()


def build(force=False, verbose=True):
    if not force and not _stale():
        return OUT
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = ["hipcc"] + FLAGS + ["-o", OUT] + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(_digest())
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
