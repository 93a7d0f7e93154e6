"""In-tree build of libmaelsim.so for gfx950 (explicit hipcc; no JIT cache, so the .so travels with the repo).

    python -m maelstrom_amd.build [--force]

hipcc cross-compiles without a GPU.  Every source is compiled to its own object (in parallel, rebuilt only when its
content or a header changed) and the objects are linked into the library.  The library is the product; the oracle is
built separately by oracle/Makefile (see __graft_entry__.build)."""
import concurrent.futures as cf
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libmaelsim.so")
SOURCES = ["config.cpp", "engine.hip",
           # the one-cluster-per-wavefront kernels (csrc/sim_kernels.h), one unit per family or part of one
           "k_general_a.hip", "k_general_b.hip", "k_general_c.hip", "k_wide_gset.hip", "k_wide_bcast.hip", "k_wide_ack.hip", "k_wide_pn.hip",
           "k_raft.hip", "k_svc.hip", "k_txn.hip", "k_mk.hip", "k_dt.hip", "k_kafka.hip", "k_hat.hip",
           "duo.hip", "raft4.hip", "svc4.hip", "txng4.hip", "dtg4.hip", "txn8.hip", "mk8.hip", "dt8.hip", "hat8.hip", "kafka8.hip", "uid8.hip", "crdt8.hip", "bcast8.hip", "checker.hip", "lin_check.cpp", "lin_check_dev.hip", "txn_check.cpp", "txn_check_dev.hip", "rw_check_dev.hip", "pn_check.cpp", "kafka_check.cpp", "kafka_check_dev.hip", "pn_check_dev.hip", "unique_check_dev.hip", "edn.cpp",
           "fressian.cpp", "gather.cpp", "guard.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
LINK_LIBS = ["-ldl"]
# Flags appended for single units — measured, not reasoned: the g-set instantiations of sim_kernel_wide<> (253 registers at -O3) run BASELINE cfg3 at
# 199.2 ms per 16384 clusters built with -O2 against 207 (5 % loss: 301 against 317); the same flag changes nothing for any other unit
# (duo, raft4, txn8, mk8, kafka8, the broadcast wide unit: within 1 %; k_general_c: worse).  profiles/r04y_ack_retry_round_split.txt, call ai.
UNIT_FLAGS = {"k_wide_gset.hip": ["-O2"]}

STAMP = OUT + ".stamp"


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(path, seen=None):
    """The files `path` includes with #include "..." (recursively), resolved like the compiler does: relative to the including file."""
    seen = set() if seen is None else seen
    with open(path, errors="ignore") as f:
        txt = f.read()
    for inc in _INC.findall(txt):
        q = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if q not in seen and os.path.exists(q):
            seen.add(q)
            _deps(q, seen)
    return seen


def _headers_digest():
    """Flags only: every source hashes the headers IT includes (_src_digest), so that a change to one kernel's .inc rebuilds the units
    that include it and nothing else."""
    return hashlib.sha256(" ".join(FLAGS).encode()).hexdigest()


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _src_digest(src, hd):
    h = hashlib.sha256(hd.encode())
    path = os.path.join(CSRC, src)
    h.update(" ".join(UNIT_FLAGS.get(src, [])).encode())
    for d in [path] + sorted(_deps(path)):
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _digest():
    """Content hash of everything the library is built from (mtimes do not survive every copy of the tree)."""
    hd = _headers_digest()
    h = hashlib.sha256(hd.encode())
    for s in _sources():
        h.update(s.encode())
        h.update(_src_digest(s, hd).encode())
    return h.hexdigest()


def _stale():
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _digest()


def _compile(src, hd, force, verbose):
    obj = os.path.join(OBJ, src + ".o")
    stamp = obj + ".stamp"
    dg = _src_digest(src, hd)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == dg:
        return obj
    cmd = ["hipcc"] + FLAGS + UNIT_FLAGS.get(src, []) + ["-c", "-o", obj, os.path.join(CSRC, src)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(dg)
    return obj


def build(force=False, verbose=True):
    if not force and not _stale():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    hd = _headers_digest()
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(s, hd, force, verbose), srcs))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + LINK_LIBS
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(_digest())
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
