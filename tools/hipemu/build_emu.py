"""Builds tools/hipemu/_build/libmaelsim_emu.so: the SOURCES of libmaelsim compiled for the host against the wavefront emulator
(tools/hipemu/hip/hip_runtime.h).  Developer / test tool — see that header; the product library is maelstrom_amd/libmaelsim.so,
built by hipcc for gfx950, and nothing in the product loads this one.

    python tools/hipemu/build_emu.py [--force] [source.hip ...]

Use:  MSIM_LIB=tools/hipemu/_build/libmaelsim_emu.so python tools/emu_compare.py ...   (engine API on the emulator vs the oracle)"""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "maelstrom_amd", "csrc")
OBJ = os.path.join(HERE, "_build")
OUT = os.path.join(OBJ, "libmaelsim_emu.so")
sys.path.insert(0, ROOT)
CXX = os.environ.get("HIPEMU_CXX", "g++")
FLAGS = ["-std=c++17", "-O2", "-g1", "-fPIC", "-fno-omit-frame-pointer", "-w", "-I", HERE, "-DMSIM_HIPEMU=1", "-pthread", "-fno-extern-tls-init"]


def _digest(path, extra):
    h = hashlib.sha256(extra.encode())
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def _deps_digest():
    h = hashlib.sha256((CXX + " ".join(FLAGS)).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")))
    deps += [os.path.join(ROOT, "include", "maelsim.h"), os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "rccl", "rccl.h")]
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, only=None):
    from maelstrom_amd.build import SOURCES
    os.makedirs(OBJ, exist_ok=True)
    dd = _deps_digest()
    # gather.cpp included: it compiles against tools/hipemu/rccl/rccl.h (types only) and binds librccl.so at run time like the product —
    # tests put tools/hipemu/_build/librccl.so (rccl_stub.cpp: the same entry points between processes, over a socket) on the path
    jobs = [(os.path.join(CSRC, s), os.path.join(OBJ, s + ".o")) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs.append((os.path.join(HERE, "hipemu.cpp"), os.path.join(OBJ, "hipemu.cpp.o")))

    def one(job):
        src, obj = job
        dg = _digest(src, dd)
        stamp = obj + ".stamp"
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg:
            return obj
        if only and os.path.basename(src) not in only and os.path.exists(obj):
            return obj
        cmd = [CXX] + FLAGS + ["-x", "c++", "-c", "-o", obj, src]
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(stamp, "w") as f:
            f.write(dg)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, jobs))
    cmd = [CXX, "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl", "-pthread"]
    print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stub_src, stub_out = os.path.join(HERE, "rccl_stub.cpp"), os.path.join(OBJ, "librccl.so")
    dg = _digest(stub_src, dd)
    if force or not os.path.exists(stub_out) or not os.path.exists(stub_out + ".stamp") or open(stub_out + ".stamp").read() != dg:
        cmd = [CXX] + FLAGS + ["-shared", "-o", stub_out, stub_src]
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(stub_out + ".stamp", "w") as f:
            f.write(dg)
    return OUT


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    build(force="--force" in sys.argv, only=set(args) or None)
