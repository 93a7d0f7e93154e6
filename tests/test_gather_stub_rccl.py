"""The N > 1 branch of msim_gather (csrc/gather.cpp:130-165: ncclAllGather of the byte counts, one group of ncclSend / ncclRecv to the
root, the root's own part device-to-device) EXECUTED on a machine without GPUs: gather.cpp itself is compiled into the host wavefront
emulator's library (tools/hipemu/build_emu.py) and binds `librccl.so` at run time exactly as the product does — here
tools/hipemu/_build/librccl.so (tools/hipemu/rccl_stub.cpp: the same nine entry points between processes of one machine, over a
Unix-domain socket hub).  Two (and three) processes, one engine context each, ragged shards, every root: the root's slabs must be every
rank's own fetched histories back to back in rank order and equal what maelstrom_amd.ensemble.gather_to_root's layout code gives;
bytes_received is the peers' bytes only.  Test infrastructure on both sides; the real RCCL run is tests/test_gather_gpu.py (two devices)
and bench.py --gpus N."""
import ctypes
import hashlib
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tools", "hipemu", "_build")

_RANK_SCRIPT = r"""
import ctypes, json, os, sys, time
sys.path.insert(0, sys.argv[1])
from maelstrom_amd import engine as E
rank, world, root, tmp = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
cfg = E.test_config("broadcast", node_count=5, rate=20, time_limit=3, latency=10, seed=5)
n = 3 + 2 * rank                     # ragged shards
first = 1000 * rank
idf = os.path.join(tmp, "rccl_id")
if rank == 0:
    with open(idf + ".tmp", "wb") as f:
        f.write(E.Engine.comm_unique_id())
    os.rename(idf + ".tmp", idf)
t0 = time.time()
while not os.path.exists(idf):        # the host's own channel for the id (here: a file)
    assert time.time() - t0 < 60
    time.sleep(0.05)
view = lambda ptr, nbytes: ctypes.string_at(ptr, int(nbytes)) if nbytes else b""   # (the emulator's device memory is host memory)
with E.Engine(cfg, device=0) as eng:
    eng.comm_init(open(idf, "rb").read(), rank, world)
    eng.run(first, n)
    g = eng.gather(root)
    out = {"rank": rank, "n": n, "world": g.world, "bytes_received": int(g.bytes_received)}
    if rank == root:
        out["n_instances"] = int(g.n_instances)
        for k in ("rows", "payload", "meta", "stats"):
            out[k] = view(getattr(g, k), getattr(g, k + "_bytes")).hex()
    g2 = eng.gather(root)              # a second exchange over the same communicator
    out["again"] = int(g2.bytes_received) == out["bytes_received"] and (rank != root or view(g2.rows, g2.rows_bytes).hex() == out["rows"])
    eng.fetch()                        # what this rank holds itself, instance by instance
    out["own_rows"] = b"".join(eng.raw_history(i)[0].tobytes() for i in range(n)).hex()
    out["own_payload"] = b"".join(eng.raw_history(i)[1].tobytes() for i in range(n)).hex()
    db = eng.device_buffers()
    out["own_meta"] = view(db.meta, db.meta_bytes).hex(); out["own_stats"] = view(db.stats, db.stats_bytes).hex()
with open(os.path.join(tmp, f"out{rank}.json"), "w") as f:
    json.dump(out, f)
"""


@pytest.fixture(scope="module")
def emu_env():
    if shutil.which(os.environ.get("HIPEMU_CXX", "g++")) is None:
        pytest.skip("no host C++ compiler for the emulator build")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hipemu", "build_emu.py")], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(os.path.join(EMU_DIR, "librccl.so"))
    return dict(os.environ, MSIM_LIB=os.path.join(EMU_DIR, "libmaelsim_emu.so"), HIPEMU_DIVERGENT="1",
                LD_LIBRARY_PATH=EMU_DIR + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,root", [(2, 0), (2, 1), (3, 1), (8, 0), (8, 3), (8, 7)])   # 8 = the ranks of one MI355X node, ragged shards of 3 .. 17 instances
def test_gather_branch_for_several_ranks_runs_and_equals_the_ranks_own_histories(emu_env, tmp_path, world, root):
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), str(root), str(tmp_path)], env=emu_env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    outs = [json.load(open(tmp_path / f"out{r}.json")) for r in range(world)]
    g = outs[root]
    assert g["world"] == world and g["n_instances"] == sum(o["n"] for o in outs)
    cat = lambda key: b"".join(bytes.fromhex(o[key]) for o in outs)
    assert bytes.fromhex(g["rows"]) == cat("own_rows") and bytes.fromhex(g["payload"]) == cat("own_payload")
    assert bytes.fromhex(g["meta"]) == cat("own_meta") and bytes.fromhex(g["stats"]) == cat("own_stats")
    peers = [o for o in outs if o["rank"] != root]
    assert g["bytes_received"] == sum(len(o["own_rows"]) + len(o["own_payload"]) + len(o["own_meta"]) + len(o["own_stats"]) for o in peers) // 2
    assert all(o["bytes_received"] == 0 for o in peers) and all(o["again"] for o in outs)
    # the same layout as the torch.distributed twin (maelstrom_amd/ensemble.py) computes for its transport
    import numpy as np
    from maelstrom_amd import ensemble
    sizes = np.array([[len(o["own_rows"]) // 2, len(o["own_payload"]) // 2, len(o["own_meta"]) // 2, len(o["own_stats"]) // 2] for o in outs], dtype=np.uint64)
    offs, totals = ensemble.gather_layout(sizes)
    assert [int(t) for t in totals] == [len(g[k]) // 2 for k in ("rows", "payload", "meta", "stats")]
    for r, o in enumerate(outs):
        assert bytes.fromhex(g["rows"])[int(offs[0][r]):int(offs[0][r]) + len(o["own_rows"]) // 2] == bytes.fromhex(o["own_rows"])


def test_stub_is_test_infrastructure_only():
    """Nothing of the product refers to the stub: libmaelsim dlopens "librccl.so" by name and the in-tree build never links or ships one."""
    for base in ("maelstrom_amd", "include", "integration"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".h", ".inc", ".c", ".clj")):
                    assert "rccl_stub" not in open(os.path.join(dp, f), errors="ignore").read(), f
