"""msim_gather (include/maelsim.h "multi-GPU ensemble", csrc/gather.cpp) on one device: the device-side compaction and the
root's own part of the exchange.  With a communicator of one rank RCCL is initialised for real (ncclCommInitRank) and the gather
takes the same code path as on N GPUs minus the peers; the N > 1 exchange itself is covered on CPU over gloo
(tests/test_ensemble_gloo.py, same layout code) and by bench.py --gpus N."""
import numpy as np
import pytest
import torch

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E

pytestmark = pytest.mark.gpu


def _view(ptr, nbytes):
    class W:
        pass
    w = W()
    w.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(w, device="cuda:0").cpu().numpy()


@pytest.mark.parametrize("with_comm", [False, True])
def test_gather_on_one_device_equals_fetch(lib, with_comm):
    cfg = E.test_config("broadcast", node_count=5, rate=20, time_limit=5, latency=10, seed=5)
    n = 37
    with E.Engine(cfg) as eng:
        if with_comm:
            eng.comm_init(E.Engine.comm_unique_id(), 0, 1)
        eng.run(100, n)
        g = eng.gather(0)
        assert (g.world, g.rank, g.n_instances, g.bytes_received) == (1, 0, n, 0) and g.ms > 0
        rows, pay = _view(g.rows, g.rows_bytes), _view(g.payload, g.payload_bytes)
        meta, stats = _view(g.meta, g.meta_bytes).view(E.META_DT), _view(g.stats, g.stats_bytes).view(E.STATS_DT)
        eng.fetch()
        want_rows = b"".join(eng.raw_history(i)[0].tobytes() for i in range(n))
        want_pay = b"".join(eng.raw_history(i)[1].tobytes() for i in range(n))
        assert rows.tobytes() == want_rows and pay.tobytes() == want_pay
        for i in range(n):
            m = eng.meta(i)
            assert (int(meta[i]["n_rows"]), int(meta[i]["n_payload_words"]), int(meta[i]["flags"])) == (m.n_rows, m.n_payload_words, m.flags)
            assert int(stats[i]["all_send"]) == eng.net_stats_raw(i).all_send
        g2 = eng.gather(0)   # buffers are reused
        assert g2.rows_bytes == g.rows_bytes


_RANK_SCRIPT = r"""
import hashlib, json, os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
from maelstrom_amd import engine as E
rank, world, root, tmp = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
torch.cuda.set_device(rank)
cfg = E.test_config("broadcast", node_count=5, rate=20, time_limit=5, latency=10, seed=5)
n = 19 + 7 * rank                     # ragged shards
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
with E.Engine(cfg, device=rank) as eng:
    eng.comm_init(open(idf, "rb").read(), rank, world)
    eng.run(first, n)
    g = eng.gather(root)
    out = {"rank": rank, "n": n, "world": g.world, "bytes_received": int(g.bytes_received)}
    if rank == root:
        def view(ptr, nbytes):
            class W: pass
            w = W(); w.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(w, device=f"cuda:{rank}").cpu().numpy().tobytes()
        out["n_instances"] = int(g.n_instances)
        out["rows"] = hashlib.sha256(view(g.rows, g.rows_bytes)).hexdigest(); out["rows_bytes"] = int(g.rows_bytes)
        out["payload"] = hashlib.sha256(view(g.payload, g.payload_bytes)).hexdigest(); out["payload_bytes"] = int(g.payload_bytes)
        out["meta"] = hashlib.sha256(view(g.meta, g.meta_bytes)).hexdigest()
        out["stats"] = hashlib.sha256(view(g.stats, g.stats_bytes)).hexdigest()
    eng.fetch()                        # what this rank holds itself, instance by instance
    out["own_rows"] = b"".join(eng.raw_history(i)[0].tobytes() for i in range(n)).hex()
    out["own_payload"] = b"".join(eng.raw_history(i)[1].tobytes() for i in range(n)).hex()
    db = eng.device_buffers()
    def dview(ptr, nbytes):
        class W: pass
        w = W(); w.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(w, device=f"cuda:{rank}").cpu().numpy().tobytes()
    out["own_meta"] = dview(db.meta, db.meta_bytes).hex(); out["own_stats"] = dview(db.stats, db.stats_bytes).hex()
with open(os.path.join(tmp, f"out{rank}.json"), "w") as f:
    json.dump(out, f)
"""


@pytest.mark.parametrize("root", [0, 1])
def test_gather_over_rccl_between_two_devices(lib, tmp_path, root):
    """The N > 1 branch of csrc/gather.cpp (ncclAllGather of the sizes, grouped ncclSend / ncclRecv to the root) on two real devices,
    one process per device: the root's slabs must be every rank's fetched histories back to back in rank order, bytes_received the
    peers' bytes only.  Skipped on a one-GPU box (gpurun's boxes are; the driver's 8-GPU node is not)."""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), root_dir, str(r), "2", str(root), str(tmp_path)], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    outs = [json.load(open(tmp_path / f"out{r}.json")) for r in range(2)]
    g = outs[root]
    assert g["world"] == 2 and g["n_instances"] == outs[0]["n"] + outs[1]["n"]
    cat = lambda key: b"".join(bytes.fromhex(o[key]) for o in outs)
    assert g["rows"] == hashlib.sha256(cat("own_rows")).hexdigest() and g["rows_bytes"] == len(cat("own_rows"))
    assert g["payload"] == hashlib.sha256(cat("own_payload")).hexdigest() and g["payload_bytes"] == len(cat("own_payload"))
    assert g["meta"] == hashlib.sha256(cat("own_meta")).hexdigest() and g["stats"] == hashlib.sha256(cat("own_stats")).hexdigest()
    peer = outs[1 - root]
    assert g["bytes_received"] == (len(peer["own_rows"]) + len(peer["own_payload"]) + len(peer["own_meta"]) + len(peer["own_stats"])) // 2
    assert peer["bytes_received"] == 0
