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
