"""world_size-2 test of the N>1 path on CPU (gloo): sharding of the instance range and the variable-length
history gather.  Per-rank outputs come from the CPU oracle here (no GPU in this tier); on the GPU box the same
code runs on engine-owned HBM over RCCL (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maelstrom_amd import engine as E, ensemble as EN
    import oracle_lib as O
    cfg = E.test_config("broadcast", node_count=5, rate=10, time_limit=5, latency=10, seed=77)
    first, count = EN.shard(n_total, rank, world)
    o = O.run(cfg, first, count)
    rows = torch.from_numpy(o.rows.view(np.int32).reshape(count, cfg.max_rows, 4).copy())
    pay = torch.from_numpy(o.payload.view(np.int32).copy())
    meta = torch.from_numpy(o.meta.view(np.int32).reshape(count, 8).copy())
    crow, cpay, nr, nw = EN.compact(rows, pay, meta)
    parts, nbytes = EN.gather_histories(crow, cpay, nr, nw, dist, world)
    msgs = torch.tensor([int(o.stats["all_send"].sum())], dtype=torch.int64)
    dist.all_reduce(msgs)
    if rank == 0:
        full = O.run(cfg, 0, n_total)
        ok = True
        inst = 0
        for r in range(world):
            rr, pp, nrr, nww = parts[r]
            ro = po = 0
            for k in range(len(nrr)):
                a, b = int(nrr[k]), int(nww[k])
                want_r, want_p = full.history(inst)
                ok &= rr[ro:ro + a].numpy().tobytes() == want_r.tobytes()
                ok &= pp[po:po + b].numpy().tobytes() == want_p.tobytes()
                ro += a; po += b; inst += 1
        ok &= inst == n_total and int(msgs) == int(full.stats["all_send"].sum())
        ok &= nbytes == int(full.meta["n_rows"].sum()) * 16 + int(full.meta["n_payload_words"].sum()) * 4
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def _worker_root(rank, world, port, n_total, q):
    """gather-to-root (the exchange of msim_gather, csrc/gather.cpp): byte counts all-gathered, every slab sent once"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maelstrom_amd import engine as E, ensemble as EN
    import oracle_lib as O
    cfg = E.test_config("broadcast", node_count=5, rate=10, time_limit=5, latency=10, seed=78)
    first, count = EN.shard(n_total, rank, world)
    o = O.run(cfg, first, count)
    rows = torch.from_numpy(o.rows.view(np.int32).reshape(count, cfg.max_rows, 4).copy())
    pay = torch.from_numpy(o.payload.view(np.int32).copy())
    meta = torch.from_numpy(o.meta.view(np.int32).reshape(count, 8).copy())
    crow, cpay, nr, nw = EN.compact(rows, pay, meta)
    as_u8 = lambda t: t.contiguous().view(torch.uint8).reshape(-1)
    parts = [as_u8(crow), as_u8(cpay), as_u8(meta), torch.from_numpy(o.stats.view(np.uint8).copy())]
    got, received = EN.gather_to_root(parts, dist, world, rank, root=1)   # a root that is not rank 0
    if rank == 1:
        full = O.run(cfg, 0, n_total)
        want_rows = b"".join(full.history(i)[0].tobytes() for i in range(n_total))
        want_pay = b"".join(full.history(i)[1].tobytes() for i in range(n_total))
        ok = got[0].numpy().tobytes() == want_rows and got[1].numpy().tobytes() == want_pay
        ok &= got[2].numpy().tobytes() == full.meta.tobytes() and got[3].numpy().tobytes() == full.stats.tobytes()
        f0, c0 = EN.shard(n_total, 0, world)
        ok &= received == sum(len(full.history(i)[0]) * 16 + len(full.history(i)[1]) * 4 for i in range(f0, f0 + c0)) + c0 * (32 + 48)
        q.put(bool(ok))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_layout():
    from maelstrom_amd import ensemble as EN
    offs, totals = EN.gather_layout([(16, 4, 32, 48), (0, 0, 0, 0), (160, 44, 64, 96)])
    assert offs == [[0, 16, 16], [0, 4, 4], [0, 32, 32], [0, 48, 48]] and totals == [176, 48, 96, 144]


def test_two_rank_gather_to_root_gloo():
    from maelstrom_amd import build
    build.build(verbose=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_root, args=(r, 2, port, 9, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_covers_range_exactly():
    from maelstrom_amd import ensemble as EN
    for total in (1, 7, 8, 4096, 65536 + 3):
        for world in (1, 2, 3, 8):
            spans = [EN.shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_two_rank_history_gather_gloo():
    from maelstrom_amd import build
    build.build(verbose=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
