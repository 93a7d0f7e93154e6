#!/usr/bin/env python3
"""Developer tool: runs a configuration through the engine API and bit-compares every instance with the CPU oracle.

    MSIM_LIB=tools/hipemu/_build/libmaelsim_emu.so python tools/emu_compare.py <case> [<case> ...]      (on the host emulator)
    python tools/emu_compare.py <case> ...                                                              (on the device)

A case is a name from CASES below or a Python dict literal of `engine.test_config` keywords (add "n": instances, "flags": dev flags).
Test infrastructure: the oracle is the checker here, nothing of the product imports this file."""
import ast
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from maelstrom_amd import engine as E  # noqa: E402
import oracle_lib as O  # noqa: E402

CASES = {
    "gset40": dict(workload="g-set", node_count=40, rate=20, time_limit=6, latency=50, latency_dist="exponential", n=2),
    "gset100": dict(workload="g-set", node_count=100, rate=50, time_limit=6, latency=100, latency_dist="exponential", n=1),
    "gset70loss": dict(workload="g-set", node_count=70, rate=30, time_limit=6, latency=30, latency_dist="uniform", p_loss=0.1, n=2),
    "gset50nem": dict(workload="g-set", node_count=50, rate=30, time_limit=12, latency=20, nemesis=["partition"], nemesis_interval=3, n=2),
    "gset64journal": dict(workload="g-set", node_count=64, rate=20, time_limit=6, latency=10, journal_capacity=60000, n=1),
    "bcast100": dict(workload="broadcast", node_count=100, rate=20, time_limit=4, latency=100, latency_dist="exponential", n=1),
    "bcast50tree": dict(workload="broadcast", node_count=50, rate=20, time_limit=4, latency=10, topology="tree3", n=2),
    "bcast40nem": dict(workload="broadcast", node_count=40, rate=20, time_limit=8, latency=10, nemesis=["partition"], nemesis_interval=2, n=2),
    "ack60": dict(workload="broadcast", bin="broadcast-ack-retry", node_count=60, rate=10, time_limit=4, latency=20, p_loss=0.1, n=1),
    "rpc40": dict(workload="broadcast", bin="broadcast-rpc-all", node_count=40, rate=10, time_limit=4, latency=20, n=1),
    "pn40": dict(workload="pn-counter", node_count=40, rate=20, time_limit=6, latency=50, latency_dist="exponential", n=1),
    "duo25": dict(workload="broadcast", node_count=25, rate=50, time_limit=4, n=4),
    "duo25exp": dict(workload="broadcast", node_count=25, rate=50, time_limit=4, latency=100, latency_dist="exponential", n=4),
    "duo25uni": dict(workload="broadcast", node_count=25, rate=50, time_limit=4, latency=30, latency_dist="uniform", n=3),
    "duo25lat10": dict(workload="broadcast", node_count=25, rate=50, time_limit=4, latency=10, n=4),
    "duo9total": dict(workload="broadcast", node_count=9, rate=50, time_limit=4, latency=100, latency_dist="exponential", topology="total", n=4),
}


def compare(kw):
    kw = dict(kw)
    n = kw.pop("n", 2)
    flags = kw.pop("flags", None)
    seed = kw.pop("seed", 7)
    cfg = E.test_config(seed=seed, **kw)
    t0 = time.time()
    ora = O.run(cfg, 0, n)
    t1 = time.time()
    bad = 0
    with E.Engine(cfg, device=0) as eng:
        if flags is not None:
            eng.set_dev_flags(flags)
        eng.run(0, n)
        eng.fetch()
        t2 = time.time()
        for i in range(n):
            rows, pay = eng.raw_history(i)
            orows, opay = ora.history(i)
            m = eng.meta(i)
            st = eng.net_stats_raw(i)
            ok = rows.tobytes() == orows.tobytes() and pay.tobytes() == opay.tobytes()
            ok = ok and m.flags == int(ora.meta[i]["flags"]) and m.n_rounds == int(ora.meta[i]["n_rounds"])
            for f in ("all_send", "all_recv", "clients_send", "clients_recv", "servers_send", "servers_recv"):
                ok = ok and int(getattr(st, f)) == int(ora.stats[i][f])
            if cfg.journal_capacity:
                ok = ok and eng.raw_journal(i).tobytes() == ora.events(i).tobytes()
            if not ok:
                bad += 1
                print(f"  instance {i}: DIFFERS rows {len(rows)}/{len(orows)} rounds {m.n_rounds}/{int(ora.meta[i]['n_rounds'])} flags {m.flags}/{int(ora.meta[i]['flags'])} "
                      f"send {int(st.all_send)}/{int(ora.stats[i]['all_send'])}")
    return bad, n, t1 - t0, t2 - t1


def main():
    rc = 0
    for a in sys.argv[1:] or list(CASES):
        kw = CASES[a] if a in CASES else ast.literal_eval(a)
        bad, n, to, te = compare(kw)
        print(f"{a}: {'OK' if not bad else 'MISMATCH'} ({n - bad}/{n} identical; oracle {to:.1f} s, engine {te:.1f} s)", flush=True)
        rc |= bad != 0
    if os.environ.get("MSIM_GUARD"):   # fenced slabs (csrc/guard.cpp): bytes written outside a slab, over every case above
        import ctypes as C
        from maelstrom_amd import _abi
        lib = _abi.load()
        lib.msim_guard_check.restype = C.c_ulonglong
        n_allocs = C.c_ulonglong(0)
        damaged = int(lib.msim_guard_check(C.byref(n_allocs)))
        print(f"guard: {damaged} damaged byte(s) around {n_allocs.value} slabs (MSIM_GUARD={os.environ['MSIM_GUARD']})", flush=True)
        rc |= damaged != 0
    sys.exit(rc)


if __name__ == "__main__":
    main()
