#!/usr/bin/env python3
"""Developer tool (GPU box): runs a few configurations through the engine and the CPU oracle and prints where they differ — meta,
net stats, the first differing history row decoded, payload — instead of a bare assertion.  Used while bringing up a kernel layout.

    python tools/duo_debug.py [name ...]      (MSIM_DEV_FLAGS=512 keeps the one-cluster-per-wavefront kernels)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from maelstrom_amd import _abi as A  # noqa: E402
from maelstrom_amd import engine as E  # noqa: E402
import oracle_lib as O  # noqa: E402

CASES = {
    "n5-lat0": (dict(workload="broadcast", node_count=5, rate=10, time_limit=5, seed=7), 0, 4),
    "n5-lat10": (dict(workload="broadcast", node_count=5, rate=20, time_limit=5, latency=10, seed=7), 0, 4),
    "n25-lat0": (dict(workload="broadcast", bin="broadcast-ff", node_count=25, rate=100, time_limit=20, inbox_capacity=6, seed=2026), 0, 9),
    "n25-lat10": (dict(workload="broadcast", node_count=25, rate=100, time_limit=20, latency=10, seed=99), 0, 4),
    "n25-lat100": (dict(workload="broadcast", node_count=25, rate=100, time_limit=20, latency=100, seed=99), 0, 4),
    "n25-total": (dict(workload="broadcast", node_count=25, rate=20, time_limit=10, latency=20, topology="total", seed=123), 1000, 3),
    "n32-grid": (dict(workload="broadcast", node_count=32, rate=100, time_limit=10, latency=5, seed=123), 1000, 3),
    "n9-echoback": (dict(workload="broadcast", bin="broadcast-ff-echoback", node_count=9, rate=100, time_limit=10, seed=123), 1000, 3),
    "n25-exp100": (dict(workload="broadcast", node_count=25, rate=100, time_limit=20, latency=100, latency_dist="exponential", seed=99), 0, 4),
    "n25-uni50": (dict(workload="broadcast", node_count=25, rate=100, time_limit=20, latency=50, latency_dist="uniform", seed=42), 100, 4),
    "n5-exp20": (dict(workload="broadcast", node_count=5, rate=20, time_limit=5, latency=20, latency_dist="exponential", seed=3), 0, 6),
    "n25-total-exp": (dict(workload="broadcast", node_count=25, rate=20, time_limit=10, latency=20, latency_dist="exponential", topology="total", seed=123), 1000, 3),
    "n5-exp200-tiny": (dict(workload="broadcast", node_count=5, rate=50, time_limit=5, latency=200, latency_dist="exponential", seed=9, inbox_capacity=2), 0, 6),
    "n9-echoback-uni": (dict(workload="broadcast", bin="broadcast-ff-echoback", node_count=9, rate=100, time_limit=10, latency=30, latency_dist="uniform", seed=123), 1000, 3),
    "raft": (dict(workload="lin-kv", bin="raft", node_count=5, rate=30, time_limit=20, seed=17), 0, 6),
    "raft-lat10": (dict(workload="lin-kv", bin="raft", node_count=5, rate=30, time_limit=20, seed=17, latency=10), 0, 6),
    "raft-exp-loss": (dict(workload="lin-kv", bin="raft", node_count=5, rate=30, time_limit=20, seed=17, latency=20, latency_dist="exponential", p_loss=0.02), 0, 6),
    "raft-part": (dict(workload="lin-kv", bin="raft", node_count=5, rate=30, time_limit=20, seed=17, nemesis=["partition"], nemesis_interval=5, latency=5), 0, 6),
    "raft-n3c6": (dict(workload="lin-kv", bin="raft", node_count=3, concurrency=6, nemesis=["partition"], nemesis_interval=4, time_limit=30, latency=10, latency_dist="uniform", rate=30, seed=17), 0, 7),
    "raft-n1": (dict(workload="lin-kv", bin="raft", node_count=1, rate=50, time_limit=10, seed=5), 0, 5),
    "raft-n4c8": (dict(workload="lin-kv", bin="raft", node_count=4, concurrency=8, rate=40, time_limit=15, latency=3, seed=5), 0, 5),
    "txn": (dict(workload="txn-list-append", node_count=5, rate=100, time_limit=10, seed=17), 0, 9),
    "txn-lat5": (dict(workload="txn-list-append", node_count=5, rate=100, time_limit=10, seed=17, latency=5), 0, 9),
    "txn-part": (dict(workload="txn-list-append", node_count=5, rate=100, time_limit=20, seed=17, latency=5, nemesis=["partition"], nemesis_interval=3), 0, 9),
    "txn-exp-loss": (dict(workload="txn-list-append", node_count=5, rate=100, time_limit=10, seed=17, latency=20, latency_dist="exponential", p_loss=0.05), 0, 9),
    "txn-n3": (dict(workload="txn-list-append", node_count=3, rate=200, time_limit=10, seed=17, latency=2), 0, 9),
    "txn-n7": (dict(workload="txn-list-append", node_count=7, rate=150, time_limit=10, seed=17, latency=10, latency_dist="uniform", key_count=3, max_txn_length=4), 0, 9),
    "txn-len6": (dict(workload="txn-list-append", node_count=5, rate=100, time_limit=10, seed=17, latency=5, key_count=3, max_txn_length=6), 0, 6),
    "txn-n1": (dict(workload="txn-list-append", node_count=1, rate=50, time_limit=5, seed=3), 0, 5),
    "n12-spill": (dict(workload="broadcast", node_count=12, latency=30, rate=300, time_limit=10, inbox_capacity=2, spill_capacity=64, seed=123), 1000, 3),
}


def decode_row(r):
    t = int(r["time_len"]) & 0xFFFFFFFFFFFF
    ln = int(r["time_len"]) >> 48
    pk = int(r["packed"])
    return dict(t_us=t // 1000, len=ln, type=pk & 3, f=(pk >> 2) & 31, err=(pk >> 7) & 15, final=(pk >> 11) & 1, process=pk >> 12, value=int(r["value"]))


def run_case(name):
    kw, first, n = CASES[name]
    cfg = E.test_config(**kw)
    ora = O.run(cfg, first, n)
    t0 = time.time()
    with E.Engine(cfg) as eng:
        eng.run(first, n)
        dt = time.time() - t0
        eng.fetch()
        nbad = 0
        for i in range(n):
            m = eng.meta(i)
            om = ora.meta[i]
            st = eng.net_stats_raw(i)
            gst = tuple(getattr(st, f) for f, _ in A.NetStats._fields_)
            ost = tuple(int(x) for x in ora.stats[i])
            gm = (m.n_rows, m.n_payload_words, m.flags, m.n_rounds)
            omt = (int(om["n_rows"]), int(om["n_payload_words"]), int(om["flags"]), int(om["n_rounds"]))
            rows, pay = eng.raw_history(i)
            orows, opay = ora.history(i)
            ok = gm == omt and gst == ost and rows.tobytes() == orows.tobytes() and pay.tobytes() == opay.tobytes()
            if ok:
                continue
            nbad += 1
            print(f"  [{name}] instance {first + i}: meta gpu {gm} oracle {omt}")
            print(f"      stats gpu {gst}\n      stats ora {ost}")
            k = min(len(rows), len(orows))
            d = np.nonzero((rows[:k]["time_len"] != orows[:k]["time_len"]) | (rows[:k]["packed"] != orows[:k]["packed"]) | (rows[:k]["value"] != orows[:k]["value"]))[0]
            if len(d):
                j = int(d[0])
                print(f"      first differing row {j} of {len(rows)}/{len(orows)} ({len(d)} differ):")
                for jj in range(max(0, j - 2), min(k, j + 3)):
                    print(f"        {jj}: gpu {decode_row(rows[jj])}\n        {' ' * len(str(jj))}  ora {decode_row(orows[jj])}")
            kp = min(len(pay), len(opay))
            dp = np.nonzero(pay[:kp] != opay[:kp])[0]
            if len(dp):
                print(f"      payload: {len(dp)} of {kp} words differ, first at {int(dp[0])}: gpu {int(pay[dp[0]]):#x} ora {int(opay[dp[0]]):#x}")
        print(f"[{name}] {n - nbad}/{n} instances identical  (engine wall {dt * 1e3:.1f} ms, sim kernel {eng.kernel_ms()[0]:.3f} ms)", flush=True)
    return nbad


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    bad = 0
    for nm in names:
        try:
            bad += run_case(nm)
        except Exception as ex:  # keep going: one GPU call should report on every case
            print(f"[{nm}] EXCEPTION {type(ex).__name__}: {ex}", flush=True)
            bad += 1
    sys.exit(1 if bad else 0)
