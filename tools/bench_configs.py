#!/usr/bin/env python3
"""Runs the non-headline BASELINE.json configs that the engine supports and prints one line each
(msgs/s, checked histories/s, kernel ms).  Parity for these is covered by tests/; this is measurement only."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maelstrom_amd import engine as E  # noqa: E402

CONFIGS = {
    "cfg1 echo n=3": (dict(workload="echo", node_count=3, rate=5, time_limit=10), 4096),
    "cfg2 broadcast n=25 grid lat0": (dict(workload="broadcast", node_count=25, rate=100, time_limit=20, inbox_capacity=6), 4096),
    "cfg2 broadcast n=25 grid lat10": (dict(workload="broadcast", node_count=25, rate=100, time_limit=20, latency=10), 4096),
    "cfg2 broadcast n=25 grid lat100": (dict(workload="broadcast", node_count=25, rate=100, time_limit=20, latency=100), 4096),
    "cfg2 broadcast n=25 grid lat100 exponential": (dict(workload="broadcast", node_count=25, rate=100, time_limit=20, latency=100, latency_dist="exponential"), 4096),
    "cfg2 broadcast n=25 total lat100": (dict(workload="broadcast", node_count=25, rate=100, time_limit=20, latency=100, topology="total"), 1024),
    "broadcast n=25 ack-retry + partitions": (dict(workload="broadcast", bin="broadcast-ack-retry", node_count=25, rate=100, time_limit=20, latency=10,
                                                   nemesis=["partition"], nemesis_interval=10), 2048),
    "g-set n=25 lat100 exponential p_loss 0.05": (dict(workload="g-set", node_count=25, rate=100, time_limit=20, latency=100, latency_dist="exponential", p_loss=0.05), 4096),
    "cfg3 g-set n=100 lat100 exponential": (dict(workload="g-set", node_count=100, rate=100, time_limit=20, latency=100, latency_dist="exponential"), 16384),
    "cfg3 g-set n=100 lat100 exponential p_loss 0.05": (dict(workload="g-set", node_count=100, rate=100, time_limit=20, latency=100, latency_dist="exponential", p_loss=0.05), 16384),
    "cfg3 g-set n=100 lat100 exponential p_loss 0.5": (dict(workload="g-set", node_count=100, rate=100, time_limit=20, latency=100, latency_dist="exponential", p_loss=0.5), 16384),
    "cfg4 lin-kv raft n=5 c=10 rate30 60s": (dict(workload="lin-kv", bin="raft", node_count=5, rate=30, time_limit=60), 8192),
    "cfg4 lin-kv raft + partitions lat10": (dict(workload="lin-kv", bin="raft", node_count=5, rate=30, time_limit=60, latency=10, nemesis=["partition"], nemesis_interval=10), 8192),
    # the reference's own demo invocation of the proxy node (core.clj:112: lin_kv_proxy.rb, concurrency 10) over the linearizable service
    "lin-kv proxy n=5 c=10 rate30 60s lat5": (dict(workload="lin-kv", bin="lin-kv-proxy", node_count=5, concurrency=10, rate=30, time_limit=60, latency=5, proxy_service="lin-kv"), 16384),
    "unique-ids over lin-tso n=3 rate1000 10s lat5 + partitions": (dict(workload="unique-ids", bin="tso-ids", node_count=3, rate=1000, time_limit=10, latency=5, nemesis=["partition"], nemesis_interval=3), 16384),
    "pn-counter n=5 rate100 20s lat100 exponential": (dict(workload="pn-counter", node_count=5, rate=100, time_limit=20, latency=100, latency_dist="exponential"), 16384),
    "g-counter n=5 rate100 20s lat10": (dict(workload="g-counter", node_count=5, rate=100, time_limit=20, latency=10), 16384),
    "unique-ids n=3 rate1000 10s lat5 + partitions": (dict(workload="unique-ids", node_count=3, rate=1000, time_limit=10, latency=5, nemesis=["partition"], nemesis_interval=3), 16384),
    "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions": (dict(workload="txn-list-append", node_count=5, rate=100, time_limit=30, latency=5,
                                                                     nemesis=["partition"], nemesis_interval=10), 32768),
    # the single-root node as the reference's own runs of this workload are invoked (doc/05-datomic/01-single-node.md:257,322: one node, --concurrency 10n)
    "txn-list-append n=1 c=10 rate100 30s lat5 (single-root node)": (dict(workload="txn-list-append", node_count=1, concurrency=10, rate=100, time_limit=30, latency=5), 16384),
    "txn-list-append n=5 c=10 rate100 30s lat5 + partitions (single-root node)": (dict(workload="txn-list-append", node_count=5, concurrency=10, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10), 16384),
    # the multi-key node (multi_key_txn.js; same architecture as core.clj:113-114's datomic_list_append.rb, a different program): thunks in lww-kv, root map in lin-kv
    "cfg5-mk txn-list-append multi-key n=5 rate100 30s lat5 + partitions": (dict(workload="txn-list-append", bin="multi-key-txn", node_count=5, rate=100, time_limit=30, latency=5,
                                                                                  nemesis=["partition"], nemesis_interval=10), 32768),
    # the node core.clj:113-114 runs (demo/ruby/datomic_list_append.rb): a persistent hash tree in lww-kv, the root pointer in lin-kv, a lock per node
    "cfg5-datomic txn-list-append datomic n=5 rate100 30s lat5 + partitions": (dict(workload="txn-list-append", bin="datomic", node_count=5, rate=100, time_limit=30, latency=5,
                                                                                    nemesis=["partition"], nemesis_interval=10), 32768),
    # ... and the same node as the reference's own runs invoke it (doc/05-datomic/01-single-node.md:257,322: one node, --concurrency 10n)
    "txn-list-append datomic n=1 c=10 rate100 30s lat0": (dict(workload="txn-list-append", bin="datomic", node_count=1, concurrency=10, rate=100, time_limit=30), 16384),
    "broadcast n=100 grid lat0": (dict(workload="broadcast", node_count=100, rate=100, time_limit=20), 2048),
    "broadcast n=100 grid lat100 exponential": (dict(workload="broadcast", node_count=100, rate=100, time_limit=20, latency=100, latency_dist="exponential"), 2048),
    # the reference's own demo invocation for this workload (core.clj:115-121): 2 nodes, rate 100, partitions, read-committed
    "kafka n=5 rate100 20s lat5 + partitions": (dict(workload="kafka", node_count=5, rate=100, time_limit=20, latency=5, nemesis=["partition"], nemesis_interval=10), 16384),
    "txn-rw-register hat n=2 rate100 30s + partitions": (dict(workload="txn-rw-register", node_count=2, rate=100, time_limit=30,
                                                               nemesis=["partition"], nemesis_interval=10), 16384),
    "txn-rw-register hat n=5 rate100 30s lat5 + partitions": (dict(workload="txn-rw-register", node_count=5, rate=100, time_limit=30, latency=5,
                                                                    nemesis=["partition"], nemesis_interval=10), 4096),
}


# configurations whose launches leave the chip half empty towards their end (one slow wavefront sets a launch's time): also measured with three
# batches in flight, one engine context and HIP stream each (tools/overlap_configs.py has the sweep over the depth)
IN_FLIGHT = {"cfg2 broadcast n=25 grid lat0", "cfg2 broadcast n=25 grid lat10", "cfg2 broadcast n=25 grid lat100", "cfg2 broadcast n=25 grid lat100 exponential",
             "cfg2 broadcast n=25 total lat100", "broadcast n=25 ack-retry + partitions"}


def amortised_ms(cfg, n, depth=3, batches=12):
    """Steady-state ms per batch (simulation + check of every history) with `depth` batches in flight."""
    engs = [E.Engine(cfg) for _ in range(depth)]
    try:
        for j, e in enumerate(engs):
            e.run(j * n, n); e.check()
        t0 = time.perf_counter()
        for k in range(batches):
            e = engs[k % depth]
            if k >= depth:
                e.check()
            e.run_async((depth + k) * n, n)
        for e in engs[: min(depth, batches)]:
            e.check()
        return (time.perf_counter() - t0) / batches * 1e3
    finally:
        for e in engs:
            e.close()


def main():
    only = sys.argv[1:] or list(CONFIGS)
    for name in only:
        kw, n = CONFIGS[name]
        cfg = E.test_config(seed=99, **kw)
        with E.Engine(cfg) as eng:
            eng.run(0, n)                      # warm-up (allocation, code load, pinned host buffers of the host-side checkers)
            eng.check()
            t0 = time.perf_counter()
            eng.run(n, n)
            eng.check()
            dt = time.perf_counter() - t0
            sim_ms, chk_ms = eng.kernel_ms()
            eng.fetch()
            msgs = sum(int(eng.net_stats_raw(i).all_send) for i in range(n))
            flagged = sum(1 for i in range(n) if eng.meta(i).flags)
            flag_or = 0
            for i in range(n):
                flag_or |= eng.meta(i).flags
            res = eng.check_results()
            valid = int((res["valid"] == 1).sum())
            host_rechecks = eng.check_host_rechecks()
        out = {"config": name, "instances": n, "msgs_per_s": msgs / dt, "histories_per_s": valid / dt, "valid": valid, "flagged": flagged, "flags_seen": flag_or,
               "msgs_per_instance": msgs / n, "sim_ms": sim_ms, "check_ms": chk_ms, "check_host_rechecks": host_rechecks}
        if name in IN_FLIGHT and not os.environ.get("MSIM_BENCH_CONFIGS_NO_OVERLAP"):
            am = amortised_ms(cfg, n)
            out.update({"three_in_flight_ms_per_batch": am, "three_in_flight_msgs_per_s": msgs / (am * 1e-3), "three_in_flight_histories_per_s": valid / (am * 1e-3)})
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
