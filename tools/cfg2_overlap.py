#!/usr/bin/env python3
"""cfg2's latency sweep with several batches in flight ("attack the tail"): the slowest wavefront of a batch sets its kernel's
duration while every other SIMD idles (profiles/r04a_latency_probe.jsonl).  D engine contexts, each with its own HIP stream (msim_run_async on
the context's stream), hold batches k, k+1, .. in flight together: the head of the next launch fills the tail of the last.  Prints, per latency
shape and per D, the steady-state ms per batch of 4096 (simulation + set-full check of every history) beside the single-batch kernel time.

    python tools/cfg2_overlap.py [--batches 24] [--instances 4096]
Measurement only; parity of these shapes is tests/test_bench_shapes_gpu.py."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maelstrom_amd import engine as E  # noqa: E402

SHAPES = {
    "lat0": dict(latency=0, inbox_capacity=6),
    "lat10": dict(latency=10),
    "lat100": dict(latency=100),
    "exp100": dict(latency=100, latency_dist="exponential"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--instances", type=int, default=4096)
    ap.add_argument("--depths", default="1,2,3,4")
    ap.add_argument("--shapes", default=",".join(SHAPES))
    a = ap.parse_args()
    n = a.instances
    for name in a.shapes.split(","):
        cfg = E.test_config("broadcast", node_count=25, rate=100, time_limit=20, topology="grid", seed=99, **SHAPES[name])
        single = None
        for depth in [int(x) for x in a.depths.split(",")]:
            engs = [E.Engine(cfg) for _ in range(depth)]
            try:
                for j, e in enumerate(engs):   # warm-up: slabs, code
                    e.run(j * n, n); e.check()
                if single is None:
                    engs[0].run(0, n); single = engs[0].kernel_ms()[0]
                valid = msgs = 0
                t0 = time.perf_counter()
                for k in range(a.batches):
                    e = engs[k % depth]
                    if k >= depth:
                        e.check()          # waits for this context's batch (k - depth), checks it
                    e.run_async((depth + k) * n, n)
                for e in engs[: min(depth, a.batches)]:
                    e.check()
                dt = time.perf_counter() - t0
                e = engs[(a.batches - 1) % depth]
                res = e.check_results()
                valid = int((res["valid"] == 1).sum())
                e.fetch()
                msgs = sum(int(e.net_stats_raw(i).all_send) for i in range(0, n, 64)) * 64
                flagged = sum(1 for i in range(n) if e.meta(i).flags)
            finally:
                for e in engs:
                    e.close()
            print(json.dumps({"shape": name, "instances_per_batch": n, "contexts_in_flight": depth, "batches": a.batches, "ms_per_batch": dt / a.batches * 1e3,
                              "single_batch_kernel_ms": single, "histories_per_sec": n * a.batches / dt, "msgs_per_sec_est": msgs * a.batches / dt,
                              "valid_last_batch": valid, "flagged_last_batch": flagged}), flush=True)


if __name__ == "__main__":
    main()
