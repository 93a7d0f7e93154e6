#!/usr/bin/env python3
"""Any tools/bench_configs.py configuration with several batches in flight (one engine context and HIP stream each): the steady-state ms per batch
(simulation + check of every history) beside the single-batch kernel times.  The generalisation of tools/cfg2_overlap.py.
    python tools/overlap_configs.py "<config name>" [...] [--depths 1,2,3] [--batches 12]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_configs  # noqa: E402
from maelstrom_amd import engine as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+")
    ap.add_argument("--depths", default="1,2,3")
    ap.add_argument("--batches", type=int, default=12)
    a = ap.parse_args()
    for name in a.configs:
        kw, n = bench_configs.CONFIGS[name]
        cfg = E.test_config(seed=99, **kw)
        for depth in [int(x) for x in a.depths.split(",")]:
            engs = [E.Engine(cfg) for _ in range(depth)]
            try:
                for j, e in enumerate(engs):
                    e.run(j * n, n); e.check()
                sim1, chk1 = engs[0].kernel_ms()
                t0 = time.perf_counter()
                for k in range(a.batches):
                    e = engs[k % depth]
                    if k >= depth:
                        e.check()
                    e.run_async((depth + k) * n, n)
                for e in engs[: min(depth, a.batches)]:
                    e.check()
                dt = time.perf_counter() - t0
                res = engs[(a.batches - 1) % depth].check_results()
                valid = int((res["valid"] == 1).sum())
            finally:
                for e in engs:
                    e.close()
            print(json.dumps({"config": name, "instances_per_batch": n, "contexts_in_flight": depth, "batches": a.batches, "ms_per_batch": round(dt / a.batches * 1e3, 2),
                              "single_batch_sim_ms": round(sim1, 2), "single_batch_check_ms": round(chk1, 2), "valid_last_batch": valid}), flush=True)


if __name__ == "__main__":
    main()
