#!/usr/bin/env python3
"""Where does svc4_kernel (four lin-kv-proxy clusters per wavefront, csrc/svc4.hip) overtake svc_kernel<> (one)?  Both layouts at several batch
sizes and shapes (MSIM_DEV_FLAGS bit 9 = one cluster per wavefront, bit 10 = the packed layout whatever the launch)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maelstrom_amd import engine as E  # noqa: E402

SHAPES = {
    "demo n=5 c=10 rate30 60s lat5 lin-kv": dict(node_count=5, concurrency=10, rate=30, time_limit=60, latency=5, proxy_service="lin-kv"),
    "n=5 c=10 rate300 20s lat20 exp p_loss 0.02 lww-kv": dict(node_count=5, concurrency=10, rate=300, time_limit=20, latency=20, latency_dist="exponential", p_loss=0.02, proxy_service="lww-kv"),
    "n=3 c=6 rate100 30s lat5 + partitions lin-kv": dict(node_count=3, concurrency=6, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=5, proxy_service="lin-kv"),
}
for name, kw in SHAPES.items():
    for n in (512, 1024, 2048, 4096, 8192, 16384):
        out = {"shape": name, "clusters": n}
        for lay, flags in (("one", 0x200), ("four", 0x400)):
            cfg = E.test_config("lin-kv", bin="lin-kv-proxy", seed=99, **kw)
            with E.Engine(cfg) as eng:
                eng.set_dev_flags(flags)
                eng.run(0, n)
                eng.run(n, n)
                out[lay] = round(eng.kernel_ms()[0], 2)
        print(json.dumps(out), flush=True)
