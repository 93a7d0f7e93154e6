"""Where the fetch-inclusive step spends its host time: per-phase wall clock of the two-context loop bench.py times."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from maelstrom_amd import engine as E

def main(steps, with_torch):
    cfg = bench.headline_config(E, 1)
    n = 4096
    engs = [E.Engine(cfg, device=0) for _ in range(2)]
    for k, e in enumerate(engs):
        e.run(k * n, n); e.check(); e.fetch()
    torch.cuda.synchronize()
    acc = {"wait": 0.0, "run": 0.0, "check": 0.0, "begin": 0.0}
    began = [False, False]
    msgs = torch.zeros(1, dtype=torch.int64, device='cuda:0'); junk = torch.ones(4096, 6, dtype=torch.int64, device='cuda:0')
    t00 = time.perf_counter()
    for k in range(steps):
        e = engs[k % 2]
        t0 = time.perf_counter()
        if began[k % 2]: e.fetch()
        t1 = time.perf_counter(); e.run((2 + k) * n, n)
        t2 = time.perf_counter(); e.check()
        if with_torch: msgs.add_(junk[:, 0].sum())
        t3 = time.perf_counter(); e.fetch_begin(); began[k % 2] = True
        t4 = time.perf_counter()
        acc["wait"] += t1 - t0; acc["run"] += t2 - t1; acc["check"] += t3 - t2; acc["begin"] += t4 - t3
    for e in engs: e.fetch()
    tot = time.perf_counter() - t00
    print(steps, with_torch, {k: round(v / steps * 1e3, 3) for k, v in acc.items()}, "ms/step", round(tot / steps * 1e3, 3), "kernel_ms", engs[0].kernel_ms())

for steps, wt in ((20, False), (20, True), (200, False), (200, True)):
    main(steps, wt)
