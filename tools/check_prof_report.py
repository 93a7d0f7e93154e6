#!/usr/bin/env python3
"""Developer tool (GPU box): cycles of check_kernel's passes on the headline shape (or cfg3: CFG=cfg3), from a build of csrc/checker.hip with -DCK_PROF
(tools/variant_lib.sh ckprof checker.hip -DCK_PROF)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
os.environ.setdefault("MSIM_LIB", os.path.join(ROOT, "maelstrom_amd", "libmaelsim_ckprof.so"))
import numpy as np
from maelstrom_amd import engine as E
if os.environ.get("CFG") == "cfg3": cfg, n = E.test_config(workload="g-set", node_count=100, rate=100, time_limit=20, latency=100, latency_dist="exponential", seed=99), 16384
else: cfg, n = E.test_config(workload="broadcast", node_count=25, rate=100, time_limit=20, seed=99), 4096
with E.Engine(cfg) as eng:
    eng.run(0, n); eng.check(); eng.run(n, n); eng.check()
    res = eng.check_results(); print("check ms", eng.kernel_ms()[1])
t = res["stable_latency_ms"].astype(np.float64)
tot = t[:, :3].sum(axis=1).mean()
for i, nm in enumerate(["pass 1 (rows: classify, pair, record)", "pass 2 (sweep over the reads' bitmaps)", "outcomes + quantiles"]): print(f"  {nm:44s} {t[:, i].mean():12.3e} cycles {100 * t[:, i].mean() / tot:5.1f} %")
print(f"  pairing rounds per history {t[:, 3].mean():.0f}, reads {t[:, 4].mean():.0f}; cycles per history {tot:.3e} = {tot / 2.4e6:.3f} ms at 2.4 GHz")
