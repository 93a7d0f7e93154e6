"""The kernels on the BASELINE configs' hot paths keep nothing in private (scratch) memory: a spill or a stack object inside a round loop is a
memory round trip per round (round 4: the colocated kernels' 48-232 B cost the acknowledged gossip 14 %).  Read from the
metadata of the built library's code objects (tools/private_memory_audit.py) — no GPU needed."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = [r"sim_kernel_duo<", r"raft4_kernel<", r"txn8_kernel<", r"mk8_kernel<", r"kafka8_kernel<", r"uid8_kernel<", r"crdt8_kernel<", r"bcast8_kernel<", r"hat8_kernel<",
       r"sim_kernel_wide<(true|false), [0-4], false, ", r"sim_kernel_colo<3, ", r"check_kernel\(", r"lin_check_kernel\("]


def test_hot_kernels_have_no_private_memory():
    lib = os.path.join(ROOT, "maelstrom_amd", "libmaelsim.so")
    if not os.path.exists(lib) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("no built library / no llvm-readelf on this machine")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "private_memory_audit.py"), lib], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    head = r.stdout.splitlines()[0]
    assert re.match(r"\d+ kernels, \d+ with private memory", head), head
    assert int(head.split()[0]) > 150   # (the parse saw the library's kernels)
    bad = [l for l in r.stdout.splitlines()[1:] if any(re.search(h, l) for h in HOT)]
    assert not bad, "\n".join(bad)
