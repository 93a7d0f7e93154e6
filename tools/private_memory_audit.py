#!/usr/bin/env python3
"""Developer tool: lists the kernels of maelstrom_amd/libmaelsim.so that use private (scratch) memory — register spills or stack objects —
with their VGPR counts, read from the code objects' metadata (--sgpr: also the scalar registers spilled to lanes of vector registers).  A spill inside a round loop is a memory round trip per round."""
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
lib = args[0] if args else os.path.join(ROOT, "maelstrom_amd", "libmaelsim.so")
data = open(lib, "rb").read()
rows = []
for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
    b = m.start()
    n = struct.unpack_from("<Q", data, b + 24)[0]
    off = b + 32
    for _ in range(n):
        o, s, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        trip = data[off:off + tl].decode()
        off += tl
        if "gfx950" not in trip or not s:
            continue
        fn = "/tmp/_audit_co.elf"
        open(fn, "wb").write(data[b + o:b + o + s])
        t = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", fn], capture_output=True, text=True).stdout
        for blk in t.split("- .agpr_count")[1:]:
            nm, pv, vg = re.search(r"\.name:\s+(\S+)", blk), re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk), re.search(r"\.vgpr_count:\s+(\d+)", blk)
            ss = re.search(r"\.sgpr_spill_count:\s+(\d+)", blk)
            if nm and pv:
                rows.append((int(pv.group(1)), int(vg.group(1)) if vg else -1, nm.group(1), int(ss.group(1)) if ss else 0))
print(f"{len(rows)} kernels, {sum(1 for r in rows if r[0])} with private memory")
names = subprocess.run(["c++filt"], input="\n".join(r[2] for r in rows), capture_output=True, text=True).stdout.splitlines()
for (pv, vg, _, _), nm in sorted(zip(rows, names), key=lambda x: -x[0][0]):
    if pv:
        print(f"{pv:6d} B  {vg:4d} VGPRs  {nm[:150]}")
if "--sgpr" in sys.argv:   # scalar registers spilled to lanes of vector registers (a v_writelane / v_readlane per use)
    print("scalar-register spills (count, VGPRs):")
    for (_, vg, _, ss), nm in sorted(zip(rows, names), key=lambda x: -x[0][3])[:40]:
        if ss:
            print(f"{ss:6d}    {vg:4d} VGPRs  {nm[:150]}")
