#!/bin/bash
# Developer tool: compiles only the headline kernel (sim_kernel_colo<BCAST_FF, no nemesis, constant latency>) to ISA and
# prints register use, spills and static instruction counts of its innermost loops (the cascade loop is VALU-issue bound,
# DESIGN.md §4.4).  Usage: tools/isa_probe.sh [out.s]
set -e
OUT=${1:-/tmp/isa_probe.s}
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMSIM_ISA_PROBE -S --cuda-device-only -o "$OUT" maelstrom_amd/csrc/engine.hip 2>/dev/null
python3 - "$OUT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
m = re.search(r"^_Z15sim_kernel_coloILi1ELb0ELb0ELb0EEv7KParams:(.*?)\.end_amdhsa_kernel", txt, re.S | re.M)
body = m.group(1)
for k in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy"):
    mm = re.search(r"; %s: (\d+)" % k, txt[m.start():])
    print(k, mm.group(1) if mm else "?")
lines = [l.strip() for l in body.splitlines()]
depth = {}
cur = 0
cnt = {}
for l in lines:
    mm = re.match(r"^\.LBB\d+_\d+:\s*;.*Depth=(\d+)", l)
    if mm:
        cur = int(mm.group(1)); continue
    if re.match(r"^\.LBB\d+_\d+:", l):
        # label without depth annotation: outside loops unless stated
        if "Depth" not in l: cur = cur
        continue
    if not l or l.startswith(";") or l.startswith("."): continue
    op = l.split()[0]
    cls = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "scratch_", "buffer_", "flat_")) else "other"
    cnt.setdefault(cur, {}).setdefault(cls, 0)
    cnt[cur][cls] += 1
    if op == "v_mov_b32_e32" or op == "v_mov_b64_e32":
        cnt[cur]["v_mov"] = cnt[cur].get("v_mov", 0) + 1
for d in sorted(cnt): print("loop depth", d, cnt[d])
PY
