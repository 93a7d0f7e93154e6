#!/bin/bash
# Developer tool: compiles one translation unit to ISA and prints register use, spills and static instruction counts by loop depth of
# one kernel of it (a regular expression over the mangled name).
# Usage: tools/isa_probe.sh [unit.hip] [kernel regex] [out.s]      default: duo.hip, the headline instantiation sim_kernel_duo<true,true,false>
set -e
UNIT=${1:-duo.hip}
KERNEL=${2:-sim_kernel_duoILb1ELb1ELb0EE}
OUT=${3:-/tmp/isa_probe.s}
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o "$OUT" maelstrom_amd/csrc/$UNIT 2>/dev/null
python3 - "$OUT" "$KERNEL" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
m = re.search(r"^(_Z\S*%s\S*):(.*?)\.end_amdhsa_kernel" % sys.argv[2], txt, re.S | re.M)
print(m.group(1))
body = m.group(2)
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
