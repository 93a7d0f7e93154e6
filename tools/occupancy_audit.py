#!/usr/bin/env python3
"""Developer tool (no GPU needed): registers and the compiler's occupancy figure of every kernel of the library, and which of the two register
files sets it.  A kernel at <= 64 vector registers whose SCALAR registers exceed 96 gets seven wavefronts per SIMD, not eight — a workgroup
of sixteen wavefronts then has a CU to itself (txn_check_lds_kernel, round 6: 252 histories in flight instead of 512).
    python tools/occupancy_audit.py [unit.hip ...]"""
import concurrent.futures as cf, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maelstrom_amd.build import FLAGS, UNIT_FLAGS, SOURCES, CSRC

def audit(unit):
    out = subprocess.run(["hipcc", *FLAGS, *UNIT_FLAGS.get(unit, []), "-c", "-o", "/dev/null", os.path.join(CSRC, unit), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)", line)
        if not m: continue
        k, v = m.groups()
        if k == "Function Name":
            cur = {"unit": unit, "name": v}; rows.append(cur)
        elif cur is not None: cur[k] = int(v)
    return rows

units = sys.argv[1:] or [s for s in SOURCES if s.endswith(".hip")]
with cf.ThreadPoolExecutor(6) as ex:
    allrows = [r for rs in ex.map(audit, units) for r in rs]
print(f"{'unit':22s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'occ':>4s} {'by vgpr':>8s} {'by sgpr':>8s} {'scratch':>8s}  kernel")
for r in allrows:
    v, a, s = r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("TotalSGPRs", 0)
    tot = ((v + 7) // 8 * 8) + ((a + 7) // 8 * 8)
    by_v = min(8, 512 // max(tot, 8)); by_s = min(8, 800 // max((s + 15) // 16 * 16, 16))
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()[:90]
    mark = "  <-- scalar registers set the occupancy" if by_s < by_v else ""
    print(f"{r['unit']:22s} {v:5d} {a:5d} {s:5d} {r.get('Occupancy [waves/SIMD]', 0):4d} {by_v:8d} {by_s:8d} {r.get('ScratchSize [bytes/lane]', 0):8d}  {name}{mark}")
