#!/bin/bash
# Resource usage (VGPRs, spills, scratch, LDS, occupancy) of every kernel of one translation unit, from the compiler's own remarks:
#   scripts/kernel_resources.sh kmu_sketch [filter]
cd "$(dirname "$0")/../kmerutils_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $KMU_BUILD_DEFS \
  -Rpass-analysis=kernel-resource-usage -c "$1.hip" -o /dev/null 2>&1 | python3 -c '
import re, sys, subprocess
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.rsplit(":", 1); cur[k.strip()] = v.strip()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception: name = r["name"]
    name = re.sub(r"\(.*", "", name)
    if flt and flt not in name: continue
    print("%-70s vgpr %3s agpr %3s sgpr %3s  spill v %3s s %3s  scratch %4s B  lds %6s B  occupancy %s" % (
        name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill", r.get("VGPR Spill")), r.get("SGPRs Spill", r.get("SGPR Spill")),
        r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
' "$2"
