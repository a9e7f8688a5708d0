#!/usr/bin/env bash
# Per-kernel register / LDS / scratch usage of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
#   tools/kernel_resources.sh pocketflow_amd/csrc/pf_conv_stream.hip
set -euo pipefail
f=$1
cd "$(dirname "$f")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -c "$(basename "$f")" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re, subprocess
cur = {}
def flush():
    if cur:
        name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        print("%-70s vgpr %3s agpr %3s sgpr %3s spill %s/%s scratch %s occ %s lds %s" % (name[:70], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"), cur.get("VGPRs Spill"), cur.get("SGPRs Spill"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?): (\S+) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        flush(); cur.clear(); cur["name"] = v
    else:
        cur[k] = v
flush()
'
