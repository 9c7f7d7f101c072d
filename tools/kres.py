#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/occupancy per kernel of one HIP source (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                    "-c", src, "-o", "/dev/null"] + extra, capture_output=True, text=True)
cur = None
rows = {}
for line in p.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("(anonymous namespace)::", "")
        cur = re.sub(r"^void ", "", cur)
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(\w[\w ]*): (\S+)", line) or re.search(r":\s+([A-Za-z][\w \[\]/]*): (\S+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    print(f"{k:60s} VGPR={v.get('VGPRs','?'):>4} AGPR={v.get('AGPRs','?'):>3} SGPR={v.get('TotalSGPRs','?'):>3} scratch={v.get('ScratchSize [bytes/lane]','?'):>4} occ={v.get('Occupancy [waves/SIMD]','?')} LDS={v.get('LDS Size [bytes/block]','?')}")
