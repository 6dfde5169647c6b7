#!/usr/bin/env python3
"""Compile one .hip file for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs / AGPRs / spills / scratch / LDS / occupancy.  Usage: tools/kres.py csrc/file.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
       "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
def flush():
    if cur and flt in cur.get("name", ""):
        n = subprocess.run(["/usr/bin/c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        n = re.sub(r"\(.*", "", n)[-90:]
        print(f"{n:90s} vgpr={cur.get('VGPRs')} agpr={cur.get('AGPRs')} spill={cur.get('VGPRs Spill')} scratch={cur.get('ScratchSize [bytes/lane]')} sgpr={cur.get('TotalSGPRs')} lds={cur.get('LDS Size [bytes/block]')} occ={cur.get('Occupancy [waves/SIMD]')}")
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?): (.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        flush(); cur = {"name": v}
    else:
        cur[k] = v
flush()
