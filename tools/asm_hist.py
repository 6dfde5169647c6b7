#!/usr/bin/env python3
"""Per-region instruction histogram of one kernel in a hipcc -S listing (where do spills / loads / MFMAs sit?).
Usage: tools/asm_hist.py file.hip mangled-substring [lines-per-region]"""
import collections, re, subprocess, sys
src, key = sys.argv[1], sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 600
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
                "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", "/tmp/asm_hist.s", src], check=True,
               stderr=subprocess.DEVNULL)
s = open("/tmp/asm_hist.s").read()
i = s.index(key + ":") if (key + ":") in s else s.index(key)
k = s[i:]
k = k[:k.index(".end_amdhsa_kernel")]
lines = k.split("\n")
print(len(lines), "lines")
for start in range(0, len(lines), step):
    c = collections.Counter()
    for l in lines[start:start + step]:
        m = re.match(r"\s*([a-z_0-9]+)", l)
        if not m:
            continue
        op = m.group(1)
        for pre, tag in (("scratch_store", "sst"), ("scratch_load", "sld"), ("v_mfma", "mfma"), ("global_load_lds", "dma"),
                         ("global_load", "gld"), ("global_atomic", "atom"), ("ds_read", "dsr"), ("ds_write", "dsw"),
                         ("ds_bpermute", "bperm"), ("v_readlane", "rdl"), ("v_writelane", "wrl"), ("v_", "valu"), ("s_", "salu")):
            if op.startswith(pre):
                c[tag] += 1
                break
    print(start, dict(c))
