#!/usr/bin/env python
"""Per-kernel register / scratch / LDS table from `hipcc -Rpass-analysis=kernel-resource-usage` (cross-compiled, no GPU).
usage: python tools/resusage.py [pattern]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "ctrl-vio_amd", "csrc", "ctvio.hip")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: [^:]*:?\s*(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"ctv::|\(.*", "", name)}
        rows.append(cur)
        continue
    m = re.search(r"\s(SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'kernel':52s} {'VGPR':>5s} {'AGPR':>5s} {'vspill':>6s} {'scratch':>7s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    if pat in r["name"]:
        print(f"{r['name'][:52]:52s} {r.get('VGPRs',0):5d} {r.get('AGPRs',0):5d} {r.get('VGPRs Spill',0):6d} {r.get('ScratchSize',0):7d} {r.get('LDS Size',0):7d} {r.get('Occupancy',0):4d}")
