#!/usr/bin/env python
"""Instruction counts of k_imu_linearize_f64's pass loop (64 samples, one per lane), read off the ISA -- no GPU needed.
  python tools/imu_isa_count.py            (compiles ctrl-vio_amd/csrc/ctvio.hip to assembly first: ~20 s)
The pass loop is the depth-2 loop of the kernel (the walk over groups is depth 1); the six MFMA chains inside it are depth-3 loops whose
trip count is kmax / 8 (two K-steps per trip).  Everything else in the loop body runs once per pass.  Printed: static counts per category,
the dynamic counts of a full pass (kmax = 64) and of the second pass of a 100-sample group (kmax = 36), and the issue-slot model of
DESIGN.md section 4: fp64 vector instructions at 4 cycles (16 lanes per cycle), v_mfma_f64_16x16x4 at 64 cycles, one datapath."""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = "/tmp/ctvio_imu.s"
KERNEL = "_ZN3ctv19k_imu_linearize_f64ENS_3DevEiii"


def cat(l):
    l = l.strip()
    if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
        return None
    op = l.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "lane"
    if op.startswith("v_accvgpr"): return "accvgpr_mov"
    if op.startswith("v_") and ("_f64" in op): return "valu_f64"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    if "--no-compile" not in sys.argv:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                               "-o", ASM, os.path.join(ROOT, "ctrl-vio_amd", "csrc", "ctvio.hip")], stderr=subprocess.DEVNULL)
    s = open(ASM).read()
    i = s.index(KERNEL + ":"); j = s.index(".Lfunc_end", i)
    b = s[i:j].split("\n")
    # loops: label lines carry LLVM's loop annotations
    hdr = [(n, l) for n, l in enumerate(b) if re.match(r"^\.LBB\d+_\d+:", l)]
    depth2 = [n for n, l in hdr if "Parent Loop" in l and "Depth=2" not in l and "Depth=3" not in l]
    # the pass loop = the depth-2 loop that contains MFMAs
    mf = [n for n, l in enumerate(b) if "v_mfma" in l]
    loops = []   # (header line, last line) of every loop found through backward branches
    lab = {l.split(":")[0]: n for n, l in hdr}
    for n, l in enumerate(b):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < n:
            loops.append((lab[m.group(1)], n))
    cand = [(a, z) for a, z in loops if a < mf[0] and z > mf[-1]]
    a, z = max(cand, key=lambda t: t[0])          # innermost loop around all MFMAs = the pass loop
    inner = sorted((x, y) for x, y in loops if x > a and y < z and any(x <= q <= y for q in mf))
    print(f"pass loop: lines {a}..{z} of the kernel's {len(b)}; {len(inner)} MFMA chain loops inside")
    stat, chain = Counter(), Counter()
    in_chain = lambda n: any(x <= n <= y for x, y in inner)
    for n in range(a, z + 1):
        k = cat(b[n])
        if k:
            (chain if in_chain(n) else stat)[k] += 1
    cats = ["valu_f64", "valu_other", "accvgpr_mov", "lane", "mfma", "lds", "vmem", "salu", "waitcnt", "nop"]
    print(f"{'category':14s} {'once per pass':>14s} {'per trip, all 6 chains':>24s} {'full pass (8 trips)':>20s} {'kmax = 36 (5 trips)':>20s}")
    dyn = {}
    for c in cats:
        dyn[c] = (stat[c] + 8 * chain[c], stat[c] + 5 * chain[c])
        print(f"{c:14s} {stat[c]:14d} {chain[c]:24d} {dyn[c][0]:20d} {dyn[c][1]:20d}")
    for name, idx in (("full pass", 0), ("kmax = 36", 1)):
        v = 4 * (dyn["valu_f64"][idx] + dyn["valu_other"][idx] + dyn["accvgpr_mov"][idx] + dyn["lane"][idx])
        m = 64 * dyn["mfma"][idx]
        print(f"issue-slot model, {name}: vector 4 x {v // 4} = {v} cycles + matrix 64 x {dyn['mfma'][idx]} = {m} cycles = {v + m} cycles on the shared fp64 datapath")
    full = 4 * sum(dyn[c][0] for c in ("valu_f64", "valu_other", "accvgpr_mov", "lane")) + 64 * dyn["mfma"][0]
    part = 4 * sum(dyn[c][1] for c in ("valu_f64", "valu_other", "accvgpr_mov", "lane")) + 64 * dyn["mfma"][1]
    # config 2: 2000 samples in 21 groups (20 of 100 samples: one full pass + one of 36 lanes): 2048 windows on 1024 SIMDs
    per_window = 20 * (full + part) + 1 * (full + part) * 0   # (the 21st group of a window is the same shape in the synthetic windows: 95..100 samples)
    per_window = 21 * (full + part)
    for ghz in (2.4, 2.1):
        print(f"2048 config-2 windows (21 groups of ~95-100 samples = 42 passes each) on 1024 SIMDs: {2 * per_window} cycles per SIMD = {2 * per_window / ghz / 1e3:.0f} us at {ghz} GHz")


if __name__ == "__main__":
    main()
