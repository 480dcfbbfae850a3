#!/bin/bash
# fp64 instruction counts of the visual evaluation bodies (series-only form), from the ISA (no GPU needed):
#   bash tools/vis_isa_count.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o /tmp/vis_isa_count.s "$R/tools/vis_isa_count.hip"
python3 - <<'PY'
import re
s = open('/tmp/vis_isa_count.s').read()
for name in ("isa_vis_block_small", "isa_vis_anchor_small"):
    i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
    b = s[i:j]
    c = lambda pat: len(re.findall(pat, b))
    fma, mul, add, other = c(r"\bv_fma_f64"), c(r"\bv_mul_f64"), c(r"\bv_add_f64"), c(r"\bv_(rcp|rsq|sqrt|div_fixup|div_fmas|div_scale|log|exp|ldexp|frexp)[a-z_]*f64")
    flops = 2 * fma + mul + add
    print(f"{name}: v_fma_f64 {fma}  v_mul_f64 {mul}  v_add_f64 {add}  (other fp64: {other})  -> {fma + mul + add} fp64 VALU instructions, {flops} flop")
PY
