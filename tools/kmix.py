"""Instruction mix of the loops of one kernel's ISA (output of tools/kisa.sh): per loop (between a label that is a loop
header and its back-edge) VALU / SALU / LDS / VMEM counts.  usage: python tools/kmix.py k.s"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().splitlines()
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
for a, b in loops:
    c = Counter()
    ops = Counter()
    for l in lines[a:b + 1]:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m or l.strip().startswith(";"):
            continue
        op = m.group(1)
        cls = ("VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_")
               else "VMEM" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else "other")
        c[cls] += 1
        ops[re.sub(r"_e32|_e64|_sdwa", "", op)] += 1
    print(f"loop lines {a}-{b} ({b - a} lines): " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())))
    if len(sys.argv) > 2:
        print("   " + ", ".join(f"{k} {v}" for k, v in ops.most_common(28)))
