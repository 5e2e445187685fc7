#!/usr/bin/env python3
"""Instruction mix of the LAST loop (the steady-state loop) of every kernel in a hipcc -S listing:
   tools/asm_count.py file.s [name filter]"""
import collections
import re
import sys

t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_ZN3xlb\S+):[^\n]*\n(.*?)^\.Lfunc_end", t, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    lines = body.split("\n")
    # the largest backward-branch loop
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = (0, 0, 0)
    for i, l in enumerate(lines):
        mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i and i - labels[mm.group(1)] > best[0]:
            best = (i - labels[mm.group(1)], labels[mm.group(1)], i)
    loop = [l.strip() for l in lines[best[1]:best[2]] if l.strip() and not l.strip().startswith((";", "."))]
    c = collections.Counter()
    for l in loop:
        op = l.split()[0]
        k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
        c[k] += 1
        if op.startswith("v_pk"):
            c["v_pk"] += 1
        if op in ("s_waitcnt", "s_barrier", "s_nop"):
            c[op] += 1
    print(name[13:60], "loop lines", len(loop), dict(c))
