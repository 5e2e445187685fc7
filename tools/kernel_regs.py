#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of every kernel in a hipcc -S listing:  tools/kernel_regs.py file.s [filter]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
    name, body = m.group(1), m.group(2)
    get = lambda k: (re.search(r"\." + k + r" (\d+)", body) or [None, "?"])[1]  # noqa: E731
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void xlb::", "")
    if flt in dem:
        print(f"{dem:90s} vgpr {get('amdhsa_next_free_vgpr'):>4s} agpr_off {get('amdhsa_accum_offset'):>4s} sgpr {get('amdhsa_next_free_sgpr'):>4s} scratch {get('amdhsa_private_segment_fixed_size'):>5s} lds {get('amdhsa_group_segment_fixed_size'):>7s}")
