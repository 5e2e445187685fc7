#!/usr/bin/env python3
"""Merge the PMC traffic of one tools/profile.sh output directory into profiles/traffic.json, stamped with the hash of the
kernel sources it was measured on (bench.py quotes a figure only for the build it belongs to).

    python tools/update_traffic.py gpurun_out/prof_r02_cavity_halfway_512 D3Q19_BGK_FP32FP32_cavity_halfway_512 profiles/r02/cavity_halfway_512_summary.md
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402


def main():
    prof, key, where = sys.argv[1], sys.argv[2], sys.argv[3]
    text = open(os.path.join(prof, "summary.md")).read()
    sec = text.split("## HBM traffic of the bench kernels")[1]
    entry = {"source_hash": kernel_source_hash(), "profile": where}
    for line in sec.splitlines():
        m = re.match(r"\| `(k_step2?)<.*?` \|.*\| ([0-9.e+]+) \|$", line)
        if m:
            entry[m.group(1)] = int(float(m.group(2)))
    path = os.path.join(ROOT, "profiles", "traffic.json")
    table = json.load(open(path))
    table[key] = entry
    table["_comment"] = ("HBM bytes PER KERNEL LAUNCH from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) calibrated on a copy kernel of known "
                         "traffic as MI355X_MICROARCH.md 'HBM' prescribes (gfx950: FETCH_SIZE unit = 2048 B measured, WRITE_SIZE unit = 1024 B). k_step2 "
                         "performs TWO steps per launch, k_step one. Each entry carries the hash of xlb_amd/csrc it was measured on (bench.py: "
                         "kernel_source_hash); bench.py reports traffic null for any other build. Written by tools/update_traffic.py.")
    json.dump(table, open(path, "w"), indent=2)
    print(key, entry)


if __name__ == "__main__":
    main()
