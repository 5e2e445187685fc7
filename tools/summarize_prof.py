#!/usr/bin/env python3
"""Condenses one tools/profile.sh output directory into a markdown summary (kernel durations from
--kernel-trace, HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes, calibrated on the
copy kernels of known traffic as MI355X_MICROARCH.md 'HBM' prescribes)."""

import csv
import glob
import os
import sys
from collections import defaultdict


def rows(d, suffix):
    for path in glob.glob(os.path.join(d, "**", f"*{suffix}.csv"), recursive=True):
        with open(path, newline="") as fh:
            yield from csv.DictReader(fh)


def short(name):
    name = name.split("(")[0]
    for a, b in (("xlb::", ""), ("void ", "")):
        name = name.replace(a, b)
    return name[:110]


def durations(d):
    acc = defaultdict(list)
    for r in rows(d, "kernel_trace"):
        acc[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc


def counters(d):
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows(d, "counter_collection"):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    out = sys.argv[1]
    print(f"# rocprofv3 summary: {os.path.basename(out)}\n")
    dur = durations(os.path.join(out, "kt"))
    print("## kernel-trace (rocprofv3 --kernel-trace --stats), bench.py workload\n")
    print("| kernel | calls | avg us | min us | max us |\n|---|---:|---:|---:|---:|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"| `{k}` | {len(v)} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} |")
    cal = {}
    print("\n## counter calibration on copy kernels of known traffic (tools/copy_bw.py, 512^3 x 19 x 4 B)\n")
    print("| kernel | counter | avg value per launch | known bytes | bytes per counter unit |\n|---|---|---:|---:|---:|")
    for sub, cname in (("cal_fetch", "FETCH_SIZE"), ("cal_write", "WRITE_SIZE")):
        c = counters(os.path.join(out, sub))
        for k, cs in c.items():
            if "k_copy" not in k or cname not in cs:
                continue
            v = cs[cname]
            avg = sum(v) / len(v)
            known = None
            for line in open(os.path.join(out, sub + ".log")):
                if "GB each way" in line:
                    known = float(line.split("GB each way")[0].split()[-1]) * 1e9
                    break
            if known and avg > 0:
                width = "u4" if ("__vector" in k or "ext_vector" in k or "Dv4" in k) else "u32"
                cal[(cname, width)] = known / avg
                print(f"| `{k}` | {cname} | {avg:.4g} | {known:.4g} | {known / avg:.1f} |")
    print("\n## HBM traffic of the bench kernels (separate --pmc passes)\n")
    print("| kernel | FETCH_SIZE avg | WRITE_SIZE avg | read bytes (calibrated, 4 B/lane factor) | write bytes | total per launch |\n|---|---:|---:|---:|---:|---:|")
    f = counters(os.path.join(out, "pmc_fetch"))
    w = counters(os.path.join(out, "pmc_write"))
    kf = cal.get(("FETCH_SIZE", "u32"), 1024.0)
    kw = cal.get(("WRITE_SIZE", "u32"), 1024.0)
    for k in f:
        if "k_step" not in k:
            continue
        fa = sum(f[k]["FETCH_SIZE"]) / max(len(f[k]["FETCH_SIZE"]), 1)
        wv = w.get(k, {}).get("WRITE_SIZE", [0.0])
        wa = sum(wv) / max(len(wv), 1)
        print(f"| `{k}` | {fa:.4g} | {wa:.4g} | {fa * kf:.4g} | {wa * kw:.4g} | {fa * kf + wa * kw:.4g} |")
    print(f"\ncalibration factors used: FETCH_SIZE x {kf:.1f} B, WRITE_SIZE x {kw:.1f} B (from the 4 B/lane copy kernel; 1024 = nominal KiB)")


if __name__ == "__main__":
    main()
