#!/usr/bin/env python3
"""One run -> the table of measured configurations quoted in DESIGN.md (every row is a `bench.py` invocation on THIS box,
same protocol, CPU-baseline leg off).  Output: markdown on stdout (commit it as profiles/rNN/design_table.md).

    python tools/design_table.py [--steps 100]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ROWS = [
    # label, bench.py arguments
    ("D3Q19 BGK fp32 256^3 periodic (configs[1])", "--workload periodic --size 256"),
    ("D3Q19 BGK fp32 256^3 periodic, single-step kernel", "--workload periodic --size 256 --opt fuse2=0"),
    ("D3Q19 BGK fp32 512^3 periodic", "--workload periodic --size 512"),
    ("D3Q19 BGK fp32 512^3 periodic, single-step kernel", "--workload periodic --size 512 --opt fuse2=0"),
    ("D3Q19 BGK fp32 512^3 cavity, halfway walls (configs[2])", "--workload cavity_halfway --size 512"),
    ("D3Q19 BGK fp32 512^3 cavity, halfway walls, single-step kernel", "--workload cavity_halfway --size 512 --opt fuse2=0"),
    ("D3Q19 BGK fp32 512^3 cavity, fullway walls (the reference harness's set-up)", "--workload cavity_fullway --size 512"),
    ("D3Q19 BGK fp32 384^3 cavity, halfway walls", "--workload cavity_halfway --size 384"),
    ("D3Q27 KBC FP64FP32 384^3 periodic, omega 1.9 (configs[4]), fast collision (default)", "--workload periodic --size 384 --lattice D3Q27 --collision KBC --policy FP64FP32 --omega 1.9"),
    ("D3Q27 KBC FP64FP32 384^3 periodic, omega 1.9, fast collision, single-step kernel", "--workload periodic --size 384 --lattice D3Q27 --collision KBC --policy FP64FP32 --omega 1.9 --opt fuse2=0"),
    ("D3Q27 KBC FP64FP32 384^3 periodic, omega 1.9, bit-exact build", "--workload periodic --size 384 --lattice D3Q27 --collision KBC --policy FP64FP32 --omega 1.9 --opt exact_math=1"),
    ("D3Q27 KBC FP32FP32 384^3 periodic, omega 1.9 (configs[4], fp32)", "--workload periodic --size 384 --lattice D3Q27 --collision KBC --policy FP32FP32 --omega 1.9"),
    ("D3Q27 BGK fp32 384^3 periodic (two-step kernel)", "--workload periodic --size 384 --lattice D3Q27"),
    ("D3Q27 BGK fp32 384^3 periodic, single-step kernel", "--workload periodic --size 384 --lattice D3Q27 --opt fuse2=0"),
    ("D3Q27 BGK fp32 384^3 cavity, halfway walls (single-step kernel: the automatic choice)", "--workload cavity_halfway --size 384 --lattice D3Q27"),
    ("D3Q27 BGK fp32 384^3 cavity, halfway walls, two-step kernel with BCs (fuse2 = 2)", "--workload cavity_halfway --size 384 --lattice D3Q27 --opt fuse2=2"),
    ("D3Q19 BGK FP32FP16 512^3 periodic", "--workload periodic --size 512 --policy FP32FP16"),
    ("D3Q19 BGK FP64FP64 384^3 periodic", "--workload periodic --size 384 --policy FP64FP64"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    print("| configuration | kernel | ms / step | MLUPS | GB/s (algorithmic) | of 8 TB/s | of the fused ideal | copy yardstick GB/s |")
    print("|---|---|---:|---:|---:|---:|---:|---:|")
    for label, extra in ROWS:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup", "10", "--cpu-baseline-seconds", "0"] + extra.split()
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            print(f"| {label} | FAILED: {out.stderr.strip().splitlines()[-1] if out.stderr.strip() else out.returncode} | | | | | | |", flush=True)
            continue
        r = json.loads(line[0])
        rf = r["roofline"]
        kern = rf["kernel"].split("<")[0] + (" (2 steps / launch)" if rf["steps_per_launch"] == 2 else "")
        print(f"| {label} | `{kern}` | {rf['kernel_ms']:.3f} | {r['value']:.0f} | {rf['achieved']:.0f} | {rf['frac']:.3f} | {rf['frac_of_fused_ideal']:.3f} | {rf['copy_yardstick_gbs']} |", flush=True)


if __name__ == "__main__":
    main()
