#!/usr/bin/env python3
"""MLUPS benchmark of the fused HIP LBM step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cavity_halfway|cavity_fullway|periodic]
                    [--size 512] [--lattice D3Q19] [--collision BGK] [--policy FP32FP32]

A "step" is one pull-stream + BC + collide pass over the whole lattice.  The default N=1
workload is BASELINE configs[2]: D3Q19 BGK 512^3 fp32 lid-driven cavity with halfway
bounce-back walls (the reference harness examples/performance/mlups_3d.py uses the same
cavity with fullway walls: --workload cavity_fullway).  For N>1 (launched with
`python -m torch.distributed.run --nproc-per-node N`) every rank owns a 512^3 slab of a
(512 N) x 512 x 512 cavity — N=8 is BASELINE configs[3] — with the ring halo exchange over
RCCL overlapped with the interior update: weak scaling.

Protocol (mirrors mlups_3d.py:225-242): W warm-up steps, device sync + barrier, K timed steps,
device sync + barrier; the time is the MAX over ranks; MLUPS = cells_total * K / t / 1e6.
Inputs are resident in HBM before the timed region.  One JSON line is printed by rank 0.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--workload", default="cavity_halfway", choices=["cavity_halfway", "cavity_fullway", "periodic"])
    p.add_argument("--size", type=int, default=512, help="cube edge per GPU")
    p.add_argument("--lattice", default="D3Q19", choices=["D3Q19", "D3Q27"])
    p.add_argument("--collision", default="BGK", choices=["BGK", "KBC"])
    p.add_argument("--policy", default="FP32FP32")
    p.add_argument("--cpu-baseline-seconds", type=float, default=12.0, help="0 disables the CPU baseline leg")
    p.add_argument("--cpu-baseline-size", type=int, default=256)
    p.add_argument("--opt", action="append", default=[], help="backend option key=value (e.g. vec=4, nt_store=0)")
    return p.parse_args()


def cavity_bcs(grid, walls_cls, EquilibriumBC):
    """The reference harness's cavity (mlups_3d.py:193-204) on the GLOBAL box."""
    box = grid.bounding_box_indices(as_numpy=True)
    box_ne = grid.bounding_box_indices(remove_edges=True, as_numpy=True)
    lid = box_ne["top"]
    walls = np.concatenate([box[f] for f in ("bottom", "left", "right", "front", "back")], axis=1).astype(np.int64)
    # the reference driver's np.unique(walls, axis=-1) (mlups_3d.py:198), done on linear keys (same set, same order)
    nx, ny, nz = grid.shape
    keys = np.unique((walls[0] * ny + walls[1]) * nz + walls[2])
    walls = np.stack([keys // (ny * nz), (keys // nz) % ny, keys % nz]).astype(np.int32)
    return [EquilibriumBC(rho=1.0, u=(0.02, 0.0, 0.0), indices=lid), walls_cls(indices=walls)]


def host_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get("OMP_NUM_THREADS", "").strip()
    if env.isdigit():
        n = min(n, int(env))
    return n


def cpu_baseline(args):
    """The C restatement of the same step (oracle/lbm_ref.c, OpenMP) timed on this host's cores on a
    bounded sample of the same workload.  Checker code timed as a baseline — never the product."""
    from oracle import lbm_ref, xlb_numpy as orc

    n = args.cpu_baseline_size
    threads = host_cores()
    lbm_ref.set_threads(threads)
    lat = orc.Lattice(args.lattice)
    if args.workload == "periodic":
        bcs, bm, mm = [], None, None
        f = orc.perturbed_init((n, n, n), lat, seed=0)
        label = "periodic"
    else:
        kind = orc.KIND_HALFWAY_BB if args.workload == "cavity_halfway" else orc.KIND_FULLWAY_BB
        lat, shape, bcs = orc.cavity_3d(n, kind, lattice=args.lattice)
        bm, mm = orc.build_masks(shape, lat, bcs)
        f = orc.initialize_eq(shape, lat)
        label = f"cavity ({kind} walls)"
    coll = args.collision
    f = lbm_ref.run(f, bm, mm, bcs, 1.0, lat, 1, "FP32FP32", coll)  # warm-up / page-in
    steps, t = 0, 0.0
    per_call = 2
    while t < args.cpu_baseline_seconds:
        t0 = time.perf_counter()
        f = lbm_ref.run(f, bm, mm, bcs, 1.0, lat, per_call, "FP32FP32", coll)
        t += time.perf_counter() - t0
        steps += per_call
    # lbm_ref.run copies its input once per call; that copy is inside the timed region (conservative)
    mlups = n**3 * steps / t / 1e6
    return {
        "value": round(mlups, 2),
        "unit": "MLUPS",
        "cores": threads,
        "kind": "port",
        "sample": f"{args.lattice} {coll} fp32 {label} {n}^3, {steps} steps in {t:.1f} s, oracle/lbm_ref.c (gcc -O2 -fopenmp), {threads} threads",
    }


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON of rank 0: everything else that writes to file descriptor 1 — the
    # reference-style no-slip warning of HalfwayBounceBackBC, RCCL's version banner (C code) — is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import xlb_amd
    from xlb_amd import ComputeBackend, PrecisionPolicy
    from xlb_amd import distribute as xdist
    from xlb_amd.default_config import get_context
    from xlb_amd.grid import grid_factory
    from xlb_amd.operator.boundary_condition import EquilibriumBC, FullwayBounceBackBC, HalfwayBounceBackBC
    from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

    # XLB_BENCH_TRANSPORT=host: rehearsal of the N > 1 code path with all ranks on ONE GPU (ghost planes through gloo;
    # RCCL refuses two ranks on a device).  Numbers from it are not benchmark results.
    transport = os.environ.get("XLB_BENCH_TRANSPORT", "rccl")
    rank, world = xdist.init_process_group(transport=transport)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run --nproc-per-node {args.gpus}")

    pp = PrecisionPolicy[args.policy]
    vs = getattr(xlb_amd.velocity_set, args.lattice)(precision_policy=pp, compute_backend=ComputeBackend.HIP)
    xlb_amd.init(velocity_set=vs, default_backend=ComputeBackend.HIP, default_precision_policy=pp)
    ctx = get_context()
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))

    n = args.size
    shape = (n * world, n, n)  # slabs along the slowest spatial axis ("axis naming")
    grid = grid_factory(shape)
    if args.workload == "periodic":
        bcs = []
    else:
        bcs = cavity_bcs(grid, HalfwayBounceBackBC if args.workload == "cavity_halfway" else FullwayBounceBackBC, EquilibriumBC)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs, collision_type=args.collision)
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    if args.workload == "periodic":
        # non-trivial synthetic state: f = feq(1 + 0.01 xi, 0.01 eta), xi, eta ~ U(-1, 1),
        # default_rng(0), drawn on a 32^3 tile and evaluated by the backend's own equilibrium operator
        from xlb_amd.operator.equilibrium import QuadraticEquilibrium

        rng = np.random.default_rng(0)
        reps = (grid.local_shape[0] // 32, n // 32, n // 32)
        T = pp.compute_precision.np_dtype
        rho_t = (1.0 + 0.01 * rng.uniform(-1, 1, (1, 32, 32, 32))).astype(T)
        u_t = (0.01 * rng.uniform(-1, 1, (3, 32, 32, 32))).astype(T)
        rho = grid.create_field(1, dtype=pp.compute_precision).assign(np.tile(rho_t, (1,) + reps))
        u = grid.create_field(3, dtype=pp.compute_precision).assign(np.tile(u_t, (1,) + reps))
        QuadraticEquilibrium()(rho, u, f_0)
        ctx.sync()
        rho.free()
        u.free()
    omega = 1.0  # mlups_3d.py:222

    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, args.warmup)
    ctx.sync()
    xdist.barrier()
    t0 = time.perf_counter()
    (f_0, f_1), dev_ms = stepper.run_timed(f_0, f_1, bc_mask, missing_mask, omega, args.steps, first_timestep=args.warmup)
    ctx.sync()
    xdist.barrier()
    elapsed = xdist.all_reduce_max(time.perf_counter() - t0)
    dev_ms = xdist.all_reduce_max(dev_ms)

    # streaming-copy yardstick on the same device and buffers (after the timed region; f_1 is scratch now)
    copy_gbs = None
    if world == 1:
        f_1.copy_kernel_from(f_0, 16)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            f_1.copy_kernel_from(f_0, 16)
        ctx.sync()
        info = f_0.info()
        copy_gbs = 2 * info["plane_stride"] * vs.q * pp.store_precision.np_dtype(0).itemsize * 5 / (time.perf_counter() - t1) / 1e9

    cells_total = float(n) ** 3 * world
    mlups = cells_total * args.steps / elapsed / 1e6
    s_bytes = pp.store_precision.np_dtype(0).itemsize
    b_alg = 2 * vs.q * s_bytes  # one read + one write of every population
    step_ms = dev_ms / args.steps  # HIP events on the compute stream around the K launches
    achieved = b_alg * float(n) ** 3 / (step_ms * 1e-3) / 1e9  # per GPU, GB/s
    if rank != 0:
        return
    # xlbhip_run fuses two steps per launch where the library's rule says so (D3Q19 BGK fp32, basic BCs,
    # ny % 8 == nz % 64 == 0, enough tile segments to fill the chip): ask it, to name the kernel that actually ran
    fused2 = args.steps >= 2 and stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
    kernel = (f"k_step2<{args.lattice}, {args.collision}, {args.policy}> (two steps per launch through LDS)" if fused2 else
              f"k_step<{args.lattice}, {args.collision}, {args.policy}, vec{ctx.get_option('vec') or 1}>")
    out = {
        "metric": f"MLUPS (million lattice updates/s) {args.lattice} {args.collision}, {n}^3 per GPU",
        "value": round(mlups, 1),
        "unit": "MLUPS",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.policy == "FP32FP32" else args.policy.lower(),
        "data": "synthetic",
        "config": {
            "workload": f"{args.lattice} {args.collision} {args.policy} {args.workload} {shape[0]}x{n}x{n} ({n}^3 per GPU), omega=1.0",
            "baseline_config": "configs[2]" if (args.workload == "cavity_halfway" and n == 512 and world == 1) else
                               ("configs[3] (long axis = slowest array axis)" if (n == 512 and world == 8) else "other"),
            "decomposition": (f"{world} x-slab(s), ring halo over RCCL" if transport == "rccl" else
                              f"{world} x-slab(s), REHEARSAL transport {transport}") if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),  # algorithmic bytes per launch / launch duration
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": measured_traffic(args, world, "k_step2" if fused2 else "k_step"),  # HBM bytes per launch (PMC)
            "algorithmic_bytes_per_launch": b_alg * n**3 * (2 if fused2 else 1),
            "kernel": kernel,
            "kernel_ms": round(step_ms, 4),  # device time per STEP (HIP events / K)
            "steps_per_launch": 2 if fused2 else 1,
            "launch_ms": round(step_ms * (2 if fused2 else 1), 4),  # what rocprofv3 reports per kernel launch
            "algorithmic_bytes_per_update": b_alg,
            "copy_yardstick_gbs": None if copy_gbs is None else round(copy_gbs, 1),
        },
    }
    if world == 1 and args.cpu_baseline_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(args)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def measured_traffic(args, world, kernel):
    """HBM bytes per launch from the committed PMC passes (profiles/traffic.json), if one matches
    this exact workload; else null.  bench.py cannot collect counters on itself."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    try:
        table = json.load(open(path))
    except Exception:
        return None
    key = f"{args.lattice}_{args.collision}_{args.policy}_{args.workload}_{args.size}"
    return (table.get(key) or {}).get(kernel)


if __name__ == "__main__":
    main()
