#!/usr/bin/env python3
"""MLUPS benchmark of the fused HIP LBM step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cavity_halfway|cavity_fullway|periodic]
                    [--size 512] [--global-shape 4096x512x512] [--lattice D3Q19] [--collision BGK] [--policy FP32FP32]

A "step" is one pull-stream + BC + collide pass over the whole lattice.  The default N=1
workload is BASELINE configs[2]: D3Q19 BGK 512^3 fp32 lid-driven cavity with halfway
bounce-back walls (the reference harness examples/performance/mlups_3d.py uses the same
cavity with fullway walls: --workload cavity_fullway).

N > 1: one process per GPU.  Either the caller starts the ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment), or — when WORLD_SIZE is not set — this script starts
them itself: the parent spawns N fresh children BEFORE anything touches HIP, waits, relays rank
0's JSON line and exits non-zero if any child did.
  * default: WEAK scaling — every rank owns a --size^3 slab of a (size N) x size x size cavity;
    N=8 at 512 is BASELINE configs[3] (the long axis is the slowest array axis);
  * --global-shape XxYxZ: STRONG scaling — the same global domain for every N (the reference
    harness's protocol, mlups_3d.py:546-556; north-star target: >= 6x at 8 GPUs on 4096x512x512).
The ring halo exchange over RCCL is overlapped with the interior update; with halfway walls on both
x faces the ring is a chain (no exchange between rank 0 and rank N-1).

Protocol (mirrors mlups_3d.py:225-242): W warm-up steps, device sync + barrier, K timed steps,
device sync + barrier; the time is the MAX over ranks; MLUPS = cells_total * K / t / 1e6.
Inputs are resident in HBM before the timed region.  One JSON line is printed by rank 0.
"""

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--workload", default="cavity_halfway", choices=["cavity_halfway", "cavity_fullway", "periodic"])
    p.add_argument("--size", type=int, default=512, help="cube edge per GPU (weak scaling)")
    p.add_argument("--global-shape", default="", help="XxYxZ: fixed global domain split over the GPUs along X (strong scaling)")
    p.add_argument("--lattice", default="D3Q19", choices=["D3Q19", "D3Q27"])
    p.add_argument("--collision", default="BGK", choices=["BGK", "KBC"])
    p.add_argument("--policy", default="FP32FP32")
    p.add_argument("--omega", type=float, default=1.0)  # mlups_3d.py:222
    p.add_argument("--cpu-baseline-seconds", type=float, default=14.0, help="budget of the CPU baseline legs; 0 disables them")
    p.add_argument("--cpu-baseline-size", type=int, default=256)
    p.add_argument("--opt", action="append", default=[], help="backend option key=value (e.g. vec=4, nt_store=0)")
    p.add_argument("--dry-run", action="store_true", help="launcher / rendezvous self-test: join the job, barrier, print a line marked dry_run; no GPU work")
    return p.parse_args(argv)


# ---- launcher: --gpus N from a plain shell ----------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    """GPUs this process would see, counted without touching HIP (the launcher must not initialise it): KFD topology
    nodes with compute units, capped by a HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES list."""
    n = 0
    top = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(top)
    except OSError:
        return 0
    for node in nodes:
        try:  # (a container that leases one GPU of a host sees every node but may read only its own: "Operation not permitted")
            for line in open(os.path.join(top, node, "properties")):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        except OSError:
            continue
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var, "").strip()
        if v:
            n = min(n, len([t for t in v.split(",") if t.strip()]))
    return n


def launch(args, argv):
    """Parent of a self-launched N-rank run.  Touches neither HIP nor xlb_amd: the children are fresh processes."""
    n = args.gpus
    port = free_port()
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), XLB_JOB_ID=f"bench-{os.getpid()}-{port}",
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    have = visible_gpus()
    if os.environ.get("XLB_BENCH_TRANSPORT", "auto") == "host" or 0 < have < n:
        # fewer GPUs than ranks (or the host-staged debugging transport): a REHEARSAL with every rank on one GPU — RCCL refuses
        # that ("Duplicate GPU"), so "auto" ends up on the ipc transport; the JSON line says so
        if "XLB_HIP_DEVICE" not in base:
            sys.stderr.write(f"bench.py launcher: {have} GPU(s) visible for {n} ranks: all ranks share device 0 (rehearsal, not a scaling measurement)\n")
        base.setdefault("XLB_HIP_DEVICE", "0")
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None))
    # rank 0's stdout is exactly the JSON line; read it to the end, then collect everyone
    out = procs[0].stdout.read()
    codes = []
    deadline = time.monotonic() + 120.0
    for p in procs:
        try:
            codes.append(p.wait(timeout=max(1.0, deadline - time.monotonic())))
        except subprocess.TimeoutExpired:  # a rank that outlives rank 0 by minutes is stuck: end exactly that process
            p.kill()
            codes.append(p.wait())
    if any(codes):
        sys.stderr.write(f"bench.py launcher: rank exit codes {codes}\n")
        return next(c for c in codes if c) or 1
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return 0


# ---- workload ---------------------------------------------------------------------------------------------------
def cavity_bcs(grid, walls_cls, EquilibriumBC):
    """The reference harness's cavity (mlups_3d.py:193-204) on the GLOBAL box."""
    import numpy as np

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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(args):
    """SURVEY 8(d) "CPU baseline beside it": CPU restatements of the same step timed on this host's cores on a bounded
    sample of the same workload —
      * the vectorised C++/OpenMP port (oracle/lbm_cpu_fast.cpp, g++ -O3 -march=native, compiled on this host; BGK) —
        or, for other collisions, oracle/lbm_ref.c built the same way — at nproc threads and at 1 thread,
      * the bit-exact checker build of oracle/lbm_ref.c (gcc -O2 -ffp-contract=off) at nproc threads,
      * the NumPy restatement (oracle/xlb_numpy.py) on a small box, for transparency.
    `value` is the fastest multi-thread leg.  Checker code timed as a baseline — never the product."""
    from oracle import lbm_ref, xlb_numpy as orc

    threads = host_cores()
    budget = args.cpu_baseline_seconds
    lattice = args.lattice
    coll = args.collision

    def case(n):
        lat = orc.Lattice(lattice)
        if args.workload == "periodic":
            return lat, (n, n, n), [], None, None, orc.perturbed_init((n, n, n), lat, seed=0), "periodic"
        kind = orc.KIND_HALFWAY_BB if args.workload == "cavity_halfway" else orc.KIND_FULLWAY_BB
        lat, shape, bcs = orc.cavity_3d(n, kind, lattice=lattice)
        bm, mm = orc.build_masks(shape, lat, bcs)
        return lat, shape, bcs, bm, mm, orc.initialize_eq(shape, lat), f"cavity ({kind} walls)"

    def time_loop(run2, n, seconds):
        run2(1)  # warm-up / page-in
        steps, t = 0, 0.0
        while t < seconds:
            t0 = time.perf_counter()
            run2(4)
            t += time.perf_counter() - t0
            steps += 4
        # every call copies its input once; that copy is inside the timed region (conservative)
        return n**3 * steps / t / 1e6, steps, t

    def time_checker(lib, n, nthreads, seconds, state):
        lat, shape, bcs, bm, mm, f, _ = state
        lbm_ref.set_threads(nthreads, lib=lib)
        box = [f]

        def run2(k):
            box[0] = lbm_ref.run(box[0], bm, mm, bcs, args.omega, lat, k, "FP32FP32", coll, lib=lib)

        return time_loop(run2, n, seconds)

    def time_port(lib, n, nthreads, seconds, state):
        lat, shape, bcs, bm, mm, f, _ = state
        box = [f]

        def run2(k):
            box[0] = lbm_ref.run_cpu_port(lib, box[0], bm, mm, bcs, args.omega, lat, k, threads=nthreads)

        return time_loop(run2, n, seconds)

    legs = []
    n = args.cpu_baseline_size
    big = case(n)
    label = big[-1]
    n1 = min(n, 128)
    small = big if n1 == n else case(n1)
    port = lbm_ref.build_cpu_port() if coll == "BGK" else None  # compiled on this host; None without g++
    if port is not None:
        name = "oracle/lbm_cpu_fast.cpp, g++ -O3 -march=native -fopenmp (vectorised CPU port, not bit-exact)"
        v, s, t = time_port(port, n, threads, 0.3 * budget, big)
        legs.append({"build": name, "threads": threads, "size": n, "steps": s, "seconds": round(t, 2), "mlups": round(v, 2)})
        v, s, t = time_port(port, n1, 1, 0.2 * budget, small)
        legs.append({"build": name, "threads": 1, "size": n1, "steps": s, "seconds": round(t, 2), "mlups": round(v, 2)})
    else:
        fast = lbm_ref.build_fast()  # the generic restatement, -O3 -march=native
        if fast is not None:
            name = "oracle/lbm_ref.c, gcc -O3 -march=native -fopenmp (not bit-exact)"
            v, s, t = time_checker(fast, n, threads, 0.3 * budget, big)
            legs.append({"build": name, "threads": threads, "size": n, "steps": s, "seconds": round(t, 2), "mlups": round(v, 2)})
            v, s, t = time_checker(fast, n1, 1, 0.2 * budget, small)
            legs.append({"build": name, "threads": 1, "size": n1, "steps": s, "seconds": round(t, 2), "mlups": round(v, 2)})
    v, s, t = time_checker(None, n, threads, 0.25 * budget, big)
    legs.append({"build": "oracle/lbm_ref.c, gcc -O2 -ffp-contract=off -fopenmp (the bit-exact checker)", "threads": threads, "size": n, "steps": s,
                 "seconds": round(t, 2), "mlups": round(v, 2)})
    # NumPy restatement, single process
    nn = 48
    lat, shape, bcs, bm, mm, f, _ = case(nn)
    if bm is None:
        bm, mm = orc.build_masks(shape, lat, [])
    steps, t = 0, 0.0
    while t < 0.15 * budget:
        t0 = time.perf_counter()
        f = orc.step(f, bm, mm, bcs, args.omega, lat, collision=coll)
        t += time.perf_counter() - t0
        steps += 1
    legs.append({"build": "NumPy restatement (oracle/xlb_numpy.py), one process", "threads": 1, "size": nn, "steps": steps, "seconds": round(t, 2),
                 "mlups": round(nn**3 * steps / t / 1e6, 3)})
    best = max((l for l in legs if l["threads"] == threads and "NumPy" not in l["build"]), key=lambda l: l["mlups"])
    return {
        "value": best["mlups"],
        "unit": "MLUPS",
        "cores": best["threads"],
        "kind": "port",
        "cpu_model": cpu_model(),
        "sample": f"{lattice} {coll} fp32 {label} {best['size']}^3, {best['steps']} steps in {best['seconds']} s, {best['build']}, "
                  f"{best['threads']} threads",
        "legs": legs,
    }


def kernel_source_hash():
    """Hash of the sources that decide what the step kernels do and how they are launched (the kernels, their launchers
    and the scheduling logic of api.hip): PMC traffic figures in profiles/traffic.json are only quoted for the build
    they were measured on.  (ops_kernels.hpp — whole-field operators, maskers —, comm.cpp and yardstick.hip — bench.py's measurement copy — are not part of it.)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "xlb_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".hip", ".cpp")) and name not in ("ops_kernels.hpp", "comm.cpp", "comm.hpp", "common.hpp", "yardstick.hip"):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def traffic_key(args, n):
    return f"{args.lattice}_{args.collision}_{args.policy}_{args.workload}_{n}"


def measured_traffic(args, world, n, kernel):
    """(HBM bytes per launch, provenance) from the committed PMC passes (profiles/traffic.json) when one matches this
    exact workload AND this build of the kernels; else (None, reason).  bench.py cannot collect counters on itself."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if world != 1 or not os.path.exists(path):
        return None, "no PMC record for multi-GPU runs"
    try:
        table = json.load(open(path))
    except Exception:
        return None, "profiles/traffic.json unreadable"
    entry = table.get(traffic_key(args, n))
    if not entry or kernel not in entry:
        return None, "no PMC record for this workload"
    if entry.get("source_hash") != kernel_source_hash():
        return None, f"stale: PMC record is for kernel sources {entry.get('source_hash')}, this build is {kernel_source_hash()}"
    return entry[kernel], f"profiles/traffic.json (offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, {entry.get('profile', 'profiles/')})"


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus > 1 and not os.environ.get("WORLD_SIZE", "").strip():
        sys.exit(launch(args, argv))
    # stdout carries exactly ONE line, the JSON of rank 0: everything else that writes to file descriptor 1 — the
    # reference-style no-slip warning of HalfwayBounceBackBC, RCCL's version banner (C code) — is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.dry_run:
        from xlb_amd import distribute as xd

        rank, world = xd.init_process_group(init_device_comm=False)
        ranks = xd.all_gather(rank)
        xd.barrier()
        if rank == 0:
            os.write(json_fd, (json.dumps({"dry_run": True, "n_gpus": world, "ranks": ranks, "value": None}) + "\n").encode())
        xd.shutdown()
        return
    import numpy as np

    import xlb_amd
    from xlb_amd import ComputeBackend, PrecisionPolicy
    from xlb_amd import distribute as xdist
    from xlb_amd.default_config import get_context
    from xlb_amd.grid import grid_factory
    from xlb_amd.grid.hip_grid import slab_bounds
    from xlb_amd.operator.boundary_condition import EquilibriumBC, FullwayBounceBackBC, HalfwayBounceBackBC
    from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

    # Halo transport (xlb_amd.distribute.init_process_group).  Default "auto": RCCL send / recv (the north-star's transport); if its
    # communicator cannot be built or fails the start-up self-check (an error, not a hang) every rank moves on to "ipc" — the
    # neighbours' fields mapped through HIP IPC and pulled by copy-engine copies on the communication stream — and only then to
    # the host-staged debugging transport; the JSON names what ran and why.  XLB_BENCH_TRANSPORT=ipc / host / rccl force one;
    # ipc and host also work with every rank on ONE GPU (rehearsals on a one-GPU box: not benchmark results).
    transport = os.environ.get("XLB_BENCH_TRANSPORT", "auto")
    # halfway walls on both x faces: no population is ever pulled across them (every such pull is a missing direction
    # that the wall redirects), so the ring is a chain.  Fullway wall cells DO exchange (inert) populations with their
    # periodic images in the reference (roll-based streaming), so that workload keeps the ring.
    periodic_x = args.workload != "cavity_halfway"
    rank, world = xdist.init_process_group(periodic_x=periodic_x, transport=transport)
    transport = xdist.transport() or transport  # what is actually in use ("host (fallback: ...)" after a failed RCCL set-up)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    pp = PrecisionPolicy[args.policy]
    vs = getattr(xlb_amd.velocity_set, args.lattice)(precision_policy=pp, compute_backend=ComputeBackend.HIP)
    xlb_amd.init(velocity_set=vs, default_backend=ComputeBackend.HIP, default_precision_policy=pp)
    ctx = get_context()
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))

    n = args.size
    if args.global_shape:
        shape = tuple(int(v) for v in args.global_shape.lower().split("x"))
        assert len(shape) == 3, "--global-shape XxYxZ"
        scaling = "strong"
    else:
        shape = (n * world, n, n)  # slabs along the slowest spatial axis ("axis naming")
        scaling = "weak"
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
        ls = grid.local_shape
        reps = tuple(-(-s // 32) for s in ls)
        T = pp.compute_precision.np_dtype
        rho_t = (1.0 + 0.01 * rng.uniform(-1, 1, (1, 32, 32, 32))).astype(T)
        u_t = (0.01 * rng.uniform(-1, 1, (3, 32, 32, 32))).astype(T)
        rho = grid.create_field(1, dtype=pp.compute_precision).assign(np.tile(rho_t, (1,) + reps)[:, : ls[0], : ls[1], : ls[2]])
        u = grid.create_field(3, dtype=pp.compute_precision).assign(np.tile(u_t, (1,) + reps)[:, : ls[0], : ls[1], : ls[2]])
        QuadraticEquilibrium()(rho, u, f_0)
        ctx.sync()
        rho.free()
        u.free()
    omega = args.omega

    def copy_yardstick():
        """streaming-copy rate on the same device and buffers (f_1 is scratch: the next step overwrites it entirely)"""
        f_1.copy_kernel_from(f_0, 16)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            f_1.copy_kernel_from(f_0, 16)
        ctx.sync()
        info = f_0.info()
        return 2 * info["plane_stride"] * vs.q * pp.store_precision.np_dtype(0).itemsize * 5 / (time.perf_counter() - t1) / 1e9

    def pattern_copy_ms():
        """ms for ONE copy of the field by a kernel with the two-step kernel's launch shape ((8 x 64) tiles, one block of 8 storing waves
        per CU, stores | barrier | pulls per plane; csrc/ops_kernels.hpp k_copy_tiles): what the memory system gives such a launch,
        without any arithmetic.  A fused pair moves the same algorithmic bytes (1 read + 1 write of the field)."""
        try:
            f_1.copy_tiles_from(f_0)
        except Exception:  # noqa: BLE001 (shapes the tile copy does not take: no yardstick)
            return None
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            f_1.copy_tiles_from(f_0)
        ctx.sync()
        return (time.perf_counter() - t1) / 5 * 1e3

    # (after the timed region by default; XLB_BENCH_YARDSTICK=before was an experiment — does a device that has just streamed 100 GB enter a
    # short timed region faster? no: the run-to-run spread, 2.35 / 2.50 ms per step on one box, is bimodal either way and moves with
    # time inside one process on unchanged allocations: tools/alloc_modes.py, profiles/r03/run_to_run_spread.md)
    yardstick_first = os.environ.get("XLB_BENCH_YARDSTICK", "after") == "before"
    copy_gbs = copy_yardstick() if (world == 1 and yardstick_first) else None
    xdist.barrier()  # (set-up takes the ranks unevenly long; the first exchange's bounded waits should not have to cover that)
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, args.warmup)
    ctx.sync()
    xdist.barrier()
    t0 = time.perf_counter()
    (f_0, f_1), dev_ms = stepper.run_timed(f_0, f_1, bc_mask, missing_mask, omega, args.steps, first_timestep=args.warmup)
    ctx.sync()
    xdist.barrier()
    elapsed = xdist.all_reduce_max(time.perf_counter() - t0)
    per_rank_ms = [round(v / args.steps, 4) for v in xdist.all_gather(float(dev_ms))]
    dev_ms = max(xdist.all_gather(float(dev_ms)))
    # slab runs: how long each rank's compute stream sat waiting for its halo exchange after the interior launch (a lost
    # overlap shows here; with a working one it is the cost of the two timing events).  The warm-up's share is small: W of W + K steps.
    stats = ctx.comm_stats() if world > 1 else {"halo_wait_ms": 0.0, "halo_waits": 0}
    halo_wait = xdist.all_gather([round(stats["halo_wait_ms"], 3), int(stats["halo_waits"])])

    if world == 1 and not yardstick_first:
        copy_gbs = copy_yardstick()
    tiles_ms = pattern_copy_ms() if world == 1 and pp.store_precision.np_dtype(0).itemsize == 4 else None

    cells_total = float(np.prod(shape))
    local_cells = float(max(slab_bounds(shape[0], r, world)[1] for r in range(world))) * shape[1] * shape[2]
    mlups = cells_total * args.steps / elapsed / 1e6
    s_bytes = pp.store_precision.np_dtype(0).itemsize
    b_alg = 2 * vs.q * s_bytes  # one read + one write of every population
    step_ms = dev_ms / args.steps  # HIP events on the compute stream around the K launches
    achieved = b_alg * local_cells / (step_ms * 1e-3) / 1e9  # per GPU (the largest slab), GB/s
    # xlbhip_run fuses two steps per launch where the library's rule says so: ask it, to name the kernel that actually ran
    fused2 = args.steps >= 2 and stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
    fused2 = bool(xdist.all_reduce_min(1.0 if fused2 else 0.0))  # (the library takes the same MIN over the ranks)
    if rank != 0:
        xdist.shutdown()
        return
    kernel = (f"k_step2<{args.lattice}, {args.collision}, {args.policy}> (two steps per launch through LDS)" if fused2 else
              f"k_step<{args.lattice}, {args.collision}, {args.policy}, vec{ctx.get_option('vec') or 1}>")
    spl = 2 if fused2 else 1
    launch_ms = step_ms * spl
    traffic, traffic_src = measured_traffic(args, world, n if not args.global_shape else 0, "k_step2" if fused2 else "k_step")
    is_c2 = args.workload == "cavity_halfway" and shape == (512, 512, 512) and world == 1
    is_c3 = args.workload == "cavity_halfway" and shape == (4096, 512, 512)
    out = {
        "metric": f"MLUPS (million lattice updates/s) {args.lattice} {args.collision}, " + (f"{n}^3 per GPU" if scaling == "weak" else "x".join(map(str, shape)) + " global"),
        "value": round(mlups, 1),
        "unit": "MLUPS",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32" if args.policy == "FP32FP32" else args.policy.lower(),
        "data": "synthetic",
        "config": {
            "workload": f"{args.lattice} {args.collision} {args.policy} {args.workload} {'x'.join(map(str, shape))}"
                        + (f" ({n}^3 per GPU)" if scaling == "weak" else f" (fixed global domain over {world} GPU(s))") + f", omega={omega}",
            "baseline_config": "configs[2]" if is_c2 else ("configs[3] (long axis = slowest array axis)" if (is_c3 and world == 8) else
                                                           ("configs[3]'s domain on fewer GPUs" if is_c3 else "other")),
            "decomposition": (f"{world} x-slab(s) of {'/'.join(str(slab_bounds(shape[0], r, world)[1]) for r in range(world))} planes, "
                              + ("chain" if not periodic_x else "ring") + (" halo over RCCL" if transport == "rccl" else f" halo, NOT over RCCL — transport {transport}")
                              + (", all ranks on ONE device (rehearsal)" if os.environ.get("XLB_HIP_DEVICE", "").strip() else ""))
                             if world > 1 else "single GPU",
            "per_rank_ms_per_step": per_rank_ms,
            # per rank: total ms the compute stream waited for halo exchanges (warm-up + timed steps), and how many exchanges
            "halo_wait_ms": [h[0] for h in halo_wait] if world > 1 else None,
            "halo_exchanges": [h[1] for h in halo_wait] if world > 1 else None,
        },
        "roofline": {
            "bound": "hbm",
            # `achieved` / `frac`: ALGORITHMIC bytes (2 q s per update) per launch / launch duration — the contract's
            # definition.  With two steps per launch the intermediate f(t+1) never reaches HBM, so this "effective"
            # rate may exceed what the memory system really moves: that is `hbm_gbs` = measured traffic / duration.
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,  # HBM bytes per launch (PMC), null unless measured on this build of the kernels
            "traffic_source": traffic_src,
            "hbm_gbs": None if traffic is None else round(traffic / (launch_ms * 1e-3) / 1e9, 1),
            "hbm_frac": None if traffic is None else round(traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_launch": int(b_alg * local_cells * spl),
            # what a launch must move at the least: one read + one write of every population per LAUNCH (f(t+1) of a fused
            # pair never reaches HBM), and the physical rate against that bound — `frac` above is the contract's effective figure
            "fused_ideal_bytes_per_launch": int(b_alg * local_cells),
            "frac_of_fused_ideal": round(b_alg * local_cells / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "kernel": kernel,
            "kernel_ms": round(step_ms, 4),  # device time per STEP (HIP events / K)
            "steps_per_launch": spl,
            "launch_ms": round(launch_ms, 4),  # what rocprofv3 reports per kernel launch
            "algorithmic_bytes_per_update": b_alg,
            "copy_yardstick_gbs": None if copy_gbs is None else round(copy_gbs, 1),
            # the second yardstick (two-step launches only): a bare copy of the field with k_step2's launch shape — tiles, occupancy,
            # phase order — and how much of its rate the launch reaches at equal algorithmic bytes (1 read + 1 write of the field)
            "pattern_copy_ms": None if (tiles_ms is None or not fused2) else round(tiles_ms, 4),
            "frac_of_pattern_copy": None if (tiles_ms is None or not fused2) else round(tiles_ms / launch_ms, 4),
            "kernel_source_hash": kernel_source_hash(),
        },
    }
    if world == 1 and args.cpu_baseline_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(args)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    xdist.shutdown()


if __name__ == "__main__":
    main()
