#!/usr/bin/env python3
"""Interleaved A/B sweep of backend options on one workload, in ONE process (methodology rule:
perf deltas come from interleaved rounds in one process).  Prints median / min ms per step.

    python tools/sweep.py --workload periodic --size 512 --rounds 3 --steps 20 \
        --variant vec=4 --variant vec=1 --variant "vec=4,nt_store=0"
"""

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="periodic")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--lattice", default="D3Q19")
    ap.add_argument("--collision", default="BGK")
    ap.add_argument("--policy", default="FP32FP32")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--omega", type=float, default=1.0)
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--halo", type=int, default=0, help="ghost planes per side (the multi-rank field layout on one rank: ring exchange onto itself)")
    ap.add_argument("--self-comm", action="store_true", help="with --halo: a real one-rank RCCL communicator instead of device copies")
    ap.add_argument("--realloc", action="store_true", help="variants change the field layout (plane_pad_bytes): rebuild fields per variant")
    args = ap.parse_args()

    import xlb_amd
    from bench import cavity_bcs
    from xlb_amd import ComputeBackend, PrecisionPolicy
    from xlb_amd.default_config import get_context
    from xlb_amd.grid import grid_factory
    from xlb_amd.operator.boundary_condition import EquilibriumBC, FullwayBounceBackBC, HalfwayBounceBackBC
    from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

    pp = PrecisionPolicy[args.policy]
    vs = getattr(xlb_amd.velocity_set, args.lattice)(precision_policy=pp, compute_backend=ComputeBackend.HIP)
    xlb_amd.init(velocity_set=vs, default_backend=ComputeBackend.HIP, default_precision_policy=pp)
    ctx = get_context()
    defaults = {k: ctx.get_option(k) for k in ("vec", "nt_store", "plane_pad_bytes", "block_threads", "block_tz", "overlap", "xcd_swizzle", "nt_load", "fuse2", "fuse2_xcd", "fuse2_lpt", "fuse2_xseg", "exact_math", "fuse2_clean", "fuse2_xcap", "fuse2_shift", "fast_bgk", "fuse2_tile", "fuse2_strips", "halo_skip", "ipc_copy", "fuse2_rowmap")}
    n = args.size
    if args.self_comm:
        from xlb_amd import _lib

        ctx.comm_init(0, 1, _lib.comm_unique_id())

    def setup():
        grid = grid_factory((n, n, n), backend_config={"halo": args.halo} if args.halo else None)
        if args.workload == "periodic":
            bcs = []
        elif args.workload == "one_cell":
            # a single fullway cell: isolates the cost of the HASBC kernel variant itself
            bcs = [FullwayBounceBackBC(indices=[[n // 2], [n // 2], [n // 2]])]
        elif args.workload == "xwalls":
            # walls on the two x faces only: whole planes of boundary cells, no boundary lanes elsewhere
            b = grid.bounding_box_indices(as_numpy=True)
            bcs = [HalfwayBounceBackBC(indices=np.concatenate([b["left"], b["right"]], axis=1))]
        elif args.workload == "ywalls":
            b = grid.bounding_box_indices(as_numpy=True)
            bcs = [HalfwayBounceBackBC(indices=np.concatenate([b["front"], b["back"]], axis=1))]
        elif args.workload == "zwalls_fw":
            b = grid.bounding_box_indices(as_numpy=True)
            bcs = [FullwayBounceBackBC(indices=np.concatenate([b["bottom"], b["top"]], axis=1))]
        elif args.workload == "zwalls":
            b = grid.bounding_box_indices(as_numpy=True)
            bcs = [HalfwayBounceBackBC(indices=np.concatenate([b["bottom"], b["top"]], axis=1))]
        else:
            bcs = cavity_bcs(grid, HalfwayBounceBackBC if args.workload == "cavity_halfway" else FullwayBounceBackBC, EquilibriumBC)
        st = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs, collision_type=args.collision)
        fields = st.prepare_fields()
        if args.workload == "periodic":
            # bench.py's synthetic state: f = feq(1 + 0.01 xi, 0.01 eta) on a tiled 32^3 pattern.  A uniform
            # f = w makes compute-bound kernels look faster than they are (fp64 KBC pairs: 2.18 ms/step on f = w, 2.31 on this
            # state — identical operands in every lane, lower power, higher clocks): timings must come from a non-trivial state
            from xlb_amd.operator.equilibrium import QuadraticEquilibrium

            rng = np.random.default_rng(0)
            ls = grid.local_shape
            reps = tuple(-(-v // 32) for v in ls)
            T = pp.compute_precision.np_dtype
            rho_t = (1.0 + 0.01 * rng.uniform(-1, 1, (1, 32, 32, 32))).astype(T)
            u_t = (0.01 * rng.uniform(-1, 1, (3, 32, 32, 32))).astype(T)
            rho = grid.create_field(1, dtype=pp.compute_precision).assign(np.tile(rho_t, (1,) + reps)[:, : ls[0], : ls[1], : ls[2]])
            u = grid.create_field(3, dtype=pp.compute_precision).assign(np.tile(u_t, (1,) + reps)[:, : ls[0], : ls[1], : ls[2]])
            QuadraticEquilibrium()(rho, u, fields[0])
            ctx.sync()
            rho.free()
            u.free()
        return st, fields

    def apply(variant):
        for k, v in defaults.items():
            ctx.set_option(k, v)
        for kv in filter(None, variant.split(",")):
            k, v = kv.split("=")
            ctx.set_option(k, int(v))

    variants = args.variant or [""]
    times = {v: [] for v in variants}
    shared = None if args.realloc else setup()
    for r in range(args.rounds):
        for v in variants:
            apply(v)
            st, (f0, f1, bm, mm) = shared if shared else setup()
            st.run(f0, f1, bm, mm, args.omega, 4)
            ctx.sync()
            _, ms = st.run_timed(f0, f1, bm, mm, args.omega, args.steps)
            times[v].append(ms / args.steps)
            if not shared:
                for fld in (f0, f1, bm, mm):
                    fld.free()
    cells = float(n) ** 3
    b_alg = 2 * vs.q * pp.store_precision.np_dtype(0).itemsize
    print(f"# {args.lattice} {args.collision} {args.policy} {args.workload} {n}^3, {args.steps} steps x {args.rounds} rounds, halo {args.halo}"
          f"{' (RCCL self communicator)' if args.self_comm else ''}")
    print(f"{'variant':44s} {'med ms':>8s} {'min ms':>8s} {'MLUPS(med)':>11s} {'GB/s':>8s} {'frac':>6s}")
    for v in variants:
        t = np.array(times[v])
        med, mn = float(np.median(t)), float(t.min())
        gbs = b_alg * cells / (med * 1e-3) / 1e9
        print(f"{(v or 'default'):44s} {med:8.4f} {mn:8.4f} {cells / med / 1e3:11.1f} {gbs:8.1f} {gbs / 8000:6.3f}")


if __name__ == "__main__":
    main()
