"""Incompressible Navier-Stokes stepper: one fused HIP kernel per time step.

Reference: xlb/operator/stepper/nse_stepper.py — constructor :65-97, ``prepare_fields``
:99-148, boundary processing :150-205, the step itself :237-282 (JAX order, the parity
target) / :427-476 (fused Warp kernel, the structural model).

    stepper = IncompressibleNavierStokesStepper(grid, boundary_conditions, collision_type="BGK")
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    for i in range(n):
        f_0, f_1 = stepper(f_0, f_1, bc_mask, missing_mask, omega, i)
        f_0, f_1 = f_1, f_0

``stepper.run(f_0, f_1, bc_mask, missing_mask, omega, n_steps)`` runs the same loop natively
(no per-step Python dispatch, which costs tens of microseconds per call in the reference's
``Operator.__call__``).

**Pairing of reference-style calls.**  The fast kernel advances TWO steps per pass (csrc/step2_kernel.hpp).  So that an
unmodified XLB driver — the loop above, every example of the reference (mlups_3d.py:237-238) — gets it too, a call on
fields the two-step kernel supports is DEFERRED: nothing is enqueued; when the next call arrives with the two
population fields swapped (same masks, same omega) both steps run as one fused pass, otherwise — any other use of either
field, ``ctx.sync()``, a different call — the deferred step runs first as a single step.  The results are bit-identical
to single steps.  After a fused pair the field that received step t+1 in the reference's protocol holds it only virtually
(the fused pass writes f(t+2) next to f(t), and the two field objects exchange their device buffers): if it is read
before the next call overwrites it, it is materialised through a temporary third field.  Fields exported through DLPack /
``__cuda_array_interface__`` are never deferred (their memory must stay put); ``backend_config={"lazy_pairs": False}``
switches the mechanism off."""

import numpy as np

from ... import _lib
from ...compute_backend import ComputeBackend
from ...helper.check_boundary_overlaps import check_bc_overlaps
from ...helper.initializers import initialize_eq
from ...helper.nse_fields import create_nse_fields
from ..boundary_condition import ImplementationStep
from ..boundary_masker import IndicesBoundaryMasker
from ..collision import BGK, KBC, SmagorinskyLESBGK
from ..equilibrium import QuadraticEquilibrium
from ..macroscopic import Macroscopic
from ..operator import Operator
from ..stream import Stream
from .stepper import Stepper


class _Materialise:
    """Hook of a field that holds f(t+1) only virtually after a fused pair (its buffer holds f(t)): see _lazy_pair."""

    def __init__(self, stepper, bcm, miss, omega, t):
        self.args = (stepper, bcm, miss, omega, t)

    def __call__(self, field):
        stepper, bcm, miss, omega, t = self.args
        stepper._materialise(field, bcm, miss, omega, t)


class IncompressibleNavierStokesStepper(Stepper):
    def __init__(self, grid, boundary_conditions=[], collision_type="BGK", streaming_scheme="pull",
                 forcing_scheme="exact_difference", force_vector=None, backend_config={}):
        self.backend_config = dict(backend_config or {})
        self.collision_type = collision_type
        self.streaming_scheme = streaming_scheme
        if streaming_scheme != "pull":
            raise AssertionError(f"Unknown or unimplemented streaming scheme for backend: {ComputeBackend.HIP}")
        if force_vector is not None:
            if forcing_scheme != "exact_difference":
                raise NotImplementedError(f"Force model {forcing_scheme} not implemented!")  # forced_collision.py:40
            force_vector = np.asarray(force_vector, dtype=np.float64)
        self.forcing_scheme = forcing_scheme
        self.force_vector = force_vector
        self._native = None
        self._deferred = None   # (f_src, f_dst, bc_mask, missing_mask, omega, timestep): a step not enqueued yet
        self._n_fused_pairs = self._n_materialised = 0  # statistics of the pairing (tests, diagnostics)
        super().__init__(grid, boundary_conditions)
        vs, pp, be = self.velocity_set, self.precision_policy, self.compute_backend
        if collision_type == "BGK":
            self.collision = BGK(vs, pp, be)
        elif collision_type == "KBC":
            self.collision = KBC(vs, pp, be)
        elif collision_type == "SmagorinskyLESBGK":
            self.collision = SmagorinskyLESBGK(vs, pp, be)
        else:
            raise NotImplementedError(f"collision_type {collision_type!r} is not available on the HIP backend (BGK, KBC, SmagorinskyLESBGK)")
        if force_vector is not None and force_vector.shape != (vs.d,):
            raise AssertionError("Check the dimensions of the input force!")  # forced_collision.py:41
        self.stream = Stream(vs, pp, be)
        self.equilibrium = QuadraticEquilibrium(vs, pp, be)
        self.macroscopic = Macroscopic(vs, pp, be)

    # -- native stepper object, created lazily (BC ids and constants are final by then)
    def _native_stepper(self):
        if self._native is None:
            descs = [bc._hip_descriptor() for bc in self.boundary_conditions]
            self._native = _lib.Stepper(self._ctx, self.velocity_set.hip_id, self.collision.hip_collision_id, self._compute_code,
                                        self._store_code, descs)
            if isinstance(self.collision, SmagorinskyLESBGK):
                self._native.set_smagorinsky(self.collision.smagorinsky_coef)
            for bc in self.boundary_conditions:
                table = bc._profile_table(self.grid) if hasattr(bc, "_profile_table") else None
                if table is not None:
                    self._native.set_bc_profile(bc.id, *table)
            for bc in self.boundary_conditions:
                table = getattr(bc, "_distance_table", None)
                if table is not None:
                    # wall-distance weights the mesh masker gathered (HybridBC with use_mesh_distance); interior linear
                    # cell index == storage cell index on the single-rank fields mesh BCs live on
                    self._native.set_bc_distances(*table)
            if self.force_vector is not None:
                f3 = np.zeros(3)
                f3[3 - self.velocity_set.d :] = self.force_vector  # internal 3-component form
                self._native.set_force(f3)
        return self._native

    def prepare_fields(self, initializer=None):
        _, f_0, f_1, missing_mask, bc_mask = create_nse_fields(
            grid=self.grid, velocity_set=self.velocity_set, compute_backend=self.compute_backend, precision_policy=self.precision_policy
        )
        f_1, bc_mask, missing_mask = self._process_boundary_conditions(self.boundary_conditions, f_1, bc_mask, missing_mask)
        if initializer is not None:
            f_0 = initializer(bc_mask, f_0)
        else:
            f_0 = initialize_eq(f_0, self.grid, self.velocity_set, self.precision_policy, self.compute_backend)
        return f_0, f_1, bc_mask, missing_mask

    def _process_boundary_conditions(self, boundary_conditions, f_1, bc_mask, missing_mask):
        import weakref

        for bc in boundary_conditions:
            bc._stepper_ref = weakref.ref(self)  # (MomentumTransfer of a HybridBC / profile wall reads this stepper's tables)
        check_bc_overlaps(boundary_conditions, self.velocity_set.d, self.compute_backend)
        masker = IndicesBoundaryMasker(self.velocity_set, self.precision_policy, self.compute_backend, grid=self.grid)
        with_indices = [bc for bc in boundary_conditions if getattr(bc, "indices", None) is not None]
        if with_indices:
            bc_mask, missing_mask = masker(with_indices, bc_mask, missing_mask)
        # mesh-based BCs (nse_stepper.py:165-203 in the reference): one masker per voxelisation method
        for bc in boundary_conditions:
            if getattr(bc, "mesh_vertices", None) is None:
                continue
            from ..boundary_masker import mesh_masker_for

            mesh_masker = mesh_masker_for(getattr(bc, "voxelization_method", None), self.velocity_set, self.precision_policy, self.compute_backend)
            f_1, bc_mask, missing_mask = mesh_masker(bc, f_1, bc_mask, missing_mask)
        # wall-velocity profiles (HybridBC(profile=...), bc_hybrid.py:163-172): evaluated at the BC's cells, known only now
        with_profile = [bc for bc in boundary_conditions if callable(getattr(bc, "_evaluate_profile", None)) and getattr(bc, "profile", None)]
        if with_profile:
            if bc_mask.halo != 0:
                raise NotImplementedError("wall-velocity profiles need fields without ghost planes (single rank)")
            ids = bc_mask.numpy()[0]
            if ids.ndim == 2:
                ids = ids[None]
            for bc in with_profile:
                cells = np.argwhere(ids == bc.id).T  # (3, n), ascending storage order
                keys = (cells[0] * ids.shape[1] + cells[1]) * ids.shape[2] + cells[2]
                bc._evaluate_profile(cells, keys)
        return f_1, bc_mask, missing_mask

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f_0, f_1, bc_mask, missing_mask, omega, timestep):
        if not self._lazy_pair(f_0, f_1, bc_mask, missing_mask, omega, timestep):
            self._native_stepper().step(f_0, f_1, bc_mask, missing_mask, omega, timestep)
        return f_0, f_1

    # -- pairing of reference-style calls (module docstring) ------------------------------------------------------
    def _lazy_eligible(self, f_0, f_1, bc_mask, missing_mask):
        if not self.backend_config.get("lazy_pairs", True) or f_0.halo != 0 or f_0._pinned or f_1._pinned:
            return False
        # (asked every time: the answer follows the backend options — fuse2, ... — and the masks' contents; host logic only)
        if not self._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask):
            return False
        return self._room_for_a_third_field(f_0)

    def _room_for_a_third_field(self, f):
        """After a fused pair one field holds f(t+1) only virtually, and reading it takes a temporary third population field
        (_materialise).  A run whose two fields just fit — the reference's protocol needs no more — must not fail with
        hipErrorOutOfMemory in an innocent read: without room for that field (plus 1/16 slack) the calls are not paired.
        hipMemGetInfo is asked at the first pairing and every 256th afterwards."""
        self._room_calls = getattr(self, "_room_calls", 0) + 1
        if getattr(self, "_room_ok", None) is None or self._room_calls % 256 == 0:
            info = f.info()
            need = info["plane_stride"] * info["cardinality"] * f.dtype.itemsize
            free, _ = self._ctx.mem_info()
            self._room_ok = free >= need + need // 16
        return self._room_ok

    def _flush_deferred(self, *_):
        """Enqueue the deferred step as a single step (some other use of its fields came first)."""
        d, self._deferred = self._deferred, None
        if d is None:
            return
        f_src, f_dst, bcm, miss, omega, t = d
        self._unhook(f_src, f_dst, bcm, miss)
        try:
            self._native_stepper().step(f_src, f_dst, bcm, miss, omega, t)
        except BaseException:
            # not enqueued: the step stays owed (the next use of any of its fields tries again and raises again)
            self._deferred = d
            for fld in (f_src, f_dst, bcm, miss):
                if fld is not None:
                    fld._hook = self._flush_deferred
            raise

    @staticmethod
    def _unhook(*fields):
        for fld in fields:
            if fld is not None:
                fld._hook = None

    def _materialise(self, field, bcm, miss, omega, t):
        """`field` should hold f(t+1) but its buffer holds f(t) (a fused pair passed it by): one single step through a
        temporary field, whose buffer the field then adopts."""
        # the temporary comes first: if it cannot be allocated this raises with the hook still owed (Field.handle puts it
        # back), so the next reader raises again instead of being handed f(t) for f(t+1)
        tmp = self.grid.create_field(cardinality=self.velocity_set.q, dtype=self.precision_policy.store_precision)
        try:
            self._native_stepper().step(field, tmp, bcm, miss, omega, t)  # (field._hook is None while its hook runs)
        except BaseException:
            tmp.free()
            raise
        self._n_materialised += 1
        field._h, tmp._h = tmp._h, field._h
        tmp.free()

    def _lazy_pair(self, f_0, f_1, bc_mask, missing_mask, omega, timestep):
        """True when this call was absorbed (deferred, or executed as the second half of a fused pair)."""
        if isinstance(f_1._hook, _Materialise):
            f_1._hook = None  # a virtual f(t+1) in the destination is about to be overwritten: nothing to materialise
        d = self._deferred
        if d is not None:
            f_src, f_dst, bcm, miss, om, t = d
            if f_0 is f_dst and f_1 is f_src and bc_mask is bcm and missing_mask is miss and float(omega) == om and not (f_0._pinned or f_1._pinned):
                # the caller swapped the fields: steps t and t + 1 in one pass, f(t) in f_src's buffer -> f(t+2) in f_dst's
                self._deferred = None
                self._unhook(f_src, f_dst, bcm, miss)
                if not self._native_stepper().step2_eligible(f_src, f_dst, bcm, miss):  # (an option changed in between)
                    self._native_stepper().step(f_src, f_dst, bcm, miss, om, t)
                    return False
                self._native_stepper().step2(f_src, f_dst, bcm, miss, om, t)
                self._n_fused_pairs += 1
                # the reference's protocol leaves f(t+2) in THIS call's f_1 (= f_src) and f(t+1) in its f_0 (= f_dst):
                # exchange the buffers; f_dst now holds f(t) instead of f(t+1) until someone looks
                f_src._h, f_dst._h = f_dst._h, f_src._h
                f_dst._hook = _Materialise(self, bcm, miss, om, t)
                return True
            self._flush_deferred()
        if not self._lazy_eligible(f_0, f_1, bc_mask, missing_mask):
            return False
        # defer: whoever touches either field first (or ctx.sync()) makes it run
        if f_0._hook is not None:  # the source itself is virtual (read it properly first)
            f_0.handle
        self._deferred = (f_0, f_1, bc_mask, missing_mask, float(omega), timestep)
        for fld in (f_0, f_1, bc_mask, missing_mask):  # the masks too: editing them must not overtake the deferred step
            if fld is not None:
                fld._hook = self._flush_deferred
        import weakref

        if not any(r() is self for r in self._ctx._flushers):
            self._ctx._flushers.append(weakref.ref(self))
        return True

    @staticmethod
    def _drop_virtual_destination(f_1, n_steps):
        """The first of n >= 1 steps overwrites f_1 entirely: a virtual f(t+1) in it (left by a fused pair of reference-style
        calls) needs no materialisation — which would allocate a third field and run an extra step for nothing."""
        if n_steps >= 1 and isinstance(f_1._hook, _Materialise):
            f_1._hook = None

    def run(self, f_0, f_1, bc_mask, missing_mask, omega, n_steps, first_timestep=0):
        """``n_steps`` x (step, swap) in native code; returns (f_current, f_other)."""
        self._flush_deferred()
        self._drop_virtual_destination(f_1, n_steps)
        if f_0.halo > 0 and self._ctx.get_option("external_halo"):
            return self._run_host_staged(f_0, f_1, bc_mask, missing_mask, omega, n_steps, first_timestep)
        in_b = self._native_stepper().run(f_0, f_1, bc_mask, missing_mask, omega, first_timestep, n_steps)
        return (f_1, f_0) if in_b else (f_0, f_1)

    def run_timed(self, f_0, f_1, bc_mask, missing_mask, omega, n_steps, first_timestep=0):
        """As :meth:`run`; also returns the device time in ms measured with HIP events."""
        self._flush_deferred()
        self._drop_virtual_destination(f_1, n_steps)
        if f_0.halo > 0 and self._ctx.get_option("external_halo"):
            import time

            self._ctx.sync()  # host-staged transport: no native loop to bracket with events; wall clock instead
            t0 = time.perf_counter()
            out = self._run_host_staged(f_0, f_1, bc_mask, missing_mask, omega, n_steps, first_timestep)
            self._ctx.sync()
            return out, (time.perf_counter() - t0) * 1e3
        in_b, ms = self._native_stepper().run_timed(f_0, f_1, bc_mask, missing_mask, omega, first_timestep, n_steps)
        return ((f_1, f_0) if in_b else (f_0, f_1)), ms

    def _run_host_staged(self, f_0, f_1, bc_mask, missing_mask, omega, n_steps, first_timestep):
        """The native run loop with the ghost planes moved by the host (``init_process_group(transport="host")``):
        same kernels and the same pairing of steps as ``xlbhip_run``, the debugging transport in between."""
        from ...distribute import HostStagedHalo, all_reduce_min

        native = self._native_stepper()
        halo = HostStagedHalo(self.grid, self.velocity_set)
        cur, oth, i = f_0, f_1, 0
        fuse = f_0.halo >= 2 and n_steps >= 2 and native.step2_eligible(cur, oth, bc_mask, missing_mask)
        # the message plan differs between pairs and single steps: the decision must be the same on every rank
        # (uneven slabs may sit on either side of the two-step kernel's chip-filling rule)
        fuse = bool(all_reduce_min(1.0 if fuse else 0.0))
        if fuse and bc_mask is not None and self.boundary_conditions:
            halo.exchange_masks(bc_mask, missing_mask)
        while i < n_steps:
            if fuse and n_steps - i >= 2:
                halo.exchange(cur, depth=2)
                native.step2(cur, oth, bc_mask, missing_mask, omega, first_timestep + i)
                i += 2
            else:
                halo.exchange(cur, depth=1)
                native.step(cur, oth, bc_mask, missing_mask, omega, first_timestep + i)
                i += 1
            cur, oth = oth, cur
        return cur, oth

    @property
    def has_post_streaming_bc(self):
        return any(bc.implementation_step == ImplementationStep.STREAMING for bc in self.boundary_conditions)
