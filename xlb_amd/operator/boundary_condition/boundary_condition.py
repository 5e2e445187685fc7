"""Boundary-condition base class, id registry, the four in-scope conditions and the Zou-He /
Regularized inlet-outlet pair.

Reference: xlb/operator/boundary_condition/boundary_condition.py:26-136 (base, flags,
``pad_indices``), boundary_condition_registry.py:6-30 (process-global id counter starting at 1,
ids assigned in CONSTRUCTION order), bc_equilibrium.py, bc_halfway_bounce_back.py,
bc_fullway_bounce_back.py, bc_do_nothing.py.

Each BC is (a) a stand-alone operator ``bc(f_pre, f_post, bc_mask, missing_mask) -> f_post``
and (b) a descriptor consumed by the fused stepper kernel (``_hip_descriptor``).
"""

from enum import Enum, auto

import numpy as np

from ... import _lib
from ...compute_backend import ComputeBackend
from ..equilibrium import Equilibrium, QuadraticEquilibrium
from ..operator import Operator


class ImplementationStep(Enum):
    COLLISION = auto()
    STREAMING = auto()


class BoundaryConditionRegistry:
    def __init__(self):
        self.id_to_bc = {}
        self.bc_to_id = {}
        self.next_id = 1  # 0 = no boundary condition

    def register_boundary_condition(self, boundary_condition):
        _id = self.next_id
        if _id > 253:
            raise ValueError("more than 253 boundary conditions registered in this process (bc_mask is uint8)")
        self.next_id += 1
        self.id_to_bc[_id] = boundary_condition
        self.bc_to_id[boundary_condition] = _id
        return _id


boundary_condition_registry = BoundaryConditionRegistry()


class _GridOfField:
    """What _profile_table needs from a grid, taken from a field (the stand-alone bc(...) call has no grid):
    single-rank fields only (a slab-decomposed field does not know its x offset)."""

    def __init__(self, field):
        if field.halo != 0:
            raise NotImplementedError("the stand-alone call of a profile BC needs fields without ghost planes; use the stepper")
        self.shape = self.local_shape = field.grid_shape
        self.x_offset, self.halo = 0, 0


class BoundaryCondition(Operator):
    hip_kind = None

    def __init__(
        self,
        implementation_step,
        velocity_set=None,
        precision_policy=None,
        compute_backend=None,
        indices=None,
        mesh_vertices=None,
        voxelization_method=None,
    ):
        self.id = boundary_condition_registry.register_boundary_condition(f"{self.__class__.__name__}_{id(self)}")
        super().__init__(velocity_set, precision_policy, compute_backend)
        if mesh_vertices is not None:
            # triangle soup in lattice units, (3 n_triangles, d) — consumed by MeshMaskerAABB (boundary_masker/mesh_boundary_masker.py)
            if indices is not None:
                raise ValueError("give either indices or mesh_vertices")
            mesh_vertices = np.asarray(mesh_vertices, dtype=np.float32)
            if mesh_vertices.ndim != 2 or mesh_vertices.shape[1] != 3 or mesh_vertices.shape[0] % 3 != 0:
                raise ValueError("Mesh points must be reshaped into an array (N, 3) where N indicates number of points (three per triangle)!")
        self.indices = indices
        self.mesh_vertices = mesh_vertices
        self.voxelization_method = voxelization_method
        self.implementation_step = implementation_step
        self.needs_padding = False
        self.needs_mesh_distance = False
        self.needs_aux_init = False
        self.is_initialized_with_aux_data = False
        self.num_of_aux_data = 0
        self.needs_aux_recovery = False

    def pad_indices(self):
        """Indices plus all their lattice neighbours (only when ``needs_padding``)."""
        idx = np.array(self.indices)
        if not self.needs_padding:
            return idx
        padded = idx[:, :, None] + self.velocity_set._c[:, None, :]
        return np.unique(padded.reshape(self.velocity_set.d, -1), axis=1)

    # descriptor for the C ABI: q constants in compute precision
    def _hip_values(self):
        return np.zeros(self.velocity_set.q)

    def _hip_descriptor(self):
        return _lib.make_bc_desc(self.id, self.hip_kind, self._hip_values())

    def _apply(self, f_pre, f_post, bc_mask, missing_mask):
        desc = self._hip_descriptor()
        if getattr(self, "prescribed_values", None) is not None:
            keys, vals = self._profile_table(_GridOfField(f_post))
            keys = np.ascontiguousarray(keys, dtype=np.uint32)
            vals = np.ascontiguousarray(vals, dtype=np.float64)
            _lib.check(
                _lib.load().xlbhip_apply_bc_profile(
                    self._ctx.handle, self.velocity_set.hip_id, self._compute_code, _lib.C.byref(desc), f_pre.handle, f_post.handle,
                    bc_mask.handle, missing_mask.handle if missing_mask is not None else None, int(keys.shape[0]), keys.ctypes.data,
                    vals.ctypes.data,
                )
            )
            return f_post
        import ctypes

        _lib.check(
            _lib.load().xlbhip_apply_bc(
                self._ctx.handle,
                self.velocity_set.hip_id,
                self._compute_code,
                ctypes.byref(desc),
                f_pre.handle,
                f_post.handle,
                bc_mask.handle,
                missing_mask.handle if missing_mask is not None else None,
            )
        )
        return f_post


class EquilibriumBC(BoundaryCondition):
    """f = feq(rho, u) at the tagged cells, after streaming."""

    hip_kind = _lib.BC_EQUILIBRIUM

    def __init__(self, rho, u, equilibrium_operator=None, velocity_set=None, precision_policy=None, compute_backend=None,
                 indices=None, mesh_vertices=None, voxelization_method=None):
        self.rho = rho
        self.u = u
        if equilibrium_operator is not None and not isinstance(equilibrium_operator, Equilibrium):
            raise ValueError("Equilibrium operator must be a subclass of Equilibrium")
        super().__init__(ImplementationStep.STREAMING, velocity_set, precision_policy, compute_backend, indices, mesh_vertices,
                         voxelization_method)
        self.equilibrium_operator = equilibrium_operator or QuadraticEquilibrium(self.velocity_set, self.precision_policy, self.compute_backend)

    def _hip_values(self):
        return self.equilibrium_operator.host_values(self.rho, self.u).astype(np.float64)

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f_pre, f_post, bc_mask, missing_mask):
        return self._apply(f_pre, f_post, bc_mask, missing_mask)


class HalfwayBounceBackBC(BoundaryCondition):
    """Missing populations are replaced by the opposite pre-streaming population of the same
    cell (+ 6 w_l c_l.u_wall for a moving wall), after streaming."""

    hip_kind = _lib.BC_HALFWAY_BB

    def __init__(self, velocity_set=None, precision_policy=None, compute_backend=None, indices=None, mesh_vertices=None,
                 voxelization_method=None, profile=None, prescribed_value=None):
        super().__init__(ImplementationStep.STREAMING, velocity_set, precision_policy, compute_backend, indices, mesh_vertices,
                         voxelization_method)
        self.needs_padding = True
        if profile is not None:
            if prescribed_value is not None:
                raise ValueError("Cannot specify both profile and prescribed_value")
            if not callable(profile):
                raise ValueError("profile must be a callable: cell indices (d, n) -> wall velocities (d, n)")
            # a time-independent wall velocity per boundary cell (the reference's profile(index) Warp function,
            # bc_halfway_bounce_back.py:76-96): evaluated by the stepper at this BC's cells, a sparse table for the kernel
            self.hip_kind = _lib.BC_HALFWAY_BB_PROFILE
        self.profile = profile
        self._profile_table_data = None
        self.needs_moving_wall_treatment = prescribed_value is not None or profile is not None
        if prescribed_value is None and profile is not None:
            prescribed_value = [0] * self.velocity_set.d
        elif prescribed_value is None:
            print(f"WARNING! Assuming no-slip condition for BC type = {self.__class__.__name__}!")
            prescribed_value = [0] * self.velocity_set.d
        if not isinstance(prescribed_value, (tuple, list, np.ndarray)):
            raise ValueError("Velocity prescribed_value must be a tuple, list, or array")
        self.prescribed_value = np.asarray(prescribed_value, dtype=np.float64)
        if self.prescribed_value.shape != (self.velocity_set.d,):
            raise ValueError(f"prescribed_value must have {self.velocity_set.d} components")

    def _evaluate_profile(self, cells, storage_keys):
        """Called by the stepper with this BC's cells ((3, n) global indices, x first) and their storage cell indices."""
        d = self.velocity_set.d
        vals = np.asarray(self.profile(cells[3 - d :]), dtype=np.float64)
        if vals.shape != (d, cells.shape[1]):
            raise ValueError(f"profile(indices) must return an array of shape (d, n) = {(d, cells.shape[1])}, got {vals.shape}")
        v3 = np.zeros((cells.shape[1], 3))
        v3[:, 3 - d :] = vals.astype(self.compute_dtype).astype(np.float64).T  # internal 3-component form
        self._profile_table_data = (np.asarray(storage_keys, dtype=np.uint32), v3)

    def _profile_table(self, grid):
        return self._profile_table_data

    def _hip_values(self):
        vs = self.velocity_set
        T, S = self.compute_dtype, self.store_dtype
        out = np.zeros(vs.q, dtype=T)
        if not self.needs_moving_wall_treatment or self.profile is not None:
            return out.astype(np.float64)
        # bc_halfway_bounce_back.py:97-102,124-128: u_wall in STORE precision, 6 * (w * (c . u_wall))
        uw = self.prescribed_value.astype(S)
        w = vs._w.astype(T)
        for l in range(vs.q):
            dot = S(0)
            for d in range(vs.d):
                dot = S(dot + S(int(vs._c[d, l])) * uw[d])
            out[l] = T(T(6.0) * T(w[l] * T(dot)))
        return out.astype(np.float64)

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f_pre, f_post, bc_mask, missing_mask):
        if self.profile is not None:
            raise NotImplementedError("a HalfwayBounceBackBC with a wall-velocity profile runs inside the stepper (its per-cell table lives there)")
        return self._apply(f_pre, f_post, bc_mask, missing_mask)


class FullwayBounceBackBC(BoundaryCondition):
    """Every population of a tagged cell is replaced by the opposite post-streaming population,
    after collision."""

    hip_kind = _lib.BC_FULLWAY_BB

    def __init__(self, velocity_set=None, precision_policy=None, compute_backend=None, indices=None, mesh_vertices=None,
                 voxelization_method=None):
        super().__init__(ImplementationStep.COLLISION, velocity_set, precision_policy, compute_backend, indices, mesh_vertices,
                         voxelization_method)

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f_pre, f_post, bc_mask, missing_mask):
        return self._apply(f_pre, f_post, bc_mask, missing_mask)


class DoNothingBC(BoundaryCondition):
    """Tagged cells keep their pre-streaming populations."""

    hip_kind = _lib.BC_DO_NOTHING

    def __init__(self, velocity_set=None, precision_policy=None, compute_backend=None, indices=None, mesh_vertices=None,
                 voxelization_method=None):
        super().__init__(ImplementationStep.STREAMING, velocity_set, precision_policy, compute_backend, indices, mesh_vertices,
                         voxelization_method)

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f_pre, f_post, bc_mask, missing_mask):
        return self._apply(f_pre, f_post, bc_mask, missing_mask)


class ZouHeBC(BoundaryCondition):
    """Zou-He inlet / outlet: non-equilibrium bounce-back of the missing populations around
    feq(rho, u) with either the normal velocity or the density prescribed
    (reference xlb/operator/boundary_condition/bc_zouhe.py:37-304, JAX semantics).

    ``prescribed_value``: a constant velocity vector / density.  ``profile``: a callable WITHOUT arguments returning an
    array ``(d, ...)`` (velocity) or ``(1, ...)`` (density) that is broadcast over the grid the way the JAX branch does
    (``_broadcast_prescribed_values``, bc_zouhe.py:179-214: missing axes are inserted after the first one, e.g. a
    ``(3, ny, nz)`` inlet profile applies to every x); NumPy instead of jax.numpy.  The values at this BC's cells go to
    the stepper as a sparse per-cell table (``xlbhip_stepper_set_bc_profile``) — the role the aux-data encoding in
    ``f_1`` plays for the reference's kernel backends."""

    _kinds = {"velocity": _lib.BC_ZOUHE_VELOCITY, "pressure": _lib.BC_ZOUHE_PRESSURE}

    def __init__(self, bc_type, profile=None, prescribed_value=None, velocity_set=None, precision_policy=None, compute_backend=None,
                 indices=None, mesh_vertices=None, voxelization_method=None):
        assert bc_type in ["velocity", "pressure"], f"type = {bc_type} not supported! Use 'pressure' or 'velocity'."
        self.bc_type = bc_type
        super().__init__(ImplementationStep.STREAMING, velocity_set, precision_policy, compute_backend, indices, mesh_vertices,
                         voxelization_method)
        self.hip_kind = self._kinds[bc_type]
        self.profile = profile
        self.prescribed_values = None
        if profile is not None:
            if prescribed_value is not None:
                raise ValueError("Cannot specify both profile and prescribed_value")
            if not callable(profile):
                raise ValueError("profile must be a callable returning an array")
            self.prescribed_values = np.asarray(profile(), dtype=np.float64)  # bc_zouhe.py:122-124
            lead = self.velocity_set.d if bc_type == "velocity" else 1
            if self.prescribed_values.ndim < 1 or self.prescribed_values.shape[0] != lead:
                raise ValueError(f"profile() must return an array whose first axis has {lead} component(s)")
            self.prescribed_value = None
            self.needs_padding = True
            if indices is None:
                raise ValueError("a profile BC needs indices (its per-cell values are evaluated there)")
            self._profile_cells = np.array(indices, dtype=np.int64)  # the masker drops bc.indices once the masks exist
            return
        if prescribed_value is None:
            raise ValueError("prescribed_value is required")
        if bc_type == "velocity":
            if not isinstance(prescribed_value, (tuple, list, np.ndarray)):
                raise ValueError("Velocity prescribed_value must be a tuple, list, or array-like")
            prescribed_value = np.asarray(prescribed_value, dtype=np.float64)
            if prescribed_value.shape != (self.velocity_set.d,):
                raise ValueError(f"prescribed_value must have {self.velocity_set.d} components")
        else:
            if not isinstance(prescribed_value, (int, float)):
                raise ValueError("Pressure prescribed_value must be a scalar (int or float)")
            prescribed_value = float(prescribed_value)
        if np.count_nonzero(prescribed_value) > 1:
            raise ValueError("This BC only supports normal prescribed values (only one non-zero element allowed)")
        self.prescribed_value = prescribed_value
        self.needs_padding = True

    def _values_at(self, cells, grid_shape):
        """The broadcast profile evaluated at ``cells`` ((d, n) global indices) -> (n, 3) values in the internal
        3-component form, rounded to the store precision first like a constant value (bc_zouhe.py:155-156)."""
        pv = self.prescribed_values
        d = self.velocity_set.d
        nd_target = d + 1
        if pv.ndim > nd_target:
            raise ValueError("prescribed_values has more dimensions than target_shape")
        if pv.ndim < nd_target:  # bc_zouhe.py:195-204: singleton axes go right after the first one
            pv = pv.reshape((pv.shape[0],) + (1,) * (nd_target - pv.ndim) + pv.shape[1:])
        for pd, td in zip(pv.shape[1:], grid_shape):
            if pd != 1 and pd != td:
                raise ValueError(f"Cannot broadcast dimension {pd} to {td}")
        idx = tuple(np.zeros_like(cells[a]) if pv.shape[1 + a] == 1 else cells[a] for a in range(d))
        vals = pv[(slice(None),) + idx]  # (lead, n)
        S = self.store_dtype
        vals = vals.astype(S).astype(np.float64)
        out = np.zeros((cells.shape[1], 3))
        if self.bc_type == "velocity":
            out[:, 3 - d :] = vals.T
        else:
            out[:, 0] = vals[0]
        return out

    def _profile_table(self, grid):
        """(storage cell indices, (n, 3) values) of this BC's cells on this rank — None for constant values."""
        if self.prescribed_values is None:
            return None
        cells = self._profile_cells
        d = self.velocity_set.d
        x0 = getattr(grid, "x_offset", 0)
        local = getattr(grid, "local_shape", grid.shape)
        halo = getattr(grid, "halo", 0)
        keep = (cells[0] >= x0) & (cells[0] < x0 + local[0])
        cells = cells[:, keep]
        vals = self._values_at(cells, grid.shape)
        s3 = (1,) + tuple(local) if d == 2 else tuple(local)
        c3 = np.vstack([np.zeros((1, cells.shape[1]), np.int64), cells]) if d == 2 else cells.copy()
        c3[0] = c3[0] - (x0 if d == 3 else 0) + halo
        keys = (c3[0] * s3[1] + c3[1]) * s3[2] + c3[2]
        return keys.astype(np.uint32), vals

    def _hip_values(self):
        S = self.store_dtype
        out = np.zeros(27)
        if self.prescribed_values is not None:
            return out  # per-cell values: see _profile_table
        if self.bc_type == "velocity":
            v = self.prescribed_value.astype(S).astype(np.float64)  # bc_zouhe.py:155-156: store precision first
            out[3 - self.velocity_set.d : 3] = v  # internal 3-component form
        else:
            out[0] = float(S(self.prescribed_value))
        return out

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f_pre, f_post, bc_mask, missing_mask):
        return self._apply(f_pre, f_post, bc_mask, missing_mask)


class RegularizedBC(ZouHeBC):
    """Zou-He followed by the regularisation of the boundary populations,
    f = feq + 9/2 w_l Q_l : Pi^neq (reference bc_regularized.py:25-137)."""

    _kinds = {"velocity": _lib.BC_REGULARIZED_VELOCITY, "pressure": _lib.BC_REGULARIZED_PRESSURE}

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f_pre, f_post, bc_mask, missing_mask):
        return self._apply(f_pre, f_post, bc_mask, missing_mask)


class ExtrapolationOutflowBC(BoundaryCondition):
    """Outflow by first-order extrapolation from the interior (Geier et al. 2015), JAX semantics of the reference
    (xlb/operator/boundary_condition/bc_extrapolation_outflow.py:36-145):

    * ONE outward normal per BC object, deduced from the index lists (``_get_normal_vectors``, :78-92);
    * streaming step: the missing populations take the cell's own opposite pre-stream population — which holds the
      auxiliary data written by the previous step;
    * after the collision (``assemble_auxiliary_data``, :104-134) the outgoing populations are overwritten with
      ``cs * f_post_stream(cell - normal) + (1 - cs) * f_post_stream(cell)``, cs = 1/sqrt(3) in the compute dtype.

    The fused stepper does the first part inside the step kernel and the second in a small pass over the outflow
    cells right after it.  The face must be a face of the (global) domain."""

    hip_kind = _lib.BC_EXTRAPOLATION_OUTFLOW

    def __init__(self, velocity_set=None, precision_policy=None, compute_backend=None, indices=None, mesh_vertices=None,
                 voxelization_method=None):
        super().__init__(ImplementationStep.STREAMING, velocity_set, precision_policy, compute_backend, indices, mesh_vertices,
                         voxelization_method)
        if indices is None:
            raise ValueError("ExtrapolationOutflowBC needs indices (its normal is deduced from them)")
        self._get_normal_vectors(indices)

    def _get_normal_vectors(self, indices):
        """bc_extrapolation_outflow.py:78-92: the axis whose most common coordinate occurs most often is the normal
        axis; it points outward (negative when that coordinate is 0)."""
        from collections import Counter

        freq_counts = [Counter(int(v) for v in coord).most_common(1)[0] for coord in indices]
        counts = np.array([count for _, count in freq_counts])
        elements = np.array([element for element, _ in freq_counts])
        self.normal = counts // counts.max()
        if elements[np.argmax(counts)] == 0:
            self.normal = self.normal * -1

    def _hip_values(self):
        T = self.compute_dtype
        out = np.zeros(27)
        out[3 - self.velocity_set.d : 3] = self.normal  # internal 3-component form
        cs = T(1.0) / np.sqrt(T(3.0))
        out[3] = float(cs)
        out[4] = float(T(1.0) - cs)
        return out

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f_pre, f_post, bc_mask, missing_mask):
        return self._apply(f_pre, f_post, bc_mask, missing_mask)


class HybridBC(BoundaryCondition):
    """Curved / moving wall boundary condition of the reference's kernel backends (bc_hybrid.py:40-391), 3-D only:

    * ``"bounceback_regularized"``  interpolated bounce-back (Bouzidi / Yu et al.) + Latt regularisation,
    * ``"bounceback_grads"``        interpolated bounce-back + Grad's approximation of the missing populations,
    * ``"nonequilibrium_regularized"``  Tao et al.'s non-equilibrium bounce-back + regularisation.

    ``prescribed_value`` = constant wall velocity (moving-wall treatment), none = no-slip; ``profile`` = a time-independent wall velocity
    per boundary cell (the reference's ``profile(index)`` Warp function, bc_hybrid.py:163-172, 265 — e.g. a rotating body,
    examples/cfd/rotating_sphere_3d.py:114-131): here a Python callable that takes the (3, n) integer array of this BC's cell indices and
    returns the (3, n) velocities; the stepper evaluates it once the masks exist and hands the values to the kernel as a sparse
    per-cell table.  ``use_mesh_distance`` needs
    ``mesh_vertices`` and a voxelisation method with distances (RAY, WINDING, AABB_CLOSE): the interpolation then uses the
    fractional distance to the surface along every cut link instead of the halfway assumption.  Time-dependent profiles
    (``profile(index, timestep)``) are out of scope."""

    _kinds = {"bounceback_regularized": _lib.BC_HYBRID_BB_REGULARIZED, "bounceback_grads": _lib.BC_HYBRID_BB_GRADS,
              "nonequilibrium_regularized": _lib.BC_HYBRID_NEQ_REGULARIZED}

    def __init__(self, bc_method, profile=None, prescribed_value=None, velocity_set=None, precision_policy=None, compute_backend=None,
                 indices=None, mesh_vertices=None, voxelization_method=None, use_mesh_distance=False):
        assert bc_method in self._kinds, (
            f"type = {bc_method} not supported! Use 'bounceback_regularized', 'bounceback_grads' or 'nonequilibrium_regularized'."
        )
        self.bc_method = bc_method
        self.hip_kind = self._kinds[bc_method]
        super().__init__(ImplementationStep.STREAMING, velocity_set, precision_policy, compute_backend, indices, mesh_vertices,
                         voxelization_method)
        if self.velocity_set.d == 2:
            raise NotImplementedError("This BC is not implemented in 2D!")
        if profile is not None:
            if prescribed_value is not None:
                raise AssertionError("Cannot specify both profile and prescribed_value")
            if not callable(profile):
                raise ValueError("profile must be a callable: cell indices (3, n) -> wall velocities (3, n)")
        self.profile = profile
        self._profile_table_data = None  # (storage cells, (n, 3) values): the stepper evaluates the profile at this BC's cells
        self.needs_moving_wall_treatment = prescribed_value is not None or profile is not None
        if prescribed_value is None and profile is not None:
            prescribed_value = [0, 0, 0]  # (placeholder: every cell of this BC reads its velocity from the table)
        elif prescribed_value is None:
            print(f"WARNING! Assuming no-slip condition for BC type = {self.__class__.__name__}_{self.bc_method}!")
            prescribed_value = [0, 0, 0]
        if not isinstance(prescribed_value, (tuple, list, np.ndarray)):
            raise ValueError("Velocity prescribed_value must be a tuple, list, or array")
        self.prescribed_value = np.asarray(prescribed_value, dtype=np.float64)
        if self.prescribed_value.shape != (3,):
            raise ValueError("prescribed_value must have 3 components")
        self.needs_mesh_distance = bool(use_mesh_distance)
        if self.needs_mesh_distance:
            self.needs_aux_recovery = True  # (reference flag; the weights live in the stepper's table here)
        if self.mesh_vertices is None:
            assert self.indices is not None
            assert self.needs_mesh_distance is False, 'To use mesh distance, please provide the mesh vertices using keyword "mesh_vertices"!'
            assert self.voxelization_method is None, "Voxelization method is only applicable when using mesh vertices!"
            self.needs_padding = True
        self._distance_table = None

    def _evaluate_profile(self, cells, storage_keys):
        """Called by the stepper with this BC's cells ((3, n) global indices) and their storage cell indices."""
        vals = np.asarray(self.profile(cells), dtype=np.float64)
        if vals.shape != (3, cells.shape[1]):
            raise ValueError(f"profile(indices) must return an array of shape (3, n) = {(3, cells.shape[1])}, got {vals.shape}")
        vals = vals.astype(self.compute_dtype).astype(np.float64)
        self._profile_table_data = (np.asarray(storage_keys, dtype=np.uint32), np.ascontiguousarray(vals.T))

    def _profile_table(self, grid):
        return self._profile_table_data

    def _hip_values(self):
        out = np.zeros(self.velocity_set.q)
        out[:3] = self.prescribed_value.astype(self.compute_dtype)
        out[3] = 1.0 if self.needs_moving_wall_treatment else 0.0
        out[4] = 1.0 if self.needs_mesh_distance else 0.0
        return out

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f_pre, f_post, bc_mask, missing_mask):
        if self.profile is not None:
            raise NotImplementedError("a HybridBC with a wall-velocity profile runs inside the stepper (its per-cell table lives there)")
        return self._apply(f_pre, f_post, bc_mask, missing_mask)
