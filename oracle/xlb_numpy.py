"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

NumPy restatement of the reference's per-timestep LBM hot path (the JAX branches
of hsalehipour/XLB, which are the parity target).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker.  The product (`xlb_amd`) never imports it.

Pinning status
--------------
* The reference ships NO golden vectors.  Its own tests hold seven closed-form
  known-answer properties for this path (equilibrium, macroscopic, BGK, stream,
  masker, EquilibriumBC, FullwayBB); `tests/test_oracle_reference_pins.py` checks
  this oracle against every one of them.
* KBC, HalfwayBounceBackBC, `missing_mask` contents and the composed multi-step
  stepper are NOT pinned by any reference test or fixture: for those the status
  is **parity unpinned by the reference**; they are cross-checked three ways
  (this file <-> oracle/lbm_ref.c <-> HIP) plus physics invariants.
* The reference itself cannot be imported here (it hard-imports `jax` and
  `warp`, neither installed) so no reference-generated vectors exist.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  Arithmetic order is fixed and explicit (sequential sums in
direction order, no FMA) so that the C restatement and the HIP kernels can
reproduce it bit for bit in the compute dtype.
"""

import itertools

import numpy as np

BC_NONE = 0  # xlb/cell_type.py:9

# implementation_step values, xlb/operator/boundary_condition/boundary_condition.py:26-30
STEP_COLLISION = 1
STEP_STREAMING = 2

KIND_EQUILIBRIUM = "equilibrium"
KIND_HALFWAY_BB = "halfway_bounce_back"
KIND_FULLWAY_BB = "fullway_bounce_back"
KIND_DO_NOTHING = "do_nothing"
KIND_EXTRAPOLATION_OUTFLOW = "extrapolation_outflow"
# SURVEY.md section 8f rank 1 ("next"): inlet / outlet family with constant prescribed values
KIND_ZOUHE_VELOCITY = "zouhe_velocity"
KIND_ZOUHE_PRESSURE = "zouhe_pressure"
KIND_REGULARIZED_VELOCITY = "regularized_velocity"
KIND_REGULARIZED_PRESSURE = "regularized_pressure"
ZOUHE_KINDS = (KIND_ZOUHE_VELOCITY, KIND_ZOUHE_PRESSURE, KIND_REGULARIZED_VELOCITY, KIND_REGULARIZED_PRESSURE)


# ----------------------------------------------------------------------------
# Lattices
# ----------------------------------------------------------------------------
class Lattice:
    """Lattice tables, re-derived with the constructions of
    xlb/velocity_set/d2q9.py:18-21, d3q19.py:19-27, d3q27.py:19-29 and
    xlb/velocity_set/velocity_set.py:63-83,139-253."""

    def __init__(self, name):
        name = name.upper()
        if name == "D2Q9":
            cx = [0, 0, 0, 1, -1, 1, -1, 1, -1]
            cy = [0, 1, -1, 0, 1, -1, 0, 1, -1]
            c = np.array(tuple(zip(cx, cy))).T
            w = np.array([4 / 9, 1 / 9, 1 / 9, 1 / 9, 1 / 36, 1 / 36, 1 / 9, 1 / 36, 1 / 36])
            d, q = 2, 9
        elif name == "D3Q19":
            c = np.array([ci for ci in itertools.product([0, -1, 1], repeat=3) if np.sum(np.abs(ci)) <= 2]).T
            d, q = 3, 19
            w = np.zeros(q)
            n1 = np.sum(np.abs(c), axis=0)
            w[n1 == 0] = 1.0 / 3.0
            w[n1 == 1] = 1.0 / 18.0
            w[n1 == 2] = 1.0 / 36.0
        elif name == "D3Q27":
            c = np.array(list(itertools.product([0, -1, 1], repeat=3))).T
            d, q = 3, 27
            w = np.zeros(q)
            n1 = np.sum(np.abs(c), axis=0)
            w[n1 == 0] = 8.0 / 27.0
            w[n1 == 1] = 2.0 / 27.0
            w[n1 == 2] = 1.0 / 54.0
            w[n1 == 3] = 1.0 / 216.0
        else:
            raise ValueError(name)
        self.name, self.d, self.q = name, d, q
        self.c = c.astype(np.int64)  # (d, q)
        self.w = w.astype(np.float64)
        ct = self.c.T.tolist()
        # velocity_set.py:182-195
        self.opp = np.array([ct.index((-self.c.T[i]).tolist()) for i in range(q)])
        # velocity_set.py:155-180: cc[:, k] = c_a * c_b for a <= b
        nt = d * (d + 1) // 2
        cc = np.zeros((q, nt))
        k = 0
        for a in range(d):
            for b in range(a, d):
                cc[:, k] = self.c[a] * self.c[b]
                k += 1
        self.cc = cc
        self.right = np.nonzero(self.c[0] == 1)[0]  # velocity_set.py:214-224
        self.left = np.nonzero(self.c[0] == -1)[0]  # velocity_set.py:226-236
        self.main = np.nonzero(np.sum(np.abs(self.c), axis=0) == 1)[0]
        self.center = int(np.nonzero(np.all(self.c == 0, axis=0))[0][0])


def compute_dtype(policy):
    """xlb/precision_policy.py:72-85"""
    return {"FP64FP64": np.float64, "FP64FP32": np.float64, "FP64FP16": np.float64, "FP32FP32": np.float32, "FP32FP16": np.float32}[policy]


def store_dtype(policy):
    """xlb/precision_policy.py:87-100"""
    return {"FP64FP64": np.float64, "FP64FP32": np.float32, "FP64FP16": np.float16, "FP32FP32": np.float32, "FP32FP16": np.float16}[policy]


# ----------------------------------------------------------------------------
# Grid helper
# ----------------------------------------------------------------------------
def bounding_box_indices(shape, remove_edges=False):
    """xlb/grid/grid.py:135-191 (verbatim semantics, np.indices based)."""
    origin = np.array([0, 0, 0])
    bounds = np.array(shape)
    if remove_edges:
        origin = origin + 1
        bounds = bounds - 1
    sx = slice(origin[0], bounds[0])
    sy = slice(origin[1], bounds[1])
    dim = len(bounds)
    grid = np.indices(shape)
    if dim == 2:
        nx, ny = shape
        box = {"bottom": grid[:, sx, 0], "top": grid[:, sx, ny - 1], "left": grid[:, 0, sy], "right": grid[:, nx - 1, sy]}
    else:
        nx, ny, nz = shape
        sz = slice(origin[2], bounds[2])
        box = {
            "bottom": grid[:, sx, sy, 0].reshape(3, -1),
            "top": grid[:, sx, sy, nz - 1].reshape(3, -1),
            "left": grid[:, 0, sy, sz].reshape(3, -1),
            "right": grid[:, nx - 1, sy, sz].reshape(3, -1),
            "front": grid[:, sx, 0, sz].reshape(3, -1),
            "back": grid[:, sx, ny - 1, sz].reshape(3, -1),
        }
    return {k: v.tolist() for k, v in box.items()}


# ----------------------------------------------------------------------------
# Operators (whole field, compute dtype T = f.dtype)
# ----------------------------------------------------------------------------
def stream(f, lat):
    """Pull streaming, periodic: out[l, x] = f[l, x - c_l].
    xlb/operator/stream/stream.py:29-62 (vmap of jnp.roll with shift c_l)."""
    out = np.empty_like(f)
    axes = tuple(range(lat.d))
    for l in range(lat.q):
        out[l] = np.roll(f[l], tuple(int(s) for s in lat.c[:, l]), axis=axes)
    return out


def zero_moment(f):
    """rho = sum_l f_l, sequential in l. xlb/operator/macroscopic/zero_moment.py:14-17"""
    rho = f[0].copy()
    for l in range(1, f.shape[0]):
        rho = rho + f[l]
    return rho[None]


def first_moment(f, rho, lat):
    """u_d = (sum_l c[d,l] f_l) / rho. xlb/operator/macroscopic/first_moment.py:14-18"""
    T = f.dtype.type
    u = np.zeros((lat.d,) + f.shape[1:], dtype=f.dtype)
    for d in range(lat.d):
        acc = np.zeros(f.shape[1:], dtype=f.dtype)
        for l in range(lat.q):
            cl = int(lat.c[d, l])
            if cl == 1:
                acc = acc + f[l]
            elif cl == -1:
                acc = acc - f[l]
        u[d] = acc / rho[0]
    del T
    return u


def macroscopic(f, lat):
    """xlb/operator/macroscopic/macroscopic.py:21-26"""
    rho = zero_moment(f)
    u = first_moment(f, rho, lat)
    return rho, u


def second_moment(fneq, lat):
    """Pi_k = sum_l cc[l,k] fneq_l. xlb/operator/macroscopic/second_moment.py:35-55"""
    nt = lat.cc.shape[1]
    pi = np.zeros((nt,) + fneq.shape[1:], dtype=fneq.dtype)
    for k in range(nt):
        acc = np.zeros(fneq.shape[1:], dtype=fneq.dtype)
        for l in range(lat.q):
            v = int(lat.cc[l, k])
            if v == 1:
                acc = acc + fneq[l]
            elif v == -1:
                acc = acc - fneq[l]
        pi[k] = acc
    return pi


def equilibrium(rho, u, lat, T):
    """feq_l = rho * w_l * (1 + cu (1 + 0.5 cu) - usqr), cu = 3 c_l.u, usqr = 1.5 u.u
    xlb/operator/equilibrium/quadratic_equilibrium.py:23-30"""
    rho = np.asarray(rho, dtype=T)
    u = np.asarray(u, dtype=T)
    w = lat.w.astype(T)
    usq = u[0] * u[0]
    for d in range(1, lat.d):
        usq = usq + u[d] * u[d]
    usqr = T(1.5) * usq
    feq = np.empty((lat.q,) + u.shape[1:], dtype=T)
    for l in range(lat.q):
        dot = np.zeros(u.shape[1:], dtype=T)
        for d in range(lat.d):
            cl = int(lat.c[d, l])
            if cl == 1:
                dot = dot + u[d]
            elif cl == -1:
                dot = dot - u[d]
        cu = T(3.0) * dot
        feq[l] = (rho[0] * w[l]) * ((T(1.0) + cu * (T(1.0) + T(0.5) * cu)) - usqr)
    return feq


def bgk(f, feq, omega):
    """fout = f - omega (f - feq). xlb/operator/collision/bgk.py:27-32"""
    T = f.dtype.type
    fneq = f - feq
    return f - T(omega) * fneq


def _shear_d3q27(pi):
    """xlb/operator/collision/kbc.py:96-145"""
    T = pi.dtype.type
    nxz = pi[0] - pi[5]
    nyz = pi[3] - pi[5]
    s = {}
    s[9] = s[18] = (T(2.0) * nxz - nyz) / T(6.0)
    s[3] = s[6] = (-nxz + T(2.0) * nyz) / T(6.0)
    s[1] = s[2] = (-nxz - nyz) / T(6.0)
    s[12] = s[24] = pi[1] / T(4.0)
    s[21] = s[15] = -pi[1] / T(4.0)
    s[10] = s[20] = pi[2] / T(4.0)
    s[19] = s[11] = -pi[2] / T(4.0)
    s[8] = s[4] = pi[4] / T(4.0)
    s[7] = s[5] = -pi[4] / T(4.0)
    return s


def _shear_d2q9(pi):
    """xlb/operator/collision/kbc.py:147-174 followed by the /4 of kbc.py:61"""
    T = pi.dtype.type
    n = pi[0] - pi[2]
    s = {}
    s[3] = s[6] = n
    s[2] = s[1] = -n
    s[8] = s[7] = pi[1]
    s[4] = s[5] = -pi[1]
    return {k: v / T(4.0) for k, v in s.items()}


def kbc(f, feq, omega, lat):
    """Entropic KBC collision. xlb/operator/collision/kbc.py:40-94"""
    T = f.dtype.type
    fneq = f - feq
    pi = second_moment(fneq, lat)
    if lat.name == "D3Q27":
        sd = _shear_d3q27(pi)
    elif lat.name == "D2Q9":
        sd = _shear_d2q9(pi)
    else:
        raise NotImplementedError("Velocity set not supported: " + lat.name)  # kbc.py:65-66
    delta_s = np.zeros_like(fneq)
    for k, v in sd.items():
        delta_s[k] = v
    beta = T(0.5) * T(omega)
    inv_beta = T(1.0) / beta
    delta_h = fneq - delta_s
    temp = delta_h / feq
    sp1 = temp[0] * delta_s[0]
    sp2 = temp[0] * delta_h[0]
    for l in range(1, lat.q):
        sp1 = sp1 + temp[l] * delta_s[l]
        sp2 = sp2 + temp[l] * delta_h[l]
    gamma = inv_beta - ((T(2.0) - inv_beta) * sp1) / (T(1e-32) + sp2)
    return f - beta * (T(2.0) * delta_s + gamma[None] * delta_h)


def smagorinsky_les_bgk(f, feq, omega, lat, smagorinsky_coef=0.17):
    """BGK with the Smagorinsky effective relaxation time.
    xlb/operator/collision/smagorinsky_les_bgk.py:44-60"""
    T = f.dtype.type
    fneq = f - feq
    pi = second_moment(fneq, lat)
    if lat.d == 3:
        diag, off = (0, 3, 5), (1, 2, 4)
    else:
        diag, off = (0, 2), (1,)
    sd = _seq_sum_arrays([pi[k] * pi[k] for k in diag])
    so = _seq_sum_arrays([pi[k] * pi[k] for k in off])
    strain = sd + T(2.0) * so
    tau0 = T(1.0) / T(omega)
    cs = T(smagorinsky_coef)
    tau = T(0.5) * (tau0 + np.sqrt(tau0 * tau0 + (T(36.0) * (cs * cs)) * np.sqrt(strain)))
    omega_eff = T(1.0) / tau
    return f - omega_eff[None] * fneq


def _seq_sum_arrays(terms):
    acc = terms[0]
    for t in terms[1:]:
        acc = acc + t
    return acc


def exact_difference_force(f_post, feq, lat, force_vector):
    """ForcedCollision + ExactDifference: moments of the post-collision populations, then
    f += feq(rho, u + F) - feq.  xlb/operator/collision/forced_collision.py:44-50,
    xlb/operator/force/exact_difference_force.py:61-83"""
    T = f_post.dtype.type
    rho, u = macroscopic(f_post, lat)
    du = np.asarray(force_vector, dtype=T).reshape((lat.d,) + (1,) * lat.d)
    feq_force = equilibrium(rho, u + du, lat, T)
    return f_post + (feq_force - feq)


# ----------------------------------------------------------------------------
# Boundary conditions
# ----------------------------------------------------------------------------
def outflow_normal(indices):
    """bc_extrapolation_outflow.py:78-92 (_get_normal_vectors)."""
    from collections import Counter

    freq = [Counter(int(v) for v in coord).most_common(1)[0] for coord in indices]
    counts = np.array([c for _, c in freq])
    elements = np.array([e for e, _ in freq])
    normal = counts // counts.max()
    if elements[np.argmax(counts)] == 0:
        normal = normal * -1
    return normal


def assemble_auxiliary_data(bc, f_pre, f_post, bc_mask, missing_mask, lat):
    """ExtrapolationOutflowBC.assemble_auxiliary_data, bc_extrapolation_outflow.py:104-134 (JAX); every other BC
    returns f_post unchanged (boundary_condition.py:138-144).  f_pre = post-stream, f_post = post-collision."""
    if bc.kind != KIND_EXTRAPOLATION_OUTFLOW:
        return f_post
    T = f_post.dtype.type
    axes = tuple(range(1, lat.d + 1))
    normal = tuple(int(v) for v in bc.normal)
    sound_speed = T(1.0) / np.sqrt(T(3.0))
    boundary = _bcast(bc_mask == bc.id, lat.q)
    neighbour = np.roll(boundary, tuple(-v for v in normal), axis=axes)
    fpop = np.where(boundary, f_pre, f_post)
    fpop_neighbour = np.where(neighbour, f_pre, f_post)
    fpop_neighbour = np.roll(fpop_neighbour, normal, axis=axes)
    fpop_extrapolated = sound_speed * fpop_neighbour + (T(1.0) - sound_speed) * fpop
    known_mask = missing_mask.astype(bool)[lat.opp]
    return np.where(np.logical_and(boundary, known_mask), fpop_extrapolated[lat.opp], f_post)


class BC:
    """Plain descriptor of an in-scope boundary condition.

    id semantics follow xlb/operator/boundary_condition/boundary_condition.py:68 and
    boundary_condition_registry.py:16-27: ids are handed out by the caller here
    (the oracle has no process-global registry)."""

    def __init__(self, kind, bc_id, indices, rho=None, u=None, u_wall=None, prescribed=None):
        self.kind = kind
        self.prescribed = prescribed  # ZouHe / Regularized: velocity vector or density scalar
        self.id = int(bc_id)
        self.indices = None if indices is None else np.asarray(indices, dtype=np.int64)
        self.rho = rho
        self.u = u
        self.u_wall = u_wall
        self.step = STEP_COLLISION if kind == KIND_FULLWAY_BB else STEP_STREAMING
        # bc_halfway_bounce_back.py:60 sets needs_padding; others keep the base False
        self.needs_padding = kind == KIND_HALFWAY_BB or kind in ZOUHE_KINDS  # bc_zouhe.py:146
        self.normal = outflow_normal(self.indices) if kind == KIND_EXTRAPOLATION_OUTFLOW else None

    def pad_indices(self, lat):
        """boundary_condition.py:123-136"""
        if self.needs_padding:
            padded = self.indices[:, :, None] + lat.c[:, None, :]
            return np.unique(padded.reshape(lat.d, -1), axis=1)
        return self.indices


def _bcast(mask, q):
    return np.broadcast_to(mask, (q,) + mask.shape[1:])


def apply_bc(bc, f_pre, f_post, bc_mask, missing_mask, lat, policy):
    """Dispatch on kind; JAX branches of bc_equilibrium.py:72-80,
    bc_halfway_bounce_back.py:116-134, bc_fullway_bounce_back.py:50-56,
    bc_do_nothing.py:50-54."""
    T = f_post.dtype.type
    boundary = bc_mask == bc.id  # (1, ...)
    if bc.kind == KIND_EQUILIBRIUM:
        feq = equilibrium(np.array([bc.rho], dtype=T), np.array(bc.u, dtype=T), lat, T)  # (q,)
        feq = feq.reshape((lat.q,) + (1,) * lat.d)
        return np.where(boundary, feq, f_post)
    if bc.kind == KIND_HALFWAY_BB:
        mw = T(0.0)
        if bc.u_wall is not None:
            S = store_dtype(policy)
            uw = np.asarray(bc.u_wall, dtype=np.float64).astype(S)
            w = lat.w.astype(T)
            comp = np.zeros(lat.q, dtype=T)
            for l in range(lat.q):
                dot = S(0)
                for d in range(lat.d):
                    dot = S(dot + S(int(lat.c[d, l])) * uw[d])
                comp[l] = T(6.0) * (w[l] * T(dot))
            mw = comp.reshape((lat.q,) + (1,) * lat.d)
        cond = np.logical_and(missing_mask.astype(bool), _bcast(boundary, lat.q))
        return np.where(cond, f_pre[lat.opp] + mw, f_post)
    if bc.kind == KIND_EXTRAPOLATION_OUTFLOW:
        # bc_extrapolation_outflow.py:137-145
        cond = np.logical_and(missing_mask.astype(bool), _bcast(boundary, lat.q))
        return np.where(cond, f_pre[lat.opp], f_post)
    if bc.kind == KIND_FULLWAY_BB:
        return np.where(_bcast(boundary, lat.q), f_pre[lat.opp], f_post)
    if bc.kind == KIND_DO_NOTHING:
        return np.where(_bcast(boundary, lat.q), f_pre, f_post)
    if bc.kind in ZOUHE_KINDS:
        return _zouhe(bc, f_post, boundary, missing_mask.astype(bool), lat, policy)
    raise ValueError(bc.kind)


def _seq_sum(terms):
    acc = terms[0]
    for t in terms[1:]:
        acc = acc + t
    return acc


def _zouhe(bc, f_post, boundary, missing, lat, policy):
    """Zou-He (bc_zouhe.py:166-304) and Regularized (bc_regularized.py:78-137) with a CONSTANT
    prescribed normal velocity vector or density; sums are sequential in direction order."""
    T = f_post.dtype.type
    S = store_dtype(policy)
    q, d = lat.q, lat.d
    opp = lat.opp
    # bc_zouhe.py:166-177
    known = missing[opp]
    middle = ~(missing | known)
    normals = [-_seq_sum([int(lat.c[a, l]) * missing[l].astype(np.int32) for l in lat.main]) for a in range(d)]
    fsum = _seq_sum([np.where(middle[l], f_post[l], T(0)) for l in range(q)]) + T(2.0) * _seq_sum(
        [np.where(known[l], f_post[l], T(0)) for l in range(q)]
    )
    velocity = bc.kind in (KIND_ZOUHE_VELOCITY, KIND_REGULARIZED_VELOCITY)
    profile = np.ndim(bc.prescribed) > 1  # the array a profile() returned, to be broadcast (bc_zouhe.py:179-214)
    if profile:
        pv = np.asarray(bc.prescribed, dtype=np.float64).astype(S).astype(T)
        if pv.ndim < d + 1:  # singleton axes are inserted right after the first one
            pv = pv.reshape((pv.shape[0],) + (1,) * (d + 1 - pv.ndim) + pv.shape[1:])
        pv = np.broadcast_to(pv, (pv.shape[0],) + f_post.shape[1:])
    if velocity and profile:
        vel = [pv[a] for a in range(d)]
        unormal = _seq_sum([normals[a].astype(T) * vel[a] for a in range(d)])
        rho = fsum / (T(1.0) + unormal)
    elif velocity:
        pv = np.asarray(bc.prescribed, dtype=np.float64).astype(S).astype(T)  # bc_zouhe.py:155-156
        vel = [np.full(f_post.shape[1:], pv[a], dtype=T) for a in range(d)]
        unormal = _seq_sum([normals[a].astype(T) * vel[a] for a in range(d)])  # :263
        rho = fsum / (T(1.0) + unormal)  # :265
    else:
        rho = pv[0] if profile else np.full(f_post.shape[1:], T(S(bc.prescribed)), dtype=T)
        unormal = T(-1.0) + fsum / rho  # :250
        vel = [unormal * normals[a].astype(T) for a in range(d)]  # :253
    feq = equilibrium(rho[None], np.stack(vel), lat, T)
    # bounceback_nonequilibrium, :280-288
    fknown = (f_post[opp] + feq) - feq[opp]
    fbd = np.where(missing, fknown, f_post)
    if bc.kind in (KIND_REGULARIZED_VELOCITY, KIND_REGULARIZED_PRESSURE):
        # regularize_fpop, bc_regularized.py:78-114; Qi = cc - cs^2 I with doubled off-diagonals (velocity_set.py:139-153)
        fneq = fbd - feq
        pi = second_moment(fneq, lat)
        qi = lat.cc.copy()
        k = 0
        for a in range(d):
            for b in range(a, d):
                if a == b:
                    qi[:, k] -= 1.0 / 3.0
                else:
                    qi[:, k] *= 2.0
                k += 1
        qi = qi.astype(T)
        w = lat.w.astype(T)
        out = np.empty_like(fbd)
        for l in range(q):
            qp = _seq_sum([qi[l, kk] * pi[kk] for kk in range(pi.shape[0])])
            out[l] = feq[l] + (T(9.0 / 2.0) * w[l]) * qp
        fbd = out
    return np.where(_bcast(boundary, q), fbd, f_post)


def are_indices_in_interior(indices, shape):
    """xlb/operator/boundary_masker/indices_boundary_masker.py:51-62"""
    d = len(shape)
    sh = np.array(shape)
    return np.all((indices[:d] > 0) & (indices[:d] < sh[:d, None] - 1), axis=0)


def build_masks(shape, lat, bcs, missing_in=None, bc_mask_in=None):
    """JAX masker, xlb/operator/boundary_masker/indices_boundary_masker.py:73-143,
    with nDevices == 1 (pad 1 in every direction).

    Returns bc_mask (1, *shape) uint8 and missing_mask (q, *shape) bool."""
    dim = len(shape)
    bc_mask = np.zeros((1,) + tuple(shape), dtype=np.uint8) if bc_mask_in is None else bc_mask_in.copy()
    missing = np.zeros((lat.q,) + tuple(shape), dtype=bool) if missing_in is None else missing_in.astype(bool)
    pad = ((1, 1),) * dim
    bm = np.pad(bc_mask[0], pad, constant_values=0)
    mm = np.pad(missing, ((0, 0),) + pad, constant_values=True)
    shift = np.ones((dim, 1), dtype=np.int64)
    for bc in bcs:
        idx = bc.indices
        if np.any(are_indices_in_interior(idx, shape)):
            solid = idx + shift
            mm[(slice(None),) + tuple(solid)] = True
            tagged = bc.pad_indices(lat) + shift
        else:
            tagged = idx + shift
        bm[tuple(tagged)] = bc.id
    mm = stream(mm, lat)
    crop = (slice(1, -1),) * dim
    bc_mask[0] = bm[crop]
    missing = mm[(slice(None),) + crop]
    return bc_mask, np.ascontiguousarray(missing)


# ----------------------------------------------------------------------------
# The step
# ----------------------------------------------------------------------------
def step(f_0, bc_mask, missing_mask, bcs, omega, lat, policy="FP32FP32", collision="BGK", force_vector=None):
    """One pull-scheme LBM step -> f_1 (store dtype).
    xlb/operator/stepper/nse_stepper.py:237-282."""
    T = compute_dtype(policy)
    S = store_dtype(policy)
    F0 = f_0.astype(T)
    post_stream = stream(F0, lat)
    for bc in bcs:
        if bc.step == STEP_STREAMING:
            post_stream = apply_bc(bc, F0, post_stream, bc_mask, missing_mask, lat, policy)
    rho, u = macroscopic(post_stream, lat)
    feq = equilibrium(rho, u, lat, T)
    if collision == "BGK":
        post_coll = bgk(post_stream, feq, omega)
    elif collision == "KBC":
        post_coll = kbc(post_stream, feq, omega, lat)
    elif collision == "SmagorinskyLESBGK":
        post_coll = smagorinsky_les_bgk(post_stream, feq, omega, lat)
    else:
        raise ValueError(collision)
    if force_vector is not None:
        post_coll = exact_difference_force(post_coll, feq, lat, force_vector)
    for bc in bcs:
        post_coll = assemble_auxiliary_data(bc, post_stream, post_coll, bc_mask, missing_mask, lat)  # nse_stepper.py:270-272
        if bc.step == STEP_COLLISION:
            post_coll = apply_bc(bc, post_stream, post_coll, bc_mask, missing_mask, lat, policy)
    return post_coll.astype(S)


def run(f_0, bc_mask, missing_mask, bcs, omega, lat, n_steps, policy="FP32FP32", collision="BGK", force_vector=None):
    """The caller's loop: step then swap (examples/cfd/lid_driven_cavity_2d.py:64-67)."""
    f = f_0
    for _ in range(n_steps):
        f = step(f, bc_mask, missing_mask, bcs, omega, lat, policy, collision, force_vector)
    return f


def initialize_eq(shape, lat, policy="FP32FP32"):
    """f = feq(rho=1, u=0) in compute dtype, stored. xlb/helper/initializers.py:25-72"""
    T = compute_dtype(policy)
    rho = np.ones((1,) + tuple(shape), dtype=T)
    u = np.zeros((lat.d,) + tuple(shape), dtype=T)
    return equilibrium(rho, u, lat, T).astype(store_dtype(policy))


# ----------------------------------------------------------------------------
# Drivers (what the reference's example scripts set up)
# ----------------------------------------------------------------------------
def cavity_2d(n, u_lid=0.05):
    """examples/cfd/lid_driven_cavity_2d.py:43-55: lid constructed first (id 1),
    walls second (id 2), listed [walls, lid]."""
    lat = Lattice("D2Q9")
    shape = (n, n)
    box = bounding_box_indices(shape)
    box_ne = bounding_box_indices(shape, remove_edges=True)
    lid = box_ne["top"]
    walls = [box["bottom"][i] + box["left"][i] + box["right"][i] for i in range(2)]
    walls = np.unique(np.array(walls), axis=-1).tolist()
    bc_top = BC(KIND_EQUILIBRIUM, 1, lid, rho=1.0, u=(u_lid, 0.0))
    bc_walls = BC(KIND_HALFWAY_BB, 2, walls)
    return lat, shape, [bc_walls, bc_top]


def cavity_3d(n, walls_kind=KIND_FULLWAY_BB, u_lid=0.02, lattice="D3Q19"):
    """examples/performance/mlups_3d.py:193-204: lid id 1 (EquilibriumBC), walls id 2,
    listed [lid, walls]."""
    lat = Lattice(lattice)
    shape = (n, n, n) if np.isscalar(n) else tuple(n)
    box = bounding_box_indices(shape)
    box_ne = bounding_box_indices(shape, remove_edges=True)
    lid = box_ne["top"]
    walls = [box["bottom"][i] + box["left"][i] + box["right"][i] + box["front"][i] + box["back"][i] for i in range(3)]
    walls = np.unique(np.array(walls), axis=-1).tolist()
    bc_lid = BC(KIND_EQUILIBRIUM, 1, lid, rho=1.0, u=(u_lid, 0.0, 0.0))
    bc_walls = BC(walls_kind, 2, walls)
    return lat, shape, [bc_lid, bc_walls]


def perturbed_init(shape, lat, policy="FP32FP32", seed=0, amp_rho=0.01, amp_u=0.01):
    """Synthetic non-equilibrium-free but non-trivial start:
    f = feq(1 + amp*xi, amp*eta), xi, eta ~ U(-1, 1), default_rng(seed)."""
    T = compute_dtype(policy)
    rng = np.random.default_rng(seed)
    rho = (1.0 + amp_rho * rng.uniform(-1, 1, size=(1,) + tuple(shape))).astype(T)
    u = (amp_u * rng.uniform(-1, 1, size=(lat.d,) + tuple(shape))).astype(T)
    return equilibrium(rho, u, lat, T).astype(store_dtype(policy))


# ---- post-processing (xlb/operator/postprocess/vorticity.py:30-84, q_criterion.py:36-131) -------------------------
def _velocity_gradient(u, bc_mask):
    """Central differences on the cells one layer inside the box; `ok` marks the cells the reference's kernels write
    (all six face neighbours fluid).  u: (3, nx, ny, nz)."""
    T = u.dtype.type
    c = (slice(1, -1),) * 3
    b = bc_mask[0]
    ok = (b[2:, 1:-1, 1:-1] == 0) & (b[1:-1, 2:, 1:-1] == 0) & (b[1:-1, 1:-1, 2:] == 0) & (b[:-2, 1:-1, 1:-1] == 0) & (b[1:-1, :-2, 1:-1] == 0) & (b[1:-1, 1:-1, :-2] == 0)

    def d(a, axis):
        hi = [slice(1, -1)] * 3
        lo = [slice(1, -1)] * 3
        hi[axis] = slice(2, None)
        lo[axis] = slice(None, -2)
        return (u[a][tuple(hi)] - u[a][tuple(lo)]) / T(2.0)

    g = [[d(a, axis) for axis in range(3)] for a in range(3)]  # g[a][axis] = du_a / dx_axis
    return g, ok, c


def vorticity(u, bc_mask, vort, mag):
    """Vorticity()(u, bc_mask, vorticity, vorticity_magnitude): returns updated copies of (vorticity, magnitude)."""
    g, ok, c = _velocity_gradient(u, bc_mask)
    vx = g[2][1] - g[1][2]
    vy = g[0][2] - g[2][0]
    vz = g[1][0] - g[0][1]
    m = np.sqrt((vx * vx + vy * vy) + vz * vz)
    vort, mag = vort.copy(), mag.copy()
    for a, v in enumerate((vx, vy, vz)):
        vort[a][c] = np.where(ok, v, vort[a][c])
    mag[0][c] = np.where(ok, m, mag[0][c])
    return vort, mag


def q_criterion(u, bc_mask, norm_mu, q):
    """QCriterion()(u, bc_mask, norm_mu, q): returns updated copies of (norm_mu, q)."""
    T = u.dtype.type
    g, ok, c = _velocity_gradient(u, bc_mask)
    vx = g[2][1] - g[1][2]
    vy = g[0][2] - g[2][0]
    vz = g[1][0] - g[0][1]
    m = np.sqrt((vx * vx + vy * vy) + vz * vz)
    h = T(0.5)
    s01, s02, s12 = h * (g[0][1] + g[1][0]), h * (g[0][2] + g[2][0]), h * (g[1][2] + g[2][1])
    ss = _seq_sum_arrays([g[0][0] * g[0][0], s01 * s01, s02 * s02, s01 * s01, g[1][1] * g[1][1], s12 * s12, s02 * s02, s12 * s12, g[2][2] * g[2][2]])
    o01, o02, o12 = h * (g[0][1] - g[1][0]), h * (g[0][2] - g[2][0]), h * (g[1][2] - g[2][1])
    z = np.zeros_like(o01)
    oo = _seq_sum_arrays([z, o01 * o01, o02 * o02, (-o01) * (-o01), z, o12 * o12, (-o02) * (-o02), (-o12) * (-o12), z])
    qv = h * (oo - ss)
    norm_mu, q = norm_mu.copy(), q.copy()
    norm_mu[0][c] = np.where(ok, m, norm_mu[0][c])
    q[0][c] = np.where(ok, qv, q[0][c])
    return norm_mu, q


def momentum_transfer(f_0, bc, bc_mask, missing_mask, lat, policy="FP32FP32"):
    """MomentumTransfer.jax_implementation, force/momentum_transfer.py:167-205 (stream-then-collide): f_0 = post-collision.
    Returns the net force (d,) in the compute dtype; the reduction order over the grid is NumPy's (the reference's is
    XLA's: unpinned), so callers compare with a tolerance."""
    T = compute_dtype(policy)
    f_pc = f_0.astype(T)
    f_ps = stream(f_pc, lat)
    f_ps = apply_bc(bc, f_pc, f_ps, bc_mask, missing_mask, lat, policy)
    mm = missing_mask.astype(bool)
    boundary = _bcast(bc_mask == bc.id, lat.q)
    is_edge = np.logical_and(boundary, ~mm[0])
    phi = f_pc[lat.opp] + f_ps
    phi = np.where(np.logical_and(mm, is_edge), phi, T(0.0))
    force = np.tensordot(lat.c[:, lat.opp].astype(T), phi, axes=(-1, 0))
    return force.reshape(lat.d, -1).sum(axis=1, dtype=np.float64).astype(T)


def sphere_channel(shape=(28, 14, 14), u_max=0.04, inlet_kind=KIND_REGULARIZED_VELOCITY):
    """The boundary-condition set of examples/cfd/flow_past_sphere_3d.py:104-112 on a small box: fullway walls (id 1), a
    velocity inlet with a parabolic PROFILE (id 2), extrapolation outflow (id 3), a halfway sphere from interior indices
    (id 4) — ids in the example's construction order.  Returns (lattice, bcs, profile array (3, ny, nz))."""
    lat = Lattice("D3Q19")
    box = bounding_box_indices(shape)
    box_ne = bounding_box_indices(shape, remove_edges=True)
    walls = [sum((list(box[f][i]) for f in ("bottom", "top", "front", "back")), []) for i in range(3)]
    walls = np.unique(np.array(walls), axis=-1)
    ny, nz = shape[1], shape[2]
    y, z = np.meshgrid(np.arange(ny), np.arange(nz), indexing="ij")
    hy, hz = ny - 1.0, nz - 1.0
    r2 = (2.0 * (y - hy / 2.0) / hy) ** 2 + (2.0 * (z - hz / 2.0) / hz) ** 2
    ux = u_max * np.maximum(0.0, 1.0 - r2)
    prof = np.stack([ux, np.zeros_like(ux), np.zeros_like(ux)])
    x, y, z = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    r = 0.23 * shape[1]
    sphere = np.array(np.where((x - shape[0] // 3) ** 2 + (y - shape[1] // 2) ** 2 + (z - shape[2] // 2) ** 2 < r * r))
    bcs = [BC(KIND_FULLWAY_BB, 1, walls), BC(inlet_kind, 2, box_ne["left"], prescribed=prof), BC(KIND_EXTRAPOLATION_OUTFLOW, 3, box_ne["right"]),
           BC(KIND_HALFWAY_BB, 4, sphere)]
    return lat, bcs, prof


def grid_to_point(grid, points):
    """GridToPoint, postprocess/grid_to_point.py:28-94: trilinear interpolation of grid[0] at points (n, 3) float32."""
    T = grid.dtype.type
    p = np.asarray(points, dtype=np.float32)
    lo = p.astype(np.int32)
    d = p - lo.astype(np.float32)
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    one = np.float32(1.0)
    g = lambda a, b, c: grid[0, lo[:, 0] + a, lo[:, 1] + b, lo[:, 2] + c]  # noqa: E731
    v = ((one - dx) * (one - dy) * (one - dz)).astype(T) * g(0, 0, 0)
    v = v + ((one - dx) * (one - dy) * dz).astype(T) * g(0, 0, 1)
    v = v + ((one - dx) * dy * (one - dz)).astype(T) * g(0, 1, 0)
    v = v + ((one - dx) * dy * dz).astype(T) * g(0, 1, 1)
    v = v + (dx * (one - dy) * (one - dz)).astype(T) * g(1, 0, 0)
    v = v + (dx * (one - dy) * dz).astype(T) * g(1, 0, 1)
    v = v + (dx * dy * (one - dz)).astype(T) * g(1, 1, 0)
    v = v + (dx * dy * dz).astype(T) * g(1, 1, 1)
    return v


# ---- MeshMaskerAABB (boundary_masker/aabb.py:38-100, mesh_boundary_masker.py:62-181) ---------------------------------
BC_SOLID = 255


def _tri_box_setup(v):
    """Schwarz & Seidel (2010) triangle / unit-box overlap, the precomputation of mesh_boundary_masker.py:62-93; fp32."""
    F = np.float32
    v = v.astype(F)
    e = np.stack([v[1] - v[0], v[2] - v[1], v[0] - v[2]])
    m = -e[2]
    n = np.array([e[0][1] * m[2] - e[0][2] * m[1], e[0][2] * m[0] - e[0][0] * m[2], e[0][0] * m[1] - e[0][1] * m[0]], dtype=F)
    ln = np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])
    if not ln > 0:
        return None
    n = (n / ln).astype(F)
    c = (n > 0).astype(F)
    d1 = n[0] * (c[0] - v[0][0]) + n[1] * (c[1] - v[0][1]) + n[2] * (c[2] - v[0][2])
    d2 = n[0] * ((F(1) - c[0]) - v[0][0]) + n[1] * ((F(1) - c[1]) - v[0][1]) + n[2] * ((F(1) - c[2]) - v[0][2])
    ne = np.zeros((3, 3, 2), F)
    de = np.zeros((3, 3), F)
    for ax0 in range(3):
        ax1, ax2 = (ax0 + 1) % 3, (ax0 + 2) % 3
        sgn = F(-1.0) if n[ax2] < 0 else F(1.0)
        for i in range(3):
            a0, a1 = -sgn * e[i][ax1], sgn * e[i][ax0]
            ne[ax0, i] = (a0, a1)
            de[ax0, i] = -(a0 * v[i][ax0] + a1 * v[i][ax1]) + max(F(0), a0) + max(F(0), a1)
    return n, F(d1), F(d2), ne, de


def _tri_box_overlap(t, low):
    n, d1, d2, ne, de = t
    F = np.float32
    low = np.asarray(low, F)
    npv = n[0] * low[0] + n[1] * low[1] + n[2] * low[2]
    if not (npv + d1) * (npv + d2) <= 0:
        return False
    for ax0 in range(3):
        ax1 = (ax0 + 1) % 3
        for i in range(3):
            if not ne[ax0, i, 0] * low[ax0] + ne[ax0, i, 1] * low[ax1] + de[ax0, i] >= 0:
                return False
    return True


def mesh_mask_aabb(shape, lat, bc_id, vertices, bc_mask, missing_mask):
    """MeshMaskerAABB on a (3 n_triangles, 3) triangle soup; returns updated copies of (bc_mask, missing_mask)."""
    verts = np.asarray(vertices, np.float32).reshape(-1, 3, 3)
    solid = np.zeros(shape, bool)
    for v in verts:
        t = _tri_box_setup(v)
        if t is None:
            continue
        lo = np.maximum(np.floor(v.min(axis=0)).astype(int) - 1, 0)
        hi = np.minimum(np.floor(v.max(axis=0)).astype(int), np.array(shape) - 1)
        for i in range(lo[0], hi[0] + 1):
            for j in range(lo[1], hi[1] + 1):
                for k in range(lo[2], hi[2] + 1):
                    if not solid[i, j, k] and _tri_box_overlap(t, (i, j, k)):
                        solid[i, j, k] = True
    bc = bc_mask.copy()
    mm = missing_mask.astype(bool).copy()
    was_solid = bc[0] == BC_SOLID
    now_solid = was_solid | solid
    nb = np.zeros((lat.q,) + tuple(shape), bool)  # nb[l]: the neighbour at +c_l is a mesh voxel
    pad = np.pad(solid, 1)
    for l in range(lat.q):
        if l == lat.opp[l]:
            continue
        cx, cy, cz = (int(v) for v in lat.c[:, l])
        nb[l] = pad[1 + cx : 1 + cx + shape[0], 1 + cy : 1 + cy + shape[1], 1 + cz : 1 + cz + shape[2]]
    boundary = nb.any(axis=0) & ~now_solid
    bc[0][now_solid] = BC_SOLID
    bc[0][boundary] = bc_id
    for l in range(lat.q):
        if l != lat.opp[l]:
            mm[lat.opp[l]] |= nb[l] & ~now_solid
    # resolve_out_of_bound_kernel: voxels of this id miss the directions pulled from outside the box
    idx = np.indices(shape)
    has_id = bc[0] == bc_id
    for l in range(lat.q):
        if l == lat.opp[l]:
            continue
        outside = np.zeros(shape, bool)
        for a in range(3):
            p = idx[a] - int(lat.c[a, l])
            outside |= (p < 0) | (p >= shape[a])
        mm[l] |= has_id & outside
    return bc, mm


def _seg_tri_hit(v, p, dvec, max_t):
    """Moeller-Trumbore, fp32, the operation order of ops_kernels.hpp: seg_tri_hit"""
    F = np.float32
    v = v.astype(F)
    e1, e2 = v[1] - v[0], v[2] - v[0]
    dx, dy, dz = (F(x) for x in dvec)
    pv = np.array([dy * e2[2] - dz * e2[1], dz * e2[0] - dx * e2[2], dx * e2[1] - dy * e2[0]], F)
    det = (e1[0] * pv[0] + e1[1] * pv[1]) + e1[2] * pv[2]
    if abs(det) < F(1e-12):
        return False
    inv = F(1.0) / det
    t = np.asarray(p, F) - v[0]
    u = ((t[0] * pv[0] + t[1] * pv[1]) + t[2] * pv[2]) * inv
    if u < 0 or u > 1:
        return False
    q = np.array([t[1] * e1[2] - t[2] * e1[1], t[2] * e1[0] - t[0] * e1[2], t[0] * e1[1] - t[1] * e1[0]], F)
    w = ((dx * q[0] + dy * q[1]) + dz * q[2]) * inv
    if w < 0 or u + w > 1:
        return False
    tt = ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]) * inv
    return bool(tt >= 0 and tt <= F(max_t))


def mesh_mask_ray(shape, lat, bc_id, vertices, bc_mask, missing_mask):
    """MeshMaskerRay (boundary_masker/ray.py:38-76); returns updated copies of (bc_mask, missing_mask)."""
    F = np.float32
    verts = np.asarray(vertices, F).reshape(-1, 3, 3)
    bc = bc_mask.copy()
    mm = missing_mask.astype(bool).copy()
    lens = {1: F(1.0), 2: F(1.41421356237309515), 3: F(1.73205080756887719)}
    for v in verts:
        lo = np.maximum(np.floor(v.min(axis=0)).astype(int) - 2, 0)
        hi = np.minimum(np.floor(v.max(axis=0)).astype(int) + 1, np.array(shape) - 1)
        for i in range(lo[0], hi[0] + 1):
            for j in range(lo[1], hi[1] + 1):
                for k in range(lo[2], hi[2] + 1):
                    p = (F(i) + F(0.5), F(j) + F(0.5), F(k) + F(0.5))
                    for l in range(lat.q):
                        if l == lat.opp[l] or mm[lat.opp[l], i, j, k] and bc[0, i, j, k] == bc_id:
                            continue
                        c = lat.c[:, l].astype(int)
                        ln = lens[int((c * c).sum())]
                        if _seg_tri_hit(v, p, (F(c[0]) / ln, F(c[1]) / ln, F(c[2]) / ln), ln):
                            bc[0, i, j, k] = bc_id
                            mm[lat.opp[l], i, j, k] = True
    idx = np.indices(shape)
    has_id = bc[0] == bc_id
    for l in range(lat.q):
        if l == lat.opp[l]:
            continue
        outside = np.zeros(shape, bool)
        for a in range(3):
            pp = idx[a] - int(lat.c[a, l])
            outside |= (pp < 0) | (pp >= shape[a])
        mm[l] |= has_id & outside
    return bc, mm

