"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement of the rest of SURVEY.md 8(f) rank 4: the WINDING and AABB_CLOSE mesh voxelisations, the wall distances
("weights") the mesh maskers hand to curved-wall boundary conditions, and HybridBC.

Reference (kernel backends only — none of this has a JAX branch or a reference test, so it is "parity unpinned by the
reference"; the functions below restate the Warp functionals):
  * xlb/operator/boundary_masker/ray.py:48-76            voxel links that cross the surface, distance along the link
  * xlb/operator/boundary_masker/winding.py:46-103       inside test by generalized winding number, rays out of solid voxels
  * xlb/operator/boundary_masker/aabb_close.py:67-365    AABB voxelisation + morphological close (dilate, erode)
  * xlb/operator/boundary_masker/mesh_boundary_masker.py:156-176   resolve_out_of_bound_kernel
  * xlb/operator/boundary_condition/bc_hybrid.py:254-358 + helper_functions_bc.py:160-340   HybridBC, three methods

Choices where the reference is a race or leaves the order open (stated once, mirrored by the HIP kernels):
  * Warp's BVH queries (`wp.mesh_query_ray`, `wp.mesh_query_point_sign_winding_number`) are replaced by exhaustive loops
    over the triangles: closest Moeller-Trumbore hit in fp32 (both faces), exact winding number in fp64 (solid angles,
    Van Oosterom & Strackee) with the same 0.5 threshold.
  * WINDING: a voxel that is inside the mesh is BC_SOLID even when a neighbouring solid voxel's ray tags it (the
    reference's two writes race); tags landing outside the box are dropped.
  * the distance along a link is the ray parameter t itself (|hit - centre| = t for a unit direction), and WINDING's
    |hit - centre(neighbour)| is len - t: no square roots, so fp32 results are reproducible bit for bit.
  * moments inside HybridBC are the plain sequential sums used everywhere else in this oracle (Warp compensates them).
"""

import numpy as np

from . import xlb_numpy as orc

BC_SOLID = orc.BC_SOLID
F = np.float32
LENS = {1: F(1.0), 2: F(1.41421356237309515), 3: F(1.73205080756887719)}

KIND_HYBRID_BB_REGULARIZED = "hybrid_bounceback_regularized"
KIND_HYBRID_BB_GRADS = "hybrid_bounceback_grads"
KIND_HYBRID_NEQ_REGULARIZED = "hybrid_nonequilibrium_regularized"
HYBRID_KINDS = (KIND_HYBRID_BB_REGULARIZED, KIND_HYBRID_BB_GRADS, KIND_HYBRID_NEQ_REGULARIZED)


def seg_tri_t(v, p, dvec, max_t):
    """Moeller-Trumbore in fp32, the operation order of ops_kernels.hpp: seg_tri_t.  Returns t (np.float32) or None."""
    v = v.astype(F)
    e1, e2 = v[1] - v[0], v[2] - v[0]
    dx, dy, dz = (F(x) for x in dvec)
    pv = np.array([dy * e2[2] - dz * e2[1], dz * e2[0] - dx * e2[2], dx * e2[1] - dy * e2[0]], F)
    det = (e1[0] * pv[0] + e1[1] * pv[1]) + e1[2] * pv[2]
    if abs(det) < F(1e-12):
        return None
    inv = F(1.0) / det
    t = np.asarray(p, F) - v[0]
    u = ((t[0] * pv[0] + t[1] * pv[1]) + t[2] * pv[2]) * inv
    if u < 0 or u > 1:
        return None
    q = np.array([t[1] * e1[2] - t[2] * e1[1], t[2] * e1[0] - t[0] * e1[2], t[0] * e1[1] - t[1] * e1[0]], F)
    w = ((dx * q[0] + dy * q[1]) + dz * q[2]) * inv
    if w < 0 or u + w > 1:
        return None
    tt = ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]) * inv
    return F(tt) if (tt >= 0 and tt <= F(max_t)) else None


def _dirs(lat):
    """[(l, (cx, cy, cz), |c| as fp32, unit direction as fp32 triple)] for every direction but the rest one"""
    out = []
    for l in range(lat.q):
        if l == lat.opp[l]:
            continue
        c = tuple(int(x) for x in lat.c[:, l])
        ln = LENS[c[0] * c[0] + c[1] * c[1] + c[2] * c[2]]
        out.append((l, c, ln, (F(c[0]) / ln, F(c[1]) / ln, F(c[2]) / ln)))
    return out


def closest_hit(verts, p, d, max_t):
    """Smallest ray parameter over all triangles (what a mesh ray query returns), or None."""
    best = None
    for v in verts:
        t = seg_tri_t(v, p, d, max_t)
        if t is not None and (best is None or t < best):
            best = t
    return best


def resolve_out_of_bound(shape, lat, bc_id, bc, mm):
    """mesh_boundary_masker.py:156-176: voxels of this id miss the directions pulled from outside the box"""
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


def mesh_mask_ray(shape, lat, bc_id, vertices, bc_mask, missing_mask, distances=None):
    """MeshMaskerRay with the wall distances (ray.py:48-76): for every voxel and direction whose link (from the voxel
    centre, length |c_l|) crosses the surface: bc id, missing[opp l], distances[l] = t / |c_l|.
    Returns (bc_mask, missing_mask, distances) copies."""
    verts = np.asarray(vertices, F).reshape(-1, 3, 3)
    bc, mm = bc_mask.copy(), missing_mask.astype(bool).copy()
    dist = None if distances is None else distances.copy()
    lo = np.maximum(np.floor(verts.reshape(-1, 3).min(axis=0)).astype(int) - 2, 0)
    hi = np.minimum(np.floor(verts.reshape(-1, 3).max(axis=0)).astype(int) + 1, np.array(shape) - 1)
    dirs = _dirs(lat)
    for i in range(lo[0], hi[0] + 1):
        for j in range(lo[1], hi[1] + 1):
            for k in range(lo[2], hi[2] + 1):
                p = (F(i) + F(0.5), F(j) + F(0.5), F(k) + F(0.5))
                near = [v for v in verts if np.all(v.min(axis=0) <= np.array(p) + 1.0) and np.all(v.max(axis=0) >= np.array(p) - 1.0)]
                for l, c, ln, d in dirs:
                    t = closest_hit(near, p, d, ln)
                    if t is not None:
                        bc[0, i, j, k] = bc_id
                        mm[lat.opp[l], i, j, k] = True
                        if dist is not None:
                            dist[l, i, j, k] = t / ln
    resolve_out_of_bound(shape, lat, bc_id, bc, mm)
    return bc, mm, dist


def winding_number(p, verts):
    """Generalized winding number of the triangle soup at p (fp64): sum of signed solid angles / 4 pi; +1 inside a
    closed surface whose triangles are counter-clockwise seen from outside."""
    a = verts[:, 0].astype(np.float64) - p
    b = verts[:, 1].astype(np.float64) - p
    c = verts[:, 2].astype(np.float64) - p
    la, lb, lc = np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1), np.linalg.norm(c, axis=1)
    num = np.einsum("ij,ij->i", a, np.cross(b, c))
    den = la * lb * lc + np.einsum("ij,ij->i", a, b) * lc + np.einsum("ij,ij->i", b, c) * la + np.einsum("ij,ij->i", c, a) * lb
    return float(np.sum(2.0 * np.arctan2(num, den)) / (4.0 * np.pi))


def mesh_mask_winding(shape, lat, bc_id, vertices, bc_mask, missing_mask, distances=None):
    """MeshMaskerWinding (winding.py:46-103): voxels whose centre has winding number > 0.5 are BC_SOLID; a ray from such a
    centre along c_l (length |c_l|) that crosses the surface tags the neighbour at +c_l with the id, missing[l] and
    distances[opp l] = |hit - centre(neighbour)| / |c_l| = (|c_l| - t) / |c_l|."""
    verts = np.asarray(vertices, F).reshape(-1, 3, 3)
    bc, mm = bc_mask.copy(), missing_mask.astype(bool).copy()
    dist = None if distances is None else distances.copy()
    lo = np.maximum(np.floor(verts.reshape(-1, 3).min(axis=0)).astype(int) - 1, 0)
    hi = np.minimum(np.floor(verts.reshape(-1, 3).max(axis=0)).astype(int) + 1, np.array(shape) - 1)
    solid = np.zeros(shape, bool)
    for i in range(lo[0], hi[0] + 1):
        for j in range(lo[1], hi[1] + 1):
            for k in range(lo[2], hi[2] + 1):
                solid[i, j, k] = winding_number(np.array([i + 0.5, j + 0.5, k + 0.5]), verts) > 0.5
    dirs = _dirs(lat)
    for i, j, k in zip(*np.nonzero(solid)):
        p = (F(i) + F(0.5), F(j) + F(0.5), F(k) + F(0.5))
        for l, c, ln, d in dirs:
            t = closest_hit(verts, p, d, ln)
            if t is None:
                continue
            n = (i + c[0], j + c[1], k + c[2])
            if min(n) < 0 or n[0] >= shape[0] or n[1] >= shape[1] or n[2] >= shape[2] or solid[n]:
                continue
            bc[(0,) + n] = bc_id
            mm[(l,) + n] = True
            if dist is not None:
                dist[(lat.opp[l],) + n] = (ln - t) / ln
    bc[0][solid] = BC_SOLID
    resolve_out_of_bound(shape, lat, bc_id, bc, mm)
    return bc, mm, dist


def aabb_close_solid(shape, vertices, close_voxels):
    """The closed solid mask of MeshMaskerAABBClose (aabb_close.py:130-154, 303-344): AABB voxelisation on a grid padded by
    2 * close_voxels per side, max filter then min filter over (2 h + 1)^3 cubes (cells within h of the padded grid's
    faces are copied unchanged), cropped back to the domain."""
    h = int(close_voxels)
    tl = 2 * h
    verts = np.asarray(vertices, F).reshape(-1, 3, 3)
    pshape = tuple(n + 2 * tl for n in shape)
    solid = np.zeros(pshape, bool)
    for v in verts:
        t = orc._tri_box_setup(v)
        if t is None:
            continue
        lo = np.maximum(np.floor(v.min(axis=0)).astype(int) - 1 + tl, 0)
        hi = np.minimum(np.floor(v.max(axis=0)).astype(int) + tl, np.array(pshape) - 1)
        for i in range(lo[0], hi[0] + 1):
            for j in range(lo[1], hi[1] + 1):
                for k in range(lo[2], hi[2] + 1):
                    if not solid[i, j, k] and orc._tri_box_overlap(t, (i - tl, j - tl, k - tl)):
                        solid[i, j, k] = True

    def morph(a, op):
        out = a.copy()
        if h == 0:
            return out
        core = tuple(slice(h, n - h) for n in pshape)
        acc = None
        for di in range(-h, h + 1):
            for dj in range(-h, h + 1):
                for dk in range(-h, h + 1):
                    sh = a[h + di : pshape[0] - h + di, h + dj : pshape[1] - h + dj, h + dk : pshape[2] - h + dk]
                    acc = sh.copy() if acc is None else op(acc, sh)
        out[core] = acc
        return out

    closed = morph(morph(solid, np.logical_or), np.logical_and)
    return closed[tl : pshape[0] - tl, tl : pshape[1] - tl, tl : pshape[2] - tl]


def mesh_mask_aabb_close(shape, lat, bc_id, vertices, close_voxels, bc_mask, missing_mask, distances=None):
    """MeshMaskerAABBClose (aabb_close.py:216-263): closed solid voxels (and voxels already BC_SOLID) are BC_SOLID; a fluid
    voxel with a solid neighbour at +c_l gets the id, missing[opp l] and distances[l] = (t - 0.5 |c_l|) / |c_l| for the
    closest hit within 1.5 |c_l| along c_l, 1.0 without a hit."""
    verts = np.asarray(vertices, F).reshape(-1, 3, 3)
    solid = aabb_close_solid(shape, vertices, close_voxels)
    bc, mm = bc_mask.copy(), missing_mask.astype(bool).copy()
    dist = None if distances is None else distances.copy()
    now_solid = solid | (bc[0] == BC_SOLID)
    pad = np.pad(solid, 1)
    for l, c, ln, d in _dirs(lat):
        nb = pad[1 + c[0] : 1 + c[0] + shape[0], 1 + c[1] : 1 + c[1] + shape[1], 1 + c[2] : 1 + c[2] + shape[2]] & ~now_solid
        bc[0][nb] = bc_id
        mm[lat.opp[l]] |= nb
        if dist is not None:
            for i, j, k in zip(*np.nonzero(nb)):
                p = (F(i) + F(0.5), F(j) + F(0.5), F(k) + F(0.5))
                t = closest_hit(verts, p, d, F(1.5) * ln)
                dist[l, i, j, k] = F(1.0) if t is None else (t - F(0.5) * ln) / ln
    bc[0][now_solid] = BC_SOLID
    resolve_out_of_bound(shape, lat, bc_id, bc, mm)
    return bc, mm, dist


# ---- HybridBC ------------------------------------------------------------------------------------------------------
class HybridBC(orc.BC):
    """Descriptor: kind in HYBRID_KINDS; u_wall = None (no-slip), the wall velocity (3,) or a per-cell field (3, nx, ny, nz) — the
    reference's profile(index), bc_hybrid.py:265; distances = None or the (q, ...)
    array of weights the mesh masker produced (weight of missing direction l sits in slot opp l, bc_hybrid.py:207-214)."""

    def __init__(self, kind, bc_id, indices, u_wall=None, distances=None):
        assert kind in HYBRID_KINDS
        super().__init__(orc.KIND_HALFWAY_BB, bc_id, indices)  # streaming step, needs_padding like the halfway wall (bc_hybrid.py:190-195)
        self.kind = kind
        self.u_wall = u_wall
        self.distances = distances


class HalfwayProfileBC(orc.BC):
    """Halfway bounce-back whose wall velocity is a per-cell field (3, nx, ny, nz): the kernel backends' HalfwayBounceBackBC(profile=...),
    bc_halfway_bounce_back.py:144-169 with moving_wall_fpop_correction (helper_functions_bc.py:230-250)."""

    def __init__(self, bc_id, indices, u_wall):
        super().__init__(orc.KIND_HALFWAY_BB, bc_id, indices)
        self.kind = "halfway_bounce_back_profile"
        self.u_wall = u_wall


def apply_halfway_profile(bc, f_pre, f_post, bc_mask, missing_mask, lat):
    T = f_post.dtype.type
    uw = np.asarray(bc.u_wall, np.float64).astype(T)
    boundary = bc_mask == bc.id
    out = f_post.copy()
    for l in range(lat.q):
        cond = np.logical_and(missing_mask[l].astype(bool), boundary[0])
        out[l] = np.where(cond, f_pre[lat.opp[l]] + _moving_term(lat, uw, l, T), f_post[l])
    return out


def _qi(lat, T):
    qi = lat.cc.astype(np.float64).copy()
    k = 0
    for a in range(lat.d):
        for b in range(a, lat.d):
            if a == b:
                qi[:, k] -= 1.0 / 3.0
            else:
                qi[:, k] *= 2.0
            k += 1
    return qi.astype(T)


def _moving_term(lat, uw, l, T):
    """helper_functions_bc.py:231-250: 6 w_l (c_l . u_wall), the sum over components in order"""
    cu = T(0.0)
    for a in range(lat.d):
        cl = int(lat.c[a, l])
        if cl == 1:
            cu = cu + uw[a]
        elif cl == -1:
            cu = cu - uw[a]
    return cu * (T(6.0) * T(lat.w[l]))


def apply_hybrid(bc, f_pre, f_post, bc_mask, missing_mask, lat, policy):
    """One HybridBC on the whole field (vectorised over cells; per-population order as in the Warp functionals)."""
    T = f_post.dtype.type
    q, opp = lat.q, lat.opp
    boundary = bc_mask == bc.id
    missing = missing_mask.astype(bool)
    moving = bc.u_wall is not None
    uw = None if not moving else np.asarray(bc.u_wall, np.float64).astype(T)
    use_dist = bc.distances is not None
    wgt = None if not use_dist else bc.distances.astype(T)
    w = lat.w.astype(T)
    one = T(1.0)
    out = f_post.copy()
    if bc.kind in (KIND_HYBRID_BB_REGULARIZED, KIND_HYBRID_BB_GRADS):
        # interpolated_bounceback, helper_functions_bc.py:253-292
        for l in range(q):
            if use_dist:
                wl = wgt[opp[l]]
                val = ((one - wl) * f_post[opp[l]] + wl * (f_pre[l] + f_pre[opp[l]])) / (one + wl)
            else:
                val = f_pre[opp[l]]
            val = np.where(missing[opp[l]], f_pre[opp[l]], val)  # sandwiched between two solid cells
            if moving:
                val = val + _moving_term(lat, uw, l, T)
            out[l] = np.where(missing[l], val, f_post[l])
    else:
        # interpolated_nonequilibrium_bounceback, helper_functions_bc.py:295-340
        rho, u = orc.macroscopic(f_pre, lat)
        feq = orc.equilibrium(rho, u, lat, T)
        if moving:
            uw_field = uw if uw.ndim > 1 else np.broadcast_to(uw.reshape((lat.d,) + (1,) * lat.d), u.shape)
            feq_wall = orc.equilibrium(rho, np.ascontiguousarray(uw_field).astype(T), lat, T)
        for l in range(q):
            wl = wgt[opp[l]] if use_dist else T(0.5)
            fneq = f_pre[opp[l]] - feq[opp[l]]
            fw = (feq_wall[l] if moving else w[l] * rho[0]) + fneq
            out[l] = np.where(missing[l], (fw + wl * f_pre[l]) / (one + wl), f_post[l])
    rho, u = orc.macroscopic(out, lat)
    qi = _qi(lat, T)
    if bc.kind == KIND_HYBRID_BB_GRADS:
        # grads_approximate_fpop, helper_functions_bc.py:186-228 (missing populations only)
        pi = orc.second_moment(out, lat)
        nt = pi.shape[0]
        res = out.copy()
        for l in range(q):
            qp = None
            for t in range(nt):
                term = qi[l, t] * ((pi[t] - rho[0] / T(3.0)) if t in (0, 3, 5) else pi[t])
                qp = term if qp is None else qp + term
            cu = T(0.0)
            for a in range(lat.d):
                cl = int(lat.c[a, l])
                if cl == 1:
                    cu = cu + u[a]
                elif cl == -1:
                    cu = cu - u[a]
            cu = cu * T(3.0)
            res[l] = np.where(missing[l], (rho[0] * w[l]) * (one + cu) + (w[l] * T(4.5)) * qp, out[l])
        out = res
    else:
        # regularize_fpop, helper_functions_bc.py:160-183 (every population)
        feq = orc.equilibrium(rho, u, lat, T)
        pi = orc.second_moment(out - feq, lat)
        res = np.empty_like(out)
        for l in range(q):
            qp = None
            for t in range(pi.shape[0]):
                term = qi[l, t] * pi[t]
                qp = term if qp is None else qp + term
            res[l] = feq[l] + (T(4.5) * w[l]) * qp
        out = res
    return np.where(np.broadcast_to(boundary, out.shape), out, f_post)


def momentum_transfer(f_0, bc, bc_mask, missing_mask, lat, policy="FP32FP32"):
    """MomentumTransfer for a HybridBC / HalfwayProfileBC descriptor: force/momentum_transfer.py:225-262 with FetchPopulations (:75-92) —
    f_post_stream = the BC applied to (f_0, stream(f_0)); the sum over the grid in NumPy's order (unpinned in the reference: atomics)."""
    T = orc.compute_dtype(policy)
    f_pc = f_0.astype(T)
    f_ps = orc.stream(f_pc, lat)
    if bc.kind in HYBRID_KINDS:
        f_ps = apply_hybrid(bc, f_pc, f_ps, bc_mask, missing_mask, lat, policy)
    else:
        f_ps = apply_halfway_profile(bc, f_pc, f_ps, bc_mask, missing_mask, lat)
    mm = missing_mask.astype(bool)
    boundary = np.broadcast_to(bc_mask == bc.id, mm.shape)
    is_edge = np.logical_and(boundary, ~mm[0])
    phi = np.where(np.logical_and(mm, is_edge), f_pc[lat.opp] + f_ps, T(0.0))
    force = np.tensordot(lat.c[:, lat.opp].astype(T), phi, axes=(-1, 0))
    return force.reshape(lat.d, -1).sum(axis=1, dtype=np.float64).astype(T)


def step(f_0, bc_mask, missing_mask, bcs, omega, lat, policy="FP32FP32", collision="BGK"):
    """orc.step with HybridBC descriptors allowed in the list (streaming step, list order; nse_stepper.py:237-282)."""
    T, S = orc.compute_dtype(policy), orc.store_dtype(policy)
    F0 = f_0.astype(T)
    post = orc.stream(F0, lat)
    for bc in bcs:
        if bc.kind in HYBRID_KINDS:
            post = apply_hybrid(bc, F0, post, bc_mask, missing_mask, lat, policy)
        elif bc.kind == "halfway_bounce_back_profile":
            post = apply_halfway_profile(bc, F0, post, bc_mask, missing_mask, lat)
        elif bc.step == orc.STEP_STREAMING:
            post = orc.apply_bc(bc, F0, post, bc_mask, missing_mask, lat, policy)
    rho, u = orc.macroscopic(post, lat)
    feq = orc.equilibrium(rho, u, lat, T)
    coll = orc.bgk(post, feq, omega) if collision == "BGK" else orc.kbc(post, feq, omega, lat)
    for bc in bcs:
        if bc.kind not in HYBRID_KINDS and bc.kind != "halfway_bounce_back_profile":
            coll = orc.assemble_auxiliary_data(bc, post, coll, bc_mask, missing_mask, lat)
            if bc.step == orc.STEP_COLLISION:
                coll = orc.apply_bc(bc, post, coll, bc_mask, missing_mask, lat, policy)
    return coll.astype(S)


def run(f_0, bc_mask, missing_mask, bcs, omega, lat, n_steps, policy="FP32FP32", collision="BGK"):
    f = f_0
    for _ in range(n_steps):
        f = step(f, bc_mask, missing_mask, bcs, omega, lat, policy, collision)
    return f
