"""Known-answer pins from THEORY for the parts of the path the reference's own tests do not pin (the composed
stepper, HalfwayBounceBackBC, forcing).  They hold for any correct implementation of the reference's formulas and are
independent of this repository's oracle / kernels agreeing with each other:

  * Taylor-Green vortex in a periodic D2Q9 box: the velocity amplitude decays as exp(-2 nu k^2 t) with the BGK viscosity
    nu = (1 / omega - 1 / 2) / 3  (bgk.py:27-32 + quadratic_equilibrium.py:23-30 + stream.py:29-62 composed as
    nse_stepper.py:237-282);
  * the same vortex under KBC (D2Q9, D3Q27): the shear part is relaxed with 2 beta = omega, so the decay rate — hence the
    hard-coded shear tables of kbc.py:120-143 / :163-172 — is pinned too;
  * Couette flow between two halfway bounce-back walls: the steady profile is the straight line through walls that sit HALF A
    CELL outside the boundary nodes, for every relaxation rate.  An independently written textbook step (own lattice order,
    own formulas) reproduces it to rounding, and the oracle's / the HIP backend's HalfwayBounceBackBC step (moving wall
    included: bc_halfway_bounce_back.py:97-134) equals that textbook step to 1e-15 — away from the x faces, where the
    reference's masker additionally bounces the directions that leave the box (App. B.2), which is why the channel itself
    cannot be run through the reference's semantics with a periodic x.

Run on the oracle (CPU) and, marked gpu, on the HIP backend through the same operator API."""

import numpy as np
import pytest

from oracle import xlb_numpy as orc


def taylor_green_init(n, u0, lat, policy):
    T = orc.compute_dtype(policy)
    k = 2.0 * np.pi / n
    x, y = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    ux = -u0 * np.cos(k * x) * np.sin(k * y)
    uy = u0 * np.sin(k * x) * np.cos(k * y)
    rho = 1.0 - 0.75 * u0 * u0 * (np.cos(2 * k * x) + np.cos(2 * k * y))
    f = orc.equilibrium(rho[None].astype(T), np.stack([ux, uy]).astype(T), lat, T)
    return f.astype(orc.store_dtype(policy)), k


def amplitude(f, lat):
    _, u = orc.macroscopic(f.astype(np.float64), lat)
    return float(np.sqrt((u**2).mean()))


# ---- an independently written textbook D2Q9 BGK step: collide, pull-stream, halfway bounce-back on the y walls -----------
# (own lattice ordering, own formulas; shares nothing with oracle/ or xlb_amd/)
TB_C = np.array([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1], [1, 1], [-1, 1], [-1, -1], [1, -1]])
TB_W = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4)
TB_OPP = [0, 3, 4, 1, 2, 7, 8, 5, 6]


def textbook_collide(f, omega):
    rho = f.sum(0)
    u = np.einsum("qd,qxy->dxy", TB_C, f) / rho
    cu = np.einsum("qd,dxy->qxy", TB_C, u)
    feq = TB_W[:, None, None] * rho * (1 + 3 * cu + 4.5 * cu**2 - 1.5 * (u**2).sum(0))
    return f - omega * (f - feq)


def textbook_stream_bounce(fs, u_top):
    """pull streaming, periodic in x; the wall below row 0 rests, the wall above row ny - 1 moves with (u_top, 0):
    f_q(b, t+1) = f*_opp(q)(b, t) + 2 w_q rho_0 (c_q . u_wall) / c_s^2 for the populations coming out of a wall"""
    ny = fs.shape[2]
    fn = np.empty_like(fs)
    for q in range(9):
        fn[q] = np.roll(fs[q], shift=(TB_C[q, 0], TB_C[q, 1]), axis=(0, 1))
    for q in range(9):
        if TB_C[q, 1] == 1:
            fn[q][:, 0] = fs[TB_OPP[q]][:, 0]
        if TB_C[q, 1] == -1:
            fn[q][:, ny - 1] = fs[TB_OPP[q]][:, ny - 1] + 6 * TB_W[q] * (TB_C[q, 0] * u_top)
    return fn


def to_textbook_order(lat):
    """index of the oracle's direction for every textbook direction"""
    return [int(np.flatnonzero((lat.c.T == c).all(axis=1))[0]) for c in TB_C]


def channel_bcs(shape, u_top):
    box = orc.bounding_box_indices(shape)
    bot = [list(box["bottom"][i]) for i in range(2)]
    top = [list(box["top"][i]) for i in range(2)]
    return [orc.BC(orc.KIND_HALFWAY_BB, 1, bot), orc.BC(orc.KIND_HALFWAY_BB, 2, top, u_wall=(u_top, 0.0))], bot, top


@pytest.mark.parametrize("omega", [1.0, 1.6])
def test_taylor_green_decay_gives_the_bgk_viscosity(omega):
    lat = orc.Lattice("D2Q9")
    n, policy = 32, "FP64FP64"
    f, k = taylor_green_init(n, 0.01, lat, policy)
    bm, mm = np.zeros((1, n, n), np.uint8), np.zeros((lat.q, n, n), bool)
    f = orc.run(f, bm, mm, [], omega, lat, 100, policy)
    a1 = amplitude(f, lat)
    f = orc.run(f, bm, mm, [], omega, lat, 200, policy)
    a2 = amplitude(f, lat)
    nu = np.log(a1 / a2) / (2.0 * k * k * 200)
    assert abs(nu / ((1.0 / omega - 0.5) / 3.0) - 1.0) < 0.01


def taylor_green_3d(n, nz, u0, lat, policy):
    """the same 2-D vortex, uniform along z, in a D3Q27 box"""
    T = orc.compute_dtype(policy)
    k = 2.0 * np.pi / n
    x, y, z = np.meshgrid(np.arange(n), np.arange(n), np.arange(nz), indexing="ij")
    ux = -u0 * np.cos(k * x) * np.sin(k * y)
    uy = u0 * np.sin(k * x) * np.cos(k * y)
    rho = 1.0 - 0.75 * u0 * u0 * (np.cos(2 * k * x) + np.cos(2 * k * y))
    f = orc.equilibrium(rho[None].astype(T), np.stack([ux, uy, np.zeros_like(ux)]).astype(T), lat, T)
    return f.astype(orc.store_dtype(policy)), k


@pytest.mark.parametrize("lattice,omega", [("D2Q9", 1.2), ("D2Q9", 1.8), ("D3Q27", 1.5)])
def test_taylor_green_decay_with_kbc_gives_the_same_viscosity(lattice, omega):
    """KBC relaxes the shear part of the non-equilibrium with 2 beta = omega whatever the entropic stabiliser gamma does to
    the rest (kbc.py:58-94: f - beta (2 ds + gamma dh)), so the vortex must decay with the BGK viscosity.  A wrong entry in
    the hard-coded shear tables (kbc.py:120-143, :163-172) or a wrong 1/4, 1/6 factor changes the decay rate."""
    lat = orc.Lattice(lattice)
    policy, n = "FP64FP64", 24
    if lattice == "D2Q9":
        f, k = taylor_green_init(n, 0.01, lat, policy)
        shape = (n, n)
    else:
        f, k = taylor_green_3d(n, 3, 0.01, lat, policy)
        shape = (n, n, 3)
    bm, mm = np.zeros((1,) + shape, np.uint8), np.zeros((lat.q,) + shape, bool)
    f = orc.run(f, bm, mm, [], omega, lat, 60, policy, "KBC")
    a1 = amplitude(f, lat)
    f = orc.run(f, bm, mm, [], omega, lat, 120, policy, "KBC")
    a2 = amplitude(f, lat)
    nu = np.log(a1 / a2) / (2.0 * k * k * 120)
    assert abs(nu / ((1.0 / omega - 0.5) / 3.0) - 1.0) < 0.02


def test_textbook_halfway_bounce_back_gives_the_exact_couette_profile():
    """Theory: between two halfway bounce-back walls the steady Couette profile is linear through walls that sit HALF A CELL
    outside the boundary nodes, for any relaxation rate — the textbook step reproduces it to rounding."""
    nx, ny, U = 4, 16, 0.01
    for omega in (0.8, 1.2, 1.7):
        f = np.tile(TB_W[:, None, None], (1, nx, ny))
        for _ in range(int(6 * ny * ny / ((1 / omega - 0.5) / 3)) // 4 + 2000):
            f = textbook_stream_bounce(textbook_collide(f, omega), U)
        ux = (np.einsum("qd,qxy->dxy", TB_C, f) / f.sum(0))[0].mean(axis=0)
        assert np.abs(ux / (U * (np.arange(ny) + 0.5) / ny) - 1.0).max() < 1e-6  # (a wall ON the nodes would be off by 3 % at the first one)


def test_oracle_halfway_bounce_back_is_the_textbook_rule():
    """The oracle's HalfwayBounceBackBC (bc_halfway_bounce_back.py:116-134 restated) + step order (nse_stepper.py:237-282) IS
    that textbook rule: one step from an arbitrary state agrees to rounding in every column that does not touch an x face.
    (On the x faces the reference's masker marks the directions leaving the box as missing as well — App. B.2: the padding
    is True — so tagged cells there bounce in x too: the reference's semantics, not the textbook channel.)"""
    lat = orc.Lattice("D2Q9")
    shape, U, omega = (6, 12), 0.013, 1.3
    bcs, _, _ = channel_bcs(shape, U)
    bm, mm = orc.build_masks(shape, lat, bcs)
    P = orc.perturbed_init(shape, lat, "FP64FP64", seed=3, amp_u=0.03)  # a post-collision state
    out = orc.step(P, bm, mm, bcs, omega, lat, "FP64FP64")
    order = to_textbook_order(lat)
    exp = textbook_collide(textbook_stream_bounce(P[order], U), omega)
    assert np.abs(out[order][:, 1:-1] - exp[:, 1:-1]).max() < 1e-15
    assert np.abs(out[order][:, 0] - exp[:, 0]).max() > 1e-4  # the x-face columns differ, as explained


@pytest.mark.gpu
@pytest.mark.parametrize("omega", [1.0, 1.6])
def test_taylor_green_decay_on_the_hip_backend(omega):
    from xlb_amd.grid import grid_factory
    from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper
    from _util import init_hip

    vs, pp = init_hip("D2Q9", "FP64FP64")
    lat = orc.Lattice("D2Q9")
    n = 32
    f_np, k = taylor_green_init(n, 0.01, lat, "FP64FP64")
    stepper = IncompressibleNavierStokesStepper(grid=grid_factory((n, n)), boundary_conditions=[])
    f_0, f_1, bm, mm = stepper.prepare_fields()
    f_0.assign(f_np)
    f_0, f_1 = stepper.run(f_0, f_1, bm, mm, omega, 100)
    a1 = amplitude(f_0.numpy(), lat)
    f_0, f_1 = stepper.run(f_0, f_1, bm, mm, omega, 200)
    a2 = amplitude(f_0.numpy(), lat)
    nu = np.log(a1 / a2) / (2.0 * k * k * 200)
    assert abs(nu / ((1.0 / omega - 0.5) / 3.0) - 1.0) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("policy", ["FP64FP64", "FP64FP32"])
def test_taylor_green_decay_with_kbc_on_the_hip_backend(policy):
    """D3Q27 KBC through the step kernel — the default fast fp64 collision included — decays with the BGK viscosity."""
    from xlb_amd.grid import grid_factory
    from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper
    from _util import init_hip

    vs, pp = init_hip("D3Q27", policy)
    lat = orc.Lattice("D3Q27")
    n, omega = 24, 1.5
    f_np, k = taylor_green_3d(n, 4, 0.01, lat, policy)
    stepper = IncompressibleNavierStokesStepper(grid=grid_factory((n, n, 4)), boundary_conditions=[], collision_type="KBC")
    f_0, f_1, bm, mm = stepper.prepare_fields()
    f_0.assign(f_np)
    f_0, f_1 = stepper.run(f_0, f_1, bm, mm, omega, 60)
    a1 = amplitude(f_0.numpy(), lat)
    f_0, f_1 = stepper.run(f_0, f_1, bm, mm, omega, 120)
    a2 = amplitude(f_0.numpy(), lat)
    nu = np.log(a1 / a2) / (2.0 * k * k * 120)
    assert abs(nu / ((1.0 / omega - 0.5) / 3.0) - 1.0) < 0.02


@pytest.mark.gpu
def test_hip_halfway_bounce_back_is_the_textbook_rule():
    from xlb_amd.grid import grid_factory
    from xlb_amd.operator.boundary_condition import HalfwayBounceBackBC
    from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper
    from _util import init_hip

    vs, pp = init_hip("D2Q9", "FP64FP64")
    lat = orc.Lattice("D2Q9")
    shape, U, omega = (6, 12), 0.013, 1.3
    _, bot, top = channel_bcs(shape, U)
    bcs = [HalfwayBounceBackBC(indices=bot), HalfwayBounceBackBC(indices=top, prescribed_value=(U, 0.0))]
    stepper = IncompressibleNavierStokesStepper(grid=grid_factory(shape), boundary_conditions=bcs)
    f_0, f_1, bm, mm = stepper.prepare_fields()
    P = orc.perturbed_init(shape, lat, "FP64FP64", seed=3, amp_u=0.03)
    f_0.assign(P)
    f_0, f_1 = stepper(f_0, f_1, bm, mm, omega, 0)
    order = to_textbook_order(lat)
    exp = textbook_collide(textbook_stream_bounce(P[order], U), omega)
    assert np.abs(f_1.numpy()[order][:, 1:-1] - exp[:, 1:-1]).max() < 1e-15
