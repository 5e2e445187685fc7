"""Zou-He / Regularized inlet-outlet BCs on the HIP backend vs the oracle.
No reference test pins these BCs ("parity unpinned by the reference"): the oracle follows
bc_zouhe.py:166-304 and bc_regularized.py:78-137 line by line."""

import numpy as np
import pytest

from oracle import xlb_numpy as orc
from xlb_amd.grid import grid_factory
from xlb_amd.helper import create_nse_fields
from xlb_amd.operator.boundary_condition import HalfwayBounceBackBC, RegularizedBC, ZouHeBC
from xlb_amd.operator.boundary_masker import IndicesBoundaryMasker
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

from _util import init_hip, max_ulp_diff

pytestmark = pytest.mark.gpu

CASES = [("D2Q9", (24, 12), "FP32FP32", "BGK"), ("D3Q19", (16, 8, 12), "FP32FP32", "BGK"), ("D3Q27", (10, 6, 8), "FP32FP32", "KBC"),
         ("D3Q19", (8, 6, 8), "FP64FP64", "BGK"), ("D2Q9", (16, 10), "FP64FP32", "KBC")]
NAMES = {(ZouHeBC, "velocity"): orc.KIND_ZOUHE_VELOCITY, (ZouHeBC, "pressure"): orc.KIND_ZOUHE_PRESSURE,
         (RegularizedBC, "velocity"): orc.KIND_REGULARIZED_VELOCITY, (RegularizedBC, "pressure"): orc.KIND_REGULARIZED_PRESSURE}


def channel(lattice, shape, policy, cls):
    vs, pp = init_hip(lattice, policy)
    lat = orc.Lattice(lattice)
    d = lat.d
    grid = grid_factory(shape)
    box = grid.bounding_box_indices()
    box_ne = grid.bounding_box_indices(remove_edges=True)
    faces = ["bottom", "top"] + (["front", "back"] if d == 3 else [])
    walls = [sum((box[f][i] for f in faces), []) for i in range(d)]
    walls = np.unique(np.array(walls), axis=-1).tolist()
    u_in = (0.03,) + (0.0,) * (d - 1)
    b_in = cls("velocity", prescribed_value=u_in, indices=box_ne["left"])
    b_out = cls("pressure", prescribed_value=1.0, indices=box_ne["right"])
    b_w = HalfwayBounceBackBC(indices=walls)
    obcs = [orc.BC(NAMES[(cls, "velocity")], b_in.id, box_ne["left"], prescribed=u_in),
            orc.BC(NAMES[(cls, "pressure")], b_out.id, box_ne["right"], prescribed=1.0), orc.BC(orc.KIND_HALFWAY_BB, b_w.id, walls)]
    return grid, [b_in, b_out, b_w], lat, obcs


@pytest.mark.parametrize("cls", [ZouHeBC, RegularizedBC])
@pytest.mark.parametrize("lattice,shape,policy,collision", CASES)
def test_channel_flow_vs_oracle(lattice, shape, policy, collision, cls):
    grid, bcs, lat, obcs = channel(lattice, shape, policy, cls)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs, collision_type=collision)
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    o_bm, o_mm = orc.build_masks(shape, lat, obcs)
    assert np.array_equal(bc_mask.numpy(), o_bm) and np.array_equal(missing_mask.numpy(), o_mm.astype(np.uint8))
    steps = 25
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.3, steps)
    with np.errstate(all="ignore"):
        exp = orc.run(orc.initialize_eq(shape, lat, policy), o_bm, o_mm, obcs, 1.3, lat, steps, policy, collision)
    out = f_0.numpy()
    tol = 1e-6 if policy != "FP64FP64" else 1e-12
    assert np.abs(out.astype(np.float64) - exp.astype(np.float64)).max() <= tol
    assert np.array_equal(out, exp), f"not bit-exact: max ulp {max_ulp_diff(out, exp)}"
    rho, u = orc.macroscopic(out.astype(orc.compute_dtype(policy)), lat)
    assert float(u[0].max()) > 0.02  # the inlet drives a flow


@pytest.mark.parametrize("cls", [ZouHeBC, RegularizedBC])
@pytest.mark.parametrize("lattice,shape", [("D2Q9", (20, 14)), ("D3Q19", (10, 8, 6))])
def test_standalone_operator_vs_oracle(lattice, shape, cls):
    """bc(f_pre, f_post, bc_mask, missing_mask) as a stand-alone operator on random populations."""
    vs, pp = init_hip(lattice)
    lat = orc.Lattice(lattice)
    d = lat.d
    rng = np.random.default_rng(6)
    a_post = (lat.w.astype(np.float32).reshape((-1,) + (1,) * d) * (1.0 + 0.1 * rng.random((lat.q,) + shape, dtype=np.float32))).astype(np.float32)
    for bc_type, value in (("velocity", (0.02,) + (0.0,) * (d - 1)), ("pressure", 1.02)):
        grid, f_0, f_1, missing_mask, bc_mask = create_nse_fields(shape)
        idx = grid.bounding_box_indices(remove_edges=True)["left"]
        bc = cls(bc_type, prescribed_value=value, indices=idx)
        bc_mask, missing_mask = IndicesBoundaryMasker(grid=grid)([bc], bc_mask, missing_mask)
        f_pre = grid.create_field(vs.q)
        f_post = grid.create_field(vs.q).assign(a_post)
        out = bc(f_pre, f_post, bc_mask, missing_mask).numpy()
        obc = orc.BC(NAMES[(cls, bc_type)], bc.id, idx, prescribed=value)
        o_bm, o_mm = orc.build_masks(shape, lat, [obc])
        with np.errstate(all="ignore"):
            exp = orc.apply_bc(obc, np.zeros_like(a_post), a_post, o_bm, o_mm, lat, "FP32FP32")
        assert np.array_equal(out, exp), (cls.__name__, bc_type, max_ulp_diff(out, exp))


def test_argument_validation():
    init_hip("D3Q19")
    with pytest.raises(AssertionError):
        ZouHeBC("temperature", prescribed_value=1.0, indices=[[0], [0], [0]])
    with pytest.raises(ValueError):
        ZouHeBC("velocity", prescribed_value=(0.1, 0.1, 0.0), indices=[[0], [0], [0]])  # only normal values
    with pytest.raises(ValueError):
        ZouHeBC("pressure", prescribed_value=(1.0, 0.0, 0.0), indices=[[0], [0], [0]])
    with pytest.raises(ValueError, match="callable"):
        RegularizedBC("velocity", profile=np.zeros((3, 1)), indices=[[0], [0], [0]])


# ---- ExtrapolationOutflowBC (bc_extrapolation_outflow.py, JAX semantics) --------------------------------------
from xlb_amd.operator.boundary_condition import ExtrapolationOutflowBC, FullwayBounceBackBC  # noqa: E402

OUTFLOW_CASES = [("D2Q9", (24, 12), "FP32FP32", "BGK"), ("D3Q19", (16, 8, 12), "FP32FP32", "BGK"), ("D3Q27", (10, 6, 8), "FP32FP32", "KBC"),
                 ("D3Q19", (8, 6, 8), "FP64FP64", "BGK"), ("D3Q19", (12, 6, 8), "FP32FP16", "BGK")]


def outflow_channel(lattice, shape, policy, walls_cls, inlet_face, outlet_face):
    """velocity inlet on one face, extrapolation outflow on the opposite one, walls on the other faces"""
    vs, pp = init_hip(lattice, policy)
    lat = orc.Lattice(lattice)
    d = lat.d
    grid = grid_factory(shape)
    box = grid.bounding_box_indices()
    box_ne = grid.bounding_box_indices(remove_edges=True)
    all_faces = ["left", "right", "bottom", "top"] + (["front", "back"] if d == 3 else [])
    faces = [f for f in all_faces if f not in (inlet_face, outlet_face)]
    walls = [sum((box[f][i] for f in faces), []) for i in range(d)]
    walls = np.unique(np.array(walls), axis=-1).tolist()
    axis = {"left": 0, "right": 0, "front": 1, "back": 1, "bottom": d - 1, "top": d - 1}[inlet_face]
    sign = 1.0 if inlet_face in ("left", "front", "bottom") else -1.0
    u_in = tuple(sign * 0.03 if a == axis else 0.0 for a in range(d))
    b_w = walls_cls(indices=walls)
    b_in = RegularizedBC("velocity", prescribed_value=u_in, indices=box_ne[inlet_face])
    b_out = ExtrapolationOutflowBC(indices=box_ne[outlet_face])
    wkind = orc.KIND_FULLWAY_BB if walls_cls is FullwayBounceBackBC else orc.KIND_HALFWAY_BB
    obcs = [orc.BC(wkind, b_w.id, walls), orc.BC(orc.KIND_REGULARIZED_VELOCITY, b_in.id, box_ne[inlet_face], prescribed=u_in),
            orc.BC(orc.KIND_EXTRAPOLATION_OUTFLOW, b_out.id, box_ne[outlet_face])]
    assert np.array_equal(b_out.normal, obcs[2].normal)
    return grid, [b_w, b_in, b_out], lat, obcs


@pytest.mark.parametrize("walls_cls", [FullwayBounceBackBC, HalfwayBounceBackBC])
@pytest.mark.parametrize("lattice,shape,policy,collision", OUTFLOW_CASES)
def test_extrapolation_outflow_vs_oracle(lattice, shape, policy, collision, walls_cls):
    """The order [walls, inlet, outlet] of examples/cfd/flow_past_sphere_3d.py:108-112.  Bit-exact against the oracle:
    the streaming part lives in the step kernel, the auxiliary data in k_outflow_aux (both per-cell code, the
    oracle is roll-based like the reference).  No reference test pins this BC."""
    grid, bcs, lat, obcs = outflow_channel(lattice, shape, policy, walls_cls, "left", "right")
    assert bcs[2].normal.tolist() == [1] + [0] * (lat.d - 1)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs, collision_type=collision)
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    o_bm, o_mm = orc.build_masks(shape, lat, obcs)
    assert np.array_equal(bc_mask.numpy(), o_bm) and np.array_equal(missing_mask.numpy(), o_mm.astype(np.uint8))
    f_np = orc.perturbed_init(shape, lat, policy, seed=41)
    f_0.assign(f_np)
    for steps in (1, 24):
        f_0.assign(f_np)
        a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.4, steps)
        with np.errstate(all="ignore"):
            exp = orc.run(f_np, o_bm, o_mm, obcs, 1.4, lat, steps, policy, collision)
        out = a.numpy()
        assert np.array_equal(out, exp), f"{steps} steps: max ulp {max_ulp_diff(out, exp)}, max abs {np.abs(out.astype(np.float64) - exp).max()}"
        f_0, f_1 = (a, b) if a is f_0 else (b, a)


@pytest.mark.parametrize("inlet_face,outlet_face,normal", [("right", "left", [-1, 0, 0]), ("bottom", "top", [0, 0, 1]), ("back", "front", [0, -1, 0])])
def test_extrapolation_outflow_other_faces(inlet_face, outlet_face, normal):
    grid, bcs, lat, obcs = outflow_channel("D3Q19", (10, 8, 12), "FP32FP32", HalfwayBounceBackBC, inlet_face, outlet_face)
    assert bcs[2].normal.tolist() == normal
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    o_bm, o_mm = orc.build_masks((10, 8, 12), lat, obcs)
    f_np = orc.perturbed_init((10, 8, 12), lat, seed=43)
    f_0.assign(f_np)
    a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.7, 15)
    exp = orc.run(f_np, o_bm, o_mm, obcs, 1.7, lat, 15)
    assert np.array_equal(a.numpy(), exp)


def test_extrapolation_outflow_operator_call():
    """bc(f_pre, f_post, bc_mask, missing_mask): the streaming-step form (bc_extrapolation_outflow.py:137-145)."""
    grid, bcs, lat, obcs = outflow_channel("D3Q19", (8, 6, 8), "FP32FP32", HalfwayBounceBackBC, "left", "right")
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    rng = np.random.default_rng(3)
    pre = rng.random((19, 8, 6, 8)).astype(np.float32)
    post = rng.random((19, 8, 6, 8)).astype(np.float32)
    f_0.assign(pre)
    f_1.assign(post)
    out = bcs[2](f_0, f_1, bc_mask, missing_mask)
    o_bm, o_mm = orc.build_masks((8, 6, 8), lat, obcs)
    exp = orc.apply_bc(obcs[2], pre, post, o_bm, o_mm, lat, "FP32FP32")
    assert np.array_equal(out.numpy(), exp)
    assert not np.array_equal(exp, post)


# ---- callable profiles: per-cell prescribed values (bc_zouhe.py:122-124, :179-232) ---------------------------------
def parabolic_inlet(shape, u_max=0.04):
    """examples/cfd/flow_past_sphere_3d.py:64-81 in NumPy: u_x(y, z) parabolic, zero at the walls -> (3, ny, nz)"""
    ny, nz = shape[1], shape[2]
    y, z = np.meshgrid(np.arange(ny), np.arange(nz), indexing="ij")
    hy, hz = ny - 1.0, nz - 1.0
    r2 = (2.0 * (y - hy / 2.0) / hy) ** 2 + (2.0 * (z - hz / 2.0) / hz) ** 2
    ux = u_max * np.maximum(0.0, 1.0 - r2)
    return np.stack([ux, np.zeros_like(ux), np.zeros_like(ux)])


@pytest.mark.parametrize("cls,policy", [(RegularizedBC, "FP32FP32"), (ZouHeBC, "FP32FP32"), (RegularizedBC, "FP64FP64")])
def test_flow_past_sphere_setup_vs_oracle(cls, policy):
    """The boundary-condition set of examples/cfd/flow_past_sphere_3d.py:104-112: fullway walls, an inlet with a parabolic
    PROFILE, extrapolation outflow and a halfway sphere in the interior (indices with padding) — against the oracle,
    bit for bit.  The pressure-profile variant runs on the outlet of a second stepper."""
    shape = (28, 14, 14)
    vs, pp = init_hip("D3Q19", policy)
    lat = orc.Lattice("D3Q19")
    grid = grid_factory(shape)
    box = grid.bounding_box_indices()
    box_ne = grid.bounding_box_indices(remove_edges=True)
    walls = [sum((box[f][i] for f in ("bottom", "top", "front", "back")), []) for i in range(3)]
    walls = np.unique(np.array(walls), axis=-1).tolist()
    x, y, z = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    sphere = np.where((x - 9) ** 2 + (y - 7) ** 2 + (z - 7) ** 2 < 3.2**2)
    sphere = [s.tolist() for s in sphere]
    prof = parabolic_inlet(shape)
    b_w = FullwayBounceBackBC(indices=walls)
    b_in = cls("velocity", profile=lambda: prof, indices=box_ne["left"])
    b_out = ExtrapolationOutflowBC(indices=box_ne["right"])
    b_s = HalfwayBounceBackBC(indices=sphere)
    bcs = [b_w, b_in, b_out, b_s]
    kind = orc.KIND_REGULARIZED_VELOCITY if cls is RegularizedBC else orc.KIND_ZOUHE_VELOCITY
    obcs = [orc.BC(orc.KIND_FULLWAY_BB, b_w.id, walls), orc.BC(kind, b_in.id, box_ne["left"], prescribed=prof),
            orc.BC(orc.KIND_EXTRAPOLATION_OUTFLOW, b_out.id, box_ne["right"]), orc.BC(orc.KIND_HALFWAY_BB, b_s.id, sphere)]
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    o_bm, o_mm = orc.build_masks(shape, lat, obcs)
    assert np.array_equal(bc_mask.numpy(), o_bm) and np.array_equal(missing_mask.numpy(), o_mm.astype(np.uint8))
    steps = 30
    a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.5, steps)
    with np.errstate(all="ignore"):
        exp = orc.run(orc.initialize_eq(shape, lat, policy), o_bm, o_mm, obcs, 1.5, lat, steps, policy)
    out = a.numpy()
    assert np.array_equal(out, exp), f"max ulp {max_ulp_diff(out, exp)}, max abs {np.abs(out.astype(np.float64) - exp).max()}"
    rho, u = orc.macroscopic(out.astype(orc.compute_dtype(policy)), lat)
    assert u[0, 1, 7, 7] > 0.03 and abs(u[0, 1, 1, 1]) < 0.01  # the profile arrived: fast in the middle, slow near the walls
    # stand-alone operator call with per-cell values
    rng = np.random.default_rng(9)
    T = orc.compute_dtype(policy)
    post = (lat.w[:, None, None, None] * (1 + 0.05 * rng.standard_normal((19,) + shape))).astype(T)
    f_1.assign(post)
    got = b_in(f_0, f_1, bc_mask, missing_mask)
    with np.errstate(all="ignore"):
        e = orc.apply_bc(obcs[1], post, post, o_bm, o_mm, lat, policy)
    assert np.array_equal(got.numpy(), e.astype(got.numpy().dtype))


def test_pressure_profile_and_errors():
    shape = (16, 8, 10)
    vs, pp = init_hip("D3Q19")
    lat = orc.Lattice("D3Q19")
    grid = grid_factory(shape)
    box_ne = grid.bounding_box_indices(remove_edges=True)
    rho_face = 1.0 + 0.01 * np.random.default_rng(2).random((1, shape[1], shape[2]))
    b_in = ZouHeBC("velocity", prescribed_value=(0.02, 0.0, 0.0), indices=box_ne["left"])
    b_out = ZouHeBC("pressure", profile=lambda: rho_face, indices=box_ne["right"])
    obcs = [orc.BC(orc.KIND_ZOUHE_VELOCITY, b_in.id, box_ne["left"], prescribed=(0.02, 0.0, 0.0)),
            orc.BC(orc.KIND_ZOUHE_PRESSURE, b_out.id, box_ne["right"], prescribed=rho_face)]
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b_in, b_out])
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    o_bm, o_mm = orc.build_masks(shape, lat, obcs)
    a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.2, 12)
    with np.errstate(all="ignore"):
        exp = orc.run(orc.initialize_eq(shape, lat), o_bm, o_mm, obcs, 1.2, lat, 12)
    assert np.array_equal(a.numpy(), exp)
    with pytest.raises(ValueError, match="both profile and prescribed_value"):
        ZouHeBC("velocity", profile=lambda: rho_face, prescribed_value=(0.1, 0, 0), indices=box_ne["left"])
    with pytest.raises(ValueError, match="first axis"):
        ZouHeBC("velocity", profile=lambda: rho_face, indices=box_ne["left"])


def test_flow_past_sphere_vs_golden():
    """HIP backend against the committed vectors of tests/golden/d3q19_sphere_channel.npz: populations bit-exact, vorticity
    magnitude / Q bit-exact, force to rounding."""
    from _util import golden
    from xlb_amd.operator.force import MomentumTransfer
    from xlb_amd.operator.macroscopic import Macroscopic
    from xlb_amd.operator.postprocess import QCriterion, Vorticity
    from xlb_amd.precision_policy import Precision

    g = golden("d3q19_sphere_channel")
    shape = (28, 14, 14)
    vs, pp = init_hip("D3Q19")
    lat, obcs, prof = orc.sphere_channel(shape)
    grid = grid_factory(shape)
    b_w = FullwayBounceBackBC(indices=obcs[0].indices.tolist())
    b_in = RegularizedBC("velocity", profile=lambda: prof, indices=obcs[1].indices.tolist())
    b_out = ExtrapolationOutflowBC(indices=obcs[2].indices.tolist())
    b_s = HalfwayBounceBackBC(indices=obcs[3].indices.tolist())
    assert [b.id for b in (b_w, b_in, b_out, b_s)] == [1, 2, 3, 4]
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b_w, b_in, b_out, b_s])
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    assert np.array_equal(bc_mask.numpy(), g["bc_mask"])
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, float(g["omega"]), int(g["steps"]))
    assert np.array_equal(f_0.numpy(), g["f"])
    rho, u = grid.create_field(1, dtype=Precision.FP32), grid.create_field(3, dtype=Precision.FP32)
    Macroscopic()(f_0, rho, u)
    _, mag = Vorticity()(u, bc_mask, grid.create_field(3, dtype=Precision.FP32), grid.create_field(1, dtype=Precision.FP32))
    _, q = QCriterion()(u, bc_mask, grid.create_field(1, dtype=Precision.FP32), grid.create_field(1, dtype=Precision.FP32))
    assert np.array_equal(mag.numpy(), g["vorticity_magnitude"]) and np.array_equal(q.numpy(), g["q"])
    force = MomentumTransfer(b_s)(f_0, f_1, bc_mask, missing_mask)
    assert np.allclose(force, g["force"], rtol=1e-5, atol=1e-5 * np.abs(g["force"]).max())


@pytest.mark.parametrize("steps", [2, 5, 8])
@pytest.mark.parametrize("outlet", ["outflow", "pressure"])
def test_two_step_kernel_with_inlet_outlet_planes(outlet, steps):
    """Steppers whose Zou-He / Regularized / outflow cells all sit in the planes x = 0 and x = nx-1 still use the two-step
    kernel for the planes 2 .. nx-3; the four end planes go through the single-step kernel twice (api.hip:
    step_twice_edge_ext).  Same bits as the oracle — i.e. as the single-step path — for every step-count parity."""
    from xlb_amd.default_config import get_context

    shape = (24, 16, 64)
    vs, pp = init_hip("D3Q19")
    lat = orc.Lattice("D3Q19")
    grid = grid_factory(shape)
    box = grid.bounding_box_indices()
    box_ne = grid.bounding_box_indices(remove_edges=True)
    walls = [sum((box[f][i] for f in ("bottom", "top", "front", "back")), []) for i in range(3)]
    walls = np.unique(np.array(walls), axis=-1).tolist()
    x, y, z = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    sphere = [s.tolist() for s in np.where((x - 8) ** 2 + (y - 8) ** 2 + (z - 30) ** 2 < 3.3**2)]
    prof = parabolic_inlet(shape)
    b_w = HalfwayBounceBackBC(indices=walls)
    b_in = RegularizedBC("velocity", profile=lambda: prof, indices=box_ne["left"])
    if outlet == "outflow":
        b_out = ExtrapolationOutflowBC(indices=box_ne["right"])
        o_out = orc.BC(orc.KIND_EXTRAPOLATION_OUTFLOW, b_out.id, box_ne["right"])
    else:
        b_out = ZouHeBC("pressure", prescribed_value=1.0, indices=box_ne["right"])
        o_out = orc.BC(orc.KIND_ZOUHE_PRESSURE, b_out.id, box_ne["right"], prescribed=1.0)
    b_s = HalfwayBounceBackBC(indices=sphere)
    obcs = [orc.BC(orc.KIND_HALFWAY_BB, b_w.id, walls), orc.BC(orc.KIND_REGULARIZED_VELOCITY, b_in.id, box_ne["left"], prescribed=prof), o_out,
            orc.BC(orc.KIND_HALFWAY_BB, b_s.id, sphere)]
    ctx = get_context()
    try:
        ctx.set_option("fuse2", 2)
        stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b_w, b_in, b_out, b_s])
        f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
        assert stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
        o_bm, o_mm = orc.build_masks(shape, lat, obcs)
        f_np = orc.perturbed_init(shape, lat, seed=51)
        f_0.assign(f_np)
        a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.5, steps)
        with np.errstate(all="ignore"):
            exp = orc.run(f_np, o_bm, o_mm, obcs, 1.5, lat, steps)
        out = a.numpy()
        assert np.array_equal(out, exp), f"max ulp {max_ulp_diff(out, exp)}, max abs {np.abs(out - exp).max()}"
    finally:
        ctx.set_option("fuse2", 1)


def test_two_step_kernel_refuses_interior_inlets():
    """an extended BC away from the end planes keeps the stepper on the single-step kernel"""
    from xlb_amd.default_config import get_context

    shape = (24, 16, 64)
    vs, pp = init_hip("D3Q19")
    grid = grid_factory(shape)
    plane3 = [[3] * 4, [4, 5, 6, 7], [10, 11, 12, 13]]
    b = ZouHeBC("velocity", prescribed_value=(0.01, 0.0, 0.0), indices=plane3)
    ctx = get_context()
    try:
        ctx.set_option("fuse2", 2)
        stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b])
        f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
        assert not stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
    finally:
        ctx.set_option("fuse2", 1)


@pytest.mark.parametrize("steps", [2, 7])
def test_two_step_kernel_with_do_nothing_outlet(steps):
    """DoNothingBC on the outlet plane: same route as the Zou-He family (end planes through the single-step kernel)."""
    from xlb_amd.default_config import get_context
    from xlb_amd.operator.boundary_condition import DoNothingBC, EquilibriumBC

    shape = (20, 16, 64)
    vs, pp = init_hip("D3Q19")
    lat = orc.Lattice("D3Q19")
    grid = grid_factory(shape)
    box = grid.bounding_box_indices()
    box_ne = grid.bounding_box_indices(remove_edges=True)
    walls = [sum((box[f][i] for f in ("bottom", "top", "front", "back")), []) for i in range(3)]
    walls = np.unique(np.array(walls), axis=-1).tolist()
    b_in = EquilibriumBC(rho=1.0, u=(0.03, 0.0, 0.0), indices=box_ne["left"])
    b_out = DoNothingBC(indices=box_ne["right"])
    b_w = HalfwayBounceBackBC(indices=walls)
    obcs = [orc.BC(orc.KIND_EQUILIBRIUM, b_in.id, box_ne["left"], rho=1.0, u=(0.03, 0.0, 0.0)), orc.BC(orc.KIND_DO_NOTHING, b_out.id, box_ne["right"]),
            orc.BC(orc.KIND_HALFWAY_BB, b_w.id, walls)]
    ctx = get_context()
    try:
        ctx.set_option("fuse2", 2)
        stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b_in, b_out, b_w])
        f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
        assert stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
        o_bm, o_mm = orc.build_masks(shape, lat, obcs)
        f_np = orc.perturbed_init(shape, lat, seed=53)
        f_0.assign(f_np)
        a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.3, steps)
        exp = orc.run(f_np, o_bm, o_mm, obcs, 1.3, lat, steps)
        assert np.array_equal(a.numpy(), exp)
    finally:
        ctx.set_option("fuse2", 1)
