"""The reference's calling protocol — `f_0, f_1 = stepper(f_0, f_1, ...); f_0, f_1 = f_1, f_0` (mlups_3d.py:237-238, every
XLB example) — reaches the two-steps-per-pass kernel through deferred pairing (operator/stepper/nse_stepper.py).  Whatever
the driver does in between, the fields it sees are bit-identical to single steps / the oracle."""

import numpy as np
import pytest

from oracle import xlb_numpy as orc
from xlb_amd.default_config import get_context
from xlb_amd.operator.boundary_condition import FullwayBounceBackBC, HalfwayBounceBackBC
from xlb_amd.operator.macroscopic import Macroscopic
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

from _util import hip_cavity_3d

pytestmark = pytest.mark.gpu
SHAPE = (12, 16, 64)


def setup(walls_cls=HalfwayBounceBackBC, **cfg):
    grid, bcs, lat, obcs = hip_cavity_3d(SHAPE, walls_cls)
    get_context().set_option("fuse2", 2)  # (the chip-filling rule would refuse a domain this small)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs, backend_config=cfg)
    fields = stepper.prepare_fields()
    f_np = orc.perturbed_init(SHAPE, lat, seed=51)
    fields[0].assign(f_np)
    o_bm, o_mm = orc.build_masks(SHAPE, lat, obcs)
    return stepper, fields, lat, obcs, f_np, o_bm, o_mm


@pytest.fixture(autouse=True)
def _restore_options():
    yield
    get_context().set_option("fuse2", 1)


@pytest.mark.parametrize("walls_cls", [HalfwayBounceBackBC, FullwayBounceBackBC])
@pytest.mark.parametrize("n", [1, 2, 3, 6, 7])
def test_reference_loop_matches_oracle(walls_cls, n):
    stepper, (f_0, f_1, bm, mm), lat, obcs, f_np, o_bm, o_mm = setup(walls_cls)
    for i in range(n):
        f_0, f_1 = stepper(f_0, f_1, bm, mm, 1.4, i)
        f_0, f_1 = f_1, f_0
    assert np.array_equal(f_0.numpy(), orc.run(f_np, o_bm, o_mm, obcs, 1.4, lat, n))
    # the steady loop is pairs only: no step ran alone except a trailing odd one, and nothing was materialised (that path
    # allocates a whole temporary field: a loop that hit it every pair once ran 50x slower)
    assert stepper._n_fused_pairs == n // 2 and stepper._n_materialised == 0


def test_pairs_really_fuse_and_previous_state_is_materialised():
    stepper, (f_0, f_1, bm, mm), lat, obcs, f_np, o_bm, o_mm = setup()
    a, b = f_0, f_1
    ptr = lambda f: f._h.value  # noqa: E731 (raw: .handle / .info() would count as a use and flush)
    pa, pb = ptr(a), ptr(b)
    stepper(a, b, bm, mm, 1.2, 0)
    assert stepper._deferred is not None  # nothing enqueued yet
    stepper(b, a, bm, mm, 1.2, 1)
    assert stepper._deferred is None and (ptr(a), ptr(b)) == (pb, pa)  # one fused pass, buffers exchanged
    exp1 = orc.run(f_np, o_bm, o_mm, obcs, 1.2, lat, 1)
    exp2 = orc.run(f_np, o_bm, o_mm, obcs, 1.2, lat, 2)
    assert np.array_equal(a.numpy(), exp2)  # f(t+2) where the protocol puts it
    assert np.array_equal(b.numpy(), exp1)  # f(t+1): materialised on demand through a temporary field
    assert np.array_equal(a.numpy(), exp2)
    # and the loop goes on from there
    stepper(a, b, bm, mm, 1.2, 2)
    assert np.array_equal(b.numpy(), orc.run(f_np, o_bm, o_mm, obcs, 1.2, lat, 3))


def test_any_other_use_flushes_the_deferred_step():
    stepper, (f_0, f_1, bm, mm), lat, obcs, f_np, o_bm, o_mm = setup()
    vs, pp = stepper.velocity_set, stepper.precision_policy
    exp = [orc.run(f_np, o_bm, o_mm, obcs, 1.3, lat, k) for k in range(6)]
    macro = Macroscopic()
    rho = stepper.grid.create_field(1, dtype=pp.compute_precision)
    u = stepper.grid.create_field(3, dtype=pp.compute_precision)
    for i in range(5):
        f_0, f_1 = stepper(f_0, f_1, bm, mm, 1.3, i)
        f_0, f_1 = f_1, f_0
        if i == 0:  # an operator on the newest field between the two calls of a would-be pair
            macro(f_0, rho, u)
            o_rho, o_u = orc.macroscopic(exp[1], lat)
            assert np.array_equal(rho.numpy(), o_rho) and np.array_equal(u.numpy(), o_u)
        if i == 2:  # a device sync (what drivers do before timing / output)
            get_context().sync()
            assert stepper._deferred is None
        if i == 3:  # the caller modifies the state
            tweak = f_0.numpy()
            assert np.array_equal(tweak, exp[4])
            f_0.assign(tweak)
    assert np.array_equal(f_0.numpy(), exp[5])


def test_changing_omega_or_not_swapping_breaks_the_pair_correctly():
    stepper, (f_0, f_1, bm, mm), lat, obcs, f_np, o_bm, o_mm = setup()
    f_0, f_1 = stepper(f_0, f_1, bm, mm, 1.1, 0)
    f_0, f_1 = f_1, f_0
    f_0, f_1 = stepper(f_0, f_1, bm, mm, 1.7, 1)  # different omega: two single steps
    f_0, f_1 = f_1, f_0
    e = orc.run(orc.run(f_np, o_bm, o_mm, obcs, 1.1, lat, 1), o_bm, o_mm, obcs, 1.7, lat, 1)
    assert np.array_equal(f_0.numpy(), e)
    # no swap: the same (source, destination) twice recomputes the same step
    f_0, f_1 = stepper(f_0, f_1, bm, mm, 1.7, 2)
    f_0, f_1 = stepper(f_0, f_1, bm, mm, 1.7, 3)
    assert np.array_equal(f_1.numpy(), orc.run(e, o_bm, o_mm, obcs, 1.7, lat, 1))


def test_lazy_pairs_can_be_switched_off_and_exported_fields_are_never_deferred():
    stepper, (f_0, f_1, bm, mm), lat, obcs, f_np, o_bm, o_mm = setup(lazy_pairs=False)
    stepper(f_0, f_1, bm, mm, 1.2, 0)
    assert stepper._deferred is None
    stepper2, (g_0, g_1, bm2, mm2), *_ = setup()
    _ = g_0.__cuda_array_interface__  # exported: its memory must stay put
    stepper2(g_0, g_1, bm2, mm2, 1.2, 0)
    assert stepper2._deferred is None
    assert np.array_equal(g_1.numpy(), orc.run(f_np, o_bm, o_mm, obcs, 1.2, lat, 1))


def test_editing_the_masks_flushes_the_deferred_step():
    """A deferred step was issued with the masks as they were: touching bc_mask afterwards must run it first."""
    stepper, (f_0, f_1, bm, mm), lat, obcs, f_np, o_bm, o_mm = setup()
    stepper(f_0, f_1, bm, mm, 1.2, 0)
    assert stepper._deferred is not None
    saved = bm.numpy()  # any access to the mask counts
    assert stepper._deferred is None and np.array_equal(saved, o_bm)
    assert np.array_equal(f_1.numpy(), orc.run(f_np, o_bm, o_mm, obcs, 1.2, lat, 1))


def test_failed_materialisation_keeps_the_step_owed(monkeypatch):
    """reading the virtual f(t+1) allocates a temporary third field.  If that allocation fails the hook
    must stay: the next reader raises again (or succeeds once memory is back) — it is never handed f(t) for f(t+1)."""
    stepper, (a, b, bm, mm), lat, obcs, f_np, o_bm, o_mm = setup()
    stepper(a, b, bm, mm, 1.2, 0)
    stepper(b, a, bm, mm, 1.2, 1)  # fused: b holds f(t+1) virtually
    assert stepper._n_fused_pairs == 1
    real = stepper.grid.create_field

    def no_memory(*args, **kw):
        raise MemoryError("hipMalloc failed (simulated)")

    monkeypatch.setattr(stepper.grid, "create_field", no_memory)
    with pytest.raises(MemoryError):
        b.numpy()
    with pytest.raises(MemoryError):  # still owed: not f(t)
        b.numpy()
    assert stepper._n_materialised == 0
    monkeypatch.setattr(stepper.grid, "create_field", real)
    assert np.array_equal(b.numpy(), orc.run(f_np, o_bm, o_mm, obcs, 1.2, lat, 1))
    assert np.array_equal(a.numpy(), orc.run(f_np, o_bm, o_mm, obcs, 1.2, lat, 2))


def test_run_after_a_pair_does_not_materialise_what_it_overwrites():
    """stepper.run(f_0, f_1, ...) overwrites f_1 with its first step; a virtual f(t+1) there is dropped, not
    materialised through a third field."""
    stepper, (a, b, bm, mm), lat, obcs, f_np, o_bm, o_mm = setup()
    stepper(a, b, bm, mm, 1.2, 0)
    stepper(b, a, bm, mm, 1.2, 1)  # a: f(t+2); b: virtual f(t+1)
    cur, oth = stepper.run(a, b, bm, mm, 1.2, 3)
    assert stepper._n_materialised == 0
    assert np.array_equal(cur.numpy(), orc.run(f_np, o_bm, o_mm, obcs, 1.2, lat, 5))
    # zero steps overwrite nothing: the virtual field stays readable
    stepper2, (c, d, bm2, mm2), *_ = setup()
    stepper2(c, d, bm2, mm2, 1.2, 0)
    stepper2(d, c, bm2, mm2, 1.2, 1)
    stepper2.run(c, d, bm2, mm2, 1.2, 0)
    assert np.array_equal(d.numpy(), orc.run(f_np, o_bm, o_mm, obcs, 1.2, lat, 1)) and stepper2._n_materialised == 1


def test_no_pairing_without_room_for_a_third_field(monkeypatch):
    """where the two population fields just fit, reference-style calls are not paired (single steps need no
    temporary), instead of failing later inside an innocent read."""
    stepper, (f_0, f_1, bm, mm), lat, obcs, f_np, o_bm, o_mm = setup()
    monkeypatch.setattr(type(stepper._ctx), "mem_info", lambda self: (1 << 20, 1 << 38))
    for i in range(4):
        f_0, f_1 = stepper(f_0, f_1, bm, mm, 1.4, i)
        f_0, f_1 = f_1, f_0
    assert stepper._n_fused_pairs == 0 and stepper._deferred is None
    assert np.array_equal(f_0.numpy(), orc.run(f_np, o_bm, o_mm, obcs, 1.4, lat, 4))
