"""Host-side logic of the path that has no device part: `Grid.bounding_box_indices` and
`check_bc_overlaps`.  Runs without a GPU."""

import numpy as np
import pytest

from oracle import xlb_numpy as orc
from xlb_amd.compute_backend import ComputeBackend
from xlb_amd.grid.grid import Grid
from xlb_amd.helper.check_boundary_overlaps import check_bc_overlaps


class _HostGrid(Grid):
    """The index bookkeeping of Grid without a device context."""

    def _initialize_backend(self):
        pass


def _expected_box(shape, remove_edges):
    """What the reference computes (xlb/grid/grid.py:135-191): slices of np.indices(shape), faces flattened in C order."""
    dim = len(shape)
    lo = 1 if remove_edges else 0
    sl = [slice(lo, n - lo) for n in shape]
    g = np.indices(shape)
    if dim == 2:
        nx, ny = shape
        box = {"bottom": g[:, sl[0], 0], "top": g[:, sl[0], ny - 1], "left": g[:, 0, sl[1]], "right": g[:, nx - 1, sl[1]]}
    else:
        nx, ny, nz = shape
        box = {"bottom": g[:, sl[0], sl[1], 0], "top": g[:, sl[0], sl[1], nz - 1], "left": g[:, 0, sl[1], sl[2]], "right": g[:, nx - 1, sl[1], sl[2]],
               "front": g[:, sl[0], 0, sl[2]], "back": g[:, sl[0], ny - 1, sl[2]]}
    return {k: v.reshape(dim, -1).tolist() for k, v in box.items()}


@pytest.mark.parametrize("shape", [(5, 7), (4, 4), (3, 9), (4, 5, 6), (7, 3, 5), (3, 3, 3), (16, 8, 12)])
@pytest.mark.parametrize("remove_edges", [False, True])
def test_bounding_box_indices_element_for_element(shape, remove_edges):
    """Same faces, same index triples, SAME ORDER as the reference — drivers concatenate and np.unique these lists, and
    the masker's "later BC wins" rule makes order observable (grid.py:135-191)."""
    exp = _expected_box(shape, remove_edges)
    got = _HostGrid(shape, ComputeBackend.HIP).bounding_box_indices(remove_edges=remove_edges)
    assert list(got.keys()) == list(exp.keys())
    for face in exp:
        assert got[face] == exp[face], face
        assert all(isinstance(v, int) for v in got[face][0][:3])  # nested Python lists of ints, like ndarray.tolist()
    # the oracle's restatement and the as_numpy form (an extension) agree with it too
    o = orc.bounding_box_indices(shape, remove_edges=remove_edges)
    a = _HostGrid(shape, ComputeBackend.HIP).bounding_box_indices(remove_edges=remove_edges, as_numpy=True)
    for face in exp:
        assert [list(map(int, r)) for r in o[face]] == exp[face], face
        assert a[face].dtype == np.int32 and a[face].tolist() == exp[face], face


def test_bounding_box_explicit_shape_argument():
    g = _HostGrid((8, 8, 8), ComputeBackend.HIP)
    assert g.bounding_box_indices(shape=(4, 5, 6), remove_edges=True) == _expected_box((4, 5, 6), True)
    # remove_edges trims EVERY tangential range by one cell: a face of an n^3 box keeps (n - 2)^2 cells
    assert len(g.bounding_box_indices(remove_edges=True)["top"][0]) == 36
    assert len(g.bounding_box_indices()["top"][0]) == 64


class _BC:
    def __init__(self, indices):
        self.indices = indices


class _MeshBC:
    indices = None


def test_check_bc_overlaps_warns_like_the_jax_backend(capsys):
    """check_boundary_overlaps.py:5-24: duplicates inside one BC and across the BC list are WARNINGS on the parity target
    (JAX; the Warp backend raises) — the later BC in the list then wins (indices_boundary_masker.py:128)."""
    clean = [_BC([[0, 1, 2], [0, 0, 0], [5, 5, 5]]), _BC([[0, 1], [1, 1], [5, 5]]), _MeshBC()]
    check_bc_overlaps(clean, 3, ComputeBackend.HIP)
    assert capsys.readouterr().out == ""
    # duplicate column inside one BC
    check_bc_overlaps([_BC([[0, 1, 0], [0, 0, 0], [5, 5, 5]])], 3, ComputeBackend.HIP)
    out = capsys.readouterr().out
    assert "WARNING: there are duplicate indices in _BC and hence the order in bc list matters!" in out
    assert "duplicate indices in the boundary condition list" in out  # the concatenated list has the duplicate as well
    # the same cell in two BCs
    check_bc_overlaps([_BC([[0, 1], [0, 0], [5, 5]]), _BC([[1, 2], [0, 0], [5, 5]])], 3, ComputeBackend.HIP)
    out = capsys.readouterr().out
    assert "in _BC" not in out
    assert "WARNING: there are duplicate indices in the boundary condition list and hence the order in this list matters!" in out
    # 2-D lists, NumPy arrays as indices, nothing to check
    check_bc_overlaps([_BC(np.array([[0, 1], [3, 3]])), _BC([[1], [3]])], 2, ComputeBackend.HIP)
    assert "boundary condition list" in capsys.readouterr().out
    check_bc_overlaps([], 3, ComputeBackend.HIP)
    check_bc_overlaps([_MeshBC()], 3, ComputeBackend.HIP)
    assert capsys.readouterr().out == ""


def test_later_bc_wins_on_true_duplicates_in_the_oracle_masker():
    """The semantics the warning announces, on the oracle's masker (App. B.2 step 2): the same cells given to two BCs end
    up with the id of the LATER one in the list, whatever the construction (id) order."""
    lat = orc.Lattice("D3Q19")
    shape = (6, 6, 6)
    cells = [[0, 0, 0], [1, 2, 3], [0, 0, 0]]
    first = orc.BC(orc.KIND_FULLWAY_BB, 2, cells)
    later = orc.BC(orc.KIND_HALFWAY_BB, 1, [c[:2] for c in cells])
    bm, _ = orc.build_masks(shape, lat, [first, later])
    assert bm[0, 0, 1, 0] == 1 and bm[0, 0, 2, 0] == 1 and bm[0, 0, 3, 0] == 2
    bm, _ = orc.build_masks(shape, lat, [later, first])
    assert bm[0, 0, 1, 0] == 2 and bm[0, 0, 2, 0] == 2 and bm[0, 0, 3, 0] == 2
