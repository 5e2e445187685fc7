"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the slab halo protocol, plus the pure
host bookkeeping of the decomposition."""

import os
import socket
import subprocess
import sys

import pytest

from oracle import xlb_numpy as orc
from xlb_amd.distribute import SlabPlan
from xlb_amd.grid.hip_grid import slab_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,walls,lattice,depth", [
    (2, orc.KIND_HALFWAY_BB, "D3Q19", 1), (2, orc.KIND_FULLWAY_BB, "D3Q19", 1), (3, orc.KIND_HALFWAY_BB, "D3Q27", 1),
    # depth 2: one exchange per PAIR of steps, f(t+1) recomputed on the inner ghost planes (the two-step kernel's protocol)
    (2, orc.KIND_HALFWAY_BB, "D3Q19", 2), (2, orc.KIND_FULLWAY_BB, "D3Q19", 2), (3, orc.KIND_HALFWAY_BB, "D3Q19", 2)])
def test_slab_protocol_over_gloo(world, walls, lattice, depth):
    env = dict(os.environ, XLB_TEST_WALLS=walls, XLB_TEST_LATTICE=lattice, XLB_TEST_DEPTH=str(depth), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_dist_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "SLAB_PROTOCOL_OK" in out.stdout


def test_slab_bounds_cover_the_domain():
    for nx in (8, 9, 512, 4096, 13):
        for n in (1, 2, 3, 4, 8):
            if nx < n:
                continue
            spans = [slab_bounds(nx, r, n) for r in range(n)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == nx
            for (a, ca), (b, _) in zip(spans, spans[1:]):
                assert a + ca == b
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_slab_plan_matches_reference_ring():
    # reference xlb/distribute/distribute.py:23-24,26-44: c_x=+1 populations of the LAST plane go to rank+1,
    # c_x=-1 populations of the FIRST plane go to rank-1, periodic ring
    lat = orc.Lattice("D3Q19")
    for world in (2, 4, 8):
        for r in range(world):
            p = SlabPlan(512 * world, r, world, lat.c[0])
            msgs = {m[0]: m for m in p.messages()}
            assert msgs["right"][1].tolist() == [14, 15, 16, 17, 18] and msgs["left"][1].tolist() == [9, 10, 11, 12, 13]
            assert msgs["right"][2] == p.nx_local and msgs["right"][3] == 0
            assert msgs["left"][2] == 1 and msgs["left"][3] == p.nx_local + 1
            assert msgs["right"][4] == (r + 1) % world and msgs["right"][5] == (r - 1) % world
            assert msgs["left"][4] == (r - 1) % world and msgs["left"][5] == (r + 1) % world
    p = SlabPlan(16, 0, 2, lat.c[0], periodic=False)
    assert p.has_right and not p.has_left


def test_slab_plan_depth_two():
    """Two fused steps: every population of the neighbour's edge plane + the crossing ones of the plane behind it;
    storage planes shift with the number of ghost planes."""
    lat = orc.Lattice("D3Q19")
    p = SlabPlan(64, 1, 4, lat.c[0], halo=2)
    nx = p.nx_local
    one = {m[0]: m for m in p.messages(1)}
    assert one["right"][2] == nx + 1 and one["right"][3] == 1 and one["left"][2] == 2 and one["left"][3] == nx + 2
    two = {m[0]: m for m in p.messages(2)}
    assert two["right"][1].tolist() == list(range(19)) and two["right"][2:4] == (nx + 1, 1)
    assert two["right2"][1].tolist() == [14, 15, 16, 17, 18] and two["right2"][2:4] == (nx, 0)
    assert two["left"][1].tolist() == list(range(19)) and two["left"][2:4] == (2, nx + 2)
    assert two["left2"][1].tolist() == [9, 10, 11, 12, 13] and two["left2"][2:4] == (3, nx + 3)
    assert all(m[4] == (2 if m[0].startswith("right") else 0) and m[5] == (0 if m[0].startswith("right") else 2) for m in p.messages(2))
    assert [m[2:4] for m in p.mask_messages()] == [(nx + 1, 1), (2, nx + 2)]
    with pytest.raises(ValueError):
        SlabPlan(64, 1, 4, lat.c[0], halo=1).messages(2)
