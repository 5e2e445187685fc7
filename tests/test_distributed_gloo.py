"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the slab halo protocol, plus the pure
host bookkeeping of the decomposition."""

import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import xlb_numpy as orc
from xlb_amd.distribute import SlabPlan
from xlb_amd.grid.hip_grid import slab_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,walls,lattice", [(2, orc.KIND_HALFWAY_BB, "D3Q19"), (2, orc.KIND_FULLWAY_BB, "D3Q19"), (3, orc.KIND_HALFWAY_BB, "D3Q27")])
def test_slab_protocol_over_gloo(world, walls, lattice):
    env = dict(os.environ, XLB_TEST_WALLS=walls, XLB_TEST_LATTICE=lattice, OMP_NUM_THREADS="1")
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
