"""The HIP slab path beyond one rank, on the one-GPU boxes of this pool:
  * two ranks sharing device 0 with the host-staged (gloo) halo transport — everything of the
    multi-rank path except the RCCL calls: per-slab masks from global indices, ghost planes,
    uneven slabs, periodic ring;
  * two and three ranks sharing device 0 with the IPC transport — device-to-device ghost planes on the communication
    stream, the native protocol end to end (overlap, pairing vote, meta planes), only the wire is not xGMI;
  * one rank with a REAL RCCL communicator (self send/recv) — the RCCL calls, the communication
    stream and the interior/edge overlap, minus the wire;
  * two ranks over RCCL proper: runs when RCCL accepts the device set, skipped on "Duplicate GPU".
"""

import os
import subprocess
import sys

import pytest

from test_distributed_gloo import ROOT, free_port

pytestmark = pytest.mark.gpu


def torchrun(script, world, **env):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", script)]
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_host_staged_halo(world):
    """(world 3: a rank with two neighbours, uneven slabs of three different sizes)"""
    out = torchrun("_gpu_rank_worker.py", world, XLB_HIP_DEVICE="0", XLB_TEST_TRANSPORT="host")
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "GPU_SLAB_OK" in out.stdout


@pytest.mark.parametrize("world", [2, 3, 4])
def test_ranks_sharing_one_gpu_ipc_halo(world):
    """The device-to-device transport that does not need RCCL: hipIpc-mapped neighbour fields pulled
    with copies on the communication stream, ordered by counters in host shared memory — with 2 and 3 ranks on ONE GPU
    the whole native slab protocol runs: depth-1 and depth-2 exchanges, chain ends, uneven slabs, overlap on / off, the
    collective pairing vote, the meta-plane exchange.  Bit-exact against the single-domain oracle.  (4 ranks: with the test runner itself that
    is as many processes as this pool lets share a card safely — its guard allows 6 and killed a 6-rank attempt.)"""
    out = torchrun("_gpu_rank_worker.py", world, XLB_HIP_DEVICE="0", XLB_TEST_TRANSPORT="ipc")
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "GPU_SLAB_OK" in out.stdout


def test_ipc_chain_three_ranks():
    """periodic_x=False: the ring opened into a chain — what bench.py runs the halfway cavity (BASELINE configs[2] / [3]) with.
    Rank 0 has no left neighbour, rank 2 no right one, rank 1 both; the result is the ring's, i.e. the single-domain oracle's."""
    out = torchrun("_gpu_rank_worker.py", 3, XLB_HIP_DEVICE="0", XLB_TEST_TRANSPORT="ipc", XLB_TEST_CHAIN="1")
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "GPU_SLAB_OK" in out.stdout


def test_one_rank_real_rccl_self_exchange():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_gpu_rccl_self_worker.py")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "RCCL_SELF_OK" in out.stdout


def test_two_ranks_rccl_or_host_falls_back_together():
    """transport="auto" (bench.py's default; "rccl_or_host" is its round-2 name): on a one-GPU box RCCL refuses the second rank on the
    device, every rank hears about it through the rendezvous, all of them move on to the next transport of the chain (ipc, then host)
    and the run is still bit-identical to the single-domain oracle.  (Where RCCL accepts the device set this is the RCCL run.)"""
    out = torchrun("_gpu_rank_worker.py", 2, XLB_HIP_DEVICE="0", XLB_TEST_TRANSPORT="rccl_or_host")
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert "GPU_SLAB_OK" in out.stdout
    if "falls back to" in text:
        assert "halo transport 'rccl' failed" in text and "every rank falls back to 'ipc'" in text


def test_two_ranks_rccl():
    out = torchrun("_gpu_rank_worker.py", 2, XLB_HIP_DEVICE="0", XLB_TEST_TRANSPORT="rccl", NCCL_DEBUG="WARN")
    text = out.stdout + out.stderr
    if out.returncode != 0 and "Duplicate GPU detected" in text:
        pytest.skip("RCCL refuses two ranks on one device (one-GPU box)")
    assert out.returncode == 0, text[-3000:]
    assert "GPU_SLAB_OK" in out.stdout


def test_config3_global_domain_decomposed_matches_one_rank(tmp_path):
    """BASELINE configs[3] names 8 GPUs; this pool has one.  What CAN be exercised is its global domain and its decomposition: the
    4096 x 512 x 512 halfway cavity (1.07 G cells, 163 GB of populations) is advanced 7 steps by one rank and then, slab-decomposed,
    by FOUR ranks (with the test runner the most processes the pool lets share a card) of 1024 planes each that exchange their ghost
    planes device to device over the ipc transport — fused pairs with the depth-2 exchange, interior / edge launches, strip buffers,
    then a single step with the depth-1 exchange.  Every population plane checked — the x walls, both sides of every rank boundary,
    planes in between — is the one-rank run's, bit for bit.  ("configs[3] ... never the decomposition".)"""
    path = str(tmp_path / "config3_planes.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", XLB_C3_FILE=path, XLB_HIP_DEVICE="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_gpu_config3_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0 and "CONFIG3_REF_OK" in one.stdout, (one.stdout + one.stderr)[-3000:]
    out = torchrun("_gpu_config3_worker.py", 4, XLB_HIP_DEVICE="0", XLB_TEST_TRANSPORT="ipc", XLB_C3_FILE=path)
    assert out.returncode == 0 and "CONFIG3_DECOMPOSED_OK" in out.stdout, (out.stdout + out.stderr)[-3000:]
