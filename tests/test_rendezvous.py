"""The torch-free control plane (xlb_amd/distribute/rendezvous.py) and bench.py's self-launcher, on CPU."""

import json
import os
import subprocess
import sys

import pytest

from test_distributed_gloo import ROOT, free_port

WORKER = os.path.join(ROOT, "tests", "_rdv_worker.py")


def spawn(world, port, extra_env=None):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    return [p.returncode for p in procs], outs


@pytest.mark.parametrize("world", [2, 3, 5])
def test_collectives_plain_processes(world):
    codes, outs = spawn(world, free_port())
    assert codes == [0] * world, outs
    assert "RDV_OK" in outs[0][0]


def test_route_through_the_hub_when_shared_memory_is_off():
    """XLB_RDV_SHM=0 (or ranks on different hosts): bulk frames travel through the TCP hub instead of the /dev/shm mailboxes — same results."""
    codes, outs = spawn(3, free_port(), {"XLB_RDV_SHM": "0"})
    assert codes == [0] * 3, outs
    assert "RDV_OK" in outs[0][0]


def test_hub_skips_a_port_that_is_taken():
    """MASTER_PORT itself occupied by somebody else (as under torch.distributed.run, whose store owns it): the hub moves to
    the next candidate and the other ranks find it there."""
    import socket

    with socket.socket() as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("127.0.0.1", 0))
        s.listen(4)  # a foreign listener that never answers the handshake
        codes, outs = spawn(2, s.getsockname()[1], {"XLB_RDV_TIMEOUT": "60"})
    assert codes == [0, 0], outs
    assert "RDV_OK" in outs[0][0]


def test_under_torch_distributed_run():
    """The driver's launch form: python -m torch.distributed.run --nproc-per-node N ... (MASTER_PORT belongs to torchrun's store)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), WORKER]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "RDV_OK" in out.stdout


def test_package_does_not_import_torch():
    code = "import sys; import xlb_amd, xlb_amd.distribute, xlb_amd.distribute.rendezvous; assert 'torch' not in sys.modules, 'torch imported'"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]


@pytest.mark.parametrize("form", ["self-launch", "torchrun"])
def test_bench_launcher_dry_run(form):
    """`python bench.py --gpus 2` from a plain shell spawns its own ranks; the driver's torchrun form
    still works.  --dry-run: rendezvous + barrier only, no GPU."""
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    if form == "self-launch":
        cmd = [sys.executable, bench, "--gpus", "2", "--dry-run"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), bench, "--gpus", "2", "--dry-run"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks"] == [0, 1] and rec["dry_run"] is True


def test_bench_launcher_propagates_failure():
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    # no GPU / bad option -> children fail; the parent must exit non-zero and print no JSON line
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "0", "--size", "16", "--opt", "no_such_option=1"],
                         capture_output=True, text=True, timeout=300, env=dict(env, XLB_RDV_TIMEOUT="20"))
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def _chain(scenario):
    port = free_port()
    worker = os.path.join(ROOT, "tests", "_chain_worker.py")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), XLB_CHAIN_SCENARIO=scenario)
        procs.append(subprocess.Popen([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    return [[l for l in o[0].splitlines() if l.startswith("CHAIN")][0] for o in outs]


def test_transport_chain_rccl_fails_on_one_rank_everybody_moves_to_ipc():
    """bench.py joins with transport="auto": RCCL, then ipc, then host.  A transport that fails on ANY rank is dropped by ALL ranks
    (a rank that kept it would wait for the others forever) — here RCCL refuses rank 1 only; both ranks tear it down, bring ipc up,
    pass its self-check and report the same transport string."""
    lines = _chain("rccl_fails_on_rank1")
    assert all("transport='ipc (fallback: rccl: rank 1: ncclCommInitRank failed: Duplicate GPU detected)'" in l and "external_halo=0" in l for l in lines), lines
    assert "calls=rccl,destroy,ipc,verify" in lines[0] and "calls=rccl,destroy,ipc,verify" in lines[1]


def test_transport_chain_ends_on_the_host_transport_when_everything_fails():
    """RCCL comes up but its self-check finds wrong bytes on rank 0; ipc cannot export memory on rank 0: both ranks end on the
    host-staged transport (external_halo = 1), with both reasons in the transport string."""
    lines = _chain("all_fail")
    for l in lines:
        assert "transport='host (fallback: rccl: rank 0: rccl self-check: ghost plane 1" in l and "ipc: rank 0: hipIpcGetMemHandle" in l and "external_halo=1" in l, l


def test_forced_transport_failure_raises_on_every_rank():
    lines = _chain("forced_ipc_fails")
    assert all("raised halo transport 'ipc' failed: rank 0: hipIpcGetMemHandle: invalid argument" in l for l in lines), lines
