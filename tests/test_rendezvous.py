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
