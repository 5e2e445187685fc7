"""Worker of tests/test_rendezvous.py::test_transport_chain_*: xlb_amd.distribute.init_process_group's fallback chain (rccl -> ipc -> host)
on CPU, with a fake device context whose transports fail where the scenario says.  What is under test is the COLLECTIVE part: every rank
makes the same calls in the same order whatever fails where, and all ranks end up on the same transport."""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from xlb_amd import _lib  # noqa: E402
from xlb_amd import default_config  # noqa: E402
import importlib  # noqa: E402

xd = importlib.import_module("xlb_amd.distribute.distribute")  # (the package re-exports a FUNCTION of that name)


class FakeCtx:
    def __init__(self, rank, scenario):
        self.rank, self.n_ranks, self._me, self.s = 0, 1, rank, scenario
        self.opts, self.log = {}, []

    def comm_init(self, rank, n, uid, periodic_x=True):
        self.log.append("rccl")
        if self.s == "rccl_fails_on_rank1" and rank == 1:
            raise RuntimeError("ncclCommInitRank failed: Duplicate GPU detected")
        self.rank, self.n_ranks = rank, n

    def comm_init_ipc(self, rank, n, token, periodic_x=True):
        self.log.append("ipc")
        assert isinstance(token, str) and len(token) == 24
        if self.s in ("all_fail", "forced_ipc_fails") and rank == 0:
            raise RuntimeError("hipIpcGetMemHandle: invalid argument")
        self.rank, self.n_ranks = rank, n

    def comm_destroy(self):
        self.log.append("destroy")

    def set_option(self, k, v):
        self.opts[k] = v

    def sync(self):
        pass


class FakeField:
    def __init__(self, *a, **k):
        pass

    def free(self):
        pass


def main():
    scenario = os.environ["XLB_CHAIN_SCENARIO"]
    rank = int(os.environ["RANK"])
    ctx = FakeCtx(rank, scenario)
    default_config.DefaultConfig.context = ctx
    _lib.comm_unique_id = lambda: bytes(128)
    _lib.Field = FakeField

    def verify(c, f, r, world, periodic):
        c.log.append("verify")
        if scenario == "all_fail" and r == 0 and c.log.count("verify") == 1:
            raise RuntimeError("ghost plane 1 of population 3 holds 0.0, expected 1031.0")  # RCCL came up and moved wrong bytes

    xd._verify_exchange = verify
    transport = {"forced_ipc_fails": "ipc"}.get(scenario, "auto")
    try:
        xd.init_process_group(periodic_x=False, transport=transport)
        print(f"CHAIN rank {rank} transport={xd.transport()!r} external_halo={ctx.opts.get('external_halo', 0)} calls={','.join(ctx.log)}", flush=True)
    except RuntimeError as e:
        print(f"CHAIN rank {rank} raised {e}", flush=True)
    xd.shutdown()


if __name__ == "__main__":
    main()
