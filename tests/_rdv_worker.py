"""Worker of tests/test_rendezvous.py: every collective of xlb_amd.distribute.rendezvous on WORLD_SIZE ranks."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xlb_amd.distribute import rendezvous  # noqa: E402


def main():
    rdv = rendezvous.from_env(timeout=60)
    r, n = rdv.rank, rdv.world
    ok = True
    # all_gather of arrays (dtype / shape preserved), scalars, bytes
    parts = rdv.all_gather(np.full((2, r + 1), r, np.float32))
    ok &= len(parts) == n and all(p.dtype == np.float32 and p.shape == (2, i + 1) and (p == i).all() for i, p in enumerate(parts))
    ok &= rdv.all_gather(0.1 * r) == [0.1 * i for i in range(n)]  # doubles survive the JSON round trip exactly
    ok &= rdv.broadcast(bytes(range(128)) if r == 0 else None, src=0) == bytes(range(128))
    ok &= rdv.all_reduce(r, "max") == n - 1 and rdv.all_reduce(r, "min") == 0 and rdv.all_reduce(r + 1, "sum") == n * (n + 1) / 2
    # ring through the hub: payload to both neighbours (the same peer twice when n == 2 is ONE frame per peer)
    out = {(r + 1) % n: rendezvous.pack(np.array([r, 1])), }
    if n > 2:
        out[(r - 1) % n] = rendezvous.pack(np.array([r, -1]))
    got = {src: rendezvous.unpack(b) for src, b in rdv.route(out).items()}
    ok &= (got[(r - 1) % n] == np.array([(r - 1) % n, 1])).all()
    if n > 2:
        ok &= (got[(r + 1) % n] == np.array([(r + 1) % n, -1])).all()
    ok &= rdv.route({}) == {}
    # bulk frames (several MB: the shared-memory mailbox has to grow; a frame to oneself; uneven sizes), twice in a row
    want_shm = os.environ.get("XLB_RDV_SHM", "1") != "0"
    ok &= rdv._shm == want_shm
    for rep in range(2):
        big = {(r + 1) % n: bytes([r + rep]) * (3_000_000 + 1000 * r), r: b"self" + bytes([rep])}
        got = rdv.route(big)
        src = (r - 1) % n
        ok &= got[r] == b"self" + bytes([rep]) and len(got[src]) == 3_000_000 + 1000 * src and got[src][:2] == bytes([src + rep]) * 2 and got[src][-1] == src + rep
        ok &= set(got) == {r, src}
    if want_shm:
        ok &= os.path.exists(rdv._box_path(r))
    rdv.barrier()
    flags = rdv.all_gather(bool(ok))
    if r == 0:
        print("RDV_OK" if all(flags) else f"RDV_FAIL {flags}", flush=True)
    path = rdv._box_path(r)
    rdv.close()
    if os.path.exists(path):  # the mailbox file goes with the rendezvous
        sys.exit(2)
    sys.exit(0 if all(flags) else 1)


if __name__ == "__main__":
    main()
