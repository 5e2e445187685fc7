"""Duplicate-index detection (reference xlb/helper/check_boundary_overlaps.py:5-24).
The HIP backend follows the JAX behaviour — the parity target — and WARNS: a later BC in the
list overwrites an earlier one at shared cells."""

import numpy as np


def check_bc_overlaps(bclist, dim, compute_backend):
    chunks = []
    for bc in bclist:
        if getattr(bc, "indices", None) is None:
            continue
        idx = np.asarray(bc.indices)
        if np.unique(idx, axis=-1).shape[-1] != idx.shape[-1]:
            print(f"WARNING: there are duplicate indices in {bc.__class__.__name__} and hence the order in bc list matters!")
        chunks.append(idx[:dim])
    if not chunks:
        return
    allidx = np.concatenate(chunks, axis=-1)
    if np.unique(allidx, axis=-1).shape[-1] != allidx.shape[-1]:
        print("WARNING: there are duplicate indices in the boundary condition list and hence the order in this list matters!")
