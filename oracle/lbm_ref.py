"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes wrapper of oracle/liblbmref.so (lbm_ref.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""

import ctypes as C
import os

import numpy as np

from . import xlb_numpy as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblbmref.so")
KIND_CODE = {orc.KIND_EQUILIBRIUM: 1, orc.KIND_HALFWAY_BB: 2, orc.KIND_FULLWAY_BB: 3, orc.KIND_DO_NOTHING: 4}


class LatticeT(C.Structure):
    _fields_ = [("d", C.c_int), ("q", C.c_int), ("c", (C.c_int * 27) * 3), ("w", C.c_double * 27), ("opp", C.c_int * 27),
                ("cc", (C.c_int * 6) * 27)]


_lib = None


def _open(path):
    lib = C.CDLL(path)
    lib.lbmref_run.restype = C.c_int
    lib.lbmref_set_threads.restype = C.c_int
    return lib


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `make oracle`")
        _lib = _open(LIB_PATH)
    return _lib


def _compile_here(src_name, out_stem, compiler, extra=()):
    """Compile one file of this directory for THE HOST IT RUNS ON into oracle/_build/ (-march=native code must not
    travel between machines: the directory is git- and gpurun-ignored); None when the compiler is unavailable."""
    import subprocess

    out_dir = os.path.join(_HERE, "_build")
    out = os.path.join(out_dir, f"{out_stem}_{os.uname().nodename}.so")
    try:
        os.makedirs(out_dir, exist_ok=True)
        src = os.path.join(_HERE, src_name)
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            subprocess.run([compiler, "-O3", "-march=native", "-fPIC", "-shared", "-fopenmp", *extra, "-o", out, src, "-lm"], check=True,
                           capture_output=True, timeout=300)
        return C.CDLL(out)
    except Exception:
        return None


def build_fast():
    """lbm_ref.c once more, tuned for this host (gcc -O3 -march=native, contraction allowed): NOT bit-exact; the generic
    optimised-build leg of bench.py's CPU baseline (any lattice / collision)."""
    lib = _compile_here("lbm_ref.c", "liblbmref_fast", "gcc")
    if lib is not None:
        lib.lbmref_run.restype = C.c_int
        lib.lbmref_set_threads.restype = C.c_int
    return lib


def build_cpu_port():
    """oracle/lbm_cpu_fast.cpp: the vectorised CPU port of the BGK step (SURVEY 8(d) optimised CPU baseline)."""
    lib = _compile_here("lbm_cpu_fast.cpp", "liblbmcpufast", "g++", extra=("-std=c++17",))
    if lib is not None:
        lib.lbmfast_run.restype = C.c_int
        lib.lbmfast_set_threads.restype = C.c_int
    return lib


def run_cpu_port(lib, f_0, bc_mask, missing_mask, bcs, omega, lat, n_steps, threads=0):
    """n_steps of the vectorised CPU port (fp32, BGK, D3Q19 / D3Q27)."""
    assert lat.q in (19, 27)
    if threads:
        lib.lbmfast_set_threads(int(threads))
    a = np.ascontiguousarray(f_0, dtype=np.float32).copy()
    b = np.empty_like(a)
    nx, ny, nz = a.shape[1:]
    ids, kinds, vals = bc_tables(bcs, lat, "FP32FP32")
    bm = None if bc_mask is None or not len(bcs) else np.ascontiguousarray(bc_mask.reshape(-1), np.uint8)
    mm = None if missing_mask is None or not len(bcs) else np.ascontiguousarray(missing_mask, np.uint8)
    rc = lib.lbmfast_run(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(bm.ctypes.data if bm is not None else None),
                         C.c_void_p(mm.ctypes.data if mm is not None else None), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int(lat.q),
                         C.c_int(len(bcs)), C.c_void_p(ids.ctypes.data), C.c_void_p(kinds.ctypes.data), C.c_void_p(vals.ctypes.data),
                         C.c_double(float(omega)), C.c_int(int(n_steps)))
    if rc:
        raise RuntimeError(f"lbmfast_run failed with code {rc}")
    return a if n_steps % 2 == 0 else b


def lattice_struct(lat):
    L = LatticeT()
    L.d, L.q = lat.d, lat.q
    off = 3 - lat.d
    for l in range(lat.q):
        for a in range(lat.d):
            L.c[a + off][l] = int(lat.c[a, l])
        L.w[l] = float(lat.w[l])
        L.opp[l] = int(lat.opp[l])
        for k in range(lat.cc.shape[1]):
            L.cc[l][k] = int(lat.cc[l, k])
    return L


def bc_tables(bcs, lat, policy):
    """ids, kinds and the per-BC constant vectors (feq of EquilibriumBC, moving-wall term of
    HalfwayBB) evaluated by the NumPy oracle's own formulas."""
    T = orc.compute_dtype(policy)
    ids = np.array([b.id for b in bcs], np.int32)
    kinds = np.array([KIND_CODE[b.kind] for b in bcs], np.int32)
    vals = np.zeros((max(len(bcs), 1), 27), np.float64)
    for i, b in enumerate(bcs):
        if b.kind == orc.KIND_EQUILIBRIUM:
            vals[i, : lat.q] = orc.equilibrium(np.array([b.rho], dtype=T), np.array(b.u, dtype=T), lat, T)
        elif b.kind == orc.KIND_HALFWAY_BB and b.u_wall is not None:
            one = np.zeros((lat.q,) + (1,) * lat.d, T)
            pre = np.zeros_like(one)
            out = orc.apply_bc(b, pre, one, np.full((1,) + (1,) * lat.d, b.id, np.uint8), np.ones(one.shape, bool), lat, policy)
            vals[i, : lat.q] = out.reshape(lat.q)
    return ids, kinds, vals


def set_threads(n, lib=None):
    return (lib or load()).lbmref_set_threads(int(n))


def run(f_0, bc_mask, missing_mask, bcs, omega, lat, n_steps, policy="FP32FP32", collision="BGK", threads=0, lib=None):
    """n_steps of the C restatement; compute == store precision only (FP32FP32 / FP64FP64).  `lib`: a build_fast() handle."""
    assert policy in ("FP32FP32", "FP64FP64")
    lib = lib or load()
    if threads:
        set_threads(threads, lib=lib)
    T = orc.compute_dtype(policy)
    a = np.ascontiguousarray(f_0, dtype=T).copy()
    b = np.empty_like(a)
    shape = a.shape[1:]
    s3 = (1,) + tuple(shape) if len(shape) == 2 else tuple(shape)
    L = lattice_struct(lat)
    ids, kinds, vals = bc_tables(bcs, lat, policy)
    bm = None if bc_mask is None or not len(bcs) else np.ascontiguousarray(bc_mask.reshape(-1), np.uint8)
    mm = None if missing_mask is None or not len(bcs) else np.ascontiguousarray(missing_mask, np.uint8)
    rc = lib.lbmref_run(
        C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(bm.ctypes.data if bm is not None else None),
        C.c_void_p(mm.ctypes.data if mm is not None else None), C.c_int(s3[0]), C.c_int(s3[1]), C.c_int(s3[2]), C.byref(L),
        C.c_int(len(bcs)), C.c_void_p(ids.ctypes.data), C.c_void_p(kinds.ctypes.data), C.c_void_p(vals.ctypes.data),
        C.c_double(float(omega)), C.c_int(0 if collision == "BGK" else 1), C.c_int(1 if T is np.float64 else 0), C.c_int(int(n_steps)),
    )
    if rc:
        raise RuntimeError(f"lbmref_run failed with code {rc}")
    return a if n_steps % 2 == 0 else b
