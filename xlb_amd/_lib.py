"""ctypes binding of libxlbhip.so (include/xlbhip.h) and thin RAII wrappers.

This is the stub a reference maintainer would add for a ``ComputeBackend.HIP`` (see
INTEGRATION.md): every ``hip_implementation`` method of the operators calls one function of
the C ABI through the objects defined here.  There is NO fallback: if the shared library is
missing or a call fails, an exception is raised.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XLBHIP_LIB") or os.path.join(_HERE, "lib", "libxlbhip.so")  # XLBHIP_LIB: A/B builds (tools/)

# element types (include/xlbhip.h)
F64, F32, F16, U8, BOOL, MISSING = 0, 1, 2, 3, 4, 5
D2Q9, D3Q19, D3Q27 = 0, 1, 2
BGK, KBC, SMAGORINSKY_LES_BGK = 0, 1, 2
BC_EQUILIBRIUM, BC_HALFWAY_BB, BC_FULLWAY_BB, BC_DO_NOTHING = 1, 2, 3, 4
BC_ZOUHE_VELOCITY, BC_ZOUHE_PRESSURE, BC_REGULARIZED_VELOCITY, BC_REGULARIZED_PRESSURE = 5, 6, 7, 8
BC_EXTRAPOLATION_OUTFLOW = 9
BC_HYBRID_BB_REGULARIZED, BC_HYBRID_BB_GRADS, BC_HYBRID_NEQ_REGULARIZED = 10, 11, 12
BC_HALFWAY_BB_PROFILE = 13
MESH_AABB, MESH_RAY, MESH_AABB_CLOSE, MESH_WINDING = 1, 2, 3, 4
UNIQUE_ID_BYTES = 128

NP_OF_DTYPE = {F64: np.float64, F32: np.float32, F16: np.float16, U8: np.uint8, BOOL: np.bool_, MISSING: np.uint8}


class HipBackendError(RuntimeError):
    pass


class BcDesc(C.Structure):
    _fields_ = [("id", C.c_int32), ("kind", C.c_int32), ("values", C.c_double * 27)]


_p = C.c_void_p
_pp = C.POINTER(C.c_void_p)
_i = C.c_int
_i64 = C.c_int64
_d = C.c_double

# name -> argtypes ; every function returns int except xlbhip_last_error
SIGNATURES = {
    "xlbhip_create": [_i, _pp],
    "xlbhip_destroy": [_p],
    "xlbhip_sync": [_p],
    "xlbhip_device_info": [_p, C.c_char_p, _i, C.POINTER(_i), C.POINTER(C.c_uint64)],
    "xlbhip_set_option": [_p, C.c_char_p, _i64],
    "xlbhip_get_option": [_p, C.c_char_p, C.POINTER(_i64)],
    "xlbhip_lattice_info": [_i, C.POINTER(_i), C.POINTER(_i), _p, _p, _p, _p],
    "xlbhip_field_create": [_p, _i, _i, _i, _i, _i, _i, _d, _pp],
    "xlbhip_field_destroy": [_p],
    "xlbhip_field_fill": [_p, _d],
    "xlbhip_field_copy": [_p, _p],
    "xlbhip_field_copy_kernel": [_p, _p, _i],
    "xlbhip_field_copy_tiles": [_p, _p],
    "xlbhip_field_upload": [_p, _p, C.c_size_t],
    "xlbhip_field_download": [_p, _p, C.c_size_t],
    "xlbhip_field_plane_download": [_p, _i, _i, _p, C.c_size_t],
    "xlbhip_field_plane_upload": [_p, _i, _i, _p, C.c_size_t],
    "xlbhip_field_touch": [_p],
    "xlbhip_mem_info": [_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "xlbhip_field_info": [_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i),
                          C.POINTER(C.c_uint64), _pp],
    "xlbhip_stream": [_p, _i, _p, _p],
    "xlbhip_equilibrium": [_p, _i, _i, _p, _p, _p],
    "xlbhip_macroscopic": [_p, _i, _i, _p, _p, _p],
    "xlbhip_second_moment": [_p, _i, _i, _p, _p],
    "xlbhip_vorticity": [_p, _p, _p, _p, _p],
    "xlbhip_mesh_mask_aabb": [_p, _i, _i, _i64, _p, _p, _p],
    "xlbhip_mesh_mask_ray": [_p, _i, _i, _i64, _p, _p, _p],
    "xlbhip_mesh_mask": [_p, _i, _i, _i, _i64, _p, _i, _p, _p, _p],
    "xlbhip_field_gather": [_p, _i64, _p, _p, C.c_size_t],
    "xlbhip_stepper_set_bc_distances": [_p, _i64, _p, _p],
    "xlbhip_stepper_momentum_transfer": [_p, _i, _p, _p, _p, C.POINTER(C.c_double)],
    "xlbhip_grid_to_point": [_p, _p, _i64, _p, _p],
    "xlbhip_momentum_transfer": [_p, _i, _i, C.POINTER(BcDesc), _p, _p, _p, C.POINTER(C.c_double)],
    "xlbhip_stepper_set_bc_profile": [_p, _i, _i64, _p, _p],
    "xlbhip_apply_bc_profile": [_p, _i, _i, C.POINTER(BcDesc), _p, _p, _p, _p, _i64, _p, _p],
    "xlbhip_q_criterion": [_p, _p, _p, _p, _p],
    "xlbhip_collide": [_p, _i, _i, _i, _p, _p, _p, _d],
    "xlbhip_apply_bc": [_p, _i, _i, C.POINTER(BcDesc), _p, _p, _p, _p],
    "xlbhip_build_masks": [_p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p],
    "xlbhip_stepper_create": [_p, _i, _i, _i, _i, _i, C.POINTER(BcDesc), _pp],
    "xlbhip_stepper_destroy": [_p],
    "xlbhip_stepper_set_force": [_p, _p],
    "xlbhip_stepper_set_smagorinsky": [_p, _d],
    "xlbhip_step": [_p, _p, _p, _p, _p, _d, _i64],
    "xlbhip_run": [_p, _p, _p, _p, _p, _d, _i64, _i64],
    "xlbhip_run_timed": [_p, _p, _p, _p, _p, _d, _i64, _i64, C.POINTER(C.c_float), C.POINTER(C.c_int)],
    "xlbhip_run_any": [_p, _p, _p, _p, _p, _d, _i64, _i64, C.POINTER(C.c_int)],
    "xlbhip_comm_unique_id": [_p],
    "xlbhip_comm_init": [_p, _i, _i, _p, _i],
    "xlbhip_comm_init_ipc": [_p, _i, _i, C.c_char_p, _i],
    "xlbhip_comm_destroy": [_p],
    "xlbhip_comm_stats": [_p, C.POINTER(C.c_double), C.POINTER(_i64), _i],
    "xlbhip_halo_exchange": [_p, _i, _p],
    "xlbhip_halo_exchange_wide": [_p, _i, _p],
    "xlbhip_step2_eligible": [_p, _p, _p, _p, _p],
    "xlbhip_step2": [_p, _p, _p, _p, _p, _d, _i64],
}

_lib = None


def load():
    """Load libxlbhip.so; fail loudly when it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipBackendError(
            f"{LIB_PATH} not found: build it with `make` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "xlb_amd has no CPU fallback."
        )
    # Multi-process runs share device memory through dmabuf IPC on this driver stack; the HIP runtime reads the switch
    # when it initialises, i.e. with the first call into the library (RCCL: "hipIpcGetMemHandle: invalid argument" otherwise)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.xlbhip_last_error.argtypes = []
    lib.xlbhip_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise HipBackendError(load().xlbhip_last_error().decode("utf-8", "replace"))


def lattice_info(lattice_id):
    lib = load()
    d, q = _i(), _i()
    c = np.zeros((3, 27), np.int32)
    w = np.zeros(27, np.float64)
    opp = np.zeros(27, np.int32)
    cc = np.zeros((27, 6), np.int32)
    cbuf = np.zeros(3 * 27, np.int32)
    ccbuf = np.zeros(27 * 6, np.int32)
    check(lib.xlbhip_lattice_info(lattice_id, C.byref(d), C.byref(q), cbuf.ctypes.data, w.ctypes.data, opp.ctypes.data, ccbuf.ctypes.data))
    qq = q.value
    c = cbuf[: 3 * qq].reshape(3, qq)
    cc = ccbuf[: qq * 6].reshape(qq, 6)
    return d.value, qq, c, w[:qq].copy(), opp[:qq].copy(), cc


class Context:
    """Device context: HIP device + compute/communication streams (xlbhip_create)."""

    def __init__(self, device=0):
        self._h = _p()
        check(load().xlbhip_create(int(device), C.byref(self._h)))
        self.device = int(device)
        self.rank, self.n_ranks = 0, 1
        self._flushers = []  # weak references to objects with deferred device work (steppers pairing reference-style calls)

    @property
    def handle(self):
        if not self._h:
            raise HipBackendError("context already destroyed")
        return self._h

    def flush_deferred(self):
        """Enqueue whatever work operators have deferred (IncompressibleNavierStokesStepper pairs reference-style calls)."""
        alive = []
        for ref in self._flushers:
            obj = ref()
            if obj is not None:
                obj._flush_deferred()
                alive.append(ref)
        self._flushers = alive

    def sync(self):
        self.flush_deferred()
        check(load().xlbhip_sync(self.handle))

    def set_option(self, key, value):
        check(load().xlbhip_set_option(self.handle, key.encode(), int(value)))

    def get_option(self, key):
        v = _i64()
        check(load().xlbhip_get_option(self.handle, key.encode(), C.byref(v)))
        return v.value

    def mem_info(self):
        """(free, total) device memory in bytes."""
        fr, tot = C.c_uint64(), C.c_uint64()
        check(load().xlbhip_mem_info(self.handle, C.byref(fr), C.byref(tot)))
        return fr.value, tot.value

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, hbm = _i(), C.c_uint64()
        check(load().xlbhip_device_info(self.handle, name, 256, C.byref(cus), C.byref(hbm)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}

    def comm_init(self, rank, n_ranks, id_bytes, periodic_x=True):
        buf = C.create_string_buffer(bytes(id_bytes), UNIQUE_ID_BYTES) if id_bytes is not None else None
        check(load().xlbhip_comm_init(self.handle, int(rank), int(n_ranks), buf, 1 if periodic_x else 0))
        self.rank, self.n_ranks = int(rank), int(n_ranks)

    def comm_init_ipc(self, rank, n_ranks, token, periodic_x=True):
        """Join the IPC halo transport of the job named by ``token`` (xlbhip_comm_init_ipc)."""
        check(load().xlbhip_comm_init_ipc(self.handle, int(rank), int(n_ranks), str(token).encode(), 1 if periodic_x else 0))
        self.rank, self.n_ranks = int(rank), int(n_ranks)

    def comm_destroy(self):
        check(load().xlbhip_comm_destroy(self.handle))

    def comm_stats(self, reset=False):
        """{"halo_wait_ms": ..., "halo_waits": ...}: time the compute stream waited for halo exchanges (xlbhip_comm_stats)."""
        ms, n = C.c_double(), _i64()
        check(load().xlbhip_comm_stats(self.handle, C.byref(ms), C.byref(n), 1 if reset else 0))
        return {"halo_wait_ms": ms.value, "halo_waits": n.value}

    def close(self):
        if self._h:
            load().xlbhip_destroy(self._h)
            self._h = _p()


def comm_unique_id():
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    check(load().xlbhip_comm_unique_id(buf))
    return buf.raw


# ---- DLPack (dlpack.h v0.8 ABI) through ctypes: no extension module needed ---------------------------------
class _DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int), ("device_id", C.c_int)]


class _DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class _DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", _DLDevice), ("ndim", C.c_int), ("dtype", _DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class _DLManagedTensor(C.Structure):
    pass


_DL_DELETER = C.CFUNCTYPE(None, C.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DL_DELETER)]
_dl_alive = {}  # address of the managed tensor -> (managed tensor, shape/stride arrays, owner): kept until the consumer's deleter runs


@_DL_DELETER
def _dl_deleter(handle):
    try:
        _dl_alive.pop(C.addressof(handle.contents), None)
    except Exception:  # interpreter shutdown: the module globals may already be gone
        pass


# A consumer may drop its last tensor during interpreter finalisation, after this module's globals were cleared: the
# native thunk of the deleter must outlive everything, so it is leaked on purpose.
C.pythonapi.Py_IncRef(C.py_object(_dl_deleter))


# The capsule destructor runs while the capsule is being deallocated (reference count 0): it must see the raw
# PyObject* — a ctypes py_object argument would INCREF / DECREF the dying object and free it twice.
_PyCapsule_IsValid = C.PYFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p)(("PyCapsule_IsValid", C.pythonapi))
_PyCapsule_GetPointer = C.PYFUNCTYPE(C.c_void_p, C.c_void_p, C.c_char_p)(("PyCapsule_GetPointer", C.pythonapi))
_PyCapsule_New = C.PYFUNCTYPE(C.py_object, C.c_void_p, C.c_char_p, C.c_void_p)(("PyCapsule_New", C.pythonapi))


@C.PYFUNCTYPE(None, C.c_void_p)
def _dl_capsule_destructor(capsule_ptr):
    # a capsule nobody consumed still carries the name "dltensor": release what it points to
    if _PyCapsule_IsValid(capsule_ptr, b"dltensor"):
        _dl_alive.pop(_PyCapsule_GetPointer(capsule_ptr, b"dltensor"), None)


def _dlpack_capsule(owner, ptr, shape, strides, dtype, device_id):
    nd = len(shape)
    shp = (C.c_int64 * nd)(*shape)
    std = (C.c_int64 * nd)(*strides)  # DLPack strides are in elements
    code = {"f": 2, "u": 1, "i": 0, "b": 6}[np.dtype(dtype).kind]
    m = _DLManagedTensor()
    m.dl_tensor = _DLTensor(C.c_void_p(ptr), _DLDevice(10, int(device_id)), nd, _DLDataType(code, np.dtype(dtype).itemsize * 8, 1),
                            C.cast(shp, C.POINTER(C.c_int64)), C.cast(std, C.POINTER(C.c_int64)), 0)
    m.manager_ctx = None
    m.deleter = _dl_deleter
    _dl_alive[C.addressof(m)] = (m, shp, std, owner)
    return _PyCapsule_New(C.addressof(m), b"dltensor", C.cast(_dl_capsule_destructor, C.c_void_p))


class Field:
    """A device-resident field.  Host-visible shape is (cardinality, nx, ny[, nz]) C-order,
    exactly the reference layout; the device layout is private."""

    def __init__(self, ctx, cardinality, shape, dtype_code, halo=0, fill_value=0.0):
        self.ctx = ctx
        self.cardinality = int(cardinality)
        self.grid_shape = tuple(int(s) for s in shape)
        assert len(self.grid_shape) in (2, 3)
        self.dtype_code = int(dtype_code)
        self.halo = int(halo)
        s3 = (1,) + self.grid_shape if len(self.grid_shape) == 2 else self.grid_shape
        self._s3 = s3
        self._hook = None     # called before any access through .handle (deferred work that involves this field)
        self._pinned = False  # exported through DLPack / CUDA array interface: its device memory must never be swapped
        self._h = _p()
        check(load().xlbhip_field_create(ctx.handle, self.cardinality, s3[0], s3[1], s3[2], self.dtype_code, self.halo,
                                         float(fill_value or 0.0), C.byref(self._h)))

    # -- reference-like attributes
    @property
    def shape(self):
        return (self.cardinality,) + self.grid_shape

    @property
    def dtype(self):
        return np.dtype(NP_OF_DTYPE[self.dtype_code])

    @property
    def handle(self):
        if self._hook is not None:
            hook, self._hook = self._hook, None
            try:
                hook(self)
            except BaseException:
                # the deferred work did not happen: whoever looks next must not see the field as if it had (a failed
                # materialisation of a virtual f(t+1) would otherwise hand out f(t))
                if self._hook is None:
                    self._hook = hook
                raise
        if not self._h:
            raise HipBackendError("field already destroyed")
        return self._h

    def touch(self):
        """The contents changed behind the library's back (a write through a zero-copy alias): invalidate what is cached on them."""
        check(load().xlbhip_field_touch(self.handle))

    @property
    def nbytes_host(self):
        return int(np.prod(self.shape)) * self.dtype.itemsize

    def numpy(self):
        """Synchronous device -> host copy in the reference layout."""
        out = np.empty(self.shape, dtype=self.dtype)
        check(load().xlbhip_field_download(self.handle, out.ctypes.data, out.nbytes))
        return out

    # -- zero-copy export (what replaces the reference's ToJAX / warp_array_to_jax helpers, utils/utils.py:340-447):
    #    the interior of the field as a strided (cardinality, nx, ny[, nz]) device array.  The population planes are
    #    `plane_stride` elements apart (padding, ghost planes), which strides express exactly.
    def _device_view(self):
        if self.dtype_code == MISSING:
            raise HipBackendError("the bit-packed missing_mask has no array view; use numpy()")
        info = self.info()
        item = self.dtype.itemsize
        ny, nz = self._s3[1], self._s3[2]
        ptr = info["device_ptr"] + self.halo * ny * nz * item
        strides = (info["plane_stride"], ny * nz, nz, 1) if len(self.grid_shape) == 3 else (info["plane_stride"], nz, 1)
        return ptr, self.shape, strides, item

    @property
    def __cuda_array_interface__(self):
        """CUDA Array Interface v3 (PyTorch-ROCm, CuPy-ROCm and Numba consume it): ``torch.as_tensor(field, device="cuda")``
        aliases the field's memory.  Work enqueued on the backend's stream is finished first."""
        ptr, shape, strides, item = self._device_view()
        self._pinned = True
        self.ctx.sync()
        return {"shape": shape, "strides": tuple(s * item for s in strides), "typestr": self.dtype.str, "data": (ptr, False), "version": 3}

    def __dlpack_device__(self):
        return (10, self.ctx.device)  # kDLROCM

    def __dlpack__(self, stream=None, **_):
        """DLPack capsule of the same strided view (``torch.from_dlpack(field)``).  The consumer's stream is not known to
        this backend, so the backend's own stream is drained before the capsule is handed out."""
        ptr, shape, strides, item = self._device_view()
        self._pinned = True
        self.ctx.sync()
        return _dlpack_capsule(self, ptr, shape, strides, self.dtype, self.ctx.device)

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def assign(self, array):
        a = np.ascontiguousarray(np.asarray(array), dtype=self.dtype)
        if a.shape != self.shape:
            raise ValueError(f"shape {a.shape} does not match field shape {self.shape}")
        check(load().xlbhip_field_upload(self.handle, a.ctypes.data, a.nbytes))
        return self

    def fill(self, value):
        check(load().xlbhip_field_fill(self.handle, float(value)))
        return self

    def copy_from(self, other):
        check(load().xlbhip_field_copy(self.handle, other.handle))
        return self

    @property
    def plane_dtype(self):
        """dtype of one device x-plane: the field's dtype, or uint32 bit-sets for the bit-packed missing_mask (population 0)"""
        return np.dtype(np.uint32) if self.dtype_code == MISSING else np.dtype(self.dtype)

    def get_plane(self, population, storage_plane):
        """One x-plane (ny, nz) of one population; storage_plane counts the left ghosts (if any) from 0."""
        out = np.empty(self._s3[1:], dtype=self.plane_dtype)
        check(load().xlbhip_field_plane_download(self.handle, int(population), int(storage_plane), out.ctypes.data, out.nbytes))
        return out

    def set_plane(self, population, storage_plane, array):
        a = np.ascontiguousarray(array, dtype=self.plane_dtype)
        assert a.shape == tuple(self._s3[1:])
        check(load().xlbhip_field_plane_upload(self.handle, int(population), int(storage_plane), a.ctypes.data, a.nbytes))

    def gather(self, cells):
        """Rows of the field at interior linear cell indices ((x * ny + y) * nz + z): an (n, cardinality) array."""
        k = np.ascontiguousarray(cells, dtype=np.uint32)
        out = np.empty((k.shape[0], self.cardinality), dtype=self.dtype)
        check(load().xlbhip_field_gather(self.handle, int(k.shape[0]), k.ctypes.data, out.ctypes.data, out.nbytes))
        return out

    def copy_kernel_from(self, other, bytes_per_lane=16):
        check(load().xlbhip_field_copy_kernel(self.handle, other.handle, int(bytes_per_lane)))
        return self

    def copy_tiles_from(self, other):
        """The copy with the two-step kernel's launch shape (bench.py's second yardstick)."""
        check(load().xlbhip_field_copy_tiles(self.handle, other.handle))
        return self

    def info(self):
        card, nx, ny, nz, dt, halo = _i(), _i(), _i(), _i(), _i(), _i()
        ps, ptr = C.c_uint64(), _p()
        check(load().xlbhip_field_info(self.handle, C.byref(card), C.byref(nx), C.byref(ny), C.byref(nz), C.byref(dt), C.byref(halo),
                                       C.byref(ps), C.byref(ptr)))
        return {"cardinality": card.value, "nx": nx.value, "ny": ny.value, "nz": nz.value, "dtype": dt.value, "halo": halo.value,
                "plane_stride": ps.value, "device_ptr": ptr.value}

    def free(self):
        if self._hook is not None:
            hook, self._hook = self._hook, None
            try:
                hook(self)
            except Exception:
                pass
        if self._h:
            load().xlbhip_field_destroy(self._h)
            self._h = _p()

    def __del__(self):
        try:
            self._hook = None
            if self._h:
                load().xlbhip_field_destroy(self._h)
                self._h = _p()
        except Exception:
            pass

    def __repr__(self):
        return f"HipField(shape={self.shape}, dtype={self.dtype}, halo={self.halo})"


def make_bc_desc(bc_id, kind, values):
    d = BcDesc()
    d.id = int(bc_id)
    d.kind = int(kind)
    v = np.zeros(27, np.float64)
    if values is not None:
        values = np.asarray(values, np.float64).ravel()
        v[: values.size] = values
    for i in range(27):
        d.values[i] = v[i]
    return d


def _h(field):
    """handle of an optional mask argument.  A mask that was exported as a writable zero-copy alias (DLPack / CUDA array
    interface) may have been edited by the consumer without any C-ABI call: it counts as modified every time it is used."""
    if field is None:
        return None
    if field._pinned:
        field.touch()
    return field.handle


def _hf(field):
    """handle of a population field argument: one that was exported as a writable zero-copy alias may have been written by its
    consumer, so what the library caches on its contents (the two-step kernel's strip buffer) is dropped first."""
    if field._pinned:
        field.touch()
    return field.handle


class Stepper:
    """Native stepper object (xlbhip_stepper_create)."""

    def __init__(self, ctx, lattice_id, collision_id, compute_code, store_code, bc_descs):
        self.ctx = ctx
        n = len(bc_descs)
        arr = (BcDesc * max(n, 1))(*bc_descs)
        self._h = _p()
        check(load().xlbhip_stepper_create(ctx.handle, lattice_id, collision_id, compute_code, store_code, n, arr, C.byref(self._h)))

    def set_force(self, force3):
        if force3 is None:
            check(load().xlbhip_stepper_set_force(self._h, None))
        else:
            arr = (C.c_double * 3)(*[float(x) for x in force3])
            check(load().xlbhip_stepper_set_force(self._h, C.cast(arr, C.c_void_p)))

    def set_bc_profile(self, bc_id, storage_cells, values):
        """Per-cell prescribed values of a Zou-He / Regularized BC: storage cell indices (uint32) and (n, 3) values."""
        k = np.ascontiguousarray(storage_cells, dtype=np.uint32)
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1, 3)
        assert k.shape[0] == v.shape[0]
        check(load().xlbhip_stepper_set_bc_profile(self._h, int(bc_id), int(k.shape[0]), k.ctypes.data, v.ctypes.data))

    def set_bc_distances(self, storage_cells, weights):
        """Wall-distance weights of HybridBC cells: storage cell indices (uint32) and (n, q) float32 weights."""
        k = np.ascontiguousarray(storage_cells, dtype=np.uint32)
        w = np.ascontiguousarray(weights, dtype=np.float32)
        assert w.ndim == 2 and k.shape[0] == w.shape[0]
        check(load().xlbhip_stepper_set_bc_distances(self._h, int(k.shape[0]), k.ctypes.data, w.ctypes.data))

    def momentum_transfer(self, bc_id, f_0, bc_mask, missing_mask):
        """Force (3,) on the solid behind a HybridBC / profile wall of this stepper (its distance and velocity tables)."""
        out = (C.c_double * 3)()
        check(load().xlbhip_stepper_momentum_transfer(self._h, int(bc_id), f_0.handle, _h(bc_mask), _h(missing_mask), out))
        return np.array(out[:], dtype=np.float64)

    def set_smagorinsky(self, coef):
        check(load().xlbhip_stepper_set_smagorinsky(self._h, float(coef)))

    def step(self, f_src, f_dst, bc_mask, missing_mask, omega, timestep):
        check(load().xlbhip_step(self._h, _hf(f_src), _hf(f_dst), _h(bc_mask), _h(missing_mask), float(omega), int(timestep)))

    def step2_eligible(self, f_src, f_dst, bc_mask, missing_mask):
        return bool(load().xlbhip_step2_eligible(self._h, f_src.handle, f_dst.handle, _h(bc_mask), _h(missing_mask)))

    def step2(self, f_src, f_dst, bc_mask, missing_mask, omega, timestep):
        """Two steps in one pass: f(t) in f_src -> f(t+2) in f_dst."""
        check(load().xlbhip_step2(self._h, _hf(f_src), _hf(f_dst), _h(bc_mask), _h(missing_mask), float(omega), int(timestep)))

    def run(self, f_a, f_b, bc_mask, missing_mask, omega, first_timestep, n_steps):
        """n steps; returns True when the result is in f_b (every pair of steps fused where the kernel exists)."""
        where = C.c_int()
        check(load().xlbhip_run_any(self._h, _hf(f_a), _hf(f_b), _h(bc_mask), _h(missing_mask), float(omega), int(first_timestep),
                                    int(n_steps), C.byref(where)))
        return bool(where.value)

    def run_timed(self, f_a, f_b, bc_mask, missing_mask, omega, first_timestep, n_steps):
        """As run(); returns (result is in f_b, device milliseconds)."""
        ms, where = C.c_float(), C.c_int()
        check(load().xlbhip_run_timed(self._h, _hf(f_a), _hf(f_b), _h(bc_mask), _h(missing_mask), float(omega), int(first_timestep),
                                      int(n_steps), C.byref(ms), C.byref(where)))
        return bool(where.value), ms.value

    def free(self):
        if self._h:
            load().xlbhip_stepper_destroy(self._h)
            self._h = _p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def build_masks(ctx, lattice_id, bc_ids, tag_lists, solid_lists, global_shape3, x_offset, bc_mask, missing_mask):
    """tag_lists / solid_lists: per BC an int32 (3, n) array (or None)."""
    n = len(bc_ids)
    ids = np.asarray(bc_ids, np.int32)
    keep = []
    tag_ptrs = (C.c_void_p * max(n, 1))()
    sol_ptrs = (C.c_void_p * max(n, 1))()
    tag_cnt = np.zeros(max(n, 1), np.int64)
    sol_cnt = np.zeros(max(n, 1), np.int64)
    for i in range(n):
        t = np.ascontiguousarray(tag_lists[i], dtype=np.int32)
        assert t.ndim == 2 and t.shape[0] == 3
        keep.append(t)
        tag_ptrs[i] = t.ctypes.data
        tag_cnt[i] = t.shape[1]
        s = solid_lists[i]
        if s is not None and np.size(s) > 0:
            s = np.ascontiguousarray(s, dtype=np.int32)
            assert s.ndim == 2 and s.shape[0] == 3
            keep.append(s)
            sol_ptrs[i] = s.ctypes.data
            sol_cnt[i] = s.shape[1]
    gs = np.asarray(global_shape3, np.int32)
    check(load().xlbhip_build_masks(ctx.handle, lattice_id, n, ids.ctypes.data, C.cast(tag_ptrs, C.c_void_p), tag_cnt.ctypes.data,
                                    C.cast(sol_ptrs, C.c_void_p), sol_cnt.ctypes.data, gs.ctypes.data, int(x_offset), bc_mask.handle,
                                    missing_mask.handle))
    del keep
