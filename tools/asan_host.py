#!/usr/bin/env python3
"""Host-side sanitizer run (SURVEY.md section 5 "sanitizers"), on the CPU — GPU AddressSanitizer is not
available on this pool.  `make asan` builds build/asan/libxlbhip_asan.so: api.hip and comm.cpp compiled with
-fsanitize=address,undefined for the HOST side only (-fno-gpu-sanitize), linked with the ordinary kernel objects.  This script
loads it (the sanitizer runtime preloaded) and walks the argument paths of the C ABI that a machine without a GPU can reach:
NULL handles, bad enums, bad ranks / tokens, the error channel's formatting, xlbhip_create's failure path, the lattice tables.
Any sanitizer report aborts the process (exit code != 0).

    python tools/asan_host.py            # re-executes itself with LD_PRELOAD set
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "build", "asan", "libxlbhip_asan.so")


def runtime():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.exists(out) else None


def main():
    if not os.path.exists(LIB):
        sys.exit(f"{LIB} not built: run `make asan` (about 12 minutes: api.hip's device code is compiled again)")
    if os.environ.get("XLB_ASAN_CHILD") != "1":
        rt = runtime()
        if rt is None:
            sys.exit("the sanitizer runtime was not found")
        env = dict(os.environ, LD_PRELOAD=rt, XLB_ASAN_CHILD="1", XLBHIP_LIB=LIB,
                   ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
        sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__)], env=env).returncode)
    sys.path.insert(0, ROOT)
    import numpy as np

    from xlb_amd import _lib

    lib = _lib.load()
    err = lambda: lib.xlbhip_last_error().decode()  # noqa: E731
    n_calls = 0

    def expect_fail(rc, what):
        nonlocal n_calls
        n_calls += 1
        assert rc != 0 and err(), f"{what}: expected an error, got rc={rc}"

    # lattice tables: the one compute-free call that succeeds without a device
    for lid in (_lib.D2Q9, _lib.D3Q19, _lib.D3Q27):
        d, q, c, w, opp, cc = _lib.lattice_info(lid)
        assert abs(float(w.sum()) - 1.0) < 1e-12 and q in (9, 19, 27)
        n_calls += 1
    d, q = C.c_int(), C.c_int()
    buf = np.zeros(3 * 27, np.int32)
    expect_fail(lib.xlbhip_lattice_info(7, C.byref(d), C.byref(q), buf.ctypes.data, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data), "bad lattice id")
    expect_fail(lib.xlbhip_lattice_info(1, None, None, None, None, None, None), "null outputs")
    # context creation on a machine without a device / with a bad index: the failure path (error string formatting)
    h = C.c_void_p()
    rc = lib.xlbhip_create(0, C.byref(h))
    have_gpu = rc == 0
    if not have_gpu:
        assert err()
    expect_fail(lib.xlbhip_create(0, None), "null out")
    expect_fail(lib.xlbhip_create(-3, C.byref(C.c_void_p())), "negative device")
    # every entry point with NULL handles: must return an error, never dereference
    null = None
    one = C.c_int64(1)
    dbl = (C.c_double * 3)()
    calls = {
        "xlbhip_sync": (null,), "xlbhip_set_option": (null, b"vec", 1), "xlbhip_get_option": (null, b"vec", C.byref(one)),
        "xlbhip_device_info": (null, None, 0, None, None), "xlbhip_field_create": (null, 1, 1, 1, 1, 1, 0, 0.0, C.byref(C.c_void_p())),
        "xlbhip_field_fill": (null, 0.0), "xlbhip_field_copy": (null, null), "xlbhip_field_copy_kernel": (null, null, 16),
        "xlbhip_field_upload": (null, null, 0), "xlbhip_field_download": (null, null, 0), "xlbhip_field_plane_download": (null, 0, 0, null, 0),
        "xlbhip_field_plane_upload": (null, 0, 0, null, 0), "xlbhip_field_info": (null, None, None, None, None, None, None, None, None),
        "xlbhip_field_touch": (null,), "xlbhip_mem_info": (null, None, None), "xlbhip_stream": (null, 1, null, null),
        "xlbhip_equilibrium": (null, 1, 1, null, null, null), "xlbhip_macroscopic": (null, 1, 1, null, null, null),
        "xlbhip_second_moment": (null, 1, 1, null, null), "xlbhip_vorticity": (null, null, null, null, null),
        "xlbhip_q_criterion": (null, null, null, null, null), "xlbhip_grid_to_point": (null, null, 0, null, null),
        "xlbhip_collide": (null, 1, 0, 1, null, null, null, 1.0), "xlbhip_apply_bc": (null, 1, 1, None, null, null, null, null),
        "xlbhip_momentum_transfer": (null, 1, 1, None, null, null, null, dbl),
        "xlbhip_build_masks": (null, 1, 0, null, null, null, null, null, null, 0, null, null),
        "xlbhip_mesh_mask": (null, 1, 1, 1, 0, null, 0, null, null, null), "xlbhip_field_gather": (null, 0, null, null, 0),
        "xlbhip_stepper_create": (null, 1, 0, 1, 1, 0, None, C.byref(C.c_void_p())),
        "xlbhip_step": (null, null, null, null, null, 1.0, 0), "xlbhip_run": (null, null, null, null, null, 1.0, 0, 1),
        "xlbhip_run_any": (null, null, null, null, null, 1.0, 0, 1, C.byref(C.c_int())), "xlbhip_step2": (null, null, null, null, null, 1.0, 0),
        "xlbhip_run_timed": (null, null, null, null, null, 1.0, 0, 1, C.byref(C.c_float()), None),
        "xlbhip_stepper_set_bc_profile": (null, 1, 0, null, null), "xlbhip_stepper_set_bc_distances": (null, 0, null, null),
        "xlbhip_stepper_momentum_transfer": (null, 1, null, null, null, dbl),
        "xlbhip_comm_init": (null, 0, 1, null, 1), "xlbhip_comm_init_ipc": (null, 0, 2, b"tok", 1), "xlbhip_comm_stats": (null, None, None, 0),
        "xlbhip_halo_exchange": (null, 1, null), "xlbhip_halo_exchange_wide": (null, 1, null), "xlbhip_comm_unique_id": (null,),
    }
    for name, args in calls.items():
        expect_fail(getattr(lib, name)(*args), name)
    assert lib.xlbhip_step2_eligible(null, null, null, null, null) == 0
    # destroy / free of NULL are no-ops
    for name in ("xlbhip_destroy", "xlbhip_field_destroy", "xlbhip_stepper_destroy", "xlbhip_comm_destroy"):
        assert getattr(lib, name)(null) == 0
        n_calls += 1
    # a very long message through the error channel (vsnprintf truncation)
    lib.xlbhip_comm_init_ipc(null, 0, 2, b"x" * 5000, 1)
    assert len(err()) < 1100
    print(f"ASAN_HOST_OK: {n_calls} calls through build/asan/libxlbhip_asan.so, no sanitizer report (device present: {have_gpu})")


if __name__ == "__main__":
    main()
