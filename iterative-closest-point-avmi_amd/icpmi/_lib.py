"""ctypes binding of libicpmi.so (include/icpmi.h).

There is no CPU fallback: if the shared library is missing or a call fails the
caller gets an exception, never a silently different code path.
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "lib", "libicpmi.so")

OK = 0
ST_CONVERGED, ST_MAXITER, ST_FEW_INLIERS, ST_EMPTY = 1, 2, 3, 4
RES_DOUBLES, RES_R, RES_T, RES_ERR, RES_DELTA, RES_ITERS, RES_STATUS = 16, 0, 9, 12, 13, 14, 15
POINT_TO_POINT, POINT_TO_LINE = 0, 1


class IcpParams(C.Structure):
    _fields_ = [("error_threshold", C.c_double), ("max_corr_dist", C.c_double),
                ("max_iterations", C.c_int32), ("method", C.c_int32),
                ("has_init", C.c_int32), ("dim", C.c_int32)]


class IcpmiError(RuntimeError):
    pass


def build(verbose=False):
    """Compile the HIP sources for gfx950 into lib/libicpmi.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4", "all"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise IcpmiError("building libicpmi.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout)
    return LIB_PATH


_SIGS = {
    # name: (restype, argtypes)
    "icpmi_version": (C.c_char_p, []),
    "icpmi_strerror": (C.c_char_p, [C.c_int]),
    "icpmi_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "icpmi_shutdown": (C.c_int, []),
    "icpmi_runtime_check": (C.c_int, [C.c_char_p, C.c_size_t]),
    "icpmi_voxel_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "icpmi_voxel_downsample_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "icpmi_nn_batch": (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                 C.c_int32, C.c_void_p]),
    "icpmi_normals_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "icpmi_normals_2d_batch": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "icpmi_p2l_solve_2d": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "icpmi_icp_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "icpmi_icp_batch": (C.c_int, [C.c_void_p] * 7 + [C.c_int32] * 4 + [C.POINTER(IcpParams), C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "icpmi_prepared_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "icpmi_prepare_targets": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 5 + [C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "icpmi_prepare_targets_ex": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 5 + [C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_int32, C.c_void_p]),
    "icpmi_nn_prepared_batch": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int32, C.c_void_p]),
    "icpmi_rotation_scores": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double,
                                        C.c_double, C.c_void_p, C.c_void_p]),
    "icpmi_rotation_search_workspace_bytes": (C.c_size_t, [C.c_int32] * 4),
    "icpmi_rotation_search": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p]),
    "icpmi_rotation_refine_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "icpmi_rotation_refine": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double,
                                        C.c_double, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "icpmi_rotation_search_batch_workspace_bytes": (C.c_size_t, [C.c_int32] * 3),
    "icpmi_rotation_search_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                              C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "icpmi_world_to_grid": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "icpmi_bresenham_cells": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "icpmi_grid_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "icpmi_grid_update_scans": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                          C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int32, C.c_void_p]),
    "icpmi_grid_update_scans_band": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                               C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double,
                                               C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int32, C.c_int32,
                                               C.c_int32, C.c_void_p]),
    "icpmi_grid_update_scans_box": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                              C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double,
                                              C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int32, C.c_int32,
                                              C.c_int32, C.c_void_p, C.c_void_p]),
    "icpmi_pose_graph_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
    "icpmi_pose_graph_optimize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.c_void_p]),
    "icpmi_pose_graph_error": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """The loaded library; raises IcpmiError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IcpmiError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(or `make -C iterative-closest-point-avmi_amd/csrc`). There is no CPU fallback.")
        # torch first: it ships its own HIP runtime (libamdhip64 inside the wheel), and the process must hold ONE — the
        # streams and allocations torch hands us have to belong to the runtime our launches go through.  Loaded before
        # torch, libicpmi.so would bring in /opt/rocm's copy and torch its own beside it: every launch then fails with
        # a HIP runtime error (seen: build() followed by smoke() in one process).
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)          # AttributeError here means the .so is stale
            f.restype, f.argtypes = res, args
        _lib = L
        why = runtime_problem()
        if why:
            _lib = None
            raise IcpmiError(why)
    return _lib


def runtime_problem():
    """'' or what icpmi_runtime_check found: two copies of libamdhip64 mapped into this process (every launch would fail)."""
    if _lib is None:
        return ""
    buf = C.create_string_buffer(1024)
    return buf.value.decode() if _lib.icpmi_runtime_check(buf, len(buf)) != OK else ""


ERR_HIP = -3


def check(code, what):
    if code != OK:
        hint = runtime_problem() if code == ERR_HIP else ""
        raise IcpmiError(f"{what}: {lib().icpmi_strerror(code).decode()} ({code})" + (f" [{hint}]" if hint else ""))


def set_option(name, value):
    """icpmi_set_option: change one of the library's ICPMI_* switches after it has read the environment (None unsets)."""
    check(lib().icpmi_set_option(name.encode(), None if value is None else str(value).encode()), f"set_option({name})")


def shutdown():
    """icpmi_shutdown: destroy the side streams and events the library made (made again on demand)."""
    if _lib is not None:
        check(_lib.icpmi_shutdown(), "shutdown")
