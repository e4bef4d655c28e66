"""ctypes binding of libsmcounter_hip.so (include/smcounter_hip.h).  Fails loudly: there is no
CPU fallback in the product path."""
from __future__ import annotations

import ctypes
import os

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMC_HIP_LIB") or os.path.join(HERE, "libsmcounter_hip.so")   # (SMC_HIP_LIB: another build of the same ABI - same-box A/B runs)

SYMBOLS = ("smc_abi_version", "smc_last_error", "smc_row_size", "smc_locus_size", "smc_device_count",
           "smc_create", "smc_destroy", "smc_plan_create", "smc_plan_create_dev", "smc_plan_create_dev_spec", "smc_plan_spec_ok", "smc_plan_spec_counts", "smc_plan_hint_reset", "smc_philox_marks", "smc_philox4x32_10_host", "smc_plan_destroy", "smc_plan_info",
           "smc_plan_run", "smc_plan_run_words", "smc_plan_run_words16", "smc_pack_words", "smc_plan_set_timing", "smc_plan_kernel_ms", "smc_call_batch_host", "smc_event_create", "smc_event_record",
           "smc_event_elapsed_ms", "smc_event_destroy", "smc_class_table", "smc_wire_row_size", "smc_pack_rows", "smc_unpack_rows",
           "smc_build_planes", "smc_build_planes_w16", "smc_build_max_depth", "smc_build_set_timing", "smc_build_kernel_ms", "smc_mem_alloc", "smc_mem_alloc_best", "smc_mem_write_probe", "smc_mem_free", "smc_mem_h2d", "smc_mem_d2h",
           "smc_mem_alloc_host", "smc_mem_free_host", "smc_pool_trim",
           "smc_device_sync")


class SmcError(RuntimeError):
    pass


def exp_env(name: str, default=None):
    """An EXPERIMENT switch of the Python host (measurements, tests): read only when SMC_EXPERIMENTAL is set to anything but 0, as
    the native libraries read theirs - a stray variable in a production environment changes nothing."""
    on = os.environ.get("SMC_EXPERIMENTAL", "0") not in ("", "0")
    return os.environ.get(name, default) if on else default


_LIB = None


def load(with_torch: bool = True):
    """dlopen + bind the C ABI.  `with_torch=False` skips importing PyTorch first (about a second of start-up): only
    for processes that will never import it afterwards - the single-process command line."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        # not built yet in this tree: build it now (hipcc, about two minutes); without hipcc this fails loudly -
        # there is no CPU fallback
        try:
            from . import build
            build.build_hip()
        except Exception as e:
            raise SmcError("HIP library missing: %s and it could not be built (%s) - run `python -c 'import "
                           "__graft_entry__ as g; g.build()'`; there is no CPU fallback" % (LIB_PATH, e))
    # PyTorch-ROCm ships its own HIP runtime; it must be the first one mapped into the process, or
    # torch later finds "No HIP GPUs" behind the system libamdhip64 this library would pull in.
    if with_torch:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.smc_abi_version.restype = ctypes.c_int
    L.smc_last_error.restype = ctypes.c_char_p
    L.smc_row_size.restype = ctypes.c_int
    L.smc_locus_size.restype = ctypes.c_int
    L.smc_device_count.restype = ctypes.c_int
    L.smc_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.smc_destroy.argtypes = [vp]
    L.smc_destroy.restype = None
    L.smc_plan_create.argtypes = [vp, vp, i64, ctypes.POINTER(vp)]
    L.smc_plan_create_dev.argtypes = [vp, vp, i64, vp, ctypes.POINTER(vp)]
    L.smc_plan_create_dev_spec.argtypes = [vp, ctypes.POINTER(abi.SmcParams), vp, i64, vp, ctypes.POINTER(vp)]
    L.smc_plan_spec_ok.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
    L.smc_plan_spec_counts.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.smc_plan_hint_reset.argtypes = [vp]
    L.smc_philox_marks.argtypes = [vp, ctypes.POINTER(abi.SmcParams), vp, i64, vp, vp, ctypes.c_int, vp, vp, vp, ctypes.c_uint64, vp, vp]
    L.smc_philox4x32_10_host.argtypes = [vp, vp, vp]
    L.smc_philox4x32_10_host.restype = None
    L.smc_plan_destroy.argtypes = [vp]
    L.smc_plan_destroy.restype = None
    L.smc_plan_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i64)]
    L.smc_plan_run.argtypes = [vp, ctypes.POINTER(abi.SmcParams), vp, vp, vp, vp, vp, vp, vp]
    L.smc_plan_run_words.argtypes = [vp, ctypes.POINTER(abi.SmcParams), vp, vp, vp, vp]
    L.smc_plan_run_words16.argtypes = [vp, ctypes.POINTER(abi.SmcParams), vp, vp, vp, vp]
    L.smc_pack_words.argtypes = [vp, vp, vp, vp, vp]
    L.smc_plan_set_timing.argtypes = [vp, ctypes.c_int]
    L.smc_plan_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(i32), ctypes.POINTER(i64),
                                     ctypes.POINTER(i64)]
    L.smc_call_batch_host.argtypes = [vp, ctypes.POINTER(abi.SmcParams), vp, i64, vp, vp, vp, vp, i64, vp, i64, vp]
    L.smc_wire_row_size.restype = ctypes.c_int
    L.smc_pack_rows.argtypes = [vp, vp, i64, vp, vp]
    L.smc_unpack_rows.argtypes = [vp, i64, vp]
    L.smc_mem_alloc.argtypes = [vp, i64, ctypes.POINTER(vp)]
    L.smc_mem_alloc_best.argtypes = [vp, i64, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_float)]
    L.smc_mem_write_probe.argtypes = [vp, vp, i64, ctypes.POINTER(ctypes.c_float)]
    L.smc_mem_free.argtypes = [vp, vp]
    L.smc_mem_free.restype = None
    L.smc_mem_alloc_host.argtypes = [vp, i64, ctypes.POINTER(vp)]
    L.smc_mem_free_host.argtypes = [vp, vp]
    L.smc_mem_free_host.restype = None
    L.smc_mem_h2d.argtypes = [vp, vp, vp, i64]
    L.smc_mem_d2h.argtypes = [vp, vp, vp, i64]
    L.smc_device_sync.argtypes = [vp]
    L.smc_pool_trim.argtypes = [vp]
    L.smc_build_max_depth.restype = ctypes.c_int
    L.smc_build_set_timing.argtypes = [vp, ctypes.c_int]
    L.smc_build_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)]
    L.smc_build_planes.argtypes = [vp, ctypes.POINTER(abi.SmcParams), ctypes.POINTER(abi.SmcBuildIn), ctypes.c_uint32,
                                   ctypes.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint32, vp, vp]
    L.smc_build_planes_w16.argtypes = [vp, ctypes.POINTER(abi.SmcParams), ctypes.POINTER(abi.SmcBuildIn), ctypes.c_uint32,
                                       ctypes.c_uint32, vp, vp, vp, vp, vp, vp, ctypes.c_uint32, vp, vp]
    L.smc_event_create.argtypes = [ctypes.POINTER(vp)]
    L.smc_event_record.argtypes = [vp, vp]
    L.smc_event_elapsed_ms.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.smc_event_destroy.argtypes = [vp]
    L.smc_event_destroy.restype = None
    if L.smc_abi_version() != 8:
        raise SmcError("ABI version mismatch")
    if L.smc_row_size() != abi.ROW_DTYPE.itemsize or L.smc_locus_size() != 32 or L.smc_wire_row_size() != abi.WIRE_DTYPE.itemsize:
        raise SmcError("struct layout mismatch between include/smcounter_hip.h and smcounter_amd/abi.py")
    _LIB = L
    return L


def check(rc: int, what: str):
    if rc != 0:
        raise SmcError("%s failed (%d): %s" % (what, rc, load().smc_last_error().decode()))
