"""ctypes binding of libyacht_hip.so — the C ABI declared in include/yacht_hip.h.

There is deliberately no fallback: if the shared library cannot be loaded, or a call reports
that no HIP device is usable, a YachtHipError is raised.  The CPU restatement under oracle/ is
test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

YH_OK = 0
YH_ERR_INVALID_ARG = -1
YH_ERR_NO_DEVICE = -2
YH_ERR_HIP = -3
YH_ERR_UNSORTED = -4
YH_ERR_CAPACITY = -5
YH_ERR_OOM = -6
YH_ERR_UNSUPPORTED = -7

YH_DB_DEFAULT = 0
YH_DB_NO_INDEX = 1
YH_DB_KEEP_CSR = 2
YH_DB_PAIRWISE_ONLY = 8
YH_DB_NO_DIRECTORY = 16
YH_RUN_SLOTS = 4
YH_BATCH_SLOTS = 3
YH_LOOKUP_AUTO, YH_LOOKUP_STREAM, YH_LOOKUP_INDEXED = 0, 1, 2

_ERR_NAMES = {
    YH_ERR_INVALID_ARG: "YH_ERR_INVALID_ARG",
    YH_ERR_NO_DEVICE: "YH_ERR_NO_DEVICE",
    YH_ERR_HIP: "YH_ERR_HIP",
    YH_ERR_UNSORTED: "YH_ERR_UNSORTED",
    YH_ERR_CAPACITY: "YH_ERR_CAPACITY",
    YH_ERR_OOM: "YH_ERR_OOM",
    YH_ERR_UNSUPPORTED: "YH_ERR_UNSUPPORTED",
}


class YachtHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {message}")
        self.code = code


class DbInfo(C.Structure):
    _fields_ = [
        ("n_refs", C.c_uint64),
        ("n_hashes", C.c_uint64),
        ("max_hash", C.c_uint64),
        ("n_distinct", C.c_uint64),
        ("n_shared_distinct", C.c_uint64),
        ("n_shared_postings", C.c_uint64),
        ("device_bytes", C.c_uint64),
        ("device_id", C.c_int32),
        ("flags", C.c_uint32),
        ("stream_layout", C.c_uint32),
        ("stream_shift", C.c_uint32),
        ("stream_bytes", C.c_uint64),
        ("n_holder_sets", C.c_uint64),
        ("filter_bytes", C.c_uint64),
        ("sort_path", C.c_uint32),
        ("reserved_", C.c_uint32),
        ("n_spilled_buckets", C.c_uint64),
        ("n_spilled_pairs", C.c_uint64),
    ]


class Timing(C.Structure):
    _fields_ = [
        ("ms_overlap_kernel", C.c_float),
        ("ms_exclusive_kernels", C.c_float),
        ("ms_pairwise_kernels", C.c_float),
        ("ms_db_build", C.c_float),
        ("ms_h2d", C.c_float),
        ("ms_d2h", C.c_float),
    ]


_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/yacht_hip.h declares
SIGNATURES = {
    "yh_last_error": (C.c_char_p, []),
    "yh_abi_version": (C.c_int, []),
    "yh_pool_release": (C.c_int, [C.POINTER(C.c_uint64)]),
    "yh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "yh_alloc_stats": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "yh_db_create": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(_vp)]),
    "yh_db_create_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(_vp)]),
    "yh_csr_pack_bound": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "yh_csr_pack": (C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]),
    "yh_csr_unpack": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "yh_csr_subset": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "yh_sig_batch_pack": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]),
    "yh_db_create_packed": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(_vp)]),
    "yh_db_destroy": (C.c_int, [_vp]),
    "yh_db_get_info": (C.c_int, [_vp, C.POINTER(DbInfo)]),
    "yh_db_set_stream": (C.c_int, [_vp, _vp]),
    "yh_db_set_batch_finish_stream": (C.c_int, [_vp, _vp]),
    "yh_db_synchronize": (C.c_int, [_vp]),
    "yh_db_set_lookup": (C.c_int, [_vp, C.c_int]),
    "yh_db_lookup_choice": (C.c_int, [_vp, C.c_uint64]),
    "yh_db_get_timing": (C.c_int, [_vp, C.POINTER(Timing)]),
    "yh_overlap": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "yh_overlap_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "yh_overlap_bsearch": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "yh_overlap_bsearch_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "yh_overlap_indexed_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "yh_run_indexed_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "yh_run_batch": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp]),
    "yh_run_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint64, _vp, _vp, _vp]),
    "yh_exclusive": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp, _vp]),
    "yh_run": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "yh_run_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "yh_run_device_pipelined": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "yh_run_device_join": (C.c_int, [_vp]),
    "yh_db_set_ghosts": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp]),
    "yh_run_local_device": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64, _vp, _vp, _vp, _vp]),
    "yh_run_finish_device": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "yh_run_local_range_device": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64, _vp, _vp, _vp]),
    "yh_run_finish_range_device": (C.c_int, [_vp, C.c_int, _vp, C.c_uint32, C.c_uint64, _vp]),
    "yh_run_batch_local_range_device": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_uint32, C.c_uint64, _vp, _vp]),
    "yh_run_batch_finish_range_device": (C.c_int, [_vp, C.c_int, C.c_uint32, _vp, C.c_uint32, _vp, _vp, _vp]),
    "yh_run_batch_rows_pack_device": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_uint64, _vp]),
    "yh_run_batch_rows_unpack_device": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64, _vp, _vp]),
    "yh_run_batch_words_packed_len": (C.c_uint64, [C.c_uint64]),
    "yh_run_batch_words_pack_device": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint64]),
    "yh_run_batch_words_unpack_device": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint64, _vp, _vp]),
    "yh_run_submit": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64, _vp, _vp, _vp]),
    "yh_run_wait": (C.c_int, [_vp, C.c_int]),
    "yh_sample_pack_bound": (C.c_uint64, [C.c_uint64]),
    "yh_sample_pack": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "yh_sample_pack_threads": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]),
    "yh_sample_unpack": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "yh_run_submit_packed": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64, _vp, C.c_uint64]),
    "yh_run_submit_rows": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64, _vp, C.c_uint64]),
    "yh_run_wait_rows": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint64)]),
    "yh_run_rows_device": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp]),
    "yh_host_alloc": (C.c_int, [C.POINTER(_vp), C.c_uint64]),
    "yh_host_free": (C.c_int, [_vp]),
    "yh_db_nshared_device": (C.c_int, [_vp, _vp]),
    "yh_pairwise": (C.c_int, [_vp, C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, _vp, _vp, _vp,
                              C.POINTER(C.c_uint64)]),
    "yh_index_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "yh_pairwise_row_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "yh_sketch_dna": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, _vp,
                                C.POINTER(C.c_uint64)]),
    "yh_sketch_dna_device": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, _vp, _vp, _vp]),
    "yh_hyp_test": (C.c_int, [C.c_uint64, _vp, _vp, C.c_int, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp]),
    "yh_train_select": (C.c_int, [_vp, C.c_uint64, _vp, _vp, C.c_uint64, _vp, C.POINTER(C.c_uint64)]),
    "yh_sig_batch_read": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint64, C.c_int, C.POINTER(_vp)]),
    "yh_sig_batch_status": (C.c_int, [_vp, _vp]),
    "yh_sig_batch_sizes": (C.c_int, [_vp, _vp]),
    "yh_sig_batch_values": (C.c_int, [_vp, _vp]),
    "yh_sig_batch_destroy": (C.c_int, [_vp]),
    "yh_gunzip_files": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint64, C.c_int, _vp]),
    "yh_sig_meta_read": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint64, C.c_int, C.c_int, C.POINTER(_vp)]),
    "yh_sig_meta_read_keep": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint64, C.c_int, C.c_int, C.POINTER(_vp)]),
    "yh_sig_meta_take_batch": (C.c_int, [_vp, C.POINTER(_vp)]),
    "yh_sig_meta_get": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "yh_sig_meta_names": (C.c_int, [_vp, _vp]),
    "yh_sig_meta_destroy": (C.c_int, [_vp]),
    "yh_zip_sig_ingest": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp)]),
    "yh_zip_extract_start": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "yh_zip_extract_wait": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "yh_sig_meta_count": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "yh_sig_meta_paths": (C.c_int, [_vp, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return os.environ.get("YACHT_HIP_LIB", _build.LIB_PATH)


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7; if
    libyacht_hip.so pulled in /opt/rocm's copy first, a later `import torch` would bring up a
    second runtime that finds no device.  So when torch is installed, its copy is loaded first
    (by path, globally): the SONAME then satisfies libyacht_hip.so's dependency too, whichever
    of the two packages the caller touches first."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> C.CDLL:
    """Load (once) and type the shared library.  Raises if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise YachtHipError(
            YH_ERR_NO_DEVICE,
            f"{path} is missing: build it with `python -m yacht_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback.",
        )
    _preload_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = the .so does not match the header
        fn.restype = res
        fn.argtypes = args
    if lib.yh_abi_version() != 8:
        raise YachtHipError(YH_ERR_INVALID_ARG, f"ABI version mismatch in {path}")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != YH_OK:
        msg = load().yh_last_error()
        raise YachtHipError(rc, msg.decode("utf-8", "replace") if msg else "")


def device_count() -> int:
    n = C.c_int(0)
    rc = load().yh_device_count(C.byref(n))
    return n.value if rc == YH_OK else 0


def pool_release() -> int:
    """Give every idle block of the library's device buffer cache back to the driver (yh_pool_release): bytes released."""
    b = C.c_uint64(0)
    check(load().yh_pool_release(C.byref(b)))
    return int(b.value)


def alloc_stats() -> dict:
    """This process's trips to the driver for device memory (yh_alloc_stats): count, host ms inside hipMalloc, idle cached bytes."""
    n, ms, idle = C.c_uint64(0), C.c_double(0.0), C.c_uint64(0)
    check(load().yh_alloc_stats(C.byref(n), C.byref(ms), C.byref(idle)))
    return {"driver_allocs": int(n.value), "ms_in_driver": round(float(ms.value), 1), "bytes_idle": int(idle.value)}
