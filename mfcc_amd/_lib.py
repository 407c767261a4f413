"""ctypes binding of libmfcc_hip.so (include/mfcc_hip.h).  No compute happens in Python.

The library is built in-tree (``mfcc_amd/libmfcc_hip.so``) by ``__graft_entry__.build()`` /
``make -C mfcc_amd/csrc``.  If it is missing this module raises -- there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MFCC_HIP_LIB: diagnostic override for A/B runs of experimental builds of the same ABI (tools/ab.sh)
LIB_PATH = os.environ.get("MFCC_HIP_LIB") or os.path.join(_HERE, "libmfcc_hip.so")

ABI_VERSION = 2

# enum mfcc_hip_error
SUCCESS = 0
ERROR_INVALID_PARAM = -101
ERROR_NOT_FOUND = -102
ERROR_NO_MEM = -103
ERROR_BUSY = -104
ERROR_UNSUPPORTED = -105
ERROR_BUFFER_SMALL = -106
ERROR_IO = -107
ERROR_OTHER = -200

PAD_NOTEBOOK = 0
PAD_STREAM = 1

IMPL_AUTO = 0
IMPL_GENERIC = 1
IMPL_FUSED512 = 2

TABLE_WINDOW_F32 = 0
TABLE_MEL_POINTS_I32 = 1
TABLE_MEL_DENSE_F32 = 2
TABLE_DCT_F32 = 3
TABLE_FX_CURVE_I32 = 4
TABLE_FX_TWIDDLE_I32 = 5
TABLE_FX_MEL_DENSE_U32 = 6


class Params(C.Structure):
    """struct mfcc_hip_params"""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("nfft", C.c_int32),
        ("hop", C.c_int32),
        ("n_mel", C.c_int32),
        ("n_cep", C.c_int32),
        ("sample_rate", C.c_int32),
        ("pad_mode", C.c_int32),
        ("power_scale", C.c_float),
        ("lifter", C.c_float),
        ("device", C.c_int32),
        ("float_impl", C.c_int32),
        ("reserved", C.c_int32 * 5),
    ]


# every symbol include/mfcc_hip.h declares: name -> (restype, argtypes)
_H = C.c_void_p
_SZ = C.c_size_t
_PSZ = C.POINTER(C.c_size_t)
SYMBOLS = {
    "mfcc_hip_serial_packed_size": (C.c_size_t, [C.c_size_t, C.c_int]),
    "mfcc_hip_serial_pack": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]),
    "mfcc_hip_serial_unpack": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "mfcc_hip_eval_power": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_longlong)]),
    "mfcc_hip_process_ragged_i16": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                              C.c_void_p]),
    "mfcc_hip_process_ragged_fixed_i16": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                                    C.c_size_t, C.c_void_p]),
    "mfcc_hip_convert_wavs": (C.c_int, [_H, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_size_t, C.c_int,
                                        C.c_void_p]),
    "mfcc_hip_process_ragged_i16_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                  C.c_void_p]),
    "mfcc_hip_process_ragged_fixed_i16_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                                        C.c_size_t, C.c_void_p]),
    "mfcc_hip_abi_version": (C.c_int, []),
    "mfcc_hip_default_params": (C.c_int, [C.POINTER(Params)]),
    "mfcc_hip_create": (C.c_int, [C.POINTER(Params), C.POINTER(_H)]),
    "mfcc_hip_destroy": (None, [_H]),
    "mfcc_hip_set_stream": (C.c_int, [_H, C.c_void_p]),
    "mfcc_hip_use_own_stream": (C.c_int, [_H]),
    "mfcc_hip_synchronize": (C.c_int, [_H]),
    "mfcc_hip_num_frames": (C.c_int, [C.POINTER(Params), _SZ, _PSZ]),
    "mfcc_hip_strerror": (C.c_char_p, [C.c_int]),
    "mfcc_hip_last_hip_error": (C.c_int, [_H]),
    "mfcc_hip_get_table": (C.c_int, [C.POINTER(Params), C.c_int, C.c_void_p, _SZ, _PSZ]),
    "mfcc_hip_process_i16": (C.c_int, [_H, C.c_void_p, _SZ, _SZ, C.c_void_p, _SZ, _PSZ]),
    "mfcc_hip_process_fixed_i16": (C.c_int, [_H, C.c_void_p, _SZ, _SZ, C.c_void_p, _SZ, _PSZ]),
    "mfcc_hip_process_i16_dev": (C.c_int, [_H, C.c_void_p, _SZ, _SZ, _SZ, C.c_int, C.c_void_p, _PSZ]),
    "mfcc_hip_process_fixed_i16_dev": (C.c_int, [_H, C.c_void_p, _SZ, _SZ, _SZ, C.c_int, C.c_void_p, _PSZ]),
    "mfcc_hip_time_dev": (C.c_int, [_H, C.c_int, C.c_void_p, _SZ, _SZ, _SZ, C.c_void_p, C.c_int, C.c_int,
                                    C.POINTER(C.c_float)]),
    "mfcc_hip_kernel_name": (C.c_char_p, [_H, C.c_int]),
    "mfcc_hip_convert_wav": (C.c_int, [_H, C.c_char_p, C.c_char_p, C.c_int, _PSZ]),
    "mfcc_hip_stream_create": (C.c_int, [_H, C.c_int, C.POINTER(_H)]),
    "mfcc_hip_stream_destroy": (None, [_H]),
    "mfcc_hip_stream_reset": (C.c_int, [_H]),
    "mfcc_hip_stream_pending": (C.c_size_t, [_H]),
    "mfcc_hip_stream_max_frames": (C.c_size_t, [_H, _SZ]),
    "mfcc_hip_stream_push": (C.c_int, [_H, C.c_void_p, _SZ, C.c_void_p, _SZ, _PSZ]),
    "mfcc_hip_stream_flush": (C.c_int, [_H, C.c_void_p, _SZ, _PSZ]),
    "mfcc_hip_lift_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_double, _PSZ]),
}

_lib = None


class MfccHipError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = load().mfcc_hip_strerror(code).decode()
        super().__init__("%s: %s (%d)" % (what, msg, code) if what else "%s (%d)" % (msg, code))


def load():
    """dlopen the in-tree library and type every entry point; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C mfcc_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7; if ours (linked
    # against /opt/rocm) were loaded first the process would end up with a runtime torch did
    # not initialise.  torch is this package's plumbing for device memory and streams, so
    # load it first and let libmfcc_hip.so bind to the runtime that is already resident.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.mfcc_hip_abi_version() != ABI_VERSION:
        raise ImportError("libmfcc_hip.so ABI %d != binding ABI %d" %
                          (lib.mfcc_hip_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def kernel_source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources in csrc/: stamps the rocprofv3 summaries kept in
    profiles/ so that bench.py can tell whether a committed counter figure still belongs to the kernels it runs."""
    import glob
    import hashlib
    hsh = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hpp")) + glob.glob(os.path.join(src, "*.hip"))):
        hsh.update(os.path.basename(f).encode())
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


def check(code, what=""):
    if code != SUCCESS:
        raise MfccHipError(code, what)
