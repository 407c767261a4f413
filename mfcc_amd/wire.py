"""Serial wire format of the FPGA's coefficient stream and the receiver's power gate -- the data
formats on the far side of the hot path.  Thin wrappers over the host-only
C-ABI functions of libmfcc_hip.so (include/mfcc_hip.h); names follow the reference:

  mfcc/misc/magic.py:9-41        MagicInserter: 0xa55a in front of every frame's coefficients
  software/serial.c:89-122       expect_magic: byte-wise resynchronisation, big endian
  software/cepstrum.c:15-71      cepstrum_get_column: magic, then n_cep big-endian int16
  software/cepstrum.c:161-183    cepstrum_eval_power: sum of c0^2 over the middle third >= 1e8
"""
import ctypes as C

import numpy as np

from . import _lib

MAGIC = 0xA55A                 # magic.py:10, serial.c:13-14
POWER_THRESHOLD = 100000000    # cepstrum.c:13


def pack_columns(cep) -> bytes:
    """int16 ``(frames, n_cep)`` (what ``MFCC.process_fixed`` returns) -> the UART byte stream."""
    cep = np.ascontiguousarray(cep, dtype=np.int16)
    if cep.ndim != 2:
        raise ValueError("cep must be (frames, n_cep)")
    lib = _lib.load()
    nf, nc = cep.shape
    out = np.empty(lib.mfcc_hip_serial_packed_size(nf, nc), dtype=np.uint8)
    _lib.check(lib.mfcc_hip_serial_pack(cep.ctypes.data_as(C.c_void_p), nf, nc,
                                        out.ctypes.data_as(C.c_void_p), out.size), "serial_pack")
    return out.tobytes()


def unpack_columns(data, n_cep, max_frames=None):
    """Byte stream -> (int16 ``(frames, n_cep)``, bytes consumed); resynchronises on the magic exactly
    like ``expect_magic`` + ``cepstrum_get_column``.  Bytes after the last whole column are left for
    the next call (``data[consumed:]``)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    lib = _lib.load()
    cap = buf.size // (2 * (n_cep + 1)) + 1 if max_frames is None else int(max_frames)
    cep = np.empty((cap, n_cep), dtype=np.int16)
    nf, used = C.c_size_t(0), C.c_size_t(0)
    _lib.check(lib.mfcc_hip_serial_unpack(buf.ctypes.data_as(C.c_void_p), buf.size, int(n_cep),
                                          cep.ctypes.data_as(C.c_void_p), cap, C.byref(nf), C.byref(used)),
               "serial_unpack")
    return cep[:nf.value].copy(), int(used.value)


def cepstrum_eval_power(window, head=0):
    """``cepstrum_eval_power`` on an int16 window ``(frames, n_cep)`` stored as the reference's circular
    buffer with its oldest element at flat index ``head``.  Returns ``(power, power >= 1e8)``."""
    window = np.ascontiguousarray(window, dtype=np.int16)
    if window.ndim != 2:
        raise ValueError("window must be (frames, n_cep)")
    lib = _lib.load()
    p = C.c_longlong(0)
    rc = lib.mfcc_hip_eval_power(window.ctypes.data_as(C.c_void_p), window.shape[1], window.shape[0],
                                 int(head), C.byref(p))
    if rc < 0:
        _lib.check(rc, "eval_power")
    return int(p.value), bool(rc)
