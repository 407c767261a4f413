"""Host-side mirror of the reference's interface for the MFCC hot path.

The reference exposes this path two ways and both are mirrored here, names and argument
meaning kept:

* ``MFCC(width=16, nfft=512, samplerate=16e3, nfilters=16, nceptrums=16)`` -- the nMigen core's
  constructor, ``mfcc/core/mfcc.py:20-21`` (stream in: ``sink``, stream out: ``source``, ``reset``).
  Here the streams become whole arrays: :meth:`MFCC.process` (float contract, fp32 on the GPU)
  and :meth:`MFCC.process_fixed` (RTL-exact int16).
* ``mfcc_open / mfcc_convert(sess, path_in, path_out) / mfcc_close`` -- the host driver,
  ``software/main.c:36,100,53``, plus the directory walker ``show_dir_content`` (:206-247).

Everything numerical happens in libmfcc_hip.so (HIP kernels); this module only marshals
buffers.  Without a GPU :class:`MFCC` construction raises ``MfccHipError(NOT_FOUND)``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import MfccHipError, Params

PAD_NOTEBOOK = _lib.PAD_NOTEBOOK
PAD_STREAM = _lib.PAD_STREAM
_PAD = {"notebook": PAD_NOTEBOOK, "stream": PAD_STREAM, PAD_NOTEBOOK: PAD_NOTEBOOK, PAD_STREAM: PAD_STREAM}
_IMPL = {"auto": _lib.IMPL_AUTO, "generic": _lib.IMPL_GENERIC, "fused512": _lib.IMPL_FUSED512}


def make_params(nfft=512, hop=None, nfilters=32, nceptrums=13, samplerate=16000, pad_mode="notebook",
                power_scale=512.0, lifter=0.0, device=-1, impl="auto") -> Params:
    lib = _lib.load()
    p = Params()
    _lib.check(lib.mfcc_hip_default_params(C.byref(p)))
    p.nfft = int(nfft)
    p.hop = 0 if hop is None else int(hop)          # 0 -> nfft // 3 (mfcc/core/mfcc.py:43)
    p.n_mel = int(nfilters)
    p.n_cep = int(nceptrums)
    p.sample_rate = int(samplerate)
    p.pad_mode = _PAD[pad_mode]
    p.power_scale = float(power_scale) if power_scale else 0.0
    p.lifter = float(lifter)
    p.device = int(device)
    p.float_impl = _IMPL[impl] if isinstance(impl, str) else int(impl)
    return p


def num_frames(n_samples, **kw) -> int:
    """Frames a stream of ``n_samples`` yields (host-only; `nframes`, software/main.c:95)."""
    lib = _lib.load()
    p = make_params(**kw)
    out = C.c_size_t(0)
    _lib.check(lib.mfcc_hip_num_frames(C.byref(p), int(n_samples), C.byref(out)), "num_frames")
    return int(out.value)


_TABLE_DTYPES = {
    _lib.TABLE_WINDOW_F32: np.float32, _lib.TABLE_MEL_POINTS_I32: np.int32,
    _lib.TABLE_MEL_DENSE_F32: np.float32, _lib.TABLE_DCT_F32: np.float32,
    _lib.TABLE_FX_CURVE_I32: np.int32, _lib.TABLE_FX_TWIDDLE_I32: np.int32,
    _lib.TABLE_FX_MEL_DENSE_U32: np.uint32,
}


def get_table(which, **kw) -> np.ndarray:
    """The constant tables as the library's host code builds them (works without a GPU)."""
    lib = _lib.load()
    p = make_params(**kw)
    n = C.c_size_t(0)
    _lib.check(lib.mfcc_hip_get_table(C.byref(p), which, None, 0, C.byref(n)), "get_table")
    buf = np.empty(n.value, dtype=np.uint8)
    _lib.check(lib.mfcc_hip_get_table(C.byref(p), which, buf.ctypes.data, buf.nbytes, C.byref(n)))
    return buf.view(_TABLE_DTYPES[which])


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class MFCC:
    """``MFCC(width=16, nfft=512, samplerate=16e3, nfilters=16, nceptrums=16)`` -- same
    constructor arguments as ``mfcc/core/mfcc.py:20-21``; extra keyword arguments select the
    host driver's framing (``pad_mode``), the float path's power scale / lifter and the device.

    ``hop`` defaults to ``nfft // 3`` like the core (``mfcc.py:43``); the notebook and the host
    driver hard-code 170 for nfft 512, which is the same number.
    """

    def __init__(self, width=16, nfft=512, samplerate=16e3, nfilters=16, nceptrums=16, *, hop=None,
                 pad_mode="notebook", power_scale=512.0, lifter=0.0, device=-1, impl="auto"):
        if width != 16:
            raise ValueError("only width=16 (int16 PCM) is supported, like every reference target")
        self.width = width
        self.nfft = int(nfft)
        self.samplerate = samplerate
        self.nfilters = int(nfilters)
        self.nceptrums = int(nceptrums)
        self._lib = _lib.load()
        self._params = make_params(nfft=nfft, hop=hop, nfilters=nfilters, nceptrums=nceptrums,
                                   samplerate=int(samplerate), pad_mode=pad_mode, power_scale=power_scale,
                                   lifter=lifter, device=device, impl=impl)
        self.hop = self._params.hop or self.nfft // 3
        h = C.c_void_p()
        self._device_index = None
        if int(device) < 0:
            # "the current device" is resolved NOW, by the library, from torch's current device
            try:
                import torch
                if torch.cuda.is_available():
                    self._device_index = torch.cuda.current_device()
            except ImportError:
                pass
        _lib.check(self._lib.mfcc_hip_create(C.byref(self._params), C.byref(h)), "mfcc_hip_create")
        self._h = h

    # -- lifetime (``reset`` of the core clears all state: every call here starts from reset)
    def close(self):
        if getattr(self, "_h", None):
            self._lib.mfcc_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def num_frames(self, n_samples) -> int:
        out = C.c_size_t(0)
        _lib.check(self._lib.mfcc_hip_num_frames(C.byref(self._params), int(n_samples), C.byref(out)))
        return int(out.value)

    def kernel_name(self, fixed=False) -> str:
        return self._lib.mfcc_hip_kernel_name(self._h, int(fixed)).decode()

    def is_fallback(self, fixed=False) -> bool:
        """True when this parameter set runs on a generic kernel (one frame per wave, 3-6 x slower than the fused kernels
        that cover the reference's own configurations): bench.py puts it into its line as ``config.fallback``."""
        name = self.kernel_name(fixed)
        return "generic" in name or name == "mfcc_fixed_kernel"

    def set_stream(self, stream_ptr):
        """Launch on a caller-provided hipStream_t (e.g. ``torch.cuda.current_stream().cuda_stream``;
        0 / None is the HIP null stream, torch's default)."""
        _lib.check(self._lib.mfcc_hip_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def use_own_stream(self):
        _lib.check(self._lib.mfcc_hip_use_own_stream(self._h))

    def synchronize(self):
        _lib.check(self._lib.mfcc_hip_synchronize(self._h))

    # -- the hot path ---------------------------------------------------------------
    def _host(self, pcm, fixed):
        pcm = np.ascontiguousarray(pcm)
        if pcm.dtype != np.int16:
            raise TypeError("pcm must be int16 (the core's sink is signed 16 bit, mfcc.py:29)")
        squeeze = pcm.ndim == 1
        if squeeze:
            pcm = pcm[None, :]
        if pcm.ndim != 2:
            raise ValueError("pcm must be (n,) or (channels, n)")
        nch, n = pcm.shape
        nf = self.num_frames(n)
        out = np.empty((nch, nf, self.nceptrums), dtype=np.int16 if fixed else np.float32)
        got = C.c_size_t(0)
        fn = self._lib.mfcc_hip_process_fixed_i16 if fixed else self._lib.mfcc_hip_process_i16
        _lib.check(fn(self._h, pcm.ctypes.data, n, nch, out.ctypes.data, out.size, C.byref(got)),
                   "process")
        assert got.value == nf
        return out[0] if squeeze else out

    def _dev(self, pcm, fixed, halo, out):
        import torch
        if pcm.dtype != torch.int16 or not pcm.is_cuda:
            raise TypeError("device path needs a CUDA(HIP) int16 tensor")
        squeeze = pcm.dim() == 1
        if squeeze:
            pcm = pcm[None, :]
        if pcm.stride(1) != 1:
            pcm = pcm.contiguous()
        nch, n_tot = pcm.shape
        if halo not in (0, 1) or n_tot < int(halo):
            raise ValueError("halo must be 0 or 1 and counted in the samples")
        n = n_tot - int(halo)
        nf = self.num_frames(n)
        self._check_device(pcm)
        odt = torch.int16 if fixed else torch.float32
        if out is None:
            out = torch.empty((nch, nf, self.nceptrums), device=pcm.device, dtype=odt)
        else:
            # the kernel gets raw pointers: a wrong shape / dtype / layout / device would be an out-of-bounds write
            want = (nf, self.nceptrums) if squeeze and out.dim() == 2 else (nch, nf, self.nceptrums)
            if tuple(out.shape) != want or out.dtype != odt or not out.is_contiguous() or out.device != pcm.device:
                raise ValueError("out must be a contiguous %s tensor of shape %s on %s" % (odt, want, pcm.device))
        fn = self._lib.mfcc_hip_process_fixed_i16_dev if fixed else self._lib.mfcc_hip_process_i16_dev
        got = C.c_size_t(0)
        with self._on_torch_stream(pcm.device):
            _lib.check(fn(self._h, C.c_void_p(pcm.data_ptr()), n, pcm.stride(0), nch, int(halo),
                          C.c_void_p(out.data_ptr()), C.byref(got)), "process_dev")
        if squeeze and out.dim() == 3:
            return out[0]
        return out

    def _check_device(self, t):
        """The handle's tables, stream and scratch live on ONE GPU: refuse tensors of another one."""
        if self._device_index is None:
            import torch
            self._device_index = torch.cuda.current_device() if self._params.device < 0 else int(self._params.device)
        if t.device.index != self._device_index:
            raise ValueError("tensor on %s but this MFCC handle was created on cuda:%d" % (t.device, self._device_index))

    def _on_torch_stream(self, device):
        """Context: launch on torch's current stream of `device`, then go back to the handle's own stream, so
        that later host-path calls do not run on (or outlive) a stream torch owns."""
        import contextlib
        import torch

        @contextlib.contextmanager
        def ctx():
            self.set_stream(torch.cuda.current_stream(device).cuda_stream)
            try:
                yield
            finally:
                self.use_own_stream()
        return ctx()

    def process(self, pcm, halo=0, out=None):
        """Float contract: int16 PCM ``(n,)`` / ``(channels, n)`` -> float32 ``(.., frames, nceptrums)``.
        NumPy in -> NumPy out (H2D, kernel, D2H); torch CUDA tensor in -> torch tensor out, asynchronous
        on the current stream.  ``halo=1`` (device path): sample 0 of every channel is history only."""
        if _is_torch(pcm):
            return self._dev(pcm, False, halo, out)
        if halo:
            raise ValueError("halo is only available on the device path")
        return self._host(pcm, False)

    def process_fixed(self, pcm, halo=0, out=None):
        """Fixed contract (RTL arithmetic): int16 PCM -> int16 coefficients, bit-exact."""
        if _is_torch(pcm):
            return self._dev(pcm, True, halo, out)
        if halo:
            raise ValueError("halo is only available on the device path")
        return self._host(pcm, True)

    def process_batch(self, utterances, fixed=False):
        """Many utterances of different lengths in ONE launch (the batched form of the driver's directory
        walk, main.c:206-247).  ``utterances``: sequence of 1-D int16 arrays.  Returns a list of
        ``(frames_u, nceptrums)`` arrays (views of one result buffer), bit-identical to calling
        ``process`` / ``process_fixed`` on each utterance."""
        if len(utterances) and _is_torch(utterances[0]):
            return self._batch_dev(utterances, fixed)
        utts = [np.ascontiguousarray(u, dtype=np.int16).reshape(-1) for u in utterances]
        n = len(utts)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        if n:
            offsets[1:] = np.cumsum([u.size for u in utts], dtype=np.uint64)
        flat = np.concatenate(utts) if n and int(offsets[-1]) else np.zeros(0, dtype=np.int16)
        fo = np.zeros(n + 1, dtype=np.uint64)
        nf = sum(self.num_frames(u.size) for u in utts)
        out = np.empty((nf, self.nceptrums), dtype=np.int16 if fixed else np.float32)
        fn = self._lib.mfcc_hip_process_ragged_fixed_i16 if fixed else self._lib.mfcc_hip_process_ragged_i16
        _lib.check(fn(self._h, flat.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p), n,
                      out.ctypes.data_as(C.c_void_p), out.size, fo.ctypes.data_as(C.c_void_p)), "process_ragged")
        assert int(fo[-1]) == nf
        return [out[int(fo[i]):int(fo[i + 1])] for i in range(n)]

    def _batch_dev(self, utterances, fixed):
        """``process_batch`` for torch CUDA int16 tensors: stays on the device, asynchronous on the current stream."""
        import torch
        utts = [u.reshape(-1) for u in utterances]
        if any(u.dtype != torch.int16 or not u.is_cuda for u in utts):
            raise TypeError("device path needs CUDA(HIP) int16 tensors")
        n = len(utts)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum([u.numel() for u in utts], dtype=np.uint64)
        # utterances that are consecutive views of one buffer (a corpus already laid out in HBM) are used in place
        in_place = n > 0 and all(u.is_contiguous() for u in utts) and all(
            utts[i].data_ptr() + 2 * utts[i].numel() == utts[i + 1].data_ptr() and
            utts[i].untyped_storage().data_ptr() == utts[0].untyped_storage().data_ptr() for i in range(n - 1))
        if in_place:
            base = utts[0]
            flat = torch.as_strided(base, (int(offsets[-1]),), (1,), storage_offset=base.storage_offset())
        else:
            flat = torch.cat(utts) if int(offsets[-1]) else torch.zeros(0, dtype=torch.int16, device=utts[0].device)
        out, fo = self.process_packed(flat, offsets, fixed=fixed)
        return [out[int(fo[i]):int(fo[i + 1])] for i in range(n)]

    def process_packed(self, flat, offsets, fixed=False, out=None):
        """A corpus that already lies in HBM: ``flat`` = all utterances back to back (1-D CUDA int16 tensor),
        utterance ``u`` = ``flat[offsets[u]:offsets[u + 1]]``.  ONE launch (``mfcc_hip_process_ragged_*_dev``),
        asynchronous on the current stream.  Returns ``(out, frame_offsets)``: the dense ``(sum frames, nceptrums)``
        result tensor and the row range of every utterance (``out[fo[u]:fo[u + 1]]``).  Equal-length utterances run
        as channels of one multi-channel launch (no packing copy); the bits are the same."""
        import torch
        if flat.dtype != torch.int16 or not flat.is_cuda or flat.dim() != 1 or not flat.is_contiguous():
            raise TypeError("flat must be a contiguous 1-D CUDA(HIP) int16 tensor")
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        if n < 0 or (n >= 0 and len(offsets) and int(offsets[-1]) > flat.numel()):
            raise ValueError("offsets run past the end of flat")
        self._check_device(flat)
        lens = np.diff(offsets.astype(np.int64))
        if n and int(lens.min()) < 0:
            raise ValueError("offsets must not decrease")
        fo = np.zeros(n + 1, dtype=np.uint64)
        # this runs once per step of a sharded corpus (bench.py enqueue_us_per_step): equal lengths -- config 5 as
        # BASELINE defines it -- need one frame count, not a table of them
        if n and int(lens.min()) == int(lens.max()):
            nf = self.num_frames(int(lens[0])) * n
        else:
            uniq, counts = np.unique(lens, return_counts=True)
            nf = int(sum(self.num_frames(int(v)) * int(c) for v, c in zip(uniq, counts)))
        odt = torch.int16 if fixed else torch.float32
        if out is None:
            out = torch.empty((nf, self.nceptrums), device=flat.device, dtype=odt)
        elif tuple(out.shape) != (nf, self.nceptrums) or out.dtype != odt or not out.is_contiguous() or \
                out.device != flat.device:
            raise ValueError("out must be a contiguous %s tensor of shape %s on %s" % (odt, (nf, self.nceptrums), flat.device))
        fn = self._lib.mfcc_hip_process_ragged_fixed_i16_dev if fixed else self._lib.mfcc_hip_process_ragged_i16_dev
        with self._on_torch_stream(flat.device):
            _lib.check(fn(self._h, C.c_void_p(flat.data_ptr()), offsets.ctypes.data_as(C.c_void_p), n,
                          C.c_void_p(out.data_ptr()), out.numel(), fo.ctypes.data_as(C.c_void_p)), "process_ragged_dev")
        assert int(fo[-1]) == nf
        return out, fo

    def time_launches(self, pcm, out, fixed=False, warmup=2, iters=10) -> float:
        """Average kernel time in ms over ``iters`` launches, HIP events on the launch stream."""
        import torch
        if pcm.dim() == 1:
            pcm = pcm[None, :]
        self._check_device(pcm)
        ms = C.c_float(0)
        with self._on_torch_stream(pcm.device):
            _lib.check(self._lib.mfcc_hip_time_dev(self._h, int(fixed), C.c_void_p(pcm.data_ptr()), pcm.shape[1],
                                                   pcm.stride(0), pcm.shape[0], C.c_void_p(out.data_ptr()),
                                                   warmup, iters, C.byref(ms)), "time_dev")
        return float(ms.value)


    # -- online mode: the core's own interface, sink / source / reset (mfcc/core/mfcc.py:28-30) -----
    def stream(self, fixed=False) -> "MfccStream":
        """A streaming session on this handle: feed chunks, get the frames they complete."""
        return MfccStream(self, fixed)

    # -- file level: mfcc_convert(sess, path_in, path_out), software/main.c:100-177 -----
    def convert_many(self, paths_in, paths_out, fixed=True):
        """``mfcc_convert`` for many files in one ragged launch; returns the frame count of each file."""
        n = len(paths_in)
        assert n == len(paths_out)
        a_in = (C.c_char_p * n)(*[os.fsencode(p) for p in paths_in])
        a_out = (C.c_char_p * n)(*[os.fsencode(p) for p in paths_out])
        nf = (C.c_size_t * n)()
        _lib.check(self._lib.mfcc_hip_convert_wavs(self._h, a_in, a_out, n, int(fixed), nf), "convert_wavs")
        return [int(v) for v in nf]

    def convert(self, path_in, path_out, fixed=True) -> int:
        """``x.wav -> x.mfcc``: raw int16 LE ``[frame][nceptrums]``.  Returns the frame count."""
        nf = C.c_size_t(0)
        _lib.check(self._lib.mfcc_hip_convert_wav(self._h, os.fsencode(path_in), os.fsencode(path_out),
                                                  int(fixed), C.byref(nf)), "convert %s" % path_in)
        return int(nf.value)


class MfccStream:
    """Online mode -- the stream interface of the nMigen core (``sink`` in, ``source`` out, ``reset``;
    mfcc/core/mfcc.py:28-30,116) and of its targets (wav2mfcc.py:27-42: bit 31 = soft reset; mic2mfcc.py:19-30).
    The session keeps the core's cross-frame state on the device: one pre-emphasis history sample and the
    samples of the frame in progress.  Any chunking gives the one-shot result, frame for frame, bit for bit."""

    def __init__(self, mfcc: MFCC, fixed=False):
        self._m = mfcc
        self._lib = mfcc._lib
        self.fixed = bool(fixed)
        s = C.c_void_p()
        _lib.check(self._lib.mfcc_hip_stream_create(mfcc._h, int(self.fixed), C.byref(s)), "stream_create")
        self._s = s

    def close(self):
        # either order is safe: a handle closed first is kept alive by the library until its last session goes
        if getattr(self, "_s", None):
            self._lib.mfcc_hip_stream_destroy(self._s)
        self._s = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def pending(self) -> int:
        return int(self._lib.mfcc_hip_stream_pending(self._s))

    def _out(self, nf):
        return np.empty((nf, self._m.nceptrums), dtype=np.int16 if self.fixed else np.float32)

    def push(self, samples) -> np.ndarray:
        """``sink``: int16 samples in; returns the ``(frames, nceptrums)`` they complete (possibly 0 rows)."""
        samples = np.ascontiguousarray(samples)
        if samples.dtype != np.int16 or samples.ndim != 1:
            raise TypeError("samples must be a 1-D int16 array (the core's sink is signed 16 bit, mfcc.py:29)")
        out = self._out(int(self._lib.mfcc_hip_stream_max_frames(self._s, samples.size)))
        nf = C.c_size_t(0)
        _lib.check(self._lib.mfcc_hip_stream_push(self._s, samples.ctypes.data, samples.size, out.ctypes.data,
                                                  out.size, C.byref(nf)), "stream_push")
        return out[:nf.value]

    def flush(self) -> np.ndarray:
        """End of the stream: the zero-padded tail frame of the host driver (main.c:134-144) with
        ``pad_mode="stream"``, nothing with ``"notebook"``; the session is reset afterwards."""
        out = self._out(1)
        nf = C.c_size_t(0)
        _lib.check(self._lib.mfcc_hip_stream_flush(self._s, out.ctypes.data, out.size, C.byref(nf)), "stream_flush")
        return out[:nf.value]

    def reset(self):
        """``MFCC.reset`` / ``mfcc_softreset`` (main.c:21-34): drop pending samples, history back to 0."""
        _lib.check(self._lib.mfcc_hip_stream_reset(self._s), "stream_reset")


def lift_file(mfcc_in, lift_out, nceptrums=32, L=22) -> int:
    """``x.mfcc -> x.lift`` like the loop of software/lift.py:28-40 (host only); returns the frame count."""
    nf = C.c_size_t(0)
    _lib.check(_lib.load().mfcc_hip_lift_file(os.fsencode(mfcc_in), os.fsencode(lift_out), int(nceptrums), float(L),
                                              C.byref(nf)), "lift_file %s" % mfcc_in)
    return int(nf.value)


# ---- software/main.c names ---------------------------------------------------------------

def mfcc_open(**kw) -> MFCC:
    """``mfcc_open`` (main.c:36): a session with the host driver's constants
    ``NFFT 512, STEPSIZE 170, NCEPSTRUMS 32, SAMPLERATE 16000`` (main.c:11-14) and its
    zero-padded tail frame (main.c:134-144)."""
    args = dict(nfft=512, samplerate=16000, nfilters=32, nceptrums=32, pad_mode="stream")
    args.update(kw)
    return MFCC(**args)


def mfcc_convert(sess: MFCC, path_in, path_out, fixed=True) -> int:
    """``mfcc_convert(sess, path_in, path_out)`` (main.c:100); 0 on success like the original."""
    sess.convert(path_in, path_out, fixed=fixed)
    return 0


def mfcc_close(sess: MFCC) -> None:
    sess.close()


def show_dir_content(sess: MFCC, path, fixed=True, batch_bytes=1 << 30):
    """Recursive ``*.wav -> *.mfcc`` walk of ``show_dir_content`` (main.c:206-247).  The files are
    converted in ragged batches -- one launch per ``batch_bytes`` of WAV data instead of one USB
    ping-pong per frame -- and the ``.mfcc`` files are byte-identical to per-file ``mfcc_convert``.
    Returns the list of (wav, mfcc) pairs converted."""
    pairs = []
    for root, _dirs, files in os.walk(path):
        for name in sorted(files):
            if name.endswith(".wav"):
                src = os.path.join(root, name)
                pairs.append((src, src[:-3] + "mfcc"))
    batch, size = [], 0
    for i, (src, dst) in enumerate(pairs):
        batch.append((src, dst))
        size += os.path.getsize(src)
        if size >= batch_bytes or i == len(pairs) - 1:
            sess.convert_many([b[0] for b in batch], [b[1] for b in batch], fixed=fixed)
            batch, size = [], 0
    return pairs


def lifter(cepstra, L=22):
    """``lifter(cepstra, L=22)`` of software/lift.py:12-26 on a ``.mfcc`` array (host-side post
    step on files; the float kernel can fold the same lifter in via ``MFCC(lifter=L)``)."""
    cepstra = np.asarray(cepstra)
    if L > 0:
        n = np.arange(cepstra.shape[1])
        return (1 + (L / 2.) * np.sin(np.pi * n / L)) * cepstra
    return cepstra
