"""Streaming / online session (mfcc_hip_stream_*): any chunking of a stream reproduces the one-shot result
bit for bit, both contracts, both framing modes -- the core's sink/source/reset interface
(mfcc/core/mfcc.py:28-30,116; wav2mfcc.py:27-42; receiver loop software/cepstrum.c:93-159)."""
import numpy as np
import pytest

from oracle import mfcc_fixed as mx
from oracle import mfcc_float as mf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mfcc_amd():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import mfcc_amd
    return mfcc_amd


def _chunks(n, kind, rng):
    pos = 0
    while pos < n:
        if kind == "driver":                      # the host driver's own pattern: 512, then 170 per round (main.c:134)
            c = 512 if pos == 0 else 170
        elif kind == "random":
            c = int(rng.choice([1, 2, 7, 169, 170, 171, 341, 511, 512, 513, 1000, 4096, 30000]))
        else:
            c = int(kind)
        yield pos, min(n, pos + c)
        pos += c


def _run(sess, pcm, kind, seed=0):
    rng = np.random.default_rng(seed)
    rows = [sess.push(pcm[a:b]) for a, b in _chunks(len(pcm), kind, rng)]
    rows.append(sess.flush())
    return np.concatenate(rows)


@pytest.mark.parametrize("fixed", [False, True])
@pytest.mark.parametrize("pad_mode", ["stream", "notebook"])
def test_any_chunking_equals_one_shot(mfcc_amd, wav_pcm, fixed, pad_mode):
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=32, pad_mode=pad_mode) as m:
        one = m.process_fixed(wav_pcm) if fixed else m.process(wav_pcm)
        with m.stream(fixed=fixed) as s:
            for kind in ("driver", 170, 171, 4096, "random", len(wav_pcm)):
                got = _run(s, wav_pcm, kind, seed=3)
                assert got.dtype == one.dtype and got.shape == one.shape, (kind, got.shape, one.shape)
                assert np.array_equal(got, one), kind
                assert s.pending == 0                                    # flush leaves a reset session
            # one sample at a time (the RTL's own rate): the first 6000 samples, then the rest in one push
            rows = [s.push(wav_pcm[i:i + 1]) for i in range(6000)]
            assert sum(len(r) for r in rows) == (6000 - 512) // 170 + 1
            rows += [s.push(wav_pcm[6000:]), s.flush()]
            assert np.array_equal(np.concatenate(rows), one)
    # and the one-shot result is the oracle's
    if fixed:
        assert np.array_equal(one, mx.mfcc_fixed_ref(wav_pcm, nceptrums=32, pad_mode=pad_mode))
    else:
        ref = mf.mfcc_float_ref(wav_pcm, n_cep=32, pad_mode=pad_mode)
        assert np.abs(one - ref).max() / np.abs(ref).max() < 2e-5


def test_reset_mid_stream_and_short_streams(mfcc_amd, wav_pcm):
    """`reset` (bit 31 of a word, wav2mfcc.py:27-36) drops the frame in progress and the pre-emphasis history;
    streams shorter than a frame give the driver's single zero-padded frame; an empty flush too."""
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=16, pad_mode="stream") as m, m.stream(fixed=True) as s:
        assert s.push(wav_pcm[:1000]).shape == (3, 16) and s.pending == 1000 - 3 * 170
        s.reset()
        assert s.pending == 0
        got = np.concatenate([s.push(wav_pcm[5000:9000]), s.flush()])
        assert np.array_equal(got, m.process_fixed(wav_pcm[5000:9000]))
        for n in (0, 1, 100, 511):
            got = np.concatenate([s.push(wav_pcm[:n]), s.flush()])
            assert got.shape == (1, 16)
            assert np.array_equal(got, m.process_fixed(wav_pcm[:n]))
            assert np.array_equal(got, mx.mfcc_fixed_ref(wav_pcm[:n], nceptrums=16, pad_mode="stream"))
        # two sessions on one handle are independent
        with m.stream(fixed=True) as s2:
            a = s.push(wav_pcm[:700])
            b = s2.push(wav_pcm[20000:20700])
            a2, b2 = s.push(wav_pcm[700:1400]), s2.push(wav_pcm[20700:21400])
            assert np.array_equal(np.concatenate([a, a2]), m.process_fixed(wav_pcm[:1400])[:len(a) + len(a2)])
            assert np.array_equal(np.concatenate([b, b2]), m.process_fixed(wav_pcm[20000:21400])[:len(b) + len(b2)])


def test_stream_feeds_the_serial_wire_format(mfcc_amd, wav_pcm):
    """The UART byte stream (0xa55a + n_cep big-endian int16 per frame, misc/magic.py:27-39) packed from the
    streamed columns equals the one packed from the one-shot result; the receiver decodes it back."""
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=16, pad_mode="stream") as m, m.stream(fixed=True) as s:
        one = m.process_fixed(wav_pcm)
        wire = b"".join(mfcc_amd.wire.pack_columns(s.push(wav_pcm[a:a + 4000])) for a in range(0, len(wav_pcm), 4000))
        wire += mfcc_amd.wire.pack_columns(s.flush())
    assert wire == mfcc_amd.wire.pack_columns(one)
    cols, used = mfcc_amd.wire.unpack_columns(wire, 16)
    assert used == len(wire) and np.array_equal(cols, one)


def test_push_with_a_small_buffer_leaves_the_session_untouched(mfcc_amd, wav_pcm):
    import ctypes as C
    lib = mfcc_amd.load_library()
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13) as m, m.stream() as s:
        x = np.ascontiguousarray(wav_pcm[:2000])
        out = np.empty((2, 13), np.float32)
        nf = C.c_size_t(0)
        rc = lib.mfcc_hip_stream_push(s._s, x.ctypes.data, x.size, out.ctypes.data, out.size, C.byref(nf))
        assert rc == -106 and nf.value == 9 and s.pending == 0           # BUFFER_SMALL, nothing consumed
        assert np.array_equal(s.push(x), m.process(x))
