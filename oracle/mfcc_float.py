"""CPU oracle for the FLOAT contract of lambdaconcept/mfcc -- TEST INFRASTRUCTURE ONLY.

This file is a restatement, in this repo's own code, of the float64 NumPy/SciPy
model the reference ships as ``notebook/MFCC.ipynb`` (the contract for the
<= 1e-4 tolerance target).  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; ``mfcc_amd`` never does.

Pinning: ``tests/golden/make_golden.py`` executes the
notebook's own cells verbatim on ``f2bjrop1.0.wav`` in the build container and
commits the outputs; ``tests/test_oracle_float.py`` requires this restatement
to reproduce them bit for bit (``np.array_equal``), plus the known answers the
notebook's stored outputs hold (filter points, row sums).

Every function cites the notebook cell it follows ("NB cell N" = 0-based index
into ``notebook/MFCC.ipynb``'s cell list, markdown cells counted).

The per-frame Python loops of the notebook (cells 9 and 20) are kept on
purpose in :func:`mfcc_notebook`: that function is also what ``bench.py`` times
as the "notebook NumPy CPU path" baseline.
"""
from __future__ import annotations

import numpy as np
import scipy.fftpack as _fftpack
from scipy.signal import get_window as _get_window

EMPHASIS_COEFF = 0.96875          # NB cell 7 line 3  (= 1 - 1/32, README "Pre-Emphasis")


# --------------------------------------------------------------------------- stages

def pre_emphasis(audio: np.ndarray) -> np.ndarray:
    """NB cell 7: ``np.append(audio[0], audio[1:] - 0.96875 * audio[:-1])`` (float64)."""
    audio = np.asarray(audio)
    return np.append(audio[0], audio[1:] - EMPHASIS_COEFF * audio[:-1])


def num_frames_notebook(n_samples: int, nfft: int = 512, hop: int = 170) -> int:
    """NB cell 9 line 9: ``int((len(audio) - FFT_size) / hop_len) + 1`` (no padding)."""
    if n_samples < nfft:
        return 0
    return int((n_samples - nfft) / hop) + 1


def frame_audio(audio: np.ndarray, nfft: int = 512, hop: int = 170) -> np.ndarray:
    """NB cell 9 (hop hard-coded to 170 there; MFCC-INT cell 3 uses ``FFT_size // 3``).

    The copy loop is the notebook's own (cell 9 lines 15-16)."""
    frame_num = num_frames_notebook(len(audio), nfft, hop)
    frames = np.zeros((frame_num, nfft))
    for n in range(frame_num):
        frames[n] = audio[n * hop:n * hop + nfft]
    return frames


def hamming_window(nfft: int = 512) -> np.ndarray:
    """NB cell 17 line 5: ``get_window("hamm", FFT_size, fftbins=True)`` (periodic)."""
    return _get_window("hamm", nfft, fftbins=True)


def fft_frames(frames_win: np.ndarray, nfft: int = 512) -> list:
    """NB cell 20 lines 19-22: per-frame ``scipy.fftpack.fft(frame)[0:nfft//2+1]``."""
    out = []
    size = int(1 + nfft // 2)
    for frame in frames_win:
        out.append(_fftpack.fft(frame, axis=0)[0:size])
    return out


def power_spectrum(audio_fft: list, scale: float = 512.0) -> np.ndarray:
    """NB cell 22 line 1: ``np.square(np.abs([a/512 for a in audio_fft]))``.

    ``scale`` is hard-coded to 512 in the notebook; config 4 (nfft 1024) uses
    ``scale = nfft``."""
    return np.square(np.abs([a / scale for a in audio_fft]))


def freq_to_mel(freq):
    """NB cell 26."""
    return 2595.0 * np.log10(1.0 + freq / 700.0)


def mel_to_freq(mels):
    """NB cell 26 (``met_to_freq`` there)."""
    return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)


def get_filter_points(fmin, fmax, mel_filter_num, nfft, sample_rate=16000):
    """NB cell 27: ``floor((FFT_size + 1) / sample_rate * freqs)``; returns (points, freqs)."""
    fmin_mel = freq_to_mel(fmin)
    fmax_mel = freq_to_mel(fmax)
    mels = np.linspace(fmin_mel, fmax_mel, num=mel_filter_num + 2)
    freqs = mel_to_freq(mels)
    return np.floor((nfft + 1) / sample_rate * freqs).astype(int), freqs


def get_filters(filter_points, nfft):
    """NB cell 30: triangular filters from linspace ramps, no area normalisation
    (the librosa-style norm of cell 33 is computed but its application is commented out)."""
    filters = np.zeros((len(filter_points) - 2, int(nfft / 2 + 1)))
    for n in range(len(filter_points) - 2):
        filters[n, filter_points[n]:filter_points[n + 1]] = \
            np.linspace(0, 1, filter_points[n + 1] - filter_points[n])
        filters[n, filter_points[n + 1]:filter_points[n + 2]] = \
            np.linspace(1, 0, filter_points[n + 2] - filter_points[n + 1])
    return filters


def mel_filterbank(nfft=512, n_mel=32, sample_rate=16000):
    """NB cells 24 + 28 + 31: fmin 0, fmax sample_rate/2."""
    points, _ = get_filter_points(0, sample_rate / 2, n_mel, nfft, sample_rate=sample_rate)
    return get_filters(points, nfft)


def dct_basis(dct_filter_num, filter_len):
    """NB cell 38: orthonormal DCT-II basis."""
    basis = np.empty((dct_filter_num, filter_len))
    basis[0, :] = 1.0 / np.sqrt(filter_len)
    samples = np.arange(1, 2 * filter_len, 2) * np.pi / (2.0 * filter_len)
    for i in range(1, dct_filter_num):
        basis[i, :] = np.cos(i * samples) * np.sqrt(2.0 / filter_len)
    return basis


def lifter(cepstra, L=22):
    """NB cell 43 / software/lift.py:12-26: ``1 + (L/2) sin(pi n / L)`` on (frames, ncoeff)."""
    if L > 0:
        _, ncoeff = np.shape(cepstra)
        n = np.arange(ncoeff)
        lift = 1 + (L / 2.) * np.sin(np.pi * n / L)
        return lift * cepstra
    return cepstra


# --------------------------------------------------------------------------- whole chain

def mfcc_notebook(audio, nfft=512, hop=170, n_mel=32, sample_rate=16000,
                  power_scale=512.0, return_stages=False):
    """The notebook path, cells 7 -> 39, in order, loops included.

    Returns ``cepstral_coefficents`` transposed to (frames, n_mel) float64 --
    the notebook's own array is (n_mel, frames) (cell 39); the product's layout is
    frame-major like the reference's ``.mfcc`` files (software/main.c:162-165).
    """
    audio = np.asarray(audio)
    audio_emphasis = pre_emphasis(audio)                                  # cell 7
    audio_framed = frame_audio(audio_emphasis, nfft=nfft, hop=hop)        # cells 9-10
    window = hamming_window(nfft)                                         # cell 17
    audio_win = audio_framed * window                                     # cell 18
    audio_fft = fft_frames(audio_win, nfft)                               # cell 20
    audio_power = power_spectrum(audio_fft, power_scale)                  # cell 22
    filters = mel_filterbank(nfft, n_mel, sample_rate)                    # cells 24-31
    audio_filtered = np.dot(filters, np.transpose(audio_power))           # cell 36 line 1
    with np.errstate(divide="ignore"):
        audio_log = np.log2(audio_filtered)                               # cell 36 line 12
    dct_filters = dct_basis(n_mel, n_mel)                                 # cells 38-39
    cepstral_coefficents = np.dot(dct_filters, audio_log)                 # cell 39 line 6
    out = np.ascontiguousarray(cepstral_coefficents.T)
    if return_stages:
        return out, dict(emphasis=audio_emphasis, framed=audio_framed, window=window,
                         windowed=audio_win, fft=audio_fft, power=audio_power,
                         filters=filters, mel=audio_filtered.T, logmel=audio_log.T,
                         dct_basis=dct_filters)
    return out


def mfcc_float_ref(pcm, n_cep=13, pad_mode="notebook", **kw):
    """Oracle entry point matching the product's output layout.

    ``pcm``: int16 array (n,) or (channels, n).  ``pad_mode``: "notebook"
    (frames = int((n - nfft)/hop) + 1, NB cell 9) or "stream" (the host driver's
    framing, software/main.c:95,134-144: zero padding after EOF until the frame that
    contains the last sample has been emitted -> (n - nfft)//hop + 2 frames).
    Returns float64 (channels?, frames, n_cep).
    """
    pcm = np.asarray(pcm)
    if pcm.ndim == 2:
        return np.stack([mfcc_float_ref(c, n_cep=n_cep, pad_mode=pad_mode, **kw) for c in pcm])
    nfft = kw.get("nfft", 512)
    hop = kw.get("hop", 170)
    if pad_mode == "stream":
        nf = num_frames_stream(len(pcm), nfft, hop)
        need = (nf - 1) * hop + nfft
        pcm = np.concatenate([pcm, np.zeros(need - len(pcm), dtype=pcm.dtype)])
    elif pad_mode != "notebook":
        raise ValueError(pad_mode)
    return mfcc_notebook(pcm, **kw)[:, :n_cep]


def num_frames_stream(n_samples: int, nfft: int = 512, hop: int = 170) -> int:
    """Frame count of the reference host driver's loop (software/main.c:128-166):
    first transfer ``nfft`` samples, then ``hop`` per frame, until a read hits EOF;
    the transfer that hits EOF is zero-padded and still produces a frame."""
    if n_samples < nfft:
        return 1
    return (n_samples - nfft) // hop + 2


def synth_pcm(n_samples: int, seed: int = 0) -> np.ndarray:
    """Synthetic 16 kHz PCM of SURVEY.md section 8d config 2: white Gaussian, sigma 3000,
    clipped to int16 (never an all-zero span, so no log2(0))."""
    x = np.random.default_rng(seed).standard_normal(n_samples) * 3000.0
    return np.clip(x, -32768, 32767).astype(np.int16)
