"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against the oracle
on the same inputs.  Float contract: max|d|/max|ref| and rel-L2 <= 1e-4 (BASELINE.json).
Fixed contract: bit-exact.  Run with `pytest -m gpu` on the MI355X box."""
import os

import numpy as np
import pytest

from oracle import mfcc_fixed as mx
from oracle import mfcc_float as mf

pytestmark = pytest.mark.gpu

TOL = 1e-4        # BASELINE.json north_star: <= 1e-4 rel-err vs the float notebook


def _err(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return (np.abs(got - ref).max() / np.abs(ref).max(),
            np.linalg.norm(got - ref) / np.linalg.norm(ref))


@pytest.fixture(scope="module")
def mfcc_amd():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import mfcc_amd
    return mfcc_amd


IMPLS = ["generic", "auto"]


@pytest.mark.parametrize("impl", IMPLS)
def test_float_golden_wav(mfcc_amd, wav_pcm, golden_dir, impl):
    """Config 1: f2bjrop1.0.wav against the fixture made by running the reference notebook."""
    ref = np.load(os.path.join(golden_dir, "f2bjrop_float64_cep32.npy"))[:, :13]
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, impl=impl) as m:
        got = m.process(wav_pcm)
    assert got.shape == (1046, 13) and got.dtype == np.float32
    e_max, e_l2 = _err(got, ref)
    assert e_max <= TOL and e_l2 <= TOL, (e_max, e_l2)
    # fp32 chain should be far inside the tolerance (~1.5e-6)
    assert e_max < 2e-5, e_max


@pytest.mark.parametrize("impl", IMPLS)
def test_float_all_32_coefficients(mfcc_amd, wav_pcm, golden_dir, impl):
    ref = np.load(os.path.join(golden_dir, "f2bjrop_float64_cep32.npy"))
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=32, impl=impl) as m:
        got = m.process(wav_pcm)
    e_max, e_l2 = _err(got, ref)
    assert e_max <= TOL and e_l2 <= TOL, (e_max, e_l2)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n", [512, 513, 681, 682, 683, 1023, 5000, 170 * 300 + 512 + 169])
def test_float_ragged_lengths_vs_oracle(mfcc_amd, n, impl):
    pcm = mf.synth_pcm(n, seed=n)
    ref = mf.mfcc_float_ref(pcm)
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, impl=impl) as m:
        got = m.process(pcm)
    assert got.shape == ref.shape
    e_max, e_l2 = _err(got, ref)
    assert e_max <= TOL and e_l2 <= TOL, (n, e_max, e_l2)


def test_float_too_short_gives_zero_frames(mfcc_amd):
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13) as m:
        assert m.process(np.zeros(511, np.int16)).shape == (0, 13)
        assert m.process(np.zeros((3, 100), np.int16)).shape == (3, 0, 13)


@pytest.mark.parametrize("impl", IMPLS)
def test_float_stream_padding(mfcc_amd, wav_pcm, impl):
    pcm = wav_pcm[:20000]
    ref = mf.mfcc_float_ref(pcm, pad_mode="stream")
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, pad_mode="stream", impl=impl) as m:
        got = m.process(pcm)
    assert got.shape == ref.shape == (mf.num_frames_stream(20000), 13)
    e_max, e_l2 = _err(got, ref)
    assert e_max <= TOL and e_l2 <= TOL


@pytest.mark.parametrize("impl", IMPLS)
def test_float_multichannel_and_device_path(mfcc_amd, impl):
    import torch
    pcm = np.stack([mf.synth_pcm(30000, seed=s) for s in range(5)])
    ref = mf.mfcc_float_ref(pcm)
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, impl=impl) as m:
        got = m.process(pcm)
        dev = m.process(torch.from_numpy(pcm).cuda())
        torch.cuda.synchronize()
    assert got.shape == ref.shape == (5, mf.num_frames_notebook(30000), 13)
    e_max, e_l2 = _err(got, ref)
    assert e_max <= TOL and e_l2 <= TOL
    assert np.array_equal(dev.cpu().numpy(), got)


@pytest.mark.parametrize("impl", IMPLS)
def test_float_halo_sharding_is_seamless(mfcc_amd, impl):
    """Frame-range shards with a 1-sample history halo reproduce the unsharded result (8e)."""
    import torch
    pcm = mf.synth_pcm(170 * 400 + 512, seed=11)
    x = torch.from_numpy(pcm).cuda()
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, impl=impl) as m:
        whole = m.process(x).cpu().numpy()
        f0, f1 = 123, 301                                   # frames [f0, f1)
        lo, hi = 170 * f0 - 1, 170 * (f1 - 1) + 512
        part = m.process(x[lo:hi].clone(), halo=1).cpu().numpy()
    assert part.shape == (f1 - f0, 13)
    assert np.array_equal(part, whole[f0:f1])


def test_fused_kernel_alignment_shifts_and_edges(mfcc_amd):
    """The fused 512/170/32 kernel fetches 16-byte aligned windows and shifts them by 0..7 samples:
    drive every shift (channel base and stride at all residues mod 8 samples), windows that stick
    out of the stream at both ends, a history halo, and enough tiles for the per-workgroup pipeline
    -- against the generic kernel (same fp32 contract) and, on a slice, the oracle."""
    import torch
    nfr = 16 * 37 + 5                                   # 37 full tiles + a ragged one per channel
    n = 170 * (nfr - 1) + 512
    nch = 9
    rng = np.random.default_rng(77)
    for base_off in range(8):
        stride = n + 3 + base_off                       # odd strides: every channel lands on another residue
        flat = torch.from_numpy(rng.integers(-32768, 32767, size=base_off + stride * nch + 64,
                                             dtype=np.int16)).cuda()
        view = torch.as_strided(flat, (nch, n), (stride, 1), storage_offset=base_off)
        assert (view.data_ptr() // 2) % 8 == (flat.data_ptr() // 2 + base_off) % 8
        for halo in (0, 1):
            with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, impl="fused512") as mfu, \
                    mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, impl="generic") as mge:
                a = mfu.process(view, halo=halo).cpu().numpy()
                b = mge.process(view, halo=halo).cpu().numpy()
            assert a.shape == b.shape == (nch, mf.num_frames_notebook(n - halo), 13)
            assert np.isfinite(a).all()
            e_max, e_l2 = _err(a, b)
            assert e_max <= 2e-5 and e_l2 <= 2e-5, (base_off, halo, e_max, e_l2)
            if halo == 0:
                a0 = a
    # oracle on one misaligned channel (uniform noise keeps every mel energy > 0)
    ref = mf.mfcc_float_ref(view[3].cpu().numpy())
    e_max, e_l2 = _err(a0[3], ref)
    assert e_max <= TOL and e_l2 <= TOL


@pytest.mark.parametrize("ncep", [1, 4, 12, 16, 17, 24, 32])
def test_fused_kernel_other_ncep_and_stream_padding(mfcc_amd, ncep):
    pcm = np.stack([mf.synth_pcm(170 * 50 + 512 + 37, seed=200 + s) for s in range(3)])
    ref = mf.mfcc_float_ref(pcm, n_cep=ncep, pad_mode="stream")
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=ncep, pad_mode="stream", impl="fused512") as m:
        assert m.kernel_name().startswith("mfcc_fused512")
        got = m.process(pcm)
    assert got.shape == ref.shape
    e_max, e_l2 = _err(got, ref)
    assert e_max <= TOL and e_l2 <= TOL


def test_float_lifter(mfcc_amd, wav_pcm):
    pcm = wav_pcm[:30000]
    ref = mf.lifter(mf.mfcc_float_ref(pcm, n_cep=32), 22)
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=32, lifter=22.0) as m:
        got = m.process(pcm)
    e_max, e_l2 = _err(got, ref)
    assert e_max <= TOL and e_l2 <= TOL


def test_float_config4_1024_40(mfcc_amd):
    """Config 4 shape at a small size: nfft 1024, hop 341, 40 mel, power scale = nfft."""
    pcm = np.stack([mf.synth_pcm(40000, seed=100 + s) for s in range(3)])
    ref = mf.mfcc_float_ref(pcm, nfft=1024, hop=341, n_mel=40, power_scale=1024.0)
    with mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=13, power_scale=0) as m:
        assert m.hop == 341
        got = m.process(pcm)
    assert got.shape == ref.shape
    e_max, e_l2 = _err(got, ref)
    assert e_max <= TOL and e_l2 <= TOL


def test_fused1024_kernel_alignment_shifts_and_edges(mfcc_amd):
    """Config-4 parameters (1024 / 341 / 40 mel, power scale = nfft) on the fused 1024 kernel: every
    alignment shift of the window fetch (odd hop: the shift changes from tile to tile too), misaligned
    channel strides, a history halo, stream ends, ragged tile counts -- against the generic kernel and,
    on one channel, the oracle."""
    import torch
    nfr = 16 * 19 + 7
    n = 341 * (nfr - 1) + 1024
    nch = 5
    rng = np.random.default_rng(41)
    kw = dict(nfft=1024, nfilters=40, nceptrums=13, power_scale=0)
    for base_off in range(8):
        stride = n + 5 + base_off
        flat = torch.from_numpy(rng.integers(-32768, 32767, size=base_off + stride * nch + 64,
                                             dtype=np.int16)).cuda()
        view = torch.as_strided(flat, (nch, n), (stride, 1), storage_offset=base_off)
        for halo in (0, 1):
            with mfcc_amd.MFCC(**kw) as mfu, mfcc_amd.MFCC(impl="generic", **kw) as mge:
                assert mfu.kernel_name().startswith("mfcc_fused1024") and mfu.hop == 341
                assert mge.kernel_name().endswith("generic_kernel")
                a = mfu.process(view, halo=halo).cpu().numpy()
                b = mge.process(view, halo=halo).cpu().numpy()
            assert a.shape == b.shape == (nch, (n - halo - 1024) // 341 + 1, 13)
            assert np.isfinite(a).all()
            e_max, e_l2 = _err(a, b)
            assert e_max <= 2e-5 and e_l2 <= 2e-5, (base_off, halo, e_max, e_l2)
            if halo == 0:
                a0 = a
    ref = mf.mfcc_float_ref(view[2].cpu().numpy(), nfft=1024, hop=341, n_mel=40, power_scale=1024.0)
    e_max, e_l2 = _err(a0[2], ref)
    assert e_max <= TOL and e_l2 <= TOL


@pytest.mark.parametrize("ncep", [1, 13, 16, 17, 32, 40])
def test_fused1024_other_ncep_stream_padding_and_lifter(mfcc_amd, ncep):
    pcm = np.stack([mf.synth_pcm(341 * 40 + 1024 + 55, seed=300 + s) for s in range(3)])
    ref = mf.mfcc_float_ref(pcm, n_cep=ncep, pad_mode="stream", nfft=1024, hop=341, n_mel=40, power_scale=1024.0)
    with mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=ncep, power_scale=0, pad_mode="stream") as m:
        # every coefficient count up to n_mel on the fused kernel (the reference tops keep nceptrums = nfilters)
        assert m.kernel_name().startswith("mfcc_fused1024")
        got = m.process(pcm)
        many = m.process_batch([pcm[0], pcm[1][:5000], pcm[2][:1023]])
        assert np.array_equal(many[0], got[0]) and many[2].shape == (1, ncep)
    assert got.shape == ref.shape
    e_max, e_l2 = _err(got, ref)
    assert e_max <= TOL and e_l2 <= TOL
    if ncep in (16, 32):
        with mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=ncep, power_scale=0, pad_mode="stream", lifter=22.0) as m:
            lif = m.process(pcm)
        e_max, e_l2 = _err(lif, np.stack([mf.lifter(r, 22) for r in ref]))
        assert e_max <= TOL and e_l2 <= TOL


def test_fused1024_config4_size_periodicity(mfcc_amd, wav_pcm):
    """Config 4 at 1 h x 4 channels of speech-like audio: frames repeat bit for bit with the period of the
    tiled input (341 q samples), one period checked against the oracle."""
    import torch
    q = 400
    period = wav_pcm[:341 * q]
    n = 16000 * 3600
    pcm = np.tile(period, n // len(period) + 1)[:n]
    x = torch.from_numpy(np.stack([pcm, pcm[::-1].copy(), pcm // 2, pcm])).cuda()
    with mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=13, power_scale=0) as m:
        out = m.process(x)
        a = out[0].cpu().numpy()
        assert torch.equal(out[0], out[3])
    nf = a.shape[0]
    assert nf == (n - 1024) // 341 + 1 == 168912
    assert np.array_equal(a[q + 1:nf - q], a[2 * q + 1:nf])
    ref = mf.mfcc_float_ref(pcm[:341 * (2 * q) + 1024], nfft=1024, hop=341, n_mel=40, power_scale=1024.0)[q:2 * q]
    e_max, e_l2 = _err(a[q:2 * q], ref)
    assert e_max <= TOL and e_l2 <= TOL


@pytest.mark.parametrize("sr", [8000, 11025, 22050, 32000, 44100, 48000])
def test_other_sample_rates_never_get_a_wrong_fused_kernel(mfcc_amd, sr):
    """The banded MFMA lists of the fused kernels are the band structure of the mel matrix at ONE sample rate; their
    table builders check every non-zero weight is covered.  The 512 kernel switches to its dense instantiation (all 32
    (k2, block) pairs, any rate).  The 1024 kernel has three set lists (<= 22.05 kHz, 32 kHz, 44.1 / 48 kHz) and picks the
    first that covers the rate's matrix."""
    x = mf.synth_pcm(30000, seed=2)
    for nfft, nmel in ((512, 32), (1024, 40)):
        with mfcc_amd.MFCC(nfft=nfft, nfilters=nmel, nceptrums=13, samplerate=sr, power_scale=0) as m:
            got = m.process(x)
            if nfft == 512:
                assert m.kernel_name().startswith("mfcc_fused512")
            else:
                assert m.kernel_name().startswith("mfcc_fused1024"), sr
        ref = mf.mfcc_float_ref(x, nfft=nfft, hop=nfft // 3, n_mel=nmel, sample_rate=sr, power_scale=float(nfft))
        e_max, e_l2 = _err(got, ref)
        assert e_max <= TOL and e_l2 <= TOL, (sr, nfft)


@pytest.mark.parametrize("sr", [8000, 11025, 22050, 32000, 44100, 48000])
def test_fused1024_schedules_of_other_rates_vs_generic_and_oracle(mfcc_amd, sr):
    """Each set list of the 1024 kernel: several channels at odd alignments, 32 coefficients, STREAM framing,
    against the float64 oracle and against the generic kernel on the device."""
    import torch
    nch, n = 3, 341 * 50 + 1024 + 77
    flat = np.zeros(5 + (n + 3) * nch + 64, np.int16)
    for c in range(nch):
        flat[5 + c * (n + 3): 5 + c * (n + 3) + n] = mf.synth_pcm(n, seed=500 + sr % 97 + c)
    view = torch.as_strided(torch.from_numpy(flat).cuda(), (nch, n), (n + 3, 1), storage_offset=5)
    kw = dict(nfft=1024, nfilters=40, nceptrums=32, samplerate=sr, power_scale=0, pad_mode="stream")
    with mfcc_amd.MFCC(**kw) as a, mfcc_amd.MFCC(impl="generic", **kw) as b:
        assert a.kernel_name().startswith("mfcc_fused1024")
        ga, gb = a.process(view).cpu().numpy(), b.process(view).cpu().numpy()
    for c in range(nch):
        ref = mf.mfcc_float_ref(flat[5 + c * (n + 3): 5 + c * (n + 3) + n], n_cep=32, nfft=1024, hop=341, n_mel=40,
                                sample_rate=sr, power_scale=1024.0, pad_mode="stream")
        for g in (ga[c], gb[c]):
            e_max, e_l2 = _err(g, ref)
            assert e_max <= TOL and e_l2 <= TOL, (sr, c)


def test_constructor_defaults_16_filters_run_on_the_fused_kernel(mfcc_amd, wav_pcm):
    """MFCC() = the core's constructor defaults (nfft 512, 16 filters, 16 coefficients, mfcc.py:20): a 16-filter
    bank is block 0 of the dense schedule with block 1 masked."""
    pcm = np.stack([wav_pcm[:40000], mf.synth_pcm(40000, seed=6)])
    for ncep, pad in ((16, "notebook"), (5, "stream")):
        ref = mf.mfcc_float_ref(pcm, n_cep=ncep, n_mel=16, pad_mode=pad)
        with mfcc_amd.MFCC(nceptrums=ncep, pad_mode=pad) as m, mfcc_amd.MFCC(nceptrums=ncep, pad_mode=pad, impl="generic") as mg:
            assert m.nfilters == 16 and m.kernel_name().startswith("mfcc_fused512")
            got = m.process(pcm)
            gen = mg.process(pcm)
        assert got.shape == ref.shape and np.isfinite(got).all()
        e_max, e_l2 = _err(got, ref)
        assert e_max <= TOL and e_l2 <= TOL
        e_max, e_l2 = _err(got, gen)
        assert e_max <= 2e-5


def test_float_linearity_property_full_size(mfcc_amd):
    """Size-independent property at config-2 size (10 min): scaling the input by 2 adds
    exactly 2*sqrt(32) to c0 (log2 of 4x power through the ortho DCT) and leaves c1.. unchanged."""
    pcm = (mf.synth_pcm(9_600_000, seed=0) // 4).astype(np.int16)
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13) as m:
        a = m.process(pcm)
        b = m.process((pcm * 2).astype(np.int16))
    assert a.shape == (56468, 13)
    assert np.isfinite(a).all()
    np.testing.assert_allclose(b[:, 0] - a[:, 0], 2 * np.sqrt(32.0), atol=2e-3)
    assert np.abs(b[:, 1:] - a[:, 1:]).max() < 2e-3
    # and a sample of frames against the oracle
    for f in [0, 1, 28000, 56467]:
        if f == 0:
            ref = mf.mfcc_float_ref(pcm[:512])[0:1]
        else:   # a chunk starting one hop earlier gives frame f the right pre-emphasis history
            ref = mf.mfcc_float_ref(pcm[170 * (f - 1): 170 * f + 512])[1:2]
        e_max, _ = _err(a[f:f + 1], ref)
        assert e_max <= TOL, (f, e_max)


def test_periodic_speech_stream_repeats_bit_for_bit(mfcc_amd, wav_pcm):
    """Size-independent property on a speech-like spectrum at config-2 size: tile a prefix of the
    golden wav whose length is a multiple of the hop (170 q samples).  Frame k + q sees exactly the
    samples of frame k (and, from the second period on, the same pre-emphasis history), so both
    kernels must repeat their output bit for bit with period q frames -- across tiles, alignment
    shifts and workgroups.  One period is checked against the oracles."""
    import torch
    q = 1000
    period = wav_pcm[:170 * q]
    reps = 9_600_000 // len(period) + 1
    pcm = np.tile(period, reps)[:9_600_000]
    x = torch.from_numpy(pcm).cuda()
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, pad_mode="stream") as m:
        fl = m.process(x).cpu().numpy()
        fx = m.process_fixed(x).cpu().numpy()
    nf = fl.shape[0]
    assert nf == 56469 and fx.shape == fl.shape
    last = nf - 5                                    # the zero-padded tail does not repeat
    # frame 0 has history 0, every later period starts with history = last sample of the period
    assert np.array_equal(fl[q + 1:last - q], fl[2 * q + 1:last])
    assert np.array_equal(fx[q + 1:last - q], fx[2 * q + 1:last])
    assert np.array_equal(fl[1:q], fl[q + 1:2 * q]) and np.array_equal(fx[1:q], fx[q + 1:2 * q])
    two = pcm[:170 * (2 * q) + 512]
    ref_fl = mf.mfcc_float_ref(two)[q:2 * q]
    e_max, e_l2 = _err(fl[q:2 * q], ref_fl)
    assert e_max <= TOL and e_l2 <= TOL
    ref_fx = mx.mfcc_fixed_ref(two, nceptrums=13, pad_mode="notebook")[q:2 * q]
    assert np.array_equal(fx[q:2 * q], ref_fx)


# ----------------------------------------------------------------------------- fixed

def test_fixed_bit_exact_golden_wav(mfcc_amd, wav_pcm):
    ref = mx.mfcc_fixed_ref(wav_pcm, nceptrums=32)
    with mfcc_amd.mfcc_open() as m:                    # host-driver constants, STREAM framing
        got = m.process_fixed(wav_pcm)
    assert got.shape == ref.shape == (1047, 32) and got.dtype == np.int16
    assert np.array_equal(got, ref)
    assert list(got[0, :13]) == [4059, 1164, -93, -353, -597, -289, -221, -201, -123, -202, -128, 40, -109]


@pytest.mark.parametrize("n", [0, 100, 512, 682, 5000])
def test_fixed_bit_exact_ragged(mfcc_amd, n):
    pcm = mf.synth_pcm(n, seed=n + 1)
    ref = mx.mfcc_fixed_ref(pcm, nceptrums=13)
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, pad_mode="stream") as m:
        got = m.process_fixed(pcm)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)


def test_fixed_bit_exact_extremes_and_channels(mfcc_amd):
    rng = np.random.default_rng(5)
    chans = [rng.integers(-32768, 32768, 6000).astype(np.int16),
             np.where(np.arange(6000) % 7 < 3, 32767, -32768).astype(np.int16),
             np.zeros(6000, np.int16),
             np.full(6000, -32768, np.int16),
             (3000 * np.sin(np.arange(6000) * 0.3)).astype(np.int16)]
    pcm = np.stack(chans)
    ref = mx.mfcc_fixed_ref(pcm, nceptrums=16)
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=16, pad_mode="stream") as m:
        got = m.process_fixed(pcm)
    assert np.array_equal(got, ref)


def test_fixed_bit_exact_config3_full_size(mfcc_amd):
    """Config 3: synthetic 10 min, STREAM framing, all 56 469 frames bit-exact."""
    import torch
    pcm = mf.synth_pcm(9_600_000, seed=0)
    ref = mx.mfcc_fixed_ref(pcm, nceptrums=13)
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, pad_mode="stream") as m:
        got = m.process_fixed(torch.from_numpy(pcm).cuda()).cpu().numpy()
    assert got.shape == ref.shape == (56469, 13)
    assert np.array_equal(got, ref)


def test_fixed_other_parameters(mfcc_amd):
    pcm = mf.synth_pcm(20000, seed=9)
    for nfft, nfil, ncep in [(256, 16, 16), (1024, 64, 32), (512, 16, 16), (256, 32, 13), (1024, 32, 32),
                             (128, 16, 16), (128, 8, 8), (64, 8, 8), (1024, 16, 5), (256, 8, 8)]:
        ref = mx.mfcc_fixed_ref(pcm, nfft=nfft, nfilters=nfil, nceptrums=ncep)
        with mfcc_amd.MFCC(nfft=nfft, nfilters=nfil, nceptrums=ncep, pad_mode="stream") as m:
            got = m.process_fixed(pcm)
        assert np.array_equal(got, ref), (nfft, nfil, ncep)


def test_fixed_fused_kernel_other_sample_rates(mfcc_amd):
    """The fused fixed-point kernel deals the filterbank's products out in pieces whose size and count follow the
    filter widths, i.e. the sample rate (10 bins / up to 4 lanes per filter at 16 kHz, 11 / 5 at 48 kHz: the third
    step of the segmented reduction): every rate bit for bit against the oracle, or refused where the oracle's
    filterbank asserts."""
    pcm = np.concatenate([mf.synth_pcm(30000, seed=21), np.full(700, -32768, np.int16), np.full(700, 32767, np.int16)])
    ran = 0
    for sr, nfil in [(r, 32) for r in (8000, 11025, 22050, 32000, 44100, 48000)] + [(r, 16) for r in (8000, 22050, 48000)]:
        with mfcc_amd.MFCC(nfft=512, nfilters=nfil, nceptrums=nfil, samplerate=sr, pad_mode="stream") as m:
            try:
                ref = mx.mfcc_fixed_ref(pcm, nfilters=nfil, nceptrums=nfil, sample_rate=float(sr))
            except AssertionError:
                with pytest.raises(mfcc_amd.MfccHipError) as e:
                    m.process_fixed(pcm)
                assert e.value.code == -105, sr
                continue
            assert m.kernel_name(fixed=True) == "mfcc_fixed512_kernel", (sr, nfil)
            assert np.array_equal(m.process_fixed(pcm), ref), (sr, nfil)
            ran += 1
    assert ran >= 6


@pytest.mark.parametrize("pad", ["stream", "notebook"])
def test_fixed_fused_kernel_with_the_constructor_default_of_16_filters(mfcc_amd, pad):
    """MFCC(nfft=512) with the constructor's 16 filters (mfcc.py:20-21): the fused fixed-point kernel's 16-filter
    instantiation -- a 64-point DCT FFT, log2 + DCT once per FOUR frames -- bit for bit, for every number of frames
    modulo four (the last pass of a wave is partly filled), several channels, full-scale inputs, n_cep 1..16."""
    rng = np.random.default_rng(23)
    for n_extra, ncep in ((0, 16), (170, 13), (341, 1), (511, 16), (3000, 7)):
        n = 512 + 170 * 9 + n_extra
        pcm = np.stack([mf.synth_pcm(n, seed=int(rng.integers(1 << 30))), rng.integers(-32768, 32768, n).astype(np.int16),
                        np.where(np.arange(n) % 7 < 3, 32767, -32768).astype(np.int16)])
        ref = mx.mfcc_fixed_ref(pcm, nfilters=16, nceptrums=ncep, pad_mode=pad)
        with mfcc_amd.MFCC(nfft=512, nfilters=16, nceptrums=ncep, pad_mode=pad) as m:
            assert m.kernel_name(fixed=True) == "mfcc_fixed512_kernel"
            got = m.process_fixed(pcm)
        assert got.shape == ref.shape and np.array_equal(got, ref), (n_extra, ncep, pad)


def test_fixed_filterbanks_the_rtl_cannot_stream_are_refused(mfcc_amd):
    """Filter points too dense for the streaming filterbank (filterbank.py:22-34,88-142): the RTL would emit fewer
    than n_mel values per frame (the oracle asserts on exactly these sets) -- UNSUPPORTED, never a made-up result."""
    for nfft, nfil in [(256, 64), (512, 64), (128, 32), (64, 16)]:
        with pytest.raises(AssertionError):
            mx.mfcc_fixed_ref(mf.synth_pcm(2000, seed=1), nfft=nfft, nfilters=nfil, nceptrums=4)
        with mfcc_amd.MFCC(nfft=nfft, nfilters=nfil, nceptrums=4, power_scale=0) as m:
            with pytest.raises(mfcc_amd.MfccHipError) as e:
                m.process_fixed(np.zeros(4000, np.int16))
            assert e.value.code == -105


def test_fixed_unsupported_is_an_error_not_a_fallback(mfcc_amd):
    with mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=13) as m:      # 4*40 is not a power of two
        with pytest.raises(mfcc_amd.MfccHipError) as e:
            m.process_fixed(np.zeros(4000, np.int16))
        assert e.value.code == -105


def test_convert_wav_to_mfcc_file(mfcc_amd, golden_dir, tmp_path, wav_pcm):
    """mfcc_convert(sess, x.wav, x.mfcc): raw int16 LE [frame][32] like software/main.c:162-165."""
    out = tmp_path / "f2.mfcc"
    with mfcc_amd.mfcc_open() as sess:
        assert mfcc_amd.mfcc_convert(sess, os.path.join(golden_dir, "f2bjrop1.0.wav"), str(out)) == 0
    raw = np.fromfile(out, dtype="<i2").reshape(-1, 32)               # view.py:24-25 / lift.py:35-36
    assert np.array_equal(raw, mx.mfcc_fixed_ref(wav_pcm, nceptrums=32))


def test_directory_walk_in_one_launch_writes_the_same_files(mfcc_amd, golden_dir, tmp_path, wav_pcm):
    """show_dir_content (main.c:206-247) as ragged batches: byte-identical .mfcc files to per-file conversion."""
    import shutil
    import wave
    d = tmp_path / "corpus" / "sub"
    d.mkdir(parents=True)
    shutil.copy(os.path.join(golden_dir, "f2bjrop1.0.wav"), d / "a.wav")
    rng = np.random.default_rng(8)
    for name, n in (("b.wav", 700), ("c.wav", 40000), ("d.wav", 300)):
        with wave.open(str(d.parent / name), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(rng.integers(-20000, 20000, size=n, dtype=np.int16).astype("<i2").tobytes())
    with mfcc_amd.mfcc_open() as sess:
        pairs = mfcc_amd.show_dir_content(sess, str(tmp_path / "corpus"))
        assert len(pairs) == 4
        for src, dst in pairs:
            one = str(tmp_path / "single.mfcc")
            assert mfcc_amd.mfcc_convert(sess, src, one) == 0
            assert open(dst, "rb").read() == open(one, "rb").read(), src
    raw = np.fromfile(d / "a.mfcc", dtype="<i2").reshape(-1, 32)
    assert np.array_equal(raw, mx.mfcc_fixed_ref(wav_pcm, nceptrums=32))


# ----------------------------------------------------------------------------- ragged batch

@pytest.mark.parametrize("pad_mode", ["notebook", "stream"])
def test_ragged_batch_is_bit_identical_to_per_utterance_calls(mfcc_amd, pad_mode, wav_pcm):
    """One launch over utterances of different lengths (packed at hop-multiple offsets with zero gaps)
    reproduces the per-utterance results bit for bit, float and fixed; too-short and empty utterances
    included; the golden wav among them is checked against the oracles."""
    rng = np.random.default_rng(12)
    lens = [0, 100, 511, 512, 513, 681, 682, 683, 5000, 170 * 37 + 512, 23456, 1, 170 * 200 + 512 + 169]
    utts = [rng.integers(-32768, 32767, size=n, dtype=np.int16) for n in lens] + [wav_pcm]
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, pad_mode=pad_mode) as m:
        fl = m.process_batch(utts)
        fx = m.process_batch(utts, fixed=True)
        assert len(fl) == len(fx) == len(utts)
        for u, a, b in zip(utts, fl, fx):
            ra, rb = m.process(u), m.process_fixed(u)
            assert a.shape == ra.shape and b.shape == rb.shape
            assert np.array_equal(a, ra, equal_nan=True), len(u)
            assert np.array_equal(b, rb), len(u)
        assert m.process_batch([]) == []
        # device-resident form: torch tensors in, views of one CUDA tensor out, same bits
        import torch
        dv = [torch.from_numpy(u).cuda() for u in utts]
        dfl = m.process_batch(dv)
        dfx = m.process_batch(dv, fixed=True)
        torch.cuda.synchronize()
        for a, b, da, db in zip(fl, fx, dfl, dfx):
            assert da.is_cuda and np.array_equal(da.cpu().numpy(), a, equal_nan=True)
            assert np.array_equal(db.cpu().numpy(), b)
    ref = mf.mfcc_float_ref(wav_pcm, pad_mode=pad_mode)
    e_max, e_l2 = _err(fl[-1], ref)
    assert e_max <= TOL and e_l2 <= TOL
    assert np.array_equal(fx[-1], mx.mfcc_fixed_ref(wav_pcm, nceptrums=13, pad_mode=pad_mode))


def test_ragged_batch_many_short_utterances(mfcc_amd):
    """Config-5 shape at a reduced count: 300 utterances of 10 s in one launch."""
    rng = np.random.default_rng(5)
    utts = [(rng.standard_normal(160000) * 3000).clip(-32768, 32767).astype(np.int16) for _ in range(300)]
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13) as m:
        got = m.process_batch(utts)
        for i in (0, 17, 299):
            assert got[i].shape == (m.num_frames(160000), 13) == (939, 13)
            assert np.array_equal(got[i], m.process(utts[i]))


@pytest.mark.gpu
@pytest.mark.parametrize("form,name", [("w12bf", "mfcc_fused1024_w12bf_kernel"), ("w12", "mfcc_fused1024_w12_kernel"),
                                       ("f32", "mfcc_fused1024_kernel"), ("bf16", "mfcc_fused1024_kernel")])
def test_fused1024_every_staging_of_the_kernel_against_the_oracle(mfcc_amd, monkeypatch, form, name):
    """The 1024 path has four forms -- twelve waves or eight in lockstep, the mel contraction on bf16-split or fp32 matrix
    instructions; the handle takes the twelve-wave bf16 form, MFCC_HIP_FUSED1024 (read when a handle is made) picks another
    for A/B runs.  All of them are held to the float contract here: ragged length, unaligned start, two channels."""
    monkeypatch.setenv("MFCC_HIP_FUSED1024", form)
    x = np.stack([mf.synth_pcm(16 * 341 * 7 + 1024 + 77, seed=11), mf.synth_pcm(16 * 341 * 7 + 1024 + 77, seed=12)])
    with mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=13, power_scale=0) as m:
        assert m.kernel_name() == name
        got = np.asarray(m.process(x))
    for c in range(2):
        ref = mf.mfcc_float_ref(x[c], nfft=1024, hop=341, n_mel=40, power_scale=1024.0)
        e_max, e_l2 = _err(got[c], ref)
        assert e_max <= TOL and e_l2 <= TOL, (form, c)


@pytest.mark.gpu
@pytest.mark.parametrize("ncep", [13, 32])
def test_a_frame_does_not_depend_on_what_else_is_in_its_tile(mfcc_amd, ncep):
    """The twelve-wave kernel's tail runs the DCT on bf16-split matrix instructions and, for a 16-frame tile that holds a
    frame with a silent band (-inf log-mel), the fp32 chain as well -- per FRAME: frames with a -inf take the fp32
    results, all others keep theirs.  So shifting the stream by five frames (other tile boundaries, other neighbours)
    must reproduce every frame bit for bit, -inf / NaN patterns included, next to stretches of silence."""
    x = mf.synth_pcm(170 * 400 + 512, seed=21).astype(np.int16)
    x[170 * 100:170 * 100 + 3000] = 0                      # frames 100..114 see zeros only, their neighbours partly
    x[170 * 250:170 * 250 + 700] = 0
    y = np.concatenate([mf.synth_pcm(170 * 5, seed=22).astype(np.int16), x])
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=ncep) as m:
        assert m.kernel_name() == "mfcc_fused512_w12_kernel"
        a = np.asarray(m.process(x))
        b = np.asarray(m.process(y))
    assert np.isneginf(a[:, 0]).sum() >= 10                # the silent frames are there: c0 = -inf like the notebook
    assert np.array_equal(a[1:], b[6:], equal_nan=True)
    ref = mf.mfcc_float_ref(x, n_cep=ncep)
    fin = np.isfinite(ref).all(axis=1)
    assert np.array_equal(np.isneginf(ref[:, 0]), np.isneginf(a[:, 0]))
    e_max, e_l2 = _err(a[fin], ref[fin])
    assert e_max <= TOL and e_l2 <= TOL
