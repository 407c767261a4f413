"""BASELINE.json configs 4 and 5 at FULL size on one GPU, inputs generated on the device.

config 4: 64 channels x 1 h x 16 kHz, nfft 1024 / hop 341 / 40 mel / 13 coeff -> 10 810 368 frames
config 5: 10 000 utterances x 10 s (160 000 samples), 512 / 170 / 32 / 13, ONE ragged launch -> 9.39 M frames

Size-independent properties (periodicity of a tiled input, duplicate channels, batch == per-utterance calls)
cover every frame; the oracle covers one period per distinct channel / a sample of utterances."""
import numpy as np
import pytest

from oracle import mfcc_fixed as mx
from oracle import mfcc_float as mf

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def mfcc_amd():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import mfcc_amd
    return mfcc_amd


def _err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / np.abs(ref).max(), np.linalg.norm(got - ref) / np.linalg.norm(ref)


def test_config4_full_size_64_channels_one_hour(mfcc_amd, wav_pcm):
    import torch
    q, nch, n = 400, 64, 16000 * 3600
    plen = 341 * q                                   # a period of q hops: frame k + q == frame k, bit for bit
    kinds = 8
    periods = [wav_pcm[:plen].copy()] + [mf.synth_pcm(plen, seed=100 + i) for i in range(1, kinds)]
    periods[2] = (periods[2] // 16).astype(np.int16)            # a quiet channel
    reps = n // plen + 1
    pcm = torch.empty((nch, n), dtype=torch.int16, device="cuda")
    for c in range(nch):                              # channel c repeats distinct input c % 8
        pcm[c] = torch.from_numpy(periods[c % kinds]).cuda().repeat(reps)[:n]
    with mfcc_amd.MFCC(nfft=1024, nfilters=40, nceptrums=13, power_scale=0) as m:
        assert m.kernel_name().startswith("mfcc_fused1024")
        out = m.process(pcm)
        torch.cuda.synchronize()
    nf = out.shape[1]
    assert tuple(out.shape) == (64, 168912, 13) and out.shape[0] * nf == 10_810_368
    assert bool(torch.isfinite(out).all())
    for c in range(kinds):
        # every duplicate of this input gives the same bits
        for d in range(c + kinds, nch, kinds):
            assert torch.equal(out[c], out[d]), (c, d)
        # periodicity over the whole hour (frame 0 has no history, so start one period in)
        assert torch.equal(out[c, q:nf - q], out[c, 2 * q:nf]), c
        # one period against the float64 notebook restatement
        a = out[c, q:2 * q].cpu().numpy()
        x = np.tile(periods[c], 3)[:341 * 2 * q + 1024]
        ref = mf.mfcc_float_ref(x, nfft=1024, hop=341, n_mel=40, power_scale=1024.0)[q:2 * q]
        e_max, e_l2 = _err(a, ref)
        assert e_max <= TOL and e_l2 <= TOL, (c, e_max, e_l2)


@pytest.fixture(scope="module")
def corpus():
    """config 5's corpus on the device: 10 000 utterances x 160 000 samples, seed = utterance id"""
    import torch
    n_utt, n = 10_000, 160_000
    g = torch.Generator(device="cuda")
    flat = torch.empty(n_utt * n, dtype=torch.int16, device="cuda")
    for u in range(n_utt):
        g.manual_seed(u)
        flat[u * n:(u + 1) * n] = (torch.randn(n, generator=g, device="cuda") * 3000.0).clamp_(-32768, 32767).to(torch.int16)
    yield flat
    del flat
    torch.cuda.empty_cache()


@pytest.mark.parametrize("fixed", [False, True])
@pytest.mark.parametrize("ragged", [False, True])
def test_config5_full_size_10k_utterances_one_launch(mfcc_amd, corpus, fixed, ragged):
    """ragged=False: the corpus as defined (equal lengths: runs as 10 000 channels of one launch, no packing copy).
    ragged=True: utterance u cut to 160 000 - 997 (u mod 5) samples -- the packed-stream path (pack kernel, one
    launch over 9.3 M frames of ONE stream, row gather) at the same scale."""
    import torch
    n_utt, n = 10_000, 160_000
    flat = corpus
    lens = [n - 997 * (u % 5) if ragged else n for u in range(n_utt)]
    utts = [flat[u * n:u * n + lens[u]] for u in range(n_utt)]
    pad = "stream" if fixed else "notebook"
    with mfcc_amd.MFCC(nfft=512, nfilters=32, nceptrums=13, pad_mode=pad) as m:
        per = [m.num_frames(v) for v in sorted(set(lens), reverse=True)]
        assert per[0] == (940 if fixed else 939)
        got = m.process_batch(utts, fixed=fixed)
        torch.cuda.synchronize()
        total = sum(m.num_frames(v) for v in lens)
        assert len(got) == n_utt and sum(len(r) for r in got) == total and total > 9_200_000
        esz = 2 if fixed else 4
        assert got[-1].data_ptr() == got[0].data_ptr() + (total - len(got[-1])) * 13 * esz    # one dense result buffer
        dense = torch.as_strided(got[0], (total, 13), (13, 1))
        assert bool(torch.isfinite(dense.float()).all())
        if not ragged:
            # equal lengths: the plain multi-channel call gives the same bits
            multi = (m.process_fixed if fixed else m.process)(flat.view(n_utt, n)[:2000])
            for u in (0, 1, 999, 1999):
                assert torch.equal(multi[u], got[u]), u
        # a sample of utterances: per-utterance calls (bit for bit) and the oracle
        for u in (0, 1, 2, 4999, 5000, 7777, 9998, 9999):
            x = utts[u].cpu().numpy()
            one = (m.process_fixed if fixed else m.process)(x)
            a = got[u].cpu().numpy()
            assert np.array_equal(a, one), u
            if fixed:
                assert np.array_equal(a, mx.mfcc_fixed_ref(x, nceptrums=13, pad_mode="stream")), u
            else:
                e_max, e_l2 = _err(a, mf.mfcc_float_ref(x))
                assert e_max <= TOL and e_l2 <= TOL, (u, e_max, e_l2)
