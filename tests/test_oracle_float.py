"""The float oracle (oracle/mfcc_float.py) against the fixtures produced by running the
reference's notebook/MFCC.ipynb verbatim (tests/golden/make_golden.py) and against the
known answers stored in the notebook's own outputs.  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import mfcc_float as mf


@pytest.fixture(scope="module")
def ka(golden_dir):
    return json.load(open(os.path.join(golden_dir, "notebook_known_answers.json")))


@pytest.fixture(scope="module")
def stages(golden_dir):
    return np.load(os.path.join(golden_dir, "f2bjrop_float64_stages.npz"))


def test_wav_fixture_is_the_reference_file(golden_dir, ka, wav_pcm):
    h = hashlib.sha256(open(os.path.join(golden_dir, "f2bjrop1.0.wav"), "rb").read()).hexdigest()
    assert h == ka["meta"]["wav_sha256"] == \
        "d0d550ec1edf96771c12026fc2cd2f5471b3d64a2f9a80f0eb633b4884cefccd"
    assert len(wav_pcm) == 178240 and wav_pcm.dtype == np.int16


def test_restatement_is_bit_identical_to_notebook(golden_dir, wav_pcm):
    ref = np.load(os.path.join(golden_dir, "f2bjrop_float64_cep32.npy"))
    out = mf.mfcc_notebook(wav_pcm)
    assert out.shape == ref.shape == (1046, 32)
    # same numpy/scipy as the generator -> exact; otherwise last-ulp slack
    if np.__version__ == "2.2.6":
        assert np.array_equal(out, ref)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10)


def test_frame0_spot_values(wav_pcm):
    # SURVEY.md section 7 step 1 / BASELINE.md section 4
    want = [39.55645148236, 7.172379444316, -1.737353723798, -3.91194310788, -5.563047655899,
            -3.15527464509, -2.24471325048, -2.168805711875, -1.341579295162, -1.728438317826,
            -1.381398118029, 0.151029148639, -0.869253025614]
    out = mf.mfcc_float_ref(wav_pcm[:512 + 170 * 2])
    np.testing.assert_allclose(out[0], want, rtol=0, atol=1e-9)


def test_stage_fixtures(stages, wav_pcm):
    out, st = mf.mfcc_notebook(wav_pcm, return_stages=True)
    sel = stages["frames_sel"]
    np.testing.assert_array_equal(st["emphasis"][:4096], stages["emphasis"])
    np.testing.assert_array_equal(st["framed"][sel], stages["framed"])
    np.testing.assert_array_equal(st["window"], stages["window"])
    np.testing.assert_array_equal(st["filters"], stages["filters"])
    np.testing.assert_array_equal(st["dct_basis"], stages["dct_basis"])
    np.testing.assert_allclose(st["power"][sel], stages["power"], rtol=1e-13)
    np.testing.assert_allclose(st["logmel"][sel], stages["logmel"], rtol=0, atol=1e-11)


def test_known_answers_from_notebook_outputs(ka):
    pts, freqs = mf.get_filter_points(0, 8000.0, 32, 512, sample_rate=16000)
    assert list(pts) == ka["filter_points"] == ka["filter_points_int_nb"]
    np.testing.assert_allclose(freqs, ka["mel_center_freqs"], rtol=1e-8)
    f = mf.get_filters(pts, 512)
    assert f.shape == (32, 257)
    # NB cell 31 accumulates value*weight in a Python loop and prints int(sum)
    totals = []
    for row in f:
        s = 0
        for w in row:
            s += w * 1234
        totals.append(int(s))
    assert totals == ka["filter_total_1234"]
    # bins 0 and 256 carry no weight in any filter (used by the kernels)
    assert not f[:, 0].any() and not f[:, 256].any()


def test_int_notebook_agrees(golden_dir):
    a = np.load(os.path.join(golden_dir, "f2bjrop_float64_cep32.npy"))
    b = np.load(os.path.join(golden_dir, "f2bjrop_float64_intnb_dct32.npy"))
    assert np.abs(a - b).max() < 1e-11


def test_lifter_fixture(golden_dir):
    cep = np.load(os.path.join(golden_dir, "f2bjrop_float64_cep32.npy"))
    lif = np.load(os.path.join(golden_dir, "f2bjrop_float64_lifter32.npy"))
    np.testing.assert_allclose(mf.lifter(cep, 22), lif, rtol=1e-14)


def test_frame_counts():
    assert mf.num_frames_notebook(178240) == 1046
    assert mf.num_frames_stream(178240) == 1047
    assert mf.num_frames_notebook(9_600_000) == 56468
    assert mf.num_frames_stream(9_600_000) == 56469
    assert mf.num_frames_notebook(511) == 0 and mf.num_frames_notebook(512) == 1
    assert mf.num_frames_stream(0) == 1 and mf.num_frames_stream(511) == 1
    assert mf.num_frames_stream(512) == 2 and mf.num_frames_stream(512 + 170) == 3
    assert mf.num_frames_notebook(57_600_000, 1024, 341) == 168912


def test_stream_padding_and_channels(wav_pcm):
    x = wav_pcm[:2000]
    a = mf.mfcc_float_ref(x, pad_mode="stream")
    assert a.shape == (mf.num_frames_stream(2000), 13)
    b = mf.mfcc_float_ref(x, pad_mode="notebook")
    np.testing.assert_array_equal(a[:len(b)], b)
    two = mf.mfcc_float_ref(np.stack([x, x[::-1]]))
    assert two.shape == (2, len(b), 13)
    np.testing.assert_array_equal(two[0], b)


def test_synth_is_deterministic():
    a = mf.synth_pcm(4096, seed=3)
    b = mf.synth_pcm(4096, seed=3)
    assert a.dtype == np.int16 and np.array_equal(a, b) and a.std() > 2500
