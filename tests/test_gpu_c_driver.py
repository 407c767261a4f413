"""A plain C program (tests/c_driver.c: the reference host driver's flow, software/main.c:36-177,249-277, on the
C ABI) linked against libmfcc_hip.so and run as a child process -- no Python, no torch in that process, so it
also shows which HIP runtime a C caller binds (the library's RUNPATH: /opt/rocm).  The `.mfcc` / `.lift` files
it writes are compared byte for byte with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import mfcc_fixed as mx
from oracle import mfcc_float as mf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "bin", "c_driver")


@pytest.fixture(scope="module")
def c_driver():
    if not os.path.exists(EXE):                  # normally built by __graft_entry__.build() and shipped with the tree
        import shutil
        if shutil.which("gcc") is None:
            pytest.skip("no gcc on this host and no prebuilt tests/bin/c_driver")
        import __graft_entry__ as g
        g.build_c_driver()
    return EXE


def _run(*args):
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    r = subprocess.run(list(args), capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    return r.stdout


def test_c_driver_converts_the_golden_wav(c_driver, golden_dir, tmp_path, wav_pcm):
    wav = os.path.join(golden_dir, "f2bjrop1.0.wav")
    out = str(tmp_path / "f2bjrop1.0.mfcc")
    log = _run(c_driver, "convert", wav, out)
    assert "frames 1047" in log and "fixed512" in log, log
    raw = np.fromfile(out, dtype=np.int16).reshape(-1, 32)                 # view.py:24-25 / lift.py:35-36
    ref = mx.mfcc_fixed_ref(wav_pcm, nceptrums=32)                         # STREAM framing: 1047 frames
    assert raw.shape == ref.shape == (1047, 32)
    assert np.array_equal(raw, ref)

    # the same file through the streaming session, with the driver's own transfer pattern (512, then 170)
    # and with a fixed chunk size
    for chunk in ("0", "1000"):
        out_s = str(tmp_path / ("stream%s.mfcc" % chunk))
        _run(c_driver, "stream", wav, out_s, chunk)
        assert open(out_s, "rb").read() == open(out, "rb").read(), chunk

    # .mfcc -> .lift (software/lift.py:28-40): lifter L = 22, astype(int16)
    lift = str(tmp_path / "f2bjrop1.0.lift")
    _run(c_driver, "lift", out, lift)
    want = mf.lifter(raw, 22).astype(np.int16)
    assert np.array_equal(np.fromfile(lift, dtype=np.int16).reshape(-1, 32), want)


def test_float_mfcc_file_truncates_like_astype_int16(c_driver, golden_dir, tmp_path, wav_pcm):
    """`fixed = 0`: the float coefficients written as int16, truncated toward zero like `astype(np.int16)`
    (software/lift.py:39).  Exact against the GPU's own fp32 output; against the float64 oracle a coefficient
    within fp32 rounding of an integer may land on the other side of it."""
    import mfcc_amd
    wav = os.path.join(golden_dir, "f2bjrop1.0.wav")
    out = str(tmp_path / "float.mfcc")
    _run(c_driver, "convertf", wav, out)
    raw = np.fromfile(out, dtype=np.int16).reshape(-1, 32)
    with mfcc_amd.mfcc_open() as m:
        f32 = m.process(wav_pcm)
        out2 = str(tmp_path / "float2.mfcc")
        assert m.convert(wav, out2, fixed=False) == 1047
    assert np.array_equal(raw, np.trunc(f32).astype(np.int16))
    assert open(out2, "rb").read() == open(out, "rb").read()
    ref = mf.mfcc_float_ref(wav_pcm, n_cep=32, pad_mode="stream")
    want = ref.astype(np.int16)
    diff = raw.astype(np.int32) - want
    assert np.abs(diff).max() <= 1 and (diff != 0).mean() < 1e-3
    # values that are not within 1e-3 of an integer must agree exactly
    safe = np.abs(ref - np.round(ref)) > 1e-3
    assert np.array_equal(raw[safe], want[safe])
