"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mfcc_hip.h declares,
its host-only helpers (frame counts, constant-table builders, error strings) agree with the oracle,
and -- with no GPU here -- the compute entry points fail loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mfcc_amd
from mfcc_amd import _lib as L
from oracle import mfcc_fixed as mx
from oracle import mfcc_float as mf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mfcc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mfcc_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(L.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), "libmfcc_hip.so does not export %s" % name
    # and the Python binding types exactly the declared set
    assert sorted(L.SYMBOLS) == declared


def test_abi_version_and_params_struct():
    lib = L.load()
    assert lib.mfcc_hip_abi_version() == 2
    p = L.Params()
    assert lib.mfcc_hip_default_params(C.byref(p)) == 0
    assert p.struct_size == C.sizeof(L.Params) == 64
    assert (p.nfft, p.hop, p.n_mel, p.n_cep, p.sample_rate) == (512, 170, 32, 13, 16000)
    assert p.power_scale == 512.0 and p.pad_mode == L.PAD_NOTEBOOK


def test_frame_counts_match_reference_conventions():
    for n in [0, 1, 511, 512, 513, 681, 682, 683, 5000, 178240, 9_600_000]:
        assert mfcc_amd.num_frames(n) == mf.num_frames_notebook(n)
        assert mfcc_amd.num_frames(n, pad_mode="stream") == mf.num_frames_stream(n)
    assert mfcc_amd.num_frames(57_600_000, nfft=1024, nfilters=40) == 168912       # config 4, hop 341


def test_invalid_parameters_are_rejected():
    lib = L.load()
    out = C.c_size_t(0)
    for kw in [dict(nfft=500), dict(nfft=2048), dict(nfilters=0), dict(nfilters=65),
               dict(nceptrums=0), dict(nceptrums=33), dict(hop=600), dict(samplerate=0)]:
        p = mfcc_amd.make_params(**kw)
        assert lib.mfcc_hip_num_frames(C.byref(p), 1000, C.byref(out)) == L.ERROR_INVALID_PARAM, kw
    p = mfcc_amd.make_params()
    p.struct_size = 12
    assert lib.mfcc_hip_num_frames(C.byref(p), 1000, C.byref(out)) == L.ERROR_INVALID_PARAM
    p = mfcc_amd.make_params()
    p.reserved[2] = 7
    assert lib.mfcc_hip_num_frames(C.byref(p), 1000, C.byref(out)) == L.ERROR_INVALID_PARAM
    assert b"invalid" in lib.mfcc_hip_strerror(L.ERROR_INVALID_PARAM)
    assert b"no CPU path" in lib.mfcc_hip_strerror(L.ERROR_NOT_FOUND)


@pytest.mark.parametrize("nfft,nmel", [(512, 32), (1024, 40), (256, 16)])
def test_float_tables_match_oracle(nfft, nmel):
    kw = dict(nfft=nfft, nfilters=nmel, nceptrums=13)
    w = mfcc_amd.get_table(L.TABLE_WINDOW_F32, **kw)
    np.testing.assert_allclose(w, mf.hamming_window(nfft), rtol=0, atol=6e-8)
    pts = mfcc_amd.get_table(L.TABLE_MEL_POINTS_I32, **kw)
    ref_pts, _ = mf.get_filter_points(0, 8000.0, nmel, nfft, sample_rate=16000)
    assert np.array_equal(pts, ref_pts)
    md = mfcc_amd.get_table(L.TABLE_MEL_DENSE_F32, **kw).reshape(nmel, nfft // 2 + 1)
    np.testing.assert_allclose(md, mf.mel_filterbank(nfft, nmel, 16000), rtol=0, atol=6e-8)
    d = mfcc_amd.get_table(L.TABLE_DCT_F32, **kw).reshape(13, nmel)
    np.testing.assert_allclose(d, mf.dct_basis(nmel, nmel)[:13], rtol=0, atol=3e-8)


def test_lifter_is_folded_into_the_dct_rows():
    d = mfcc_amd.get_table(L.TABLE_DCT_F32, nceptrums=32, lifter=22.0).reshape(32, 32)
    ref = mf.dct_basis(32, 32) * (1 + 11.0 * np.sin(np.pi * np.arange(32) / 22.0))[:, None]
    np.testing.assert_allclose(d, ref, rtol=0, atol=1e-6)


@pytest.mark.parametrize("nfft,nmel", [(512, 32), (256, 16), (1024, 64)])
def test_fixed_tables_match_oracle(nfft, nmel):
    kw = dict(nfft=nfft, nfilters=nmel, nceptrums=nmel // 2)
    assert np.array_equal(mfcc_amd.get_table(L.TABLE_FX_CURVE_I32, **kw), mx.window_curve(nfft))
    re, im = mx.twiddle_rom(nfft)
    tw = mfcc_amd.get_table(L.TABLE_FX_TWIDDLE_I32, **kw).reshape(-1, 2)
    assert np.array_equal(tw[:, 0], re) and np.array_equal(tw[:, 1], im)
    # the closed-form filterbank weights reproduce the RTL's streaming accumulators bit for bit
    fm = mfcc_amd.get_table(L.TABLE_FX_MEL_DENSE_U32, **kw)
    shift, W = int(fm[0]), fm[1:].reshape(nmel, nfft // 2).astype(np.uint64)
    P = np.random.default_rng(nfft).integers(0, 1 << 29, size=(40, nfft // 2)).astype(np.uint64)
    with np.errstate(over="ignore"):
        acc = (P[:, None, :] * W[None]).sum(-1)
    got = ((acc >> np.uint64(shift)) & np.uint64(0xFFFF)).astype(np.int64)
    assert np.array_equal(got, mx.filterbank(P.astype(np.int64), nfft, nmel))


def test_fixed_table_unsupported_parameters():
    with pytest.raises(mfcc_amd.MfccHipError) as e:
        mfcc_amd.get_table(L.TABLE_FX_MEL_DENSE_U32, nfft=1024, nfilters=40)     # 160-point FFT: not a power of two
    assert e.value.code == L.ERROR_UNSUPPORTED


def test_no_gpu_means_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mfcc_amd.MfccHipError) as e:
        mfcc_amd.MFCC()
    assert e.value.code == L.ERROR_NOT_FOUND
    with pytest.raises(mfcc_amd.MfccHipError):
        mfcc_amd.mfcc_open()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mfcc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_lifter_helper_matches_reference_formula():
    c = np.random.default_rng(0).standard_normal((7, 32))
    np.testing.assert_allclose(mfcc_amd.lifter(c, 22), mf.lifter(c, 22))
    assert mfcc_amd.lifter(c, 0) is not None and np.array_equal(mfcc_amd.lifter(c, 0), c)


def test_header_is_plain_c_and_the_integration_snippet_compiles(tmp_path):
    """include/mfcc_hip.h must be usable from the reference's C driver: C99, no C++; the replacement
    functions shown in INTEGRATION.md compile against it (syntax only: linking needs the GPU library)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "mfcc_hip.h")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                   check=True)
    src = tmp_path / "driver.c"
    src.write_text(r"""
#include <stdio.h>
#include <stdlib.h>
#include "mfcc_hip.h"
#define NFFT 512
#define STEPSIZE 170
#define NCEPSTRUMS 32
#define SAMPLERATE 16000
struct mfcc_s { mfcc_hip_handle *h; };
int mfcc_open(struct mfcc_s *sess) {
    mfcc_hip_params p;
    mfcc_hip_default_params(&p);
    p.nfft = NFFT; p.hop = STEPSIZE; p.n_mel = 32; p.n_cep = NCEPSTRUMS; p.sample_rate = SAMPLERATE;
    p.pad_mode = MFCC_HIP_PAD_STREAM;
    int ret = mfcc_hip_create(&p, &sess->h);
    if (ret) printf("Error mfcc_hip_create %d (%s)\n", ret, mfcc_hip_strerror(ret));
    return ret;
}
int mfcc_convert(struct mfcc_s *sess, const char *in, const char *out) {
    return mfcc_hip_convert_wav(sess->h, in, out, 1, NULL) ? -1 : 0;
}
int convert_dir(struct mfcc_s *sess, const char *const *in, const char *const *out, size_t n) {
    return mfcc_hip_convert_wavs(sess->h, in, out, n, 1, NULL);
}
int ragged(struct mfcc_s *sess, const int16_t *pcm, const size_t *off, size_t n, int16_t *cep, size_t cap, size_t *fo) {
    return mfcc_hip_process_ragged_fixed_i16(sess->h, pcm, off, n, cep, cap, fo);
}
int serial(const int16_t *cep, size_t nf, uint8_t *wire, size_t cap, long long *power) {
    if (mfcc_hip_serial_pack(cep, nf, NCEPSTRUMS, wire, cap)) return -1;
    return mfcc_hip_eval_power(cep, NCEPSTRUMS, (int)nf, 0, power);
}
void mfcc_close(struct mfcc_s *sess) { mfcc_hip_destroy(sess->h); }
""")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                    str(src)], check=True)


def test_lift_file_is_lift_py_on_a_file(tmp_path):
    """`.mfcc` -> `.lift` (software/lift.py:28-40), host only: lifter L = 22 in double, then `astype(np.int16)` --
    truncation toward zero, low 16 bits beyond int16 (full-range input makes n = 11's factor 12 overflow)."""
    rng = np.random.default_rng(3)
    cep = rng.integers(-32768, 32768, (70, 32)).astype(np.int16)
    cep[0, :] = 32767
    cep[1, :] = -32768
    src, dst = tmp_path / "x.mfcc", tmp_path / "x.lift"
    cep.tofile(src)
    assert mfcc_amd.lift_file(src, dst, nceptrums=32, L=22) == 70
    with np.errstate(invalid="ignore"):
        want = mf.lifter(cep, 22).astype(np.int16)                       # what lift.py writes
    assert np.array_equal(np.fromfile(dst, dtype=np.int16).reshape(-1, 32), want)
    # L <= 0: unchanged; a file that is not a whole number of rows is refused like np.reshape would
    assert mfcc_amd.lift_file(src, dst, nceptrums=32, L=0) == 70
    assert np.array_equal(np.fromfile(dst, dtype=np.int16).reshape(-1, 32), cep)
    cep.reshape(-1)[:-5].tofile(src)
    with pytest.raises(mfcc_amd.MfccHipError) as e:
        mfcc_amd.lift_file(src, dst, nceptrums=32)
    assert e.value.code == -101
    with pytest.raises(mfcc_amd.MfccHipError) as e:
        mfcc_amd.lift_file(tmp_path / "missing.mfcc", dst)
    assert e.value.code == -107


def test_stream_entry_points_check_their_arguments_without_a_gpu():
    import ctypes as C
    lib = mfcc_amd.load_library()
    s = C.c_void_p()
    assert lib.mfcc_hip_stream_create(None, 0, C.byref(s)) == -101 and not s.value
    assert lib.mfcc_hip_stream_push(None, None, 0, None, 0, None) == -101
    assert lib.mfcc_hip_stream_flush(None, None, 0, None) == -101
    assert lib.mfcc_hip_stream_reset(None) == -101
    assert lib.mfcc_hip_stream_pending(None) == 0 and lib.mfcc_hip_stream_max_frames(None, 100) == 0
    lib.mfcc_hip_stream_destroy(None)
    assert lib.mfcc_hip_abi_version() == 2


def test_fixed_support_follows_the_rtl_filterbank():
    """The fixed path exists where the RTL's streaming filterbank emits n_mel values per frame (the oracle asserts
    on exactly the other sets)."""
    from mfcc_amd import _lib
    for nfft, nmel, ok in ((512, 32, True), (256, 16, True), (1024, 64, True), (128, 16, True), (64, 8, True),
                           (256, 64, False), (512, 64, False), (128, 32, False), (64, 16, False)):
        try:
            mfcc_amd.get_table(_lib.TABLE_FX_MEL_DENSE_U32, nfft=nfft, nfilters=nmel, nceptrums=4)
            got = True
        except mfcc_amd.MfccHipError as e:
            assert e.code == -105
            got = False
        assert got == ok, (nfft, nmel)
        if ok:
            mx.mfcc_fixed_ref(np.zeros(nfft * 2, np.int16), nfft=nfft, nfilters=nmel, nceptrums=4)
        else:
            with pytest.raises(AssertionError):
                mx.mfcc_fixed_ref(np.zeros(nfft * 2, np.int16), nfft=nfft, nfilters=nmel, nceptrums=4)
