"""Serial wire format + power gate: the C-ABI host functions against a
plain-Python restatement of software/serial.c:89-122, cepstrum.c:15-71,161-183 and
mfcc/misc/magic.py:9-41.  No GPU needed: these entry points never touch HIP."""
import numpy as np
import pytest

from mfcc_amd import wire


def _ref_pack(cep):
    out = bytearray()
    for row in cep:
        out += b"\xa5\x5a"                                      # magic.py: 0xa55a first
        out += row.astype(">i2").tobytes()                       # cepstrum.c:54-58: ntohs on receive
    return bytes(out)


def _ref_unpack(data, ncep):
    """expect_magic + cepstrum_get_column, byte by byte like the C (serial.c:97-119)."""
    pos, rows = 0, []
    while True:
        aligned = False
        while not aligned:
            while pos < len(data) and data[pos] != 0xA5:
                pos += 1
            if pos + 1 >= len(data):
                return np.array(rows, dtype=np.int16).reshape(-1, ncep)
            aligned = data[pos + 1] == 0x5A
            pos += 2
        if pos + 2 * ncep > len(data):
            return np.array(rows, dtype=np.int16).reshape(-1, ncep)
        rows.append(np.frombuffer(data[pos:pos + 2 * ncep], dtype=">i2").astype(np.int16))
        pos += 2 * ncep


def _ref_power(window, head):
    flat = window.reshape(-1).astype(np.int64)
    size, ncep = flat.size, window.shape[1]
    power = 0
    for i in range(size // 3, 2 * size // 3, ncep):              # cepstrum.c:166-179
        k = head + i
        if k >= size:
            k -= size
        power += int(flat[k]) ** 2
    return power


@pytest.mark.parametrize("ncep", [13, 16, 32])
def test_pack_matches_reference_layout_and_round_trips(ncep):
    rng = np.random.default_rng(ncep)
    cep = rng.integers(-32768, 32768, size=(57, ncep)).astype(np.int16)
    cep[3, 0] = np.int16(-23206)                                  # 0xa55a as a coefficient
    data = wire.pack_columns(cep)
    assert data == _ref_pack(cep)
    assert len(data) == 57 * 2 * (ncep + 1)
    got, used = wire.unpack_columns(data, ncep)
    assert used == len(data)
    assert np.array_equal(got, cep)


def test_unpack_resynchronises_like_expect_magic():
    rng = np.random.default_rng(5)
    cep = rng.integers(-2000, 2000, size=(20, 16)).astype(np.int16)
    clean = wire.pack_columns(cep)
    # garbage in front (with a lone 0xa5), a truncated column at the end
    dirty = bytes([0x00, 0xA5, 0x11, 0xA5, 0xA5, 0x77]) + clean + clean[:11]
    got, used = wire.unpack_columns(dirty, 16)
    ref = _ref_unpack(dirty, 16)
    assert np.array_equal(got, ref)
    assert len(got) >= 19                                         # at most one column lost to the garbage
    assert used <= len(dirty) - 11 + 2
    # streaming: feeding the rest later continues where the first call stopped
    got2, _ = wire.unpack_columns(dirty[used:] + clean[11:34], 16)
    assert np.array_equal(got2, _ref_unpack(dirty[used:] + clean[11:34], 16))
    # max_frames stops early
    got3, used3 = wire.unpack_columns(clean, 16, max_frames=4)
    assert np.array_equal(got3, cep[:4]) and used3 == 4 * 34
    empty, used0 = wire.unpack_columns(b"", 16)
    assert empty.shape == (0, 16) and used0 == 0


@pytest.mark.parametrize("ncep,nframes", [(16, 93), (13, 50), (32, 7)])
def test_eval_power_matches_cepstrum_c(ncep, nframes):
    rng = np.random.default_rng(ncep * nframes)
    win = rng.integers(-3000, 3000, size=(nframes, ncep)).astype(np.int16)
    for head in (0, ncep * 5 % (ncep * nframes), ncep * (nframes - 1)):
        power, loud = wire.cepstrum_eval_power(win, head)
        assert power == _ref_power(win, head)
        assert loud == (power >= wire.POWER_THRESHOLD)
    quiet = np.zeros((nframes, ncep), dtype=np.int16)
    assert wire.cepstrum_eval_power(quiet) == (0, False)
    loud = np.full((nframes, ncep), 30000, dtype=np.int16)
    p, ok = wire.cepstrum_eval_power(loud)
    assert ok and p == _ref_power(loud, 0)
    with pytest.raises(Exception):
        wire.cepstrum_eval_power(win, head=ncep * nframes)        # head outside the buffer
