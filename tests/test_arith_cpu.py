"""The kernels replace divisions by multiplications with reciprocals made on the host; the
identities they rely on, checked over the ranges the kernels use them on."""
import numpy as np


def test_small_division_by_reciprocal_is_exact():
    """FAST: n / d == (n * (65536 / d + 1)) >> 16 for 0 <= n < 256, 1 <= d < 256
    (column within a run of cells, divided by the cell width)."""
    n = np.arange(256, dtype=np.int64)[None, :]
    d = np.arange(1, 256, dtype=np.int64)[:, None]
    m = 65536 // d + 1
    assert np.array_equal((n * m) >> 16, n // d)


def test_umulhi_division_is_exact_while_the_product_fits():
    """Level-0 copy: t / n == umulhi(t, 2^32 / n + 1) as long as t * n < 2^32 (piece index
    divided by the pieces per row: at most a few hundred, with up to 2^21 pieces)."""
    rng = np.random.default_rng(0)
    for n in list(range(2, 400)) + [1023, 4096, 65535]:
        m = (1 << 32) // n + 1
        assert m < (1 << 32)
        tmax = ((1 << 32) - 1) // n
        t = np.unique(np.concatenate([np.arange(0, min(tmax, 70000)), rng.integers(0, tmax + 1, 20000),
                                      np.array([tmax, tmax - 1, max(tmax - n, 0)])])).astype(object)
        got = np.array([(int(x) * m) >> 32 for x in t[:: max(1, len(t) // 3000)]])
        want = np.array([int(x) // n for x in t[:: max(1, len(t) // 3000)]])
        assert np.array_equal(got, want), n


def test_division_by_the_narrow_lds_pitch():
    """FAST: a byte offset in a view staged at 224 bytes per row -> its row,
    off / 224 == (off * 18725) >> 22 for every offset a 15-bit queue entry can hold."""
    off = np.arange(32768, dtype=np.int64)
    assert np.array_equal((off * 18725) >> 22, off // 224)
    assert int(off[-1]) * 18725 < 2 ** 32
