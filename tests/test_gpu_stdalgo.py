"""GPU: the wave-parallel replay of libstdc++ std::sort / std::nth_element (slam_amd/csrc/wave_stdalgo.h) against
the real functions (called through the oracle library on the host), on key distributions with heavy ties."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(42)
    out = []
    for t in range(280):
        # 160 general sizes, then 120 small ones (4..200: the ranges the three-round-trip partition of round 3 takes)
        n = (int(rng.integers(1, 1800)) if t % 8 else int(rng.integers(2000, 8000))) if t < 160 else int(rng.integers(4, 200))
        spread = [3, 30, 250, n][t % 4]
        if t % 11 == 0:
            k = np.arange(n) % max(1, spread)            # sawtooth
        elif t % 13 == 0:
            k = (n - np.arange(n)) * spread // max(n, 1)  # descending runs
        elif t % 17 == 0:
            k = np.sort(rng.integers(0, spread + 1, n))   # ascending (worst case for desc comparator)
        elif t % 19 == 0:
            h = np.arange(n)
            k = np.where(h < n // 2, h, n - h)            # organ pipe
            k = np.where(h % 2 == 1, n - k, k) % 1000
        else:
            k = rng.integers(0, spread + 1, n)
        out.append(np.asarray(k, np.int64))
    return out


def test_wave_sort_and_select_replay_libstdcxx():
    from oracle import pyoracle
    from slam_amd import _lib
    L = _lib.load()
    O = pyoracle.lib()
    rng = np.random.default_rng(7)
    for t, k in enumerate(_cases()):
        n = len(k)
        for shift, kmax in ((16, 60000), (22, 1000)):
            keys = ((np.minimum(k, kmax).astype(np.uint32)) << shift) | (np.arange(n, dtype=np.uint32) & ((1 << shift) - 1))
            for nth in ([-1] if shift == 16 else [min(25, n - 1), int(rng.integers(0, n))]):
                want = keys.copy()
                O.orc_std_select_or_sort(want.ctypes.data_as(C.c_void_p), n, shift, nth)
                got = np.zeros_like(keys)
                rc = L.rvo_selftest_stdalgo(0, keys.ctypes.data_as(C.c_void_p), n, shift, nth, got.ctypes.data_as(C.c_void_p))
                assert rc == 0, L.rvo_last_error()
                assert np.array_equal(got, want), (t, n, shift, nth)
