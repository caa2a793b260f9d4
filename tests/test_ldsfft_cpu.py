"""Host check of the workgroup FFT (hmvec_amd/csrc/ldsfft.hpp) that the fused radial-profile
kernel runs in LDS: the same per-thread code, sequenced thread by thread on the CPU, must
reproduce numpy's rfft imaginary part (what hmvec/fft.py:49 consumes)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("ldsfft") / "libldsfft_host.so"
    src = os.path.join(REPO, "tests", "cpp", "ldsfft_host.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", str(out)], check=True)
    lib = ctypes.CDLL(str(out))
    lib.ldsfft_rfft_imag.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.ldsfft_rfft_imag.restype = ctypes.c_int
    lib.ldsfft_rfft_imag_spec2500.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.ldsfft_rfft_imag_spec2500.restype = ctypes.c_int
    lib.ldsfft_pruned_rfft_imag.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p]
    lib.ldsfft_pruned_rfft_imag.restype = ctypes.c_int
    lib.ldsfft_chirp_rfft_imag.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.ldsfft_chirp_rfft_imag.restype = ctypes.c_int
    lib.ldsfft_band_rfft_imag.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p]
    lib.ldsfft_band_rfft_imag.restype = ctypes.c_int
    return lib


@pytest.mark.parametrize("n,threads", [(5000, 512), (5000, 256), (4, 64), (8, 64), (600, 128), (1000, 512),
                                       (4000, 512), (2 * 3 * 5 * 7 * 2, 64), (12, 64), (20000, 512), (64, 64)])
def test_matches_numpy_rfft(lib, n, threads):
    rng = np.random.default_rng(n)
    y = rng.standard_normal(n) * np.exp(-np.linspace(0, 6, n))
    out = np.zeros(n // 2 + 1)
    rc = lib.ldsfft_rfft_imag(y.ctypes.data, n, threads, out.ctypes.data)
    has_big_prime = any(n // 2 % p == 0 for p in (7, 11, 13))
    if has_big_prime:
        assert rc == 2          # unsupported factor -> caller falls back to rocFFT
        return
    assert rc == 0
    ref = np.fft.rfft(y).imag
    scale = np.max(np.abs(np.fft.rfft(y)))
    assert np.max(np.abs(out - ref)) < 4e-15 * scale * max(1.0, np.log2(n))


def test_odd_length_rejected(lib):
    y = np.ones(15)
    out = np.zeros(8)
    assert lib.ldsfft_rfft_imag(y.ctypes.data, 15, 64, out.ctypes.data) == 1


@pytest.mark.parametrize("nonzero", [1, 300, 674, 749, 750, 900, 1249, 1250, 1251, 3000, 5000])
def test_compile_time_plan_sequence_with_truncated_rows(lib, nonzero):
    """The nxs = 5000 plan as the fused kernel sequences it (hmgrid.hip, SPECM = 2500): rows that are zero from real
    sample `nonzero` on take the pruned first pass with the compact source and - below sample 750 - the 3-of-5 butterfly
    over samples 0..374 only; the rest of the buffer holds garbage there.  Same Im F as the run-time plan, to rounding."""
    rng = np.random.default_rng(nonzero)
    y = np.zeros(5000)
    y[:nonzero] = rng.standard_normal(nonzero) * np.exp(-np.linspace(0, 3, nonzero))
    nz_from = (nonzero + 1) // 2                      # first packed sample that is zero
    out, gen = np.zeros(2501), np.zeros(2501)
    assert lib.ldsfft_rfft_imag_spec2500(y.ctypes.data, nz_from, 512, out.ctypes.data) == 0
    assert lib.ldsfft_rfft_imag(y.ctypes.data, 5000, 512, gen.ctypes.data) == 0
    ref = np.fft.rfft(y)
    scale = np.max(np.abs(ref))
    assert np.max(np.abs(out - ref.imag)) < 4e-15 * scale * np.log2(5000)
    # pruning only drops terms that are exactly zero: the two sequences agree as numbers
    assert np.array_equal(out, gen)


@pytest.mark.parametrize("n,LP,nonzero,jn", [
    (30000, 1000, 1639, 394), (30000, 1000, 2000, 5522), (30000, 1000, 1200, 3), (30000, 1000, 1999, 15000),
    (30000, 1000, 900, 7499), (30000, 1000, 900, 7500), (30000, 1000, 900, 7501), (30000, 1250, 2500, 800),
    (30000, 1500, 3000, 53), (30000, 2500, 4100, 2000), (40000, 1250, 2400, 9000), (40000, 2000, 3999, 16000),
    (40000, 2500, 5000, 20000), (40000, 1000, 1, 0), (10000, 1000, 1366, 1220), (10000, 2500, 683, 2210),
    (32768, 1024, 2048, 700), (32768, 2048, 4000, 8191), (5000, 1250, 2500, 1249), (6000, 1000, 2000, 33),
    (8000, 2000, 1, 1999), (4096, 1024, 7, 1024)])
def test_pruned_decomposition_matches_numpy(lib, n, LP, nonzero, jn):
    """A row of n real samples that is zero from sample `nonzero` on, transformed as R = n/2/LP pairs of length-LP
    transforms (the long-grid route of the fused profile kernel): every mode j <= jn - and every mirror M - j <= jn -
    equals numpy's rfft, for odd and even R, LP, the self-paired residues 0 and R/2, and any jn."""
    rng = np.random.default_rng(n + LP + nonzero)
    y = np.zeros(n)
    y[:nonzero] = rng.standard_normal(nonzero) * np.exp(-np.linspace(0, 3, nonzero))
    M = n // 2
    out = np.zeros(M + 1)
    assert lib.ldsfft_pruned_rfft_imag(y.ctypes.data, n, LP, 512, jn, out.ctypes.data) == 0
    ref = np.fft.rfft(y)
    scale = np.max(np.abs(ref))
    need = np.arange(1, min(jn, M - 1) + 1)
    assert need.size == 0 or np.max(np.abs(out[need] - ref.imag[need])) < 4e-15 * scale * np.log2(n)
    assert not np.any(np.isnan(out[need]))


def test_pruned_decomposition_rejects_what_it_cannot_take(lib):
    y = np.ones(30000)
    out = np.zeros(15001)
    assert lib.ldsfft_pruned_rfft_imag(y.ctypes.data, 30000, 1000, 512, 100, out.ctypes.data) == 4    # support too long
    assert lib.ldsfft_pruned_rfft_imag(y.ctypes.data, 30000, 2000, 512, 100, out.ctypes.data) == 2    # LP does not divide M


@pytest.mark.parametrize("n,LP,p0,nonzero,nwin,jn", [
    (30000, 1000, 820, 1639, 0, 394), (30000, 1000, 820, 1640, 0, 590), (30000, 1000, 820, 100, 0, 1),
    (30000, 1000, 1000, 2000, 0, 500), (30000, 1000, 700, 1399, 0, 650), (40000, 1250, 1200, 2400, 0, 650),
    (40000, 1250, 1250, 2500, 0, 625), (40000, 1250, 900, 1777, 0, 0), (10000, 1000, 683, 1366, 0, 658),
    (30000, 1000, 820, 1639, 2, 591), (30000, 1000, 820, 1639, 2, 1771), (30000, 1000, 820, 1639, 2, 1772),
    (30000, 1000, 820, 1639, 2, 2952), (30000, 1000, 832, 1660, 1, 1200), (40000, 1250, 1216, 2431, 2, 3200),
    (30000, 1000, 820, 1639, 2, 300)])
def test_chirp_route_matches_numpy(lib, n, LP, p0, nonzero, nwin, jn):
    """Rows that need few modes: the chirp transform (a forward transform of length 2 LP, one more per window) gives
    every mode j <= jn of a row that is zero from packed sample p0 on - numpy's rfft to rounding (relative to the
    largest mode) - in the central window (jn <= Jw) and in the one-sided window pairs beyond it."""
    rng = np.random.default_rng(n + LP + p0 + jn)
    y = np.zeros(n)
    y[:nonzero] = rng.standard_normal(nonzero) * np.exp(-np.linspace(0, 3, nonzero))
    M = n // 2
    out = np.zeros(M + 1)
    assert jn <= lib.ldsfft_chirp_window(n, LP, p0, nwin)
    assert lib.ldsfft_chirp_rfft_imag(y.ctypes.data, n, LP, p0, nwin, 512, jn, out.ctypes.data) == 0
    ref = np.fft.rfft(y)
    scale = np.max(np.abs(ref))
    need = np.arange(1, jn + 1)
    assert need.size == 0 or np.max(np.abs(out[need] - ref.imag[need])) < 1e-14 * scale * np.log2(n)


def test_chirp_route_refuses_modes_outside_its_window(lib):
    y = np.zeros(30000)
    y[:1600] = 1.0
    out = np.zeros(15001)
    jw = lib.ldsfft_chirp_window(30000, 1000, 820, 0)
    assert jw == 590 and lib.ldsfft_chirp_window(30000, 1000, 820, 2) == 590 + 2 * 1181
    assert lib.ldsfft_chirp_rfft_imag(y.ctypes.data, 30000, 1000, 820, 0, 512, jw + 1, out.ctypes.data) == 5
    assert lib.ldsfft_chirp_rfft_imag(y.ctypes.data, 30000, 1000, 700, 0, 512, 10, out.ctypes.data) == 4    # support > p0


@pytest.mark.parametrize("n,LB,nonzero,jn", [(30000, 1000, 20491, 223), (30000, 1000, 30000, 498), (30000, 1000, 12000, 1),
                                             (30000, 1000, 25000, 0), (40000, 1250, 33333, 600), (32768, 1024, 32768, 255),
                                             (4000, 1000, 4000, 499), (30000, 1000, 17, 300)])
def test_narrow_band_route_matches_numpy(lib, n, LB, nonzero, jn):
    """Rows whose support does not prune (the tSZ notebook's pressure profile fills two thirds of the grid) but which
    need few modes: D = n/2/LB transforms of length LB of the decimated row, accumulated per mode with a running
    twiddle, give every mode j <= jn (2 jn + 2 <= LB) - numpy's rfft to rounding."""
    rng = np.random.default_rng(n + LB + nonzero + jn)
    y = np.zeros(n)
    y[:nonzero] = rng.standard_normal(nonzero) * np.exp(-np.linspace(0, 2, nonzero))
    M = n // 2
    out = np.zeros(M + 1)
    assert lib.ldsfft_band_rfft_imag(y.ctypes.data, n, LB, 512, jn, out.ctypes.data) == 0
    ref = np.fft.rfft(y)
    scale = np.max(np.abs(ref))
    need = np.arange(1, jn + 1)
    assert need.size == 0 or np.max(np.abs(out[need] - ref.imag[need])) < 1e-14 * scale * np.log2(n)
    assert lib.ldsfft_band_rfft_imag(y.ctypes.data, n, LB, 512, LB // 2, out.ctypes.data) == 5      # band too wide
