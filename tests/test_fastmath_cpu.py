"""Host check of hmvec_amd/csrc/fastmath.hpp - the short fp64 log / exp / log1p the fused
radial-profile kernel evaluates the Battaglia integrand with (hmvec/hmvec.py:844-860,906-927) -
against 80-bit long double arithmetic.  The same source is compiled for the GPU; this test
also builds it with -fsanitize=address,undefined (sanitizers on the CPU build)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("fastmath") / "libfastmath_host.so"
    src = os.path.join(REPO, "tests", "cpp", "fastmath_host.cpp")
    # no contraction: the host build must only use the FMAs the source writes
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", str(out)],
                   check=True)
    return ctypes.CDLL(str(out))


def call(lib, name, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    getattr(lib, name)(ctypes.c_void_p(x.ctypes.data), ctypes.c_int(x.size), ctypes.c_void_p(out.ctypes.data))
    return out


def ulps(got, ref):
    ref = np.asarray(ref, dtype=np.longdouble)
    sp = np.spacing(np.abs(ref).astype(np.float64)).astype(np.longdouble)
    return float(np.max(np.abs(got.astype(np.longdouble) - ref) / sp))


def test_log_exp_log1p_within_two_ulp(lib):
    if np.finfo(np.longdouble).eps >= np.finfo(np.float64).eps:
        pytest.skip("no extended-precision long double on this host")
    rng = np.random.default_rng(1)
    x = np.concatenate([10 ** rng.uniform(-300, 300, 100000), rng.uniform(0.5, 2.0, 100000),
                        1 + rng.uniform(-1e-3, 1e-3, 50000), [1.0, 0.5, 2.0, np.sqrt(0.5)]])
    assert ulps(call(lib, "fm_log", x), np.log(x.astype(np.longdouble))) < 2.0
    y = np.concatenate([rng.uniform(-700, 700, 100000), rng.uniform(-1, 1, 100000),
                        rng.uniform(-1e-8, 1e-8, 1000), [0.0]])
    assert ulps(call(lib, "fm_exp", y), np.exp(y.astype(np.longdouble))) < 2.0
    a = np.concatenate([10 ** rng.uniform(-300, 300, 100000), rng.uniform(0, 3, 100000),
                        [1e-17, 2.0 ** -53, 2.0 ** -52]])
    assert ulps(call(lib, "fm_log1p", a), np.log1p(a.astype(np.longdouble))) < 2.0
    assert call(lib, "fm_log1p", np.zeros(1))[0] == 0.0
    sat = call(lib, "fm_exp", np.array([800.0, -800.0]))
    assert sat[0] == np.inf and sat[1] == 0.0
    # the integrand's forms: same bits without the clamp inside its range (and the same saturation far outside
    # the clamp's window); ln(1 + a) without the correction term is good to 1.2e-16 absolute + 2 ulp
    assert np.array_equal(call(lib, "fm_exp_nc", y), call(lib, "fm_exp", y))
    far = call(lib, "fm_exp_nc", np.array([5000.0, -5000.0, 1e8, -1e8]))
    assert np.all(far[[0, 2]] == np.inf) and np.all(far[[1, 3]] == 0.0)
    got = call(lib, "fm_log1p_abs", a).astype(np.longdouble)
    ref = np.log1p(a.astype(np.longdouble))
    assert float(np.max(np.abs(got - ref) - 2 * np.spacing(np.abs(ref).astype(np.float64)))) < 1.2e-16


def test_gnfw_integrand_through_fast_functions(lib):
    """amp t^g (1+t^a)^(-e) = amp exp(g ln t - e log1p(exp(a ln t))) at Battaglia-like parameters."""
    rng = np.random.default_rng(2)
    t = 10 ** rng.uniform(-4, 1.5, 20000)
    g, a, e = -0.2, rng.uniform(0.5, 2.5, t.size), rng.uniform(1.0, 6.0, t.size)
    lt = call(lib, "fm_log", t)
    rho = call(lib, "fm_exp_nc", g * lt - e * call(lib, "fm_log1p_abs", call(lib, "fm_exp_nc", a * lt)))
    tl = t.astype(np.longdouble)
    ref = tl ** g * (1 + tl ** a.astype(np.longdouble)) ** (-e.astype(np.longdouble))
    assert np.max(np.abs(rho / ref.astype(np.float64) - 1)) < 2e-14


@pytest.mark.parametrize("src,entry", [("fastmath_host.cpp", "fm_log"), ("ldsfft_host.cpp", "ldsfft_rfft_imag")])
def test_host_builds_are_clean_under_asan_ubsan(tmp_path, src, entry):
    """The host-compilable pieces of the native code, driven by a small C main under
    AddressSanitizer + UBSan (GPU ASan is not available on the pool)."""
    main = tmp_path / "main.cpp"
    if entry == "fm_log":
        body = """
extern "C" void fm_log(const double*, int, double*); extern "C" void fm_exp(const double*, int, double*);
extern "C" void fm_log1p(const double*, int, double*); extern "C" void fm_exp_nc(const double*, int, double*);
extern "C" void fm_log1p_abs(const double*, int, double*);
int main() { std::vector<double> x(4097), o(4097); for (int i = 0; i < 4097; ++i) x[i] = 1e-3 * (i + 1);
  fm_log(x.data(), 4097, o.data()); fm_exp(x.data(), 4097, o.data()); fm_log1p(x.data(), 4097, o.data());
  fm_exp_nc(x.data(), 4097, o.data()); fm_log1p_abs(x.data(), 4097, o.data());
  return o[7] > 0 ? 0 : 1; }"""
    else:
        body = """
extern "C" int ldsfft_rfft_imag(const double*, int, int, double*);
extern "C" int ldsfft_rfft_imag_spec2500(const double*, int, int, double*);
extern "C" int ldsfft_pruned_rfft_imag(const double*, int, int, int, int, double*);
extern "C" int ldsfft_chirp_rfft_imag(const double*, int, int, int, int, int, int, double*);
int main() { int rc = 0; for (int n : {4, 12, 600, 5000, 20000}) { std::vector<double> y(n), o(n / 2 + 1);
  for (int i = 0; i < n; ++i) y[i] = 1.0 / (1 + i); rc |= ldsfft_rfft_imag(y.data(), n, 512, o.data()); }
  // the compile-time plan's sequences: 3-of-5 butterflies on the compact source, full ones on it, all five passes
  for (int nz_from : {1, 375, 376, 625, 626, 2500}) { std::vector<double> y(5000, 0.0), o(2501);
    for (int i = 0; i < 2 * nz_from && i < 5000; ++i) y[i] = 1.0 / (1 + i);
    rc |= ldsfft_rfft_imag_spec2500(y.data(), nz_from, 512, o.data()); }
  // the long-grid routes (round 4): residue pairs of the pruned decomposition (odd and even R, mirrors needed or not)
  // and the chirp route with and without one-sided windows; buffers sized exactly, so any stray index trips ASan
  struct P { int n, LP, nz, jn; };
  for (P c : {P{30000, 1000, 1640, 394}, P{30000, 1000, 2000, 15000}, P{40000, 1250, 2400, 9000}, P{8000, 2000, 1, 1999},
              P{32768, 2048, 4000, 8191}, P{30000, 1500, 3000, 53}}) {
    std::vector<double> y(c.n, 0.0), o(c.n / 2 + 1);
    for (int i = 0; i < c.nz; ++i) y[i] = 1.0 / (1 + i);
    rc |= ldsfft_pruned_rfft_imag(y.data(), c.n, c.LP, 512, c.jn, o.data()); }
  struct Q { int n, LP, p0, nwin, jn; };
  for (Q c : {Q{30000, 1000, 820, 0, 590}, Q{30000, 1000, 820, 2, 2952}, Q{40000, 1250, 1216, 1, 1900}, Q{30000, 1000, 1000, 0, 0}}) {
    std::vector<double> y(c.n, 0.0), o(c.n / 2 + 1);
    for (int i = 0; i < 2 * c.p0; ++i) y[i] = 1.0 / (1 + i);
    rc |= ldsfft_chirp_rfft_imag(y.data(), c.n, c.LP, c.p0, c.nwin, 512, c.jn, o.data()); }
  return rc; }"""
    main.write_text("#include <vector>\n" + body)
    exe = tmp_path / "san"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    str(main), os.path.join(REPO, "tests", "cpp", src), "-o", str(exe)], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1")
    r = subprocess.run([str(exe)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
