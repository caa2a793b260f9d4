// Host harness for hmvec_amd/csrc/fastmath.hpp (the short log/exp/log1p of the profile integrand):
// array entry points for tests/test_fastmath_cpu.py.  Built with g++; not part of the product.
#include "../../hmvec_amd/csrc/fastmath.hpp"

extern "C" void fm_log(const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = hmg::log_fast(x[i]); }
extern "C" void fm_exp(const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = hmg::exp_fast(x[i]); }
extern "C" void fm_log1p(const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = hmg::log1p_fast(x[i]); }
// the forms the integrand uses: no exponent clamp, ln(1 + a) to absolute accuracy
extern "C" void fm_exp_nc(const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = hmg::exp_fast<false>(x[i]); }
extern "C" void fm_log1p_abs(const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = hmg::log1p_abs(x[i]); }
