"""A tabulated linear P(k, z) of the kind one saves from a Boltzmann-code run, whose z and k dependence does
NOT factorise (scale-dependent growth, as massive neutrinos give).  Shared by tools/make_golden.py - where a
stand-in ``camb.get_matter_power_interpolator`` serves it to the UNMODIFIED reference for accuracy='medium'
and 'high' (fixture case_e) - and by the tests that feed the same table to this repo's TabulatedBackground."""
import numpy as np


def table(ns):
    zt = np.linspace(0.0, 3.2, 17)
    kt = np.geomspace(5e-5, 3000.0, 600)
    x = kt / 0.02
    shape = 2.0e4 * x ** ns / (1.0 + x ** 2.9) ** 1.05 * (1.0 + 0.04 * np.sin(14.0 * np.log(kt)) * np.exp(-(kt / 0.3)))
    growth = np.exp(-0.75 * zt) / (1.0 + 0.1 * zt)
    nonsep = 1.0 + 0.25 * np.tanh(zt[:, None] - 1.0) * np.log10(1.0 + kt[None, :] / 0.05) / 5.0
    return zt, kt, shape[None, :] * growth[:, None] ** 2 * nonsep
