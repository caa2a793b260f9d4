"""Default parameter tables for the halo-model hot path.

These are *inputs* to every kernel, so the numeric values are restated 1:1 from
the reference (hmvec/params.py:2-113); only values on the hot path
plus the cosmology scalars the analytic provider needs are carried.

``default_params`` is a flat dict merged with the user's ``params=`` at
construction (reference: hmvec/cosmology.py:60-62).
"""

# Battaglia 2016 gas-density fits (AGN / SH feedback families) and the
# Battaglia 2012 pressure fit.  Each triple is (A0, alpha_m, alpha_z) of
#   X = A0 * (M200c / 1e14 Msun)**alpha_m * (1+z)**alpha_z
battaglia_defaults = {
    "AGN": dict(
        rho0_A0=4000.0, rho0_alpham=0.29, rho0_alphaz=-0.66,
        alpha_A0=0.88, alpha_alpham=-0.03, alpha_alphaz=0.19,
        beta_A0=3.83, beta_alpham=0.04, beta_alphaz=-0.025,
    ),
    "SH": dict(
        rho0_A0=19000.0, rho0_alpham=0.09, rho0_alphaz=-0.95,
        alpha_A0=0.70, alpha_alpham=-0.017, alpha_alphaz=0.27,
        beta_A0=4.43, beta_alpham=0.005, beta_alphaz=0.037,
    ),
    "pres": dict(
        P0_A0=18.1, P0_alpham=0.154, P0_alphaz=-0.758,
        xc_A0=0.497, xc_alpham=-0.00865, xc_alphaz=0.731,
        beta_A0=4.35, beta_alpham=0.0393, beta_alphaz=0.415,
    ),
}

_mass_function = dict(
    st_A=0.3222, st_a=0.707, st_p=0.3, st_deltac=1.686,
    sigma2_kmin=1e-4, sigma2_kmax=2000, sigma2_numks=10000,
    Wkr_taylor_switch=0.01,
)

_profiles = dict(
    duffy_A_vir=7.85, duffy_alpha_vir=-0.081, duffy_beta_vir=-0.71,
    duffy_A_mean=10.14, duffy_alpha_mean=-0.081, duffy_beta_mean=-1.01,
    nfw_integral_numxs=40000, nfw_integral_xmax=200,
    electron_density_profile_integral_numxs=5000,
    electron_density_profile_integral_xmax=20,
    electron_pressure_profile_integral_numxs=5000,
    electron_pressure_profile_integral_xmax=20,
    battaglia_gas_gamma=-0.2, battaglia_gas_family="AGN",
    battaglia_pres_gamma=-0.3, battaglia_pres_alpha=1.0,
    battaglia_pres_family="pres",
)

_power = dict(kstar_damping=0.01, default_halofit="mead")

_cosmology = dict(
    omch2=0.1198, ombh2=0.02225, H0=67.3, ns=0.9645, As=2.2e-9,
    mnu=0.0, omk=0.0, pivot_scalar=0.05, w0=-1.0, tau=0.06, nnu=3.046,
    wa=0.0, num_massive_neutrinos=3, T_CMB=2.7255e6,
    parsec=3.08567758e16, mSun=1.989e30, thompson_SI=6.6524e-29,
    meterToMegaparsec=3.241e-23, Yp=0.24,
)

_hod = dict(
    hod_A_log10mthresh=1.0, hod_sig_log_mstellar=0.2, hod_alphasat=1.0,
    hod_Bsat=9.04, hod_betasat=0.74, hod_Bcut=1.65, hod_betacut=0.59,
    hod_bisection_search_min_log10mthresh=7.0,
    hod_bisection_search_max_log10mthresh=14.0,
    hod_bisection_search_rtol=1e-4,
    hod_bisection_search_warn_iter=20,
)

default_params = {}
for _grp in (_mass_function, _profiles, _power, _cosmology, _hod):
    default_params.update(_grp)
default_params["class_output"] = ""
del _grp
