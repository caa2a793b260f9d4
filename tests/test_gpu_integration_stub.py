"""INTEGRATION.md section 2 is executable documentation: this test lifts the ctypes stub out of the
markdown verbatim, points it at the in-tree library, feeds it numpy arrays the way a patched
hmvec.HaloModel would (nothing from hmvec_amd's Python layer), and checks the spectra it returns
against the CPU oracle."""
import os
import re
import types

import numpy as np
import pytest

from conftest import REPO, merged_params, power_close

pytestmark = pytest.mark.gpu


def test_integration_md_stub_runs_as_written():
    from oracle import hmref
    from hmvec_amd.quadrature import trapz_weights          # host helper named in INTEGRATION.md's mapping table
    import hmvec_amd as hm                                   # only for the analytic cosmology inputs
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# hmvec/_hmgrid\.py.*?)```", md, re.S).group(1)
    block = block.replace('C.CDLL("libhmgrid.so")', f'C.CDLL("{os.path.join(REPO, "hmvec_amd", "libhmgrid.so")}")')
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)

    zs = np.array([0.3, 1.2])
    ms = np.geomspace(1e11, 1e16, 48)
    ks = np.geomspace(1e-3, 20, 33)
    p = merged_params()
    cos = hm.Cosmology(p, accuracy="low", engine="analytic")
    ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
    ci = hmref.CosmoInputs(h=cos.h, omm0=cos.omm0, ombh2=p["ombh2"], rho_crit_0=float(cos.rho_critical_z(0.0)),
                           rho_crit_zs=cos.rho_critical_z(zs), Pzk=cos.P_lin_approx(ks, zs),
                           sPzk=cos.P_lin_approx(ksig, zs), ks_sigma2=ksig, h_of_z_zs=cos.h_of_z(zs))
    o = hmref.RefHaloModel(ci, zs, ks, ms, p)               # numpy state a reference HaloModel would hold
    to_dev = ns["to_dev"]
    fake = types.SimpleNamespace(zs=zs, ms=ms, ks=ks, p=p, rho_matter_z=cos.rho_matter_z,
                                 _d_uk={"nfw": to_dev(o.uk_profiles["nfw"])}, _d_nzm=to_dev(o.nzm), _d_bh=to_dev(o.bh),
                                 _d_ms=to_dev(ms), _d_wm=to_dev(trapz_weights(ms)), _d_ks=to_dev(ks), _d_Pzk=to_dev(o.Pzk))
    P1, P2 = ns["power_matter"](fake, "nfw", "nfw")
    ok, w = power_close(P1, o.get_power_1halo("nfw"))
    assert ok, w
    ok, w = power_close(P2, o.get_power_2halo("nfw"))
    assert ok, w
