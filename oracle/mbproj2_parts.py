"""Restatement of the mbproj2 pieces on the JoXSZ log-posterior path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  **Parity unpinned**:
mbproj2 is a third-party dependency named, unpinned, in
``/root/reference/requirements.txt:1``; its source is neither under
``/root/reference`` nor installed here.  Each function below names the mbproj2
entry point whose published behaviour it restates and the reference call site
that reaches it.
"""
import math
import numpy as np

# mbproj2.physconstants (imported at joxsz_funcs.py:6)
kpc_cm = 3.0856776e21
keV_erg = 1.6021765e-9
mu_g = 1.6605389e-24
G_cgs = 6.67428e-8
solar_mass_g = 1.9891e33


def param_prior(val, minval, maxval):
    """``mbproj2.Param.prior`` (summed at joxsz_funcs.py:518): flat box."""
    if val < minval or val > maxval:
        return -np.inf
    return 0.0


def param_gaussian_prior(val, mu, sigma):
    """``mbproj2.ParamGaussian.prior`` (joxsz_funcs.py:518, parameters
    ``backscale`` / ``calibration`` of joxsz_main.py:156-157): normalised
    Gaussian log-density; -inf for a non-positive sigma."""
    if sigma <= 0:
        return -np.inf
    return (-0.5 * math.log(2 * math.pi) - math.log(sigma)
            - 0.5 * ((val - mu) / sigma) ** 2)


def projection_volume_matrix(edges):
    """``mbproj2.utils.projectionVolumeMatrix`` (built in ``Annuli``,
    joxsz_main.py:116).  ``V[i, j]`` = volume (front and back) of the spherical
    shell ``edges[j]..edges[j+1]`` seen through the annulus
    ``edges[i]..edges[i+1]`` on the sky, same length unit cubed as ``edges``."""
    e = np.asarray(edges, dtype=np.float64)
    n = e.size - 1
    V = np.zeros((n, n))

    def tsq(x):
        return np.sqrt(np.clip(x, 0.0, 1e200))

    for i in range(n):          # annulus on the sky
        y1, y2 = e[i], e[i + 1]
        for j in range(n):      # shell
            R1, R2 = e[j], e[j + 1]
            p1 = tsq(R1 ** 2 - y2 ** 2)
            p2 = tsq(R1 ** 2 - y1 ** 2)
            p3 = tsq(R2 ** 2 - y2 ** 2)
            p4 = tsq(R2 ** 2 - y1 ** 2)
            V[i, j] = 2.0 * (2.0 / 3.0) * math.pi * ((p1 ** 3 - p2 ** 3) + (p4 ** 3 - p3 ** 3))
    return V


def count_rate(lnT_grid, lnrate_Z0, lnrate_Z1, T_keV, Z_solar, ne_cm3):
    """``mbproj2.countrate.CountRate.getCountRate`` (via ``Band.calcProjProfile``,
    reached from ``Fit.calcProfiles`` at joxsz_funcs.py:527).  The cached table
    layout is the one the reference's ``addCountCache`` documents at
    joxsz_funcs.py:667-680: ln(count rate) for Z=0 and Z=1 solar on the ln T
    grid ``CountRate.Tlogvals``, unit density.  Linear interpolation in ln T
    (clamped to the grid ends), linear in Z, times n_e^2."""
    lnT = np.log(np.asarray(T_keV, dtype=np.float64))
    z0 = np.exp(np.interp(lnT, lnT_grid, lnrate_Z0))
    z1 = np.exp(np.interp(lnT, lnT_grid, lnrate_Z1))
    return (z0 + (z1 - z0) * Z_solar) * np.asarray(ne_cm3) ** 2


def band_proj_profile(projvols, rates, areascales, exposures, backrates, geomarea, backscale):
    """``mbproj2.data.Band.calcProjProfile``: project the shell emissivities on
    the sky, convert to counts, add the scaled background (band constants from
    ``loadBand``, joxsz_funcs.py:184-211)."""
    proj = projvols.dot(rates)
    proj = proj * areascales * exposures
    proj = proj + backrates * geomarea * areascales * exposures * backscale
    return proj


def cash_log_likelihood(data, model):
    """``mbproj2.utils.cashLogLikelihood`` (joxsz_funcs.py:504)."""
    like = np.sum(data * np.log(model)) - np.sum(model)
    if np.isfinite(like):
        return float(like)
    return -np.inf


def kpc_per_arcsec(z, H0=67.32, WM=0.3158, WV=0.6842, n=20000):
    """``mbproj2.Cosmology.kpc_per_arcsec`` for the cosmology of
    joxsz_main.py:27-31 (angular-diameter distance by numerical integration of
    1/E(z); curvature term kept for generality)."""
    c = 299792.458
    WK = 1.0 - WM - WV
    zz = np.linspace(0.0, z, n + 1)
    Ez = np.sqrt(WM * (1 + zz) ** 3 + WK * (1 + zz) ** 2 + WV)
    f = 1.0 / Ez
    dz = z / n
    integ = dz * (f[0] + f[-1] + 4 * f[1:-1:2].sum() + 2 * f[2:-1:2].sum()) / 3.0
    DCMR = integ                          # in units of c/H0
    if abs(WK) < 1e-12:
        DM = DCMR
    elif WK > 0:
        DM = math.sinh(math.sqrt(WK) * DCMR) / math.sqrt(WK)
    else:
        DM = math.sin(math.sqrt(-WK) * DCMR) / math.sqrt(-WK)
    DA_Mpc = c / H0 * DM / (1 + z)
    return DA_Mpc * 1000.0 * math.pi / (180.0 * 3600.0)
