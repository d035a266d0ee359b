"""Radial thermodynamic and mass profiles over a chain, all samples at once.

The reference walks the chain one sample at a time (joxsz_plots.py:249-273, 341-376, 451-478), each visit re-evaluating
closed-form profiles on ``r_pp``.  Here every function takes ``thetas[M, ndim]`` (thawed values, order of
``fit.thawed``) and returns ``[M, N]`` arrays; the closed forms are the ones the device prep kernel evaluates at the
X-ray shells, written out for an arbitrary radius vector.  Nothing here is on the log-posterior hot path.

Physical constants are mbproj2's ``physconstants`` (imported at joxsz_plots.py:5); mbproj2 is absent from the
reference tree, so the values are the published ones, unpinned.
"""
import numpy as np

from .chain import chain_subset, equal_tailed

kpc_cm = 3.0856776e21
Mpc_cm = 3.0856776e24
Mpc_km = 3.0856776e19
keV_erg = 1.6021765e-9
mu_g = 1.6605389e-24
G_cgs = 6.67428e-8
solar_mass_g = 1.9891e33
mu_e = 1.18
ne_nH = 1.21                         # electrons per hydrogen atom (solar abundances); [MEM] of mbproj2.physconstants, unpinned
yr_s = 31556926.0
MU_GAS = 0.61                        # joxsz_funcs.py:428 default


def par_table(pb, thetas):
    """``Fit.updateThawed`` for every sample: dict name -> [M, 1] column (frozen parameters broadcast)."""
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    if thetas.shape[1] != pb.ndim:
        raise ValueError('expected %d thawed values per sample' % pb.ndim)
    full = np.tile(np.asarray(pb.par_vals, np.float64), (thetas.shape[0], 1))
    full[:, pb.thawed_idx] = thetas
    return {n: full[:, k:k + 1] for k, n in enumerate(pb.par_names)}


def _radii(r_kpc):
    """[N] -> one row shared by all samples; [M, K] -> sample i evaluated at its own K radii."""
    r = np.asarray(r_kpc, np.float64)
    return r[None, :] if r.ndim == 1 else r


def density(pb, p, r_kpc):
    """joxsz_funcs.py:375-395 (modified-beta electron density, cm^-3)."""
    r = _radii(r_kpc)
    n0, rc, rs = 10 ** p['log(n_0)'], 10 ** p['log(r_c)'], 10 ** p['log(r_s)']
    al, be, ep, ga = p[r'\alpha'], p[r'\beta'], p[r'\epsilon'], p[r'\gamma']
    sq = n0 ** 2 * (r / rc) ** (-al) / ((1 + (r / rc) ** 2) ** (3 * be - al / 2) * (1 + (r / rs) ** ga) ** (ep / ga))
    if pb.ne_mode == 'double':
        sq = sq + (10 ** p['log(n_{02})']) ** 2 / (1 + (r / 10 ** p['log(r_{c2})']) ** 2) ** (3 * p[r'\beta_2'])
    return np.sqrt(sq)


def pressure(p, r_kpc):
    """joxsz_funcs.py:275-287 (gNFW electron pressure, keV cm^-3)."""
    x = _radii(r_kpc) / p['r_p']
    return p['P_0'] / (x ** p['c'] * (1 + x ** p['a']) ** ((p['b'] - p['c']) / p['a']))


def pressure_derivative(p, r_kpc):
    """joxsz_funcs.py:289-301 (dP/dr, keV cm^-3 kpc^-1)."""
    x = _radii(r_kpc) / p['r_p']
    a, b, c = p['a'], p['b'], p['c']
    return -p['P_0'] * (c + b * x ** a) / (p['r_p'] * x ** (c + 1) * (1 + x ** a) ** ((b - c + a) / a))


def hydrostatic_mass(pb, p, r_kpc, mu_gas=MU_GAS):
    """joxsz_funcs.py:428-437 (solar masses within r)."""
    r_cm = _radii(r_kpc) * kpc_cm
    dpr_cm = pressure_derivative(p, r_kpc) * keV_erg / kpc_cm
    return -dpr_cm * r_cm ** 2 / (mu_gas * mu_g * density(pb, p, r_kpc) * G_cgs) / solar_mass_g


def inner_fraction(edges):
    """joxsz_plots.py:194-206."""
    lo, hi = edges[:-1], edges[1:]
    vin = (lo + hi) ** 3 / 24 - lo ** 3 / 3
    vout = hi ** 3 / 3 - (lo + hi) ** 3 / 24
    return vin / (vin + vout)


def cumulative_gas_mass(r_kpc, dens):
    """joxsz_plots.py:208-217 for dens [M, N] (solar masses)."""
    r_kpc = np.asarray(r_kpc, np.float64)
    edg = np.append(r_kpc[0] / 2, r_kpc + r_kpc[0] / 2) * kpc_cm
    mgas = dens * mu_e * mu_g / solar_mass_g * 4 / 3 * np.pi * (edg[1:] ** 3 - edg[:-1] ** 3)
    inside = np.concatenate((np.zeros((mgas.shape[0], 1)), np.cumsum(mgas, axis=1)[:, :-1]), axis=1)
    return mgas * inner_fraction(edg) + inside


def bolometric_flux(lnT_grid, lnflux_Z0, lnflux_Z1, T_keV, Z_solar, ne_cm3):
    """``CountRate.getFlux(T, Z, ne)`` of mbproj2 as joxsz_plots.py:243 calls it: the flux at the observer (erg cm^-2 s^-1)
    per cm^3 of emitting plasma.  mbproj2 tabulates it with XSPEC the way it tabulates the count rates the reference's
    ``addCountCache`` documents (joxsz_funcs.py:667-680): ln(flux) for Z = 0 and Z = 1 solar at unit density on the ln T
    grid ``CountRate.Tlogvals``; linear in ln T (clamped to the grid ends), linear in Z, times n_e^2.  XSPEC is not
    available here, so the two tables are inputs, like the count-rate tables of the likelihood ([MEM], unpinned)."""
    lnT = np.log(np.asarray(T_keV, dtype=np.float64))
    f0 = np.exp(np.interp(lnT, lnT_grid, lnflux_Z0))
    f1 = np.exp(np.interp(lnT, lnT_grid, lnflux_Z1))
    return (f0 + (f1 - f0) * Z_solar) * np.asarray(ne_cm3) ** 2


def cooling_time(dens, temp, Z_solar, flux_table, D_L_Mpc):
    """joxsz_plots.py:242-244 (years): enthalpy per volume (5/2) n_e (1 + 1/ne_nH) k T over the bolometric emissivity, the
    latter recovered from the flux at the observer as flux * 4 pi D_L^2.  ``flux_table`` = (lnT_grid, lnflux_Z0, lnflux_Z1)."""
    flux = bolometric_flux(flux_table[0], flux_table[1], flux_table[2], temp, Z_solar, dens)
    return 2.5 * dens * (1. + 1. / ne_nH) * temp * keV_erg / (flux * 4. * np.pi * (D_L_Mpc * Mpc_cm) ** 2) / yr_s


def thermodynamic_profs(pb, thetas, r_kpc=None, flux_table=None, D_L_Mpc=None):
    """joxsz_plots.py:219-247: dict of [M, N] arrays ``dens, temp, press, entr, cmgas, tempx`` and, when the bolometric flux
    tables (``flux_table`` = (lnT_grid, lnflux_Z0, lnflux_Z1), see ``bolometric_flux``) and the luminosity distance are
    given, ``cool`` (the cooling time in years)."""
    r = pb.r_pp if r_kpc is None else np.asarray(r_kpc, np.float64)
    p = par_table(pb, thetas)
    dens, press = density(pb, p, r), pressure(p, r)
    temp = press / dens
    out = dict(dens=dens, temp=temp, press=press, entr=temp / dens ** (2 / 3), cmgas=cumulative_gas_mass(r, dens),
               tempx=temp * 10 ** p['log(T_X/T_{SZ})'])
    if flux_table is not None:
        if D_L_Mpc is None:
            raise ValueError('the cooling time needs the luminosity distance (cosmology.D_L, Mpc)')
        out['cool'] = cooling_time(dens, temp, p['Z'], flux_table, D_L_Mpc)
    return out


def critical_mass(r_kpc, z, H0, WM, WV, delta=500):
    """joxsz_plots.py:378-399: mass of a sphere of radius r at ``delta`` times the critical density at redshift z."""
    HZ = H0 / Mpc_km * np.sqrt(WM * (1. + z) ** 3 + WV)
    rho_c = 3. * HZ ** 2 / (8. * np.pi * G_cgs)
    return 4 / 3 * np.pi * rho_c * delta * (np.asarray(r_kpc, np.float64) * kpc_cm) ** 3 / solar_mass_g


def overdensity_radius(pb, thetas, cosmo, delta=500, start_opt=700., tol=1.48e-8, maxiter=50):
    """joxsz_plots.py:335-337: the root of M_hydro(r) - M_crit(r) from ``start_opt`` kpc for every sample, by the secant
    iteration ``scipy.optimize.newton`` runs without a derivative (same starting pair, same stopping rule), all samples
    advanced together.  ``cosmo`` = dict(z, H0, WM, WV) (joxsz_main.py:28-31).  Returns (r_delta[M], m_delta[M]); NaN
    where the iteration does not converge."""
    p = par_table(pb, thetas)
    M = next(iter(p.values())).shape[0]

    def f(r):
        with np.errstate(all='ignore'):
            return hydrostatic_mass(pb, p, r[:, None])[:, 0] - critical_mass(r, cosmo['z'], cosmo['H0'], cosmo['WM'], cosmo['WV'], delta)

    x0 = np.full(M, float(start_opt))
    eps = 1e-4
    x1 = x0 * (1 + eps) + np.where(x0 >= 0, eps, -eps)
    q0, q1 = f(x0), f(x1)
    sw = np.abs(q1) < np.abs(q0)                                  # newton's initial ordering of the starting pair
    x0, x1, q0, q1 = np.where(sw, x1, x0), np.where(sw, x0, x1), np.where(sw, q1, q0), np.where(sw, q0, q1)
    done = np.zeros(M, bool)
    root = np.full(M, np.nan)
    for _ in range(maxiter):
        with np.errstate(all='ignore'):
            xn = np.where(q1 == q0, (x1 + x0) / 2, x1 - q1 * (x1 - x0) / (q1 - q0))
        conv = ~done & (np.abs(xn - x1) < tol)
        root[conv] = xn[conv]
        done |= conv | ~np.isfinite(xn)
        if done.all():
            break
        x0, q0 = x1, q1
        x1 = np.where(done, x1, xn)
        q1 = f(x1)
    with np.errstate(all='ignore'):
        return root, hydrostatic_mass(pb, p, root[:, None])[:, 0]


# ---- chain summaries (equal-tailed intervals), joxsz_plots.py:249-273, 341-376, 451-478 ----
def comp_rad_profs(cube, pb, num='all', seed=None, ci=95, flux_table=None, D_L_Mpc=None):
    """Returns dict name -> [3, N] (lower, median, upper) for dens, temp, press, entr, cmgas, tempx (and cool, given the
    flux tables)."""
    prof = thermodynamic_profs(pb, chain_subset(cube, num, seed), flux_table=flux_table, D_L_Mpc=D_L_Mpc)
    return {k: equal_tailed(v, ci) for k, v in prof.items()}


def comp_mass_prof(cube, pb, cosmo=None, num='all', seed=None, overdens=True, delta=500, start_opt=700., ci=95):
    thetas = chain_subset(cube, num, seed)
    mass = equal_tailed(hydrostatic_mass(pb, par_table(pb, thetas), pb.r_pp), ci)
    if not overdens:
        return mass
    r_d, m_d = overdensity_radius(pb, thetas, cosmo, delta, start_opt)
    return mass, equal_tailed(r_d, ci), equal_tailed(m_d, ci)


def frac_gas_prof(cube, pb, num='all', seed=None, ci=95):
    thetas = chain_subset(cube, num, seed)
    p = par_table(pb, thetas)
    return equal_tailed(cumulative_gas_mass(pb.r_pp, density(pb, p, pb.r_pp)) / hydrostatic_mass(pb, p, pb.r_pp), ci)
