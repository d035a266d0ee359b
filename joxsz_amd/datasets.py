"""Problem builders: CL J1226.9+3332-shaped synthetic inputs and the loader
for the reference's bundled data directory.

``synthetic_problem`` follows SURVEY.md section 8(d): same step, plate scale,
beam width, transfer-function shape, flux radii/errors, annuli and bands as the
bundled cluster, at an arbitrary map side S and radial grid length N.
``bundled_problem`` performs the wiring of joxsz_main.py:95-125 on the
reference's data files (only usable where those files exist).
"""
import math
import os
import numpy as np

from .problem import Problem, default_par_table
from . import setup_host as sh

# data/SZ/Compton_to_Jy_per_beam.dat (6-point table; joxsz_main.py:108-109 multiplies by 1e3)
CONV_T_KEV = np.array([1., 5., 10., 15., 20., 25.])
CONV_JY_PER_BEAM = np.array([-11.63, -11.34, -11.00, -10.71, -10.38, -10.17])

# data/SZ/press_data_cl1226_flagsource_Xraycent.dat, third column (statistical errors, mJy/beam)
FLUX_ERR = np.array([0.1768, 0.106, 0.08111, 0.07474, 0.06696, 0.06008, 0.05522, 0.05077, 0.04688,
                     0.04299, 0.03916, 0.03745, 0.03583, 0.03467, 0.03345, 0.03239, 0.03139, 0.0304,
                     0.02953])

# joxsz_main.py:73-74
BAND_EDGES_EV = [[700, 1000], [1000, 1300], [1300, 1600], [1600, 2000], [2000, 2700],
                 [2700, 3400], [3400, 3800], [3800, 4300], [4300, 5000], [5000, 7000]]

KPC_AS_CLJ1226 = 8.0012          # flat LCDM, z=0.888, H0=67.32, Om=0.3158 (joxsz_main.py:27-31)

# fiducial thawed vector (order of Problem.thawed): a physically sensible cluster,
# central T ~ 9 keV, monotone hydrostatic mass
THETA0 = np.array([-1.9, 0.6, 2.0, 2.9, 3.0, 0.0, 0.3, 0.12, 1.33, 4.13, 400., 1.0, 1.0])


# log(n_02), beta_2, log(r_c2) of the 'double' density mode (joxsz_funcs.py:367-372): a faint compact core
THETA0_DOUBLE_EXTRA = np.array([-3.0, 0.6, 1.2])


def synthetic_count_rate_tables(nband=10, ntab=100):
    """Smooth bremsstrahlung-like ln(rate) tables on mbproj2's ln T grid
    (0.06-60 keV, 100 points; layout of joxsz_funcs.py:667-680).  XSPEC is not
    available, so the real tables cannot be regenerated."""
    lnT = np.linspace(math.log(0.06), math.log(60.), ntab)
    T = np.exp(lnT)
    out = np.zeros((nband, 2, ntab))
    for b in range(nband):
        lo, hi = BAND_EDGES_EV[b % len(BAND_EDGES_EV)]
        e_mid = 0.5e-3 * (lo + hi) * (1 + 0.888)          # rest-frame keV
        width = (hi - lo) / 300.
        base = -157.5 + math.log(width) - 0.35 * b - 0.5 * lnT - e_mid / T
        out[b, 0] = base
        out[b, 1] = base + 0.45 / (1. + T / 2.5)          # metal lines matter at low T
    return lnT, np.maximum(out, math.log(1e-300))


def synthetic_problem(S=512, N=500, seed=0, sz_only=False, ne_mode='single', step=2.0,
                      kpc_as=KPC_AS_CLJ1226, fwhm=18.5, nann=15, nband=10, tf_scale=0.02, tf_c=0.95):
    """CL J1226.9+3332-shaped constants at map side ``S`` and grid length ``N``.

    The SZ flux values and X-ray counts are smooth placeholders of the right
    magnitude; ``fill_data`` replaces them with model-generated values at a
    fiducial parameter vector (plus noise) once a model evaluator is at hand.
    """
    rng = np.random.default_rng(seed)
    h = step * kpc_as
    r_pp = h * np.arange(1, N + 1, dtype=np.float64)
    c = S // 2
    radius = step * (np.arange(S, dtype=np.float64) - c)
    d_mat = sh.pixel_radius_matrix(radius * kpc_as)
    flux_r = 3.136 + 6.273 * np.arange(19)
    beam_2d, _ = sh.beam_image(step, flux_r[-1], approx=True, fwhm=fwhm)
    wn = np.linspace(0., 0.4967, 76)
    wn, tf = sh.transfer_function(wn, None, approx=True, loc=0., scale=tf_scale, c=tf_c)
    filtering = sh.filter_image(wn, tf, S, step)
    flux = -2.6 * np.exp(-flux_r / 32.) + 0.02
    flux_data = np.vstack((flux_r, flux + FLUX_ERR * rng.standard_normal(19), FLUX_ERR))

    edges_am = 0.05 * np.arange(nann + 1)
    geo = sh.annuli_geometry(edges_am, kpc_as)
    lnT, lnrate = synthetic_count_rate_tables(nband)
    exposures = 2.5e4 * (1. + 0.06 * rng.standard_normal((nband, nann)))
    areascales = 1.0 + 0.08 * rng.random((nband, nann))
    backrates = 1.3e-4 * (1. + 0.2 * rng.random((nband, nann)))
    mid = geo['midpt_kpc']
    cts = np.floor(40. / (1. + (mid / 120.) ** 2)[None, :] * np.exp(-0.3 * np.arange(nband))[:, None]
                   * (geo['geomarea'] / geo['geomarea'][0])[None, :] ** 0.5 + 1.)

    tab = default_par_table(ne_mode, logr_max=float(geo['edges_logkpc'][-2]))
    pb = Problem(step=step, kpc_as=kpc_as, conv_T=CONV_T_KEV.copy(), conv_v=1e3 * CONV_JY_PER_BEAM,
                 flux_data=flux_data, beam_2d=beam_2d, radius=radius, r_pp=r_pp, d_mat=d_mat,
                 filtering=filtering, x_r_ne_kpc=mid.copy(), x_r_T_kpc=mid.copy(),
                 projvols=geo['projvols'], cts=cts, areascales=areascales, exposures=exposures,
                 backrates=backrates, geomarea=geo['geomarea'], lnT=lnT, lnrate=lnrate,
                 ne_mode=ne_mode, sz_only=sz_only, **tab)
    pb.meta = dict(kind='synthetic', seed=seed, fwhm=fwhm, tf_scale=tf_scale, tf_c=tf_c)
    return pb.validate()


def fiducial_theta(pb):
    """THETA0 extended to the problem's thawed vector (double-beta mode adds three)."""
    if pb.ndim == THETA0.size:
        return THETA0.copy()
    return np.concatenate((THETA0, THETA0_DOUBLE_EXTRA))


def fill_data(pb, bright, xprofs, seed=0):
    """Replace the placeholder observations by model-generated ones.

    ``bright`` = model surface-brightness profile at the fiducial vector
    (``get_sz_like(output='bright')``, joxsz_funcs.py:474) on ``radius[S//2:]``;
    ``xprofs`` = predicted counts [nband, nann] (``calcProfiles``, joxsz_funcs.py:527).
    SZ flux <- cubic interpolation of ``bright`` at the data radii + N(0, err);
    counts <- Poisson(xprofs)."""
    from scipy.interpolate import CubicSpline
    rng = np.random.default_rng(seed + 1)
    S = pb.S
    g = CubicSpline(pb.radius[S // 2:], np.asarray(bright, dtype=np.float64), bc_type='not-a-knot')
    pb.flux_data = pb.flux_data.copy()
    pb.flux_data[1] = g(pb.flux_data[0]) + pb.flux_data[2] * rng.standard_normal(pb.flux_data.shape[1])
    if xprofs is not None:
        pb.cts = rng.poisson(np.clip(xprofs, 0., 1e9)).astype(np.float64)
    return pb


def walker_ball(pb, nwalkers, spread=0.1, seed=0, theta0=None):
    """Initial walker positions as ``_generateInitPars`` draws them
    (joxsz_funcs.py:567, spread joxsz_main.py:209): theta0*(1+spread*N(0,1)).
    No rejection here (the caller filters on finite log-probability)."""
    rng = np.random.default_rng(seed + 2)
    t0 = fiducial_theta(pb) if theta0 is None else np.asarray(theta0, dtype=np.float64)
    return t0[None, :] * (1. + spread * rng.standard_normal((nwalkers, t0.size)))


def bundled_problem(data_dir, R_b=5000., step=2.0, kpc_as=KPC_AS_CLJ1226, beam_approx=False,
                    tf_approx=False, lnrate_tables=None):
    """joxsz_main.py:95-125 on the reference's ``data/`` directory.

    The XSPEC count-rate tables cannot be rebuilt here, so ``lnrate_tables`` =
    (lnT, lnrate[nband, 2, ntab]) must be supplied or the synthetic ones are used.
    """
    szd = os.path.join(data_dir, 'SZ')
    xd = os.path.join(data_dir, 'X')
    flux_data = np.array(sh.read_columns(os.path.join(szd, 'press_data_cl1226_flagsource_Xraycent.dat'), 3))
    maxr = flux_data[0][-1]
    prof = sh.clip_beam_profile(*sh.read_columns(os.path.join(szd, 'Beam150GHz.fits'), 2))
    beam_2d, fwhm = sh.beam_image(step, maxr, approx=beam_approx, profile=prof, fwhm=18.5 if beam_approx else None)
    radius, sep, r_pp = sh.sz_axes(step, kpc_as, maxr, fwhm, R_b)
    d_mat = sh.pixel_radius_matrix(radius * kpc_as)
    wn, tf = sh.read_columns(os.path.join(szd, 'TransferFunction150GHz_CLJ1227.fits'), 2)
    wn, tf = sh.transfer_function(wn, tf, approx=tf_approx)
    filtering = sh.filter_image(wn, tf, d_mat.shape[0], step)
    t_keV, cjy = np.loadtxt(os.path.join(szd, 'Compton_to_Jy_per_beam.dat'), skiprows=1, unpack=True)

    fgs = [np.loadtxt(os.path.join(xd, 'fg_profnew_%04i_%04i.dat' % tuple(b))) for b in BAND_EDGES_EV]
    bgs = [np.loadtxt(os.path.join(xd, 'bg_profnew_%04i_%04i.dat' % tuple(b))) for b in BAND_EDGES_EV]
    geo = sh.annuli_geometry(sh.annuli_edges(fgs[0]), kpc_as)
    bands = [sh.band_from_profiles(f, b) for f, b in zip(fgs, bgs)]
    if lnrate_tables is None:
        lnrate_tables = synthetic_count_rate_tables(len(bands))
    lnT, lnrate = lnrate_tables
    tab = default_par_table('single', logr_max=float(geo['edges_logkpc'][-2]))
    pb = Problem(step=step, kpc_as=kpc_as, conv_T=t_keV, conv_v=1e3 * cjy, flux_data=flux_data,
                 beam_2d=beam_2d, radius=radius, r_pp=r_pp, d_mat=d_mat, filtering=filtering,
                 x_r_ne_kpc=geo['midpt_kpc'].copy(), x_r_T_kpc=geo['midpt_kpc'].copy(),
                 projvols=geo['projvols'], cts=np.array([b['cts'] for b in bands]),
                 areascales=np.array([b['areascales'] for b in bands]),
                 exposures=np.array([b['exposures'] for b in bands]),
                 backrates=np.array([b['backrates'] for b in bands]), geomarea=geo['geomarea'],
                 lnT=lnT, lnrate=lnrate, **tab)
    pb.meta = dict(kind='bundled', fwhm=float(fwhm), sep=int(sep))
    return pb.validate()
