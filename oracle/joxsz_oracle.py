"""CPU restatement of the JoXSZ per-walker log-posterior on plain arrays.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``): this module is the parity
checker and the timed CPU baseline; the product (``joxsz_amd``) never imports
it.

Each function cites the reference lines it follows (``funcs:`` =
``/root/reference/joxsz_funcs.py``, ``main:`` = ``/root/reference/joxsz_main.py``).
It uses the same scipy primitives as the reference (``interp1d``,
``fftconvolve``, ``fftpack.fft2/ifft2``) so that the reference's own outputs,
committed under ``tests/golden/`` by ``oracle/make_golden.py``, pin it.

The "problem" argument ``pb`` is any object exposing the attributes listed in
``PROBLEM_FIELDS`` (``joxsz_amd.problem.Problem`` is one; the oracle does not
import it).
"""
import numpy as np
from scipy.interpolate import interp1d
from scipy.signal import fftconvolve
from scipy.fftpack import fft2, ifft2
from scipy.stats import norm
from scipy.integrate import simpson
from scipy import optimize

from . import pyabel_direct
from . import mbproj2_parts as mbp

PROBLEM_FIELDS = (
    # SZ constants (funcs:136-170, main:95-111)
    'm_e', 'sigma_T', 'kpc_cm', 'step', 'kpc_as', 'conv_T', 'conv_v',
    'flux_data', 'beam_2d', 'radius', 'r_pp', 'd_mat', 'filtering',
    # parameter table (funcs:241-246,266-272,318,358-372; main:156-175)
    'par_names', 'par_vals', 'par_min', 'par_max', 'par_frozen', 'par_kind',
    'par_mu', 'par_sigma', 'ne_mode',
    # X-ray constants (main:116-125, funcs:184-211)
    'x_r_ne_kpc', 'x_r_T_kpc', 'projvols', 'cts', 'areascales', 'exposures',
    'backrates', 'geomarea', 'lnT', 'lnrate',
    # switches
    'exclude_unphy_mass', 'sz_only',
)


# --------------------------------------------------------------------------
# L1 setup (funcs:46-134) -- restated so the golden setup tensors can pin it
# --------------------------------------------------------------------------

def centdistmat(r, offset=0.0):
    """funcs:78-88."""
    x, y = np.meshgrid(r, r)
    return np.sqrt(x ** 2 + y ** 2) + offset


def dist(naxis):
    """funcs:104-116."""
    axis = np.linspace(-naxis // 2 + 1, naxis // 2, naxis)
    result = np.sqrt(axis ** 2 + axis[:, np.newaxis] ** 2)
    return np.roll(result, naxis // 2 + 1, axis=(0, 1))


def filt_image(wn_as, tf, side, step):
    """funcs:118-134."""
    f = interp1d(wn_as, tf, 'cubic', bounds_error=False, fill_value=tuple([tf[0], tf[-1]]))
    kmax = 1 / step
    karr = dist(side) / side
    karr /= karr.max()
    karr *= kmax
    return f(karr)


def mybeam(step, maxr_data, approx=False, beam_profile=None, normalize=True, fwhm_beam=None):
    """funcs:46-76; ``beam_profile`` = (radius, beam) as ``read_beam`` returns
    them (funcs:30-44) when ``approx`` is False."""
    if not approx:
        r_irreg, b = beam_profile
        f = interp1d(np.append(-r_irreg, r_irreg), np.append(b, b), 'cubic',
                     bounds_error=False, fill_value=(0., 0.))
        inv_f = lambda x: f(x) - f(0.) / 2
        fwhm_beam = 2 * optimize.newton(inv_f, x0=5.)
    maxr = (maxr_data + 3 * fwhm_beam) // step * step
    rad = np.arange(0., maxr + step, step)
    rad = np.append(-rad[:0:-1], rad)
    rad_cut = rad[np.where(abs(rad) <= 3 * fwhm_beam)]
    beam_mat = centdistmat(rad_cut)
    if approx:
        sigma_beam = fwhm_beam / (2 * np.sqrt(2 * np.log(2)))
        beam_2d = norm.pdf(beam_mat, loc=0., scale=sigma_beam)
    else:
        beam_2d = f(beam_mat)
    if normalize:
        beam_2d /= beam_2d.sum() * step ** 2
    return beam_2d, fwhm_beam


def read_beam_clip(radius, beam_prof):
    """funcs:30-44 applied to already-read columns."""
    radius = np.asarray(radius, dtype=np.float64)
    beam_prof = np.asarray(beam_prof, dtype=np.float64)
    if np.isnan(beam_prof).sum() > 0.:
        first_nan = np.where(np.isnan(beam_prof))[0][0]
        radius = radius[:first_nan]
        beam_prof = beam_prof[:first_nan]
    if beam_prof.min() < 0.:
        first_neg = np.where(beam_prof < 0.)[0][0]
        radius = radius[:first_neg]
        beam_prof = beam_prof[:first_neg]
    return radius, beam_prof


# --------------------------------------------------------------------------
# L2 profile components
# --------------------------------------------------------------------------

def pars_dict(pb, theta=None):
    """funcs:515-516 / mbproj2 ``Fit.updateThawed``: scatter the thawed values
    (in the order of ``fit.thawed``, main:179) over the full parameter table."""
    vals = np.array(pb.par_vals, dtype=np.float64).copy()
    if theta is not None:
        thawed = [k for k in range(len(pb.par_names)) if not pb.par_frozen[k]]
        theta = np.asarray(theta, dtype=np.float64)
        assert theta.size == len(thawed)
        vals[thawed] = theta
    return {n: float(v) for n, v in zip(pb.par_names, vals)}


def press_fun(p, r):
    """funcs:275-287."""
    P_0, r_p, a, b, c = p['P_0'], p['r_p'], p['a'], p['b'], p['c']
    return P_0 / ((r / r_p) ** c * (1 + (r / r_p) ** a) ** ((b - c) / a))


def press_derivative(p, r):
    """funcs:289-301."""
    P_0, r_p, a, b, c = p['P_0'], p['r_p'], p['a'], p['b'], p['c']
    return -P_0 * (c + b * (r / r_p) ** a) / (r_p * (r / r_p) ** (c + 1) * (1 + (r / r_p) ** a) ** ((b - c + a) / a))


def vikh_function(p, r, mode='single'):
    """funcs:375-395."""
    n_0 = 10 ** p['log(n_0)']
    beta = p[r'\beta']
    r_c = 10 ** p['log(r_c)']
    r_s = 10 ** p['log(r_s)']
    alpha = p[r'\alpha']
    epsilon = p[r'\epsilon']
    gamma = p[r'\gamma']
    res_sq = n_0 ** 2 * (r / r_c) ** (-alpha) / ((1 + (r / r_c) ** 2) ** (3 * beta - alpha / 2) * (1 + (r / r_s) ** gamma) ** (epsilon / gamma))
    if mode == 'double':
        n_02 = 10 ** p['log(n_{02})']
        r_c2 = 10 ** p['log(r_{c2})']
        beta_2 = p[r'\beta_2']
        res_sq = res_sq + n_02 ** 2 / (1 + (r / r_c2) ** 2) ** (3 * beta_2)
    return np.sqrt(res_sq)


def dens_prior(p):
    """funcs:397-407."""
    if 10 ** p['log(r_c)'] > 10 ** p['log(r_s)']:
        return -np.inf
    return 0.0


def temp_fun(p, r, mode='single', getT_SZ=False):
    """funcs:321-336."""
    T_SZ = press_fun(p, r) / vikh_function(p, r, mode)
    if getT_SZ:
        return T_SZ
    return T_SZ * 10 ** p['log(T_X/T_{SZ})']


def mass_fun(p, r, mode='single', mu_gas=0.61):
    """funcs:428-437."""
    dpr_kpc = press_derivative(p, r)
    dpr_cm = dpr_kpc * mbp.keV_erg / mbp.kpc_cm
    ne = vikh_function(p, r, mode)
    r_cm = r * mbp.kpc_cm
    return -dpr_cm * r_cm ** 2 / (mu_gas * mbp.mu_g * ne * mbp.G_cgs) / mbp.solar_mass_g


# --------------------------------------------------------------------------
# L3 SZ likelihood (funcs:439-493)
# --------------------------------------------------------------------------

def nrow_of(S):
    """Number of profile points: ``map_out[S//2, S//2:]`` (funcs:472).  For the
    reference's odd S this is sep+1; even S is this build's generalisation
    (SURVEY.md section 8d)."""
    return S - S // 2


def row_chain(pb, pp, abel=None):
    """funcs:457-467 and the row of funcs:472 for a GIVEN pressure profile ``pp`` on ``r_pp``: Abel integral, Compton y,
    mirrored cubic spline onto the pixel radii, beam convolution, transfer function, central row.  Every step is linear
    in ``pp`` with constant coefficients (which ``sz_operator`` tabulates)."""
    if abel is None:
        abel = pyabel_direct.direct_transform_forward
    r_pp = pb.r_pp
    S = pb.d_mat.shape[0]
    out = {}
    ab = abel(pp, r_pp)                              # funcs:457
    out['ab'] = ab
    y = pb.kpc_cm * pb.sigma_T / pb.m_e * ab         # funcs:459
    out['y'] = y
    f = interp1d(np.append(-r_pp, r_pp), np.append(y, y), 'cubic',
                 bounds_error=False, fill_value=(0., 0.))   # funcs:460
    y_2d = f(pb.d_mat)                               # funcs:462
    out['y_2d'] = y_2d
    conv_2d = fftconvolve(y_2d, pb.beam_2d, 'same') * pb.step ** 2   # funcs:464
    out['conv_2d'] = conv_2d
    FT_map_in = fft2(conv_2d)                        # funcs:466
    map_out = np.real(ifft2(FT_map_in * pb.filtering))   # funcs:467
    out['map_row'] = map_out[S // 2, S // 2:].copy()
    return out


def sz_operator(pb, abel=None):
    """G [N, nrow]: row j = ``row_chain`` of the unit profile e_j, so that map_row = pp @ G for every pp (the matrix the
    HIP library's operator route builds with its own kernels, ``jx_set_route``)."""
    N = pb.r_pp.size
    eye = np.eye(N)
    return np.array([row_chain(pb, eye[j], abel)['map_row'] for j in range(N)])


def sz_stages(pb, p, abel=None):
    """All intermediate quantities of ``get_sz_like`` (funcs:453-479) in a dict."""
    r_pp = pb.r_pp
    S = pb.d_mat.shape[0]
    nrow = nrow_of(S)
    nt = nrow - 1                                    # == sep for odd S
    pp = press_fun(p, r_pp)                          # funcs:453
    out = {'pp': pp}
    out.update(row_chain(pb, pp, abel))
    map_row = out['map_row']
    t_prof = temp_fun(p, r_pp[:nt], pb.ne_mode, getT_SZ=True)   # funcs:469
    out['t_prof'] = t_prof
    h = interp1d(np.append(-r_pp[:nt], r_pp[:nt]), np.append(t_prof, t_prof), 'cubic',
                 bounds_error=False, fill_value=(t_prof[-1], t_prof[-1]))   # funcs:470-471
    t0 = h(0.)
    out['t0'] = float(t0)
    convert = interp1d(pb.conv_T, pb.conv_v, 'linear', fill_value='extrapolate')   # main:109
    map_prof = map_row * convert(np.append(t0, t_prof)) * p['calibration']   # funcs:472-473
    out['bright'] = map_prof
    g = interp1d(pb.radius[S // 2:], map_prof, 'cubic', fill_value='extrapolate')   # funcs:476
    model = g(pb.flux_data[0])
    out['flux_model'] = model
    chisq = np.nansum(((pb.flux_data[1] - model) / pb.flux_data[2]) ** 2)   # funcs:478
    out['chisq'] = float(chisq)
    log_lik = -chisq / 2                             # funcs:479
    if getattr(pb, 'calc_integ', False):             # funcs:480-484 (`simps` is today's scipy.integrate.simpson)
        y = out['y']
        f = interp1d(np.append(-r_pp, r_pp), np.append(y, y), 'cubic', bounds_error=False, fill_value=(0., 0.))
        xs = np.arange(0., r_pp[-1] / pb.kpc_as / 60 + pb.step / 60, pb.step / 60)
        cint = simpson(np.concatenate((f(0.), y), axis=None) * xs, xs) * 2 * np.pi
        new_chi = np.nansum(((cint - pb.integ_mu) / pb.integ_sig) ** 2)
        log_lik -= new_chi / 2
        out['integ'] = float(cint)
    out['ll'] = float(log_lik)
    return out


def get_sz_like(pb, p, output='ll', abel=None):
    """funcs:439-493."""
    st = sz_stages(pb, p, abel=abel)
    if output in ('pp', 'bright', 'chisq', 'll') or (output == 'integ' and 'integ' in st):
        return st[output]
    raise RuntimeError('Unrecognised output name')


# --------------------------------------------------------------------------
# L3 X-ray likelihood (funcs:495-505, 527-532; mbproj2 Fit.calcProfiles)
# --------------------------------------------------------------------------

def calc_profiles(pb, p):
    """mbproj2 ``Fit.calcProfiles`` (funcs:527): predicted counts [nband, nann]."""
    ne = vikh_function(p, pb.x_r_ne_kpc, pb.ne_mode)
    T = temp_fun(p, pb.x_r_T_kpc, pb.ne_mode)        # funcs:338-339 (T_X)
    Z = p['Z']
    nband = pb.cts.shape[0]
    profs = np.zeros_like(pb.cts, dtype=np.float64)
    for b in range(nband):
        rates = mbp.count_rate(pb.lnT, pb.lnrate[b, 0], pb.lnrate[b, 1], T, Z, ne)
        profs[b] = mbp.band_proj_profile(pb.projvols, rates, pb.areascales[b], pb.exposures[b],
                                         pb.backrates[b], pb.geomarea, p['backscale'])
    return profs


def like_from_profs(pb, profs):
    """funcs:495-505."""
    like = 0.0
    for b in range(pb.cts.shape[0]):
        ok = ~np.isnan(pb.cts[b])
        like += mbp.cash_log_likelihood(pb.cts[b][ok], profs[b][ok])
    return like


# --------------------------------------------------------------------------
# L3 joint log-posterior (funcs:507-546)
# --------------------------------------------------------------------------

def par_prior(pb, p):
    """funcs:518: sum of ``prior()`` over ALL parameters, frozen ones included."""
    tot = 0.0
    for k, name in enumerate(pb.par_names):
        if pb.par_kind[k] == 1:
            tot += mbp.param_gaussian_prior(p[name], pb.par_mu[k], pb.par_sigma[k])
        else:
            tot += mbp.param_prior(p[name], pb.par_min[k], pb.par_max[k])
    return tot


def get_likelihood(pb, theta=None, abel=None, parts=False):
    """funcs:507-546.  Returns a Python float (``-inf`` on rejection).  With
    ``parts=True`` returns (total, xray_like, sz_like, prior)."""
    p = pars_dict(pb, theta)
    parprior = par_prior(pb, p)                      # funcs:518
    if not np.isfinite(parprior):                    # funcs:519-520
        return (-np.inf, None, None, None) if parts else -np.inf
    if pb.exclude_unphy_mass:                        # funcs:522-525
        m_prof = mass_fun(p, pb.r_pp, pb.ne_mode)
        if not (all(np.gradient(m_prof, 1) > 0.)):
            return (-np.inf, None, None, None) if parts else -np.inf
    if pb.sz_only:                                   # build extension (BASELINE configs[1])
        like = 0.0
    else:
        profs = calc_profiles(pb, p)                 # funcs:527
        if np.array(profs).min() > 0.:               # funcs:529-532
            like = like_from_profs(pb, profs)
        else:
            like = -np.inf
    sz_like = get_sz_like(pb, p, 'll', abel=abel)    # funcs:534
    prior = dens_prior(p) + parprior                 # funcs:536 (T and Z components contribute 0)
    totlike = float(like + prior + sz_like)          # funcs:538
    if parts:
        return totlike, like, sz_like, prior
    return totlike


def log_posterior_batch(pb, thetas):
    """Serial map of ``get_likelihood`` over the rows of ``thetas`` -- what
    emcee does with the reference callable (main:206)."""
    thetas = np.atleast_2d(thetas)
    return np.array([get_likelihood(pb, t) for t in thetas], dtype=np.float64)
