"""Host-side container of the constant tensors the log-posterior consumes.

Mirror of the reference's ``SZ_data`` (joxsz_funcs.py:136-170) plus the pieces
of ``mbproj2.Data`` / ``Annuli`` / ``Band`` / ``Fit.pars`` that
``getLikelihood`` (joxsz_funcs.py:507-546) reads, flattened to plain numpy
arrays so they can be handed to the C-ABI (``include/joxsz_hip.h``) unchanged.
"""
from dataclasses import dataclass, field, fields
import numpy as np

# mbproj2.physconstants.kpc_cm (joxsz_funcs.py:6, used at joxsz_funcs.py:459)
KPC_CM = 3.0856776e21
# joxsz_main.py:22-23
M_E_KEV = 0.5109989 * 1e3
SIGMA_T_CM2 = 6.6524587158 * 1e-25

# Semantic slot of every physical parameter in the device-side parameter
# vector.  Order = insertion order of ``fit.pars`` in the reference
# (ne block joxsz_funcs.py:358-366, T joxsz_funcs.py:318, Z joxsz_funcs.py:246,
# pressure joxsz_funcs.py:266-272, joxsz_main.py:156-157), then the three
# extra parameters of the 'double' density mode (joxsz_funcs.py:367-372).
PAR_SLOTS = (
    'log(n_0)', r'\beta', 'log(r_c)', 'log(r_s)', r'\alpha', r'\epsilon', r'\gamma',
    'log(T_X/T_{SZ})', 'Z', 'P_0', 'a', 'b', 'c', 'r_p', 'backscale', 'calibration',
    'log(n_{02})', r'\beta_2', 'log(r_{c2})',
)
NPAR_SINGLE = 16
NPAR_DOUBLE = 19


def default_par_table(ne_mode='single', logr_max=3.7):
    """The parameter table ``joxsz_main.py`` ends up with at line 178.

    Defaults and bounds: joxsz_funcs.py:266-272 (pressure), :318 (T ratio),
    :358-372 (density), :246 + joxsz_main.py:131 (Z); overrides
    joxsz_main.py:156-175.  ``logr_max`` is ``annuli.edges_logkpc[-2]``
    (joxsz_main.py:160-161).
    Returns dict of arrays: names, vals, min, max, frozen, kind, mu, sigma.
    """
    rows = [
        # name, val, min, max, frozen, kind(0 box / 1 gauss), mu, sigma
        ('log(n_0)', -3., -7., 2., False, 0, 0., 0.),
        (r'\beta', 2 / 3, 0., 4., False, 0, 0., 0.),
        ('log(r_c)', 2., -1., logr_max, False, 0, 0., 0.),
        ('log(r_s)', 2.7, 0., logr_max, False, 0, 0., 0.),
        (r'\alpha', 0., -1., 2., True, 0, 0., 0.),
        (r'\epsilon', 3., 0., 10., False, 0, 0., 0.),
        (r'\gamma', 3., 0., 10., True, 0, 0., 0.),
        ('log(T_X/T_{SZ})', 0., -1., 1., False, 0, 0., 0.),
        ('Z', 0.3, 0., 1., False, 0, 0., 0.),
        ('P_0', 0.4, 0., 2., False, 0, 0., 0.),
        ('a', 1.33, 0.1, 20., False, 0, 0., 0.),
        ('b', 4.13, 0.1, 15., False, 0, 0., 0.),
        ('c', 0.014, 0., 3., True, 0, 0., 0.),
        ('r_p', 300., 100., 3000., False, 0, 0., 0.),
        ('backscale', 1., -1e99, 1e99, False, 1, 1., 0.1),
        ('calibration', 1., -1e99, 1e99, False, 1, 1., 0.07),
    ]
    if ne_mode == 'double':
        rows += [
            ('log(n_{02})', -1., -7., 2., False, 0, 0., 0.),
            (r'\beta_2', 0.5, 0., 4., False, 0, 0., 0.),
            ('log(r_{c2})', 1.7, -1., 3.7, False, 0, 0., 0.),
        ]
    return dict(
        par_names=[r[0] for r in rows],
        par_vals=np.array([r[1] for r in rows], dtype=np.float64),
        par_min=np.array([r[2] for r in rows], dtype=np.float64),
        par_max=np.array([r[3] for r in rows], dtype=np.float64),
        par_frozen=np.array([r[4] for r in rows], dtype=bool),
        par_kind=np.array([r[5] for r in rows], dtype=np.int32),
        par_mu=np.array([r[6] for r in rows], dtype=np.float64),
        par_sigma=np.array([r[7] for r in rows], dtype=np.float64),
    )


@dataclass
class Problem:
    """Everything constant over an MCMC run.  Shapes: S = map side, N = radial
    grid points, B = beam side, nflux = SZ data points, nann = X-ray annuli,
    nband = X-ray bands, ntab = count-rate table length."""
    # --- SZ (SZ_data, joxsz_funcs.py:157-170) ---
    step: float                       # arcsec
    kpc_as: float
    conv_T: np.ndarray                # [nconv] keV          (joxsz_main.py:108)
    conv_v: np.ndarray                # [nconv] 1e3*Jy/beam per unit y (joxsz_main.py:109)
    flux_data: np.ndarray             # [3, nflux] radius(arcsec), flux, err
    beam_2d: np.ndarray               # [B, B]
    radius: np.ndarray                # [S] arcsec
    r_pp: np.ndarray                  # [N] kpc
    d_mat: np.ndarray                 # [S, S] kpc
    filtering: np.ndarray             # [S, S] transfer function, FFT layout
    # --- parameter table ---
    par_names: list
    par_vals: np.ndarray
    par_min: np.ndarray
    par_max: np.ndarray
    par_frozen: np.ndarray
    par_kind: np.ndarray
    par_mu: np.ndarray
    par_sigma: np.ndarray
    # --- X-ray ---
    x_r_ne_kpc: np.ndarray            # [nann] radii where n_e is evaluated
    x_r_T_kpc: np.ndarray             # [nann] radii where T is evaluated (annuli.midpt_kpc, joxsz_funcs.py:339)
    projvols: np.ndarray              # [nann, nann] cm^3
    cts: np.ndarray                   # [nband, nann] (NaN = missing, joxsz_funcs.py:504)
    areascales: np.ndarray            # [nband, nann]
    exposures: np.ndarray             # [nband, nann]
    backrates: np.ndarray             # [nband, nann]
    geomarea: np.ndarray              # [nann] arcmin^2
    lnT: np.ndarray                   # [ntab]
    lnrate: np.ndarray                # [nband, 2, ntab] ln(rate) at Z=0, Z=1 solar
    # --- scalars / switches ---
    m_e: float = M_E_KEV
    sigma_T: float = SIGMA_T_CM2
    kpc_cm: float = KPC_CM
    ne_mode: str = 'single'
    exclude_unphy_mass: bool = True   # joxsz_main.py:88
    sz_only: bool = False             # build extension: skip the X-ray term (BASELINE configs[1])
    calc_integ: bool = False          # SZ_data.calc_integ (joxsz_main.py:65): integrated-Compton term, joxsz_funcs.py:480-484
    integ_mu: float = .94 / 1e3       # joxsz_main.py:66
    integ_sig: float = .36 / 1e3      # joxsz_main.py:67
    meta: dict = field(default_factory=dict)

    # ---- derived sizes ----
    @property
    def S(self):
        return int(self.d_mat.shape[0])

    @property
    def N(self):
        return int(self.r_pp.size)

    @property
    def B(self):
        return int(self.beam_2d.shape[0])

    @property
    def nrow(self):
        """Length of the extracted profile ``map_out[S//2, S//2:]`` (joxsz_funcs.py:472)."""
        return self.S - self.S // 2

    @property
    def thawed_idx(self):
        """Indices into the parameter table in the order of ``fit.thawed``
        (joxsz_main.py:179)."""
        return np.array([k for k in range(len(self.par_names)) if not self.par_frozen[k]], dtype=np.int32)

    @property
    def thawed(self):
        return [self.par_names[k] for k in self.thawed_idx]

    @property
    def ndim(self):
        return int(self.thawed_idx.size)

    def thawed_vals(self):
        """mbproj2 ``Fit.thawedParVals`` (used at joxsz_funcs.py:555,585)."""
        return self.par_vals[self.thawed_idx].copy()

    def integ_weights(self):
        """w [N+1] with  cint = sum_j w_j v_j,  v = [f(0), y_0 .. y_{N-1}]: joxsz_funcs.py:481-483 is Simpson's rule over
        the arcmin grid x = arange(0, r_pp[-1]/kpc_as/60 + step/60, step/60) of the integrand v x, times 2 pi -- a fixed
        linear functional of v.  The rule is scipy's (``simps`` of the reference = ``scipy.integrate.simpson`` today),
        applied here to the unit vectors, so whatever it does for an even number of samples is reproduced."""
        from scipy.integrate import simpson
        x = np.arange(0., self.r_pp[-1] / self.kpc_as / 60 + self.step / 60, self.step / 60)
        if x.size != self.N + 1:
            raise ValueError('calc_integ: the arcmin grid of joxsz_funcs.py:482 has %d points for %d samples '
                             '(r_pp must be step*kpc_as*(1..N))' % (x.size, self.N + 1))
        return 2 * np.pi * x * simpson(np.eye(x.size), x=x, axis=1)

    def validate(self):
        S = self.S
        assert self.d_mat.shape == (S, S) and self.filtering.shape == (S, S)
        assert self.radius.shape == (S,)
        assert self.beam_2d.shape[0] == self.beam_2d.shape[1] and self.B % 2 == 1
        assert self.r_pp.ndim == 1 and self.N >= self.nrow + 1, \
            'radial grid must cover the extracted profile (r_pp[:nrow-1], joxsz_funcs.py:469)'
        assert np.all(np.diff(self.r_pp) > 0) and self.r_pp[0] > 0
        assert self.flux_data.shape[0] == 3
        nb, na = self.cts.shape
        for a in (self.areascales, self.exposures, self.backrates):
            assert a.shape == (nb, na)
        assert self.projvols.shape == (na, na) and self.geomarea.shape == (na,)
        assert self.lnrate.shape == (nb, 2, self.lnT.size)
        assert np.all(np.diff(self.lnT) > 0)
        npar = len(self.par_names)
        assert npar == (NPAR_DOUBLE if self.ne_mode == 'double' else NPAR_SINGLE)
        assert list(self.par_names) == list(PAR_SLOTS[:npar])
        return self

    # ---- (de)serialisation for fixtures ----
    _ARRAYS = ('conv_T', 'conv_v', 'flux_data', 'beam_2d', 'radius', 'r_pp', 'd_mat', 'filtering',
               'par_vals', 'par_min', 'par_max', 'par_frozen', 'par_kind', 'par_mu', 'par_sigma',
               'x_r_ne_kpc', 'x_r_T_kpc', 'projvols', 'cts', 'areascales', 'exposures', 'backrates',
               'geomarea', 'lnT', 'lnrate')
    _SCALARS = ('step', 'kpc_as', 'm_e', 'sigma_T', 'kpc_cm', 'ne_mode', 'exclude_unphy_mass', 'sz_only',
                'calc_integ', 'integ_mu', 'integ_sig')

    def to_dict(self, prefix='pb_'):
        d = {prefix + k: np.asarray(getattr(self, k)) for k in self._ARRAYS}
        for k in self._SCALARS:
            d[prefix + k] = np.asarray(getattr(self, k))
        d[prefix + 'par_names'] = np.array([n.encode('utf-8') for n in self.par_names])
        return d

    @classmethod
    def from_dict(cls, d, prefix='pb_'):
        kw = {k: np.array(d[prefix + k]) for k in cls._ARRAYS}
        kw['par_frozen'] = kw['par_frozen'].astype(bool)
        kw['par_kind'] = kw['par_kind'].astype(np.int32)
        for k in ('step', 'kpc_as', 'm_e', 'sigma_T', 'kpc_cm'):
            kw[k] = float(d[prefix + k])
        kw['ne_mode'] = str(d[prefix + 'ne_mode'])
        kw['exclude_unphy_mass'] = bool(d[prefix + 'exclude_unphy_mass'])
        kw['sz_only'] = bool(d[prefix + 'sz_only'])
        if prefix + 'calc_integ' in d:                      # (fixtures written before the branch existed lack these)
            kw['calc_integ'] = bool(d[prefix + 'calc_integ'])
            kw['integ_mu'], kw['integ_sig'] = float(d[prefix + 'integ_mu']), float(d[prefix + 'integ_sig'])
        kw['par_names'] = [n.decode('utf-8') for n in d[prefix + 'par_names']]
        return cls(**kw)


_ = fields  # (dataclasses.fields kept importable for callers that introspect)
