"""ctypes binding of the C-ABI in ``include/joxsz_hip.h``.

There is no CPU fallback: if ``joxsz_amd/csrc/libjoxsz_hip.so`` is missing or
no gfx950 device answers, construction raises ``JoxszHipError``.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('JOXSZ_LIB') or os.path.join(_HERE, 'csrc', 'libjoxsz_hip.so')    # JOXSZ_LIB: A/B builds
ABI_VERSION = 5

# must list every function include/joxsz_hip.h declares (tests/test_abi.py checks this)
EXPORTS = (
    'jx_create', 'jx_upload', 'jx_finalize', 'jx_eval', 'jx_eval_device', 'jx_sync', 'jx_set_stream', 'jx_sample', 'jx_eval_stage',
    'jx_set_route', 'jx_get_route', 'jx_get_operator', 'jx_set_option', 'jx_audit',
    'jx_set_par_vals', 'jx_dev_alloc', 'jx_dev_free', 'jx_memcpy_h2d', 'jx_memcpy_d2h',
    'jx_timing_reset', 'jx_timing_enable', 'jx_timing_get', 'jx_get_info', 'jx_get_conv_mode', 'jx_get_conv_layout', 'jx_get_fft_info', 'jx_debug_workspace', 'jx_device_count',
    'jx_device_name', 'jx_strerror', 'jx_last_error', 'jx_destroy',
    'jx_get_truncation', 'jx_get_output_pruning', 'jx_get_sampling', 'jx_get_radial_sampling', 'jx_comm_unique_id', 'jx_comm_init_rank', 'jx_allgather_logp', 'jx_comm_allreduce_max', 'jx_comm_destroy',
    'jx_comm_count', 'jx_comm_set_overlap', 'jx_comm_gather_time', 'jx_map_kernel_time', 'jx_event_bracket_time', 'jx_fastmath_eval', 'jx_copy_bandwidth', 'jx_stream_bandwidth',
)

TENSORS = ('r_pp', 'd_mat', 'beam_2d', 'filtering', 'radius', 'flux_data', 'conv_T', 'conv_v',
           'par_vals', 'par_min', 'par_max', 'par_kind', 'par_mu', 'par_sigma', 'thawed_idx',
           'x_r_ne_kpc', 'x_r_T_kpc', 'projvols', 'cts', 'areascales', 'exposures', 'backrates',
           'geomarea', 'lnT', 'lnrate', 'integ_w')
_INT_TENSORS = ('par_kind', 'thawed_idx')
_XRAY_FIRST = TENSORS.index('x_r_ne_kpc')

STAGES = ('pp', 'ab', 'y', 'y_2d', 'conv_2d', 'map_row', 'bright', 'chisq', 'tprof', 'xprofs', 'parts', 'integ')


class JoxszHipError(RuntimeError):
    pass


class JoxszTruncationWarning(UserWarning):
    """The truncation guard of the low-rank form rebuilt the tables at a higher rank (slower), or sits close to its bound."""


class JxConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        'abi_version', 'S', 'N', 'B', 'nflux', 'nconv', 'nann', 'nband', 'ntab', 'npar', 'ndim',
        'ne_mode', 'exclude_unphy_mass', 'sz_only', 'device', 'max_batch', 'fft_pad', 'map_split',
        'conv_mode', 'dtype', 'calc_integ', 'reserved1')] + \
        [(n, ctypes.c_double) for n in ('step', 'kpc_as', 'm_e', 'sigma_T', 'kpc_cm', 'integ_mu', 'integ_sig')]


class JxTiming(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in (
        'prep_ms', 'abel_map_ms', 'beam_fft_ms', 'tf_fft_ms', 'tail_ms', 'total_ms')] + \
        [('launches', ctypes.c_int64), ('walkers', ctypes.c_int64), ('gemm_ms', ctypes.c_double)]


_lib = None


def load_library(path=None):
    """dlopen the HIP library and declare the prototypes.  Raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise JoxszHipError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                            '(there is no CPU fallback)' % path)
    lib = ctypes.CDLL(path)
    vp, ci, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    dp = ctypes.POINTER(ctypes.c_double)
    lib.jx_create.argtypes = [ctypes.POINTER(JxConfig), ctypes.POINTER(vp)]
    lib.jx_upload.argtypes = [vp, ci, vp, cs]
    lib.jx_finalize.argtypes = [vp]
    lib.jx_eval.argtypes = [vp, dp, ci, dp]
    lib.jx_eval_device.argtypes = [vp, vp, ci, vp]
    lib.jx_sync.argtypes = [vp]
    lib.jx_set_stream.argtypes = [vp, vp]
    lib.jx_set_route.argtypes = [vp, ci]
    lib.jx_get_route.argtypes = [vp]
    lib.jx_get_operator.argtypes = [vp, dp, cs]
    lib.jx_sample.argtypes = [vp, dp, ci, ci, ctypes.c_double, ctypes.c_uint64, dp, dp, ctypes.POINTER(ctypes.c_int64)]
    lib.jx_eval_stage.argtypes = [vp, dp, ci, ci, dp, cs]
    lib.jx_set_par_vals.argtypes = [vp, dp, ci]
    lib.jx_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
    lib.jx_audit.argtypes = [vp, dp, ci, dp]
    lib.jx_dev_alloc.argtypes = [vp, cs, ctypes.POINTER(vp)]
    lib.jx_dev_free.argtypes = [vp, vp]
    lib.jx_memcpy_h2d.argtypes = [vp, vp, vp, cs]
    lib.jx_memcpy_d2h.argtypes = [vp, vp, vp, cs]
    lib.jx_timing_reset.argtypes = [vp]
    lib.jx_timing_enable.argtypes = [vp, ci]
    lib.jx_timing_get.argtypes = [vp, ctypes.POINTER(JxTiming)]
    i32p, i64p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
    lib.jx_get_info.argtypes = [vp, i32p, i32p, i32p, i32p, i64p]
    lib.jx_get_conv_mode.argtypes = [vp]
    lib.jx_get_conv_layout.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    lib.jx_get_truncation.argtypes = [vp, dp]
    lib.jx_get_output_pruning.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    lib.jx_get_sampling.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ci]
    lib.jx_get_radial_sampling.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ci]
    lib.jx_debug_workspace.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int32)]
    lib.jx_comm_unique_id.argtypes = [vp]
    lib.jx_comm_init_rank.argtypes = [vp, vp, ci, ci]
    lib.jx_allgather_logp.argtypes = [vp, vp, vp, ci]
    lib.jx_comm_allreduce_max.argtypes = [vp, vp, ci]
    lib.jx_comm_destroy.argtypes = [vp]
    lib.jx_comm_count.argtypes = [vp]
    lib.jx_comm_set_overlap.argtypes = [vp, ci]
    lib.jx_comm_gather_time.argtypes = [vp, dp, i64p]
    lib.jx_map_kernel_time.argtypes = [vp, vp, ci, ci, dp]
    lib.jx_event_bracket_time.argtypes = [vp, ci, dp]
    lib.jx_fastmath_eval.argtypes = [vp, dp, ci, dp, dp]
    lib.jx_copy_bandwidth.argtypes = [vp, cs, ci, dp]
    lib.jx_stream_bandwidth.argtypes = [vp, ci, cs, ci, dp]
    lib.jx_device_count.argtypes = []
    lib.jx_device_name.argtypes = [vp]
    lib.jx_device_name.restype = ctypes.c_char_p
    lib.jx_strerror.argtypes = [ci]
    lib.jx_strerror.restype = ctypes.c_char_p
    lib.jx_last_error.argtypes = [vp]
    lib.jx_last_error.restype = ctypes.c_char_p
    lib.jx_destroy.argtypes = [vp]
    lib.jx_destroy.restype = None
    for name in EXPORTS:
        if name not in ('jx_device_name', 'jx_strerror', 'jx_last_error', 'jx_destroy'):
            getattr(lib, name).restype = ci
    if path == LIB_PATH:
        _lib = lib
    return lib


CONV_MODES = {'auto': 0, 'rocfft': 1, 'custom': 2, 'mix': 2}        # 'custom' = 'mix': the contracted route (hand-written kernels)
ROUTES = {'map': 0, 'operator': 1}


DTYPES = {'f64': 0, 'f32': 1, 'f32c': 2}        # 'f32': fp32 spline arrays, fp64 sums; 'f32c': fp32 arithmetic in stages 1 and 2 as well


def config_from_problem(pb, device=0, max_batch=0, fft_pad=0, map_split=0, conv='auto', dtype='f64'):
    cfg = JxConfig()
    cfg.abi_version = ABI_VERSION
    cfg.S, cfg.N, cfg.B = pb.S, pb.N, pb.B
    cfg.nflux = int(pb.flux_data.shape[1])
    cfg.nconv = int(np.size(pb.conv_T))
    cfg.nband, cfg.nann = (int(v) for v in pb.cts.shape)
    cfg.ntab = int(np.size(pb.lnT))
    cfg.npar = len(pb.par_names)
    cfg.ndim = pb.ndim
    cfg.ne_mode = 1 if pb.ne_mode == 'double' else 0
    cfg.exclude_unphy_mass = int(bool(pb.exclude_unphy_mass))
    cfg.sz_only = int(bool(pb.sz_only))
    cfg.device, cfg.max_batch, cfg.fft_pad, cfg.map_split = device, max_batch, fft_pad, map_split
    cfg.conv_mode = CONV_MODES[conv]
    cfg.dtype = DTYPES[dtype]
    cfg.calc_integ = int(bool(getattr(pb, 'calc_integ', False)))
    cfg.integ_mu, cfg.integ_sig = float(getattr(pb, 'integ_mu', 0.0)), float(getattr(pb, 'integ_sig', 1.0))
    cfg.step, cfg.kpc_as = float(pb.step), float(pb.kpc_as)
    cfg.m_e, cfg.sigma_T, cfg.kpc_cm = float(pb.m_e), float(pb.sigma_T), float(pb.kpc_cm)
    return cfg


class HipContext:
    """Thin owner of one ``jx_ctx``: uploads a ``Problem`` and evaluates batches."""

    def __init__(self, pb, device=0, max_batch=0, fft_pad=0, map_split=0, conv='auto', lib_path=None, route=None, dtype='f64', options=None):
        self._h = ctypes.c_void_p()
        self.lib = load_library(lib_path)
        pb.validate()
        self.pb = pb
        self.ndim = pb.ndim
        cfg = config_from_problem(pb, device, max_batch, fft_pad, map_split, conv, dtype)
        self.dtype = dtype
        rc = self.lib.jx_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc != 0:
            self._h = ctypes.c_void_p()
            raise JoxszHipError('jx_create: %s' % self.lib.jx_strerror(rc).decode())
        try:
            # switches of the library for THIS context (jx_set_option; names with or without their JOXSZ_ prefix): the process environment
            # is only the default of each
            for k, v in (options or {}).items():
                self._chk(self.lib.jx_set_option(self._h, str(k).encode(), None if v is None else str(v).encode()), 'jx_set_option(%s)' % k)
            for tid, name in enumerate(TENSORS):
                if name == 'integ_w':
                    if not getattr(pb, 'calc_integ', False):
                        continue
                    a = pb.integ_weights()
                elif pb.sz_only and tid >= _XRAY_FIRST:
                    continue
                else:
                    a = pb.thawed_idx if name == 'thawed_idx' else getattr(pb, name)
                a = np.ascontiguousarray(a, dtype=np.int32 if name in _INT_TENSORS else np.float64)
                self._chk(self.lib.jx_upload(self._h, tid, a.ctypes.data_as(ctypes.c_void_p), a.nbytes),
                          'jx_upload(%s)' % name)
            self._chk(self.lib.jx_finalize(self._h), 'jx_finalize')
        except Exception:
            self.close()
            raise
        f, c, b, n = (ctypes.c_int32() for _ in range(4))
        nb = ctypes.c_int64()
        self._chk(self.lib.jx_get_info(self._h, f, c, b, n, nb), 'jx_get_info')
        self.fft_pad, self.chunk, self.spline_band, self.nrow, self.device_bytes = f.value, c.value, b.value, n.value, nb.value
        self.device_name = self.lib.jx_device_name(self._h).decode()
        self.conv = {1: 'rocfft', 2: 'custom'}.get(self.lib.jx_get_conv_mode(self._h), '?')
        self.conv_layout = None
        if self.conv == 'custom':
            self.conv_layout = self._layout()
        pr = (ctypes.c_int32 * 6)()
        self._chk(self.lib.jx_get_output_pruning(self._h, pr), 'jx_get_output_pruning')
        self.output_pruning = dict(nrow=int(pr[0]), outputs_read_by_the_tail=int(pr[1]), outputs_computed=int(pr[2]), tiles_per_block=int(pr[3]),
                                   k_slices=int(pr[4]), active=bool(pr[5]))
        sm = (ctypes.c_int32 * 8)()
        rows = (ctypes.c_int32 * 4096)()
        self._chk(self.lib.jx_get_sampling(self._h, sm, rows, 4096), 'jx_get_sampling')
        self.sampling = dict(rows_of_the_quadrant=int(sm[0]), rows_evaluated=int(sm[1]), full_below=int(sm[2]), every_second_up_to=int(sm[3]),
                             interpolation_points=int(sm[4]), active=bool(sm[5]), removed_by_the_guard=int(sm[6]),
                             rows=np.array(rows[:min(int(sm[1]), 4096)], dtype=np.int64))
        rm = (ctypes.c_int32 * 6)()
        rrows = (ctypes.c_int32 * 8192)()
        removed = self.lib.jx_get_radial_sampling(self._h, rm, rrows, 8192)
        if removed < 0:
            self._chk(removed, 'jx_get_radial_sampling')
        self.radial_sampling = dict(radii_of_the_grid=int(rm[0]), radii_in_use=int(rm[1]), full_below=int(rm[2]), every_second_up_to=int(rm[3]),
                                    interpolation_points=int(rm[4]), active=bool(rm[5]), removed_by_the_guard=int(removed),
                                    rows=np.array(rrows[:min(int(rm[1]), 8192)], dtype=np.int64))
        self.truncation = self._truncation()
        if self.truncation['warning'] and not os.environ.get('JOXSZ_QUIET'):
            import warnings
            warnings.warn(self.truncation['warning'], JoxszTruncationWarning, stacklevel=3)
        self.route = 'map'
        route = route or os.environ.get('JOXSZ_ROUTE')
        if route and route != 'map':
            try:
                self.set_route(route)
            except Exception:
                self.close()
                raise

    def fft_info(self):
        """Transforms of the rocFFT sequence (this context's own route, or its reference facility once built): hand-written columns / rows or
        rocFFT plans, padded side, radices of the two transform lengths (jx_get_fft_info)."""
        o = (ctypes.c_int32 * 34)()
        self._chk(self.lib.jx_get_fft_info(self._h, o), 'jx_get_fft_info')
        v = [int(x) for x in o]
        if not v[0]:
            return {'built': False}
        d = {'built': True, 'columns': 'custom' if v[1] else 'rocfft', 'rows': 'custom' if v[2] else 'rocfft', 'fft_pad': v[3]}
        if v[1]:
            d.update(ld_padded=v[4], ld_window=v[5], columns_per_block=v[6], radices_padded=v[8:8 + v[7]], radices_window=v[21:21 + v[20]])
        return d

    def _layout(self):
        lay = (ctypes.c_int32 * 12)()
        self._chk(self.lib.jx_get_conv_layout(self._h, lay), 'jx_get_conv_layout')
        d = dict(zip(('form', 'NU', 'rank', 'beam_terms', 'R', 'RT', 'nxt', 'ntile', 'ksteps', 'tW', 'ldx', 'ksplit'), [int(v) for v in lay]))
        d['form'] = ('lowrank', 'full', 'exact')[d['form']]
        return d

    def _truncation(self):
        tr = (ctypes.c_double * 12)()
        self._chk(self.lib.jx_get_truncation(self._h, tr), 'jx_get_truncation')
        d = dict(tol=tr[0], est_rel_row_err=tr[1], rank=int(tr[2]), retried=int(tr[3]), points=int(tr[4]), bound=tr[5],
                 est_rel_row_err_box=tr[6], est_rel_sz_like_err_box=tr[7], rank_above_cut=int(tr[8]), bound_sz_like=tr[9],
                 stage1_on_matrix_cores=bool(tr[10]), cap_removed=int(tr[11]))
        d['warning'] = self._truncation_warning(d)
        return d

    def _truncation_warning(self, d):
        """One line when the guard changed the tables or has little room left (jx_finalize, joxsz_hip.hip): what happened and
        what it costs.  Stages 1 and 2 of the low-rank form cost about (4 + rank) and rank multiply-adds per map sample, so
        the step scales roughly with (4 + 2 rank) / (4 + 2 * 16) of the 16-term tables a smooth transfer function gets."""
        if self.conv != 'custom':
            return None
        msgs = []
        if self.radial_sampling['removed_by_the_guard']:
            msgs.append('the truncation guard took the radial sub-grid of the spline-array product away on these inputs: every radius of the profile '
                        'is multiplied (%d instead of %d: that kernel costs about 2x)' % (self.radial_sampling['radii_of_the_grid'], self.radial_sampling['radii_of_the_grid'] * 4 // 9))
        if self.sampling['removed_by_the_guard']:
            msgs.append('the truncation guard took the sub-grid of map samples away on these inputs: every distinct sample is evaluated '
                        '(%d rows and columns instead of about %d: the SZ stages cost 2-4x)' % (self.sampling['rows_of_the_quadrant'], self.sampling['rows_of_the_quadrant'] // 2))
        if d['retried'] > 0 and d['rank'] <= 0:
            msgs.append('the truncation guard found the low-rank tables of this beam / transfer function outside its bounds and rebuilt them until the '
                        'exact full form was the cheaper one: nothing is truncated now; the SZ stages cost about 1.6x those of the 16-term tables')
        elif d['rank'] > 0 and d['est_rel_row_err'] >= 0:
            rel = (4.0 + 2.0 * d['rank']) / (4.0 + 2.0 * 16)
            if d['retried'] > d['cap_removed']:
                msgs.append('the truncation guard tightened the singular-value cut to %.0e on this beam / transfer function: %d terms kept, '
                            'the SZ stages cost about %.1fx those of the 16-term tables' % (d['tol'], d['rank'], rel))
            elif d['cap_removed']:
                msgs.append('the truncation guard took the 16-term cap away on this beam / transfer function: %d terms kept '
                            '(SZ stages about %.2fx)' % (d['rank'], rel))
            near = max(d['est_rel_row_err'] / d['bound'], d['est_rel_sz_like_err_box'] / d['bound_sz_like'])
            if self.dtype != 'f64' and near > 1.0:
                # (an fp32 context is measured but never rebuilt -- its rounding is of the bounds' size -- so nothing takes its 16-term cap or
                #  its sub-grids away: say what was measured)
                msgs.append('fp32 context: the contracted tables read %.1e on the row and %.1e on the SZ log-likelihood over the prior box (bounds %.0e / %.0e) and are '
                            'NOT rebuilt for fp32 contexts; build the f64 context (the exact form) or check the walkers with audit()'
                            % (d['est_rel_row_err'], d['est_rel_sz_like_err_box'], d['bound'], d['bound_sz_like']))
            if not msgs and near > 0.5 and self.dtype == 'f64':
                msgs.append('the truncation of the low-rank form sits at %.0f %% of its bound (SZ log-likelihood over the prior box %.1e of %.0e): '
                            'slightly different inputs rebuild the tables with more terms (about -35 %% throughput at 31 terms)'
                            % (100 * near, d['est_rel_sz_like_err_box'], d['bound_sz_like']))
        return '; '.join(msgs) if msgs else None

    # -- plumbing --
    def _chk(self, rc, what):
        if rc != 0:
            detail = self.lib.jx_last_error(self._h).decode() if self._h else ''
            raise JoxszHipError('%s failed: %s%s' % (what, self.lib.jx_strerror(rc).decode(),
                                                     (' -- ' + detail) if detail else ''))

    def close(self):
        if getattr(self, '_h', None):
            self.lib.jx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _theta(self, theta):
        t = np.ascontiguousarray(theta, dtype=np.float64)
        if t.ndim == 1:
            t = t[None, :]
        if t.ndim != 2 or t.shape[1] != self.ndim:
            raise ValueError('theta must have shape (nwalkers, %d); got %r' % (self.ndim, np.shape(theta)))
        return t

    # -- evaluation --
    def eval(self, theta):
        """theta [W, ndim] float64 (host) -> log-posterior [W] float64 (host)."""
        t = self._theta(theta)
        out = np.empty(t.shape[0], dtype=np.float64)
        dp = ctypes.POINTER(ctypes.c_double)
        self._chk(self.lib.jx_eval(self._h, t.ctypes.data_as(dp), t.shape[0], out.ctypes.data_as(dp)), 'jx_eval')
        return out

    def eval_stage(self, theta, stage):
        t = self._theta(theta)
        W = t.shape[0]
        pb = self.pb
        shapes = {'pp': (W, pb.N), 'ab': (W, pb.N), 'y': (W, pb.N), 'y_2d': (W, pb.S, pb.S),
                  'conv_2d': (W, pb.S, pb.S), 'map_row': (W, self.nrow), 'bright': (W, self.nrow),
                  'chisq': (W,), 'tprof': (W, self.nrow), 'xprofs': (W,) + tuple(pb.cts.shape), 'parts': (W, 4), 'integ': (W,)}
        out = np.empty(shapes[stage], dtype=np.float64)
        dp = ctypes.POINTER(ctypes.c_double)
        self._chk(self.lib.jx_eval_stage(self._h, t.ctypes.data_as(dp), W, STAGES.index(stage),
                                         out.ctypes.data_as(dp), out.nbytes), 'jx_eval_stage(%s)' % stage)
        return out

    def set_option(self, name, value):
        """One switch of the library (``jx_set_option``): after construction only the two JOXSZ_SAMPLE_ options can still change."""
        self._chk(self.lib.jx_set_option(self._h, str(name).encode(), None if value is None else str(value).encode()), 'jx_set_option(%s)' % name)

    def audit(self, theta):
        """Run-time assurance (``jx_audit``): the walkers through this context's route and through the rocFFT sequence held inside it."""
        t = self._theta(theta)
        out = (ctypes.c_double * 4)()
        self._chk(self.lib.jx_audit(self._h, t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), t.shape[0], out), 'jx_audit')
        return dict(max_abs_sz_loglike_diff=out[0], max_rel_row_diff=out[1], worst_walker=int(out[2]), walkers_compared=int(out[3]))

    def set_par_vals(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.float64)
        self._chk(self.lib.jx_set_par_vals(self._h, v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), v.size),
                  'jx_set_par_vals')

    # -- device-resident path (bench, multi-GPU) --
    def dev_alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._chk(self.lib.jx_dev_alloc(self._h, nbytes, ctypes.byref(p)), 'jx_dev_alloc')
        return p.value

    def dev_free(self, ptr):
        self._chk(self.lib.jx_dev_free(self._h, ctypes.c_void_p(ptr)), 'jx_dev_free')

    def h2d(self, ptr, arr):
        a = np.ascontiguousarray(arr)
        self._chk(self.lib.jx_memcpy_h2d(self._h, ctypes.c_void_p(ptr), a.ctypes.data_as(ctypes.c_void_p), a.nbytes), 'jx_memcpy_h2d')

    def workspace(self, which):
        """Test hook: copy of a work buffer of the contracted route (``jx_debug_workspace``), holding the last evaluated
        chunk: 'y_map' [chunk, NU, ld] quadrant of the Compton-y map (after a y_2d tap), 'splines' [N, tW, 2] walker-minor
        (y_k, M_k), 'stage1' [NU, R, tW] rows kept per map column, 'partials' [ksplit, tW, ldx] partial rows; the constant
        operators 'stage1_op' [1, wld, cld] (C[u][j]) and 'product_op' [K, 16, ntile] (Op[kappa][x & 15][x >> 4])."""
        ids = {'y_map': 0, 'splines': 1, 'stage1': 2, 'partials': 3, 'stage1_op': 4, 'product_op': 5, 'ordinates': 6, 'row_op': 7}
        ptr = ctypes.c_void_p()
        geom = (ctypes.c_int32 * 4)()
        self._chk(self.lib.jx_debug_workspace(self._h, ids[which], ctypes.byref(ptr), geom), 'jx_debug_workspace')
        self.sync()
        if which == 'partials':                              # slices are 272 doubles apart beyond their [tW][ldx] rows
            ks, tW, ldx = geom[0], geom[1], geom[2]
            raw = np.empty(ks * (tW * ldx + 272))
            self.d2h(raw, ptr.value)
            return np.ascontiguousarray(raw.reshape(ks, tW * ldx + 272)[:, :tW * ldx]).reshape(ks, tW, ldx)
        out = np.empty((geom[0], geom[1], geom[2]), np.float32 if geom[3] == 4 else np.float64)
        self.d2h(out, ptr.value)
        return out

    def d2h(self, arr, ptr):
        assert arr.flags['C_CONTIGUOUS']
        self._chk(self.lib.jx_memcpy_d2h(self._h, arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), arr.nbytes), 'jx_memcpy_d2h')

    def eval_device(self, theta_ptr, nwalkers, logp_ptr):
        self._chk(self.lib.jx_eval_device(self._h, ctypes.c_void_p(theta_ptr), nwalkers, ctypes.c_void_p(logp_ptr)), 'jx_eval_device')

    def sample(self, theta0, nsteps, a=2.0, seed=0):
        """Device-resident stretch-move run (``jx_sample``): returns (chain[nsteps, W, ndim], logp[nsteps, W], naccept[W])."""
        th = np.ascontiguousarray(theta0, dtype=np.float64)
        W, ndim = th.shape
        chain = np.empty((nsteps, W, ndim)); lps = np.empty((nsteps, W)); nacc = np.zeros(W, np.int64)
        dp = ctypes.POINTER(ctypes.c_double)
        self._chk(self.lib.jx_sample(self._h, th.ctypes.data_as(dp), W, int(nsteps), float(a), int(seed) & (2 ** 64 - 1),
                                     chain.ctypes.data_as(dp), lps.ctypes.data_as(dp),
                                     nacc.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))), 'jx_sample')
        return chain, lps, nacc

    def set_route(self, route):
        """'map': the reference's own sequence of steps for every walker (default).  'operator': the SZ side as one
        constant nrow x N matrix applied to the pressure profile (``jx_set_route``; built on first use by sending the
        unit profiles through the 'map' kernels).  ``eval_stage`` always takes the 'map' route."""
        if route not in ROUTES:
            raise ValueError("route must be 'map' or 'operator'")
        self._chk(self.lib.jx_set_route(self._h, ROUTES[route]), 'jx_set_route')
        self.route = route

    def operator(self):
        """G [N, nrow]: row j is the map row (before conversion) of the unit pressure profile e_j."""
        out = np.empty((self.pb.N, self.nrow))
        self._chk(self.lib.jx_get_operator(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), out.nbytes), 'jx_get_operator')
        return out

    def set_stream(self, hip_stream):
        """Enqueue on the caller's hipStream_t (integer handle, e.g. ``torch.cuda.current_stream().cuda_stream``);
        ``None`` or 0 returns to the context's own stream."""
        self._chk(self.lib.jx_set_stream(self._h, ctypes.c_void_p(hip_stream or 0)), 'jx_set_stream')

    def sync(self):
        self._chk(self.lib.jx_sync(self._h), 'jx_sync')

    # -- RCCL through the C-ABI (no torch): see joxsz_amd/dist.py::RcclGather --
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        rc = self.lib.jx_comm_unique_id(buf)
        if rc != 0:
            detail = self.lib.jx_last_error(None).decode()
            raise JoxszHipError('jx_comm_unique_id failed: %s%s' % (self.lib.jx_strerror(rc).decode(), (' -- ' + detail) if detail else ''))
        return buf.raw

    def comm_init_rank(self, uid, nranks, rank):
        assert len(uid) == 128
        self._chk(self.lib.jx_comm_init_rank(self._h, ctypes.c_char_p(uid), nranks, rank), 'jx_comm_init_rank')

    def allgather_logp(self, send_ptr, recv_ptr, count):
        self._chk(self.lib.jx_allgather_logp(self._h, ctypes.c_void_p(send_ptr), ctypes.c_void_p(recv_ptr), count), 'jx_allgather_logp')

    def comm_allreduce_max(self, ptr, count=1):
        self._chk(self.lib.jx_comm_allreduce_max(self._h, ctypes.c_void_p(ptr), count), 'jx_comm_allreduce_max')

    def comm_set_overlap(self, on=True):
        """Collectives on a second stream: the next evaluation overlaps the gather unless it writes the buffer being sent."""
        self._chk(self.lib.jx_comm_set_overlap(self._h, int(bool(on))), 'jx_comm_set_overlap')

    def comm_gather_time(self):
        """(ms, calls): summed duration of the all-gathers on their own stream since the last call (timing must be enabled)."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        self._chk(self.lib.jx_comm_gather_time(self._h, ctypes.byref(ms), ctypes.byref(n)), 'jx_comm_gather_time')
        return ms.value, n.value

    def comm_destroy(self):
        self._chk(self.lib.jx_comm_destroy(self._h), 'jx_comm_destroy')

    def comm_count(self):
        n = self.lib.jx_comm_count(self._h)
        if n < 0:
            self._chk(n, 'jx_comm_count')
        return n

    def fastmath_eval(self, x):
        """Test hook: (exp(x), log(x)) as the per-walker kernel's table-driven functions compute them (``jx_fastmath_eval``)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        e, l = np.empty_like(x), np.empty_like(x)
        dp = ctypes.POINTER(ctypes.c_double)
        self._chk(self.lib.jx_fastmath_eval(self._h, x.ctypes.data_as(dp), int(x.size), e.ctypes.data_as(dp), l.ctypes.data_as(dp)), 'jx_fastmath_eval')
        return e, l

    def event_bracket_time(self, repeats=64):
        """What a pair of HIP events around one kernel of a dependent chain reads when the kernel does nothing (``jx_event_bracket_time``), ms."""
        ms = ctypes.c_double()
        self._chk(self.lib.jx_event_bracket_time(self._h, int(repeats), ctypes.byref(ms)), 'jx_event_bracket_time')
        return ms.value

    def map_kernel_time(self, theta_ptr, nwalkers, repeats=10):
        """Mean duration (ms) of the Abel + map kernel writing the full S x S map of ``nwalkers`` walkers (``jx_map_kernel_time``)."""
        ms = ctypes.c_double()
        self._chk(self.lib.jx_map_kernel_time(self._h, ctypes.c_void_p(theta_ptr), nwalkers, repeats, ctypes.byref(ms)), 'jx_map_kernel_time')
        return ms.value

    def stream_bandwidth(self, kind, nbytes=1 << 30, repeats=10):
        """What this GPU's HBM gives a plain stream, GB/s (``jx_stream_bandwidth``): kind 'copy' (bytes read + written), 'read'
        or 'write'.  The full-map kernel is a write stream."""
        g = ctypes.c_double()
        self._chk(self.lib.jx_stream_bandwidth(self._h, {'copy': 0, 'read': 1, 'write': 2}[kind], nbytes, repeats, ctypes.byref(g)),
                  'jx_stream_bandwidth')
        return g.value

    def copy_bandwidth(self, nbytes=1 << 30, repeats=10):
        """Device-to-device copy bandwidth in GB/s, bytes read + written (``jx_copy_bandwidth``): the practical HBM roofline."""
        g = ctypes.c_double()
        self._chk(self.lib.jx_copy_bandwidth(self._h, nbytes, repeats, ctypes.byref(g)), 'jx_copy_bandwidth')
        return g.value

    # -- timing --
    def timing_enable(self, on=True):
        self._chk(self.lib.jx_timing_enable(self._h, int(on)), 'jx_timing_enable')

    def timing_reset(self):
        self._chk(self.lib.jx_timing_reset(self._h), 'jx_timing_reset')

    def timing(self):
        t = JxTiming()
        self._chk(self.lib.jx_timing_get(self._h, ctypes.byref(t)), 'jx_timing_get')
        return {n: getattr(t, n) for n, _ in JxTiming._fields_}
