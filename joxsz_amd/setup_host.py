"""One-off host-side setup of the constant tensors (per run, not per walker).

Own numpy/scipy formulation of the reference's setup layer
(joxsz_funcs.py:16-134, 172-211 and joxsz_main.py:95-125): beam image, pixel
radius matrix, transfer-function image, X-ray annulus geometry, and a
dependency-free reader for the two FITS binary tables the reference ships
(astropy is not available on the target image).  Nothing here is on the
per-walker path.
"""
import math
import re
import numpy as np
from scipy.interpolate import CubicSpline
from scipy.special import ndtr

from .problem import KPC_CM


# ----------------------------------------------------------------------------
# readers (joxsz_funcs.py:16-44, 90-102)
# ----------------------------------------------------------------------------

_TFORM = re.compile(r"^\s*(\d*)([A-Z])")
_FITS_DTYPES = {'D': '>f8', 'E': '>f4', 'J': '>i4', 'I': '>i2', 'K': '>i8', 'B': 'u1'}


def read_fits_bintable_row(path, row=0):
    """Columns of one row of the first BINTABLE extension of a FITS file, as a
    list of float64/int arrays.  Replaces ``fits.open(f)[''].data[0]``
    (joxsz_funcs.py:23)."""
    blob = open(path, 'rb').read()
    pos = 0
    hdu = 0
    while pos < len(blob):
        cards = {}
        order = []
        while True:
            block = blob[pos:pos + 2880]
            pos += 2880
            done = False
            for k in range(0, 2880, 80):
                card = block[k:k + 80].decode('ascii', 'replace')
                key = card[:8].strip()
                if key == 'END':
                    done = True
                    break
                if card[8:10] == '= ':
                    val = card[10:].split('/')[0].strip()
                    cards[key] = val.strip("'").strip()
                    order.append(key)
            if done:
                break
        naxis = int(cards.get('NAXIS', 0))
        nbytes = 0
        if naxis:
            nbytes = abs(int(cards['BITPIX'])) // 8
            for a in range(1, naxis + 1):
                nbytes *= int(cards['NAXIS%d' % a])
            nbytes += int(cards.get('PCOUNT', 0))
        if hdu > 0 and cards.get('XTENSION', '').startswith('BINTABLE'):
            rowlen = int(cards['NAXIS1'])
            base = pos + row * rowlen
            cols = []
            off = 0
            for c in range(1, int(cards['TFIELDS']) + 1):
                m = _TFORM.match(cards['TFORM%d' % c])
                rep = int(m.group(1)) if m.group(1) else 1
                dt = np.dtype(_FITS_DTYPES[m.group(2)])
                arr = np.frombuffer(blob, dtype=dt, count=rep, offset=base + off)
                off += rep * dt.itemsize
                cols.append(arr.astype(np.float64) if dt.kind == 'f' else arr.astype(np.int64))
            return cols
        pos += (nbytes + 2879) // 2880 * 2880
        hdu += 1
    raise RuntimeError('no BINTABLE extension in %s' % path)


def read_columns(path, ncol):
    """First ``ncol`` columns of a FITS table row or of an ASCII table
    (joxsz_funcs.py:16-28)."""
    ext = path.rsplit('.', 1)[-1]
    if ext == 'fits':
        return read_fits_bintable_row(path)[:ncol]
    if ext in ('txt', 'dat'):
        return list(np.loadtxt(path, unpack=True)[:ncol])
    raise RuntimeError('Unrecognised file extension (not in fits, dat, txt)')


def clip_beam_profile(radius, prof):
    """Keep the beam profile up to its first NaN, then up to its first negative
    value (joxsz_funcs.py:30-44)."""
    radius = np.asarray(radius, dtype=np.float64)
    prof = np.asarray(prof, dtype=np.float64)
    bad = np.flatnonzero(np.isnan(prof))
    if bad.size:
        radius, prof = radius[:bad[0]], prof[:bad[0]]
    neg = np.flatnonzero(prof < 0.)
    if neg.size:
        radius, prof = radius[:neg[0]], prof[:neg[0]]
    return radius, prof


# ----------------------------------------------------------------------------
# images (joxsz_funcs.py:46-134)
# ----------------------------------------------------------------------------

def pixel_radius_matrix(axis):
    """``centdistmat`` (joxsz_funcs.py:78-88): M[i, j] = hypot-like sqrt(axis[j]^2 + axis[i]^2)."""
    a2 = np.asarray(axis, dtype=np.float64) ** 2
    return np.sqrt(a2[None, :] + a2[:, None])


def fft_frequency_radius(n):
    """``dist`` (joxsz_funcs.py:104-116): radial index distance in FFT layout."""
    ax = np.linspace(-n // 2 + 1, n // 2, n)
    m = np.sqrt(ax[None, :] ** 2 + ax[:, None] ** 2)
    return np.roll(m, n // 2 + 1, axis=(0, 1))


def _sym_cubic(x, y, fill):
    """Not-a-knot cubic through the mirrored samples (-x, y), (x, y); ``fill``
    outside -- what ``interp1d(append(-x, x), append(y, y), 'cubic', fill_value=fill)``
    evaluates (joxsz_funcs.py:61)."""
    xs = np.concatenate((-x[::-1], x))
    ys = np.concatenate((y[::-1], y))
    cs = CubicSpline(xs, ys, bc_type='not-a-knot')

    def f(q):
        q = np.asarray(q, dtype=np.float64)
        v = cs(q)
        v = np.where(q < xs[0], fill[0], v)
        v = np.where(q > xs[-1], fill[1], v)
        return v
    return f


def beam_image(step, maxr_data, approx=False, profile=None, fwhm=None, normalize=True):
    """``mybeam`` (joxsz_funcs.py:46-76).  ``profile`` = clipped (radius, beam)
    when ``approx`` is False.  Returns (beam_2d, fwhm)."""
    if not approx:
        r, b = profile
        f = _sym_cubic(r, b, (0., 0.))
        half = float(f(0.)) / 2
        # Newton from x0=5 as joxsz_funcs.py:62-63 does (secant form: no derivative given)
        x0, x1 = 5.0, 5.0 * (1 + 1e-4) + 1e-4
        f0, f1 = float(f(x0)) - half, float(f(x1)) - half
        for _ in range(50):
            if f1 == f0:
                break
            x2 = x1 - f1 * (x1 - x0) / (f1 - f0)
            x0, f0, x1 = x1, f1, x2
            f1 = float(f(x1)) - half
            if abs(x1 - x0) < 1.48e-8:
                break
        fwhm = 2 * x1
    maxr = (maxr_data + 3 * fwhm) // step * step
    rad = np.arange(0., maxr + step, step)
    rad = np.concatenate((-rad[:0:-1], rad))
    cut = rad[np.abs(rad) <= 3 * fwhm]
    dm = pixel_radius_matrix(cut)
    if approx:
        sig = fwhm / (2 * math.sqrt(2 * math.log(2)))
        img = np.exp(-0.5 * (dm / sig) ** 2) / (sig * math.sqrt(2 * math.pi))
    else:
        img = f(dm)
    if normalize:
        img = img / (img.sum() * step ** 2)
    return img, fwhm


def transfer_function(wn, tf, approx=False, loc=0., scale=0.02, c=0.95):
    """``read_tf`` (joxsz_funcs.py:90-102) after the columns are read."""
    wn = np.asarray(wn, dtype=np.float64)
    if approx:
        tf = c * ndtr((wn - loc) / scale)
    return wn, np.asarray(tf, dtype=np.float64)


def filter_image(wn, tf, side, step):
    """``filt_image`` (joxsz_funcs.py:118-134)."""
    cs = CubicSpline(wn, tf, bc_type='not-a-knot')
    k = fft_frequency_radius(side) / side
    k = k / k.max() * (1 / step)
    v = cs(k)
    v = np.where(k < wn[0], tf[0], v)
    v = np.where(k > wn[-1], tf[-1], v)
    return v


def sz_axes(step, kpc_as, maxr_data, fwhm, R_b):
    """joxsz_main.py:100-105: the map axis, centre index and radial grid."""
    mymaxr = (maxr_data + 3 * fwhm) // step * step
    radius = np.arange(0., mymaxr + step, step)
    radius = np.concatenate((-radius[:0:-1], radius))
    sep = radius.size // 2
    r_pp = np.arange(step * kpc_as, R_b + step * kpc_as, step * kpc_as)
    return radius, sep, r_pp


# ----------------------------------------------------------------------------
# X-ray geometry (mbproj2 Annuli, joxsz_main.py:116; joxsz_funcs.py:172-211)
# ----------------------------------------------------------------------------

def projection_volumes(edges_cm):
    """Volume of shell j seen through annulus i, both sides of the sky plane
    (mbproj2 ``utils.projectionVolumeMatrix``), vectorised."""
    e = np.asarray(edges_cm, dtype=np.float64)
    lo, hi = e[:-1], e[1:]
    y1, y2 = lo[:, None], hi[:, None]
    R1, R2 = lo[None, :], hi[None, :]

    def cube_root_term(R, y):
        return np.sqrt(np.clip(R * R - y * y, 0., None)) ** 3
    return (4. * math.pi / 3.) * ((cube_root_term(R1, y2) - cube_root_term(R1, y1))
                                  + (cube_root_term(R2, y1) - cube_root_term(R2, y2)))


def annuli_geometry(edges_arcmin, kpc_as):
    """mbproj2 ``Annuli``: mid-point radii (kpc), projection volumes (cm^3),
    geometric areas (arcmin^2), ln(edges/kpc)."""
    e_am = np.asarray(edges_arcmin, dtype=np.float64)
    e_kpc = e_am * 60. * kpc_as
    e_cm = e_kpc * KPC_CM
    mid_kpc = 0.5 * (e_kpc[1:] + e_kpc[:-1])
    geom = math.pi * (e_am[1:] ** 2 - e_am[:-1] ** 2)
    with np.errstate(divide='ignore'):
        edges_logkpc = np.log(e_kpc)          # natural log, as mbproj2 stores it
    return dict(midpt_kpc=mid_kpc, projvols=projection_volumes(e_cm), geomarea=geom,
                edges_logkpc=edges_logkpc)


def band_from_profiles(fg, bg):
    """``loadBand`` (joxsz_funcs.py:184-211) on already-loaded tables."""
    radii, hws, cts, areas, exps = (fg[:, k] for k in range(5))
    geom = math.pi * ((radii + hws) ** 2 - (radii - hws) ** 2)
    back = bg[:radii.size, 4]
    if abs(bg[:radii.size, 0][-1] - radii[-1]) > .001:
        raise RuntimeError('Problem while reading bg file', bg[radii.size - 1, 0], radii[-1])
    return dict(cts=cts, exposures=exps, areascales=areas / geom, backrates=back)


def annuli_edges(fg):
    """``getEdges`` (joxsz_funcs.py:172-182)."""
    return np.hstack((fg[0, 0] - fg[0, 1], fg[:, 0] + fg[:, 1]))
