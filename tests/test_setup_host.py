"""The product's own setup layer (joxsz_amd/setup_host.py, datasets.py) against the
tensors the REFERENCE's setup functions produced (tests/golden, oracle/make_golden.py),
and the oracle's restatement of the same functions."""
import os

import numpy as np

from joxsz_amd import setup_host as sh, datasets
from oracle import joxsz_oracle as orc, mbproj2_parts as mbp

from conftest import GOLDEN


def _inputs():
    return np.load(os.path.join(GOLDEN, 'bundled_inputs.npz'))


def test_dist_and_centdistmat():
    z = _inputs()
    for n, key in ((9, 'dist9'), (8, 'dist8')):
        np.testing.assert_array_equal(sh.fft_frequency_radius(n), z[key])
        np.testing.assert_array_equal(orc.dist(n), z[key])
    r = 2.0 * (np.arange(171) - 85) * 8.0012
    np.testing.assert_array_equal(sh.pixel_radius_matrix(r), orc.centdistmat(r))


def test_beam_and_filter_images(golden_bundled):
    pb, _ = golden_bundled                       # beam_2d, filtering, d_mat, radius as the reference built them
    z = _inputs()
    beam, fwhm = sh.beam_image(2.0, z['flux_data'][0][-1], approx=False, profile=(z['beam_r'], z['beam_prof']))
    assert abs(fwhm - float(z['fwhm'])) < 1e-9 * fwhm
    assert beam.shape == pb.beam_2d.shape == (55, 55)
    np.testing.assert_allclose(beam, pb.beam_2d, rtol=1e-9, atol=1e-12 * pb.beam_2d.max())
    ob, ofw = orc.mybeam(2.0, z['flux_data'][0][-1], approx=False, beam_profile=(z['beam_r'], z['beam_prof']))
    np.testing.assert_allclose(ob, pb.beam_2d, rtol=1e-12, atol=1e-15)
    filt = sh.filter_image(z['wn_as'], z['tf'], 171, 2.0)
    np.testing.assert_allclose(filt, pb.filtering, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(orc.filt_image(z['wn_as'], z['tf'], 171, 2.0), pb.filtering, rtol=1e-13, atol=1e-15)
    radius, sep, r_pp = sh.sz_axes(2.0, pb.kpc_as, z['flux_data'][0][-1], fwhm, 5000.)
    assert sep == 85
    np.testing.assert_array_equal(radius, pb.radius)
    np.testing.assert_array_equal(r_pp, pb.r_pp)
    np.testing.assert_array_equal(sh.pixel_radius_matrix(radius * pb.kpc_as), pb.d_mat)


def test_gaussian_beam_branch(golden_tiny):
    pb, _ = golden_tiny                          # reference mybeam(approx=True), filt_image on the normal-cdf TF
    beam, _ = sh.beam_image(6.0, 0., approx=True, fwhm=8.5)
    np.testing.assert_allclose(beam, pb.beam_2d, rtol=1e-12)
    wn, tf = sh.transfer_function(np.linspace(0., 0.4967, 76), None, approx=True)
    np.testing.assert_allclose(sh.filter_image(wn, tf, 31, 6.0), pb.filtering, rtol=1e-10, atol=1e-14)


def test_xray_geometry(golden_bundled):
    pb, _ = golden_bundled
    z = _inputs()
    np.testing.assert_array_equal(sh.annuli_edges(z['fg'][0]), z['edges_arcmin'])
    geo = sh.annuli_geometry(z['edges_arcmin'], pb.kpc_as)
    e_cm = z['edges_arcmin'] * 60. * pb.kpc_as * mbp.kpc_cm
    V = mbp.projection_volume_matrix(e_cm)
    np.testing.assert_allclose(geo['projvols'], V, rtol=1e-10, atol=1e-13 * V.max())   # differences of cubes cancel
    # every shell inside the outermost annulus is seen whole: column sums = shell volumes
    vol = 4. / 3. * np.pi * (e_cm[1:] ** 3 - e_cm[:-1] ** 3)
    np.testing.assert_allclose(V.sum(axis=0), vol, rtol=1e-10)
    band = sh.band_from_profiles(z['fg'][3], z['bg'][3])
    np.testing.assert_array_equal(band['cts'], pb.cts[3])
    np.testing.assert_allclose(band['areascales'], pb.areascales[3], rtol=1e-14)


def test_cosmology_plate_scale():
    # joxsz_main.py:27-31; the survey quotes 8.0012 kpc/arcsec
    assert abs(mbp.kpc_per_arcsec(0.888) - datasets.KPC_AS_CLJ1226) < 2e-4


def test_fits_reader_matches_reference_columns():
    ref = '/root/reference/data/SZ/Beam150GHz.fits'
    if not os.path.exists(ref):
        import pytest
        pytest.skip('reference data not present on this box')
    z = _inputs()
    r, b = sh.clip_beam_profile(*sh.read_columns(ref, 2))
    np.testing.assert_array_equal(r, z['beam_r'])
    np.testing.assert_array_equal(b, z['beam_prof'])
