"""Restatement of PyAbel's direct forward Abel transform, Python backend.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  **Parity unpinned**: PyAbel
is a third-party dependency named, unpinned, in
``/root/reference/requirements.txt:2``; its source is neither under
``/root/reference`` nor installed here, and the reference holds no test or
golden vector at this boundary.  What follows restates the published algorithm
of ``abel.direct.direct_transform`` / ``_pyabel_direct_integral`` (PyAbel
0.8.x, ``abel/direct.py``) as called at ``joxsz_funcs.py:457``::

    direct_transform(pp, r=r_pp, direction='forward', backend='Python')

Algorithm (forward direction, ``correction=True``, ``int_func=np.trapz``):

1. ``F_j = 2 r_j f_j``.
2. ``P[i, j] = F_j / sqrt(r_j^2 - r_i^2)`` for ``j > i`` and 0 otherwise.
3. ``out_i = trapz(P[i, :], r)``  (``dx=r[1]-r[0]`` when the grid passes
   ``is_uniform_sampling``: all second differences within 1e-13 of zero).
4. "Extra triangle" correction: subtract half the trapezoid integral of the
   row restricted to the columns ``j in {i, i+1}``.  Because a lone non-zero
   sample at an interior index contributes to two cells, this removes
   ``0.5*dx*P[i, i+1]`` for ``i+1 < N-1`` and ``0.25*dx*P[i, N-1]`` for
   ``i = N-2``.
5. Analytic integral of the singular cell assuming ``F`` linear on
   ``[r_i, r_{i+1}]``:  ``out_i += s_i F'_i + acosh(r_{i+1}/r_i) (F_i - F'_i r_i)``
   with ``s_i = sqrt(r_{i+1}^2 - r_i^2)``, ``F'_i = (F_{i+1}-F_i)/(r_{i+1}-r_i)``;
   for ``r_0 = 0`` the first ratio is replaced by ``cosh(1)``.
6. ``out_{N-1} = 0`` (no cell beyond the grid: the profile is truncated at
   ``r_N``, i.e. at ``R_b`` of ``joxsz_main.py:24``).
"""
import numpy as np


def is_uniform_sampling(r):
    """True when every second difference of ``r`` is within 1e-13 of zero."""
    dr = np.diff(r)
    ddr = np.diff(dr)
    return bool(np.allclose(ddr, 0.0, atol=1e-13))


def _trapz(P, r, uniform):
    if uniform:
        dx = abs(r[1] - r[0])
        return dx * (P[:, 1:] + P[:, :-1]).sum(axis=1) / 2.0
    d = np.diff(r)
    return (d[None, :] * (P[:, 1:] + P[:, :-1]) / 2.0).sum(axis=1)


def direct_transform_forward(f, r):
    """Forward Abel transform of the 1-D profile ``f`` sampled at radii ``r``.

    Mirrors ``direct_transform(f, r=r, direction='forward', backend='Python')``.
    Returns an array of the shape of ``f``.
    """
    f = np.asarray(f, dtype=np.float64)
    r = np.asarray(r, dtype=np.float64)
    n = r.size
    F = 2.0 * r * f
    uniform = is_uniform_sampling(r)

    R, Y = np.meshgrid(r, r, indexing='ij')          # R[i,j]=r_i, Y[i,j]=r_j
    ii = np.arange(n)
    II, JJ = np.meshgrid(ii, ii, indexing='ij')
    mask = II < JJ
    I_sqrt = np.zeros((n, n))
    I_sqrt[mask] = np.sqrt((Y ** 2 - R ** 2)[mask])
    I_isqrt = np.zeros((n, n))
    I_isqrt[mask] = 1.0 / I_sqrt[mask]
    mask2 = (II > JJ - 2) & (II < JJ + 1)            # columns j in {i, i+1}

    P = F[None, :] * I_isqrt
    out = _trapz(P, r, uniform)
    out = out - 0.5 * _trapz(P * mask2, r, uniform)

    f_r = (F[1:] - F[:-1]) / np.diff(r)
    isqrt = I_sqrt[II + 1 == JJ]
    if r[0] < r[1] * 1e-8:
        ratio = np.append(np.cosh(1.0), r[2:] / r[1:-1])
    else:
        ratio = r[1:] / r[:-1]
    acr = np.arccosh(ratio)
    out[:-1] += isqrt * f_r + acr * (F[:-1] - f_r * r[:-1])
    return out


def abel_weight_matrix(r):
    """The same transform as an explicit upper-triangular matrix ``A`` with
    ``direct_transform_forward(f, r) == A @ f`` (the transform is linear in
    ``f``).  Used by the tests as an independent cross-check."""
    r = np.asarray(r, dtype=np.float64)
    n = r.size
    A = np.zeros((n, n))
    eye = np.eye(n)
    for j in range(n):
        A[:, j] = direct_transform_forward(eye[j], r)
    return A
