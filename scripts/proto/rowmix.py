"""numpy statement of the contracted route (round 3): the sum over map rows is taken BEFORE any transform.

Reference lines restated: joxsz_funcs.py:462-467 and the row of :472.

    map   y2d[m][n]            (S x S, y2d[m][n] = Q[|m-c|][|n-c|], Q = the quadrant of distinct samples)
    conv  = fftconvolve(y2d, beam, 'same') * step^2                                  (funcs:464)
    out   = Re ifft2(fft2(conv) * filtering)[c, c:]                                  (funcs:466-467, 472)

With the beam image in separable form  step^2 beam[a][b] = sum_s by_s[a] bx_s[b]  (one term for the Gaussian branch of
mybeam, funcs:69-71; a few for a measured radial beam) and the transfer-function weights of the extracted row in
low-rank form  Hy[q][kx] = sum_rho U[rho][q] v_rho[kx]  (q = conv row, kx = column wavenumber):

    stage 1   D[(rho,s)][x'] = sum_u C[(rho,s)][u] Q[u][x'],   C[(rho,s)][u] = sum_q U[rho][q] sum_{m: |m-c|=u} by_s[q-m+o]
    stage 2   out[x] = sum_{(rho,s),x'} G[x][(rho,s)][x'] D[(rho,s)][x'],
              G[x][(rho,s)][x'] = sum_{n: |n-c|=x'} sum_{x''} k_rho[(c+x-x'') mod S] bx_s[x''-n+o],
              k_rho[j] = sum_kx v_rho[kx] cos(2 pi kx j / S)   (the real circular kernel of term rho along the row)

Stage 1 is the only pass over the S^2/4 samples; it keeps R = r*s numbers per column instead of NU.  No FFT anywhere.
Run:  python scripts/proto/rowmix.py [S] [N]
"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))


def hy_table(filtering):
    """Hy[q][kx] (real for a real, point-symmetric filter): weight of conv row q, column wavenumber kx (0..S/2) in
    out[c + x] = sum_kx Re( Z[kx] e^{2 pi i kx (c+x)/S} ),  Z[kx] = sum_q Hy[q][kx] rfft(conv[q])[kx]."""
    S = filtering.shape[0]
    c = S // 2
    Sh = S // 2 + 1
    Fs = 0.5 * (filtering + np.roll(filtering[::-1, ::-1], 1, axis=(0, 1)))   # (F[k] + F[-k]) / 2
    g = np.fft.ifft(Fs[:, :Sh], axis=0) * S                                   # g[n][kx] = sum_kr Fs e^{+2 pi i kr n/S}
    n = (c - np.arange(S)) % S
    w = np.full(Sh, 2.0)
    w[0] = 1.0
    if S % 2 == 0:
        w[-1] = 1.0
    Hy = g[n, :] * w[None, :] / S ** 2
    return Hy


def factor_tables(pb, tol=1e-8, beam_tol=1e-14):
    S = pb.d_mat.shape[0]
    c = S // 2
    Sh = S // 2 + 1
    nrow = S - c
    B = pb.beam_2d.shape[0]
    o = (B - 1) // 2
    umap = np.abs(np.arange(S) - c)
    NU = umap.max() + 1
    # beam: symmetric matrix -> separable terms
    bm = pb.beam_2d * pb.step ** 2
    ub, sb, vbt = np.linalg.svd(bm)
    ns = int(np.sum(sb > beam_tol * sb[0]))
    by = ub[:, :ns] * sb[:ns]          # [B][s]
    bx = vbt[:ns, :].T                 # [B][s]
    # transfer function: Hy = U v
    Hy = hy_table(pb.filtering)
    assert np.abs(Hy.imag).max() <= 1e-15 * np.abs(Hy.real).max()
    uu, sv, vt = np.linalg.svd(Hy.real, full_matrices=False)
    r = int(np.sum(sv > tol * sv[0]))
    U = (uu[:, :r] * sv[:r]).T          # [r][S]
    V = vt[:r]                          # [r][Sh]
    # stage 1 operator
    C = np.zeros((r, ns, NU))
    for s in range(ns):
        T = np.zeros((S, NU))           # T[q][u] = sum_{m: umap[m]=u, |q-m|<=o} by_s[q-m+o]
        for q in range(S):
            for m in range(max(0, q - o), min(S - 1, q + o) + 1):
                T[q, umap[m]] += by[q - m + o, s]
        C[:, s, :] = U @ T
    # stage 2 operator
    j = np.arange(S)
    kk = np.arange(Sh)
    cosm = np.cos(2 * np.pi * np.outer(kk, j) / S)                              # [Sh][S]
    krho = V @ cosm                                                             # [r][S]
    G = np.zeros((nrow, r, ns, NU))
    for s in range(ns):
        Fx = np.zeros((S, NU))          # Fx[x''][x'] = sum_{n: |n-c|=x', |x''-n|<=o} bx_s[x''-n+o]
        for xx in range(S):
            for n in range(max(0, xx - o), min(S - 1, xx + o) + 1):
                Fx[xx, umap[n]] += bx[xx - n + o, s]
        for x in range(nrow):
            kern = krho[:, (c + x - j) % S]                                     # [r][x'']
            G[x, :, s, :] = kern @ Fx
    return dict(C=C.reshape(r * ns, NU), G=G.reshape(nrow, r * ns * NU), r=r, ns=ns, NU=NU, sv=sv, sb=sb)


def main():
    from joxsz_amd import datasets
    from oracle import joxsz_oracle as orc
    from scipy.interpolate import interp1d
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    N = int(sys.argv[2]) if len(sys.argv) > 2 else max(S - S // 2, 100)
    pb = datasets.synthetic_problem(S=S, N=N, sz_only=True)
    p = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    pp = orc.press_fun(p, pb.r_pp)
    ref = orc.row_chain(pb, pp)
    c = S // 2
    NU = max(c, S - 1 - c) + 1
    iy = np.array([c + b if c + b < S else c - b for b in range(NU)])
    Q = ref['y_2d'][np.ix_(iy, iy)]                                              # quadrant of distinct samples [u][x']
    for tol in (1e-13, 1e-8):
        t = factor_tables(pb, tol=tol)
        D = t['C'] @ Q                                                          # [R][NU]
        out = t['G'] @ D.reshape(-1)
        err = np.abs(out - ref['map_row']).max() / np.abs(ref['map_row']).max()
        print(f"S={S} tol={tol:g}: beam terms {t['ns']} (sigma2/sigma1 {t['sb'][1] / t['sb'][0]:.1e}), TF rank {t['r']} of {S // 2 + 1}, "
              f"stage-1 rows R={t['r'] * t['ns']}, row error {err:.2e} of max")


if __name__ == '__main__':
    main()
