"""numpy prototype of the quarter-length real-even transform used by jx_rowdct_kernel.

R(k) = sum_{n=-amax}^{amax} q[|n|] cos(2 pi k n / P),  k = 0..P/2,  through ONE complex FFT of length Q = P/4.
"""
import numpy as np


def dct_direct(q, P, nk):
    amax = len(q) - 1
    n = np.arange(-amax, amax + 1)
    k = np.arange(nk)[:, None]
    return (q[np.abs(n)][None, :] * np.cos(2 * np.pi * k * n[None, :] / P)).sum(1)


def dct_quarter(q, P):
    LP, Q = P // 2, P // 4
    amax = len(q) - 1
    assert amax < LP - 1

    def xq(n):
        n = abs(n)
        return q[n] if n <= amax else 0.0

    z = np.zeros(Q, complex)
    for g in range(Q // 2 + 1):                       # groups of four samples
        if 2 * g + 1 <= Q:                            # first kind: j = g <= (Q-1)//2
            if g <= (Q - 1) // 2:
                z[g] = (xq(4 * g) + xq(4 * g + 1) - xq(4 * g - 1)) + 1j * (xq(4 * g + 2) + xq(4 * g + 3) - xq(4 * g + 1))
        if g >= 1 and Q - g > (Q - 1) // 2:           # second kind: j = Q - g
            z[Q - g] = (xq(4 * g) - xq(4 * g + 1) + xq(4 * g - 1)) + 1j * (xq(4 * g - 2) - xq(4 * g - 1) + xq(4 * g - 3))
    Z = np.fft.fft(z)
    Zx = np.append(Z, Z[0])
    k = np.arange(Q + 1)
    Zc = np.conj(Zx[Q - k])
    w = np.exp(-2j * np.pi * k / LP)
    Y = 0.5 * (Zx + Zc) - 0.5j * w * (Zx - Zc)
    A = Y.real
    B = np.zeros(Q + 1)
    B[1:Q] = Y.imag[1:Q] / (2 * np.sin(2 * np.pi * k[1:Q] / P))
    B[0] = 2 * q[1::2].sum()
    X = np.zeros(LP + 1)
    X[:Q + 1] = A + B
    X[LP - k] = A - B
    X[Q] = A[Q]
    return X


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    for P, amax in ((576, 255), (576, 256), (288, 127), (288, 85), (1152, 511), (36, 15), (96, 31), (96, 24), (192, 63), (36, 16)):
        a = np.arange(amax + 1)
        q = np.exp(-(a / (0.3 * amax)) ** 2) + 0.01 * rng.standard_normal(amax + 1)
        want = dct_direct(q, P, P // 2 + 1)
        got = dct_quarter(q, P)
        print(P, amax, np.abs(got - want).max() / np.abs(want).max())
