"""numpy check of the index scheme of the wave-level 25 000-point inverse transform:
 decimation in time by 10 (q), each 2 500-point transform as 50 x 50 with register-resident 50-point DFTs (PFA 2 x 25, 25 = 5 x 5)."""
import numpy as np
rng = np.random.default_rng(1)
N, M = 25000, 2500
P = rng.standard_normal(N) + 1j * rng.standard_normal(N)
ref = np.fft.ifft(P) * N            # unnormalised inverse
w = lambda n, m: np.exp(2j * np.pi * n / m)

def idft5(x):  # x: list of 5 -> list of 5 (inverse, unnormalised)
    return [sum(x[n] * w(n * k, 5) for n in range(5)) for k in range(5)]

def idft25_inplace(v):
    """v[25] -> in place; position k1 + 5 k2 holds y[5 k1 + k2]"""
    for n1 in range(5):
        o = idft5([v[n1 + 5 * n2] for n2 in range(5)])
        for k2 in range(5): v[n1 + 5 * k2] = o[k2] * w(n1 * k2, 25)
    for k2 in range(5):
        o = idft5([v[n1 + 5 * k2] for n1 in range(5)])
        for k1 in range(5): v[k1 + 5 * k2] = o[k1]
    return v
def pos25(k):  # register position of output k of idft25_inplace
    return (k // 5) + 5 * (k % 5)

def idft50(x):
    """x[50] -> dict: output index -> value, via PFA 2 x 25"""
    S = [x[(2 * n2) % 50] + x[(25 + 2 * n2) % 50] for n2 in range(25)]
    D = [x[(2 * n2) % 50] - x[(25 + 2 * n2) % 50] for n2 in range(25)]
    idft25_inplace(S); idft25_inplace(D)
    y = [None] * 50
    for k2 in range(25):
        y[(26 * k2) % 50] = S[pos25(k2)]
        y[(25 + 26 * k2) % 50] = D[pos25(k2)]
    return y
x = rng.standard_normal(50) + 1j * rng.standard_normal(50)
print("idft50 err", np.abs(np.array(idft50(list(x))) - np.fft.ifft(x) * 50).max())

surf_ref = np.abs(ref.reshape(10, M)).sum(axis=0)      # sum_n |y[j + 2500 n]|
U = np.zeros((10, M), complex)
for q in range(10):
    Pq = P[q::10]                                      # P[10 k' + q]
    A = np.zeros((50, 50), complex)                    # A[a][c]
    for a in range(50):
        y = idft50([Pq[a + 50 * b] for b in range(50)])
        for c in range(50): A[a][c] = y[c] * w(a * c, 2500)
    for c in range(50):
        y = idft50([A[a][c] for a in range(50)])
        for d in range(50): U[q][c + 50 * d] = y[d]
    print("q", q, "err", np.abs(U[q] - np.fft.ifft(Pq) * M).max())
surf = np.zeros(M)
for j in range(M):
    v = [U[q][j] * w(q * j, N) for q in range(10)]
    yn = np.fft.ifft(v) * 10
    surf[j] = np.abs(yn).sum()
print("surface err", np.abs(surf - surf_ref).max() / surf_ref.max())
