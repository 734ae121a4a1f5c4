import numpy as np
def idft8(x):
    j = 1j
    a0, a1, a2, a3 = x[0] + x[4], x[0] - x[4], x[2] + x[6], x[2] - x[6]
    a4, a5, a6, a7 = x[1] + x[5], x[1] - x[5], x[3] + x[7], x[3] - x[7]
    b0, b2, b1, b3 = a0 + a2, a0 - a2, a1 + j * a3, a1 - j * a3
    b4, b6, b5, b7 = a4 + a6, a4 - a6, a5 + j * a7, a5 - j * a7
    w1, w3 = np.exp(1j * np.pi / 4), np.exp(3j * np.pi / 4)
    return [b0 + b4, b1 + w1 * b5, b2 + j * b6, b3 + w3 * b7, b0 - b4, b1 - w1 * b5, b2 - j * b6, b3 - w3 * b7]
x = np.random.randn(8) + 1j * np.random.randn(8)
print("idft8", np.abs(np.array(idft8(x)) - np.fft.ifft(x) * 8).max())
def mixed(P, radices):
    M = P.size; A = P.copy().astype(complex)
    LS = M
    for s, R in enumerate(radices):
        per = LS // R
        for bf in range(M // R):
            sub, tt = divmod(bf, per)
            base = sub * LS + tt
            v = [A[base + per * q] for q in range(R)]
            y = np.fft.ifft(v) * R
            for k in range(R):
                A[base + per * k] = y[k] * np.exp(2j * np.pi * tt * k / LS)
        LS = per
    # natural index of position
    out = np.zeros(M, complex)
    for p in range(M):
        digs = []; rem = p; L = M
        for R in radices:
            L //= R
            digs.append(rem // L); rem %= L
        n = 0; mul = 1
        for d, R in zip(digs, radices):
            n += d * mul; mul *= R
        out[n] = A[p]
    return out
for M, rad in ((4000, (10, 10, 8, 5)), (5000, (10, 10, 10, 5)), (2500, (10, 10, 5, 5))):
    P = np.random.randn(M) + 1j * np.random.randn(M)
    print(M, np.abs(mixed(P, rad) - np.fft.ifft(P) * M).max())
