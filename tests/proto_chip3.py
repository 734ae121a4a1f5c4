"""numpy prototype of the chip3 formulation (SV-independent integer prefix moments + per-flip difference sums)
against the oracle's direct sums.  Development aid kept beside the tests because it uses oracle/ (not collected by pytest: run it directly)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import navlab_dpe_sdr_amd as dpe
from oracle import oracle as o

def chip3(iq, fs, prn, rc, ri, fc, fi, cp_ela, cp_ref, C, B, f32=True, order=2):
    S = iq.size // 2
    xr = iq[0::2].astype(np.int64); xi = iq[1::2].astype(np.int64)
    x = xr + 1j * xi
    chips = o.ca_code(prn).astype(np.float64)
    m = np.arange(S)
    chipidx = np.floor(m * (fc / fs) + rc).astype(np.int64)
    c0, cE = chipidx[0], chipidx[-1]
    # first index of each chip
    e = np.searchsorted(chipidx, np.arange(c0, cE + 2), side="left")   # e[j] = first m with chip >= c0+j ; last = S
    e[0] = 0
    nC = cE - c0 + 1
    a, b = e[:-1], e[1:]
    ln = b - a
    r = chips[np.mod(np.arange(c0, cE + 1), 1023)]
    idxNext = o.nav_bit_boundary(cp_ela, cp_ref, rc, fc, fs)
    hasFlip = 0 < idxNext < S
    side = (a >= idxNext) if hasFlip else np.zeros(nC, bool)
    mean = x.sum() / np.float64(np.float32(S))
    phi = 2 * np.pi * fi / fs
    # integer prefix sums (exact)
    C0 = np.concatenate([[0], np.cumsum(x)])
    W1 = np.concatenate([[0], np.cumsum(m * x)])
    W2 = np.concatenate([[0], np.cumsum(m * m * x)])
    c2 = a + b - 1          # doubled centre
    dC0 = C0[b] - C0[a]; dW1 = W1[b] - W1[a]; dW2 = W2[b] - W2[a]
    R0 = dC0
    R1 = (2 * dW1 - c2 * dC0) / 2.0
    R2 = (4 * dW2 - 4 * c2 * dW1 + c2 * c2 * dC0) / 4.0
    ft = np.float32 if f32 else np.float64
    ct = np.complex64 if f32 else np.complex128
    R0 = R0.astype(ct); R1 = R1.astype(ct); R2 = R2.astype(ct)
    lnf = ln.astype(ft)
    mean_c = ct(mean)
    phi_t = ft(phi)
    # raw (no mean) and mean-removed chip sums
    def taylor(R0, R1, R2):
        E0 = R0 - 1j * phi_t * R1
        E1 = R1.copy()
        if order >= 2:
            E0 = E0 - (phi_t * phi_t / 2) * R2
            E1 = E1 - 1j * phi_t * R2
        return E0.astype(ct), E1.astype(ct)
    T0, _ = taylor(R0, R1, R2)
    g2 = lnf * (lnf * lnf - 1) / 12
    P0, P1 = taylor(R0 - lnf * mean_c, R1, R2 - g2 * mean_c)
    cc = (a + b - 1) / 2.0
    ph = cc * (fi / fs) + ri
    ph -= np.floor(ph)
    wc = np.exp(-2j * np.pi * ph).astype(ct)
    # direct lag 0 per side
    tot = (wc * T0 * r.astype(ft)).astype(ct)
    corr0 = [tot[~side].sum(dtype=np.complex128), tot[side].sum(dtype=np.complex128)]
    # carrier bank: F[b] = sum_j r_j wc e^{-j th cc}(P0 - j th P1)   (fp64 reference combination of the per-chip terms)
    bins = np.arange(-B, B + 1)
    th = 2 * np.pi * bins / C
    carr = np.zeros((2, bins.size), complex)
    for s in (0, 1):
        sel = side == bool(s)
        if not sel.any(): continue
        z0 = (wc * P0 * r)[sel].astype(complex); z1 = (wc * P1 * r)[sel].astype(complex)
        for i, t in enumerate(th):
            carr[s, i] = np.sum(np.exp(-1j * t * cc[sel]) * (z0 - 1j * t * z1))
    # lag path: D'[l] = sum_boundaries c_j x[(e_j + l) mod S]
    lags = np.arange(-32, 31)
    Dp = np.zeros((2, lags.size), ct)
    # boundary list: interior chip starts a[1:], plus the circular one at 0
    rs = [np.where(~side, r, 0.0), np.where(side, r, 0.0)]
    for s in (0, 1):
        rr = rs[s]
        J = np.empty(nC); J[1:] = rr[:-1] - rr[1:]; J[0] = rr[-1] - rr[0]
        nz = np.nonzero(J)[0]
        pe = a[nz] * (fi / fs) + ri
        pe -= np.floor(pe)
        cj = (J[nz] * np.exp(-2j * np.pi * pe)).astype(ct)
        acc = np.zeros(lags.size, ct)
        xs = x.astype(ct)
        for jj, ej in enumerate(a[nz]):
            idx = ej + lags
            fix = np.where(idx < 0, np.exp(-1j * phi * S), np.where(idx >= S, np.exp(1j * phi * S), 1.0)).astype(ct)
            acc = (acc + cj[jj] * fix * xs[idx % S]).astype(ct)
        Dp[s] = acc
    D = Dp.astype(complex) * np.exp(-1j * phi * lags)[None, :]
    corr = np.zeros((2, 64), complex)
    for s in (0, 1):
        corr[s, 32] = corr0[s]
        for j in range(33, 64): corr[s, j] = corr[s, j - 1] + D[s, j - 1]
        for j in range(31, -1, -1): corr[s, j] = corr[s, j + 1] - D[s, j]
    return corr, carr, hasFlip, idxNext

if __name__ == "__main__":
    fs, S, K = 25e6, 125000, 4
    fi_scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    dc = (900.0, -700.0) if len(sys.argv) > 2 else (3.0, -2.0)
    ch = dpe.synth.random_channels(5, K)
    ch["fi"] = ch["fi"] * fi_scale
    ch["fc"] = 1.023e6 * (1 + ch["fi"] / 1.57542e9)
    iq = dpe.synth.gen_iq(1, fs, S, ch, amp=40.0, dc=dc)
    C = o.carr_fft_len(S); B = 16; L = 31
    for k in range(K):
        code, carr, inf = o.bcs_sv(iq, fs, int(ch["prn"][k]), ch["rc"][k], ch["ri"][k], ch["fc"][k], ch["fi"][k], int(ch["cp"][k]), int(ch["cp_ref"][k]), -32, 31, -B, B, C)
        for f32 in (False, True):
            for order in (1, 2):
                corr, cb, hf, idn = chip3(iq, fs, int(ch["prn"][k]), ch["rc"][k], ch["ri"][k], ch["fc"][k], ch["fi"][k], int(ch["cp"][k]), int(ch["cp_ref"][k]), C, B, f32, order)
                sg = 1.0 if inf["no_flip_larger"] else -1.0
                cg = corr[0] + sg * corr[1]
                fg = cb[0] + sg * cb[1]
                print("k", k, "fi %.0f" % ch["fi"][k], "f32", f32, "order", order, "code err %.3g" % (np.abs(cg - code).max() / np.abs(code).max()),
                      "carr err %.3g" % (np.abs(fg - carr).max() / np.abs(carr).max()), "flip", hf)
