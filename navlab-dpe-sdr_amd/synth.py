"""Seeded synthetic inputs for the sampleblock->BCM path.

Product-side utility used by bench.py, smoke() and the tests; no oracle dependency.
The C/A generator here is an independent restatement (G2 delay-table form, as in
pygnss correlator.py:474-525) of the LFSR that csrc/ implements in the tap-selector form
(batchcorrscores.cu:117-177); tests cross-check the two.
"""
import numpy as np

F_CA = 1.023e6
F_L1 = 1.57542e9
L_CA = 1023
PRNS_R = [2, 3, 6, 12, 17, 19, 24, 28]          # handoff_params_usrp6.csv:5
PRNS_H = PRNS_R + [1, 5, 10, 25]                # SURVEY.md 8d, K=12

_G2_DELAY = [5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258, 469, 470, 471,
             472, 473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862, 863, 950, 947, 948, 950]


def ca_code(prn):
    """+/-1 C/A chips, chip = +1 where G1 xor G2 == 1 (correlator.py:515)."""
    g1 = np.zeros(1023, dtype=np.int64)
    g2 = np.zeros(1023, dtype=np.int64)
    r1 = [1] * 10
    r2 = [1] * 10
    for i in range(1023):
        g1[i] = r1[9]
        g2[i] = r2[9]
        f1 = r1[2] ^ r1[9]
        f2 = r2[1] ^ r2[2] ^ r2[5] ^ r2[7] ^ r2[8] ^ r2[9]
        r1 = [f1] + r1[:9]
        r2 = [f2] + r2[:9]
    g2 = np.roll(g2, _G2_DELAY[prn - 1])
    return np.where((g1 + g2) % 2 == 0, -1, 1).astype(np.int8)


def time_idx(S, fs):
    return np.round(np.arange(S) / fs * 1.0e9) / 1.0e9


def nav_bit_boundary(cp_ela, cp_ref, rc, fc, fs):
    since = (int(cp_ela) - int(cp_ref)) % 20
    return int(np.floor((L_CA * (20 - since) - rc) * (fs / fc))) + 1


def random_channels(seed, K, prns=None):
    """Channel parameters per SURVEY.md 8d: fi~U(-4e3,4e3), rc~U(0,1023), ri~U(0,1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    prns = list(prns) if prns is not None else (PRNS_R if K <= 8 else PRNS_H)[:K]
    fi = rng.uniform(-4e3, 4e3, K)
    fc = F_CA * (1.0 + fi / F_L1)
    rc = rng.uniform(0, L_CA, K)
    ri = rng.uniform(0, 1, K)
    cp = np.full(K, 1000, dtype=np.int32)
    cp_ref = (1000 + rng.integers(0, 20, K)).astype(np.int32)
    return dict(prn=np.array(prns, dtype=np.int32), rc=rc, ri=ri, fc=fc, fi=fi, cp=cp, cp_ref=cp_ref)


def gen_iq(seed, fs, S, ch, amp=48.0, sigma=300.0, dc=(3.0, -2.0), flip=None):
    """One window of interleaved int16 I/Q consistent with channel params `ch`.

    flip[k] = True puts a nav-bit sign change at the predicted boundary sample."""
    rng = np.random.Generator(np.random.PCG64(seed))
    K = len(ch["prn"])
    t = time_idx(S, fs)
    x = np.zeros(S, dtype=np.complex128)
    if flip is None:
        flip = rng.integers(0, 2, K).astype(bool)
    amp = np.broadcast_to(np.asarray(amp, dtype=np.float64), (K,))
    for k in range(K):
        chips = ca_code(int(ch["prn"][k])).astype(np.float64)
        r = chips[np.mod(np.floor(t * ch["fc"][k] + ch["rc"][k]).astype(np.int64), L_CA)]
        nb = nav_bit_boundary(ch["cp"][k], ch["cp_ref"][k], ch["rc"][k], ch["fc"][k], fs)
        d = np.ones(S)
        if flip[k] and 0 < nb < S:
            d[nb:] = -1.0
        x += amp[k] * d * r * np.exp(2j * np.pi * (ch["fi"][k] * t + ch["ri"][k]))
    x += sigma * (rng.standard_normal(S) + 1j * rng.standard_normal(S))
    x += complex(dc[0], dc[1])
    iq = np.empty(2 * S, dtype=np.int16)
    iq[0::2] = np.clip(np.rint(x.real), -32768, 32767).astype(np.int16)
    iq[1::2] = np.clip(np.rint(x.imag), -32768, 32767).astype(np.int16)
    return iq


def rand_grid(seed, G, half=(110.0, 110.0, 110.0, 132.0)):
    """rngrid3-format random ENU-dt grid (SURVEY.md 8d iii): columns x,y,z,delta_t (m)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = rng.uniform(-1.0, 1.0, (G, 4)) * np.asarray(half)[None, :]
    return g


def uniform_grid(dim, spacing):
    """Uniform tensor grid, x slowest / t fastest (batchcorrmanifold.cu:165-182)."""
    half = (dim - 1) // 2
    ax = spacing * (np.arange(dim) - half)
    X, Y, Z, T = np.meshgrid(ax, ax, ax, ax, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel(), T.ravel()], axis=1)


def spread_grid():
    """PyGNSS spread grids (receiver.py:995-1026): returns (pos[G,4], vel[G,4])."""
    a = np.array([-22, -19, -16, -13, -10, -7, -6, -5, -4, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7, 10, 13,
                  16, 19, 22], dtype=np.float64)
    b = np.arange(-12, 13, dtype=np.float64)
    X, Y, Z, T = np.meshgrid(a * 5, a * 5, a * 5, a * 6, indexing="ij")
    pos = np.stack([X.ravel(), Y.ravel(), Z.ravel(), T.ravel()], axis=1)
    X, Y, Z, T = np.meshgrid(b * 0.5, b * 0.5, b * 0.5, b * 0.25, indexing="ij")
    vel = np.stack([X.ravel(), Y.ravel(), Z.ravel(), T.ravel()], axis=1)
    return pos, vel


def write_grid_csv(path, grid):
    """x,y,z,delta_t per line with CRLF (batchcorrmanifold.cu:2433-2444 loader format)."""
    with open(path, "w", newline="") as f:
        for row in grid:
            f.write("%.17g,%.17g,%.17g,%.17g\r\n" % tuple(row))
