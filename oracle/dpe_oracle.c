/*
 * dpe_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 * See dpe_oracle.h for scope, citation convention and the parity pin.
 * All arithmetic is fp64, as in the reference (cufftDoubleComplex / double everywhere).
 */
#include "dpe_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int posmod_i(int a, int b) { return ((a % b) + b) % b; } /* auxil.h:11 POSMOD */

/* ---------------------------------------------------------------- C/A code ------- */
/* batchcorrscores.cu:117-177.  Two 10-stage LFSRs in +/-1 notation (binary 1 == -1),
 * G2 output = product of two phase-selector taps (table :130-133), chip = -(g1*g2). */
void dpo_gen_ca_code(int prn, int8_t chips[1023])
{
    static const int8_t sel1[37] = { 2, 3, 4, 5, 1, 2, 1, 2, 3, 2, 3, 5, 6, 7, 8, 9, 1, 2,
                                     3, 4, 5, 6, 1, 4, 5, 6, 7, 8, 1, 2, 3, 4, 5, 4, 1, 2, 4 };
    static const int8_t sel2[37] = { 6, 7, 8, 9, 9, 10, 8, 9, 10, 3, 4, 6, 7, 8, 9, 10, 4,
                                     5, 6, 7, 8, 9, 3, 6, 7, 8, 9, 10, 6, 7, 8, 9, 10, 10, 7, 8, 10 };
    int a[10], b[10];
    for (int i = 0; i < 10; i++) { a[i] = -1; b[i] = -1; }
    const int s1 = sel1[prn - 1] - 1, s2 = sel2[prn - 1] - 1;
    for (int i = 0; i < 1023; i++) {
        int g1 = a[9];
        int g2 = b[s1] * b[s2];
        int fa = a[2] * a[9];
        int fb = b[1] * b[2] * b[5] * b[7] * b[8] * b[9];
        memmove(a + 1, a, 9 * sizeof(int));
        memmove(b + 1, b, 9 * sizeof(int));
        a[0] = fa;
        b[0] = fb;
        chips[i] = (int8_t)(-(g1 * g2));
    }
}

/* batchcorrscores.cu:247-253 */
int dpo_nav_bit_boundary(int cpElapsed, int cpReference, double codePhase,
                         double codeFreq, double fs)
{
    int since = posmod_i(cpElapsed - cpReference, 20);
    int toNext = 20 - since;
    return (int)(floor((DPO_L_CA * toNext - codePhase) * (fs / codeFreq)) + 1);
}

/* batchcorrscores.cu:191-193 */
double dpo_time_idx(int64_t i, double fs)
{
    double t = (double)i / fs;
    return round(t * 1.0e9) / 1.0e9;
}

/* ---------------------------------------------------------------- BCS ------------ */
int dpo_bcs_sv(const int16_t *iq, int S, double fs, const int8_t *chips,
               double rc, double ri, double fc, double fi, int cpElapsed, int cpReference,
               int lagLo, int lagHi, int64_t C, int binLo, int binHi,
               double *code, double *carr, int *info, double *meanOut)
{
    if (S <= 0 || lagHi < lagLo || binHi < binLo) return -1;
    double *bre = (double *)malloc(sizeof(double) * S * 6);
    if (!bre) return -1;
    double *bim = bre + S, *wre = bim + S, *wim = wre + S, *r = wim + S, *rf = r + S;

    /* BCS_Load :209-221 + thrust::reduce / ComplexDivide :1065-1066,1210-1216 */
    double sre = 0.0, sim = 0.0;
    for (int n = 0; n < S; n++) { sre += (double)iq[2 * n]; sim += (double)iq[2 * n + 1]; }
    const double mre = sre / (float)S, mim = sim / (float)S;
    if (meanOut) { meanOut[0] = mre; meanOut[1] = mim; }

    /* BCS_NavBitBoundary :237-258 */
    const int idxNext = dpo_nav_bit_boundary(cpElapsed, cpReference, rc, fc, fs);
    const int hasFlip = (idxNext > 0) && (idxNext < S);

    for (int n = 0; n < S; n++) {
        const double t = dpo_time_idx(n, fs);
        /* BCS_ComputeDopplerWipeoff :294-300: conj(exp(j 2pi (f t + phi))) */
        const double ph = 2 * DPO_PI * (fi * t + ri);
        wre[n] = cos(ph);
        wim[n] = -sin(ph);
        /* BCS_BatchMultiply :402 rawWiped = raw * wipe */
        const double xr = (double)iq[2 * n], xi = (double)iq[2 * n + 1];
        bre[n] = xr * wre[n] - xi * wim[n];
        bim[n] = xr * wim[n] + xi * wre[n];
        /* BCS_ComputeCodeReplica :347-367 */
        const int chip = posmod_i((int)floor(t * fc + rc), DPO_L_CA);
        r[n] = (double)chips[chip];
        rf[n] = hasFlip ? ((n >= idxNext) ? -r[n] : r[n]) : 0.0;
    }

    /* circular correlation == ifft(conj(fft(rep)) .* fft(rawWiped)) with BCS_NormalizeFFT
     * (:1099-1144): corr[l] = sum_n b[n] * rep[(n-l) mod S]  (rep real). */
    const int nLag = lagHi - lagLo + 1;
    double *tmp = (double *)malloc(sizeof(double) * 4 * (nLag + 1));
    if (!tmp) { free(bre); return -1; }
    for (int li = 0; li <= nLag; li++) {
        const int lag = (li < nLag) ? (lagLo + li) : 0; /* extra slot: lag 0 for the choice */
        int m = posmod_i(-lag, S);
        double a0 = 0, a1 = 0, f0 = 0, f1 = 0;
        for (int n = 0; n < S; n++) {
            a0 += bre[n] * r[m];  a1 += bim[n] * r[m];
            f0 += bre[n] * rf[m]; f1 += bim[n] * rf[m];
            if (++m == S) m = 0;
        }
        tmp[4 * li] = a0; tmp[4 * li + 1] = a1; tmp[4 * li + 2] = f0; tmp[4 * li + 3] = f1;
    }
    /* BCS_ChooseCodeCorr :512-516 (decision at lag 0 only) */
    const double *z = tmp + 4 * nLag;
    const int noFlipLarger = (!hasFlip) || (hypot(z[0], z[1]) > hypot(z[2], z[3]));
    for (int li = 0; li < nLag; li++) {
        code[2 * li]     = noFlipLarger ? tmp[4 * li]     : tmp[4 * li + 2];
        code[2 * li + 1] = noFlipLarger ? tmp[4 * li + 1] : tmp[4 * li + 3];
    }
    free(tmp);
    if (info) { info[0] = idxNext; info[1] = noFlipLarger; }

    /* BCS_SubtractDCOffset :480 + BCS_ChoosyBatchMultiplyAndPad :440-448, then the
     * zero-padded length-C forward DFT :1179 and fftshift :1180 (bin b at C/2+b). */
    const double *rch = noFlipLarger ? r : rf;
    for (int n = 0; n < S; n++) {
        const double dr = mre * wre[n] - mim * wim[n];
        const double di = mre * wim[n] + mim * wre[n];
        bre[n] = (bre[n] - dr) * rch[n];
        bim[n] = (bim[n] - di) * rch[n];
    }
    const double twoPi = 6.283185307179586476925286766559; /* FFT twiddles use true pi */
    for (int b = binLo; b <= binHi; b++) {
        const int64_t bb = ((b % C) + C) % C;
        double ar = 0, ai = 0, tr = 1, ti = 0;
        const double sr = cos(twoPi * (double)bb / (double)C), si = -sin(twoPi * (double)bb / (double)C);
        for (int n = 0; n < S; n++) {
            if ((n & 63) == 0) { /* exact resync of the rotation recurrence */
                const int64_t mm = ((int64_t)n * bb) % C;
                const double ang = twoPi * (double)mm / (double)C;
                tr = cos(ang); ti = -sin(ang);
            }
            ar += bre[n] * tr - bim[n] * ti;
            ai += bre[n] * ti + bim[n] * tr;
            const double nr = tr * sr - ti * si;
            ti = tr * si + ti * sr;
            tr = nr;
        }
        carr[2 * (b - binLo)] = ar;
        carr[2 * (b - binLo) + 1] = ai;
    }
    free(bre);
    return 0;
}

/* ---------------------------------------------------------------- BCM ------------ */
/* Index of one (point, SV) pair in extended precision: same formula as :1779-1797, but carried
 * in long double so that the reference's own fp64 cancellation (rxTime ~4e5 s minus pr/C rounds
 * at ulp = 5.8e-11 s = 17 mm = 1.4e-4 samples) is removed.  Used only to MEASURE that noise. */
static long double pos_base_ld(const double *s, const double *g, const double *c, const double *R,
                               double rxTime, int tow, int dcp, double rcEnd, double fs, double fc,
                               int numSamps)
{
    const long double px = (long double)R[0] * g[0] + (long double)R[1] * g[1] + (long double)R[2] * g[2] + c[0];
    const long double py = (long double)R[3] * g[0] + (long double)R[4] * g[1] + (long double)R[5] * g[2] + c[1];
    const long double pz = (long double)R[6] * g[0] + (long double)R[7] * g[1] + (long double)R[8] * g[2] + c[2];
    const long double pdt = (long double)g[3] + c[3];
    const long double lx = s[0] - px, ly = s[1] - py, lz = s[2] - pz;
    const long double range = sqrtl(lx * lx + ly * ly + lz * lz);
    const long double pr = range - (long double)DPO_C * s[3] + pdt;
    const long double txT = (long double)rxTime - pr / (long double)DPO_C;
    const long double cfd = txT - tow - ((long double)dcp * (long double)DPO_T_CA);
    const long double rc0 = cfd * (long double)DPO_F_CA - rcEnd;
    return ((long double)fs / fc) * (-rc0) + numSamps / 2.0L;
}

/* Grid points of the last dpo_bcm_pos call at which the reference's two floors (:1798-1799) were TWO apart: an index
 * within one rounding step below an integer has floor(idx) = n-1 but idx+1 rounds up to n+1, so the weights
 * (idx - fidx, cidx - idx) are both ~1 and the pair contributes c[n+1] + c[n-1] instead of ~c[n].  Only met where the
 * predicted index is an integer by construction (a grid point with zero offset and self-consistent channel
 * parameters); which side of the integer the fp64 rounding falls decides the reference's value there. */
#define DPO_MAX_QUIRKS 256
static int64_t g_quirk_idx[DPO_MAX_QUIRKS];
static int64_t g_quirk_n = 0;

int64_t dpo_bcm_pos_quirks(int64_t *idx, int64_t max)
{
    const int64_t n = g_quirk_n < DPO_MAX_QUIRKS ? g_quirk_n : DPO_MAX_QUIRKS;
    for (int64_t i = 0; i < n && i < max; i++) idx[i] = g_quirk_idx[i];
    return g_quirk_n;
}

int dpo_bcm_pos(const double *sat, const double *codeWin, int winLo, int winLen,
                const double *c, const double *grid, int64_t G, const double *R,
                const double *codeFreq, const int *cpRefTOW, const int *cpElapsedEnd,
                const int *cpRef, const double *codePhase, double rxTime, int K, double fs,
                int numSamps, int LPower, double *scores, int64_t *oob)
{
    /* LPower < 0 selects the extended-precision index (|LPower| is the exponent) */
    const int extended = LPower < 0;
    if (extended) LPower = -LPower;
    int64_t nOob = 0;
    g_quirk_n = 0;
    for (int64_t i = 0; i < G; i++) {
        const double *g = grid + 4 * i;
        /* :1760-1763 */
        const double px = R[0] * g[0] + R[1] * g[1] + R[2] * g[2] + c[0];
        const double py = R[3] * g[0] + R[4] * g[1] + R[5] * g[2] + c[1];
        const double pz = R[6] * g[0] + R[7] * g[1] + R[8] * g[2] + c[2];
        const double pdt = g[3] + c[3];
        double score = 0.0;
        for (int k = 0; k < K; k++) {
            const double *s = sat + 8 * k; /* caller passes the mid-time state, :1775 */
            const double lx = s[0] - px, ly = s[1] - py, lz = s[2] - pz;   /* :1779-1781 */
            const double range = sqrt(lx * lx + ly * ly + lz * lz);           /* :1782 */
            const double pr = range - DPO_C * s[3] + pdt;                     /* :1783 */
            const double txT = rxTime - pr / DPO_C;                           /* :1784 */
            const double cfd = txT - cpRefTOW[k] - ((cpElapsedEnd[k] - cpRef[k]) * DPO_T_CA);
            const double rcbc = cfd * DPO_F_CA;                               /* :1786 */
            const double rc0 = rcbc - codePhase[k];                           /* :1790 */
            double base = (fs / codeFreq[k]) * (-rc0) + numSamps / 2.0;       /* :1791 */
            long double baseLd = base;
            if (extended) {
                baseLd = pos_base_ld(s, g, c, R, rxTime, cpRefTOW[k], cpElapsedEnd[k] - cpRef[k],
                                     codePhase[k], fs, codeFreq[k], numSamps);
                base = (double)baseLd;
            }
            if (!(base < numSamps && base > 0)) { nOob++; continue; }         /* :1795 */
            const double idx = base + ((double)numSamps * k);                 /* :1797 */
            const double fi_ = floor(idx), ci_ = floor(idx + 1);              /* :1798-1799 */
            int64_t fin = (int64_t)fi_ - (int64_t)numSamps * k - winLo;
            int64_t cin = (int64_t)ci_ - (int64_t)numSamps * k - winLo;
            if (extended) {   /* neighbours and weight from the SAME long-double index (an index within a double rounding
                               * step of an integer would otherwise pair floor(double) with frac(long double)) */
                fin = (int64_t)floorl(baseLd) - winLo;
                cin = fin + 1;
            }
            if (!extended && cin - fin != 1 && (g_quirk_n == 0 || g_quirk_idx[(g_quirk_n - 1) % DPO_MAX_QUIRKS] != i)) {
                if (g_quirk_n < DPO_MAX_QUIRKS) g_quirk_idx[g_quirk_n] = i;
                g_quirk_n++;
            }
            if (fin < 0 || cin < 0 || fin >= winLen || cin >= winLen) { nOob++; continue; }
            const double *row = codeWin + 2 * (int64_t)winLen * k;
            double wc = idx - fi_, wf = ci_ - idx;                            /* :1810-1811 */
            if (extended) { wc = (double)(baseLd - floorl(baseLd)); wf = 1.0 - wc; }
            const double vr = row[2 * cin] * wc + row[2 * fin] * wf;
            const double vi = row[2 * cin + 1] * wc + row[2 * fin + 1] * wf;
            score += pow(hypot(vr, vi), (double)LPower);                      /* :1816 */
        }
        scores[i] = score;
    }
    if (oob) *oob = nOob;
    return 0;
}

int dpo_bcm_vel(const double *sat, const double *carrWin, int64_t winLo, int winLen,
                const double *c, const double *grid, int64_t G, const double *R,
                const double *carrFreq, double rxTime, int K, double fs, int64_t numfftPts,
                int dopplerSign, int LPower, double *scores, int64_t *oob)
{
    (void)rxTime;
    int64_t nOob = 0;
    for (int64_t i = 0; i < G; i++) {
        const double *g = grid + 4 * i;
        /* :1903-1906 */
        const double vx = R[0] * g[0] + R[1] * g[1] + R[2] * g[2] + c[4];
        const double vy = R[3] * g[0] + R[4] * g[1] + R[5] * g[2] + c[5];
        const double vz = R[6] * g[0] + R[7] * g[1] + R[8] * g[2] + c[6];
        const double vdt = g[3] + c[7];
        double score = 0.0;
        for (int k = 0; k < K; k++) {
            const double *s = sat + 8 * k;
            /* :1917-1920 */
            const double ex = vx - DPO_OEDOT * c[1], ey = vy + DPO_OEDOT * c[0], ez = vz;
            /* :1923-1930 */
            const double lx = s[0] - c[0], ly = s[1] - c[1], lz = s[2] - c[2];
            const double range = sqrt(lx * lx + ly * ly + lz * lz);
            const double lrr = ((lx / range) * (ex - s[4])) + ((ly / range) * (ey - s[5])) +
                               ((lz / range) * (ez - s[6]));
            const double fbc = DPO_F_L1 * ((lrr - vdt) / DPO_C + s[7]) / dopplerSign;
            const double f0 = fbc - carrFreq[k];                               /* :1933 */
            const double base = ((double)numfftPts / fs) * f0 + numfftPts / 2.0; /* :1936 */
            if (!(base < numfftPts && base > 0)) { nOob++; continue; }
            const double idx = base + ((double)numfftPts * k);
            const double fi_ = floor(idx), ci_ = floor(idx + 1);
            const int64_t fin = (int64_t)fi_ - numfftPts * k - winLo;
            const int64_t cin = (int64_t)ci_ - numfftPts * k - winLo;
            if (fin < 0 || cin < 0 || fin >= winLen || cin >= winLen) { nOob++; continue; }
            const double *row = carrWin + 2 * (int64_t)winLen * k;
            const double wc = idx - fi_, wf = ci_ - idx;                       /* :1952-1953 */
            const double vr = row[2 * cin] * wc + row[2 * fin] * wf;
            const double vi = row[2 * cin + 1] * wc + row[2 * fin + 1] * wf;
            score += pow(hypot(vr, vi), (double)LPower);                       /* :1954 */
        }
        scores[i] = score;
    }
    if (oob) *oob = nOob;
    return 0;
}

int64_t dpo_argmax_first(const double *v, int64_t n)
{
    int64_t best = 0;
    for (int64_t i = 1; i < n; i++) if (v[i] > v[best]) best = i;
    return best;
}

void dpo_make_meas(int64_t pi, int64_t vi, const double *c, const double *pg,
                   const double *vg, const double *R, double z[8], double Rv[64])
{
    const double *p = pg + 4 * pi, *v = vg + 4 * vi;
    /* :1990-1999 */
    z[0] = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + c[0];
    z[1] = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + c[1];
    z[2] = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + c[2];
    z[3] = p[3] + c[3];
    /* :2042-2051 */
    z[4] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2] + c[4];
    z[5] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2] + c[5];
    z[6] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2] + c[6];
    z[7] = v[3] + c[7];
    /* :2003-2011 and :2055-2063 -> 8x8 identity */
    for (int i = 0; i < 64; i++) Rv[i] = 0.0;
    for (int i = 0; i < 8; i++) Rv[9 * i] = 1.0;
}

/* batchcorrmanifold.cu:163-246 */
static double arthur_axis(int idx, int dim, int half, double sp)
{
    if (idx < half / 2 || (dim - idx) < half / 2) {
        if (idx < half) return 3 * sp * (idx - half) + sp * ((half / 2) + 1) * 2;
        return 3 * sp * (idx - half) - sp * ((half / 2) + 1) * 2;
    }
    return sp * (idx - half);
}

void dpo_init_grid(int gridType, const int dims[4], const double spacing[4],
                   double *grid, double *timeGrid)
{
    int half[4];
    for (int d = 0; d < 4; d++) half[d] = (dims[d] - 1) / 2; /* :2331 */
    const int64_t G = (int64_t)dims[0] * dims[1] * dims[2] * dims[3];
    for (int64_t i = 0; i < G; i++) {
        int ix[4];
        int64_t t = i;
        ix[3] = (int)(t % dims[3]); t /= dims[3];
        ix[2] = (int)(t % dims[2]); t /= dims[2];
        ix[1] = (int)(t % dims[1]); ix[0] = (int)(t / dims[1]);
        for (int d = 0; d < 4; d++) {
            grid[4 * i + d] = (gridType == 2) ? arthur_axis(ix[d], dims[d], half[d], spacing[d])
                                              : spacing[d] * (ix[d] - half[d]);
        }
        if (timeGrid && i == ix[3]) timeGrid[i] = grid[4 * i + 3];
    }
}

/* ---------------------------------------------------------------- chanmgr -------- */
static double week_crossover(double t) /* cuchanmgr.cu:26-31 */
{
    if (t > 302400.0) return t - 604800.0;
    if (t < -302400.0) return t + 604800.0;
    return t;
}

static double kepler(double M0, double n, double tk, double e, int *ok) /* :98-107 */
{
    double E, M, dE = 1;
    E = M = fmod(M0 + n * tk, DPO_2PI);
    for (int it = 0; it < 10 && fabs(dE) > 1e-12; it++) {
        const double f = M - E + e * sin(E);
        const double dfdE = -1.0 + e * cos(E);
        dE = -f / dfdE;
        E = fmod(E + dE, DPO_2PI);
    }
    *ok = !(fabs(dE) > 1e-12);
    return E;
}

int dpo_sat_pos(const double eph[DPO_EPH_N], double txTime, double st[8])
{
    const double sqA = eph[DPO_EPH_SQRT_A], A = sqA * sqA, e = eph[DPO_EPH_E];
    const double n = sqrt(DPO_MU / (A * A * A)) + eph[DPO_EPH_DELN];            /* :89 */
    double tc = week_crossover(txTime - eph[DPO_EPH_TOCS]);                      /* :92 */
    double clkb = eph[DPO_EPH_F2] * tc * tc + eph[DPO_EPH_F1] * tc + eph[DPO_EPH_F0] - eph[DPO_EPH_TGD];
    double tk = week_crossover(txTime - clkb - eph[DPO_EPH_TOES]);               /* :94 */
    int ok;
    double E = kepler(eph[DPO_EPH_M0], n, tk, e, &ok);
    if (!ok) return -1;
    const double dtr = DPO_F * e * sqA * sin(E);                                  /* :110 */
    tc = txTime - (clkb + dtr) - eph[DPO_EPH_TOCS];
    clkb = eph[DPO_EPH_F2] * tc * tc + eph[DPO_EPH_F1] * tc + eph[DPO_EPH_F0] + dtr - eph[DPO_EPH_TGD];
    const double clkd = eph[DPO_EPH_F1] + 2.0 * eph[DPO_EPH_F2] * tc;
    tk = week_crossover(txTime - clkb - eph[DPO_EPH_TOES]);                      /* :117 */
    E = kepler(eph[DPO_EPH_M0], n, tk, e, &ok);
    if (!ok) return -1;
    const double sinE = sin(E), cosE = cos(E);
    const double v = atan2(sqrt(1.0 - e * e) * sinE / (1.0 - e * cosE), (cosE - e) / (1.0 - e * cosE));
    double u = fmod(v + eph[DPO_EPH_OMG], DPO_2PI);
    double cos2u = cos(2.0 * u), sin2u = sin(2.0 * u);
    u += eph[DPO_EPH_CUC] * cos2u + eph[DPO_EPH_CUS] * sin2u;
    const double r = A * (1.0 - e * cosE) + eph[DPO_EPH_CRC] * cos2u + eph[DPO_EPH_CRS] * sin2u;
    const double inc = eph[DPO_EPH_I0] + eph[DPO_EPH_IDOT] * tk + eph[DPO_EPH_CIC] * cos2u + eph[DPO_EPH_CIS] * sin2u;
    const double omegak = fmod(eph[DPO_EPH_OMG0] + (eph[DPO_EPH_OMGD] - DPO_OEDOT) * tk - DPO_OEDOT * eph[DPO_EPH_TOES], DPO_2PI);
    const double xop = r * cos(u), yop = r * sin(u);
    const double co = cos(omegak), so = sin(omegak), ci = cos(inc), si = sin(inc);
    st[0] = xop * co - yop * so * ci;
    st[1] = xop * so + yop * co * ci;
    st[2] = yop * si;
    st[3] = clkb;
    cos2u = cos(2.0 * u); sin2u = sin(2.0 * u);                                   /* :180-181 */
    const double edot = n / (1.0 - e * cosE);
    const double vdot = sinE * edot * (1.0 + e * cos(v)) / (sin(v) * (1.0 - e * cosE));
    const double udot = vdot + 2.0 * (eph[DPO_EPH_CUS] * cos2u - eph[DPO_EPH_CUC] * sin2u) * vdot;
    const double rdot = A * e * sinE * edot + 2.0 * (eph[DPO_EPH_CRS] * cos2u - eph[DPO_EPH_CRC] * sin2u) * vdot;
    const double idd = eph[DPO_EPH_IDOT] + (eph[DPO_EPH_CIS] * cos2u - eph[DPO_EPH_CIC] * sin2u) * 2 * vdot;
    const double vxop = rdot * cos(u) - yop * udot, vyop = rdot * sin(u) + xop * udot;
    const double omd = eph[DPO_EPH_OMGD] - DPO_OEDOT;
    const double ta = vxop - yop * ci * omd, tb = xop * omd + vyop * ci - yop * si * idd;
    st[4] = ta * co - tb * so;
    st[5] = ta * so + tb * co;
    st[6] = vyop * si + yop * ci * idd;
    st[7] = clkd;
    return 0;
}

void dpo_ecef2ll(const double p[3], double ll[2]) /* cuchanmgr.cu:37-50 */
{
    const double pp = sqrt(p[0] * p[0] + p[1] * p[1]);
    const double th = atan2(p[2] * DPO_WGS84_A, pp * DPO_WGS84_B);
    ll[0] = atan2(p[2] + pow(DPO_WGS84_EP, 2) * DPO_WGS84_B * pow(sin(th), 3),
                  pp - pow(DPO_WGS84_E, 2) * DPO_WGS84_A * pow(cos(th), 3));
    ll[1] = atan2(p[1], p[0]);
}

void dpo_enu2ecef(const double ll[2], double R[9]) /* cuchanmgr.cu:54-73 */
{
    const double sa = sin(ll[0]), so = sin(ll[1]), ca = cos(ll[0]), co = cos(ll[1]);
    R[0] = -so; R[1] = -sa * co; R[2] = ca * co;
    R[3] = co;  R[4] = -sa * so; R[5] = ca * so;
    R[6] = 0.0; R[7] = ca;       R[8] = sa;
}

static double tx_time(int tow, double cpEla, int cpRef, double rc) /* :258-260 */
{
    return tow + ((cpEla - cpRef) * DPO_T_CA) + (rc / DPO_F_CA);
}

void dpo_chm_compute_sat_states(dpo_chan_t *ch)
{
    for (int i = 0; i < ch->K; i++) {
        ch->txTime[i] = tx_time(ch->cpRefTOW[i], ch->cpElaEnd[i], ch->cpRef[i], ch->rcEnd[i]);
        dpo_sat_pos(ch->eph + DPO_EPH_N * i, ch->txTime[i], ch->satStates + 8 * i);
    }
}

static double posfmod(double a, double b) { double t = fmod(a, b); return (t < 0.0) ? t + b : t; }

/* shared tail of :675-823 / :451-602 */
static void time_update_one(dpo_chan_t *ch, int i, const double *c, double rxTime, double T)
{
    const double *eph = ch->eph + DPO_EPH_N * i;
    const double adv = ch->fc[i] * T + ch->rcEnd[i];
    const double cpPred = ch->cpElaEnd[i] + floor(adv / DPO_L_CA);               /* :679-683 */
    const double rcPred = posfmod(adv, (double)DPO_L_CA);                          /* :685-689 */
    const double txPred = tx_time(ch->cpRefTOW[i], cpPred, ch->cpRef[i], rcPred);  /* :695-697 */
    double sp[8];
    dpo_sat_pos(eph, txPred, sp);                                                  /* :737 */
    const double tau = rxTime + T - (txPred + (c[3] / DPO_C)) + sp[3];            /* :742 */
    const double ct = cos(-DPO_OEDOT * tau), st = sin(-DPO_OEDOT * tau);
    const double sx = ct * sp[0] - st * sp[1], sy = st * sp[0] + ct * sp[1], sz = sp[2];
    const double lx = sx - c[0], ly = sy - c[1], lz = sz - c[2];
    const double range = sqrt(lx * lx + ly * ly + lz * lz);
    const double pr = range - DPO_C * sp[3] + c[3];                                /* :763 */
    const double bctx = rxTime + T - pr / DPO_C;                                   /* :765 */
    const double cfd = bctx - ch->cpRefTOW[i] - ((ch->cpElaEnd[i] - ch->cpRef[i]) * DPO_T_CA);
    const double bcrc = cfd * DPO_F_CA;                                            /* :774 */
    ch->cpElaStart[i] = ch->cpElaEnd[i];                                           /* :789 */
    const double t1 = floor(bcrc / DPO_L_CA);
    ch->rcStart[i] = ch->rcEnd[i];
    ch->cpElaEnd[i] += t1;                                                         /* :799 */
    ch->rcEnd[i] = posfmod(bcrc, (double)DPO_L_CA);
    ch->riStart[i] = ch->riEnd[i];
    ch->riEnd[i] = posfmod(ch->fi[i] * T + ch->riEnd[i], 1.0);                     /* :803-807 */
    ch->txTime[i] = tx_time(ch->cpRefTOW[i], ch->cpElaEnd[i], ch->cpRef[i], ch->rcEnd[i]);
    dpo_sat_pos(eph, ch->txTime[i], ch->satStates + 8 * i);                        /* :823 */
}

void dpo_chm_time_update(dpo_chan_t *ch, const double *c, double rxTime, double T)
{
    for (int i = 0; i < ch->K; i++) time_update_one(ch, i, c, rxTime, T);
}

void dpo_chm_propagate(dpo_chan_t *ch, const double *c, double rxTime, double T)
{
    for (int i = 0; i < ch->K; i++) {
        const double *s = ch->satStates + 8 * i;
        /* measurement update :380-447 */
        const double tau = rxTime - (ch->txTime[i] + (c[3] / DPO_C)) + s[3];
        const double ct = cos(-DPO_OEDOT * tau), st = sin(-DPO_OEDOT * tau);
        const double sx = ct * s[0] - st * s[1], sy = st * s[0] + ct * s[1], sz = s[2];
        const double sxd = ct * s[4] - st * s[5] - DPO_OEDOT * st * s[0] - DPO_OEDOT * ct * s[1];
        const double syd = st * s[4] + ct * s[5] + DPO_OEDOT * ct * s[0] - DPO_OEDOT * st * s[1];
        const double szd = s[6];
        const double ex = c[4] - DPO_OEDOT * c[1], ey = c[5] + DPO_OEDOT * c[0], ez = c[6];
        const double lx = sx - c[0], ly = sy - c[1], lz = sz - c[2];
        const double range = sqrt(lx * lx + ly * ly + lz * lz);
        const double lrr = ((lx / range) * (ex - sxd)) + ((ly / range) * (ey - syd)) + ((lz / range) * (ez - szd));
        const double bcfi = DPO_F_L1 * ((lrr - c[7]) / DPO_C + s[7]) / ch->dopplerSign;
        const double pr = range - DPO_C * s[3] + c[3];
        const double bctx = rxTime - pr / DPO_C;
        const double cfd = bctx - ch->cpRefTOW[i] - ((ch->cpElaEnd[i] - ch->cpRef[i]) * DPO_T_CA);
        const double bcrc = cfd * DPO_F_CA;
        const double bcfc = DPO_F_CA + (ch->dopplerSign * DPO_F_CA / DPO_F_L1) * bcfi + (bcrc - ch->rcEnd[i]) / T;
        ch->fi[i] = bcfi;
        ch->fc[i] = bcfc;
        time_update_one(ch, i, c, rxTime, T);
    }
}

void dpo_chm_grid_prep(double rxTime, const double *txTime, const double *c,
                       const double *sat, int K, const double *timeGrid, int dimT,
                       double *batch, double *R)
{
    for (int k = 0; k < K; k++) {
        const double *s = sat + 8 * k;
        for (int t = 0; t < dimT; t++) {
            /* :892-916 */
            const double tau = rxTime - (txTime[k] + ((timeGrid[t] + c[3]) / DPO_C)) + s[3];
            const double ct = cos(-DPO_OEDOT * tau), st = sin(-DPO_OEDOT * tau);
            double *o = batch + 8 * ((int64_t)k * dimT + t);
            o[0] = ct * s[0] - st * s[1];
            o[1] = st * s[0] + ct * s[1];
            o[2] = s[2];
            o[3] = s[3];
            o[4] = ct * s[4] - st * s[5] - DPO_OEDOT * st * s[0] - DPO_OEDOT * ct * s[1];
            o[5] = st * s[4] + ct * s[5] + DPO_OEDOT * ct * s[0] - DPO_OEDOT * st * s[1];
            o[6] = s[6];
            o[7] = s[7];
        }
    }
    double ll[2];
    dpo_ecef2ll(c, ll);       /* :881-883 */
    dpo_enu2ecef(ll, R);
}
