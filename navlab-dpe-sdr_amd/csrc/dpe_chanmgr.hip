// dpe_chanmgr.hip -- host-side (fp64) channel manager feeding the hot path.
//
// Restates dsp::cuChanMgr (cudarecv/modules/src/cuchanmgr.cu:85-210,240-306,338-608,641-829,
// 853-923,1004-1268).  In the reference these are <<<1,64>>> kernels over K <= 37 channels whose
// outputs live in device arrays; here they run on the host (a few microseconds of fp64 per window)
// and directly fill the structs that dpe_bcs_update / dpe_bcm_update take.
#include "dpe_common.h"

namespace dpe {

struct Eph {  // subset of eph_t (cudarecv/utils/inc/ephhelper.h:98-125) used by CHM_Get_Sat_Pos
    double sqrtA, e, i0, OMG0, omg, M0, deln, OMGd, idot, crc, crs, cuc, cus, cic, cis, toes, tocs, f0, f1, f2, tgd;
};

constexpr double kMu = 3.9860050e14;       // ephhelper.h MU_GPS
constexpr double kRelF = -4.442807633e-10; // consthelper.h CONST_F
constexpr double k2Pi = 6.2831853071796;   // consthelper.h CONST_2PI
constexpr double kWgsA = 6378137.0, kWgsB = 6356752.314245, kWgsE = 0.08181919084262149, kWgsEp = 0.08209443794969568;

static double half_week(double t)  // CHM_Correct_Week_Crossover :26-31
{
    return t > 302400.0 ? t - 604800.0 : (t < -302400.0 ? t + 604800.0 : t);
}

static bool solve_kepler(double M, double e, double &E)  // :97-107
{
    E = M;
    double dE = 1.0;
    for (int it = 0; it < 10 && std::fabs(dE) > 1e-12; ++it) {
        double sE, cE;
        sincos(E, &sE, &cE);   // glibc: bit-identical to sin() and cos(), one argument reduction
        dE = (M - E + e * sE) / (1.0 - e * cE);
        E = std::fmod(E + dE, k2Pi);
    }
    return std::fabs(dE) <= 1e-12;
}

// CHM_Get_Sat_Pos :85-210 -> state {x,y,z,clk bias, vx,vy,vz, clk drift}
static int sat_state(const Eph &p, double tx, double out[8])
{
    const double A = p.sqrtA * p.sqrtA;
    const double n = std::sqrt(kMu / (A * A * A)) + p.deln;
    double tc = half_week(tx - p.tocs);
    double clkb = p.f2 * tc * tc + p.f1 * tc + p.f0 - p.tgd;
    double tk = half_week(tx - clkb - p.toes);
    double E;
    if (!solve_kepler(std::fmod(p.M0 + n * tk, k2Pi), p.e, E)) return -1;
    const double dtr = kRelF * p.e * p.sqrtA * std::sin(E);
    tc = tx - (clkb + dtr) - p.tocs;
    clkb = p.f2 * tc * tc + p.f1 * tc + p.f0 + dtr - p.tgd;
    const double clkd = p.f1 + 2.0 * p.f2 * tc;
    tk = half_week(tx - clkb - p.toes);
    if (!solve_kepler(std::fmod(p.M0 + n * tk, k2Pi), p.e, E)) return -1;
    double sE, cE;
    sincos(E, &sE, &cE);
    const double den = 1.0 - p.e * cE;
    const double nu = std::atan2(std::sqrt(1.0 - p.e * p.e) * sE / den, (cE - p.e) / den);
    double u = std::fmod(nu + p.omg, k2Pi);
    double c2, s2;
    sincos(2.0 * u, &s2, &c2);
    u += p.cuc * c2 + p.cus * s2;
    const double r = A * den + p.crc * c2 + p.crs * s2;
    const double inc = p.i0 + p.idot * tk + p.cic * c2 + p.cis * s2;
    const double Om = std::fmod(p.OMG0 + (p.OMGd - kOEDot) * tk - kOEDot * p.toes, k2Pi);
    double su, cu, sO, cO, si, ci;
    sincos(u, &su, &cu);
    sincos(Om, &sO, &cO);
    sincos(inc, &si, &ci);
    const double xo = r * cu, yo = r * su;
    out[0] = xo * cO - yo * sO * ci;
    out[1] = xo * sO + yo * cO * ci;
    out[2] = yo * si;
    out[3] = clkb;
    sincos(2.0 * u, &s2, &c2);  // recomputed with the corrected u (:180-181)
    const double Ed = n / den;
    double snu, cnu;
    sincos(nu, &snu, &cnu);
    const double nud = sE * Ed * (1.0 + p.e * cnu) / (snu * den);
    const double ud = nud + 2.0 * (p.cus * c2 - p.cuc * s2) * nud;
    const double rd = A * p.e * sE * Ed + 2.0 * (p.crs * c2 - p.crc * s2) * nud;
    const double id = p.idot + (p.cis * c2 - p.cic * s2) * 2 * nud;
    const double vxo = rd * cu - yo * ud, vyo = rd * su + xo * ud;
    const double Omd = p.OMGd - kOEDot;
    const double ta = vxo - yo * ci * Omd, tb = xo * Omd + vyo * ci - yo * si * id;
    out[4] = ta * cO - tb * sO;
    out[5] = ta * sO + tb * cO;
    out[6] = vyo * si + yo * ci * id;
    out[7] = clkd;
    return 0;
}

struct Chan {
    int prn, cpElaStart, cpElaEnd, cpRef, cpRefTOW;
    double rcStart, rcEnd, riStart, riEnd, fc, fi, txTime;
    double sat[8];
    Eph eph;
};

static double wrap_pos(double v, double m)
{
    double t = std::fmod(v, m);
    return t < 0.0 ? t + m : t;
}

static double tx_of(const Chan &c, double cpEla, double rc)  // :258-260
{
    return c.cpRefTOW + ((cpEla - c.cpRef) * kTCA) + (rc / kFCA);
}

// Earth-rotation of a satellite state by the signal time of flight (:383-404, :895-916)
static void rotate_state_cs(const double s[8], double ct, double st, double o[8]);
static void rotate_state(const double s[8], double tau, double o[8])
{
    double ct, st;
    sincos(-kOEDot * tau, &st, &ct);
    rotate_state_cs(s, ct, st, o);
}
static void rotate_state_cs(const double s[8], double ct, double st, double o[8])
{
    o[0] = ct * s[0] - st * s[1];
    o[1] = st * s[0] + ct * s[1];
    o[2] = s[2];
    o[3] = s[3];
    o[4] = ct * s[4] - st * s[5] - kOEDot * st * s[0] - kOEDot * ct * s[1];
    o[5] = st * s[4] + ct * s[5] + kOEDot * ct * s[0] - kOEDot * st * s[1];
    o[6] = s[6];
    o[7] = s[7];
}

// back-calculated code phase (chips since the reference code period) for a receiver state x at
// receive time t and a rotated satellite state (:429-432, :763-774)
static double back_calc_rc(const Chan &c, const double sat[8], const double *x, double t, double *rangeOut)
{
    const double lx = sat[0] - x[0], ly = sat[1] - x[1], lz = sat[2] - x[2];
    const double range = std::sqrt(lx * lx + ly * ly + lz * lz);
    const double pr = range - kC * sat[3] + x[3];
    const double bcTx = t - pr / kC;
    const double frac = bcTx - c.cpRefTOW - ((c.cpElaEnd - c.cpRef) * kTCA);
    if (rangeOut) *rangeOut = range;
    return frac * kFCA;
}

// time update shared by CHM_TimeUpdateChannels (:675-823) and the tail of CHM_PropagateChannels (:451-602)
static int advance(Chan &c, const double *x, double rxTime, double T)
{
    const double adv = c.fc * T + c.rcEnd;
    const double cpPred = c.cpElaEnd + std::floor(adv / kLCA);
    const double rcPred = wrap_pos(adv, (double)kLCA);
    const double txPred = tx_of(c, cpPred, rcPred);
    double sp[8], sr[8];
    if (sat_state(c.eph, txPred, sp)) return -1;
    const double tau = rxTime + T - (txPred + (x[3] / kC)) + sp[3];
    rotate_state(sp, tau, sr);
    const double bcRc = back_calc_rc(c, sr, x, rxTime + T, nullptr);
    c.cpElaStart = c.cpElaEnd;
    c.rcStart = c.rcEnd;
    c.cpElaEnd += std::floor(bcRc / kLCA);
    c.rcEnd = wrap_pos(bcRc, (double)kLCA);
    c.riStart = c.riEnd;
    c.riEnd = wrap_pos(c.fi * T + c.riEnd, 1.0);
    c.txTime = tx_of(c, c.cpElaEnd, c.rcEnd);
    return sat_state(c.eph, c.txTime, c.sat);
}

}  // namespace dpe

struct dpe_chanmgr {
    std::vector<dpe::Chan> ch;
    double rxTime, T;
    int dopplerSign;
    std::vector<double> batch;  // [K][dimT][8]
    double R[9];
    double xkk1[8];
    int dimT = 0;
    bool started = false;
};

static void grid_prep(dpe_chanmgr *h, const double *xkk1, const double *timeGrid, int dimT)
{
    using namespace dpe;
    const int K = (int)h->ch.size();
    h->dimT = dimT;
    h->batch.resize((size_t)K * dimT * 8);
    memcpy(h->xkk1, xkk1, sizeof(double) * 8);
    const int mid = dimT / 2;
    for (int k = 0; k < K; ++k) {  // CHM_GridPrep :892-916
        const Chan &c = h->ch[k];
        // The entry the ML kernels read (dimT/2, BCM :1775) takes the reference's own evaluation.  The others differ from it
        // by a clock-offset step of metres / c in the time of flight, i.e. by d <= 1e-10 rad of Earth rotation: their
        // rotation is the mid one advanced to first order, cos(a+d) = cos a - d sin a, sin(a+d) = sin a + d cos a -- the
        // d^2/2 <= 1e-20 remainder is far below the last bit -- instead of 2 (dimT-1) more sin/cos evaluations per SV.
        const double tau0 = h->rxTime - (c.txTime + ((timeGrid[mid] + xkk1[3]) / kC)) + c.sat[3];
        double ct0, st0;
        sincos(-kOEDot * tau0, &st0, &ct0);
        for (int t = 0; t < dimT; ++t) {
            double *o = &h->batch[((size_t)k * dimT + t) * 8];
            if (t == mid) { rotate_state_cs(c.sat, ct0, st0, o); continue; }
            const double tau = h->rxTime - (c.txTime + ((timeGrid[t] + xkk1[3]) / kC)) + c.sat[3];
            const double d = -kOEDot * (tau - tau0);
            if (std::fabs(d) < 1e-8) rotate_state_cs(c.sat, ct0 - d * st0, st0 + d * ct0, o);
            else rotate_state(c.sat, tau, o);   // a time grid of kilometres: evaluate directly
        }
    }
    // CHM_Dev_ECEF2LL_Rad :37-50 + CHM_Dev_R_ENU2ECEF :54-73
    const double p = std::sqrt(xkk1[0] * xkk1[0] + xkk1[1] * xkk1[1]);
    const double th = std::atan2(xkk1[2] * kWgsA, p * kWgsB);
    const double lat = std::atan2(xkk1[2] + std::pow(kWgsEp, 2) * kWgsB * std::pow(std::sin(th), 3),
                                  p - std::pow(kWgsE, 2) * kWgsA * std::pow(std::cos(th), 3));
    const double lon = std::atan2(xkk1[1], xkk1[0]);
    double sa, ca, so, co;
    sincos(lat, &sa, &ca);
    sincos(lon, &so, &co);
    const double Rm[9] = {-so, -sa * co, ca * co, co, -sa * so, ca * so, 0.0, ca, sa};
    memcpy(h->R, Rm, sizeof(Rm));
}

extern "C" {

int dpe_chm_create(const dpe_chm_config *cfg, const dpe_chm_init_chan *chans, dpe_chanmgr **out)
{
    using namespace dpe;
    DPE_REQUIRE(cfg && chans && out, "[cuChanMgr] create: null argument");
    DPE_REQUIRE(cfg->nChan >= 1 && cfg->nChan <= DPE_MAX_CHAN, "[cuChanMgr] create: nChan out of range");
    DPE_REQUIRE(cfg->dopplerSign == 1 || cfg->dopplerSign == -1, "[cuChanMgr] create: DopplerSign must be +/-1");
    DPE_REQUIRE(cfg->sampleLength > 0, "[cuChanMgr] create: SampleLength must be positive");
    dpe_chanmgr *h = new dpe_chanmgr();
    h->rxTime = cfg->rxTime;
    h->T = std::round(cfg->sampleLength * 1.0e6) / 1.0e6;  // :1037
    h->dopplerSign = cfg->dopplerSign;
    h->ch.resize(cfg->nChan);
    for (int k = 0; k < cfg->nChan; ++k) {
        const dpe_chm_init_chan &s = chans[k];
        Chan &c = h->ch[k];
        c.prn = s.prn;
        c.cpElaStart = 0; c.cpElaEnd = s.cpElapsed; c.cpRef = s.cpReference; c.cpRefTOW = s.cpRefTOW;  // :1046-1071
        c.rcStart = 0; c.rcEnd = s.codePhase; c.riStart = 0; c.riEnd = s.carrierPhase;
        c.fc = s.codeFrequency; c.fi = s.carrierFrequency; c.txTime = 0;
        const double *e = s.eph;
        c.eph = Eph{e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8], e[9], e[10], e[11], e[12], e[13], e[14],
                    e[15], e[16], e[17], e[18], e[19], e[20]};
    }
    *out = h;
    return 0;
}

int dpe_chm_destroy(dpe_chanmgr *h)
{
    delete h;
    return 0;
}

int dpe_chm_start(dpe_chanmgr *h, const double *xk1k1, const double *xkk1, const double *timeGrid, int32_t dimT)
{
    using namespace dpe;
    DPE_REQUIRE(h && xk1k1 && xkk1 && timeGrid && dimT >= 1, "[cuChanMgr] Start: bad arguments");
    if (h->started) return 0;  // "Start: Already Started." (:1007-1010)
    for (Chan &c : h->ch) {    // CHM_ComputeSatStates :258-301
        c.txTime = tx_of(c, c.cpElaEnd, c.rcEnd);
        DPE_REQUIRE(sat_state(c.eph, c.txTime, c.sat) == 0, "[cuChanMgr] Start: Kepler iteration failed (PRN %d)", c.prn);
    }
    for (Chan &c : h->ch)      // CHM_TimeUpdateChannels
        DPE_REQUIRE(advance(c, xk1k1, h->rxTime, h->T) == 0, "[cuChanMgr] Start: Kepler iteration failed (PRN %d)", c.prn);
    h->rxTime += h->T;         // :1121
    grid_prep(h, xkk1, timeGrid, dimT);
    h->started = true;
    return 0;
}

int dpe_chm_update(dpe_chanmgr *h, const double *xk1k1, const double *xkk1, const double *timeGrid, int32_t dimT)
{
    using namespace dpe;
    DPE_REQUIRE(h && h->started, "[cuChanMgr] Error: Update() Failed due to SatPos not initialized");
    DPE_REQUIRE(xk1k1 && xkk1 && timeGrid && dimT >= 1, "[cuChanMgr] Update: bad arguments");
    const double *x = xk1k1;
    for (Chan &c : h->ch) {
        // measurement update of fi / fc from the new fix (CHM_PropagateChannels :380-447)
        double sr[8], range;
        const double tau = h->rxTime - (c.txTime + (x[3] / kC)) + c.sat[3];
        rotate_state(c.sat, tau, sr);
        const double bcRc = back_calc_rc(c, sr, x, h->rxTime, &range);
        const double ex = x[4] - kOEDot * x[1], ey = x[5] + kOEDot * x[0], ez = x[6];
        const double lx = sr[0] - x[0], ly = sr[1] - x[1], lz = sr[2] - x[2];
        const double lrr = ((lx / range) * (ex - sr[4])) + ((ly / range) * (ey - sr[5])) + ((lz / range) * (ez - sr[6]));
        const double bcFi = kFL1 * ((lrr - x[7]) / kC + sr[7]) / h->dopplerSign;
        const double bcFc = kFCA + (h->dopplerSign * kFCA / kFL1) * bcFi + (bcRc - c.rcEnd) / h->T;
        c.fi = bcFi;
        c.fc = bcFc;
        DPE_REQUIRE(advance(c, x, h->rxTime, h->T) == 0, "[cuChanMgr] Update: Kepler iteration failed (PRN %d)", c.prn);
    }
    h->rxTime += h->T;  // :1249
    grid_prep(h, xkk1, timeGrid, dimT);
    return 0;
}

int dpe_chm_outputs(dpe_chanmgr *h, dpe_chan_start *start, dpe_chan_end *end, dpe_bcm_window *win,
                    double *batchSatStates)
{
    using namespace dpe;
    DPE_REQUIRE(h && h->started, "[cuChanMgr] outputs: not started");
    const int K = (int)h->ch.size();
    for (int k = 0; k < K; ++k) {
        const Chan &c = h->ch[k];
        if (start) {
            dpe_chan_start &s = start[k];
            s.codePhaseStart = c.rcStart; s.carrierPhaseStart = c.riStart;
            s.codeFrequency = c.fc; s.carrierFrequency = c.fi;
            s.cpElapsedStart = c.cpElaStart; s.cpReference = c.cpRef; s.prn = c.prn; s.reserved = 0;
        }
        if (end) {
            dpe_chan_end &e = end[k];
            memcpy(e.satState, &h->batch[((size_t)k * h->dimT + h->dimT / 2) * 8], sizeof(double) * 8);  // mid-time, BCM :1775
            e.codePhaseEnd = c.rcEnd; e.codeFrequency = c.fc; e.carrierFrequency = c.fi;
            e.cpRefTOW = c.cpRefTOW; e.cpElapsedEnd = c.cpElaEnd; e.cpRef = c.cpRef; e.reserved = 0;
        }
    }
    if (win) {
        memcpy(win->xCurrkk1, h->xkk1, sizeof(double) * 8);
        memcpy(win->enu2ecef, h->R, sizeof(double) * 9);
        win->rxTime = h->rxTime;
        win->dopplerSign = h->dopplerSign;
        win->reserved = 0;
    }
    if (batchSatStates) memcpy(batchSatStates, h->batch.data(), sizeof(double) * h->batch.size());
    return 0;
}

}  // extern "C"
