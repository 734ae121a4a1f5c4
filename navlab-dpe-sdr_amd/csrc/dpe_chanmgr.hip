// dpe_chanmgr.hip -- fp64 channel manager feeding the hot path, in two forms that share one set of functions.
//
// Restates dsp::cuChanMgr (cudarecv/modules/src/cuchanmgr.cu:85-210,240-306,338-608,641-829,
// 853-923,1004-1268).  In the reference these are <<<1,64>>> kernels over K <= 37 channels whose
// outputs live in device arrays.
//   * dpe_chm_*      : on the host (a few microseconds of fp64 per window); fills the structs that dpe_bcs_update /
//                      dpe_bcm_update take;
//   * dpe_chm_dev_*  : on the device, as in the reference -- state and port arrays live in HBM, one <<<1,128>>> kernel per
//                      window that ALSO forms the measurement from the scan's arg-max keys (BCM_MakePosMeas / MakeVelMeas,
//                      batchcorrmanifold.cu:1977-2068), passes it through (EKF_PassMeas, cuekf.cu:147-159) and writes the
//                      parameter blocks of the attached BatchCorrScores / BatchCorrManifold handles, so that a closed loop
//                      enqueues  bank -> finalize -> scan -> this kernel  per window and reads nothing back: the host polls the
//                      fixes from a pinned ring.  The two Kepler evaluations per channel and window (:85-210) are taken off the
//                      critical path: a second kernel on a side stream evaluates the satellite state and its time derivative
//                      at the nominal transmit time of the NEXT window while that window's correlator kernels run, and the
//                      per-window kernel advances it over the <= 1e-6 s that the fix moves the transmit time (second-order
//                      remainder 3e-11 m; direct evaluation beyond 1e-5 s, flagged).
#include "dpe_common.h"
#include "dpe_prep.h"

#ifdef __HIPCC__
#define DPE_HD __host__ __device__
#else
#define DPE_HD
#endif

namespace dpe {

struct Eph {  // subset of eph_t (cudarecv/utils/inc/ephhelper.h:98-125) used by CHM_Get_Sat_Pos
    double sqrtA, e, i0, OMG0, omg, M0, deln, OMGd, idot, crc, crs, cuc, cus, cic, cis, toes, tocs, f0, f1, f2, tgd;
};

constexpr double kMu = 3.9860050e14;       // ephhelper.h MU_GPS
constexpr double kRelF = -4.442807633e-10; // consthelper.h CONST_F
constexpr double k2Pi = 6.2831853071796;   // consthelper.h CONST_2PI
constexpr double kWgsA = 6378137.0, kWgsB = 6356752.314245, kWgsE = 0.08181919084262149, kWgsEp = 0.08209443794969568;

DPE_HD static inline double half_week(double t)  // CHM_Correct_Week_Crossover :26-31
{
    return t > 302400.0 ? t - 604800.0 : (t < -302400.0 ? t + 604800.0 : t);
}

DPE_HD static inline bool solve_kepler(double M, double e, double &E)  // :97-107
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
DPE_HD static inline int sat_state(const Eph &p, double tx, double out[8])
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

DPE_HD static inline double wrap_pos(double v, double m)
{
    double t = std::fmod(v, m);
    return t < 0.0 ? t + m : t;
}

DPE_HD static inline double tx_of(const Chan &c, double cpEla, double rc)  // :258-260
{
    return c.cpRefTOW + ((cpEla - c.cpRef) * kTCA) + (rc / kFCA);
}

// Earth-rotation of a satellite state by the signal time of flight (:383-404, :895-916)
DPE_HD static inline void rotate_state_cs(const double s[8], double ct, double st, double o[8]);
DPE_HD static inline void rotate_state(const double s[8], double tau, double o[8])
{
    double ct, st;
    sincos(-kOEDot * tau, &st, &ct);
    rotate_state_cs(s, ct, st, o);
}
DPE_HD static inline void rotate_state_cs(const double s[8], double ct, double st, double o[8])
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
DPE_HD static inline double back_calc_rc(const Chan &c, const double sat[8], const double *x, double t, double *rangeOut)
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
struct SatDirect {   // the reference's evaluation (CHM_Get_Sat_Pos)
    DPE_HD int operator()(const Chan &c, double tx, double out[8]) const { return sat_state(c.eph, tx, out); }
};
template <class SatFn>
DPE_HD static inline int advance(Chan &c, const double *x, double rxTime, double T, const SatFn &sat_at)
{
    const double adv = c.fc * T + c.rcEnd;
    const double cpPred = c.cpElaEnd + std::floor(adv / kLCA);
    const double rcPred = wrap_pos(adv, (double)kLCA);
    const double txPred = tx_of(c, cpPred, rcPred);
    double sp[8], sr[8];
    if (sat_at(c, txPred, sp)) return -1;
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
    return sat_at(c, c.txTime, c.sat);
}

// measurement update of fi / fc from the new fix (CHM_PropagateChannels :380-447), then the time update
template <class SatFn>
DPE_HD static inline int propagate(Chan &c, const double *x, double rxTime, double T, int dopplerSign, const SatFn &sat_at)
{
    double sr[8], range;
    const double tau = rxTime - (c.txTime + (x[3] / kC)) + c.sat[3];
    rotate_state(c.sat, tau, sr);
    const double bcRc = back_calc_rc(c, sr, x, rxTime, &range);
    const double ex = x[4] - kOEDot * x[1], ey = x[5] + kOEDot * x[0], ez = x[6];
    const double lx = sr[0] - x[0], ly = sr[1] - x[1], lz = sr[2] - x[2];
    const double lrr = ((lx / range) * (ex - sr[4])) + ((ly / range) * (ey - sr[5])) + ((lz / range) * (ez - sr[6]));
    const double bcFi = kFL1 * ((lrr - x[7]) / kC + sr[7]) / dopplerSign;
    const double bcFc = kFCA + (dopplerSign * kFCA / kFL1) * bcFi + (bcRc - c.rcEnd) / T;
    c.fi = bcFi;
    c.fc = bcFc;
    return advance(c, x, rxTime, T, sat_at);
}

// CHM_GridPrep :892-916 for one channel: the K x dimT batch satellite states.  The entry the ML kernels read (dimT / 2,
// BCM :1775) takes the reference's own evaluation.  The others differ from it by a clock-offset step of metres / c in the
// time of flight, i.e. by d <= 1e-10 rad of Earth rotation: their rotation is the mid one advanced to first order,
// cos(a + d) = cos a - d sin a, sin(a + d) = sin a + d cos a -- the d^2 / 2 <= 1e-20 remainder is far below the last bit --
// instead of 2 (dimT - 1) more sin/cos evaluations per SV.
DPE_HD static inline void batch_states(const Chan &c, double rxTime, const double *xkk1, const double *timeGrid, int dimT, double *out /* [dimT][8] */)
{
    const int mid = dimT / 2;
    const double tau0 = rxTime - (c.txTime + ((timeGrid[mid] + xkk1[3]) / kC)) + c.sat[3];
    double ct0, st0;
    sincos(-kOEDot * tau0, &st0, &ct0);
    for (int t = 0; t < dimT; ++t) {
        double *o = out + (size_t)t * 8;
        if (t == mid) { rotate_state_cs(c.sat, ct0, st0, o); continue; }
        const double tau = rxTime - (c.txTime + ((timeGrid[t] + xkk1[3]) / kC)) + c.sat[3];
        const double d = -kOEDot * (tau - tau0);
        if (std::fabs(d) < 1e-8) rotate_state_cs(c.sat, ct0 - d * st0, st0 + d * ct0, o);
        else rotate_state(c.sat, tau, o);   // a time grid of kilometres: evaluate directly
    }
}

// CHM_Dev_ECEF2LL_Rad :37-50 + CHM_Dev_R_ENU2ECEF :54-73: row-major ENU -> ECEF at the grid centre
DPE_HD static inline void enu2ecef_matrix(const double *xkk1, double Rm[9])
{
    const double p = std::sqrt(xkk1[0] * xkk1[0] + xkk1[1] * xkk1[1]);
    const double th = std::atan2(xkk1[2] * kWgsA, p * kWgsB);
    const double lat = std::atan2(xkk1[2] + std::pow(kWgsEp, 2) * kWgsB * std::pow(std::sin(th), 3),
                                  p - std::pow(kWgsE, 2) * kWgsA * std::pow(std::cos(th), 3));
    const double lon = std::atan2(xkk1[1], xkk1[0]);
    double sa, ca, so, co;
    sincos(lat, &sa, &ca);
    sincos(lon, &so, &co);
    Rm[0] = -so; Rm[1] = -sa * co; Rm[2] = ca * co;
    Rm[3] = co;  Rm[4] = -sa * so; Rm[5] = ca * so;
    Rm[6] = 0.0; Rm[7] = ca;       Rm[8] = sa;
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
    for (int k = 0; k < K; ++k) batch_states(h->ch[k], h->rxTime, xkk1, timeGrid, dimT, &h->batch[(size_t)k * dimT * 8]);
    enu2ecef_matrix(xkk1, h->R);
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
        DPE_REQUIRE(advance(c, xk1k1, h->rxTime, h->T, SatDirect{}) == 0, "[cuChanMgr] Start: Kepler iteration failed (PRN %d)", c.prn);
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
    for (Chan &c : h->ch)
        DPE_REQUIRE(propagate(c, xk1k1, h->rxTime, h->T, h->dopplerSign, SatDirect{}) == 0, "[cuChanMgr] Update: Kepler iteration failed (PRN %d)", c.prn);
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

// ============================================================================================
// Device-resident form (dpe_chm_dev_*)
#ifdef __HIPCC__
namespace dpe {

struct ChmRef {   // satellite state and its time derivative at a nominal transmit time (written by chm_ephem_kernel)
    double t0, s0[8], ds[8];
    int ok, pad;
};
struct ChmDevState {
    Chan ch[DPE_MAX_CHAN];
    ChmRef ref[DPE_MAX_CHAN];
    double rxTime, T;
    int dopplerSign, K, dimT;
    int status;        // sticky: 1 Kepler iteration failed, 2 a transmit time left the expansion's range (evaluated directly), 4 an arg-max key was 0 / out of range
    long long window;  // windows completed (Start does not count)
};
// The reference's output ports of cuChanMgr (cuchanmgr.cu:973-990,1136-1171) and cuEKF (cuekf.cu:277-279) as device arrays,
// plus zVal (BatchCorrManifold, :2297) and the TimeGrid input
struct ChmPorts {
    double *rxTime, *txTime, *rcStart, *riStart, *rcEnd, *riEnd, *fc, *fi, *sat, *enu2ecef, *xk1k1, *xkk1, *zVal, *timeGrid;
    int *dopplerSign, *cpRef, *cpElaStart, *cpElaEnd, *cpRefTOW;
    unsigned char *prn;
};
struct ChmKArgs {
    ChmDevState *st;
    ChmPorts p;
    int mode;                       // 0 Start (CHM_ComputeSatStates + time update), 1 Update (measurement + time update)
    const double *xk1k1, *xkk1;     // inputs 10 / 12 on the device; ignored when meas != 0
    // measurement from the attached BatchCorrManifold's keys (BCM_MakePosMeas / MakeVelMeas + EKF_PassMeas)
    int meas;
    const unsigned long long *keys; // {posKey, velKey, posOutOfWindow, velOutOfWindow} of the window just scanned
    const double *posGrid, *velGrid;
    long long posG, velG, posOff, velOff;
    dpe_fix_record *ring;           // pinned, device address
    int ringDepth;
    // parameter blocks of the attached handles for the NEXT window (nullptr: not attached)
    BcsChanDev *bcsChan;
    int *bcsStatus;
    double fs;
    int S;
    BcmSvDev *svPos, *svVel;
    BcmDevWin *devWin;
    double Cf;
    int L, B;
    long long C;
};

// satellite state at tx from the expansion about ref.t0; beyond +-1e-5 s (or without an expansion) the reference's evaluation
struct SatExpanded {
    const ChmRef *ref;
    int *flags;
    __device__ int operator()(const Chan &c, double tx, double out[8]) const
    {
        const double dt = tx - ref->t0;
        if (ref->ok && fabs(dt) < 1e-5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) out[i] = fma(ref->ds[i], dt, ref->s0[i]);
            return 0;
        }
        if (ref->ok) *flags |= 2;
        return sat_state(c.eph, tx, out);
    }
};

__global__ __launch_bounds__(128) void chm_dev_kernel(ChmKArgs a)
{
    __shared__ double sX1[8], sXk[8], sR[9];
    __shared__ int sFlags;
    ChmDevState *st = a.st;
    const int tid = threadIdx.x, K = st->K, dimT = st->dimT;
    const double rxTime = st->rxTime, T = st->T;
    const int ds = st->dopplerSign;
    if (tid == 0) sFlags = 0;
    // ---- 1. the state the channels are propagated from (xCurrk1k1) and the next grid centre (xCurrkk1)
    if (tid == 64) {
        if (a.meas) {
            // ML grid points -> ECEF measurement (BCM_MakePosMeas / MakeVelMeas :1990-1999, 2042-2051) about the grid centre and
            // with the ENU matrix the scan of this window used = the ports as the previous call left them
            const unsigned long long kp = a.keys[0], kv = a.keys[1];
            long long ip = (long long)(0xFFFFFFFFu - (unsigned)(kp & 0xFFFFFFFFull)) - a.posOff;
            long long iv = (long long)(0xFFFFFFFFu - (unsigned)(kv & 0xFFFFFFFFull)) - a.velOff;
            int bad = 0;
            if (kp == 0ull || ip < 0 || ip >= a.posG) { bad = 4; ip = 0; }
            if (kv == 0ull || iv < 0 || iv >= a.velG) { bad = 4; iv = 0; }
            const double *g = a.posGrid + 4 * ip, *v = a.velGrid + 4 * iv, *R = a.p.enu2ecef, *c = a.p.xkk1;
            double z[8];
            z[0] = R[0] * g[0] + R[1] * g[1] + R[2] * g[2] + c[0];
            z[1] = R[3] * g[0] + R[4] * g[1] + R[5] * g[2] + c[1];
            z[2] = R[6] * g[0] + R[7] * g[1] + R[8] * g[2] + c[2];
            z[3] = g[3] + c[3];
            z[4] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2] + c[4];
            z[5] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2] + c[5];
            z[6] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2] + c[6];
            z[7] = v[3] + c[7];
            if (bad) {   // no valid score this window: hold the state (and say so)
                for (int i = 0; i < 8; ++i) z[i] = c[i];
                atomicOr(&sFlags, bad);
            }
            for (int i = 0; i < 8; ++i) { sX1[i] = z[i]; sXk[i] = z[i]; a.p.zVal[i] = z[i]; }   // EKF_PassMeas: both state ports
            // the fix for the host: one record of the pinned ring, sequence word last
            dpe_fix_record *r = a.ring + (st->window % a.ringDepth);
            for (int i = 0; i < 8; ++i) __hip_atomic_store(&r->zVal[i], z[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&r->rxTime, rxTime, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&r->posIndex, (long long)(ip + a.posOff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&r->velIndex, (long long)(iv + a.velOff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&r->posOutOfWindow, (long long)a.keys[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&r->velOutOfWindow, (long long)a.keys[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&r->posScore, __uint_as_float((unsigned)(kp >> 32)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&r->velScore, __uint_as_float((unsigned)(kv >> 32)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&r->status, st->status | bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&r->seq, (unsigned long long)(st->window + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            for (int i = 0; i < 8; ++i) { sX1[i] = a.xk1k1[i]; sXk[i] = a.xkk1[i]; }
        }
    }
    __syncthreads();
    // ---- 2. wave 1: the ENU -> ECEF matrix of the next grid centre (CHM_Dev_R_ENU2ECEF), beside the channels of wave 0
    if (tid == 64) {
        double Rm[9];
        enu2ecef_matrix(sXk, Rm);
        for (int i = 0; i < 9; ++i) { sR[i] = Rm[i]; a.p.enu2ecef[i] = Rm[i]; }
        for (int i = 0; i < 8; ++i) { a.p.xk1k1[i] = sX1[i]; a.p.xkk1[i] = sXk[i]; }
        a.p.rxTime[0] = rxTime + T;
        a.p.dopplerSign[0] = ds;
    }
    // ---- 3. wave 0: one channel per lane
    Chan c;
    const bool live = tid < K;
    if (live) {
        c = st->ch[tid];
        int flags = 0;
        const SatExpanded sat_at{&st->ref[tid], &flags};
        int rc;
        if (a.mode == 0) {   // CHM_ComputeSatStates :258-301, then CHM_TimeUpdateChannels
            c.txTime = tx_of(c, c.cpElaEnd, c.rcEnd);
            rc = sat_state(c.eph, c.txTime, c.sat);
            if (!rc) rc = advance(c, sX1, rxTime, T, sat_at);
        } else {
            rc = propagate(c, sX1, rxTime, T, ds, sat_at);
        }
        if (rc) flags |= 1;
        if (flags) atomicOr(&sFlags, flags);
        st->ch[tid] = c;
        // ports (cuchanmgr.cu:1136-1171)
        a.p.txTime[tid] = c.txTime;
        a.p.rcStart[tid] = c.rcStart; a.p.riStart[tid] = c.riStart;
        a.p.rcEnd[tid] = c.rcEnd;     a.p.riEnd[tid] = c.riEnd;
        a.p.fc[tid] = c.fc;           a.p.fi[tid] = c.fi;
        a.p.cpRef[tid] = c.cpRef;     a.p.cpElaStart[tid] = c.cpElaStart;
        a.p.cpElaEnd[tid] = c.cpElaEnd; a.p.cpRefTOW[tid] = c.cpRefTOW;
        a.p.prn[tid] = (unsigned char)c.prn;
        // CHM_GridPrep :892-916 at the new receive time
        batch_states(c, rxTime + T, sXk, a.p.timeGrid, dimT, a.p.sat + (size_t)tid * dimT * 8);
        // BatchCorrScores' block for the next window
        if (a.bcsChan) {
            int bad;
            a.bcsChan[tid] = bcs_prep_one(c.rcStart, c.riStart, c.fc, c.fi, c.cpElaStart, c.cpRef, c.prn, a.fs, a.S, bad);
            if (bad) atomicOr(&sFlags, bad << 4);
        }
    }
    __syncthreads();   // the matrix of wave 1
    if (live && a.svPos) {
        BcmSvDev ap, av;
        bcm_prep_one(sXk, sR, a.p.sat + ((size_t)tid * dimT + dimT / 2) * 8, c.rcEnd, c.fc, c.fi, c.cpRefTOW, c.cpElaEnd, c.cpRef, ds, rxTime + T,
                     a.fs, a.Cf, a.S, a.L, a.B, a.C, ap, av);
        a.svPos[tid] = ap;
        a.svVel[tid] = av;
    }
    if (tid == 0) {
        st->rxTime = rxTime + T;   // :1121, :1249
        if (a.mode == 1) st->window += 1;
        if (sFlags) st->status |= sFlags & 7;
        if (a.bcsStatus) *a.bcsStatus = (sFlags >> 4) & 3;
        if (a.devWin) {   // the window frame dpe_bcm_results would read (pinned)
            for (int i = 0; i < 8; ++i) a.devWin->xCurrkk1[i] = sXk[i];
            for (int i = 0; i < 9; ++i) a.devWin->enu2ecef[i] = sR[i];
            a.devWin->dopplerSign = ds;
            a.devWin->bad = 0;
        }
    }
}

// Off the critical path: satellite state and its derivative (forward difference over 2^-10 s: the step error a h / 2 x dt is
// 3e-11 m in position for dt = 1e-7 s) at the nominal transmit time of the next window, t0 = txTime + T.  The per-window
// kernel's two evaluations (predicted and back-calculated transmit time, :451-602) lie within ~1e-7 s of it.
__global__ void chm_ephem_kernel(ChmDevState *st)
{
    const int k = threadIdx.x;
    if (k >= st->K) return;
    const Chan &c = st->ch[k];
    ChmRef r;
    r.t0 = c.txTime + st->T;
    const double h = 0.0009765625;
    double s1[8];
    r.ok = (sat_state(c.eph, r.t0, r.s0) == 0 && sat_state(c.eph, r.t0 + h, s1) == 0) ? 1 : 0;
    for (int i = 0; i < 8; ++i) r.ds[i] = (s1[i] - r.s0[i]) / h;
    r.pad = 0;
    st->ref[k] = r;
}

}  // namespace dpe

struct dpe_chm_dev {
    dpe::ChmDevState *st_d = nullptr;
    char *portBuf_d = nullptr;
    dpe::ChmPorts p{};
    int K = 0, dimT = 0;
    bool started = false, aheadOff = false;
    dpe_bcs *bcs = nullptr;
    dpe_bcm *bcm = nullptr;
    dpe_bcs_hook hb{};
    dpe_bcm_hook hm{};
    dpe_fix_record *ring_h = nullptr, *ring_hd = nullptr;
    int ringDepth = 0;
    long long enqueued = 0;          // Updates enqueued since Start
    hipStream_t side = nullptr;      // the expansion kernel's stream
    hipEvent_t evState = nullptr, evRef = nullptr;
};

static int chm_dev_launch(dpe_chm_dev *h, int mode, int meas, const double *xk1k1, const double *xkk1, hipStream_t stream)
{
    using namespace dpe;
    ChmKArgs a{};
    a.st = h->st_d;
    a.p = h->p;
    a.mode = mode;
    a.xk1k1 = xk1k1;
    a.xkk1 = xkk1;
    a.meas = meas;
    if (meas) {
        const uint64_t *keys = nullptr;
        if (dpe_bcm_keys(h->bcm, &keys)) return -1;
        a.keys = reinterpret_cast<const unsigned long long *>(keys);
        a.posGrid = h->hm.posGrid64_d; a.velGrid = h->hm.velGrid64_d;
        a.posG = h->hm.posG; a.velG = h->hm.velG; a.posOff = h->hm.posOffset; a.velOff = h->hm.velOffset;
        a.ring = h->ring_hd;
        a.ringDepth = h->ringDepth;
    }
    if (h->bcs) { a.bcsChan = h->hb.chan_d; a.bcsStatus = h->hb.status_d; a.fs = h->hb.fs; a.S = h->hb.S; }
    if (h->bcm) {
        a.svPos = h->hm.svPos_d; a.svVel = h->hm.svVel_d; a.devWin = h->hm.devWin_hd;
        a.fs = h->hm.fs; a.Cf = h->hm.Cf; a.S = h->hm.S; a.L = h->hm.L; a.B = h->hm.B; a.C = h->hm.C;
    }
    // the expansion for this call must be there; the one for the next call starts as soon as this call's state is
    if (!h->aheadOff) DPE_CHECK_HIP(hipStreamWaitEvent(stream, h->evRef, 0));
    hipLaunchKernelGGL(chm_dev_kernel, dim3(1), dim3(128), 0, stream, a);
    if (!h->aheadOff) {
        DPE_CHECK_HIP(hipEventRecord(h->evState, stream));
        DPE_CHECK_HIP(hipStreamWaitEvent(h->side, h->evState, 0));
        hipLaunchKernelGGL(chm_ephem_kernel, dim3(1), dim3(64), 0, h->side, h->st_d);
        DPE_CHECK_HIP(hipEventRecord(h->evRef, h->side));
    }
    DPE_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" {

int dpe_chm_dev_create(const dpe_chm_config *cfg, const dpe_chm_init_chan *chans, const double *timeGrid, int32_t dimT, dpe_chm_dev **out)
{
    using namespace dpe;
    DPE_REQUIRE(cfg && chans && timeGrid && out, "[cuChanMgr] create: null argument");
    DPE_REQUIRE(cfg->nChan >= 1 && cfg->nChan <= DPE_MAX_CHAN, "[cuChanMgr] create: nChan out of range");
    DPE_REQUIRE(cfg->dopplerSign == 1 || cfg->dopplerSign == -1, "[cuChanMgr] create: DopplerSign must be +/-1");
    DPE_REQUIRE(cfg->sampleLength > 0 && dimT >= 1, "[cuChanMgr] create: SampleLength / dimT must be positive");
    std::vector<ChmDevState> st(1);
    memset(&st[0], 0, sizeof(ChmDevState));
    ChmDevState &s = st[0];
    s.rxTime = cfg->rxTime;
    s.T = std::round(cfg->sampleLength * 1.0e6) / 1.0e6;  // :1037
    s.dopplerSign = cfg->dopplerSign;
    s.K = cfg->nChan;
    s.dimT = dimT;
    for (int k = 0; k < cfg->nChan; ++k) {
        const dpe_chm_init_chan &in = chans[k];
        Chan &c = s.ch[k];
        c.prn = in.prn;
        c.cpElaStart = 0; c.cpElaEnd = in.cpElapsed; c.cpRef = in.cpReference; c.cpRefTOW = in.cpRefTOW;  // :1046-1071
        c.rcStart = 0; c.rcEnd = in.codePhase; c.riStart = 0; c.riEnd = in.carrierPhase;
        c.fc = in.codeFrequency; c.fi = in.carrierFrequency; c.txTime = 0;
        const double *e = in.eph;
        c.eph = Eph{e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8], e[9], e[10], e[11], e[12], e[13], e[14],
                    e[15], e[16], e[17], e[18], e[19], e[20]};
    }
    dpe_chm_dev *h = new dpe_chm_dev();
    h->K = cfg->nChan;
    h->dimT = dimT;
    h->aheadOff = getenv("DPE_CHM_NO_AHEAD") != nullptr;   // (A/B: every satellite state evaluated inside the per-window kernel)
    const size_t K = DPE_MAX_CHAN;
    // one buffer for all port arrays, 8-byte aligned pieces
    const size_t nD = 1 + 7 * K + K * (size_t)dimT * 8 + 9 + 8 + 8 + 8 + (size_t)dimT, nI = 1 + 4 * K;
    const size_t bytes = nD * 8 + nI * 4 + K + 64;
    h->st_d = dev_alloc<ChmDevState>(1);
    h->portBuf_d = dev_alloc<char>(bytes);
    const auto fail = [&](const char *msg) { set_error("%s", msg); dpe_chm_dev_destroy(h); return -1; };
    if (!h->st_d || !h->portBuf_d) return fail("[cuChanMgr] create: device allocation failed");
    double *d = reinterpret_cast<double *>(h->portBuf_d);
    ChmPorts &p = h->p;
    p.rxTime = d; d += 1;
    p.txTime = d; d += K; p.rcStart = d; d += K; p.riStart = d; d += K; p.rcEnd = d; d += K; p.riEnd = d; d += K; p.fc = d; d += K; p.fi = d; d += K;
    p.sat = d; d += K * (size_t)dimT * 8;
    p.enu2ecef = d; d += 9; p.xk1k1 = d; d += 8; p.xkk1 = d; d += 8; p.zVal = d; d += 8; p.timeGrid = d; d += dimT;
    int *ip = reinterpret_cast<int *>(d);
    p.dopplerSign = ip; ip += 1; p.cpRef = ip; ip += K; p.cpElaStart = ip; ip += K; p.cpElaEnd = ip; ip += K; p.cpRefTOW = ip; ip += K;
    p.prn = reinterpret_cast<unsigned char *>(ip);
    if (hipMemset(h->portBuf_d, 0, bytes) != hipSuccess || hipMemcpy(h->st_d, &s, sizeof(ChmDevState), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p.timeGrid, timeGrid, sizeof(double) * dimT, hipMemcpyHostToDevice) != hipSuccess ||
        hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->evState, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evRef, hipEventDisableTiming) != hipSuccess)
        return fail("[cuChanMgr] create: device initialisation failed");
    *out = h;
    return 0;
}

int dpe_chm_dev_destroy(dpe_chm_dev *h)
{
    if (!h) return 0;
    if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
    if (h->evState) (void)hipEventDestroy(h->evState);
    if (h->evRef) (void)hipEventDestroy(h->evRef);
    (void)hipFree(h->st_d);
    (void)hipFree(h->portBuf_d);
    if (h->ring_h) (void)hipHostFree(h->ring_h);
    delete h;
    return 0;
}

int dpe_chm_dev_attach(dpe_chm_dev *h, dpe_bcs *bcs, dpe_bcm *bcm, int32_t fixRingDepth)
{
    DPE_REQUIRE(h && !h->started, "[cuChanMgr] attach: before Start");
    DPE_REQUIRE(fixRingDepth >= 2 && fixRingDepth <= 4096, "[cuChanMgr] attach: fixRingDepth %d not in [2, 4096]", fixRingDepth);
    if (bcs) {
        if (dpe_bcs_hook_get(bcs, &h->hb)) return -1;
        DPE_REQUIRE(h->hb.maxChannels >= h->K, "[cuChanMgr] attach: the BatchCorrScores handle holds %d channels, %d tracked", h->hb.maxChannels, h->K);
        h->bcs = bcs;
    }
    if (bcm) {
        if (dpe_bcm_hook_get(bcm, &h->hm)) return -1;
        DPE_REQUIRE(h->hm.maxWindows == 1, "[cuChanMgr] attach: the BatchCorrManifold handle must be a single-window one (the closed loop)");
        DPE_REQUIRE(h->hm.maxChannels >= h->K, "[cuChanMgr] attach: the BatchCorrManifold handle holds %d channels, %d tracked", h->hm.maxChannels, h->K);
        h->bcm = bcm;
        if (!h->ring_h) {
            DPE_CHECK_HIP(hipHostMalloc((void **)&h->ring_h, sizeof(dpe_fix_record) * (size_t)fixRingDepth, hipHostMallocDefault));
            memset(h->ring_h, 0, sizeof(dpe_fix_record) * (size_t)fixRingDepth);
            DPE_CHECK_HIP(hipHostGetDevicePointer((void **)&h->ring_hd, h->ring_h, 0));
            h->ringDepth = fixRingDepth;
        }
    }
    return 0;
}

int dpe_chm_dev_ports(dpe_chm_dev *h, dpe_bcs_ports_dev *bcs, dpe_bcm_ports_dev *bcm, const double **rxTime_dev, double **xk1k1_dev,
                      double **xkk1_dev, const double **zVal_dev)
{
    DPE_REQUIRE(h, "[cuChanMgr] ports: null handle");
    const dpe::ChmPorts &p = h->p;
    if (bcs) *bcs = dpe_bcs_ports_dev{p.rcStart, p.riStart, p.fc, p.fi, p.cpElaStart, p.cpRef, p.prn};
    if (bcm) *bcm = dpe_bcm_ports_dev{p.xkk1, p.enu2ecef, p.sat, p.rcEnd, p.fc, p.fi, p.cpRefTOW, p.cpElaEnd, p.cpRef, p.dopplerSign, h->dimT, 0};
    if (rxTime_dev) *rxTime_dev = p.rxTime;
    if (xk1k1_dev) *xk1k1_dev = p.xk1k1;
    if (xkk1_dev) *xkk1_dev = p.xkk1;
    if (zVal_dev) *zVal_dev = p.zVal;
    return 0;
}

int dpe_chm_dev_start(dpe_chm_dev *h, const double *x0_host, dpe_stream_t stream_)
{
    DPE_REQUIRE(h && x0_host, "[cuChanMgr] Start: bad arguments");
    if (h->started) return 0;  // "Start: Already Started." (:1007-1010)
    hipStream_t stream = (hipStream_t)stream_;
    DPE_CHECK_HIP(hipMemcpyAsync(h->p.xk1k1, x0_host, sizeof(double) * 8, hipMemcpyHostToDevice, stream));
    DPE_CHECK_HIP(hipMemcpyAsync(h->p.xkk1, x0_host, sizeof(double) * 8, hipMemcpyHostToDevice, stream));
    DPE_CHECK_HIP(hipStreamSynchronize(stream));   // (x0_host may be pageable)
    DPE_CHECK_HIP(hipEventRecord(h->evRef, h->side));   // nothing to wait for yet: Start evaluates directly
    if (chm_dev_launch(h, 0, 0, h->p.xk1k1, h->p.xkk1, stream)) return -1;
    h->started = true;
    return 0;
}

int dpe_chm_dev_update(dpe_chm_dev *h, const double *xk1k1_dev, const double *xkk1_dev, dpe_stream_t stream)
{
    DPE_REQUIRE(h && h->started, "[cuChanMgr] Error: Update() Failed due to SatPos not initialized");
    DPE_REQUIRE(xk1k1_dev && xkk1_dev, "[cuChanMgr] Update: null state pointer");
    if (chm_dev_launch(h, 1, 0, xk1k1_dev, xkk1_dev, (hipStream_t)stream)) return -1;
    h->enqueued += 1;
    return 0;
}

int dpe_chm_dev_step(dpe_chm_dev *h, dpe_stream_t stream)
{
    DPE_REQUIRE(h && h->started, "[cuChanMgr] Error: Update() Failed due to SatPos not initialized");
    DPE_REQUIRE(h->bcm && h->ring_h, "[cuChanMgr] step: no BatchCorrManifold attached (dpe_chm_dev_attach)");
    if (chm_dev_launch(h, 1, 1, nullptr, nullptr, (hipStream_t)stream)) return -1;
    h->enqueued += 1;
    return 0;
}

int dpe_chm_dev_fix(dpe_chm_dev *h, int64_t window, dpe_fix_record *out, int32_t timeoutMicros)
{
    DPE_REQUIRE(h && out && h->ring_h, "[cuChanMgr] fix: no fix ring (dpe_chm_dev_attach)");
    DPE_REQUIRE(window >= 0 && window < h->enqueued, "[cuChanMgr] fix: window %lld not enqueued yet", (long long)window);
    const dpe_fix_record *r = h->ring_h + (window % h->ringDepth);
    const unsigned long long want = (unsigned long long)window + 1ull;
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        const unsigned long long seq = __atomic_load_n(&r->seq, __ATOMIC_ACQUIRE);
        if (seq == want) break;
        DPE_REQUIRE(seq < want, "[cuChanMgr] fix: window %lld was overwritten (the ring holds %d fixes)", (long long)window, h->ringDepth);
        if (timeoutMicros >= 0) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double us = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
            if (us > (double)timeoutMicros) return 1;   // not there yet
        }
    }
    memcpy(out, r, sizeof(dpe_fix_record));
    DPE_REQUIRE(__atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) == want, "[cuChanMgr] fix: window %lld was overwritten while it was read", (long long)window);
    return 0;
}

int dpe_chm_dev_read(dpe_chm_dev *h, dpe_chan_start *start, dpe_chan_end *end, dpe_bcm_window *win, double *batchSatStates, int32_t *status,
                     dpe_stream_t stream)
{
    using namespace dpe;
    DPE_REQUIRE(h && h->started, "[cuChanMgr] read: not started");
    DPE_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    std::vector<ChmDevState> st(1);
    DPE_CHECK_HIP(hipMemcpy(&st[0], h->st_d, sizeof(ChmDevState), hipMemcpyDeviceToHost));
    const int K = h->K, dimT = h->dimT;
    std::vector<double> batch((size_t)K * dimT * 8), fr(9 + 8);
    DPE_CHECK_HIP(hipMemcpy(batch.data(), h->p.sat, sizeof(double) * batch.size(), hipMemcpyDeviceToHost));
    DPE_CHECK_HIP(hipMemcpy(fr.data(), h->p.enu2ecef, sizeof(double) * 9, hipMemcpyDeviceToHost));
    DPE_CHECK_HIP(hipMemcpy(fr.data() + 9, h->p.xkk1, sizeof(double) * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < K; ++k) {
        const Chan &c = st[0].ch[k];
        if (start) {
            dpe_chan_start &s = start[k];
            s.codePhaseStart = c.rcStart; s.carrierPhaseStart = c.riStart;
            s.codeFrequency = c.fc; s.carrierFrequency = c.fi;
            s.cpElapsedStart = c.cpElaStart; s.cpReference = c.cpRef; s.prn = c.prn; s.reserved = 0;
        }
        if (end) {
            dpe_chan_end &e = end[k];
            memcpy(e.satState, &batch[((size_t)k * dimT + dimT / 2) * 8], sizeof(double) * 8);
            e.codePhaseEnd = c.rcEnd; e.codeFrequency = c.fc; e.carrierFrequency = c.fi;
            e.cpRefTOW = c.cpRefTOW; e.cpElapsedEnd = c.cpElaEnd; e.cpRef = c.cpRef; e.reserved = 0;
        }
    }
    if (win) {
        memcpy(win->xCurrkk1, fr.data() + 9, sizeof(double) * 8);
        memcpy(win->enu2ecef, fr.data(), sizeof(double) * 9);
        win->rxTime = st[0].rxTime;
        win->dopplerSign = st[0].dopplerSign;
        win->reserved = 0;
    }
    if (batchSatStates) memcpy(batchSatStates, batch.data(), sizeof(double) * batch.size());
    if (status) *status = st[0].status;
    return 0;
}

}  // extern "C"
#endif  // __HIPCC__
