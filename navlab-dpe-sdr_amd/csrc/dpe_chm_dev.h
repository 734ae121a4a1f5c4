// dpe_chm_dev.h -- the channel manager's arithmetic (cudarecv/modules/src/cuchanmgr.cu:26-923), shared by its host form
// (dpe_chm_*, dpe_chanmgr.hip), its device form (dpe_chm_dev_*) and the stage-1 kernels of dpe_bcs.hip that carry the device
// form's time update as an extra block of their launch.
#pragma once
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
    const double x = -kOEDot * tau;
#ifdef __HIP_DEVICE_COMPILE__
    // The angle is the Earth's rotation over a signal's travel time, ~5e-6 rad: below 2^-10 the series to x^5 / x^6 are exact to the last
    // bit (next terms x^7 / 5040 and x^8 / 40320 are < 2e-22 of sin x and < 3e-29), and the device library's sincos -- argument
    // reduction and all, ~230 instructions of the lone wave of chm_k1, which pays ~5 clocks for each -- is not needed.
    if (fabs(x) < 0x1p-10) {
        const double x2 = x * x;
        st = x * fma(x2, fma(x2, 1.0 / 120.0, -1.0 / 6.0), 1.0);
        ct = fma(x2, fma(x2, fma(x2, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
    } else
#endif
        sincos(x, &st, &ct);
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

#ifdef __HIPCC__
namespace dpe {

struct ChmDevState {
    Chan ch[DPE_MAX_CHAN];
    double warmE[DPE_MAX_CHAN], warmM[DPE_MAX_CHAN];   // Kepler warm start: eccentric / mean anomaly of the channel's last evaluation
    int warmOk[DPE_MAX_CHAN];
    double rxTime, T;
    int dopplerSign, K, dimT;
    int status;        // sticky: 1 Kepler iteration failed, 4 an arg-max key was 0 / out of range, 8 / 16 BatchCorrScores input flags
    long long window;  // windows completed (Start does not count)
    // ---- speculative satellite states (chm_k3): the Kepler solutions of the NEXT time update, computed one window ahead beside
    // the current one (two buffers, alternating with the window parity the host passes in ChmKArgs::specPar)
    Eph ephC[DPE_MAX_CHAN];                 // the ephemerides again, never written after create (chm_k3 reads them while chm_k2 rewrites ch[])
    double snapFc[DPE_MAX_CHAN], snapRcEnd[DPE_MAX_CHAN];   // chm_k1's snapshot of what chm_k3 needs of a channel (chm_k2 overwrites ch[] beside it)
    int snapCpElaEnd[DPE_MAX_CHAN], snapCpRef[DPE_MAX_CHAN], snapCpRefTOW[DPE_MAX_CHAN];
    double specTx[2][DPE_MAX_CHAN], specSat[2][2][DPE_MAX_CHAN][8];   // [buffer][at specTx | at specTx + kChmH]
    int specOk[2][DPE_MAX_CHAN];
    double specWE[2][DPE_MAX_CHAN], specWM[2][DPE_MAX_CHAN];          // chm_k3's own Kepler warm start, per wave
    int specWOk[2][DPE_MAX_CHAN];
};
// The reference's output ports of cuChanMgr (cuchanmgr.cu:973-990,1136-1171) and cuEKF (cuekf.cu:277-279) as device arrays,
// plus zVal (BatchCorrManifold, :2297) and the TimeGrid input
struct ChmPorts {
    double *rxTime, *txTime, *rcStart, *riStart, *rcEnd, *riEnd, *fc, *fi, *sat, *enu2ecef, *xk1k1, *xkk1, *zVal, *timeGrid;
    int *dopplerSign, *cpRef, *cpElaStart, *cpElaEnd, *cpRefTOW;
    unsigned char *prn;
};
// cuEKF's filter (EnableEKF = true) as device state: dsp::cuEKF::StepUpdate / StepPredict (cuekf.cu:626-742) run inside the
// measurement kernel, one lane per matrix element.  Same fields, same row-major layout as the host form (dpe_ekf.hip).
struct EkfDev {
    double F[64], H[64], Q[64], K[64], Pk1k1[64], Pkk1[64], xk1k1[8], xkk1[8];
    double lpfVals[20];
    double lpfAvg;
    int lpfIdx, failed;   // failed: S was singular in some window (the state is then held for that window, bit 32 of the status)
    // the structure dpe_ekf_create gives F and H, which the device form exploits (see ekf_dev_step): H = I; F = I, plus Tc on [i][i + 4],
    // i < 4, when `coupled`
    int coupled, pad;
    double Tc;
};
struct ChmKArgs {
    ChmDevState *st;
    ChmPorts p;
    int mode;                       // 0 Start (CHM_ComputeSatStates + time update), 1 Update (measurement + time update)
    int specPar;                    // which speculative buffer chm_k2 reads; chm_k3 (same launch) fills the other one
    const double *xk1k1, *xkk1;     // inputs 10 / 12 on the device; ignored when meas != 0
    // measurement from the attached BatchCorrManifold's keys (BCM_MakePosMeas / MakeVelMeas + EKF_PassMeas)
    int meas;
    const unsigned long long *keys; // {posKey, velKey, posOutOfWindow, velOutOfWindow} of the window just scanned
    const double *posGrid, *velGrid;
    long long posG, velG, posOff, velOff;
    dpe_fix_record *ring;           // pinned, device address
    int ringDepth;
    dpe_fix_record *stage;          // device memory: chm_k1 leaves the window's record here, the chm_k2 that follows it sends it over the host link
    EkfDev *ekf;                    // nullptr: EKF_PassMeas (the shipped flow); else the filter runs on the measurement (dpe_chm_dev_set_ekf)
    // parameter blocks of the attached handles for the NEXT window (nullptr: not attached)
    BcsChanDev *bcsChan;
    int *bcsStatus;
    int hintL1;                     // > 0: the attached BatchCorrScores chooses its kernel from nominal values (dpe_bcs_set_dev_hint): check the promise here
    double hintStepMax;
    int *hintViol;
    double fs;
    int S;
    BcmSvDev *svPos, *svVel;
    BcmDevWin *devWin;
    double Cf;
    int L, B;
    long long C;
};

// CHM_Get_Sat_Pos with the Kepler iterations started near their fixed point: the first solve from the channel's previous
// eccentric anomaly advanced by the change of the mean anomaly (two iterations instead of five at e = 0.01), the second -- whose
// mean anomaly differs by n x 1e-8 s -- from the first.  The iteration's fixed point does not depend on where it starts; the
// values agree with the cold start to the last bits (1e-16 in E).
__device__ static inline bool kepler_from(double M, double e, double &E)
{
    double dE = 1.0;
    for (int it = 0; it < 10 && fabs(dE) > 1e-12; ++it) {
        double sE, cE;
        sincos(E, &sE, &cE);
        dE = (M - E + e * sE) / (1.0 - e * cE);
        E = fmod(E + dE, k2Pi);
    }
    return fabs(dE) <= 1e-12;
}
__device__ static inline int sat_state_warm(const Eph &p, double tx, double out[8], double &wE, double &wM, int &wOk)
{
    const double A = p.sqrtA * p.sqrtA;
    const double n = sqrt(kMu / (A * A * A)) + p.deln;
    double tc = half_week(tx - p.tocs);
    double clkb = p.f2 * tc * tc + p.f1 * tc + p.f0 - p.tgd;
    double tk = half_week(tx - clkb - p.toes);
    double M = fmod(p.M0 + n * tk, k2Pi);
    double E = M;
    if (wOk && fabs(M - wM) < 1e-2) E = wE + (M - wM);
    if (!kepler_from(M, p.e, E)) return -1;
    const double dtr = kRelF * p.e * p.sqrtA * sin(E);
    tc = tx - (clkb + dtr) - p.tocs;
    clkb = p.f2 * tc * tc + p.f1 * tc + p.f0 + dtr - p.tgd;
    const double clkd = p.f1 + 2.0 * p.f2 * tc;
    tk = half_week(tx - clkb - p.toes);
    M = fmod(p.M0 + n * tk, k2Pi);
    if (!kepler_from(M, p.e, E)) return -1;
    wE = E; wM = M; wOk = 1;
    double sE, cE;
    sincos(E, &sE, &cE);
    const double den = 1.0 - p.e * cE;
    const double nu = atan2(sqrt(1.0 - p.e * p.e) * sE / den, (cE - p.e) / den);
    double u = fmod(nu + p.omg, k2Pi);
    double c2, s2;
    sincos(2.0 * u, &s2, &c2);
    u += p.cuc * c2 + p.cus * s2;
    const double r = A * den + p.crc * c2 + p.crs * s2;
    const double inc = p.i0 + p.idot * tk + p.cic * c2 + p.cis * s2;
    const double Om = fmod(p.OMG0 + (p.OMGd - kOEDot) * tk - kOEDot * p.toes, k2Pi);
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

// The per-window work in two pieces, so that only the first sits between the scan of window n and the correlator of window n+1:
//
//   chm_k1 (one wave, a kernel of its own behind the scan): the measurement from the scan's keys (BCM_MakePosMeas / MakeVelMeas,
//       batchcorrmanifold.cu:1977-2068), the pass-through (EKF_PassMeas, cuekf.cu:147-159), the measurement update of fi / fc
//       (CHM_PropagateChannels :380-447), BatchCorrScores' parameter block of the next window, the fix for the host (the
//       record's fields leave for the pinned ring as soon as the measurement exists, its sequence word at the kernel's end).
//   chm_k2 (one block of 256 threads): the time update (CHM_TimeUpdateChannels :675-823) with its two Kepler evaluations
//       (:85-210), CHM_GridPrep (:853-923), BatchCorrManifold's coefficient blocks.  Nothing the next window's stage 1 reads
//       -- it runs as an extra block of that window's stage-1 LAUNCH (bcs_bank_kernel, FUSE form), beside the correlator blocks,
//       and is complete when the scan starts.  (As a kernel of its own for a host that keeps the reference's module order.)
//       Waves 0 / 1: a channel's two evaluations lie ~1e-7 s apart, so the second is the first advanced along the difference
//       quotient over h = 2^-10 s (step error a h / 2 x dt = 3e-11 m) and the two run side by side.  Wave 2: the ENU matrix.
//       Then all threads: the K x dimT batch states.
constexpr double kChmH = 0.0009765625;

// ---- cuEKF on the device: the 8 x 8 fp64 steps of dpe_ekf.hip with one lane per matrix element (a block of 64 threads).
// Every element is formed by the same operations in the same order as the host form (sums over k = 0 .. 7 ascending, no fused
// multiply-add), so the two give the same doubles.  sm: 6 x 64 doubles of LDS scratch.
#if defined(DPE_EXPERIMENTS) && defined(DPE_EKF_STAMPS)   // (timing attribution: shader-clock stamps of the phases of chm_k1, left in the fix record's two
#define DPE_EKF_STAMP(i) do { if (threadIdx.x == 0) dpe_ekf_stamps[i] = __builtin_readcyclecounter(); } while (0)   // out-of-window counts)
__shared__ unsigned long long dpe_ekf_stamps[10];
#else
#define DPE_EKF_STAMP(i) do { } while (0)
#endif
__device__ static inline double ekf_dev_mul(const double *a, const double *b, bool bT, double *c)
{
#pragma clang fp contract(off)
    const int l = threadIdx.x, i = l >> 3, j = l & 7;
    double s = 0.0;
    for (int k = 0; k < 8; ++k) s += a[i * 8 + k] * (bT ? b[j * 8 + k] : b[k * 8 + j]);
    __syncthreads();   // (c may alias a or b)
    c[l] = s;
    __syncthreads();
    return s;          // (this lane's element: the caller need not read it back)
}
// inverse through LU with partial pivoting (getrf + getri, cuekf.cu:681-694); false: singular.
// Round 6: the FACTORISATION runs on registers -- lane l = (r, kk) holds lu[r][kk]; the block is one wave, so a column's values at or
// below the diagonal are v_readlane pairs (wave-uniform: the pivot search is the host form's, first largest |.| at or below the
// diagonal), the row swap and the pivot row's element are one ds_bpermute round trip per column, a lane's own row's element of the
// column is a DPP row broadcast (rows c and p change places, so behind the swap row p holds the old diagonal element -- a uniform
// value -- and every other row below the diagonal what it held before), and nothing waits at a barrier.  Through LDS it was five
// dependent round trips and four barriers per column.  A lone wave is ISSUE-bound here (shader-clock stamps, scripts/ekf_stamps.py:
// 954 clocks per column at ~150 instructions, most of them selects): the DPP form of the own-row element replaced a 38-instruction
// select chain per column, and the pivot search became a scalar chain (below): chm_k1 with the filter 9.7-9.9 -> 8.8-9.0 us.  (Measured and
// dropped: the divisions with the divisor's reciprocal refined once, off the chain -- three dependent operations per quotient instead
// of thirteen, bit-identical under an exponent-range guard -- were SLOWER, 9.65-9.9 against 8.75-9.0 us on one box: the guards and
// selects add instructions, and what this wave pays for is instructions, ~5 clocks each, not the latency of a chain.)  Every element still sees the host form's operations in the host form's order: lu[r][c] /= lu[c][c];
// lu[r][k] -= lu[r][c] * lu[c][k] (no fused multiply-add).  The two substitutions (one column of the inverse per lane, 28 dependent
// multiply-subtracts each) read the factors back from LDS as before.
__device__ __forceinline__ double ekf_readlane_d(double v, int lane)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// lane N of every 16-lane row to all lanes of the row (row_newbcast); lane l + 4 of the row to lane l (row_shl:4, zero beyond the row)
template <int N>
__device__ __forceinline__ double ekf_row_bcast_d(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x150 + N, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x150 + N, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double ekf_row_shl4_d(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x104, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x104, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// column C of the factorisation; false: the column is zero at and below the diagonal (singular)
template <int C>
__device__ __forceinline__ bool ekf_lu_column(double &x, int &pv, int l, int r, int kk)
{
#pragma clang fp contract(off)
    // The pivot row: the first largest |.| at or below the diagonal, as the host form finds it.  The candidates are wave-uniform
    // (v_readlane pairs), and |a| > |b| on finite doubles is the unsigned order of their magnitude bits: the search is a chain of scalar
    // subtract-with-borrow / select steps -- five scalar instructions per candidate; as v_cmp_gt_f64 on scalar operands every step
    // was a vector compare between two scalar selects, each waiting for the other pipe.  (NaN-free input: a NaN in S makes every
    // later state NaN in either form.)
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    unsigned bLo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, C * 8 + C);
    unsigned bHi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), C * 8 + C) & 0x7fffffffu;
    int p = C;
#pragma unroll
    for (int q = C + 1; q < 8; ++q) {
        const unsigned cLo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, q * 8 + C);
        const unsigned cHi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), q * 8 + C) & 0x7fffffffu;
        unsigned t;
        asm("s_sub_u32 %3, %0, %4\n\ts_subb_u32 %3, %1, %5\n\ts_cselect_b32 %0, %4, %0\n\ts_cselect_b32 %1, %5, %1\n\ts_cselect_b32 %2, %6, %2"
            : "+s"(bLo), "+s"(bHi), "+s"(p), "=&s"(t)
            : "s"(cLo), "s"(cHi), "s"(q)
            : "scc");
    }
    if ((bLo | bHi) == 0u) return false;
    const double pivVal = ekf_readlane_d(x, p * 8 + C);
    const double diagOld = ekf_readlane_d(x, C * 8 + C);   // (row p holds it behind the swap)
    // this lane's own row's element of column C before the swap: lane (r, C) = lane C or 8 + C of the lane's 16-lane row
    const double ownEven = ekf_row_bcast_d<C>(x), ownOdd = ekf_row_bcast_d<8 + C>(x);
    const double own = (r & 1) ? ownOdd : ownEven;
    // the row swap and the pivot row's element of this lane's column, both from the values before the swap: one round trip
    const int src = (r == C) ? p * 8 + kk : (r == p) ? C * 8 + kk : l;
    const double xs = __shfl(x, src, 64), xck = __shfl(x, p * 8 + kk, 64);
    const int ps = (l == C) ? p : (l == p) ? C : l;
    pv = __shfl(pv, ps, 64);
    x = xs;
    if (r > C) {
        const double xc = (r == p) ? diagOld : own;   // this lane's row's element of column C after the swap
        const double f = xc / pivVal;
        if (kk == C) x = f;
        else if (kk > C) x -= f * xck;
    }
    return true;
}
__device__ static inline bool ekf_dev_invert(const double *a, double *lu, double *inv, int *piv)
{
#pragma clang fp contract(off)
    const int l = threadIdx.x, r = l >> 3, kk = l & 7;
    double x = a[l];
    int pv = l;                      // lanes 0 .. 7: piv[l]
    const bool singular = !(ekf_lu_column<0>(x, pv, l, r, kk) && ekf_lu_column<1>(x, pv, l, r, kk) && ekf_lu_column<2>(x, pv, l, r, kk) &&
                            ekf_lu_column<3>(x, pv, l, r, kk) && ekf_lu_column<4>(x, pv, l, r, kk) && ekf_lu_column<5>(x, pv, l, r, kk) &&
                            ekf_lu_column<6>(x, pv, l, r, kk) && ekf_lu_column<7>(x, pv, l, r, kk));
    if (singular) return false;      // (wave-uniform)
    DPE_EKF_STAMP(3);
    lu[l] = x;
    (void)piv;
    int pvRow[8];                    // (wave-uniform: the row permutation, from lanes 0 .. 7)
#pragma unroll
    for (int i = 0; i < 8; ++i) pvRow[i] = __builtin_amdgcn_readlane(pv, i);
    __syncthreads();
    if (l < 8) {   // one column of the inverse per lane
        const int cl = l;
        double y[8], xx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double s = (pvRow[i] == cl) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) s -= lu[i * 8 + k] * y[k];
            y[i] = s;
        }
#pragma unroll
        for (int i = 7; i >= 0; --i) {
            double s = y[i];
#pragma unroll
            for (int k = i + 1; k < 8; ++k) s -= lu[i * 8 + k] * xx[k];
            xx[i] = s / lu[i * 8 + i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) inv[i * 8 + cl] = xx[i];
    }
    __syncthreads();
    return true;
}
// StepUpdate (:660-721) with the measurement z and R = I as BatchCorrManifold emits it (:2003-2011), then StepPredict (:626-656).
// Returns false when S is singular (state untouched).  sm: kEkfDevScratch doubles of LDS.
// Round 6: eight of the ten 8 x 8 products have H or F as a factor, and dpe_ekf_create fixes their structure -- H = I, F = I (+ Tc on
// [i][i + 4], i < 4).  A dot product of the host form, s = 0.0; s += a[i][k] * b[k][j] for k = 0 .. 7 without fused multiply-add, then
// reduces to its structural terms IN THE SAME ORDER: the exact-zero factors contribute +-0.0, which leaves a sum that started at +0.0
// unchanged (and +0.0 + x = x exactly, -0.0 included: it becomes +0.0 in both forms -- hence the literal "0.0 +" below).  So
//   H X = X H^T = 0.0 + X,      (F X)[i][j] = (0.0 + X[i][j]) + Tc X[i + 4][j]  (i < 4),      (X F^T)[i][j] = (0.0 + X[i][j]) + X[i][j + 4] Tc  (j < 4)
// give the host form's doubles bit for bit with one or two operations per element instead of sixteen LDS reads and eight dependent
// multiply-adds; the two dense products (K = P S^-1 and (I - K) P) and the LU inverse stay as they were.  chm_k1 with the filter:
// 11.7 -> see profiles/r6_closed_loop_device.txt.
constexpr int kEkfDevScratch = 11 * 64, kEkfDevXk = 640, kEkfDevX1 = 648;   // (x_k+1|k and x_k|k stay at these offsets of the scratch)
// the filter's operands of lane l, requested at the top of chm_k1 -- beside the keys and the grid rows of the measurement, not behind them
// (one ~1.5 us round trip to device memory off the kernel's dependent chain)
struct EkfPre {
    double p, xk, lpf, lpfAvg, Tc;
    int lpfIdx, coupled;
};
__device__ static inline EkfPre ekf_dev_prefetch(const EkfDev *e)
{
    const int l = threadIdx.x;
    EkfPre r;
    r.p = e->Pkk1[l];
    r.xk = l < 8 ? e->xkk1[l] : 0.0;
    r.lpf = l < 20 ? e->lpfVals[l] : 0.0;
    r.lpfAvg = e->lpfAvg;
    r.Tc = e->Tc;
    r.lpfIdx = e->lpfIdx;
    r.coupled = e->coupled;
    return r;
}
__device__ static inline bool ekf_dev_step(EkfDev *e, const EkfPre &pre, const double *z, double *sm, int *piv)
{
#pragma clang fp contract(off)
    const int l = threadIdx.x, i = l >> 3, j = l & 7;
    double *T = sm, *S = sm + 64, *Sinv = sm + 128, *lu = sm + 192, *y = sm + 256, *tmp = sm + 320;
    double *P = sm + 512, *P1 = sm + 576, *xk = sm + 640, *x1 = sm + 648;
    const bool coupled = pre.coupled != 0;
    const double Tc = pre.Tc;
    const double p = pre.p;
    P[l] = p;
    const int lpfIdx = __builtin_amdgcn_readfirstlane(pre.lpfIdx);
    const double lpfAvg0 = pre.lpfAvg;
    const double pz = 0.0 + p;                                     // H P = P H^T = 0.0 + P
    S[l] = (0.0 + pz) + ((l % 9 == 0) ? 1.0 : 0.0);                // S = (H P) H^T + R
    T[l] = pz;                                                     // P H^T, the first factor of K
    if (l < 8) y[l] = z[l] - pre.xk;                               // y = z - H x_k|k-1
    __syncthreads();
    DPE_EKF_STAMP(2);
#if defined(DPE_EXPERIMENTS) && defined(DPE_EKF_SKIP_INV)   // (timing attribution only: wrong results)
    Sinv[l] = S[l];
    __syncthreads();
#else
    if (!ekf_dev_invert(S, lu, Sinv, piv)) return false;
#endif
    DPE_EKF_STAMP(4);
    const double kv = ekf_dev_mul(T, Sinv, false, tmp);            // K = (P H^T) S^-1
    e->K[l] = kv;
    double xs1 = 0.0;                                              // lanes 0 .. 7: x_k|k
    if (l < 8) {                                                   // x_k|k = x_k|k-1 + K y
        double s = pre.xk;
        for (int k = 0; k < 8; ++k) s += tmp[l * 8 + k] * y[k];
        x1[l] = s;                                                 // (left in the scratch for the caller: kEkfDevX1)
        e->xk1k1[l] = s;
        xs1 = s;
    }
    {                                                              // P_k|k = (I - K H) P_k|k-1, K H = 0.0 + K
        double t = -(0.0 + kv);
        if (l % 9 == 0) t += 1.0;
        __syncthreads();                                           // (the reads of tmp above)
        T[l] = t;
    }
    __syncthreads();
    DPE_EKF_STAMP(5);
    const double p1 = ekf_dev_mul(T, P, false, P1);
    DPE_EKF_STAMP(6);
    e->Pk1k1[l] = p1;
    // ---- StepPredict with GetQVal (:733-742) and EKF_Update_Q (:42-78).  The block is one wave: the low-pass state is wave-uniform
    // (every lane forms it from lanes 4 .. 6 of x_k|k and the ring entry of lane lpfIdx), so nothing is broadcast through LDS.
    const double v4 = ekf_readlane_d(xs1, 4), v5 = ekf_readlane_d(xs1, 5), v6 = ekf_readlane_d(xs1, 6);
    const double v = sqrt(v4 * v4 + v5 * v5 + v6 * v6);
    const double av = lpfAvg0 - ekf_readlane_d(pre.lpf, lpfIdx) + (v / 20.0);
    if (l == 0) {
        e->lpfVals[lpfIdx] = v / 20.0;
        e->lpfAvg = av;
        e->lpfIdx = lpfIdx + 1 >= 20 ? 0 : lpfIdx + 1;
    }
    DPE_EKF_STAMP(7);
    const double q = 1.0 + 250.0 / fmin(fmax(av * av, 50.0), 125.0);
    const auto q0 = [&](int r, int c) -> double {                  // Q0: diagonal
        if (r != c) return 0.0;
        if (r == 4 || r == 5 || r == 6) return q;
        if (r == 7) return (2.5e-10) * (2.5e-10) * kC * kC;        // Q_CLOCK_DRIFT, cuekf.h:28
        return 0.0;
    };
    const auto fq = [&](int r, int c) -> double {                  // (F Q0)[r][c]
        double s = 0.0 + q0(r, c);
        if (coupled && r < 4) s += Tc * q0(r + 4, c);
        return s;
    };
    double qv = 0.0 + fq(i, j);                                    // Q = (F Q0) F^T
    if (coupled && j < 4) qv += fq(i, j + 4) * Tc;
    e->Q[l] = qv;
    {                                                              // x_k+1|k = F x_k|k
        const double up = ekf_row_shl4_d(xs1);                     // lane l + 4's value (lanes 0 .. 3 read lanes 4 .. 7)
        if (l < 8) {
            double s = 0.0 + xs1;
            if (coupled && l < 4) s += Tc * up;
            xk[l] = s;                                             // (left in the scratch for the caller: kEkfDevXk)
            e->xkk1[l] = s;
        }
    }
    {                                                              // P_k+1|k = (F P) F^T + Q
        double t = 0.0 + p1;
        if (coupled && i < 4) t += Tc * P1[l + 32];
        const double tr = ekf_row_shl4_d(t);                       // (F P)[i][j + 4]: four lanes on in the lane's own matrix row
        double s2 = 0.0 + t;
        if (coupled && j < 4) s2 += tr * Tc;
        e->Pkk1[l] = s2 + qv;
    }
    __syncthreads();
    return true;
}

__device__ static inline void chm_k1(const ChmKArgs &a)
{
    __shared__ double sX1[8], sXk[8], sZ[8];   // x_k|k, x_k+1|k, the measurement
    __shared__ double sEkf[kEkfDevScratch];
    __shared__ int sPiv[8];
    __shared__ int sFlags;
    ChmDevState *st = a.st;
    const int k = threadIdx.x, K = st->K;
    const double rxTime = st->rxTime, T = st->T;
    const int ds = st->dopplerSign;
    if (k == 0) sFlags = 0;
    DPE_EKF_STAMP(0);
    EkfPre ekfPre{};
    if (a.meas && a.ekf) ekfPre = ekf_dev_prefetch(a.ekf);
    Chan c;
    if (k < K) c = st->ch[k];
    double z[8];
    unsigned long long kp = 0ull, kv = 0ull;
    long long ip = 0, iv = 0;
    int measBad = 0;
    if (k == 63) {
        if (a.meas) {
            // ML grid points -> ECEF measurement about the grid centre and with the ENU matrix the scan of this window used
            // = the ports as the previous window's chm_k2 left them
            kp = a.keys[0]; kv = a.keys[1];
            ip = (long long)(0xFFFFFFFFu - (unsigned)(kp & 0xFFFFFFFFull)) - a.posOff;
            iv = (long long)(0xFFFFFFFFu - (unsigned)(kv & 0xFFFFFFFFull)) - a.velOff;
            if (kp == 0ull || ip < 0 || ip >= a.posG) { measBad = 4; ip = 0; }
            if (kv == 0ull || iv < 0 || iv >= a.velG) { measBad = 4; iv = 0; }
            const double *g = a.posGrid + 4 * ip, *v = a.velGrid + 4 * iv, *R = a.p.enu2ecef, *cc = a.p.xkk1;
            {
#pragma clang fp contract(off)   // (the host form's BCM_MakePosMeas / MakeVelMeas run without fused multiply-adds: the same doubles here)
                z[0] = R[0] * g[0] + R[1] * g[1] + R[2] * g[2] + cc[0];   // :1990-1999
                z[1] = R[3] * g[0] + R[4] * g[1] + R[5] * g[2] + cc[1];
                z[2] = R[6] * g[0] + R[7] * g[1] + R[8] * g[2] + cc[2];
                z[3] = g[3] + cc[3];
                z[4] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2] + cc[4];   // :2042-2051
                z[5] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2] + cc[5];
                z[6] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2] + cc[6];
                z[7] = v[3] + cc[7];
            }
            if (measBad) {   // no valid score this window: hold the state (and say so)
                for (int i = 0; i < 8; ++i) z[i] = cc[i];
                atomicOr(&sFlags, measBad);
            }
            for (int i = 0; i < 8; ++i) { sX1[i] = z[i]; sXk[i] = z[i]; sZ[i] = z[i]; a.p.zVal[i] = z[i]; }   // EKF_PassMeas: z to both state ports (below)
        }
    }
    if (a.meas && a.ekf) {   // EnableEKF = true: StepUpdate + StepPredict on the measurement (block-uniform branch; all 64 lanes work)
        __syncthreads();
        DPE_EKF_STAMP(1);
        const bool ok = sFlags == 0 && ekf_dev_step(a.ekf, ekfPre, sZ, sEkf, sPiv);
        __syncthreads();
        if (k < 8) {
            if (ok) { sX1[k] = sEkf[kEkfDevX1 + k]; sXk[k] = sEkf[kEkfDevXk + k]; }
            else { sX1[k] = a.p.xkk1[k]; sXk[k] = a.p.xkk1[k]; }   // (no measurement, or S singular: hold the predicted state)
        }
        if (!ok && k == 0 && sFlags == 0) { atomicOr(&sFlags, 32); a.ekf->failed = 1; }
        DPE_EKF_STAMP(8);
        __syncthreads();
    }
    if (k == 63) {
        if (a.meas) {
            for (int i = 0; i < 8; ++i) { z[i] = sX1[i]; a.p.xk1k1[i] = sX1[i]; a.p.xkk1[i] = sXk[i]; }   // the state ports (cuekf.cu:277-279)
            // the fix for the host: staged in device memory; the chm_k2 behind this kernel (which has time to spare) sends it over
            // the host link -- stores to the pinned ring and the wait for them cost this kernel, which the next window's
            // correlator waits for, ~2 us
            dpe_fix_record *r = a.stage;
            for (int i = 0; i < 8; ++i) r->zVal[i] = z[i];
            r->rxTime = rxTime;
            r->posIndex = (long long)(ip + a.posOff);
            r->velIndex = (long long)(iv + a.velOff);
            r->posOutOfWindow = (long long)a.keys[2];
            r->velOutOfWindow = (long long)a.keys[3];
#if defined(DPE_EXPERIMENTS) && defined(DPE_EKF_STAMPS)
            {   // eight phase lengths in units of 4 shader clocks, 16 bits each
                unsigned long long lo = 0ull, hi = 0ull;
                for (int i = 0; i < 4; ++i) lo |= (((dpe_ekf_stamps[i + 1] - dpe_ekf_stamps[i]) >> 2) & 0xFFFFull) << (16 * i);
                for (int i = 0; i < 4; ++i) hi |= (((dpe_ekf_stamps[i + 5] - dpe_ekf_stamps[i + 4]) >> 2) & 0xFFFFull) << (16 * i);
                r->posOutOfWindow = (long long)lo;
                r->velOutOfWindow = (long long)hi;
            }
#endif
            r->posScore = __uint_as_float((unsigned)(kp >> 32));
            r->velScore = __uint_as_float((unsigned)(kv >> 32));
            r->status = st->status | measBad | (sFlags & 32);
            r->seq = (unsigned long long)(st->window + 1);
        } else {
            for (int i = 0; i < 8; ++i) {
                const double x1 = a.xk1k1[i], xk = a.xkk1[i];
                sX1[i] = x1; a.p.xk1k1[i] = x1; a.p.xkk1[i] = xk;
            }
        }
    }
    __syncthreads();
    if (k < K && a.mode == 1) {   // measurement update (CHM_PropagateChannels :380-447)
        double sr[8], range;
        const double tau = rxTime - (c.txTime + (sX1[3] / kC)) + c.sat[3];
        rotate_state(c.sat, tau, sr);
        const double bcRc = back_calc_rc(c, sr, sX1, rxTime, &range);
        const double ex = sX1[4] - kOEDot * sX1[1], ey = sX1[5] + kOEDot * sX1[0], ez = sX1[6];
        const double lx = sr[0] - sX1[0], ly = sr[1] - sX1[1], lz = sr[2] - sX1[2];
        const double lrr = ((lx / range) * (ex - sr[4])) + ((ly / range) * (ey - sr[5])) + ((lz / range) * (ez - sr[6]));
        const double bcFi = kFL1 * ((lrr - sX1[7]) / kC + sr[7]) / ds;
        const double bcFc = kFCA + (ds * kFCA / kFL1) * bcFi + (bcRc - c.rcEnd) / T;
        c.fi = bcFi;
        c.fc = bcFc;
        st->ch[k].fi = bcFi;
        st->ch[k].fc = bcFc;
    }
    if (k < K) {   // what chm_k3 needs of the channel (it runs beside chm_k2, which rewrites ch[])
        st->snapFc[k] = c.fc; st->snapRcEnd[k] = c.rcEnd;
        st->snapCpElaEnd[k] = c.cpElaEnd; st->snapCpRef[k] = c.cpRef; st->snapCpRefTOW[k] = c.cpRefTOW;
    }
    if (k < K && a.bcsChan) {   // the next window starts where this one ended: rcEnd, riEnd, cpElaEnd with the new frequencies
        int bad;
        const BcsChanDev d = bcs_prep_one(c.rcEnd, c.riEnd, c.fc, c.fi, c.cpElaEnd, c.cpRef, c.prn, a.fs, a.S, bad);
        a.bcsChan[k] = d;
        if (a.hintL1 > 0 && hint_broken(d, a.hintL1, a.hintStepMax)) {   // (the prepared form launches no prep kernel: the promise is checked here)
            bad |= 8;
            __hip_atomic_store(a.hintViol, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (bad) atomicOr(&sFlags, bad << 3);
    }
    __syncthreads();
    if (k == 0) {
        if (sFlags) st->status |= sFlags;
        if (a.bcsStatus) {   // BatchCorrScores' input flags: bits 0, 1 and 3 replaced; bit 2 (the batch kernels' own) and the sticky bit 4 kept, as bcs_prep_kernel does
            const int v = (int)((sFlags >> 3) & 11u);
            const int old = *a.bcsStatus;          // (one thread, stream order: nothing else writes the word while this kernel runs)
            if ((old & ~11) != old || v) *a.bcsStatus = (old & ~11) | v;
        }
    }
}

__device__ __forceinline__ static void chm_k2(const ChmKArgs &a)
{
    __shared__ double sXk[8], sR[9];
    __shared__ double sTx[DPE_MAX_CHAN], sS1[DPE_MAX_CHAN][8], sSat[DPE_MAX_CHAN][8], sTau0[DPE_MAX_CHAN], sCt[DPE_MAX_CHAN], sSt[DPE_MAX_CHAN];
    __shared__ int sFlags;
    ChmDevState *st = a.st;
    const int tid = threadIdx.x, nThreads = blockDim.x, wave = tid >> 6, k = tid & 63, K = st->K, dimT = st->dimT;
    const double rxTime = st->rxTime, T = st->T;
    const int ds = st->dopplerSign;
    const bool live = wave < 2 && k < K;
    if (tid == 0) sFlags = 0;
    if (tid < 8) sXk[tid] = a.p.xkk1[tid];
    // the fix record chm_k1 staged: its fields go out over the host link now (one otherwise idle lane), the sequence word at the end
    // of this kernel, when they have long arrived
    dpe_fix_record *ringRec = nullptr;
    unsigned long long ringSeq = 0ull;
    if (a.meas && tid == nThreads - 1) {
        const dpe_fix_record rec = *a.stage;
        ringSeq = rec.seq;
        ringRec = a.ring + ((long long)(ringSeq - 1ull) % a.ringDepth);
        for (int i = 0; i < 8; ++i) __hip_atomic_store(&ringRec->zVal[i], rec.zVal[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ringRec->rxTime, rec.rxTime, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ringRec->posIndex, rec.posIndex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ringRec->velIndex, rec.velIndex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ringRec->posOutOfWindow, rec.posOutOfWindow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ringRec->velOutOfWindow, rec.velOutOfWindow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ringRec->posScore, rec.posScore, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ringRec->velScore, rec.velScore, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ringRec->status, rec.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    Chan c;
    double wE = 0.0, wM = 0.0;
    int wOk = 0;
    if (live) {
        c = st->ch[k];
        wE = st->warmE[k]; wM = st->warmM[k]; wOk = st->warmOk[k];
    }
    __syncthreads();
    // ---- 1. predicted transmit time (CHM_TimeUpdateChannels :675-823); at Start first CHM_ComputeSatStates :258-301
    double x1[8];
    for (int i = 0; i < 8; ++i) x1[i] = a.p.xk1k1[i];
    double txPred = 0.0;
    if (wave == 0 && live) {
        if (a.mode == 0) {
            c.txTime = tx_of(c, c.cpElaEnd, c.rcEnd);
            if (sat_state_warm(c.eph, c.txTime, c.sat, wE, wM, wOk)) atomicOr(&sFlags, 1);
        }
        const double adv = c.fc * T + c.rcEnd;
        const double cpPred = c.cpElaEnd + floor(adv / kLCA);
        const double rcPred = wrap_pos(adv, (double)kLCA);
        txPred = tx_of(c, cpPred, rcPred);
        sTx[k] = txPred;
    }
    __syncthreads();
    // ---- 2. the satellite state at the predicted transmit time (wave 0) and a step later (wave 1), side by side; wave 2: the
    //         ENU -> ECEF matrix of the next grid centre (CHM_Dev_R_ENU2ECEF :54-73)
    // A chm_k3 one window earlier has left both states at a transmit time within ~1e-7 s of this one (it could not know this window's
    // fix): they are advanced along their own difference quotient -- the remainder, a delta^2 / 2 with a = 0.6 m/s^2, is 1e-14 m --
    // and the two Kepler solutions, 5 us of a lone wave's fp64 libm calls, are off this kernel's path.  Without a usable
    // speculation (Start, a standalone launch, a jump of the code phase) the solutions are computed here as before.
    double sp[8];
    if (live) {
        const int sb = a.specPar & 1;
        const double dSpec = sTx[k] - st->specTx[sb][k];
        if (a.mode == 1 && st->specOk[sb][k] && fabs(dSpec) < 1e-5) {
            if (wave == 0) {
                for (int i = 0; i < 8; ++i) {
                    const double s0 = st->specSat[sb][0][k][i], s1 = st->specSat[sb][1][k][i];
                    const double adv = (s1 - s0) / kChmH * dSpec;
                    sp[i] = s0 + adv;
                    sS1[k][i] = s1 + adv;
                }
            }
        } else {
            const double tx = sTx[k] + (wave == 1 ? kChmH : 0.0);
            if (sat_state_warm(c.eph, tx, sp, wE, wM, wOk)) atomicOr(&sFlags, 1);
            if (wave == 1)
                for (int i = 0; i < 8; ++i) sS1[k][i] = sp[i];
        }
    }
    if (tid == 128) {
        double Rm[9];
        enu2ecef_matrix(sXk, Rm);
        for (int i = 0; i < 9; ++i) { sR[i] = Rm[i]; a.p.enu2ecef[i] = Rm[i]; }
        a.p.rxTime[0] = rxTime + T;
        a.p.dopplerSign[0] = ds;
        if (a.devWin) {   // the window frame dpe_bcm_results would read (pinned): sent now, it has arrived long before the kernel ends
            for (int i = 0; i < 8; ++i) a.devWin->xCurrkk1[i] = sXk[i];
            for (int i = 0; i < 9; ++i) a.devWin->enu2ecef[i] = Rm[i];
            a.devWin->dopplerSign = ds;
            a.devWin->bad = 0;
        }
    }
    __syncthreads();
    // ---- 3. wave 0: the rest of the time update; the state at the back-calculated transmit time along the difference quotient
    if (wave == 0 && live) {
        double sr[8];
        const double tau = rxTime + T - (txPred + (x1[3] / kC)) + sp[3];
        rotate_state(sp, tau, sr);
        const double bcRc = back_calc_rc(c, sr, x1, rxTime + T, nullptr);
        c.cpElaStart = c.cpElaEnd;
        c.rcStart = c.rcEnd;
        c.cpElaEnd += floor(bcRc / kLCA);
        c.rcEnd = wrap_pos(bcRc, (double)kLCA);
        c.riStart = c.riEnd;
        c.riEnd = wrap_pos(c.fi * T + c.riEnd, 1.0);
        c.txTime = tx_of(c, c.cpElaEnd, c.rcEnd);
        const double dt = c.txTime - txPred;
        for (int i = 0; i < 8; ++i) c.sat[i] = fma((sS1[k][i] - sp[i]) / kChmH, dt, sp[i]);
        st->ch[k] = c;
        st->warmE[k] = wE; st->warmM[k] = wM; st->warmOk[k] = wOk;
        // ports (cuchanmgr.cu:1136-1171)
        a.p.txTime[k] = c.txTime;
        a.p.rcStart[k] = c.rcStart; a.p.riStart[k] = c.riStart;
        a.p.rcEnd[k] = c.rcEnd;     a.p.riEnd[k] = c.riEnd;
        a.p.fc[k] = c.fc;           a.p.fi[k] = c.fi;
        a.p.cpRef[k] = c.cpRef;     a.p.cpElaStart[k] = c.cpElaStart;
        a.p.cpElaEnd[k] = c.cpElaEnd; a.p.cpRefTOW[k] = c.cpRefTOW;
        a.p.prn[k] = (unsigned char)c.prn;
        // CHM_GridPrep :892-916 at the new receive time: the mid-time rotation here, the K x dimT entries by all threads below
        const double tau0 = rxTime + T - (c.txTime + ((a.p.timeGrid[dimT / 2] + sXk[3]) / kC)) + c.sat[3];
        double ct0, st0;
        sincos(-kOEDot * tau0, &st0, &ct0);
        sCt[k] = ct0; sSt[k] = st0; sTau0[k] = tau0;
        for (int i = 0; i < 8; ++i) sSat[k][i] = c.sat[i];
        if (a.svPos) {
            double mid[8];
            rotate_state_cs(c.sat, ct0, st0, mid);
            BcmSvDev ap, av;
            bcm_prep_one(sXk, sR, mid, c.rcEnd, c.fc, c.fi, c.cpRefTOW, c.cpElaEnd, c.cpRef, ds, rxTime + T, a.fs, a.Cf, a.S, a.L, a.B, a.C, ap, av);
            a.svPos[k] = ap;
            a.svVel[k] = av;
        }
    }
    __syncthreads();
    // ---- 4. batch satellite states, entry (channel, time-grid index) per thread (see batch_states: the mid entry takes the
    //         reference's own evaluation, the others its first-order neighbourhood)
    for (int e = tid; e < K * dimT; e += nThreads) {
        const int kk = e / dimT, t = e - kk * dimT;
        double *o = a.p.sat + (size_t)e * 8;
        const double ct0 = sCt[kk], st0 = sSt[kk];
        if (t == dimT / 2) { rotate_state_cs(sSat[kk], ct0, st0, o); continue; }
        const double dTau = -(a.p.timeGrid[t] - a.p.timeGrid[dimT / 2]) / kC;   // tau - tau0
        const double d = -kOEDot * dTau;
        if (fabs(d) < 1e-8) rotate_state_cs(sSat[kk], ct0 - d * st0, st0 + d * ct0, o);
        else rotate_state(sSat[kk], sTau0[kk] + dTau, o);
    }
    if (tid == 0) {
        st->rxTime = rxTime + T;   // :1121, :1249
        if (a.mode == 1) st->window += 1;
        if (sFlags) st->status |= sFlags;
    }
    if (ringRec) {   // the record is complete once its fields have arrived: sequence word last
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&ringRec->seq, ringSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// chm_k3 (two waves, beside chm_k2 in the stage-1 launch): the satellite states the NEXT time update will ask for.  That update's
// transmit time follows from the channel as chm_k2 is leaving it right now and from the next fix; predicted from chm_k1's snapshot
// -- two windows of code-phase advance at the present code frequency -- it is off by the next fix's correction, ~1e-7 s.
__device__ __forceinline__ static void chm_k3(const ChmKArgs &a)
{
    ChmDevState *st = a.st;
    const int tid = threadIdx.x, wave = tid >> 6, k = tid & 63, K = st->K;
    const bool active = wave < 2 && k < K;
    const int wb = (a.specPar & 1) ^ 1;
    int bad = 0;
    if (active) {
        const double adv2 = 2.0 * (st->snapFc[k] * st->T) + st->snapRcEnd[k];
        const double cp2 = st->snapCpElaEnd[k] + floor(adv2 / kLCA), rc2 = wrap_pos(adv2, (double)kLCA);
        const double tx0 = st->snapCpRefTOW[k] + ((cp2 - st->snapCpRef[k]) * kTCA) + (rc2 / kFCA);
        double wE = st->specWE[wave][k], wM = st->specWM[wave][k];
        int wOk = st->specWOk[wave][k];
        double out[8];
        bad = sat_state_warm(st->ephC[k], tx0 + (wave == 1 ? kChmH : 0.0), out, wE, wM, wOk);
        for (int i = 0; i < 8; ++i) st->specSat[wb][wave][k][i] = out[i];
        st->specWE[wave][k] = wE; st->specWM[wave][k] = wM; st->specWOk[wave][k] = wOk;
        if (wave == 0) { st->specTx[wb][k] = tx0; st->specOk[wb][k] = bad ? 0 : 1; }
    }
    __syncthreads();
    if (active && wave == 1 && bad) st->specOk[wb][k] = 0;   // (behind wave 0's store: both waves of a channel run in this block)
}

__global__ __launch_bounds__(64) static void chm_k1_kernel(ChmKArgs a) { chm_k1(a); }
__global__ __launch_bounds__(256) static void chm_k2_kernel(ChmKArgs a) { chm_k2(a); }

}  // namespace dpe
#endif  // __HIPCC__
