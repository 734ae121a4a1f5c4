// dpe_prep.h -- per-channel constants of the two hot-path modules, derived ON THE DEVICE from the reference's port arrays.
//
// Shared by the one-block prep kernels of dpe_bcs_update_dev / dpe_bcm_update_dev (the ports of a host that keeps the reference's
// own cuChanMgr / cuEKF) and by the device-resident channel manager (dpe_chanmgr.hip), whose per-window kernel writes the same
// blocks itself so that no prep launch sits between it and the next window's kernels.  fp64, expression for expression what the
// host loops of dpe_bcs_update / dpe_bcm_update compute (fp contraction off: a fused multiply-add would round differently).
#pragma once
#include "dpe_common.h"

namespace dpe {

struct BcsChanDev {
    double rc;        // code phase at sample 0 (chips)
    double codeStep;  // chips per sample  = fc / fs
    double ri;        // carrier phase at sample 0 (cycles)
    double carrStep;  // cycles per sample = fi / fs
    float rotRe, rotIm;  // exp(-j 2 pi carrStep)
    int32_t idxNext;  // BCS_NavBitBoundary
    int32_t hasFlip;  // 0 < idxNext < S
    int32_t prn;
    int32_t pad;
    double fc, fi;    // raw code / carrier frequency (time-table mode)
    double invStep;   // samples per chip = fs / fc (chip-boundary kernel)
};
static_assert(sizeof(BcsChanDev) % 16 == 0, "the parameter upload copies 16-byte words");

struct BcmSvDev {
    float ue, un, uu;  // unit line of sight (receiver -> SV) in the ENU frame of the grid
    float g;           // bank entries per metre of (delta_t + d_rho)   [vel: -(entries per m/s)]
    float h;           // position manifold: 1 / (2 range)
    float idx0;        // bank-relative fractional index at the grid centre
    float pad0, pad1;
    // velocity manifold (no second-order term): {h, pad0, pad1} = g {ue, un, uu}, so that the index is four chained FMAs
    // idx0 + g dt - (g ue) dx - (g un) dy - (g uu) dz instead of dot product, difference, scale
};

struct BcmDevWin {   // pinned mirror of the window inputs of a device-parameter Update
    double xCurrkk1[8], enu2ecef[9];
    int dopplerSign, bad;
};

#ifdef __HIPCC__
#pragma clang fp contract(off)
// BatchCorrScores, one channel: batchcorrscores.cu:237-258 (nav-bit boundary) and the per-sample steps of :277-305, :323-372.
// bad: bit 0 = PRN outside 1..37 (clamped), bit 1 = non-positive code frequency / negative code phase (nominal values
// substituted so that the kernels' chip-table indices stay inside the table; the flag is what the caller reports).
__device__ __forceinline__ BcsChanDev bcs_prep_one(double rc, double ri, double fc, double fi, int cpEla, int cpRef, int prn, double fs, int S, int &bad)
{
    BcsChanDev d;
    bad = 0;
    if (prn < 1 || prn > kPrnMax) { bad |= 1; prn = prn < 1 ? 1 : kPrnMax; }
    if (!(fc > 0.0) || !(rc >= 0.0)) {
        bad |= 2;
        if (!(fc > 0.0)) fc = kFCA;
        if (!(rc >= 0.0)) rc = 0.0;
    }
    d.rc = rc;
    d.codeStep = fc / fs;
    d.ri = ri;
    d.carrStep = fi / fs;
    d.fc = fc;
    d.fi = fi;
    d.invStep = fs / fc;
    const double ang = -6.283185307179586476925286766559 * d.carrStep;
    double sa, ca;
    sincos(ang, &sa, &ca);   // (one argument reduction; the host's std::cos / std::sin give the same floats)
    d.rotRe = (float)ca;
    d.rotIm = (float)sa;
    const int since = (((cpEla - cpRef) % 20) + 20) % 20;                                   // BCS_NavBitBoundary :247-253
    d.idxNext = (int)(floor((kLCA * (20 - since) - rc) * (fs / fc)) + 1);
    d.hasFlip = (d.idxNext > 0 && d.idxNext < S) ? 1 : 0;
    d.prn = prn;
    d.pad = 0;
    return d;
}

// BatchCorrManifold, one channel: the expansion coefficients of both manifolds about the grid centre
// (batchcorrmanifold.cu:1779-1791, 1917-1936).  c = xCurrkk1 [8], R = ENU2ECEFMat [9] row-major, s = the SV's mid-time batch
// state [8].  The host form carries the centre index in long double because rxTime - pr / C (rxTime ~ 4e5 s) rounds at 5.8e-11 s
// in fp64; here the same difference is kept as an unevaluated sum (TwoSum), which is more than the 64-bit significand gives.
__device__ __forceinline__ void bcm_prep_one(const double *c, const double *R, const double *s, double rcEnd, double fck, double fik, int cpRefTOW,
                                             int cpElaEnd, int cpRef, int ds, double rxTime, double fs, double Cf, int S, int L, int B, long long C,
                                             BcmSvDev &a, BcmSvDev &v)
{
    const double dx = s[0] - c[0], dy = s[1] - c[1], dz = s[2] - c[2];                    // :1779-1781
    const double range = sqrt(dx * dx + dy * dy + dz * dz);                               // :1782
    const double ux = dx / range, uy = dy / range, uz = dz / range;
    const double ue = R[0] * ux + R[3] * uy + R[6] * uz;                                  // R^T u
    const double un = R[1] * ux + R[4] * uy + R[7] * uz;
    const double uu = R[2] * ux + R[5] * uy + R[8] * uz;
    // position manifold, centre index (:1783-1791)
    const double pr = range - kC * s[3] + c[3];
    const double t = pr / kC;
    const double hi = rxTime - t, bb = hi - rxTime;
    double lo = (rxTime - (hi - bb)) + (-t - bb);                                         // rxTime - t = hi + lo exactly
    const double d1 = hi - (double)cpRefTOW;                                              // exact: both are multiples of ulp(rxTime)
    const double n = (double)(cpElaEnd - cpRef);
    const double pn = n * kTCA;
    lo -= fma(n, kTCA, -pn);                                                              // the product's own rounding
    const double cfd = (d1 - pn) + lo;
    const double rc0 = cfd * kFCA - rcEnd;
    const double basePos = (fs / fck) * (-rc0) + (double)S / 2.0;
    a.ue = (float)ue; a.un = (float)un; a.uu = (float)uu;
    a.g = (float)(fs * kFCA / (fck * kC));
    a.h = (float)(0.5 / range);
    a.idx0 = (float)(basePos - (double)(S / 2 - L));
    a.pad0 = a.pad1 = 0.f;
    // velocity manifold, centre index (:1917-1936)
    const double ex = c[4] - kOEDot * c[1], ey = c[5] + kOEDot * c[0], ez = c[6];
    const double lrr = ux * (ex - s[4]) + uy * (ey - s[5]) + uz * (ez - s[6]);
    const double fbc = kFL1 * ((lrr - c[7]) / kC + s[7]) / ds;
    const double baseVel = (Cf / fs) * (fbc - fik) + Cf / 2.0;
    const double gv = (Cf / fs) * kFL1 / (kC * ds);
    v.ue = (float)ue; v.un = (float)un; v.uu = (float)uu;
    v.g = (float)(-gv);
    v.idx0 = (float)(baseVel - (double)(C / 2 - B));
    v.h = (float)(-gv * ue); v.pad0 = (float)(-gv * un); v.pad1 = (float)(-gv * uu);
}
#pragma clang fp contract(fast)
#endif  // __HIPCC__

}  // namespace dpe

// ---- hooks between the translation units (not part of include/dpe_hip.h: the device-resident channel manager writes the
// parameter blocks of the handles it is attached to, and reads the scan's keys and the fp64 grids for the measurement)
struct dpe_bcs;
struct dpe_bcm;
extern "C" {
struct dpe_bcs_hook {
    dpe::BcsChanDev *chan_d;   // [maxWindows][maxChannels]; window 0 is the single-window block
    int *status_d;
    double fs;
    int S, maxChannels;
    // dpe_bcs_set_dev_hint in force (hintL1 > 0): whoever writes the parameter block checks the promise (hint_broken below), flags it
    // in the status word (bit 3) and raises *hintViol (pinned, device address): the handle drops the hint at its next call
    int hintL1;
    double hintStepMax;
    int *hintViol;
};
// the conditions behind DPE_DEV_HINT_CHIP for one channel: chips of hintL1 or hintL1 + 1 samples, code step within the assumed bound,
// carrier offset inside the chip kernels' closed-form DC term, nav-bit boundary on a chip boundary of the replica
__host__ __device__ inline bool hint_broken(const dpe::BcsChanDev &d, int hintL1, double hintStepMax)
{
    const bool offBoundary = d.hasFlip && (int)fma((double)d.idxNext, d.codeStep, d.rc) == (int)fma((double)(d.idxNext - 1), d.codeStep, d.rc);
    return (int)d.invStep != hintL1 || d.codeStep > hintStepMax || 6.283185307179586 * fabs(d.fi) > 0.25 * d.fc || offBoundary;
}
int dpe_bcs_hook_get(dpe_bcs *h, dpe_bcs_hook *out);
// Destruction in any order: a device-resident channel manager attached to a handle registers itself here, and the handle's destroy
// calls `detach(owner, which)` FIRST (which = 0 BatchCorrScores, 1 BatchCorrManifold), while the handle is still whole -- the manager
// runs what it parked there and forgets the handle.  owner = nullptr unregisters (the manager's own destroy).
typedef void (*dpe_owner_detach_fn)(void *owner, int which);
int dpe_bcs_hook_set_owner(dpe_bcs *h, dpe_owner_detach_fn detach, void *owner);
int dpe_bcm_hook_set_owner(dpe_bcm *h, dpe_owner_detach_fn detach, void *owner);
// The device-resident channel manager's time update (chm_k2, dpe_chm_dev.h) as a task for this handle's next stage-1 launch:
// `args` = a dpe::ChmKArgs; the launch carries it as an extra block when its kernel form can (single-window bcs_bank_kernel),
// otherwise -- and from dpe_bcs_cotask_flush -- it runs as a kernel of its own first.
int dpe_bcs_cotask_set(dpe_bcs *h, const void *args, size_t bytes);
int dpe_bcs_cotask_flush(dpe_bcs *h, void *stream);
struct dpe_bcm_hook {
    dpe::BcmSvDev *svPos_d, *svVel_d;          // [maxWindows][maxChannels] each; window 0 is the single-window block
    dpe::BcmDevWin *devWin_hd;                 // pinned window frame (device address) the NEXT Update's results will be decoded with
    const unsigned long long *keys_d[2];       // the two alternating key sets: {pos, vel} keys then {pos, vel} out-of-window counts per window
    const double *posGrid64_d, *velGrid64_d;   // fp64 copies of the local grids (made by the first call of dpe_bcm_hook_get)
    long long posG, velG, posOffset, velOffset;
    double fs, Cf;
    int S, L, B, maxWindows, maxChannels;
    long long C;
};
int dpe_bcm_hook_get(dpe_bcm *h, dpe_bcm_hook *out);
// dpe_bcm_create for a further lane of a dpe_pipe (dpe_pipe.hip): the fp32 device grids are donor's (which must outlive the handle)
int dpe_bcm_create_sharing(const dpe_bcm_config *cfg, dpe_bcm *donor, dpe_bcm **out);
// enable = 0: the device-parameter Updates of this handle leave keys and counts in device memory only (no ticket, no stores over the
// host link at the end of the scan); dpe_bcm_results then fetches them with a copy.  Set by dpe_chm_dev_attach.
int dpe_bcm_hook_set_publish(dpe_bcm *h, int enable);
// referencePair with dpe_bcm_update_prepared: the attached channel manager's port arrays and its rxTime port, for the fp64 re-evaluation
// of the affected grid points (ports = nullptr: detached)
int dpe_bcm_hook_set_ref_ports(dpe_bcm *h, const dpe_bcm_ports_dev *ports, const double *rxTime_dev);
}
