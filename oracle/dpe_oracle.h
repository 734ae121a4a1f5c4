/*
 * dpe_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 *
 * Plain-C fp64 restatement of the sampleblock -> BatchCorrScores -> BatchCorrManifold
 * hot path of Stanford-NavLab/NavLab-DPE-SDR (CUDARecv semantics), plus the cuChanMgr
 * side inputs that feed it.  Every function cites the reference file:line it restates.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (navlab-dpe-sdr_amd/) never links it.
 *
 * PARITY PIN: the reference ships no tests/golden vectors and the
 * CUDA build needs nvcc/cuFFT (unbuildable here).  The oracle is therefore pinned against
 * fixtures generated in the dev container by importing the reference's Python twin
 * (pygnss, via a scratch lib2to3 conversion) -- tests/golden/make_golden.py -- which the
 * CUDA authors themselves used as their oracle.
 */
#ifndef DPE_ORACLE_H_
#define DPE_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* constants: cudarecv/utils/inc/consthelper.h:5-27 (bit-for-bit) */
#define DPO_C        (299792458.0)
#define DPO_PI       (3.1415926535898)
#define DPO_2PI      (6.2831853071796)
#define DPO_F_L1     (1.57542e9)
#define DPO_F_CA     (1.023e6)
#define DPO_L_CA     (1023)
#define DPO_T_CA     (0.001)
#define DPO_PRN_MAX  (37)
#define DPO_MU       (3.9860050e14)   /* ephhelper.h MU_GPS */
#define DPO_F        (-4.442807633e-10)
#define DPO_OEDOT    (7.2921151467e-5)
#define DPO_WGS84_A  (6378137.0)
#define DPO_WGS84_B  (6356752.314245)
#define DPO_WGS84_E  (0.08181919084262149)
#define DPO_WGS84_EP (0.08209443794969568)

/* ephemeris as a flat double[DPO_EPH_N] (subset of eph_t, ephhelper.h:98-125) */
enum {
    DPO_EPH_SQRT_A = 0, DPO_EPH_E, DPO_EPH_I0, DPO_EPH_OMG0, DPO_EPH_OMG, DPO_EPH_M0,
    DPO_EPH_DELN, DPO_EPH_OMGD, DPO_EPH_IDOT, DPO_EPH_CRC, DPO_EPH_CRS, DPO_EPH_CUC,
    DPO_EPH_CUS, DPO_EPH_CIC, DPO_EPH_CIS, DPO_EPH_TOES, DPO_EPH_TOCS, DPO_EPH_F0,
    DPO_EPH_F1, DPO_EPH_F2, DPO_EPH_TGD, DPO_EPH_N
};

/* batchcorrscores.cu:117-177 BCS_GenCACode (== correlator.py:474-515) */
void dpo_gen_ca_code(int prn, int8_t chips[1023]);

/* batchcorrscores.cu:237-258 BCS_NavBitBoundary */
int dpo_nav_bit_boundary(int cpElapsed, int cpReference, double codePhase,
                         double codeFreq, double fs);

/* batchcorrscores.cu:185-196 BCS_GenTimeIdcs */
double dpo_time_idx(int64_t i, double fs);

/*
 * One SV of BatchCorrScores::Update (batchcorrscores.cu:1043-1180), direct-sum form.
 * Produces the fft-shifted CodeScores entries for lags [lagLo,lagHi] (array index
 * S/2+lag in the reference's row) and the fft-shifted CarrScores entries for bins
 * [binLo,binHi] (array index C/2+bin).  Interleaved re,im doubles.
 * info[0]=idxNextNavBit, info[1]=noFlipIsLarger; meanOut = DC mean (re,im).
 * meanIn: if non-NULL use this mean instead of computing it (mean is per window).
 */
int dpo_bcs_sv(const int16_t *iq, int S, double fs, const int8_t *chips,
               double rc, double ri, double fc, double fi, int cpElapsed, int cpReference,
               int lagLo, int lagHi, int64_t C, int binLo, int binHi,
               double *code, double *carr, int *info, double *meanOut);

/*
 * batchcorrmanifold.cu:1710-1828 BCM_PosMeasML.  Score rows are given as windows:
 * row k covers the reference's absolute in-row indices [winLo, winLo+winLen) of
 * codeScores[k*S ...]; an access outside the window (or outside (0,S), which the
 * reference leaves undefined, :1795-1804) contributes 0 and bumps *oob.
 * grid: G x 4 doubles (x,y,z,delta_t ENU offsets).  satStates: K x 8 (mid-time state).
 * LPower < 0: evaluate the index in long double (exponent |LPower|) -- not the reference's
 * arithmetic, only used to measure the reference's own fp64 cancellation noise.
 */
int64_t dpo_bcm_pos_quirks(int64_t *idx, int64_t max);   /* see dpe_oracle.c: double-counted points of the last call */
int dpo_bcm_pos(const double *satStates, const double *codeWin, int winLo, int winLen,
                const double *centerPt, const double *grid, int64_t G,
                const double *enu2ecef, const double *codeFreq, const int *cpRefTOW,
                const int *cpElapsedEnd, const int *cpRef, const double *codePhase,
                double rxTime, int K, double fs, int numSamps, int LPower,
                double *scores, int64_t *oob);

/* batchcorrmanifold.cu:1861-1963 BCM_VelMeasML (windowed rows as above, over C bins) */
int dpo_bcm_vel(const double *satStates, const double *carrWin, int64_t winLo, int winLen,
                const double *centerPt, const double *grid, int64_t G,
                const double *enu2ecef, const double *carrFreq, double rxTime, int K,
                double fs, int64_t numfftPts, int dopplerSign, int LPower,
                double *scores, int64_t *oob);

/* thrust::max_element semantics: first maximum (batchcorrmanifold.cu:2589-2590) */
int64_t dpo_argmax_first(const double *v, int64_t n);

/* batchcorrmanifold.cu:1977-2068 BCM_MakePosMeas / BCM_MakeVelMeas */
void dpo_make_meas(int64_t posIdx, int64_t velIdx, const double *centerPt,
                   const double *posGrid, const double *velGrid, const double *enu2ecef,
                   double zVal[8], double RVal[64]);

/* batchcorrmanifold.cu:148-255 BCM_InitPosGrid (gridType 0=Uniform, 2=ArthurBasis),
 * dims[4], spacing[4]; t fastest.  timeGrid (dims[3]) may be NULL. */
void dpo_init_grid(int gridType, const int dims[4], const double spacing[4],
                   double *grid, double *timeGrid);

/* cuchanmgr.cu:85-210 CHM_Get_Sat_Pos.  returns 0 ok, -1 no Kepler convergence */
int dpo_sat_pos(const double eph[DPO_EPH_N], double txTime, double state[8]);

/* cuchanmgr.cu:36-73 ECEF->lat/lon (rad) and the row-major ENU->ECEF matrix */
void dpo_ecef2ll(const double posECEF[3], double ll[2]);
void dpo_enu2ecef(const double ll[2], double R[9]);

/* Channel state block used by the chanmgr restatement (arrays of length K) */
typedef struct {
    int K;
    double *rcStart, *rcEnd, *riStart, *riEnd, *fc, *fi, *txTime;
    int *cpElaStart, *cpElaEnd, *cpRef, *cpRefTOW;
    double *satStates;        /* K x 8 */
    const double *eph;        /* K x DPO_EPH_N */
    int dopplerSign;
} dpo_chan_t;

/* cuchanmgr.cu:240-306 CHM_ComputeSatStates (ephemeris already selected per SV) */
void dpo_chm_compute_sat_states(dpo_chan_t *ch);
/* cuchanmgr.cu:641-829 CHM_TimeUpdateChannels */
void dpo_chm_time_update(dpo_chan_t *ch, const double *centerPt, double rxTime, double T);
/* cuchanmgr.cu:338-608 CHM_PropagateChannels */
void dpo_chm_propagate(dpo_chan_t *ch, const double *centerPt, double rxTime, double T);
/* cuchanmgr.cu:853-923 CHM_GridPrep: batch K x dimT x 8, R[9] */
void dpo_chm_grid_prep(double rxTime, const double *txTime, const double *centerPt,
                       const double *satStates, int K, const double *timeGrid, int dimT,
                       double *batchSatStates, double *enu2ecef);

#ifdef __cplusplus
}
#endif
#endif
