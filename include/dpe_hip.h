/*
 * dpe_hip.h -- C-ABI of the MI355X-native DPE correlator engine (libdpe_hip.so).
 *
 * Drop-in boundary for the sampleblock -> BatchCorrScores -> BatchCorrManifold path of
 * Stanford-NavLab/NavLab-DPE-SDR (cudarecv).  Plain pointers, sizes and int status codes
 * (0 ok, -1 failure -- the reference's convention, cudarecv/auxil/inc/errorhandler.h:43-45,
 * 100-102); no C++ or torch types.  dpe_last_error() returns the message the reference
 * would have printed to std::cerr.
 *
 * Differences from the reference interface that are deliberate:
 *  - lengths are 32/64-bit (reference: unsigned short, dsp.h:109, sampleblock.h:81);
 *  - several windows ("batch") can be processed per call, each with its own channel state;
 *  - BatchCorrScores produces WINDOWED score banks (code lags [-L,+L] about the fftshift
 *    centre S/2, Doppler bins [-B,+B] about C/2) in fp32 instead of the dense K*S / K*C
 *    complex128 arrays; dpe_bcs_export_dense() writes the reference layout on request;
 *  - per-channel parameters are passed from HOST memory by dpe_bcs_update / dpe_bcm_update (any number of windows per
 *    call); dpe_bcs_update_dev / dpe_bcm_update_dev take them, for one window, from the DEVICE arrays the reference's
 *    cuChanMgr kernels write (dpeflow.cpp:169-191).
 */
#ifndef DPE_HIP_H_
#define DPE_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic environment switches, read ONCE when a handle is created and never on the Update path (all default off;
 * results stay within the stated tolerances, only kernel selection / tile shapes change).  Each selects another of the forms the
 * parity tests hold against the oracle and against each other:  DPE_BCS_NO_BANK16=1 (batches use the single-window bank kernel),
 * DPE_BCS_NO_CHIP=1 / DPE_BCS_NO_CHIP2=1 (no chip-boundary kernel / not its second form), DPE_BCS_NO_FUSE=1 (single windows run the
 * separate DC-sum kernel), DPE_BCS_NO_SUMRIDE=1 / DPE_BCS_SUMRIDE_MIN=n / DPE_BCS_RIDE_SPIN=n (DC sums of chip2 batches: separate
 * kernel / smallest riding batch / polls before a block sums its window itself), DPE_BCS_FORCE_FFT=1 (full-length FFT form),
 * DPE_BCS_TPB16=n / DPE_BCS_CHIP_TPB=n / DPE_BCS_CHIP2_P=n (tiles or passes per block of the batch / chip kernels),
 * DPE_BCM_NO_POLL=1 (dpe_bcm_results always waits for the stream), DPE_ACQ_NO_FUSED=1 (the acquisition searches keep the rocFFT
 * chain instead of the fused transform kernels), DPE_ACQ_NO_PACK=1 / DPE_ACQ_NO_FWD_PACK=1 (non-coherent search: radix-10 kernel +
 * four-pass transforms / the forward transform left to rocFFT), DPE_COMM_TIMEOUT_S (rendezvous time-out).  Switches whose A/B is
 * settled (DPE_BCS_NO_WIDE, DPE_BCS_RIDE_LA, DPE_ACQ_NO_FUSED_FWD, DPE_ACQ_STATS_LDS) and the ablations that skip work
 * (DPE_BCS_CHIP_DBG, DPE_BCS_FAT, DPE_BCM_SPLIT) exist only in builds with -DDPE_EXPERIMENTS. */
#define DPE_MAX_CHAN 37            /* CONST_PRN_MAX, consthelper.h:13 */
#define DPE_MAX_LAG_HALF_WIDTH 292  /* widest code-lag bank of the windowed stage-1 kernels: +-32 and four 65-lag chunks per side */
#define DPE_ABI_VERSION 4   /* 2: dpe_bcm_config.referencePair, pitched score rows, sizes in dpe_bcm_results_from_keys, *_update_dev;
                             * 3: the dpe_chm_dev_ family -- device-resident cuChanMgr + measurement hand-over -- and the _update_prepared forms;
                             * 4: dpe_chm_dev_set_shard, dpe_chm_dev_set_ekf; dpe_bcs_dev_status bit 2 is per launch (was sticky), bit 4 the
                             *    sticky one; dpe_fix_record.status bits 32 / 64; the dpe_pipe_ family (batches in flight) */

typedef void *dpe_stream_t;        /* hipStream_t (reference: cudaStream_t* flow stream, module.h:23) */
typedef struct dpe_bcs dpe_bcs;    /* opaque: one BatchCorrScores instance */
typedef struct dpe_bcm dpe_bcm;    /* opaque: one BatchCorrManifold instance */

/* ------------------------------------------------------------------ general -------- */
int dpe_abi_version(void);
const char *dpe_last_error(void);                 /* thread-local message of the last failure */
int dpe_device_info(char *name, int nameLen, int *cuCount, int64_t *hbmBytes);
/* hipSetDevice for the calling thread (one process per GPU: call before anything allocates). */
int dpe_set_device(int32_t device);
/* Measured HBM ceiling of this device (SURVEY 8d "verify on the box"): a float4 stream copy (2 arrays) and a
 * triad a = b + s*c (3 arrays) over `bytesPerArray`-byte arrays, `iters` timed launches each after two untimed ones;
 * GB/s = bytes read + written per launch / launch time.  Diagnostic only -- nothing on the hot path calls it. */
int dpe_hbm_ceiling(int64_t bytesPerArray, int iters, dpe_stream_t stream, double *copyGBs, double *triadGBs);

/* 37 x 1023 C/A chips (+1/-1), PRN p at row p-1.  Replaces BCS_GenCACode
 * (batchcorrscores.cu:117-177); unlike the reference, PRN 37 is generated too. */
int dpe_gen_ca_code(int8_t *chips /* [37*1023] host */);

/* ------------------------------------------------------------------ SampleBlock ---- */
/* Replaces the device side of dsp::SampleBlock (sampleblock.cu:356-410,465-515): copies one
 * block of interleaved little-endian int16 I/Q (4 bytes/sample) from host to a device
 * buffer on `stream` (asynchronous when `src` is page-locked, e.g. from dpe_host_alloc_pinned;
 * reference: cudaMallocHost). */
int dpe_sampleblock_upload(int16_t *dst_dev, const int16_t *src_host, int64_t nSamples,
                           dpe_stream_t stream);
int dpe_host_alloc_pinned(void **ptr, int64_t bytes);
int dpe_host_free_pinned(void *ptr);
/* Plain device-memory plumbing for hosts that do not bring their own allocator. */
int dpe_device_alloc(void **ptr_dev, int64_t bytes);
int dpe_device_free(void *ptr_dev);
int dpe_memcpy_h2d(void *dst_dev, const void *src_host, int64_t bytes, dpe_stream_t stream);
int dpe_memcpy_d2h(void *dst_host, const void *src_dev, int64_t bytes, dpe_stream_t stream); /* synchronises */
int dpe_stream_create(dpe_stream_t *stream);
int dpe_stream_destroy(dpe_stream_t stream);
int dpe_stream_synchronize(dpe_stream_t stream);

/* ------------------------------------------------------------------ BatchCorrScores - */
typedef struct dpe_bcs_config {
    int32_t samplesPerWindow;   /* S  = SamplingFrequency*SampleLength (sampleblock.cu:169) */
    int32_t lagHalfWidth;       /* L: code lags [-L,+L] kept about the fftshift centre S/2; 1..292 (beyond 32: chunks of 65 lags) */
    int32_t binHalfWidth;       /* B: Doppler bins [-B,+B] kept about C/2; B <= ~2.86e-4 * C (C = 8*2^ceil(log2 S)): 149 at
                                 * S = 50000, 18 at S = 8192 -- the bins come from a 6th-order moment expansion over
                                 * 256-sample blocks and create refuses a B whose remainder would reach fp32 rounding */
    int32_t maxWindows;         /* windows per dpe_bcs_update call (>=1) */
    int32_t maxChannels;        /* <= DPE_MAX_CHAN */
    int32_t reserved;
    double samplingFrequency;   /* fs, Hz (BCS input 9, batchcorrscores.cu:690,754) */
} dpe_bcs_config;

/* One tracked SV of one window, referenced to the START of the window: BCS inputs 1-7
 * (batchcorrscores.cu:682-688; produced by cuChanMgr, dpeflow.cpp:169-176). */
typedef struct dpe_chan_start {
    double codePhaseStart;      /* chips  */
    double carrierPhaseStart;   /* cycles */
    double codeFrequency;       /* chips/s */
    double carrierFrequency;    /* Hz */
    int32_t cpElapsedStart;     /* code periods since tracking start */
    int32_t cpReference;        /* code period of the reference Z-count */
    int32_t prn;                /* 1..37 (ValidPRNs) */
    int32_t reserved;
} dpe_chan_start;

int dpe_bcs_create(const dpe_bcs_config *cfg, dpe_bcs **out);     /* BatchCorrScores::Start  :710-886 */
int dpe_bcs_destroy(dpe_bcs *h);                                   /* BatchCorrScores::Stop   :888-973 */
/* BatchCorrScores::Update :975-1208.  samples_dev: nWindows blocks of 2*S int16, block w at
 * samples_dev + w*windowStrideSamples*2.  chan_host: [nWindows][nChan], consumed before the call returns.
 * Asynchronous on `stream`; outputs are valid once the stream is synchronised.  Calls may be issued back to
 * back: parameters of up to DPE_MAX_CHAN (window, channel) pairs travel as kernel arguments, larger batches
 * through a pinned staging block that the next call waits for. */
int dpe_bcs_update(dpe_bcs *h, const int16_t *samples_dev, int64_t windowStrideSamples,
                   int32_t nWindows, int32_t nChan, const dpe_chan_start *chan_host,
                   dpe_stream_t stream);
/* The same for ONE window with the channel parameters where the reference keeps them: in DEVICE arrays written by cuChanMgr
 * (dpeflow.cpp:169-176; BatchCorrScores captures the pointers once, batchcorrscores.cu:991-1004).  A one-block kernel derives
 * the per-channel constants on the device in fp64 (same expressions as the host form), so a host that keeps the reference's
 * cuChanMgr needs no device-to-host copy per window.  One exception: at sampling rates of >= 16 samples per chip the
 * chip-boundary stage-1 kernels are selected from the channel values, and this call then reads back the <= 3 KB block its
 * prep kernel derived (one stream wait per window) rather than fall back to the 4-5 x slower per-sample kernels. */
typedef struct dpe_bcs_ports_dev {
    const double *codePhaseStart;      /* [K] chips   (input 2) */
    const double *carrierPhaseStart;   /* [K] cycles  (input 3) */
    const double *codeFrequency;       /* [K] chips/s (input 4) */
    const double *carrierFrequency;    /* [K] Hz      (input 5) */
    const int32_t *cpElapsedStart;     /* [K]         (input 6) */
    const int32_t *cpReference;        /* [K]         (input 7) */
    const uint8_t *validPRNs;          /* [K] PRN numbers 1..37 (input 1) */
} dpe_bcs_ports_dev;
int dpe_bcs_update_dev(dpe_bcs *h, const int16_t *samples_dev, int32_t nChan, const dpe_bcs_ports_dev *ports_dev_ptrs,
                       dpe_stream_t stream);
/* (The device-parameter forms -- *_update_dev, *_update_prepared -- always launch eagerly: dpe_bcs_set_graph / dpe_bcm_set_graph
 * apply to the host-parameter Updates only.) */
/* Input check of the last dpe_bcs_update_dev (synchronises): bit 0 = a PRN outside 1..37 (clamped), bit 1 = a non-positive
 * code frequency or negative code phase.  0 = clean.  Bits 2 and 4 belong to the host-parameter batches of the high-rate chip kernel
 * (>= 48 windows at >= 16 samples per chip), whose DC sums are computed by blocks of the stage-1 launch itself: a correlator block
 * that waited too long for them added up its window's samples itself -- slower, the same exact sums, the banks are right either way.
 * Bit 2 = happened in the last such launch, bit 4 = happened at least once since create (a diagnostic: never observed outside the
 * test that forces it).  DPE_BCS_NO_SUMRIDE=1 at create keeps the separate DC-sum kernel. */
int dpe_bcs_dev_status(dpe_bcs *h, int32_t *status, dpe_stream_t stream);
/* A promise about the values behind the device ports, for sampling rates of >= 16 samples per chip: which stage-1 kernel runs is
 * a property of the channel values there, and without a promise dpe_bcs_update_dev reads back the <= 3 KB block its prep kernel
 * derived -- one small copy and one stream wait per window.  flags bit 0 (DPE_DEV_HINT_CHIP): every channel has 2 pi |fi| <= 0.25 fc
 * (|fi| below ~40 kHz) and a code frequency within 1e-5 of 1.023 MHz.  The call then decides from nominal values, reads nothing
 * back and does not wait for the stream; the prep kernel checks the promise (and that the nav-bit boundary falls on a chip boundary
 * of the replica, which it does unless fp64 rounding separates the two expressions -- about once in 1e10 windows): bit 3 of
 * dpe_bcs_dev_status = it did not hold for that window, AND THE WINDOW WAS RE-RUN: behind every hinted chip-kernel launch the call
 * enqueues the general per-sample kernels guarded by that bit (two launches whose blocks leave at once when it is clear), so the
 * banks of a flagged window are the per-sample kernels' -- within the stated tolerance like every other window's.  A broken promise
 * also withdraws the hint: the device-side check raises a pinned word, and from the next call that finds it raised the handle reads
 * the derived block back and chooses the kernel from the real values again, for the rest of its life.  The same check and the same
 * guarded re-run serve dpe_bcs_update_prepared, whose parameter block the device-resident channel manager's kernel writes (bit 6 =
 * 64 of the fix record's status).  0 clears the promise. */
#define DPE_DEV_HINT_CHIP 1
int dpe_bcs_set_dev_hint(dpe_bcs *h, int32_t flags);
/* Output ports CodeScores / CarrScores / NumFFTPoints (batchcorrscores.cu:696-698,869-874).
 * codeBank_dev: float2 [maxWindows][maxChannels][2L+1]; entry j = reference
 * codeCorrOut_d[k*S + S/2 - L + j].  carrBank_dev: float2 [..][..][2B+1]; entry j = reference
 * carrfftOut_d[k*C + C/2 - B + j].  Pointers are owned by the handle. */
int dpe_bcs_outputs(dpe_bcs *h, const float **codeBank_dev, const float **carrBank_dev,
                    int32_t *nLag, int32_t *nBin, int64_t *numFFTPoints);
/* Diagnostics of the last update, host arrays [nWindows*nChan]: nav-bit boundary sample
 * index (BCS_NavBitBoundary :237-258), replica choice (BCS_ChooseCodeCorr :512-516) and the
 * per-window DC mean [nWindows*2] (:1065-1066).  Synchronises `stream`. */
int dpe_bcs_read_info(dpe_bcs *h, int32_t *idxNext, int32_t *noFlipLarger, double *mean,
                      dpe_stream_t stream);
/* Writes window `window` of the banks into the reference's dense layout: complex128
 * codeScores_dev[K*S], carrScores_dev[K*C] (zero outside the banks).  Either may be NULL. */
int dpe_bcs_export_dense(dpe_bcs *h, int32_t window, double *codeScores_dev,
                         double *carrScores_dev, dpe_stream_t stream);

/* ------------------------------------------------------------------ BatchCorrManifold */
typedef struct dpe_bcm_config {
    int32_t samplesPerWindow;   /* S  (numSamps, batchcorrmanifold.cu:2536) */
    int32_t lagHalfWidth;       /* must equal the L of the banks passed to dpe_bcm_update */
    int32_t binHalfWidth;       /* must equal the B of the banks.  The scan keeps all maxChannels banks in LDS:
                                 * maxChannels * ((2 max(L,B) + 1) * 16 + 32) <= 150 KB, i.e. max(L,B) <= 129 at 37
                                 * channels, 599 at 8; wider ones go through a slower variant with 12-byte entries
                                 * (max(L,B) <= 174 at 37 channels); create refuses more */
    int32_t lPower;             /* LPower param (batchcorrmanifold.cu:2290) */
    int32_t maxWindows;
    int32_t maxChannels;
    int64_t numFFTPoints;       /* C  (BCM input 11) */
    double samplingFrequency;   /* fs (BCM input 7) */
    /* Manifold grids, HOST, row i = {x,y,z,delta_t} ENU offsets in metres (m/s for vel):
     * statePosManifold_t / stateVelManifold_t (gridhelper.h:11-25).  This is the LOCAL shard
     * of the global grid; gridIndexOffset is the global index of local point 0 (multi-GPU). */
    const double *posGrid;
    const double *velGrid;
    int64_t posGridSize;
    int64_t velGridSize;
    int64_t posGridIndexOffset;
    int64_t velGridIndexOffset;
    int32_t writeScores;        /* 1: keep per-point fp32 scores (PosScores port), 0: arg-max only */
    int32_t weightedMean;       /* 1: also accumulate the "Method 1" score-weighted mean (see dpe_bcm_result) */
    int32_t referencePair;      /* 1: reproduce the reference's floor(idx) / floor(idx + 1) neighbour pair bit for bit where it
                                 * differs from the continuous interpolation (batchcorrmanifold.cu:1798-1812): when an fp64 index
                                 * sits one rounding step below 2^m, idx + 1 rounds UP and the two neighbours are two apart, both
                                 * with weight ~1.  Only possible for the first channel and only when S / 2 is a power of two;
                                 * the Update then finds the affected grid points with one device pass, re-evaluates them in fp64
                                 * on the host, patches their scores with one scatter launch and re-derives the arg-max
                                 * (synchronous: a few stream waits per Update, cost proportional to the number of affected points
                                 * -- a handful; no fixed limit; needs writeScores).  With the device ports (dpe_bcm_update_dev) the
                                 * whole fix-up stays on the device -- candidates, the reference's expression in fp64 from the port
                                 * arrays, scores patched in place, arg-max re-derived -- with nothing read back; there it needs
                                 * weightedMean = 0 and holds at most 65 536 candidate points per window (more: an error from the
                                 * results call).  Not with dpe_bcm_update_prepared (its blocks hold expansion coefficients only).
                                 * 0 (default): the continuous interpolation everywhere.  No effect when S / 2 is not a power of two. */
    int32_t reserved;
} dpe_bcm_config;

/* Per-window inputs of BatchCorrManifold::Update (batchcorrmanifold.cu:2261-2279,2512-2540). */
typedef struct dpe_bcm_window {
    double xCurrkk1[8];         /* grid centre [x,y,z,c*dt, vx,vy,vz,c*dt_dot] ECEF (input 2) */
    double enu2ecef[9];         /* row-major ENU->ECEF (input 12, cuchanmgr.cu:54-73) */
    double rxTime;              /* receiver clock at window end (input 5) */
    int32_t dopplerSign;        /* +1/-1 (input 10) */
    int32_t reserved;
} dpe_bcm_window;

/* One tracked SV of one window, referenced to the END of the window (inputs 4,8,9,14,16-18). */
typedef struct dpe_chan_end {
    double satState[8];         /* mid-time entry of SatStates: batch[k*dimT + dimT/2] (:1775) */
    double codePhaseEnd;        /* chips */
    double codeFrequency;
    double carrierFrequency;
    int32_t cpRefTOW;
    int32_t cpElapsedEnd;
    int32_t cpRef;
    int32_t reserved;
} dpe_chan_end;

typedef struct dpe_bcm_result {
    double zVal[8];             /* BCM_MakePosMeas/MakeVelMeas :1977-2068 */
    int64_t posIndex;           /* global grid index of the ML point (first maximum) */
    int64_t velIndex;
    float posScore;
    float velScore;
    int64_t posOutOfWindow;     /* (point,SV) pairs whose index left the bank (reference: UB) */
    int64_t velOutOfWindow;
    /* "Method 1" of BatchCorrManifold::Update (:2546-2567, kernels :816-1056,1365-1510; what PyGNSS'
     * folded dp_measurement_estimation does, receiver.py:317-318): score-weighted mean state.
     * weightedSums[m] = {sum s, sum s*x, sum s*y, sum s*z, sum s*t} over the local shard of manifold m
     * (ENU offsets; all-reduce(SUM) them across shards), zValMean = the resulting ECEF state (NaN, as in the
     * reference, when every score of a manifold is 0, e.g. every pair outside the banks). */
    double zValMean[8];
    double weightedSums[2][5];
} dpe_bcm_result;

int dpe_bcm_create(const dpe_bcm_config *cfg, dpe_bcm **out);     /* BatchCorrManifold::Start :2315-2463 */
int dpe_bcm_destroy(dpe_bcm *h);                                   /* ::Stop :2467-2498 */
/* BatchCorrManifold::Update :2502-2635 for nWindows windows.  Asynchronous on `stream`; win_host / chan_host are
 * consumed before the call returns.  One launch scores both manifolds; its last block writes the arg-max keys and
 * out-of-window counts into pinned host memory, so dpe_bcm_results only synchronises.  A handle serves ONE stream at a
 * time (its key sets, ticket counter and result mirror are per handle): give concurrent streams their own handles. */
int dpe_bcm_update(dpe_bcm *h, const float *codeBank_dev, const float *carrBank_dev,
                   int32_t nWindows, int32_t nChan, const dpe_bcm_window *win_host,
                   const dpe_chan_end *chan_host, dpe_stream_t stream);
/* One window with the inputs where the reference keeps them: DEVICE arrays of cuChanMgr / cuEKF (dpeflow.cpp:178-191,212;
 * captured once at batchcorrmanifold.cu:2512-2533, xCurrkk1 re-read per Update :2540); rxTime is a host scalar in the
 * reference too (:2536-2537).  A one-block kernel forms the per-SV expansion coefficients in fp64 on the device (the centre
 * index as a compensated sum where the host form uses long double); the scan runs with its range clamps on, since the host
 * cannot prove the indices inside the banks.  referencePair: see dpe_bcm_config. */
typedef struct dpe_bcm_ports_dev {
    const double *xCurrkk1;            /* [8]  (input 2) */
    const double *enu2ecef;            /* [9]  row-major (input 12) */
    const double *satStates;           /* [K][dimT][8] batch satellite states (input 4); entry dimT/2 is read (:1775) */
    const double *codePhaseEnd;        /* [K]  (input 14) */
    const double *codeFrequency;       /* [K]  (input 8) */
    const double *carrierFrequency;    /* [K]  (input 9) */
    const int32_t *cpRefTOW;           /* [K]  (input 16) */
    const int32_t *cpElapsedEnd;       /* [K]  (input 17) */
    const int32_t *cpRef;              /* [K]  (input 18) */
    const int32_t *dopplerSign;        /* [1]  (input 10) */
    int32_t dimT;                      /* entries of the time grid */
    int32_t reserved;
} dpe_bcm_ports_dev;
int dpe_bcm_update_dev(dpe_bcm *h, const float *codeBank_dev, const float *carrBank_dev, int32_t nChan,
                       const dpe_bcm_ports_dev *ports_dev_ptrs, double rxTime, dpe_stream_t stream);
/* Waits for the last Update (single windows without the weighted-mean estimator: polls the sequence word the kernel writes
 * behind the results in pinned memory; otherwise synchronises `stream`) and returns the per-window ML results (zVal/RVal ports; RVal is the
 * 8x8 identity the reference writes, :2003-2011,2055-2063).  results: [nWindows]. */
int dpe_bcm_results(dpe_bcm *h, dpe_bcm_result *results, dpe_stream_t stream);
/* PosScores port (:2300,2404) and its velocity twin: one row of gridSize floats per window; window w's row starts at
 * scores_dev + w * pitch, where pitch (dpe_bcm_scores_pitch) is the grid size rounded up to a multiple of 32 floats so that
 * every row starts on a 128-byte line (the reference's port is the single row of a one-window handle: unchanged).  The
 * pitch - gridSize floats behind a row are never written. */
int dpe_bcm_scores(dpe_bcm *h, const float **posScores_dev, const float **velScores_dev);
int dpe_bcm_scores_pitch(dpe_bcm *h, int64_t *posPitch, int64_t *velPitch);
/* The port in the reference's own type -- ConfigOutput(3, "PosScores", DOUBLE_t, GRID, ...), batchcorrmanifold.cu:2300, read by
 * a DataLogger tap (datalogger.cu:215-278): window `window`'s scores as dense fp64 rows [gridSize] (either pointer may be NULL).
 * Asynchronous on `stream`, after the Update that produced them. */
int dpe_bcm_export_scores_f64(dpe_bcm *h, int32_t window, double *posScores_dev, double *velScores_dev, dpe_stream_t stream);
/* Packed arg-max keys of the LAST update (two sets alternate: query after every Update), device uint64 [maxWindows][2] (pos,vel):
 * (score bits << 32) | (0xFFFFFFFF - globalIndex): an integer max over shards reproduces
 * the "first maximum" tie-break of thrust::max_element (:2589-2590).  For RCCL all-reduce. */
int dpe_bcm_keys(dpe_bcm *h, const uint64_t **keys_dev);
/* Measurement from externally reduced keys (multi-GPU): host keys [nWindows][2]; the GLOBAL grids with their sizes (a
 * decoded index outside them, or a key of 0 = "no valid score" -- every score NaN, or a key set that was never reduced --
 * is an error, not a wild read). */
int dpe_bcm_results_from_keys(dpe_bcm *h, const uint64_t *keys_host, int32_t nWindows,
                              const double *posGridGlobal, int64_t posGridGlobalSize,
                              const double *velGridGlobal, int64_t velGridGlobalSize,
                              dpe_bcm_result *results);

/* ------------------------------------------------------------------ batches in flight ------ */
/* Several batches of the path on the device at once -- what the reference gets from SampleBlock's 32-slot ring and reader thread
 * (sampleblock.cu:327-447) and from the side streams of BatchCorrScores / BatchCorrManifold (batchcorrscores.h:60-64,
 * batchcorrmanifold.cu:2573-2586): the correlations of batch n + 1 run beside the grid scan of batch n.  A dpe_bcs / dpe_bcm handle
 * serves one stream, so a dpe_pipe owns `inFlight` LANES (a handle pair created from the two configurations, a non-blocking
 * stream, events) and deals consecutive batches to them round robin; the lanes' BatchCorrManifold handles share one device copy
 * of the grids.  Every lane runs exactly the launches a lone handle pair would: results are bit-identical to the one-stream path.
 * A batch is named by its TICKET (0, 1, 2, ...); its banks, scores, keys and results live in its lane until the lane is dealt
 * again, i.e. until `inFlight` later batches have been issued -- collect them before that.  One thread drives a pipe. */
typedef struct dpe_pipe dpe_pipe;
/* inputStream value for "the samples are already there" (resident since before anything in flight: no event, no cross-stream wait --
 * a dependency between two hardware queues costs the lane ~10 us of start latency per batch). */
#define DPE_STREAM_NONE ((dpe_stream_t)(intptr_t)-1)
int dpe_pipe_create(const dpe_bcs_config *bcsCfg, const dpe_bcm_config *bcmCfg, int32_t inFlight /* 1..8 */, dpe_pipe **out);
int dpe_pipe_destroy(dpe_pipe *p);                       /* waits for the lanes, then destroys their handles */
int dpe_pipe_in_flight(const dpe_pipe *p);
/* Deal to the first `inFlight` lanes only from now on (1 = one stream: the latency form; at most the lanes made at create). */
int dpe_pipe_set_in_flight(dpe_pipe *p, int32_t inFlight);
/* The handles and stream of lane `lane` (0 .. lanes - 1) whatever it holds: set-up calls on the lanes' handles (dpe_*_profile,
 * dpe_*_set_graph, ...), never an Update. */
int dpe_pipe_lane_at(dpe_pipe *p, int32_t lane, dpe_bcs **bcs, dpe_bcm **bcm, dpe_stream_t *laneStream);
/* One batch: dpe_bcs_update + dpe_bcm_update (same arguments) on the next lane, which first waits -- on the device -- for
 * everything enqueued so far on `inputStream` (the stream that produced samples_dev).  Asynchronous; never waits for the lane's
 * previous batch on the host.  samples_dev must stay valid until stage 1 of the batch is through (dpe_pipe_samples_consumed). */
int dpe_pipe_submit(dpe_pipe *p, const int16_t *samples_dev, int64_t windowStrideSamples, int32_t nWindows, int32_t nChan,
                    const dpe_chan_start *chanStart_host, const dpe_bcm_window *win_host, const dpe_chan_end *chanEnd_host,
                    dpe_stream_t inputStream, int64_t *ticket);
/* The same in pieces, for a host that puts its own work between the two stages (multi-GPU: the bank all-gather and the key
 * exchange, on the lane's stream): acquire hands out the next lane's handles and stream (which waits for `inputStream` as above);
 * the host calls dpe_bcs_update / ... / dpe_bcm_update on them with that stream, optionally dpe_pipe_mark_stage1 once the samples
 * are consumed, and dpe_pipe_commit at the end. */
int dpe_pipe_acquire(dpe_pipe *p, dpe_stream_t inputStream, int64_t *ticket, dpe_bcs **bcs, dpe_bcm **bcm, dpe_stream_t *laneStream);
int dpe_pipe_mark_stage1(dpe_pipe *p, int64_t ticket);
int dpe_pipe_commit(dpe_pipe *p, int64_t ticket, int32_t nWindows);
/* Handles and stream of the lane that holds `ticket` (error once the lane has been dealt again): dpe_bcs_outputs, dpe_bcm_scores,
 * dpe_bcm_keys ... of that batch. */
int dpe_pipe_lane(dpe_pipe *p, int64_t ticket, dpe_bcs **bcs, dpe_bcm **bcm, dpe_stream_t *laneStream);
/* dpe_bcm_results of that batch: waits for ITS lane only -- later batches keep running. */
int dpe_pipe_results(dpe_pipe *p, int64_t ticket, dpe_bcm_result *results);
/* Stream-ordered hand-backs (no host wait): `stream` continues once stage 1 of the batch has read its samples (a SampleBlock ring
 * slot may then be refilled: sampleblock.cu:421-447) / once every committed batch is complete.  samples_consumed also takes a
 * ticket whose lane has been dealt again (a ring deeper than the lanes): it then waits for stage 1 of the later batch on that
 * lane, which is behind this batch's in the lane's stream order. */
int dpe_pipe_samples_consumed(dpe_pipe *p, int64_t ticket, dpe_stream_t stream);
int dpe_pipe_join(dpe_pipe *p, dpe_stream_t stream);
int dpe_pipe_synchronize(dpe_pipe *p);                   /* host wait for all lanes */

/* ------------------------------------------------------------------ multi-GPU exchange ------ */
/* One process per GPU; the manifold grid is sharded (posGridIndexOffset / velGridIndexOffset), stage 1 may be sharded by
 * window.  The only data-path collectives are the arg-max exchange -- all-reduce(MAX) of the packed keys, 16 B per window --
 * and, with stage 1 sharded, the all-gather of the banks (the reference is single-GPU).  A dpe_comm is
 * either RCCL over xGMI (librccl is bound at run time; the unique id travels through <rendezvousPath>/nccl_id, rank 0
 * writes it) or a host-file transport for functional tests with several ranks on ONE GPU. */
#define DPE_COMM_RCCL 0
#define DPE_COMM_HOSTFILES 1
typedef struct dpe_comm dpe_comm;
int dpe_comm_create(int32_t rank, int32_t nRanks, const char *rendezvousPath, int32_t backend, dpe_comm **out);
/* adopt the host application's own ncclComm_t (not destroyed by dpe_comm_destroy) */
int dpe_comm_wrap_nccl(void *ncclComm, int32_t rank, int32_t nRanks, dpe_comm **out);
int dpe_comm_destroy(dpe_comm *c);
int dpe_comm_rank(const dpe_comm *c, int32_t *rank, int32_t *nRanks);
/* One thread drives a communicator.  Its collectives may go to different streams from call to call (the lanes of a dpe_pipe): on the RCCL
 * backend every call first makes its stream wait, on the device, for the previous call's completion when that one was enqueued on another
 * stream -- two collectives of one communicator are never in flight at once; every rank must issue them in the same order. */
int dpe_comm_allreduce_max_u64(dpe_comm *c, uint64_t *data_dev, int64_t count, dpe_stream_t stream);
int dpe_comm_allgather(dpe_comm *c, const void *send_dev, void *recv_dev, int64_t bytesPerRank, dpe_stream_t stream);
/* Arg-max exchange of the LAST Update: all-reduce(MAX) in place on the handle's device keys ([nWindows][2]); with keys_host
 * != NULL the reduced keys are also copied there (synchronises) -- feed them to dpe_bcm_results_from_keys with the GLOBAL
 * grids.  Every rank ends with the same keys. */
int dpe_bcm_exchange_keys(dpe_bcm *h, dpe_comm *c, uint64_t *keys_host, dpe_stream_t stream);
/* Stage 1 sharded by window: gathers the banks of this rank's LAST Update (nWindowsLocal windows, the same on every rank)
 * into codeAll_dev [nRanks * nWindowsLocal][maxChannels][2L+1] and carrAll_dev [..][2B+1] (float2), rank-major = window order
 * when rank r holds windows [r n, (r+1) n). */
int dpe_bcs_allgather_banks(dpe_bcs *h, dpe_comm *c, float *codeAll_dev, float *carrAll_dev, dpe_stream_t stream);

/* ------------------------------------------------------------------ cuChanMgr ------ */
/* Host-side (fp64) restatement of dsp::cuChanMgr (cuchanmgr.cu:1004-1268): owns the per-SV
 * channel parameters, start- and end-referenced, the Kepler satellite states and the
 * Earth-rotation-corrected batch states / ENU->ECEF matrix that feed BCS and BCM.
 * (Reference: <<<1,64>>> kernels writing device arrays; K <= 37, so this runs on the host.) */
typedef struct dpe_chanmgr dpe_chanmgr;
#define DPE_EPH_N 21
typedef struct dpe_chm_config {
    int32_t nChan;
    int32_t dopplerSign;        /* param "DopplerSign" (cuchanmgr.cu:959, dpeflow.cpp:88) */
    double sampleLength;        /* T, input 11 */
    double rxTime;              /* InitRXTime, input 9 */
} dpe_chm_config;
/* Handoff state of one SV (inputs 1-8, dpeflow.cpp:144-153) + its broadcast ephemeris in the
 * order sqrt_A,e,i_0,OMEGA_0,omega,M_0,delta_n,OMEGADOT,IDOT,C_rc,C_rs,C_uc,C_us,C_ic,C_is,
 * t_oe,t_oc,a_f0,a_f1,a_f2,T_GD (handoff_params_usrp6.csv rows 13-40). */
typedef struct dpe_chm_init_chan {
    int32_t prn, cpElapsed, cpReference, cpRefTOW;
    double codePhase, carrierPhase, codeFrequency, carrierFrequency;
    double eph[DPE_EPH_N];
} dpe_chm_init_chan;
int dpe_chm_create(const dpe_chm_config *cfg, const dpe_chm_init_chan *chans, dpe_chanmgr **out);
int dpe_chm_destroy(dpe_chanmgr *h);
/* cuChanMgr::Start :1100-1132 / ::Update :1237-1264.  xk1k1: state the channels are propagated
 * from (input 10); xkk1: grid centre (input 12); timeGrid: BCM's TimeGrid port (input 13). */
int dpe_chm_start(dpe_chanmgr *h, const double *xk1k1, const double *xkk1, const double *timeGrid, int32_t dimT);
int dpe_chm_update(dpe_chanmgr *h, const double *xk1k1, const double *xkk1, const double *timeGrid, int32_t dimT);
/* Output ports in the form BCS/BCM take (any pointer may be NULL): start[K], end[K] (with the
 * mid-time batch satellite state), win, and the full SatStates batch [K][dimT][8]. */
int dpe_chm_outputs(dpe_chanmgr *h, dpe_chan_start *start, dpe_chan_end *end, dpe_bcm_window *win,
                    double *batchSatStates);

/* ------------------------------------------------------------------ cuChanMgr on the device ------ */
/* The reference's form of the module: state and output ports live in DEVICE memory and one small kernel per window advances
 * them (cuchanmgr.cu:1100-1132 Start, :1237-1264 Update -- CHM_ComputeSatStates / PropagateChannels / TimeUpdateChannels /
 * GridPrep :240-306,338-608,641-829,853-923).  Same arithmetic as dpe_chm_* (the two forms share their functions), laid out for
 * latency: a channel's two Kepler evaluations per window (:85-210; their transmit times lie ~1e-7 s apart) run side by side
 * on two waves -- the second is the first advanced along a difference quotient over 2^-10 s, remainder 3e-11 m -- with the
 * iterations started from the previous window's anomaly (same fixed point), beside the ENU matrix and the fix hand-over.
 *
 * Attached to a single-window BatchCorrScores / BatchCorrManifold pair (dpe_chm_dev_attach) the same kernel also
 *   - forms the measurement from the scan's arg-max keys on the device (BCM_MakePosMeas / MakeVelMeas, batchcorrmanifold.cu:
 *     1977-2068) and passes it through to both state ports (EKF_PassMeas, cuekf.cu:147-159; EnableEKF = false, as shipped),
 *   - writes the channel-parameter blocks of the two handles for the NEXT window (what dpe_bcs_update_dev / dpe_bcm_update_dev
 *     derive with a prep kernel each), so that a closed loop is, per window,
 *         dpe_bcs_update_prepared -> dpe_bcm_update_prepared -> dpe_chm_dev_step        (four kernels, nothing read back)
 *   - and leaves the fix in a pinned ring that the host polls (dpe_chm_dev_fix) while later windows are already enqueued. */
typedef struct dpe_chm_dev dpe_chm_dev;
typedef struct dpe_fix_record {
    uint64_t seq;               /* window index + 1, written last (0: never written) */
    double zVal[8];             /* the window's measurement = xCurrk1k1 = xCurrkk1 of the pass-through filter */
    double rxTime;              /* receiver time at the END of the window the fix belongs to */
    int64_t posIndex, velIndex; /* global grid indices of the ML points */
    int64_t posOutOfWindow, velOutOfWindow;
    float posScore, velScore;
    int32_t status;             /* sticky bits: 1 Kepler iteration failed, 4 an arg-max key was 0 / outside the grid (state held for
                                 * that window), 8 / 16 the BatchCorrScores input flags of dpe_bcs_dev_status, 32 the filter's S was singular
                                 * (dpe_chm_dev_set_ekf; state held), 64 a dpe_bcs_set_dev_hint promise did not hold */
    int32_t reserved;
} dpe_fix_record;
int dpe_chm_dev_create(const dpe_chm_config *cfg, const dpe_chm_init_chan *chans, const double *timeGrid_host, int32_t dimT,
                       dpe_chm_dev **out);
int dpe_chm_dev_destroy(dpe_chm_dev *h);    /* in any order with the handles it is attached to: whichever goes first detaches */
/* Before Start: bcs / bcm (either may be NULL) get their parameter blocks from this channel manager; with bcm a ring of
 * fixRingDepth fixes is set up and dpe_chm_dev_step becomes available. */
int dpe_chm_dev_attach(dpe_chm_dev *h, dpe_bcs *bcs, dpe_bcm *bcm, int32_t fixRingDepth);
/* Sharded manifold grid in the device-resident loop (the reference takes its arg-max at batchcorrmanifold.cu:
 * 2589-2596): the attached BatchCorrManifold scans ITS shard of the grids (dpe_bcm_config posGridIndexOffset / velGridIndexOffset),
 * and dpe_chm_dev_step puts the exchange between the scan and the measurement kernel --
 *     dpe_bcs_update_prepared -> dpe_bcm_update_prepared -> [ all-reduce(MAX) of the device keys over `comm` ] -> measurement ...
 * stream-ordered on the RCCL backend, nothing copied to the host; the measurement kernel decodes the REDUCED keys against the
 * GLOBAL fp64 grids given here (copied to the device).  Every rank runs the same channel manager on the same reduced keys and
 * ends every window with the same fix.  After dpe_chm_dev_attach, before Start; `comm` stays the caller's. */
int dpe_chm_dev_set_shard(dpe_chm_dev *h, dpe_comm *comm, const double *posGridGlobal_host, int64_t posGridGlobalSize,
                          const double *velGridGlobal_host, int64_t velGridGlobalSize);
/* EnableEKF = true in the device-resident loop: dsp::cuEKF::StepUpdate / StepPredict (cuekf.cu:626-742; the host form is dpe_ekf_*
 * below) run inside the measurement kernel -- 8 x 8 fp64, one lane per matrix element, the same operations in the same order as the
 * host form -- on the measurement just formed (R = I, as BatchCorrManifold emits it): the state ports carry x_k|k and x_k+1|k
 * instead of the passed-through measurement, the fix record's zVal is x_k|k (what the X-file logs).  Bit 5 (32) of the record's
 * status: S was singular in some window (state held).  After dpe_chm_dev_attach, before Start.  (dpe_ekf_config is declared below.) */
struct dpe_ekf_config;
int dpe_chm_dev_set_ekf(dpe_chm_dev *h, const struct dpe_ekf_config *cfg);
/* The device port arrays, in the structs dpe_bcs_update_dev / dpe_bcm_update_dev take, plus rxTime and the two state ports
 * (any pointer may be NULL): what dpeflow.cpp:169-191,212 connects. */
int dpe_chm_dev_ports(dpe_chm_dev *h, dpe_bcs_ports_dev *bcs, dpe_bcm_ports_dev *bcm, const double **rxTime_dev,
                      double **xk1k1_dev, double **xkk1_dev, const double **zVal_dev);
int dpe_chm_dev_start(dpe_chm_dev *h, const double *x0_host /* InitX [8] */, dpe_stream_t stream);   /* ::Start :1100-1132 */
/* ::Update :1237-1264 with the two state inputs as device arrays (inputs 10 and 12; a host that keeps the reference's cuEKF). */
int dpe_chm_dev_update(dpe_chm_dev *h, const double *xk1k1_dev, const double *xkk1_dev, dpe_stream_t stream);
/* Attached form: measurement from the keys of the attached BatchCorrManifold's LAST Update, pass-through, ::Update, parameter
 * blocks of the next window, fix -> ring.  Asynchronous; never waits. */
int dpe_chm_dev_step(dpe_chm_dev *h, dpe_stream_t stream);
/* Fix of window `window` (0 = the first dpe_chm_dev_step).  timeoutMicros < 0: wait; otherwise returns 1 when the fix has not
 * arrived within that time (0 = just look).  -1: never enqueued, already overwritten (the host fell fixRingDepth behind), or the
 * loop died (its streams report an error, or have drained on two consecutive probes without the record arriving).  One thread
 * drives a manager: dpe_chm_dev_step and dpe_chm_dev_fix come from the same thread (or are serialised by the caller); the
 * streams passed to dpe_chm_dev_step must outlive the manager and the handles attached to it. */
int dpe_chm_dev_fix(dpe_chm_dev *h, int64_t window, dpe_fix_record *out, int32_t timeoutMicros);
/* Diagnostics / tests: the state in the form of dpe_chm_outputs (synchronises `stream`). */
int dpe_chm_dev_read(dpe_chm_dev *h, dpe_chan_start *start, dpe_chan_end *end, dpe_bcm_window *win, double *batchSatStates,
                     int32_t *status, dpe_stream_t stream);
/* One window whose channel-parameter block is already on the device, written by an attached device-resident channel manager (no prep kernel, no host
 * work on the values).  As the *_update_dev forms otherwise. */
int dpe_bcs_update_prepared(dpe_bcs *h, const int16_t *samples_dev, int32_t nChan, dpe_stream_t stream);
int dpe_bcm_update_prepared(dpe_bcm *h, const float *codeBank_dev, const float *carrBank_dev, int32_t nChan, dpe_stream_t stream);

/* ------------------------------------------------------------------ cuEKF ---- */
/* The EnableEKF=true path of dsp::cuEKF (cudarecv/modules/src/cuekf.cu): 8-state Kalman filter, host fp64.
 * The shipped flow disables it (dpeflow.cpp:90) and passes zVal through (EKF_PassMeas :147-159).
 * All matrices ROW-major, 8 x 8. */
typedef struct dpe_ekf dpe_ekf;
typedef struct dpe_ekf_config {
    double sampleLength;      /* T */
    int32_t coupleVelocity;   /* 1: F = I + T on [i][i+4] (EKF_MakeDPERandomWalkFMatrix :111-143); 0: F = I (ekf.py:47) */
    int32_t reserved;
    double x0[8];             /* InitX */
    double P0[64];            /* InitP (P_k-1|k-1; P_k|k-1 starts as I, cuekf.cu:464) */
} dpe_ekf_config;
int dpe_ekf_create(const dpe_ekf_config *cfg, dpe_ekf **out);
int dpe_ekf_destroy(dpe_ekf *h);
int dpe_ekf_step_update(dpe_ekf *h, const double *z /* [8] */, const double *R /* [64] */);   /* StepUpdate :660-721 */
int dpe_ekf_step_predict(dpe_ekf *h);                                                        /* StepPredict :626-656 */
int dpe_ekf_state(dpe_ekf *h, double *xk1k1, double *xkk1, double *Pk1k1, double *Pkk1, double *Q, double *K);

/* ------------------------------------------------------------------ Acquisition ---- */
/* Cold-start coarse acquisition.  Only the reference's Python twin implements it:
 * Correlator.coarse_acquisition, pygnss/pythonreceiver/scalar/correlator.py:53-103 (CUDARecv only
 * forward-declares the classes, cudarecv/dsp/inc/dsp.h:207-209).  Full code-delay x Doppler search
 * with hand-written fused transform kernels where the window shape allows -- coherent / textbook searches at 2 500, 4 000 and 5 000
 * delays per code period (2.5 / 4 / 5 Msps), the reference's non-coherent search at 10 x 2 500 samples -- and batched FFTs (rocFFT,
 * called directly: csrc/dpe_fft.h) for every other shape. */
typedef struct dpe_acq dpe_acq;
typedef struct dpe_acq_config {
    int32_t samplesPerWindow;   /* S = round(T fs), e.g. 25000 for 10 ms at 2.5 Msps (rawfile.py:162) */
    int32_t nCodePeriods;       /* N = round(T / 1 ms) (rawfile.py:161); S % N must be 0 */
    int32_t nBins;              /* Doppler bins: binStartHz + i binStepHz (correlator.py:13-14) */
    int32_t nPrn;
    int32_t mode;               /* 0: reference coherent=True; 1: reference coherent=False (sum |.| over the N
                                 * lag aliases of one S-long correlation); 2: textbook 1 ms coherent x N
                                 * non-coherent (BASELINE.json wording; not a reference algorithm) */
    int32_t prnChunk;           /* PRNs per inverse-FFT batch (0 -> 8) */
    double samplingFrequency;
    double binStartHz;
    double binStepHz;
    double dopplerSign;         /* rawfile.ds; fcaid = ds F_CA / F_L1 (rawfile.py:95) */
    int32_t prn[DPE_MAX_CHAN];
    int32_t reserved;
} dpe_acq_config;
/* cppr / cppm mask +-ceil(fs/F_CA) delays about the peak (correlator.py:96-99).  The reference's index array wraps at
 * the low end only and raises IndexError for a peak that close to the LAST delay; here the mask wraps at both ends. */
typedef struct dpe_acq_result {  /* return values of coarse_acquisition, correlator.py:86-103 */
    int32_t prn, found, maxCodeIdx, maxDoppIdx;
    double rc, fc, fi, cppr, cppm, peak;
} dpe_acq_result;
int dpe_acq_create(const dpe_acq_config *cfg, dpe_acq **out);
int dpe_acq_destroy(dpe_acq *h);
int dpe_acq_search(dpe_acq *h, const int16_t *samples_dev, dpe_stream_t stream);   /* one window, asynchronous */
int dpe_acq_results(dpe_acq *h, dpe_acq_result *results /* [nPrn] */, dpe_stream_t stream);   /* synchronises */
/* Correlator.fine_frequency_acquisition (correlator.py:105-133): code wipe-off with the coarse (rc, fc),
 * zero-padded FFT of rawfile.carr_fftpts = 8 << S.bit_length() points (rawfile.py:173), first maximum of
 * |.| inside [min(bins), max(bins)] -> ri = angle / 2 pi (cycles), fi, fc.  Synchronises. */
typedef struct dpe_acq_fine_result {
    int32_t prn, maxCarrIdx;    /* index into the fft-shifted spectrum */
    double rc, ri, fc, fi, peakRe, peakIm;
} dpe_acq_fine_result;
int dpe_acq_fine(dpe_acq *h, const int16_t *samples_dev, const dpe_acq_result *coarse /* [nPrn] */,
                 dpe_acq_fine_result *fine /* [nPrn] */, dpe_stream_t stream);
/* Receiver.scalar_acquisition (receiver.py:452-520): search_signal (coarse + fine) on two consecutive
 * windows, keep per PRN the window with the larger cppm; a second-window result is propagated back by one
 * window (rc - fc T mod 1023, ri - fi T mod 1), so the parameters always refer to the first window's start. */
typedef struct dpe_acq_track_init {
    int32_t prn, found, fromSecondWindow, reserved;
    double rc, ri, fc, fi, cppr, cppm;
    double cppmWindow[2];
} dpe_acq_track_init;
int dpe_acq_scalar_acquisition(dpe_acq *h, const int16_t *window0_dev, const int16_t *window1_dev,
                               dpe_acq_track_init *out /* [nPrn] */, dpe_stream_t stream);
/* |coarse_result_matrix| as float [nPrn][nBins][S/N] and its per-lag maximum over bins [nPrn][S/N] */
int dpe_acq_surface(dpe_acq *h, const float **surface_dev, const float **maxPerCode_dev);

/* Per-kernel timing (HIP events recorded on the launch stream around each kernel).  Returns and
 * resets the totals accumulated since the previous call, then sets the enable flag.
 * BCS slots: 0 DC-sum (not launched for single windows, where the bank kernel carries the sums), 1 bank (one launch per
 * 65-lag chunk), 2 finalize (ms[3], count[3]); BCM: slot 0 = the fused position + velocity scan (slot 1 unused). */
int dpe_bcs_profile(dpe_bcs *h, int32_t enable, float *ms, int32_t *count);
/* enable = 1: events around every kernel; enable = 2 * m: only around the slots of bit mask m (e.g. 4 = the bank kernel
 * alone -- bench.py times just the dominant kernel inside its timed region). */
/* Name of the stage-1 kernel the last Update launched ("bcs_bank_kernel", "bcs_bank16_kernel", "bcs_bank_wide_kernel",
 * "bcs_bank_chip_kernel", "bcs_bank_chip2_kernel", or the rocFFT full-lag path): which of the forms of DESIGN.md 2.2 the
 * shape selected. */
const char *dpe_bcs_stage1_kernel(dpe_bcs *h);
int dpe_bcm_profile(dpe_bcm *h, int32_t enable, float *ms, int32_t *count);

/* Graph replay (an option, NOT the low-latency path on this ROCm: measured on MI355X / ROCm 7.2 a replayed single-window step
 * takes 75 us against 47 us for the eager launches, profiles/r6_closed_loop_device.txt -- hipGraphLaunch costs more than the 2 + 1
 * launches it replaces; leave it off unless a later runtime changes that).  With enable != 0 an Update whose shape and device pointers repeat (the per-window
 * call of a running receiver, one entry per SampleBlock ring slot) is captured once as a hipGraph and
 * replayed with a single launch afterwards; the pinned parameter blocks are re-read on every replay, so
 * results are identical to the eager path.  Needs a created stream (not the null stream, which cannot be
 * captured -- eager launches are used there) and is bypassed while dpe_*_profile is enabled.  The
 * reference has no counterpart: its Update enqueues ~20 kernels per window (batchcorrscores.cu:1026-1190,
 * batchcorrmanifold.cu:2520-2632). */
int dpe_bcs_set_graph(dpe_bcs *h, int32_t enable);
int dpe_bcm_set_graph(dpe_bcm *h, int32_t enable);

/* Timing helper for bench.py: HIP events on the stream the kernels run on. */
int dpe_event_create(void **ev);
int dpe_event_record(void *ev, dpe_stream_t stream);
int dpe_event_elapsed_ms(void *start, void *stop, float *ms);  /* synchronises `stop` */
int dpe_event_destroy(void *ev);

#ifdef __cplusplus
}
#endif
#endif
