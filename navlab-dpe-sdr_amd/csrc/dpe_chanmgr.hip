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
//                      fixes from a pinned ring.  The kernel is a latency chain, so it is laid out across three waves: the two
//                      Kepler evaluations per channel and window (:85-210; ~1e-7 s apart) run side by side -- the second is the
//                      first advanced along a difference quotient, remainder 3e-11 m -- with warm-started iterations, beside the
//                      ENU matrix and the fix hand-over.
#include "dpe_chm_dev.h"

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
// Device-resident form (dpe_chm_dev_*): handle and entry points; the kernels are in dpe_chm_dev.h
#ifdef __HIPCC__
struct dpe_chm_dev {
    dpe::ChmDevState *st_d = nullptr;
    char *portBuf_d = nullptr;
    dpe::ChmPorts p{};
    int K = 0, dimT = 0;
    bool started = false;
    dpe_bcs *bcs = nullptr;
    dpe_bcm *bcm = nullptr;
    dpe_bcs_hook hb{};
    dpe_bcm_hook hm{};
    dpe_fix_record *ring_h = nullptr, *ring_hd = nullptr;
    dpe_fix_record *stage_d = nullptr;   // the window's record between chm_k1 (which forms it) and the chm_k2 behind it (which sends it)
    int ringDepth = 0;
    long long enqueued = 0;          // Updates enqueued since Start
    hipStream_t lastStream = nullptr;
    std::vector<hipStream_t> slotStream;   // per ring slot: the stream its window's dpe_chm_dev_step was enqueued on (liveness probe of dpe_chm_dev_fix)
    // sharded grid (dpe_chm_dev_set_shard): the keys are all-reduced before the measurement kernel, which decodes them against the global grids
    dpe_comm *comm = nullptr;
    double *gPos_d = nullptr, *gVel_d = nullptr;
    long long gPosG = 0, gVelG = 0;
    dpe::EkfDev *ekf_d = nullptr;    // cuEKF's filter inside the measurement kernel (dpe_chm_dev_set_ekf); nullptr: pass-through
};

static int chm_dev_args(dpe_chm_dev *h, int mode, int meas, const double *xk1k1, const double *xkk1, dpe::ChmKArgs &a)
{
    using namespace dpe;
    a = ChmKArgs{};
    a.st = h->st_d;
    a.p = h->p;
    a.mode = mode;
    a.xk1k1 = xk1k1;
    a.xkk1 = xkk1;
    a.meas = meas;
    a.specPar = (int)(h->enqueued & 1);
    if (meas) {
        const uint64_t *keys = nullptr;
        if (dpe_bcm_keys(h->bcm, &keys)) return -1;
        a.keys = reinterpret_cast<const unsigned long long *>(keys);
        a.posGrid = h->hm.posGrid64_d; a.velGrid = h->hm.velGrid64_d;
        a.posG = h->hm.posG; a.velG = h->hm.velG; a.posOff = h->hm.posOffset; a.velOff = h->hm.velOffset;
        if (h->comm) {   // the keys carry GLOBAL indices and have been reduced over the ranks: decode against the global grids
            a.posGrid = h->gPos_d; a.velGrid = h->gVel_d;
            a.posG = h->gPosG; a.velG = h->gVelG; a.posOff = 0; a.velOff = 0;
        }
        a.ring = h->ring_hd;
        a.ringDepth = h->ringDepth;
        a.stage = h->stage_d;
        a.ekf = h->ekf_d;
    }
    if (h->bcs) {
        if (dpe_bcs_hook_get(h->bcs, &h->hb)) return -1;   // (the hint may have been set, or withdrawn, since the last window)
        a.bcsChan = h->hb.chan_d; a.bcsStatus = h->hb.status_d; a.fs = h->hb.fs; a.S = h->hb.S;
        a.hintL1 = h->hb.hintL1; a.hintStepMax = h->hb.hintStepMax; a.hintViol = h->hb.hintViol;
    }
    if (h->bcm) {
        if (dpe_bcm_hook_get(h->bcm, &h->hm)) return -1;   // (the window frame alternates with the key sets)
        a.svPos = h->hm.svPos_d; a.svVel = h->hm.svVel_d; a.devWin = h->hm.devWin_hd;
        a.fs = h->hm.fs; a.Cf = h->hm.Cf; a.S = h->hm.S; a.L = h->hm.L; a.B = h->hm.B; a.C = h->hm.C;
    }
    return 0;
}

// K1 now; K2 either now (a kernel of its own) or, `ride` set and a BatchCorrScores attached, as an extra block of that handle's
// next stage-1 launch
static int chm_dev_launch(dpe_chm_dev *h, int mode, int meas, const double *xk1k1, const double *xkk1, bool ride, hipStream_t stream)
{
    using namespace dpe;
    ChmKArgs a;
    if (chm_dev_args(h, mode, meas, xk1k1, xkk1, a)) return -1;
    hipLaunchKernelGGL(chm_k1_kernel, dim3(1), dim3(64), 0, stream, a);
    if (ride && h->bcs) {
        if (dpe_bcs_cotask_set(h->bcs, &a, sizeof(a))) return -1;
    } else {
        hipLaunchKernelGGL(chm_k2_kernel, dim3(1), dim3(256), 0, stream, a);
    }
    DPE_CHECK_HIP(hipGetLastError());
    return 0;
}

// Called by dpe_bcs_destroy / dpe_bcm_destroy when the handle goes first (dpe_*_hook_set_owner): the handle is still whole.
static void chm_dev_detach(void *owner, int which)
{
    dpe_chm_dev *h = static_cast<dpe_chm_dev *>(owner);
    if (which == 0 && h->bcs) {
        // a time update may be parked in the handle for its next stage-1 launch: run it now (it writes this manager's state and ports
        // and sends the last window's fix) and let the stream drain -- the handle's buffers are freed next
        (void)dpe_bcs_cotask_flush(h->bcs, h->lastStream);
        (void)hipStreamSynchronize(h->lastStream);
        h->bcs = nullptr;
    } else if (which == 1 && h->bcm) {
        // the parked time update also writes the NEXT window's coefficient blocks of this BatchCorrManifold handle: run it while they
        // exist; a measurement kernel may still be reading the handle's keys
        if (h->bcs) (void)dpe_bcs_cotask_flush(h->bcs, h->lastStream);
        (void)hipStreamSynchronize(h->lastStream);
        h->bcm = nullptr;
    }
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
        s.ephC[k] = c.eph;
    }
    dpe_chm_dev *h = new dpe_chm_dev();
    h->K = cfg->nChan;
    h->dimT = dimT;
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
        hipMemcpy(p.timeGrid, timeGrid, sizeof(double) * dimT, hipMemcpyHostToDevice) != hipSuccess)
        return fail("[cuChanMgr] create: device initialisation failed");
    *out = h;
    return 0;
}

// The manager and the handles it is attached to may be destroyed in ANY order: whichever goes first tells the other (the handles
// through dpe_*_hook_set_owner -> chm_dev_detach above, the manager here).
int dpe_chm_dev_destroy(dpe_chm_dev *h)
{
    if (!h) return 0;
    // a time update may still be parked in the BatchCorrScores handle (dpe_chm_dev_step hands it to the next stage-1 launch): it
    // points into the buffers freed below -- run it now and let the stream drain; give the manifold handle its published results back
    if (h->bcs) {
        (void)dpe_bcs_cotask_flush(h->bcs, h->lastStream);
        (void)hipStreamSynchronize(h->lastStream);
        (void)dpe_bcs_hook_set_owner(h->bcs, nullptr, nullptr);
    }
    if (h->bcm) {
        (void)dpe_bcm_hook_set_publish(h->bcm, 1);
        (void)dpe_bcm_hook_set_ref_ports(h->bcm, nullptr, nullptr);
        (void)dpe_bcm_hook_set_owner(h->bcm, nullptr, nullptr);
    }
    (void)hipFree(h->st_d);
    (void)hipFree(h->portBuf_d);
    (void)hipFree(h->gPos_d);
    (void)hipFree(h->gVel_d);
    (void)hipFree(h->ekf_d);
    if (h->ring_h) (void)hipHostFree(h->ring_h);
    (void)hipFree(h->stage_d);
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
        if (dpe_bcs_hook_set_owner(bcs, chm_dev_detach, h)) return -1;
        h->bcs = bcs;
    }
    if (bcm) {
        if (dpe_bcm_hook_get(bcm, &h->hm)) return -1;
        DPE_REQUIRE(h->hm.maxWindows == 1, "[cuChanMgr] attach: the BatchCorrManifold handle must be a single-window one (the closed loop)");
        DPE_REQUIRE(h->hm.maxChannels >= h->K, "[cuChanMgr] attach: the BatchCorrManifold handle holds %d channels, %d tracked", h->hm.maxChannels, h->K);
        if (dpe_bcm_hook_set_owner(bcm, chm_dev_detach, h)) return -1;
        h->bcm = bcm;
        {
            dpe_bcm_ports_dev pm{};
            const double *rx = nullptr;
            if (dpe_chm_dev_ports(h, nullptr, &pm, &rx, nullptr, nullptr, nullptr) || dpe_bcm_hook_set_ref_ports(bcm, &pm, rx)) return -1;
        }
        if (dpe_bcm_hook_set_publish(bcm, 0)) return -1;   // chm_k1 reads the keys on the device; the fix goes out through the ring
        if (!h->ring_h) {
            DPE_CHECK_HIP(hipHostMalloc((void **)&h->ring_h, sizeof(dpe_fix_record) * (size_t)fixRingDepth, hipHostMallocDefault));
            memset(h->ring_h, 0, sizeof(dpe_fix_record) * (size_t)fixRingDepth);
            DPE_CHECK_HIP(hipHostGetDevicePointer((void **)&h->ring_hd, h->ring_h, 0));
            h->ringDepth = fixRingDepth;
            h->slotStream.assign((size_t)fixRingDepth, nullptr);
            h->stage_d = dpe::dev_alloc<dpe_fix_record>(1);
            DPE_REQUIRE(h->stage_d, "[cuChanMgr] attach: device allocation failed");
        }
    }
    return 0;
}

int dpe_chm_dev_set_shard(dpe_chm_dev *h, dpe_comm *comm, const double *posGridGlobal, int64_t posG, const double *velGridGlobal, int64_t velG)
{
    DPE_REQUIRE(h && !h->started, "[cuChanMgr] set_shard: before Start");
    DPE_REQUIRE(h->bcm && h->ring_h, "[cuChanMgr] set_shard: attach a BatchCorrManifold first (dpe_chm_dev_attach)");
    DPE_REQUIRE(comm && posGridGlobal && velGridGlobal && posG >= 1 && velG >= 1, "[cuChanMgr] set_shard: bad arguments");
    // the shard this rank's scan covers must lie inside the global grids (its keys then decode to rows of them)
    DPE_REQUIRE(h->hm.posOffset >= 0 && h->hm.posOffset + h->hm.posG <= posG && h->hm.velOffset >= 0 && h->hm.velOffset + h->hm.velG <= velG,
                "[cuChanMgr] set_shard: the attached BatchCorrManifold scans [%lld, %lld) / [%lld, %lld), outside the global grids (%lld, %lld)",
                (long long)h->hm.posOffset, (long long)(h->hm.posOffset + h->hm.posG), (long long)h->hm.velOffset,
                (long long)(h->hm.velOffset + h->hm.velG), (long long)posG, (long long)velG);
    (void)hipFree(h->gPos_d); (void)hipFree(h->gVel_d);
    h->gPos_d = dpe::dev_alloc<double>((size_t)posG * 4);
    h->gVel_d = dpe::dev_alloc<double>((size_t)velG * 4);
    DPE_REQUIRE(h->gPos_d && h->gVel_d, "[cuChanMgr] set_shard: device allocation failed");
    DPE_CHECK_HIP(hipMemcpy(h->gPos_d, posGridGlobal, sizeof(double) * 4 * (size_t)posG, hipMemcpyHostToDevice));
    DPE_CHECK_HIP(hipMemcpy(h->gVel_d, velGridGlobal, sizeof(double) * 4 * (size_t)velG, hipMemcpyHostToDevice));
    h->gPosG = posG; h->gVelG = velG;
    h->comm = comm;
    return 0;
}

int dpe_chm_dev_set_ekf(dpe_chm_dev *h, const dpe_ekf_config *cfg)
{
    using namespace dpe;
    DPE_REQUIRE(h && cfg && !h->started, "[cuChanMgr] set_ekf: before Start");
    DPE_REQUIRE(h->bcm && h->ring_h, "[cuChanMgr] set_ekf: attach a BatchCorrManifold first (dpe_chm_dev_attach)");
    DPE_REQUIRE(cfg->sampleLength > 0, "[cuEKF] create: SampleLength must be positive");
    std::vector<EkfDev> e(1);
    memset(&e[0], 0, sizeof(EkfDev));
    const auto eye = [](double *m) { for (int i = 0; i < 64; ++i) m[i] = (i % 9 == 0) ? 1.0 : 0.0; };
    eye(e[0].F);                                                              // the host form's create, dpe_ekf.hip (cuekf.cu:338-352,460-477)
    if (cfg->coupleVelocity)
        for (int j = 0; j < 4; ++j) e[0].F[j * 8 + j + 4] = cfg->sampleLength;
    eye(e[0].H); eye(e[0].Q); eye(e[0].K); eye(e[0].Pkk1);
    memcpy(e[0].Pk1k1, cfg->P0, sizeof(e[0].Pk1k1));
    memcpy(e[0].xk1k1, cfg->x0, sizeof(e[0].xk1k1));
    memcpy(e[0].xkk1, cfg->x0, sizeof(e[0].xkk1));
    e[0].coupled = cfg->coupleVelocity ? 1 : 0;                               // (the structure the device step exploits: dpe_chm_dev.h, ekf_dev_step)
    e[0].Tc = cfg->sampleLength;
    (void)hipFree(h->ekf_d);
    h->ekf_d = dev_alloc<EkfDev>(1);
    DPE_REQUIRE(h->ekf_d, "[cuChanMgr] set_ekf: device allocation failed");
    DPE_CHECK_HIP(hipMemcpy(h->ekf_d, &e[0], sizeof(EkfDev), hipMemcpyHostToDevice));
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
    if (chm_dev_launch(h, 0, 0, h->p.xk1k1, h->p.xkk1, false, stream)) return -1;
    h->started = true;
    return 0;
}

int dpe_chm_dev_update(dpe_chm_dev *h, const double *xk1k1_dev, const double *xkk1_dev, dpe_stream_t stream)
{
    DPE_REQUIRE(h && h->started, "[cuChanMgr] Error: Update() Failed due to SatPos not initialized");
    DPE_REQUIRE(xk1k1_dev && xkk1_dev, "[cuChanMgr] Update: null state pointer");
    if (h->bcs && dpe_bcs_cotask_flush(h->bcs, stream)) return -1;
    if (chm_dev_launch(h, 1, 0, xk1k1_dev, xkk1_dev, false, (hipStream_t)stream)) return -1;
    h->enqueued += 1;
    return 0;
}

int dpe_chm_dev_step(dpe_chm_dev *h, dpe_stream_t stream)
{
    DPE_REQUIRE(h && h->started, "[cuChanMgr] Error: Update() Failed due to SatPos not initialized");
    DPE_REQUIRE(h->bcs && h->bcm && h->ring_h, "[cuChanMgr] step: no BatchCorrScores / BatchCorrManifold attached (dpe_chm_dev_attach)");
    if (dpe_bcs_cotask_flush(h->bcs, stream)) return -1;   // (a time update nobody picked up: no stage-1 launch since the last step)
    // sharded grid: the arg-max keys of the scan just enqueued are reduced over the ranks, in place and in stream order, before the
    // measurement kernel reads them (batchcorrmanifold.cu:2589-2596 is where the reference takes its arg-max)
    if (h->comm && dpe_bcm_exchange_keys(h->bcm, h->comm, nullptr, stream)) return -1;
    if (chm_dev_launch(h, 1, 1, nullptr, nullptr, true, (hipStream_t)stream)) return -1;
    h->lastStream = (hipStream_t)stream;
    h->slotStream[(size_t)(h->enqueued % h->ringDepth)] = (hipStream_t)stream;
    h->enqueued += 1;
    return 0;
}

int dpe_chm_dev_fix(dpe_chm_dev *h, int64_t window, dpe_fix_record *out, int32_t timeoutMicros)
{
    DPE_REQUIRE(h && out && h->ring_h, "[cuChanMgr] fix: no fix ring (dpe_chm_dev_attach)");
    DPE_REQUIRE(window >= 0 && window < h->enqueued, "[cuChanMgr] fix: window %lld not enqueued yet", (long long)window);
    // (the record of a window is sent by the time update that FOLLOWS its measurement kernel -- normally an extra block of the next
    //  window's stage-1 launch; the newest window's has not been picked up yet: run it now)
    if (window == h->enqueued - 1 && h->bcs && dpe_bcs_cotask_flush(h->bcs, h->lastStream)) return -1;
    const dpe_fix_record *r = h->ring_h + (window % h->ringDepth);
    const unsigned long long want = (unsigned long long)window + 1ull;
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int idleProbes = 0;
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long seq = __atomic_load_n(&r->seq, __ATOMIC_ACQUIRE);
        if (seq == want) break;
        DPE_REQUIRE(seq < want, "[cuChanMgr] fix: window %lld was overwritten (the ring holds %d fixes)", (long long)window, h->ringDepth);
        if (timeoutMicros >= 0) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double us = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
            if (us > (double)timeoutMicros) return 1;   // not there yet
        }
        // liveness (every 16 384 polls, ~100 us: the probe takes the stream's lock, which the thread that enqueues the loop wants too):
        // a kernel of the loop that faulted never sends the record -- report the stream's error instead of spinning for ever.  The
        // record of window w is sent by the time update BEHIND its measurement kernel, which rides in window w + 1's stage-1 launch:
        // both windows' streams are looked at, and only when both have drained on two consecutive probes without the record having
        // arrived is that an error (a single "idle" answer may race the write landing, or a step another stream has yet to flush).
        if ((spins & 0x3fff) == 0x3fff) {
            bool idle = true;
            for (long long w = window; w <= window + 1 && w < h->enqueued; ++w) {
                const hipError_t q = hipStreamQuery(h->slotStream[(size_t)(w % h->ringDepth)]);
                if (q != hipSuccess && q != hipErrorNotReady) {
                    dpe::set_error("[cuChanMgr] fix: the loop's stream reports %s while window %lld is awaited", hipGetErrorString(q), (long long)window);
                    return -1;
                }
                idle = idle && q == hipSuccess;
            }
            if (idle && __atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) != want) {
                if (++idleProbes >= 2) {
                    dpe::set_error("[cuChanMgr] fix: the loop's streams are idle and window %lld's record never arrived", (long long)window);
                    return -1;
                }
            } else {
                idleProbes = 0;
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
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
    if (h->bcs && dpe_bcs_cotask_flush(h->bcs, stream)) return -1;   // a time update still waiting for a stage-1 launch: run it now
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
