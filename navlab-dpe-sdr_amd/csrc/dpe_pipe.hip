// dpe_pipe.hip -- several batches of the sampleblock -> BatchCorrScores -> BatchCorrManifold path in flight (gfx950 only).
//
// What the reference gets from SampleBlock's 32-slot ring with its reader thread (cudarecv/modules/src/sampleblock.cu:327-447)
// and from the side streams of BatchCorrScores / BatchCorrManifold (batchcorrscores.h:60-64; batchcorrmanifold.cu:2573-2586):
// the work of batch n + 1 is on the device while batch n is still being scored.  A dpe_bcs / dpe_bcm handle serves one stream
// (its banks, key sets, ticket counter and pinned result mirror are per handle), so a pipe owns `inFlight` LANES -- a handle
// pair, a non-blocking stream and three events each -- and deals consecutive batches to them round robin: stage 1 of batch
// n + 1 (lane B) runs beside the grid scan of batch n (lane A), the tail of one kernel under the head of the next.  The lanes'
// BatchCorrManifold handles share ONE device copy of the manifold grids (the first lane owns it).
//
// Nothing here touches the arithmetic: a lane runs exactly the launches a lone handle pair would, so results are bit-identical to
// the one-stream path (tests/test_gpu_pipe.py).
#include "dpe_common.h"
#include "dpe_prep.h"

struct dpe_pipe {
    struct Lane {
        dpe_bcs *bcs = nullptr;
        dpe_bcm *bcm = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t in = nullptr, stage1 = nullptr, done = nullptr;
        long long ticket = -1;     // the batch this lane holds (its banks, scores, keys and result mirror)
        bool committed = false, stage1Marked = false;
        int nWindows = 0;
    };
    std::vector<Lane> lanes;
    long long next = 0;            // ticket of the next batch
    int active = 0;                // lanes batches are dealt to (dpe_pipe_set_in_flight; <= lanes.size())
    int nextLane = 0;
    static constexpr int kHist = 256;
    int laneHist[kHist] = {};      // lane of ticket t at [t % kHist], for the last kHist tickets (samples_consumed on an overtaken ticket)
};

static dpe_pipe::Lane *lane_of(dpe_pipe *p, int64_t ticket, const char *who)
{
    if (!p || ticket < 0 || ticket >= p->next) {
        dpe::set_error("[Pipe] %s: ticket %lld was never issued", who, (long long)ticket);
        return nullptr;
    }
    for (auto &l : p->lanes)
        if (l.ticket == ticket) return &l;
    dpe::set_error("[Pipe] %s: ticket %lld is gone -- its lane was handed to a later batch (collect a batch before %d later ones are "
                   "issued)", who, (long long)ticket, p->active);
    return nullptr;
}

extern "C" {

int dpe_pipe_destroy(dpe_pipe *p)
{
    if (!p) return 0;
    for (auto &l : p->lanes)
        if (l.stream) (void)hipStreamSynchronize(l.stream);
    // reverse order: lane 0's BatchCorrManifold owns the grids the others borrow
    for (size_t i = p->lanes.size(); i-- > 0;) {
        dpe_pipe::Lane &l = p->lanes[i];
        if (l.bcm) dpe_bcm_destroy(l.bcm);
        if (l.bcs) dpe_bcs_destroy(l.bcs);
        for (hipEvent_t e : {l.in, l.stage1, l.done})
            if (e) (void)hipEventDestroy(e);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    delete p;
    return 0;
}

int dpe_pipe_create(const dpe_bcs_config *bcsCfg, const dpe_bcm_config *bcmCfg, int32_t inFlight, dpe_pipe **out)
{
    DPE_REQUIRE(bcsCfg && bcmCfg && out, "[Pipe] create: null argument");
    DPE_REQUIRE(inFlight >= 1 && inFlight <= 8, "[Pipe] create: inFlight %d not in 1..8", inFlight);
    DPE_REQUIRE(bcmCfg->maxWindows >= bcsCfg->maxWindows && bcmCfg->maxChannels == bcsCfg->maxChannels,
                "[Pipe] create: the BatchCorrManifold must hold the BatchCorrScores' windows (%d < %d) and channels (%d vs %d)",
                bcmCfg->maxWindows, bcsCfg->maxWindows, bcmCfg->maxChannels, bcsCfg->maxChannels);
    dpe_pipe *p = new dpe_pipe();
    p->lanes.resize((size_t)inFlight);
    for (int i = 0; i < inFlight; ++i) {
        dpe_pipe::Lane &l = p->lanes[(size_t)i];
        if (dpe_bcs_create(bcsCfg, &l.bcs) || dpe_bcm_create_sharing(bcmCfg, i ? p->lanes[0].bcm : nullptr, &l.bcm)) {
            dpe_pipe_destroy(p);
            return -1;
        }
        if (hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&l.in, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&l.stage1, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess) {
            dpe::set_error("[Pipe] create: stream / event creation failed");
            dpe_pipe_destroy(p);
            return -1;
        }
    }
    p->active = inFlight;
    *out = p;
    return 0;
}

int dpe_pipe_in_flight(const dpe_pipe *p)
{
    return p ? p->active : 0;
}

int dpe_pipe_set_in_flight(dpe_pipe *p, int32_t inFlight)
{
    DPE_REQUIRE(p && inFlight >= 1 && inFlight <= (int)p->lanes.size(), "[Pipe] set_in_flight: %d not in 1..%d (the lanes made at create)",
                inFlight, p ? (int)p->lanes.size() : 0);
    p->active = inFlight;
    if (p->nextLane >= inFlight) p->nextLane = 0;
    return 0;
}

int dpe_pipe_lane_at(dpe_pipe *p, int32_t lane, dpe_bcs **bcs, dpe_bcm **bcm, dpe_stream_t *laneStream)
{
    DPE_REQUIRE(p && lane >= 0 && lane < (int)p->lanes.size(), "[Pipe] lane_at: lane %d out of range", lane);
    dpe_pipe::Lane &l = p->lanes[(size_t)lane];
    if (bcs) *bcs = l.bcs;
    if (bcm) *bcm = l.bcm;
    if (laneStream) *laneStream = (dpe_stream_t)l.stream;
    return 0;
}

int dpe_pipe_acquire(dpe_pipe *p, dpe_stream_t inputStream, int64_t *ticket, dpe_bcs **bcs, dpe_bcm **bcm, dpe_stream_t *laneStream)
{
    DPE_REQUIRE(p && ticket, "[Pipe] acquire: null argument");
    dpe_pipe::Lane &l = p->lanes[(size_t)p->nextLane];
    DPE_REQUIRE(l.ticket < 0 || l.committed, "[Pipe] acquire: batch %lld on this lane was acquired but never committed", l.ticket);
    // the lane starts once everything the caller enqueued on inputStream so far (the samples' producer) is done
    if (inputStream != DPE_STREAM_NONE) {
        DPE_CHECK_HIP(hipEventRecord(l.in, (hipStream_t)inputStream));
        DPE_CHECK_HIP(hipStreamWaitEvent(l.stream, l.in, 0));
    }
    l.ticket = p->next++;
    p->laneHist[l.ticket % dpe_pipe::kHist] = p->nextLane;
    p->nextLane = (p->nextLane + 1) % p->active;
    l.committed = false;
    l.stage1Marked = false;
    l.nWindows = 0;
    *ticket = l.ticket;
    if (bcs) *bcs = l.bcs;
    if (bcm) *bcm = l.bcm;
    if (laneStream) *laneStream = (dpe_stream_t)l.stream;
    return 0;
}

int dpe_pipe_mark_stage1(dpe_pipe *p, int64_t ticket)
{
    dpe_pipe::Lane *l = lane_of(p, ticket, "mark_stage1");
    if (!l) return -1;
    DPE_CHECK_HIP(hipEventRecord(l->stage1, l->stream));
    l->stage1Marked = true;
    return 0;
}

int dpe_pipe_commit(dpe_pipe *p, int64_t ticket, int32_t nWindows)
{
    dpe_pipe::Lane *l = lane_of(p, ticket, "commit");
    if (!l) return -1;
    DPE_REQUIRE(!l->committed, "[Pipe] commit: batch %lld was committed before", (long long)ticket);
    if (!l->stage1Marked) DPE_CHECK_HIP(hipEventRecord(l->stage1, l->stream));   // (no separate stage-1 mark: both events at the end)
    DPE_CHECK_HIP(hipEventRecord(l->done, l->stream));
    l->committed = true;
    l->nWindows = nWindows;
    return 0;
}

int dpe_pipe_submit(dpe_pipe *p, const int16_t *samples_dev, int64_t windowStrideSamples, int32_t nWindows, int32_t nChan,
                    const dpe_chan_start *chanStart_host, const dpe_bcm_window *win_host, const dpe_chan_end *chanEnd_host,
                    dpe_stream_t inputStream, int64_t *ticket)
{
    DPE_REQUIRE(p && ticket, "[Pipe] submit: null argument");
    int64_t t = -1;
    dpe_bcs *bcs = nullptr;
    dpe_bcm *bcm = nullptr;
    dpe_stream_t st = nullptr;
    if (dpe_pipe_acquire(p, inputStream, &t, &bcs, &bcm, &st)) return -1;
    dpe_pipe::Lane *l = lane_of(p, t, "submit");
    const float *code = nullptr, *carr = nullptr;
    int32_t nLag = 0, nBin = 0;
    int64_t nfft = 0;
    int rc = dpe_bcs_update(bcs, samples_dev, windowStrideSamples, nWindows, nChan, chanStart_host, st);
    if (!rc) rc = dpe_pipe_mark_stage1(p, t);     // the samples are free again once stage 1 is through
    if (!rc) rc = dpe_bcs_outputs(bcs, &code, &carr, &nLag, &nBin, &nfft);
    if (!rc) rc = dpe_bcm_update(bcm, code, carr, nWindows, nChan, win_host, chanEnd_host, st);
    if (rc) {      // the lane stays usable: whatever was enqueued is ordinary stream work
        l->committed = true;
        l->nWindows = 0;
        l->ticket = -1;
        return -1;
    }
    if (dpe_pipe_commit(p, t, nWindows)) return -1;
    *ticket = t;
    return 0;
}

int dpe_pipe_lane(dpe_pipe *p, int64_t ticket, dpe_bcs **bcs, dpe_bcm **bcm, dpe_stream_t *laneStream)
{
    dpe_pipe::Lane *l = lane_of(p, ticket, "lane");
    if (!l) return -1;
    if (bcs) *bcs = l->bcs;
    if (bcm) *bcm = l->bcm;
    if (laneStream) *laneStream = (dpe_stream_t)l->stream;
    return 0;
}

int dpe_pipe_results(dpe_pipe *p, int64_t ticket, dpe_bcm_result *results)
{
    dpe_pipe::Lane *l = lane_of(p, ticket, "results");
    if (!l) return -1;
    DPE_REQUIRE(l->committed && l->nWindows > 0, "[Pipe] results: batch %lld was not committed", (long long)ticket);
    return dpe_bcm_results(l->bcm, results, (dpe_stream_t)l->stream);
}

int dpe_pipe_samples_consumed(dpe_pipe *p, int64_t ticket, dpe_stream_t stream)
{
    DPE_REQUIRE(p && ticket >= 0 && ticket < p->next, "[Pipe] samples_consumed: ticket %lld was never issued", (long long)ticket);
    for (auto &l : p->lanes)
        if (l.ticket == ticket) {
            DPE_REQUIRE(l.committed, "[Pipe] samples_consumed: batch %lld was not committed", (long long)ticket);
            DPE_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, l.stage1, 0));
            return 0;
        }
    // An overtaken ticket (a ring deeper than the lanes): its lane has been dealt again, and a lane takes a new batch only after the one
    // before was committed, so the latest record of that lane's stage-1 event is at or after this batch's in the lane's stream order.
    if (ticket >= p->next - dpe_pipe::kHist) {
        DPE_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, p->lanes[(size_t)p->laneHist[ticket % dpe_pipe::kHist]].stage1, 0));
        return 0;
    }
    for (auto &l : p->lanes) DPE_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, l.stage1, 0));   // older than the history: every lane
    return 0;
}

int dpe_pipe_join(dpe_pipe *p, dpe_stream_t stream)
{
    DPE_REQUIRE(p, "[Pipe] join: null argument");
    for (auto &l : p->lanes)
        if (l.ticket >= 0 && l.committed) DPE_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, l.done, 0));
    return 0;
}

int dpe_pipe_synchronize(dpe_pipe *p)
{
    DPE_REQUIRE(p, "[Pipe] synchronize: null argument");
    for (auto &l : p->lanes) DPE_CHECK_HIP(hipStreamSynchronize(l.stream));
    return 0;
}

}  // extern "C"
