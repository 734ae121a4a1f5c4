// test_pipe.cpp -- a C++ host driving the batches-in-flight form of the C-ABI (include/dpe_hip.h, dpe_pipe_*) the way a
// reference-side shim would: nothing but the header, plain pointers and int status codes.  It replays a recording open-loop in
// batches -- SampleBlock's ring (sampleblock.cu:327-447: slots refilled while earlier ones are processed) in front of two lanes of
// BatchCorrScores / BatchCorrManifold handle pairs -- and checks every batch's results against the one-stream path
// (dpe_bcs_update + dpe_bcm_update on one handle pair) BIT FOR BIT.
//
//   test_pipe <dir>     <dir> holds the inputs a test wrote (tests/test_gpu_pipe.py::test_cpp_host_drives_the_pipe):
//       meta.txt   "fs S K W nBatches L B Gp Gv"        iq.bin     int16 [nBatches][W][2 S]
//       cs.bin     dpe_chan_start [nBatches][W][K]      ce.bin     dpe_chan_end [nBatches][W][K]
//       win.bin    dpe_bcm_window [nBatches][W]         pos.bin / vel.bin   double [G][4]
// Built by __graft_entry__.build() as navlab-dpe-sdr_amd/test_pipe.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dpe_hip.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        if ((call) != 0) {                                                                       \
            std::fprintf(stderr, "test_pipe: %s failed: %s\n", #call, dpe_last_error());         \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

template <class T>
static bool slurp(const std::string &path, std::vector<T> &out, size_t count)
{
    out.resize(count);
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const size_t got = std::fread(out.data(), sizeof(T), count, f);
    std::fclose(f);
    return got == count;
}

static bool same(const dpe_bcm_result &a, const dpe_bcm_result &b)
{
    return a.posIndex == b.posIndex && a.velIndex == b.velIndex && std::memcmp(&a.posScore, &b.posScore, sizeof(float)) == 0 &&
           std::memcmp(&a.velScore, &b.velScore, sizeof(float)) == 0 && std::memcmp(a.zVal, b.zVal, sizeof(a.zVal)) == 0 &&
           a.posOutOfWindow == b.posOutOfWindow && a.velOutOfWindow == b.velOutOfWindow;
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: test_pipe <dir>\n"); return 2; }
    const std::string dir = argv[1];
    if (dpe_abi_version() != DPE_ABI_VERSION) { std::fprintf(stderr, "test_pipe: ABI %d, built against %d\n", dpe_abi_version(), DPE_ABI_VERSION); return 1; }
    double fs = 0;
    int S = 0, K = 0, W = 0, nB = 0, L = 0, B = 0;
    long long Gp = 0, Gv = 0;
    {
        FILE *f = std::fopen((dir + "/meta.txt").c_str(), "r");
        if (!f || std::fscanf(f, "%lf %d %d %d %d %d %d %lld %lld", &fs, &S, &K, &W, &nB, &L, &B, &Gp, &Gv) != 9) { std::fprintf(stderr, "test_pipe: bad meta.txt\n"); return 2; }
        std::fclose(f);
    }
    std::vector<int16_t> iq;
    std::vector<dpe_chan_start> cs;
    std::vector<dpe_chan_end> ce;
    std::vector<dpe_bcm_window> win;
    std::vector<double> pos, vel;
    if (!slurp(dir + "/iq.bin", iq, (size_t)nB * W * 2 * S) || !slurp(dir + "/cs.bin", cs, (size_t)nB * W * K) || !slurp(dir + "/ce.bin", ce, (size_t)nB * W * K) ||
        !slurp(dir + "/win.bin", win, (size_t)nB * W) || !slurp(dir + "/pos.bin", pos, (size_t)Gp * 4) || !slurp(dir + "/vel.bin", vel, (size_t)Gv * 4)) {
        std::fprintf(stderr, "test_pipe: short input files in %s\n", dir.c_str());
        return 2;
    }
    long long C = 1;
    while (C < S) C <<= 1;
    C *= 8;                                                   // NumFFTPoints (batchcorrscores.cu:761)
    dpe_bcs_config bc = {};
    bc.samplesPerWindow = S; bc.lagHalfWidth = L; bc.binHalfWidth = B; bc.maxWindows = W; bc.maxChannels = K; bc.samplingFrequency = fs;
    dpe_bcm_config mc = {};
    mc.samplesPerWindow = S; mc.lagHalfWidth = L; mc.binHalfWidth = B; mc.lPower = 1; mc.maxWindows = W; mc.maxChannels = K;
    mc.numFFTPoints = C; mc.samplingFrequency = fs; mc.posGrid = pos.data(); mc.velGrid = vel.data(); mc.posGridSize = Gp; mc.velGridSize = Gv;
    mc.writeScores = 1;

    // ---- the one-stream path: one handle pair, one stream
    dpe_stream_t st = nullptr;
    CHECK(dpe_stream_create(&st));
    const size_t blockSamples = (size_t)W * S;               // complex samples per batch
    int16_t *buf = nullptr;
    CHECK(dpe_device_alloc((void **)&buf, (int64_t)(blockSamples * 4)));
    dpe_bcs *bcs = nullptr;
    dpe_bcm *bcm = nullptr;
    CHECK(dpe_bcs_create(&bc, &bcs));
    CHECK(dpe_bcm_create(&mc, &bcm));
    const float *code = nullptr, *carr = nullptr;
    int32_t nLag = 0, nBin = 0;
    int64_t nfft = 0;
    CHECK(dpe_bcs_outputs(bcs, &code, &carr, &nLag, &nBin, &nfft));
    std::vector<dpe_bcm_result> ref((size_t)nB * W), got((size_t)nB * W);
    for (int n = 0; n < nB; ++n) {
        CHECK(dpe_memcpy_h2d(buf, iq.data() + (size_t)n * blockSamples * 2, (int64_t)(blockSamples * 4), st));
        CHECK(dpe_bcs_update(bcs, buf, S, W, K, cs.data() + (size_t)n * W * K, st));
        CHECK(dpe_bcm_update(bcm, code, carr, W, K, win.data() + (size_t)n * W, ce.data() + (size_t)n * W * K, st));
        CHECK(dpe_bcm_results(bcm, ref.data() + (size_t)n * W, st));
    }
    CHECK(dpe_bcm_destroy(bcm));
    CHECK(dpe_bcs_destroy(bcs));

    // ---- two batches in flight behind a three-slot ring of pinned host blocks and device slots (SampleBlock's ring in small)
    constexpr int kSlots = 3;
    int16_t *slot_d[kSlots] = {}, *slot_h[kSlots] = {};
    for (int i = 0; i < kSlots; ++i) {
        CHECK(dpe_device_alloc((void **)&slot_d[i], (int64_t)(blockSamples * 4)));
        CHECK(dpe_host_alloc_pinned((void **)&slot_h[i], (int64_t)(blockSamples * 4)));
    }
    dpe_pipe *pipe = nullptr;
    CHECK(dpe_pipe_create(&bc, &mc, 2, &pipe));
    if (dpe_pipe_in_flight(pipe) != 2) { std::fprintf(stderr, "test_pipe: in_flight\n"); return 1; }
    std::vector<int64_t> ticket((size_t)nB, -1);
    for (int n = 0; n < nB; ++n) {
        const int s = n % kSlots;
        // the slot's previous user: its upload must not start before stage 1 of that batch has read the slot -- a stream-ordered wait
        if (n >= kSlots) CHECK(dpe_pipe_samples_consumed(pipe, ticket[(size_t)(n - kSlots)], st));
        if (n >= kSlots) CHECK(dpe_stream_synchronize(st));   // (... and the PINNED block is rewritten by the host: that needs the host to know)
        std::memcpy(slot_h[s], iq.data() + (size_t)n * blockSamples * 2, blockSamples * 4);
        CHECK(dpe_sampleblock_upload(slot_d[s], slot_h[s], (int64_t)blockSamples, st));
        CHECK(dpe_pipe_submit(pipe, slot_d[s], S, W, K, cs.data() + (size_t)n * W * K, win.data() + (size_t)n * W, ce.data() + (size_t)n * W * K, st,
                              &ticket[(size_t)n]));
        if (n >= 1) CHECK(dpe_pipe_results(pipe, ticket[(size_t)(n - 1)], got.data() + (size_t)(n - 1) * W));   // waits for batch n - 1 only
    }
    CHECK(dpe_pipe_results(pipe, ticket[(size_t)(nB - 1)], got.data() + (size_t)(nB - 1) * W));
    // a ticket whose lane has been dealt again is refused
    if (nB > 2 && dpe_pipe_results(pipe, ticket[0], got.data()) == 0) { std::fprintf(stderr, "test_pipe: an overtaken ticket was answered\n"); return 1; }
    CHECK(dpe_pipe_join(pipe, st));
    CHECK(dpe_stream_synchronize(st));
    CHECK(dpe_pipe_destroy(pipe));
    int bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += same(ref[i], got[i]) ? 0 : 1;
    for (int i = 0; i < kSlots; ++i) { CHECK(dpe_device_free(slot_d[i])); CHECK(dpe_host_free_pinned(slot_h[i])); }
    CHECK(dpe_device_free(buf));
    CHECK(dpe_stream_destroy(st));
    std::printf("test_pipe: %d batches x %d windows, %d results differ from the one-stream path\n", nB, W, bad);
    return bad ? 1 : 0;
}
