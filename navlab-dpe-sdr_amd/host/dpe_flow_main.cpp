// dpe_flow -- batch driver that loads the DPE flow exactly as DPEFlow::LoadFlow does
// (cudarecv/dsp/src/dpeflow.cpp:26-222: 7 modules, SetModParam table, ConnectPort table) on top of
// libdpe_hip.so, runs it for N windows and writes the X-file (xCurrk1k1 per window, CSV).
// Replaces the interactive console (NEWFlow / LOADFlow / STARTFlow) for this path.
//
//   dpe_flow --samples f.dat --handoff handoff.csv --out X.csv [--fs 2.5e6] [--T 0.02] [--iters 3000]
//            [--grid-dim 25] [--spacing 1.0] [--grid-type 0|2] [--load-grid rngrid.csv] [--lpower 1]
//            [--init-delta dx dy dz dt]
//            [--device-loop [--fix-lag n]]       cuChanMgr on the device (dpe_chm_dev_*): measurement, pass-through and channel
//                                                update in one kernel behind the scan, nothing read back per window; the flow
//                                                thread enqueues up to n (default 8) windows ahead of the fixes it has collected
//            [--ranks N --rank r --rendezvous DIR [--comm rccl|files] [--device d] [--shard-stage1]]
//                                                one flow per GPU: grid shard r of N, arg-max exchanged through
//                                                dpe_bcm_exchange_keys (RCCL, or host files for tests); --shard-stage1: each
//                                                flow correlates K / N of the channels, dpe_bcs_allgather_banks completes the banks.
//                                                With --device-loop the exchange is enqueued between the scan and the channel
//                                                manager's measurement kernel (dpe_chm_dev_set_shard): still nothing read back
//   dpe_flow --dump-grid <type> <dim> <spacing> <out.bin>        (grid builders only, no GPU)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "modules.hpp"

#define CHECK(x)                                                              \
    do {                                                                      \
        if ((x) != 0) {                                                       \
            std::fprintf(stderr, "[DPEFlow] failed: %s\n", #x);               \
            return 1;                                                         \
        }                                                                     \
    } while (0)

// The flow of dpeflow.cpp:55-213 with cuEKF (pass-through), cuChanMgr and the X-file logger replaced by cuChanMgrDev.
static int run_device_loop(const std::string &samples, const std::string &handoff, const std::string &out, const std::string &rinex,
                           const std::string &loadGrid, double fs, double T, int iters, int gridDim, int gridType, int lpower, float spacing,
                           const float *delta, int L, int B, int fixLag, int device, bool timing, int ranks, int rank,
                           const std::string &rendezvous, const std::string &commName, bool enableEkf)
{
    dsp::Flow flow;
    auto *bcs = new dsp::BatchCorrScores;
    auto *bcm = new dsp::BatchCorrManifold;
    auto *chm = new dsp::cuChanMgrDev(bcs, bcm);
    flow.Add(new dsp::DPInit);
    flow.Add(new dsp::SampleBlock);
    flow.Add(bcs);
    flow.Add(bcm);
    flow.Add(chm);
    CHECK(flow.SetModParam("SampleBlock", "SamplingFrequency", fs));
    CHECK(flow.SetModParam("SampleBlock", "RunLive", false));
    CHECK(flow.SetModParam("SampleBlock", "Filename", samples.c_str()));
    CHECK(flow.SetModParam("SampleBlock", "SampleLength", T));
    CHECK(flow.SetModParam("SampleBlock", "ReleaseLag", fixLag + 2));
    CHECK(flow.SetModParam("DPInit", "HandoffFilename", handoff.c_str()));
    if (!rinex.empty()) CHECK(flow.SetModParam("DPInit", "RINEXFilename", rinex.c_str()));
    CHECK(flow.SetModParam("DPInit", "InitDeltaX", delta[0]));
    CHECK(flow.SetModParam("DPInit", "InitDeltaY", delta[1]));
    CHECK(flow.SetModParam("DPInit", "InitDeltaZ", delta[2]));
    CHECK(flow.SetModParam("DPInit", "InitDeltaT", delta[3]));
    CHECK(flow.SetModParam("DPInit", "MaxIterations", iters + 1));
    CHECK(flow.SetModParam("BatchCorrManifold", "PosGridDimSize", gridDim));
    CHECK(flow.SetModParam("BatchCorrManifold", "VelGridDimSize", gridDim));
    CHECK(flow.SetModParam("BatchCorrManifold", "GridDimSpacing", spacing));
    CHECK(flow.SetModParam("BatchCorrManifold", "GridType", gridType));
    CHECK(flow.SetModParam("BatchCorrManifold", "LPower", lpower));
    CHECK(flow.SetModParam("BatchCorrManifold", "DeviceLoop", true));
    CHECK(flow.SetModParam("BatchCorrScores", "DeviceLoop", true));
    CHECK(flow.SetModParam("BatchCorrScores", "LagHalfWidth", L));
    CHECK(flow.SetModParam("BatchCorrScores", "BinHalfWidth", B));
    CHECK(flow.SetModParam("cuChanMgr", "DopplerSign", 1));
    CHECK(flow.SetModParam("cuChanMgr", "FixLag", fixLag));
    CHECK(flow.SetModParam("cuChanMgr", "EnableEKF", enableEkf));   // cuEKF's filter inside the measurement kernel (dpe_chm_dev_set_ekf)
    CHECK(flow.SetModParam("cuChanMgr", "XFilename", out.c_str()));
    if (!loadGrid.empty()) {
        CHECK(flow.SetModParam("BatchCorrManifold", "LoadPosGrid", true));
        CHECK(flow.SetModParam("BatchCorrManifold", "LoadPosGridFilename", loadGrid.c_str()));
    }
    if (ranks > 1 || !rendezvous.empty()) {
        // one flow per GPU, each scanning grid shard `rank` of `ranks`; the device-resident channel manager puts the key exchange
        // between the scan and its measurement kernel (dpe_chm_dev_set_shard) -- every rank ends every window with the same fix
        if (commName != "rccl" && commName != "files") { std::fprintf(stderr, "--comm rccl|files\n"); return 2; }
        CHECK(flow.SetModParam("BatchCorrManifold", "ShardRank", rank));
        CHECK(flow.SetModParam("BatchCorrManifold", "ShardCount", ranks));
        CHECK(flow.SetModParam("BatchCorrManifold", "CommBackend", commName == "files" ? DPE_COMM_HOSTFILES : DPE_COMM_RCCL));
        CHECK(flow.SetModParam("BatchCorrManifold", "CommRendezvous", rendezvous.c_str()));
    }
    static const char *wires[][4] = {
        {"DPInit", "StartByte", "SampleBlock", "StartByte"},
        {"DPInit", "InitX", "cuChanMgr", "InitX"},
        {"DPInit", "InitP", "cuChanMgr", "InitP"},
        {"DPInit", "InitEph", "cuChanMgr", "InitEph"},
        {"DPInit", "InitPRN", "cuChanMgr", "InitPRN"},
        {"DPInit", "InitCodePhase", "cuChanMgr", "InitCodePhase"},
        {"DPInit", "InitCarrierPhase", "cuChanMgr", "InitCarrierPhase"},
        {"DPInit", "InitCodeFrequency", "cuChanMgr", "InitCodeFrequency"},
        {"DPInit", "InitCarrierFrequency", "cuChanMgr", "InitCarrierFrequency"},
        {"DPInit", "InitElapsedCodePeriods", "cuChanMgr", "InitElapsedCodePeriods"},
        {"DPInit", "InitReferenceCodePeriods", "cuChanMgr", "InitReferenceCodePeriods"},
        {"DPInit", "InitCPRefTOW", "cuChanMgr", "InitCPRefTOW"},
        {"DPInit", "InitRXTime", "cuChanMgr", "InitRXTime"},
        {"SampleBlock", "Samples", "BatchCorrScores", "Samples"},
        {"SampleBlock", "SamplingFrequency", "BatchCorrScores", "SamplingFrequency"},
        {"SampleBlock", "SampleLength", "BatchCorrScores", "SampleLength"},
        {"SampleBlock", "SamplingFrequency", "BatchCorrManifold", "SamplingFrequency"},
        {"SampleBlock", "SampleLength", "BatchCorrManifold", "SampleLength"},
        {"SampleBlock", "SampleLength", "cuChanMgr", "SampleLength"},
        {"BatchCorrScores", "CodeScores", "BatchCorrManifold", "CodeScores"},
        {"BatchCorrScores", "CarrScores", "BatchCorrManifold", "CarrScores"},
        {"BatchCorrScores", "NumFFTPoints", "BatchCorrManifold", "NumFFTPoints"},
        {"cuChanMgr", "ValidPRNs", "BatchCorrScores", "ValidPRNs"},              // (device arrays, dpeflow.cpp:169-191)
        {"cuChanMgr", "CodeFrequency", "BatchCorrManifold", "CodeFrequency"},
    };
    for (auto &w : wires) CHECK(flow.ConnectPort(w[0], w[1], w[2], w[3]));
    std::clog << "[DPEFlow] Completed LoadFlow, device loop. (L=" << L << ", B=" << B << ", fix lag " << fixLag << ")" << std::endl;
    if (device >= 0) CHECK(dpe_set_device(device));
    dpe_stream_t stream = nullptr;
    CHECK(dpe_stream_create(&stream));
    CHECK(flow.Start(stream));
    flow.EnableTiming(timing);
    int n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto tHalf = t0;
    while (n < iters && flow.Step() == 0) {
        if (++n == iters / 2) tHalf = std::chrono::steady_clock::now();   // (the host runs FixLag windows ahead of the device at most)
    }
    chm->Drain();                                            // the fixes still on their way (before any module stops)
    dpe_stream_synchronize(stream);
    const auto t1 = std::chrono::steady_clock::now();
    const double dt = std::chrono::duration<double>(t1 - t0).count(), dt2 = std::chrono::duration<double>(t1 - tHalf).count();
    flow.ReportTiming(std::clog);
    flow.Stop();
    dpe_stream_destroy(stream);
    std::clog << "[DPEFlow] " << n << " iterations, " << (n ? dt / n * 1e6 : 0.0) << " us per iteration ("
              << (n ? n * T / dt : 0.0) << " x real time, closed loop on the device, one window per Update)" << std::endl;
    if (n == iters && iters >= 2)   // the first calls carry one-off costs (code-object loads, clock ramp): the second half alone
        std::clog << "[DPEFlow] second half: " << dt2 / (n - iters / 2) * 1e6 << " us per iteration" << std::endl;
    return n > 0 ? 0 : 1;
}

int main(int argc, char **argv)
{
    std::string samples, handoff, out = "XFile.csv", loadGrid, rinex, rendezvous, commName = "rccl";
    int ranks = 1, rank = 0, device = -1;
    double fs = 2.5e6, T = 0.02;
    int iters = 3000, gridDim = 25, gridType = 0, lpower = 1;
    bool useGraph = false, timing = false, enableEkf = false, shardStage1 = false, deviceLoop = false;
    int fixLag = 8;
    float spacing = 1.0f, delta[4] = {0, 0, 0, 0};
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&](int n = 1) { if (i + n >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[i + 1]; };
        if (a == "--dump-grid") {
            if (i + 4 >= argc) return 2;
            const int type = std::atoi(argv[i + 1]), dim = std::atoi(argv[i + 2]);
            const double sp = std::atof(argv[i + 3]);
            const int d[4] = {dim, dim, dim, dim};
            const double s[4] = {sp, sp, sp, sp};
            std::vector<double> g, tg;
            dsp::utils::build_grid((dsp::utils::ManifoldGridTypes)type, d, s, g, &tg, false);
            FILE *f = std::fopen(argv[i + 4], "wb");
            if (!f) return 1;
            std::fwrite(g.data(), sizeof(double), g.size(), f);
            std::fwrite(tg.data(), sizeof(double), tg.size(), f);
            std::fclose(f);
            return 0;
        } else if (a == "--dump-eph") {   // --dump-eph <rinex nav file> <seconds of week> <prn> [<prn> ...]: the ephemerides DPInit would pick (no GPU)
            next(3);
            std::vector<dsp::utils::RinexNavRecord> nav;
            std::string err;
            if (dsp::utils::read_rinex_nav(argv[i + 1], nav, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
            std::vector<int> prns;
            for (int j = i + 3; j < argc; ++j) prns.push_back(std::atoi(argv[j]));
            std::vector<double> eph;
            if (dsp::utils::select_ephemerides(nav, prns, std::atof(argv[i + 2]), eph, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
            std::printf("%zu records\n", nav.size());
            for (size_t p = 0; p < prns.size(); ++p) {
                std::printf("%d", prns[p]);
                for (int j = 0; j < 21; ++j) std::printf(",%.17g", eph[p * 21 + j]);
                std::printf("\n");
            }
            return 0;
        } else if (a == "--samples") { samples = next(); ++i; }
        else if (a == "--rinex") { rinex = next(); ++i; }
        else if (a == "--handoff") { handoff = next(); ++i; }
        else if (a == "--out") { out = next(); ++i; }
        else if (a == "--load-grid") { loadGrid = next(); ++i; }
        else if (a == "--fs") { fs = std::atof(next()); ++i; }
        else if (a == "--T") { T = std::atof(next()); ++i; }
        else if (a == "--iters") { iters = std::atoi(next()); ++i; }
        else if (a == "--grid-dim") { gridDim = std::atoi(next()); ++i; }
        else if (a == "--grid-type") { gridType = std::atoi(next()); ++i; }
        else if (a == "--spacing") { spacing = (float)std::atof(next()); ++i; }
        else if (a == "--lpower") { lpower = std::atoi(next()); ++i; }
        else if (a == "--ranks") { ranks = std::atoi(next()); ++i; }
        else if (a == "--rank") { rank = std::atoi(next()); ++i; }
        else if (a == "--rendezvous") { rendezvous = next(); ++i; }
        else if (a == "--comm") { commName = next(); ++i; }
        else if (a == "--device") { device = std::atoi(next()); ++i; }
        else if (a == "--graph") { useGraph = true; }
        else if (a == "--shard-stage1") { shardStage1 = true; }
        else if (a == "--timing") { timing = true; }
        else if (a == "--ekf") { enableEkf = true; }
        else if (a == "--device-loop") { deviceLoop = true; }
        else if (a == "--fix-lag") { fixLag = std::atoi(next()); ++i; }
        else if (a == "--init-delta") { next(4); for (int j = 0; j < 4; ++j) delta[j] = (float)std::atof(argv[i + 1 + j]); i += 4; }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (samples.empty() || handoff.empty()) { std::fprintf(stderr, "usage: see the header of dpe_flow_main.cpp\n"); return 2; }

    // bank widths from the grids this run will use
    int L = 8, B = 48;
    {
        const int d[4] = {gridDim, gridDim, gridDim, gridDim};
        const double s[4] = {spacing, spacing, spacing, spacing};
        std::vector<double> pg, vg;
        dsp::utils::build_grid((dsp::utils::ManifoldGridTypes)gridType, d, s, pg, nullptr, false);
        dsp::utils::build_grid((dsp::utils::ManifoldGridTypes)gridType, d, s, vg, nullptr, true);
        if (!loadGrid.empty() && dsp::utils::load_grid_csv(loadGrid, (long long)pg.size() / 4, pg)) {
            std::fprintf(stderr, "[DPEFlow] cannot load %s (needs %d^4 rows)\n", loadGrid.c_str(), gridDim);
            return 1;
        }
        long long nfft = 8;
        while (nfft / 8 < (long long)(fs * T + 0.5)) nfft <<= 1;
        dsp::utils::bank_half_widths(pg, vg, fs, nfft, &L, &B);
        if (L > DPE_MAX_LAG_HALF_WIDTH) {
            std::fprintf(stderr, "[DPEFlow] grid needs +-%d code lags (> %d)\n", L, DPE_MAX_LAG_HALF_WIDTH);
            return 1;
        }
    }

    if (deviceLoop && (useGraph || shardStage1)) {
        std::fprintf(stderr, "[DPEFlow] --device-loop launches eagerly and correlates every channel on every rank (not with --graph / --shard-stage1)\n");
        return 2;
    }
    if (deviceLoop) return run_device_loop(samples, handoff, out, rinex, loadGrid, fs, T, iters, gridDim, gridType, lpower, spacing, delta, L, B,
                                           fixLag, device, timing, ranks, rank, rendezvous, commName, enableEkf);
    dsp::Flow flow;                                             // dpeflow.cpp:55-62
    flow.Add(new dsp::DPInit);
    flow.Add(new dsp::SampleBlock);
    flow.Add(new dsp::BatchCorrScores);
    flow.Add(new dsp::BatchCorrManifold);
    flow.Add(new dsp::cuEKF);
    flow.Add(new dsp::cuChanMgr);
    flow.Add(new dsp::DataLogger("XECEFLogger"));

    CHECK(flow.SetModParam("SampleBlock", "SamplingFrequency", fs));          // dpeflow.cpp:67-90
    CHECK(flow.SetModParam("SampleBlock", "RunLive", false));
    CHECK(flow.SetModParam("SampleBlock", "Filename", samples.c_str()));
    CHECK(flow.SetModParam("DPInit", "HandoffFilename", handoff.c_str()));
    if (!rinex.empty()) CHECK(flow.SetModParam("DPInit", "RINEXFilename", rinex.c_str()));   // dpeflow.cpp:41-45,131
    CHECK(flow.SetModParam("DPInit", "InitDeltaX", delta[0]));
    CHECK(flow.SetModParam("DPInit", "InitDeltaY", delta[1]));
    CHECK(flow.SetModParam("DPInit", "InitDeltaZ", delta[2]));
    CHECK(flow.SetModParam("DPInit", "InitDeltaT", delta[3]));
    CHECK(flow.SetModParam("DPInit", "MaxIterations", iters + 1));
    CHECK(flow.SetModParam("SampleBlock", "SampleLength", T));
    CHECK(flow.SetModParam("cuEKF", "SampleLength", T));
    CHECK(flow.SetModParam("BatchCorrManifold", "PosGridDimSize", gridDim));
    CHECK(flow.SetModParam("BatchCorrManifold", "VelGridDimSize", gridDim));
    CHECK(flow.SetModParam("BatchCorrManifold", "GridDimSpacing", spacing));
    CHECK(flow.SetModParam("BatchCorrManifold", "GridType", gridType));
    CHECK(flow.SetModParam("BatchCorrManifold", "LPower", lpower));
    CHECK(flow.SetModParam("cuChanMgr", "DopplerSign", 1));
    CHECK(flow.SetModParam("cuEKF", "EnableEKF", enableEkf));   // dpeflow.cpp:90 ships false
    CHECK(flow.SetModParam("XECEFLogger", "Filename", out.c_str()));
    CHECK(flow.SetModParam("XECEFLogger", "CSV", true));
    if (!loadGrid.empty()) {
        CHECK(flow.SetModParam("BatchCorrManifold", "LoadPosGrid", true));
        CHECK(flow.SetModParam("BatchCorrManifold", "LoadPosGridFilename", loadGrid.c_str()));
    }
    if (ranks > 1 || !rendezvous.empty()) {
        if (commName != "rccl" && commName != "files") { std::fprintf(stderr, "--comm rccl|files\n"); return 2; }
        CHECK(flow.SetModParam("BatchCorrManifold", "ShardRank", rank));
        CHECK(flow.SetModParam("BatchCorrManifold", "ShardCount", ranks));
        CHECK(flow.SetModParam("BatchCorrManifold", "CommBackend", commName == "files" ? DPE_COMM_HOSTFILES : DPE_COMM_RCCL));
        CHECK(flow.SetModParam("BatchCorrManifold", "CommRendezvous", rendezvous.c_str()));
        if (shardStage1) {
            CHECK(flow.SetModParam("BatchCorrScores", "ShardStage1", true));
            CHECK(flow.SetModParam("BatchCorrScores", "ShardRank", rank));
            CHECK(flow.SetModParam("BatchCorrScores", "ShardCount", ranks));
            CHECK(flow.SetModParam("BatchCorrScores", "CommBackend", commName == "files" ? DPE_COMM_HOSTFILES : DPE_COMM_RCCL));
            CHECK(flow.SetModParam("BatchCorrScores", "CommRendezvous", rendezvous.c_str()));
        }
    }
    CHECK(flow.SetModParam("BatchCorrScores", "LagHalfWidth", L));
    CHECK(flow.SetModParam("BatchCorrScores", "BinHalfWidth", B));
    CHECK(flow.SetModParam("BatchCorrScores", "UseGraph", useGraph));
    CHECK(flow.SetModParam("BatchCorrManifold", "UseGraph", useGraph));

    // port table, dpeflow.cpp:140-213 
    static const char *wires[][4] = {
        {"DPInit", "StartByte", "SampleBlock", "StartByte"},
        {"DPInit", "InitX", "cuEKF", "InitX"},
        {"DPInit", "InitP", "cuEKF", "InitP"},
        {"DPInit", "InitK", "cuEKF", "InitK"},
        {"DPInit", "InitEph", "cuChanMgr", "InitEph"},
        {"DPInit", "InitPRN", "cuChanMgr", "InitPRN"},
        {"DPInit", "InitCodePhase", "cuChanMgr", "InitCodePhase"},
        {"DPInit", "InitCarrierPhase", "cuChanMgr", "InitCarrierPhase"},
        {"DPInit", "InitCodeFrequency", "cuChanMgr", "InitCodeFrequency"},
        {"DPInit", "InitCarrierFrequency", "cuChanMgr", "InitCarrierFrequency"},
        {"DPInit", "InitElapsedCodePeriods", "cuChanMgr", "InitElapsedCodePeriods"},
        {"DPInit", "InitReferenceCodePeriods", "cuChanMgr", "InitReferenceCodePeriods"},
        {"DPInit", "InitCPRefTOW", "cuChanMgr", "InitCPRefTOW"},
        {"DPInit", "InitRXTime", "cuChanMgr", "InitRXTime"},
        {"SampleBlock", "Samples", "BatchCorrScores", "Samples"},
        {"SampleBlock", "SamplingFrequency", "BatchCorrScores", "SamplingFrequency"},
        {"SampleBlock", "SampleLength", "BatchCorrScores", "SampleLength"},
        {"SampleBlock", "SamplingFrequency", "BatchCorrManifold", "SamplingFrequency"},
        {"SampleBlock", "SampleLength", "BatchCorrManifold", "SampleLength"},
        {"SampleBlock", "SampleLength", "cuChanMgr", "SampleLength"},
        {"BatchCorrScores", "CodeScores", "BatchCorrManifold", "CodeScores"},
        {"BatchCorrScores", "CarrScores", "BatchCorrManifold", "CarrScores"},
        {"BatchCorrScores", "NumFFTPoints", "BatchCorrManifold", "NumFFTPoints"},
        {"cuChanMgr", "CodePhaseStart", "BatchCorrScores", "CodePhaseStart"},
        {"cuChanMgr", "CodeFrequency", "BatchCorrScores", "CodeFrequency"},
        {"cuChanMgr", "CarrierPhaseStart", "BatchCorrScores", "CarrierPhaseStart"},
        {"cuChanMgr", "CarrierFrequency", "BatchCorrScores", "CarrierFrequency"},
        {"cuChanMgr", "cpReference", "BatchCorrScores", "cpReference"},
        {"cuChanMgr", "cpElapsedStart", "BatchCorrScores", "cpElapsedStart"},
        {"cuChanMgr", "DopplerSign", "BatchCorrScores", "DopplerSign"},
        {"cuChanMgr", "ValidPRNs", "BatchCorrScores", "ValidPRNs"},
        {"cuChanMgr", "CodeFrequency", "BatchCorrManifold", "CodeFrequency"},
        {"cuChanMgr", "CarrierFrequency", "BatchCorrManifold", "CarrierFrequency"},
        {"cuChanMgr", "rxTime", "BatchCorrManifold", "rxTime"},
        {"cuChanMgr", "txTime", "BatchCorrManifold", "txTime"},
        {"cuChanMgr", "DopplerSign", "BatchCorrManifold", "DopplerSign"},
        {"cuChanMgr", "SatStates", "BatchCorrManifold", "SatStates"},
        {"cuChanMgr", "ENU2ECEFMat", "BatchCorrManifold", "ENU2ECEFMat"},
        {"cuChanMgr", "SatStatesOld", "BatchCorrManifold", "SatStatesOld"},
        {"cuChanMgr", "CodePhaseEnd", "BatchCorrManifold", "CodePhase"},
        {"cuChanMgr", "CarrierPhaseEnd", "BatchCorrManifold", "CarrierPhase"},
        {"cuChanMgr", "cpRefTOW", "BatchCorrManifold", "cpRefTOW"},
        {"cuChanMgr", "cpRef", "BatchCorrManifold", "cpRef"},
        {"cuChanMgr", "cpElapsedEnd", "BatchCorrManifold", "cpElapsedEnd"},
        {"BatchCorrManifold", "zVal", "cuEKF", "zVal"},
        {"BatchCorrManifold", "RVal", "cuEKF", "RVal"},
        {"BatchCorrManifold", "TimeGrid", "cuChanMgr", "TimeGrid"},
        {"cuEKF", "xCurrk1k1", "cuChanMgr", "xCurrk1k1"},
        {"cuEKF", "xCurrkk1", "cuChanMgr", "xCurrkk1"},
        {"cuEKF", "xCurrkk1", "BatchCorrManifold", "xCurrkk1"},
        {"cuEKF", "xCurrk1k1", "XECEFLogger", "Data"},
    };
    for (auto &w : wires) CHECK(flow.ConnectPort(w[0], w[1], w[2], w[3]));
    std::clog << "[DPEFlow] Completed LoadFlow. (L=" << L << ", B=" << B << ")" << std::endl;

    if (device >= 0) CHECK(dpe_set_device(device));            // one process per GPU: before anything allocates
    dpe_stream_t stream = nullptr;
    CHECK(dpe_stream_create(&stream));
    CHECK(flow.Start(stream));
    flow.EnableTiming(timing);
    int n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto tHalf = t0;
    while (n < iters && flow.Step() == 0) {                     // FlowThread loop, flow.cu:122-137
        if (++n == iters / 2) tHalf = std::chrono::steady_clock::now();
    }
    const auto t1 = std::chrono::steady_clock::now();
    const double dt = std::chrono::duration<double>(t1 - t0).count(), dt2 = std::chrono::duration<double>(t1 - tHalf).count();
    flow.ReportTiming(std::clog);
    flow.Stop();
    dpe_stream_destroy(stream);
    std::clog << "[DPEFlow] " << n << " iterations, " << (n ? dt / n * 1e6 : 0.0) << " us per iteration ("
              << (n ? n * T / dt : 0.0) << " x real time, closed loop, one window per Update)" << std::endl;
    if (n == iters && iters >= 2)   // the first calls carry one-off costs (code-object loads, clock ramp): the second half alone
        std::clog << "[DPEFlow] second half: " << dt2 / (n - iters / 2) * 1e6 << " us per iteration" << std::endl;
    return n > 0 ? 0 : 1;
}
