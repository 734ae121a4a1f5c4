// modules.hpp -- the hot path's modules (and the thin callers either side of it) written against the
// C-ABI of libdpe_hip.so, with the reference's class names, port names, parameter keys and error
// behaviour (0 ok / -1 + "[Module] ..." on stderr; no exceptions across Start/Update/Stop).
//
//   DPInit            handoff CSV -> initial state             cudarecv/modules/src/dpinit.cpp:118-400
//   SampleBlock       file -> pinned ring -> device block      cudarecv/modules/src/sampleblock.cu:23-515
//   BatchCorrScores   -> dpe_bcs_*                             cudarecv/modules/src/batchcorrscores.cu:674-1208
//   BatchCorrManifold -> dpe_bcm_*                             cudarecv/modules/src/batchcorrmanifold.cu:2247-2635
//   cuEKF             pass-through only (EnableEKF=false)      cudarecv/modules/src/cuekf.cu:147-159,560-599
//   cuChanMgr         -> dpe_chm_*                             cudarecv/modules/src/cuchanmgr.cu:930-1268
//   DataLogger        CSV rows of one port                     cudarecv/modules/src/datalogger.cu:156-203
//
//   cuChanMgrDev      -> dpe_chm_dev_*  (device loop)          cuchanmgr.cu:1100-1132,1237-1264 + cuekf.cu:147-159 + batchcorrmanifold.cu:1977-2068
//
// Port-level difference from the reference: with cuChanMgr the channel-parameter ports are HOST arrays; with cuChanMgrDev
// (dpe_flow --device-loop) they are device arrays as in the reference, the measurement / pass-through / channel update run in
// one kernel behind the scan, and no module waits for the GPU: the fixes arrive through a pinned ring a few windows later.
// Sample / score-bank ports are device pointers in both.
#pragma once
#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <fstream>
#include <mutex>
#include <sstream>
#include <sys/stat.h>
#include <thread>

#include "../../include/dpe_hip.h"
#include "dsp.hpp"
#include "grids.hpp"
#include "rinex.hpp"

namespace dsp {

#define DPE_MOD_FAIL(msg)                                                       \
    do {                                                                        \
        std::cerr << "[" << ModuleName << "] " << msg << std::endl;             \
        return -1;                                                              \
    } while (0)

static inline dpe_stream_t flow_stream(void *flowStream) { return *(dpe_stream_t *)flowStream; }

// The library this host was compiled against: a module that calls into libdpe_hip.so checks at Start that the loaded library
// speaks the header's ABI (struct layouts, status-bit meanings, entry points; DPE_ABI_VERSION in include/dpe_hip.h).
#define DPE_MOD_CHECK_ABI()                                                                                                  \
    do {                                                                                                                     \
        if (dpe_abi_version() != DPE_ABI_VERSION) {                                                                          \
            std::cerr << "[" << ModuleName << "] Start: libdpe_hip.so has ABI version " << dpe_abi_version()                 \
                      << ", this host was built against version " << DPE_ABI_VERSION << std::endl;                          \
            return -1;                                                                                                       \
        }                                                                                                                    \
    } while (0)

// ------------------------------------------------------------------------------------------------
class DPInit : public Module {
  public:
    DPInit()
    {
        ModuleName = "DPInit";
        AllocateOutputs(14);
        InsertParam("HandoffFilename", handoffFilename, CHAR_t, sizeof(handoffFilename), 0);
        InsertParam("RINEXFilename", rinexFilename, CHAR_t, sizeof(rinexFilename), 0);   // dpinit.cpp:95
        InsertParam("InitDeltaX", &delta[0], FLOAT_t, sizeof(float), sizeof(float));
        InsertParam("InitDeltaY", &delta[1], FLOAT_t, sizeof(float), sizeof(float));
        InsertParam("InitDeltaZ", &delta[2], FLOAT_t, sizeof(float), sizeof(float));
        InsertParam("InitDeltaT", &delta[3], FLOAT_t, sizeof(float), sizeof(float));
        InsertParam("MaxIterations", &maxIter, INT_t, sizeof(int), sizeof(int));
        const char *names[12] = {"StartByte", "InitX", "InitPRN", "InitCodePhase", "InitCarrierPhase", "InitCodeFrequency",
                                 "InitCarrierFrequency", "InitElapsedCodePeriods", "InitReferenceCodePeriods",
                                 "InitCPRefTOW", "InitRXTime", "InitEph"};
        const DataType_t dt[12] = {INT_t, DOUBLE_t, CHAR_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, INT_t, INT_t, INT_t, DOUBLE_t, UNDEFINED_t};
        const ValueType_t vt[12] = {VALUE, STATE, VALUE, VALUE, VALUE, FREQUENCY_HZ, FREQUENCY_HZ, VALUE, VALUE, VALUE, VALUE, EPHEMS};
        for (int i = 0; i < 12; ++i) ConfigOutput(i, names[i], dt[i], vt[i], HOST, 1, nullptr, 0);
        ConfigOutput(12, "InitP", DOUBLE_t, COVARIANCE, HOST, 64, initP, 0);   // dpinit.cpp:85,162: identity
        ConfigOutput(13, "InitK", INT_t, VALUE, HOST, 1, &initK, 0);           // dpinit.cpp:86,164: 0
    }
    int Start(void *) override
    {
        std::ifstream f(handoffFilename);
        if (!f) DPE_MOD_FAIL("Unable to open handoff file: " << handoffFilename);
        std::map<std::string, std::vector<std::string>> rows;
        std::string line;
        while (std::getline(f, line)) {   // "key,v0,v1,...\r?\n" (dpinit.cpp:247-400)
            while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
            std::stringstream ss(line);
            std::string tok, key;
            std::getline(ss, key, ',');
            while (std::getline(ss, tok, ',')) rows[key].push_back(tok);
        }
        static const char *need[] = {"rxTime", "X_ECEF", "bytes_read", "prn_list", "rc", "ri", "fc", "fi", "cp", "cp_timestamp", "TOW"};
        for (const char *k : need)
            if (rows.find(k) == rows.end()) DPE_MOD_FAIL("handoff file lacks row " << k);
        K = (int)rows["prn_list"].size();
        if (K < 1 || K > DPE_MAX_CHAN || rows["X_ECEF"].size() != 8) DPE_MOD_FAIL("malformed handoff file");
        rxTime = std::atof(rows["rxTime"][0].c_str());
        startByte = std::atoll(rows["bytes_read"][0].c_str());
        for (int i = 0; i < 8; ++i) X[i] = std::atof(rows["X_ECEF"][i].c_str());
        for (int i = 0; i < 4; ++i) X[i] += delta[i];   // PerturbInitialization (dpinit.cpp:205-212)
        prn.resize(K); rc.resize(K); ri.resize(K); fc.resize(K); fi.resize(K); cp.resize(K); cpRef.resize(K); tow.resize(K);
        eph.assign((size_t)K * DPE_EPH_N, 0.0);
        static const char *ephKeys[DPE_EPH_N] = {"sqrt_A", "e", "i_0", "OMEGA_0", "omega", "M_0", "delta_n", "OMEGADOT", "IDOT",
                                                 "C_rc", "C_rs", "C_uc", "C_us", "C_ic", "C_is", "t_oe", "t_oc", "a_f0", "a_f1", "a_f2", "T_GD"};
        for (int k = 0; k < K; ++k) {
            prn[k] = (uint8_t)std::atoi(rows["prn_list"][k].c_str());
            rc[k] = std::atof(rows["rc"][k].c_str()); ri[k] = std::atof(rows["ri"][k].c_str());
            fc[k] = std::atof(rows["fc"][k].c_str()); fi[k] = std::atof(rows["fi"][k].c_str());
            cp[k] = (int)std::atof(rows["cp"][k].c_str()); cpRef[k] = (int)std::atof(rows["cp_timestamp"][k].c_str());
            tow[k] = (int)std::atof(rows["TOW"][k].c_str());
            for (int j = 0; j < DPE_EPH_N && !rinexFilename[0]; ++j) {   // with a RINEX file the rows are not needed
                if (rows.find(ephKeys[j]) == rows.end() || (int)rows[ephKeys[j]].size() < K) DPE_MOD_FAIL("handoff file lacks ephemeris row " << ephKeys[j]);
                eph[(size_t)k * DPE_EPH_N + j] = std::atof(rows[ephKeys[j]][k].c_str());
            }
        }
        // Ephemerides: the reference takes them from a RINEX nav file (dpinit.cpp:130-144) and lets cuChanMgr pick the set
        // closest in toe (cuchanmgr.cu:269-299); PyGNSS' handoff file carries the same broadcast values as rows.  With a
        // RINEXFilename set the RINEX data win (checked against the handoff rows to 1e-11 on the reference's demofiles).
        if (rinexFilename[0]) {
            std::vector<dsp::utils::RinexNavRecord> nav;
            std::string err;
            if (dsp::utils::read_rinex_nav(rinexFilename, nav, err)) DPE_MOD_FAIL(err);
            std::vector<int> prns(prn.begin(), prn.end());
            std::vector<double> sel;
            if (dsp::utils::select_ephemerides(nav, prns, rxTime, sel, err)) DPE_MOD_FAIL(err);
            eph = sel;
        }
        void *data[12] = {&startByte, X, prn.data(), rc.data(), ri.data(), fc.data(), fi.data(), cp.data(), cpRef.data(), tow.data(), &rxTime, eph.data()};
        const uint32_t len[12] = {1, 8, (uint32_t)K, (uint32_t)K, (uint32_t)K, (uint32_t)K, (uint32_t)K, (uint32_t)K, (uint32_t)K, (uint32_t)K, 1, (uint32_t)K};
        for (int i = 0; i < 12; ++i) UpdateOutput(i, len[i], data[i], 0);
        for (int i = 0; i < 64; ++i) initP[i] = (i % 9 == 0) ? 1.0 : 0.0;
        loop = 0;
        return 0;
    }
    int Update(void *) override { return (++loop >= maxIter) ? -1 : 0; }   // dpinit.cpp:224-235 (3000 there)

  private:
    char handoffFilename[512] = "", rinexFilename[512] = "";
    float delta[4] = {0, 0, 0, 0};
    int maxIter = 3000, loop = 0, K = 0;
    long long startByte = 0;
    double rxTime = 0, X[8] = {}, initP[64] = {};
    int initK = 0;
    std::vector<uint8_t> prn;
    std::vector<double> rc, ri, fc, fi, eph;
    std::vector<int> cp, cpRef, tow;
};

// ------------------------------------------------------------------------------------------------
class SampleBlock : public Module {
  public:
    SampleBlock()
    {
        ModuleName = "SampleBlock";
        AllocateInputs(1);
        AllocateOutputs(3);
        ConfigExpectedInput(0, "StartByte", INT_t, VALUE, 1);
        ConfigOutput(0, "Samples", UNDEFINED_t, VALUE_CMPX, HIP_DEVICE, 2, nullptr, 0);
        ConfigOutput(1, "SamplingFrequency", DOUBLE_t, FREQUENCY_HZ, HOST, 1, &SamplingFrequency, 0);
        ConfigOutput(2, "SampleLength", DOUBLE_t, VALUE, HOST, 1, &SampleLength, 0);
        InsertParam("Filename", Filename, CHAR_t, sizeof(Filename), 0);
        InsertParam("SamplingFrequency", &SamplingFrequency, DOUBLE_t, sizeof(double), sizeof(double));
        InsertParam("SampleLength", &SampleLength, DOUBLE_t, sizeof(double), sizeof(double));
        InsertParam("RunLive", &RunLive, BOOL_t, sizeof(bool), sizeof(bool));
        // network source of the reference (sampleblock.cu:51-52,57): the keys are accepted so that a parameter script
        // written for it loads; a non-file source is refused at Start (networking is outside this build's scope)
        InsertParam("Hostname", Hostname, CHAR_t, sizeof(Hostname), 0);
        InsertParam("PortNo", &PortNo, INT_t, sizeof(int), sizeof(int));
        InsertParam("InputSourceType", &InputSourceType, CHAR_t, sizeof(char), sizeof(char));
        // device loop: the consumer may be this many windows behind the flow thread, so a block goes back to the reader that
        // many Updates after it was handed out (0: at the next Update, the reference's behaviour)
        InsertParam("ReleaseLag", &ReleaseLag, INT_t, sizeof(int), sizeof(int));
    }
    ~SampleBlock() override { Stop(); }
    int Start(void *) override
    {
        if (running) return 0;
        DPE_MOD_CHECK_ABI();
        if (SamplingFrequency <= 0 || SampleLength <= 0) DPE_MOD_FAIL("SamplingFrequency / SampleLength not set");
        if (InputSourceType != 0) DPE_MOD_FAIL("InputSourceType " << (int)InputSourceType << ": only the file source (0) is built");
        fd = ::open(Filename, O_RDONLY);
        if (fd < 0) DPE_MOD_FAIL("Unable to open file: " << Filename);
        const long long start = inputs[0] ? *(long long *)inputs[0]->Data : 0;
        if (::lseek(fd, start, SEEK_SET) != start) DPE_MOD_FAIL("Failed to skip ahead in file: " << Filename);
        BlockLength = (uint32_t)(SamplingFrequency * SampleLength + 0.5);   // sampleblock.cu:169
        bytes = (size_t)BlockLength * 4;
        if (dpe_stream_create(&copyStream)) DPE_MOD_FAIL(dpe_last_error());
        for (Slot &s : ring) {
            if (dpe_host_alloc_pinned((void **)&s.host, (int64_t)bytes) || dpe_device_alloc((void **)&s.dev, (int64_t)bytes))
                DPE_MOD_FAIL("Unable to allocate sample buffers: " << dpe_last_error());
            s.ready = false;
        }
        outputs[0].VectorLength = BlockLength;
        stop = eof = false;
        load = 0; proc = -1; handed = 0;
        running = true;
        reader = std::thread(&SampleBlock::ReaderLoop, this);
        return 0;
    }
    int Update(void *) override
    {
        if (!running) DPE_MOD_FAIL("Update: not started");
        std::unique_lock<std::mutex> lk(mtx);
        if (ReleaseLag < 0 || ReleaseLag > kNumBlocks - 4) DPE_MOD_FAIL("ReleaseLag " << ReleaseLag << " not in [0, " << kNumBlocks - 4 << "]");
        if (handed > ReleaseLag) {   // hand the block consumed ReleaseLag Updates ago back
            ring[(proc - ReleaseLag + kNumBlocks) % kNumBlocks].ready = false;
            cvFree.notify_one();
        }
        ++handed;
        proc = (proc + 1) % kNumBlocks;
        // 1.5 s watchdog as in the reference (sampleblock.cu:484)
        if (!cvReady.wait_for(lk, std::chrono::milliseconds(1500), [&] { return ring[proc].ready || eof; }))
            DPE_MOD_FAIL("Update: timed out waiting for samples");
        if (!ring[proc].ready) DPE_MOD_FAIL("Update: end of sample file");
        outputs[0].Data = ring[proc].dev;
        return 0;
    }
    int Stop() override
    {
        if (!running) return 0;
        { std::lock_guard<std::mutex> lk(mtx); stop = true; }
        cvFree.notify_all();
        if (reader.joinable()) reader.join();
        for (Slot &s : ring) { dpe_host_free_pinned(s.host); dpe_device_free(s.dev); s.host = s.dev = nullptr; }
        dpe_stream_destroy(copyStream);
        ::close(fd);
        fd = -1;
        running = false;
        return 0;
    }

  private:
    static constexpr int kNumBlocks = 32;   // NumBlocksDefault, sampleblock.h:78
    struct Slot { int16_t *host = nullptr, *dev = nullptr; bool ready = false; };
    void ReaderLoop()   // GetSamplesThread (sampleblock.cu:312-463): read -> pinned -> async H2D on a private stream
    {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mtx);
                // RunLive: no free buffer at once means samples of a live source are being lost (sampleblock.cu:421-425:
                // the reference logs it and carries on)
                if (RunLive && !stop && ring[load].ready) std::clog << "[SampleBlock] Fail real-time." << std::endl;
                cvFree.wait(lk, [&] { return stop || !ring[load].ready; });
                if (stop) return;
            }
            size_t got = 0;
            while (got < bytes) {
                const ssize_t r = ::read(fd, (char *)ring[load].host + got, bytes - got);
                if (r <= 0) break;
                got += (size_t)r;
            }
            if (got < bytes || dpe_sampleblock_upload(ring[load].dev, ring[load].host, BlockLength, copyStream) ||
                dpe_stream_synchronize(copyStream)) {
                std::lock_guard<std::mutex> lk(mtx);
                eof = true;
                cvReady.notify_all();
                return;
            }
            {
                std::lock_guard<std::mutex> lk(mtx);
                ring[load].ready = true;
            }
            cvReady.notify_all();
            load = (load + 1) % kNumBlocks;
        }
    }
    char Filename[512] = "";
    double SamplingFrequency = 0, SampleLength = 0;
    bool RunLive = false, running = false, stop = false, eof = false;
    char Hostname[64] = "";
    int PortNo = 0;
    char InputSourceType = 0;   // 0 = file
    int fd = -1, load = 0, proc = -1, ReleaseLag = 0;
    long long handed = 0;
    uint32_t BlockLength = 0;
    size_t bytes = 0;
    Slot ring[kNumBlocks];
    dpe_stream_t copyStream = nullptr;
    std::thread reader;
    std::mutex mtx;
    std::condition_variable cvReady, cvFree;
};

// ------------------------------------------------------------------------------------------------
class BatchCorrScores : public Module {
  public:
    BatchCorrScores()
    {
        ModuleName = "BatchCorrScores";
        AllocateInputs(11);
        AllocateOutputs(3);
        ConfigExpectedInput(0, "Samples", UNDEFINED_t, VALUE_CMPX, VECTORLENGTH_ANY);
        ConfigExpectedInput(1, "ValidPRNs", CHAR_t, VALUE, VECTORLENGTH_ANY);
        ConfigExpectedInput(2, "CodePhaseStart", DOUBLE_t, VALUE, VECTORLENGTH_ANY);
        ConfigExpectedInput(3, "CarrierPhaseStart", DOUBLE_t, VALUE, VECTORLENGTH_ANY);
        ConfigExpectedInput(4, "CodeFrequency", DOUBLE_t, FREQUENCY_HZ, VECTORLENGTH_ANY);
        ConfigExpectedInput(5, "CarrierFrequency", DOUBLE_t, FREQUENCY_HZ, VECTORLENGTH_ANY);
        ConfigExpectedInput(6, "cpElapsedStart", INT_t, VALUE, VECTORLENGTH_ANY);
        ConfigExpectedInput(7, "cpReference", INT_t, VALUE, VECTORLENGTH_ANY);
        ConfigExpectedInput(8, "DopplerSign", INT_t, VALUE, VECTORLENGTH_ANY);
        ConfigExpectedInput(9, "SamplingFrequency", DOUBLE_t, FREQUENCY_HZ, 1);
        ConfigExpectedInput(10, "SampleLength", DOUBLE_t, VALUE, 1);
        ConfigOutput(0, "CodeScores", UNDEFINED_t, VALUE_CMPX, HIP_DEVICE, VECTORLENGTH_ANY, nullptr, 0);
        ConfigOutput(1, "CarrScores", UNDEFINED_t, VALUE_CMPX, HIP_DEVICE, VECTORLENGTH_ANY, nullptr, 0);
        ConfigOutput(2, "NumFFTPoints", INT_t, VALUE, HOST, 1, nullptr, 0);
        InsertParam("LagHalfWidth", &lagHalf, INT_t, sizeof(int), sizeof(int));
        InsertParam("BinHalfWidth", &binHalf, INT_t, sizeof(int), sizeof(int));
        InsertParam("UseGraph", &useGraph, BOOL_t, sizeof(bool), sizeof(bool));
        InsertParam("SyncOutputs", &syncOutputs, BOOL_t, sizeof(bool), sizeof(bool));
        // device loop: the channel block is written on the device by cuChanMgrDev (dpe_bcs_update_prepared)
        InsertParam("DeviceLoop", &deviceLoop, BOOL_t, sizeof(bool), sizeof(bool));
        // multi-GPU (one flow per GPU, SURVEY 8e "shard SVs in stage 1 and all-gather the banks"): with ShardStage1 this flow
        // correlates channels [ShardRank K / ShardCount, (ShardRank + 1) K / ShardCount) and dpe_bcs_allgather_banks fills the
        // full banks the ports point at.  A closed loop has one window per Update, so the stage-1 shard is by channel.
        InsertParam("ShardStage1", &shardStage1, BOOL_t, sizeof(bool), sizeof(bool));
        InsertParam("ShardRank", &shardRank, INT_t, sizeof(int), sizeof(int));
        InsertParam("ShardCount", &shardCount, INT_t, sizeof(int), sizeof(int));
        InsertParam("CommBackend", &commBackend, INT_t, sizeof(int), sizeof(int));          // DPE_COMM_RCCL / DPE_COMM_HOSTFILES
        InsertParam("CommRendezvous", commRendezvous, CHAR_t, sizeof(commRendezvous), 0);
    }
    ~BatchCorrScores() override { Stop(); }
    int Start(void *) override
    {
        if (Started) { std::clog << "[" << ModuleName << "] Start: Already Started." << std::endl; return 0; }
        DPE_MOD_CHECK_ABI();
        if (!inputs[0] || !inputs[9]) DPE_MOD_FAIL("Start: inputs not connected");
        dpe_bcs_config cfg = {};
        cfg.samplesPerWindow = (int32_t)inputs[0]->VectorLength;          // batchcorrscores.cu:752
        cfg.samplingFrequency = *(double *)inputs[9]->Data;               // :754
        cfg.lagHalfWidth = lagHalf; cfg.binHalfWidth = binHalf;
        cfg.maxWindows = 1; cfg.maxChannels = DPE_MAX_CHAN;
        S = cfg.samplesPerWindow;
        sharded = shardStage1 && shardCount > 1;
        if (sharded) {
            if (shardRank < 0 || shardRank >= shardCount) DPE_MOD_FAIL("Start: ShardRank " << shardRank << " not in [0, " << shardCount << ")");
            // The channel count is known at the first Update (cuChanMgr starts after this module, as in the reference, where
            // the plans are made there: batchcorrscores.cu:991-1038): the handle for this rank's share is created then.  The
            // ports point at the gathered banks from the start.
            cfgKeep = cfg;
            long long nfft = 8;
            while (nfft / 8 < (long long)S) nfft <<= 1;                       // batchcorrscores.cu:761
            carrSTot = (int)nfft;
            const int64_t nLag = 2 * lagHalf + 1, nBin = 2 * binHalf + 1;
            if (dpe_device_alloc((void **)&codeAll, sizeof(float) * 2 * DPE_MAX_CHAN * nLag) ||
                dpe_device_alloc((void **)&carrAll, sizeof(float) * 2 * DPE_MAX_CHAN * nBin)) {
                ReleaseShard();
                DPE_MOD_FAIL("Start: " << dpe_last_error());
            }
            const std::string dir = std::string(commRendezvous) + "/stage1";   // a rendezvous of its own beside BatchCorrManifold's
            (void)::mkdir(dir.c_str(), 0777);
            if (dpe_comm_create(shardRank, shardCount, dir.c_str(), commBackend, &comm)) {
                ReleaseShard();   // (Started is still false: Stop() would return at once and leave the buffers behind)
                DPE_MOD_FAIL("Start: " << dpe_last_error());
            }
            UpdateOutput(0, (uint32_t)nLag, (void *)codeAll, lagHalf);
            UpdateOutput(1, (uint32_t)nBin, (void *)carrAll, binHalf);
            UpdateOutput(2, 1, &carrSTot, 0);
            Started = true;
            return 0;
        }
        if (dpe_bcs_create(&cfg, &h)) return -1;
        dpe_bcs_set_graph(h, useGraph ? 1 : 0);
        const float *code, *carr; int32_t nLag, nBin; int64_t nfft;
        dpe_bcs_outputs(h, &code, &carr, &nLag, &nBin, &nfft);
        carrSTot = (int)nfft;
        UpdateOutput(0, (uint32_t)nLag, (void *)code, lagHalf);   // AuxValue carries the half width
        UpdateOutput(1, (uint32_t)nBin, (void *)carr, binHalf);
        UpdateOutput(2, 1, &carrSTot, 0);
        Started = true;
        return 0;
    }
    int Update(void *flowStream) override
    {
        if (!Started) DPE_MOD_FAIL("Error: Update() Failed due to batch correlator not initialized");
        const int K = (int)inputs[1]->VectorLength;                        // numChan, :994
        if (K < 1 || K > DPE_MAX_CHAN) DPE_MOD_FAIL("Update: bad channel count");
        if (deviceLoop) {
            if (dpe_bcs_update_prepared(h, (const int16_t *)inputs[0]->Data, K, flow_stream(flowStream))) { Stop(); return -1; }
            return 0;
        }
        dpe_chan_start ch[DPE_MAX_CHAN];
        for (int k = 0; k < K; ++k) {
            ch[k].prn = ((const uint8_t *)inputs[1]->Data)[k];
            ch[k].codePhaseStart = ((const double *)inputs[2]->Data)[k];
            ch[k].carrierPhaseStart = ((const double *)inputs[3]->Data)[k];
            ch[k].codeFrequency = ((const double *)inputs[4]->Data)[k];
            ch[k].carrierFrequency = ((const double *)inputs[5]->Data)[k];
            ch[k].cpElapsedStart = ((const int *)inputs[6]->Data)[k];
            ch[k].cpReference = ((const int *)inputs[7]->Data)[k];
            ch[k].reserved = 0;
        }
        dpe_stream_t st = flow_stream(flowStream);
        if (sharded) {
            if (K % shardCount) { Stop(); DPE_MOD_FAIL("Update: ShardStage1 needs the " << K << " channels to divide over " << shardCount << " ranks"); }
            const int Kl = K / shardCount;
            if (!h) {
                cfgKeep.maxChannels = Kl;       // the gathered rows are then channel-major: rank r's block holds channels r Kl ...
                if (dpe_bcs_create(&cfgKeep, &h)) { Stop(); return -1; }
                shardK = K;
            }
            if (K != shardK) { Stop(); DPE_MOD_FAIL("Update: channel count changed from " << shardK << " to " << K << " under ShardStage1"); }
            if (dpe_bcs_update(h, (const int16_t *)inputs[0]->Data, S, 1, Kl, ch + shardRank * Kl, st) ||
                dpe_bcs_allgather_banks(h, comm, codeAll, carrAll, st)) {
                std::cerr << "[" << ModuleName << "] Update: " << dpe_last_error() << std::endl;
                Stop();
                return -1;
            }
            if (syncOutputs && dpe_stream_synchronize(st)) { Stop(); return -1; }
            return 0;
        }
        if (dpe_bcs_update(h, (const int16_t *)inputs[0]->Data, S, 1, K, ch, st)) { Stop(); return -1; }
        // The reference synchronises here (:1192-1195).  The banks are consumed by BatchCorrManifold on the
        // same flow stream, so stream order already guarantees they are complete there; not waiting lets
        // BatchCorrManifold's host-side geometry overlap the correlator kernels.  SyncOutputs = true restores the
        // reference's contract (outputs complete when Update returns) for a consumer on another stream.
        if (syncOutputs && dpe_stream_synchronize(st)) { Stop(); return -1; }
        return 0;
    }
    int Stop() override
    {
        if (!Started) return 0;
        dpe_bcs_destroy(h);
        if (comm) dpe_comm_destroy(comm);
        if (codeAll) dpe_device_free(codeAll);
        if (carrAll) dpe_device_free(carrAll);
        comm = nullptr; codeAll = carrAll = nullptr;
        h = nullptr;
        Started = false;
        return 0;
    }

    dpe_bcs *Handle() const { return h; }

  private:
    void ReleaseShard()
    {
        if (comm) dpe_comm_destroy(comm);
        if (codeAll) dpe_device_free(codeAll);
        if (carrAll) dpe_device_free(carrAll);
        comm = nullptr; codeAll = carrAll = nullptr;
    }
    dpe_bcs *h = nullptr;
    bool Started = false, deviceLoop = false;
    bool shardStage1 = false, sharded = false;
    int shardRank = 0, shardCount = 1, commBackend = DPE_COMM_RCCL, shardK = 0;
    char commRendezvous[512] = "";
    dpe_comm *comm = nullptr;
    dpe_bcs_config cfgKeep = {};
    float *codeAll = nullptr, *carrAll = nullptr;
    int lagHalf = 8, binHalf = 48, carrSTot = 0, S = 0;
    bool useGraph = false;  // replay the per-window launch sequence as one hipGraph (dpe_hip.h; measured slower)
    bool syncOutputs = false;
};

// ------------------------------------------------------------------------------------------------
class BatchCorrManifold : public Module {
  public:
    BatchCorrManifold()
    {
        ModuleName = "BatchCorrManifold";
        AllocateInputs(19);
        AllocateOutputs(4);
        const char *names[19] = {"CodeScores", "CarrScores", "xCurrkk1", "txTime", "SatStates", "rxTime", "SampleLength",
                                 "SamplingFrequency", "CodeFrequency", "CarrierFrequency", "DopplerSign", "NumFFTPoints",
                                 "ENU2ECEFMat", "SatStatesOld", "CodePhase", "CarrierPhase", "cpRefTOW", "cpElapsedEnd", "cpRef"};
        const DataType_t dt[19] = {UNDEFINED_t, UNDEFINED_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t,
                                   DOUBLE_t, INT_t, INT_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, INT_t, INT_t, INT_t};
        const ValueType_t vt[19] = {VALUE_CMPX, VALUE_CMPX, STATE, VALUE, STATE, VALUE, VALUE, FREQUENCY_HZ, FREQUENCY_HZ,
                                    FREQUENCY_HZ, VALUE, VALUE, VALUE, STATE, VALUE, VALUE, VALUE, VALUE, VALUE};
        const uint32_t len[19] = {0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 1, 1, 9, 0, 0, 0, 0, 0, 0};   // batchcorrmanifold.cu:2261-2279
        for (int i = 0; i < 19; ++i) ConfigExpectedInput(i, names[i], dt[i], vt[i], len[i]);
        InsertParam("PosGridDimSize", &posDim, INT_t, sizeof(int), sizeof(int));
        InsertParam("VelGridDimSize", &velDim, INT_t, sizeof(int), sizeof(int));
        InsertParam("GridDimSpacing", &spacing, FLOAT_t, sizeof(float), sizeof(float));
        InsertParam("GridType", &gridType, INT_t, sizeof(int), sizeof(int));
        InsertParam("LPower", &LPower, INT_t, sizeof(int), sizeof(int));
        InsertParam("GridLogFileName", gridLog, CHAR_t, sizeof(gridLog), 0);
        InsertParam("LoadPosGrid", &loadPosGrid, BOOL_t, sizeof(bool), sizeof(bool));
        InsertParam("LoadPosGridFilename", loadPosGridFilename, CHAR_t, sizeof(loadPosGridFilename), 0);
        InsertParam("UseGraph", &useGraph, BOOL_t, sizeof(bool), sizeof(bool));
        InsertParam("ReferencePair", &referencePair, BOOL_t, sizeof(bool), sizeof(bool));   // dpe_bcm_config.referencePair
        InsertParam("DeviceLoop", &deviceLoop, BOOL_t, sizeof(bool), sizeof(bool));         // coefficient blocks from cuChanMgrDev, no wait for the fix
        // multi-GPU (one flow per GPU): this flow scores grid shard ShardRank of ShardCount and exchanges the arg-max
        InsertParam("ShardRank", &shardRank, INT_t, sizeof(int), sizeof(int));
        InsertParam("ShardCount", &shardCount, INT_t, sizeof(int), sizeof(int));
        InsertParam("CommBackend", &commBackend, INT_t, sizeof(int), sizeof(int));          // DPE_COMM_RCCL / DPE_COMM_HOSTFILES
        InsertParam("CommRendezvous", commRendezvous, CHAR_t, sizeof(commRendezvous), 0);
        ConfigOutput(0, "zVal", DOUBLE_t, STATE, HOST, 8, zVal, 0);
        ConfigOutput(1, "RVal", DOUBLE_t, COVARIANCE, HOST, 64, RVal, 0);
        ConfigOutput(2, "TimeGrid", DOUBLE_t, VALUE, HOST, VECTORLENGTH_ANY, nullptr, 0);
        ConfigOutput(3, "PosScores", FLOAT_t, GRID, HIP_DEVICE, VECTORLENGTH_ANY, nullptr, 0);
    }
    ~BatchCorrManifold() override { Stop(); }
    int Start(void *) override
    {
        if (Started) { std::clog << "[" << ModuleName << "] Start: Already Started." << std::endl; return 0; }
        DPE_MOD_CHECK_ABI();
        if (!inputs[0] || !inputs[1] || !inputs[6] || !inputs[7] || !inputs[11]) DPE_MOD_FAIL("Start: inputs not connected");
        const int dimP[4] = {posDim, posDim, posDim, posDim}, dimV[4] = {velDim, velDim, velDim, velDim};   // :2328-2329
        const double sp[4] = {spacing, spacing, spacing, spacing};                                         // :2332
        utils::build_grid((utils::ManifoldGridTypes)gridType, dimP, sp, posGrid, &timeGrid, false);
        utils::build_grid((utils::ManifoldGridTypes)gridType, dimV, sp, velGrid, nullptr, true);
        if (loadPosGrid) {                                                                                 // :2422-2448
            const int r = utils::load_grid_csv(loadPosGridFilename, (long long)posGrid.size() / 4, posGrid);
            if (r == -1) DPE_MOD_FAIL("Open loadGridFile failed: " << loadPosGridFilename);
            if (r) DPE_MOD_FAIL("loadGridFile " << loadPosGridFilename << " does not hold PosGridDimSize^4 rows of x,y,z,delta_t");
        }
        dpe_bcm_config cfg = {};
        cfg.samplingFrequency = *(double *)inputs[7]->Data;
        cfg.samplesPerWindow = (int32_t)((*(double *)inputs[6]->Data) * cfg.samplingFrequency);            // numSamps, :2536
        cfg.numFFTPoints = *(int *)inputs[11]->Data;
        cfg.lagHalfWidth = inputs[0]->AuxValue; cfg.binHalfWidth = inputs[1]->AuxValue;
        cfg.lPower = LPower; cfg.maxWindows = 1; cfg.maxChannels = DPE_MAX_CHAN;
        cfg.posGrid = posGrid.data(); cfg.velGrid = velGrid.data();
        cfg.posGridSize = (int64_t)posGrid.size() / 4; cfg.velGridSize = (int64_t)velGrid.size() / 4;
        if (shardCount > 1 || comm_requested()) {
            if (shardRank < 0 || shardRank >= shardCount) DPE_MOD_FAIL("Start: ShardRank " << shardRank << " not in [0, " << shardCount << ")");
            // contiguous index ranges, remainder to the first ranks (keeps the reference's index order, "t fastest")
            auto range = [&](int64_t G, int64_t &b, int64_t &e) {
                const int64_t base = G / shardCount, rem = G % shardCount;
                b = shardRank * base + (shardRank < rem ? shardRank : rem);
                e = b + base + (shardRank < rem ? 1 : 0);
            };
            int64_t pb, pe, vb, ve;
            range(cfg.posGridSize, pb, pe);
            range(cfg.velGridSize, vb, ve);
            cfg.posGrid = posGrid.data() + 4 * pb; cfg.posGridSize = pe - pb; cfg.posGridIndexOffset = pb;
            cfg.velGrid = velGrid.data() + 4 * vb; cfg.velGridSize = ve - vb; cfg.velGridIndexOffset = vb;
            if (dpe_comm_create(shardRank, shardCount, commRendezvous, commBackend, &comm)) DPE_MOD_FAIL("Start: " << dpe_last_error());
        }
        cfg.writeScores = 1;
        cfg.referencePair = referencePair ? 1 : 0;
        if (dpe_bcm_create(&cfg, &h)) return -1;
        dpe_bcm_set_graph(h, useGraph ? 1 : 0);
        const float *ps, *vs;
        dpe_bcm_scores(h, &ps, &vs);
        UpdateOutput(2, (uint32_t)timeGrid.size(), timeGrid.data(), 0);
        UpdateOutput(3, (uint32_t)cfg.posGridSize, (void *)ps, 0);
        for (int i = 0; i < 64; ++i) RVal[i] = (i % 9 == 0) ? 1.0 : 0.0;                                  // :2003-2011,2055-2063
        Started = true;
        return 0;
    }
    int Update(void *flowStream) override
    {
        if (!Started) DPE_MOD_FAIL("Error: Update() Failed due to SatPos not initialized");
        const int K = (int)inputs[8]->VectorLength;
        const int dimT = (int)timeGrid.size();
        if (deviceLoop) {   // the measurement is formed behind the scan by cuChanMgrDev's kernel; nothing to wait for here
            if (dpe_bcm_update_prepared(h, (const float *)inputs[0]->Data, (const float *)inputs[1]->Data, K, flow_stream(flowStream))) { Stop(); return -1; }
            return 0;
        }
        dpe_bcm_window win = {};
        std::memcpy(win.xCurrkk1, inputs[2]->Data, sizeof(double) * 8);                                    // re-read every Update, :2540
        std::memcpy(win.enu2ecef, inputs[12]->Data, sizeof(double) * 9);
        win.rxTime = *(double *)inputs[5]->Data;
        win.dopplerSign = *(int *)inputs[10]->Data;
        dpe_chan_end ch[DPE_MAX_CHAN];
        for (int k = 0; k < K; ++k) {
            std::memcpy(ch[k].satState, (const double *)inputs[4]->Data + ((size_t)k * dimT + dimT / 2) * 8, sizeof(double) * 8);   // :1775
            ch[k].codePhaseEnd = ((const double *)inputs[14]->Data)[k];
            ch[k].codeFrequency = ((const double *)inputs[8]->Data)[k];
            ch[k].carrierFrequency = ((const double *)inputs[9]->Data)[k];
            ch[k].cpRefTOW = ((const int *)inputs[16]->Data)[k];
            ch[k].cpElapsedEnd = ((const int *)inputs[17]->Data)[k];
            ch[k].cpRef = ((const int *)inputs[18]->Data)[k];
            ch[k].reserved = 0;
        }
        dpe_stream_t st = flow_stream(flowStream);
        if (dpe_bcm_update(h, (const float *)inputs[0]->Data, (const float *)inputs[1]->Data, 1, K, &win, ch, st)) { Stop(); return -1; }
        dpe_bcm_result r;
        if (comm) {
            // sharded grid: every rank ends with the same reduced keys and decodes the same global ML point
            uint64_t keys[2];
            if (dpe_bcm_exchange_keys(h, comm, keys, st)) DPE_MOD_FAIL("Update: " << dpe_last_error());
            if (dpe_bcm_results_from_keys(h, keys, 1, posGrid.data(), (int64_t)posGrid.size() / 4, velGrid.data(),
                                          (int64_t)velGrid.size() / 4, &r)) DPE_MOD_FAIL("Update: " << dpe_last_error());
        } else if (dpe_bcm_results(h, &r, st)) return -1;                                                  // synchronises, :2606-2632
        std::memcpy(zVal, r.zVal, sizeof(zVal));
        last = r;
        return 0;
    }
    int Stop() override
    {
        if (!Started) return 0;
        dpe_bcm_destroy(h);
        if (comm) dpe_comm_destroy(comm);
        comm = nullptr;
        h = nullptr;
        Started = false;
        return 0;
    }
    const dpe_bcm_result &LastResult() const { return last; }
    dpe_bcm *Handle() const { return h; }
    dpe_comm *Comm() const { return comm; }      // the communicator of a sharded grid (nullptr: one GPU scans the whole grid)
    const std::vector<double> &TimeGrid() const { return timeGrid; }
    const std::vector<double> &PosGrid() const { return posGrid; }
    const std::vector<double> &VelGrid() const { return velGrid; }

  private:
    dpe_bcm *h = nullptr;
    bool Started = false, loadPosGrid = false;
    int posDim = 25, velDim = 25, gridType = 0, LPower = 1;
    bool useGraph = false, deviceLoop = false;
    bool referencePair = false;   // reproduce the reference's floor(idx) / floor(idx + 1) pair where it double-counts (dpe_hip.h)
    int shardRank = 0, shardCount = 1, commBackend = DPE_COMM_RCCL;
    char commRendezvous[512] = "";
    dpe_comm *comm = nullptr;
    bool comm_requested() const { return commRendezvous[0] != 0; }
    float spacing = 1.0f;
    char gridLog[512] = "", loadPosGridFilename[512] = "";
    std::vector<double> posGrid, velGrid, timeGrid;
    double zVal[8] = {}, RVal[64] = {};
    dpe_bcm_result last = {};
};

// ------------------------------------------------------------------------------------------------
class cuEKF : public Module {   // EnableEKF=false: EKF_PassMeas copies zVal to both state ports; true: the filter
  public:
    cuEKF()
    {
        ModuleName = "cuEKF";
        AllocateInputs(5);
        AllocateOutputs(3);
        ConfigExpectedInput(0, "InitX", DOUBLE_t, STATE, VECTORLENGTH_ANY);         // cuekf.cu:238-244
        ConfigExpectedInput(1, "zVal", DOUBLE_t, STATE, VECTORLENGTH_ANY);
        ConfigExpectedInput(2, "RVal", DOUBLE_t, COVARIANCE, VECTORLENGTH_ANY);
        ConfigExpectedInput(3, "InitP", DOUBLE_t, COVARIANCE, VECTORLENGTH_ANY);
        ConfigExpectedInput(4, "InitK", INT_t, VALUE, 1);
        ConfigOutput(0, "xCurrk1k1", DOUBLE_t, STATE, HOST, 8, xk1k1, 0);           // :277-279 (host memory here)
        ConfigOutput(1, "xCurrkk1", DOUBLE_t, STATE, HOST, 8, xkk1, 0);
        ConfigOutput(2, "PCurrkk1", DOUBLE_t, COVARIANCE, HOST, 64, Pkk1, 0);
        InsertParam("EnableEKF", &enable, BOOL_t, sizeof(bool), sizeof(bool));
        InsertParam("SampleLength", &T, DOUBLE_t, sizeof(double), sizeof(double));
    }
    ~cuEKF() override { Stop(); }
    int Start(void *) override
    {
        if (!inputs[0]) DPE_MOD_FAIL("Start: InitX not connected");
        if (inputs[0]->VectorLength != 8) DPE_MOD_FAIL("Start: the filter is built for the 8-state DPE model");
        std::memcpy(xk1k1, inputs[0]->Data, sizeof(xk1k1));   // cuekf.cu:338-344
        std::memcpy(xkk1, inputs[0]->Data, sizeof(xkk1));
        for (int i = 0; i < 64; ++i) Pkk1[i] = (i % 9 == 0) ? 1.0 : 0.0;
        if (enable) {
            dpe_ekf_config cfg = {};
            cfg.sampleLength = T;
            cfg.coupleVelocity = 1;                            // EKF_MakeDPERandomWalkFMatrix, :460
            std::memcpy(cfg.x0, xk1k1, sizeof(cfg.x0));
            if (inputs[3]) std::memcpy(cfg.P0, inputs[3]->Data, sizeof(cfg.P0));   // :352
            else for (int i = 0; i < 64; ++i) cfg.P0[i] = (i % 9 == 0) ? 1.0 : 0.0;
            if (dpe_ekf_create(&cfg, &h)) return -1;
        }
        return 0;
    }
    int Update(void *) override
    {
        if (!enable) {
            std::memcpy(xk1k1, inputs[1]->Data, sizeof(xk1k1));   // EKF_PassMeas, cuekf.cu:147-159
            std::memcpy(xkk1, inputs[1]->Data, sizeof(xkk1));
            return 0;
        }
        if (!h) DPE_MOD_FAIL("Error: Update() Failed due to EKF not initialized");
        // StepUpdate, then StepPredict once per new measurement (:575-588)
        if (dpe_ekf_step_update(h, (const double *)inputs[1]->Data, (const double *)inputs[2]->Data)) return -1;
        if (dpe_ekf_step_predict(h)) return -1;
        return dpe_ekf_state(h, xk1k1, xkk1, nullptr, Pkk1, nullptr, nullptr);
    }
    int Stop() override
    {
        if (h) dpe_ekf_destroy(h);
        h = nullptr;
        return 0;
    }

  private:
    dpe_ekf *h = nullptr;
    bool enable = false;
    double T = 0.02, xk1k1[8] = {}, xkk1[8] = {}, Pkk1[64] = {};
};

// ------------------------------------------------------------------------------------------------
class cuChanMgr : public Module {
  public:
    cuChanMgr()
    {
        ModuleName = "cuChanMgr";
        AllocateInputs(14);
        AllocateOutputs(18);
        const char *in[14] = {"InitEph", "InitPRN", "InitCodePhase", "InitCarrierPhase", "InitCodeFrequency", "InitCarrierFrequency",
                              "InitElapsedCodePeriods", "InitReferenceCodePeriods", "InitCPRefTOW", "InitRXTime", "xCurrk1k1",
                              "SampleLength", "xCurrkk1", "TimeGrid"};
        const DataType_t idt[14] = {UNDEFINED_t, CHAR_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, INT_t, INT_t, INT_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t};
        const ValueType_t ivt[14] = {EPHEMS, VALUE, VALUE, VALUE, FREQUENCY_HZ, FREQUENCY_HZ, VALUE, VALUE, VALUE, VALUE, STATE, VALUE, STATE, VALUE};
        for (int i = 0; i < 14; ++i) ConfigExpectedInput(i, in[i], idt[i], ivt[i], i == 11 ? 1 : VECTORLENGTH_ANY);   // cuchanmgr.cu:942-955
        InsertParam("DopplerSign", &dopplerSign, INT_t, sizeof(int), sizeof(int));
        const char *out[18] = {"rxTime", "txTime", "CodePhaseStart", "CarrierPhaseStart", "CodePhaseEnd", "CarrierPhaseEnd",
                               "CodeFrequency", "CarrierFrequency", "SatStates", "DopplerSign", "ValidPRNs", "cpReference",
                               "cpElapsedStart", "cpElapsedEnd", "ENU2ECEFMat", "SatStatesOld", "cpRef", "cpRefTOW"};
        const DataType_t odt[18] = {DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, INT_t,
                                    CHAR_t, INT_t, INT_t, INT_t, DOUBLE_t, DOUBLE_t, INT_t, INT_t};
        const ValueType_t ovt[18] = {VALUE, VALUE, VALUE, VALUE, VALUE, VALUE, FREQUENCY_HZ, FREQUENCY_HZ, STATE, VALUE, VALUE, VALUE,
                                     VALUE, VALUE, VALUE, STATE, VALUE, VALUE};
        for (int i = 0; i < 18; ++i) ConfigOutput(i, out[i], odt[i], ovt[i], HOST, i == 0 || i == 9 ? 1 : (i == 14 ? 9 : VECTORLENGTH_ANY), nullptr, 0);   // :973-990
    }
    ~cuChanMgr() override { Stop(); }
    int Start(void *) override
    {
        if (h) { std::clog << "[" << ModuleName << "] Start: Already Started." << std::endl; return 0; }
        DPE_MOD_CHECK_ABI();
        for (int i = 0; i < 14; ++i)
            if (!inputs[i]) DPE_MOD_FAIL("Start: input " << expectedInputs[i].Name << " not connected");
        K = (int)inputs[1]->VectorLength;
        std::vector<dpe_chm_init_chan> init(K);
        for (int k = 0; k < K; ++k) {
            init[k].prn = ((const uint8_t *)inputs[1]->Data)[k];
            init[k].codePhase = ((const double *)inputs[2]->Data)[k];
            init[k].carrierPhase = ((const double *)inputs[3]->Data)[k];
            init[k].codeFrequency = ((const double *)inputs[4]->Data)[k];
            init[k].carrierFrequency = ((const double *)inputs[5]->Data)[k];
            init[k].cpElapsed = ((const int *)inputs[6]->Data)[k];
            init[k].cpReference = ((const int *)inputs[7]->Data)[k];
            init[k].cpRefTOW = ((const int *)inputs[8]->Data)[k];
            std::memcpy(init[k].eph, (const double *)inputs[0]->Data + (size_t)k * DPE_EPH_N, sizeof(double) * DPE_EPH_N);
        }
        dpe_chm_config cfg = {K, dopplerSign, *(double *)inputs[11]->Data, *(double *)inputs[9]->Data};
        if (dpe_chm_create(&cfg, init.data(), &h)) return -1;
        dimT = (int)inputs[13]->VectorLength;
        if (dpe_chm_start(h, (const double *)inputs[10]->Data, (const double *)inputs[12]->Data, (const double *)inputs[13]->Data, dimT)) return -1;
        return Publish();
    }
    int Update(void *) override
    {
        if (!h) DPE_MOD_FAIL("Error: Update() Failed due to SatPos not initialized");
        if (dpe_chm_update(h, (const double *)inputs[10]->Data, (const double *)inputs[12]->Data, (const double *)inputs[13]->Data, dimT)) return -1;
        return Publish();
    }
    int Stop() override
    {
        if (h) dpe_chm_destroy(h);
        h = nullptr;
        return 0;
    }

  private:
    int Publish()
    {
        start.resize(K); end.resize(K); batch.resize((size_t)K * dimT * 8);
        if (dpe_chm_outputs(h, start.data(), end.data(), &win, batch.data())) return -1;
        rcS.resize(K); riS.resize(K); rcE.resize(K); fc.resize(K); fi.resize(K); prn.resize(K);
        cpRef.resize(K); cpS.resize(K); cpE.resize(K); tow.resize(K);
        for (int k = 0; k < K; ++k) {
            rcS[k] = start[k].codePhaseStart; riS[k] = start[k].carrierPhaseStart; rcE[k] = end[k].codePhaseEnd;
            fc[k] = start[k].codeFrequency; fi[k] = start[k].carrierFrequency; prn[k] = (uint8_t)start[k].prn;
            cpRef[k] = start[k].cpReference; cpS[k] = start[k].cpElapsedStart; cpE[k] = end[k].cpElapsedEnd; tow[k] = end[k].cpRefTOW;
        }
        rxTime = win.rxTime;
        const uint32_t uK = (uint32_t)K;
        UpdateOutput(0, 1, &rxTime, 0);           UpdateOutput(2, uK, rcS.data(), 0);   UpdateOutput(3, uK, riS.data(), 0);
        UpdateOutput(4, uK, rcE.data(), 0);       UpdateOutput(6, uK, fc.data(), 0);    UpdateOutput(7, uK, fi.data(), 0);
        UpdateOutput(8, uK * dimT, batch.data(), 0);   UpdateOutput(9, 1, &dopplerSign, 0);   UpdateOutput(10, uK, prn.data(), 0);
        UpdateOutput(11, uK, cpRef.data(), 0);    UpdateOutput(12, uK, cpS.data(), 0);  UpdateOutput(13, uK, cpE.data(), 0);
        UpdateOutput(14, 9, win.enu2ecef, 0);     UpdateOutput(16, uK, cpRef.data(), 0); UpdateOutput(17, uK, tow.data(), 0);
        // txTime (1), CarrierPhaseEnd (5), SatStatesOld (15): not consumed by the active kernels
        UpdateOutput(1, uK, rcE.data(), 0); UpdateOutput(5, uK, riS.data(), 0); UpdateOutput(15, uK, batch.data(), 0);
        return 0;
    }
    dpe_chanmgr *h = nullptr;
    int K = 0, dimT = 1, dopplerSign = 1;
    double rxTime = 0;
    std::vector<dpe_chan_start> start;
    std::vector<dpe_chan_end> end;
    dpe_bcm_window win = {};
    std::vector<double> batch, rcS, riS, rcE, fc, fi;
    std::vector<uint8_t> prn;
    std::vector<int> cpRef, cpS, cpE, tow;
};

// ------------------------------------------------------------------------------------------------
// cuChanMgr as the reference has it -- device state, device port arrays -- with the measurement hand-over (BCM_MakePosMeas /
// MakeVelMeas), the pass-through filter (EKF_PassMeas, EnableEKF = false) and the channel update in ONE kernel behind the scan
// (dpe_chm_dev_step).  Update() only enqueues; the fix of window n - FixLag is collected from the pinned ring and written to the
// X-file (the XECEFLogger's "%f, " rows, datalogger.cu:160-203), the rest at Stop().
class cuChanMgrDev : public Module {
  public:
    cuChanMgrDev(BatchCorrScores *bcs_, BatchCorrManifold *bcm_) : bcs(bcs_), bcm(bcm_)
    {
        ModuleName = "cuChanMgr";
        AllocateInputs(13);
        AllocateOutputs(18);
        const char *in[13] = {"InitEph", "InitPRN", "InitCodePhase", "InitCarrierPhase", "InitCodeFrequency", "InitCarrierFrequency",
                              "InitElapsedCodePeriods", "InitReferenceCodePeriods", "InitCPRefTOW", "InitRXTime", "InitX", "SampleLength",
                              "InitP"};   // InitP: cuEKF's input (cuekf.cu:352), optional, read with EnableEKF
        const DataType_t idt[13] = {UNDEFINED_t, CHAR_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, INT_t, INT_t, INT_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t};
        const ValueType_t ivt[13] = {EPHEMS, VALUE, VALUE, VALUE, FREQUENCY_HZ, FREQUENCY_HZ, VALUE, VALUE, VALUE, VALUE, STATE, VALUE, COVARIANCE};
        for (int i = 0; i < 13; ++i) ConfigExpectedInput(i, in[i], idt[i], ivt[i], i == 11 ? 1 : VECTORLENGTH_ANY);
        InsertParam("DopplerSign", &dopplerSign, INT_t, sizeof(int), sizeof(int));
        InsertParam("EnableEKF", &enableEkf, BOOL_t, sizeof(bool), sizeof(bool));   // cuEKF's parameter: the filter runs in this module's measurement kernel
        InsertParam("FixLag", &fixLag, INT_t, sizeof(int), sizeof(int));
        InsertParam("XFilename", xFilename, CHAR_t, sizeof(xFilename), 0);
        const char *out[18] = {"rxTime", "txTime", "CodePhaseStart", "CarrierPhaseStart", "CodePhaseEnd", "CarrierPhaseEnd",
                               "CodeFrequency", "CarrierFrequency", "SatStates", "DopplerSign", "ValidPRNs", "cpReference",
                               "cpElapsedStart", "cpElapsedEnd", "ENU2ECEFMat", "SatStatesOld", "cpRef", "cpRefTOW"};
        const DataType_t odt[18] = {DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, DOUBLE_t, INT_t,
                                    CHAR_t, INT_t, INT_t, INT_t, DOUBLE_t, DOUBLE_t, INT_t, INT_t};
        const ValueType_t ovt[18] = {VALUE, VALUE, VALUE, VALUE, VALUE, VALUE, FREQUENCY_HZ, FREQUENCY_HZ, STATE, VALUE, VALUE, VALUE,
                                     VALUE, VALUE, VALUE, STATE, VALUE, VALUE};
        for (int i = 0; i < 18; ++i) ConfigOutput(i, out[i], odt[i], ovt[i], HIP_DEVICE, i == 0 || i == 9 ? 1 : (i == 14 ? 9 : VECTORLENGTH_ANY), nullptr, 0);   // cuchanmgr.cu:973-990
    }
    ~cuChanMgrDev() override { Stop(); }
    int Start(void *flowStream) override
    {
        if (h) { std::clog << "[" << ModuleName << "] Start: Already Started." << std::endl; return 0; }
        DPE_MOD_CHECK_ABI();
        for (int i = 0; i < 12; ++i)
            if (!inputs[i]) DPE_MOD_FAIL("Start: input " << expectedInputs[i].Name << " not connected");
        if (!bcs || !bcm || !bcs->Handle() || !bcm->Handle()) DPE_MOD_FAIL("Start: BatchCorrScores / BatchCorrManifold must be started first");
        if (fixLag < 1 || fixLag > 24) DPE_MOD_FAIL("Start: FixLag " << fixLag << " not in [1, 24]");
        K = (int)inputs[1]->VectorLength;
        std::vector<dpe_chm_init_chan> init(K);
        for (int k = 0; k < K; ++k) {
            init[k].prn = ((const uint8_t *)inputs[1]->Data)[k];
            init[k].codePhase = ((const double *)inputs[2]->Data)[k];
            init[k].carrierPhase = ((const double *)inputs[3]->Data)[k];
            init[k].codeFrequency = ((const double *)inputs[4]->Data)[k];
            init[k].carrierFrequency = ((const double *)inputs[5]->Data)[k];
            init[k].cpElapsed = ((const int *)inputs[6]->Data)[k];
            init[k].cpReference = ((const int *)inputs[7]->Data)[k];
            init[k].cpRefTOW = ((const int *)inputs[8]->Data)[k];
            std::memcpy(init[k].eph, (const double *)inputs[0]->Data + (size_t)k * DPE_EPH_N, sizeof(double) * DPE_EPH_N);
        }
        const std::vector<double> &tg = bcm->TimeGrid();
        dpe_chm_config cfg = {K, dopplerSign, *(double *)inputs[11]->Data, *(double *)inputs[9]->Data};
        if (dpe_chm_dev_create(&cfg, init.data(), tg.data(), (int32_t)tg.size(), &h)) DPE_MOD_FAIL("Start: " << dpe_last_error());
        if (dpe_chm_dev_attach(h, bcs->Handle(), bcm->Handle(), fixLag + 8)) DPE_MOD_FAIL("Start: " << dpe_last_error());
        // sharded grid (BatchCorrManifold's ShardRank / ShardCount / Comm* parameters): the exchange of the arg-max keys sits between
        // the scan and this module's measurement kernel, which decodes the reduced keys against the global grids
        if (enableEkf) {   // dsp::cuEKF with EnableEKF = true (cuekf.cu:338-352,460): F couples the velocities over one window
            dpe_ekf_config ec = {};
            ec.sampleLength = *(double *)inputs[11]->Data;
            ec.coupleVelocity = 1;
            std::memcpy(ec.x0, inputs[10]->Data, sizeof(ec.x0));
            if (inputs[12]) std::memcpy(ec.P0, inputs[12]->Data, sizeof(ec.P0));
            else for (int i = 0; i < 64; ++i) ec.P0[i] = (i % 9 == 0) ? 1.0 : 0.0;
            if (dpe_chm_dev_set_ekf(h, &ec)) DPE_MOD_FAIL("Start: " << dpe_last_error());
        }
        if (bcm->Comm() &&
            dpe_chm_dev_set_shard(h, bcm->Comm(), bcm->PosGrid().data(), (int64_t)bcm->PosGrid().size() / 4, bcm->VelGrid().data(),
                                  (int64_t)bcm->VelGrid().size() / 4))
            DPE_MOD_FAIL("Start: " << dpe_last_error());
        if (xFilename[0]) {
            fp = std::fopen(xFilename, "w");
            if (!fp) DPE_MOD_FAIL("Unable to open file: " << xFilename);
        }
        if (dpe_chm_dev_start(h, (const double *)inputs[10]->Data, flow_stream(flowStream))) DPE_MOD_FAIL("Start: " << dpe_last_error());
        dpe_bcs_ports_dev pb;
        dpe_bcm_ports_dev pm;
        const double *rx = nullptr, *z = nullptr;
        double *x1 = nullptr, *xk = nullptr;
        dpe_chm_dev_ports(h, &pb, &pm, &rx, &x1, &xk, &z);
        const uint32_t uK = (uint32_t)K, dimT = (uint32_t)tg.size();
        UpdateOutput(0, 1, (void *)rx, 0);                       UpdateOutput(2, uK, (void *)pb.codePhaseStart, 0);
        UpdateOutput(3, uK, (void *)pb.carrierPhaseStart, 0);    UpdateOutput(4, uK, (void *)pm.codePhaseEnd, 0);
        UpdateOutput(6, uK, (void *)pb.codeFrequency, 0);        UpdateOutput(7, uK, (void *)pb.carrierFrequency, 0);
        UpdateOutput(8, uK * dimT, (void *)pm.satStates, 0);     UpdateOutput(9, 1, (void *)pm.dopplerSign, 0);
        UpdateOutput(10, uK, (void *)pb.validPRNs, 0);           UpdateOutput(11, uK, (void *)pb.cpReference, 0);
        UpdateOutput(12, uK, (void *)pb.cpElapsedStart, 0);      UpdateOutput(13, uK, (void *)pm.cpElapsedEnd, 0);
        UpdateOutput(14, 9, (void *)pm.enu2ecef, 0);             UpdateOutput(16, uK, (void *)pm.cpRef, 0);
        UpdateOutput(17, uK, (void *)pm.cpRefTOW, 0);
        // txTime (1), CarrierPhaseEnd (5), SatStatesOld (15): not consumed by the active kernels
        UpdateOutput(1, uK, (void *)pm.codePhaseEnd, 0); UpdateOutput(5, uK, (void *)pb.carrierPhaseStart, 0); UpdateOutput(15, uK * dimT, (void *)pm.satStates, 0);
        enq = got = 0;
        return 0;
    }
    int Update(void *flowStream) override
    {
        if (!h) DPE_MOD_FAIL("Error: Update() Failed due to SatPos not initialized");
        if (dpe_chm_dev_step(h, flow_stream(flowStream))) DPE_MOD_FAIL("Update: " << dpe_last_error());
        ++enq;
        while (enq - got > fixLag)
            if (Collect(-1)) return -1;
        return 0;
    }
    // Collects the fixes still on their way.  Call before the flow stops.
    int Drain()
    {
        while (h && got < enq)
            if (Collect(2000000)) return -1;
        return 0;
    }
    int Stop() override
    {
        if (h) {
            if (got < enq) std::clog << "[" << ModuleName << "] Stop: " << (enq - got) << " fixes not collected (Drain() before Stop)" << std::endl;
            dpe_chm_dev_destroy(h);
        }
        h = nullptr;
        if (fp) std::fclose(fp);
        fp = nullptr;
        return 0;
    }
    const dpe_fix_record &LastFix() const { return last; }
    long long Collected() const { return got; }

  private:
    int Collect(int timeoutMicros)
    {
        dpe_fix_record r;
        const int rc = dpe_chm_dev_fix(h, got, &r, timeoutMicros);
        if (rc) { std::cerr << "[" << ModuleName << "] fix " << got << ": " << (rc == 1 ? "timed out" : dpe_last_error()) << std::endl; return -1; }
        if (r.status && !warned) { std::clog << "[" << ModuleName << "] status " << r.status << " at window " << got << std::endl; warned = true; }
        if (fp)
            for (int i = 0; i < 8; ++i) std::fprintf(fp, i + 1 < 8 ? "%f, " : "%f\n", r.zVal[i]);
        last = r;
        ++got;
        return 0;
    }
    BatchCorrScores *bcs;
    BatchCorrManifold *bcm;
    dpe_chm_dev *h = nullptr;
    int K = 0, dopplerSign = 1, fixLag = 8;
    bool enableEkf = false;
    long long enq = 0, got = 0;
    bool warned = false;
    char xFilename[512] = "";
    FILE *fp = nullptr;
    dpe_fix_record last = {};
};

// ------------------------------------------------------------------------------------------------
class DataLogger : public Module {   // CSV rows "%f, %f, ... %f\n" of a DOUBLE_t host port (datalogger.cu:156-203)
  public:
    explicit DataLogger(const char *name)
    {
        ModuleName = name;
        AllocateInputs(1);
        ConfigExpectedInput(0, "Data", DATATYPE_ANY, VALUETYPE_ANY, VECTORLENGTH_ANY);
        InsertParam("Filename", Filename, CHAR_t, sizeof(Filename), 0);
        InsertParam("CSV", &csv, BOOL_t, sizeof(bool), sizeof(bool));
    }
    ~DataLogger() override { Stop(); }
    int Start(void *) override
    {
        fp = std::fopen(Filename, csv ? "w" : "wb");
        if (!fp) DPE_MOD_FAIL("Unable to open file: " << Filename);
        return 0;
    }
    int Update(void *) override
    {
        if (!fp || !inputs[0]) DPE_MOD_FAIL("Update: not started");
        const double *d = (const double *)inputs[0]->Data;
        const uint32_t n = inputs[0]->VectorLength;
        if (!csv) return std::fwrite(d, sizeof(double), n, fp) == n ? 0 : -1;
        for (uint32_t i = 0; i < n; ++i) std::fprintf(fp, i + 1 < n ? "%f, " : "%f\n", d[i]);
        return 0;
    }
    int Stop() override
    {
        if (fp) std::fclose(fp);
        fp = nullptr;
        return 0;
    }

  private:
    char Filename[512] = "";
    bool csv = true;
    FILE *fp = nullptr;
};

}  // namespace dsp
