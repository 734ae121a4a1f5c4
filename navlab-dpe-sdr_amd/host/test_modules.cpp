// test_modules.cpp -- host-side checks of the dsp::Module / dsp::Flow mirror against the reference's interface rules
// (cudarecv/modules/src/module.cpp:21-63,284-320; cudarecv/dsp/src/flow.cu:28-87,212-324).  No GPU work: parameter
// typing, port validation by ValueType + VectorLength, wiring by names, Start roll-back, Update-before-Start errors.
// Built by __graft_entry__.build() as navlab-dpe-sdr_amd/test_modules; run by tests/test_abi_cpu.py.
#include <cstdio>

#include "modules.hpp"

using namespace dsp;

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

struct Probe : Module {   // records the order of Start / Stop calls; can be told to fail
    static std::string log;
    bool failStart = false;
    int ivalue = 0;
    float fvalue = 0.f;
    char text[8] = "";
    double out8[8] = {};
    explicit Probe(const char *name)
    {
        ModuleName = name;
        AllocateInputs(2);
        AllocateOutputs(1);
        ConfigExpectedInput(0, "State", DOUBLE_t, STATE, 8);
        ConfigExpectedInput(1, "Anything", DOUBLE_t, VALUE, VECTORLENGTH_ANY);
        ConfigOutput(0, "State", DOUBLE_t, STATE, HOST, 8, out8, 0);
        InsertParam("Count", &ivalue, INT_t, sizeof(int), sizeof(int));
        InsertParam("Gain", &fvalue, FLOAT_t, sizeof(float), sizeof(float));
        InsertParam("Label", text, CHAR_t, sizeof(text), 0);
    }
    int Start(void *) override { log += "S" + ModuleName; return failStart ? -1 : 0; }
    int Stop() override { log += "X" + ModuleName; return 0; }
    int Update(void *) override { return 0; }
    int dupParam() { return InsertParam("Count", &ivalue, INT_t, sizeof(int), sizeof(int)); }
};
std::string Probe::log;

int main()
{
    std::cerr.setstate(std::ios::failbit);   // the modules report to std::cerr like the reference; keep the test output clean
    std::clog.setstate(std::ios::failbit);
    {   // parameters: key must exist, datatype must match, size must fit the capacity (module.cpp:40-63)
        Probe p("A");
        EXPECT(p.SetParam("Count", 7) == 0 && p.ivalue == 7);
        EXPECT(p.SetParam("Count", 7.0f) == -1);
        EXPECT(p.SetParam("Gain", 2.5f) == 0 && p.fvalue == 2.5f);
        EXPECT(p.SetParam("Gain", 2.5) == -1);                      // double into a FLOAT_t parameter
        EXPECT(p.SetParam("Nope", 1) == -1);
        EXPECT(p.SetParam("Label", "short") == 0 && std::string(p.text) == "short");
        EXPECT(p.SetParam("Label", "much too long") == -1 && std::string(p.text) == "short");
        EXPECT(p.dupParam() == -1);
        int v = 0;
        EXPECT(p.GetParam("Count", &v) == 0 && v == 7);
    }
    {   // ports: by name, ValueType and VectorLength are checked, the datatype is not (module.cpp:284-310)
        Flow f;
        Probe *a = new Probe("A"), *b = new Probe("B");
        f.Add(a); f.Add(b);
        EXPECT(f.ConnectPort("A", "State", "B", "State") == 0);
        EXPECT(f.ConnectPort("A", "State", "B", "Anything") == -1);   // STATE into a VALUE input
        EXPECT(f.ConnectPort("A", "Missing", "B", "State") == -1);
        EXPECT(f.ConnectPort("A", "State", "B", "Missing") == -1);
        EXPECT(f.ConnectPort("Nobody", "State", "B", "State") == -1);
        EXPECT(f.SetModParam("B", "Count", 3) == 0 && b->ivalue == 3);
        EXPECT(f.SetModParam("Nobody", "Count", 3) == -1);
    }
    {   // Flow::Start rolls back the modules already started when one fails (flow.cu:35-44)
        Flow f;
        Probe *a = new Probe("A"), *b = new Probe("B"), *c = new Probe("C");
        b->failStart = true;
        f.Add(a); f.Add(b); f.Add(c);
        Probe::log.clear();
        EXPECT(f.Start(nullptr) == -1);
        EXPECT(Probe::log == "SASBXA");
        Probe::log.clear();
        b->failStart = false;
        EXPECT(f.Start(nullptr) == 0 && f.Step() == 0);
        f.Stop();
        EXPECT(Probe::log == "SASBSCXAXBXC");
    }
    {   // the path's modules: vector-length contract of their ports, Update before Start, missing files
        BatchCorrScores bcs;
        BatchCorrManifold bcm;
        cuEKF ekf;
        Port *p = nullptr;
        EXPECT(bcs.GetOutput((unsigned char)bcs.GetOutputID("CodeScores"), &p) == 0 && p->ValueType == VALUE_CMPX);
        EXPECT(bcs.GetOutputID("NoSuchPort") == -1);
        EXPECT(bcm.GetInputID("xCurrkk1") >= 0 && bcm.GetInputID("ENU2ECEFMat") >= 0 && bcm.GetInputID("SatStates") >= 0);
        EXPECT(bcm.GetOutputID("zVal") == 0 && bcm.GetOutputID("RVal") == 1 && bcm.GetOutputID("PosScores") == 3);
        EXPECT(bcs.Update(nullptr) == -1);                            // "batch correlator not initialized"
        EXPECT(bcm.Update(nullptr) == -1);
        EXPECT(bcs.SetParam("LagHalfWidth", 5) == 0 && bcs.SetParam("LagHalfWidth", 5.0) == -1);
        EXPECT(bcm.SetParam("GridType", 1) == 0 && bcm.SetParam("LoadPosGridFilename", "/nonexistent/rngrid3.csv") == 0);
        EXPECT(ekf.SetParam("EnableEKF", true) == 0 && ekf.SetParam("EnableEKF", 1) == -1);   // BOOL_t, not INT_t
        DPInit init;
        EXPECT(init.SetParam("HandoffFilename", "/nonexistent/handoff.csv") == 0);
        EXPECT(init.Start(nullptr) == -1);
        SampleBlock sb;
        EXPECT(sb.SetParam("Filename", "/nonexistent/samples.dat") == 0 && sb.SetParam("SamplingFrequency", 2.5e6) == 0);
        EXPECT(sb.Update(nullptr) == -1);                             // not started
        // the reference's network-source keys are accepted (sampleblock.cu:51-52,57); a non-file source is refused at Start
        EXPECT(sb.SetParam("Hostname", "192.168.10.7") == 0 && sb.SetParam("PortNo", 49152) == 0);
        EXPECT(sb.SetParam("SampleLength", 0.02) == 0 && sb.SetParam("InputSourceType", (char)1) == 0);
        EXPECT(sb.Start(nullptr) == -1);
    }
    std::printf(failures ? "%d failure(s)\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
