// dsp.hpp -- host-side mirror of the reference's module/port interface for the hot path.
//
// Same names and semantics as cudarecv/dsp/inc/dsp.h:28-121 (enums, Param, Port, ExpectedPort) and
// cudarecv/modules/inc/module.h:13-144 (dsp::Module), so that DPEFlow::LoadFlow's SetModParam /
// ConnectPort table (cudarecv/dsp/src/dpeflow.cpp:67-213) applies unchanged.  Differences:
// Port::VectorLength is 32-bit (reference: unsigned short, dsp.h:109 -- S = 500000 at 25 Msps does not
// fit) and MemLoc_t gains HIP_DEVICE as an alias of the reference's CUDA_DEVICE value.
#pragma once
#include <cstdint>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <chrono>
#include <vector>

namespace dsp {

const uint32_t VECTORLENGTH_ANY = 0;

enum ValueType_t : uint8_t {
    VALUETYPE_ANY = 0, VALUE, VALUE_CMPX, RATIO, RATIO_DB, FREQUENCY_HZ, FREQUENCY_RAD, PHASE, MAGNITUDE,
    RS_CORR_OUT, SS_CORR_OUT, CHANNEL, STATE, COVARIANCE, FUNCTION_PTR, EPHEMS, GRID
};
enum MemLoc_t : uint8_t { HOST = 0, CUDA_DEVICE = 1, HIP_DEVICE = 1 };
enum DataType_t : uint8_t {
    DATATYPE_ANY = 0, UNDEFINED_t, FLOAT_t, DOUBLE_t, FIXED_Q15_t, FIXED_Q31_t, FIXED_I15Q16_t, CHAR_t, STRING_t,
    INT_t, BOOL_t, CUFFTCOMP_t
};

struct Param {
    void *Ptr;
    DataType_t Datatype;
    unsigned int Capacity;  // bytes
    unsigned int Size;      // bytes
};

struct Port {
    char Name[32];
    DataType_t Datatype;
    signed char Exponent;
    ValueType_t ValueType;
    MemLoc_t MemLoc;
    uint32_t VectorLength;
    void *Data;
    int AuxValue;
};

struct ExpectedPort {
    char Name[32];
    DataType_t Datatype;
    ValueType_t ValueType;
    uint32_t VectorLength;
};

// Base class of every module: Start / Update / Stop, ports by index or name, typed parameter map.
class Module {
  public:
    virtual ~Module()
    {
        delete[] expectedInputs;
        delete[] inputs;
        delete[] outputs;
    }
    virtual int Update(void *flowStream) = 0;
    virtual int Start(void * /*flowStream*/) { return 0; }
    virtual int Stop() { return 0; }

    std::string GetModuleName() const { return ModuleName; }

    int GetInputID(const char *name) const
    {
        for (unsigned i = 0; i < NumInputs; ++i)
            if (std::strcmp(name, expectedInputs[i].Name) == 0) return (int)i;
        std::cerr << "[" << ModuleName << "] GetInputID: " << name << " does not exist." << std::endl;
        return -1;
    }
    int GetOutputID(const char *name) const
    {
        for (unsigned i = 0; i < NumOutputs; ++i)
            if (std::strcmp(name, outputs[i].Name) == 0) return (int)i;
        std::cerr << "[" << ModuleName << "] GetOutputID: " << name << " does not exist." << std::endl;
        return -1;
    }
    // accepts the connection if the value type matches (or is ANY) and the vector length is equal or unconstrained
    int SetInput(unsigned char id, Port *in)
    {
        if (id >= NumInputs) {
            std::cerr << "[" << ModuleName << "] No such input." << std::endl;
            return -1;
        }
        const ExpectedPort &e = expectedInputs[id];
        if (e.ValueType != in->ValueType && e.ValueType != VALUETYPE_ANY) {
            std::cerr << "[" << ModuleName << "] Invalid input value type." << std::endl;
            return -1;
        }
        if (e.VectorLength != VECTORLENGTH_ANY && e.VectorLength != in->VectorLength) {
            std::cerr << "[" << ModuleName << "] Invalid input vector length." << std::endl
                      << "Input size: " << in->VectorLength << std::endl
                      << "Expected size: " << e.VectorLength << std::endl;
            return -1;
        }
        inputs[id] = in;
        return 0;
    }
    int GetOutput(unsigned char id, Port **out)
    {
        if (id >= NumOutputs) {
            std::cerr << "[" << ModuleName << "] No such output." << std::endl;
            return -1;
        }
        *out = &outputs[id];
        return 0;
    }

    int SetParam(const std::string &key, Param *p)
    {
        auto it = Params.find(key);
        if (it == Params.end()) {
            std::cerr << "[" << ModuleName << "] SetParam: Parameter with key \"" << key << "\" does not exist." << std::endl;
            return -1;
        }
        if (it->second.Datatype != p->Datatype) {
            std::cerr << "[" << ModuleName << "] SetParam: Parameter with key \"" << key << "\" datatype does not match." << std::endl;
            return -1;
        }
        if (it->second.Capacity < p->Size) {
            std::cerr << "[" << ModuleName << "] SetParam: Parameter with key \"" << key << "\" has size larger than capacity." << std::endl;
            return -1;
        }
        it->second.Size = p->Size;
        std::memcpy(it->second.Ptr, p->Ptr, p->Size);
        return 0;
    }
    int SetParam(const std::string &key, int v) { return setTyped(key, &v, INT_t, sizeof(int)); }
    int SetParam(const std::string &key, char v) { return setTyped(key, &v, CHAR_t, sizeof(char)); }
    int SetParam(const std::string &key, float v) { return setTyped(key, &v, FLOAT_t, sizeof(float)); }
    int SetParam(const std::string &key, double v) { return setTyped(key, &v, DOUBLE_t, sizeof(double)); }
    int SetParam(const std::string &key, bool v) { return setTyped(key, &v, BOOL_t, sizeof(bool)); }
    int SetParam(const std::string &key, const char *s) { return setTyped(key, s, CHAR_t, (unsigned)std::strlen(s) + 1); }

    int GetParam(const std::string &key, Param *p)
    {
        auto it = Params.find(key);
        if (it == Params.end()) {
            std::cerr << "[" << ModuleName << "] GetParam: Parameter with key \"" << key << "\" does not exist." << std::endl;
            return -1;
        }
        if (p->Capacity < it->second.Size) {
            std::cerr << "[" << ModuleName << "] GetParam: Parameter with key \"" << key << "\" has size larger than capacity." << std::endl;
            return -1;
        }
        p->Datatype = it->second.Datatype;
        p->Size = it->second.Size;
        std::memcpy(p->Ptr, it->second.Ptr, p->Size);
        return 0;
    }
    int GetParam(const std::string &key, int *v) { return getTyped(key, v, sizeof(int)); }
    int GetParam(const std::string &key, float *v) { return getTyped(key, v, sizeof(float)); }
    int GetParam(const std::string &key, double *v) { return getTyped(key, v, sizeof(double)); }
    int GetParam(const std::string &key, bool *v) { return getTyped(key, v, sizeof(bool)); }

  protected:
    std::string ModuleName;
    std::map<std::string, Param> Params;
    unsigned char NumInputs = 0, NumOutputs = 0;
    ExpectedPort *expectedInputs = nullptr;
    Port **inputs = nullptr;
    Port *outputs = nullptr;

    int InsertParam(const std::string &key, void *ptr, DataType_t dtype, unsigned capacity, unsigned size)
    {
        Param p{ptr, dtype, capacity, size};
        if (!Params.insert({key, p}).second) {
            std::cerr << "[" << ModuleName << "] InsertParam: Parameter with key \"" << key << "\" already exists." << std::endl;
            return -1;
        }
        return 0;
    }
    int AllocateInputs(unsigned char n)
    {
        if (NumInputs || expectedInputs || inputs) {
            std::cerr << "[" << ModuleName << "] AllocateInputs: Inputs already allocated" << std::endl;
            return -1;
        }
        NumInputs = n;
        if (n) {
            expectedInputs = new ExpectedPort[n]();
            inputs = new Port *[n]();
        }
        return 0;
    }
    int AllocateOutputs(unsigned char n)
    {
        if (NumOutputs || outputs) {
            std::cerr << "[" << ModuleName << "] AllocateOutputs: Outputs already allocated" << std::endl;
            return -1;
        }
        NumOutputs = n;
        if (n) outputs = new Port[n]();
        return 0;
    }
    int ConfigExpectedInput(unsigned char id, const char *name, DataType_t dtype, ValueType_t vt, uint32_t len)
    {
        if (id >= NumInputs) {
            std::cerr << "[" << ModuleName << "] ConfigExpectedInput: id " << name << " out of range." << std::endl;
            return -1;
        }
        std::strncpy(expectedInputs[id].Name, name, 31);
        expectedInputs[id].Datatype = dtype;
        expectedInputs[id].ValueType = vt;
        expectedInputs[id].VectorLength = len;
        return 0;
    }
    int ConfigOutput(unsigned char id, const char *name, DataType_t dtype, ValueType_t vt, MemLoc_t loc, uint32_t len,
                     void *data, int aux)
    {
        if (id >= NumOutputs) {
            std::cerr << "[" << ModuleName << "] ConfigOutput: id out of range." << std::endl;
            return -1;
        }
        Port &o = outputs[id];
        std::strncpy(o.Name, name, 31);
        o.Datatype = dtype; o.Exponent = 0; o.ValueType = vt; o.MemLoc = loc; o.VectorLength = len; o.Data = data; o.AuxValue = aux;
        return 0;
    }
    int UpdateOutput(unsigned char id, uint32_t len, void *data, int aux)
    {
        if (id >= NumOutputs) {
            std::cerr << "[" << ModuleName << "] UpdateOutput: id out of range." << std::endl;
            return -1;
        }
        outputs[id].VectorLength = len; outputs[id].Data = data; outputs[id].AuxValue = aux;
        return 0;
    }

  private:
    int setTyped(const std::string &key, const void *v, DataType_t t, unsigned size)
    {
        Param p{const_cast<void *>(v), t, 0, size};
        return SetParam(key, &p);
    }
    int getTyped(const std::string &key, void *v, unsigned cap)
    {
        Param p{v, DATATYPE_ANY, cap, 0};
        return GetParam(key, &p);
    }
};

// Minimal dsp::Flow (cudarecv/dsp/src/flow.cu:28-87,105-197,212-324): owns the modules and one stream,
// starts them in order, runs Update() in order each iteration, any non-zero return stops the flow.
class Flow {
  public:
    ~Flow()
    {
        for (Module *m : Mods) delete m;
    }
    void Add(Module *m) { Mods.push_back(m); }
    Module *Find(const char *name)
    {
        for (Module *m : Mods)
            if (m->GetModuleName() == name) return m;
        std::cerr << "[Flow] Module " << name << " not found." << std::endl;
        return nullptr;
    }
    template <typename T>
    int SetModParam(const char *mod, const char *key, T val)
    {
        Module *m = Find(mod);
        return m ? m->SetParam(key, val) : -1;
    }
    int ConnectPort(const char *srcMod, const char *srcPort, const char *dstMod, const char *dstPort)
    {
        Module *s = Find(srcMod), *d = Find(dstMod);
        if (!s || !d) return -1;
        const int so = s->GetOutputID(srcPort), di = d->GetInputID(dstPort);
        if (so < 0 || di < 0) return -1;
        Port *p = nullptr;
        if (s->GetOutput((unsigned char)so, &p)) return -1;
        return d->SetInput((unsigned char)di, p);
    }
    int Start(void *stream)
    {
        Stream = stream;
        for (size_t i = 0; i < Mods.size(); ++i)
            if (Mods[i]->Start(&Stream)) {   // modules get a pointer to the flow's stream handle (flow.cu:35-44)
                for (size_t j = 0; j < i; ++j) Mods[j]->Stop();
                return -1;
            }
        return 0;
    }
    // one iteration of FlowThread's loop (flow.cu:122-137); returns the first non-zero Update()
    int Step()
    {
        if (Timing) return TimedStep();
        for (Module *m : Mods) {
            const int r = m->Update(&Stream);
            if (r) return r;
        }
        return 0;
    }
    // per-module wall time of Update() (host side; a module that waits on the stream is charged the wait)
    void EnableTiming(bool on)
    {
        Timing = on;
        Spent.assign(Mods.size(), 0.0);
        Steps = 0;
    }
    void ReportTiming(std::ostream &os) const
    {
        if (!Timing || !Steps) return;
        for (size_t i = 0; i < Mods.size(); ++i)
            os << "[Flow] " << Mods[i]->GetModuleName() << ": " << Spent[i] / Steps * 1e6 << " us per Update" << std::endl;
    }
    void Stop()
    {
        for (Module *m : Mods) m->Stop();
    }

  private:
    int TimedStep()
    {
        for (size_t i = 0; i < Mods.size(); ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            const int r = Mods[i]->Update(&Stream);
            Spent[i] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (r) return r;
        }
        ++Steps;
        return 0;
    }
    std::vector<Module *> Mods;
    std::vector<double> Spent;
    long long Steps = 0;
    bool Timing = false;
    void *Stream = nullptr;
};

}  // namespace dsp
