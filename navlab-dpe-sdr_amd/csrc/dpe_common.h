// dpe_common.h -- internal helpers shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/dpe_hip.h"

namespace dpe {

// cudarecv/utils/inc/consthelper.h:5-27 -- bit-for-bit
constexpr double kC = 299792458.0;
constexpr double kPi = 3.1415926535898;
constexpr double kFL1 = 1.57542e9;
constexpr double kFCA = 1.023e6;
constexpr double kTCA = 0.001;
constexpr double kOEDot = 7.2921151467e-5;
constexpr int kLCA = 1023;
constexpr int kPrnMax = 37;

void set_error(const char *fmt, ...);

#define DPE_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            dpe::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                 \
                           hipGetErrorString(e_));                                       \
            return -1;                                                                   \
        }                                                                                \
    } while (0)

#define DPE_REQUIRE(cond, ...)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            dpe::set_error(__VA_ARGS__);                                                 \
            return -1;                                                                   \
        }                                                                                \
    } while (0)

void gen_ca_code_host(int prn, int8_t *chips);  // dpe_util.hip

// 64-lane butterfly sum; every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg).
struct KernelProfiler {
    bool enabled = false;
    static constexpr int kSlots = 8;
    std::vector<hipEvent_t> ev[kSlots];   // pairs: start, stop
    void begin(int slot, hipStream_t st)
    {
        if (!enabled) return;
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        ev[slot].push_back(a);
        ev[slot].push_back(b);
        (void)hipEventRecord(a, st);
    }
    void end(int slot, hipStream_t st)
    {
        if (!enabled || ev[slot].empty()) return;
        (void)hipEventRecord(ev[slot].back(), st);
    }
    // total milliseconds and launch count per slot since the last call; frees the events
    void collect(float *ms, int *count)
    {
        for (int s = 0; s < kSlots; ++s) {
            ms[s] = 0.f;
            count[s] = (int)ev[s].size() / 2;
            for (size_t i = 0; i + 1 < ev[s].size(); i += 2) {
                float t = 0.f;
                (void)hipEventSynchronize(ev[s][i + 1]);
                if (hipEventElapsedTime(&t, ev[s][i], ev[s][i + 1]) == hipSuccess) ms[s] += t;
                (void)hipEventDestroy(ev[s][i]);
                (void)hipEventDestroy(ev[s][i + 1]);
            }
            ev[s].clear();
        }
    }
};

template <typename T>
static inline T *dev_alloc(size_t n)
{
    void *p = nullptr;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
    return static_cast<T *>(p);
}

}  // namespace dpe
