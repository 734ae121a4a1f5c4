// dpe_util.hip -- error reporting, C/A code generator, SampleBlock H2D, HIP-event timing.
#include "dpe_common.h"

namespace dpe {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    fprintf(stderr, "[dpe_hip] %s\n", g_err);  // reference: message to std::cerr tagged [Module]
}

// C/A chips of one PRN.  Same sequence as BCS_GenCACode (batchcorrscores.cu:117-177): G1 taps
// 3,10; G2 taps 2,3,6,8,9,10; G2 output from the two phase-selector stages of the PRN;
// chip = +1 where G1 xor G2 == 1.  Written on bits rather than the reference's +/-1 products.
void gen_ca_code_host(int prn, int8_t *chips)
{
    static const uint8_t selA[37] = {2, 3, 4, 5, 1, 2, 1, 2, 3, 2, 3, 5, 6, 7, 8, 9, 1, 2, 3,
                                     4, 5, 6, 1, 4, 5, 6, 7, 8, 1, 2, 3, 4, 5, 4, 1, 2, 4};
    static const uint8_t selB[37] = {6, 7, 8, 9, 9, 10, 8, 9, 10, 3, 4, 6, 7, 8, 9, 10, 4, 5, 6,
                                     7, 8, 9, 3, 6, 7, 8, 9, 10, 6, 7, 8, 9, 10, 10, 7, 8, 10};
    // stage s (1..10) lives in bit s-1; all ones initially
    uint32_t g1 = 0x3FF, g2 = 0x3FF;
    const int a = selA[prn - 1] - 1, b = selB[prn - 1] - 1;
    for (int i = 0; i < kLCA; ++i) {
        const uint32_t o1 = (g1 >> 9) & 1u;
        const uint32_t o2 = ((g2 >> a) ^ (g2 >> b)) & 1u;
        chips[i] = (o1 ^ o2) ? 1 : -1;
        const uint32_t f1 = ((g1 >> 2) ^ (g1 >> 9)) & 1u;
        const uint32_t f2 = ((g2 >> 1) ^ (g2 >> 2) ^ (g2 >> 5) ^ (g2 >> 7) ^ (g2 >> 8) ^ (g2 >> 9)) & 1u;
        g1 = ((g1 << 1) | f1) & 0x3FF;
        g2 = ((g2 << 1) | f2) & 0x3FF;
    }
}

// Stream copy / triad over float4: the measured HBM ceiling bench.py quotes beside the 8 TB/s nominal peak.
__global__ __launch_bounds__(256) void hbm_copy_kernel(float4 *__restrict__ dst, const float4 *__restrict__ src, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void hbm_triad_kernel(float4 *__restrict__ a, const float4 *__restrict__ b,
                                                        const float4 *__restrict__ c, float s, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const float4 x = b[i], y = c[i];
        a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}

}  // namespace dpe

extern "C" {

int dpe_abi_version(void) { return DPE_ABI_VERSION; }

const char *dpe_last_error(void) { return dpe::g_err; }

int dpe_set_device(int32_t device)
{
    DPE_CHECK_HIP(hipSetDevice(device));
    return 0;
}

int dpe_device_info(char *name, int nameLen, int *cuCount, int64_t *hbmBytes)
{
    int dev = 0;
    DPE_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    DPE_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    if (name && nameLen > 0) snprintf(name, nameLen, "%s (%s)", p.name, p.gcnArchName);
    if (cuCount) *cuCount = p.multiProcessorCount;
    if (hbmBytes) *hbmBytes = (int64_t)p.totalGlobalMem;
    return 0;
}

int dpe_hbm_ceiling(int64_t bytesPerArray, int iters, dpe_stream_t stream, double *copyGBs, double *triadGBs)
{
    if (bytesPerArray < 4096 || iters < 1 || !copyGBs || !triadGBs) {
        dpe::set_error("dpe_hbm_ceiling: bad arguments");
        return -1;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t n4 = bytesPerArray / 16;
    float4 *buf[3] = {nullptr, nullptr, nullptr};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = -1;
    float ms = 0.f;
    const unsigned blocks = (unsigned)((n4 + 255) / 256);
    for (int i = 0; i < 3; ++i)
        if (hipMalloc((void **)&buf[i], n4 * 16) != hipSuccess) { dpe::set_error("dpe_hbm_ceiling: hipMalloc failed"); goto done; }
    for (int i = 0; i < 3; ++i)
        if (hipMemsetAsync(buf[i], 0, n4 * 16, st) != hipSuccess) goto done;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) goto done;
    for (int pass = 0; pass < 2; ++pass) {
        // two untimed launches first (clock ramp), then `iters` timed ones
        for (int it = -2; it < iters; ++it) {
            if (it == 0 && hipEventRecord(e0, st) != hipSuccess) goto done;
            if (pass == 0) dpe::hbm_copy_kernel<<<blocks, 256, 0, st>>>(buf[0], buf[1], n4);
            else dpe::hbm_triad_kernel<<<blocks, 256, 0, st>>>(buf[0], buf[1], buf[2], 0.5f, n4);
        }
        if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
            hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
            dpe::set_error("dpe_hbm_ceiling: timing failed");
            goto done;
        }
        const double gbs = (double)(pass == 0 ? 2 : 3) * (double)(n4 * 16) * iters / (ms * 1e-3) / 1e9;
        if (pass == 0) *copyGBs = gbs; else *triadGBs = gbs;
    }
    rc = 0;
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (int i = 0; i < 3; ++i) if (buf[i]) (void)hipFree(buf[i]);
    return rc;
}

int dpe_gen_ca_code(int8_t *chips)
{
    DPE_REQUIRE(chips != nullptr, "dpe_gen_ca_code: null output");
    for (int prn = 1; prn <= dpe::kPrnMax; ++prn) dpe::gen_ca_code_host(prn, chips + (prn - 1) * dpe::kLCA);
    return 0;
}

int dpe_sampleblock_upload(int16_t *dst_dev, const int16_t *src_host, int64_t nSamples, dpe_stream_t stream)
{
    DPE_REQUIRE(dst_dev && src_host && nSamples > 0, "dpe_sampleblock_upload: bad arguments");
    DPE_CHECK_HIP(hipMemcpyAsync(dst_dev, src_host, (size_t)nSamples * 2 * sizeof(int16_t), hipMemcpyHostToDevice,
                                 (hipStream_t)stream));
    return 0;
}

int dpe_host_alloc_pinned(void **ptr, int64_t bytes)
{
    DPE_REQUIRE(ptr && bytes > 0, "dpe_host_alloc_pinned: bad arguments");
    DPE_CHECK_HIP(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
    return 0;
}

int dpe_host_free_pinned(void *ptr)
{
    if (ptr) DPE_CHECK_HIP(hipHostFree(ptr));
    return 0;
}

int dpe_device_alloc(void **ptr_dev, int64_t bytes)
{
    DPE_REQUIRE(ptr_dev && bytes > 0, "dpe_device_alloc: bad arguments");
    DPE_CHECK_HIP(hipMalloc(ptr_dev, (size_t)bytes));
    return 0;
}

int dpe_device_free(void *ptr_dev)
{
    if (ptr_dev) DPE_CHECK_HIP(hipFree(ptr_dev));
    return 0;
}

int dpe_memcpy_h2d(void *dst_dev, const void *src_host, int64_t bytes, dpe_stream_t stream)
{
    DPE_CHECK_HIP(hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}

int dpe_memcpy_d2h(void *dst_host, const void *src_dev, int64_t bytes, dpe_stream_t stream)
{
    DPE_CHECK_HIP(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    DPE_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int dpe_stream_create(dpe_stream_t *stream)
{
    hipStream_t s;
    DPE_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (dpe_stream_t)s;
    return 0;
}

int dpe_stream_destroy(dpe_stream_t stream)
{
    DPE_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

int dpe_stream_synchronize(dpe_stream_t stream)
{
    DPE_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int dpe_event_create(void **ev)
{
    hipEvent_t e;
    DPE_CHECK_HIP(hipEventCreate(&e));
    *ev = (void *)e;
    return 0;
}

int dpe_event_record(void *ev, dpe_stream_t stream)
{
    DPE_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}

int dpe_event_elapsed_ms(void *start, void *stop, float *ms)
{
    DPE_CHECK_HIP(hipEventSynchronize((hipEvent_t)stop));
    DPE_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

int dpe_event_destroy(void *ev)
{
    DPE_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
    return 0;
}

}  // extern "C"
