// dpe_fft.h -- batched 1-D complex transforms on rocFFT, called directly (rocfft_plan_create / rocfft_execute; no cuFFT-shaped
// front end): in place, single precision, interleaved complex, rows of `len` elements `len` apart, unnormalised in both
// directions (as cuFFT's, which the reference divides out itself: batchcorrscores.cu:594-606).  Host-side helper of the
// full-length FFT form of stage 1 (dpe_bcs_fft.h) and of the acquisition search (dpe_acq.hip).
//
// Load / store callbacks (rocfft_execution_info_set_load_callback / _store_callback) were measured for the acquisition's
// inverse transforms -- the spectrum product fused into the load, |.| into the store -- and dropped: 4000 transforms of 2500
// points took 0.087 ms with the callbacks against 0.080 ms as three kernels (product, transform, magnitude) whose
// intermediate 80 MB surface stays in the Infinity Cache; the indirect call per element costs more than the passes it saves.
#pragma once
#include <rocfft/rocfft.h>

#include "dpe_common.h"

namespace dpe {

struct FftPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void *work = nullptr;
    size_t workBytes = 0;

    bool ok() const { return plan != nullptr; }
    // 0 on success; on failure everything is released again and dpe::set_error has the message
    int create(size_t len, size_t batch, bool inverse)
    {
        static const bool once = (rocfft_setup() == rocfft_status_success);
        (void)once;
        destroy();
        size_t lengths[1] = {len};
        rocfft_status st = rocfft_plan_create(&plan, rocfft_placement_inplace,
                                              inverse ? rocfft_transform_type_complex_inverse : rocfft_transform_type_complex_forward,
                                              rocfft_precision_single, 1, lengths, batch, nullptr);
        if (st != rocfft_status_success) {
            plan = nullptr;
            set_error("rocfft_plan_create(n = %zu, batch = %zu, %s) -> status %d", len, batch, inverse ? "inverse" : "forward", (int)st);
            return -1;
        }
        if (rocfft_execution_info_create(&info) != rocfft_status_success) { info = nullptr; destroy(); set_error("rocfft_execution_info_create failed"); return -1; }
        if (rocfft_plan_get_work_buffer_size(plan, &workBytes) != rocfft_status_success) workBytes = 0;
        if (workBytes) {
            if (hipMalloc(&work, workBytes) != hipSuccess || rocfft_execution_info_set_work_buffer(info, work, workBytes) != rocfft_status_success) {
                const size_t wb = workBytes;
                destroy();
                set_error("rocFFT work buffer (%zu bytes) for n = %zu, batch = %zu", wb, len, batch);
                return -1;
            }
        }
        return 0;
    }
    int exec(hipStream_t st, void *buf)
    {
        if (!plan) { set_error("rocFFT plan missing"); return -1; }
        if (rocfft_execution_info_set_stream(info, (void *)st) != rocfft_status_success) { set_error("rocfft_execution_info_set_stream failed"); return -1; }
        void *in[1] = {buf};
        const rocfft_status r = rocfft_execute(plan, in, nullptr, info);
        if (r != rocfft_status_success) { set_error("rocfft_execute -> status %d", (int)r); return -1; }
        return 0;
    }
    void destroy()
    {
        if (info) (void)rocfft_execution_info_destroy(info);
        if (plan) (void)rocfft_plan_destroy(plan);
        if (work) (void)hipFree(work);
        info = nullptr; plan = nullptr; work = nullptr; workBytes = 0;
    }
};

}  // namespace dpe
