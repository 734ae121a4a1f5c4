// micro-checks for an MFMA-assisted scan: (1) operand / result layout of v_mfma_f32_4x4x1_16b_f32, (2) does MFMA work hide under VALU work
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void layout_kernel(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}
template <int NM>
__global__ __launch_bounds__(256) void mix_kernel(float *out, int iters, float s)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    float x0 = t * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    const float av = s + (threadIdx.x & 3), bv = x0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // 32 VALU per iteration (8 independent chains x 4)
            x0 = fmaf(x0, s, 1.f); x1 = fmaf(x1, s, 1.f); x2 = fmaf(x2, s, 1.f); x3 = fmaf(x3, s, 1.f);
            x4 = fmaf(x4, s, 1.f); x5 = fmaf(x5, s, 1.f); x6 = fmaf(x6, s, 1.f); x7 = fmaf(x7, s, 1.f);
            if (NM > 0 && r < NM) {
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c3, 0, 0, 0);
            }
        }
    }
    out[t] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0[0] + c1[1] + c2[2] + c3[3];
}
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NM>
__global__ __launch_bounds__(256) void mix32_kernel(float *out, int iters, float s)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    float x0 = t * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    f16v c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    const float av = s + (threadIdx.x & 3), bv = x0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // 32 VALU per iteration; NM 32x32x2 MFMAs per iteration
            x0 = fmaf(x0, s, 1.f); x1 = fmaf(x1, s, 1.f); x2 = fmaf(x2, s, 1.f); x3 = fmaf(x3, s, 1.f);
            x4 = fmaf(x4, s, 1.f); x5 = fmaf(x5, s, 1.f); x6 = fmaf(x6, s, 1.f); x7 = fmaf(x7, s, 1.f);
            if (r < NM) { if (r & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0); }
        }
    }
    out[t] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0[0] + c1[1];
}
template <int NM>
static float time_mix32(float *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mix32_kernel<NM>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(mix32_kernel<NM>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
template <int NM>
static float time_mix(float *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mix_kernel<NM>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(mix_kernel<NM>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main()
{
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    std::vector<float> ha(64), hb(64), hd(256);
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + 3.f * l; }
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const float want = ha[4 * (l / 4) + i] * hb[l];   // assumed: lane (b, j) register i = A_b[i] * B_b[j]
            if (std::fabs(hd[l * 4 + i] - want) > 1e-3f * std::fabs(want)) ++bad;
        }
    printf("layout: %d mismatches of 256 (lane 5: %g %g %g %g; A[4..7] = %g %g %g %g, B[5] = %g)\n", bad, hd[20], hd[21], hd[22], hd[23], ha[4], ha[5], ha[6], ha[7], hb[5]);
    float *o; hipMalloc(&o, 4 * 256 * 8192);
    const int blocks = 256 * 8, iters = 2000;
    printf("32 VALU per iteration, 8 waves per SIMD, %d blocks x %d iterations\n", blocks, iters);
    printf("  + 0 MFMA : %.3f ms\n", time_mix<0>(o, blocks, iters));
    printf("  + 4 MFMA (4x4x1) per 32 VALU : %.3f ms\n", time_mix<1>(o, blocks, iters));
    printf("  + 8 MFMA per 32 VALU : %.3f ms\n", time_mix<2>(o, blocks, iters));
    printf("  + 16 MFMA per 32 VALU : %.3f ms\n", time_mix<4>(o, blocks, iters));
    printf("  32x32x2: + 0 : %.3f ms, + 1 per 32 VALU : %.3f ms, + 2 : %.3f ms, + 4 : %.3f ms\n", time_mix32<0>(o, blocks, iters), time_mix32<1>(o, blocks, iters), time_mix32<2>(o, blocks, iters), time_mix32<4>(o, blocks, iters));
    return 0;
}
