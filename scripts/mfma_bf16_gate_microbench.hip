// Gate micro-benchmark (round 5): does bf16 matrix-pipe work hide under vector work on gfx950?
//   v_mfma_f32_32x32x16_bf16 (16 384 MAC per instruction) issued
//     (a) by the SAME waves that issue the vector instructions (interleaved),
//     (b) by MFMA-ONLY waves beside VALU-ONLY waves on the same SIMDs (blocks of 8 waves: waves 0..3 vector, waves 4..7 matrix --
//         the arrangement the microarchitecture guide says runs concurrently),
//   against the two kinds of work alone.  Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_bf16_gate_microbench.hip -o /tmp/mfma_gate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void valu32(float (&x)[8], float s)
{
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], s, 1.f);
}

// mode 0: vector only; 1: matrix only; 2: both from every wave (NM MFMAs per 32 vector FMAs); 3: waves 0..3 vector, waves 4..7 matrix
template <int MODE, int NM>
__global__ __launch_bounds__(512) void gate_kernel(float *out, int iters, float s)
{
    const int t = blockIdx.x * 512 + threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = t * 1e-3f + i;
    f16v c0, c1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + 0.001f * (threadIdx.x & 7)); b[i] = (__bf16)(0.5f + 0.01f * i); }
    const bool doV = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool doM = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (doV) valu32(x, s);
        if (doM) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (m & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += x[i];
    out[t] = r + c0[0] + c1[1];
}

template <int MODE, int NM>
static float run(float *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((gate_kernel<MODE, NM>), dim3(blocks), dim3(512), 0, 0, d, iters, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((gate_kernel<MODE, NM>), dim3(blocks), dim3(512), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main()
{
    const int blocks = 256 * 4, iters = 2000;   // 4 blocks of 8 waves per CU: 8 waves per SIMD
    float *o;
    hipMalloc(&o, sizeof(float) * 512 * blocks);
    printf("blocks of 8 waves, %d blocks x %d iterations; 32 v_fma_f32 (16 v_pk_fma_f32) and / or NM v_mfma_f32_32x32x16_bf16 per iteration\n", blocks, iters);
    const float v = run<0, 1>(o, blocks, iters);
    printf("vector only (all 8 waves)                        : %.3f ms\n", v);
    printf("matrix only (all 8 waves), NM = 1 / 2 / 4        : %.3f / %.3f / %.3f ms\n", run<1, 1>(o, blocks, iters), run<1, 2>(o, blocks, iters), run<1, 4>(o, blocks, iters));
    printf("both from every wave,      NM = 1 / 2 / 4        : %.3f / %.3f / %.3f ms\n", run<2, 1>(o, blocks, iters), run<2, 2>(o, blocks, iters), run<2, 4>(o, blocks, iters));
    printf("waves 0-3 vector, 4-7 matrix, NM = 1 / 2 / 4     : %.3f / %.3f / %.3f ms\n", run<3, 1>(o, blocks, iters), run<3, 2>(o, blocks, iters), run<3, 4>(o, blocks, iters));
    // rate check: blocks x 8 waves x iters x NM instructions of 32 x 32 x 16 MAC
    const double flop = (double)blocks * 8 * iters * 4 * 32768.0;
    printf("matrix rate at NM = 4, matrix only: %.0f TFLOP/s (dense bf16 peak ~2500)\n", flop / (run<1, 4>(o, blocks, iters) * 1e-3) / 1e12);
    return 0;
}
