// dpe_bcs_fft.h -- full-length FFT form of stage 1, the fallback for windows the streaming kernels do not take: code-lag
// banks wider than DPE_MAX_LAG_HALF_WIDTH, Doppler banks beyond the moment expansion.  Included by dpe_bcs.hip.
//
// This IS the reference's formulation (cudarecv/modules/src/batchcorrscores.cu): wiped samples and both masked replicas
// (:277-305, :323-372), forward transforms of length S, conj(F r) F b, inverse transform and 1/S (:1099-1144, :594), the
// replica choice at lag 0 (:499-543); then (raw - mean) wipe x chosen replica zero-padded to C, one forward transform of
// length C (:422-452, :1161-1180).  rocFFT (called directly, dpe_fft.h) in fp32 (batched over the SVs and a chunk of windows); only the requested
// window of lags / bins is copied into the banks, in the same layout the streaming kernels produce.
#pragma once

#include "dpe_fft.h"

namespace dpe {

// b[n] = raw wipe, rX[n] / rY[n] = replica masked to the two sides of the nav-bit boundary.  Layout: three planes of
// [nW][K][S] complex each (b, rX, rY), so that one batched transform covers them all.
template <bool TABLE>
__global__ __launch_bounds__(256) void bcs_fft_prep_code_kernel(const int16_t *__restrict__ iq, long long winStride, int S, int K, int nW, int w0,
                                                                const BcsChanDev *__restrict__ chan, const int8_t *__restrict__ chipTable,
                                                                const double *__restrict__ tT, float2 *__restrict__ work)
{
    const int k = blockIdx.y, wl = blockIdx.z, w = w0 + wl;     // nW = windows per chunk (plane stride), w0 = first window of this chunk
    const BcsChanDev ch = chan[w * K + k];
    const int *x = reinterpret_cast<const int *>(iq + (size_t)w * winStride * 2);
    const int8_t *chips = chipTable + (ch.prn - 1) * 1024;
    const size_t plane = (size_t)nW * K * S, row = ((size_t)wl * K + k) * S;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < S; n += gridDim.x * 256) {
        const int v = x[n];
        double ph = carr_phase<TABLE>(ch, tT, n);
        ph -= floor(ph);
        const f2 wv = wipe_seed((float)ph);
        const f2 b = cmul(f2{(float)(short)(v & 0xFFFF), (float)(v >> 16)}, wv);
        const int ci = ((int)floor(code_phase<TABLE>(ch, tT, n))) % kLCA;
        const float r = (float)chips[ci];
        const int sd = ch.hasFlip ? (n >= ch.idxNext) : 0;
        work[row + n] = make_float2(b.x, b.y);
        work[plane + row + n] = make_float2(sd == 0 ? r : 0.f, 0.f);
        work[2 * plane + row + n] = make_float2(sd == 1 ? r : 0.f, 0.f);
    }
}

// in place on the replica planes: conj(F r) F b
__global__ __launch_bounds__(256) void bcs_fft_mul_kernel(float2 *__restrict__ work, size_t plane)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (size_t)gridDim.x * 256) {
        const float2 fb = work[i];
#pragma unroll
        for (int s = 1; s <= 2; ++s) {
            const float2 fr = work[s * plane + i];
            work[s * plane + i] = make_float2(fr.x * fb.x + fr.y * fb.y, fr.x * fb.y - fr.y * fb.x);
        }
    }
}

// lags [-L, L] of both sides out of the inverse transforms (circular: lag l at index l mod S), 1/S, the replica choice
// at lag 0 (BCS_ChooseCodeCorr :512-516), bank [w][maxK][2L+1]
__global__ __launch_bounds__(256) void bcs_fft_extract_code_kernel(const float2 *__restrict__ work, size_t plane, int S, int K, int L, int w0,
                                                                   int maxK, const BcsChanDev *__restrict__ chan,
                                                                   float2 *__restrict__ codeBank, int *__restrict__ info)
{
    const int k = blockIdx.x, wl = blockIdx.y, w = w0 + wl;
    const float2 *X = work + plane + ((size_t)wl * K + k) * S, *Y = work + 2 * plane + ((size_t)wl * K + k) * S;
    const float inv = 1.0f / (float)S;
    const BcsChanDev ch = chan[w * K + k];
    const float2 x0 = X[0], y0 = Y[0];
    const float nr = x0.x + y0.x, ni = x0.y + y0.y, fr = x0.x - y0.x, fi = x0.y - y0.y;
    const int noFlip = (!ch.hasFlip) || (nr * nr + ni * ni > fr * fr + fi * fi);
    const float sgn = noFlip ? 1.f : -1.f;
    if (threadIdx.x == 0) info[w * K + k] = noFlip;
    for (int j = threadIdx.x; j < 2 * L + 1; j += 256) {
        int idx = j - L;
        if (idx < 0) idx += S;
        const float2 a = X[idx], b = Y[idx];
        codeBank[((size_t)w * maxK + k) * (2 * L + 1) + j] = make_float2((a.x + sgn * b.x) * inv, (a.y + sgn * b.y) * inv);
    }
}

// carrier path: (raw - mean) wipe x chosen replica, zero-padded to C  (work: [nW][K][C])
template <bool TABLE>
__global__ __launch_bounds__(256) void bcs_fft_prep_carr_kernel(const int16_t *__restrict__ iq, long long winStride, int S, long long C, int K,
                                                                int w0, int nSumBlk, const BcsChanDev *__restrict__ chan,
                                                                const long long *__restrict__ sums, const int8_t *__restrict__ chipTable,
                                                                const double *__restrict__ tT, const int *__restrict__ info,
                                                                float2 *__restrict__ work)
{
    const int k = blockIdx.y, wl = blockIdx.z, w = w0 + wl;
    const BcsChanDev ch = chan[w * K + k];
    const int *x = reinterpret_cast<const int *>(iq + (size_t)w * winStride * 2);
    const int8_t *chips = chipTable + (ch.prn - 1) * 1024;
    float mRe, mIm;
    window_mean(sums, w, nSumBlk, S, mRe, mIm);
    const float sgnY = info[w * K + k] ? 1.f : -1.f;
    float2 *o = work + ((size_t)wl * K + k) * C;
    for (long long n = (long long)blockIdx.x * 256 + threadIdx.x; n < C; n += (long long)gridDim.x * 256) {
        float2 c = make_float2(0.f, 0.f);
        if (n < S) {
            const int v = x[n];
            double ph = carr_phase<TABLE>(ch, tT, (int)n);
            ph -= floor(ph);
            const f2 wv = wipe_seed((float)ph);
            const int ci = ((int)floor(code_phase<TABLE>(ch, tT, (int)n))) % kLCA;
            const float r = (float)chips[ci] * ((ch.hasFlip && n >= ch.idxNext) ? sgnY : 1.f);
            const f2 b = cmul(f2{(float)(short)(v & 0xFFFF) - mRe, (float)(v >> 16) - mIm}, wv);   // (raw - mean) wipe (:480, :440-448)
            c = make_float2(b.x * r, b.y * r);
        }
        o[n] = c;
    }
}

// bins [-B, B] out of the forward transform (bin b at index b mod C), bank [w][maxK][2B+1]
__global__ __launch_bounds__(256) void bcs_fft_extract_carr_kernel(const float2 *__restrict__ work, long long C, int K, int B, int w0, int maxK,
                                                                   float2 *__restrict__ carrBank)
{
    const int k = blockIdx.x, wl = blockIdx.y, w = w0 + wl;
    const float2 *F = work + ((size_t)wl * K + k) * C;
    for (int j = threadIdx.x; j < 2 * B + 1; j += 256) {
        long long idx = j - B;
        if (idx < 0) idx += C;
        carrBank[((size_t)w * maxK + k) * (2 * B + 1) + j] = F[idx];
    }
}

}  // namespace dpe
