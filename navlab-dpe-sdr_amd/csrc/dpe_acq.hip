// dpe_acq.hip -- cold-start coarse acquisition (BASELINE.json configs[4]).
//
// The reference has acquisition only in its Python twin: Correlator.coarse_acquisition
// (pygnss/pythonreceiver/scalar/correlator.py:53-103) -- per Doppler bin: wipe-off, FFT, multiply by the
// conjugate FFT of the nominal-rate replica, IFFT (all S lags), fold the N code periods of the window
// (complex sum = "coherent", sum of magnitudes = "non-coherent"), then peak statistics
// (cppr, cppm = peak / 10 %-trimmed mean, found <=> cppm > 2).
//
// Here: the bin loop becomes one batched FFT (rocFFT, called directly: dpe_fft.h -- an FFT is intrinsic to a
// full code-delay search), the replica spectra (conj, 1/S folded in) are precomputed once per PRN at
// create, and HIP kernels do wipe-off, spectrum multiply and fold + per-lag max over bins.  The coherent search at
// 2 500 delays per code period (the reference's 2.5 Msps) replaces multiply + inverse rocFFT + fold by ONE fused kernel
// with its own 2 500-point transform (acq_corr2500_kernel, below).
// mode 0 / 1 = the reference's coherent / non-coherent semantics (pinned by fixture O8);
// mode 2 = the textbook "1 ms coherent x N non-coherent" of BASELINE.json (NOT in the reference: parity
// unpinned, checked against the oracle's own restatement only).
#include "dpe_fft.h"

#include <algorithm>

#include "dpe_common.h"

namespace dpe {

// X[b][n] = raw[n] * exp(-j 2 pi f_b n / fs)   (correlator.py:63)
// (both wipe-off kernels also clear max_percode for the fold kernel's atomic maxima: first kernel of a search)
__device__ __forceinline__ void acq_clear(float *__restrict__ mp, long long n)
{
    const long long gt = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    const long long gn = (long long)gridDim.x * gridDim.y * blockDim.x;
    for (long long i = gt; i < n; i += gn) mp[i] = 0.f;
}

__global__ __launch_bounds__(256) void acq_wipe_kernel(const int16_t *__restrict__ iq, int S, int B, double binStart,
                                                       double binStep, double invFs, float2 *__restrict__ X,
                                                       float *__restrict__ mp, long long mpLen)
{
    acq_clear(mp, mpLen);
    const int b = blockIdx.y;
    const double cyclesPerSample = (binStart + binStep * b) * invFs;
    const int *x = reinterpret_cast<const int *>(iq);
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < S; n += gridDim.x * blockDim.x) {
        const int v = x[n];
        const float re = (float)(short)(v & 0xFFFF), im = (float)(v >> 16);
        double ph = cyclesPerSample * (double)n;
        ph -= floor(ph);
        const float f = (float)ph;
        const float c = __builtin_amdgcn_cosf(f), s = -__builtin_amdgcn_sinf(f);
        X[(size_t)b * S + n] = make_float2(re * c - im * s, re * s + im * c);
    }
}

// Coherent mode: X[b][j] = sum_n raw[j + n M] exp(-j 2 pi f_b (j + n M) / fs), j < M.  Summing the N lag aliases
// of a length-S circular correlation (correlator.py:77-80) equals a length-M circular correlation of the
// time-folded inputs (sampling the product spectrum at every N-th bin), so the coherent search runs
// length-M transforms on folded data: N times less FFT work and memory than the literal formulation.
__global__ __launch_bounds__(256) void acq_wipe_fold_kernel(const int16_t *__restrict__ iq, int M, int N, double binStart,
                                                            double binStep, double invFs, float2 *__restrict__ X,
                                                            float *__restrict__ mp, long long mpLen)
{
    acq_clear(mp, mpLen);
    const int b = blockIdx.y;
    const double cyclesPerSample = (binStart + binStep * b) * invFs;
    const int *x = reinterpret_cast<const int *>(iq);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) {
        float ar = 0.f, ai = 0.f;
        for (int n = 0; n < N; ++n) {
            const int i = j + n * M;
            const int v = x[i];
            const float re = (float)(short)(v & 0xFFFF), im = (float)(v >> 16);
            double ph = cyclesPerSample * (double)i;
            ph -= floor(ph);
            const float f = (float)ph;
            const float c = __builtin_amdgcn_cosf(f), s = -__builtin_amdgcn_sinf(f);
            ar += re * c - im * s;
            ai += re * s + im * c;
        }
        X[(size_t)b * M + j] = make_float2(ar, ai);
    }
}

// Rc = conj(FFT(replica)) / len   (correlator.py:67; the 1/len is numpy's ifft normalisation)
__global__ void acq_conj_scale_kernel(float2 *__restrict__ R, long long n, float scale)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 v = R[i];
        R[i] = make_float2(v.x * scale, -v.y * scale);
    }
}

// Y[p][b][i] = X[b][i] * Rc[p][i mod len]   (correlator.py:75)
__global__ __launch_bounds__(256) void acq_mul_kernel(const float2 *__restrict__ X, const float2 *__restrict__ Rc, int S,
                                                      int len, int B, float2 *__restrict__ Y)
{
    const int b = blockIdx.y, p = blockIdx.z;
    const float2 *xr = X + (size_t)b * S, *rr = Rc + (size_t)p * len;
    float2 *yr = Y + ((size_t)p * B + b) * S;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S; i += gridDim.x * blockDim.x) {
        const float2 a = xr[i], r = rr[i % len];
        yr[i] = make_float2(a.x * r.x - a.y * r.y, a.x * r.y + a.y * r.x);
    }
}

// surface[p][b][j] = | sum_n Y[j + n M] |  (coherent)  or  sum_n |Y[j + n M]|   (correlator.py:77-84), and
// max_percode[p][j] = max_b surface[p][b][j]   (correlator.py:87) in the same pass: a block walks kAcqBinGroup bins of its
// (PRN, delay range), keeps the running maximum per delay and merges it with one atomicMax on the value's bit pattern
// (values are >= 0, so unsigned order is float order; a maximum does not depend on the order of the merges).  mp holds
// zeros on entry (the wipe-off kernel clears it).  The separate column-maximum pass this replaces re-read the whole surface
// with 320 blocks: 34 us of a 155 us search.
constexpr int kAcqBinGroup = 25;
__global__ __launch_bounds__(256) void acq_fold_kernel(const float2 *__restrict__ Y, int S, int M, int N, int coherent, int B,
                                                       float *__restrict__ surf, unsigned int *__restrict__ mpBits)
{
    const int p = blockIdx.z, b0 = blockIdx.y * kAcqBinGroup;
    const int b1 = b0 + kAcqBinGroup < B ? b0 + kAcqBinGroup : B;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) {
        float mx = 0.f;
        for (int b = b0; b < b1; ++b) {
            const size_t row = (size_t)p * B + b;
            const float2 *y = Y + row * S;
            float ar = 0.f, ai = 0.f, am = 0.f;
            for (int n = 0; n < N; ++n) {
                const float2 v = y[j + n * M];
                ar += v.x; ai += v.y;
                am += sqrtf(v.x * v.x + v.y * v.y);
            }
            const float sv = coherent ? sqrtf(ar * ar + ai * ai) : am;
            surf[row * M + j] = sv;
            mx = fmaxf(mx, sv);
        }
        atomicMax(&mpBits[(size_t)p * M + j], __float_as_uint(mx));
    }
}


// ---- the coherent search's inner product at the reference's rate (2.5 Msps: 2 500 delays per code period), fused:
//   surface[p][b][:] = | IFFT_2500( X[b][:] * Rc[p][:] ) |,   max_percode[p][:] = max_b surface[p][b][:]
// in ONE kernel per PRN chunk -- spectrum multiply, a hand-written 2 500-point inverse transform in LDS, magnitude and the
// running maximum over the block's bins.  The rocFFT form writes the 80 MB product, transforms it in place and reads it back
// for the magnitudes (three kernels, 240 MB of traffic per 32-PRN search); here the product never leaves the registers and
// the only large write is the surface itself.
// Transform: 2500 = 10 x 10 x 5 x 5, decimation in frequency, 250 threads.  The 10-point butterflies are prime-factor
// (2 x 5: no inner twiddles), W = exp(+j 2 pi n / 2500) comes from one table for all four passes, data ping-pongs between two
// 20 KB LDS buffers (one barrier per pass).  Index scheme checked against numpy.fft.ifft before it was written in HIP
// (5e-16 in fp64); on the GPU the search results equal the rocFFT path's (tests/test_gpu_acq.py, fixtures O8 / O9).
typedef float af2 __attribute__((ext_vector_type(2)));
// complex product in two packed instructions (the op_sel / neg forms of v_pk_fma_f32, as in dpe_bcs.hip)
__device__ __forceinline__ af2 acq_cmul(af2 a, af2 b)
{
    af2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
__device__ __forceinline__ af2 acq_jrot(af2 b) { return af2{-b.y, b.x}; }   // j * b
// inverse 5-point DFT (e^{+j 2 pi n k / 5}, unnormalised), in place
__device__ __forceinline__ void acq_idft5(af2 &x0, af2 &x1, af2 &x2, af2 &x3, af2 &x4)
{
    constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f, s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const af2 sa = x1 + x4, sb = x2 + x3, da = x1 - x4, db = x2 - x3;
    const af2 y0 = x0 + sa + sb;
    const af2 a1 = x0 + sa * c1 + sb * c2, a2 = x0 + sa * c2 + sb * c1;
    const af2 b1 = acq_jrot(da * s1 + db * s2), b2 = acq_jrot(da * s2 - db * s1);
    x0 = y0; x1 = a1 + b1; x4 = a1 - b1; x2 = a2 + b2; x3 = a2 - b2;
}
// inverse 10-point DFT as 2 x 5 prime-factor: inputs n = 5 n1 + 2 n2, outputs k = 5 k1 + 6 k2 (mod 10); in place, natural order
__device__ __forceinline__ void acq_idft10(af2 (&v)[10])
{
    af2 e0 = v[0] + v[5], e1 = v[2] + v[7], e2 = v[4] + v[9], e3 = v[6] + v[1], e4 = v[8] + v[3];
    af2 o0 = v[0] - v[5], o1 = v[2] - v[7], o2 = v[4] - v[9], o3 = v[6] - v[1], o4 = v[8] - v[3];
    acq_idft5(e0, e1, e2, e3, e4);
    acq_idft5(o0, o1, o2, o3, o4);
    v[0] = e0; v[6] = e1; v[2] = e2; v[8] = e3; v[4] = e4;
    v[5] = o0; v[1] = o1; v[7] = o2; v[3] = o3; v[9] = o4;
}
}  // namespace dpe
#include "dpe_acq_pack.h"
#include "dpe_acq_mixed.h"
namespace dpe {
constexpr int kAcqFusedLen = 2500;
constexpr int kAcqFusedBins = 6;     // bins per block: 21 x 32 blocks are resident at once (three per CU) on 256 CUs
constexpr int kAcqSubStride = 281;   // LDS stride of the ten 250-point sub-sequences: = 25 (mod 32), so that the 25-lane groups of
                                     // passes 1 / 2 fall on distinct 8-byte bank slots
// nSeg = 1: X[b][2500] holds the time-folded window (coherent mode).  nSeg = N: X[b][N][2500] holds the N code periods
// of the window (textbook mode, "1 ms coherent x N non-coherent"): the magnitudes of the N transforms are summed in
// registers -- a thread's ten outputs have the same indices in every segment -- before they go out.
// ALIAS (the reference's coherent = False at 10 x 2 500 samples, correlator.py:77-82): one 25 000-point correlation per (PRN,
// bin), surface[j] = sum_n |y[j + 2500 n]|.  X = Z[p][b][k0][2500], the output of acq_radix10_kernel -- product and first
// decimation-in-frequency stage already applied -- so y[10 m' + k0] = IFFT2500(Z[..][k0])[m'] and the ten lag aliases of delay
// j = 10 r + k0 are the outputs r + 250 n of transform k0: exactly the ten values thread r holds.  nSeg = 10 transforms per
// (PRN, bin), no multiply, each transform's ten-fold sums written at stride 10.
template <bool ALIAS>
__global__ __launch_bounds__(256, 3) void acq_corr2500_kernel(const float2 *__restrict__ X, const float2 *__restrict__ Rc,
                                                              const float2 *__restrict__ tw, int B, int nSeg, int binsPerBlock, int pOffset,
                                                              float *__restrict__ surf, unsigned int *__restrict__ mpBits)
{
    constexpr int N = kAcqFusedLen, SS = kAcqSubStride;
    __shared__ float2 sA[10 * SS], sB[10 * SS];
    __shared__ float2 sW250[256], sW25[32];   // W250^n = tw[10 n], W25^n = tw[100 n]: the twiddles of passes 2 and 3
    const int t = threadIdx.x, p = blockIdx.y, bx = blockIdx.x;
    const bool act = t < 250;
    const int tt = act ? t : 0;
    if (act) sW250[t] = tw[10 * t];
    if (t < 25) sW25[t] = tw[100 * t];
    // the PRN's spectrum and the pass-1 twiddles W^(t k1) of this thread's ten elements stay in registers across the bins
    af2 rc[ALIAS ? 1 : 10], w1[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        if constexpr (!ALIAS) {
            const float2 r = Rc[(size_t)p * N + tt + 250 * q];
            rc[q] = af2{r.x, r.y};
        }
        const float2 a = tw[tt * q];   // t k1 <= 249 * 9 < 2500
        w1[q] = af2{a.x, a.y};
    }
    const int k1b = tt / 25, t1 = tt - 25 * k1b;        // pass 2: sub-sequence and position
    float mx[10], macc[2][5];
#pragma unroll
    for (int q = 0; q < 10; ++q) mx[q] = 0.f;
    const int b0 = bx * binsPerBlock;
    const int nb = (B - b0) < binsPerBlock ? (B - b0) : binsPerBlock;
    const int nTr = nb * nSeg;   // transforms of this block: rows b0 nSeg .. of X (ALIAS: of this PRN's Z), consecutive
    const float2 *x0 = X + ((ALIAS ? (size_t)p * B : (size_t)0) + (size_t)b0) * nSeg * N;
    float2 xn[10];   // the next transform's spectrum, fetched under this one
#pragma unroll
    for (int q = 0; q < 10; ++q) xn[q] = x0[tt + 250 * q];
    __syncthreads();
    int seg = 0, b = b0;
    for (int e = 0; e < nTr; ++e) {
        if (act) {
            // spectrum product (correlator.py:75) and pass 1: radix 10 over the stride-250 elements, twiddle W^(t k1)
            af2 v[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) v[q] = ALIAS ? af2{xn[q].x, xn[q].y} : acq_cmul(af2{xn[q].x, xn[q].y}, rc[ALIAS ? 0 : q]);
            if (e + 1 < nTr) {
#pragma unroll
                for (int q = 0; q < 10; ++q) xn[q] = x0[(size_t)(e + 1) * N + t + 250 * q];
            }
            acq_idft10(v);
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const af2 y = k ? acq_cmul(v[k], w1[k]) : v[k];
                sA[k * SS + t] = make_float2(y.x, y.y);
            }
        }
        __syncthreads();
        if (act) {
            // pass 2: radix 10 inside each 250-point sub-sequence (stride 25), twiddle W250^(t1 k2)
            af2 v[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const float2 a = sA[k1b * SS + t1 + 25 * q];
                v[q] = af2{a.x, a.y};
            }
            acq_idft10(v);
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const float2 w = sW250[t1 * k];   // t1 k2 <= 24 * 9 < 250
                const af2 y = k ? acq_cmul(v[k], af2{w.x, w.y}) : v[k];
                sB[k1b * SS + k * 25 + t1] = make_float2(y.x, y.y);
            }
        }
        __syncthreads();
        if (act) {
            // pass 3: radix 5 inside each 25-point sub-sequence (stride 5), twiddle W25^(t2 k3); two butterflies per thread
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int g = t + 250 * h, sq = g / 5, t2 = g - 5 * sq;
                const int base = (sq / 10) * SS + (sq % 10) * 25;
                af2 c[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const float2 a = sB[base + t2 + 5 * q];
                    c[q] = af2{a.x, a.y};
                }
                acq_idft5(c[0], c[1], c[2], c[3], c[4]);
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float2 w = sW25[t2 * k];   // t2 k3 <= 16
                    const af2 y = k ? acq_cmul(c[k], af2{w.x, w.y}) : c[k];
                    sA[sq * 25 + k * 5 + t2] = make_float2(y.x, y.y);
                }
            }
        }
        __syncthreads();
        float *sMag = reinterpret_cast<float *>(sB);   // (pass 2's output is consumed: its buffer takes the magnitudes in natural order)
        const bool last = ALIAS || seg == nSeg - 1;     // block-uniform
        if (act) {
            // pass 4: the last radix 5; output index k = k1 + 10 (k2 + 10 (k3a + 5 k3b)), sq = 10 k1 + k2
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int g = t + 250 * h, sq = g / 5, k3a = g - 5 * sq;
                af2 d[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const float2 a = sA[sq * 25 + k3a * 5 + q];
                    d[q] = af2{a.x, a.y};
                }
                acq_idft5(d[0], d[1], d[2], d[3], d[4]);
                const int k1 = sq / 10, k2 = sq - 10 * k1;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float mg = __builtin_amdgcn_sqrtf(d[k].x * d[k].x + d[k].y * d[k].y);   // | . |  (correlator.py:80; v_sqrt_f32, 1 ulp)
                    macc[h][k] = (ALIAS || seg == 0) ? mg : macc[h][k] + mg;
                    if (last) sMag[k1 + 10 * (k2 + 10 * (k3a + 5 * k))] = macc[h][k];
                }
            }
        }
        __syncthreads();   // (also orders this pass's reads of sA before the next transform's pass-1 writes)
        if constexpr (ALIAS) {
            if (act) {
                float sv = 0.f;
#pragma unroll
                for (int q = 0; q < 10; ++q) sv += sMag[t + 250 * q];   // the ten lag aliases of delay 10 t + seg (correlator.py:80-82)
                surf[((size_t)(pOffset + p) * B + b) * N + 10 * t + seg] = sv;
                atomicMax(&mpBits[(size_t)(pOffset + p) * N + 10 * t + seg], __float_as_uint(sv));   // max over the bins (:87)
            }
            if (++seg == nSeg) { seg = 0; ++b; }
        } else if (last) {
            if (act) {
                float *o = surf + ((size_t)p * B + b) * N;
#pragma unroll
                for (int q = 0; q < 10; ++q) {
                    const float sv = sMag[t + 250 * q];
                    o[t + 250 * q] = sv;
                    mx[q] = fmaxf(mx[q], sv);
                }
            }
            seg = 0; ++b;
            // (sB / sMag is rewritten only after the next transform's pass-1 barrier)
        } else ++seg;
    }
    if (!ALIAS && act) {
#pragma unroll
        for (int q = 0; q < 10; ++q) atomicMax(&mpBits[(size_t)p * N + t + 250 * q], __float_as_uint(mx[q]));
    }
}

// Wipe-off (+ fold) AND the forward 2 500-point transform of a row in one block: X[b][seg][k] = FFT2500(row)[k], row[j] =
// sum_{n < nFold} raw[i] exp(-j 2 pi f_b i / fs), i = j + (seg + n) M (coherent mode: nFold = N, one row per bin; textbook mode:
// nFold = 1, N rows per bin).  It replaces acq_wipe_fold_kernel / acq_wipe_kernel + the batched rocFFT forward pass of the fused
// modes: one launch less, and the rows never travel through memory between the two.  The transform is the inverse one of
// acq_corr2500_kernel run on the conjugate (FFT(x) = conj(IFFT_unnormalised(conj(x)))): same passes, same tables; its last pass
// knows every output's natural index, so X is written in natural order.  512 threads: all of them wipe (five samples each), the
// first 250 transform.
__global__ __launch_bounds__(512) void acq_wipe_fft2500_kernel(const int16_t *__restrict__ iq, int nFold, double binStart, double binStep, double invFs,
                                                               const float2 *__restrict__ tw, float2 *__restrict__ X, float *__restrict__ mp,
                                                               long long mpLen)
{
    constexpr int N = kAcqFusedLen, SS = kAcqSubStride;
    __shared__ float2 sA[10 * SS], sB[10 * SS];
    __shared__ float2 sW250[256], sW25[32];
    acq_clear(mp, mpLen);
    const int t = threadIdx.x, b = blockIdx.x, seg = blockIdx.y, nSeg = gridDim.y;
    const bool act = t < 250;
    if (act) sW250[t] = tw[10 * t];
    if (t < 25) sW25[t] = tw[100 * t];
    const double cyclesPerSample = (binStart + binStep * b) * invFs;
    const int *x = reinterpret_cast<const int *>(iq);
    float2 *row = sB;   // the wiped row in natural order (2 500 of the 2 810 entries)
    for (int j = t; j < N; j += 512) {
        float ar = 0.f, ai = 0.f;
        for (int n = 0; n < nFold; ++n) {
            const int i = j + (seg + n) * N;
            const int v = x[i];
            const float re = (float)(short)(v & 0xFFFF), im = (float)(v >> 16);
            double ph = cyclesPerSample * (double)i;
            ph -= floor(ph);
            const float f = (float)ph;
            const float c = __builtin_amdgcn_cosf(f), sn = -__builtin_amdgcn_sinf(f);
            ar += re * c - im * sn;
            ai += re * sn + im * c;
        }
        row[j] = make_float2(ar, -ai);   // conjugated: the inverse passes below then give conj(FFT(row))
    }
    __syncthreads();
    if (act) {   // pass 1: radix 10 over the stride-250 elements, twiddle W^(t k1)
        af2 v[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const float2 a = row[t + 250 * q];
            v[q] = af2{a.x, a.y};
        }
        acq_idft10(v);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float2 w = tw[t * k];
            const af2 y = k ? acq_cmul(v[k], af2{w.x, w.y}) : v[k];
            sA[k * SS + t] = make_float2(y.x, y.y);
        }
    }
    __syncthreads();
    const int k1b = (act ? t : 0) / 25, t1 = (act ? t : 0) - 25 * k1b;
    if (act) {   // pass 2: radix 10 inside each 250-point sub-sequence (stride 25), twiddle W250^(t1 k2)
        af2 v[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const float2 a = sA[k1b * SS + t1 + 25 * q];
            v[q] = af2{a.x, a.y};
        }
        acq_idft10(v);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float2 w = sW250[t1 * k];
            const af2 y = k ? acq_cmul(v[k], af2{w.x, w.y}) : v[k];
            sB[k1b * SS + k * 25 + t1] = make_float2(y.x, y.y);
        }
    }
    __syncthreads();
    if (act) {   // pass 3: radix 5 inside each 25-point sub-sequence (stride 5), twiddle W25^(t2 k3)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int g = t + 250 * h, sq = g / 5, t2 = g - 5 * sq;
            const int base = (sq / 10) * SS + (sq % 10) * 25;
            af2 c[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const float2 a = sB[base + t2 + 5 * q];
                c[q] = af2{a.x, a.y};
            }
            acq_idft5(c[0], c[1], c[2], c[3], c[4]);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float2 w = sW25[t2 * k];
                const af2 y = k ? acq_cmul(c[k], af2{w.x, w.y}) : c[k];
                sA[sq * 25 + k * 5 + t2] = make_float2(y.x, y.y);
            }
        }
    }
    __syncthreads();
    if (act) {   // pass 4: the last radix 5; output index k = k1 + 10 (k2 + 10 (k3a + 5 k3b)), sq = 10 k1 + k2 -- natural order into sB
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int g = t + 250 * h, sq = g / 5, k3a = g - 5 * sq;
            af2 d[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const float2 a = sA[sq * 25 + k3a * 5 + q];
                d[q] = af2{a.x, a.y};
            }
            acq_idft5(d[0], d[1], d[2], d[3], d[4]);
            const int k1 = sq / 10, k2 = sq - 10 * k1;
#pragma unroll
            for (int k = 0; k < 5; ++k) sB[k1 + 10 * (k2 + 10 * (k3a + 5 * k))] = make_float2(d[k].x, -d[k].y);   // conjugated back
        }
    }
    __syncthreads();
    float2 *o = X + ((size_t)b * nSeg + seg) * N;
    for (int k = t; k < N; k += 512) o[k] = sB[k];
}

// The reference's non-coherent search at S = 10 x 2 500: spectrum product (correlator.py:75) and the first, radix-10
// decimation-in-frequency stage of the 25 000-point inverse transform, twiddles included:
//     Z[p][b][k0][m] = W^(m k0) sum_q X[b][m + 2500 q] Rc[p][m + 2500 q] W10^(q k0),   W = exp(+j 2 pi / 25000)
// (n = m + 2500 q, k = k0 + 10 m':  W^(n k) = W^(m k0) W2500^(m m') W10^(q k0)).  The 25 000-point rocFFT pass, the product
// buffer it transformed in place and the fold pass over it are replaced by this kernel and ten 2 500-point transforms per
// (PRN, bin) in acq_corr2500_kernel<true>.
__global__ __launch_bounds__(256) void acq_radix10_kernel(const float2 *__restrict__ X, const float2 *__restrict__ Rc, const float2 *__restrict__ tw25k,
                                                         int B, int M, float2 *__restrict__ Z)   // (M = 2 500, or 4 000 / 5 000 with tw25k = exp(+j 2 pi n / 10 M))
{
    const int S = 10 * M;
    const int m = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, p = blockIdx.z;
    if (m >= M) return;
    const float2 *x = X + (size_t)b * S + m, *r = Rc + (size_t)p * S + m;
    af2 v[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        const float2 a = x[M * q], c = r[M * q];
        v[q] = acq_cmul(af2{a.x, a.y}, af2{c.x, c.y});
    }
    acq_idft10(v);
    float2 *z = Z + ((size_t)p * B + b) * S + m;
#pragma unroll
    for (int k0 = 0; k0 < 10; ++k0) {
        af2 y = v[k0];
        if (k0) {
            const float2 w = tw25k[m * k0];   // m k0 <= (M - 1) * 9 < 10 M
            y = acq_cmul(y, af2{w.x, w.y});
        }
        // (streaming store: the 800 MB of Z written per 32 x 125 search must not push X and Rc -- re-read by every block -- out of L2)
        __builtin_nontemporal_store(y.x, &z[(size_t)k0 * M].x);
        __builtin_nontemporal_store(y.y, &z[(size_t)k0 * M].y);
    }
}

}  // namespace dpe

namespace dpe {

// ---- fine frequency (correlator.py:105-133)
struct AcqFineChan {
    double rc, fc;   // coarse code phase (chips) and code frequency
};

__global__ __launch_bounds__(256) void acq_sum_kernel(const int16_t *__restrict__ iq, int S, long long *__restrict__ sums)
{
    const int *x = reinterpret_cast<const int *>(iq);
    long long sI = 0, sQ = 0;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < S; n += gridDim.x * blockDim.x) {
        const int v = x[n];
        sI += (short)(v & 0xFFFF);
        sQ += v >> 16;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sI += __shfl_xor(sI, off, 64);
        sQ += __shfl_xor(sQ, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&sums[0]), (unsigned long long)sI);
        atomicAdd(reinterpret_cast<unsigned long long *>(&sums[1]), (unsigned long long)sQ);
    }
}

// carr[p][n] = (raw[n] - mean) * chips_p[floor(t_n fc_p + rc_p) mod 1023] for n < S, 0 up to the FFT length (:114-121)
__global__ __launch_bounds__(256) void acq_fine_build_kernel(const int16_t *__restrict__ iq, int S, int C, double fs,
                                                             const AcqFineChan *__restrict__ chan,
                                                             const long long *__restrict__ sums,
                                                             const int8_t *__restrict__ chips, float2 *__restrict__ out)
{
    const int p = blockIdx.y;
    const AcqFineChan ch = chan[p];
    const float mRe = (float)((double)sums[0] / (double)S), mIm = (float)((double)sums[1] / (double)S);
    const int *x = reinterpret_cast<const int *>(iq);
    const int8_t *cp = chips + (size_t)p * 1024;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < C; n += gridDim.x * blockDim.x) {
        float2 o = make_float2(0.f, 0.f);
        if (n < S) {
            const int v = x[n];
            const double t = (double)n / fs;                                   // rawfile.time_idc
            const double ph = __dadd_rn(__dmul_rn(t, ch.fc), ch.rc);           // time_idc * fc + rc, two roundings as numpy
            const long long ci = (long long)floor(ph);
            const float r = (float)cp[(int)(((ci % kLCA) + kLCA) % kLCA)];
            o = make_float2(((float)(short)(v & 0xFFFF) - mRe) * r, ((float)(v >> 16) - mIm) * r);
        }
        out[(size_t)p * C + n] = o;
    }
}

// first maximum of |fftshift(X)| over the shifted indices [iLo, iHi] (:124-127); one block per PRN
__global__ __launch_bounds__(256) void acq_fine_peak_kernel(const float2 *__restrict__ X, int C, int iLo, int iHi,
                                                            int *__restrict__ idxOut, float2 *__restrict__ valOut)
{
    const int p = blockIdx.x;
    const float2 *x = X + (size_t)p * C;
    unsigned long long best = 0ull;
    for (int i = iLo + threadIdx.x; i <= iHi; i += 256) {
        const float2 v = x[(i + C / 2) % C];
        const float m = v.x * v.x + v.y * v.y;
        const unsigned long long key = ((unsigned long long)__float_as_uint(m) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        best = key > best ? key : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    __shared__ unsigned long long sB[4];
    if ((threadIdx.x & 63) == 0) sB[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = sB[0];
        for (int q = 1; q < 4; ++q) b = sB[q] > b ? sB[q] : b;
        const int i = (int)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFull));
        idxOut[p] = i;
        valOut[p] = x[(i + C / 2) % C];
    }
}

// Peak statistics of one PRN's max_percode row (correlator.py:86-103, _trim_mean :546-564) on the device, so that
// dpe_acq_results copies 48 bytes per PRN instead of the row: the masked row's maximum (cppr), and the mean of the values
// strictly between the 5 % and 95 % percentiles (cppm).  Percentiles as numpy / scipy define them (linear interpolation
// between the order statistics floor(pos) and floor(pos)+1, pos = (M-1) q / 100): the order statistics are EXACT -- an
// 8-bit-digit radix select over the bit patterns (non-negative floats order like unsigned integers) -- and the
// interpolation repeats the host expression; only the fp64 summation order of the trimmed mean differs from a serial loop.
struct AcqStats {
    float peak, maxRest;
    int ci, di;
    double sum;
    long long cnt;
    double lo, hi;
};

__device__ __forceinline__ unsigned int block_sum_u32(unsigned int v, unsigned int *sTmp)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sTmp[threadIdx.x >> 6] = v;
    __syncthreads();
    return sTmp[0] + sTmp[1] + sTmp[2] + sTmp[3];
}

// Per PRN, one block: max_code_idx = first maximum of max_percode, max_dopp_idx = first maximum of that column of the
// surface (correlator.py:88-89), then the peak statistics.  The row is staged in LDS when it fits (rowInLds: M floats of
// dynamic shared memory) -- the two rank selections and their neighbours make about a dozen passes over it.
__global__ __launch_bounds__(256) void acq_stats_kernel(const float *__restrict__ surf, const float *__restrict__ mp, int B, int M,
                                                        int rowInLds, int maskS, int iLo, double fLo, int iHi, double fHi,
                                                        int *__restrict__ codeIdx, int *__restrict__ doppIdx,
                                                        AcqStats *__restrict__ out)
{
    const int p = blockIdx.x, tid = threadIdx.x;
    extern __shared__ float sRow[];
    const float *m = mp + (size_t)p * M;
    if (rowInLds) {
        for (int j = tid; j < M; j += 256) sRow[j] = m[j];
        __syncthreads();
        m = sRow;
    }
    __shared__ unsigned int hist[2 * 2048];
    __shared__ unsigned int sTmp[8];
    __shared__ unsigned int sSel[4];
    __shared__ float sF[4];
    __shared__ double sD[4];
    __shared__ unsigned long long sB[4];
    __shared__ float sCand[2][64];
    __shared__ unsigned int sCnt[2];
    auto block_argmax = [&](unsigned long long best) -> int {   // packed (value bits, ~index): larger value, then smaller index
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        __syncthreads();
        if ((tid & 63) == 0) sB[tid >> 6] = best;
        __syncthreads();
        unsigned long long b = sB[0];
        for (int q = 1; q < 4; ++q) b = sB[q] > b ? sB[q] : b;
        return (int)(0xFFFFFFFFu - (unsigned)(b & 0xFFFFFFFFull));
    };
    unsigned long long best = 0ull;
    float rowSum = 0.f;   // (only sets the bin width of the selection below: any positive value gives the same result)
    for (int j = tid; j < M; j += 256) {
        const float mj = m[j];
        rowSum += mj;
        const unsigned long long key = ((unsigned long long)__float_as_uint(mj) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)j);
        best = key > best ? key : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rowSum += __shfl_xor(rowSum, off, 64);
    if ((tid & 63) == 0) sF[tid >> 6] = rowSum;
    for (unsigned int i = tid; i < 2048u; i += 256) hist[i] = 0u;
    if (tid < 2) sCnt[tid] = 0u;
    const int ci = block_argmax(best);   // (its barriers also publish sF, the cleared histogram and the counters)
    const float binScale = 256.0f * (float)M / fmaxf((sF[0] + sF[1]) + (sF[2] + sF[3]), 1e-30f);   // 256 bins per row mean
    // the peak's column of the surface: loads issued here, reduced at the end (they arrive under the selection passes)
    unsigned long long bestD = 0ull;
    for (int b = tid; b < B; b += 256) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(surf[((size_t)p * B + b) * M + ci]) << 32) |
                                       (unsigned long long)(0xFFFFFFFFu - (unsigned)b);
        bestD = key > bestD ? key : bestD;
    }
    // row value with the +-maskS delays about the peak zeroed (indices wrap at both ends, see dpe_hip.h)
    auto val = [&](int j) -> float {
        int d = j - ci;
        if (d < 0) d += M;
        const int c = d < M - d ? d : M - d;
        return c <= maskS ? 0.f : m[j];
    };
    // the iLo-th and the iHi-th smallest (0-based) of the masked row, found TOGETHER: radix selection over the values' bit
    // patterns (>= 0, so unsigned order is float order) in three passes of 11 / 11 / 10 bits with one histogram per rank --
    // two ranks x four 8-bit passes one after the other were ~50 block barriers of a 26 us kernel
    float selV[2], mxT = 0.f;
    bool selected = false;
    {
        // Fast path: ONE histogram over 2048 linear bins of width mean / 256 (monotone in the value, so the k-th smallest lies in
        // the bin where the running count passes k), then the handful of values of that bin -- a few per bin for a noise-like row --
        // are ranked by counting inside one wave.  Exact like the radix passes below, which remain for rows whose selected bins
        // hold more than 64 values (constant rows, heavy ties): five block barriers instead of twelve.
        auto bin_of = [&](float v) -> unsigned int {
            const float t = v * binScale;
            return t >= 2047.f ? 2047u : (unsigned int)t;
        };
        for (int j = tid; j < M; j += 256) {
            const float vj = val(j);
            mxT = fmaxf(mxT, vj);
            atomicAdd(&hist[bin_of(vj)], 1u);
        }
        __syncthreads();
        unsigned int tot = 0u;
        for (unsigned int q = 0; q < 8u; ++q) tot += hist[tid * 8u + q];
        unsigned int inc = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(inc, off, 64);
            if ((tid & 63) >= off) inc += o;
        }
        if ((tid & 63) == 63) sTmp[tid >> 6] = inc;
        __syncthreads();
        unsigned int excl0 = inc - tot;
        for (int q = 0; q < (tid >> 6); ++q) excl0 += sTmp[q];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned int kr = (unsigned int)(r ? iHi : iLo);
            if (tot && kr >= excl0 && kr < excl0 + tot) {   // the rank lies among this thread's bins
                unsigned int excl = excl0;
                for (unsigned int q = 0; q < 8u; ++q) {
                    const unsigned int c = hist[tid * 8u + q];
                    if (kr < excl + c) { sSel[2 * r] = tid * 8u + q; sSel[2 * r + 1] = kr - excl; break; }
                    excl += c;
                }
            }
        }
        __syncthreads();
        const unsigned int selBin[2] = {sSel[0], sSel[2]}, kIn[2] = {sSel[1], sSel[3]};
        const unsigned int nC[2] = {hist[selBin[0]], hist[selBin[1]]};
        if (nC[0] <= 64u && nC[1] <= 64u) {   // (block-uniform)
            for (int j = tid; j < M; j += 256) {
                const float vj = val(j);
                const unsigned int b = bin_of(vj);
                if (b == selBin[0]) sCand[0][atomicAdd(&sCnt[0], 1u)] = vj;
                if (b == selBin[1]) sCand[1][atomicAdd(&sCnt[1], 1u)] = vj;
            }
            __syncthreads();
            const int r = tid >> 6, l = tid & 63;
            if (r < 2) {
                const int n = (int)nC[r];
                const float v = l < n ? sCand[r][l] : 3.0e38f;
                unsigned int below = 0u;
                for (int j = 0; j < n; ++j) {
                    const float cj = sCand[r][j];
                    below += (cj < v || (cj == v && j < l)) ? 1u : 0u;
                }
                if (l < n && below == kIn[r]) sF[r] = v;
            }
            __syncthreads();
            selV[0] = sF[0];
            selV[1] = sF[1];
            selected = true;
        }
    }
    if (!selected) {
        __syncthreads();   // (the fast path's histogram is dead: the passes below clear and reuse it)
        unsigned int prefix[2] = {0u, 0u}, kk[2] = {(unsigned int)iLo, (unsigned int)iHi}, mask = 0u;
        const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            const int shift = shifts[pass];
            const unsigned int nbin = 1u << widths[pass];
            for (unsigned int i = tid; i < 2u * nbin; i += 256) hist[i] = 0u;
            __syncthreads();
            for (int j = tid; j < M; j += 256) {
                const float vj = val(j);
                mxT = fmaxf(mxT, vj);   // (the masked maximum rides along; three times the same value)
                const unsigned int u = __float_as_uint(vj), dgt = (u >> shift) & (nbin - 1u);
                if ((u & mask) == prefix[0]) atomicAdd(&hist[dgt], 1u);
                if ((u & mask) == prefix[1]) atomicAdd(&hist[nbin + dgt], 1u);
            }
            __syncthreads();
            const unsigned int per = nbin / 256u;   // 8 or 4 consecutive digits per thread
            unsigned int tot[2] = {0u, 0u};
            for (unsigned int q = 0; q < per; ++q) { tot[0] += hist[tid * per + q]; tot[1] += hist[nbin + tid * per + q]; }
            unsigned int inc[2] = {tot[0], tot[1]};
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned int o0 = __shfl_up(inc[0], off, 64), o1 = __shfl_up(inc[1], off, 64);
                if ((tid & 63) >= off) { inc[0] += o0; inc[1] += o1; }
            }
            if ((tid & 63) == 63) { sTmp[tid >> 6] = inc[0]; sTmp[4 + (tid >> 6)] = inc[1]; }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                unsigned int excl = inc[r] - tot[r];
                for (int q = 0; q < (tid >> 6); ++q) excl += sTmp[4 * r + q];
                if (tot[r] && kk[r] >= excl && kk[r] < excl + tot[r]) {   // the rank lies among this thread's digits
                    for (unsigned int q = 0; q < per; ++q) {
                        const unsigned int c = hist[r * nbin + tid * per + q];
                        if (kk[r] < excl + c) { sSel[2 * r] = prefix[r] | ((tid * per + q) << shift); sSel[2 * r + 1] = kk[r] - excl; break; }
                        excl += c;
                    }
                }
            }
            __syncthreads();
            prefix[0] = sSel[0]; kk[0] = sSel[1]; prefix[1] = sSel[2]; kk[1] = sSel[3];
            mask |= (nbin - 1u) << shift;
        }
        selV[0] = __uint_as_float(prefix[0]);
        selV[1] = __uint_as_float(prefix[1]);
    }
    // the order statistics after the two selected ones (x itself if it occurs beyond rank k, else the smallest larger value)
    // and the masked maximum: one pass over the row, one round of reductions
    float mx, nxt[2];
    {
        unsigned int le[2] = {0u, 0u};
        float mn[2] = {3.0e38f, 3.0e38f};
        for (int j = tid; j < M; j += 256) {
            const float v = val(j);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                le[r] += v <= selV[r] ? 1u : 0u;
                mn[r] = (v > selV[r] && v < mn[r]) ? v : mn[r];
            }
        }
        float mxw = mxT;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            le[0] += __shfl_xor(le[0], off, 64);
            le[1] += __shfl_xor(le[1], off, 64);
            const float o0 = __shfl_xor(mn[0], off, 64), o1 = __shfl_xor(mn[1], off, 64);
            mn[0] = o0 < mn[0] ? o0 : mn[0];
            mn[1] = o1 < mn[1] ? o1 : mn[1];
            mxw = fmaxf(mxw, __shfl_xor(mxw, off, 64));
        }
        __shared__ float sR[4][3];
        __shared__ unsigned int sL[4][2];
        if ((tid & 63) == 0) {
            sR[tid >> 6][0] = mn[0]; sR[tid >> 6][1] = mn[1]; sR[tid >> 6][2] = mxw;
            sL[tid >> 6][0] = le[0]; sL[tid >> 6][1] = le[1];
        }
        __syncthreads();
        mx = fmaxf(fmaxf(sR[0][2], sR[1][2]), fmaxf(sR[2][2], sR[3][2]));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned int nLe = sL[0][r] + sL[1][r] + sL[2][r] + sL[3][r];
            const float mnr = fminf(fminf(sR[0][r], sR[1][r]), fminf(sR[2][r], sR[3][r]));
            nxt[r] = nLe > (unsigned int)(r ? iHi : iLo) + 1u ? selV[r] : mnr;
        }
    }
    // percentiles: lo + (hi - lo) * f with a float difference, as the host expression (and scipy) evaluate it
    double pLo = (double)selV[0], pHi = (double)selV[1];
    if (iLo + 1 < M) pLo = (double)selV[0] + (double)(nxt[0] - selV[0]) * fLo;
    if (iHi + 1 < M) pHi = (double)selV[1] + (double)(nxt[1] - selV[1]) * fHi;
    double sum = 0.0;
    unsigned int cnt = 0u;
    for (int j = tid; j < M; j += 256) {
        const float v = val(j);
        if ((double)v > pLo && (double)v < pHi) { sum += (double)v; ++cnt; }
    }
    const unsigned int nCnt = block_sum_u32(cnt, sTmp);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if ((tid & 63) == 0) sD[tid >> 6] = sum;
    __syncthreads();
    const int di = block_argmax(bestD);
    if (tid == 0) {
        codeIdx[p] = ci; doppIdx[p] = di;
        AcqStats r;
        r.peak = m[ci]; r.maxRest = mx; r.ci = ci; r.di = di;
        r.sum = ((sD[0] + sD[1]) + sD[2]) + sD[3]; r.cnt = (long long)nCnt; r.lo = pLo; r.hi = pHi;
        out[p] = r;
    }
}

// ---- the statistics of rows of up to 2 560 delays (the reference's 2.5 Msps) with the row in REGISTERS: same selections, same results
// as acq_stats_kernel, whose chain of six LDS passes over the row and ~80 ds_bpermute shuffle steps was 10 of its 14.6 us.  Each of the
// 256 threads owns ten delays; the masked values are formed once; every pass is ten register operations and a reduction on the vector
// ALU (DPP row operations: ~8 cycles a step where a shuffle through the LDS crossbar takes ~100), waves meet through LDS eight times.
#define DPE_RED63(T, v, ident, OPEXPR)                                                                                                          \
    {                                                                                                                                           \
        const int id_ = __builtin_bit_cast(int, (T)(ident));                                                                                    \
        T o_;                                                                                                                                   \
        o_ = __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(id_, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)); v = OPEXPR;           \
        o_ = __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(id_, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)); v = OPEXPR;           \
        o_ = __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(id_, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false)); v = OPEXPR;          \
        o_ = __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(id_, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false)); v = OPEXPR;          \
        o_ = __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(id_, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false)); v = OPEXPR;          \
        o_ = __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(id_, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false)); v = OPEXPR;          \
        v = __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));                                                   \
    }
__device__ __forceinline__ float wave_max63(float v) { DPE_RED63(float, v, 0.f, fmaxf(v, o_)); return v; }               // (values >= 0)
__device__ __forceinline__ float wave_min63(float v) { DPE_RED63(float, v, 3.0e38f, fminf(v, o_)); return v; }
__device__ __forceinline__ float wave_sum63(float v) { DPE_RED63(float, v, 0.f, v + o_); return v; }
__device__ __forceinline__ unsigned int wave_minu63(unsigned int v) { DPE_RED63(unsigned int, v, 0xFFFFFFFFu, (v < o_ ? v : o_)); return v; }
__device__ __forceinline__ unsigned int wave_addu63(unsigned int v) { DPE_RED63(unsigned int, v, 0u, v + o_); return v; }
#undef DPE_RED63
__device__ __forceinline__ double wave_sumd63(double v)
{
#define DPE_STEP(CTRL, RM)                                                                                                                      \
    {                                                                                                                                           \
        const long long b_ = __builtin_bit_cast(long long, v);                                                                                  \
        const unsigned int lo_ = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)b_, CTRL, RM, 0xf, false);                     \
        const unsigned int hi_ = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(b_ >> 32), CTRL, RM, 0xf, false);             \
        v += __builtin_bit_cast(double, ((unsigned long long)hi_ << 32) | lo_);                                                                 \
    }
    DPE_STEP(0xB1, 0xf) DPE_STEP(0x4E, 0xf) DPE_STEP(0x141, 0xf) DPE_STEP(0x140, 0xf) DPE_STEP(0x142, 0xa) DPE_STEP(0x143, 0xc)
#undef DPE_STEP
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)b, 63), hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// inclusive prefix sums over the 64 lanes
__device__ __forceinline__ unsigned int wave_scan_addu(unsigned int v)
{
#define DPE_STEP(CTRL, RM) v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, RM, 0xf, false);
    DPE_STEP(0x111, 0xf) DPE_STEP(0x112, 0xf) DPE_STEP(0x114, 0xf) DPE_STEP(0x118, 0xf) DPE_STEP(0x142, 0xa) DPE_STEP(0x143, 0xc)
#undef DPE_STEP
    return v;
}

constexpr int kAcqStatsR = 10;   // delays per thread: rows up to 2 560
__global__ __launch_bounds__(256) void acq_stats_small_kernel(const float *__restrict__ surf, const float *__restrict__ mp, int B, int M, int maskS, int iLo,
                                                              double fLo, int iHi, double fHi, int *__restrict__ codeIdx, int *__restrict__ doppIdx,
                                                              AcqStats *__restrict__ out)
{
    constexpr int R = kAcqStatsR;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    __shared__ unsigned int hist[2 * 2048];
    __shared__ unsigned int sTmp[8], sSel[4], sCnt[2], sWu[4][4];
    __shared__ float sCand[2][64], sWf[4][4], sF[2];
    __shared__ double sWd[4];
    const float *m = mp + (size_t)p * M;
    float rv[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int j = tid + 256 * i;
        rv[i] = j < M ? m[j] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) hist[tid + 256 * i] = 0u;
    if (tid < 2) sCnt[tid] = 0u;
    // max_code_idx: the FIRST maximum of the row (correlator.py:88) -- largest value, then smallest index
    float lsum = 0.f, lmax = -1.f;
    unsigned int lidx = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int j = tid + 256 * i;
        if (j < M) {
            lsum += rv[i];
            if (rv[i] > lmax) { lmax = rv[i]; lidx = (unsigned int)j; }
        }
    }
    {
        const float wmax = wave_max63(fmaxf(lmax, 0.f));
        const unsigned int widx = wave_minu63(lmax == wmax ? lidx : 0xFFFFFFFFu);
        const float wsum = wave_sum63(lsum);
        if (lane == 0) { sWf[w][0] = wmax; sWu[w][0] = widx; sWf[w][1] = wsum; }
    }
    __syncthreads();   // (also: the cleared histogram and counters)
    float peak = sWf[0][0];
#pragma unroll
    for (int q = 1; q < 4; ++q) peak = fmaxf(peak, sWf[q][0]);
    unsigned int ciU = 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q < 4; ++q) ciU = (sWf[q][0] == peak && sWu[q][0] < ciU) ? sWu[q][0] : ciU;
    const int ci = (int)ciU;
    const float binScale = 256.0f * (float)M / fmaxf((sWf[0][1] + sWf[1][1]) + (sWf[2][1] + sWf[3][1]), 1e-30f);   // 256 bins per row mean
    // the peak's column of the surface: loads issued here, reduced at the end (they arrive under the selection passes)
    float dmax = -1.f;
    unsigned int didx = 0xFFFFFFFFu;
    for (int b = tid; b < B; b += 256) {
        const float v = surf[((size_t)p * B + b) * M + ci];
        if (v > dmax) { dmax = v; didx = (unsigned int)b; }
    }
    // row values with the +-maskS delays about the peak zeroed (indices wrap at both ends, see dpe_hip.h), formed once
    float vm[R], mxT = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int j = tid + 256 * i;
        int d = j - ci;
        if (d < 0) d += M;
        const int c = d < M - d ? d : M - d;
        vm[i] = c <= maskS ? 0.f : rv[i];
        if (j < M) mxT = fmaxf(mxT, vm[i]);
    }
    auto bin_of = [&](float v) -> unsigned int {
        const float t = v * binScale;
        return t >= 2047.f ? 2047u : (unsigned int)t;
    };
    // the iLo-th and the iHi-th smallest (0-based) of the masked row: one histogram over 2048 linear bins of width mean / 256, the handful
    // of values of the selected bins ranked inside one wave; rows with more than 64 values in a selected bin take the radix passes below
    float selV[2];
    bool selected = false;
    {
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (tid + 256 * i < M) atomicAdd(&hist[bin_of(vm[i])], 1u);
        __syncthreads();
        unsigned int tot = 0u;
#pragma unroll
        for (unsigned int q = 0; q < 8u; ++q) tot += hist[tid * 8u + q];
        const unsigned int inc = wave_scan_addu(tot);
        if (lane == 63) sTmp[w] = inc;
        __syncthreads();
        unsigned int excl0 = inc - tot;
        for (int q = 0; q < w; ++q) excl0 += sTmp[q];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned int kr = (unsigned int)(r ? iHi : iLo);
            if (tot && kr >= excl0 && kr < excl0 + tot) {   // the rank lies among this thread's bins
                unsigned int excl = excl0;
                for (unsigned int q = 0; q < 8u; ++q) {
                    const unsigned int c = hist[tid * 8u + q];
                    if (kr < excl + c) { sSel[2 * r] = tid * 8u + q; sSel[2 * r + 1] = kr - excl; break; }
                    excl += c;
                }
            }
        }
        __syncthreads();
        const unsigned int selBin[2] = {sSel[0], sSel[2]}, kIn[2] = {sSel[1], sSel[3]};
        const unsigned int nC[2] = {hist[selBin[0]], hist[selBin[1]]};
        if (nC[0] <= 64u && nC[1] <= 64u) {   // (block-uniform)
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (tid + 256 * i < M) {
                    const unsigned int b = bin_of(vm[i]);
                    if (b == selBin[0]) sCand[0][atomicAdd(&sCnt[0], 1u)] = vm[i];
                    if (b == selBin[1]) sCand[1][atomicAdd(&sCnt[1], 1u)] = vm[i];
                }
            }
            __syncthreads();
            if (w < 2) {
                const int n = __builtin_amdgcn_readfirstlane((int)nC[w]);
                const float v = lane < n ? sCand[w][lane] : 3.0e38f;
                unsigned int below = 0u;
                for (int j = 0; j < n; ++j) {
                    const float cj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j));
                    below += (cj < v || (cj == v && j < lane)) ? 1u : 0u;
                }
                if (lane < n && below == kIn[w]) sF[w] = v;
            }
            __syncthreads();
            selV[0] = sF[0];
            selV[1] = sF[1];
            selected = true;
        }
    }
    if (!selected) {
        // radix selection over the values' bit patterns (>= 0, so unsigned order is float order), 11 / 11 / 10 bits, one histogram per rank
        __syncthreads();   // (the fast path's histogram is dead: the passes below clear and reuse it)
        unsigned int prefix[2] = {0u, 0u}, kk[2] = {(unsigned int)iLo, (unsigned int)iHi}, mask = 0u;
        const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            const int shift = shifts[pass];
            const unsigned int nbin = 1u << widths[pass];
            for (unsigned int i = tid; i < 2u * nbin; i += 256) hist[i] = 0u;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (tid + 256 * i < M) {
                    const unsigned int u = __float_as_uint(vm[i]), dgt = (u >> shift) & (nbin - 1u);
                    if ((u & mask) == prefix[0]) atomicAdd(&hist[dgt], 1u);
                    if ((u & mask) == prefix[1]) atomicAdd(&hist[nbin + dgt], 1u);
                }
            }
            __syncthreads();
            const unsigned int per = nbin / 256u;   // 8 or 4 consecutive digits per thread
            unsigned int tot[2] = {0u, 0u};
            for (unsigned int q = 0; q < per; ++q) { tot[0] += hist[tid * per + q]; tot[1] += hist[nbin + tid * per + q]; }
            const unsigned int inc[2] = {wave_scan_addu(tot[0]), wave_scan_addu(tot[1])};
            if (lane == 63) { sTmp[w] = inc[0]; sTmp[4 + w] = inc[1]; }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                unsigned int excl = inc[r] - tot[r];
                for (int q = 0; q < w; ++q) excl += sTmp[4 * r + q];
                if (tot[r] && kk[r] >= excl && kk[r] < excl + tot[r]) {   // the rank lies among this thread's digits
                    for (unsigned int q = 0; q < per; ++q) {
                        const unsigned int c = hist[r * nbin + tid * per + q];
                        if (kk[r] < excl + c) { sSel[2 * r] = prefix[r] | ((tid * per + q) << shift); sSel[2 * r + 1] = kk[r] - excl; break; }
                        excl += c;
                    }
                }
            }
            __syncthreads();
            prefix[0] = sSel[0]; kk[0] = sSel[1]; prefix[1] = sSel[2]; kk[1] = sSel[3];
            mask |= (nbin - 1u) << shift;
        }
        selV[0] = __uint_as_float(prefix[0]);
        selV[1] = __uint_as_float(prefix[1]);
    }
    // the order statistics after the two selected ones (x itself if it occurs beyond rank k, else the smallest larger value)
    // and the masked maximum
    float mx, nxt[2];
    {
        unsigned int le[2] = {0u, 0u};
        float mn[2] = {3.0e38f, 3.0e38f};
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (tid + 256 * i < M) {
                const float v = vm[i];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    le[r] += v <= selV[r] ? 1u : 0u;
                    mn[r] = (v > selV[r] && v < mn[r]) ? v : mn[r];
                }
            }
        }
        const unsigned int l0 = wave_addu63(le[0]), l1 = wave_addu63(le[1]);
        const float m0 = wave_min63(mn[0]), m1 = wave_min63(mn[1]), mw = wave_max63(mxT);
        if (lane == 0) { sWu[w][1] = l0; sWu[w][2] = l1; sWf[w][2] = m0; sWf[w][3] = m1; sWf[w][0] = mw; }
        __syncthreads();
        mx = fmaxf(fmaxf(sWf[0][0], sWf[1][0]), fmaxf(sWf[2][0], sWf[3][0]));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned int nLe = sWu[0][1 + r] + sWu[1][1 + r] + sWu[2][1 + r] + sWu[3][1 + r];
            const float mnr = fminf(fminf(sWf[0][2 + r], sWf[1][2 + r]), fminf(sWf[2][2 + r], sWf[3][2 + r]));
            nxt[r] = nLe > (unsigned int)(r ? iHi : iLo) + 1u ? selV[r] : mnr;
        }
    }
    // percentiles: lo + (hi - lo) * f with a float difference, as the host expression (and scipy) evaluate it
    double pLo = (double)selV[0], pHi = (double)selV[1];
    if (iLo + 1 < M) pLo = (double)selV[0] + (double)(nxt[0] - selV[0]) * fLo;
    if (iHi + 1 < M) pHi = (double)selV[1] + (double)(nxt[1] - selV[1]) * fHi;
    double sum = 0.0;
    unsigned int cnt = 0u;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (tid + 256 * i < M) {
            const float v = vm[i];
            if ((double)v > pLo && (double)v < pHi) { sum += (double)v; ++cnt; }
        }
    }
    {
        const double ws = wave_sumd63(sum);
        const unsigned int wc = wave_addu63(cnt);
        const float wd = wave_max63(fmaxf(dmax, 0.f));
        const unsigned int wi = wave_minu63((dmax == wd && didx != 0xFFFFFFFFu) ? didx : 0xFFFFFFFFu);
        __syncthreads();   // (the reads of sWf / sWu above)
        if (lane == 0) { sWd[w] = ws; sWu[w][0] = wc; sWf[w][0] = wd; sWu[w][3] = wi; }
    }
    __syncthreads();
    if (tid == 0) {
        float dpk = sWf[0][0];
        for (int q = 1; q < 4; ++q) dpk = fmaxf(dpk, sWf[q][0]);
        unsigned int di = 0xFFFFFFFFu;
        for (int q = 0; q < 4; ++q) di = (sWf[q][0] == dpk && sWu[q][3] < di) ? sWu[q][3] : di;
        codeIdx[p] = ci; doppIdx[p] = (int)di;
        AcqStats r;
        r.peak = peak; r.maxRest = mx; r.ci = ci; r.di = (int)di;
        r.sum = ((sWd[0] + sWd[1]) + sWd[2]) + sWd[3]; r.cnt = (long long)(sWu[0][0] + sWu[1][0] + sWu[2][0] + sWu[3][0]); r.lo = pLo; r.hi = pHi;
        out[p] = r;
    }
}

}  // namespace dpe

struct dpe_acq {
    dpe_acq_config cfg;
    int S, N, M, B, P, len, chunk;   // len = FFT length (S in mode 1; M in modes 0 and 2)
    int SX;                          // samples per Doppler row after the wipe-off (M in mode 0: time-folded)
    dpe::FftPlan planFwd, planInv;
    float2 *X_d = nullptr, *Rc_d = nullptr, *Y_d = nullptr;
    float *surf_d = nullptr, *mp_d = nullptr;
    int *peakIdx_d = nullptr;   // [2][P]: max_code_idx, max_dopp_idx
    float2 *tw_d = nullptr;     // exp(+j 2 pi n / 2500): the fused searches (acq_corr2500_kernel), else null
    float2 *tw25k_d = nullptr;  // exp(+j 2 pi n / 25000): the radix-10 stage of the fused non-coherent search
    bool fused = false, fusedAlias = false;
    int fusedLen = 2500;        // fused coherent / textbook search: 2 500 (acq_corr2500_kernel) or 4 000 / 5 000 (acq_corr_mixed_kernel, dpe_acq_mixed.h)
    bool fwdPack = true;        // packed form: wipe-off + forward transform in acq_fwd25k_pack_kernel (DPE_ACQ_NO_FWD_PACK=1: wipe kernel + rocFFT + decimation)
    int cus = 256;              // compute units of the device (the packed form launches one persistent block per CU)
    bool packForm = false;      // fusedAlias through acq_corr25k_pack_kernel (dpe_acq_pack.h); DPE_ACQ_NO_PACK=1 keeps acq_radix10_kernel + the four-pass transform (A/B runs)
    float2 *Rcq_d = nullptr, *tw2_d = nullptr;   // packed form: the replicas' spectra decimated by ten, W2500^(a c) as [c][a]
    bool fusedFwd = true;   // fused modes: wipe-off and forward transform in one kernel (-DDPE_EXPERIMENTS builds: DPE_ACQ_NO_FUSED_FWD=1 keeps wipe kernel + rocFFT)
    dpe::AcqStats *stats_hd = nullptr, *stats_h = nullptr;  // per-PRN peak statistics: pinned host memory the statistics kernel writes itself (_hd: its device address)
    bool searched = false;
    // fine-frequency stage, allocated on first use
    int fineC = 0, fineLo = 0, fineHi = -1;
    dpe::FftPlan planFine;
    bool haveFine = false;
    float2 *F_d = nullptr, *fineVal_d = nullptr;
    int *fineIdx_d = nullptr;
    long long *fineSums_d = nullptr;
    dpe::AcqFineChan *fineChan_d = nullptr;
    int8_t *chips_d = nullptr;
};

extern "C" {

int dpe_acq_destroy(dpe_acq *h)
{
    if (!h) return 0;
    h->planFwd.destroy();
    h->planInv.destroy();
    h->planFine.destroy();
    void *bufs[] = {h->tw25k_d, h->tw_d, h->X_d, h->Rc_d, h->Y_d, h->surf_d, h->mp_d, h->peakIdx_d, h->F_d, h->fineVal_d, h->fineIdx_d, h->fineSums_d, h->fineChan_d, h->chips_d, h->Rcq_d, h->tw2_d};
    for (void *b : bufs) (void)hipFree(b);
    if (h->stats_h) (void)hipHostFree(h->stats_h);
    delete h;
    return 0;
}

int dpe_acq_create(const dpe_acq_config *cfg, dpe_acq **out)
{
    using namespace dpe;
    DPE_REQUIRE(cfg && out, "[Acquisition] create: null argument");
    DPE_REQUIRE(cfg->samplesPerWindow > 0 && cfg->nCodePeriods >= 1 && cfg->samplesPerWindow % cfg->nCodePeriods == 0,
                "[Acquisition] create: samplesPerWindow must be a positive multiple of nCodePeriods");
    DPE_REQUIRE(cfg->nBins >= 1 && cfg->nPrn >= 1 && cfg->nPrn <= DPE_MAX_CHAN, "[Acquisition] create: nBins / nPrn out of range");
    DPE_REQUIRE(cfg->mode >= 0 && cfg->mode <= 2, "[Acquisition] create: mode must be 0, 1 or 2");
    DPE_REQUIRE(cfg->samplingFrequency > 0, "[Acquisition] create: bad samplingFrequency");
    for (int i = 0; i < cfg->nPrn; ++i)
        DPE_REQUIRE(cfg->prn[i] >= 1 && cfg->prn[i] <= kPrnMax, "[Acquisition] create: PRN %d out of range", cfg->prn[i]);
    dpe_acq *h = new dpe_acq();
    h->cfg = *cfg;
    h->S = cfg->samplesPerWindow; h->N = cfg->nCodePeriods; h->M = h->S / h->N; h->B = cfg->nBins; h->P = cfg->nPrn;
    h->len = (cfg->mode == 1) ? h->S : h->M;
    h->SX = (cfg->mode == 0) ? h->M : h->S;
    h->chunk = cfg->prnChunk > 0 ? std::min(cfg->prnChunk, h->P) : std::min(8, h->P);
    // (the fused non-coherent form wants every PRN in one launch pair -- 32 PRNs x 25 bins: 0.19 ms at one chunk, 0.24 at chunks of
    //  8, 1.16 at chunks of 1 -- while its stage buffer stays modest: 160 MB at the reference's raster)
    if (cfg->prnChunk <= 0 && cfg->mode == 1 && h->M == 2500 && h->N == 10 && (size_t)h->P * h->B * h->S * sizeof(float2) <= ((size_t)1 << 30)) h->chunk = h->P;
    const size_t S = h->SX, B = h->B, P = h->P;
    // coherent / textbook search at 2 500 delays per code period: the fused kernel, which needs neither the product buffer nor
    // the inverse plan (DPE_ACQ_NO_FUSED=1 keeps the rocFFT chain, for A/B runs and as the cross-check of the parity tests)
    const bool wantFused = (cfg->mode == 0 || cfg->mode == 2) && (h->M == kAcqFusedLen || h->M == 4000 || h->M == 5000) && !getenv("DPE_ACQ_NO_FUSED");
    // the reference's non-coherent search at 10 x 2 500 samples: radix-10 stage + ten fused 2 500-point transforms per (PRN, bin);
    // the product buffer holds the radix-10 stage's output, no inverse plan
    const bool wantAlias = cfg->mode == 1 && (h->M == kAcqFusedLen || h->M == 4000 || h->M == 5000) && h->N == 10 && !getenv("DPE_ACQ_NO_FUSED");
    h->X_d = dev_alloc<float2>(B * S);
    h->Rc_d = dev_alloc<float2>(P * (size_t)h->len);
    bool wantPack = wantAlias && h->M == kAcqFusedLen && !(getenv("DPE_ACQ_NO_PACK") && atoi(getenv("DPE_ACQ_NO_PACK")) != 0);
    // the packed form keeps ten transforms of a (PRN, bin) in 128 000 B of dynamic LDS: a device (or partition) that does not grant
    // that much per work group runs the radix-10 kernel + four-pass transforms instead -- decided here, before the buffers are sized
    if (wantPack && (hipFuncSetAttribute(reinterpret_cast<const void *>(acq_corr25k_pack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPkLdsBytes) != hipSuccess ||
                     hipFuncSetAttribute(reinterpret_cast<const void *>(acq_fwd25k_pack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPkLdsBytes) != hipSuccess)) {
        (void)hipGetLastError();
        wantPack = false;
    }
    h->Y_d = dev_alloc<float2>(wantFused ? 1 : wantPack ? B * S : (size_t)h->chunk * B * S);   // (packed form: the bins' decimated spectra)
    h->surf_d = dev_alloc<float>(P * B * (size_t)h->M);
    h->mp_d = dev_alloc<float>(P * (size_t)h->M);
    h->peakIdx_d = dev_alloc<int>(2 * P);
    if (hipHostMalloc((void **)&h->stats_h, P * sizeof(AcqStats), hipHostMallocDefault) != hipSuccess) h->stats_h = nullptr;
    if (h->stats_h && hipHostGetDevicePointer((void **)&h->stats_hd, h->stats_h, 0) != hipSuccess) h->stats_hd = nullptr;
    if (!h->X_d || !h->Rc_d || !h->Y_d || !h->surf_d || !h->mp_d || !h->peakIdx_d || !h->stats_hd || !h->stats_h) {
        set_error("[Acquisition] create: device allocation failed");
        dpe_acq_destroy(h);
        return -1;
    }
    const int batchFwd = (int)(B * (S / h->len)), batchInv = (int)(h->chunk * B * (S / h->len));
    if (h->planFwd.create((size_t)h->len, (size_t)batchFwd, false) ||
        (!wantFused && !wantAlias && h->planInv.create((size_t)h->len, (size_t)batchInv, true))) {
        dpe_acq_destroy(h);   // (the message is rocFFT's, from dpe_fft.h)
        return -1;
    }
    // nominal-rate replica: chips[floor(n / fs * F_CA) mod 1023] (correlator.py:66, rawfile.py:164-166), fp64 on the host
    std::vector<float2> rep(P * (size_t)h->len);
    int8_t chips[kLCA];
    for (size_t p = 0; p < P; ++p) {
        gen_ca_code_host(cfg->prn[p], chips);
        for (int i = 0; i < h->len; ++i) {
            // mode 0: the replica folded over the N code periods of the window (see acq_wipe_fold_kernel)
            double acc = 0.0;
            for (int n = 0; n < (cfg->mode == 0 ? h->N : 1); ++n) {
                const double t = (double)(i + n * h->M) / cfg->samplingFrequency;
                const long long ci = (long long)std::floor(t * kFCA);
                acc += (double)chips[ci % kLCA];
            }
            rep[p * h->len + i] = make_float2((float)acc, 0.f);
        }
    }
    FftPlan pr;
    const auto finish = [&]() -> int {   // a failure here must leak neither the handle nor the temporary plan
        DPE_CHECK_HIP(hipMemcpy(h->Rc_d, rep.data(), sizeof(float2) * rep.size(), hipMemcpyHostToDevice));
        if (pr.create((size_t)h->len, P, false) || pr.exec(nullptr, h->Rc_d)) return -1;
        hipLaunchKernelGGL(acq_conj_scale_kernel, dim3(256), dim3(256), 0, 0, h->Rc_d, (long long)(P * h->len), 1.0f / (float)h->len);
        DPE_CHECK_HIP(hipDeviceSynchronize());
        return 0;
    };
    int rc = finish();
    pr.destroy();
    if (!rc && wantAlias) {
        std::vector<float2> tw((size_t)10 * h->M);
        for (int n = 0; n < 10 * h->M; ++n) {
            const double a = 6.283185307179586476925286766559 * (double)n / (double)(10 * h->M);
            tw[n] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        h->tw25k_d = dev_alloc<float2>(tw.size());
        if (!h->tw25k_d || hipMemcpy(h->tw25k_d, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("[Acquisition] create: twiddle table");
            rc = -1;
        } else h->fusedAlias = true;
    }
    if (!rc && h->fusedAlias && wantPack) {
        std::vector<float2> t2(2500);
        for (int c = 0; c < 50; ++c)
            for (int a = 0; a < 50; ++a) {
                const double ang = 6.283185307179586476925286766559 * (double)(a * c) / 2500.0;
                t2[c * 50 + a] = make_float2((float)std::cos(ang), (float)std::sin(ang));
            }
        h->tw2_d = dev_alloc<float2>(t2.size());
        h->Rcq_d = dev_alloc<float2>(P * (size_t)h->len);
        if (!h->tw2_d || !h->Rcq_d || hipMemcpy(h->tw2_d, t2.data(), sizeof(float2) * t2.size(), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("[Acquisition] create: packed-form tables");
            rc = -1;
        } else {
            hipLaunchKernelGGL(acq_decimate10_kernel, dim3(40, (unsigned)P), dim3(640), 0, 0, h->Rc_d, h->Rcq_d);
            if (hipDeviceSynchronize() != hipSuccess) { set_error("[Acquisition] create: decimating the replica spectra"); rc = -1; }
            else { h->packForm = true; h->fwdPack = !(getenv("DPE_ACQ_NO_FWD_PACK") && atoi(getenv("DPE_ACQ_NO_FWD_PACK")) != 0); }
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 1)
                h->cus = prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : prop.multiProcessorCount;
        }
    }
#ifdef DPE_EXPERIMENTS
    h->fusedFwd = !(getenv("DPE_ACQ_NO_FUSED_FWD") && atoi(getenv("DPE_ACQ_NO_FUSED_FWD")) != 0);
#endif
    if (!rc && (wantFused || wantAlias)) {
        const int L = h->M;   // (2 500 for wantAlias)
        std::vector<float2> tw((size_t)L);
        for (int n = 0; n < L; ++n) {
            const double a = 6.283185307179586476925286766559 * (double)n / (double)L;
            tw[n] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        h->tw_d = dev_alloc<float2>((size_t)L);
        if (!h->tw_d || hipMemcpy(h->tw_d, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("[Acquisition] create: twiddle table");
            rc = -1;
        } else {
            h->fused = wantFused;
            h->fusedLen = L;
        }
        if (!rc && L != kAcqFusedLen &&
            (hipFuncSetAttribute(reinterpret_cast<const void *>(acq_corr_mixed_kernel<4000, 10, 8, 5, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)AcqMixedShape<4000, 10, 8, 5>::ldsBytes) != hipSuccess ||
             hipFuncSetAttribute(reinterpret_cast<const void *>(acq_corr_mixed_kernel<5000, 10, 10, 5, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)AcqMixedShape<5000, 10, 10, 5>::ldsBytes) != hipSuccess ||
             hipFuncSetAttribute(reinterpret_cast<const void *>(acq_corr_mixed_kernel<4000, 10, 8, 5, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)AcqMixedShape<4000, 10, 8, 5>::ldsBytes) != hipSuccess ||
             hipFuncSetAttribute(reinterpret_cast<const void *>(acq_corr_mixed_kernel<5000, 10, 10, 5, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)AcqMixedShape<5000, 10, 10, 5>::ldsBytes) != hipSuccess)) {
            set_error("[Acquisition] create: LDS size of the fused search");
            rc = -1;
        }
    }
    if (rc) {
        dpe_acq_destroy(h);
        return -1;
    }
    *out = h;
    return 0;
}

int dpe_acq_search(dpe_acq *h, const int16_t *samples_dev, dpe_stream_t stream_)
{
    using namespace dpe;
    DPE_REQUIRE(h && samples_dev, "[Acquisition] search: null argument");
    hipStream_t st = (hipStream_t)stream_;
    const int S = h->SX, B = h->B, P = h->P, M = h->M;
    if (h->packForm && h->fwdPack) {
        hipLaunchKernelGGL(acq_fwd25k_pack_kernel, dim3(B), dim3(512), kPkLdsBytes, st, samples_dev, h->cfg.binStartHz, h->cfg.binStepHz,
                           1.0 / h->cfg.samplingFrequency, h->tw2_d, h->tw25k_d, h->Y_d, h->mp_d, (long long)P * M);
    } else if (h->fused && h->fusedLen == kAcqFusedLen && h->fusedFwd && h->cfg.mode != 0) {
        // textbook mode (N rows of 2 500 per bin): wipe-off and the forward transform in one launch, 0.271 -> 0.260 ms per 32-PRN window.
        // (Coherent mode keeps the two launches: one block per bin would have to fold ten periods -- a hundred sin / cos pairs per
        // thread on 125 blocks -- and measured 0.066 against 0.060 ms.)
        hipLaunchKernelGGL(acq_wipe_fft2500_kernel, dim3(B, h->cfg.mode == 0 ? 1 : h->N), dim3(512), 0, st, samples_dev, h->cfg.mode == 0 ? h->N : 1,
                           h->cfg.binStartHz, h->cfg.binStepHz, 1.0 / h->cfg.samplingFrequency, h->tw_d, h->X_d, h->mp_d, (long long)P * M);
    } else {
        if (h->cfg.mode == 0)
            hipLaunchKernelGGL(acq_wipe_fold_kernel, dim3((M + 255) / 256, B), dim3(256), 0, st, samples_dev, M, h->N,
                               h->cfg.binStartHz, h->cfg.binStepHz, 1.0 / h->cfg.samplingFrequency, h->X_d, h->mp_d, (long long)P * M);
        else
            hipLaunchKernelGGL(acq_wipe_kernel, dim3((S + 1023) / 1024, B), dim3(256), 0, st, samples_dev, S, B, h->cfg.binStartHz,
                               h->cfg.binStepHz, 1.0 / h->cfg.samplingFrequency, h->X_d, h->mp_d, (long long)P * M);
        if (h->planFwd.exec(st, h->X_d)) return -1;
    }
    if (h->fused && h->fusedLen != kAcqFusedLen) {
        // 4 / 5 Msps: the generic four-pass transform (dpe_acq_mixed.h), two blocks resident per CU; bins per block measured over 1 .. 8
        // (profiles/r5_ab_acq_rates.txt)
        const int nSeg = h->cfg.mode == 0 ? 1 : h->N;
        const int bpb = h->cfg.mode == 0 ? 8 : 4;   // 16 x 32 = 512 blocks = one round at two per CU / 32 x 32 = two rounds of ten transforms per bin
        constexpr size_t lds4 = AcqMixedShape<4000, 10, 8, 5>::ldsBytes, lds5 = AcqMixedShape<5000, 10, 10, 5>::ldsBytes;
        if (h->fusedLen == 4000)
            hipLaunchKernelGGL((acq_corr_mixed_kernel<4000, 10, 8, 5, false>), dim3((B + bpb - 1) / bpb, P), dim3(400), lds4, st, h->X_d,
                               h->Rc_d, h->tw_d, B, nSeg, bpb, 0, h->surf_d, reinterpret_cast<unsigned int *>(h->mp_d));
        else
            hipLaunchKernelGGL((acq_corr_mixed_kernel<5000, 10, 10, 5, false>), dim3((B + bpb - 1) / bpb, P), dim3(500), lds5, st, h->X_d,
                               h->Rc_d, h->tw_d, B, nSeg, bpb, 0, h->surf_d, reinterpret_cast<unsigned int *>(h->mp_d));
    } else if (h->fused) {
        // bins per block: six in the coherent mode (672 blocks = one round of the 768 resident ones); ONE in the textbook mode, whose blocks
        // already run ten transforms per bin -- 4 000 short blocks balance the tail (0.229 -> 0.206 ms per 32 x 125 search; five bins per
        // block, 800 blocks = one round plus 32 stragglers, measured 0.265)
        const int bpb = h->cfg.mode == 0 ? kAcqFusedBins : 1;
        hipLaunchKernelGGL(acq_corr2500_kernel<false>, dim3((B + bpb - 1) / bpb, P), dim3(256), 0, st, h->X_d, h->Rc_d, h->tw_d,
                           B, h->cfg.mode == 0 ? 1 : h->N, bpb, 0, h->surf_d, reinterpret_cast<unsigned int *>(h->mp_d));
    }
    for (int p0 = 0; !h->fused && p0 < P; p0 += h->chunk) {
        const int pc = std::min(h->chunk, P - p0);
        if (h->packForm) {
            if (p0 == 0 && !h->fwdPack) hipLaunchKernelGGL(acq_decimate10_kernel, dim3(40, (unsigned)B), dim3(640), 0, st, h->X_d, h->Y_d);
            const int items = pc * B, nBlk = std::min(h->cus, pc % 8 == 0 ? (items + 7) / 8 * 8 : items);
            hipLaunchKernelGGL(acq_corr25k_pack_kernel, dim3(nBlk), dim3(512), kPkLdsBytes, st, h->Y_d, h->Rcq_d + (size_t)p0 * h->len, h->tw2_d, h->tw25k_d,
                               B, pc, p0, h->surf_d, reinterpret_cast<unsigned int *>(h->mp_d), (pc % 8 == 0 && nBlk % 8 == 0) ? 1 : 0);
            continue;
        }
        if (h->fusedAlias) {
            hipLaunchKernelGGL(acq_radix10_kernel, dim3((M + 255) / 256, B, pc), dim3(256), 0, st, h->X_d, h->Rc_d + (size_t)p0 * h->len, h->tw25k_d,
                               B, M, h->Y_d);
            // one bin per block: 10 transforms each, B x pc blocks (25 x 32 = 800 at the reference's raster: one round of the chip)
            if (M == 4000) {
                constexpr size_t lds4 = AcqMixedShape<4000, 10, 8, 5>::ldsBytes;
                hipLaunchKernelGGL((acq_corr_mixed_kernel<4000, 10, 8, 5, true>), dim3(B, pc), dim3(400), lds4, st, h->Y_d, (const float2 *)nullptr, h->tw_d, B, h->N, 1, p0,
                                   h->surf_d, reinterpret_cast<unsigned int *>(h->mp_d));
            } else if (M == 5000) {
                constexpr size_t lds5 = AcqMixedShape<5000, 10, 10, 5>::ldsBytes;
                hipLaunchKernelGGL((acq_corr_mixed_kernel<5000, 10, 10, 5, true>), dim3(B, pc), dim3(500), lds5, st, h->Y_d, (const float2 *)nullptr, h->tw_d, B, h->N, 1, p0,
                                   h->surf_d, reinterpret_cast<unsigned int *>(h->mp_d));
            } else
                hipLaunchKernelGGL(acq_corr2500_kernel<true>, dim3(B, pc), dim3(256), 0, st, h->Y_d, (const float2 *)nullptr, h->tw_d, B, h->N, 1, p0,
                                   h->surf_d, reinterpret_cast<unsigned int *>(h->mp_d));
            continue;
        }
        hipLaunchKernelGGL(acq_mul_kernel, dim3((S + 1023) / 1024, B, pc), dim3(256), 0, st, h->X_d,
                           h->Rc_d + (size_t)p0 * h->len, S, h->len, B, h->Y_d);
        // a short last chunk still runs the full-batch plan over stale rows; they are never read
        if (h->planInv.exec(st, h->Y_d)) return -1;
        // mode 0 arrives already folded: one term, |.|
        hipLaunchKernelGGL(acq_fold_kernel, dim3((M + 255) / 256, (B + kAcqBinGroup - 1) / kAcqBinGroup, pc), dim3(256), 0, st, h->Y_d, S, M,
                           h->cfg.mode == 0 ? 1 : h->N, h->cfg.mode == 0 ? 1 : 0, B, h->surf_d + (size_t)p0 * B * M,
                           reinterpret_cast<unsigned int *>(h->mp_d) + (size_t)p0 * M);
    }
    {
        // percentile positions of _trim_mean(max_percode, 10): pos = (M - 1) q / 100, q = 5 and 95 (numpy.percentile)
        const double posLo = (double)(M - 1) * 5.0 / 100.0, posHi = (double)(M - 1) * 95.0 / 100.0;
        const int iLo = (int)std::floor(posLo), iHi = (int)std::floor(posHi);
        const int maskS = (int)std::ceil(h->cfg.samplingFrequency / kFCA);                      // :96-99
        // (the kernel's static LDS is ~17 KB: the row joins it only while the sum stays below the 64 KB a launch gets without
        //  an opt-in -- M <= 10 240 -- and is read from memory beyond)
        const int rowInLds = (size_t)M * sizeof(float) <= 40 * 1024 ? 1 : 0;
#ifdef DPE_EXPERIMENTS
        const bool statsLds = getenv("DPE_ACQ_STATS_LDS") && atoi(getenv("DPE_ACQ_STATS_LDS")) != 0;
#else
        constexpr bool statsLds = false;
#endif
        // (folding these statistics into the search kernel -- the last block of a PRN, a ticket per PRN, the maxima as 64-bit {value, ~bin}
        //  keys -- was built in round 6 and is slower: all blocks of a coherent search are resident at once, so every PRN's last block ends
        //  with the launch and nothing overlaps; 0.0544 -> 0.0584 ms coherent, 0.244 -> 0.2535 textbook.  With a device-scope fence per
        //  block instead of atomics-only hand-over: 0.131 / 0.684 -- an L2 write-back per block.  DESIGN_LOG.md A.15)
        if (M <= 256 * kAcqStatsR && !statsLds)
            hipLaunchKernelGGL(acq_stats_small_kernel, dim3(P), dim3(256), 0, st, h->surf_d, h->mp_d, B, M, maskS, iLo, posLo - (double)iLo, iHi,
                               posHi - (double)iHi, h->peakIdx_d, h->peakIdx_d + P, h->stats_hd);
        else
        hipLaunchKernelGGL(acq_stats_kernel, dim3(P), dim3(256), rowInLds ? (size_t)M * sizeof(float) : 0, st, h->surf_d, h->mp_d, B, M,
                           rowInLds, maskS, iLo, posLo - (double)iLo, iHi, posHi - (double)iHi, h->peakIdx_d, h->peakIdx_d + P,
                           h->stats_hd);   // (48 bytes per PRN straight into the pinned mirror: a D2H copy command behind the kernel cost ~5 us of stream time)
    }
    DPE_CHECK_HIP(hipGetLastError());
    h->searched = true;
    return 0;
}

int dpe_acq_results(dpe_acq *h, dpe_acq_result *out, dpe_stream_t stream)
{
    using namespace dpe;
    DPE_REQUIRE(h && out && h->searched, "[Acquisition] results: no search yet");
    DPE_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));   // the statistics kernel wrote stats_h itself
    const int P = h->P;
    const double fs = h->cfg.samplingFrequency;
    for (int p = 0; p < P; ++p) {
        const AcqStats &st = h->stats_h[p];
        const int ci = st.ci, di = st.di;                                          // first maxima, correlator.py:88-89
        dpe_acq_result &r = out[p];
        r.prn = h->cfg.prn[p];
        r.maxCodeIdx = ci; r.maxDoppIdx = di;
        r.rc = (double)kLCA - ((double)ci / fs) * kFCA;                            // :90
        r.fi = h->cfg.binStartHz + h->cfg.binStepHz * di;                          // :91
        r.fc = kFCA + (h->cfg.dopplerSign * kFCA / kFL1) * r.fi;                   // :92
        r.peak = st.peak;
        r.cppr = r.peak / (double)st.maxRest;                                           // :96-100 (mask: acq_stats_kernel)
        r.cppm = st.cnt ? r.peak / (st.sum / (double)st.cnt) : 0.0;               // _trim_mean :546-564
        r.found = r.cppm > 2.0 ? 1 : 0;                                            // :103
    }
    return 0;
}

static int fine_prepare(dpe_acq *h)
{
    using namespace dpe;
    if (h->haveFine) return 0;
    const int S = h->S, P = h->P;
    int bl = 0;
    for (int v = S; v; v >>= 1) ++bl;            // S.bit_length()
    const long long C = 8ll * (1ll << bl);       // rawfile.carr_fftpts, rawfile.py:173
    DPE_REQUIRE(C <= (1ll << 27), "[Acquisition] fine: FFT length %lld too large", C);
    h->fineC = (int)C;
    // shifted indices kept by the mask (:124-125): min(bins) <= fftfreq <= max(bins), fftfreq = k * (1 / (C * (1 / fs)))
    const double fs = h->cfg.samplingFrequency, val = 1.0 / ((double)C * (1.0 / fs));
    const double b0 = h->cfg.binStartHz, b1 = h->cfg.binStartHz + h->cfg.binStepHz * (h->B - 1);
    const double fmin = std::min(b0, b1), fmax = std::max(b0, b1);
    long long lo = (long long)std::floor(fmin / val) - 2, hi = (long long)std::ceil(fmax / val) + 2;
    while ((double)lo * val < fmin) ++lo;
    while ((double)hi * val > fmax) --hi;
    lo = std::max(lo, -C / 2); hi = std::min(hi, C / 2 - 1);
    DPE_REQUIRE(lo <= hi, "[Acquisition] fine: empty search range");
    h->fineLo = (int)(lo + C / 2); h->fineHi = (int)(hi + C / 2);
    h->F_d = dev_alloc<float2>((size_t)P * C);
    h->fineVal_d = dev_alloc<float2>(P);
    h->fineIdx_d = dev_alloc<int>(P);
    h->fineSums_d = dev_alloc<long long>(2);
    h->fineChan_d = dev_alloc<AcqFineChan>(P);
    h->chips_d = dev_alloc<int8_t>((size_t)P * 1024);
    DPE_REQUIRE(h->F_d && h->fineVal_d && h->fineIdx_d && h->fineSums_d && h->fineChan_d && h->chips_d,
                "[Acquisition] fine: device allocation failed");
    std::vector<int8_t> chips((size_t)P * 1024, 0);
    for (int p = 0; p < P; ++p) gen_ca_code_host(h->cfg.prn[p], chips.data() + (size_t)p * 1024);
    DPE_CHECK_HIP(hipMemcpy(h->chips_d, chips.data(), chips.size(), hipMemcpyHostToDevice));
    if (h->planFine.create((size_t)C, (size_t)P, false)) return -1;
    h->haveFine = true;
    return 0;
}

int dpe_acq_fine(dpe_acq *h, const int16_t *samples_dev, const dpe_acq_result *coarse, dpe_acq_fine_result *fine,
                 dpe_stream_t stream_)
{
    using namespace dpe;
    DPE_REQUIRE(h && samples_dev && coarse && fine, "[Acquisition] fine: null argument");
    if (fine_prepare(h)) return -1;
    hipStream_t st = (hipStream_t)stream_;
    const int S = h->S, P = h->P, C = h->fineC;
    std::vector<AcqFineChan> ch(P);
    for (int p = 0; p < P; ++p) {
        DPE_REQUIRE(coarse[p].fc > 0, "[Acquisition] fine: bad coarse code frequency (PRN %d)", h->cfg.prn[p]);
        ch[p].rc = coarse[p].rc; ch[p].fc = coarse[p].fc;
    }
    DPE_CHECK_HIP(hipMemcpyAsync(h->fineChan_d, ch.data(), sizeof(AcqFineChan) * P, hipMemcpyHostToDevice, st));
    DPE_CHECK_HIP(hipStreamSynchronize(st));   // ch is a stack-lifetime staging buffer
    DPE_CHECK_HIP(hipMemsetAsync(h->fineSums_d, 0, sizeof(long long) * 2, st));
    hipLaunchKernelGGL(acq_sum_kernel, dim3(32), dim3(256), 0, st, samples_dev, S, h->fineSums_d);
    hipLaunchKernelGGL(acq_fine_build_kernel, dim3((C + 1023) / 1024, P), dim3(256), 0, st, samples_dev, S, C,
                       h->cfg.samplingFrequency, h->fineChan_d, h->fineSums_d, h->chips_d, h->F_d);
    if (h->planFine.exec(st, h->F_d)) return -1;
    hipLaunchKernelGGL(acq_fine_peak_kernel, dim3(P), dim3(256), 0, st, h->F_d, C, h->fineLo, h->fineHi, h->fineIdx_d, h->fineVal_d);
    DPE_CHECK_HIP(hipGetLastError());
    std::vector<int> idx(P);
    std::vector<float2> val(P);
    DPE_CHECK_HIP(hipMemcpyAsync(idx.data(), h->fineIdx_d, sizeof(int) * P, hipMemcpyDeviceToHost, st));
    DPE_CHECK_HIP(hipMemcpyAsync(val.data(), h->fineVal_d, sizeof(float2) * P, hipMemcpyDeviceToHost, st));
    DPE_CHECK_HIP(hipStreamSynchronize(st));
    const double fs = h->cfg.samplingFrequency, fval = 1.0 / ((double)C * (1.0 / fs));
    for (int p = 0; p < P; ++p) {
        dpe_acq_fine_result &r = fine[p];
        r.prn = h->cfg.prn[p];
        r.maxCarrIdx = idx[p];
        r.peakRe = val[p].x; r.peakIm = val[p].y;
        r.rc = coarse[p].rc;                                                      // :133 (rc passes through)
        r.ri = std::atan2((double)val[p].y, (double)val[p].x) / (2.0 * kPi);      // :129
        r.fi = (double)(idx[p] - C / 2) * fval;                                   // :130
        r.fc = kFCA + (h->cfg.dopplerSign * kFCA / kFL1) * r.fi;                  // :131
    }
    return 0;
}

static double pos_mod(double v, double m)   // np.mod for m > 0
{
    const double t = std::fmod(v, m);
    return t < 0.0 ? t + m : t;
}

int dpe_acq_scalar_acquisition(dpe_acq *h, const int16_t *window0_dev, const int16_t *window1_dev,
                               dpe_acq_track_init *out, dpe_stream_t stream)
{
    using namespace dpe;
    DPE_REQUIRE(h && window0_dev && window1_dev && out, "[Acquisition] scalar_acquisition: null argument");
    const int P = h->P;
    const double T = (double)h->S / h->cfg.samplingFrequency;
    std::vector<dpe_acq_result> c0(P), c1(P);
    std::vector<dpe_acq_fine_result> f0(P), f1(P);
    if (dpe_acq_search(h, window0_dev, stream) || dpe_acq_results(h, c0.data(), stream) ||
        dpe_acq_fine(h, window0_dev, c0.data(), f0.data(), stream))
        return -1;
    if (dpe_acq_search(h, window1_dev, stream) || dpe_acq_results(h, c1.data(), stream) ||
        dpe_acq_fine(h, window1_dev, c1.data(), f1.data(), stream))
        return -1;
    for (int p = 0; p < P; ++p) {
        dpe_acq_track_init &r = out[p];
        r.prn = h->cfg.prn[p];
        r.fromSecondWindow = c1[p].cppm > c0[p].cppm ? 1 : 0;                     // receiver.py:493
        if (r.fromSecondWindow) {
            r.rc = pos_mod(f1[p].rc - f1[p].fc * T, (double)kLCA);               // :495-496: back to the first window's start
            r.ri = pos_mod(f1[p].ri - f1[p].fi * T, 1.0);
            r.fc = f1[p].fc; r.fi = f1[p].fi;
            r.found = c1[p].found; r.cppr = c1[p].cppr; r.cppm = c1[p].cppm;
        } else {
            r.rc = f0[p].rc; r.ri = f0[p].ri; r.fc = f0[p].fc; r.fi = f0[p].fi;  // :507
            r.found = c0[p].found; r.cppr = c0[p].cppr; r.cppm = c0[p].cppm;
        }
        r.cppmWindow[0] = c0[p].cppm; r.cppmWindow[1] = c1[p].cppm;
    }
    return 0;
}

int dpe_acq_surface(dpe_acq *h, const float **surface_dev, const float **maxPerCode_dev)
{
    DPE_REQUIRE(h, "[Acquisition] surface: null handle");
    if (surface_dev) *surface_dev = h->surf_d;
    if (maxPerCode_dev) *maxPerCode_dev = h->mp_d;
    return 0;
}

}  // extern "C"
