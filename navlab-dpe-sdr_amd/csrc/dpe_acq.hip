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
// create, and three small HIP kernels do wipe-off, spectrum multiply and fold + per-lag max over bins.
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
    __shared__ unsigned int hist[256];
    __shared__ unsigned int sTmp[4];
    __shared__ unsigned int sSel[2];
    __shared__ float sF[4];
    __shared__ double sD[4];
    __shared__ unsigned long long sB[4];
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
    for (int j = tid; j < M; j += 256) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(m[j]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)j);
        best = key > best ? key : best;
    }
    const int ci = block_argmax(best);
    best = 0ull;
    for (int b = tid; b < B; b += 256) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(surf[((size_t)p * B + b) * M + ci]) << 32) |
                                       (unsigned long long)(0xFFFFFFFFu - (unsigned)b);
        best = key > best ? key : best;
    }
    const int di = block_argmax(best);
    if (tid == 0) { codeIdx[p] = ci; doppIdx[p] = di; }
    __syncthreads();
    // row value with the +-maskS delays about the peak zeroed (indices wrap at both ends, see dpe_hip.h)
    auto val = [&](int j) -> float {
        int d = j - ci;
        if (d < 0) d += M;
        const int c = d < M - d ? d : M - d;
        return c <= maskS ? 0.f : m[j];
    };
    // k-th smallest (0-based) of the masked row
    auto select = [&](unsigned int k) -> float {
        unsigned int prefix = 0u, mask = 0u;
        for (int shift = 24; shift >= 0; shift -= 8) {
            hist[tid] = 0u;
            __syncthreads();
            for (int j = tid; j < M; j += 256) {
                const unsigned int u = __float_as_uint(val(j));
                if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
            }
            __syncthreads();
            // exclusive prefix over the 256 digit counts: the digit whose range holds rank k
            const unsigned int c = hist[tid];
            unsigned int inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned int o = __shfl_up(inc, off, 64);
                if ((tid & 63) >= off) inc += o;
            }
            if ((tid & 63) == 63) sTmp[tid >> 6] = inc;
            __syncthreads();
            unsigned int base = 0u;
            for (int q = 0; q < (tid >> 6); ++q) base += sTmp[q];
            const unsigned int excl = base + inc - c;
            if (c && k >= excl && k < excl + c) { sSel[0] = prefix | ((unsigned int)tid << shift); sSel[1] = k - excl; }
            __syncthreads();
            prefix = sSel[0]; k = sSel[1];
            mask |= 0xFFu << shift;
            __syncthreads();
        }
        return __uint_as_float(prefix);
    };
    // the order statistic after x = select(k): x itself if it occurs beyond rank k, else the smallest larger value
    auto next_after = [&](float x, unsigned int k) -> float {
        unsigned int le = 0u;
        float mn = 3.0e38f;
        for (int j = tid; j < M; j += 256) {
            const float v = val(j);
            le += v <= x ? 1u : 0u;
            mn = (v > x && v < mn) ? v : mn;
        }
        const unsigned int nLe = block_sum_u32(le, sTmp);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(mn, off, 64); mn = o < mn ? o : mn; }
        __syncthreads();
        if ((tid & 63) == 0) sF[tid >> 6] = mn;
        __syncthreads();
        mn = fminf(fminf(sF[0], sF[1]), fminf(sF[2], sF[3]));
        return nLe > k + 1u ? x : mn;
    };
    // masked maximum
    float mx = 0.f;
    for (int j = tid; j < M; j += 256) mx = fmaxf(mx, val(j));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((tid & 63) == 0) sF[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sF[0], sF[1]), fmaxf(sF[2], sF[3]));
    __syncthreads();
    // percentiles: lo + (hi - lo) * f with a float difference, as the host expression (and scipy) evaluate it
    double pLo, pHi;
    {
        const float a = select((unsigned int)iLo);
        pLo = (double)a;
        if (iLo + 1 < M) { const float b = next_after(a, (unsigned int)iLo); pLo = (double)a + (double)(b - a) * fLo; }
        const float c = select((unsigned int)iHi);
        pHi = (double)c;
        if (iHi + 1 < M) { const float d = next_after(c, (unsigned int)iHi); pHi = (double)c + (double)(d - c) * fHi; }
    }
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
    if (tid == 0) {
        AcqStats r;
        r.peak = m[ci]; r.maxRest = mx; r.ci = ci; r.di = di;
        r.sum = ((sD[0] + sD[1]) + sD[2]) + sD[3]; r.cnt = (long long)nCnt; r.lo = pLo; r.hi = pHi;
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
    dpe::AcqStats *stats_d = nullptr, *stats_h = nullptr;   // per-PRN peak statistics; pinned host copy
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
    void *bufs[] = {h->X_d, h->Rc_d, h->Y_d, h->surf_d, h->mp_d, h->peakIdx_d, h->F_d, h->fineVal_d, h->fineIdx_d, h->fineSums_d, h->fineChan_d, h->chips_d, h->stats_d};
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
    const size_t S = h->SX, B = h->B, P = h->P;
    h->X_d = dev_alloc<float2>(B * S);
    h->Rc_d = dev_alloc<float2>(P * (size_t)h->len);
    h->Y_d = dev_alloc<float2>((size_t)h->chunk * B * S);
    h->surf_d = dev_alloc<float>(P * B * (size_t)h->M);
    h->mp_d = dev_alloc<float>(P * (size_t)h->M);
    h->peakIdx_d = dev_alloc<int>(2 * P);
    h->stats_d = dev_alloc<AcqStats>(P);
    if (hipHostMalloc((void **)&h->stats_h, P * sizeof(AcqStats), hipHostMallocDefault) != hipSuccess) h->stats_h = nullptr;
    if (!h->X_d || !h->Rc_d || !h->Y_d || !h->surf_d || !h->mp_d || !h->peakIdx_d || !h->stats_d || !h->stats_h) {
        set_error("[Acquisition] create: device allocation failed");
        dpe_acq_destroy(h);
        return -1;
    }
    const int batchFwd = (int)(B * (S / h->len)), batchInv = (int)(h->chunk * B * (S / h->len));
    if (h->planFwd.create((size_t)h->len, (size_t)batchFwd, false) || h->planInv.create((size_t)h->len, (size_t)batchInv, true)) {
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
    const int rc = finish();
    pr.destroy();
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
    if (h->cfg.mode == 0)
        hipLaunchKernelGGL(acq_wipe_fold_kernel, dim3((M + 255) / 256, B), dim3(256), 0, st, samples_dev, M, h->N,
                           h->cfg.binStartHz, h->cfg.binStepHz, 1.0 / h->cfg.samplingFrequency, h->X_d, h->mp_d, (long long)P * M);
    else
        hipLaunchKernelGGL(acq_wipe_kernel, dim3((S + 1023) / 1024, B), dim3(256), 0, st, samples_dev, S, B, h->cfg.binStartHz,
                           h->cfg.binStepHz, 1.0 / h->cfg.samplingFrequency, h->X_d, h->mp_d, (long long)P * M);
    if (h->planFwd.exec(st, h->X_d)) return -1;
    for (int p0 = 0; p0 < P; p0 += h->chunk) {
        const int pc = std::min(h->chunk, P - p0);
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
        const int rowInLds = (size_t)M * sizeof(float) <= 64 * 1024 ? 1 : 0;
        hipLaunchKernelGGL(acq_stats_kernel, dim3(P), dim3(256), rowInLds ? (size_t)M * sizeof(float) : 0, st, h->surf_d, h->mp_d, B, M,
                           rowInLds, maskS, iLo, posLo - (double)iLo, iHi, posHi - (double)iHi, h->peakIdx_d, h->peakIdx_d + P,
                           h->stats_d);
        DPE_CHECK_HIP(hipMemcpyAsync(h->stats_h, h->stats_d, sizeof(AcqStats) * P, hipMemcpyDeviceToHost, st));
    }
    DPE_CHECK_HIP(hipGetLastError());
    h->searched = true;
    return 0;
}

int dpe_acq_results(dpe_acq *h, dpe_acq_result *out, dpe_stream_t stream)
{
    using namespace dpe;
    DPE_REQUIRE(h && out && h->searched, "[Acquisition] results: no search yet");
    DPE_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));   // the statistics were copied to stats_h behind the search
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
