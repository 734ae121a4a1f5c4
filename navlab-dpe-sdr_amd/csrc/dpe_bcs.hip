// dpe_bcs.hip -- BatchCorrScores for MI355X (gfx950): int16 I/Q -> windowed code-lag bank +
// windowed Doppler-bin bank per SV, for a batch of windows.
//
// Replaces the reference chain  BCS_Load -> thrust::reduce(mean) -> BCS_ComputeDopplerWipeoff
// -> BCS_ComputeCodeReplica -> 5x cufftExecZ2Z(S) -> BCS_ChooseCodeCorr -> fftshift ->
// BCS_SubtractDCOffset -> BCS_ChoosyBatchMultiplyAndPad -> cufftExecZ2Z(C) -> fftshift
// (cudarecv/modules/src/batchcorrscores.cu:1043-1180) by ONE streaming pass over the samples:
//
//   code bank   corr[l] = sum_n b[n] r[(n-l) mod S]         (== ifft(conj(fft r) fft b), :1099-1144)
//               for the lags l in [-L,+L] the manifold grid can reach, split at the nav-bit
//               boundary (X: replica index < idxNext, Y: >=) so that no-flip = X+Y, flip = X-Y;
//   carr bank   F[b] = sum_n c[n] exp(-j 2 pi n b / C)      (== zero-padded FFT bin b, :1179)
//               for bins b in [-B,+B], from per-256-sample power moments M_p = sum x^p c[n]
//               (x = n - block centre) and a degree-5 Taylor factor per block -- exact to
//               (2 pi 127.5 B / C)^6 / 720 (checked at create), ~15x fewer flops than direct.
//
// Data layout (HBM): samples int16 I,Q interleaved, read once per SV with 16-byte lane loads;
// chip table int8 [37][1024] -> LDS per block; replica +/-1 (nav-bit side masked) per wave
// sub-tile in LDS; per-block partial lag sums and per-sub-tile moments in global scratch,
// reduced in fixed order (no float atomics: bit-reproducible run to run).
#include <type_traits>

#include "dpe_common.h"
#include "dpe_prep.h"
#include "dpe_chm_dev.h"   // chm_k2: the device-resident channel manager's time update, carried as an extra block (FUSE form)

namespace dpe {

constexpr int kSub = 256;   // samples per wave sub-tile (4 per lane) == moment block
constexpr int kNMomMax = 6;  // power moments 0..5 at most; 4 when the bin window is narrow (chosen at create)
typedef float f2 __attribute__((ext_vector_type(2)));

// Complex product {a.x b.x - a.y b.y, a.x b.y + a.y b.x} in TWO packed instructions.  The compiler's own lowering of
// the same expression is pk_mul + two pk_fma (one per sign) + a v_mov to stitch the halves; the operand-select and
// lane-negate modifiers of v_pk_fma_f32 do it directly:  t = a.x * b;  r = {-a.y, a.y} * {b.y, b.x} + t.
// Inline asm is opaque to the compiler's hazard recogniser: operands that come straight from a transcendental
// (v_sin / v_cos need one wait state before a VALU consumer on gfx94x/95x) must go through wipe_seed() first.
__device__ __forceinline__ f2 cmul(f2 a, f2 b)
{
    f2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}

// conj(exp(j 2 pi f)) from the hardware sin/cos (argument in revolutions), safe to feed into cmul()
__device__ __forceinline__ f2 wipe_seed(float f)
{
    f2 w = f2{__builtin_amdgcn_cosf(f), -__builtin_amdgcn_sinf(f)};
    asm volatile("s_nop 1" : "+v"(w));   // covers the trans -> VALU forwarding hazard for the asm consumers
    return w;
}

// Time of sample n: n/fs, or (TABLE) the reference's ns-rounded table (BCS_GenTimeIdcs,
// batchcorrscores.cu:185-196) when the sampling period is not an integer number of nanoseconds.
template <bool TABLE>
__device__ __forceinline__ double code_phase(const BcsChanDev &ch, const double *tT, int m)
{
    if (TABLE) return fma(tT[m], ch.fc, ch.rc);
    return fma((double)m, ch.codeStep, ch.rc);
}
template <bool TABLE>
__device__ __forceinline__ double carr_phase(const BcsChanDev &ch, const double *tT, int n)
{
    if (TABLE) return fma(tT[n], ch.fi, ch.ri);
    return fma((double)n, ch.carrStep, ch.ri);
}

// TABLE mode: the wipe-off of sample nSeed + i is reached from the seed at nSeed by i uniform rotations exp(-j 2 pi fi / fs), but the
// reference's sample times are rounded to 1 ns (BCS_GenTimeIdcs :191-193), i.e. NOT uniformly spaced: the phase that is still owed,
// eps = 2 pi fi ((t[n] - t[nSeed]) - i / fs) <= 2 pi fi 1 ns, is applied as the first-order rotation w (1 - j eps).  For rates
// whose period is a half-integer number of ns (16 Msps: 62.5 ns) the rounding is systematic (every odd sample +0.5 ns) and the
// omitted term was a first-order error of pi fi 0.5 ns of the peak (8.6e-5 at fi = 60 kHz, 3.6e-6 at 2.5 kHz; found by the round-2
// fuzz sweep -- at 2.046 Msps the rounding is quasi-random and averaged out).
template <bool TABLE>
__device__ __forceinline__ f2 table_fix(f2 wv, const BcsChanDev &ch, const double *tT, int nSeed, int i, int S)
{
    if (!TABLE) return wv;
    int n = nSeed + i;
    n = n < S ? n : S - 1;
    const double d = fma(tT[n] - tT[nSeed < S ? nSeed : S - 1], ch.fi, -(double)(n - (nSeed < S ? nSeed : S - 1)) * ch.carrStep);
    const float eps = (float)(6.283185307179586476925286766559 * d);
    return f2{fmaf(eps, wv.y, wv.x), fmaf(-eps, wv.x, wv.y)};
}

// ------------------------------------------------------------------------------------------
// DC sum (thrust::reduce at batchcorrscores.cu:1065): exact int64 sums.  Each block writes its own slot
// sums[w][blockIdx.x][2] (no atomics, nothing to zero beforehand -- the whole Update stays free of memset
// nodes, which matters for the hipGraph replay); consumers add the <= kSumSlots slots (window_mean).
constexpr int kSumSlots = 64;
__device__ __forceinline__ void sum_body(const int16_t *__restrict__ iq, long long winStride, int S, long long *__restrict__ sums)
{
    const int w = blockIdx.y;
    const int *x = reinterpret_cast<const int *>(iq + (size_t)w * winStride * 2);
    int sI = 0, sQ = 0;  // each thread sums < 64k samples: no int32 overflow
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    if (((reinterpret_cast<uintptr_t>(x)) & 15) == 0) {
        const int4 *x4 = reinterpret_cast<const int4 *>(x);
        const int n4 = S >> 2;
        for (int n = gtid; n < n4; n += gsz) {
            const int4 v = x4[n];
            sI += (short)(v.x & 0xFFFF) + (short)(v.y & 0xFFFF) + (short)(v.z & 0xFFFF) + (short)(v.w & 0xFFFF);
            sQ += (v.x >> 16) + (v.y >> 16) + (v.z >> 16) + (v.w >> 16);
        }
        for (int n = (n4 << 2) + gtid; n < S; n += gsz) {
            const int v = x[n];
            sI += (short)(v & 0xFFFF);
            sQ += v >> 16;
        }
    } else {
        for (int n = gtid; n < S; n += gsz) {
            const int v = x[n];
            sI += (short)(v & 0xFFFF);
            sQ += v >> 16;
        }
    }
    long long tI = sI, tQ = sQ;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tI += __shfl_xor(tI, off, 64);
        tQ += __shfl_xor(tQ, off, 64);
    }
    __shared__ long long sW[4][2];
    if ((threadIdx.x & 63) == 0) { sW[threadIdx.x >> 6][0] = tI; sW[threadIdx.x >> 6][1] = tQ; }
    __syncthreads();
    if (threadIdx.x < 2) {
        long long *o = sums + ((size_t)w * kSumSlots + blockIdx.x) * 2;
        o[threadIdx.x] = sW[0][threadIdx.x] + sW[1][threadIdx.x] + sW[2][threadIdx.x] + sW[3][threadIdx.x];
    }
}

// (upDst / upSrc / upN16: the batch's channel parameters ride along -- the blocks copy the pinned staging block to the device,
//  one launch boundary less than a separate upload kernel; upN16 = 0: nothing to copy)
__global__ __launch_bounds__(256) void bcs_sum_kernel(const int16_t *__restrict__ iq, long long winStride, int S,
                                                      long long *__restrict__ sums, uint4 *__restrict__ upDst,
                                                      const uint4 *__restrict__ upSrc, int upN16)
{
    const int nThreads = gridDim.x * gridDim.y * 256;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < upN16; i += nThreads) upDst[i] = upSrc[i];
    sum_body(iq, winStride, S, sums);
}

// Closed-loop calls (one window, <= 37 channels) pass the channel parameters in the kernel-argument
// segment of the bank and finalize kernels (params_ptr, dpe_common.h): no H2D copy command, no staging
// buffer that must outlive the call.  pb must stay the FIRST argument of those kernels.
struct BcsParamBlock {
    BcsChanDev c[DPE_MAX_CHAN];
    // Re-run of a window whose dpe_bcs_set_dev_hint promise broke (the chip kernel's banks are then out of tolerance): the general
    // kernels are ALWAYS enqueued behind a hinted chip-kernel launch with guard = the handle's device status word, and their blocks
    // leave at once unless its bit 3 is set (written by bcs_prep_kernel / chm_k1 for this window).  nullptr: an ordinary launch.
    const int *guard;
};
__device__ __forceinline__ bool rerun_not_wanted(const BcsParamBlock &pb)
{
    return pb.guard != nullptr && (__hip_atomic_load(pb.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 8) == 0;
}

// DC mean = sum / (float)S in fp64 (batchcorrscores.cu:1065-1066,1210-1216), then fp32.  The <= 64 slots
// are fetched by one vector load (lane <-> slot) and added across the wave: a single memory latency,
// and integer sums are order-independent.
__device__ __forceinline__ void window_mean(const long long *__restrict__ sums, int w, int nSumBlk, int S, float &mRe, float &mIm)
{
    const int lane = threadIdx.x & 63;
    long long tI = 0, tQ = 0;
    if (lane < nSumBlk) {
        const longlong2 v = *reinterpret_cast<const longlong2 *>(sums + ((size_t)w * kSumSlots + lane) * 2);
        tI = v.x; tQ = v.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tI += __shfl_xor(tI, off, 64);
        tQ += __shfl_xor(tQ, off, 64);
    }
    mRe = (float)((double)tI / (double)(float)S);
    mIm = (float)((double)tQ / (double)(float)S);
}

// ------------------------------------------------------------------------------------------
// FUSE (single windows, closed loop): the DC sum rides along instead of running as a kernel of its own.  The mean only
// enters the Doppler moments, and linearly: M_p[(raw - mean) w r] = M_p[raw w r] - mean * M_p[w r].  So this kernel
// accumulates both moment sets and the k = 0 blocks also write the int64 sample sums of their tiles (one slot per
// block); bcs_finalize_kernel forms the mean from the slots and combines.  One launch and ~4 us less per window.
template <int LH, int kNMom, bool TABLE, bool FUSE, bool CO = false>
__global__ __launch_bounds__(256) void bcs_bank_kernel(BcsParamBlock pb, int inl, const int16_t *__restrict__ iq, long long winStride, int S,
                                                       int K, int nSub, int tilesPerBlock, int nBlk, int vecOK, int nSumBlk, int lagShift,
                                                       const BcsChanDev *__restrict__ chan,
                                                       const long long *__restrict__ sums,
                                                       const int8_t *__restrict__ chipTable,
                                                       const double *__restrict__ tT,
                                                       float2 *__restrict__ part, float2 *__restrict__ mom,
                                                       float2 *__restrict__ momRep, long long *__restrict__ sumSlots, ChmKArgs co)
{
    constexpr int NL = 2 * LH + 1;      // lags
    constexpr int NREP = kSub + 2 * LH;  // replica entries per sub-tile (with halo)
    constexpr int NRR = 4 + 2 * LH;      // entries one lane touches
    __shared__ float sChips[2048];   // chips as +/-1.0f, periodically extended: sChips[i] = chip[i mod 1023]
    __shared__ __align__(16) float sRep[4][NREP + 4];
    __shared__ float2 sAcc[4][NL];
    if (rerun_not_wanted(pb)) return;

    // Closed loop on the device (dpe_chm_dev_*): the channel manager's time update for THIS window's scan rides along as block
    // (0, 0, 0) of the launch -- it needs nothing stage 1 produces and stage 1 nothing of it -- beside the correlator blocks
    // instead of 10 us in front of them; the blocks of row x = 0 are then not correlator blocks.
    // (CO: an instantiation of its own -- the time update's registers and code stay out of the kernel every other caller runs)
    int blkX = blockIdx.x;
    if constexpr (CO) {
        static_assert(FUSE && !TABLE, "the co-block rides in the single-window form");
        if (blkX == 0) {
            if (blockIdx.y == 0 && blockIdx.z == 0) chm_k2(co);
            else if (blockIdx.y == 1 && blockIdx.z == 0) chm_k3(co);   // the NEXT time update's Kepler solutions, one window ahead
            return;
        }
        blkX -= 1;
    }
    const int blk = blkX, k = blockIdx.y, w = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    (void)pb;
    const BcsChanDev ch = params_ptr(chan, inl)[w * K + k];
    for (int i = tid; i < 2048; i += 256) sChips[i] = (float)chipTable[(ch.prn - 1) * 1024 + (i >= kLCA ? i - kLCA : i) % kLCA];
    const bool fastIdx = (double)NREP * ch.codeStep < 1000.0;   // chip span of one sub-tile fits the extended table
    float mRe = 0.f, mIm = 0.f;
    if (!FUSE) window_mean(sums, w, nSumBlk, S, mRe, mIm);
    int sumI = 0, sumQ = 0;   // FUSE, k == 0: exact sums of this lane's samples (< 64k samples per lane: no overflow)
    __shared__ int sSum[4][2];
    const int16_t *x = iq + (size_t)w * winStride * 2;
    float xp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xp[i] = (float)(4 * lane + i) - 127.5f;
    __syncthreads();

    for (int side = 0; side < 2; ++side) {
        f2 acc[NL];   // (re, im) pairs: packed fp32 FMAs against the real replica
#pragma unroll
        for (int j = 0; j < NL; ++j) acc[j] = f2{0.f, 0.f};

        for (int t = 0; t < tilesPerBlock; ++t) {
            const int sub = (blk * tilesPerBlock + t) * 4 + wave;
            const int sub0 = sub * kSub;
            // lagShift: this launch produces the lags [lagShift - LH, lagShift + LH] -- the same sums against the
            // replica delayed by lagShift samples, i.e. every replica index below is taken lagShift earlier
            const int lo = sub0 - LH - lagShift, hi = sub0 + kSub - 1 + LH - lagShift;
            bool active = sub < nSub;
            if (active) {
                if (!ch.hasFlip) active = (side == 0);
                else if (lo >= 0 && hi < S) active = (side == 0) ? (lo < ch.idxNext) : (hi >= ch.idxNext);
            }
            // samples first: the global-load latency is covered by the replica build below
            const int n0 = sub0 + 4 * lane;
            float re[4], im[4];
            const bool countRaw = FUSE && k == 0 && side == 0 && sub < nSub;   // every sample of the window exactly once
            if (active || countRaw) {
                int rawv[4];
                if (vecOK && n0 + 3 < S) {
                    const int4 v = *reinterpret_cast<const int4 *>(x + 2 * (size_t)n0);
                    rawv[0] = v.x; rawv[1] = v.y; rawv[2] = v.z; rawv[3] = v.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) rawv[i] = (n0 + i < S) ? *reinterpret_cast<const int *>(x + 2 * (size_t)(n0 + i)) : 0;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    re[i] = (float)(short)(rawv[i] & 0xFFFF);
                    im[i] = (float)(rawv[i] >> 16);
                    if (countRaw) { sumI += (short)(rawv[i] & 0xFFFF); sumQ += rawv[i] >> 16; }
                }
            }
            if (active) {
                // replica r[m] = chip[floor(t_m fc + rc) mod 1023] (BCS_ComputeCodeReplica :347-349),
                // masked to this side of the nav-bit boundary (:352-367), m wrapped circularly.
                if (fastIdx && lo >= 0 && hi < S) {
                    // common case (no circular wrap inside the sub-tile): the chip index relative to the
                    // sub-tile's first entry indexes the periodically extended table -- no modulo, no branch
                    const int ci0 = (int)floor(code_phase<TABLE>(ch, tT, lo));
                    const int shift = (ci0 % kLCA) - ci0;
                    const bool straddle = ch.hasFlip && lo < ch.idxNext && hi >= ch.idxNext;
                    for (int e = lane; e < NREP; e += 64) {
                        const int m = lo + e;
                        const int ci = (int)floor(code_phase<TABLE>(ch, tT, m)) + shift;
                        float r = sChips[ci];
                        if (straddle) r = ((m >= ch.idxNext) == (side == 1)) ? r : 0.f;
                        sRep[wave][e] = r;
                    }
                } else {
                    for (int e = lane; e < NREP; e += 64) {
                        int m = lo + e;
                        if (m < 0) m += S; else if (m >= S) m -= S;
                        const double cph = code_phase<TABLE>(ch, tT, m);
                        const int ci = ((int)floor(cph)) % kLCA;
                        const int sd = ch.hasFlip ? (m >= ch.idxNext) : 0;
                        sRep[wave][e] = (sd == side) ? sChips[ci] : 0.f;
                    }
                }
            }
            // no barrier: sRep[wave] is private to this wave and a wave's DS operations complete in order
            f2 M[kNMom], Mq[FUSE ? kNMom : 1];
#pragma unroll
            for (int p = 0; p < kNMom; ++p) M[p] = f2{0.f, 0.f};
#pragma unroll
            for (int p = 0; p < (FUSE ? kNMom : 1); ++p) Mq[p] = f2{0.f, 0.f};
            if (active) {
                // Doppler wipe-off conj(exp(j 2 pi (fi t + ri))) (BCS_ComputeDopplerWipeoff :294-300):
                // fp64 phase seed per lane, hardware sin/cos in revolutions, 3 fp32 rotations.
                double ph = carr_phase<TABLE>(ch, tT, n0 < S ? n0 : S - 1);   // lanes past the window carry zero samples
                ph -= floor(ph);
                const float f = (float)ph;
                f2 wv = wipe_seed(f);
                const f2 meanv = f2{mRe, mIm}, rotv = f2{ch.rotRe, ch.rotIm};
                float rr[NRR];
#pragma unroll
                for (int q = 0; q < NRR / 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(&sRep[wave][4 * lane + 4 * q]);
                    rr[4 * q] = v.x; rr[4 * q + 1] = v.y; rr[4 * q + 2] = v.z; rr[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // rawWiped = raw * wipe (BCS_BatchMultiply :402); complex products through cmul()
                    const f2 wrot = wv;                                       // the rotation chain stays uniform
                    wv = table_fix<TABLE>(wrot, ch, tT, n0, i, S);           // this sample's wipe-off (ns-rounded time table)
                    const f2 bb = cmul(f2{re[i], im[i]}, wv);
                    // sample n, lag l = j-LH uses replica index n-l -> rr[i - j + 2 LH]
#pragma unroll
                    for (int j = 0; j < NL; ++j) {
                        const float r = rr[i + 2 * LH - j];
                        acc[j] = __builtin_elementwise_fma(bb, f2{r, r}, acc[j]);
                    }
                    // carrier path: (raw - mean) * wipe * replica (:480, :440-448)
                    const float r0 = (n0 + i < S) ? rr[i + LH] : 0.f;  // no sample beyond the window
                    if (FUSE) {
                        f2 cp = bb * r0, cq = wv * r0;         // raw w r and w r: the mean is applied in the finalize kernel
#pragma unroll
                        for (int p = 0; p < kNMom; ++p) {
                            M[p] += cp; Mq[p] += cq;
                            cp *= xp[i]; cq *= xp[i];
                        }
                    } else {
                        f2 cp = (bb - cmul(meanv, wv)) * r0;   // x^p * c, built up by one packed multiply per order
#pragma unroll
                        for (int p = 0; p < kNMom; ++p) {
                            M[p] += cp;
                            cp *= xp[i];
                        }
                    }
                    wv = cmul(wrot, rotv);
                }
            }
            if (sub < nSub && lagShift == 0) {   // the Doppler path belongs to the unshifted replica only
                const size_t mo = ((((size_t)w * K + k) * 2 + side) * nSub + sub) * kNMom;
                float2 *o = mom + mo;
                if (active) {
                    float mm[2 * kNMom];
#pragma unroll
                    for (int p = 0; p < kNMom; ++p) { mm[2 * p] = M[p].x; mm[2 * p + 1] = M[p].y; }
                    dpp_sum_lane63(mm);
                    if (lane == 63) {
#pragma unroll
                        for (int p = 0; p < kNMom; ++p) o[p] = make_float2(mm[2 * p], mm[2 * p + 1]);
                    }
                    if (FUSE) {
#pragma unroll
                        for (int p = 0; p < kNMom; ++p) { mm[2 * p] = Mq[p].x; mm[2 * p + 1] = Mq[p].y; }
                        dpp_sum_lane63(mm);
                        if (lane == 63) {
#pragma unroll
                            for (int p = 0; p < kNMom; ++p) momRep[mo + p] = make_float2(mm[2 * p], mm[2 * p + 1]);
                        }
                    }
                } else if (lane < kNMom) {
                    o[lane] = make_float2(0.f, 0.f);
                    if (FUSE) momRep[mo + lane] = make_float2(0.f, 0.f);
                }
            }
        }
        // block partial of the lag sums, fixed reduction order
        {
            float aa[2 * NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) { aa[2 * j] = acc[j].x; aa[2 * j + 1] = acc[j].y; }
            dpp_sum_lane63(aa);
            if (lane == 63) {
#pragma unroll
                for (int j = 0; j < NL; ++j) sAcc[wave][j] = make_float2(aa[2 * j], aa[2 * j + 1]);
            }
        }
        if (FUSE && k == 0 && side == 0) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                sumI += __shfl_xor(sumI, off, 64);   // per-wave totals stay far below 2^31 (<= 16 tiles x 256 x 32767)
                sumQ += __shfl_xor(sumQ, off, 64);
            }
            if (lane == 0) { sSum[wave][0] = sumI; sSum[wave][1] = sumQ; }
        }
        __syncthreads();
        if (FUSE && k == 0 && side == 0 && tid < 2)
            sumSlots[((size_t)w * kSumSlots + blk) * 2 + tid] =
                (long long)sSum[0][tid] + (long long)sSum[1][tid] + (long long)sSum[2][tid] + (long long)sSum[3][tid];
        for (int j = tid; j < NL; j += 256) {
            float2 s = sAcc[0][j];
            s.x += sAcc[1][j].x; s.y += sAcc[1][j].y;
            s.x += sAcc[2][j].x; s.y += sAcc[2][j].y;
            s.x += sAcc[3][j].x; s.y += sAcc[3][j].y;
            part[((((size_t)w * K + k) * nBlk + blk) * 2 + side) * NL + j] = s;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Batch variant of bcs_bank_kernel for narrow lag windows (LH <= 8): one wave pass covers 1024 consecutive
// samples, each lane 16 of them, so that a DPP row (16 lanes) is exactly one 256-sample moment sub-tile.
// The arithmetic per sample is the dense kernel's; what changes is the amortisation -- index arithmetic,
// carrier-phase seeds (two per lane, so a rotation chain is never longer than 7 steps), side bookkeeping and the
// cross-lane moment sums (4 row-local DPP steps per 4 sub-tiles instead of 6 wave steps per sub-tile) are paid
// once per 16 samples instead of once per 4.  Used when the batch offers enough passes to fill the chip
// (dpe_bcs_update); single windows keep the 4-samples-per-lane kernel, whose blocks are 4x shorter.
// Same partial / moment layouts, so bcs_finalize_kernel is shared.
constexpr int kB16Blocks = 4;   // blocks per CU the narrow, table-free form is held to (128 registers; five -> 96 with spills measured 1 % slower)
template <int LH, int kNMom, bool TABLE>
__global__ __launch_bounds__(256, (LH <= 4 && !TABLE) ? kB16Blocks : 3) void bcs_bank16_kernel(BcsParamBlock pb, int inl, const int16_t *__restrict__ iq, long long winStride,
                                                         int S, int K, int nW, int nSub, int tilesPerBlock, int nBlk, int vecOK, int nSumBlk,
                                                         const BcsChanDev *__restrict__ chan,
                                                         const long long *__restrict__ sums,
                                                         const int8_t *__restrict__ chipTable,
                                                         const double *__restrict__ tT,
                                                         float2 *__restrict__ part, float2 *__restrict__ mom)
{
    constexpr int NL = 2 * LH + 1;       // lags
    constexpr int kPass = 1024;          // samples per wave pass
    constexpr int NREP = kPass + 2 * LH;  // replica entries per pass (with halo)
    constexpr int NRR = 16 + 2 * LH;     // entries one lane touches
    static_assert((8 + 2 * LH) % 4 == 0 && LH <= 8, "a segment's replica window is read as float4s");
    __shared__ float sChips[2048];   // chips as +/-1.0f, periodically extended: sChips[i] = chip[i mod 1023]
#ifdef DPE_B16_PAD   // experiment (round 3, measured and dropped -- DESIGN.md 9): every 16 replica entries take 20 floats -- a lane's float4 window reads then fall on 16 distinct 4-bank slots
    constexpr int kRepLen = ((NREP + 8 + 15) / 16) * 20;
#define DPE_PAD16(e) ((e) + 4 * ((e) >> 4))
#else
    constexpr int kRepLen = NREP + 8;
#define DPE_PAD16(e) (e)
#endif
    __shared__ __align__(16) float sRep[4][kRepLen];
    __shared__ float2 sAcc[4][NL];

    // Block -> (window, tile group, SV), XCD-aware like the chip kernel's: the K blocks that read the same samples get linear
    // ids that are congruent mod 8 (same XCD) and consecutive there, so the XCD's L2 serves K - 1 of the K reads
    const int slot = blockIdx.x >> 3, k = slot % K, tg = (slot / K) * 8 + (blockIdx.x & 7);
    if (tg >= nBlk * nW) return;
    const int w = tg / nBlk, blk = tg - w * nBlk;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    (void)pb;
    const BcsChanDev ch = params_ptr(chan, inl)[w * K + k];
    for (int i = tid; i < 2048; i += 256) sChips[i] = (float)chipTable[(ch.prn - 1) * 1024 + (i >= kLCA ? i - kLCA : i) % kLCA];
    const bool fastIdx = (double)NREP * ch.codeStep < 1000.0;   // chip span of one pass fits the extended table
    float mRe, mIm;
    window_mean(sums, w, nSumBlk, S, mRe, mIm);
    const f2 meanv = f2{mRe, mIm}, rotv = f2{ch.rotRe, ch.rotIm};
    const int16_t *x = iq + (size_t)w * winStride * 2;
    const int padLane = DPE_PAD16(lane);   // (64 j + lane pads as pad(64 j) + pad(lane): 64 j is a multiple of 16)
    (void)padLane;
    const float xbase = (float)(16 * (lane & 15)) - 127.5f;      // moment abscissa of the lane's first sample
    __syncthreads();

    for (int side = 0; side < 2; ++side) {
        f2 acc[NL];   // (re, im) pairs: packed fp32 FMAs against the real replica
#pragma unroll
        for (int j = 0; j < NL; ++j) acc[j] = f2{0.f, 0.f};
        bool anyActive = false;   // wave-uniform: most blocks lie entirely on one side of the nav-bit boundary

        for (int t = 0; t < tilesPerBlock; ++t) {
            const int pass = (blk * tilesPerBlock + t) * 4 + wave;
            const int c0 = pass * kPass;
            const int lo = c0 - LH, hi = c0 + kPass - 1 + LH;
            bool active = c0 < S;
            if (active) {
                if (!ch.hasFlip) active = (side == 0);
                else if (lo >= 0 && hi < S) active = (side == 0) ? (lo < ch.idxNext) : (hi >= ch.idxNext);
            }
            anyActive |= active;
            const int n0 = c0 + 16 * lane;
            if (active) {
                // replica r[m] = chip[floor(t_m fc + rc) mod 1023] (BCS_ComputeCodeReplica :347-349), masked to this
                // side of the nav-bit boundary (:352-367), m wrapped circularly -- as in bcs_bank_kernel
                if (fastIdx && lo >= 0 && hi < S) {
                    // (phases are >= 0 here -- rc >= 0, lo >= 0 -- so the truncating conversion IS the floor)
                    const int ci0 = (int)code_phase<TABLE>(ch, tT, lo);
                    const int shift = (ci0 % kLCA) - ci0;
                    const bool straddle = ch.hasFlip && lo < ch.idxNext && hi >= ch.idxNext;
                    if (!straddle) {   // (two loops: the side mask of the one straddling pass costs 3 of 17 slots per entry)
                        // batches of eight independent entries: the fp64 phase -> table read -> store chains overlap
                        // instead of running one after the other (the loop is latency, not issue)
                        static_assert(NREP >= 1024 && NREP < 1024 + 64, "sixteen full rounds of 64 entries and a partial one");
                        if constexpr (TABLE) {   // (the time-table form reads its sample times from memory: keep the short live range)
                            for (int e = lane; e < NREP; e += 64) sRep[wave][DPE_PAD16(e)] = sChips[(int)code_phase<TABLE>(ch, tT, lo + e) + shift];
                        } else {
#pragma unroll
                        for (int e0 = 0; e0 < 1024; e0 += 512) {
                            float v[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = sChips[(int)code_phase<TABLE>(ch, tT, lo + e0 + 64 * j + lane) + shift];
#pragma unroll
                            for (int j = 0; j < 8; ++j) sRep[wave][DPE_PAD16(e0 + 64 * j) + padLane] = v[j];
                        }
                        if (1024 + lane < NREP) sRep[wave][DPE_PAD16(1024) + padLane] = sChips[(int)code_phase<TABLE>(ch, tT, lo + 1024 + lane) + shift];
                        }
                    } else {
                        for (int e = lane; e < NREP; e += 64) {
                            const int m = lo + e;
                            const float r = sChips[(int)code_phase<TABLE>(ch, tT, m) + shift];
                            sRep[wave][DPE_PAD16(e)] = ((m >= ch.idxNext) == (side == 1)) ? r : 0.f;
                        }
                    }
                } else {
                    for (int e = lane; e < NREP; e += 64) {
                        int m = lo + e;
                        if (m < 0) m += S; else if (m >= S) m -= S;
                        const double cph = code_phase<TABLE>(ch, tT, m);
                        const int ci = ((int)floor(cph)) % kLCA;
                        const int sd = ch.hasFlip ? (m >= ch.idxNext) : 0;
                        sRep[wave][DPE_PAD16(e)] = (sd == side) ? sChips[ci] : 0.f;
                    }
                }
            }
            // no barrier: sRep[wave] is private to this wave and a wave's DS operations complete in order
            f2 M[kNMom];
#pragma unroll
            for (int p = 0; p < kNMom; ++p) M[p] = f2{0.f, 0.f};
            // two segments of 8 samples, deliberately NOT unrolled: the live set stays at one segment's samples and replica
            // window (occupancy), and each segment re-seeds the carrier phase.  allIn: the whole pass lies inside the window
            // (every pass but the window's last) -- no per-sample bound on the carrier path
            auto segments = [&](auto allInTag) {
                constexpr bool kAllIn = decltype(allInTag)::value;
#pragma unroll 1
                for (int seg = 0; seg < 2; ++seg) {
                    const int ns = n0 + 8 * seg;
                    int raw[8];   // packed I/Q
                    // (fetching a segment ahead -- under the replica build / the previous segment -- measured 5 % slower: the
                    // eight extra live registers cost more than the exposed load latency, which the other waves cover)
                    if (vecOK && (kAllIn || ns + 7 < S)) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int4 v = *reinterpret_cast<const int4 *>(x + 2 * (size_t)(ns + 4 * q));
                            raw[4 * q] = v.x; raw[4 * q + 1] = v.y; raw[4 * q + 2] = v.z; raw[4 * q + 3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) raw[i] = (ns + i < S) ? *reinterpret_cast<const int *>(x + 2 * (size_t)(ns + i)) : 0;
                    }
                    float rr[8 + 2 * LH];
#pragma unroll
                    for (int q = 0; q < (8 + 2 * LH) / 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4 *>(&sRep[wave][DPE_PAD16(16 * lane + 8 * seg + 4 * q)]);
                        rr[4 * q] = v.x; rr[4 * q + 1] = v.y; rr[4 * q + 2] = v.z; rr[4 * q + 3] = v.w;
                    }
                    // Doppler wipe-off conj(exp(j 2 pi (fi t + ri))) (BCS_ComputeDopplerWipeoff :294-300): fp64 phase
                    // seed, hardware sin/cos in revolutions, then 7 fp32 rotations -- complex products through cmul()
                    double ph = carr_phase<TABLE>(ch, tT, (kAllIn || ns < S) ? ns : S - 1);   // lanes past the window carry zero samples
                    ph -= floor(ph);
                    f2 wv = wipe_seed((float)ph);
                    const bool inside = kAllIn || ns + 7 < S;   // false only for lanes of the window's last pass
                    // moments of the segment about ITS centre (the powers of i - 3.5 are literals: one packed FMA per order
                    // and sample), moved to the sub-tile's abscissa once per segment
                    f2 m[kNMom];
#pragma unroll
                    for (int p = 0; p < kNMom; ++p) m[p] = f2{0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const f2 rawv = f2{(float)(short)(raw[i] & 0xFFFF), (float)(raw[i] >> 16)};
                        const f2 wrot = wv;                                // the rotation chain stays uniform
                        wv = table_fix<TABLE>(wrot, ch, tT, ns, i, S);    // this sample's wipe-off (ns-rounded time table)
                        const f2 bb = cmul(rawv, wv);                      // rawWiped = raw * wipe (BCS_BatchMultiply :402)
                        // sample n, lag l = j-LH uses replica index n-l -> rr[i - j + 2 LH]
#pragma unroll
                        for (int j = 0; j < NL; ++j) {
                            const float r = rr[i + 2 * LH - j];
                            acc[j] = __builtin_elementwise_fma(bb, f2{r, r}, acc[j]);
                        }
                        // carrier path: (raw - mean) * wipe * replica (:480, :440-448)
                        const float r0 = (inside || ns + i < S) ? rr[i + LH] : 0.f;  // no sample beyond the window
                        const f2 cp = (bb - cmul(meanv, wv)) * r0;
                        const float d = (float)i - 3.5f;
                        float dp = 1.f;
                        m[0] += cp;
#pragma unroll
                        for (int p = 1; p < kNMom; ++p) {
                            dp *= d;   // folded at compile time
                            m[p] = __builtin_elementwise_fma(cp, f2{dp, dp}, m[p]);
                        }
                        wv = cmul(wrot, rotv);
                    }
                    // (x + d)^p = sum_q C(p,q) x^(p-q) d^q, x = the segment centre in sub-tile coordinates
                    const float xc = xbase + (float)(8 * seg) + 3.5f;
                    float xpow[kNMom];
                    xpow[0] = 1.f;
#pragma unroll
                    for (int e = 1; e < kNMom; ++e) xpow[e] = xpow[e - 1] * xc;
                    constexpr float bin[6][6] = {{1, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0}, {1, 3, 3, 1, 0, 0}, {1, 4, 6, 4, 1, 0}, {1, 5, 10, 10, 5, 1}};
#pragma unroll
                    for (int p = 0; p < kNMom; ++p) {
                        M[p] += m[p];
#pragma unroll
                        for (int q = 0; q < p; ++q) {
                            const float cf = bin[p][q] * xpow[p - q];
                            M[p] = __builtin_elementwise_fma(m[q], f2{cf, cf}, M[p]);
                        }
                    }
                }
            };
            if (active) {
                if (c0 + kPass <= S) segments(std::true_type{});
                else segments(std::false_type{});
            }
            {
                // one moment set per DPP row = per 256-sample sub-tile
                const int sub = pass * 4 + (lane >> 4);
                float2 *o = mom + ((((size_t)w * K + k) * 2 + side) * nSub + sub) * kNMom;
                if (active) {
                    float mm[2 * kNMom];
#pragma unroll
                    for (int p = 0; p < kNMom; ++p) { mm[2 * p] = M[p].x; mm[2 * p + 1] = M[p].y; }
                    dpp_sum_rows(mm);
                    if (sub < nSub && (lane & 15) == 15) {
#pragma unroll
                        for (int p = 0; p < kNMom; ++p) o[p] = make_float2(mm[2 * p], mm[2 * p + 1]);
                    }
                } else if (sub < nSub && (lane & 15) < kNMom) {
                    o[lane & 15] = make_float2(0.f, 0.f);
                }
            }
        }
        // block partial of the lag sums, fixed reduction order (a wave without a pass on this side contributes zeros)
        if (anyActive) {
            float aa[2 * NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) { aa[2 * j] = acc[j].x; aa[2 * j + 1] = acc[j].y; }
            dpp_sum_lane63(aa);
            if (lane == 63) {
#pragma unroll
                for (int j = 0; j < NL; ++j) sAcc[wave][j] = make_float2(aa[2 * j], aa[2 * j + 1]);
            }
        } else if (lane < NL) {
            sAcc[wave][lane] = make_float2(0.f, 0.f);
        }
        __syncthreads();
        for (int j = tid; j < NL; j += 256) {
            float2 s = sAcc[0][j];
            s.x += sAcc[1][j].x; s.y += sAcc[1][j].y;
            s.x += sAcc[2][j].x; s.y += sAcc[2][j].y;
            s.x += sAcc[3][j].x; s.y += sAcc[3][j].y;
            part[((((size_t)w * K + k) * nBlk + blk) * 2 + side) * NL + j] = s;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
#undef DPE_PAD16

// Wide lag windows (|lag| <= 32, i.e. high sampling rates where a chip spans many samples): instead of
// 65 dense multiply-accumulates per sample, use that the replica is piecewise constant:
//     corr[l+1] - corr[l] = sum_m J[m] b[(m + l) mod S],   J[m] = r[m-1] - r[m]
// is a sum over the chip (and nav-bit / mask) boundaries only.  Per sub-tile: one direct lag (l = -32,
// one FMA pair per sample) plus, for each of the ~codeStep*128 boundaries of the tile, one LDS read and
// one FMA per LANE with lanes <-> the 64 lag steps.  The finalize kernel prefix-sums the steps.
// part layout per (block, side): [0] = corr[-32], [1 + i] = D[-32 + i], i = 0..63  (65 entries, as NL).
template <int kNMom, bool TABLE>
__global__ __launch_bounds__(256) void bcs_bank_wide_kernel(BcsParamBlock pb, int inl, const int16_t *__restrict__ iq, long long winStride, int S,
                                                            int K, int nSub, int tilesPerBlock, int nBlk, int vecOK, int nSumBlk, int lagShift,
                                                            const BcsChanDev *__restrict__ chan,
                                                            const long long *__restrict__ sums,
                                                            const int8_t *__restrict__ chipTable,
                                                            const double *__restrict__ tT,
                                                            float2 *__restrict__ part, float2 *__restrict__ mom)
{
    constexpr int LH = 32, NL = 2 * LH + 1;
    constexpr int NREP = kSub + LH + 1;   // replica entries: m = sub0-1 .. sub0+255+LH
    constexpr int NB = kSub + 2 * LH;     // wiped samples:   n = sub0-LH .. sub0+255+LH
    __shared__ float sChips[2048];
    __shared__ float sRep[4][NREP + 3];
    __shared__ float2 sB[4][NB];
    __shared__ float2 sAcc[4][NL];
    if (rerun_not_wanted(pb)) return;

    const int blk = blockIdx.x, k = blockIdx.y, w = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    (void)pb;
    const BcsChanDev ch = params_ptr(chan, inl)[w * K + k];
    for (int i = tid; i < 2048; i += 256) sChips[i] = (float)chipTable[(ch.prn - 1) * 1024 + (i >= kLCA ? i - kLCA : i) % kLCA];
    const bool fastIdx = (double)NREP * ch.codeStep < 1000.0;
    float mRe, mIm;
    window_mean(sums, w, nSumBlk, S, mRe, mIm);
    const int16_t *x = iq + (size_t)w * winStride * 2;
    float xp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xp[i] = (float)(4 * lane + i) - 127.5f;
    __syncthreads();

    for (int side = 0; side < 2; ++side) {
        f2 acc0 = f2{0.f, 0.f};   // direct sum at lag -LH (per-lane partial)
        f2 D = f2{0.f, 0.f};      // this lane's lag step  l = lane - LH
        for (int t = 0; t < tilesPerBlock; ++t) {
            const int sub = (blk * tilesPerBlock + t) * 4 + wave;
            const int sub0 = sub * kSub;
            const int lo = sub0 - 1 - lagShift, hi = sub0 + kSub - 1 + LH - lagShift;   // replica index range (see lagShift in bcs_bank_kernel)
            bool active = sub < nSub;
            if (active) {
                if (!ch.hasFlip) active = (side == 0);
                else if (lo >= 0 && hi < S) active = (side == 0) ? (lo < ch.idxNext) : (hi >= ch.idxNext);
            }
            f2 M[kNMom];
#pragma unroll
            for (int p = 0; p < kNMom; ++p) M[p] = f2{0.f, 0.f};
            if (active) {
                // ---- masked replica, entries e <-> m = lo + e (circular)
                if (fastIdx && lo >= 0 && hi < S) {
                    const int ci0 = (int)floor(code_phase<TABLE>(ch, tT, lo));
                    const int shift = (ci0 % kLCA) - ci0;
                    const bool straddle = ch.hasFlip && lo < ch.idxNext && hi >= ch.idxNext;
                    for (int e = lane; e < NREP; e += 64) {
                        const int m = lo + e;
                        float r = sChips[(int)floor(code_phase<TABLE>(ch, tT, m)) + shift];
                        if (straddle) r = ((m >= ch.idxNext) == (side == 1)) ? r : 0.f;
                        sRep[wave][e] = r;
                    }
                } else {
                    for (int e = lane; e < NREP; e += 64) {
                        int m = lo + e;
                        if (m < 0) m += S; else if (m >= S) m -= S;
                        const int ci = ((int)floor(code_phase<TABLE>(ch, tT, m))) % kLCA;
                        const int sd = ch.hasFlip ? (m >= ch.idxNext) : 0;
                        sRep[wave][e] = (sd == side) ? sChips[ci] : 0.f;
                    }
                }
                // ---- wiped samples b[n] (circular), own 4 + one halo sample per lane -> LDS
                const int n0 = sub0 + 4 * lane;
                float re[4], im[4];
                if (vecOK && n0 + 3 < S) {
                    const int4 v = *reinterpret_cast<const int4 *>(x + 2 * (size_t)n0);
                    re[0] = (float)(short)(v.x & 0xFFFF); im[0] = (float)(v.x >> 16);
                    re[1] = (float)(short)(v.y & 0xFFFF); im[1] = (float)(v.y >> 16);
                    re[2] = (float)(short)(v.z & 0xFFFF); im[2] = (float)(v.z >> 16);
                    re[3] = (float)(short)(v.w & 0xFFFF); im[3] = (float)(v.w >> 16);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int nn = n0 + i;
                        if (nn >= S) nn -= S;   // beyond the window: the circular continuation (masked below where it is not a sample)
                        const int v = *reinterpret_cast<const int *>(x + 2 * (size_t)nn);
                        re[i] = (float)(short)(v & 0xFFFF);
                        im[i] = (float)(v >> 16);
                    }
                }
                f2 bown[4], wown[4];
                {
                    int nn = n0 >= S ? n0 - S : n0;
                    double ph = carr_phase<TABLE>(ch, tT, nn);
                    ph -= floor(ph);
                    const float f = (float)ph;
                    f2 wv = wipe_seed(f);
                    const f2 rotv = f2{ch.rotRe, ch.rotIm};
                    int seedIdx = nn, steps = 0;                               // the chain's seed sample and the rotations since
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (n0 + i == S) {   // phase restarts where the circular continuation begins
                            double p2 = ch.ri - floor(ch.ri);
                            wv = wipe_seed((float)p2);
                            seedIdx = 0; steps = 0;
                        }
                        const f2 wrot = wv;                                   // the rotation chain stays uniform
                        wv = table_fix<TABLE>(wrot, ch, tT, seedIdx, steps, S);   // this sample's wipe-off (ns-rounded time table)
                        ++steps;
                        wown[i] = wv;
                        bown[i] = cmul(f2{re[i], im[i]}, wv);
                        sB[wave][LH + 4 * lane + i] = make_float2(bown[i].x, bown[i].y);
                        wv = cmul(wrot, rotv);
                    }
                }
                {   // halo: lanes 0..31 -> n = sub0-LH+lane ; lanes 32..63 -> n = sub0+256+(lane-32)
                    int nh = (lane < LH) ? (sub0 - LH + lane) : (sub0 + kSub + lane - LH);
                    const int e = (lane < LH) ? lane : (kSub + lane);
                    if (nh < 0) nh += S; else if (nh >= S) nh -= S;
                    const int v = *reinterpret_cast<const int *>(x + 2 * (size_t)nh);
                    const float hr = (float)(short)(v & 0xFFFF), hi2 = (float)(v >> 16);
                    double ph = carr_phase<TABLE>(ch, tT, nh);
                    ph -= floor(ph);
                    const float f = (float)ph;
                    const float wr = __builtin_amdgcn_cosf(f), wi = -__builtin_amdgcn_sinf(f);
                    sB[wave][e] = make_float2(hr * wr - hi2 * wi, hr * wi + hi2 * wr);
                }
                // ---- own samples: direct lag -LH, carrier moments (only real samples n < S)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool real = n0 + i < S;
                    const float rl = real ? sRep[wave][4 * lane + i + 1 + LH] : 0.f;   // r[n + LH]
                    acc0 = __builtin_elementwise_fma(bown[i], f2{rl, rl}, acc0);
                    const float r0 = real ? sRep[wave][4 * lane + i + 1] : 0.f;        // r[n]
                    f2 cp = (bown[i] - cmul(f2{mRe, mIm}, wown[i])) * r0;
#pragma unroll
                    for (int p = 0; p < kNMom; ++p) {
                        M[p] += cp;
                        cp *= xp[i];
                    }
                }
                // ---- boundaries m = sub0 + 4*lane + i (only m < S exist): J = r[m-1] - r[m]
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * lane + i;   // rep[e] = r[m-1], rep[e+1] = r[m]
                    const float J = (sub0 + e < S) ? (sRep[wave][e] - sRep[wave][e + 1]) : 0.f;
                    unsigned long long mask = __ballot(J != 0.f);
                    while (mask) {
                        const int src = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        const float Jv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, J), src));
                        // b index for lag l = lane - LH:  n = m + l  ->  entry (m - sub0 + LH) + (lane - LH)
                        const float2 bv = sB[wave][4 * src + i + lane];
                        D = __builtin_elementwise_fma(f2{bv.x, bv.y}, f2{Jv, Jv}, D);
                    }
                }
            }
            if (sub < nSub && lagShift == 0) {
                float2 *o = mom + ((((size_t)w * K + k) * 2 + side) * nSub + sub) * kNMom;
                if (active) {
                    float mm[2 * kNMom];
#pragma unroll
                    for (int p = 0; p < kNMom; ++p) { mm[2 * p] = M[p].x; mm[2 * p + 1] = M[p].y; }
                    dpp_sum_lane63(mm);
                    if (lane == 63) {
#pragma unroll
                        for (int p = 0; p < kNMom; ++p) o[p] = make_float2(mm[2 * p], mm[2 * p + 1]);
                    }
                } else if (lane < kNMom) {
                    o[lane] = make_float2(0.f, 0.f);
                }
            }
        }
        {
            float aa[3] = {acc0.x, acc0.y, 0.f};
            dpp_sum_lane63(aa);
            if (lane == 63) sAcc[wave][0] = make_float2(aa[0], aa[1]);
            sAcc[wave][1 + lane] = make_float2(D.x, D.y);
        }
        __syncthreads();
        for (int j = tid; j < NL; j += 256) {
            float2 sum = sAcc[0][j];
            sum.x += sAcc[1][j].x; sum.y += sAcc[1][j].y;
            sum.x += sAcc[2][j].x; sum.y += sAcc[2][j].y;
            sum.x += sAcc[3][j].x; sum.y += sAcc[3][j].y;
            part[((((size_t)w * K + k) * nBlk + blk) * 2 + side) * NL + j] = sum;
        }
        __syncthreads();
    }
}

}  // namespace dpe

#include "dpe_bcs_chip.h"   // chip-boundary form of stage 1 (high sampling rates)
#include "dpe_bcs_chip2.h"  // second form: lanes <-> chips in the prefix stage too (16 .. 25 samples per chip)
#include "dpe_bcs_fft.h"    // full-length FFT form (fallback for very wide lag / bin windows)

namespace dpe {

// ------------------------------------------------------------------------------------------
// blockIdx.x == 0: code bank (+ replica choice); blockIdx.x >= 1: 64 Doppler bins each.
// momLen = samples per moment block (kSub for the per-sample kernels, kPass for the chip kernel); nValid = entries of a
// block partial that hold lags (65, or 64 for the chip kernel whose lanes cover lagShift - 32 .. lagShift + 31).
// (batch form with 4 moments: held to 128 registers = four blocks per CU -- the kernel is a chain of latencies, 0.033 -> 0.030 ms at R)
// (NG = bin groups a block walks: 1 for the split shape, the handle's group count for the fat one -- the per-group registers of
//  unused groups made the 4-moment batch form spill 28 bytes under its 128-register bound)
template <int kNMom, bool FUSE, int NG>
__global__ __launch_bounds__(256, (kNMom == 4 && !FUSE) ? 4 : 1) void bcs_finalize_kernel(BcsParamBlock pb, int inl, int S, int K, int nSub, int nBlk, int LH, int L, int B, int wide, int lagShift,
                                                           int momLen, int nValid, long long C, const BcsChanDev *__restrict__ chan,
                                                           const float2 *__restrict__ part,
                                                           const float2 *__restrict__ mom,
                                                           float2 *__restrict__ codeBank, float2 *__restrict__ carrBank,
                                                           int *__restrict__ info, int maxK,
                                                           const float2 *__restrict__ momRep,     // fuse: moments of wipe x replica
                                                           const long long *__restrict__ sums, int nSumBlk)
{
    const int k = blockIdx.y, w = blockIdx.z, tid = threadIdx.x;
    const int NL = 2 * LH + 1;
    if (rerun_not_wanted(pb)) return;
    const BcsChanDev ch = params_ptr(chan, inl)[w * K + k];
    const float2 *pp = part + ((size_t)w * K + k) * nBlk * 2 * NL;
    constexpr int kChunk = 256;         // sub-tiles staged in LDS at a time
    __shared__ float2 sXY[2 * 65];      // [side][lag] totals of the per-block partials (NL <= 65)
    __shared__ float2 sTmp[256];
    __shared__ float2 sRed[16][16];
    __shared__ float2 sMom[kChunk * kNMom];   // flip-combined moments of the current chunk
    // A single window keeps only a few of these blocks in flight, so the kernel is a chain of memory
    // latencies: every stage issues ALL of its loads before consuming any.  Doppler blocks fetch their
    // first chunk of moments up front, under the partial-sum stage.
    // Two launch shapes: gridDim.x = 1 + nBinBlk (block 0: code bank, block 1+g: Doppler bins 16g..16g+15) for a few
    // windows, where more blocks mean a shorter latency chain; gridDim.x = 1 ("fat": one block per (window, SV)
    // does the code bank and then ALL bin groups from one staging of the moments) for batches, where the
    // redundant partial sums / moment reads of the split shape cost more than they hide.
    // lagShift != 0: a further chunk of a lag window wider than +-32 (one block per (window, SV): partial sums and
    // code-bank entries only; the replica choice was made by the lagShift == 0 launch and is read back from info[])
    const bool fat = gridDim.x == 1 && lagShift == 0;
    const bool carrBlk = lagShift == 0 && (fat || blockIdx.x != 0);
    const float2 *m0 = mom + (((size_t)w * K + k) * 2) * nSub * kNMom;
    const float2 *m1 = m0 + (size_t)nSub * kNMom;
    // fuse (single windows): the stage-1 kernel stored M_p[raw w r] in mom and M_p[w r] in momRep; the DC mean comes
    // from the per-block sample sums it left in `sums`:  M_p[(raw - mean) w r] = M_p[raw w r] - mean * M_p[w r]
    const float2 *q0 = momRep + (((size_t)w * K + k) * 2) * nSub * kNMom;
    const float2 *q1 = q0 + (size_t)nSub * kNMom;
    float mRe = 0.f, mIm = 0.f;
    if (FUSE) window_mean(sums, w, nSumBlk, S, mRe, mIm);
    float2 r0[kNMom], r1[kNMom], s0[FUSE ? kNMom : 1], s1[FUSE ? kNMom : 1];
    auto load_chunk = [&](int c0) {
        const int n = (nSub - c0 < kChunk ? nSub - c0 : kChunk) * kNMom;
#pragma unroll
        for (int i = 0; i < kNMom; ++i) {
            const int idx = tid + 256 * i;
            const bool ok = idx < n;
            r0[i] = ok ? m0[(size_t)c0 * kNMom + idx] : make_float2(0.f, 0.f);
            r1[i] = ok ? m1[(size_t)c0 * kNMom + idx] : make_float2(0.f, 0.f);
            if (FUSE) {
                s0[i] = ok ? q0[(size_t)c0 * kNMom + idx] : make_float2(0.f, 0.f);
                s1[i] = ok ? q1[(size_t)c0 * kNMom + idx] : make_float2(0.f, 0.f);
            }
        }
    };
    if (carrBlk) load_chunk(0);
    {
        // fixed-order two-level sum (bit-reproducible, identical in every block): T threads per
        // (side, lag) pair each add a strided subset of the blocks, then one thread adds the T partials
        const int pairs = 2 * NL;
        int np2 = 1;
        while (np2 < pairs) np2 <<= 1;
        const int T = 256 / np2;                     // pairs <= 130 -> T >= 1
        const int pair = tid / T, sub = tid - pair * T;
        // (compensated: a window's lag sums reach ~1e7 while a block partial is ~1e5 -- added plainly, a few hundred partials cost
        //  the total 4e-7 of its value; the third chip form writes one partial per group of tiles)
        float2 acc = make_float2(0.f, 0.f), comp = make_float2(0.f, 0.f);
        if (pair < pairs) {
            const int side = pair / NL, lag = pair - side * NL;
            for (int b0 = sub; b0 < nBlk; b0 += 8 * T) {
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int bq = b0 + u * T;
                    v[u] = bq < nBlk ? pp[(size_t)(bq * 2 + side) * NL + lag] : make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {   // Kahan steps (no fast-math in this build: the compiler keeps them)
                    const float yx = v[u].x - comp.x, yy = v[u].y - comp.y;
                    const float tx = acc.x + yx, ty = acc.y + yy;
                    comp.x = (tx - acc.x) - yx; comp.y = (ty - acc.y) - yy;
                    acc.x = tx; acc.y = ty;
                }
            }
        }
        sTmp[tid] = acc;
        __syncthreads();
        if (pair < pairs && sub == 0) {
            float2 t = sTmp[tid];
            for (int q = 1; q < T; ++q) { t.x += sTmp[tid + q].x; t.y += sTmp[tid + q].y; }
            sXY[pair] = t;
        }
        __syncthreads();
    }
    if (wide) {   // wide-lag layout: [0] = corr[-LH], [1+i] = step D[-LH+i]  ->  corr[j] by prefix sum
        if (tid < 2) {
            float2 run = sXY[tid * NL];
            for (int j = 1; j < NL; ++j) {
                run.x += sXY[tid * NL + j].x; run.y += sXY[tid * NL + j].y;
                sXY[tid * NL + j] = run;
            }
        }
        __syncthreads();
    }
    // BCS_ChooseCodeCorr :512-516 -- decision at lag 0 only
    const float2 X0 = sXY[LH], Y0 = sXY[NL + LH];
    const float nr0 = X0.x + Y0.x, ni0 = X0.y + Y0.y, fr0 = X0.x - Y0.x, fi0 = X0.y - Y0.y;
    const int noFlip = lagShift == 0 ? ((!ch.hasFlip) || (nr0 * nr0 + ni0 * ni0 > fr0 * fr0 + fi0 * fi0)) : info[w * K + k];
    const float sgn = noFlip ? 1.f : -1.f;

    if (blockIdx.x == 0) {
        if (tid == 0 && lagShift == 0) info[w * K + k] = noFlip;
        for (int jj = tid; jj < NL; jj += 256) {
            const int lag = lagShift + jj - LH;        // this chunk holds the lags lagShift - LH .. lagShift + LH
            if (lag < -L || lag > L || jj >= nValid) continue;
            const float2 X = sXY[jj], Y = sXY[NL + jj];
            codeBank[((size_t)w * maxK + k) * (2 * L + 1) + lag + L] = make_float2(X.x + sgn * Y.x, X.y + sgn * Y.y);
        }
        if (!fat) return;
    }
    // ---- Doppler bins: F[b] = sum_sub tw(sub,b) * sum_p (-j theta)^p / p! * M_p[sub]
    // 16 bins per group, 16 thread groups striding the sub-tiles; a fat block walks all bin groups per chunk
    constexpr int kMaxFatGroups = NG;
    const int nGroups = (2 * B + 1 + 15) / 16;
    const int gFirst = fat ? 0 : (int)blockIdx.x - 1, gCount = fat ? nGroups : 1;   // host: fat only if nGroups <= kMaxFatGroups
    const int grp = tid >> 4;
    const float invC = 1.0f / (float)C;  // C is a power of two: exact
    const double twoPiOverC = 6.283185307179586476925286766559 / (double)C;   // exact scaling of 2 pi (power of two)
    float2 F[kMaxFatGroups];
    float theta[kMaxFatGroups], stepS[kMaxFatGroups], stepC[kMaxFatGroups], sn[kMaxFatGroups], cs[kMaxFatGroups];
    bool live[kMaxFatGroups];
    int bb[kMaxFatGroups];
#pragma unroll
    for (int g = 0; g < kMaxFatGroups; ++g) {
        const int bi = (gFirst + g) * 16 + (tid & 15);  // bank entry
        bb[g] = bi - B;
        live[g] = g < gCount && bi < 2 * B + 1;
        F[g] = make_float2(0.f, 0.f);
        theta[g] = (float)((double)bb[g] * twoPiOverC);
        // centre twiddle exp(-j 2 pi n_c b / C), n_c = momLen sub + (momLen - 1) / 2: exact (integer-reduced phase +
        // sincospif) every 8th step of this thread, one complex rotation by exp(-j 2 pi 16 momLen b / C) between
        stepS[g] = 0.f; stepC[g] = 1.f; sn[g] = 0.f; cs[g] = 1.f;
        if (live[g]) {
            // (C = 8 next_pow2(S) is a power of two: the reduction mod 2C is a mask -- a 64-bit `%` by a run-time value costs
            // some 300 instructions per call and was 60 % of this kernel)
            const long long ts = ((long long)(32 * momLen) * (long long)bb[g]) & (2 * C - 1);   // 2 * (16 * momLen) b  (phase unit: pi / C)
            sincospif((float)ts * invC, &stepS[g], &stepC[g]);
        }
    }
    int it = 0;
    for (int c0 = 0; c0 < nSub; c0 += kChunk) {
        if (c0) {
            __syncthreads();   // the previous chunk is fully consumed
            load_chunk(c0);
        }
#pragma unroll
        for (int i = 0; i < kNMom; ++i) {
            float ax = fmaf(sgn, r1[i].x, r0[i].x), ay = fmaf(sgn, r1[i].y, r0[i].y);
            if (FUSE) {
                const float qx = fmaf(sgn, s1[i].x, s0[i].x), qy = fmaf(sgn, s1[i].y, s0[i].y);
                ax = fmaf(-mRe, qx, fmaf(mIm, qy, ax));
                ay = fmaf(-mRe, qy, fmaf(-mIm, qx, ay));
            }
            sMom[tid + 256 * i] = make_float2(ax, ay);
        }
        __syncthreads();
        const int cEnd = nSub - c0 < kChunk ? nSub - c0 : kChunk;
        for (int sl = grp; sl < cEnd; sl += 16, ++it) {   // kChunk is a multiple of 16: the stride continues across chunks
            const int sub = c0 + sl;
            const float2 *a = sMom + sl * kNMom;
#pragma unroll
            for (int g = 0; g < kMaxFatGroups; ++g) {
                if (!live[g]) continue;
                float ar = a[kNMom - 1].x, ai = a[kNMom - 1].y;
#pragma unroll
                for (int p = kNMom - 1; p >= 1; --p) {
                    const float sc = theta[g] * (1.0f / (float)p);   // (constant reciprocals: no division sequence in the loop set-up)
                    const float mr = a[p - 1].x, mi = a[p - 1].y;
                    const float nr = fmaf(sc, ai, mr);
                    ai = fmaf(-sc, ar, mi);
                    ar = nr;
                }
                if ((it & 7) == 0) {
                    const long long tt = (((long long)(2 * momLen) * sub + (momLen - 1)) * (long long)bb[g]) & (2 * C - 1);
                    sincospif((float)tt * invC, &sn[g], &cs[g]);  // angle = pi * tt / C
                } else {
                    // (explicit FMAs, here and below: the two launch shapes are different instantiations (NG), and left to the
                    // compiler's contraction their sums would round differently -- the shapes are bit-identical by test)
                    const float nc = fmaf(cs[g], stepC[g], -(sn[g] * stepS[g]));
                    sn[g] = fmaf(sn[g], stepC[g], cs[g] * stepS[g]);
                    cs[g] = nc;
                }
                // (cs - j sn) * (ar + j ai)
                F[g].x = fmaf(cs[g], ar, fmaf(sn[g], ai, F[g].x));
                F[g].y = fmaf(cs[g], ai, fmaf(-sn[g], ar, F[g].y));
            }
        }
    }
#pragma unroll
    for (int g = 0; g < kMaxFatGroups; ++g) {
        if (g >= gCount) break;   // block-uniform
        if (g) __syncthreads();
        sRed[grp][tid & 15] = F[g];
        __syncthreads();
        if (grp == 0 && live[g]) {
            float2 t = sRed[0][tid];
            for (int q = 1; q < 16; ++q) { t.x += sRed[q][tid].x; t.y += sRed[q][tid].y; }
            carrBank[((size_t)w * maxK + k) * (2 * B + 1) + (gFirst + g) * 16 + tid] = t;
        }
    }
}

// Device-resident channel parameters (dpe_bcs_update_dev): what the host loop of dpe_bcs_update computes per channel, from
// the reference's own port arrays on the device (cuChanMgr's outputs, dpeflow.cpp:169-176; captured once by the reference at
// batchcorrscores.cu:991-1004).  One block; fp64 throughout, expression for expression the host form (bcs_prep_one, dpe_prep.h).
// status: bit 0 = a PRN outside 1..37 (clamped so that the kernels stay inside the chip table), bit 1 = a non-positive code
// frequency / negative code phase (nominal values substituted: the kernels index the chip table with them).
struct BcsPortsDev {
    const double *rc, *ri, *fc, *fi;
    const int *cpEla, *cpRef;
    const unsigned char *prn;
};
// hintL1 > 0 (dpe_bcs_set_dev_hint): the host has chosen the chip kernels from NOMINAL channel values without reading this block back;
// the conditions it could not check are checked here, bit 3 of the status = one of them does not hold (chips of hintL1 or hintL1 + 1
// samples, code step within the assumed bound, 2 pi |fi| <= 0.25 fc, the nav-bit boundary on a chip boundary of the replica).
__global__ void bcs_prep_kernel(BcsPortsDev p, int K, double fs, int S, BcsChanDev *__restrict__ out, int *__restrict__ status, int hintL1,
                                double hintStepMax, int *__restrict__ hintViol)
{
    const int k = threadIdx.x;
    if (k == 0) atomicAnd(status, ~15);   // (bits 0 .. 3 are this kernel's; bits 2 and 4 belong to the batches whose DC sums ride in the stage-1 launch)
    __syncthreads();
    if (k >= K) return;
    int bad;
    const BcsChanDev d = bcs_prep_one(p.rc[k], p.ri[k], p.fc[k], p.fi[k], p.cpEla[k], p.cpRef[k], (int)p.prn[k], fs, S, bad);
    out[k] = d;
    if (hintL1 > 0 && hint_broken(d, hintL1, hintStepMax)) {
        bad |= 8;
        __hip_atomic_store(hintViol, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (pinned: the host drops the hint at its next call)
    }
    if (bad) atomicOr(status, bad);
}

// Dense export in the reference layout (complex128, fft-shifted rows), zero outside the banks.
__global__ void bcs_export_kernel(const float2 *__restrict__ bank, int n, long long rowLen, long long centre, int K,
                                  int maxK, int half, double2 *__restrict__ dense)
{
    const int k = blockIdx.y;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float2 v = bank[(size_t)k * n + j];
        dense[(size_t)k * rowLen + centre - half + j] = make_double2((double)v.x, (double)v.y);
    }
    (void)K; (void)maxK;
}

}  // namespace dpe

// ============================================================================================
struct dpe_bcs {
    dpe_bcs_config cfg;
    int LH;             // internal lag half width (4,8,16,32)
    int nSub, nBlk, tilesPerBlock, nMom;
    bool wideAllowed = true;     // (-DDPE_EXPERIMENTS builds: DPE_BCS_NO_WIDE=1 at create keeps the dense kernel)
    bool bank16Allowed = true;   // DPE_BCS_NO_BANK16=1: never the 16-samples-per-lane batch kernel (A/B tests)
    bool fuseAllowed = true;     // DPE_BCS_NO_FUSE=1: always the separate DC-sum kernel (A/B tests)
    bool chipAllowed = true;     // DPE_BCS_NO_CHIP=1: never the chip-boundary kernel (A/B tests)
    bool chipOK = false;         // create-time eligibility of the chip-boundary kernel (dpe_bcs_chip.h)
    bool chip2Allowed = true;    // DPE_BCS_NO_CHIP2=1: never the lanes-as-chips form (dpe_bcs_chip2.h; A/B tests)
    int chip2Resident = 0;       // co-resident waves of that kernel on the whole device
    int chip2PForce = 0;         // DPE_BCS_CHIP2_P at create: passes per tile of that kernel (experiments)
    int cus = 256;
    int nPassChip = 0, nBlkAlloc = 0;
    // full-length FFT fallback (dpe_bcs_fft.h): lag windows beyond DPE_MAX_LAG_HALF_WIDTH, bin windows beyond the moment
    // expansion, or DPE_BCS_FORCE_FFT=1 at create (A/B tests)
    bool fftMode = false, havePlans = false;
    dpe::FftPlan planS3, planS2, planC;   // rocFFT: length S, batch 3 chunk K (forward) / 2 chunk K (inverse); length C, batch chunk K
    int planK = 0;        // channels per window the plans are made for (the Update's nChan: re-planned when it changes)
    int fftChunkW = 1;
    float2 *fftWork_d = nullptr;
    int chipDbg = 0;                   // -DDPE_EXPERIMENTS builds only, DPE_BCS_CHIP_DBG: skips parts of the chip kernel (timing; wrong results)
    int fatForce = -1;                 // -DDPE_EXPERIMENTS builds only, DPE_BCS_FAT at create: finalize block shape (0 split, 1 fat)
    int chipTpbForce = 0;              // DPE_BCS_CHIP_TPB at create: passes per wave of the chip kernel (experiments)
    int tpb16Force = 0;                // DPE_BCS_TPB16 at create: tiles per block of the 16-samples-per-lane kernel (experiments)
    int resident16 = 1024;             // co-resident blocks of the 16-samples-per-lane kernel on the whole device
    int chipResident = 0;              // co-resident waves of the chip kernel on the whole device
    int chipTpbMax = 1;                // longest tile (passes) whose 6-moment block still meets the Taylor bound
    long long C;
    int8_t *chipTable_d = nullptr;
    uint32_t *chipBits_d = nullptr;   // the same chips as sign bits, periodically extended (bcs_bank_chip2_kernel: scalar loads)
    double *tTable_d = nullptr;   // ns-rounded sample times (always allocated; used only when useTable)
    bool useTable = false;
    long long *sums_d = nullptr;
    unsigned long long *rideWord_d = nullptr;   // [maxWindows][kSumSlots] {epoch, I, Q} words of the sums computed inside the chip2 launch
    unsigned rideEpoch = 0;                     // cycles 1 .. 3
    int rideW = 0, rideSlots = 0;               // the slot set the last such launch wrote
    bool rideAllowed = true;                    // DPE_BCS_NO_SUMRIDE=1: DC-sum kernel in front of the chip2 kernel, as before (A/B runs)
    int rideLA = 4;                             // look-ahead of the sum blocks in windows (-DDPE_EXPERIMENTS builds: DPE_BCS_RIDE_LA)
    int rideSpin = dpe::kRideSpinDefault;       // polls after which a correlator block sums its window itself (DPE_BCS_RIDE_SPIN; tests force 1)
    int rideMinW = 48;                          // smallest batch that takes the riding form (measured at H: 8 / 16 windows slower, 32 equal, 64 -1.2 %, 128 -2.6 %); DPE_BCS_SUMRIDE_MIN
    dpe::BcsChanDev *chan_d = nullptr;
    // pinned parameter staging: a ring of kStaging blocks, each guarded by an event recorded once its H2D copy (or the
    // graph that contains it) has been enqueued -- Updates may be issued kStaging - 1 deep without waiting
    static constexpr int kStaging = 4;
    dpe::BcsChanDev *chanBase_h = nullptr, *chan_h = nullptr, *chanBase_hd = nullptr;   // _hd: the pinned block's device address
    hipEvent_t stagingFree[kStaging] = {};
    int slot = 0;
    float2 *part_d = nullptr, *mom_d = nullptr, *momRep_d = nullptr, *codeBank_d = nullptr, *carrBank_d = nullptr;
    int *info_d = nullptr;
    int lastW = 0, lastK = 0, lastSumBlocks = 1;
    bool lastDev = false;              // the last Update took its channel parameters from device arrays (dpe_bcs_update_dev)
    int *status_d = nullptr;           // that form's input check, written by bcs_prep_kernel
    const char *lastKernel = "";   // stage-1 kernel of the last Update (dpe_bcs_stage1_kernel)
    dpe::ChmKArgs co{};            // a task of the device-resident channel manager for the next stage-1 launch (dpe_bcs_cotask_set)
    bool coPending = false;
    dpe_owner_detach_fn ownerDetach = nullptr;   // an attached device-resident channel manager: told first when this handle is destroyed
    void *owner = nullptr;
    int devHint = 0;               // dpe_bcs_set_dev_hint: bit 0 = the caller promises the chip kernels' conditions for the device-parameter form
    int *hintViol_h = nullptr, *hintViol_hd = nullptr;   // pinned word a device-side check raises when the promise did not hold: the hint is then dropped
    std::vector<int32_t> idxNext_h;
    dpe::KernelProfiler prof;  // slots: 0 sum, 1 bank, 2 finalize
    dpe::GraphCache graphs;
};

// Blocks per window of the DC-sum kernel: ~8192 samples per block for batches; few windows get up to
// one int4 load per thread (a lone window is latency-bound, not bandwidth-bound).
static int sum_blocks(int S, int nWindows)
{
    const int base = S / 8192 > 0 ? S / 8192 : 1;
    const int fine = (S / 4 + 255) / 256;
    int want = 512 / nWindows < fine ? 512 / nWindows : fine;
    if (want < base) want = base;
    return want > dpe::kSumSlots ? dpe::kSumSlots : want;
}

// widest supported lag window: the centre chunk (+-32) plus 4 chunks of 65 lags on each side
static constexpr int kMaxLagHalfWidth = DPE_MAX_LAG_HALF_WIDTH;
static_assert(kMaxLagHalfWidth == 32 + 4 * 65, "centre chunk plus four 65-lag chunks per side");

static long long next_pow2(long long x)
{
    long long p = 1;
    while (p < x) p <<= 1;
    return p;
}

extern "C" {

int dpe_bcs_create(const dpe_bcs_config *cfg, dpe_bcs **out)
{
    using namespace dpe;
    DPE_REQUIRE(cfg && out, "[BatchCorrScores] create: null argument");
    DPE_REQUIRE(cfg->samplesPerWindow >= 1024, "[BatchCorrScores] create: samplesPerWindow %d < 1024", cfg->samplesPerWindow);
    DPE_REQUIRE(cfg->samplingFrequency > 0, "[BatchCorrScores] create: bad samplingFrequency");
    DPE_REQUIRE(cfg->maxWindows >= 1 && cfg->maxChannels >= 1 && cfg->maxChannels <= DPE_MAX_CHAN,
                "[BatchCorrScores] create: maxWindows/maxChannels out of range");
    DPE_REQUIRE(cfg->lagHalfWidth >= 1 && 2 * (long long)cfg->lagHalfWidth + 1 < cfg->samplesPerWindow,
                "[BatchCorrScores] create: lagHalfWidth %d not in [1, S/2)", cfg->lagHalfWidth);
    DPE_REQUIRE(cfg->binHalfWidth >= 1, "[BatchCorrScores] create: binHalfWidth < 1");
    bool fftMode = getenv("DPE_BCS_FORCE_FFT") != nullptr;   // full-length FFT form instead of the streaming kernels
    if (cfg->lagHalfWidth > kMaxLagHalfWidth || cfg->lagHalfWidth + 32 + dpe::kSub >= cfg->samplesPerWindow) fftMode = true;
    const int S = cfg->samplesPerWindow;
    const long long C = 8 * next_pow2(S);  // batchcorrscores.cu:761
    // Taylor remainder of the moment expansion must stay below fp32 rounding (see file header)
    const double th = 6.283185307179586 * 127.5 * cfg->binHalfWidth / (double)C;
    const int nMom = (std::pow(th, 4) / 24.0 < 1e-7) ? 4 : 6;   // Taylor order of the moment expansion
    DPE_REQUIRE(2 * (long long)cfg->binHalfWidth + 1 < C, "[BatchCorrScores] create: binHalfWidth %d not below C/2 = %lld",
                cfg->binHalfWidth, C / 2);
    if (!(std::pow(th, 6) / 720.0 < 2e-7)) fftMode = true;   // bin window too wide for the moment expansion at this C
    // The reference rounds the sample times to 1 ns (BCS_GenTimeIdcs :191-193).  For integer-ns sampling
    // periods (all usual SDR rates) that is a no-op and the kernels use n/fs; otherwise they read the
    // reference's own table.
    const double fs = cfg->samplingFrequency;
    bool needTable = false;
    for (int n = 0; n < S && !needTable; ++n) {
        const double t = (double)n / fs, tr = std::round(t * 1.0e9) / 1.0e9;
        if (std::fabs(t - tr) > 4e-16 * (t + 1e-9)) needTable = true;
    }
    dpe_bcs *h = new dpe_bcs();
    h->cfg = *cfg;
    h->C = C;
    h->nMom = nMom;
    h->fftMode = fftMode;
    h->LH = cfg->lagHalfWidth <= 4 ? 4 : cfg->lagHalfWidth <= 8 ? 8 : cfg->lagHalfWidth <= 16 ? 16 : 32;
    h->nSub = (S + kSub - 1) / kSub;
    const int nTiles = (h->nSub + 3) / 4;
    h->tilesPerBlock = (nTiles + 63) / 64;   // fewest tiles per block ever used -> sizes the partial buffer
    h->nBlk = (nTiles + h->tilesPerBlock - 1) / h->tilesPerBlock;
    const size_t W = cfg->maxWindows, K = cfg->maxChannels;
    std::vector<int8_t> table(37 * 1024, 0);
    for (int prn = 1; prn <= 37; ++prn) gen_ca_code_host(prn, table.data() + (prn - 1) * 1024);
    h->chipTable_d = dev_alloc<int8_t>(table.size());
    std::vector<uint32_t> bitTable((size_t)37 * k2BitWords, 0u);
    for (int prn = 1; prn <= 37; ++prn)
        for (int b = 0; b < 32 * k2BitWords; ++b)
            if (table[(size_t)(prn - 1) * 1024 + b % kLCA] > 0) bitTable[(size_t)(prn - 1) * k2BitWords + (b >> 5)] |= 1u << (b & 31);
    h->chipBits_d = dev_alloc<uint32_t>(bitTable.size());
    {
        std::vector<double> tt(S);
        for (int n = 0; n < S; ++n) tt[n] = std::round(((double)n / fs) * 1.0e9) / 1.0e9;
        h->tTable_d = dev_alloc<double>(S);
        if (h->tTable_d) (void)hipMemcpy(h->tTable_d, tt.data(), sizeof(double) * S, hipMemcpyHostToDevice);
        h->useTable = needTable;
    }
    h->sums_d = dev_alloc<long long>(2 * W * kSumSlots);
    h->rideWord_d = dev_alloc<unsigned long long>(W * kSumSlots);
    if (h->rideWord_d) (void)hipMemset(h->rideWord_d, 0, sizeof(unsigned long long) * W * kSumSlots);
    h->rideAllowed = !(getenv("DPE_BCS_NO_SUMRIDE") && atoi(getenv("DPE_BCS_NO_SUMRIDE")) != 0);
    if (getenv("DPE_BCS_SUMRIDE_MIN") && atoi(getenv("DPE_BCS_SUMRIDE_MIN")) >= 1) h->rideMinW = atoi(getenv("DPE_BCS_SUMRIDE_MIN"));
#ifdef DPE_EXPERIMENTS
    if (getenv("DPE_BCS_RIDE_LA") && atoi(getenv("DPE_BCS_RIDE_LA")) >= 1) h->rideLA = atoi(getenv("DPE_BCS_RIDE_LA"));
#endif
    if (getenv("DPE_BCS_RIDE_SPIN") && atoi(getenv("DPE_BCS_RIDE_SPIN")) >= -1) h->rideSpin = atoi(getenv("DPE_BCS_RIDE_SPIN"));
    h->chan_d = dev_alloc<BcsChanDev>(W * K);
    // chip-boundary kernel (dpe_bcs_chip.h): lag windows of 17..31 samples (wider: chunks of 64 lags) at sampling rates where a
    // sub-tile holds few chips, plain n/fs sample times; its moment block is one pass of kPass samples, and a chip's
    // second-order term (theta len)^2 / 24 must stay below fp32 rounding
    h->nPassChip = (S + kPass - 1) / kPass;
    {
        const double thetaB = 6.283185307179586 * cfg->binHalfWidth / (double)C;   // phase step per sample of the outermost bin
        const double chipLen = fs / kFCA * 1.001;
        // the moment block is the wave's tile of tpb passes, 6 moments: (thetaB tpb kPass / 2)^6 / 720 < 2e-7
        const double halfMax = std::pow(2e-7 * 720.0, 1.0 / 6.0) / thetaB;
        h->chipTpbMax = (int)(2.0 * halfMax / kPass);
        if (h->chipTpbMax > 32) h->chipTpbMax = 32;
        h->chipOK = h->LH == 32 && !needTable && (kFCA / fs) * 128.0 < 40.0 && h->chipTpbMax >= 1 && chipLen < 31.0 &&
                    (thetaB * chipLen) * (thetaB * chipLen) / 24.0 < 5e-7 && S >= 2 * kPass;
    }
    h->nBlkAlloc = h->nBlk;
    {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        h->resident16 = ((h->LH <= 4 && !needTable) ? dpe::kB16Blocks : 3) * cus;   // the kernel's launch bounds
        h->cus = cus;
    }
    if (h->chipOK) {
        // partial sums of up to 128 blocks per (window, SV); a handle for a few windows gets one block per pass
        int want = (int)(2048 / W) > 128 ? (int)(2048 / W) : 128;
        if (want > h->nPassChip) want = h->nPassChip;
        if (want < (h->nPassChip + h->chipTpbMax - 1) / h->chipTpbMax) want = (h->nPassChip + h->chipTpbMax - 1) / h->chipTpbMax;
        if (h->nBlkAlloc < want) h->nBlkAlloc = want;
        int dev = 0, cus = 256, nb = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        h->chipResident = 12 * cus;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)bcs_bank_chip_kernel<6>, 64, 0) == hipSuccess && nb > 0)
            h->chipResident = nb * cus;
        h->chip2Resident = 11 * cus;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)bcs_bank_chip2_kernel<6, 24>, 64, 0) == hipSuccess && nb > 0)
            h->chip2Resident = nb * cus;
    }
    h->part_d = dev_alloc<float2>(W * K * h->nBlkAlloc * 2 * (2 * h->LH + 1));
    h->mom_d = dev_alloc<float2>(W * K * 2 * h->nSub * kNMomMax);
    h->momRep_d = dev_alloc<float2>(W * K * 2 * h->nSub * kNMomMax);
    h->codeBank_d = dev_alloc<float2>(W * K * (2 * cfg->lagHalfWidth + 1));
    h->carrBank_d = dev_alloc<float2>(W * K * (2 * cfg->binHalfWidth + 1));
    h->info_d = dev_alloc<int>(W * K);
    h->status_d = dev_alloc<int>(1);
    if (h->status_d) (void)hipMemset(h->status_d, 0, sizeof(int));
    if (hipHostMalloc((void **)&h->hintViol_h, sizeof(int), hipHostMallocDefault) == hipSuccess) {
        *h->hintViol_h = 0;
        (void)hipHostGetDevicePointer((void **)&h->hintViol_hd, h->hintViol_h, 0);
    }
    if (!h->status_d || !h->hintViol_hd || !h->rideWord_d || !h->tTable_d || !h->chipTable_d || !h->chipBits_d || !h->sums_d || !h->chan_d || !h->part_d || !h->mom_d || !h->momRep_d || !h->codeBank_d || !h->carrBank_d ||
        !h->info_d || hipHostMalloc((void **)&h->chanBase_h, dpe_bcs::kStaging * W * K * sizeof(BcsChanDev), hipHostMallocDefault) != hipSuccess) {
        set_error("[BatchCorrScores] create: device allocation failed");
        dpe_bcs_destroy(h);
        return -1;
    }
    const auto finish = [&]() -> int {   // a failure from here on must not leak the handle
        DPE_CHECK_HIP(hipMemcpy(h->chipTable_d, table.data(), table.size(), hipMemcpyHostToDevice));
        DPE_CHECK_HIP(hipMemcpy(h->chipBits_d, bitTable.data(), bitTable.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        DPE_CHECK_HIP(hipMemset(h->codeBank_d, 0, W * K * (2 * cfg->lagHalfWidth + 1) * sizeof(float2)));
        DPE_CHECK_HIP(hipMemset(h->carrBank_d, 0, W * K * (2 * cfg->binHalfWidth + 1) * sizeof(float2)));
        for (hipEvent_t &e : h->stagingFree) DPE_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        DPE_CHECK_HIP(hipHostGetDevicePointer((void **)&h->chanBase_hd, h->chanBase_h, 0));
        if (h->fftMode) {
            // windows per chunk: at most 2^27 complex work elements (1 GB); planes b, rX, rY of [chunk][K][S], reused as [chunk][K][C]
            const size_t perW = K * (size_t)(3 * (size_t)S > (size_t)C ? 3 * (size_t)S : (size_t)C);
            size_t chunk = ((size_t)1 << 27) / perW;
            if (chunk < 1) chunk = 1;
            if (chunk > W) chunk = W;
            h->fftChunkW = (int)chunk;
            h->fftWork_d = dev_alloc<float2>(chunk * perW);
            DPE_REQUIRE(h->fftWork_d, "[BatchCorrScores] create: FFT work buffer (%zu MB) allocation failed", chunk * perW * 8 >> 20);
            DPE_REQUIRE(C < (1ll << 31), "[BatchCorrScores] create: carrier transform length %lld too long for the FFT path", C);
            // (rows of a short last chunk / of unused channels are transformed but never read: keep them finite)
            DPE_CHECK_HIP(hipMemset(h->fftWork_d, 0, chunk * perW * sizeof(float2)));
            // the batched plans are made at the first Update, for the channel count actually used (and again when it changes)
        }
        return 0;
    };
    if (finish()) {
        dpe_bcs_destroy(h);
        return -1;
    }
    h->idxNext_h.assign(W * K, 0);
    h->chan_h = h->chanBase_h;
#ifdef DPE_EXPERIMENTS
    h->wideAllowed = getenv("DPE_BCS_NO_WIDE") == nullptr;
#endif
    h->bank16Allowed = getenv("DPE_BCS_NO_BANK16") == nullptr;
    h->fuseAllowed = getenv("DPE_BCS_NO_FUSE") == nullptr;
    h->chipAllowed = getenv("DPE_BCS_NO_CHIP") == nullptr;
    h->chip2Allowed = getenv("DPE_BCS_NO_CHIP2") == nullptr;
    if (const char *e = getenv("DPE_BCS_CHIP2_P")) h->chip2PForce = atoi(e);
    if (const char *e = getenv("DPE_BCS_CHIP_TPB")) h->chipTpbForce = atoi(e);
    if (const char *e = getenv("DPE_BCS_TPB16")) h->tpb16Force = atoi(e);
#ifdef DPE_EXPERIMENTS   // ablation switches: never in the product library (the first one makes the banks wrong)
    if (const char *e = getenv("DPE_BCS_CHIP_DBG")) h->chipDbg = atoi(e);
    if (const char *e = getenv("DPE_BCS_FAT")) h->fatForce = e[0] == '1' ? 1 : 0;
#endif
    *out = h;
    return 0;
}

int dpe_bcs_destroy(dpe_bcs *h)
{
    if (!h) return 0;
    if (h->ownerDetach) h->ownerDetach(h->owner, 0);   // (runs the parked time update, forgets this handle)
    void *bufs[] = {h->tTable_d, h->chipTable_d, h->chipBits_d, h->sums_d, h->rideWord_d, h->chan_d, h->part_d, h->mom_d, h->momRep_d, h->codeBank_d, h->carrBank_d, h->info_d, h->status_d};
    for (void *b : bufs) (void)hipFree(b);
    if (h->chanBase_h) (void)hipHostFree(h->chanBase_h);
    if (h->hintViol_h) (void)hipHostFree(h->hintViol_h);
    h->planS3.destroy(); h->planS2.destroy(); h->planC.destroy();
    (void)hipFree(h->fftWork_d);
    for (hipEvent_t e : h->stagingFree)
        if (e) (void)hipEventDestroy(e);
    h->graphs.clear();
    delete h;
    return 0;
}

// chan_host == nullptr: the channel parameters of this (single-window) call are already in h->chan_d, written by
// bcs_prep_kernel earlier on `stream` (dpe_bcs_update_dev) -- nothing the host decides below may then depend on their values,
// except the choice of a chip-boundary kernel at high sampling rates (which fetches them first, see there).
// rerun = 1 (internal): the guarded general kernels behind a hinted chip-kernel launch (see BcsParamBlock::guard).
static int bcs_update_impl(dpe_bcs *h, const int16_t *samples_dev, int64_t windowStrideSamples, int32_t nWindows,
                           int32_t nChan, const dpe_chan_start *chan_host, dpe_stream_t stream_, int rerun = 0)
{
    using namespace dpe;
    const bool dev = chan_host == nullptr;
    DPE_REQUIRE(h && samples_dev, "[BatchCorrScores] Update: null argument");
    DPE_REQUIRE(nWindows >= 1 && nWindows <= h->cfg.maxWindows, "[BatchCorrScores] Update: nWindows %d out of range", nWindows);
    DPE_REQUIRE(nChan >= 1 && nChan <= h->cfg.maxChannels, "[BatchCorrScores] Update: nChan %d out of range", nChan);
    const int S = h->cfg.samplesPerWindow;
    DPE_REQUIRE(nWindows == 1 || windowStrideSamples >= S, "[BatchCorrScores] Update: window stride < S");
    hipStream_t stream = (hipStream_t)stream_;
    const double fs = h->cfg.samplingFrequency;
    // next staging block of the ring; an Update issued kStaging calls ago may still be copying it (no-op otherwise)
    h->slot = (h->slot + 1) % dpe_bcs::kStaging;
    DPE_CHECK_HIP(hipEventSynchronize(h->stagingFree[h->slot]));
    h->chan_h = h->chanBase_h + (size_t)h->slot * h->cfg.maxWindows * h->cfg.maxChannels;
    h->lastDev = dev;
    for (int i = 0; !dev && i < nWindows * nChan; ++i) {
        const dpe_chan_start &c = chan_host[i];
        DPE_REQUIRE(c.prn >= 1 && c.prn <= kPrnMax, "[BatchCorrScores] Update: PRN %d out of range", c.prn);
        DPE_REQUIRE(c.codeFrequency > 0 && c.codePhaseStart >= 0, "[BatchCorrScores] Update: bad code phase/frequency");
        BcsChanDev &d = h->chan_h[i];
        d.rc = c.codePhaseStart;
        d.codeStep = c.codeFrequency / fs;
        d.ri = c.carrierPhaseStart;
        d.carrStep = c.carrierFrequency / fs;
        d.fc = c.codeFrequency;
        d.fi = c.carrierFrequency;
        d.invStep = fs / c.codeFrequency;
        const double ang = -6.283185307179586476925286766559 * d.carrStep;
        d.rotRe = (float)std::cos(ang);
        d.rotIm = (float)std::sin(ang);
        // BCS_NavBitBoundary, batchcorrscores.cu:247-253
        const int since = (((c.cpElapsedStart - c.cpReference) % 20) + 20) % 20;
        d.idxNext = (int)(std::floor((kLCA * (20 - since) - c.codePhaseStart) * (fs / c.codeFrequency)) + 1);
        d.hasFlip = (d.idxNext > 0 && d.idxNext < S) ? 1 : 0;
        d.prn = c.prn;
        d.pad = 0;
        h->idxNext_h[i] = d.idxNext;
    }
    // chip-boundary kernel: the closed-form DC term of a chip needs 2 pi |fi| / fc <= 0.25 (|fi| below ~40 kHz).  Its
    // eligibility (and the second form's below) is a property of the channel VALUES.  The device-parameter form has them
    // only on the device: at sampling rates where these kernels exist at all (>= 16 samples per chip) it reads back the
    // block bcs_prep_kernel has just derived -- <= 3 KB and one stream wait per window, against a 4-5 x slower stage 1;
    // at lower rates nothing is fetched and nothing the host decides depends on the values.
    bool chip = h->chipOK && h->chipAllowed && !rerun;
    bool hintedCall = false;   // this call chose the chip kernels on the caller's promise: the guarded re-run follows it
    if (dev && (h->devHint & 1) && __atomic_load_n(h->hintViol_h, __ATOMIC_RELAXED)) {
        // a device-side check found the promise broken in an earlier window (flagged there: status bit 3): the hint is withdrawn for the
        // life of the handle -- from here on the call reads the derived block back and chooses the kernel from the real values
        h->devHint &= ~1;
    }
    if (chip && dev && (h->devHint & 1)) {
        hintedCall = true;
        // the caller's promise (dpe_bcs_set_dev_hint): decide from nominal values -- code frequency F_CA within 1e-5 (ten times the
        // largest code Doppler), carrier offset inside the closed-form DC term's range -- and let bcs_prep_kernel check what was
        // promised (status bit 3); nothing is read back, the host does not wait for the stream
        for (int i = 0; i < nWindows * nChan; ++i) {
            BcsChanDev &d = h->chan_h[i];
            d = BcsChanDev{};
            d.fc = kFCA; d.fi = 0.0;
            d.codeStep = kFCA / fs * (1.0 + 1e-5);
            d.invStep = fs / kFCA;
            d.hasFlip = 0;
        }
    } else if (chip && dev) {
        DPE_CHECK_HIP(hipMemcpyAsync(h->chan_h, h->chan_d, sizeof(BcsChanDev) * nWindows * nChan, hipMemcpyDeviceToHost, stream));
        DPE_CHECK_HIP(hipStreamSynchronize(stream));
    }
    for (int i = 0; chip && i < nWindows * nChan; ++i)
        if (6.283185307179586 * std::fabs(h->chan_h[i].fi) > 0.25 * h->chan_h[i].fc) chip = false;
    // second form of the chip kernel (dpe_bcs_chip2.h): every chip 16 .. 25 samples long, and the nav-bit boundary
    // (BCS_NavBitBoundary :247-253) on a chip boundary of the replica (:347-349) -- it is, unless fp64 rounding separates them
    bool chip2 = chip && h->chip2Allowed;
    double stepMax = 0.0;
    int c2L1 = 0;
    for (int i = 0; chip2 && i < nWindows * nChan; ++i) {
        const BcsChanDev &d = h->chan_h[i];
        const int l1 = (int)d.invStep;   // chips of l1 or l1 + 1 samples; instantiated for every l1 in 16 .. 24 (16.4 .. 25.6 Msps)
        if (i == 0) c2L1 = l1;
        if (l1 != c2L1 || l1 < k2MinL1 || l1 > k2MaxL1) chip2 = false;
        if (d.hasFlip && (int)std::fma((double)d.idxNext, d.codeStep, d.rc) == (int)std::fma((double)(d.idxNext - 1), d.codeStep, d.rc)) chip2 = false;
        if (d.codeStep > stepMax) stepMax = d.codeStep;
    }
    h->lastW = nWindows;
    h->lastK = nChan;
    if (h->fftMode) {
        if (h->coPending) {   // (a pending task of the device-resident channel manager: a kernel of its own in front)
            hipLaunchKernelGGL(chm_k2_kernel, dim3(1), dim3(256), 0, stream, h->co);
            h->coPending = false;
        }
        // ---- full-length FFT form (dpe_bcs_fft.h): DC sum, then per chunk of windows the reference's own sequence
        const long long C = h->C;
        const int L = h->cfg.lagHalfWidth, B = h->cfg.binHalfWidth, maxK = h->cfg.maxChannels;
        const int sumBlocks = sum_blocks(S, nWindows);
        h->lastSumBlocks = sumBlocks;
        h->lastKernel = "rocfft full-lag path (bcs_fft_*_kernel)";
        if (dev) {}
        else if (h->graphs.capturing) DPE_CHECK_HIP(hipMemcpyAsync(h->chan_d, h->chan_h, sizeof(BcsChanDev) * nWindows * nChan, hipMemcpyHostToDevice, stream));
        else upload_params(h->chan_d, h->chanBase_hd + (h->chan_h - h->chanBase_h), sizeof(BcsChanDev) * nWindows * nChan, stream);
        DPE_CHECK_HIP(hipEventRecord(h->stagingFree[h->slot], stream));
        h->prof.begin(0, stream);
        hipLaunchKernelGGL(bcs_sum_kernel, dim3(sumBlocks, nWindows), dim3(256), 0, stream, samples_dev, (long long)windowStrideSamples, S, h->sums_d,
                           (uint4 *)nullptr, (const uint4 *)nullptr, 0);
        h->prof.end(0, stream);
        const int chunk = h->fftChunkW;
        if (!h->havePlans || h->planK != nChan) {   // batched over the channels this host really tracks, not over maxChannels
            h->havePlans = false;
            const size_t b3 = (size_t)3 * chunk * nChan, b2 = (size_t)2 * chunk * nChan, b1 = (size_t)chunk * nChan;
            if (h->planS3.create((size_t)S, b3, false) || h->planS2.create((size_t)S, b2, true) || h->planC.create((size_t)C, b1, false)) {
                h->planS3.destroy(); h->planS2.destroy(); h->planC.destroy();
                return -1;   // (the message is rocFFT's, from dpe_fft.h)
            }
            h->havePlans = true;
            h->planK = nChan;
        }
        h->prof.begin(1, stream);
        const size_t plane = (size_t)chunk * nChan * S;
        const int gxS = S / 256 < 64 ? (S / 256 > 0 ? S / 256 : 1) : 64;
#define DPE_FFT_EXEC(plan, ptr)                  \
    if ((plan).exec(stream, (void *)(ptr))) {    \
        h->prof.end(1, stream);                  \
        return -1;                               \
    }
        for (int w0 = 0; w0 < nWindows; w0 += chunk) {
            const int nw = nWindows - w0 < chunk ? nWindows - w0 : chunk;
            // the plans transform `chunk` windows of rows; a short last chunk carries stale (finite) rows that nobody reads
            if (h->useTable)
                hipLaunchKernelGGL((bcs_fft_prep_code_kernel<true>), dim3(gxS, nChan, nw), dim3(256), 0, stream, samples_dev, (long long)windowStrideSamples,
                                   S, nChan, chunk, w0, h->chan_d, h->chipTable_d, h->tTable_d, h->fftWork_d);
            else
                hipLaunchKernelGGL((bcs_fft_prep_code_kernel<false>), dim3(gxS, nChan, nw), dim3(256), 0, stream, samples_dev, (long long)windowStrideSamples,
                                   S, nChan, chunk, w0, h->chan_d, h->chipTable_d, h->tTable_d, h->fftWork_d);
            DPE_FFT_EXEC(h->planS3, h->fftWork_d);
            hipLaunchKernelGGL(bcs_fft_mul_kernel, dim3(2048), dim3(256), 0, stream, h->fftWork_d, plane);
            DPE_FFT_EXEC(h->planS2, h->fftWork_d + plane);
            hipLaunchKernelGGL(bcs_fft_extract_code_kernel, dim3(nChan, nw), dim3(256), 0, stream, h->fftWork_d, plane, S, nChan, L, w0, maxK,
                               h->chan_d, h->codeBank_d, h->info_d);
            const int gxC = 256;
            if (h->useTable)
                hipLaunchKernelGGL((bcs_fft_prep_carr_kernel<true>), dim3(gxC, nChan, nw), dim3(256), 0, stream, samples_dev, (long long)windowStrideSamples,
                                   S, C, nChan, w0, sumBlocks, h->chan_d, h->sums_d, h->chipTable_d, h->tTable_d, h->info_d, h->fftWork_d);
            else
                hipLaunchKernelGGL((bcs_fft_prep_carr_kernel<false>), dim3(gxC, nChan, nw), dim3(256), 0, stream, samples_dev, (long long)windowStrideSamples,
                                   S, C, nChan, w0, sumBlocks, h->chan_d, h->sums_d, h->chipTable_d, h->tTable_d, h->info_d, h->fftWork_d);
            DPE_FFT_EXEC(h->planC, h->fftWork_d);
            hipLaunchKernelGGL(bcs_fft_extract_carr_kernel, dim3(nChan, nw), dim3(256), 0, stream, h->fftWork_d, C, nChan, B, w0, maxK, h->carrBank_d);
        }
#undef DPE_FFT_EXEC
        h->prof.end(1, stream);
        DPE_CHECK_HIP(hipGetLastError());
        return 0;
    }
    // the per-kernel event timing and the graph replay exclude each other
    const bool useGraph = h->graphs.enabled && !h->prof.enabled && !dev;
    GraphCache::Guard graphGuard{h->graphs, stream};
    int sumBlocks = sum_blocks(S, nWindows);
    if (useGraph) {
        const int rc = h->graphs.begin({samples_dev, nullptr, (long long)windowStrideSamples, nWindows, nChan,
                                        (h->wideAllowed ? 1 : 0) | (h->bank16Allowed ? 2 : 0) | (h->slot << 8), stream}, stream);
        DPE_REQUIRE(rc >= 0, "[BatchCorrScores] Update: hipGraph capture/replay failed");
        if (rc == 1) {
            DPE_CHECK_HIP(hipEventRecord(h->stagingFree[h->slot], stream));   // the replayed graph reads this slot
            return 0;
        }
    }
    // few channels (the per-window call of a running receiver): parameters travel as kernel arguments of
    // the bank / finalize kernels; batches go through one H2D copy from the pinned staging block.  A
    // captured graph would freeze by-value arguments, so the graph path always copies.
    const int inl = (!dev && !h->graphs.capturing && nWindows * nChan <= DPE_MAX_CHAN) ? 1 : 0;
    BcsParamBlock pb{};
    pb.guard = rerun ? h->status_d : nullptr;
    if (inl) memcpy(pb.c, h->chan_h, sizeof(BcsChanDev) * nWindows * nChan);
    // (batches: the DC-sum kernel below carries the parameter upload; a captured graph keeps a copy node)
    else if (h->graphs.capturing)
        DPE_CHECK_HIP(hipMemcpyAsync(h->chan_d, h->chan_h, sizeof(BcsChanDev) * nWindows * nChan, hipMemcpyHostToDevice, stream));
    const bool upInSum = !dev && !inl && !h->graphs.capturing;
    const int vecOK = (((uintptr_t)samples_dev & 15) == 0 && (windowStrideSamples % 4) == 0) ? 1 : 0;
    // |lag| <= 32 windows: boundary-difference kernel when a sub-tile holds few chip boundaries
    // (~128 codeStep per lag step against 4 x 65 dense FMAs per lane), else a dense kernel
    const bool wide = !chip && h->LH == 32 && h->wideAllowed && (kFCA / fs) * 128.0 < 40.0;
    // narrow lag windows in batches: the 16-samples-per-lane kernel (tiles of 16 sub-tiles) once the batch
    // offers >= 2048 of its tiles; fewer (a single window in particular) keep the short 4-sub-tile tiles
    const int nTiles16 = (h->nSub + 15) / 16;
    const bool use16 = !wide && h->LH <= 8 && h->bank16Allowed && (long long)nTiles16 * nChan * nWindows >= 2048;
    const int nTiles = use16 ? nTiles16 : (h->nSub + 3) / 4;
    int tpb;   // tiles per block: amortise the end-of-block lag reduction while keeping >= ~4096 blocks in flight
    if (use16) {
        // the choice that minimises rounds x (tiles + fixed cost): a block's start-up (chip table into LDS, parameters, the
        // end-of-side lag reductions) measures ~0.4 tile times, and a launch whose blocks are all resident in two rounds beats
        // eight rounds of short ones (config R: 4 tiles per block 0.237 ms, all 13 in one block 0.181 ms)
        const int least = (nTiles + h->nBlk - 1) / h->nBlk;   // the partial buffer holds h->nBlk blocks per (window, SV)
        double best = -1.0;
        tpb = least < 1 ? 1 : least;
        for (int c = tpb; c <= nTiles; ++c) {
            const long long blocks = (long long)((nTiles + c - 1) / c) * nChan * nWindows;
            const double cost = (double)((blocks + h->resident16 - 1) / h->resident16) * ((double)c + 0.4);
            if (best < 0.0 || cost < best) { best = cost; tpb = c; }
        }
        if (h->tpb16Force >= least && h->tpb16Force >= 1) tpb = h->tpb16Force;
    } else {
        tpb = (int)(((long long)nTiles * nChan * nWindows) / 4096);
        if (tpb < h->tilesPerBlock) tpb = h->tilesPerBlock;
        if (tpb > 16) tpb = 16;
    }
    // chip-boundary kernel: one wave per block walks tpb passes of one (window, SV)
    int chipNMom = 6;
    if (chip) {
        // passes per wave: the choice that leaves the smallest tail -- ceil(waves / resident waves) rounds of tpb passes each
        const int least = (h->nPassChip + h->nBlkAlloc - 1) / h->nBlkAlloc;
        long long best = -1;
        tpb = least;
        for (int c = least; c <= h->chipTpbMax; ++c) {
            const long long waves = (long long)((h->nPassChip + c - 1) / c) * nChan * nWindows;
            // + the finalize kernel's share: its time grows by ~0.06 pass-rounds per partial block (measured, config H)
            const long long cost = 100 * ((waves + h->chipResident - 1) / h->chipResident) * c + 6 * ((h->nPassChip + c - 1) / c);
            if (best < 0 || cost < best) { best = cost; tpb = c; }
        }
        if (h->chipTpbForce >= least && h->chipTpbForce <= h->chipTpbMax) tpb = h->chipTpbForce;
        const double thc = 6.283185307179586 * h->cfg.binHalfWidth / (double)h->C * 0.5 * ((double)tpb * kPass - 1.0);
        chipNMom = (std::pow(thc, 4) / 24.0 < 1e-7) ? 4 : 6;
    }
    const int nBlk = chip ? (h->nPassChip + tpb - 1) / tpb : (nTiles + tpb - 1) / tpb;
    // second form: a tile = the chips that start inside a nominal range of Lt samples, at most P passes of k2Own chips
    int c2Lt = 0, c2nBlk = 0, c2NMom = 6;
    if (chip2) {
        const double thetaB = 6.283185307179586 * h->cfg.binHalfWidth / (double)h->C;
        const double halfMax = std::pow(2e-7 * 720.0, 1.0 / 6.0) / thetaB;   // 6-moment Taylor radius (samples)
        long long best = -1;
        for (int P = 1; P <= 64; ++P) {
            const int Lt = (int)std::floor((double)(k2Own * P - 2) / stepMax);
            if (Lt < 64 || 0.5 * Lt + 26.0 > halfMax) continue;
            const int nb = (S + Lt - 1) / Lt;
            if (nb > h->nBlkAlloc) continue;
            if (h->chip2PForce > 0 && P != h->chip2PForce) continue;
            const long long waves = (long long)nb * nChan * nWindows;
            const long long cost = 100 * ((waves + h->chip2Resident - 1) / h->chip2Resident) * P + 6 * nb;
            if (best < 0 || cost < best) { best = cost; c2Lt = Lt; c2nBlk = nb; }
        }
        if (c2Lt == 0) chip2 = false;
        else {
            const double thc = thetaB * (0.5 * c2Lt + 26.0);
            c2NMom = (std::pow(thc, 4) / 24.0 < 1e-7) ? 4 : 6;
        }
    }
    h->lastKernel = chip2 ? "bcs_bank_chip2_kernel" : chip ? "bcs_bank_chip_kernel" : use16 ? "bcs_bank16_kernel" : wide ? "bcs_bank_wide_kernel" : "bcs_bank_kernel";
    const dim3 grid(nBlk, nChan, nWindows), block(256);
    // single windows (<= 37 (window, channel) pairs) with a dense stage-1 kernel: no separate DC-sum launch, the
    // sums ride along in the bank kernel (FUSE) and the finalize kernel applies the mean
    const bool fuse = nWindows * nChan <= DPE_MAX_CHAN && !use16 && !wide && !chip && h->LH <= 16 && h->cfg.lagHalfWidth <= 32 && h->fuseAllowed;
    // chip2 batches: the DC sums ride in the chip2 launch (sum blocks interleaved ahead of the correlator blocks, dpe_bcs_chip2.h);
    // the parameter upload keeps a small kernel of its own (riding as well, every correlator block had to poll for it first thing:
    // 0.7035 against 0.696 ms per step)
    bool ride = chip2 && !dev && !h->graphs.capturing && vecOK && h->rideAllowed && nWindows >= h->rideMinW;
    int rideF = 0, rideSB = 0, rideGS = 1;
    if (ride) {   // one sum slot per correlator tile (the sum block then shares its tile's XCD): <= 64 slots, 31-bit sum fields
        if (c2nBlk > kSumSlots || c2Lt + 8 >= 32768) ride = false;
        else sumBlocks = c2nBlk;
    }
    if (ride) {
        const long long totalSum = (long long)nWindows * sumBlocks, NG = ((long long)c2nBlk * nWindows + 7) / 8;
        const int lookAhead = h->rideLA;   // windows between a sum block and the correlator blocks of its tile
        rideF = (int)(((long long)lookAhead * sumBlocks + 7) / 8 * 8);
        const long long groupsAvail = NG - ((long long)lookAhead * c2nBlk + 7) / 8 - 1;
        if (rideF >= totalSum || groupsAvail < 1) rideF = (int)((totalSum + 7) / 8 * 8);
        else {
            // sum blocks per group of 8 K correlator blocks at a STEADY lead (a lead that grows puts the samples a sum block fetched
            // out of its XCD's L2 before the correlator blocks of the same tile read them): whole eights per group, or one eight
            // every rideGS groups; what does not fit behind the look-ahead joins the blocks in front
            const long long rest = totalSum - rideF;
            if (rest >= 8 * groupsAvail) { rideSB = (int)(rest / (8 * groupsAvail)) * 8; rideGS = 1; }
            else { rideSB = 8; rideGS = (int)((8 * groupsAvail) / rest); }
            const long long cap = (long long)rideSB * (groupsAvail / rideGS);
            if (totalSum - cap > rideF) rideF = (int)((totalSum - cap + 7) / 8 * 8);
        }
        if (nWindows > h->rideW || sumBlocks > h->rideSlots)   // slots this launch reads that the last one did not write: clear the words
            DPE_CHECK_HIP(hipMemsetAsync(h->rideWord_d, 0, sizeof(unsigned long long) * (size_t)h->cfg.maxWindows * kSumSlots, stream));
        h->rideW = nWindows;
        h->rideSlots = sumBlocks;
        h->rideEpoch = h->rideEpoch % 3 + 1;
        if (upInSum) {
            h->prof.begin(0, stream);
            // (the same kernel clears the status word's "a block of this launch gave up waiting" bit: a block behind such a block skips its own wait)
            upload_params(h->chan_d, h->chanBase_hd + (h->chan_h - h->chanBase_h), sizeof(BcsChanDev) * nWindows * nChan, stream, h->status_d, 4);
            h->prof.end(0, stream);
            DPE_CHECK_HIP(hipEventRecord(h->stagingFree[h->slot], stream));
        }
    }
    const int sumSlotsUsed = fuse ? nBlk : sumBlocks;
    // a pending task of the device-resident channel manager: an extra block of the FUSE form's launch, a kernel of its own in
    // front of every other form
    const bool coRide = h->coPending && fuse && !h->useTable && !h->graphs.capturing;
    if (h->coPending && !coRide) hipLaunchKernelGGL(chm_k2_kernel, dim3(1), dim3(256), 0, stream, h->co);
    h->coPending = false;
    h->lastSumBlocks = sumSlotsUsed;
    if (!fuse && !ride && !rerun) {   // (a re-run uses the sums of the launch it follows: same window, same slots)
        h->prof.begin(0, stream);
        hipLaunchKernelGGL(bcs_sum_kernel, dim3(sumBlocks, nWindows), dim3(256), 0, stream, samples_dev,
                           (long long)windowStrideSamples, S, h->sums_d, (uint4 *)h->chan_d,
                           (const uint4 *)(h->chanBase_hd + (h->chan_h - h->chanBase_h)),
                           upInSum ? (int)(sizeof(BcsChanDev) * nWindows * nChan / 16) : 0);
        h->prof.end(0, stream);
        if (upInSum) DPE_CHECK_HIP(hipEventRecord(h->stagingFree[h->slot], stream));   // the staging block is free again
    }
#define DPE_LAUNCH_BANK5(LHV, NM, TB, FS, COV, GRID)                                                                     \
    hipLaunchKernelGGL((bcs_bank_kernel<LHV, NM, TB, FS, COV>), GRID, block, 0, stream, pb, inl, samples_dev, (long long)windowStrideSamples, \
                       S, nChan, h->nSub, tpb, nBlk, vecOK, sumBlocks, lagShift, h->chan_d, h->sums_d, h->chipTable_d, \
                       h->tTable_d, h->part_d, h->mom_d, h->momRep_d, h->sums_d, COV ? h->co : ChmKArgs{})
#define DPE_LAUNCH_BANK3(LHV, NM, TB)                                                                                    \
    do {                                                                                                                \
        if (fuse && LHV <= 16) {                                                                                        \
            if (coRide) DPE_LAUNCH_BANK5(LHV <= 16 ? LHV : 16, NM, false, true, true, dim3(nBlk + 1, nChan, nWindows)); \
            else DPE_LAUNCH_BANK5(LHV <= 16 ? LHV : 16, NM, TB, true, false, grid);                                     \
        } else DPE_LAUNCH_BANK5(LHV, NM, TB, false, false, grid);                                                       \
    } while (0)
#define DPE_LAUNCH_BANK2(LHV, NM)                           \
    do {                                                    \
        if (h->useTable) DPE_LAUNCH_BANK3(LHV, NM, true);   \
        else DPE_LAUNCH_BANK3(LHV, NM, false);              \
    } while (0)
    // Lag windows wider than +-32: further chunks of 65 lags, centred 65 samples apart -- the same kernels run
    // against the replica delayed by lagShift samples (centre chunk first: it makes the replica choice and the
    // Doppler bank).  One chunk when L <= 32.
    // (the chip kernel's lanes cover lagShift - 32 .. lagShift + 31: chunks of 64 lags, 64 apart)
    const int chunkLags = chip ? 64 : 65;
    const int nSideChunks = chip ? (h->cfg.lagHalfWidth > 31 ? (h->cfg.lagHalfWidth - 31 + 63) / 64 : 0)
                                 : (h->cfg.lagHalfWidth > 32 ? (h->cfg.lagHalfWidth - 32 + 64) / 65 : 0);
    for (int chunk = 0; chunk <= 2 * nSideChunks; ++chunk) {
    const int lagShift = chunk == 0 ? 0 : ((chunk + 1) / 2) * chunkLags * ((chunk & 1) ? 1 : -1);
    const bool c2 = chip2 && lagShift == 0;   // the second form produces the centre chunk; side chunks of a wider window use the first
    const int nMomUse = c2 ? c2NMom : chip ? chipNMom : h->nMom;
    const int nBlkUse = c2 ? c2nBlk : nBlk;
    const int momLenUse = c2 ? c2Lt : chip ? tpb * kPass : kSub;
    h->prof.begin(1, stream);
#define DPE_LAUNCH_BANK(LHV)            \
    do {                                \
        if (h->nMom == 4) DPE_LAUNCH_BANK2(LHV, 4); \
        else DPE_LAUNCH_BANK2(LHV, 6);  \
    } while (0)
    if (c2) {
        const int c2Groups = (c2nBlk * nWindows + 7) / 8;
        const dim3 cgrid(ride ? rideF + ((c2Groups + rideGS - 1) / rideGS) * (rideSB + rideGS * 8 * nChan) : c2Groups * 8 * nChan);   // one block per (tile, SV), tiles dealt to the XCDs (see the kernel)
#define DPE_LAUNCH_CHIP2(NM, LV, RD)                                                                                           \
    hipLaunchKernelGGL((bcs_bank_chip2_kernel<NM, LV, RD>), cgrid, dim3(64), 0, stream, pb, inl, samples_dev, (long long)windowStrideSamples, S, \
                       nChan, nWindows, c2Lt, c2nBlk, sumBlocks, h->chan_d, h->sums_d, h->chipTable_d, h->chipBits_d, h->part_d, h->mom_d,      \
                       h->rideWord_d, h->rideEpoch, rideF, rideSB, rideGS, h->rideSpin, h->status_d)
#define DPE_LAUNCH_CHIP2_L(LV)                                                                        \
    case LV:                                                                                          \
        if (ride) { if (c2NMom == 4) DPE_LAUNCH_CHIP2(4, LV, true); else DPE_LAUNCH_CHIP2(6, LV, true); }   \
        else { if (c2NMom == 4) DPE_LAUNCH_CHIP2(4, LV, false); else DPE_LAUNCH_CHIP2(6, LV, false); }      \
        break
        switch (c2L1) {
            DPE_LAUNCH_CHIP2_L(16); DPE_LAUNCH_CHIP2_L(17); DPE_LAUNCH_CHIP2_L(18); DPE_LAUNCH_CHIP2_L(19); DPE_LAUNCH_CHIP2_L(20);
            DPE_LAUNCH_CHIP2_L(21); DPE_LAUNCH_CHIP2_L(22); DPE_LAUNCH_CHIP2_L(23); DPE_LAUNCH_CHIP2_L(24);
        }
#undef DPE_LAUNCH_CHIP2_L
#undef DPE_LAUNCH_CHIP2
    } else if (chip) {
        const dim3 cgrid(((nBlk * nWindows + 7) / 8) * 8 * nChan);   // one block per (tile, SV), tiles dealt to the XCDs (see the kernel)
#define DPE_LAUNCH_CHIP(NM)                                                                                                    \
    hipLaunchKernelGGL((bcs_bank_chip_kernel<NM>), cgrid, dim3(64), 0, stream, pb, inl, samples_dev, (long long)windowStrideSamples, S, \
                       nChan, nWindows, h->nPassChip, tpb, nBlk, sumBlocks, lagShift, h->chipDbg, h->chan_d, h->sums_d, h->chipTable_d, h->part_d, h->mom_d)
        if (chipNMom == 4) DPE_LAUNCH_CHIP(4); else DPE_LAUNCH_CHIP(6);
#undef DPE_LAUNCH_CHIP
    } else if (use16) {
#define DPE_LAUNCH_B16(LHV, NM, TB)                                                                                    \
    hipLaunchKernelGGL((bcs_bank16_kernel<LHV, NM, TB>), dim3(((nBlk * nWindows + 7) / 8) * 8 * nChan), block, 0, stream, pb, inl, samples_dev, \
                       (long long)windowStrideSamples, S, nChan, nWindows, h->nSub, tpb, nBlk, vecOK, sumBlocks, h->chan_d, h->sums_d,         \
                       h->chipTable_d, h->tTable_d, h->part_d, h->mom_d)
#define DPE_LAUNCH_B16_2(LHV)                                                                           \
    do {                                                                                                \
        if (h->nMom == 4) { if (h->useTable) DPE_LAUNCH_B16(LHV, 4, true); else DPE_LAUNCH_B16(LHV, 4, false); } \
        else { if (h->useTable) DPE_LAUNCH_B16(LHV, 6, true); else DPE_LAUNCH_B16(LHV, 6, false); }         \
    } while (0)
        if (h->LH == 4) DPE_LAUNCH_B16_2(4); else DPE_LAUNCH_B16_2(8);
#undef DPE_LAUNCH_B16_2
#undef DPE_LAUNCH_B16
    } else if (wide) {
#define DPE_LAUNCH_WIDE(NM, TB)                                                                                    \
    hipLaunchKernelGGL((bcs_bank_wide_kernel<NM, TB>), grid, block, 0, stream, pb, inl, samples_dev, (long long)windowStrideSamples, \
                       S, nChan, h->nSub, tpb, nBlk, vecOK, sumBlocks, lagShift, h->chan_d, h->sums_d, h->chipTable_d, h->tTable_d, h->part_d, h->mom_d)
        if (h->nMom == 4) { if (h->useTable) DPE_LAUNCH_WIDE(4, true); else DPE_LAUNCH_WIDE(4, false); }
        else { if (h->useTable) DPE_LAUNCH_WIDE(6, true); else DPE_LAUNCH_WIDE(6, false); }
#undef DPE_LAUNCH_WIDE
    } else
    switch (h->LH) {
        case 4: DPE_LAUNCH_BANK(4); break;
        case 8: DPE_LAUNCH_BANK(8); break;
        case 16: DPE_LAUNCH_BANK(16); break;
        default: DPE_LAUNCH_BANK(32); break;
    }
#undef DPE_LAUNCH_BANK
#undef DPE_LAUNCH_BANK2
#undef DPE_LAUNCH_BANK3
#undef DPE_LAUNCH_BANK5
    h->prof.end(1, stream);
    h->prof.begin(2, stream);
    const int nBinBlk = (2 * h->cfg.binHalfWidth + 1 + 15) / 16;
    // batches: one fat block per (window, SV); few windows: 1 + nBinBlk short blocks (see the kernel);
    // side chunks of a wide lag window: one block per (window, SV), code-bank entries only
    // (crossover measured in round 2: 96 / 128 (window, SV) pairs are faster split, 192 / 256 / 384 fat -- H at 32 windows 0.0205 -> 0.0113 ms)
    bool fatFinalize = nBinBlk <= 4 && (long long)nChan * nWindows >= 192;
    if (h->fatForce >= 0) fatFinalize = nBinBlk <= 4 && h->fatForce == 1;   // (experiment builds)
    const dim3 fgrid((fatFinalize || lagShift != 0) ? 1 : 1 + nBinBlk, nChan, nWindows);
#define DPE_LAUNCH_FIN(NM, FS)                                                                                          \
    do {                                                                                                                \
        const int ng = (fatFinalize && lagShift == 0) ? nBinBlk : 1;                                                   \
        if (ng <= 1) DPE_LAUNCH_FIN_G(NM, FS, 1); else if (ng == 2) DPE_LAUNCH_FIN_G(NM, FS, 2);                        \
        else if (ng == 3) DPE_LAUNCH_FIN_G(NM, FS, 3); else DPE_LAUNCH_FIN_G(NM, FS, 4);                                \
    } while (0)
#define DPE_LAUNCH_FIN_G(NM, FS, NGV)                                                                                   \
    hipLaunchKernelGGL((bcs_finalize_kernel<NM, FS, NGV>), fgrid, dim3(256), 0, stream, pb, inl, S, nChan, chip ? nBlkUse : h->nSub, nBlkUse, h->LH, \
                       h->cfg.lagHalfWidth, h->cfg.binHalfWidth, wide ? 1 : 0, lagShift, momLenUse, chip ? 64 : 65, h->C, h->chan_d, h->part_d, h->mom_d,  \
                       h->codeBank_d, h->carrBank_d, h->info_d, h->cfg.maxChannels, h->momRep_d, h->sums_d, sumSlotsUsed)
    const bool fuseFin = fuse && lagShift == 0;
    if (nMomUse == 4) { if (fuseFin) DPE_LAUNCH_FIN(4, true); else DPE_LAUNCH_FIN(4, false); }
    else { if (fuseFin) DPE_LAUNCH_FIN(6, true); else DPE_LAUNCH_FIN(6, false); }
#undef DPE_LAUNCH_FIN
#undef DPE_LAUNCH_FIN_G
    h->prof.end(2, stream);
    }   // chunk
    const bool captured = h->graphs.capturing;
    DPE_REQUIRE(h->graphs.end(stream) == 0, "[BatchCorrScores] Update: hipGraph instantiate/launch failed");
    if (captured) DPE_CHECK_HIP(hipEventRecord(h->stagingFree[h->slot], stream));
    DPE_CHECK_HIP(hipGetLastError());
    if (hintedCall && (chip2 || chip)) {
        // the chip kernels ran on a promise nobody has checked yet on the host: the general kernels follow, guarded by the device
        // status word -- two launches whose blocks exit at once in all but the flagged windows (~1e-10 of them with honest callers)
        const char *ran = h->lastKernel;
        const int rc = bcs_update_impl(h, samples_dev, windowStrideSamples, nWindows, nChan, nullptr, stream_, 1);
        h->lastKernel = ran;
        return rc;
    }
    return 0;
}

int dpe_bcs_update(dpe_bcs *h, const int16_t *samples_dev, int64_t windowStrideSamples, int32_t nWindows,
                   int32_t nChan, const dpe_chan_start *chan_host, dpe_stream_t stream)
{
    DPE_REQUIRE(chan_host, "[BatchCorrScores] Update: null argument");
    return bcs_update_impl(h, samples_dev, windowStrideSamples, nWindows, nChan, chan_host, stream);
}

int dpe_bcs_update_dev(dpe_bcs *h, const int16_t *samples_dev, int32_t nChan, const dpe_bcs_ports_dev *ports, dpe_stream_t stream)
{
    using namespace dpe;
    DPE_REQUIRE(h && samples_dev && ports, "[BatchCorrScores] Update: null argument");
    DPE_REQUIRE(nChan >= 1 && nChan <= h->cfg.maxChannels, "[BatchCorrScores] Update: nChan %d out of range", nChan);
    DPE_REQUIRE(ports->codePhaseStart && ports->carrierPhaseStart && ports->codeFrequency && ports->carrierFrequency &&
                ports->cpElapsedStart && ports->cpReference && ports->validPRNs, "[BatchCorrScores] Update: a device port pointer is null");
    const BcsPortsDev p = {ports->codePhaseStart, ports->carrierPhaseStart, ports->codeFrequency, ports->carrierFrequency,
                           ports->cpElapsedStart, ports->cpReference, ports->validPRNs};
    if ((h->devHint & 1) && __atomic_load_n(h->hintViol_h, __ATOMIC_RELAXED)) h->devHint &= ~1;   // a broken promise withdraws the hint (see bcs_update_impl)
    const bool hinted = (h->devHint & 1) && h->chipOK && h->chipAllowed;
    const double nomStep = kFCA / h->cfg.samplingFrequency;
    hipLaunchKernelGGL(bcs_prep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, (int)nChan, h->cfg.samplingFrequency,
                       h->cfg.samplesPerWindow, h->chan_d, h->status_d, hinted ? (int)(1.0 / nomStep) : 0, nomStep * (1.0 + 1e-5), h->hintViol_hd);
    return bcs_update_impl(h, samples_dev, h->cfg.samplesPerWindow, 1, nChan, nullptr, stream);
}

int dpe_bcs_set_dev_hint(dpe_bcs *h, int32_t flags)
{
    DPE_REQUIRE(h, "[BatchCorrScores] set_dev_hint: null handle");
    h->devHint = flags;
    return 0;
}

int dpe_bcs_update_prepared(dpe_bcs *h, const int16_t *samples_dev, int32_t nChan, dpe_stream_t stream)
{
    DPE_REQUIRE(h && samples_dev, "[BatchCorrScores] Update: null argument");
    DPE_REQUIRE(nChan >= 1 && nChan <= h->cfg.maxChannels, "[BatchCorrScores] Update: nChan %d out of range", nChan);
    return bcs_update_impl(h, samples_dev, h->cfg.samplesPerWindow, 1, nChan, nullptr, stream);
}

int dpe_bcs_cotask_set(dpe_bcs *h, const void *args, size_t bytes)
{
    DPE_REQUIRE(h && args && bytes == sizeof(dpe::ChmKArgs), "[BatchCorrScores] cotask: bad arguments");
    DPE_REQUIRE(!h->coPending, "[BatchCorrScores] cotask: the previous task was never launched");
    memcpy(&h->co, args, bytes);
    h->coPending = true;
    return 0;
}

int dpe_bcs_cotask_flush(dpe_bcs *h, void *stream)
{
    DPE_REQUIRE(h, "[BatchCorrScores] cotask: null handle");
    if (!h->coPending) return 0;
    hipLaunchKernelGGL(dpe::chm_k2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, h->co);
    h->coPending = false;
    DPE_CHECK_HIP(hipGetLastError());
    return 0;
}

int dpe_bcs_hook_set_owner(dpe_bcs *h, dpe_owner_detach_fn detach, void *owner)
{
    DPE_REQUIRE(h, "[BatchCorrScores] hook: null handle");
    DPE_REQUIRE(!owner || !h->owner || h->owner == owner, "[BatchCorrScores] hook: the handle is attached to another channel manager");
    h->ownerDetach = owner ? detach : nullptr;
    h->owner = owner;
    if (!owner) h->coPending = false;   // (the manager is going away: whatever it parked points into its buffers)
    return 0;
}

int dpe_bcs_hook_get(dpe_bcs *h, dpe_bcs_hook *out)
{
    DPE_REQUIRE(h && out, "[BatchCorrScores] hook: null argument");
    out->chan_d = h->chan_d;
    out->status_d = h->status_d;
    out->fs = h->cfg.samplingFrequency;
    out->S = h->cfg.samplesPerWindow;
    out->maxChannels = h->cfg.maxChannels;
    const bool hinted = (h->devHint & 1) && h->chipOK && h->chipAllowed && !__atomic_load_n(h->hintViol_h, __ATOMIC_RELAXED);
    const double nomStep = dpe::kFCA / h->cfg.samplingFrequency;
    out->hintL1 = hinted ? (int)(1.0 / nomStep) : 0;
    out->hintStepMax = nomStep * (1.0 + 1e-5);
    out->hintViol = h->hintViol_hd;
    return 0;
}

int dpe_bcs_dev_status(dpe_bcs *h, int32_t *status, dpe_stream_t stream)
{
    DPE_REQUIRE(h && status && (h->lastDev || h->rideEpoch != 0), "[BatchCorrScores] dev_status: no device-parameter Update (and no batch whose DC sums ride in the chip2 launch) yet");
    DPE_CHECK_HIP(hipMemcpyAsync(status, h->status_d, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    DPE_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int dpe_bcs_set_graph(dpe_bcs *h, int32_t enable)
{
    DPE_REQUIRE(h, "[BatchCorrScores] set_graph: null handle");
    DPE_REQUIRE(!(enable && h->fftMode), "[BatchCorrScores] set_graph: the full-length FFT form (this handle's lag / bin windows) is not captured as a hipGraph");
    h->graphs.enabled = enable != 0;
    if (!enable) h->graphs.clear();
    return 0;
}

int dpe_bcs_outputs(dpe_bcs *h, const float **codeBank_dev, const float **carrBank_dev, int32_t *nLag, int32_t *nBin,
                    int64_t *numFFTPoints)
{
    DPE_REQUIRE(h, "[BatchCorrScores] outputs: null handle");
    if (codeBank_dev) *codeBank_dev = reinterpret_cast<const float *>(h->codeBank_d);
    if (carrBank_dev) *carrBank_dev = reinterpret_cast<const float *>(h->carrBank_d);
    if (nLag) *nLag = 2 * h->cfg.lagHalfWidth + 1;
    if (nBin) *nBin = 2 * h->cfg.binHalfWidth + 1;
    if (numFFTPoints) *numFFTPoints = h->C;
    return 0;
}

int dpe_bcs_allgather_banks(dpe_bcs *h, dpe_comm *c, float *codeAll_dev, float *carrAll_dev, dpe_stream_t stream)
{
    DPE_REQUIRE(h && c && codeAll_dev && carrAll_dev && h->lastW > 0, "[BatchCorrScores] allgather_banks: no update yet / null argument");
    const size_t K = h->cfg.maxChannels, nLag = 2 * (size_t)h->cfg.lagHalfWidth + 1, nBin = 2 * (size_t)h->cfg.binHalfWidth + 1;
    if (dpe_comm_allgather(c, h->codeBank_d, codeAll_dev, (int64_t)(h->lastW * K * nLag * sizeof(float2)), stream)) return -1;
    return dpe_comm_allgather(c, h->carrBank_d, carrAll_dev, (int64_t)(h->lastW * K * nBin * sizeof(float2)), stream);
}

int dpe_bcs_profile(dpe_bcs *h, int32_t enable, float *ms, int32_t *count)
{
    DPE_REQUIRE(h, "[BatchCorrScores] profile: null handle");
    float m[dpe::KernelProfiler::kSlots];
    int c[dpe::KernelProfiler::kSlots];
    h->prof.collect(m, c);
    for (int i = 0; i < 3; ++i) {
        if (ms) ms[i] = m[i];
        if (count) count[i] = c[i];
    }
    h->prof.enabled = enable != 0;
    h->prof.mask = enable <= 1 ? ~0u : (unsigned)(enable >> 1);   // enable = 1: every kernel; 2 * m: the slots of bit mask m
    return 0;
}

const char *dpe_bcs_stage1_kernel(dpe_bcs *h)
{
    return h ? h->lastKernel : "";
}

int dpe_bcs_read_info(dpe_bcs *h, int32_t *idxNext, int32_t *noFlipLarger, double *mean, dpe_stream_t stream)
{
    DPE_REQUIRE(h && h->lastW > 0, "[BatchCorrScores] read_info: no update yet");
    DPE_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    const int n = h->lastW * h->lastK;
    if (idxNext && h->lastDev) {   // the parameters never were on the host: fetch what bcs_prep_kernel derived
        std::vector<dpe::BcsChanDev> c((size_t)n);
        DPE_CHECK_HIP(hipMemcpy(c.data(), h->chan_d, sizeof(dpe::BcsChanDev) * n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) idxNext[i] = c[i].idxNext;
    } else if (idxNext) memcpy(idxNext, h->idxNext_h.data(), sizeof(int32_t) * n);
    if (noFlipLarger) DPE_CHECK_HIP(hipMemcpy(noFlipLarger, h->info_d, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (mean) {
        using dpe::kSumSlots;
        const int S = h->cfg.samplesPerWindow;
        const int sumBlocks = h->lastSumBlocks;
        std::vector<long long> s((size_t)2 * kSumSlots * h->lastW);
        DPE_CHECK_HIP(hipMemcpy(s.data(), h->sums_d, sizeof(long long) * s.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < 2 * h->lastW; ++i) {
            long long t = 0;
            for (int b = 0; b < sumBlocks; ++b) t += s[((size_t)(i / 2) * kSumSlots + b) * 2 + (i & 1)];
            mean[i] = (double)t / (double)(float)S;
        }
    }
    return 0;
}

int dpe_bcs_export_dense(dpe_bcs *h, int32_t window, double *codeScores_dev, double *carrScores_dev, dpe_stream_t stream_)
{
    using namespace dpe;
    DPE_REQUIRE(h && window >= 0 && window < h->lastW, "[BatchCorrScores] export_dense: bad window");
    hipStream_t stream = (hipStream_t)stream_;
    const int K = h->lastK, maxK = h->cfg.maxChannels, S = h->cfg.samplesPerWindow;
    const int nLag = 2 * h->cfg.lagHalfWidth + 1, nBin = 2 * h->cfg.binHalfWidth + 1;
    if (codeScores_dev) {
        DPE_CHECK_HIP(hipMemsetAsync(codeScores_dev, 0, sizeof(double2) * (size_t)K * S, stream));
        hipLaunchKernelGGL(bcs_export_kernel, dim3(1, K), dim3(256), 0, stream,
                           h->codeBank_d + (size_t)window * maxK * nLag, nLag, (long long)S, (long long)(S / 2), K, maxK,
                           h->cfg.lagHalfWidth, reinterpret_cast<double2 *>(codeScores_dev));
    }
    if (carrScores_dev) {
        DPE_CHECK_HIP(hipMemsetAsync(carrScores_dev, 0, sizeof(double2) * (size_t)K * h->C, stream));
        hipLaunchKernelGGL(bcs_export_kernel, dim3(1, K), dim3(256), 0, stream,
                           h->carrBank_d + (size_t)window * maxK * nBin, nBin, h->C, h->C / 2, K, maxK,
                           h->cfg.binHalfWidth, reinterpret_cast<double2 *>(carrScores_dev));
    }
    DPE_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
