// dpe_bcs_chip2.h -- stage 1 for high sampling rates, second form: LANES <-> CHIPS in the prefix stage as well.
// Included by dpe_bcs.hip after dpe_bcs_chip.h (whose helpers it shares).
//
// Reference semantics (cudarecv/modules/src/batchcorrscores.cu): corr[l] = sum_n b[n] r[(n-l) mod S] with b = raw * wipe
// (:277-305, :402) and r the +-1 code replica (:323-372), both nav-bit sides (:237-258); carrier path
// c[n] = (raw[n] - mean) wipe[n] r[n] -> zero-padded C-point FFT (:422-452, :1179).
//
// bcs_bank_chip_kernel (dpe_bcs_chip.h) gives every lane 17 consecutive samples and then needs three more mappings per
// pass (chips for the boundary positions, chips again for the Doppler moments -- which re-read every prefix value from
// LDS for the first moment of each chip -- and lags for the gather).  Here a lane owns one CHIP of the replica:
//   * its L1 or L1 + 1 samples (L1 = floor(fs / fc), 16 .. 24) are accumulated in the lane's own rotating frame,
//     u_i = u_(i-1) + raw_i T_(i-CI)  (T_j = exp(-j 2 pi j fi / fs) about the chip's middle sample CI: block constants in
//     scalar registers, two packed FMAs per sample), together with the running sum of the u_i: zeroth AND first moment of
//     the chip come out of registers (Abel summation) -- no second pass over LDS;
//   * one 64-lane scan of the chip totals gives every chip its offset; Q = w_chip u_i + offset goes to LDS (two packed FMAs
//     and one ds_write_b64 per sample) as the pass-local prefix array, centred (Q runs from -T/2 to +T/2: half the magnitude,
//     half the fp32 rounding of the boundary differences under a DC offset);
//   * the chip boundaries ARE the lanes' first samples: positions, replica values and sign changes need no second mapping;
//     lag sums as in the first form, corr_pass[l] = (r_first + r_last) T/2 + sum_i J_i Q[e_i + l] with lanes <-> the 64 lags.
// A pass = 58 chips (lanes 3 .. 60); lanes 0 .. 2 and 61 .. 63 carry the chips either side of the pass whose boundaries the
// +-32-lag window still reaches (two before and three after are needed where the window wraps around circularly and the
// partial chips at its ends are short) -- they own no samples.  The circular wrap and the nav-bit boundary need no special
// path: a lane beyond the window's last chip takes the chip the circular continuation puts there, and the two nav-bit
// sides are two masked copies of the replica values (the host only selects this kernel when the nav-bit boundary falls on
// a chip boundary, which it does unless fp64 rounding separates BCS_NavBitBoundary :247-253 from the chip index :347-349).
// A wave walks the passes of one TILE: the chips that start inside [blk Lt, (blk+1) Lt); the tile's moment block is taken
// about the centre of that nominal range, so bcs_finalize_kernel sees the layout of the first form (momLen = Lt).
// Per pass of ~1420 samples: 478 VALU + 71 LDS instructions (SQ counters) against 448 + 78 per 1088 samples of the first form.
#pragma once

namespace dpe {

#ifndef DPE_C2_WAVES
#define DPE_C2_WAVES 3
#endif
constexpr int k2Own0 = 3;      // first owner lane
#ifndef DPE_C2_OWN
#define DPE_C2_OWN 58
#endif
constexpr int k2Own = DPE_C2_OWN;   // owner lanes per pass: lanes 3 .. 60 (measured at H: 58 owners at 11 blocks per CU 0.593 ms per 128 windows,
                                    // 57 at 12 blocks 0.604, 50 at 13 blocks 0.635 -- lanes that own samples matter more than resident waves)
constexpr int k2MaxL1 = 24;    // L1 = floor(fs / fc) <= 24: chips of at most 25 samples
constexpr int k2MinL1 = 16;    // the margins reach +-32 samples: two regular chips must cover them
constexpr int k2Pad = 64;
constexpr int k2QLen = k2Pad + k2Own * (k2MaxL1 + 1) + 1 + 64;
constexpr int k2Round = 16;    // list entries per gather round: two lane halves x eight entries (the list is zero-padded to a whole round)
constexpr int k2BitWords = 40; // words per PRN of the chip-sign bit table (dpe_bcs_create): bit b = [chip (b mod 1023) is +1], b < 1280

// a * t + c and a * conj(t) + c with t a block constant in scalar registers (the lane-frame twiddles): two packed FMAs each
#define DPE_C2_TWC "s"
__device__ __forceinline__ f2 tw_mul_add(f2 a, f2 t, f2 c)
{
    f2 x, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(x) : "v"(a), DPE_C2_TWC(t), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), DPE_C2_TWC(t), "v"(x));
    return r;
}
__device__ __forceinline__ f2 tw_conj_mul_add(f2 a, f2 t, f2 c)
{
    f2 x, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(x) : "v"(a), DPE_C2_TWC(t), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), DPE_C2_TWC(t), "v"(x));
    return r;
}

// One ds_read_b64 that stays one: the compiler fuses two 8-byte LDS reads off one address register into ds_read2_b64, which the LDS
// serves as two accesses of 4 x 16 lanes (8 array cycles per wave-instruction, 128 B/clk); a plain ds_read_b64 of 32 lanes x 8
// consecutive bytes covers all 64 banks once -- 2 array cycles, 256 B/clk (MI355X_MICROARCH.md, LDS table).  The result is NOT
// tracked by the compiler's s_waitcnt insertion: lds_wait<N>() below orders the uses.
template <int OFF>
__device__ __forceinline__ f2 lds_read_b64(unsigned addr)
{
    f2 r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// s_waitcnt lgkmcnt(N) that the uses of a .. h cannot be moved above (LDS reads return in order)
template <int N>
__device__ __forceinline__ void lds_wait(f2 &a, f2 &b, f2 &c, f2 &d, f2 &e, f2 &f, f2 &g, f2 &h)
{
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N));
}

// +1.0f / -1.0f per lane from a 64-bit lane mask held in scalar registers: one v_cndmask, the mask IS the condition operand
__device__ __forceinline__ float mask_pm1(unsigned long long m)
{
    float r;
    asm("v_cndmask_b32_e64 %0, -1.0, 1.0, %1" : "=v"(r) : "s"(m));
    return r;
}

// RIDE: the DC sums of the batch are computed by extra one-wave blocks of THIS launch instead of a kernel in front of it
// (bcs_sum_kernel reads all samples at the HBM rate while the chip kernel, which is arithmetic-bound, waits behind it: 0.04 of a
// 0.72 ms step at config H).  The launch's linear block index interleaves them with the correlator blocks -- rideF sum blocks
// first, then rideSB of them in front of every rideGS groups of 8 K correlator blocks (both multiples of 8, so a correlator block's
// index mod 8 is still its XCD) -- a steady ~4 windows ahead of their consumers (the host's choice of rideF / rideSB / rideGS), so that
// a correlator block all but never waits.  Hand-over: ONE 64-bit word per slot of kSumSlots, {epoch : 2, I sum : 31, Q sum : 31} (the host takes this form only
// when a slot holds < 32768 samples), written with an agent-scope atomic store and read with agent-scope atomic loads (lane <->
// slot): tag and data arrive together, in one round trip that the block issues first thing and consumes after its set-up.  The
// sum block also leaves the int64 sums in the ordinary slots for the kernels behind this launch.  Epochs cycle 1 .. 3 and every
// launch rewrites every word it will read; the host clears the words when the set of slots grows, so no older word can carry
// the current epoch.  Forward progress: work groups are dispatched in index order per XCD and a sum block waits for nothing, so the
// lowest-index undispatched sum block is never behind a full house of waiting blocks -- a property of the dispatcher that HIP does
// not promise, so nothing depends on it for correctness: a correlator block whose wait exceeds `rideSpin` polls (200 000 by default;
// DPE_BCS_RIDE_SPIN at create, the tests force 1) adds up its window's samples ITSELF -- slow (one wave reads the whole window) and
// right: the same exact integer sums.  It also sets bit 2 of *status (this launch: cleared by the parameter-upload kernel in front of
// every such launch; the blocks behind it then skip their own wait when their word is not there yet) and bit 4 (sticky since create:
// the diagnostic dpe_bcs_dev_status reports).
constexpr int kRideSpinDefault = 200000;
#ifndef DPE_RIDE_LOADS
#define DPE_RIDE_LOADS 16
#endif
constexpr int kRideLoads = DPE_RIDE_LOADS;
   // loads in flight per lane of a sum block (24 or 32 push the kernel over its 170-register budget: scratch, 0.675 ms)
__device__ __forceinline__ void ride_sum_block(const int16_t *__restrict__ iq, long long winStride, int S, int nW, int nSumBlk, int Lt, int sidx,
                                               long long *__restrict__ sums, unsigned long long *__restrict__ rideWord, unsigned epoch)
{
    const int w = sidx / nSumBlk, b = sidx - w * nSumBlk;
    if (w >= nW) return;
    const int lane = threadIdx.x;
    const int *x = reinterpret_cast<const int *>(iq + (size_t)w * winStride * 2);   // (the host takes this form only for 16-byte aligned windows)
    const int4 *x4 = reinterpret_cast<const int4 *>(x);
    // slot b = the samples of correlator tile b (Lt of them, cut at multiples of four): sum block (w, b) has the linear index of
    // tile (w, b), mod 8 -- it runs on the XCD whose L2 that tile's K correlator blocks will read the same samples through
    const int n4 = S >> 2;
    const int lo = b == 0 ? 0 : (int)(((long long)b * Lt + 3) >> 2);
    int hi = b == nSumBlk - 1 ? n4 : (int)(((long long)(b + 1) * Lt + 3) >> 2);
    if (hi > n4) hi = n4;
    int sI = 0, sQ = 0;   // a block adds < 32768 samples
    auto add4 = [&](const int4 v) {
        sI += (short)(v.x & 0xFFFF) + (short)(v.y & 0xFFFF) + (short)(v.z & 0xFFFF) + (short)(v.w & 0xFFFF);
        sQ += (v.x >> 16) + (v.y >> 16) + (v.z >> 16) + (v.w >> 16);
    };
    // kRideLoads loads in flight per lane, each under its own range check: the block is a short latency chain, and it holds a
    // correlator block's slot for that time
    int n0 = lo;
    // (round 4, measured and dropped: twelve of a chunk's rounds straight into the block's LDS with global_load_lds_dwordx4 -- three memory
    // round trips per sum block instead of five, but the LDS-direct writes land in an array the correlator waves keep 65 % busy: config H
    // step 0.7135 against 0.700 ms.)
    for (; n0 + kRideLoads * 64 <= hi; n0 += kRideLoads * 64) {   // whole rounds (wave-uniform bound)
        int4 v[kRideLoads];
#pragma unroll
        for (int j = 0; j < kRideLoads; ++j) v[j] = x4[n0 + lane + 64 * j];
#pragma unroll
        for (int j = 0; j < kRideLoads; ++j) add4(v[j]);
    }
    {   // the rest: one predicated round (a slot is a tile: its length is no multiple of the round)
        int4 v[kRideLoads];
#pragma unroll
        for (int j = 0; j < kRideLoads; ++j) v[j] = (n0 + lane + 64 * j < hi) ? x4[n0 + lane + 64 * j] : int4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < kRideLoads; ++j) add4(v[j]);
    }
    if (b == nSumBlk - 1)
        for (int m = (n4 << 2) + lane; m < S; m += 64) { const int v = x[m]; sI += (short)(v & 0xFFFF); sQ += v >> 16; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sI += __shfl_xor(sI, off, 64);
        sQ += __shfl_xor(sQ, off, 64);
    }
    if (lane == 0) {
        long long *o = sums + ((size_t)w * kSumSlots + b) * 2;
        o[0] = sI; o[1] = sQ;
        const unsigned long long word = ((unsigned long long)epoch << 62) | ((unsigned long long)((unsigned)sI & 0x7FFFFFFFu) << 31) | (unsigned long long)((unsigned)sQ & 0x7FFFFFFFu);
        __hip_atomic_store(rideWord + (size_t)w * kSumSlots + b, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the consumer's side: window_mean (dpe_bcs.hip) from the words that blocks of the same launch publish; `first` = the word this
// lane fetched when the block started.  Gives up after rideSpin polls and then sums the window's samples itself (see above).
__device__ __forceinline__ void ride_window_mean(const unsigned long long *__restrict__ rideWord, unsigned long long first, unsigned epoch, int *__restrict__ status,
                                                 int rideSpin, const int *__restrict__ x, int w, int nSumBlk, int S, float &mRe, float &mIm)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long *wd = rideWord + (size_t)w * kSumSlots + lane;
    unsigned long long v = first;
    bool gaveUp = false;
    for (int it = 0;; ++it) {
        if (rideSpin >= 0 && __ballot(lane < nSumBlk && (unsigned)(v >> 62) != epoch) == 0ull) break;   // (rideSpin < 0, tests: every block takes the fallback)
        if (it >= rideSpin || (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4)) {
            if (lane == 0) atomicOr(status, 4 | 16);
            gaveUp = true;
            break;
        }
        __builtin_amdgcn_s_sleep(32);
        if (lane < nSumBlk) v = __hip_atomic_load(wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    long long tI = 0, tQ = 0;
    if (!gaveUp) {
        if (lane < nSumBlk) {   // 31-bit two's-complement fields
            tI = (long long)((int)((unsigned)(v >> 31) << 1) >> 1);
            tQ = (long long)((int)((unsigned)v << 1) >> 1);
        }
    } else {
        // the window's sums by this wave alone (the riding form is only taken for 16-byte aligned windows)
        const int4 *x4 = reinterpret_cast<const int4 *>(x);
        const int n4 = S >> 2;
        for (int n0 = 0; n0 < n4; n0 += 64 * 256) {   // partial sums of < 64 k samples per lane stay inside 32 bits
            int sI = 0, sQ = 0;
            const int hi = n0 + 64 * 256 < n4 ? n0 + 64 * 256 : n4;
            for (int n = n0 + lane; n < hi; n += 64) {
                const int4 q = x4[n];
                sI += (short)(q.x & 0xFFFF) + (short)(q.y & 0xFFFF) + (short)(q.z & 0xFFFF) + (short)(q.w & 0xFFFF);
                sQ += (q.x >> 16) + (q.y >> 16) + (q.z >> 16) + (q.w >> 16);
            }
            tI += sI; tQ += sQ;
        }
        for (int m = (n4 << 2) + lane; m < S; m += 64) { const int q = x[m]; tI += (short)(q & 0xFFFF); tQ += q >> 16; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tI += __shfl_xor(tI, off, 64);
        tQ += __shfl_xor(tQ, off, 64);
    }
    mRe = (float)((double)tI / (double)(float)S);
    mIm = (float)((double)tQ / (double)(float)S);
}

template <int kNMom, int L1, bool RIDE = false>
__global__ __launch_bounds__(64, DPE_C2_WAVES) void bcs_bank_chip2_kernel(BcsParamBlock pb, int inl, const int16_t *__restrict__ iq, long long winStride,
                                                               int S, int K, int nW, int Lt, int nBlk, int nSumBlk,
                                                               const BcsChanDev *__restrict__ chan,
                                                               const long long *__restrict__ sums,
                                                               const int8_t *__restrict__ chipTable,
                                                               const uint32_t *__restrict__ chipBits,
                                                               float2 *__restrict__ part, float2 *__restrict__ mom,
                                                               unsigned long long *__restrict__ rideWord, unsigned epoch, int rideF, int rideSB, int rideGS,
                                                               int rideSpin, int *__restrict__ status)
{
    constexpr int NL = 65;   // partial layout shared with the other stage-1 kernels: entry j <-> lag j - 32 (j = 64 unused)
    __shared__ float2 sQ[k2QLen];
    __shared__ float2 sList[64 + k2Round];   // flips of a pass: {J, byte offset into sQ}, zero-padded to a whole round

    // Block -> (window, tile, SV), XCD-aware as in the first form: the K blocks of a tile are congruent mod 8
    const int lane = threadIdx.x;
    int bx = blockIdx.x;
    if constexpr (RIDE) {
        int sidx = -1;
        if (bx < rideF) sidx = bx;
        else {
            const int bankPer = rideGS * 8 * K, per = rideSB + bankPer, r = bx - rideF, sg = r / per, o = r - sg * per;
            if (o < rideSB) sidx = rideF + sg * rideSB + o;
            else bx = sg * bankPer + (o - rideSB);
        }
        if (sidx >= 0) {
            ride_sum_block(iq, winStride, S, nW, nSumBlk, Lt, sidx, const_cast<long long *>(sums), rideWord, epoch);
            return;
        }
    }
    const int slot = bx >> 3, k = slot % K, tg = (slot / K) * 8 + (bx & 7);
    if (tg >= nBlk * nW) return;
    const int w = tg / nBlk, blk = tg - w * nBlk;
    (void)pb;
    unsigned long long rideFirst = 0ull;
    if constexpr (RIDE)
        if (lane < nSumBlk) rideFirst = __hip_atomic_load(rideWord + (size_t)w * kSumSlots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const BcsChanDev ch = params_ptr(chan, inl)[(size_t)w * K + k];
    const int8_t *chips = chipTable + (ch.prn - 1) * 1024;
    // the same chips as sign bits: lane l keeps word l of the PRN's (periodically extended) bit table for the whole tile, and a
    // pass fetches the three words it needs with v_readlane at a scalar index -- no memory operation inside the pass loop.
    // (A first version read them with scalar loads: the s_waitcnt that follows an s_load is lgkmcnt(0), which also drains the
    // wave's LDS queue -- 0.81 against 0.57 ms per 128 windows at H.)
    const int bitsV = (int)(lane < k2BitWords ? chipBits[(ch.prn - 1) * k2BitWords + lane] : 0u);
    const int *x = reinterpret_cast<const int *>(iq + (size_t)w * winStride * 2);

    // chip index of replica index m (phases are >= 0: the truncating conversion is the floor, :347-349)
    auto chip_at = [&](int m) -> int { return (int)fma((double)m, ch.codeStep, ch.rc); };
    // first replica index of chip c: min { m : chip_at(m) >= c }.  The estimate ceil((c - rc) fs / fc) is off the exact threshold
    // by < 1e-9 samples (three roundings at magnitude <= 2^20), so the two-sided check against the reference's own expression is
    // only needed when the estimate lies within 1e-6 of an integer -- a wave-uniform branch that is all but never taken
    auto first_index = [&](int c) -> int {
        const double xq = ((double)c - ch.rc) * ch.invStep;
        const double md = ceil(xq);
        int m = (int)md;
        if (__ballot(fabs((md - xq) - 0.5) > 0.499999) != 0ull) {
            if ((int)fma(md - 1.0, ch.codeStep, ch.rc) >= c) m -= 1;
            else if ((int)fma(md, ch.codeStep, ch.rc) < c) m += 1;
        }
        return m;
    };
    static_assert(L1 >= k2MinL1 && L1 <= k2MaxL1, "chips of L1 or L1 + 1 samples, L1 = floor(fs / fc) (the host checks every channel)");
    const int c0 = __builtin_amdgcn_readfirstlane(chip_at(0)), cEnd = __builtin_amdgcn_readfirstlane(chip_at(S - 1));
    const int Nc = cEnd - c0 + 1;                                            // chips (partial ones included) of the window
    // the tile's chips: those that START inside [blk Lt, (blk + 1) Lt) -- the window's first, partial chip starts at 0
    const int cLo = blk == 0 ? c0 : __builtin_amdgcn_readfirstlane(chip_at(blk * Lt - 1)) + 1;
    const int cHi = blk == nBlk - 1 ? cEnd + 1 : __builtin_amdgcn_readfirstlane(chip_at((blk + 1) * Lt - 1)) + 1;
    const int nPassT = (cHi - cLo + k2Own - 1) / k2Own;

    // ---- once per block: the SV's twiddles.  The lane frame has its origin at sample CI of the chip, so the samples either
    // side of it share T_j = exp(-j 2 pi j fi / fs), j = 1 .. CI (conjugated before the origin, none at it); they stay in
    // scalar registers for the whole tile (as LDS reads inside the pass they were twelve exposed round trips: 0.594 -> 0.565 ms
    // per 128 windows at H; in vector registers 0.567 at 160 instead of 136 VGPRs).  thA / thB: origin -> centre of a chip of
    // L1 / L1 + 1 samples.
    constexpr int CI = (L1 + 1) / 2;
    f2 tw[CI + 1], thA, thB;
    {
        const double a = lane < 26 ? (double)lane : (lane == 26 ? 0.5 * (double)(L1 - 1) - (double)CI : 0.5 * (double)L1 - (double)CI);
        double ph = a * ch.carrStep;
        ph -= floor(ph);
        const f2 t = wipe_seed((float)ph);   // lane j: T_j; lanes 26, 27: the two centre twiddles
#pragma unroll
        for (int j = 0; j <= CI; ++j) tw[j] = f2{readlane_f(t.x, j), readlane_f(t.y, j)};
        thA = f2{readlane_f(t.x, 26), readlane_f(t.y, 26)};
        thB = f2{readlane_f(t.x, 27), readlane_f(t.y, 27)};
    }
    float mRe, mIm;   // (behind the block's set-up: with RIDE the word fetched first thing has arrived by now)
    if constexpr (RIDE) ride_window_mean(rideWord, rideFirst, epoch, status, rideSpin, x, w, nSumBlk, S, mRe, mIm);
    else window_mean(sums, w, nSumBlk, S, mRe, mIm);
    const f2 meanv = f2{mRe, mIm};
    float2 *momOut = mom + ((((size_t)w * K + k) * 2) * nBlk + blk) * kNMom;   // [side][nBlk][kNMom]
    const size_t momSide = (size_t)nBlk * kNMom;

    // gather geometry (round 6): lane (h, j) = (lane >> 5, lane & 31) sums the two lags j - 32 and j over the list entries of its
    // HALF h.  The 32 lanes of a half read 256 consecutive bytes per LDS access -- all 64 banks once: a plain ds_read_b64 then takes
    // the LDS array 2 cycles per wave-instruction (256 B/clk).  Rounds 4-5 gave a lane four lags (16 lanes x 128 bytes per list
    // entry), which the compiler fetched with ds_read2_b64: 8 array cycles per instruction, 128 B/clk -- the gather's reads were
    // 128 of a pass's ~400 LDS-array cycles; now 64.
    const int lh = lane >> 5, lj = lane & 31;
    const int qLaneBytes = 8 * ((k2Pad - 32) + lj);   // byte offset in sQ of the lane's first lag for a boundary at the pass start
    const float phi = (float)(6.283185307179586476925286766559 * ch.carrStep);   // wipe-off phase step per sample (rad)
    // DC-mean sums over a chip of len samples about its centre: G0 = sum exp(-j phi d) (real), j G1 = sum d exp(-j phi d)
    auto mean_sums = [&](float fl, float &G0, float &G1) {
        const float l2 = fl * fl, h2 = 0.25f * phi * phi;
        G0 = fl * (1.f - (l2 - 1.f) * h2 * (1.f / 6.f) * (1.f - (3.f * l2 - 7.f) * h2 * (1.f / 60.f)));
        G1 = -phi * fl * (l2 - 1.f) * (1.f / 12.f) * (1.f - phi * phi * (3.f * l2 - 7.f) * (1.f / 120.f));
    };
    float G0a, G1a, G0b, G1b;
    mean_sums((float)L1, G0a, G1a);
    mean_sums((float)(L1 + 1), G0b, G1b);
    const float xOrigin = 0.5f * (float)(Lt - 1);   // moment abscissa origin relative to the tile's nominal start blk Lt

    // Lag sums: ONE accumulator set, for the nav-bit side being accumulated (accSide) -- two lags per lane, one half of the flips per
    // 32 lanes, the end terms shared out between the halves.  A change of side (twice per tile at most, in the tile that holds
    // the nav-bit boundary and in those whose margins wrap around the window's ends) goes through spill(): the two halves
    // are added up and lanes 0 .. 31 add them into the block partial in global memory.
    f2 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = f2{0.f, 0.f};
    int accSide = 0, spilled = 0;
    float2 *partOut = part + ((((size_t)w * K + k) * nBlk + blk) * 2) * NL;   // [side][NL]
    auto spill = [&](int side) {
        float2 *o = partOut + side * NL;
        const bool again = (spilled >> side) & 1;
        spilled |= 1 << side;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f2 v = acc[t];
            v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64);
            if (lane < 32) {
                if (again) { const float2 old = o[lj + 32 * t]; v += f2{old.x, old.y}; }   // (the same lane wrote it: program order)
                o[lj + 32 * t] = make_float2(v.x, v.y);
            }
            acc[t] = f2{0.f, 0.f};
        }
        if (lane == 0) o[64] = make_float2(0.f, 0.f);
    };
    f2 M[kNMom];                                  // per-lane moment sums of the current side
#pragma unroll
    for (int p = 0; p < kNMom; ++p) M[p] = f2{0.f, 0.f};
    int curSide = 0, flushed = 0;
    auto flush = [&](int side) {   // reduce the lanes' moment sums and store the (tile, side) block
        flushed |= 1 << side;
        float mm[2 * kNMom];
#pragma unroll
        for (int p = 0; p < kNMom; ++p) { mm[2 * p] = M[p].x; mm[2 * p + 1] = M[p].y; }
        dpp_sum_lane63(mm);
        if (lane == 63) {
#pragma unroll
            for (int p = 0; p < kNMom; ++p) momOut[side * momSide + p] = make_float2(mm[2 * p], mm[2 * p + 1]);
        }
#pragma unroll
        for (int p = 0; p < kNMom; ++p) M[p] = f2{0.f, 0.f};
    };
    // a chip's zeroth / first moment about its own centre xb:  M_p += xb^p E0 + p xb^(p-1) E1 = xb^(p-1) (xb E0 + p E1)
    // (two packed FMAs per order on three scalar powers; the running-product form it replaces took four)
    auto add_moments = [&](f2 E0, f2 E1, float xb) {
        M[0] += E0;
        const f2 xE0 = E0 * xb;
        if (kNMom > 1) M[1] += xE0 + E1;
        float xp = xb;
#pragma unroll
        for (int p = 2; p < kNMom; ++p) {
            const float pf = (float)p;
            const f2 F = __builtin_elementwise_fma(E1, f2{pf, pf}, xE0);
            M[p] = __builtin_elementwise_fma(F, f2{xp, xp}, M[p]);
            xp *= xb;
        }
    };

    // ---- lane state of a pass: chip c = cBase - k2Own0 + lane, circularly continued beyond the window's ends.
    // (Chip starts come from the fp64 expression per lane.  An integer DDA -- 32.32 fixed-point p(c) = (c - rc) fs / fc per lane,
    // the fp64 check only for lanes within 2^-24 of a sample boundary -- was measured in round 3: 0.608 against 0.601 ms per 128
    // windows at H.  fp64 FMAs issue at the full vector rate on this part and the two 64-bit multiply-adds of the DDA do not.)
    int eN = 0, offN = 0, lenN = 0, eAN = 0, eBN = 0;
    unsigned long long rMaskN = 0ull;   // wave-uniform: bit l = [lane l's chip is +1]
    unsigned long long sideMaskN = 0ull;   // wave-uniform: bit l = [lane l's chip lies behind the nav-bit boundary]
    bool ownN = false, edgeN = false;
    int raw[L1], rawX = 0;
    auto setup = [&](int cBase, auto pureTag) {
        const int cF = cBase - k2Own0;   // lane 0's chip
        const int c = cF + lane;
        int e, off = 0;
        if (cF > c0 && cF + 63 <= cEnd) {
            // every chip of the pass lies inside the window (all passes but its first and last): no circular continuation, and
            // the 64 chip signs are bits [cF mod 1023, + 64) of the PRN's periodically extended sign table -- three scalar
            // v_readlane and two scalar shifts instead of a modulo, an address and a byte load per lane
            e = first_index(c);
            const int st = cF % kLCA, wi = st >> 5, sh = st & 31;
            const unsigned w0 = (unsigned)__builtin_amdgcn_readlane(bitsV, wi), w1 = (unsigned)__builtin_amdgcn_readlane(bitsV, wi + 1),
                           w2 = (unsigned)__builtin_amdgcn_readlane(bitsV, wi + 2);
            const unsigned long long lo = ((unsigned long long)w1 << 32) | (unsigned long long)w0;
            rMaskN = (lo >> sh) | ((((unsigned long long)w2) << 32) << (32 - sh));
        } else {
            int cw = c;
            if (c < c0) { cw = c + Nc; off = -S; }
            else if (c > cEnd) { cw = c - Nc; off = S; }
            e = (cw == c0) ? 0 : first_index(cw);   // the window's first chip begins before sample 0: clipped
            rMaskN = __ballot(chips[cw % kLCA] > 0);
        }
        eN = e + off;
        offN = off;
        if constexpr (!decltype(pureTag)::value)
            sideMaskN = ch.hasFlip ? __ballot(e >= ch.idxNext) : 0ull;   // (e = the chip's replica index: the nav-bit side, :352-367)
        const int nOwn = (cHi - cBase < k2Own) ? cHi - cBase : k2Own;
        const int eNext = __builtin_amdgcn_update_dpp(0, eN, 0x130, 0xf, 0xf, true);   // wave_shl:1 -> lane + 1
        ownN = lane >= k2Own0 && lane < k2Own0 + nOwn;
        lenN = ownN ? eNext - eN : 0;
        eAN = __builtin_amdgcn_readlane(eN, k2Own0);
        eBN = __builtin_amdgcn_readlane(eN, k2Own0 + nOwn);
        // regular pass: every owner has L1 or L1 + 1 samples and may read L1 + 1 of them without leaving the window
        edgeN = __ballot(ownN && ((lenN != L1 && lenN != L1 + 1) || eN + L1 + 1 > S)) != 0ull;
    };
    auto fetch = [&]() {
        if (!edgeN) {
            int eb = ownN ? eN : 0;
            const int *px = x + eb;
#pragma unroll
            for (int i = 0; i < L1; ++i) raw[i] = px[i];
            rawX = px[L1];
        } else {
            asm volatile("" ::: "memory");   // (keeps the two forms apart: merged, every load carries its own selected address)
            const int eb = ownN ? eN : 0;
#pragma unroll
            for (int i = 0; i < L1; ++i) raw[i] = x[(eb + i < S) ? eb + i : S - 1];
            rawX = x[(eb + L1 < S) ? eb + L1 : S - 1];
            asm volatile("" ::: "memory");
        }
    };
    // The lag sums over the list of a pass.  Two list forms, both zero-padded to whole rounds of 16 entries, half h of the lanes
    // taking the entries 8 h .. 8 h + 7 of a round as four PAIRS of consecutive flips:
    //  * one nav-bit side in reach (all passes but a few): the flips alternate in sign, J_(2p+1) = -J_2p, so a pair contributes
    //    J_2p (Q[e_2p + l] - Q[e_(2p+1) + l]) -- the list holds byte offsets only (int), the differences are summed and J of the pass's
    //    first flip is applied once.  Per round and lane: two 16-byte list reads, 8 addresses, 16 ds_read_b64 (volatile: kept apart --
    //    fused into ds_read2_b64 they are served at half the rate), 8 packed subtractions + 8 packed additions.
    //  * general (a pass about the nav-bit boundary or the window's ends): entries {J, offset} as before, two packed FMAs per pair and lag.
    // Padding entries come in pairs with equal offsets: their difference is exactly zero.
    const unsigned qLaneAddr = (unsigned)(size_t)(__attribute__((address_space(3))) char *)reinterpret_cast<char *>(sQ) + (unsigned)qLaneBytes;
    auto gather_pairs = [&](f2 (&d2)[2], int nList) {
        const int *lst = reinterpret_cast<const int *>(sList);
        for (int i0 = 0; i0 < nList; i0 += k2Round) {
            const int4 oA = *reinterpret_cast<const int4 *>(lst + i0 + 8 * lh), oB = *reinterpret_cast<const int4 *>(lst + i0 + 8 * lh + 4);
            const int ofs[8] = {oA.x, oA.y, oA.z, oA.w, oB.x, oB.y, oB.z, oB.w};
            f2 qv[8][2];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const unsigned a = qLaneAddr + (unsigned)ofs[s];
                qv[s][0] = lds_read_b64<0>(a);
                qv[s][1] = lds_read_b64<256>(a);
            }
            lds_wait<8>(qv[0][0], qv[0][1], qv[1][0], qv[1][1], qv[2][0], qv[2][1], qv[3][0], qv[3][1]);
#pragma unroll
            for (int s = 0; s < 4; s += 2)
#pragma unroll
                for (int t = 0; t < 2; ++t) d2[t] += qv[s][t] - qv[s + 1][t];
            lds_wait<0>(qv[4][0], qv[4][1], qv[5][0], qv[5][1], qv[6][0], qv[6][1], qv[7][0], qv[7][1]);
#pragma unroll
            for (int s = 4; s < 8; s += 2)
#pragma unroll
                for (int t = 0; t < 2; ++t) d2[t] += qv[s][t] - qv[s + 1][t];
        }
    };
    auto gather = [&](f2 (&a2)[2], int nList) {
        for (int i0 = 0; i0 < nList; i0 += k2Round) {
            float2 ent[8];
            f2 qv[8][2];
#pragma unroll
            for (int s = 0; s < 8; ++s) ent[s] = sList[i0 + 8 * lh + s];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const unsigned a = qLaneAddr + (unsigned)__builtin_bit_cast(int, ent[s].y);
                qv[s][0] = lds_read_b64<0>(a);
                qv[s][1] = lds_read_b64<256>(a);
            }
            lds_wait<8>(qv[0][0], qv[0][1], qv[1][0], qv[1][1], qv[2][0], qv[2][1], qv[3][0], qv[3][1]);
            lds_wait<0>(qv[4][0], qv[4][1], qv[5][0], qv[5][1], qv[6][0], qv[6][1], qv[7][0], qv[7][1]);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const f2 ej = f2{ent[s].x, ent[s].y};
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a2[t]) : "v"(qv[s][t]), "v"(ej));   // a2 += q * J (J = low half of the entry)
            }
        }
    };
    // The pass loop in two instantiations.  kPure: every chip the tile touches -- owners and margin lanes of all its passes -- lies
    // on ONE side of the nav-bit boundary (all tiles but the one that holds the boundary and, with a boundary in the window, the
    // first and the last, whose margins wrap around): no side masks, no flush / spill inside the loop.
    auto passes = [&](auto pureTag) {
    constexpr bool kPure = decltype(pureTag)::value;
    setup(cLo, pureTag);
    fetch();
    // (a pass's flips are summed from zero and join the running sums with one addition: with a DC offset the prefix values
    // are ~1e6 and alternate in sign -- added one by one to a running sum of the peak's size they cost it digits)
    // Precision: with a DC offset the prefix values are a ramp of +-1e6 and the flips alternate in sign, so what a pass adds up
    // cancels to a few percent of its terms.  A group's entries are PAIRS of consecutive flips, an unpaired last flip is shared
    // out in quarters, and so is the end term -- every group's pass sum then cancels by itself -- and the pass sum is formed
    // from zero and joins the running sums with one addition.  ((+900, -700) LSB of DC on a 40 LSB signal: code bank within
    // 1.5e-6 of the oracle; 2.0e-6 with the flips dealt out one by one, 1.8e-6 with round 3's single-lag lanes.)
    auto gather_pass = [&](int nList, f2 endHalf) {
        f2 t2[2] = {endHalf, endHalf};
        gather(t2, nList);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] += t2[t];
    };

    for (int p = 0; p < nPassT; ++p) {
        const int e = eN, off = offN, len = lenN, eA = eAN, eB = eBN;
        const bool own = ownN, edge = edgeN;
        const unsigned long long rMask = rMaskN, sideMask = sideMaskN;
        const float r = mask_pm1(rMask);
        const bool sd1 = !kPure && __builtin_amdgcn_inverse_ballot_w64(sideMask);

        // ---- 1. the chip's samples in the lane's rotating frame: u_i, and vv = sum of all u_i (first moment, by Abel summation)
        f2 u[L1], uX, run = f2{0.f, 0.f}, vv = f2{0.f, 0.f};
        {
            auto slot = [&](int rv, int i) {   // i is a literal after unrolling
                const f2 si = f2{(float)(short)(rv & 0xFFFF), (float)(rv >> 16)};
                if (i < CI) run = tw_conj_mul_add(si, tw[CI - i], run);
                else if (i == CI) run += si;
                else run = tw_mul_add(si, tw[i - CI], run);
            };
            auto slots = [&](auto edgeTag) {
                constexpr bool kEdge = decltype(edgeTag)::value;
#pragma unroll
                for (int i = 0; i < L1; ++i) {
                    slot((!kEdge || i < len) ? raw[i] : 0, i);
                    u[i] = run;
                    vv += run;
                    // (keeps the later slots' sample conversions behind this group's arithmetic: hoisted they hold registers)
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (!edge) slots(std::false_type{});
            else slots(std::true_type{});
            slot(len > L1 ? rawX : 0, L1);
            uX = run;
            vv += run;
        }
        // issue priority for the latency-bound stages of the pass (scan, LDS stores, moments, gather): their few instructions between
        // waits go out ahead of another wave's sample stage, which has plenty more to issue (round 6: 0.5388 -> 0.5350 ms per 128 windows
        // at H in alternating runs; the priority on the sample stage instead: 0.545)
        __builtin_amdgcn_s_setprio(2);
        // ---- 2. chip offsets: wipe-off at the chip's frame origin, 64-lane scan of the chip totals, Q -> LDS (centred)
        double ph = fma((double)((own ? e : 0) + CI), ch.carrStep, ch.ri);   // wipe-off at the lane frame's origin
        ph -= floor(ph);
        const f2 wl = wipe_seed((float)ph);
        f2 tot = cmul(wl, run);
        tot = own ? tot : f2{0.f, 0.f};
        float incRe = tot.x, incIm = tot.y;
        wave_scan_incl2(incRe, incIm);
        const f2 T = f2{readlane_f(incRe, 63), readlane_f(incIm, 63)};
        const f2 half = T * 0.5f;
        const f2 qoff = f2{incRe, incIm} - tot - half;
        sQ[lane] = make_float2(-half.x, -half.y);                           // lower clamp pad: entries 0 .. k2Pad
        if (lane == 0) sQ[k2Pad] = make_float2(-half.x, -half.y);
        sQ[k2Pad + (eB - eA) + 1 + lane] = make_float2(half.x, half.y);     // upper clamp pad
        if (own) {
            float2 *q = sQ + k2Pad + (e - eA) + 1;
            auto store = [&](auto edgeTag) {
                constexpr bool kEdge = decltype(edgeTag)::value;
#pragma unroll
                for (int i = 0; i < L1; ++i) {
                    if (!kEdge || i < len) {
                        const f2 v = cmul_add(wl, u[i], qoff);
                        q[i] = make_float2(v.x, v.y);
                    }
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (four products in flight, not twenty-four)
                }
            };
            if (!edge) store(std::false_type{});
            else store(std::true_type{});
            if (len > L1) {
                const f2 v = cmul_add(wl, uX, qoff);
                q[L1] = make_float2(v.x, v.y);
            }
        }
        __builtin_amdgcn_wave_barrier();   // same-wave DS operations complete in order; this only pins the compiler
        // ---- the next pass's chips and samples: issued here (u[] is dead, its registers take the samples), they arrive under
        // the moments and the gather of this pass
        if (p + 1 < nPassT) {
            setup(cLo + (p + 1) * k2Own, pureTag);
            fetch();
        }

        // ---- 3. Doppler moments of the owned chips, from registers
        {
            const float lenf = (float)len;
            // first moment about the chip centre: w ((L1 + 1 + (1 - len) / 2) u_last - vv)   (vv holds L1 + 1 terms; slots beyond
            // the chip repeat u_last)
            const float kf = (float)(L1 + 1) + 0.5f * (1.f - lenf);
            const f2 D1 = cmul(wl, run * kf - vv);
            f2 wc;
            float G0, G1;
            if (!edge) {
                wc = cmul(wl, len == L1 ? thA : thB);
                G0 = len == L1 ? G0a : G0b;
                G1 = len == L1 ? G1a : G1b;
            } else {
                double pc = fma((double)(own ? e : 0) + 0.5 * (double)(len - 1), ch.carrStep, ch.ri);
                pc -= floor(pc);
                wc = wipe_seed((float)pc);
                mean_sums(lenf, G0, G1);
            }
            const f2 mw = cmul(meanv, wc);
            const f2 P0 = tot - mw * G0, P1 = D1 - f2{-mw.y, mw.x} * G1;
            const float xb = (float)(e - blk * Lt) - xOrigin + 0.5f * (lenf - 1.f);
            if constexpr (kPure) {
                const float rOwn = own ? r : 0.f;
                add_moments(P0 * rOwn, P1 * rOwn, xb);
            } else {
                // sides among the owned chips (uniform masks): all passes but the one that holds the nav-bit boundary have one
                const unsigned long long ownMask = __ballot(own);
                const bool any0 = (~sideMask & ownMask) != 0ull, any1 = (sideMask & ownMask) != 0ull;
                if (any1 && !any0 && curSide == 0) { flush(0); curSide = 1; }
                const float rA = (own && (sd1 == !any0)) ? r : 0.f;   // the owners on the first side present
                add_moments(P0 * rA, P1 * rA, xb);
                if (any0 && any1) {
                    flush(0);
                    curSide = 1;
                    const float rB = (own && sd1) ? r : 0.f;
                    add_moments(P0 * rB, P1 * rB, xb);
                }
            }
        }

        // ---- 4. lag sums: the flips among the boundaries the lag window reaches
        const bool inReach = lane >= 1 && e > eA - 32 && e < eB + 32;
        const unsigned long long rm = __ballot(inReach);
        const int firstIn = __builtin_ctzll(rm), lastIn = 63 - __builtin_clzll(rm);
        // sides present among the chips in reach (and the one before them, whose value enters the first difference)
        const unsigned long long near = rm | (rm >> 1);
        const unsigned long long s1m = sideMask & near;
        const int ofs8 = 8 * (e - eA);
        if (kPure || s1m == 0ull || s1m == near) {
            // one side (all passes but those about the nav-bit boundary or the window's ends): the flips are the chips in reach
            // whose sign differs from the chip before -- a scalar mask; J = r_prev - r = -2 r there
            if constexpr (!kPure) {
                const int side = s1m != 0ull ? 1 : 0;
                if (side != accSide) { spill(accSide); accSide = side; }
            }
            const unsigned long long bm = (rMask ^ (rMask << 1)) & rm;
            const int nb = __builtin_popcountll(bm);
            // list: the flips in pairs; with an odd count the last one stays out of it -- each half adds half of its term
            const bool odd = (nb & 1) != 0;
            const int nbE = nb & ~1, lastFlip = 63 - __builtin_clzll(bm | 1ull);
            const unsigned long long bmE = odd ? bm & ~(1ull << lastFlip) : bm;
            const int nList = (nbE + k2Round - 1) & ~(k2Round - 1);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bmE >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bmE, 0));
            int *lst = reinterpret_cast<int *>(sList);
            if (lane < k2Round) lst[nbE + lane] = 0;                         // padding: equal offsets, in pairs (nbE is even)
            if (__builtin_amdgcn_inverse_ballot_w64(bmE)) lst[rank] = ofs8;
            __builtin_amdgcn_wave_barrier();
            // J of the pass's first flip (J = r_prev - r = -2 r at a flip; the flips alternate)
            const float J0 = ((rMask >> __builtin_ctzll(bmE | (1ull << 63))) & 1ull) ? -2.f : 2.f;
            // end terms: the centred prefix is -T/2 before the first and +T/2 behind the last boundary in reach
            const float rEndsH = (float)((int)((rMask >> (firstIn - 1)) & 1ull) + (int)((rMask >> lastIn) & 1ull) - 1);
            f2 t0 = half * rEndsH;
            f2 t2[2] = {t0, t0};
            if (odd) {
                const float jh = ((rMask >> lastFlip) & 1ull) ? -1.f : 1.f;   // J / 2, J = -2 r
                const float2 *q = reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(sQ) + (qLaneBytes + __builtin_amdgcn_readlane(ofs8, lastFlip)));
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float2 v = q[32 * t];
                    t2[t] = __builtin_elementwise_fma(f2{v.x, v.y}, f2{jh, jh}, t2[t]);
                }
            }
            f2 d2[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}};
            gather_pairs(d2, nList);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] += __builtin_elementwise_fma(d2[t], f2{J0, J0}, t2[t]);
            __builtin_amdgcn_wave_barrier();   // the next pass rewrites the list
        } else {
            for (int sI = 0; sI < 2; ++sI) {
                if (sI != accSide) { spill(accSide); accSide = sI; }
                const float rs = (sd1 == (sI == 1)) ? r : 0.f;   // the replica masked to this side (:352-367)
                const float rsPrev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rs), 0x138, 0xf, 0xf, true));   // wave_shr:1 -> lane - 1
                const float J = inReach ? rsPrev - rs : 0.f;
                const unsigned long long bm = __ballot(J != 0.f);
                const float rF = readlane_f(rs, firstIn - 1), rL = readlane_f(rs, lastIn);
                const int nb = __builtin_popcountll(bm);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
                if ((bm >> lane) & 1ull) sList[rank] = make_float2(J, __builtin_bit_cast(float, ofs8));
                if (lane < k2Round) sList[nb + lane] = make_float2(0.f, 0.f);
                __builtin_amdgcn_wave_barrier();
                gather_pass((nb + k2Round - 1) & ~(k2Round - 1), half * (0.5f * (rF + rL)));
                __builtin_amdgcn_wave_barrier();   // the next side / pass rewrites the list
            }
        }
        // (Measured in round 3 and dropped: an offsets-only list with two batches of reads in flight: 0.578 against 0.565 ms per
        // 128 windows at H.  The LDS array is 60 % busy; what a wave waits for is its queue, not one round trip.)
        __builtin_amdgcn_wave_barrier();   // the next pass overwrites sQ
        __builtin_amdgcn_s_setprio(0);
    }
    };   // passes
    // side of a pure tile: chips below cB = the chip that starts at the nav-bit boundary lie before it (the host selects this
    // kernel only when the boundary IS a chip boundary); the tile's lanes span the chips cLo - 3 .. cLo + k2Own (nPassT - 1) + 60
    {
        const int cFirst = cLo - k2Own0, cLast = cLo + k2Own * (nPassT - 1) + (63 - k2Own0);
        const int cB = ch.hasFlip ? __builtin_amdgcn_readfirstlane(chip_at(ch.idxNext)) : 0;
        const bool pure = !ch.hasFlip || (cFirst > c0 && cLast <= cEnd && (cB <= cFirst || cB > cLast));
        if (pure) {
            curSide = accSide = (ch.hasFlip && cB <= cFirst) ? 1 : 0;
            passes(std::true_type{});
        } else {
            passes(std::false_type{});
        }
    }
    flush(curSide);
    for (int side = 0; side < 2; ++side)   // a side without samples in this tile: zero block
        if (!((flushed >> side) & 1) && lane < kNMom) momOut[side * momSide + lane] = make_float2(0.f, 0.f);
    // ---- block partial of the lag sums: what is still in registers, then zeros for a side that never occurred
    spill(accSide);
    for (int side = 0; side < 2; ++side)
        if (!((spilled >> side) & 1)) partOut[side * NL + lane] = make_float2(0.f, 0.f), partOut[side * NL + 64] = make_float2(0.f, 0.f);
}

}  // namespace dpe
