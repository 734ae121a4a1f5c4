// dpe_bcs_chip.h -- stage 1 for high sampling rates (a chip spans many samples, lag windows of +-31 samples):
// per-SV work scales with the chip boundaries of the replica, not with the samples.  Included by dpe_bcs.hip.
//
// Reference semantics (cudarecv/modules/src/batchcorrscores.cu): corr[l] = sum_n b[n] r[(n-l) mod S] with
// b = raw * wipe (:277-305, :402) and r the +-1 code replica (:323-372), both nav-bit sides (:237-258); carrier path
// c[n] = (raw[n] - mean) wipe[n] r[n] -> zero-padded C-point FFT (:422-452, :1179).
//
// One WAVE (= one block) walks `tpb` passes of kPass = 64 x 17 samples of one (window, SV).  Per pass
//   1. wiped prefix sums  Q[p] = sum_{n < c0+p} raw[n] wipe[n]  : lane-local in the lane's own rotating frame
//      (u_i = u_{i-1} + raw_i T_i, T_i = exp(-j 2 pi i fi/fs): two packed FMAs per sample, no rotation chain), one
//      64-lane DPP scan of the lane totals, then Q = w_lane u_i + offset (two more) -> LDS, clamp-padded at both ends;
//   2. the replica is piecewise constant:  corr_pass[l] = r_end T + sum_i J_i Q[e_i - c0 + l]  over the chip
//      boundaries e_i of the pass (J_i = r[e_i - 1] - r[e_i] = +-2, only where the code actually flips) -- one LDS read
//      and one packed FMA per boundary with LANES <-> the 64 lags l = -32 .. +31;
//   3. Doppler moments from per-CHIP sums (lanes <-> chips): zeroth moment Q[b] - Q[a], first moment about the chip
//      centre by Abel summation over the chip's prefix values, DC-mean term in closed form (sum over a chip of
//      exp(-j phi d) and d exp(-j phi d)); the chip's contribution to the block moments is xbar^p E0 + p xbar^(p-1) E1
//      -- the d^2 term is (theta len)^2 / 24 of E0 (checked at create).  The moment block is the wave's whole tile of
//      tpb passes: the per-lane moment sums stay in registers across the passes and are reduced once per (tile, side).
// Passes that touch the circular wrap or straddle the nav-bit boundary take a general per-sample path (rare).
// 17 samples per lane (not 16): the lane's LDS chunk stride is 17 entries, so the chunk writes are bank-conflict free
// while the lag gathers read consecutive entries.
#pragma once

namespace dpe {

#ifndef DPE_CHIP_SPL
#define DPE_CHIP_SPL 17
#endif
constexpr int kSPL = DPE_CHIP_SPL;   // samples per lane (odd: the chunk stride in LDS entries must be odd for conflict-free writes)
constexpr int kPass = 64 * kSPL;     // samples per wave pass
constexpr int kPad = 64;             // clamp pads of the prefix array (|boundary offset| + |lag| < 64; a chip + 8 < 64)
constexpr int kQLen = kPass + 2 * kPad + 1;

// a * b + c (complex) in two packed FMAs
__device__ __forceinline__ f2 cmul_add(f2 a, f2 b, f2 c)
{
    f2 t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(b), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}

// inclusive prefix sum of (re, im) over the 64 lanes: row_shr 1, 2, 4, 8 inside the rows of 16, then the two row
// broadcasts.  In-place v_add_f32 with a DPP source: lanes whose source lies outside the row keep their value.
__device__ __forceinline__ void wave_scan_incl2(float &a, float &b)
{
#define DPE_SCAN_STEP(mod) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " mod "\n\tv_add_f32_dpp %1, %1, %1 " mod : "+v"(a), "+v"(b));
    DPE_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
    DPE_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
    DPE_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
    DPE_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
    DPE_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
    DPE_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef DPE_SCAN_STEP
}

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

template <int kNMom>
__global__ __launch_bounds__(64, 4) void bcs_bank_chip_kernel(BcsParamBlock pb, int inl, const int16_t *__restrict__ iq, long long winStride,
                                                              int S, int K, int nW, int nPass, int tpb, int nBlk, int nSumBlk, int lagShift, int dbgArg,
                                                              const BcsChanDev *__restrict__ chan,
                                                              const long long *__restrict__ sums,
                                                              const int8_t *__restrict__ chipTable,
                                                              float2 *__restrict__ part, float2 *__restrict__ mom)
{
    constexpr int NL = 65;   // partial layout shared with the other stage-1 kernels: entry j <-> lag lagShift + j - 32 (j = 64 unused)
    __shared__ float2 sQ[kQLen];
    __shared__ __align__(16) float2 sRot[kSPL + 1];
    __shared__ float2 sList[64 + 8];   // boundaries of a round: {J, byte offset into sQ}

    // Block -> (window, tile, SV), XCD-aware: workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8, observed;
    // used for speed only), so the K blocks that read the SAME samples (one per SV of a tile) are given linear ids that
    // are congruent mod 8 and consecutive on that XCD: one HBM fetch, K - 1 hits in the XCD's L2.
    const int lane = threadIdx.x;
    const int slot = blockIdx.x >> 3, k = slot % K, tg = (slot / K) * 8 + (blockIdx.x & 7);
    if (tg >= nBlk * nW) return;
    const int w = tg / nBlk, blk = tg - w * nBlk;
    (void)pb;
    const BcsChanDev ch = params_ptr(chan, inl)[(size_t)w * K + k];
    const int8_t *chips = chipTable + (ch.prn - 1) * 1024;
    float mRe, mIm;
    window_mean(sums, w, nSumBlk, S, mRe, mIm);
    const f2 meanv = f2{mRe, mIm};
    const int *x = reinterpret_cast<const int *>(iq + (size_t)w * winStride * 2);
#ifdef DPE_EXPERIMENTS
    const int dbg = dbgArg;   // timing ablations (DPE_BCS_CHIP_DBG): stages are skipped, the results are then wrong
#else
    constexpr int dbg = 0;    // the product library has no such switch
    (void)dbgArg;
#endif
    const bool doMom = lagShift == 0 && !(dbg & 1);

    // ---- once per block: lower clamp pad (Q = 0 at and before the pass start), the SV's twiddles T_i, zeroed moment slots
    sQ[lane] = make_float2(0.f, 0.f);
    if (lane == 0) sQ[kPad] = make_float2(0.f, 0.f);
    if (lane <= kSPL) {
        double ph = (double)lane * ch.carrStep;
        ph -= floor(ph);
        const f2 t = wipe_seed((float)ph);
        sRot[lane] = make_float2(t.x, t.y);
    }
    float2 *momOut = mom + ((((size_t)w * K + k) * 2) * nBlk + blk) * kNMom;   // [side][nBlk][kNMom]
    const size_t momSide = (size_t)nBlk * kNMom;
    __builtin_amdgcn_wave_barrier();

    // raw samples of the first pass; every later pass is fetched while the previous one is processed
    int rawNext[kSPL];
    auto fetch = [&](int nn) {   // nn = first sample of the lane; only the window's last pass needs the bound check
        const int *px = x + nn;
        if (nn - kSPL * lane + kPass <= S) {
#pragma unroll
            for (int i = 0; i < kSPL; ++i) rawNext[i] = px[i];
        } else {
#pragma unroll
            for (int i = 0; i < kSPL; ++i) rawNext[i] = (nn + i < S) ? px[i] : 0;
        }
    };
    fetch(blk * tpb * kPass + kSPL * lane);
    const int qLaneBytes = 8 * ((kPad - 32) + lane);   // byte offset in sQ of lag (lane - 32)'s entry for a boundary at the pass start
    const float phi = (float)(6.283185307179586476925286766559 * ch.carrStep);   // wipe-off phase step per sample (rad)
    const int minLen = (int)ch.invStep - 1;   // every chip has minLen + 1 or + 2 samples (minLen itself only if fs / fc rounds across an integer)
    // DC-mean sums over a chip of len samples about its centre: G0 = sum exp(-j phi d) (real), j G1 = sum d exp(-j phi d)
    auto mean_sums = [&](float fl, float &G0, float &G1) {
        const float l2 = fl * fl, h2 = 0.25f * phi * phi;
        G0 = fl * (1.f - (l2 - 1.f) * h2 * (1.f / 6.f) * (1.f - (3.f * l2 - 7.f) * h2 * (1.f / 60.f)));
        G1 = -phi * fl * (l2 - 1.f) * (1.f / 12.f) * (1.f - phi * phi * (3.f * l2 - 7.f) * (1.f / 120.f));
    };
    float G0a, G1a, G0b, G1b;   // the two regular chip lengths minLen + 1, minLen + 2
    mean_sums((float)(minLen + 1), G0a, G1a);
    mean_sums((float)(minLen + 2), G0b, G1b);
    const double tileCentre = (double)(blk * tpb) * kPass + 0.5 * ((double)tpb * kPass - 1.0);   // moment abscissa origin: the tile's centre

    f2 accS[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}};   // lag sums of the two nav-bit sides, lane <-> lag
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
    // contribution of one chip / sample group to the tile moments: E0 = zeroth, E1 = first moment about its own centre xb
    auto add_moments = [&](f2 E0, f2 E1, float xb) {
        M[0] += E0;
        f2 xp = E0 * xb;          // xbar^p E0
        f2 xq = E1;               // p xbar^(p-1) E1
#pragma unroll
        for (int p = 1; p < kNMom; ++p) {
            M[p] += xp + xq * (float)p;
            xp *= xb;
            xq *= xb;
        }
    };

    for (int t = 0; t < tpb; ++t) {
        const int pass = blk * tpb + t;
        if (pass >= nPass) break;
        const int c0 = pass * kPass;
        const int n0 = c0 + kSPL * lane;
        const int cEnd = (c0 + kPass < S) ? c0 + kPass : S;   // first sample index past this pass
        // replica index range the pass can touch for any of the 64 lags: m in [lo, hi]  (m = n - lag - lagShift)
        const int lo = c0 - 32 - lagShift, hi = c0 + kPass - 1 + 32 - lagShift;
        const bool wrap = lo < 0 || hi >= S;
        const bool straddle = ch.hasFlip && lo < ch.idxNext && hi >= ch.idxNext;
        const bool fast = !wrap && !straddle;
        if (dbg & 4) continue;
        // lane <-> chip cj = ciLo + base + lane; its first replica index e(cj) = min { m : floor(m codeStep + rc) >= cj }
        // (phases are >= 0 on the fast path, so the truncating conversions are floors)
        auto first_index = [&](int c) -> int {
            const double md = ceil(((double)c - ch.rc) * ch.invStep);
            int m = (int)md;
            if ((int)fma(md - 1.0, ch.codeStep, ch.rc) >= c) m -= 1;
            else if ((int)fma(md, ch.codeStep, ch.rc) < c) m += 1;
            return m;
        };
        int ciLo = 0, ciHi = 0, ciLoMod = 0, e = 0;
        int8_t rRaw = 0;
        auto chip_round = [&](int base) {   // positions and values of the 64 chips of a round
            e = (base + lane == 0) ? lo : first_index(ciLo + base + lane);
            int cm = ciLoMod + base + lane;
            cm -= cm >= kLCA ? kLCA : 0;
            cm -= cm >= kLCA ? kLCA : 0;
            rRaw = chips[cm];
        };
        if (fast) {
            ciLo = (int)fma((double)lo, ch.codeStep, ch.rc);
            ciHi = (int)fma((double)hi, ch.codeStep, ch.rc);
            ciLoMod = ciLo % kLCA;
            chip_round(0);   // issued before the prefix work: the chip-table load completes under it
        }
        // ---- 1. wiped prefix sums
        f2 u[kSPL];
        {
            f2 run = f2{0.f, 0.f};
            float4 tw[(kSPL + 1) / 2];                   // twiddles, two per 16-byte broadcast read (wave-uniform address)
#pragma unroll
            for (int i = 0; i < (kSPL + 1) / 2; ++i) tw[i] = reinterpret_cast<const float4 *>(sRot)[i];
#pragma unroll
            for (int i = 0; i < kSPL; ++i) {
                const f2 tt = (i & 1) ? f2{tw[i / 2].z, tw[i / 2].w} : f2{tw[i / 2].x, tw[i / 2].y};
                const f2 si = f2{(float)(short)(rawNext[i] & 0xFFFF), (float)(rawNext[i] >> 16)};
                run = cmul_add(si, tt, run);
                u[i] = run;
            }
        }
        if (t + 1 < tpb && pass + 1 < nPass) fetch(n0 + kPass);
        double ph = fma((double)n0, ch.carrStep, ch.ri);
        ph -= floor(ph);
        const f2 wl = wipe_seed((float)ph);
        const f2 tot = cmul(wl, u[kSPL - 1]);
        float incRe = tot.x, incIm = tot.y;
        wave_scan_incl2(incRe, incIm);
        const f2 inc = f2{incRe, incIm};
        const f2 qoff = inc - tot;
        const f2 T = f2{readlane_f(inc.x, 63), readlane_f(inc.y, 63)};
#pragma unroll
        for (int i = 0; i < kSPL; ++i) {
            const f2 q = cmul_add(wl, u[i], qoff);
            sQ[kPad + kSPL * lane + i + 1] = make_float2(q.x, q.y);
        }
        sQ[kPad + kPass + 1 + lane] = make_float2(T.x, T.y);   // upper clamp pad
        __builtin_amdgcn_wave_barrier();   // same-wave DS operations complete in order; this only pins the compiler

        if (dbg & 8) continue;
        if (fast) {
            // ================= fast path: one side, chips by lanes (lanes 0..62 work; lane 63 only supplies the end of
            // lane 62's chip and comes back as lane 0 of the next round)
            const int side = (ch.hasFlip && lo >= ch.idxNext) ? 1 : 0;
            if (doMom && side != curSide) { flush(curSide); curSide = side; }
            f2 acc = f2{0.f, 0.f};
            float carryPrev = 0.f, rEnd = 0.f;
            for (int base = 0; base <= ciHi - ciLo; base += 63) {
                if (base) chip_round(base);
                const int cj = ciLo + base + lane;
                const float rCur = (float)rRaw;
                const int eNext = __builtin_amdgcn_update_dpp(0, e, 0x130, 0xf, 0xf, true);            // wave_shl:1 -> lane + 1
                float rPrev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rCur), 0x138, 0xf, 0xf, true));   // wave_shr:1 -> lane - 1
                if (lane == 0) rPrev = base ? carryPrev : rCur;
                carryPrev = readlane_f(rCur, 62);
                if (ciHi - ciLo - base < 64) rEnd = readlane_f(rCur, ciHi - ciLo - base);
                // ---- 2. lag sums: boundaries inside (lo, hi] where the code flips.  They are compacted into an LDS list
                // {J, byte offset}; the gather loop then runs 8 independent LDS reads deep
                const float J = rPrev - rCur;
                const bool work = lane < 63 && cj <= ciHi;
                const unsigned long long bm = __ballot(work && J != 0.f);
                const int nb = __builtin_popcountll(bm);
                {
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
                    if ((bm >> lane) & 1ull) sList[rank] = make_float2(J, __builtin_bit_cast(float, 8 * (e + lagShift - c0)));
                    if (lane < 8) sList[nb + lane] = make_float2(0.f, 0.f);
                    __builtin_amdgcn_wave_barrier();
                }
                // moments first half (issued ahead of the gather loop so that their LDS latency overlaps it): the chip's
                // FULL extent [e, eNext) on the clamp-padded prefix array (Q = 0 before the pass, = T after it) -- the pads
                // make the sums below those of the samples inside the pass, and every chip has minLen + 1 or + 2 entries:
                // a uniform trip count, no lane predicates
                const int a = e < c0 ? c0 : e;
                const int b = eNext > cEnd ? cEnd : eNext;
                const int len = (doMom && work) ? b - a : 0;
                const int full = eNext - e;
                const float2 *qa = sQ + kPad + (len > 0 ? e - c0 : 0);    // idle lanes read the pass start (in bounds, unused)
                const float2 Qa = qa[0], Qb = qa[len > 0 ? full : 0];
                f2 sq = f2{0.f, 0.f};
                if (doMom) {
                    // first moment about e (Abel summation): sum_n (n - e) b[n] = (full-1) Q[eNext] - sum_{u=e+1}^{eNext-1} Q[u];
                    // entries e + 1 .. e + minLen - 1 are interior for every chip (wave-uniform trip count, eight LDS reads in
                    // flight), the next two only for the longer chips
                    const int nInt = minLen - 1;
                    int u0 = 1;
                    for (; u0 + 7 <= nInt; u0 += 8) {
                        float2 v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = qa[u0 + j];
#pragma unroll
                        for (int j = 0; j < 8; ++j) sq += f2{v[j].x, v[j].y};
                    }
                    {
                        float2 v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = qa[u0 + j];
                        const float2 x0 = qa[minLen], x1 = qa[minLen + 1];
                        const int rem = nInt - u0 + 1;   // 0..7 entries of this batch are interior (wave-uniform)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float wt = j < rem ? 1.f : 0.f;
                            sq = __builtin_elementwise_fma(f2{v[j].x, v[j].y}, f2{wt, wt}, sq);
                        }
                        const float w0 = minLen < full ? 1.f : 0.f, w1 = minLen + 1 < full ? 1.f : 0.f;
                        sq = __builtin_elementwise_fma(f2{x0.x, x0.y}, f2{w0, w0}, sq);
                        sq = __builtin_elementwise_fma(f2{x1.x, x1.y}, f2{w1, w1}, sq);
                    }
                }
                for (int i0 = 0; i0 < ((dbg & 2) ? 0 : nb); i0 += 8) {
                    float2 ent[8], qv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) ent[j] = sList[i0 + j];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        qv[j] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(sQ) + (qLaneBytes + __builtin_bit_cast(int, ent[j].y)));
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f2 ej = f2{ent[j].x, ent[j].y}, qj = f2{qv[j].x, qv[j].y};
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(qj), "v"(ej));   // acc += q * J (J = low half of the entry)
                    }
                }
                __builtin_amdgcn_wave_barrier();   // the next round rewrites the list
                // ---- 3. Doppler moments from the chips of the pass at lag 0 (centre chunk only): chip cj covers
                // [a, b) = [max(e, c0), min(eNext, cEnd))
                if (len > 0) {
                    const f2 dQ = f2{Qb.x - Qa.x, Qb.y - Qa.y};
                    const f2 S1 = f2{Qb.x, Qb.y} * (float)(full - 1) - sq;
                    const double nc = (double)a + 0.5 * (double)(len - 1);   // centre of the part inside the pass
                    const f2 D1 = S1 - dQ * (float)(nc - (double)e);
                    // DC-mean term: mean * sum_{n in chip} {1, d} wipe[n],  d = n - centre
                    double pc = fma(nc, ch.carrStep, ch.ri);
                    pc -= floor(pc);
                    const f2 wc = wipe_seed((float)pc);
                    float G0 = len == minLen + 1 ? G0a : G0b, G1 = len == minLen + 1 ? G1a : G1b;
                    if (len != minLen + 1 && len != minLen + 2) mean_sums((float)len, G0, G1);   // chips clamped at the pass ends
                    const f2 mw = cmul(meanv, wc);
                    const f2 E0 = (dQ - mw * G0) * rCur;
                    const f2 E1 = (D1 - f2{-mw.y, mw.x} * G1) * rCur;
                    add_moments(E0, E1, (float)(nc - tileCentre));
                }
            }
            // end term: replica value at the far end of the range times the pass total
            acc = __builtin_elementwise_fma(T, f2{rEnd, rEnd}, acc);
            accS[0] += side == 0 ? acc : f2{0.f, 0.f};
            accS[1] += side == 1 ? acc : f2{0.f, 0.f};
        } else {
            // ================= general path (circular wrap inside the range, or the nav-bit boundary): the masked
            // replica entry by entry, as bcs_bank_kernel builds it (BCS_ComputeCodeReplica :347-367)
            const int nSides = ch.hasFlip ? 2 : 1;
            for (int side = 0; side < nSides; ++side) {
                auto rep = [&](int mm) -> float {   // replica index (unwrapped) -> masked value
                    if (mm < 0) mm += S; else if (mm >= S) mm -= S;
                    const int ci = ((int)floor(fma((double)mm, ch.codeStep, ch.rc))) % kLCA;
                    const int sd = ch.hasFlip ? (mm >= ch.idxNext) : 0;
                    return (sd == side) ? (float)chips[ci] : 0.f;
                };
                // entries q = 0 .. 64*18-1 <-> m = lo + q; lane owns q = 18 lane .. 18 lane + 17
                constexpr int EPL = (kPass + 64) / 64;
                static_assert(EPL * 64 == kPass + 64, "the replica range divides evenly over the lanes");
                f2 acc = f2{0.f, 0.f};
                float prev = rep(lo + EPL * lane - 1);   // lane 0: entry before the range (its J is not used: q = 0 is r0)
                for (int tq = 0; tq < EPL; ++tq) {
                    const int q = EPL * lane + tq;
                    const float cur = rep(lo + q);
                    const float J = (q == 0) ? 0.f : prev - cur;
                    prev = cur;
                    unsigned long long bm = __ballot(J != 0.f);
                    while (bm) {
                        const int src = __builtin_ctzll(bm);
                        bm &= bm - 1;
                        const int er = EPL * src + tq - 32;   // (lo + q) + lagShift - c0
                        const float Jv = readlane_f(J, src);
                        const float2 qv = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(sQ) + (qLaneBytes + 8 * er));
                        acc = __builtin_elementwise_fma(f2{qv.x, qv.y}, f2{Jv, Jv}, acc);
                    }
                }
                {
                    const float rEnd = readlane_f(prev, 63);   // entry hi
                    acc = __builtin_elementwise_fma(T, f2{rEnd, rEnd}, acc);
                }
                accS[0] += side == 0 ? acc : f2{0.f, 0.f};
                accS[1] += side == 1 ? acc : f2{0.f, 0.f};
                // moments sample by sample: c = (raw w - mean w) r  (:480, :440-448) -- only for a side that has samples in
                // this pass; the sides come in increasing sample order, so the accumulators switch 0 -> 1 at most once
                const bool hasSamples = !ch.hasFlip || (side == 0 ? c0 < ch.idxNext : cEnd > ch.idxNext);
                if (doMom && hasSamples) {
                    if (side != curSide) { flush(curSide); curSide = side; }
                    const float xb0 = (float)((double)n0 - tileCentre);
#pragma unroll 1
                    for (int i = 0; i < kSPL; ++i) {   // (rolled: the wiped sample comes back from the prefix array)
                        const float2 tt = sRot[i];
                        const f2 wv = cmul(wl, f2{tt.x, tt.y});
                        const float2 q0 = sQ[kPad + kSPL * lane + i], q1 = sQ[kPad + kSPL * lane + i + 1];
                        const float r0 = (n0 + i < S) ? rep(n0 + i) : 0.f;
                        const f2 cp = (f2{q1.x - q0.x, q1.y - q0.y} - cmul(meanv, wv)) * r0;
                        add_moments(cp, f2{0.f, 0.f}, xb0 + (float)i);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // the next pass overwrites sQ
    }
    if (doMom) {
        flush(curSide);
        for (int side = 0; side < 2; ++side)   // a side without samples in this tile: zero block
            if (!((flushed >> side) & 1) && lane < kNMom) momOut[side * momSide + lane] = make_float2(0.f, 0.f);
    }
    // ---- block partial of the lag sums (one wave: nothing to reduce)
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        float2 *o = part + ((((size_t)w * K + k) * nBlk + blk) * 2 + side) * NL;
        o[lane] = make_float2(accS[side].x, accS[side].y);
        if (lane == 0) o[64] = make_float2(0.f, 0.f);
    }
}

}  // namespace dpe
