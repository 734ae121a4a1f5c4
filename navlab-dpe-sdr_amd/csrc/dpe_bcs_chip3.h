// dpe_bcs_chip3.h -- stage 1 for high sampling rates, third form: NO per-(sample, SV) work at all.
// Included by dpe_bcs.hip after dpe_bcs_chip2.h (whose helpers it shares).
//
// Reference semantics (cudarecv/modules/src/batchcorrscores.cu): corr[l] = sum_n b[n] r[(n-l) mod S] with b = raw * wipe
// (:277-305, :402) and r the +-1 code replica (:323-372), both nav-bit sides (:237-258); carrier path
// c[n] = (raw[n] - mean) wipe[n] r[n] -> zero-padded C-point FFT (:422-452, :1179).
//
// The first two chip forms wipe every sample once per SV (5 packed FMAs, 2 converts and an LDS write per sample and SV in the
// second form).  Here a BLOCK owns a tile of samples and serves ALL SVs from it:
//   phase 0 (once per tile, shared by the SVs): the block forms three SV-INDEPENDENT prefix arrays of z = raw - round(mean) in
//     LDS -- C0[n] = sum_{m<n} z[m] (integers: exact in fp32 below 2^24), W1 / W2 = the same sums weighted with (m mod 64) - 32
//     and its square (segment-centred weights keep them small);
//   per SV and chip (lanes <-> chips): the chip's raw moments R0, R1, R2 about its own centre are DIFFERENCES of the prefix
//     arrays at its two boundaries (one more read where the chip crosses a 64-sample segment), and its wiped sums follow from
//     the second-order expansion of the carrier over the chip,  E0 = R0 - j phi R1 - phi^2/2 R2,  E1 = R1 - j phi R2
//     (phi = 2 pi fi / fs; |phi| (L1 + 1) / 2 <= 0.0375 is checked per Update: 12 kHz at 25 Msps; remainder < 1e-7 of the peak) --
//     ~90 instructions per 63 chips instead of 5 per sample;
//   per SV and FLIP (lanes <-> lags): corr[l+1] - corr[l] = sum_j J_j b[e_j + l] over the chip boundaries e_j where the replica
//     changes sign (J = +-2; +-1 at the nav-bit boundary and at the circular wrap of a side), and with c_j = J_j exp(-j phase(e_j))
//       D'[l] = sum_j c_j raw[e_j + l] = (G[l+1] - G[l]) + round(mean) sum_j c_j,   G[l] = sum_j c_j C0[e_j + l]:
//     one LDS read of 64 consecutive prefix values and two packed FMAs per flip for all 64 lags -- no conversion, no wiped
//     sample; exp(-j phi l), the running sum over l and the directly computed lag 0 (= the chips' zeroth moments) are applied
//     once per block partial.
// A block walks T consecutive tiles (its moment block = T Lt samples, bcs_finalize_kernel's layout); its NW waves share every
// tile's prefix arrays and each serves NSV SVs (k = kg NW NSV + wave + NW s); per-SV accumulators stay in registers across the
// tiles.  The nav-bit boundary must be a chip boundary of the replica (the host's condition for the second form as well).
#pragma once

namespace dpe {

constexpr int k3HL = 32;       // samples in front of a tile that its flips' lag windows reach
constexpr int k3HR = 34;       // ... and behind it (lags up to +31 and the chip that straddles the tile's end)
constexpr int k3MaxOwn = 63;   // chips that start inside a tile (lane nOwn looks one chip ahead)
constexpr int k3Mid = 800;     // origin of the W1 / W2 weights (tile-local sample index)
constexpr int k3List = 64 + 4; // flip list of a wave, zero-padded to a multiple of four
#ifndef DPE_C3_WAVES
#define DPE_C3_WAVES 3
#endif
constexpr float k3PhiMax = 0.0375f;   // |phi| (L1 + 1) / 2 bound of the second-order chip expansion

// entries per thread of phase 0 (prefix entries 0 .. NTs of a tile of <= 1600 + k3HL + k3HR samples)
template <int NW> struct Chip3Shape { static constexpr int kEPT = (1700 + 64 * NW - 1) / (64 * NW); static constexpr int kMaxEntries = kEPT * 64 * NW; };

__device__ __forceinline__ float dpp_shl1_f(float v)   // lane l <- lane l + 1 (lane 63 <- 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shr1_f(float v)   // lane l <- lane l - 1 (lane 0 <- 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ f2 dpp_shl1(f2 v) { return f2{dpp_shl1_f(v.x), dpp_shl1_f(v.y)}; }

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) -- the per-SV register arrays must be
// indexed by constants (a run-time index would send them to scratch memory)
template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// a + s * (-j b) = {a.x + s b.y, a.y - s b.x}
__device__ __forceinline__ f2 add_mjs(f2 a, float s, f2 b) { return f2{fmaf(s, b.y, a.x), fmaf(-s, b.x, a.y)}; }

// inclusive prefix sums of N floats over the 64 lanes (row shifts, then the two row broadcasts), step-major so that the
// VALU -> DPP hazard of one chain is covered by the others
template <int N>
__device__ __forceinline__ void wave_scan_incl(float (&v)[N])
{
    static_assert(N >= 3, "the step-major interleave must cover the 2 wait states of a VALU->DPP hazard");
#define DPE_SCAN_STEP(mod) \
    _Pragma("unroll") for (int i = 0; i < N; ++i) asm volatile("v_add_f32_dpp %0, %0, %0 " mod : "+v"(v[i]));
    DPE_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
    DPE_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
    DPE_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
    DPE_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
    DPE_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
    DPE_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef DPE_SCAN_STEP
}


template <int kNMom, int NW, int NSV>
__global__ __launch_bounds__(64 * NW, DPE_C3_WAVES) void bcs_bank_chip3_kernel(BcsParamBlock pb, int inl, const int16_t *__restrict__ iq, long long winStride,
                                                                 int S, int K, int nW, int Lt, int T, int nBlk, int nKG, int nSumBlk,
                                                                 const BcsChanDev *__restrict__ chan,
                                                                 const long long *__restrict__ sums,
                                                                 const uint32_t *__restrict__ chipBits,
                                                                 float2 *__restrict__ part, float2 *__restrict__ mom)
{
    constexpr int NL = 65;   // partial layout shared with the other stage-1 kernels: entry j <-> lag j - 32 (j = 64 unused)
    constexpr int kEPT = Chip3Shape<NW>::kEPT;
    extern __shared__ __align__(16) unsigned char smem3[];
    // LDS: C0 [nEnt] float2 | W [nEnt] float4 {W1, W2} | flip lists [NW][k3List] float4 | scan totals [NW][6]
    const int nEnt = Lt + k3HL + k3HR + 1;
    float2 *sC0 = reinterpret_cast<float2 *>(smem3);
    float4 *sW = reinterpret_cast<float4 *>(smem3 + (((size_t)nEnt * 8 + 15) & ~(size_t)15));
    float4 *sListAll = sW + nEnt;
    float *sScan = reinterpret_cast<float *>(sListAll + NW * k3List);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform for the compiler too: the per-SV parameter loads become scalar loads)
    float4 *sList = sListAll + wave * k3List;
    int bx = blockIdx.x;
    const int kg = bx % nKG;
    bx /= nKG;
    const int blk = bx % nBlk, w = bx / nBlk;
    (void)pb; (void)nW;
    const int *x = reinterpret_cast<const int *>(iq + (size_t)w * winStride * 2);

    // ---- the window's mean (batchcorrscores.cu:1065-1066): m = sum / (float) S in fp64; z = raw - mI with mI = round(m) keeps
    // the prefix sums integers; mu = m - mI enters the chips' carrier-path moments in closed form
    f2 mFull, mInt, mu;
    {
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
        const double mr = (double)tI / (double)(float)S, mi = (double)tQ / (double)(float)S;
        const double ir = rint(mr), ii = rint(mi);
        mFull = f2{(float)mr, (float)mi};
        mInt = f2{(float)ir, (float)ii};
        mu = f2{(float)(mr - ir), (float)(mi - ii)};
    }
    const int mIntRe = (int)mInt.x, mIntIm = (int)mInt.y;

    // ---- per-SV state of this wave
    BcsChanDev ch[NSV];
    bool live[NSV];
    int c0[NSV], cEnd[NSV], cB[NSV], curSide[NSV], spilled[NSV], flushed[NSV], bitsV[NSV];
    float phi[NSV];
    f2 thP[NSV];                    // exp(-j phi L1 / 2): chip boundary -> expansion point of the chip (its first sample + L1 / 2)
    int L1[NSV];
    f2 G[NSV], SC[NSV], ZG[NSV], DE[NSV], M[NSV][kNMom];
#pragma unroll
    for (int s = 0; s < NSV; ++s) {
        const int k = (kg * NSV + s) * NW + wave;
        live[s] = k < K;
        const int kk = live[s] ? k : 0;
        ch[s] = params_ptr(chan, inl)[(size_t)w * K + kk];
        bitsV[s] = (int)(lane < k2BitWords ? chipBits[(ch[s].prn - 1) * k2BitWords + lane] : 0u);
        c0[s] = __builtin_amdgcn_readfirstlane((int)fma(0.0, ch[s].codeStep, ch[s].rc));
        cEnd[s] = __builtin_amdgcn_readfirstlane((int)fma((double)(S - 1), ch[s].codeStep, ch[s].rc));
        cB[s] = ch[s].hasFlip ? __builtin_amdgcn_readfirstlane((int)fma((double)ch[s].idxNext, ch[s].codeStep, ch[s].rc)) : 0x7fffffff;
        curSide[s] = -1; spilled[s] = 0; flushed[s] = 0;   // curSide < 0: nothing accumulated yet
        phi[s] = (float)(6.283185307179586476925286766559 * ch[s].carrStep);
        L1[s] = (int)ch[s].invStep;
        {
            double ph = 0.5 * (double)L1[s] * ch[s].carrStep;
            ph -= floor(ph);
            const f2 t = wipe_seed((float)ph);
            thP[s] = f2{readlane_f(t.x, 0), readlane_f(t.y, 0)};
        }
        G[s] = SC[s] = ZG[s] = DE[s] = f2{0.f, 0.f};
#pragma unroll
        for (int p = 0; p < kNMom; ++p) M[s][p] = f2{0.f, 0.f};
    }

    const int stBase = blk * T * Lt;                       // first sample of the block's moment block
    const float xOrigin = 0.5f * (float)(T * Lt - 1);      // moment abscissa origin relative to stBase (bcs_finalize_kernel: momLen = T Lt)

    // ---- block partial of one (SV, side): lag sums from G / SC / DE and the direct lag 0, moments; accumulators cleared
    auto flush = [&](auto sTag, int side) __attribute__((always_inline)) {
        constexpr int s = decltype(sTag)::value;
        const int k = (kg * NSV + s) * NW + wave;
        float2 *partOut = part + ((((size_t)w * K + k) * nBlk + blk) * 2) * NL + side * NL;
        float2 *momOut = mom + ((((size_t)w * K + k) * 2) * nBlk + blk) * kNMom + (size_t)side * nBlk * kNMom;
        float red[2 * kNMom + 4];
#pragma unroll
        for (int p = 0; p < kNMom; ++p) { red[2 * p] = M[s][p].x; red[2 * p + 1] = M[s][p].y; }
        red[2 * kNMom] = SC[s].x; red[2 * kNMom + 1] = SC[s].y; red[2 * kNMom + 2] = ZG[s].x; red[2 * kNMom + 3] = ZG[s].y;
        dpp_sum_lane63(red);
        float tot[2 * kNMom + 4];
#pragma unroll
        for (int i = 0; i < 2 * kNMom + 4; ++i) tot[i] = lane63(red[i]);
        const f2 sc = f2{tot[2 * kNMom], tot[2 * kNMom + 1]}, zg = f2{tot[2 * kNMom + 2], tot[2 * kNMom + 3]};
        // direct lag 0: sum of the chips' raw zeroth moments = carrier-path M_0 + mean * sum r w_c G(len)
        const f2 corr0 = f2{tot[0], tot[1]} + cmul(mFull, zg);
        // D[l] = exp(-j phi l) ((G[l+1] - G[l]) + mI sum c + edge terms), l = lane - 32 (lanes 0 .. 62)
        const f2 gn = dpp_shl1(G[s]);
        f2 dp = gn - G[s] + cmul(mInt, sc) + DE[s];
        double ph = (double)(lane - 32) * ch[s].carrStep;
        ph -= floor(ph);
        f2 d = cmul(wipe_seed((float)ph), dp);
        if (lane == 63) d = f2{0.f, 0.f};
        float pr = d.x, pi = d.y;
        wave_scan_incl2(pr, pi);
        const float p31r = readlane_f(pr, 31), p31i = readlane_f(pi, 31);
        f2 v = f2{corr0.x + dpp_shr1_f(pr) - p31r, corr0.y + dpp_shr1_f(pi) - p31i};
        const bool again = (spilled[s] >> side) & 1;
        if (again) { const float2 old = partOut[lane]; v += f2{old.x, old.y}; }   // (the same lane wrote it: program order)
        partOut[lane] = make_float2(v.x, v.y);
        if (lane == 0) partOut[64] = make_float2(0.f, 0.f);
        spilled[s] |= 1 << side;
        if (lane < kNMom) {
            float2 o = make_float2(0.f, 0.f);
#pragma unroll
            for (int p = 0; p < kNMom; ++p) if (lane == p) o = make_float2(tot[2 * p], tot[2 * p + 1]);
            if ((flushed[s] >> side) & 1) { const float2 old = momOut[lane]; o.x += old.x; o.y += old.y; }
            momOut[lane] = o;
        }
        flushed[s] |= 1 << side;
        G[s] = SC[s] = ZG[s] = DE[s] = f2{0.f, 0.f};
#pragma unroll
        for (int p = 0; p < kNMom; ++p) M[s][p] = f2{0.f, 0.f};
    };

    // ---- phase 0 of a tile: raw samples -> z = raw - mI -> prefix arrays in LDS.  Thread tid owns the kEPT consecutive
    // samples n0 .. n0 + kEPT - 1 of the tile's range [tb, tb + NTs) (tile-local n <-> window sample tb + n, circularly continued)
    int rawN[kEPT];
    auto fetch_tile = [&](int t0) __attribute__((always_inline)) {
        const int tLen = (S - t0 < Lt) ? S - t0 : Lt, tb = t0 - k3HL, NTs = tLen + k3HL + k3HR;
        const int n0 = kEPT * tid;
        if (tb >= 0 && tb + NTs <= S) {
#pragma unroll
            for (int i = 0; i < kEPT; ++i) rawN[i] = (n0 + i < NTs) ? x[tb + n0 + i] : 0;
        } else {
#pragma unroll
            for (int i = 0; i < kEPT; ++i) {
                int g = tb + n0 + i;
                g = g < 0 ? g + S : (g >= S ? g - S : g);
                rawN[i] = (n0 + i < NTs) ? x[g] : 0;
            }
        }
    };
    auto phase0 = [&](int t0) __attribute__((always_inline)) {
        const int tLen = (S - t0 < Lt) ? S - t0 : Lt, NTs = tLen + k3HL + k3HR;
        const int n0 = kEPT * tid;
        f2 z[kEPT];
        float tot[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < kEPT; ++i) {
            const int rv = rawN[i];
            const bool ok = n0 + i < NTs;
            z[i] = ok ? f2{(float)((int)(short)(rv & 0xFFFF) - mIntRe), (float)((rv >> 16) - mIntIm)} : f2{0.f, 0.f};
            const float wt = (float)(n0 + i - k3Mid);
            tot[0] += z[i].x; tot[1] += z[i].y;
            tot[2] = fmaf(wt, z[i].x, tot[2]); tot[3] = fmaf(wt, z[i].y, tot[3]);
            tot[4] = fmaf(wt * wt, z[i].x, tot[4]); tot[5] = fmaf(wt * wt, z[i].y, tot[5]);
        }
        float inc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) inc[i] = tot[i];
        wave_scan_incl(inc);
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < 6; ++i) sScan[wave * 6 + i] = inc[i];
        }
        __syncthreads();   // (also: every wave is done with the previous tile's arrays)
        float run[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) run[i] = inc[i] - tot[i];
        for (int q = 0; q < wave; ++q) {
#pragma unroll
            for (int i = 0; i < 6; ++i) run[i] += sScan[q * 6 + i];
        }
#pragma unroll
        for (int i = 0; i < kEPT; ++i) {
            const int n = n0 + i;
            if (n <= NTs) {
                sC0[n] = make_float2(run[0], run[1]);
                sW[n] = make_float4(run[2], run[3], run[4], run[5]);
            }
            const float wt = (float)(n - k3Mid);
            run[0] += z[i].x; run[1] += z[i].y;
            run[2] = fmaf(wt, z[i].x, run[2]); run[3] = fmaf(wt, z[i].y, run[3]);
            run[4] = fmaf(wt * wt, z[i].x, run[4]); run[5] = fmaf(wt * wt, z[i].y, run[5]);
        }
        __syncthreads();
    };

    // a chip's zeroth / first moment about its own centre xb:  M_p += xb^(p-1) (xb E0 + p E1)
    auto add_moments = [&](f2 (&Mv)[kNMom], f2 E0, f2 E1, float xb) __attribute__((always_inline)) {
        Mv[0] += E0;
        const f2 xE0 = E0 * xb;
        if (kNMom > 1) Mv[1] += xE0 + E1;
        float xp = xb;
#pragma unroll
        for (int p = 2; p < kNMom; ++p) {
            const float pf = (float)p;
            const f2 F = __builtin_elementwise_fma(E1, f2{pf, pf}, xE0);
            Mv[p] = __builtin_elementwise_fma(F, f2{xp, xp}, Mv[p]);
            xp *= xb;
        }
    };

    // ---- one SV on the tile whose arrays are in LDS
    auto serve = [&](auto sTag, int t0, int cLo, int cHi, auto pureTag) __attribute__((always_inline)) {
        constexpr int s = decltype(sTag)::value;
        constexpr bool kPure = decltype(pureTag)::value;
        const BcsChanDev &c = ch[s];
        const int tLen = (S - t0 < Lt) ? S - t0 : Lt, tb = t0 - k3HL, NTs = tLen + k3HL + k3HR;
        const int nOwn = cHi - cLo;   // chips that start inside the tile (<= k3MaxOwn by the host's choice of Lt)
        if (nOwn <= 0) return;
        const int cc = cLo + lane;
        // first replica index of chip cc (see the second form): fp64 estimate, two-sided check only near an integer
        int e;
        {
            const double xq = ((double)cc - c.rc) * c.invStep;
            const double md = ceil(xq);
            int m = (int)md;
            if (__ballot(fabs((md - xq) - 0.5) > 0.499999) != 0ull) {
                if ((int)fma(md - 1.0, c.codeStep, c.rc) >= cc) m -= 1;
                else if ((int)fma(md, c.codeStep, c.rc) < cc) m += 1;
            }
            e = cc <= c0[s] ? 0 : (cc > cEnd[s] ? S : m);
            const int eMax = tb + NTs;
            e = e > eMax ? eMax : e;   // (lanes beyond the look-ahead lane: keeps their LDS reads inside the arrays)
        }
        const bool own = lane < nOwn;
        const int eNext = __builtin_amdgcn_update_dpp(0, e, 0x130, 0xf, 0xf, true);
        const int len = own ? eNext - e : 0;
        const int la = e - tb;
        // prefix values at the chip's start and -- from the next lane -- at its end
        const float2 c0a_ = sC0[la];
        const float4 wa_ = sW[la];
        const f2 c0a = f2{c0a_.x, c0a_.y}, w1a = f2{wa_.x, wa_.y}, w2a = f2{wa_.z, wa_.w};
        const f2 c0b = dpp_shl1(c0a), w1b = dpp_shl1(w1a), w2b = dpp_shl1(w2a);
        // chip signs: bit i of rMask <-> chip cLo - 1 + i, from the PRN's periodically extended sign-bit table (lane l keeps word l)
        unsigned long long rMask;
        {
            const int st = (cLo - 1 + kLCA) % kLCA, wi = st >> 5, sh = st & 31;
            const unsigned w0 = (unsigned)__builtin_amdgcn_readlane(bitsV[s], wi), w1 = (unsigned)__builtin_amdgcn_readlane(bitsV[s], wi + 1),
                           w2 = (unsigned)__builtin_amdgcn_readlane(bitsV[s], wi + 2);
            const unsigned long long lo = ((unsigned long long)w1 << 32) | (unsigned long long)w0;
            rMask = (lo >> sh) | ((((unsigned long long)w2) << 32) << (32 - sh));
        }
        const unsigned long long ownMask = nOwn >= 64 ? ~0ull : ((1ull << nOwn) - 1ull);
        const float r = mask_pm1(rMask >> 1);   // lane l: sign of chip cLo + l
        // raw moments of the chip about its EXPANSION POINT p = first sample + L1 / 2 (the centre of a chip of L1 + 1 samples, half a
        // sample behind the centre of one of L1: a fixed offset from the boundary, so that one block constant turns the boundary's
        // wipe-off into the point's), from the prefix differences; W1 / W2 carry the weights (n - k3Mid), (n - k3Mid)^2
        const float lenf = (float)len, hL = 0.5f * (float)L1[s];
        const float pc = (float)(la - k3Mid) + hL;                  // p in the weights' coordinate
        const f2 dC = c0b - c0a, dW1 = w1b - w1a, dW2 = w2b - w2a;
        const float dm = 0.5f * (lenf - 1.f) - hL;                   // chip centre - p (0 or -1/2 for the regular lengths)
        const float g2 = lenf * (lenf * lenf - 1.f) * (1.f / 12.f);  // sum over the chip of (n - centre)^2
        const float s1 = lenf * dm, s2 = g2 + s1 * dm;               // sums of (n - p), (n - p)^2 over the chip
        const f2 R0 = dC - mu * lenf;
        const f2 R1 = dW1 - dC * pc - mu * s1;
        const f2 R2 = dW2 - dW1 * (2.f * pc) + dC * (pc * pc) - mu * s2;
        const float ph1 = phi[s], hph2 = 0.5f * ph1 * ph1;
        const f2 E0 = add_mjs(R0 - R2 * hph2, ph1, R1);
        const f2 E1 = add_mjs(R1, ph1, R2);
        const f2 Gj = f2{lenf - hph2 * s2, -ph1 * s1};   // sum over the chip of exp(-j phi (n - p)), to the same order
        // wipe-off at the chip's first sample (the flip coefficient) and at p
        double phe = fma((double)e, c.carrStep, c.ri);
        phe -= floor(phe);
        const f2 we = wipe_seed((float)phe);
        const f2 wc = cmul(we, thP[s]);
        const f2 P0 = cmul(wc, E0), P1 = cmul(wc, E1);
        const float xb = (float)(e - stBase) + hL - xOrigin;
        // the lag sums of a list of flips: G[l] += c C0[e + l] for the 64 lags of the wave (lane <-> l + 32)
        // (list entry = {c.re, c.im, byte offset of C0[e - 32], -}: the coefficient is the entry's first, 64-bit aligned register pair.
        //  All reads of a round of four flips are issued before any is consumed, and two running sums halve the dependent chain.)
        auto lag_sums = [&](f2 &Gv, int nList) __attribute__((always_inline)) {
            const char *base = reinterpret_cast<const char *>(sC0) + 8 * lane;
            f2 Gb = f2{0.f, 0.f};
            for (int i0 = 0; i0 < nList; i0 += 4) {
                float4 ent[4];
                float2 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) ent[u] = sList[i0 + u];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const float2 *>(base + __builtin_bit_cast(int, ent[u].z));
                __builtin_amdgcn_sched_barrier(0);
                // (consecutive flips alternate in sign: a pair stays in ONE sum, so that each sum keeps cancelling as it goes)
                Gv = cmul_add(f2{ent[0].x, ent[0].y}, f2{q[0].x, q[0].y}, Gv);
                Gb = cmul_add(f2{ent[2].x, ent[2].y}, f2{q[2].x, q[2].y}, Gb);
                Gv = cmul_add(f2{ent[1].x, ent[1].y}, f2{q[1].x, q[1].y}, Gv);
                Gb = cmul_add(f2{ent[3].x, ent[3].y}, f2{q[3].x, q[3].y}, Gb);
            }
            Gv += Gb;
        };
        if constexpr (kPure) {
            const float rOwn = own ? r : 0.f;
            add_moments(M[s], P0 * rOwn, P1 * rOwn, xb);
            ZG[s] += cmul(wc, Gj) * rOwn;
            // flips: owned chips whose sign differs from the chip before; J = r_prev - r = -2 r
            const unsigned long long bm = (rMask ^ (rMask >> 1)) & ownMask;
            const int nb = __builtin_popcountll(bm);
            const bool isFlip = __builtin_amdgcn_inverse_ballot_w64(bm);
            const f2 cj = we * (isFlip ? -2.f * r : 0.f);
            SC[s] += cj;
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
            if (lane < 4) sList[nb + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (isFlip) sList[rank] = make_float4(cj.x, cj.y, __builtin_bit_cast(float, 8 * (la - 32)), 0.f);
            __builtin_amdgcn_wave_barrier();
            lag_sums(G[s], (nb + 3) & ~3);
            __builtin_amdgcn_wave_barrier();   // the next tile / SV rewrites the list
        } else {
            // general tile: the nav-bit boundary lies among its chips, or it is the window's first / last tile (circular wrap).
            // The two sides are two masked copies of the replica (:352-367); a flip whose lag window leaves [0, S) takes the
            // explicit difference form with the circular continuation's phase per lane.
            const bool sd1 = c.hasFlip && (cc >= cB[s]);
            // the chip before lane 0's: cLo - 1, or circularly the window's last chip (side 1 when a nav-bit boundary exists)
            const bool first = t0 == 0;
            float rPrev0;
            bool sdPrev0;
            if (first) {
                const int cw = cEnd[s] % kLCA;
                rPrev0 = ((unsigned)__builtin_amdgcn_readlane(bitsV[s], cw >> 5) >> (cw & 31)) & 1u ? 1.f : -1.f;
                sdPrev0 = c.hasFlip != 0;
            } else {
                rPrev0 = (rMask & 1ull) ? 1.f : -1.f;
                sdPrev0 = c.hasFlip && (cLo - 1 >= cB[s]);
            }
            double pS = (double)S * c.carrStep;
            pS -= floor(pS);
            const f2 rhoS = wipe_seed((float)pS);   // exp(-j phi S)
            for (int side = 0; side < 2; ++side) {
                const float rs = (own && (sd1 == (side == 1))) ? r : 0.f;
                float rsPrev = dpp_shr1_f(rs);
                if (lane == 0) rsPrev = (sdPrev0 == (side == 1)) ? rPrev0 : 0.f;
                const float J = own ? rsPrev - rs : 0.f;
                const unsigned long long anyOwn = __ballot(rs != 0.f), bm = __ballot(J != 0.f);
                if (anyOwn == 0ull && bm == 0ull) continue;
                if (side != curSide[s]) {
                    if (curSide[s] >= 0) flush(sTag, curSide[s]);
                    curSide[s] = side;
                }
                add_moments(M[s], P0 * rs, P1 * rs, xb);
                ZG[s] += cmul(wc, Gj) * rs;
                const f2 cj = we * J;
                // flips whose lag window stays inside the window: list + G form
                const bool inside = e >= 32 && e + 31 <= S;
                const unsigned long long bmIn = __ballot(J != 0.f && inside), bmEdge = bm & ~bmIn;
                SC[s] += (J != 0.f && inside) ? cj : f2{0.f, 0.f};
                const int nb = __builtin_popcountll(bmIn);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bmIn >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bmIn, 0));
                if (lane < 4) sList[nb + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((bmIn >> lane) & 1ull) sList[rank] = make_float4(cj.x, cj.y, __builtin_bit_cast(float, 8 * (la - 32)), 0.f);
                __builtin_amdgcn_wave_barrier();
                lag_sums(G[s], (nb + 3) & ~3);
                __builtin_amdgcn_wave_barrier();
                // the others, one at a time: D'[l] += c_lane (z[e + l] + mI), c_lane = c exp(-+ j phi S) where e + l wraps
                unsigned long long rest = bmEdge;
                while (rest) {
                    const int fl = __builtin_ctzll(rest);
                    rest &= rest - 1;
                    const int ef = __builtin_amdgcn_readlane(e, fl), laf = ef - tb;
                    const f2 cf = f2{readlane_f(cj.x, fl), readlane_f(cj.y, fl)};
                    const int idx = ef + lane - 32;
                    const float2 qa = sC0[laf + lane - 32], qb = sC0[laf + lane - 31];
                    const f2 xs = f2{qb.x - qa.x, qb.y - qa.y} + mInt;
                    f2 cl = cf;
                    if (idx < 0) cl = cmul(cf, rhoS);
                    else if (idx >= S) cl = cmul(cf, f2{rhoS.x, -rhoS.y});
                    if (lane < 63) DE[s] = cmul_add(cl, xs, DE[s]);
                }
            }
        }
    };

    // ---- the block's tiles
    const int tFirst = blk * T;
    const int nTilesW = (S + Lt - 1) / Lt;
    int tEnd = tFirst + T;
    if (tEnd > nTilesW) tEnd = nTilesW;
    if (tFirst < tEnd) fetch_tile(tFirst * Lt);
    for (int t = tFirst; t < tEnd; ++t) {
        const int t0 = t * Lt;
        phase0(t0);
        if (t + 1 < tEnd) fetch_tile((t + 1) * Lt);   // the next tile's samples arrive under this tile's per-SV work
        const int tLen = (S - t0 < Lt) ? S - t0 : Lt;
        static_for<NSV>([&](auto sTag) __attribute__((always_inline)) {
            constexpr int s = decltype(sTag)::value;
            __builtin_amdgcn_sched_barrier(0);   // (the SVs of a wave one after the other: interleaved, their temporaries add up)
            if (!live[s]) return;
            // pure tile: every chip it touches (the one before its first included) lies on one side of the nav-bit boundary, and
            // no flip's lag window leaves the window
            const BcsChanDev &c = ch[s];
            const int cLo = t0 == 0 ? c0[s] : __builtin_amdgcn_readfirstlane((int)fma((double)(t0 - 1), c.codeStep, c.rc)) + 1;
            const int cHi = t0 + tLen >= S ? cEnd[s] + 1 : __builtin_amdgcn_readfirstlane((int)fma((double)(t0 + tLen - 1), c.codeStep, c.rc)) + 1;
            const bool edgeTile = t0 < 32 || t0 + tLen + 31 > S;
            const bool oneSide = !c.hasFlip || cB[s] > cHi - 1 || cB[s] <= cLo - 1;
            if (oneSide && !edgeTile) {
                const int side = (c.hasFlip && cB[s] <= cLo - 1) ? 1 : 0;
                if (side != curSide[s]) {
                    if (curSide[s] >= 0) flush(sTag, curSide[s]);
                    curSide[s] = side;
                }
                serve(sTag, t0, cLo, cHi, std::true_type{});
            } else {
#ifndef DPE_C3_NOGENERAL
                serve(sTag, t0, cLo, cHi, std::false_type{});
#endif
            }
        });
    }
    // ---- block partials: what is still in registers, then zeros for a side that never occurred
    static_for<NSV>([&](auto sTag) __attribute__((always_inline)) {
        constexpr int s = decltype(sTag)::value;
        if (!live[s]) return;
        if (curSide[s] >= 0) flush(sTag, curSide[s]);
        const int k = (kg * NSV + s) * NW + wave;
        for (int side = 0; side < 2; ++side) {
            if (!((spilled[s] >> side) & 1)) {
                float2 *partOut = part + ((((size_t)w * K + k) * nBlk + blk) * 2) * NL + side * NL;
                partOut[lane] = make_float2(0.f, 0.f);
                if (lane == 0) partOut[64] = make_float2(0.f, 0.f);
                float2 *momOut = mom + ((((size_t)w * K + k) * 2) * nBlk + blk) * kNMom + (size_t)side * nBlk * kNMom;
                if (lane < kNMom) momOut[lane] = make_float2(0.f, 0.f);
            }
        }
    });
}

}  // namespace dpe
