// dpe_acq_pack.h -- the reference's non-coherent search (coarse_acquisition(coherent=False), correlator.py:77-82) at 10 x 2 500
// samples: the forward and the inverse 25 000-point transforms as ten PACKED 2 500-point transforms per block.  Included by dpe_acq.hip
// (namespace dpe, after acq_cmul / acq_idft5 / acq_idft10).
//
//   surface[p][b][j] = sum_{n<10} | y[j + 2500 n] |,   y = IFFT_25000( X_b .* Rc_p )            (N = 25 000, M = 2 500)
//
// Decimation in TIME by ten: with k = 10 k' + q,
//   y[j + M n] = sum_q e^{2 pi i q n / 10} * [ e^{2 pi i q j / N} * U_q[j] ],   U_q = IFFT_2500( P[10 k' + q] )   (unnormalised; 1/N is in Rc)
// so a (PRN, bin) is ten independent 2 500-point transforms of the ten residue classes of the product spectrum, a twiddle, and a
// ten-point transform ACROSS them at every delay j -- whose ten outputs are exactly the ten lag aliases of j, summed in magnitude
// on the spot.  (The form this replaces decimated in frequency: its radix-10 stage came FIRST and its 200 KB of output per (PRN,
// bin) had to pass through memory.  Here nothing but the surface leaves the block.)  The spectra arrive decimated --
// Xq[b][q][k'] = X_b[10 k' + q], written in that order by acq_fwd25k_pack_kernel; the replicas' the same at create -- so every load is
// contiguous.
//
// One 2 500-point transform = 50 lanes x 50 registers: lane a < 50 holds P[a + 50 b'] (b' < 50), runs a 50-point transform over b'
// (prime-factor 2 x 25, the 25 as 5 x 5: every index is a compile-time constant, the data never leaves the registers), multiplies by
// W2500^(a c) (table in LDS, [c][a]), transposes through LDS (by halves: the even output columns c first, then the odd ones, 10 KB per
// transform) and runs the second 50-point transform over a: lane c then holds U[c + 50 d], d < 50.  The 500 lanes of an item's ten
// transforms are threads 0 .. 499 of a 512-thread block (eight waves, two per SIMD, up to 256 registers each).  The ten transforms meet
// in the exchange buffer (laid over the transposes: E[q][1250] for the even d, then the odd d), after which each thread takes delays
// from the flat range and does twiddle, ten-point transform, magnitudes, sum, store, atomic maximum over the bins.  Eight barriers per
// (PRN, bin) in all (the four-pass 250-thread transform of acq_corr2500_kernel needs five per TRANSFORM).
// Measured on MI355X, 32 PRNs x 125 bins (profiles/r5_ab_acq_noncoherent.txt): radix-10 stage + four-pass transforms 0.62 ms per search
// (0.57 with an XCD-aware block order); this decomposition with one WAVE per transform (ten waves of 50 lanes, 3 + 3 + 2 + 2 on the
// SIMDs, 168 registers: the transposes' 150 live values spill) 0.43; packed 0.36; with the forward kernel below 0.31.
#pragma once

namespace dpe {

constexpr int kPkRow = 50;                 // row stride (float2) of a half transpose: 400 bytes, so a row is read back in 16-byte pieces, and the rows
                                           // r .. r + 15 of a quarter wave start on distinct groups of four banks (100 r mod 64 = 36 r: 0.3045 -> 0.296 ms
                                           // against a stride of 51 with 8-byte reads)
constexpr int kPkBuf = 25 * kPkRow;        // float2 per transform
constexpr int kPkTransforms = 10;
constexpr int kPkHalf = 1250;              // delays per exchange round
constexpr size_t kPkLdsBytes = ((size_t)kPkTransforms * kPkBuf + 2500 + 1000) * sizeof(float2);   // 128 000 B: one block per CU
static_assert(kPkTransforms * kPkBuf >= kPkTransforms * kPkHalf, "the exchange buffer lies over the transposes");

// complex product with the second factor in scalar registers (compile-time twiddles: two literals, no vector register)
__device__ __forceinline__ af2 acq_cmul_s(af2 a, af2 w)
{
    af2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}

// exp(+j 2 pi m / 25), m <= 16 (= 4 x 4)
__device__ constexpr float kW25c[17] = {1.f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899655f,
                                        0.30901699437494745f, 0.062790519529313527f, -0.1873813145857246f, -0.42577929156507272f,
                                        -0.63742398974868975f, -0.80901699437494734f, -0.92977648588825135f, -0.99211470131447776f,
                                        -0.99211470131447788f, -0.92977648588825146f, -0.80901699437494778f, -0.63742398974868952f};
__device__ constexpr float kW25s[17] = {0.f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868862f, 0.84432792550201508f,
                                        0.95105651629515353f, 0.99802672842827156f, 0.98228725072868872f, 0.90482705246601947f,
                                        0.77051324277578925f, 0.58778525229247325f, 0.36812455268467814f, 0.12533323356430454f,
                                        -0.12533323356430429f, -0.36812455268467792f, -0.58778525229247269f, -0.77051324277578936f};

// inverse 25-point DFT (unnormalised) in place, 5 x 5: input n at v[n]; OUTPUT k sits at v[acq_pos25(k)]
__device__ __forceinline__ void acq_idft25(af2 (&v)[25])
{
#pragma unroll
    for (int n1 = 0; n1 < 5; ++n1) {
        acq_idft5(v[n1], v[n1 + 5], v[n1 + 10], v[n1 + 15], v[n1 + 20]);   // over n2: v[n1 + 5 k2]
        if (n1) {
#pragma unroll
            for (int k2 = 1; k2 < 5; ++k2) v[n1 + 5 * k2] = acq_cmul_s(v[n1 + 5 * k2], af2{kW25c[n1 * k2], kW25s[n1 * k2]});
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) acq_idft5(v[5 * k2], v[5 * k2 + 1], v[5 * k2 + 2], v[5 * k2 + 3], v[5 * k2 + 4]);   // over n1: v[k1 + 5 k2]
}
__host__ __device__ constexpr int acq_pos25(int k) { return k / 5 + 5 * (k % 5); }
// 50 = 2 x 25 prime-factor: inputs n = (25 n1 + 2 n2) mod 50, outputs k = (25 k1 + 26 k2) mod 50 -- S = the n1 sums feeds k1 = 0
// (the EVEN outputs), D = the differences feeds k1 = 1 (the ODD outputs)
__host__ __device__ constexpr int acq_in50(int n1, int n2) { return (25 * n1 + 2 * n2) % 50; }
__host__ __device__ constexpr int acq_out50(int k1, int k2) { return (25 * k1 + 26 * k2) % 50; }

// Xq[row][q][k'] = X[row][10 k' + q]: 640 consecutive values through LDS, ten runs of 64 out
__global__ __launch_bounds__(640) void acq_decimate10_kernel(const float2 *__restrict__ X, float2 *__restrict__ Xq)
{
    __shared__ float2 s[640];
    const int t = threadIdx.x, k0 = blockIdx.x * 64;
    const size_t row = (size_t)blockIdx.y * 25000;
    const int i = 10 * k0 + t;
    if (i < 25000) s[t] = X[row + i];
    __syncthreads();
    const int q = t >> 6, kk = t & 63;
    if (k0 + kk < 2500) Xq[row + (size_t)q * 2500 + k0 + kk] = s[kk * 10 + q];
}

// The packed 2 500-point inverse transforms of a block (unnormalised; thread L < 500 = lane a = L mod 50 of transform q = L / 50; 512 threads).
// In: S[n2] / D[n2] = sum / difference of the inputs P[a + 50 acq_in50(0 / 1, n2)] of this lane (the radix-2 stage of the first 50-point
// transform).  Out: lane c of transform q holds U_q[c + 50 d]: S[acq_pos25(k2)] for the even d = acq_out50(0, k2), D[...] for the odd d =
// acq_out50(1, k2).  lds = the block's transposes (kPkTransforms x kPkBuf), sTw = W2500^(a c) as [c][a].  Four barriers; the first one
// separates whatever the caller last did with the transposes' LDS from the first store here.
#define DPE_PK_LANE_(q, a, act)                     \
    int L_ = tid;                                   \
    asm volatile("" : "+v"(L_));                    \
    const bool act = L_ < 500;                      \
    L_ = act ? L_ : 499;                            \
    const int q = L_ / 50, a = L_ - 50 * q;         \
    (void)a; (void)act; (void)q
__device__ __forceinline__ void acq_pack_transform2500(af2 (&S)[25], af2 (&D)[25], const int tid, float2 *acqPkLds, const float2 *sTw)
{
    acq_idft25(S);
    acq_idft25(D);
    __syncthreads();   // the readers of the previous item are through with the exchange buffer, which lies over the transposes
    // even columns out, twiddled; every lane reads back the row of its (even) column -- the odd lanes read again below.  (All the twiddles
    // of a half are fetched BEFORE its first store: the table and the transposes are one LDS array to the compiler, and a load behind a
    // store that may alias it waits for nothing but is not moved up either -- 25 exposed LDS round trips per half otherwise.)
    af2 Bv[50];
    {
        DPE_PK_LANE_(q, a, act);
        float2 *sT = acqPkLds + q * kPkBuf;
        float2 w[25];
#pragma unroll
        for (int k2 = 0; k2 < 25; ++k2) w[k2] = sTw[acq_out50(0, k2) * 50 + a];
#pragma unroll
        for (int k2 = 0; k2 < 25; ++k2) S[acq_pos25(k2)] = acq_cmul(S[acq_pos25(k2)], af2{w[k2].x, w[k2].y});
#pragma unroll
        for (int k2 = 0; k2 < 25; ++k2) w[k2] = sTw[acq_out50(1, k2) * 50 + a];
        if (act) {
#pragma unroll
            for (int k2 = 0; k2 < 25; ++k2) sT[(acq_out50(0, k2) >> 1) * kPkRow + a] = make_float2(S[acq_pos25(k2)].x, S[acq_pos25(k2)].y);
        }
#pragma unroll
        for (int k2 = 0; k2 < 25; ++k2) D[acq_pos25(k2)] = acq_cmul(D[acq_pos25(k2)], af2{w[k2].x, w[k2].y});
        __syncthreads();   // the even columns of every transform are in place
        const float4 *row = reinterpret_cast<const float4 *>(sT + (a >> 1) * kPkRow);   // (16-byte aligned: 400 bytes per row)
#pragma unroll
        for (int i = 0; i < 25; ++i) {
            const float4 v = row[i];
            Bv[2 * i] = af2{v.x, v.y};
            Bv[2 * i + 1] = af2{v.z, v.w};
        }
        __syncthreads();   // ... and read
        if (act) {
#pragma unroll
            for (int k2 = 0; k2 < 25; ++k2) sT[(acq_out50(1, k2) >> 1) * kPkRow + a] = make_float2(D[acq_pos25(k2)].x, D[acq_pos25(k2)].y);
        }
        __syncthreads();
        if (a & 1) {
#pragma unroll
            for (int i = 0; i < 25; ++i) {
                const float4 v = row[i];
                Bv[2 * i] = af2{v.x, v.y};
                Bv[2 * i + 1] = af2{v.z, v.w};
            }
        }
    }
    // second 50-point transform, over a: S -> U[c + 50 d] for the even d, D -> the odd d
#pragma unroll
    for (int n2 = 0; n2 < 25; ++n2) {
        S[n2] = Bv[acq_in50(0, n2)] + Bv[acq_in50(1, n2)];
        D[n2] = Bv[acq_in50(0, n2)] - Bv[acq_in50(1, n2)];
    }
    acq_idft25(S);
    acq_idft25(D);
}
#undef DPE_PK_LANE_

// Wipe-off AND the forward 25 000-point transform of a Doppler bin in one block, written in the decimated order the search kernel reads:
//   Xq[b][q][k'] = sum_n raw[n] e^{-2 pi i f_b n / fs} e^{-2 pi i n (10 k' + q) / 25000}
// (replaces acq_wipe_kernel + rocFFT's four kernels + acq_decimate10_kernel: 73 us of a 0.36 ms search).  Decimation in FREQUENCY by
// ten: with n = n' + 2500 m,  X[10 k' + q] = FFT_2500( z_q )[k'],  z_q[n'] = W25000^(n' q) sum_m x[n' + 2500 m] W10^(m q)  -- a ten-point
// transform over the ten samples 2 500 apart (thread n' of a chunk of 500 wipes them itself: the same fp64 phase and v_sin / v_cos as
// acq_wipe_kernel), a twiddle (the search kernel's two tables), then the ten packed 2 500-point transforms.  Everything runs on the
// CONJUGATE through the inverse machinery (FFT(x) = conj(IFFT_unnormalised(conj(x)))).  The ten outputs of a thread belong to ten
// different transforms: they cross through LDS in five chunks of 500 n' (two 40 KB buffers over the transposes, one barrier per chunk).
__global__ __launch_bounds__(512) void acq_fwd25k_pack_kernel(const int16_t *__restrict__ iq, double binStart, double binStep, double invFs,
                                                              const float2 *__restrict__ tw2, const float2 *__restrict__ tw25k, float2 *__restrict__ Xq,
                                                              float *__restrict__ mp, long long mpLen)
{
    extern __shared__ float2 acqPkLds[];
    float2 *sTw = acqPkLds + kPkTransforms * kPkBuf;
    float2 *sT1 = sTw + 2500, *sT2 = sT1 + 500;
    acq_clear(mp, mpLen);
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int i = tid; i < 2500; i += 512) sTw[i] = tw2[i];
    if (tid < 500) {
        const int tq = tid / 50, tc = tid - 50 * tq;
        sT1[tid] = tw25k[tq * tc];
        sT2[tid] = tw25k[50 * tq * tc];
    }
    const double cyclesPerSample = (binStart + binStep * b) * invFs;
    const int *x = reinterpret_cast<const int *>(iq);
    const bool act = tid < 500;
    const int L = act ? tid : 499, q = L / 50, a = L - 50 * q;
    af2 Z[50];
    int rawN[10];   // the next chunk's samples, requested a chunk ahead (one exposed memory latency instead of five)
#pragma unroll
    for (int m = 0; m < 10; ++m) rawN[m] = x[(act ? tid : 0) + 2500 * m];
    __syncthreads();   // (the tables)
#pragma unroll
    for (int ch = 0; ch < 5; ++ch) {
        float2 *E = acqPkLds + (ch & 1) * 5000;   // [q][500]
        int rawC[10];
#pragma unroll
        for (int m = 0; m < 10; ++m) rawC[m] = rawN[m];
        if (ch < 4) {
#pragma unroll
            for (int m = 0; m < 10; ++m) rawN[m] = x[500 * (ch + 1) + (act ? tid : 0) + 2500 * m];
        }
        if (act) {
            const int n1 = 500 * ch + tid;
            af2 v[10];
#pragma unroll
            for (int m = 0; m < 10; ++m) {
                const int i = n1 + 2500 * m;
                const int raw = rawC[m];
                const float re = (float)(short)(raw & 0xFFFF), im = (float)(raw >> 16);
                double ph = cyclesPerSample * (double)i;
                ph -= floor(ph);
                const float f = (float)ph;
                const float c = __builtin_amdgcn_cosf(f), sn = -__builtin_amdgcn_sinf(f);
                v[m] = af2{re * c - im * sn, -(re * sn + im * c)};   // conj( raw exp(-j 2 pi f n / fs) )   (correlator.py:63)
            }
            acq_idft10(v);
            const int bb = tid / 50, aa = tid - 50 * bb;   // n' = aa + 50 (10 ch + bb)
#pragma unroll
            for (int qq = 1; qq < 10; ++qq) {
                const float2 f1 = sT1[qq * 50 + aa], f2 = sT2[qq * 50 + 10 * ch + bb];
                v[qq] = acq_cmul(v[qq], acq_cmul(af2{f1.x, f1.y}, af2{f2.x, f2.y}));   // W25000^(+n' q), conjugate domain
            }
#pragma unroll
            for (int qq = 0; qq < 10; ++qq) E[qq * 500 + tid] = make_float2(v[qq].x, v[qq].y);
        }
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 10; ++bb) {
            const float2 u = E[q * 500 + 50 * bb + a];
            Z[10 * ch + bb] = af2{u.x, u.y};
        }
    }
    af2 S[25], D[25];
#pragma unroll
    for (int n2 = 0; n2 < 25; ++n2) {
        S[n2] = Z[acq_in50(0, n2)] + Z[acq_in50(1, n2)];
        D[n2] = Z[acq_in50(0, n2)] - Z[acq_in50(1, n2)];
    }
    acq_pack_transform2500(S, D, tid, acqPkLds, sTw);
    if (act) {
        float2 *xo = Xq + ((size_t)b * 10 + q) * 2500 + a;
#pragma unroll
        for (int k2 = 0; k2 < 25; ++k2) {
            xo[50 * acq_out50(0, k2)] = make_float2(S[acq_pos25(k2)].x, -S[acq_pos25(k2)].y);
            xo[50 * acq_out50(1, k2)] = make_float2(D[acq_pos25(k2)].x, -D[acq_pos25(k2)].y);
        }
    }
}

constexpr int kPkNA = 16;   // radix-2 pairs of the NEXT item whose spectra are requested before the last reader phase of the current one
__global__ __launch_bounds__(512) void acq_corr25k_pack_kernel(const float2 *__restrict__ Xq, const float2 *__restrict__ Rcq,
                                                               const float2 *__restrict__ tw2, const float2 *__restrict__ tw25k, int B, int nP, int pOffset,
                                                               float *__restrict__ surf, unsigned int *__restrict__ mpBits, int xcdMap)
{
    extern __shared__ float2 acqPkLds[];
    float2 *sTw = acqPkLds + kPkTransforms * kPkBuf;   // [c][a] = W2500^(a c)
    float2 *sT1 = sTw + 2500, *sT2 = sT1 + 500;   // W25000^(q c) as [q][c] and W500^(q d) as [q][d]: the twiddle of delay j = c + 50 d in front of the ten-point transform
    const int tid = threadIdx.x;
    // THE FIVE HUNDRED LANES OF AN ITEM'S TEN TRANSFORMS PACKED INTO EIGHT WAVES: thread L < 500 is lane a = L mod 50 of transform q = L / 50
    // (one wave per transform would use 50 of 64 lanes and put ten waves on four SIMDs, 3 + 3 + 2 + 2).  A transform straddles two
    // waves, so the transposes are ordered by block barriers; with two waves per SIMD a wave may use 256 registers, which the transposes
    // (150 live) and the prefetch of the next item's spectra want.
    // Lane-dependent indices and addresses are re-derived inside each phase from an opaque copy of tid: hoisted out of the item loop (as
    // the compiler does with anything loop-invariant) they would occupy ~20 registers around the whole loop.
#define DPE_PK_LANE(q, a, act)                      \
    int L_ = tid;                                   \
    asm volatile("" : "+v"(L_));                    \
    const bool act = L_ < 500;                      \
    L_ = act ? L_ : 499;                            \
    const int q = L_ / 50, a = L_ - 50 * q;         \
    (void)a; (void)act; (void)q
    // PERSISTENT blocks, one per CU (the 130 KB of LDS see to that), each walking its share of the (PRN, bin) items.  xcdMap: the block's
    // index mod 8 is its XCD; XCD x takes the PRNs p = x (mod 8) and walks the bins with all of them together -- a PRN's spectrum stays
    // in that XCD's L2 for the whole launch, a bin's is fetched once per XCD.
    unsigned s, sStep, sEnd, ppx, pBase, pMul;
    if (xcdMap) { s = blockIdx.x >> 3; sStep = gridDim.x >> 3; ppx = (unsigned)nP >> 3; pBase = blockIdx.x & 7u; pMul = 8; }
    else { s = blockIdx.x; sStep = gridDim.x; ppx = (unsigned)nP; pBase = 0; pMul = 1; }
    sEnd = ppx * (unsigned)B;
    if (s >= sEnd) return;
    for (int i = tid; i < 2500; i += 512) sTw[i] = tw2[i];
    if (tid < 500) {
        const int tq = tid / 50, tc = tid - 50 * tq;
        sT1[tid] = tw25k[tq * tc];
        sT2[tid] = tw25k[50 * tq * tc];
    }
    const int L0 = tid < 500 ? tid : 499, q0 = L0 / 50, a0 = L0 - 50 * q0;
    int p = (int)(pBase + pMul * (s % ppx)), b = (int)(s / ppx);
    const float2 *xq = Xq + ((size_t)b * 10 + q0) * 2500 + a0;
    const float2 *rq = Rcq + ((size_t)p * 10 + q0) * 2500 + a0;
    float2 ax0[kPkNA], ar0[kPkNA], ax1[kPkNA], ar1[kPkNA];
#pragma unroll
    for (int n2 = 0; n2 < kPkNA; ++n2) {
        ax0[n2] = xq[50 * acq_in50(0, n2)]; ar0[n2] = rq[50 * acq_in50(0, n2)];
        ax1[n2] = xq[50 * acq_in50(1, n2)]; ar1[n2] = rq[50 * acq_in50(1, n2)];
    }
    float mx[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};   // running maxima over the bins of this thread's six delays (the same six for every item)
    __syncthreads();   // (the tables)
    for (;;) {
        af2 S[25], D[25];
        // product spectrum of this lane's residue class, and the radix-2 stage of the first 50-point transform on the way in.  The first
        // kPkNA pairs were requested during the previous item; the rest comes in three groups, each requested before the group before it is
        // consumed (the scheduling fences keep the compiler from requesting all sixty at once, which does not fit the registers)
        {
            constexpr int G1 = kPkNA + (25 - kPkNA + 2) / 3, G2 = G1 + (25 - kPkNA + 1) / 3;
            float2 bx0[25], br0[25], bx1[25], br1[25];   // (indexed by n2; only [kPkNA, 25) is used)
            const auto request = [&](int lo, int hi) {
#pragma unroll
                for (int n2 = lo; n2 < hi; ++n2) {
                    bx0[n2] = xq[50 * acq_in50(0, n2)]; br0[n2] = rq[50 * acq_in50(0, n2)];
                    bx1[n2] = xq[50 * acq_in50(1, n2)]; br1[n2] = rq[50 * acq_in50(1, n2)];
                }
            };
            const auto consume = [&](int lo, int hi) {
#pragma unroll
                for (int n2 = lo; n2 < hi; ++n2) {
                    const af2 p0 = acq_cmul(af2{bx0[n2].x, bx0[n2].y}, af2{br0[n2].x, br0[n2].y}), p1 = acq_cmul(af2{bx1[n2].x, bx1[n2].y}, af2{br1[n2].x, br1[n2].y});
                    S[n2] = p0 + p1;
                    D[n2] = p0 - p1;
                }
            };
#pragma unroll
            for (int n2 = 0; n2 < kPkNA; ++n2) { bx0[n2] = ax0[n2]; br0[n2] = ar0[n2]; bx1[n2] = ax1[n2]; br1[n2] = ar1[n2]; }
            request(kPkNA, G1);
            __builtin_amdgcn_sched_barrier(0);
            consume(0, kPkNA);
            request(G1, G2);
            __builtin_amdgcn_sched_barrier(0);
            consume(kPkNA, G1);
            request(G2, 25);
            __builtin_amdgcn_sched_barrier(0);
            consume(G1, G2);
            consume(G2, 25);
        }
        acq_pack_transform2500(S, D, tid, acqPkLds, sTw);   // (its first barrier: the readers of the previous item are through with the exchange buffer)
        const size_t rowOut = ((size_t)(pOffset + p) * B + b) * 2500;
        unsigned int *mpRow = mpBits + (size_t)(pOffset + p) * 2500;
        const unsigned sNext = s + sStep;
        const bool more = sNext < sEnd;
        const int pCur = p;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            __syncthreads();   // r = 0: every transform is through with its transposes; r = 1: every reader is through with the even half
            DPE_PK_LANE(q, a, act);
            if (act) {
                float2 *sE = acqPkLds + q * kPkHalf;
#pragma unroll
                for (int k2 = 0; k2 < 25; ++k2) {
                    const int e = acq_out50(r, k2) >> 1;   // d = 2 e + r
                    const af2 y = r ? D[acq_pos25(k2)] : S[acq_pos25(k2)];
                    sE[e * 50 + a] = make_float2(y.x, y.y);
                }
            }
            if (r == 1) {
                // the transform's registers are free: the first spectra of the next item travel under the last reader phase.  (Unconditional --
                // the last item requests its own again: a conditional request would keep the OLD values alive around the whole loop.)
                const unsigned sn = more ? sNext : s;
                p = (int)(pBase + pMul * (sn % ppx)); b = (int)(sn / ppx);
                xq = Xq + ((size_t)b * 10 + q) * 2500 + a;
                rq = Rcq + ((size_t)p * 10 + q) * 2500 + a;
#pragma unroll
                for (int n2 = 0; n2 < kPkNA; ++n2) {
                    ax0[n2] = xq[50 * acq_in50(0, n2)]; ar0[n2] = rq[50 * acq_in50(0, n2)];
                    ax1[n2] = xq[50 * acq_in50(1, n2)]; ar1[n2] = rq[50 * acq_in50(1, n2)];
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                int i = tid;
                asm volatile("" : "+v"(i));
                i += it * 512;
                if (i < kPkHalf) {
                    const int e = i / 50, c = i - 50 * e, d = 2 * e + r, j = c + 50 * d;   // i = 50 e + c
                    af2 v[10];
                    float2 f1[10], f2[10];
#pragma unroll
                    for (int qq = 0; qq < 10; ++qq) {
                        const float2 u = acqPkLds[qq * kPkHalf + i];
                        v[qq] = af2{u.x, u.y};
                    }
#pragma unroll
                    for (int qq = 1; qq < 10; ++qq) { f1[qq] = sT1[qq * 50 + c]; f2[qq] = sT2[qq * 50 + d]; }
#pragma unroll
                    for (int qq = 1; qq < 10; ++qq) v[qq] = acq_cmul(v[qq], acq_cmul(af2{f1[qq].x, f1[qq].y}, af2{f2[qq].x, f2[qq].y}));   // W25000^(q j)
                    acq_idft10(v);
                    float sv = 0.f;
#pragma unroll
                    for (int n = 0; n < 10; ++n) sv += __builtin_amdgcn_sqrtf(v[n].x * v[n].x + v[n].y * v[n].y);   // the ten lag aliases of delay j (correlator.py:80-82)
                    surf[rowOut + j] = sv;
                    mx[r][it] = fmaxf(mx[r][it], sv);   // max over the bins (:87)
                }
            }
        }
        // the maxima go out when the block leaves the PRN (with the XCD-aware order a block keeps ONE PRN for all its bins: 2 500 atomics
        // per block instead of 2 500 per item -- ten million per 32 x 125 search)
        if (!more || p != pCur) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    int i = tid;
                    asm volatile("" : "+v"(i));
                    i += it * 512;
                    if (i < kPkHalf) {
                        const int e = i / 50;
                        atomicMax(&mpRow[i + 50 * e + 50 * r], __float_as_uint(mx[r][it]));
                    }
                    mx[r][it] = 0.f;
                }
        }
        if (!more) break;
        s = sNext;
    }
#undef DPE_PK_LANE
}

}  // namespace dpe
