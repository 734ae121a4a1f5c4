// dpe_acq_mixed.h -- the fused coherent / textbook search (product spectrum, inverse transform, |.|, column maximum in one kernel) at
// code-period lengths other than the reference's 2 500: 4 000 (4 Msps) and 5 000 (5 Msps).  Included by dpe_acq.hip (namespace dpe,
// after acq_cmul / acq_idft5 / acq_idft10).  Correlator.coarse_acquisition (pygnss/pythonreceiver/scalar/correlator.py:53-103) is
// rate-agnostic; without this those rates ran multiply kernel + inverse rocFFT + fold kernel (2.5 - 3 x the fused time at 2.5 Msps).
//
// One generic four-pass decimation-in-frequency transform M = R1 R2 R3 R4 in LDS, IN PLACE (a butterfly reads and writes the same R
// positions, so a pass needs one barrier and no second buffer), M / 10 threads:
//   pass s, radix R on sub-sequences of length LS (LS = M, M / R1, ...):  butterfly (sub, tt), tt < LS / R, over the elements
//   sub LS + tt + (LS / R) q;  output k is multiplied by W_LS^(tt k) and stored at sub LS + tt + (LS / R) k.
//   After the last pass position p = k1 (M / R1) + k2 (M / (R1 R2)) + k3 R4 + k4 holds output k1 + R1 (k2 + R2 (k3 + R3 k4)): the last
//   pass knows every output's natural index and writes the magnitudes in natural order.
// (Index scheme checked against numpy.fft.ifft in fp64 before it was written in HIP: scripts/proto/acq_mixed_radix_proto.py, 2e-13.)  The tuned 2 500-point
// kernel (acq_corr2500_kernel: padded sub-sequence strides, ping-pong buffers) stays for the reference's rate.
#pragma once

namespace dpe {

// inverse 8-point DFT (e^{+j 2 pi n k / 8}, unnormalised), in place, natural order
__device__ __forceinline__ void acq_idft8(af2 (&v)[8])
{
    constexpr float h = 0.70710678118654752f;
    const af2 a0 = v[0] + v[4], a1 = v[0] - v[4], a2 = v[2] + v[6], a3 = v[2] - v[6];
    const af2 a4 = v[1] + v[5], a5 = v[1] - v[5], a6 = v[3] + v[7], a7 = v[3] - v[7];
    const af2 b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + acq_jrot(a3), b3 = a1 - acq_jrot(a3);
    const af2 b4 = a4 + a6, b6 = a4 - a6, b5 = a5 + acq_jrot(a7), b7 = a5 - acq_jrot(a7);
    const af2 c5 = af2{(b5.x - b5.y) * h, (b5.x + b5.y) * h};     // e^{+j pi / 4} b5
    const af2 c7 = af2{(-b7.x - b7.y) * h, (b7.x - b7.y) * h};    // e^{+j 3 pi / 4} b7
    const af2 c6 = acq_jrot(b6);
    v[0] = b0 + b4; v[4] = b0 - b4;
    v[1] = b1 + c5; v[5] = b1 - c5;
    v[2] = b2 + c6; v[6] = b2 - c6;
    v[3] = b3 + c7; v[7] = b3 - c7;
}
template <int R> __device__ __forceinline__ void acq_idft(af2 (&v)[R]);
template <> __device__ __forceinline__ void acq_idft<5>(af2 (&v)[5]) { acq_idft5(v[0], v[1], v[2], v[3], v[4]); }
template <> __device__ __forceinline__ void acq_idft<8>(af2 (&v)[8]) { acq_idft8(v); }
template <> __device__ __forceinline__ void acq_idft<10>(af2 (&v)[10]) { acq_idft10(v); }

template <int M, int R2, int R3, int R4>
struct AcqMixedShape {
    static constexpr int R1 = 10, T = M / 10, L2 = M / R1, L3 = L2 / R2;
    static_assert(R1 * R2 * R3 * R4 == M, "radices");
    static_assert((M / R4) % T == 0, "the last pass gives every thread the same number of outputs (the textbook mode's accumulators)");
    static constexpr int NB4 = (M / R4) / T;
    static constexpr size_t ldsBytes = sizeof(float2) * (size_t)(M + L2 + L3) + sizeof(float) * (size_t)M;
};

// nSeg = 1: X[b][M] holds the time-folded window (coherent mode); nSeg = N: X[b][N][M] the N code periods (textbook mode), magnitudes
// summed in registers before they go out -- as acq_corr2500_kernel<false>.  tw = exp(+j 2 pi n / M), M entries.
// ALIAS (the reference's coherent = False at 10 x M samples, correlator.py:77-82): X = Z[p][b][k0][M], the output of acq_radix10_kernel -- product and
// first decimation-in-frequency stage of the 10 M-point transform already applied -- nSeg = 10 transforms per (PRN, bin), no multiply; the ten lag
// aliases of delay j = 10 r + k0 are the outputs r + (M / 10) n of transform k0: their magnitudes are summed and written at stride 10 (as
// acq_corr2500_kernel<true>).  pOffset: first PRN of the chunk Z holds.
template <int M, int R2, int R3, int R4, bool ALIAS>
__global__ __launch_bounds__(M / 10) __attribute__((amdgpu_waves_per_eu(4))) void acq_corr_mixed_kernel(const float2 *__restrict__ X, const float2 *__restrict__ Rc, const float2 *__restrict__ tw, int B,
                                                                int nSeg, int binsPerBlock, int pOffset, float *__restrict__ surf, unsigned int *__restrict__ mpBits)
{
    using Sh = AcqMixedShape<M, R2, R3, R4>;
    constexpr int T = Sh::T, L2 = Sh::L2, L3 = Sh::L3, NB4 = Sh::NB4;
    extern __shared__ float2 acqMxLds[];
    float2 *sA = acqMxLds, *sW2 = sA + M, *sW3 = sW2 + L2;   // sW2[n] = W_L2^n, sW3[n] = W_L3^n
    float *sMag = reinterpret_cast<float *>(sW3 + L3);
    const int t = threadIdx.x, p = blockIdx.y;
    for (int n = t; n < L2; n += T) sW2[n] = tw[(M / L2) * n];
    for (int n = t; n < L3; n += T) sW3[n] = tw[(M / L3) * n];
    // the PRN's spectrum and the pass-1 twiddles W_M^(t k) of this thread's ten elements stay in registers across the bins
    af2 rc[ALIAS ? 1 : 10], w1[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        if constexpr (!ALIAS) {
            const float2 r = Rc[(size_t)p * M + t + T * q];
            rc[q] = af2{r.x, r.y};
        }
        const float2 a = tw[t * q];   // t k <= (M / 10 - 1) * 9 < M
        w1[q] = af2{a.x, a.y};
    }
    float mx[10], macc[NB4 * R4];
#pragma unroll
    for (int q = 0; q < 10; ++q) mx[q] = 0.f;
    const int b0 = blockIdx.x * binsPerBlock;
    const int nb = (B - b0) < binsPerBlock ? (B - b0) : binsPerBlock;
    const int nTr = nb * nSeg;
    const float2 *x0 = X + ((ALIAS ? (size_t)p * B : (size_t)0) + (size_t)b0) * nSeg * M;
    float2 xn[10];   // the next transform's spectrum, fetched under this one
#pragma unroll
    for (int q = 0; q < 10; ++q) xn[q] = x0[t + T * q];
    __syncthreads();
    int seg = 0, b = b0;
    for (int e = 0; e < nTr; ++e) {
        {   // spectrum product (correlator.py:75) and pass 1: radix 10 over the stride-M/10 elements, twiddle W_M^(t k)
            af2 v[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) v[q] = ALIAS ? af2{xn[q].x, xn[q].y} : acq_cmul(af2{xn[q].x, xn[q].y}, rc[ALIAS ? 0 : q]);
            if (e + 1 < nTr) {
#pragma unroll
                for (int q = 0; q < 10; ++q) xn[q] = x0[(size_t)(e + 1) * M + t + T * q];
            }
            acq_idft10(v);
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const af2 y = k ? acq_cmul(v[k], w1[k]) : v[k];
                sA[t + T * k] = make_float2(y.x, y.y);
            }
        }
        __syncthreads();
        // pass 2: radix R2 inside each L2-point sub-sequence
        for (int bf = t; bf < M / R2; bf += T) {
            constexpr int per = L2 / R2;
            const int sub = bf / per, tt = bf - sub * per, base = sub * L2 + tt;
            af2 v[R2];
#pragma unroll
            for (int q = 0; q < R2; ++q) {
                const float2 a = sA[base + per * q];
                v[q] = af2{a.x, a.y};
            }
            acq_idft<R2>(v);
#pragma unroll
            for (int k = 0; k < R2; ++k) {
                const float2 w = sW2[tt * k];   // tt k < L2
                const af2 y = k ? acq_cmul(v[k], af2{w.x, w.y}) : v[k];
                sA[base + per * k] = make_float2(y.x, y.y);
            }
        }
        __syncthreads();
        // pass 3: radix R3 inside each L3-point sub-sequence
        for (int bf = t; bf < M / R3; bf += T) {
            constexpr int per = L3 / R3;
            static_assert(per == R4, "four passes");
            const int sub = bf / per, tt = bf - sub * per, base = sub * L3 + tt;
            af2 v[R3];
#pragma unroll
            for (int q = 0; q < R3; ++q) {
                const float2 a = sA[base + per * q];
                v[q] = af2{a.x, a.y};
            }
            acq_idft<R3>(v);
#pragma unroll
            for (int k = 0; k < R3; ++k) {
                const float2 w = sW3[tt * k];
                const af2 y = k ? acq_cmul(v[k], af2{w.x, w.y}) : v[k];
                sA[base + per * k] = make_float2(y.x, y.y);
            }
        }
        __syncthreads();
        const bool last = ALIAS || seg == nSeg - 1;   // block-uniform
        // pass 4: the last radix, no twiddle; position sub R4 + k with sub = k1 R2 R3 + k2 R3 + k3 holds output k1 + 10 (k2 + R2 (k3 + R3 k))
#pragma unroll
        for (int h = 0; h < NB4; ++h) {
            const int sub = t + T * h;
            af2 d[R4];
#pragma unroll
            for (int q = 0; q < R4; ++q) {
                const float2 a = sA[sub * R4 + q];
                d[q] = af2{a.x, a.y};
            }
            acq_idft<R4>(d);
            const int k1 = sub / (R2 * R3), r23 = sub - k1 * (R2 * R3), k2 = r23 / R3, k3 = r23 - k2 * R3;
            const int n0 = k1 + 10 * (k2 + R2 * k3);
#pragma unroll
            for (int k = 0; k < R4; ++k) {
                const float mg = __builtin_amdgcn_sqrtf(d[k].x * d[k].x + d[k].y * d[k].y);   // | . |  (correlator.py:80)
                macc[h * R4 + k] = (ALIAS || seg == 0) ? mg : macc[h * R4 + k] + mg;
                if (last) sMag[n0 + 10 * R2 * R3 * k] = macc[h * R4 + k];
            }
        }
        __syncthreads();   // (also orders this pass's reads of sA before the next transform's pass-1 writes)
        if constexpr (ALIAS) {
            float sv = 0.f;
#pragma unroll
            for (int q = 0; q < 10; ++q) sv += sMag[t + T * q];   // the ten lag aliases of delay 10 t + seg (correlator.py:80-82)
            surf[((size_t)(pOffset + p) * B + b) * M + 10 * t + seg] = sv;
            atomicMax(&mpBits[(size_t)(pOffset + p) * M + 10 * t + seg], __float_as_uint(sv));   // max over the bins (:87)
            if (++seg == nSeg) { seg = 0; ++b; }
        } else if (last) {
            float *o = surf + ((size_t)p * B + b) * M;
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const float sv = sMag[t + T * q];
                o[t + T * q] = sv;
                mx[q] = fmaxf(mx[q], sv);
            }
            seg = 0; ++b;
            // (sMag is rewritten only after the next transform's three barriers)
        } else ++seg;
    }
    if constexpr (!ALIAS) {
#pragma unroll
        for (int q = 0; q < 10; ++q) atomicMax(&mpBits[(size_t)p * M + t + T * q], __float_as_uint(mx[q]));   // max over the bins (:87)
    }
}

}  // namespace dpe
