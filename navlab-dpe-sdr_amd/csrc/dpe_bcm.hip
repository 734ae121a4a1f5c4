// dpe_bcm.hip -- BatchCorrManifold for MI355X (gfx950): score every point of the ENU-dt
// (and velocity-drift) manifold grid against the per-SV score banks, fused arg-max.
//
// Replaces BCM_PosMeasML / BCM_VelMeasML + thrust::max_element + BCM_MakePosMeas/MakeVelMeas
// (cudarecv/modules/src/batchcorrmanifold.cu:1710-1828,1861-1963,2589-2596).
//
// The reference evaluates, per (grid point, SV), an fp64 range with ~2e7 m magnitude and maps it
// to a fractional index into the SV's score row.  Here the geometry is expanded about the grid
// centre on the HOST in fp64 (per window, per SV: unit line of sight in ENU, index at the centre,
// index-per-metre scale, 1/(2 range)); the kernel then needs only fp32 DIFFERENCES:
//   pos:  d_rho = |d - R delta| - |d| = -a + (q - a^2) h + O(|delta|^3 / range^2),
//         a = u_enu . delta, q = |delta|^2, h = 1/(2 range)       (error < 1.3e-5 m for |delta| < 3 km)
//         idx  = idx0 + g (delta_t + d_rho)                         (g = fs F_CA / (fc C), :1783-1791)
//   vel:  idx  = idx0 + g_v (u_enu . delta_v - delta_tdot)          (exactly linear, :1917-1936)
// followed by the reference's floor / floor(+1) linear interpolation (:1798-1812) and |.|^L.
//
// HBM traffic per window per manifold: 16 B/point grid read (float4, coalesced) + 4 B/point score
// write; banks (K x (2L+1) float4 pairs) and SV coefficients live in LDS.
#include <algorithm>
#include <atomic>

#include "dpe_common.h"
#include "dpe_prep.h"

namespace dpe {

// Single-window calls pass both manifolds' coefficients in the kernel-argument segment of the scan
// kernels (params_ptr, dpe_common.h); pb must stay their FIRST argument.
struct BcmParamBlock {
    BcmSvDev s[2][DPE_MAX_CHAN];
};

typedef float f2 __attribute__((ext_vector_type(2)));

#ifndef DPE_PTS_PER_THREAD
#define DPE_PTS_PER_THREAD 4
#endif
#ifndef DPE_SV_UNROLL
#define DPE_SV_UNROLL 2
#endif
constexpr int kPtsPerThread = DPE_PTS_PER_THREAD;
constexpr int kPtsPerBlock = 256 * kPtsPerThread;

// One manifold's share of the fused launch
struct ScanSide {
    const float4 *grid;       // [G] ENU-dt offsets of this rank's shard
    const float2 *bank;       // [W][maxK][nEnt] score bank
    const BcmSvDev *sv;       // [W][maxK] coefficients (ignored when they ride in the kernel arguments)
    float *scores;            // [W][pitch] or nullptr (rows start on 128-byte lines, see dpe_bcm_scores_pitch)
    double *wsum;             // weighted-sum partials (WMEAN) or nullptr
    long long G, indexOffset, pitch;
    int nEnt, split;          // bank entries per SV; blocks along x that work on this manifold
};

// COMPACT: 12-byte LDS entries {A, B, C} instead of 16 (three dword reads per pair instead of b64 + b32): the layout for
// bank sets that do not fit the LDS otherwise (37 channels with +-130 .. +-172 entries) -- slower, always with the clamps.
template <int LP, bool SECOND, bool CLAMP, bool WMEAN, bool COMPACT = false>
__device__ __forceinline__ void scan_body(const ScanSide &sd, int inl, int K, int maxK, int lpower,
                                          unsigned long long *__restrict__ keys, unsigned long long *__restrict__ oob,
                                          int keyStride, int keySlot)
{
    const float4 *__restrict__ grid = sd.grid;
    const float2 *__restrict__ bank = sd.bank;
    float *__restrict__ scores = sd.scores;
    double *__restrict__ wsum = sd.wsum;
    const long long G = sd.G, indexOffset = sd.indexOffset;
    const int nEnt = sd.nEnt;
    const unsigned nBlkX = (unsigned)sd.split;   // persistent stride of this manifold's blocks
    extern __shared__ __align__(16) unsigned char smem[];
    // |lerp|^2 = A + w (B + w C) per bank entry, stored as {A, B, 0, C}: one address register serves the
    // ds_read_b64 {A,B} and the ds_read_b32 {C} (immediate offset 12; the gap keeps the compiler from fusing
    // them into a slower ds_read_b96).  A wave's 64 points land on a few
    // neighbouring entries (the index moves by << 1 entry per grid step), so the 16-byte stride is
    // conflict-free as long as a wave spans < 16 entries.
    float4 *sE = reinterpret_cast<float4 *>(smem);                           // [K][nEnt]
    float *sF = reinterpret_cast<float *>(smem);                             // COMPACT: [K][nEnt][3]
    __shared__ unsigned long long sKey[4];
    __shared__ unsigned int sOob[4];
    __shared__ double sW[4][5];

    const int w = blockIdx.y, tid = threadIdx.x;
    // first tile's grid points: issued before the bank fill so both latencies overlap
    float4 nxt[kPtsPerThread];
    {
        const long long b0 = (long long)blockIdx.x * kPtsPerBlock + tid;
#pragma unroll
        for (int it = 0; it < kPtsPerThread; ++it)
            nxt[it] = (b0 + it * 256 < G) ? grid[b0 + it * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float2 *bw = bank + (size_t)w * maxK * nEnt;
    for (int i = tid; i < K * nEnt; i += 256) {
        const int k = i / nEnt, j = i - k * nEnt;
        // |c0 + w (c1 - c0)|^2 = A + w (B + w C): the interpolated magnitude needs two FMAs per pair.
        // Entry nEnt-1 can never be a valid lower neighbour: it is the all-zero slot that out-of-window
        // indices are clamped to (contribution exactly 0).
        if (j + 1 < nEnt) {
            const float2 c0 = bw[(size_t)k * nEnt + j], c1 = bw[(size_t)k * nEnt + j + 1];
            const float dr = c1.x - c0.x, di = c1.y - c0.y;
            const float eA = c0.x * c0.x + c0.y * c0.y, eB = 2.f * (c0.x * dr + c0.y * di), eC = dr * dr + di * di;
            if (COMPACT) { sF[3 * i] = eA; sF[3 * i + 1] = eB; sF[3 * i + 2] = eC; }
#ifdef DPE_SCAN_SOA   // experiment (round 3, measured and dropped -- DESIGN.md 9): {A, B} as an 8-byte-stride array, C as a 4-byte-stride array behind it
            else { reinterpret_cast<float2 *>(smem)[i] = make_float2(eA, eB); sF[2 * K * nEnt + i] = eC; }
#else
            else sE[i] = make_float4(eA, eB, 0.f, eC);
#endif
        } else {
            if (COMPACT) { sF[3 * i] = 0.f; sF[3 * i + 1] = 0.f; sF[3 * i + 2] = 0.f; }
#ifdef DPE_SCAN_SOA
            else { reinterpret_cast<float2 *>(smem)[i] = make_float2(0.f, 0.f); sF[2 * K * nEnt + i] = 0.f; }
#else
            else sE[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        }
    }
    __syncthreads();

    // Persistent over point tiles: block (x, w) scores tiles x, x+gridDim.x, ... of window w, so the bank
    // fill above is paid once per block and the next tile's grid points are prefetched under the
    // current tile's arithmetic.  gridDim.x is a multiple of 8: blocks are dealt to the 8 XCDs round
    // robin, so each XCD's L2 keeps re-serving the same 1/8 of the grid for every window.
    // A thread's points: tile*1024 + tid + 256*it (each load instruction: 64 lanes x 16 B contiguous),
    // held as PAIRS so that the geometry runs on packed fp32 (v_pk_fma_f32: two points per instruction).
    constexpr int kPairs = kPtsPerThread / 2;
    const unsigned last = (unsigned)(nEnt - 1);
    // wave-uniform address -> scalar loads; inl: this manifold's coefficients come from pb.s[SECOND ? 0 : 1]
    const BcmSvDev *svw = params_ptr(sd.sv + (size_t)w * maxK, inl) + ((inl && !SECOND) ? DPE_MAX_CHAN : 0);
    const long long nTiles = (G + kPtsPerBlock - 1) / kPtsPerBlock;
    // per-lane running maximum as (score, index): a lane visits its points in increasing index order, so a strict
    // "greater" keeps the first maximum; the packed 64-bit key is built once, after the tile loop
    float bestSc = -1.f;          // scores are >= 0
    unsigned int bestIdx = 0u;
    unsigned int nOob = 0;
    // "Method 1" weighted-mean estimator (optional: wsum != nullptr): sum s, sum s*{x,y,z,t}, pair-packed fp32
    f2 w0 = f2{0.f, 0.f}, w1 = w0, w2 = w0, w3 = w0, w4 = w0;
    for (long long tile = blockIdx.x; tile < nTiles; tile += nBlkX) {
        const long long base = tile * kPtsPerBlock + tid;
        f2 dx[kPairs], dy[kPairs], dz[kPairs], dw[kPairs], q[kPairs], score[kPairs];
#pragma unroll
        for (int p = 0; p < kPairs; ++p) {
            const float4 d0 = nxt[2 * p], d1 = nxt[2 * p + 1];
            dx[p] = f2{d0.x, d1.x}; dy[p] = f2{d0.y, d1.y}; dz[p] = f2{d0.z, d1.z}; dw[p] = f2{d0.w, d1.w};
            q[p] = dx[p] * dx[p] + dy[p] * dy[p] + dz[p] * dz[p];
            score[p] = f2{0.f, 0.f};
        }
        {   // prefetch the next tile of this block (whole lane in range: plain loads, no per-point predicate)
            const long long b1 = base + (long long)nBlkX * kPtsPerBlock;
            if (b1 + (kPtsPerThread - 1) * 256 < G) {
#pragma unroll
                for (int it = 0; it < kPtsPerThread; ++it) nxt[it] = grid[b1 + it * 256];
            } else {
#pragma unroll
                for (int it = 0; it < kPtsPerThread; ++it)
                    nxt[it] = (b1 + it * 256 < G) ? grid[b1 + it * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        unsigned emax = 0;
#pragma unroll DPE_SV_UNROLL
        for (int k = 0; k < K; ++k) {
            const BcmSvDev s = svw[k];
            const float4 *bk = sE + k * nEnt;
#pragma unroll
            for (int p = 0; p < kPairs; ++p) {
                f2 idx;
                if (SECOND) {
                    f2 a = dx[p] * s.ue;
                    a = __builtin_elementwise_fma(dy[p], f2{s.un, s.un}, a);
                    a = __builtin_elementwise_fma(dz[p], f2{s.uu, s.uu}, a);
                    f2 x = dw[p] - a;
                    const f2 t = __builtin_elementwise_fma(-a, a, q[p]);          // q - a^2
                    x = __builtin_elementwise_fma(t, f2{s.h, s.h}, x);             // + (q - a^2) / (2 range)
                    idx = __builtin_elementwise_fma(x, f2{s.g, s.g}, f2{s.idx0, s.idx0});
                } else {
                    idx = __builtin_elementwise_fma(dw[p], f2{s.g, s.g}, f2{s.idx0, s.idx0});
                    idx = __builtin_elementwise_fma(dx[p], f2{-s.h, -s.h}, idx);
                    idx = __builtin_elementwise_fma(dy[p], f2{-s.pad0, -s.pad0}, idx);
                    idx = __builtin_elementwise_fma(dz[p], f2{-s.pad1, -s.pad1}, idx);
                }
                float c[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float id = idx[j];
                    const float wgt = __builtin_amdgcn_fractf(id);                 // id - floor(id)
                    int ei;
                    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ei) : "v"(id));          // (int)floor(id), saturating
                    unsigned e = (unsigned)ei;
                    if (CLAMP) {                                                   // negative -> huge -> zero slot
                        e = min(e, last);
                        emax = max(emax, e);
                    }
                    float m2;
                    if (COMPACT) {
                        const float *bf = sF + ((size_t)k * nEnt + e) * 3;
                        m2 = fmaf(wgt, fmaf(wgt, bf[2], bf[1]), bf[0]);
                    } else {
#if defined(DPE_SCAN_SOA)
                        const float2 ab = reinterpret_cast<const float2 *>(smem)[k * nEnt + e];
                        m2 = fmaf(wgt, fmaf(wgt, sF[2 * K * nEnt + k * nEnt + e], ab.y), ab.x);
#else
                        const float2 ab = *reinterpret_cast<const float2 *>(&bk[e]);
                        m2 = fmaf(wgt, fmaf(wgt, bk[e].w, ab.y), ab.x);
#endif
                    }
                    if (LP == 1) c[j] = __builtin_amdgcn_sqrtf(__builtin_fabsf(m2));    // raw v_sqrt_f32 (1 ulp)
                    else if (LP == 2) c[j] = m2;
                    else c[j] = powf(__builtin_amdgcn_sqrtf(__builtin_fabsf(m2)), (float)lpower);
                }
                score[p] += f2{c[0], c[1]};
            }
        }
        // out-of-window bookkeeping off the fast path: recount only if this thread ever hit the zero
        // slot (or owns padding beyond G)
        if (CLAMP && (emax == last || base + (kPtsPerThread - 1) * 256 >= G)) {
            for (int it = 0; it < kPtsPerThread; ++it) {
                if (base + it * 256 >= G) continue;
                const float px = dx[it >> 1][it & 1], py = dy[it >> 1][it & 1], pz = dz[it >> 1][it & 1];
                const float pw = dw[it >> 1][it & 1], pq = q[it >> 1][it & 1];
                for (int k = 0; k < K; ++k) {
                    const BcmSvDev s = svw[k];
                    float id;
                    if (SECOND) {
                        const float a = fmaf(pz, s.uu, fmaf(py, s.un, px * s.ue));
                        float x = pw - a;
                        x = fmaf(fmaf(-a, a, pq), s.h, x);
                        id = fmaf(x, s.g, s.idx0);
                    } else {   // the fast path's own expression, operation for operation
                        id = fmaf(pz, -s.pad1, fmaf(py, -s.pad0, fmaf(px, -s.h, fmaf(pw, s.g, s.idx0))));
                    }
                    int ei;
                    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ei) : "v"(id));
                    nOob += (min((unsigned)ei, last) == last) ? 1u : 0u;
                }
            }
        }
        if (WMEAN) {
#pragma unroll
            for (int p = 0; p < kPairs; ++p) {
                f2 sc = score[p];
                if (base + (kPtsPerThread - 1) * 256 >= G) {   // ragged last tile: padded points must not count
                    if (base + (2 * p) * 256 >= G) sc.x = 0.f;
                    if (base + (2 * p + 1) * 256 >= G) sc.y = 0.f;
                }
                w0 += sc;
                w1 = __builtin_elementwise_fma(sc, dx[p], w1);
                w2 = __builtin_elementwise_fma(sc, dy[p], w2);
                w3 = __builtin_elementwise_fma(sc, dz[p], w3);
                w4 = __builtin_elementwise_fma(sc, dw[p], w4);
            }
        }
        const unsigned int gi0 = (unsigned int)(base + indexOffset);   // global index of the lane's first point (< 2^32, checked at create)
        if (base + (kPtsPerThread - 1) * 256 < G) {
            // whole lane inside the grid (every tile but the last): no per-point bounds checks, one store address
            float *srow = scores ? scores + (size_t)w * sd.pitch + base : nullptr;
#pragma unroll
            for (int it = 0; it < kPtsPerThread; ++it) {
                const float sc = score[it >> 1][it & 1];
                // scores are written once and never read back on this path: non-temporal stores keep ~0.8 GB per 256-window
                // call from lingering as dirty L2 / Infinity-Cache lines whose write-back would run into the NEXT call's
                // first kernels (measured: the following DC-sum kernel 25 -> 13 us, step 0.863 -> 0.851 ms)
                if (scores) __builtin_nontemporal_store(sc, &srow[it * 256]);
                if (sc > bestSc) { bestSc = sc; bestIdx = gi0 + it * 256; }
            }
        } else {
#pragma unroll
            for (int it = 0; it < kPtsPerThread; ++it) {
                const long long i = base + it * 256;
                if (i < G) {
                    const float sc = score[it >> 1][it & 1];
                    if (scores) __builtin_nontemporal_store(sc, &scores[(size_t)w * sd.pitch + i]);
                    if (sc > bestSc) { bestSc = sc; bestIdx = gi0 + it * 256; }
                }
            }
        }
    }
    unsigned long long best = bestSc < 0.f ? 0ull
                                           : (((unsigned long long)__float_as_uint(bestSc) << 32) |
                                              (unsigned long long)(0xFFFFFFFFu - bestIdx));
    // block arg-max: larger score wins, ties -> smaller global index (thrust::max_element, :2589)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
        nOob += __shfl_xor(nOob, off, 64);
    }
    double ws[5] = {(double)w0.x + (double)w0.y, (double)w1.x + (double)w1.y, (double)w2.x + (double)w2.y,
                    (double)w3.x + (double)w3.y, (double)w4.x + (double)w4.y};
    if (WMEAN) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int j = 0; j < 5; ++j) ws[j] += __shfl_xor(ws[j], off, 64);
        }
    }
    if ((tid & 63) == 0) {
        sKey[tid >> 6] = best; sOob[tid >> 6] = nOob;
        if (WMEAN) {
#pragma unroll
            for (int j = 0; j < 5; ++j) sW[tid >> 6][j] = ws[j];
        }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = sKey[0];
        b = sKey[1] > b ? sKey[1] : b;
        b = sKey[2] > b ? sKey[2] : b;
        b = sKey[3] > b ? sKey[3] : b;
        // integer max: order-independent.  RETURNING atomics: their results are waited for before this
        // block takes its ticket (see the fused kernel), which orders them without a release fence -- on
        // gfx950 an agent-scope fence writes back the XCD's whole L2 (all the dirty score lines)
        unsigned long long seen = atomicMax(&keys[(size_t)w * keyStride + keySlot], b);
        // per-block partial of the weighted sums, reduced on the host in block order (deterministic)
        if (WMEAN) {
            double *o = wsum + ((size_t)w * nBlkX + blockIdx.x) * 5;   // wsum already points at this manifold's half
#pragma unroll
            for (int j = 0; j < 5; ++j) o[j] = ((sW[0][j] + sW[1][j]) + sW[2][j]) + sW[3][j];
        }
        const unsigned int n = sOob[0] + sOob[1] + sOob[2] + sOob[3];
        if (n) seen += atomicAdd(&oob[(size_t)w * keyStride + keySlot], (unsigned long long)n);
        asm volatile("" ::"v"(seen) : "memory");   // keep the returns (and the s_waitcnt they imply) alive, and the ticket below them
    }
}

// Both manifolds in ONE launch: blockIdx.z = 0 scores the position grid, 1 the velocity grid (a lone
// window then pays one launch instead of two and the two scans overlap).  Around the scans the kernel
//  * clears the key / counter set of the NEXT Update (the two sets alternate), so that no clearing
//    launch or memset node sits on the critical path, and
//  * lets the last block to finish publish keys and out-of-window counts of all windows straight
//    into the pinned host mirror (system-scope stores): dpe_bcm_results needs no D2H copy command.
// `done` cycles 0 .. total-1 through atomicInc and is back at 0 when the kernel ends.
template <int LP, bool CLAMP_P, bool CLAMP_V, bool WMEAN, bool COMPACT = false>
__global__ __launch_bounds__(256) void bcm_scan_kernel(BcmParamBlock pb, int inl, ScanSide sp, ScanSide sv, int K, int maxK,
                                                       int lpower, unsigned long long *__restrict__ keys,
                                                       unsigned long long *__restrict__ oob,
                                                       unsigned long long *__restrict__ clearPtr, int clearN,
                                                       unsigned int *__restrict__ done,
                                                       unsigned long long *__restrict__ hostKeys,
                                                       unsigned long long *__restrict__ hostOob,
                                                       unsigned long long seqValue)
{
    (void)pb;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int i = threadIdx.x; i < clearN; i += 256) clearPtr[i] = 0ull;
    if (blockIdx.z == 0) {
        if (blockIdx.x < (unsigned)sp.split) scan_body<LP, true, CLAMP_P, WMEAN, COMPACT>(sp, inl, K, maxK, lpower, keys, oob, 2, 0);
    } else {
        if (blockIdx.x < (unsigned)sv.split) scan_body<LP, false, CLAMP_V, WMEAN, COMPACT>(sv, inl, K, maxK, lpower, keys, oob, 2, 1);
    }
    // ---- last block out publishes the results
    // Thread 0 issued this block's key / counter atomics and has their return values, i.e. they are
    // performed at the device's coherence point; its ticket follows in program order.  The block that
    // draws the last ticket therefore reads final values with agent-scope loads.
    // (done == nullptr: nobody reads the host mirror -- the device-resident loop's next kernel takes the keys from device memory
    //  behind the kernel boundary -- so the ticket, the agent-scope reads and the stores over the host link are skipped)
    if (!done) return;
    __shared__ unsigned int sLast;
    if (threadIdx.x == 0) {
        const unsigned int total = gridDim.x * gridDim.y * gridDim.z;
        sLast = (atomicInc(done, total - 1) == total - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (sLast) {
        const int n = 2 * (int)gridDim.y;
        if (gridDim.y == 1) {
            // One window: keys, counts and a sequence word share one 64-byte line of the host mirror and are written by
            // ONE lane in program order (same line -> same memory channel -> they arrive in that order), the sequence
            // word last.  The host may poll it instead of waiting on the stream (dpe_bcm_results).
            if (threadIdx.x == 0) {
                unsigned long long v[4];
                v[0] = __hip_atomic_load(&keys[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[1] = __hip_atomic_load(&keys[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[2] = __hip_atomic_load(&oob[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[3] = __hip_atomic_load(&oob[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&hostKeys[0], v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&hostKeys[1], v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&hostOob[0], v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&hostOob[1], v[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&hostOob[2], seqValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // sequence word
            }
        } else {
            for (int i = threadIdx.x; i < n; i += 256) {
                const unsigned long long kv = __hip_atomic_load(&keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long ov = __hip_atomic_load(&oob[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&hostKeys[i], kv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&hostOob[i], ov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Device-resident inputs (dpe_bcm_update_dev): the per-SV coefficients of both manifolds from the reference's own port arrays
// on the device (cuChanMgr / cuEKF outputs, dpeflow.cpp:178-191,212; captured once by the reference at
// batchcorrmanifold.cu:2512-2540) -- what the host loop of dpe_bcm_update computes, in fp64.  The host form carries the centre
// index in long double because rxTime - pr / C (rxTime ~ 4e5 s) rounds at 5.8e-11 s in fp64; here the same difference is kept
// as an unevaluated sum (TwoSum, bcm_prep_one in dpe_prep.h).  One block, thread <-> channel; thread 0 also leaves the window's
// frame (xCurrkk1, ENU2ECEFMat, DopplerSign) in pinned host memory for dpe_bcm_results.  rxTimeDev != nullptr: the receive
// time is read from the device (a channel manager that lives there), else it is the host scalar of the reference (:2536-2537).
struct BcmPortsDev {
    const double *x, *R, *sat, *rcEnd, *fc, *fi;
    const int *cpRefTOW, *cpElaEnd, *cpRef, *dopplerSign;
    int dimT;
};
__global__ void bcm_prep_kernel(BcmPortsDev p, int K, double rxTime, const double *__restrict__ rxTimeDev, double fs, double Cf, int S, int L, int B,
                                long long C, BcmSvDev *__restrict__ svPos, BcmSvDev *__restrict__ svVel, BcmDevWin *__restrict__ hostWin)
{
    const int k = threadIdx.x;
    const double *c = p.x, *R = p.R;
    const int dsRaw = p.dopplerSign[0];
    const int ds = (dsRaw == 1 || dsRaw == -1) ? dsRaw : 1;   // (a bad value is flagged; the scan must stay finite)
    if (rxTimeDev) rxTime = *rxTimeDev;
    if (k == 0) {
        for (int i = 0; i < 8; ++i) hostWin->xCurrkk1[i] = c[i];
        for (int i = 0; i < 9; ++i) hostWin->enu2ecef[i] = R[i];
        hostWin->dopplerSign = ds;
        hostWin->bad = (dsRaw == 1 || dsRaw == -1) ? 0 : 1;
    }
    if (k >= K) return;
    const double *s = p.sat + ((size_t)k * p.dimT + p.dimT / 2) * 8;                      // mid-time entry, :1775
    BcmSvDev a, v;
    bcm_prep_one(c, R, s, p.rcEnd[k], p.fc[k], p.fi[k], p.cpRefTOW[k], p.cpElaEnd[k], p.cpRef[k], ds, rxTime, fs, Cf, S, L, B, C, a, v);
    svPos[k] = a;
    svVel[k] = v;
}

// PosScores in the reference's port type: one dense fp64 row (ConfigOutput(3, "PosScores", DOUBLE_t, GRID, ...), :2300)
__global__ void bcm_export_f64_kernel(const float *__restrict__ row, long long G, double *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (long long)gridDim.x * blockDim.x) out[i] = (double)row[i];
}

// referencePair mode, step 1: grid points whose FIRST-channel index lies within `tol` entries of the centre lag (bank entry L)
// -- the only ones the reference's floor(idx) / floor(idx + 1) pair can treat differently (see ref_pair_fixup).  Candidates
// are appended as (window << 40 | point); cand[0] counts them (it may run past the capacity: the host then grows the list).
__global__ __launch_bounds__(256) void bcm_refpair_candidates_kernel(const float4 *__restrict__ grid, long long G, const BcmSvDev *__restrict__ sv,
                                                                     int maxK, int L, double tol, unsigned long long *__restrict__ cand,
                                                                     unsigned long long capacity)
{
    const int w = blockIdx.y;
    const BcmSvDev p0 = sv[(size_t)w * maxK];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < G; i += (long long)gridDim.x * 256) {
        const float4 g = grid[i];
        const double a = (double)p0.ue * g.x + (double)p0.un * g.y + (double)p0.uu * g.z;
        const double q = (double)g.x * g.x + (double)g.y * g.y + (double)g.z * g.z;
        const double idx0 = (double)p0.idx0 + (double)p0.g * ((double)g.w - a + (q - a * a) * (double)p0.h);
        if (fabs(idx0 - (double)L) <= tol) {
            const unsigned long long slot = atomicAdd(&cand[0], 1ull);
            if (slot < capacity) cand[1 + slot] = ((unsigned long long)w << 40) | (unsigned long long)i;
        }
    }
}

// step 3: the re-evaluated scores go in with one launch; the scores they replace come back (weighted-mean correction)
__global__ void bcm_patch_kernel(float *__restrict__ scores, long long pitch, const unsigned long long *__restrict__ where,
                                 const float *__restrict__ value, float *__restrict__ old, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const unsigned long long wi = where[j];
    float *dst = scores + (size_t)(wi >> 40) * pitch + (size_t)(wi & ((1ull << 40) - 1));
    old[j] = *dst;
    *dst = value[j];
}

// step 4: arg-max of the position manifold re-derived from the patched score array.
__global__ void bcm_zero_pos_keys_kernel(unsigned long long *__restrict__ keys, int nWindows)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nWindows) keys[2 * (size_t)w] = 0ull;
}

__global__ __launch_bounds__(256) void bcm_rekey_kernel(const float *__restrict__ scores, long long G, long long pitch, long long indexOffset,
                                                        unsigned long long *__restrict__ keys)
{
    const int w = blockIdx.y;
    const float *row = scores + (size_t)w * pitch;
    float bestSc = -1.f;
    unsigned int bestIdx = 0u;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < G; i += (long long)gridDim.x * 256) {
        const float sc = row[i];
        if (sc > bestSc) { bestSc = sc; bestIdx = (unsigned int)(i + indexOffset); }   // increasing i per lane: first maximum kept
    }
    unsigned long long best = bestSc < 0.f ? 0ull
                                           : (((unsigned long long)__float_as_uint(bestSc) << 32) | (unsigned long long)(0xFFFFFFFFu - bestIdx));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(&keys[2 * (size_t)w], best);
}

// referencePair for the device-parameter form (dpe_bcm_update_dev), step 2 on the device: one thread per candidate re-evaluates its
// point with the reference's own expression in fp64 (batchcorrmanifold.cu:1760-1816, the host form's ref_pair_index / ref_pair_fixup
// operation for operation) from the port arrays, and where the neighbour pair is two apart writes the score in place.  A candidate
// list that overflowed its capacity cannot be grown without the host: bit 1 of the window frame's `bad` (the results call reports it).
#pragma clang fp contract(off)
__device__ static inline bool ref_pair_index_dev(const double *R, const double *c, const double *s, double rxTime, double rcEnd, double fc,
                                                 int cpRefTOW, int cpElaEnd, int cpRef, const double *g, double fs, int S, int k,
                                                 double *idxOut, double *fiOut, double *ciOut)
{
    const double px = R[0] * g[0] + R[1] * g[1] + R[2] * g[2] + c[0];        // :1760-1763
    const double py = R[3] * g[0] + R[4] * g[1] + R[5] * g[2] + c[1];
    const double pz = R[6] * g[0] + R[7] * g[1] + R[8] * g[2] + c[2];
    const double pdt = g[3] + c[3];
    const double lx = s[0] - px, ly = s[1] - py, lz = s[2] - pz;            // :1779-1781
    const double range = sqrt(lx * lx + ly * ly + lz * lz);                 // :1782
    const double pr = range - kC * s[3] + pdt;                              // :1783
    const double txT = rxTime - pr / kC;                                    // :1784
    const double cfd = txT - cpRefTOW - ((cpElaEnd - cpRef) * kTCA);
    const double rcbc = cfd * kFCA;                                         // :1786
    const double rc0 = rcbc - rcEnd;                                        // :1790
    const double base = (fs / fc) * (-rc0) + S / 2.0;                       // :1791
    if (!(base < S && base > 0)) return false;                              // :1795
    const double idx = base + ((double)S * k);                              // :1797
    *idxOut = idx;
    *fiOut = floor(idx);                                                    // :1798
    *ciOut = floor(idx + 1);                                                // :1799
    return true;
}
__global__ __launch_bounds__(64) void bcm_refpair_eval_kernel(BcmPortsDev p, int K, double rxTime, const double *__restrict__ rxTimeDev, double fs, int S, int L, int lPower,
                                                               const double *__restrict__ grid64, const float2 *__restrict__ bank, int maxK,
                                                               const unsigned long long *__restrict__ cand, unsigned long long capacity,
                                                               float *__restrict__ scores, long long pitch, BcmDevWin *__restrict__ hostWin,
                                                               unsigned long long *__restrict__ patched)
{
    const unsigned long long nAll = cand[0], n = nAll < capacity ? nAll : capacity;
    if (nAll > capacity && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&hostWin->bad, 2);
    if (rxTimeDev) rxTime = *rxTimeDev;   // (the prepared form: the receiver time is the channel manager's device port)
    const int nLag = 2 * L + 1;
    for (unsigned long long j = (unsigned long long)blockIdx.x * 64 + threadIdx.x; j < n; j += (unsigned long long)gridDim.x * 64) {
        const unsigned long long wi = cand[1 + j];
        const long long i = (long long)(wi & ((1ull << 40) - 1));   // (one window in this form)
        const double *g = grid64 + 4 * i;
        double idx, fi_, ci_;
        const double *s0 = p.sat + ((size_t)0 * p.dimT + p.dimT / 2) * 8;
        if (!ref_pair_index_dev(p.R, p.x, s0, rxTime, p.rcEnd[0], p.fc[0], p.cpRefTOW[0], p.cpElaEnd[0], p.cpRef[0], g, fs, S, 0, &idx, &fi_, &ci_)) continue;
        if (ci_ - fi_ != 2.0) continue;                                   // the ordinary pair: the scan's value stands
        double score = 0.0;
        for (int k = 0; k < K; ++k) {
            const double *s = p.sat + ((size_t)k * p.dimT + p.dimT / 2) * 8;
            if (!ref_pair_index_dev(p.R, p.x, s, rxTime, p.rcEnd[k], p.fc[k], p.cpRefTOW[k], p.cpElaEnd[k], p.cpRef[k], g, fs, S, k, &idx, &fi_, &ci_)) continue;
            const long long fin = (long long)fi_ - (long long)S * k - (S / 2 - L);
            const long long cin = (long long)ci_ - (long long)S * k - (S / 2 - L);
            if (fin < 0 || cin < 0 || fin >= nLag || cin >= nLag) continue;
            const float2 *row = bank + (size_t)k * nLag;
            const double wc = idx - fi_, wf = ci_ - idx;                 // :1810-1811
            const double vr = (double)row[cin].x * wc + (double)row[fin].x * wf;
            const double vi = (double)row[cin].y * wc + (double)row[fin].y * wf;
            score += pow(hypot(vr, vi), (double)lPower);                  // :1816
        }
        scores[(size_t)i] = (float)score;
        atomicAdd(patched, 1ull);
        (void)maxK; (void)pitch;
    }
}

}  // namespace dpe

// ============================================================================================
struct dpe_bcm {
    dpe_bcm_config cfg;
    std::vector<double> posGrid_h, velGrid_h;  // local shard, fp64 (for zVal)
    float4 *posGrid_d = nullptr, *velGrid_d = nullptr;
    dpe_owner_detach_fn ownerDetach = nullptr;   // an attached device-resident channel manager: told first when this handle is destroyed
    void *owner = nullptr;
    bool gridsBorrowed = false;   // the fp32 device grids belong to another handle of the same dpe_pipe (dpe_bcm_create_sharing)
    double *posGrid64_d = nullptr, *velGrid64_d = nullptr;   // fp64 copies for a measurement formed on the device (dpe_bcm_hook_get)
    float *posScores_d = nullptr, *velScores_d = nullptr;
    long long posPitch = 0, velPitch = 0;   // floats between the score rows of consecutive windows (grid size rounded up to 32)
    dpe::BcmSvDev *sv_d = nullptr;  // [2][W][maxK]  (manifold-major)
    // pinned staging ring (see dpe_bcs): Updates may be issued kStaging - 1 deep without waiting
    static constexpr int kStaging = 4;
    dpe::BcmSvDev *svBase_h = nullptr, *sv_h = nullptr, *svBase_hd = nullptr;   // _hd: the pinned block's device address
    hipEvent_t stagingFree[kStaging] = {};
    int slot = 0;
    unsigned long long *keys_d = nullptr;  // [2 sets][{keys [W][2], counts [W][2]}], alternating between Updates
    int cur = 1;                           // set of the latest Update
    unsigned int *done_d = nullptr;        // finished-block ticket of the scan kernel (returns to 0 by itself)
    unsigned long long *keys_hd = nullptr; // device view of the pinned mirror keys_h
    unsigned long long seq = 0;            // single-window Updates: sequence number the kernel writes behind the results
    bool pollable = false, pollAllowed = true;   // last Update was a single eager window; DPE_BCM_NO_POLL=1 disables
    double *wsum_d = nullptr;   // [W][2][split][5] per-block weighted sums
    unsigned long long *keys_h = nullptr, *oob_h = nullptr;   // pinned mirrors, filled by async copies at the end of Update
    unsigned lastSplit[2] = {0, 0};
    int splitForce = 0;                     // -DDPE_EXPERIMENTS builds only: DPE_BCM_SPLIT at create
    static constexpr unsigned kMaxSplit = 4096;
    size_t wsumHalf = 0;
    // referencePair mode (dpe_bcm_config): active only when S / 2 is a power of two
    bool compact = false;                   // 12-byte LDS bank entries (the banks of all channels would not fit otherwise)
    bool refPair = false;
    float2 *refBank_h = nullptr;            // pinned copy of the code banks of the last Update
    unsigned long long *refCand_d = nullptr;   // [1 + refCap] candidate list of the device prefilter (entry 0: count)
    unsigned long long refCap = 0;
    unsigned long long *refWhere_d = nullptr;  // [refPatchCap] patches: (window << 40 | point), value, replaced value
    float *refValue_d = nullptr, *refOld_d = nullptr;
    size_t refPatchCap = 0;
    std::vector<double> refWsum;            // [W][5] corrections of the weighted sums (patched - scanned score at offset x,y,z,t)
    long long refPatched = 0;               // points patched by the last Update (diagnostic)
    dpe::BcmPortsDev refPorts{};            // the device ports of the dpe_bcm_update_dev in progress (referencePair on the device)
    double refRxTime = 0.0;
    const double *refRxTime_d = nullptr;    // the prepared form: rxTime as a device port of the attached channel manager (dpe_bcm_hook_set_ref_ports)
    bool refPortsAttached = false;
    unsigned long long *refPatched_d = nullptr;
    std::vector<dpe_bcm_window> win_h;
    dpe::BcmDevWin *devWin_h = nullptr, *devWin_hd = nullptr;   // pinned [2]: window frame of a device-parameter Update (written by bcm_prep_kernel),
                                                                // one per alternating key set, so that the results of Update n can still be
                                                                // decoded after Update n + 1 has been enqueued
    bool lastDev = false;
    int lastW = 0;
    bool publish = true;        // false: the scan leaves keys / counts in device memory only (set by the device-resident channel manager's attach)
    bool lastPublished = true;  // what the last Update did
    double posExtent = 0, velExtent = 0;
    dpe::KernelProfiler prof;  // slot 0: the fused position + velocity scan
    dpe::GraphCache graphs;
};

static int upload_grid(const double *src, int64_t G, std::vector<double> &keep, float4 **dst)
{
    using namespace dpe;
    keep.assign(src, src + 4 * G);
    std::vector<float4> f((size_t)G);
    for (int64_t i = 0; i < G; ++i)
        f[i] = make_float4((float)src[4 * i], (float)src[4 * i + 1], (float)src[4 * i + 2], (float)src[4 * i + 3]);
    *dst = dev_alloc<float4>((size_t)G);
    DPE_REQUIRE(*dst, "[BatchCorrManifold] create: grid allocation failed (%lld points)", (long long)G);
    DPE_CHECK_HIP(hipMemcpy(*dst, f.data(), sizeof(float4) * (size_t)G, hipMemcpyHostToDevice));
    return 0;
}

// Blocks along x per window: ~4096 blocks in flight overall for batches, a multiple of 8 (XCD round robin, see the
// kernel) and never more than the number of 1024-point tiles.
static unsigned scan_split(long long G, int nWindows, int forced)
{
    const long long nTiles = (G + dpe::kPtsPerBlock - 1) / dpe::kPtsPerBlock;
    long long s = 4096 / nWindows;
    // one or two windows (closed loop): latency, not throughput -- 128 blocks per manifold (one block per CU over both
    // manifolds), each walking several tiles, beat one block per tile: the per-block costs (bank staging, key atomics,
    // ticket) are paid fewer times.  Measured on the 25^4 grids: 53.5 -> 49.3 us per window.
    if (nWindows <= 2 && s > 128) s = 128;
    // batches: ~768 blocks per manifold are enough to fill the chip, and each extra block stages the banks again (K x nEnt
    // entries).  Measured: config H (12 SVs x 63 entries, 32 windows) scan 0.090 ms at 96 blocks per window, 0.040 at 24;
    // config R (256 windows) 0.538 at 16, 0.533 at 24.
    if (nWindows > 2) {
        const long long cap = 768 / nWindows > 24 ? 768 / nWindows : 24;
        if (s > cap) s = cap;
    }
    if (forced > 0) s = forced;   // (-DDPE_EXPERIMENTS builds: DPE_BCM_SPLIT, read at create)
    if (s < 8) s = 8;
    s = (s + 7) / 8 * 8;
    if (s > nTiles) s = nTiles;
    return (unsigned)s;
}

struct ScanLaunch {
    dpe::BcmParamBlock pb;
    int inl, K, maxK, lp;
    dpe::ScanSide sp, sv;
    unsigned long long *keys, *oob, *clr;
    int clrN;
    unsigned int *done;
    unsigned long long *hostKeys, *hostOob, seq;
    dim3 grid;
    size_t lds;
    hipStream_t st;
};

template <int LP, bool WM>
static void launch_scan_compact(const ScanLaunch &a)
{
    hipLaunchKernelGGL((dpe::bcm_scan_kernel<LP, true, true, WM, true>), a.grid, dim3(256), a.lds, a.st, a.pb, a.inl, a.sp, a.sv, a.K, a.maxK,
                       a.lp, a.keys, a.oob, a.clr, a.clrN, a.done, a.hostKeys, a.hostOob, a.seq);
}

template <int LP, bool CP, bool CV, bool WM>
static void launch_scan4(const ScanLaunch &a)
{
    hipLaunchKernelGGL((dpe::bcm_scan_kernel<LP, CP, CV, WM>), a.grid, dim3(256), a.lds, a.st, a.pb, a.inl, a.sp, a.sv, a.K, a.maxK,
                       a.lp, a.keys, a.oob, a.clr, a.clrN, a.done, a.hostKeys, a.hostOob, a.seq);
}

template <bool CP, bool CV, bool WM>
static void launch_scan3(const ScanLaunch &a)
{
    if (a.lp == 1) launch_scan4<1, CP, CV, WM>(a);
    else if (a.lp == 2) launch_scan4<2, CP, CV, WM>(a);
    else launch_scan4<0, CP, CV, WM>(a);
}

// clampP / clampV = false only when the host has proved that every index of every (point, SV) pair of
// that manifold stays inside the bank (then the kernel drops the range clamp and the out-of-window
// bookkeeping); wmean selects the variant that also accumulates the weighted-mean sums
static void launch_scan(bool clampP, bool clampV, bool wmean, bool compact, const ScanLaunch &a)
{
    if (compact) {
#define DPE_SCAN_C(LPV) do { if (wmean) launch_scan_compact<LPV, true>(a); else launch_scan_compact<LPV, false>(a); } while (0)
        if (a.lp == 1) DPE_SCAN_C(1); else if (a.lp == 2) DPE_SCAN_C(2); else DPE_SCAN_C(0);
#undef DPE_SCAN_C
        return;
    }
#define DPE_SCAN_PICK(CP, CV) do { if (wmean) launch_scan3<CP, CV, true>(a); else launch_scan3<CP, CV, false>(a); } while (0)
    if (clampP) { if (clampV) DPE_SCAN_PICK(true, true); else DPE_SCAN_PICK(true, false); }
    else { if (clampV) DPE_SCAN_PICK(false, true); else DPE_SCAN_PICK(false, false); }
#undef DPE_SCAN_PICK
}

template <int LP, bool CP, bool CV>
static void allow_big_lds()
{
    (void)hipFuncSetAttribute((const void *)dpe::bcm_scan_kernel<LP, CP, CV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024);
    (void)hipFuncSetAttribute((const void *)dpe::bcm_scan_kernel<LP, CP, CV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024);
    if (CP && CV) {
        (void)hipFuncSetAttribute((const void *)dpe::bcm_scan_kernel<LP, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024);
        (void)hipFuncSetAttribute((const void *)dpe::bcm_scan_kernel<LP, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024);
    }
}

// ---- referencePair mode ---------------------------------------------------------------------
// The reference forms the neighbour pair as floor(idx) and floor(idx + 1) on idx = base + S k (batchcorrmanifold.cu:1797-1799).
// For k = 0 and S / 2 = 2^m an index one fp64 step below 2^m has idx + 1 rounding UP to 2^m + 1: the neighbours are two apart and
// both weights ~1, so the pair contributes c[n+1] + c[n-1] instead of ~c[n].  Because the reference's rxTime - pr / C rounds at
// 1.4e-4 samples, every grid point within that distance of the centre lag collapses onto the same fp64 value.  This pass finds
// those points (cheap fp64 prefilter on the first channel, then the reference's own expression), evaluates their scores exactly
// as :1760-1816 do, patches the score array and re-derives the arg-max from it.
#pragma clang fp contract(off)
static bool ref_pair_index(const dpe_bcm_window &win, const dpe_chan_end &ch, const double *g, double fs, int S, int k,
                           double *idxOut, double *fiOut, double *ciOut)
{
    using namespace dpe;
    const double *R = win.enu2ecef, *c = win.xCurrkk1, *s = ch.satState;
    const double px = R[0] * g[0] + R[1] * g[1] + R[2] * g[2] + c[0];        // :1760-1763
    const double py = R[3] * g[0] + R[4] * g[1] + R[5] * g[2] + c[1];
    const double pz = R[6] * g[0] + R[7] * g[1] + R[8] * g[2] + c[2];
    const double pdt = g[3] + c[3];
    const double lx = s[0] - px, ly = s[1] - py, lz = s[2] - pz;            // :1779-1781
    const double range = std::sqrt(lx * lx + ly * ly + lz * lz);            // :1782
    const double pr = range - kC * s[3] + pdt;                              // :1783
    const double txT = win.rxTime - pr / kC;                                // :1784
    const double cfd = txT - ch.cpRefTOW - ((ch.cpElapsedEnd - ch.cpRef) * kTCA);
    const double rcbc = cfd * kFCA;                                         // :1786
    const double rc0 = rcbc - ch.codePhaseEnd;                              // :1790
    const double base = (fs / ch.codeFrequency) * (-rc0) + S / 2.0;         // :1791
    if (!(base < S && base > 0)) return false;                              // :1795
    const double idx = base + ((double)S * k);                              // :1797
    *idxOut = idx;
    *fiOut = std::floor(idx);                                               // :1798
    *ciOut = std::floor(idx + 1);                                           // :1799
    return true;
}

static int ref_pair_fixup(dpe_bcm *h, const float *codeBank_dev, int nWindows, int nChan, const dpe_chan_end *chan_host,
                          unsigned long long *keys_d, hipStream_t stream)
{
    using namespace dpe;
    const int S = h->cfg.samplesPerWindow, L = h->cfg.lagHalfWidth, nLag = 2 * L + 1;
    const int maxK = h->cfg.maxChannels, W = h->cfg.maxWindows;
    const long long G = h->cfg.posGridSize;
    const double fs = h->cfg.samplingFrequency;
    h->refPatched = 0;
    h->refWsum.assign((size_t)nWindows * 5, 0.0);
    // ---- 1. candidates, on the device: one pass over the grid per window instead of W x G fp64 iterations on the host.
    // The coefficients of this Update are in sv_d for batches; a single window passed them as kernel arguments: upload.
    if (nWindows == 1) DPE_CHECK_HIP(hipMemcpyAsync(h->sv_d, h->sv_h, sizeof(BcmSvDev) * maxK, hipMemcpyHostToDevice, stream));
    std::vector<unsigned long long> cand;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!h->refCand_d) {
            if (h->refCap == 0) h->refCap = 1 << 16;
            h->refCand_d = dev_alloc<unsigned long long>(1 + h->refCap);
            DPE_REQUIRE(h->refCand_d, "[BatchCorrManifold] Update: referencePair: candidate list allocation failed (%llu entries)", h->refCap);
        }
        DPE_CHECK_HIP(hipMemsetAsync(h->refCand_d, 0, sizeof(unsigned long long), stream));
        const unsigned gx = (unsigned)((G + 256 * 8 - 1) / (256 * 8));
        hipLaunchKernelGGL(bcm_refpair_candidates_kernel, dim3(gx > 1024 ? 1024 : gx, nWindows), dim3(256), 0, stream, h->posGrid_d, G,
                           h->sv_d, maxK, L, 5e-4, h->refCand_d, h->refCap);
        unsigned long long n = 0;
        DPE_CHECK_HIP(hipMemcpyAsync(&n, h->refCand_d, sizeof(n), hipMemcpyDeviceToHost, stream));
        DPE_CHECK_HIP(hipStreamSynchronize(stream));
        if (n > h->refCap) {   // more candidates than the list holds: grow it and run the pass again (nothing was modified yet)
            DPE_REQUIRE(attempt == 0, "[BatchCorrManifold] Update: referencePair: candidate list overflow after growing");
            (void)hipFree(h->refCand_d);
            h->refCand_d = nullptr;
            h->refCap = n + n / 4;
            continue;
        }
        cand.resize((size_t)n);
        if (n) DPE_CHECK_HIP(hipMemcpy(cand.data(), h->refCand_d + 1, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
        break;
    }
    if (cand.empty()) return 0;
    std::sort(cand.begin(), cand.end());   // (the append order is not deterministic; the patches are)
    // ---- 2. the reference's own expression (:1760-1816) for the candidates, fp64 on the host
    DPE_CHECK_HIP(hipMemcpyAsync(h->refBank_h, codeBank_dev, sizeof(float2) * (size_t)nWindows * maxK * nLag, hipMemcpyDeviceToHost, stream));
    DPE_CHECK_HIP(hipStreamSynchronize(stream));
    std::vector<unsigned long long> where;
    std::vector<float> value;
    for (unsigned long long wi : cand) {
        const int w = (int)(wi >> 40);
        const long long i = (long long)(wi & ((1ull << 40) - 1));
        const dpe_bcm_window &win = h->win_h[w];
        const double *g = h->posGrid_h.data() + 4 * i;
        double idx, fi_, ci_;
        if (!ref_pair_index(win, chan_host[(size_t)w * nChan], g, fs, S, 0, &idx, &fi_, &ci_)) continue;
        if (ci_ - fi_ != 2.0) continue;                                   // the ordinary pair: the scan's value stands
        double score = 0.0;
        for (int k = 0; k < nChan; ++k) {
            if (!ref_pair_index(win, chan_host[(size_t)w * nChan + k], g, fs, S, k, &idx, &fi_, &ci_)) continue;
            const long long fin = (long long)fi_ - (long long)S * k - (S / 2 - L);
            const long long cin = (long long)ci_ - (long long)S * k - (S / 2 - L);
            if (fin < 0 || cin < 0 || fin >= nLag || cin >= nLag) continue;
            const float2 *row = h->refBank_h + ((size_t)w * maxK + k) * nLag;
            const double wc = idx - fi_, wf = ci_ - idx;                 // :1810-1811
            const double vr = (double)row[cin].x * wc + (double)row[fin].x * wf;
            const double vi = (double)row[cin].y * wc + (double)row[fin].y * wf;
            score += std::pow(std::hypot(vr, vi), (double)h->cfg.lPower); // :1816
        }
        where.push_back(wi);
        value.push_back((float)score);
    }
    h->refPatched = (long long)where.size();
    if (where.empty()) return 0;
    // ---- 3. one upload, one scatter launch (the buffers grow with the patch count: no fixed limit)
    const size_t n = where.size();
    if (n > h->refPatchCap) {
        (void)hipFree(h->refWhere_d); (void)hipFree(h->refValue_d); (void)hipFree(h->refOld_d);
        h->refPatchCap = n + n / 4 + 256;
        h->refWhere_d = dev_alloc<unsigned long long>(h->refPatchCap);
        h->refValue_d = dev_alloc<float>(h->refPatchCap);
        h->refOld_d = dev_alloc<float>(h->refPatchCap);
        if (!h->refWhere_d || !h->refValue_d || !h->refOld_d) {
            h->refPatchCap = 0;
            set_error("[BatchCorrManifold] Update: referencePair: patch buffers (%zu entries) allocation failed; the published scores are the scan's, unpatched", n);
            return -1;
        }
    }
    DPE_CHECK_HIP(hipMemcpyAsync(h->refWhere_d, where.data(), sizeof(unsigned long long) * n, hipMemcpyHostToDevice, stream));
    DPE_CHECK_HIP(hipMemcpyAsync(h->refValue_d, value.data(), sizeof(float) * n, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(bcm_patch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, h->posScores_d, h->posPitch, h->refWhere_d,
                       h->refValue_d, h->refOld_d, (int)n);
    if (h->cfg.weightedMean) {
        std::vector<float> old(n);
        DPE_CHECK_HIP(hipMemcpyAsync(old.data(), h->refOld_d, sizeof(float) * n, hipMemcpyDeviceToHost, stream));
        DPE_CHECK_HIP(hipStreamSynchronize(stream));
        for (size_t j = 0; j < n; ++j) {
            const int w = (int)(where[j] >> 40);
            const double *g = h->posGrid_h.data() + 4 * (long long)(where[j] & ((1ull << 40) - 1));
            const double d = (double)value[j] - (double)old[j];
            double *c = h->refWsum.data() + (size_t)w * 5;
            c[0] += d; c[1] += d * g[0]; c[2] += d * g[1]; c[3] += d * g[2]; c[4] += d * g[3];
        }
    }
    // ---- 4. arg-max from the patched scores
    hipLaunchKernelGGL(bcm_zero_pos_keys_kernel, dim3((nWindows + 63) / 64), dim3(64), 0, stream, keys_d, nWindows);
    const unsigned gx = (unsigned)((G + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(bcm_rekey_kernel, dim3(gx > 512 ? 512 : gx, nWindows), dim3(256), 0, stream, h->posScores_d, G, h->posPitch,
                       (long long)h->cfg.posGridIndexOffset, keys_d);
    DPE_CHECK_HIP(hipMemcpyAsync(h->keys_h, keys_d, sizeof(unsigned long long) * 2 * (size_t)nWindows, hipMemcpyDeviceToHost, stream));
    DPE_CHECK_HIP(hipStreamSynchronize(stream));
    return 0;
}

extern "C" {

int dpe_bcm_create(const dpe_bcm_config *cfg, dpe_bcm **out)
{
    return dpe_bcm_create_sharing(cfg, nullptr, out);
}

// donor != nullptr: a handle for the same configuration and grids as `donor` (a further lane of a dpe_pipe) that uses donor's device
// copy of the fp32 grids; donor must outlive it.
int dpe_bcm_create_sharing(const dpe_bcm_config *cfg, dpe_bcm *donor, dpe_bcm **out)
{
    using namespace dpe;
    DPE_REQUIRE(cfg && out, "[BatchCorrManifold] create: null argument");
    DPE_REQUIRE(!donor || ((int64_t)donor->posGrid_h.size() == 4 * cfg->posGridSize && (int64_t)donor->velGrid_h.size() == 4 * cfg->velGridSize &&
                           !donor->gridsBorrowed), "[BatchCorrManifold] create: the grids to share are not these grids");
    DPE_REQUIRE(cfg->samplesPerWindow > 0 && (cfg->samplesPerWindow % 2) == 0,
                "[BatchCorrManifold] create: samplesPerWindow must be even and positive");
    DPE_REQUIRE(cfg->samplingFrequency > 0 && cfg->numFFTPoints > 0, "[BatchCorrManifold] create: bad fs / numFFTPoints");
    DPE_REQUIRE(cfg->maxWindows >= 1 && cfg->maxChannels >= 1 && cfg->maxChannels <= DPE_MAX_CHAN,
                "[BatchCorrManifold] create: maxWindows/maxChannels out of range");
    DPE_REQUIRE(cfg->lagHalfWidth >= 1 && cfg->binHalfWidth >= 1 && cfg->lPower >= 1, "[BatchCorrManifold] create: bad L/B/LPower");
    DPE_REQUIRE(cfg->posGrid && cfg->velGrid && cfg->posGridSize > 0 && cfg->velGridSize > 0,
                "[BatchCorrManifold] create: grids missing");
    DPE_REQUIRE(cfg->posGridSize + cfg->posGridIndexOffset < 0xFFFFFFFFll &&
                cfg->velGridSize + cfg->velGridIndexOffset < 0xFFFFFFFFll,
                "[BatchCorrManifold] create: global grid index exceeds 32 bits");
    const size_t nEntMax = (size_t)(2 * (cfg->lagHalfWidth > cfg->binHalfWidth ? cfg->lagHalfWidth : cfg->binHalfWidth) + 1);
    const size_t ldsNeed = (size_t)cfg->maxChannels * (nEntMax * 16 + 32);
    const bool compact = ldsNeed > 150 * 1024;   // 12-byte entries (slower scan variant) when the 16-byte ones do not fit
    DPE_REQUIRE(!compact || (size_t)cfg->maxChannels * nEntMax * 12 <= 152 * 1024,
                "[BatchCorrManifold] create: score banks (%zu B even as 12-byte entries) exceed the 160 KB LDS",
                (size_t)cfg->maxChannels * nEntMax * 12);
    // validity of the range expansion (file header): |delta| must stay far below the SV range
    double maxR2 = 0, posExt = 0, velExt = 0;   // extents: max(|delta_xyz| + |delta_t|), bounds |delta_t - u.delta|
    for (int64_t i = 0; i < cfg->posGridSize; ++i) {
        const double *p = cfg->posGrid + 4 * i;
        const double r2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
        if (r2 > maxR2) maxR2 = r2;
        const double e = std::sqrt(r2) + std::fabs(p[3]);
        if (e > posExt) posExt = e;
    }
    for (int64_t i = 0; i < cfg->velGridSize; ++i) {
        const double *p = cfg->velGrid + 4 * i;
        const double e = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + std::fabs(p[3]);
        if (e > velExt) velExt = e;
    }
    DPE_REQUIRE(maxR2 < 3.0e3 * 3.0e3, "[BatchCorrManifold] create: position grid extends beyond 3 km from its centre");
    dpe_bcm *h = new dpe_bcm();
    h->cfg = *cfg;
    h->cfg.posGrid = h->cfg.velGrid = nullptr;
    h->compact = compact;
    {
        const int half = cfg->samplesPerWindow / 2;
        h->refPair = cfg->referencePair != 0 && (half & (half - 1)) == 0;   // only then can floor(idx + 1) - floor(idx) be 2
        if (cfg->referencePair && !cfg->writeScores) {
            set_error("[BatchCorrManifold] create: referencePair needs writeScores (the arg-max is re-derived from the patched scores)");
            delete h;
            return -1;
        }
    }
    h->posExtent = posExt * 1.000001 + maxR2 / 2.0e7 + 1e-3;   // + second-order term bound (range > 2e7 m) + fp32 slack
    h->velExtent = velExt * 1.000001 + 1e-6;
    const size_t W = cfg->maxWindows, K = cfg->maxChannels;
    if (donor) {      // a further lane of a dpe_pipe: the same grids, one device copy
        h->posGrid_h = donor->posGrid_h; h->velGrid_h = donor->velGrid_h;
        h->posGrid_d = donor->posGrid_d; h->velGrid_d = donor->velGrid_d;
        h->gridsBorrowed = true;
    } else if (upload_grid(cfg->posGrid, cfg->posGridSize, h->posGrid_h, &h->posGrid_d) ||
               upload_grid(cfg->velGrid, cfg->velGridSize, h->velGrid_h, &h->velGrid_d)) {
        dpe_bcm_destroy(h);
        return -1;
    }
    // score rows start on 128-byte lines: a wave's 64 consecutive scores are then two whole lines instead of one whole and two
    // partial ones (config R, 390 625-point rows: 0.779 -> 0.755 ms per step, the write-back of a step's 0.8 GB drains faster)
    h->posPitch = (cfg->posGridSize + 31) / 32 * 32;
    h->velPitch = (cfg->velGridSize + 31) / 32 * 32;
    if (cfg->writeScores) {
        h->posScores_d = dev_alloc<float>(W * (size_t)h->posPitch);
        h->velScores_d = dev_alloc<float>(W * (size_t)h->velPitch);
    }
    h->sv_d = dev_alloc<BcmSvDev>(2 * W * K);
    h->keys_d = dev_alloc<unsigned long long>(8 * W);   // two alternating sets of {keys [W][2], out-of-window counts [W][2]}
    h->wsumHalf = (size_t)(dpe_bcm::kMaxSplit + 8 * W) * 5;   // >= nWindows * blocks-per-window of any launch
    h->wsum_d = dev_alloc<double>(2 * h->wsumHalf);
    if ((cfg->writeScores && (!h->posScores_d || !h->velScores_d)) || !h->sv_d || !h->keys_d || !h->wsum_d ||
        hipHostMalloc((void **)&h->svBase_h, dpe_bcm::kStaging * 2 * W * K * sizeof(BcmSvDev), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&h->devWin_h, 2 * sizeof(BcmDevWin), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&h->keys_h, (4 * W + 8) * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
        set_error("[BatchCorrManifold] create: device allocation failed");
        dpe_bcm_destroy(h);
        return -1;
    }
    // dynamic LDS above 64 KB needs the opt-in attribute
    allow_big_lds<0, true, true>();  allow_big_lds<1, true, true>();  allow_big_lds<2, true, true>();
    allow_big_lds<0, false, true>(); allow_big_lds<1, false, true>(); allow_big_lds<2, false, true>();
    allow_big_lds<0, true, false>();  allow_big_lds<1, true, false>();  allow_big_lds<2, true, false>();
    allow_big_lds<0, false, false>(); allow_big_lds<1, false, false>(); allow_big_lds<2, false, false>();
    if (h->refPair &&
        hipHostMalloc((void **)&h->refBank_h, W * K * (size_t)(2 * cfg->lagHalfWidth + 1) * sizeof(float2), hipHostMallocDefault) != hipSuccess) {
        set_error("[BatchCorrManifold] create: host allocation failed");
        dpe_bcm_destroy(h);
        return -1;
    }
    h->oob_h = h->keys_h + 2 * W;
    for (size_t i = 0; i < 4 * W + 8; ++i) h->keys_h[i] = 0ull;
    h->pollAllowed = getenv("DPE_BCM_NO_POLL") == nullptr;
#ifdef DPE_EXPERIMENTS
    if (const char *e = getenv("DPE_BCM_SPLIT")) h->splitForce = atoi(e) > 0 && atoi(e) <= (int)dpe_bcm::kMaxSplit ? atoi(e) : 0;
#endif
    h->sv_h = h->svBase_h;
    const auto finish = [&]() -> int {   // a failure from here on must not leak the handle
        for (hipEvent_t &e : h->stagingFree) DPE_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        DPE_CHECK_HIP(hipMemset(h->keys_d, 0, 8 * W * sizeof(unsigned long long)));
        h->done_d = dev_alloc<unsigned int>(1);
        DPE_REQUIRE(h->done_d, "[BatchCorrManifold] create: device allocation failed");
        DPE_CHECK_HIP(hipMemset(h->done_d, 0, sizeof(unsigned int)));
        DPE_CHECK_HIP(hipHostGetDevicePointer((void **)&h->keys_hd, h->keys_h, 0));
        DPE_CHECK_HIP(hipHostGetDevicePointer((void **)&h->svBase_hd, h->svBase_h, 0));
        DPE_CHECK_HIP(hipHostGetDevicePointer((void **)&h->devWin_hd, h->devWin_h, 0));
        return 0;
    };
    if (finish()) {
        dpe_bcm_destroy(h);
        return -1;
    }
    h->win_h.resize(W);
    *out = h;
    return 0;
}

int dpe_bcm_destroy(dpe_bcm *h)
{
    if (!h) return 0;
    if (h->ownerDetach) h->ownerDetach(h->owner, 1);
    if (h->gridsBorrowed) h->posGrid_d = h->velGrid_d = nullptr;
    void *bufs[] = {h->posGrid64_d, h->velGrid64_d, h->posGrid_d, h->velGrid_d, h->posScores_d, h->velScores_d, h->sv_d, h->keys_d, h->wsum_d, h->done_d, h->refPatched_d};
    for (void *b : bufs) (void)hipFree(b);
    if (h->svBase_h) (void)hipHostFree(h->svBase_h);
    if (h->keys_h) (void)hipHostFree(h->keys_h);
    if (h->devWin_h) (void)hipHostFree(h->devWin_h);
    if (h->refBank_h) (void)hipHostFree(h->refBank_h);
    (void)hipFree(h->refCand_d); (void)hipFree(h->refWhere_d); (void)hipFree(h->refValue_d); (void)hipFree(h->refOld_d);
    for (hipEvent_t e : h->stagingFree)
        if (e) (void)hipEventDestroy(e);
    h->graphs.clear();
    delete h;
    return 0;
}

// win_host == nullptr: one window whose coefficients bcm_prep_kernel has already written to h->sv_d on `stream`
// (dpe_bcm_update_dev) -- nothing the host decides below may then depend on their values.
static int bcm_update_impl(dpe_bcm *h, const float *codeBank_dev, const float *carrBank_dev, int32_t nWindows, int32_t nChan,
                           const dpe_bcm_window *win_host, const dpe_chan_end *chan_host, dpe_stream_t stream_)
{
    using namespace dpe;
    const bool dev = win_host == nullptr;
    DPE_REQUIRE(h && codeBank_dev && carrBank_dev, "[BatchCorrManifold] Update: null argument");
    DPE_REQUIRE(nWindows >= 1 && nWindows <= h->cfg.maxWindows, "[BatchCorrManifold] Update: nWindows %d out of range", nWindows);
    DPE_REQUIRE(nChan >= 1 && nChan <= h->cfg.maxChannels, "[BatchCorrManifold] Update: nChan %d out of range", nChan);
    hipStream_t stream = (hipStream_t)stream_;
    const int S = h->cfg.samplesPerWindow, L = h->cfg.lagHalfWidth, B = h->cfg.binHalfWidth;
    const int maxK = h->cfg.maxChannels, W = h->cfg.maxWindows;
    const double fs = h->cfg.samplingFrequency, Cf = (double)h->cfg.numFFTPoints;
    bool posInside = !dev, velInside = !dev;   // every index provably inside the banks?  (device parameters: not known here)
    h->lastDev = dev;
    h->slot = (h->slot + 1) % dpe_bcm::kStaging;          // next staging block; an Update kStaging calls ago may still be copying it
    DPE_CHECK_HIP(hipEventSynchronize(h->stagingFree[h->slot]));
    h->sv_h = h->svBase_h + (size_t)h->slot * 2 * h->cfg.maxWindows * h->cfg.maxChannels;
    for (int w = 0; !dev && w < nWindows; ++w) {
        const dpe_bcm_window &win = win_host[w];
        DPE_REQUIRE(win.dopplerSign == 1 || win.dopplerSign == -1, "[BatchCorrManifold] Update: dopplerSign must be +/-1");
        h->win_h[w] = win;
        const double *c = win.xCurrkk1, *R = win.enu2ecef;
        for (int k = 0; k < nChan; ++k) {
            const dpe_chan_end &ch = chan_host[(size_t)w * nChan + k];
            const double *s = ch.satState;
            const double dx = s[0] - c[0], dy = s[1] - c[1], dz = s[2] - c[2];       // :1779-1781
            const double range = std::sqrt(dx * dx + dy * dy + dz * dz);               // :1782
            const double ux = dx / range, uy = dy / range, uz = dz / range;
            const double ue = R[0] * ux + R[3] * uy + R[6] * uz;                        // R^T u
            const double un = R[1] * ux + R[4] * uy + R[7] * uz;
            const double uu = R[2] * ux + R[5] * uy + R[8] * uz;
            // position manifold, centre index (:1783-1791).  Carried in long double: the reference's
            // fp64 "rxTime - pr/C" (rxTime ~4e5 s) rounds at 5.8e-11 s = 1.4e-4 samples PER POINT; the
            // expansion below is exact about the centre, so the centre itself is kept exact too.
            const long double pr = (long double)range - (long double)kC * s[3] + c[3];
            const long double txT = (long double)win.rxTime - pr / (long double)kC;
            const long double cfd = txT - ch.cpRefTOW - ((long double)(ch.cpElapsedEnd - ch.cpRef) * (long double)kTCA);
            const long double rc0 = cfd * (long double)kFCA - ch.codePhaseEnd;
            const double basePos = (double)(((long double)fs / ch.codeFrequency) * (-rc0) + S / 2.0L);
            BcmSvDev &p = h->sv_h[(size_t)(0 * W + w) * maxK + k];
            p.ue = (float)ue; p.un = (float)un; p.uu = (float)uu;
            p.g = (float)(fs * kFCA / (ch.codeFrequency * kC));
            p.h = (float)(0.5 / range);
            p.idx0 = (float)(basePos - (double)(S / 2 - L));
            p.pad0 = p.pad1 = 0.f;
            {
                const double reach = std::fabs((double)p.g) * h->posExtent + 1e-3;
                if (!((double)p.idx0 - reach >= 0.0 && (double)p.idx0 + reach < (double)(2 * L))) posInside = false;
            }
            // velocity manifold, centre index (:1917-1936)
            const double ex = c[4] - kOEDot * c[1], ey = c[5] + kOEDot * c[0], ez = c[6];
            const double lrr = ux * (ex - s[4]) + uy * (ey - s[5]) + uz * (ez - s[6]);
            const double fbc = kFL1 * ((lrr - c[7]) / kC + s[7]) / win.dopplerSign;
            const double baseVel = (Cf / fs) * (fbc - ch.carrierFrequency) + Cf / 2.0;
            const double gv = (Cf / fs) * kFL1 / (kC * win.dopplerSign);
            BcmSvDev &v = h->sv_h[(size_t)(1 * W + w) * maxK + k];
            v.ue = (float)ue; v.un = (float)un; v.uu = (float)uu;
            v.g = (float)(-gv);   // index = idx0 + g (delta_tdot - u . delta_v)
            v.idx0 = (float)(baseVel - (double)(h->cfg.numFFTPoints / 2 - B));
            v.h = (float)(-gv * ue); v.pad0 = (float)(-gv * un); v.pad1 = (float)(-gv * uu);   // g u, products formed in fp64
            {
                const double reach = std::fabs(gv) * h->velExtent + 1e-3;
                if (!((double)v.idx0 - reach >= 0.0 && (double)v.idx0 + reach < (double)(2 * B))) velInside = false;
            }
        }
    }
    h->lastW = nWindows;
    h->lastSplit[0] = scan_split(h->cfg.posGridSize, nWindows, h->splitForce);
    h->lastSplit[1] = scan_split(h->cfg.velGridSize, nWindows, h->splitForce);
    // Two key / counter sets alternate between Updates: this call reduces into set `cur` (zero since it
    // was cleared by the previous call's position scan, or by create) and clears the other one.
    // (h->cur only advances once the launch is enqueued: a call that fails earlier must not skip a clearing)
    const int use = h->cur ^ 1;
    unsigned long long *keys = h->keys_d + (size_t)use * 4 * W, *oob = keys + 2 * W;
    unsigned long long *other = h->keys_d + (size_t)(use ^ 1) * 4 * W;
    GraphCache::Guard graphGuard{h->graphs, stream};
    if (h->graphs.enabled && !h->prof.enabled && !h->refPair && !dev) {
        const int rc = h->graphs.begin({codeBank_dev, carrBank_dev, 0, nWindows, nChan,
                                        (posInside ? 1 : 0) | (velInside ? 2 : 0) | (use << 2) | (h->slot << 8), stream}, stream);
        DPE_REQUIRE(rc >= 0, "[BatchCorrManifold] Update: hipGraph capture/replay failed");
        if (rc == 1) {
            DPE_CHECK_HIP(hipEventRecord(h->stagingFree[h->slot], stream));   // the replayed graph reads this slot
            h->cur = use;
            h->pollable = false;
            return 0;
        }
    }
    // one window: coefficients as kernel arguments of the scans (a captured graph would freeze them, so
    // that path copies); batches: one copy covers both manifolds' coefficient blocks
    const bool inlineParams = nWindows == 1 && !h->graphs.capturing && !dev;
    BcmParamBlock pb{};
    if (dev) {}
    else if (inlineParams) {
        memcpy(pb.s[0], h->sv_h, sizeof(BcmSvDev) * nChan);
        memcpy(pb.s[1], h->sv_h + (size_t)W * maxK, sizeof(BcmSvDev) * nChan);
    } else {
        if (h->graphs.capturing) DPE_CHECK_HIP(hipMemcpyAsync(h->sv_d, h->sv_h, sizeof(BcmSvDev) * 2 * (size_t)W * maxK, hipMemcpyHostToDevice, stream));
        else upload_params(h->sv_d, h->svBase_hd + (h->sv_h - h->svBase_h), sizeof(BcmSvDev) * 2 * (size_t)W * maxK, stream);
        if (!h->graphs.capturing) DPE_CHECK_HIP(hipEventRecord(h->stagingFree[h->slot], stream));
    }
    const int nLag = 2 * L + 1, nBin = 2 * B + 1;
    ScanLaunch a;
    a.pb = pb; a.inl = inlineParams ? 1 : 0; a.K = nChan; a.maxK = maxK; a.lp = h->cfg.lPower;
    a.sp = ScanSide{h->posGrid_d, reinterpret_cast<const float2 *>(codeBank_dev), h->sv_d, h->posScores_d,
                    h->cfg.weightedMean ? h->wsum_d : nullptr, h->cfg.posGridSize, h->cfg.posGridIndexOffset, h->posPitch, nLag,
                    (int)h->lastSplit[0]};
    a.sv = ScanSide{h->velGrid_d, reinterpret_cast<const float2 *>(carrBank_dev), h->sv_d + (size_t)W * maxK, h->velScores_d,
                    h->cfg.weightedMean ? h->wsum_d + h->wsumHalf : nullptr, h->cfg.velGridSize, h->cfg.velGridIndexOffset, h->velPitch, nBin,
                    (int)h->lastSplit[1]};
    a.keys = keys; a.oob = oob; a.clr = other; a.clrN = 4 * W;
    const bool publishNow = h->publish || !dev || h->cfg.weightedMean;   // (only the device-parameter forms may skip it)
    a.done = publishNow ? h->done_d : nullptr; a.hostKeys = h->keys_hd; a.hostOob = h->keys_hd + 2 * W;
    h->lastPublished = publishNow;
    a.seq = ++h->seq;
    // (a replayed graph carries a stale sequence argument: polling only for eager single-window launches whose mirror
    //  has the sequence word right behind the results, i.e. maxWindows == 1)
    // (with the weighted-mean estimator the per-block sums are fetched with a copy after the results arrive: that copy must
    //  see the finished kernel, so the stream is waited for instead)
    h->pollable = nWindows == 1 && W == 1 && !h->graphs.capturing && h->pollAllowed && !h->cfg.weightedMean && publishNow;
    a.grid = dim3(h->lastSplit[0] > h->lastSplit[1] ? h->lastSplit[0] : h->lastSplit[1], nWindows, 2);
    a.lds = (size_t)nChan * (nLag > nBin ? nLag : nBin) * (h->compact ? 12 : 16);
    a.st = stream;
    // the kernel's last block writes keys and counts into the pinned host mirror: dpe_bcm_results only
    // has to synchronise
    h->prof.begin(0, stream);
    launch_scan(!posInside, !velInside, h->cfg.weightedMean != 0, h->compact, a);
    h->prof.end(0, stream);
    const bool captured = h->graphs.capturing;
    DPE_REQUIRE(h->graphs.end(stream) == 0, "[BatchCorrManifold] Update: hipGraph instantiate/launch failed");
    if (captured) DPE_CHECK_HIP(hipEventRecord(h->stagingFree[h->slot], stream));
    {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            h->pollable = false;   // nothing will ever write the sequence word of this Update
            dpe::set_error("%s:%d: launch failed -> %s", __FILE__, __LINE__, hipGetErrorString(le));
            return -1;
        }
    }
    h->cur = use;
    if (h->refPair && !dev) {
        h->pollable = false;
        if (ref_pair_fixup(h, codeBank_dev, nWindows, nChan, chan_host, keys, stream)) return -1;
    } else if (h->refPair) {
        // device-parameter form (dpe_bcm_update_dev): candidates, re-evaluation, patch and arg-max all on the device, nothing read back
        h->pollable = false;
        const long long G = h->cfg.posGridSize;
        if (!h->refCand_d) {
            if (h->refCap == 0) h->refCap = 1 << 16;
            h->refCand_d = dev_alloc<unsigned long long>(1 + h->refCap);
        }
        if (!h->refPatched_d) h->refPatched_d = dev_alloc<unsigned long long>(1);
        DPE_REQUIRE(h->refCand_d && h->refPatched_d, "[BatchCorrManifold] Update: referencePair: candidate list allocation failed");
        if (!h->posGrid64_d) { dpe_bcm_hook hk; if (dpe_bcm_hook_get(h, &hk)) return -1; }   // (makes the fp64 copies of the grids)
        DPE_CHECK_HIP(hipMemsetAsync(h->refCand_d, 0, sizeof(unsigned long long), stream));
        DPE_CHECK_HIP(hipMemsetAsync(h->refPatched_d, 0, sizeof(unsigned long long), stream));
        const unsigned gx = (unsigned)((G + 256 * 8 - 1) / (256 * 8));
        hipLaunchKernelGGL(bcm_refpair_candidates_kernel, dim3(gx > 1024 ? 1024 : gx, 1), dim3(256), 0, stream, h->posGrid_d, G, h->sv_d,
                           h->cfg.maxChannels, h->cfg.lagHalfWidth, 5e-4, h->refCand_d, h->refCap);
        hipLaunchKernelGGL(bcm_refpair_eval_kernel, dim3(64), dim3(64), 0, stream, h->refPorts, (int)nChan, h->refRxTime, h->refRxTime_d, h->cfg.samplingFrequency,
                           h->cfg.samplesPerWindow, h->cfg.lagHalfWidth, h->cfg.lPower, h->posGrid64_d, reinterpret_cast<const float2 *>(codeBank_dev),
                           h->cfg.maxChannels, h->refCand_d, h->refCap, h->posScores_d, h->posPitch, h->devWin_hd + use, h->refPatched_d);
        hipLaunchKernelGGL(bcm_zero_pos_keys_kernel, dim3(1), dim3(64), 0, stream, keys, 1);
        const unsigned gr = (unsigned)((G + 256 * 16 - 1) / (256 * 16));
        hipLaunchKernelGGL(bcm_rekey_kernel, dim3(gr > 512 ? 512 : gr, 1), dim3(256), 0, stream, h->posScores_d, G, h->posPitch,
                           (long long)h->cfg.posGridIndexOffset, keys);
        DPE_CHECK_HIP(hipGetLastError());
        h->lastPublished = false;   // (the pinned mirror holds the scan's keys from before the patch: the results call fetches the re-derived ones)
    }
    return 0;
}

int dpe_bcm_update(dpe_bcm *h, const float *codeBank_dev, const float *carrBank_dev, int32_t nWindows, int32_t nChan,
                   const dpe_bcm_window *win_host, const dpe_chan_end *chan_host, dpe_stream_t stream)
{
    DPE_REQUIRE(win_host && chan_host, "[BatchCorrManifold] Update: null argument");
    return bcm_update_impl(h, codeBank_dev, carrBank_dev, nWindows, nChan, win_host, chan_host, stream);
}

int dpe_bcm_update_dev(dpe_bcm *h, const float *codeBank_dev, const float *carrBank_dev, int32_t nChan, const dpe_bcm_ports_dev *ports,
                       double rxTime, dpe_stream_t stream)
{
    using namespace dpe;
    DPE_REQUIRE(h && ports, "[BatchCorrManifold] Update: null argument");
    DPE_REQUIRE(nChan >= 1 && nChan <= h->cfg.maxChannels, "[BatchCorrManifold] Update: nChan %d out of range", nChan);
    DPE_REQUIRE(!h->refPair || (h->cfg.writeScores && !h->cfg.weightedMean),
                "[BatchCorrManifold] Update: referencePair with the device ports patches the scores on the device: it needs writeScores and no weightedMean "
                "(the weighted sums are corrected on the host in the host form only)");
    DPE_REQUIRE(ports->xCurrkk1 && ports->enu2ecef && ports->satStates && ports->codePhaseEnd && ports->codeFrequency &&
                ports->carrierFrequency && ports->cpRefTOW && ports->cpElapsedEnd && ports->cpRef && ports->dopplerSign && ports->dimT >= 1,
                "[BatchCorrManifold] Update: a device port pointer is null / dimT < 1");
    const BcmPortsDev p = {ports->xCurrkk1, ports->enu2ecef, ports->satStates, ports->codePhaseEnd, ports->codeFrequency, ports->carrierFrequency,
                           ports->cpRefTOW, ports->cpElapsedEnd, ports->cpRef, ports->dopplerSign, ports->dimT};
    const size_t W = h->cfg.maxWindows, maxK = h->cfg.maxChannels;
    h->refPorts = p;
    h->refRxTime = rxTime;
    h->refRxTime_d = nullptr;
    hipLaunchKernelGGL(bcm_prep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, (int)nChan, rxTime, (const double *)nullptr, h->cfg.samplingFrequency,
                       (double)h->cfg.numFFTPoints, h->cfg.samplesPerWindow, h->cfg.lagHalfWidth, h->cfg.binHalfWidth, (long long)h->cfg.numFFTPoints,
                       h->sv_d, h->sv_d + W * maxK, h->devWin_hd + (h->cur ^ 1));   // (the frame of the key set this Update reduces into)
    return bcm_update_impl(h, codeBank_dev, carrBank_dev, 1, nChan, nullptr, nullptr, stream);
}

int dpe_bcm_update_prepared(dpe_bcm *h, const float *codeBank_dev, const float *carrBank_dev, int32_t nChan, dpe_stream_t stream)
{
    DPE_REQUIRE(h, "[BatchCorrManifold] Update: null argument");
    DPE_REQUIRE(nChan >= 1 && nChan <= h->cfg.maxChannels, "[BatchCorrManifold] Update: nChan %d out of range", nChan);
    // referencePair (batchcorrmanifold.cu:1798-1812): the prepared blocks hold expansion coefficients only, so the fp64 re-evaluation
    // reads the port arrays of the channel manager that wrote them (handed over at dpe_chm_dev_attach) -- the same device-side
    // candidates / re-evaluation / patch / arg-max kernels as dpe_bcm_update_dev, nothing read back
    DPE_REQUIRE(!h->refPair || h->refPortsAttached, "[BatchCorrManifold] Update: referencePair needs the port arrays: attach the device-resident channel "
                                                    "manager (dpe_chm_dev_attach), or use dpe_bcm_update_dev / the host form of the inputs");
    DPE_REQUIRE(!h->refPair || (h->cfg.writeScores && !h->cfg.weightedMean),
                "[BatchCorrManifold] Update: referencePair on the device patches the scores in place: it needs writeScores and no weightedMean");
    return bcm_update_impl(h, codeBank_dev, carrBank_dev, 1, nChan, nullptr, nullptr, stream);
}

int dpe_bcm_hook_set_ref_ports(dpe_bcm *h, const dpe_bcm_ports_dev *ports, const double *rxTime_dev)
{
    using namespace dpe;
    DPE_REQUIRE(h, "[BatchCorrManifold] hook: null handle");
    if (!ports) { h->refPortsAttached = false; h->refRxTime_d = nullptr; return 0; }
    h->refPorts = BcmPortsDev{ports->xCurrkk1, ports->enu2ecef, ports->satStates, ports->codePhaseEnd, ports->codeFrequency, ports->carrierFrequency,
                              ports->cpRefTOW, ports->cpElapsedEnd, ports->cpRef, ports->dopplerSign, ports->dimT};
    h->refRxTime_d = rxTime_dev;
    h->refPortsAttached = true;
    return 0;
}

int dpe_bcm_hook_get(dpe_bcm *h, dpe_bcm_hook *out)
{
    using namespace dpe;
    DPE_REQUIRE(h && out, "[BatchCorrManifold] hook: null argument");
    const size_t W = h->cfg.maxWindows, maxK = h->cfg.maxChannels;
    if (!h->posGrid64_d) {
        h->posGrid64_d = dev_alloc<double>(h->posGrid_h.size());
        h->velGrid64_d = dev_alloc<double>(h->velGrid_h.size());
        DPE_REQUIRE(h->posGrid64_d && h->velGrid64_d, "[BatchCorrManifold] hook: fp64 grid allocation failed");
        DPE_CHECK_HIP(hipMemcpy(h->posGrid64_d, h->posGrid_h.data(), sizeof(double) * h->posGrid_h.size(), hipMemcpyHostToDevice));
        DPE_CHECK_HIP(hipMemcpy(h->velGrid64_d, h->velGrid_h.data(), sizeof(double) * h->velGrid_h.size(), hipMemcpyHostToDevice));
    }
    out->svPos_d = h->sv_d;
    out->svVel_d = h->sv_d + W * maxK;
    out->devWin_hd = h->devWin_hd + (h->cur ^ 1);   // the frame of the NEXT Update's key set (re-queried per window by the channel manager)
    out->keys_d[0] = h->keys_d;
    out->keys_d[1] = h->keys_d + 4 * W;
    out->posGrid64_d = h->posGrid64_d;
    out->velGrid64_d = h->velGrid64_d;
    out->posG = h->cfg.posGridSize; out->velG = h->cfg.velGridSize;
    out->posOffset = h->cfg.posGridIndexOffset; out->velOffset = h->cfg.velGridIndexOffset;
    out->fs = h->cfg.samplingFrequency; out->Cf = (double)h->cfg.numFFTPoints;
    out->S = h->cfg.samplesPerWindow; out->L = h->cfg.lagHalfWidth; out->B = h->cfg.binHalfWidth;
    out->maxWindows = h->cfg.maxWindows; out->maxChannels = h->cfg.maxChannels;
    out->C = h->cfg.numFFTPoints;
    return 0;
}

int dpe_bcm_hook_set_owner(dpe_bcm *h, dpe_owner_detach_fn detach, void *owner)
{
    DPE_REQUIRE(h, "[BatchCorrManifold] hook: null handle");
    DPE_REQUIRE(!owner || !h->owner || h->owner == owner, "[BatchCorrManifold] hook: the handle is attached to another channel manager");
    h->ownerDetach = owner ? detach : nullptr;
    h->owner = owner;
    return 0;
}

int dpe_bcm_hook_set_publish(dpe_bcm *h, int enable)
{
    DPE_REQUIRE(h, "[BatchCorrManifold] hook: null argument");
    h->publish = enable != 0;
    return 0;
}

int dpe_bcm_set_graph(dpe_bcm *h, int32_t enable)
{
    DPE_REQUIRE(h, "[BatchCorrManifold] set_graph: null handle");
    h->graphs.enabled = enable != 0;
    if (!enable) h->graphs.clear();
    return 0;
}

static void make_meas(const dpe_bcm_window &win, const double *p, const double *v, double z[8])
{
    const double *R = win.enu2ecef, *c = win.xCurrkk1;
    z[0] = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + c[0];   // BCM_MakePosMeas :1990-1999
    z[1] = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + c[1];
    z[2] = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + c[2];
    z[3] = p[3] + c[3];
    z[4] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2] + c[4];   // BCM_MakeVelMeas :2042-2051
    z[5] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2] + c[5];
    z[6] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2] + c[6];
    z[7] = v[3] + c[7];
}

static void decode_key(unsigned long long key, float *score, int64_t *index)
{
    const unsigned int bits = (unsigned int)(key >> 32);
    memcpy(score, &bits, sizeof(float));
    *index = (int64_t)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
}

// Device-parameter Updates: the window's frame (xCurrkk1, ENU2ECEFMat, DopplerSign) came from device arrays; bcm_prep_kernel
// left a copy in pinned memory ahead of the scan, valid once the stream has been waited for (or the result polled).
static int fetch_device_frame(dpe_bcm *h)
{
    if (!h->lastDev) return 0;
    const dpe::BcmDevWin &fw = h->devWin_h[h->cur];
    DPE_REQUIRE(!(fw.bad & 1), "[BatchCorrManifold] results: DopplerSign on the device is not +/-1");
    DPE_REQUIRE(!(fw.bad & 2), "[BatchCorrManifold] results: referencePair: more candidate points than the device list holds (%llu): scores partly unpatched",
                (unsigned long long)h->refCap);
    memcpy(h->win_h[0].xCurrkk1, fw.xCurrkk1, sizeof(double) * 8);
    memcpy(h->win_h[0].enu2ecef, fw.enu2ecef, sizeof(double) * 9);
    h->win_h[0].dopplerSign = fw.dopplerSign;
    return 0;
}

int dpe_bcm_results(dpe_bcm *h, dpe_bcm_result *results, dpe_stream_t stream)
{
    DPE_REQUIRE(h && results && h->lastW > 0, "[BatchCorrManifold] results: no update yet");
    // Single-window Updates: the scan's last block writes a sequence word right behind the results in the pinned
    // mirror.  Polling it returns the fix as soon as it lands, without the stream-wait wake-up (a few us of a ~58 us
    // closed-loop window); anything unexpected falls back to the stream wait.
    bool arrived = false;
    if (h->pollable) {
        const unsigned long long *seqWord = h->oob_h + 2;
        for (int spin = 0; spin < 200000 && !arrived; ++spin)
            arrived = (__atomic_load_n(seqWord, __ATOMIC_ACQUIRE) == h->seq);   // acquire: the results are read after it
    }
    if (!arrived) DPE_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    const int W = h->lastW;
    if (!h->lastPublished) {   // the scan did not write the host mirror (device-resident loop): fetch keys and counts of the set it used
        const size_t Wm = h->cfg.maxWindows;
        DPE_CHECK_HIP(hipMemcpy(h->keys_h, h->keys_d + (size_t)h->cur * 4 * Wm, sizeof(unsigned long long) * 2 * Wm, hipMemcpyDeviceToHost));
        DPE_CHECK_HIP(hipMemcpy(h->oob_h, h->keys_d + (size_t)h->cur * 4 * Wm + 2 * Wm, sizeof(unsigned long long) * 2 * Wm, hipMemcpyDeviceToHost));
    }
    if (fetch_device_frame(h)) return -1;
    const unsigned long long *keys = h->keys_h, *oob = h->oob_h;
    std::vector<double> ws;   // per-block weighted sums: fetched only when the estimator is on (this call sits on the
                              // closed loop's critical path: no 300 KB of scratch per window otherwise)
    if (h->cfg.weightedMean) {
        ws.resize(2 * h->wsumHalf);
        DPE_CHECK_HIP(hipMemcpy(ws.data(), h->wsum_d, sizeof(double) * 2 * h->wsumHalf, hipMemcpyDeviceToHost));
    }
    for (int w = 0; w < W; ++w) {
        dpe_bcm_result &r = results[w];
        // "Method 1" score-weighted mean of the manifold (BCM_PosMeasReduction / BCM_ReduceAndPosMeas,
        // batchcorrmanifold.cu:816-1056,1365-1510; PyGNSS receiver.py:317-318): LOCAL shard only
        double m[2][5] = {};
        for (int slot = 0; slot < 2; ++slot) {
            if (h->cfg.weightedMean) {
                const double *base = ws.data() + slot * h->wsumHalf + (size_t)w * h->lastSplit[slot] * 5;
                for (unsigned b = 0; b < h->lastSplit[slot]; ++b)
                    for (int j = 0; j < 5; ++j) m[slot][j] += base[(size_t)b * 5 + j];
                if (slot == 0 && h->refPair && !h->refWsum.empty())
                    for (int j = 0; j < 5; ++j) m[0][j] += h->refWsum[(size_t)w * 5 + j];
            }
            for (int j = 0; j < 5; ++j) r.weightedSums[slot][j] = m[slot][j];
        }
        if (!h->cfg.weightedMean) {
            for (int j = 0; j < 8; ++j) r.zValMean[j] = 0.0;
        } else {
            const double pm[4] = {m[0][1] / m[0][0], m[0][2] / m[0][0], m[0][3] / m[0][0], m[0][4] / m[0][0]};
            const double vm[4] = {m[1][1] / m[1][0], m[1][2] / m[1][0], m[1][3] / m[1][0], m[1][4] / m[1][0]};
            make_meas(h->win_h[w], pm, vm, r.zValMean);
        }
        decode_key(keys[2 * w], &r.posScore, &r.posIndex);
        decode_key(keys[2 * w + 1], &r.velScore, &r.velIndex);
        r.posOutOfWindow = (int64_t)oob[2 * w];
        r.velOutOfWindow = (int64_t)oob[2 * w + 1];
        const int64_t pl = r.posIndex - h->cfg.posGridIndexOffset, vl = r.velIndex - h->cfg.velGridIndexOffset;
        DPE_REQUIRE(pl >= 0 && pl < h->cfg.posGridSize && vl >= 0 && vl < h->cfg.velGridSize,
                    "[BatchCorrManifold] results: arg-max index outside the local shard");
        make_meas(h->win_h[w], h->posGrid_h.data() + 4 * pl, h->velGrid_h.data() + 4 * vl, r.zVal);
    }
    return 0;
}

int dpe_bcm_profile(dpe_bcm *h, int32_t enable, float *ms, int32_t *count)
{
    DPE_REQUIRE(h, "[BatchCorrManifold] profile: null handle");
    float m[dpe::KernelProfiler::kSlots];
    int c[dpe::KernelProfiler::kSlots];
    h->prof.collect(m, c);
    for (int i = 0; i < 2; ++i) {
        if (ms) ms[i] = m[i];
        if (count) count[i] = c[i];
    }
    h->prof.enabled = enable != 0;
    return 0;
}

int dpe_bcm_scores(dpe_bcm *h, const float **posScores_dev, const float **velScores_dev)
{
    DPE_REQUIRE(h, "[BatchCorrManifold] scores: null handle");
    DPE_REQUIRE(h->cfg.writeScores, "[BatchCorrManifold] scores: created with writeScores=0");
    if (posScores_dev) *posScores_dev = h->posScores_d;
    if (velScores_dev) *velScores_dev = h->velScores_d;
    return 0;
}

int dpe_bcm_export_scores_f64(dpe_bcm *h, int32_t window, double *posScores_dev, double *velScores_dev, dpe_stream_t stream)
{
    DPE_REQUIRE(h && h->cfg.writeScores, "[BatchCorrManifold] export_scores_f64: created with writeScores=0");
    DPE_REQUIRE(window >= 0 && window < h->lastW, "[BatchCorrManifold] export_scores_f64: bad window %d", window);
    const long long Gp = h->cfg.posGridSize, Gv = h->cfg.velGridSize;
    if (posScores_dev)
        hipLaunchKernelGGL(dpe::bcm_export_f64_kernel, dim3((unsigned)((Gp + 1023) / 1024 > 1024 ? 1024 : (Gp + 1023) / 1024)), dim3(256), 0,
                           (hipStream_t)stream, h->posScores_d + (size_t)window * h->posPitch, Gp, posScores_dev);
    if (velScores_dev)
        hipLaunchKernelGGL(dpe::bcm_export_f64_kernel, dim3((unsigned)((Gv + 1023) / 1024 > 1024 ? 1024 : (Gv + 1023) / 1024)), dim3(256), 0,
                           (hipStream_t)stream, h->velScores_d + (size_t)window * h->velPitch, Gv, velScores_dev);
    DPE_CHECK_HIP(hipGetLastError());
    return 0;
}

int dpe_bcm_scores_pitch(dpe_bcm *h, int64_t *posPitch, int64_t *velPitch)
{
    DPE_REQUIRE(h, "[BatchCorrManifold] scores_pitch: null handle");
    if (posPitch) *posPitch = h->posPitch;
    if (velPitch) *velPitch = h->velPitch;
    return 0;
}

int dpe_bcm_keys(dpe_bcm *h, const uint64_t **keys_dev)
{
    DPE_REQUIRE(h && keys_dev, "[BatchCorrManifold] keys: null argument");
    *keys_dev = reinterpret_cast<const uint64_t *>(h->keys_d + (size_t)h->cur * 4 * h->cfg.maxWindows);
    return 0;
}

int dpe_bcm_exchange_keys(dpe_bcm *h, dpe_comm *c, uint64_t *keys_host, dpe_stream_t stream)
{
    DPE_REQUIRE(h && c && h->lastW > 0, "[BatchCorrManifold] exchange_keys: no update yet / null communicator");
    unsigned long long *keys = h->keys_d + (size_t)h->cur * 4 * h->cfg.maxWindows;
    if (dpe_comm_allreduce_max_u64(c, reinterpret_cast<uint64_t *>(keys), 2 * (int64_t)h->lastW, stream)) return -1;
    if (keys_host) {
        DPE_CHECK_HIP(hipMemcpyAsync(keys_host, keys, sizeof(uint64_t) * 2 * (size_t)h->lastW, hipMemcpyDeviceToHost, (hipStream_t)stream));
        DPE_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    }
    h->pollable = false;   // the host mirror still holds this rank's LOCAL keys
    return 0;
}

int dpe_bcm_results_from_keys(dpe_bcm *h, const uint64_t *keys_host, int32_t nWindows, const double *posGridGlobal,
                              int64_t posGridGlobalSize, const double *velGridGlobal, int64_t velGridGlobalSize,
                              dpe_bcm_result *results)
{
    DPE_REQUIRE(h && keys_host && posGridGlobal && velGridGlobal && results, "[BatchCorrManifold] results_from_keys: null argument");
    DPE_REQUIRE(nWindows >= 1 && nWindows <= h->lastW, "[BatchCorrManifold] results_from_keys: bad nWindows");
    if (fetch_device_frame(h)) return -1;   // (the reduced keys came through a synchronising copy: the scan has finished)
    for (int w = 0; w < nWindows; ++w) {
        dpe_bcm_result &r = results[w];
        DPE_REQUIRE(keys_host[2 * w] != 0 && keys_host[2 * w + 1] != 0,
                    "[BatchCorrManifold] results_from_keys: window %d has no valid score (key 0)", w);
        decode_key(keys_host[2 * w], &r.posScore, &r.posIndex);
        decode_key(keys_host[2 * w + 1], &r.velScore, &r.velIndex);
        DPE_REQUIRE(r.posIndex >= 0 && r.posIndex < posGridGlobalSize && r.velIndex >= 0 && r.velIndex < velGridGlobalSize,
                    "[BatchCorrManifold] results_from_keys: window %d: arg-max index (%lld, %lld) outside the global grids (%lld, %lld)",
                    w, (long long)r.posIndex, (long long)r.velIndex, (long long)posGridGlobalSize, (long long)velGridGlobalSize);
        r.posOutOfWindow = r.velOutOfWindow = -1;
        for (int j = 0; j < 8; ++j) r.zValMean[j] = 0.0;   // needs the all-reduced weightedSums; see sharding.py
        for (int j = 0; j < 10; ++j) (&r.weightedSums[0][0])[j] = 0.0;
        make_meas(h->win_h[w], posGridGlobal + 4 * r.posIndex, velGridGlobal + 4 * r.velIndex, r.zVal);
    }
    return 0;
}

}  // extern "C"
