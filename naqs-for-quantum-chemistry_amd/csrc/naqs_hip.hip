// naqs_hip.hip — gfx950 (MI355X / CDNA4) kernels + the C ABI declared in include/naqs_hip.h.
//
// The path: for a batch of M unique sampled occupation bit-strings (keys) with their wave-function
// values, generate the connected states key ^ xy_g of the packed Pauli Hamiltonian, look each one
// up among the samples, regenerate the matrix element on the fly and reduce the local energy
//   E_loc[i] = conj( sum_j H[i,j] psi[j] / psi[i] )
// (reference: src/optimizer/energy.py:219-263, src/optimizer/hamiltonian.py:272-370,
//  src_cpp/hamiltonian_math.pyx:85-100, src_cpp/sparse_math.pyx:85-100).
//
// Design (see DESIGN.md for the numbers):
//   * terms are packed CSR-by-unique-XY mask; masks are 32-bit when n_qubits <= 32 (all BASELINE
//     molecules), 64-bit otherwise (template parameter KT);
//   * groups are stored in three classes: the diagonal (xy == 0), "heavy" groups (single excitations,
//     tens of terms each) and "light" groups (double excitations, 4-6 terms each);
//   * one 64-lane wavefront owns one sample at a time (rows are dealt to the waves of a workgroup
//     round-robin); lanes stride over the XY groups, whose masks (and, when they fit, the
//     per-term YZ masks + coefficients) are staged once per workgroup in LDS;
//   * candidates that break particle-number conservation are rejected with two popcounts
//     (replaces the reference's 2^N look-up table), the rest are probed in an open-addressing hash
//     table of the sample keys that a small prep kernel rebuilds every call (L2 resident);
//   * light hits are compacted across the wave (ballot + mbcnt) into a per-wave LDS queue so that the
//     sign-sum loops run with full lanes; the diagonal and the heavy hits are evaluated
//     wave-cooperatively (lanes stride the terms of one group);
//   * per-sample result is a wave reduction (DPP shuffles) + one complex division.
// Everything is integer/bit work plus a handful of f64 adds per hit: the bound is cache/HBM
// traffic and issue rate, not MFMA, so there is deliberately no matrix-core code here.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <vector>

#include "naqs_common.hpp"
#include "naqs_hash.hpp"

namespace naqs { thread_local hipError_t g_last_hip = hipSuccess; }

namespace {

using naqs::WAVE;
using naqs::DeviceGuard;
using naqs::env_int;

constexpr int BLOCK = 256;                 // helper kernels: 4 waves, one per SIMD of a CU
#ifndef NAQS_LIGHT_BATCH
#define NAQS_LIGHT_BATCH 2
#endif
constexpr int LIGHT_BATCH = NAQS_LIGHT_BATCH;             // 64-group chunks probed together (all loads in flight before any is used)
constexpr int LIGHT_BATCH_BLOOM = 4;                      // Bloom variant (VALU-bound candidate loop): amortise the per-batch scaffolding
constexpr int queue_cap(bool bloom) { return 64 * ((bloom ? LIGHT_BATCH_BLOOM : LIGHT_BATCH) + 1); }   // per-wave hit queue: < 64 carried + batch x 64 pushed
constexpr int LDS_BUDGET = 78 * 1024;      // per-workgroup dynamic LDS budget (160 KiB/CU -> 2 WGs/CU)
constexpr int LDS_BUDGET_BLOOM = 152 * 1024;   // Bloom variant: one workgroup per CU
constexpr int HEAVY_TERMS = 8;             // groups with more terms than this are "heavy"

using namespace naqs;   // Slot, hash_insert, probe_load, probe_resolve, hash_find, popc (naqs_hash.hpp)

constexpr int PREP_BLOCK = 64;   // one wave per workgroup: M = 10^4 still spreads over 157 CUs
template <typename KT>
__global__ __launch_bounds__(PREP_BLOCK) void prep_kernel(int64_t M, const uint64_t *__restrict__ keys,
                                                     const void *__restrict__ psi_in, int psi_kind,
                                                     KT *__restrict__ keys_out, double2 *__restrict__ psi_out,
                                                     Slot<KT> *__restrict__ tab, int bits, uint32_t tag,
                                                     uint32_t *__restrict__ bloom) {
    for (int64_t i = blockIdx.x * (int64_t)PREP_BLOCK + threadIdx.x; i < M; i += (int64_t)gridDim.x * PREP_BLOCK) {
        const KT k = (KT)keys[i];
        keys_out[i] = k;
        hash_insert(tab, bits, tag, k, (uint32_t)i);
        if (bloom != nullptr) naqs::bloom_insert<KT>(bloom, k);
        double a, b;
        if (psi_kind == NAQS_PSI_F32 || psi_kind == NAQS_LOGPSI_F32) {
            const float2 v = reinterpret_cast<const float2 *>(psi_in)[i];
            a = (double)v.x; b = (double)v.y;
        } else {
            const double2 v = reinterpret_cast<const double2 *>(psi_in)[i];
            a = v.x; b = v.y;
        }
        if (psi_kind == NAQS_LOGPSI_F32 || psi_kind == NAQS_LOGPSI_F64) {
            const double amp = exp(a);
            double s, c;
            sincos(b, &s, &c);
            a = amp * c; b = amp * s;
        }
        psi_out[i] = make_double2(a, b);
    }
}

// ------------------------------------------------------------------------------------------------
// main E_loc kernel
// ------------------------------------------------------------------------------------------------
template <typename KT>
struct ElocParams {
    // packed Hamiltonian, device group order: [diagonal (0|1)] [heavy ...] [light ...]
    const KT *xy_g;          // [Kxy]
    const int32_t *row_ptr;  // [Kxy+1]
    const KT *yz_t;          // [K]
    const double *c_t;       // [K]
    int32_t Kxy, K;
    int32_t has_diag;        // 1 when group 0 is the xy == 0 group
    int32_t light_begin;     // first light group = has_diag + n_heavy
    KT alpha_mask, beta_mask;
    int32_t n_alpha, n_beta; // < 0: no particle-number filter
    // sample table
    const KT *keys;          // [M]
    const double2 *psi;      // [M]
    const Slot<KT> *tab;
    int32_t bits;
    uint32_t tag;            // epoch << 24 of this call's hash table entries
    const uint32_t *bloom;   // BLOOM variant: the call's Bloom filter (global copy, staged into LDS)
    // rows to produce
    int64_t row_begin, n_rows;
    int32_t rows_per_block;
    int32_t matvec;          // 1: store sum_j H_ij psi_j itself (naqs_hmatvec) instead of conj(. / psi_i)
    double2 *eloc;           // [n_rows]
};

// sum over the 64 lanes, valid in lane 0 (every lane of row 0..3 holds its row's sum on the way): four DPP exchanges
// inside the rows of 16 lanes (xor 1, xor 2, mirror in 8, mirror in 16) and three scalar adds of the row sums.  Fixed
// order, no LDS traffic (__shfl_xor on a double is two ds_bpermute per step: 24 LDS round trips for a (re, im) pair).
template <int CTRL>
__device__ __forceinline__ double dpp_swap(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_swap<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_swap<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_swap<0x141>(v);         // row_half_mirror
    v += dpp_swap<0x140>(v);         // row_mirror
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// the builtin returns a signed int: go through uint32_t before widening, or bit 31 smears upwards
__device__ __forceinline__ uint32_t to_sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t to_sgpr(uint64_t v) {
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// a / b for complex f64 (Smith's algorithm, as numpy's complex128 division)
__device__ __forceinline__ double2 cdiv(double2 a, double2 b) {
    double2 q;
    if (fabs(b.x) >= fabs(b.y)) {
        const double rat = b.y / b.x, scl = 1.0 / (b.x + b.y * rat);
        q.x = (a.x + a.y * rat) * scl;
        q.y = (a.y - a.x * rat) * scl;
    } else {
        const double rat = b.x / b.y, scl = 1.0 / (b.y + b.x * rat);
        q.x = (a.x * rat + a.y) * scl;
        q.y = (a.y * rat - a.x) * scl;
    }
    return q;
}

// sum_{t in [t0,t1)} c_t * (-1)^{popcount(key & yz_t)}, sequential, ascending t: the reference's
// summation order for one matrix element (hamiltonian_math.pyx:95-98), so H_ij is bit-identical.
// c * (-1)^parity without a select: flip the sign bit of the high word
__device__ __forceinline__ double signed_term(double c, int parity) {
    return __hiloint2double(__double2hiint(c) ^ (int)((uint32_t)(parity & 1) << 31), __double2loint(c));
}

template <typename KT>
__device__ __forceinline__ double sign_sum(KT key, const KT *__restrict__ yz, const double *__restrict__ c,
                                           int t0, int t1) {
    // two terms per trip (loads of both issued together), still added in ascending t
    double h = 0.0;
    int t = t0;
    for (; t + 1 < t1; t += 2) {
        const double c0 = c[t], c1 = c[t + 1];
        const KT y0 = yz[t], y1 = yz[t + 1];
        h += signed_term(c0, popc((KT)(key & y0)));
        h += signed_term(c1, popc((KT)(key & y1)));
    }
    if (t < t1) h += signed_term(c[t], popc((KT)(key & yz[t])));
    return h;
}

// the same sum with the lanes of a wave striding the terms (partial sum per lane)
template <typename KT>
__device__ __forceinline__ double sign_sum_strided(KT key, const KT *__restrict__ yz, const double *__restrict__ c,
                                                   int t0, int t1, int lane) {
    double h = 0.0;
    for (int t = t0 + lane; t < t1; t += WAVE) h += signed_term(c[t], popc((KT)(key & yz[t])));
    return h;
}

template <typename KT, int STAGE, int NT, bool BLOOM>
__global__ __launch_bounds__(NT) void eloc_kernel(const ElocParams<KT> p) {
    constexpr int NWAVES = NT / WAVE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS layout: [c_t: K doubles][queue: NWAVES*QUEUE_CAP int2][row_ptr: Kxy+1 (+pad)][xy: Kxy KT][yz: K KT]
    double *s_c = reinterpret_cast<double *>(smem);
    int2 *s_queue = reinterpret_cast<int2 *>(s_c + (STAGE >= 2 ? p.K : 0));
    constexpr int QUEUE_CAP = queue_cap(BLOOM);
    int32_t *s_rp = reinterpret_cast<int32_t *>(s_queue + NWAVES * QUEUE_CAP);
    KT *s_xy = reinterpret_cast<KT *>(s_rp + (STAGE >= 1 ? (p.Kxy + 2) & ~1 : 0));
    KT *s_yz = s_xy + (STAGE >= 1 ? p.Kxy : 0);
    // BLOOM: 64 KiB filter after everything else (offset rounded up to 16 bytes)
    uint32_t *s_bloom = reinterpret_cast<uint32_t *>(
        smem + ((reinterpret_cast<unsigned char *>(s_yz + (STAGE >= 2 ? p.K : 0)) - smem + 15) & ~15ull));

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;

    if (BLOOM) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.bloom);
        uint4 *dst = reinterpret_cast<uint4 *>(s_bloom);
#pragma unroll 4
        for (int e = tid; e < BLOOM_WORDS / 4; e += NT) dst[e] = src[e];
    }
    if (STAGE >= 1) {
#pragma unroll 4
        for (int g = tid; g < p.Kxy; g += NT) s_xy[g] = p.xy_g[g];
#pragma unroll 4
        for (int g = tid; g <= p.Kxy; g += NT) s_rp[g] = p.row_ptr[g];
    }
    if (STAGE >= 2) {
#pragma unroll 4
        for (int t = tid; t < p.K; t += NT) s_yz[t] = p.yz_t[t];
#pragma unroll 4
        for (int t = tid; t < p.K; t += NT) s_c[t] = p.c_t[t];
    }
    __syncthreads();

    const KT *xy = STAGE >= 1 ? s_xy : p.xy_g;
    const int32_t *rp = STAGE >= 1 ? s_rp : p.row_ptr;
    const KT *yz = STAGE >= 2 ? s_yz : p.yz_t;
    const double *cf = STAGE >= 2 ? s_c : p.c_t;
    int2 *queue = s_queue + wave * QUEUE_CAP;

    const int64_t blk_begin = (int64_t)blockIdx.x * p.rows_per_block;
    const int blk_rows = (int)min((int64_t)p.rows_per_block, p.n_rows - blk_begin);
    const bool filter = p.n_alpha >= 0;
    const Slot<KT> *__restrict__ tab = p.tab;
    const double2 *__restrict__ psi = p.psi;

    // candidate worth a look-up in the L2-resident hash table: particle numbers conserved and (BLOOM) the
    // LDS-resident Bloom filter does not rule it out
    auto physical = [&](KT j) {
        bool ok = !filter || (popc((KT)(j & p.alpha_mask)) == p.n_alpha && popc((KT)(j & p.beta_mask)) == p.n_beta);
        if (BLOOM && ok) ok = naqs::bloom_test<KT>(s_bloom, j);
        return ok;
    };

    // Rows of the workgroup are dealt to its waves round-robin.  (A dynamic deal through an LDS
    // counter — `if (lane == 0) rl = atomicAdd(s_next, 1); rl = readfirstlane(rl)` — is miscompiled by
    // ROCm 7.2's hipcc into a loop that re-reads a stale register and never terminates on gfx950;
    // measured, then removed.)
    for (int rl = wave; rl < blk_rows; rl += NWAVES) {
        const int64_t r = blk_begin + rl;
        const int64_t i = p.row_begin + r;
        const KT key = to_sgpr(p.keys[i]);
        const double2 psi_i = psi[i];
        double sr = 0.0, si = 0.0;

        // ---- diagonal group: lanes stride its terms
        if (p.has_diag) {
            const double h = sign_sum_strided<KT>(key, yz, cf, rp[0], rp[1], lane);
            sr = h * psi_i.x;
            si = h * psi_i.y;
        }

        // ---- heavy groups: one probe per lane, then each hit is summed by the whole wave
        for (int g0 = p.has_diag; g0 < p.light_begin; g0 += WAVE) {
            const int g = g0 + lane;
            int idx = -1;
            if (g < p.light_begin) {
                const KT j = key ^ xy[g];
                if (physical(j)) idx = hash_find<KT>(tab, p.bits, p.tag, j);
            }
            // every hit lane fetches its psi_j now (one round trip for all hits of the chunk); the wave then
            // walks the hits and gets (psi_j) by lane broadcast instead of a dependent load per hit
            double2 pj_mine = make_double2(0.0, 0.0);
            if (idx >= 0) pj_mine = psi[idx];
            unsigned long long m = __ballot(idx >= 0);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const double pjx = __shfl(pj_mine.x, b, WAVE), pjy = __shfl(pj_mine.y, b, WAVE);
                const double h = sign_sum_strided<KT>(key, yz, cf, rp[g0 + b], rp[g0 + b + 1], lane);
                sr += h * pjx;
                si += h * pjy;
            }
        }

        // ---- light groups: two probes in flight per lane, hits compacted into the wave's queue
        int qn = 0;  // wave-uniform number of queued hits
        auto drain = [&](int first, bool active) {
            if (active) {
                const int2 e = queue[first + lane];
                const double2 pj = psi[e.y];
                const double h = sign_sum<KT>(key, yz, cf, rp[e.x], rp[e.x + 1]);
                sr += h * pj.x;
                si += h * pj.y;
            }
        };
        auto push = [&](int g, int idx) {
            const unsigned long long hits = __ballot(idx >= 0);
            if (hits) {
                if (idx >= 0) {
                    const int pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0));
                    queue[pos] = make_int2(g, idx);
                }
                qn += __popcll(hits);
            }
        };
        constexpr int LB = BLOOM ? LIGHT_BATCH_BLOOM : LIGHT_BATCH;
        for (int g0 = p.light_begin; g0 < p.Kxy; g0 += LB * WAVE) {
            // all LB probes of a lane are issued before the first one is looked at: one L2 round trip
            // per 256 groups instead of one per 64
            KT j[LB];
            uint32_t hh[LB];
            bool ph[LB];
            decltype(probe_load(tab, 0u)) first[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int g = g0 + u * WAVE + lane;
                j[u] = 0; hh[u] = 0; ph[u] = false;
                if (g < p.Kxy) { j[u] = key ^ xy[g]; ph[u] = physical(j[u]); }
                if (ph[u]) { hh[u] = hash_key(j[u], p.bits); first[u] = probe_load(tab, hh[u]); }
            }
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int idx = ph[u] ? probe_resolve(tab, p.bits, p.tag, j[u], hh[u], first[u]) : -1;
                push(g0 + u * WAVE + lane, idx);
            }
            __builtin_amdgcn_wave_barrier();
            while (qn >= WAVE) {
                qn -= WAVE;
                drain(qn, true);
                __builtin_amdgcn_wave_barrier();
            }
        }
        drain(0, lane < qn);
        __builtin_amdgcn_wave_barrier();

        sr = wave_sum(sr);
        si = wave_sum(si);
        if (lane == 0) {
            if (p.matvec) {
                p.eloc[r] = make_double2(sr, si);
            } else {
                const double2 q = cdiv(make_double2(sr, si), psi_i);
                p.eloc[r] = make_double2(q.x, -q.y);   // conj, energy.py:248
            }
        }
    }

}

// ------------------------------------------------------------------------------------------------
// eloc_kernel2: the same row walk with the candidates compacted TWICE.  In eloc_kernel a lane owns a group through
// filter, probe and push, so after the particle-number filter (N2: 24 % pass, Li2O: ~15 %) the probe code — the bulk
// of the ~75 instructions a candidate slot costs — issues for a quarter-full wave.  Here
//   A. every lane filters one group per pass and the survivors' group numbers are compacted (ballot + mbcnt) into a
//      per-wave LDS "pass queue";
//   B. whenever 64 are queued, one dense pass probes them (Bloom test first in the BLOOM variant); hits go to the
//      hit queue as (term range, sample index) entries;
//   C. whenever 64 hits are queued every lane sums one entry's terms sequentially (as before).
// Heavy groups (single excitations, tens of terms) no longer stall the wave per hit: the kernel walks a view of the
// tables in which they are cut into chunks of <= HEAVY_TERMS terms, each a group of its own with the parent's flip mask
// (naqs_ham::d_xy2 / d_rp2), so only the diagonal keeps the lanes-stride-terms form.
// Light groups have <= HEAVY_TERMS terms by definition: one entry, summed in ascending term order = the reference's
// order for a matrix element.  A chunked heavy element is the sum of its chunks' partial sums, each multiplied by
// psi_j separately (equal to ~1 ulp; the strided form was not sequential either).
// ------------------------------------------------------------------------------------------------
constexpr int PASS_BATCH = 3;                              // filter passes between two looks at the pass queue
constexpr int PQ_CAP = 64 * (PASS_BATCH + 1), HQ_CAP = 128;  // < 64 carried + what one batch / one probe pass can push
constexpr size_t queue_bytes_v2(int nwaves) { return (size_t)nwaves * (HQ_CAP * sizeof(int2) + PQ_CAP * sizeof(int32_t)) + 16; }

template <typename KT, int STAGE, int NT, bool BLOOM>
__global__ __launch_bounds__(NT) void eloc_kernel2(const ElocParams<KT> p) {
    constexpr int NWAVES = NT / WAVE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS layout: [c_t: K doubles][hit queues: NWAVES*HQ_CAP int2][pass queues: NWAVES*PQ_CAP int][row_ptr][xy][yz][bloom]
    double *s_c = reinterpret_cast<double *>(smem);
    int2 *s_hq = reinterpret_cast<int2 *>(s_c + (STAGE >= 2 ? p.K : 0));
    int32_t *s_pq = reinterpret_cast<int32_t *>(s_hq + NWAVES * HQ_CAP);
    int32_t *s_rp = s_pq + NWAVES * PQ_CAP + 4;
    KT *s_xy = reinterpret_cast<KT *>(s_rp + (STAGE >= 1 ? (p.Kxy + 2) & ~1 : 0));
    KT *s_yz = s_xy + (STAGE >= 1 ? p.Kxy : 0);
    uint32_t *s_bloom = reinterpret_cast<uint32_t *>(
        smem + ((reinterpret_cast<unsigned char *>(s_yz + (STAGE >= 2 ? p.K : 0)) - smem + 15) & ~15ull));

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;

    if (BLOOM) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.bloom);
        uint4 *dst = reinterpret_cast<uint4 *>(s_bloom);
#pragma unroll 4
        for (int e = tid; e < BLOOM_WORDS / 4; e += NT) dst[e] = src[e];
    }
    if (STAGE >= 1) {
#pragma unroll 4
        for (int g = tid; g < p.Kxy; g += NT) s_xy[g] = p.xy_g[g];
#pragma unroll 4
        for (int g = tid; g <= p.Kxy; g += NT) s_rp[g] = p.row_ptr[g];
    }
    if (STAGE >= 2) {
#pragma unroll 4
        for (int t = tid; t < p.K; t += NT) s_yz[t] = p.yz_t[t];
#pragma unroll 4
        for (int t = tid; t < p.K; t += NT) s_c[t] = p.c_t[t];
    }
    __syncthreads();

    const KT *xy = STAGE >= 1 ? s_xy : p.xy_g;
    const int32_t *rp = STAGE >= 1 ? s_rp : p.row_ptr;
    const KT *yz = STAGE >= 2 ? s_yz : p.yz_t;
    const double *cf = STAGE >= 2 ? s_c : p.c_t;
    int2 *hq = s_hq + wave * HQ_CAP;
    int32_t *pq = s_pq + wave * PQ_CAP;

    const int64_t blk_begin = (int64_t)blockIdx.x * p.rows_per_block;
    const int blk_rows = (int)min((int64_t)p.rows_per_block, p.n_rows - blk_begin);
    const bool filter = p.n_alpha >= 0;
    const Slot<KT> *__restrict__ tab = p.tab;
    const double2 *__restrict__ psi = p.psi;

    for (int rl = wave; rl < blk_rows; rl += NWAVES) {
        const int64_t r = blk_begin + rl;
        const int64_t i = p.row_begin + r;
        const KT key = to_sgpr(p.keys[i]);
        const double2 psi_i = psi[i];
        double sr = 0.0, si = 0.0;

        // ---- diagonal group: lanes stride its terms
        if (p.has_diag) {
            const double h = sign_sum_strided<KT>(key, yz, cf, rp[0], rp[1], lane);
            sr = h * psi_i.x;
            si = h * psi_i.y;
        }

        int qn = 0, pn = 0;                                  // wave-uniform fill of the hit / pass queue
        // C: one queued (term range, sample) entry per lane
        auto drain = [&](int first, bool active) {
            if (active) {
                const int2 e = hq[first + lane];
                const double2 pj = psi[e.y];
                const int t0 = e.x & 0xFFFFFF;
                const double h = sign_sum<KT>(key, yz, cf, t0, t0 + (int)((uint32_t)e.x >> 24));
                sr += h * pj.x;
                si += h * pj.y;
            }
        };
        // hits of one dense pass -> hit queue (every group of this view has <= HEAVY_TERMS terms: one entry per hit)
        auto push_hits = [&](int g, int idx) {
            const unsigned long long m = __ballot(idx >= 0);
            if (m) {
                if (idx >= 0) {
                    const int pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                    const int t0 = rp[g];
                    hq[pos] = make_int2(t0 | ((rp[g + 1] - t0) << 24), idx);
                }
                qn += __popcll(m);
                __builtin_amdgcn_wave_barrier();
                if (qn >= WAVE) {
                    qn -= WAVE;
                    drain(qn, true);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        };
        // B: probe the queued candidates [first, first + 64)
        auto probe_pass = [&](int first, bool active) {
            int idx = -1, g = 0;
            if (active) {
                g = pq[first + lane];
                const KT j = key ^ xy[g];
                if (!BLOOM || naqs::bloom_test<KT>(s_bloom, j)) idx = hash_find<KT>(tab, p.bits, p.tag, j);
            }
            push_hits(g, idx);
        };
        // A: particle-number filter, one group per lane and pass; whole batches first (no bound check), then the tail
        auto filter_pass = [&](int g, bool in_range) {
            bool ok = false;
            if (in_range) {
                const KT j = key ^ xy[g];
                ok = !filter || (popc((KT)(j & p.alpha_mask)) == p.n_alpha && popc((KT)(j & p.beta_mask)) == p.n_beta);
            }
            const unsigned long long m = __ballot(ok);
            if (ok) pq[pn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = g;
            pn += __popcll(m);
        };
        int g0 = p.has_diag;
        for (; g0 + PASS_BATCH * WAVE <= p.Kxy; g0 += PASS_BATCH * WAVE) {
#pragma unroll
            for (int u = 0; u < PASS_BATCH; ++u) filter_pass(g0 + u * WAVE + lane, true);
            __builtin_amdgcn_wave_barrier();
            while (pn >= WAVE) {
                pn -= WAVE;
                probe_pass(pn, true);
            }
        }
        for (; g0 < p.Kxy; g0 += WAVE) {
            filter_pass(g0 + lane, g0 + lane < p.Kxy);
            __builtin_amdgcn_wave_barrier();
            if (pn >= WAVE) {
                pn -= WAVE;
                probe_pass(pn, true);
            }
        }
        probe_pass(0, lane < pn);
        drain(0, lane < qn);
        __builtin_amdgcn_wave_barrier();

        sr = wave_sum(sr);
        si = wave_sum(si);
        if (lane == 0) {
            if (p.matvec) {
                p.eloc[r] = make_double2(sr, si);
            } else {
                const double2 q = cdiv(make_double2(sr, si), psi_i);
                p.eloc[r] = make_double2(q.x, -q.y);   // conj, energy.py:248
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weighted reduction (deterministic: fixed grid-stride order + LDS tree), one workgroup
// ------------------------------------------------------------------------------------------------
constexpr int RED_BLOCK = 1024;
__global__ __launch_bounds__(RED_BLOCK) void reduce_kernel(int64_t n, const double *__restrict__ w,
                                                           const double2 *__restrict__ e, double *__restrict__ out4) {
    __shared__ double s[4][RED_BLOCK / WAVE];
    double a = 0, b = 0, c = 0, d = 0;
    for (int64_t i = threadIdx.x; i < n; i += RED_BLOCK) {
        const double wi = w[i];
        const double2 ei = e[i];
        a += wi * ei.x; b += wi * ei.y; c += wi * ei.x * ei.x; d += wi;
    }
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); d = wave_sum(d);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { s[0][wv] = a; s[1][wv] = b; s[2][wv] = c; s[3][wv] = d; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0;
        for (int k = 0; k < RED_BLOCK / WAVE; ++k) t += s[threadIdx.x][k];
        out4[threadIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// inner-ring kernels
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BLOCK) void parity_kernel(const T *__restrict__ a, int64_t n, int8_t *__restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t v = (uint64_t)(int64_t)a[i];   // sign-extends like the C promotion in the reference
        out[i] = (int8_t)(1 - 2 * (__popcll(v) & 1));
    }
}

// get_Hij_cy with the reference's own argument list (hamiltonian_math.pyx:85-100): the caller hands over the parity
// table P[M][Kyz] and the term -> (column, parity column) maps instead of bit masks.  Terms arrive grouped by output
// column (stable, i.e. ascending term index inside a column), so the sequential sum of one thread reproduces the
// reference's `H_ij[i*Kxy + g(k)] += P[i, y(k)] * c[k]` for k = 0..K-1 addend by addend.
template <typename T>
__global__ __launch_bounds__(BLOCK) void hij_parity_kernel(int64_t M, int32_t Kxy, int64_t Kyz, const int8_t *__restrict__ P,
                                                           const int32_t *__restrict__ gp, const int32_t *__restrict__ yt,
                                                           const T *__restrict__ ct, T *__restrict__ out) {
    const int64_t total = M * (int64_t)Kxy;
    for (int64_t e = blockIdx.x * (int64_t)BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * BLOCK) {
        const int64_t i = e / Kxy;
        const int g = (int)(e - i * Kxy);
        const int8_t *row = P + i * Kyz;
        T acc = (T)0;
        for (int t = gp[g]; t < gp[g + 1]; ++t) acc += (T)row[yt[t]] * ct[t];
        out[e] = acc;
    }
}

template <typename KT>
__global__ __launch_bounds__(BLOCK) void hij_kernel(int64_t M, int32_t Kxy, const uint64_t *__restrict__ keys,
                                                    const int32_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                    const KT *__restrict__ yz, const double *__restrict__ c,
                                                    double *__restrict__ out) {
    const int64_t total = M * (int64_t)Kxy;
    for (int64_t e = blockIdx.x * (int64_t)BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * BLOCK) {
        const int64_t i = e / Kxy;
        const int g = (int)(e - i * Kxy);       // device group order; col[g] = reference (ascending-xy) column
        out[i * Kxy + col[g]] = sign_sum<KT>((KT)keys[i], yz, c, rp[g], rp[g + 1]);
    }
}

__global__ __launch_bounds__(BLOCK) void csr_mv_kernel(int64_t rows, const double *__restrict__ data,
                                                       const int32_t *__restrict__ indices,
                                                       const int32_t *__restrict__ indptr,
                                                       const double2 *__restrict__ v, double2 *__restrict__ out) {
    // one wavefront per row; lanes stride the row, fixed-order tree at the end
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (blockIdx.x * (int64_t)BLOCK + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
    for (int64_t r = wave_id; r < rows; r += n_waves) {
        double re = 0, im = 0;
        for (int pidx = indptr[r] + lane; pidx < indptr[r + 1]; pidx += WAVE) {
            const double d = data[pidx];
            const double2 x = v[indices[pidx]];
            re += d * x.x; im += d * x.y;
        }
        re = wave_sum(re); im = wave_sum(im);
        if (lane == 0) out[r] = make_double2(re, im);
    }
}

}  // namespace

// ================================================================================================
// host side
// ================================================================================================
struct naqs_ham {
    int device = 0;
    int n_qubits = 0, n_alpha = -1, n_beta = -1;
    int key_bits = 32;
    int64_t K = 0, Kxy = 0;
    int32_t has_diag = 0, diag_terms = 0, n_heavy = 0;
    uint64_t alpha_mask = 0, beta_mask = 0;
    // device tables, group order [diagonal][heavy][light]; d_col[g] = reference (ascending-xy) column
    void *d_xy = nullptr, *d_yz = nullptr;
    int32_t *d_rp = nullptr, *d_col = nullptr;
    double *d_c = nullptr;
    // eloc_kernel2's view of the same term arrays: every non-diagonal group cut into chunks of <= HEAVY_TERMS terms, each
    // chunk a group of its own with the parent's flip mask (d_rp2 refines d_rp)
    int64_t Kxy2 = 0;
    void *d_xy2 = nullptr;
    int32_t *d_rp2 = nullptr;
    // scratch
    int64_t cap_M = 0;
    void *d_keys = nullptr;
    double2 *d_psi = nullptr;
    void *d_tab = nullptr;
    int64_t tab_slots = 0;
    uint32_t epoch = 0;                      // of the hash table entries (1..255; 0 = freshly zeroed)
    uint32_t *d_bloom = nullptr;             // Bloom filter of the current call's keys (large batches only)
    int cu_count = 256;
    naqs::EventRing prof;
};

namespace {

int table_bits(int64_t M) {
    int bits = 10;
    while ((1ll << bits) < 2 * M) ++bits;
    return bits;
}

int ensure_scratch(naqs_ham *h, int64_t M) {
    if (M <= h->cap_M) return NAQS_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (h->d_keys) (void)hipFree(h->d_keys);
    if (h->d_psi) (void)hipFree(h->d_psi);
    if (h->d_tab) (void)hipFree(h->d_tab);
    h->d_keys = nullptr; h->d_psi = nullptr; h->d_tab = nullptr; h->cap_M = 0;
    const int64_t cap = std::max<int64_t>(1024, M + M / 4);
    const size_t kb = h->key_bits / 8;
    const size_t slot = h->key_bits == 32 ? sizeof(Slot<uint32_t>) : sizeof(Slot<uint64_t>);
    h->tab_slots = 1ll << table_bits(cap);
    HIP_TRY(hipMalloc(&h->d_keys, cap * kb));
    HIP_TRY(hipMalloc((void **)&h->d_psi, cap * sizeof(double2)));
    HIP_TRY(hipMalloc(&h->d_tab, h->tab_slots * slot));
    HIP_TRY(hipMemset(h->d_tab, 0, h->tab_slots * slot));
    h->epoch = 0;
    h->cap_M = cap;
    return NAQS_OK;
}

template <typename KT>
int upload_tables(naqs_ham *h, const std::vector<uint64_t> &xy_g, const std::vector<int32_t> &rp,
                  const std::vector<int32_t> &col, const std::vector<uint64_t> &yz_t,
                  const std::vector<double> &c_t) {
    std::vector<KT> xy_n(xy_g.begin(), xy_g.end()), yz_n(yz_t.begin(), yz_t.end());
    HIP_TRY(hipMalloc(&h->d_xy, std::max<size_t>(1, xy_n.size()) * sizeof(KT)));
    HIP_TRY(hipMalloc(&h->d_yz, std::max<size_t>(1, yz_n.size()) * sizeof(KT)));
    HIP_TRY(hipMalloc((void **)&h->d_rp, rp.size() * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void **)&h->d_col, std::max<size_t>(1, col.size()) * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(h->d_col, col.data(), col.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&h->d_c, std::max<size_t>(1, c_t.size()) * sizeof(double)));
    HIP_TRY(hipMemcpy(h->d_xy, xy_n.data(), xy_n.size() * sizeof(KT), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_yz, yz_n.data(), yz_n.size() * sizeof(KT), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_rp, rp.data(), rp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_c, c_t.data(), c_t.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<KT> xy2;
    std::vector<int32_t> rp2;
    for (size_t g = 0; g + 1 < rp.size(); ++g) {
        const int32_t step = (int32_t)g < h->has_diag ? rp[g + 1] - rp[g] : HEAVY_TERMS;
        for (int32_t t = rp[g]; t < rp[g + 1]; t += std::max(step, 1)) { xy2.push_back(xy_n[g]); rp2.push_back(t); }
    }
    rp2.push_back(rp.back());
    h->Kxy2 = (int64_t)xy2.size();
    HIP_TRY(hipMalloc(&h->d_xy2, std::max<size_t>(1, xy2.size()) * sizeof(KT)));
    HIP_TRY(hipMalloc((void **)&h->d_rp2, rp2.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(h->d_xy2, xy2.data(), xy2.size() * sizeof(KT), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_rp2, rp2.data(), rp2.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return NAQS_OK;
}

// one call = begin (scratch, epoch) -> feed (prep_kernel here, or a producer kernel of naqs_logpsi.hip) -> main
int eloc_begin_impl(naqs_ham *h, int64_t M, hipStream_t s, naqs::ElocFeed *feed) {
    int st = ensure_scratch(h, M);
    if (st != NAQS_OK) return st;
    const size_t slot = h->key_bits == 32 ? sizeof(Slot<uint32_t>) : sizeof(Slot<uint64_t>);
    // no per-call clearing: entries are tagged with the call's epoch; zero the table when the 8-bit epoch wraps
    if (++h->epoch > 255u) {
        HIP_TRY(hipMemsetAsync(h->d_tab, 0, (size_t)h->tab_slots * slot, s));
        h->epoch = 1;
    }
    // Bloom filter for large batches in large spaces (most candidates absent): NAQS_BLOOM=1/0 forces it on/off
    const int bloom_env = env_int("NAQS_BLOOM", -1);
    const bool use_bloom = bloom_env >= 0 ? bloom_env != 0 : (M >= 20000);
    feed->bloom = nullptr;
    if (use_bloom) {
        if (!h->d_bloom) HIP_TRY(hipMalloc((void **)&h->d_bloom, BLOOM_WORDS * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(h->d_bloom, 0, BLOOM_WORDS * sizeof(uint32_t), s));
        feed->bloom = h->d_bloom;
    }
    feed->tab = h->d_tab;
    feed->keys_narrow = h->d_keys;
    feed->psi = h->d_psi;
    feed->bits = table_bits(M);
    feed->tag = h->epoch << 24;
    feed->key_bits = h->key_bits;
    return NAQS_OK;
}

template <typename KT>
int launch_prep(naqs_ham *h, int64_t M, const uint64_t *keys_dev, const void *psi_dev, int psi_kind,
                const naqs::ElocFeed &f, hipStream_t s) {
    const int grid = (int)std::min<int64_t>((M + PREP_BLOCK - 1) / PREP_BLOCK, 16 * h->cu_count);
    hipLaunchKernelGGL(prep_kernel<KT>, dim3(grid), dim3(PREP_BLOCK), 0, s, M, keys_dev, psi_dev, psi_kind,
                       reinterpret_cast<KT *>(h->d_keys), h->d_psi, reinterpret_cast<Slot<KT> *>(f.tab), f.bits, f.tag, f.bloom);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

template <typename KT>
int launch_main(naqs_ham *h, int64_t M, const naqs::ElocFeed &f, int64_t row_begin, int64_t n_rows, double *eloc_dev,
                const double *w_dev, double *out4_dev, hipStream_t s, bool matvec = false) {
    const int bits = f.bits;
    const uint32_t tag = f.tag;
    auto *tab = reinterpret_cast<Slot<KT> *>(f.tab);

    ElocParams<KT> p;
    p.xy_g = reinterpret_cast<const KT *>(h->d_xy);
    p.row_ptr = h->d_rp;
    p.yz_t = reinterpret_cast<const KT *>(h->d_yz);
    p.c_t = h->d_c;
    p.Kxy = (int32_t)h->Kxy; p.K = (int32_t)h->K;
    p.has_diag = h->has_diag;
    p.light_begin = h->has_diag + h->n_heavy;
    p.alpha_mask = (KT)h->alpha_mask; p.beta_mask = (KT)h->beta_mask;
    p.n_alpha = h->n_alpha; p.n_beta = h->n_beta;
    p.keys = reinterpret_cast<const KT *>(h->d_keys);
    p.psi = h->d_psi;
    p.tab = tab; p.bits = bits; p.tag = tag;
    p.row_begin = row_begin; p.n_rows = n_rows;
    p.eloc = reinterpret_cast<double2 *>(eloc_dev);
    p.matvec = matvec ? 1 : 0;

    // Workgroup shape.  The term tables are staged per workgroup, so big workgroups amortise the
    // staging; small batches still want every CU busy.  1024 threads = 16 waves = 4 per SIMD.
    int nt = env_int("NAQS_BLOCK", 0);
    if (nt != 256 && nt != 512 && nt != 1024) nt = n_rows >= 4096 ? 1024 : 256;
    const int nwaves = nt / WAVE;
    int rpb = env_int("NAQS_ROWS_PER_BLOCK", 0);
    if (rpb <= 0) {
        const int64_t target_blocks = (int64_t)h->cu_count * (2048 / nt);     // fill every CU's wave slots
        rpb = (int)std::max<int64_t>(nwaves, (n_rows + target_blocks - 1) / target_blocks);
    }
    p.rows_per_block = rpb;

    // Bloom variant (filter built by the feed stage for this call): 1024-thread workgroups, one per CU
    const bool bloom = f.bloom != nullptr && nt == 1024;
    p.bloom = bloom ? f.bloom : nullptr;
    const size_t b_bytes = bloom ? (size_t)BLOOM_WORDS * sizeof(uint32_t) + 16 : 0;
    const size_t budget = bloom ? (size_t)LDS_BUDGET_BLOOM : (size_t)LDS_BUDGET;
    if (bloom) {      // one workgroup per CU: give it more rows
        const int64_t target_blocks = (int64_t)h->cu_count;
        if (env_int("NAQS_ROWS_PER_BLOCK", 0) <= 0)
            rpb = (int)std::max<int64_t>(nwaves, (n_rows + target_blocks - 1) / target_blocks);
        p.rows_per_block = rpb;
    }
    const int grid2 = (int)((n_rows + rpb - 1) / rpb);
    // NAQS_ELOC_V=1: the single-compaction kernel (kept for A/B); the term range of a queue entry is 24 + 8 bits
    const bool v2 = env_int("NAQS_ELOC_V", 2) != 1 && h->K < (1 << 24);
    const int64_t Kxy_k = v2 ? h->Kxy2 : h->Kxy;
    if (v2) { p.xy_g = reinterpret_cast<const KT *>(h->d_xy2); p.row_ptr = h->d_rp2; p.Kxy = (int32_t)h->Kxy2; }
    const size_t q_bytes = v2 ? queue_bytes_v2(nwaves) : (size_t)nwaves * queue_cap(bloom) * sizeof(int2) + 16;
    const size_t g_bytes = (size_t)((Kxy_k + 2) & ~1ll) * sizeof(int32_t) + (size_t)Kxy_k * sizeof(KT);
    const size_t t_bytes = (size_t)h->K * (sizeof(double) + sizeof(KT));
    const int force = env_int("NAQS_STAGE", -1);   // tuning/testing: 0 none, 1 groups, 2 groups+terms
    int stage = (q_bytes + g_bytes + t_bytes + b_bytes <= budget) ? 2 : (q_bytes + g_bytes + b_bytes <= budget ? 1 : 0);
    if (force >= 0 && force < stage) stage = force;
    const size_t lds = q_bytes + (stage >= 1 ? g_bytes : 0) + (stage >= 2 ? t_bytes : 0) + b_bytes;

    const bool prof = h->prof.armed();
    if (prof) { int st = h->prof.begin(s); if (st != NAQS_OK) return st; }
#define NAQS_LAUNCH_K(KERNEL, ST, NTHREADS, BL)                                                                     \
    do {                                                                                                            \
        if (lds > 64 * 1024)  /* above the default dynamic-LDS limit */                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL<KT, ST, NTHREADS, BL>),                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget);                      \
        hipLaunchKernelGGL((KERNEL<KT, ST, NTHREADS, BL>), dim3(grid2), dim3(NTHREADS), lds, s, p);                 \
    } while (0)
#define NAQS_LAUNCH(ST, NTHREADS, BL)                                                                               \
    do {                                                                                                            \
        if (v2) NAQS_LAUNCH_K(eloc_kernel2, ST, NTHREADS, BL);                                                      \
        else NAQS_LAUNCH_K(eloc_kernel, ST, NTHREADS, BL);                                                          \
    } while (0)
#define NAQS_LAUNCH_NT(ST)                                          \
    do {                                                            \
        if (bloom) NAQS_LAUNCH(ST, 1024, true);                     \
        else if (nt == 1024) NAQS_LAUNCH(ST, 1024, false);          \
        else if (nt == 512) NAQS_LAUNCH(ST, 512, false);            \
        else NAQS_LAUNCH(ST, 256, false);                           \
    } while (0)
    if (stage == 2) NAQS_LAUNCH_NT(2);
    else if (stage == 1) NAQS_LAUNCH_NT(1);
    else NAQS_LAUNCH_NT(0);
#undef NAQS_LAUNCH_NT
#undef NAQS_LAUNCH
#undef NAQS_LAUNCH_K
    HIP_TRY(hipGetLastError());
    if (prof) { int st = h->prof.end(s); if (st != NAQS_OK) return st; }
    if (w_dev) {
        // Measured alternatives, both slower than this separate ~6 us launch (any tiny kernel costs 4-6 us of
        // queue time here, whatever it does): (1) the sums fused into eloc_kernel behind a last-workgroup ticket
        // (per-workgroup partials with write-through stores, parallel final add): 36.9 us vs 20.4 + 6.2 — every
        // workgroup pays a serial tail (weight load, LDS row sums, store drain, ticket round trip); with
        // agent-scope release fences instead of write-through stores 94 us (an L2 write-back per workgroup);
        // (2) a two-level multi-workgroup reduce kernel: 6.2 us, no better than one workgroup.
        hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(RED_BLOCK), 0, s, n_rows, w_dev,
                           reinterpret_cast<const double2 *>(eloc_dev), out4_dev);
        HIP_TRY(hipGetLastError());
    }
    return NAQS_OK;
}

}  // namespace

NAQS_API int naqs_abi_version(void) { return NAQS_ABI_VERSION; }

NAQS_API const char *naqs_strerror(int status) {
    switch (status) {
        case NAQS_OK: return "ok";
        case NAQS_ERR_INVALID: return "invalid argument";
        case NAQS_ERR_HIP: return "HIP runtime error (see naqs_last_hip_error_string)";
        case NAQS_ERR_NOMEM: return "out of memory";
        case NAQS_ERR_UNSUPPORTED: return "unsupported configuration";
        case NAQS_ERR_NO_DEVICE: return "no usable HIP device";
        default: return "unknown naqs status";
    }
}

NAQS_API int naqs_last_hip_error(void) { return (int)naqs::g_last_hip; }
NAQS_API const char *naqs_last_hip_error_string(void) { return hipGetErrorString(naqs::g_last_hip); }

NAQS_API int naqs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

NAQS_API int naqs_terms_group(int64_t K, const uint64_t *xy, const uint64_t *yz, const double *coeff,
                              int64_t *Kxy_out, uint64_t *xy_g, int32_t *row_ptr,
                              uint64_t *yz_t, double *c_t, int64_t *order) {
    if (K < 0 || !Kxy_out || !row_ptr) return NAQS_ERR_INVALID;
    if (K > 0 && (!xy || !yz || !coeff || !xy_g || !yz_t || !c_t)) return NAQS_ERR_INVALID;
    if (K >= (1ll << 31)) return NAQS_ERR_UNSUPPORTED;
    std::vector<int64_t> perm((size_t)K);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return xy[a] < xy[b]; });
    int64_t ng = 0;
    for (int64_t t = 0; t < K; ++t) {
        const int64_t k = perm[(size_t)t];
        if (t == 0 || xy[k] != xy_g[ng - 1]) { xy_g[ng] = xy[k]; row_ptr[ng] = (int32_t)t; ++ng; }
        yz_t[t] = yz[k];
        c_t[t] = coeff[k];
        if (order) order[t] = k;
    }
    row_ptr[ng] = (int32_t)K;
    *Kxy_out = ng;
    return NAQS_OK;
}

NAQS_API int naqs_ham_create(int n_qubits, int n_alpha, int n_beta, int64_t K,
                             const uint64_t *xy, const uint64_t *yz, const double *coeff,
                             int device, naqs_ham_t **out) {
    if (!out) return NAQS_ERR_INVALID;
    *out = nullptr;
    if (n_qubits <= 0 || K < 0 || (K > 0 && (!xy || !yz || !coeff))) return NAQS_ERR_INVALID;
    if (n_qubits > 64 || K >= (1ll << 31)) return NAQS_ERR_UNSUPPORTED;
    if ((n_alpha < 0) != (n_beta < 0)) return NAQS_ERR_INVALID;
    if (n_alpha > (n_qubits + 1) / 2 || n_beta > n_qubits / 2) return NAQS_ERR_INVALID;
    const uint64_t full = n_qubits == 64 ? ~0ull : ((1ull << n_qubits) - 1ull);
    for (int64_t k = 0; k < K; ++k)
        if ((xy[k] | yz[k]) & ~full) return NAQS_ERR_INVALID;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NAQS_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return NAQS_ERR_INVALID;

    naqs_ham *h = new (std::nothrow) naqs_ham();
    if (!h) return NAQS_ERR_NOMEM;
    h->device = device;
    h->n_qubits = n_qubits; h->n_alpha = n_alpha; h->n_beta = n_beta;
    h->key_bits = n_qubits <= 32 ? 32 : 64;
    h->K = K;
    for (int q = 0; q < n_qubits; ++q) ((q & 1) ? h->beta_mask : h->alpha_mask) |= 1ull << q;

    std::vector<uint64_t> xy_g((size_t)std::max<int64_t>(K, 1)), yz_t((size_t)std::max<int64_t>(K, 1));
    std::vector<int32_t> rp((size_t)K + 1);
    std::vector<double> c_t((size_t)std::max<int64_t>(K, 1));
    int64_t Kxy = 0;
    int st = naqs_terms_group(K, xy, yz, coeff, &Kxy, xy_g.data(), rp.data(), yz_t.data(), c_t.data(), nullptr);
    if (st != NAQS_OK) { delete h; return st; }
    xy_g.resize((size_t)Kxy); rp.resize((size_t)Kxy + 1); yz_t.resize((size_t)K); c_t.resize((size_t)K);
    h->Kxy = Kxy;

    // device group order: [diagonal][heavy: more than HEAVY_TERMS terms][light], each class in ascending xy
    std::vector<int32_t> col;                       // device position -> reference (ascending-xy) column
    col.reserve((size_t)Kxy);
    if (Kxy > 0 && xy_g[0] == 0) { h->has_diag = 1; h->diag_terms = rp[1] - rp[0]; col.push_back(0); }
    for (int64_t g = h->has_diag; g < Kxy; ++g)
        if (rp[(size_t)g + 1] - rp[(size_t)g] > HEAVY_TERMS) col.push_back((int32_t)g);
    h->n_heavy = (int32_t)col.size() - h->has_diag;
    for (int64_t g = h->has_diag; g < Kxy; ++g)
        if (rp[(size_t)g + 1] - rp[(size_t)g] <= HEAVY_TERMS) col.push_back((int32_t)g);
    std::vector<uint64_t> xy_d((size_t)Kxy), yz_d((size_t)K);
    std::vector<int32_t> rp_d((size_t)Kxy + 1);
    std::vector<double> c_d((size_t)K);
    int32_t pos = 0;
    for (int64_t d = 0; d < Kxy; ++d) {
        const int32_t g = col[(size_t)d];
        xy_d[(size_t)d] = xy_g[(size_t)g];
        rp_d[(size_t)d] = pos;
        for (int32_t t = rp[(size_t)g]; t < rp[(size_t)g + 1]; ++t, ++pos) {
            yz_d[(size_t)pos] = yz_t[(size_t)t];
            c_d[(size_t)pos] = c_t[(size_t)t];
        }
    }
    rp_d[(size_t)Kxy] = pos;

    DeviceGuard guard;
    st = guard.init(device);
    if (st == NAQS_OK) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->cu_count = prop.multiProcessorCount;
        st = h->key_bits == 32 ? upload_tables<uint32_t>(h, xy_d, rp_d, col, yz_d, c_d)
                               : upload_tables<uint64_t>(h, xy_d, rp_d, col, yz_d, c_d);
    }
    if (st != NAQS_OK) { naqs_ham_destroy(h); return st; }
    *out = h;
    return NAQS_OK;
}

NAQS_API int naqs_ham_destroy(naqs_ham_t *h) {
    if (!h) return NAQS_OK;
    DeviceGuard guard;
    (void)guard.init(h->device);
    (void)h->prof.enable(0);
    void *ptrs[] = {h->d_xy, h->d_yz, h->d_rp, h->d_col, h->d_c, h->d_xy2, h->d_rp2, h->d_keys, h->d_psi, h->d_tab, h->d_bloom};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete h;
    return NAQS_OK;
}

NAQS_API int naqs_ham_info(const naqs_ham_t *h, int64_t info[8]) {
    if (!h || !info) return NAQS_ERR_INVALID;
    info[0] = h->K; info[1] = h->Kxy; info[2] = h->n_qubits; info[3] = h->n_alpha; info[4] = h->n_beta;
    info[5] = h->key_bits; info[6] = h->diag_terms; info[7] = h->device;
    return NAQS_OK;
}

NAQS_API int naqs_ham_reserve(naqs_ham_t *h, int64_t M) {
    if (!h || M < 0) return NAQS_ERR_INVALID;
    if (M >= (1ll << 31)) return NAQS_ERR_UNSUPPORTED;
    DeviceGuard guard;
    int st = guard.init(h->device);
    if (st != NAQS_OK) return st;
    return ensure_scratch(h, M);
}

static int eloc_common(naqs_ham_t *h, int64_t M, const uint64_t *keys_dev, const void *psi_dev, int psi_kind,
                       int64_t row_begin, int64_t n_rows, double *eloc_dev, const double *w_dev, double *out4_dev,
                       void *stream, bool matvec = false) {
    if (!h || M < 0 || row_begin < 0 || n_rows < 0 || row_begin + n_rows > M) return NAQS_ERR_INVALID;
    if (psi_kind < NAQS_PSI_F32 || psi_kind > NAQS_LOGPSI_F64) return NAQS_ERR_INVALID;
    if (M > (int64_t)IDX_MASK) return NAQS_ERR_UNSUPPORTED;          // 24-bit sample index in a hash slot
    if ((w_dev == nullptr) != (out4_dev == nullptr)) return NAQS_ERR_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DeviceGuard guard;
    int st = guard.init(h->device);
    if (st != NAQS_OK) return st;
    if (n_rows == 0) {
        if (out4_dev) HIP_TRY(hipMemsetAsync(out4_dev, 0, 4 * sizeof(double), s));
        return NAQS_OK;
    }
    if (!keys_dev || !psi_dev || !eloc_dev) return NAQS_ERR_INVALID;
    naqs::ElocFeed feed;
    st = eloc_begin_impl(h, M, s, &feed);
    if (st != NAQS_OK) return st;
    st = h->key_bits == 32 ? launch_prep<uint32_t>(h, M, keys_dev, psi_dev, psi_kind, feed, s)
                           : launch_prep<uint64_t>(h, M, keys_dev, psi_dev, psi_kind, feed, s);
    if (st != NAQS_OK) return st;
    return h->key_bits == 32 ? launch_main<uint32_t>(h, M, feed, row_begin, n_rows, eloc_dev, w_dev, out4_dev, s, matvec)
                             : launch_main<uint64_t>(h, M, feed, row_begin, n_rows, eloc_dev, w_dev, out4_dev, s, matvec);
}

NAQS_API int naqs_hmatvec(naqs_ham_t *h, int64_t M, const uint64_t *keys_dev, const double *v_dev, int64_t row_begin,
                          int64_t n_rows, double *out_dev, void *stream) {
    return eloc_common(h, M, keys_dev, v_dev, NAQS_PSI_F64, row_begin, n_rows, out_dev, nullptr, nullptr, stream, true);
}

// hooks for the fused log-psi + E_loc entry point in naqs_logpsi.hip
int naqs::eloc_begin(naqs_ham *h, int64_t M, hipStream_t s, naqs::ElocFeed *feed) {
    if (!h || !feed || M <= 0 || M > (int64_t)IDX_MASK) return NAQS_ERR_INVALID;
    return eloc_begin_impl(h, M, s, feed);
}
int naqs::eloc_main(naqs_ham *h, int64_t M, const naqs::ElocFeed &feed, double *eloc_dev, const double *w_dev,
                    double *out4_dev, hipStream_t s) {
    return h->key_bits == 32 ? launch_main<uint32_t>(h, M, feed, 0, M, eloc_dev, w_dev, out4_dev, s)
                             : launch_main<uint64_t>(h, M, feed, 0, M, eloc_dev, w_dev, out4_dev, s);
}
int naqs::ham_device(const naqs_ham *h) { return h->device; }

NAQS_API int naqs_eloc(naqs_ham_t *h, int64_t M, const uint64_t *keys_dev, const void *psi_dev, int psi_kind,
                       int64_t row_begin, int64_t n_rows, double *eloc_dev, void *stream) {
    return eloc_common(h, M, keys_dev, psi_dev, psi_kind, row_begin, n_rows, eloc_dev, nullptr, nullptr, stream);
}

NAQS_API int naqs_eloc_reduced(naqs_ham_t *h, int64_t M, const uint64_t *keys_dev, const void *psi_dev, int psi_kind,
                               int64_t row_begin, int64_t n_rows, const double *w_dev, double *eloc_dev,
                               double *out4_dev, void *stream) {
    if (!w_dev || !out4_dev) return NAQS_ERR_INVALID;
    return eloc_common(h, M, keys_dev, psi_dev, psi_kind, row_begin, n_rows, eloc_dev, w_dev, out4_dev, stream);
}

NAQS_API int naqs_eloc_reduce(naqs_ham_t *h, int64_t n, const double *w_dev, const double *eloc_dev,
                              double *out4_dev, void *stream) {
    if (!h || n < 0 || !out4_dev || (n > 0 && (!w_dev || !eloc_dev))) return NAQS_ERR_INVALID;
    DeviceGuard guard;
    int st = guard.init(h->device);
    if (st != NAQS_OK) return st;
    hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(RED_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), n, w_dev,
                       reinterpret_cast<const double2 *>(eloc_dev), out4_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

NAQS_API int naqs_popcount_parity(const void *arr_dev, int elem_bytes, int64_t n, int8_t *out_dev, void *stream) {
    if (n < 0 || (n > 0 && (!arr_dev || !out_dev))) return NAQS_ERR_INVALID;
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return NAQS_ERR_UNSUPPORTED;
    if (n == 0) return NAQS_OK;
    const int grid = (int)std::min<int64_t>((n + BLOCK - 1) / BLOCK, 8192);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (elem_bytes == 1)
        hipLaunchKernelGGL(parity_kernel<int8_t>, dim3(grid), dim3(BLOCK), 0, s, (const int8_t *)arr_dev, n, out_dev);
    else if (elem_bytes == 2)
        hipLaunchKernelGGL(parity_kernel<int16_t>, dim3(grid), dim3(BLOCK), 0, s, (const int16_t *)arr_dev, n, out_dev);
    else if (elem_bytes == 4)
        hipLaunchKernelGGL(parity_kernel<int32_t>, dim3(grid), dim3(BLOCK), 0, s, (const int32_t *)arr_dev, n, out_dev);
    else
        hipLaunchKernelGGL(parity_kernel<int64_t>, dim3(grid), dim3(BLOCK), 0, s, (const int64_t *)arr_dev, n, out_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

NAQS_API int naqs_get_hij(naqs_ham_t *h, int64_t M, const uint64_t *keys_dev, double *hij_dev, void *stream) {
    if (!h || M < 0 || (M > 0 && (!keys_dev || !hij_dev))) return NAQS_ERR_INVALID;
    if (M == 0 || h->Kxy == 0) return NAQS_OK;
    DeviceGuard guard;
    int st = guard.init(h->device);
    if (st != NAQS_OK) return st;
    const int64_t total = M * h->Kxy;
    const int grid = (int)std::min<int64_t>((total + BLOCK - 1) / BLOCK, 16384);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (h->key_bits == 32)
        hipLaunchKernelGGL(hij_kernel<uint32_t>, dim3(grid), dim3(BLOCK), 0, s, M, (int32_t)h->Kxy, keys_dev, h->d_rp,
                           h->d_col, (const uint32_t *)h->d_yz, h->d_c, hij_dev);
    else
        hipLaunchKernelGGL(hij_kernel<uint64_t>, dim3(grid), dim3(BLOCK), 0, s, M, (int32_t)h->Kxy, keys_dev, h->d_rp,
                           h->d_col, (const uint64_t *)h->d_yz, h->d_c, hij_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

NAQS_API int naqs_hij_from_parity(int64_t M, int64_t Kxy, int64_t Kyz, int64_t K, const int8_t *parity_dev,
                                  const int32_t *group_ptr_dev, const int32_t *term_yz_dev, const void *term_coeff_dev,
                                  int coeff_bytes, void *hij_dev, void *stream) {
    if (M < 0 || Kxy < 0 || Kyz < 0 || K < 0 || Kxy > INT32_MAX || K > INT32_MAX) return NAQS_ERR_INVALID;
    if (coeff_bytes != 4 && coeff_bytes != 8) return NAQS_ERR_UNSUPPORTED;
    if (M == 0 || Kxy == 0) return NAQS_OK;
    if (!group_ptr_dev || !hij_dev || (K > 0 && (!parity_dev || !term_yz_dev || !term_coeff_dev))) return NAQS_ERR_INVALID;
    const int64_t total = M * Kxy;
    const int grid = (int)std::min<int64_t>((total + BLOCK - 1) / BLOCK, 16384);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (coeff_bytes == 8)
        hipLaunchKernelGGL(hij_parity_kernel<double>, dim3(grid), dim3(BLOCK), 0, s, M, (int32_t)Kxy, Kyz, parity_dev,
                           group_ptr_dev, term_yz_dev, (const double *)term_coeff_dev, (double *)hij_dev);
    else
        hipLaunchKernelGGL(hij_parity_kernel<float>, dim3(grid), dim3(BLOCK), 0, s, M, (int32_t)Kxy, Kyz, parity_dev,
                           group_ptr_dev, term_yz_dev, (const float *)term_coeff_dev, (float *)hij_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

NAQS_API int naqs_csr_mv(int64_t rows, const double *data_dev, const int32_t *indices_dev,
                         const int32_t *indptr_dev, const double *v_dev, double *out_dev, void *stream) {
    if (rows < 0 || (rows > 0 && (!indptr_dev || !v_dev || !out_dev))) return NAQS_ERR_INVALID;
    if (rows == 0) return NAQS_OK;
    const int grid = (int)std::min<int64_t>((rows + BLOCK / WAVE - 1) / (BLOCK / WAVE), 8192);
    hipLaunchKernelGGL(csr_mv_kernel, dim3(grid), dim3(BLOCK), 0, reinterpret_cast<hipStream_t>(stream), rows,
                       data_dev, indices_dev, indptr_dev, reinterpret_cast<const double2 *>(v_dev),
                       reinterpret_cast<double2 *>(out_dev));
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

NAQS_API int naqs_prof_enable(naqs_ham_t *h, int max_records) {
    if (!h || max_records < 0) return NAQS_ERR_INVALID;
    DeviceGuard guard;
    int st = guard.init(h->device);
    if (st != NAQS_OK) return st;
    return h->prof.enable(max_records);
}

NAQS_API int naqs_prof_read(naqs_ham_t *h, double *total_ms, int64_t *launches) {
    if (!h || !total_ms || !launches) return NAQS_ERR_INVALID;
    DeviceGuard guard;
    int st = guard.init(h->device);
    if (st != NAQS_OK) return st;
    return h->prof.read(total_ms, launches);
}

NAQS_API int naqs_prof_stride(naqs_ham_t *h, int stride) {
    if (!h || stride < 1) return NAQS_ERR_INVALID;
    h->prof.stride = stride;
    h->prof.tick = 0;
    return NAQS_OK;
}
