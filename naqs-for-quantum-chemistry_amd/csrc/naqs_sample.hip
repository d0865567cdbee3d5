// naqs_sample.hip — the autoregressive tree sampler on gfx950 (MI355X).
//
// Semantics: ComplexAutoregressiveMachine1D_OrbitalNade._forward_sample (src/naqs/network/nade.py:632-736) with
// multinomial_arr (:20-37) of the reference: n_samples draws are represented as a tree of unique prefixes with
// counts.  Per orbital pair n every live prefix evaluates its conditional (amplitude block n -> symmetrise ->
// electron-budget mask -> 0.5 log_softmax(2x)), p = exp(.)^2 in float32 renormalised in float64 (:673-683), splits
// its count over the four outcomes with the conditional-binomial chain, drops un-physical children (:695) and
// the survivors become the next level.  The reference does this with ~25 host-driven array operations and a
// host binomial per level; here a level is two launches and nothing returns to the host until the end:
//
//   sample_expand_kernel   one thread per prefix: block-n MLP with the weights staged in LDS (the code path of
//                          amp_kernel), conditional, binomial chain (naqs_rng.hpp: Philox-keyed by the prefix
//                          itself, so the draw does not depend on where the prefix sits in the arrays), children
//                          counts -> scratch, survivors per workgroup -> wg_total.
//   sample_scatter_kernel  stream compaction in (prefix, outcome) order: workgroup offset = sum of the preceding
//                          totals, in-workgroup exclusive scan, children written to the other half of the
//                          ping-pong arrays.  With qubit_ordering = -1 this order IS ascending key order, which
//                          is how the reference's samples come out.
//
// The level sizes live on the device (U[n]); grids are sized for the worst case min(4^n, cap) and surplus
// workgroups exit at once.  More than max_unique live prefixes at any level sets the overflow flag (the
// reference raises MaxBatchSizeExceededError, nade.py:710-712) and the remaining launches fall through.
// Statistical, not bitwise, parity with the reference (its generator is numpy's): tests/test_sampler_gpu.py.

#include <algorithm>
#include <cstdint>
#include <atomic>
#include <cstdio>

#include "naqs_common.hpp"
#include "naqs_net.hpp"
#include "naqs_amp_mfma.hpp"
#include "naqs_rng.hpp"
#include "naqs_pack.hpp"

#if defined(NAQS_SAMPLE_STATS)
namespace naqs { __device__ unsigned long long g_sample_stats[2][4][4]; }
// (developer build only) -> out[32]: [G == 4 ? 0 : 1][class][calls, BTRS rounds, exact tests, steps of the longest inversion], and reset
extern "C" __attribute__((visibility("default"))) int naqs_debug_sample_stats(unsigned long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(naqs::g_sample_stats), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long zero[32] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(naqs::g_sample_stats), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

#if defined(NAQS_HEAD_CLOCKS)
// (developer build only, tools/head_clock_probe.py) cycle stamps of sample_head_kernel's thread 0, accumulated over calls:
// [level][0 level begins, 1 probabilities in registers, 2 both draws done, 3 children compacted and written] + [7][0] = calls
__device__ long long g_head_clk[8][4];
extern "C" __attribute__((visibility("default"))) int naqs_debug_head_clocks(long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_head_clk), 32 * sizeof(long long)) != hipSuccess) return -1;
    long long zero[32] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_head_clk), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#define HEAD_MARK(level, stage) do { if (threadIdx.x == 0) g_head_clk[level][stage] += clock64() - head_t0; } while (0)
// ... and of sample_multi_kernel's first and last workgroup: [launch: 0 not the last of the call, 1 the last][workgroup: 0 first,
// 1 last][row: level 0..3 of the launch, 4 = look-back / children written / kernel ends, 5 = (launches counted, NL summed)][stage]
__device__ long long g_multi_clk[2][2][6][4];
extern "C" __attribute__((visibility("default"))) int naqs_debug_multi_clocks(long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_multi_clk), 96 * sizeof(long long)) != hipSuccess) return -1;
    long long zero[96] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_multi_clk), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#define MULTI_MARK(row, stage) do { if (threadIdx.x == 0 && mm_wg >= 0) g_multi_clk[last ? 1 : 0][mm_wg][row][stage] += clock64() - mm_t0; } while (0)
#else
#define HEAD_MARK(level, stage) do {} while (0)
#define MULTI_MARK(row, stage) do {} while (0)
#endif

namespace {

using naqs::MAXP;
using naqs::NetDims;
using naqs::WAVE;
using naqs::DeviceGuard;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SB = 256;                     // threads per workgroup; prefixes per workgroup of the scatter kernel
constexpr int EXP_PARENTS = SB / 4;         // prefixes per workgroup of the expand kernel (a quad of lanes each)
constexpr int U_SLOTS = MAXP + 2;           // U[0..P] level sizes, U[MAXP + 1] overflow flag

struct SampleBufs {
    uint32_t *ab[2];                        // prefix occupations in model order: alpha string | beta string << 16
    int64_t *cnt[2];
    float *prob[2];
    int64_t *child_cnt;                     // [cap][4]
    float *child_prob;                      // [cap][4]
    uint32_t *wg_total;                     // survivors per workgroup of the current level
    unsigned long long *wg_state;           // fused level kernel: (tag << 32 | survivors) per workgroup, never cleared
    int64_t *U;
    const naqs::PollCtl *ctl;               // bounded waits of the look-back (naqs_poll.hpp)
};

__global__ void sample_init_kernel(SampleBufs b, int64_t n_samples) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < U_SLOTS; ++i) b.U[i] = 0;
        b.U[0] = 1;
        b.ab[0][0] = 0u;
        b.cnt[0][0] = n_samples;
        b.prob[0][0] = 1.0f;
    }
}

// (publish_info — (M, overflow) for a host that polls mapped memory — is naqs_net.hpp's: written as soon as the size of the last
// level is known (the workgroup that closes the look-back chain), i.e. while the rest of that launch and the weights launch are
// still running, and again by the finish job (the paths that do not end in the fused level kernel).)
using naqs::publish_info;

// inclusive prefix sum over the 64 lanes of a wave with DPP moves: shifts by 1, 2, 4, 8 within the rows of 16 lanes, then lane 15
// of rows 0 / 2 to rows 1 / 3 and lane 31 to the upper half (row_bcast) — seven VALU instructions where six __shfl_up rounds go
// through the LDS crossbar (ds_bpermute: ~100 cycles each on the chain that ends every tree level; round 5: 2.2-3.3 k cycles of
// compaction per level in tools/head_clock_probe.py).  Integers: the same sums in any order.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
#define NAQS_SCAN_STEP(CTRL, ROW_MASK) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false)
    NAQS_SCAN_STEP(0x111, 0xF);      // row_shr:1 (lanes without a source add the `old` operand, 0)
    NAQS_SCAN_STEP(0x112, 0xF);      // row_shr:2
    NAQS_SCAN_STEP(0x114, 0xF);      // row_shr:4
    NAQS_SCAN_STEP(0x118, 0xF);      // row_shr:8
    NAQS_SCAN_STEP(0x142, 0xA);      // row_bcast:15 -> rows 1 and 3
    NAQS_SCAN_STEP(0x143, 0xC);      // row_bcast:31 -> rows 2 and 3
#undef NAQS_SCAN_STEP
    return v;
}

// Where lane c of a quad writes the prefix's child c: all four lanes hold the children counts `out` (split_quad) and the quad's
// first lane holds the position `first` of the prefix's first surviving child (exclusive scan of the survivors); the children
// stay in (prefix, outcome) order.  live: child c survived.  One store round per level instead of the first lane's loop.
struct ChildSlot {
    bool live; int pos; int64_t cnt;
    __device__ __forceinline__ float p(const float (&pp)[4]) const { const int c = threadIdx.x & 3; return c == 0 ? pp[0] : (c == 1 ? pp[1] : (c == 2 ? pp[2] : pp[3])); }
};
struct ChildSlot64 {
    bool live; int64_t pos; int64_t cnt;
    __device__ __forceinline__ float p(const float (&pp)[4]) const { const int c = threadIdx.x & 3; return c == 0 ? pp[0] : (c == 1 ? pp[1] : (c == 2 ? pp[2] : pp[3])); }
};
__device__ __forceinline__ ChildSlot child_slot(const int64_t (&out)[4], const int first, const int c) {
    const int f = naqs::dpp_quad<0x00>(first);            // the first lane's value to the whole quad
    const int rank = (c > 0 && out[0] > 0 ? 1 : 0) + (c > 1 && out[1] > 0 ? 1 : 0) + (c > 2 && out[2] > 0 ? 1 : 0);
    const int64_t mine = c == 0 ? out[0] : (c == 1 ? out[1] : (c == 2 ? out[2] : out[3]));
    return ChildSlot{mine > 0, f + rank, mine};
}
__device__ __forceinline__ ChildSlot64 child_slot64(const int64_t (&out)[4], const int64_t first, const int c) {
    const int lo = naqs::dpp_quad<0x00>((int)(uint32_t)first), hi = naqs::dpp_quad<0x00>((int)(uint32_t)((uint64_t)first >> 32));
    const int64_t f = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    const int rank = (c > 0 && out[0] > 0 ? 1 : 0) + (c > 1 && out[1] > 0 ? 1 : 0) + (c > 2 && out[2] > 0 ? 1 : 0);
    const int64_t mine = c == 0 ? out[0] : (c == 1 ? out[1] : (c == 2 ? out[2] : out[3]));
    return ChildSlot64{mine > 0, f + rank, mine};
}

// quad (4 consecutive lanes) sum: every lane of the quad gets the total
__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(naqs::dpp_quad<0xB1>(__float_as_int(v)));      // quad_perm [1, 0, 3, 2]: lane ^ 1
    v += __int_as_float(naqs::dpp_quad<0x4E>(__float_as_int(v)));      // quad_perm [2, 3, 0, 1]: lane ^ 2
    return v;
}

using naqs::binomial_group;

// raw outputs of block n for one prefix -> the float32 conditional probabilities p[c] = exp(log-amp)^2 (nade.py:673) and
// the electron-budget mask of the four outcomes (nade.py:695)
__device__ __forceinline__ void probs_from_outputs(const NetDims &d, const int n, const float (&t)[5], const uint32_t abits,
                                                   const uint32_t bbits, float (&p)[4], bool (&phys)[4]) {
    float la[4];
    bool ok[4];
    naqs::amp_conditional(d, n, t, abits, bbits, la, ok);
    naqs::amp_budget_mask(d, n, abits, bbits, phys);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float e = ok[c] ? expf(la[c]) : 0.0f;
        p[c] = e * e;
    }
}

__device__ __forceinline__ void split_quad(const int n, const uint32_t ab, const int64_t cnt, const uint32_t k0, const uint32_t k1,
                                           const float (&p)[4], const bool (&phys)[4], int64_t (&out)[4], long long *clk);

// Matrix-core mode: a block's MLP depends on the prefix bits only, not on the counts, so the kernels that run several levels
// per launch let their idle waves evaluate every possible descendant ("candidate") of the launch's entry prefixes while
// wave 0 is in the binomial rounds of a level that fits it; the later levels then start from the probabilities.  Same
// item, same arithmetic as expand_quad's: the same bits.
//
// The schedule: items of 16 candidates (one matrix-core item of one wave), level n + 1 first; item j runs in pass
// min(its level - 1, j / (n_waves - 1)) on wave 1 + j % (n_waves - 1) — at most one item per wave and pass while the list is
// short enough (five items in three levels from four entries; six in the four-level head), same-level items of a wave
// sharing one fetch of the pair's fragments.  Candidate ci of level n + lj is entry ci >> 2 lj followed by the outcomes
// (ci >> 2 (lj - 1)) & 3, ..., ci & 3; its probabilities go to cp[(lj - 1) * 64 + ci] (float4) and its physical mask to
// cphys[...] (bit c) — at most 64 candidates per level by construction (E * 4^lj <= 64).
template <int CT>
__device__ __forceinline__ void cand_jobs_t(const NetDims &d, const naqs::ushort_t *__restrict__ wamp, const int n, const int LV,
                                            const int E, const int entries, const uint32_t *s_entry, float *outs, f32x4 *cp,
                                            uint8_t *cphys, const int wave, const int n_waves, const int pass) {
    const int lane = threadIdx.x & 63;
    naqs::AmpFrag<CT> f;
    int have = -1, job = 0;
    for (int lj = 1; lj < LV; ++lj) {
        const int items = ((E << (2 * lj)) + 15) >> 4;            // (the last item of a short level carries unused candidates)
        for (int it = 0; it < items; ++it, ++job) {
            if (min(lj - 1, job / (n_waves - 1)) != pass || job % (n_waves - 1) != wave - 1) continue;
            if (((it * 16) >> (2 * lj)) >= entries) continue;     // all 16 candidates descend from absent entries
            const int ci = it * 16 + (lane & 15);
            const int e = ci >> (2 * lj);
            uint32_t ab16 = e < entries ? s_entry[e] : 0u;
            for (int k = 0; k < lj; ++k) {
                const uint32_t c = (uint32_t)(ci >> (2 * (lj - 1 - k))) & 3u;
                ab16 |= ((c & 1u) << (n + k)) | ((c >> 1) << (16 + n + k));
            }
            const int lev = n + lj;
            if (have != lev) { naqs::amp_mfma_load<CT>(wamp + (size_t)lev * naqs::amp_mfma_pair_elems(CT * 16), lane, f); have = lev; }
            naqs::amp_mfma_item<CT>(d, f, lev, ab16, lane, outs);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 16) {
                float t[5], p[4];
                bool phys[4];
#pragma unroll
                for (int c = 0; c < 5; ++c) t[c] = outs[lane * 8 + c];
                probs_from_outputs(d, lev, t, ab16 & 0xffffu, ab16 >> 16, p, phys);
                cp[(lj - 1) * 64 + ci] = (f32x4){p[0], p[1], p[2], p[3]};
                cphys[(lj - 1) * 64 + ci] = (uint8_t)((phys[0] ? 1 : 0) | (phys[1] ? 2 : 0) | (phys[2] ? 4 : 0) | (phys[3] ? 8 : 0));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}
__device__ __forceinline__ void cand_jobs(const NetDims &d, const naqs::ushort_t *__restrict__ wamp, const int n, const int LV,
                                          const int E, const int entries, const uint32_t *s_entry, float *outs, f32x4 *cp,
                                          uint8_t *cphys, const int wave, const int n_waves, const int pass) {
    if (d.Ha == 128) cand_jobs_t<8>(d, wamp, n, LV, E, entries, s_entry, outs, cp, cphys, wave, n_waves, pass);
    else if (d.Ha == 64) cand_jobs_t<4>(d, wamp, n, LV, E, entries, s_entry, outs, cp, cphys, wave, n_waves, pass);
    else cand_jobs_t<2>(d, wamp, n, LV, E, entries, s_entry, outs, cp, cphys, wave, n_waves, pass);
}

// One prefix, one quad of lanes (q = lane & 3): the hidden units of block n are split four ways, then the quad draws the
// first-level binomial of the multinomial split together and its two halves the two independent second-level ones.  Weights of pair n are in s_w (staged and
// synchronised by the caller).  On return lane q == 0 holds the children counts (un-physical ones zeroed, nade.py:695)
// and the float32 conditional probabilities p[c] = exp(log-amp)^2 (nade.py:673).
// (expand_probs: the first half — block MLP, conditional, probabilities and physical mask; expand_quad: both halves)
__device__ __forceinline__ void expand_probs(const NetDims &d, const float *__restrict__ s_w, const int n, const uint32_t ab,
                                             float (&p)[4], bool (&phys)[4], long long *clk = nullptr,
                                             const naqs::ushort_t *__restrict__ wamp = nullptr) {
    const int q = threadIdx.x & 3;
    const int nin = n == 0 ? 1 : 2 * n;
    const int S = (nin + 1 + 5 + 3) & ~3;
    const uint32_t abits = ab & 0xffffu, bbits = ab >> 16;
    const bool swap = d.sym && abits > bbits;                                  // nade.py:519-530
    const uint32_t first = swap ? bbits : abits, second = swap ? abits : bbits;
    const int per = (d.Ha + 3) / 4;
    const int j0 = min(d.Ha, q * per), j1 = min(d.Ha, j0 + per);
    float t[5];
    if (wamp != nullptr) {
        // the block's MLP on the matrix cores: a wave's 64 lanes are 16 prefixes = one (tile, pair) item of the log-psi
        // kernel's prologue (naqs_amp_mfma.hpp); s_w is then per-wave scratch (the
        // raw outputs), not the pair's weights.  ~2.5 k cycles instead of the 5-8 k of four lanes walking 16 hidden units
        // each through LDS reads nothing hides (one wave per SIMD).
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float *outs = const_cast<float *>(s_w) + (size_t)wave * 128;
        const uint32_t ab16 = (uint32_t)__shfl((int)ab, 4 * (lane & 15), 64);
        if (d.Ha == 128) {
            naqs::AmpFrag<8> f;
            naqs::amp_mfma_load<8>(wamp + (size_t)n * naqs::amp_mfma_pair_elems(128), lane, f);
            naqs::amp_mfma_item<8>(d, f, n, ab16, lane, outs);
        } else if (d.Ha == 64) {
            naqs::AmpFrag<4> f;
            naqs::amp_mfma_load<4>(wamp + (size_t)n * naqs::amp_mfma_pair_elems(64), lane, f);
            naqs::amp_mfma_item<4>(d, f, n, ab16, lane, outs);
        } else {
            naqs::AmpFrag<2> f;
            naqs::amp_mfma_load<2>(wamp + (size_t)n * naqs::amp_mfma_pair_elems(32), lane, f);
            naqs::amp_mfma_item<2>(d, f, n, ab16, lane, outs);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = 0; c < 5; ++c) t[c] = outs[(lane >> 2) * 8 + c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) clk[0] = clock64();
    } else {
        float o[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        switch (n) {
#define CASE(NB) case NB: naqs::amp_partial<NB>(d, s_w, first, second, j0, j1, o); break;
            CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7)
            CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
#undef CASE
            default: break;
        }
        if (clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) clk[0] = clock64();     // MLP partial sums done
        const float *b2 = s_w + d.Ha * S;
#pragma unroll
        for (int c = 0; c < 5; ++c) t[c] = (c < d.n_out_amp ? b2[c] : 0.0f) + quad_sum(o[c]);
    }
    probs_from_outputs(d, n, t, abits, bbits, p, phys);
    if (clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) clk[1] = clock64();     // conditional + probabilities done
}
__device__ __forceinline__ void expand_quad(const NetDims &d, const float *__restrict__ s_w, const int n, const uint32_t ab,
                                            const int64_t cnt, const uint32_t k0, const uint32_t k1, int64_t (&out)[4],
                                            float (&p)[4], long long *clk = nullptr,
                                            const naqs::ushort_t *__restrict__ wamp = nullptr) {
    bool phys[4];
    expand_probs(d, s_w, n, ab, p, phys, clk, wamp);
    split_quad(n, ab, cnt, k0, k1, p, phys, out, clk);
}


// the second half of expand_quad: the count's multinomial split given the conditional probabilities and the physical mask
__device__ __forceinline__ void split_quad(const int n, const uint32_t ab, const int64_t cnt, const uint32_t k0, const uint32_t k1,
                                           const float (&p)[4], const bool (&phys)[4], int64_t (&out)[4], long long *clk) {
    const int q = threadIdx.x & 3;
    // multinomial(count; p) as a binary tree of binomials (same distribution as the reference's conditional chain,
    // nade.py:31-35, two dependent rounds instead of three; the float64 renormalisation of :682-683 cancels in the
    // ratios): first {2,3} against {0,1}, then 1 within {0,1} and 3 within {2,3}
    const double p01 = (double)p[0] + (double)p[1], p23 = (double)p[2] + (double)p[3], tot = p01 + p23;
    // first split by the whole quad (four attempts per round), then lanes {0,1} draw outcome 1 within {0,1} and lanes
    // {2,3} outcome 3 within {2,3} (two attempts per round each); every lane of a group ends up with the group's variate
    const int64_t n23 = binomial_group<4>(tot > 0.0, cnt, fmin(1.0, p23 / tot), k0, k1, ab, (uint32_t)n | (1u << 8));
    if (clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) clk[2] = clock64();     // first binomial done
    const int64_t n01 = tot > 0.0 ? cnt - n23 : 0;
    const int sub = q >> 1;
    const int64_t m_sub = sub == 0 ? n01 : n23;
    const double den = sub == 0 ? p01 : p23, num = sub == 0 ? (double)p[1] : (double)p[3];
    const int64_t hi_sub = binomial_group<2>(den > 0.0, m_sub, fmin(1.0, num / den), k0, k1, ab,
                                             (uint32_t)n | ((uint32_t)(2 + sub) << 8));
    const int64_t hi = naqs::group_bcast<4>(hi_sub, 0);
    const int64_t n3 = naqs::group_bcast<4>(hi_sub, 2);
    out[0] = n01 - hi; out[1] = hi; out[2] = n23 - n3; out[3] = n3;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (!phys[c]) out[c] = 0;
}

__device__ __forceinline__ void stage_pair_weights(const NetDims &d, const float *__restrict__ w, const int n, float *s_w,
                                                   const int nthreads) {
    const int nin = n == 0 ? 1 : 2 * n;
    const int total = d.Ha * ((nin + 1 + 5 + 3) & ~3) + 8;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(w + d.amp_off[n]);
    f32x4 *dst = reinterpret_cast<f32x4 *>(s_w);
    for (int e = threadIdx.x; e < total / 4; e += nthreads) dst[e] = src[e];
}

__global__ __launch_bounds__(SB) void sample_expand_kernel(const NetDims d, const float *__restrict__ w, const int n,
                                                           const SampleBufs b, const int cur, const uint32_t k0,
                                                           const uint32_t k1, const naqs::ushort_t *__restrict__ wamp) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    __shared__ uint32_t s_red[SB / WAVE];
    const int64_t U = b.U[n];
    if (b.U[MAXP + 1] != 0 || (int64_t)blockIdx.x * EXP_PARENTS >= U) return;          // workgroup-uniform
    if (wamp == nullptr) stage_pair_weights(d, w, n, s_w, SB);
    const int64_t u = (int64_t)blockIdx.x * EXP_PARENTS + (threadIdx.x >> 2);
    const bool active = u < U;
    const uint32_t ab = active ? b.ab[cur][u] : 0u;
    const int64_t cnt = active ? b.cnt[cur][u] : 0;
    __syncthreads();
    int64_t out[4];
    float p[4];
    expand_quad(d, s_w, n, ab, cnt, k0, k1, out, p, nullptr, wamp);
    uint32_t survivors = 0;
    if (active && (threadIdx.x & 3) == 0) {
        const float pr = b.prob[cur][u];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            b.child_cnt[u * 4 + c] = out[c];
            b.child_prob[u * 4 + c] = pr * p[c];
            survivors += out[c] > 0 ? 1u : 0u;
        }
    }
    // survivors of this workgroup
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) survivors += __shfl_down(survivors, off, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = survivors;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot_s = 0;
        for (int i = 0; i < SB / WAVE; ++i) tot_s += s_red[i];
        b.wg_total[blockIdx.x] = tot_s;
    }
}

// One tree level in ONE launch (expand + compaction): as sample_expand_kernel, then instead of leaving the children
// counts for sample_scatter_kernel the workgroup finds its place in the next level itself.  Its survivors count goes
// out as one 8-byte word (tag << 32 | count, a relaxed agent-scope store: flag and value travel in the same naturally
// aligned granule, so no fence is needed) and the sum over the preceding workgroups comes from polling their words
// (relaxed agent-scope loads, which bypass this CU's L1) — "decoupled look-back" over the aggregates.  The tag is
// (sampling call, level), so the words are never cleared.  Forward progress: a workgroup only waits for workgroups
// with LOWER indices, which the dispatcher starts no later than itself.  Children are written straight to the other
// half of the ping-pong arrays (or, on the last level, to the caller's outputs) in (prefix, outcome) order: the same
// positions sample_scatter_kernel assigns, bit-identical results.
__global__ __launch_bounds__(SB) void sample_level_kernel(const NetDims d, const float *__restrict__ w, const int n,
                                                          const SampleBufs b, const int cur, const uint32_t k0,
                                                          const uint32_t k1, const uint32_t tag, const int64_t cap,
                                                          const int last, uint64_t *__restrict__ keys_out,
                                                          int64_t *__restrict__ counts_out, float *__restrict__ probs_out,
                                                          long long *__restrict__ clk, const naqs::ushort_t *__restrict__ wamp,
                                                          int64_t *__restrict__ early, const int64_t seq) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    __shared__ uint32_t s_wave[SB / WAVE];
    __shared__ long long s_base;
#define SMARK(k) do { if (clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) clk[n * 12 + (k)] = clock64(); } while (0)
    SMARK(0);
    const int64_t U = b.U[n];
    if (b.U[MAXP + 1] != 0 || (int64_t)blockIdx.x * EXP_PARENTS >= U) return;          // workgroup-uniform
    const int64_t nwg = (U + EXP_PARENTS - 1) / EXP_PARENTS;
    SMARK(1);
    if (wamp == nullptr) stage_pair_weights(d, w, n, s_w, SB);
    const int64_t u = (int64_t)blockIdx.x * EXP_PARENTS + (threadIdx.x >> 2);
    const bool active = u < U;
    const uint32_t ab = active ? b.ab[cur][u] : 0u;
    const int64_t cnt = active ? b.cnt[cur][u] : 0;
    const float pr = active ? b.prob[cur][u] : 0.0f;
    __syncthreads();
    SMARK(2);
    int64_t out[4];
    float p[4];
    expand_quad(d, s_w, n, ab, cnt, k0, k1, out, p, clk ? clk + n * 12 + 6 : nullptr, wamp);
    SMARK(3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool owner = active && (threadIdx.x & 3) == 0;
    uint32_t mine = 0;
    if (owner)
#pragma unroll
        for (int c = 0; c < 4; ++c) mine += out[c] > 0 ? 1u : 0u;
    // inclusive scan of `mine` over the workgroup (only quad owners contribute)
    const uint32_t incl = wave_inclusive_scan(mine);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int i = 0; i < SB / WAVE; ++i) { if (i < wave) before += s_wave[i]; total += s_wave[i]; }
    if (wave == 0) {
        if (lane == 0 && !(naqs::poll_drop(b.ctl, naqs::POLL_SAMPLE_LOOKBACK) && blockIdx.x == 0))
            __hip_atomic_store(&b.wg_state[blockIdx.x], ((unsigned long long)tag << 32) | total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // look back: lanes poll the words of the preceding workgroups, 64 at a time (bounded: naqs_poll.hpp)
        long long part = 0;
        bool ok = true;
        for (int64_t j0 = 0; j0 < (int64_t)blockIdx.x; j0 += WAVE) {
            const int64_t j = j0 + lane;
            if (j < (int64_t)blockIdx.x) {
                unsigned long long v;
                ok = naqs::poll_tagged<1>(&b.wg_state[j], tag, v, b.ctl, naqs::POLL_SAMPLE_LOOKBACK, (uint32_t)j) && ok;
                part += (long long)(uint32_t)v;
            }
        }
        const bool all_ok = __all(ok);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_down(part, off, 64);
        if (lane == 0) s_base = all_ok ? part : -1;
    }
    __syncthreads();
    SMARK(4);
    const int64_t base = s_base;
    if (base < 0) {                                        // a look-back wait ran out: no children are written, the rest of the call
        if (threadIdx.x == 0) b.U[MAXP + 1] = 1;           // falls through (overflow flag) and the host reports NAQS_ERR_HIP
        return;
    }
    int64_t pos = base + before + (incl - mine);
    if (owner && mine) {
        const int nxt = cur ^ 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (out[c] > 0) {
                if (pos < cap) {
                    const uint32_t child = ab | ((uint32_t)(c & 1) << n) | ((uint32_t)(c >> 1) << (16 + n));
                    const float prc = pr * p[c];
                    if (last) {
                        uint64_t key = 0;
                        for (int k = 0; k < d.P; ++k) {
                            key |= (uint64_t)((child >> k) & 1u) << d.qa[k];
                            key |= (uint64_t)((child >> (16 + k)) & 1u) << d.qb[k];
                        }
                        keys_out[pos] = key;
                        counts_out[pos] = out[c];
                        if (probs_out) probs_out[pos] = prc;
                    } else {
                        b.ab[nxt][pos] = child;
                        b.cnt[nxt][pos] = out[c];
                        b.prob[nxt][pos] = prc;
                    }
                }
                ++pos;
            }
        }
    }
    if ((int64_t)blockIdx.x == nwg - 1 && threadIdx.x == 0) {
        const int64_t all = base + total;
        b.U[n + 1] = all < cap ? all : cap;
        if (all > cap) b.U[MAXP + 1] = 1;
        if (last && early != nullptr) publish_info(early, all > cap ? 0 : all, all > cap ? 1 : 0, seq);
    }
    SMARK(5);
#undef SMARK
}

// NL (2 or 3) tree levels in ONE launch.  A level kernel lasts ~16 us whatever its size — ~5 us of launch and drain around a
// ~10 us chain (prefix load, block MLP, two dependent binomial rounds, look-back, scatter) that only a handful of CUs take
// part in — so a workgroup here enters with E = 64 / 4^(NL-1) prefixes of level n, expands them, compacts the children in
// LDS (sample_head_kernel's way; at most 4 E, 16 E, ... <= 64 of them) and goes on to the next level itself; only the last
// level of the launch is compacted across workgroups (sample_level_kernel's look-back) and written to global memory.
// Children stay in (prefix, outcome) order at every step and a draw is keyed by its prefix, so the samples are the ones
// every other cut of the tree into launches produces (tests/test_sampler_gpu.py).
// The look-back word of a workgroup carries, beside the survivors of the launch's last level (9 bits), its prefix counts
// at the intermediate levels (7 bits each, up to three of them): the workgroup that closes the chain then knows those levels' total sizes too
// (U[n + 1], ...; more than `cap` prefixes alive at ANY level is the overflow the reference raises, nade.py:710-712).
// Every workgroup adds up ALL its predecessors' words (up to 256 loads in flight per round of its first wave): fine for
// the ~10^2..10^3 active workgroups this kernel is chosen for (the host picks NL from the previous call's level sizes,
// net_sample_impl), slow but correct beyond.
template <int NL>
__global__ __launch_bounds__(SB) void sample_multi_kernel(const NetDims d, const float *__restrict__ w, const int n,
                                                          const SampleBufs b, const int cur, const uint32_t k0,
                                                          const uint32_t k1, const uint32_t tag, const int64_t cap,
                                                          const int last, uint64_t *__restrict__ keys_out,
                                                          int64_t *__restrict__ counts_out, float *__restrict__ probs_out,
                                                          const naqs::ushort_t *__restrict__ wamp, int64_t *__restrict__ early,
                                                          const int64_t seq) {
    static_assert(NL >= 2 && NL <= 4, "two to four levels per launch");
    constexpr int E = 64 >> (2 * (NL - 1));
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    __shared__ uint32_t s_ab[2][64];
    __shared__ int64_t s_cnt[2][64];
    __shared__ float s_prob[2][64];
    __shared__ uint32_t s_wave[SB / WAVE];
    __shared__ long long s_base[4];
    // matrix-core mode: probabilities of every possible descendant of the entry prefixes (cand_jobs), the candidate index
    // of each live prefix, the entry prefixes' strings
    __shared__ __attribute__((aligned(16))) f32x4 s_cp[(NL - 1) * 64];
    __shared__ uint8_t s_cphys[(NL - 1) * 64];
    __shared__ uint8_t s_cand[2][64];
    __shared__ uint32_t s_entry[E];
    const bool ahead = wamp != nullptr;
    const int64_t U = b.U[n];
    if (b.U[MAXP + 1] != 0 || (int64_t)blockIdx.x * E >= U) return;                    // workgroup-uniform
    const int64_t nwg = (U + E - 1) / E;
    const int tid = threadIdx.x, u = tid >> 2, lane = tid & 63, wave = tid >> 6;
    const int entries = (int)min((int64_t)E, U - (int64_t)blockIdx.x * E);
    int U_loc = entries;
#if defined(NAQS_HEAD_CLOCKS)
    const long long mm_t0 = clock64();
    const int mm_wg = blockIdx.x == 0 ? 0 : ((int64_t)blockIdx.x == nwg - 1 ? 1 : -1);
    if (threadIdx.x == 0 && mm_wg >= 0) { g_multi_clk[last ? 1 : 0][mm_wg][5][0] += 1; g_multi_clk[last ? 1 : 0][mm_wg][5][1] += NL; g_multi_clk[last ? 1 : 0][mm_wg][5][2] += nwg; }
#endif
    uint32_t mid_a = 0u, mid_b = 0u, mid_c = 0u;           // this workgroup's prefixes at the launch's 2nd (3rd, 4th) level
    // (not unrolled, and ONE call site of the binomial draws per level: the level body with its float64 generator inlined is
    // ~5 k instructions; NL unrolled copies with two call sites each were 190 KB of straight-line code for NL = 4, and no
    // instruction cache holds that — every level of every workgroup streamed its code from L2.  N2 step -6 us, same samples)
#pragma unroll 1
    for (int li = 0; li < NL; ++li) {
        const int lev = n + li;
        const bool active = u < U_loc;
        uint32_t ab = 0u;
        int64_t cnt = 0;
        float pr = 0.0f;
        int cand = u;                                      // candidate index of this quad's prefix at its level
        if (li == 0) {
            const int64_t gu = (int64_t)blockIdx.x * E + u;
            if (active) { ab = b.ab[cur][gu]; cnt = b.cnt[cur][gu]; pr = b.prob[cur][gu]; }
            if (ahead && active && (tid & 3) == 0) s_entry[u] = ab;
        } else {
            __syncthreads();                               // the previous level's children are in LDS
            if (active) { ab = s_ab[li & 1][u]; cnt = s_cnt[li & 1][u]; pr = s_prob[li & 1][u]; cand = s_cand[li & 1][u]; }
        }
        if (wamp == nullptr) {
            __syncthreads();                               // everyone done with the previous pair's rows
            stage_pair_weights(d, w, lev, s_w, SB);
        }
        __syncthreads();
        MULTI_MARK(li, 0);
        int64_t out[4] = {0, 0, 0, 0};
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        bool phys[4] = {false, false, false, false};
        bool draws = true;                                 // (wave-uniform)
        if (ahead && wave > 0 && (E << (2 * li)) <= 16) {
            // this level's prefixes fill at most wave 0: the other waves evaluate the later levels' candidates meanwhile
            cand_jobs(d, wamp, n, NL, E, entries, s_entry, s_w + (size_t)wave * 128, s_cp, s_cphys, wave, SB / WAVE, li);
            draws = false;
        } else if (ahead && li > 0) {
            if (active) {
                const f32x4 pv = s_cp[(li - 1) * 64 + cand];
                const uint32_t pm = s_cphys[(li - 1) * 64 + cand];
#pragma unroll
                for (int c = 0; c < 4; ++c) { p[c] = pv[c]; phys[c] = ((pm >> c) & 1u) != 0u; }
            }
        } else {
            expand_probs(d, s_w, lev, ab, p, phys, nullptr, wamp);
        }
        MULTI_MARK(li, 1);
        if (draws) split_quad(lev, ab, cnt, k0, k1, p, phys, out, nullptr);
        MULTI_MARK(li, 2);
        const bool owner = active && (tid & 3) == 0;
        uint32_t mine = 0;
        if (owner)
#pragma unroll
            for (int c = 0; c < 4; ++c) mine += out[c] > 0 ? 1u : 0u;
        const uint32_t incl = wave_inclusive_scan(mine);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        MULTI_MARK(li, 3);
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < SB / WAVE; ++i) { if (i < wave) before += s_wave[i]; total += s_wave[i]; }
        if (li + 1 < NL) {
            // next level of this launch: children -> LDS, in (prefix, outcome) order
            {   // (every lane of the quad writes its own child: child_slot)
                const int c = tid & 3;
                const ChildSlot cs = child_slot(out, (int)(before + (incl - mine)), c);
                if (active && cs.live) {
                    const int pos = cs.pos;
                    s_ab[(li + 1) & 1][pos] = ab | ((uint32_t)(c & 1) << lev) | ((uint32_t)(c >> 1) << (16 + lev));
                    s_cnt[(li + 1) & 1][pos] = cs.cnt;
                    s_prob[(li + 1) & 1][pos] = pr * cs.p(p);
                    s_cand[(li + 1) & 1][pos] = (uint8_t)(cand * 4 + c);
                }
            }
            U_loc = (int)total;
            if (li == 0) mid_a = total; else if (li == 1) mid_b = total; else mid_c = total;
            continue;
        }
        // last level of the launch: place among the other workgroups' children (look-back over their words)
        if (wave == 0) {
            if (lane == 0 && !(naqs::poll_drop(b.ctl, naqs::POLL_SAMPLE_LOOKBACK_MULTI) && blockIdx.x == 0))
                __hip_atomic_store(&b.wg_state[blockIdx.x],
                                   ((unsigned long long)tag << 32) | (unsigned long long)(total | (mid_a << 9) | (mid_b << 16) | (mid_c << 23)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long part = 0, part_a = 0, part_b = 0, part_c = 0;
            bool ok = true;
            for (int64_t j0 = 0; j0 < (int64_t)blockIdx.x; j0 += 4 * WAVE) {
                unsigned long long v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t j = j0 + q * WAVE + lane;
                    v[q] = j < (int64_t)blockIdx.x ? __hip_atomic_load(&b.wg_state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                  : ((unsigned long long)tag << 32);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t j = j0 + q * WAVE + lane;
                    if ((uint32_t)(v[q] >> 32) != tag)         // not there yet: poll (bounded, naqs_poll.hpp)
                        ok = naqs::poll_tagged<1>(&b.wg_state[j], tag, v[q], b.ctl, naqs::POLL_SAMPLE_LOOKBACK_MULTI, (uint32_t)j) && ok;
                    const uint32_t x = (uint32_t)v[q];
                    part += (long long)(x & 0x1ffu);
                    part_a += (long long)((x >> 9) & 0x7fu);
                    part_b += (long long)((x >> 16) & 0x7fu);
                    part_c += (long long)((x >> 23) & 0x7fu);
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                part += __shfl_down(part, off, 64);
                part_a += __shfl_down(part_a, off, 64);
                part_b += __shfl_down(part_b, off, 64);
                part_c += __shfl_down(part_c, off, 64);
            }
            const bool all_ok = __all(ok);
            if (lane == 0) { s_base[0] = all_ok ? part : -1; s_base[1] = part_a; s_base[2] = part_b; s_base[3] = part_c; }
        }
        __syncthreads();
        MULTI_MARK(4, 0);
        const int64_t base = s_base[0];
        if (base < 0) {                                    // a look-back wait ran out (see sample_level_kernel)
            if (tid == 0) b.U[MAXP + 1] = 1;
            return;
        }
        {
            // the OTHER half of the global ping-pong arrays, whatever NL: this launch's workgroups read their entry prefixes
            // from half `cur` at their own pace (a workgroup may start after others have finished), so nothing may be written
            // there — with `(cur + NL) & 1` a two-level launch wrote its output over its own input
            // (every lane of the quad writes its own child — child_slot — and forms its own key: these stores are the tail of
            // the whole sampler call, 5-9 k cycles as one lane's loop over four children with the bit scatter inside)
            const int nxt = cur ^ 1;
            const int c = tid & 3;
            const ChildSlot64 cs = child_slot64(out, base + before + (incl - mine), c);
            if (active && cs.live && cs.pos < cap) {
                const int64_t pos = cs.pos;
                const uint32_t child = ab | ((uint32_t)(c & 1) << lev) | ((uint32_t)(c >> 1) << (16 + lev));
                const float prc = pr * cs.p(p);
                if (last) {
                    uint64_t key = 0;
                    for (int k = 0; k < d.P; ++k) {
                        key |= (uint64_t)((child >> k) & 1u) << d.qa[k];
                        key |= (uint64_t)((child >> (16 + k)) & 1u) << d.qb[k];
                    }
                    keys_out[pos] = key;
                    counts_out[pos] = cs.cnt;
                    if (probs_out) probs_out[pos] = prc;
                } else {
                    b.ab[nxt][pos] = child;
                    b.cnt[nxt][pos] = cs.cnt;
                    b.prob[nxt][pos] = prc;
                }
            }
        }
        if ((int64_t)blockIdx.x == nwg - 1 && tid == 0) {
            const int64_t all = base + total, all_a = s_base[1] + mid_a, all_b = NL > 2 ? s_base[2] + mid_b : 0;
            const int64_t all_c = NL > 3 ? s_base[3] + mid_c : 0;
            b.U[n + 1] = all_a < cap ? all_a : cap;
            if (NL > 2) b.U[n + 2] = all_b < cap ? all_b : cap;
            if (NL > 3) b.U[n + 3] = all_c < cap ? all_c : cap;
            b.U[n + NL] = all < cap ? all : cap;
            const bool over = all > cap || all_a > cap || all_b > cap || all_c > cap;
            if (over) b.U[MAXP + 1] = 1;
            if (last && early != nullptr) publish_info(early, over ? 0 : all, over ? 1 : 0, seq);
        }
        MULTI_MARK(4, 1);
    }
}

// The first HL levels of the tree in ONE launch: level n has at most 4^n <= HT / 4 prefixes there, so a single
// workgroup of HT threads (a quad of lanes per prefix) expands, compacts in LDS and moves on — these levels are pure
// latency (~30 us each as separate expand + scatter launches whatever their size).  Leaves the level-HL prefixes
// (<= HT) in the global ping-pong arrays where the per-level kernels continue.  HT = 1024 keeps five levels in the
// launch, HT = 256 four (small max_unique).
// Workgroups 1 .. pk.n_wgs of the 256-thread form (naqs_vmc_step): this launch is one workgroup for ~30 us that reads nothing but
// the amplitude blocks, so the phase layers' share of the last update's re-pack — weight maxima, f16x2 planes, the backward's
// row-major copies: 13 us as launches of their own between the update and this call — runs beside it (naqs_pack.hpp).
template <int HT, int HL>
__global__ __launch_bounds__(HT) void sample_head_kernel(const NetDims d, const float *__restrict__ w, const SampleBufs b,
                                                         const int64_t n_samples, const uint32_t k0, const uint32_t k1,
                                                         const naqs::ushort_t *__restrict__ wamp, const naqs::PackPhaseArgs pk) {
    if constexpr (HT == 256) {
        if (blockIdx.x > 0) {                              // (workgroup-uniform)
            naqs::pack_phase_dispatch(d, pk, (int)blockIdx.x - 1);
            return;
        }
    }
    constexpr int HP = HT / 4;                             // prefixes the workgroup can hold
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    __shared__ uint32_t s_ab[2][HP];
    __shared__ int64_t s_cnt[2][HP];
    __shared__ float s_prob[2][HP];
    __shared__ uint32_t s_wave[HT / WAVE];
    // matrix-core mode, four-level head: while wave 0 expands the root, the other waves evaluate every possible prefix of
    // levels 1-3 (4 + 16 + 64 candidates: cand_jobs with the root as the one entry); those levels start from the probabilities
    constexpr bool AHEAD_OK = HT == 256 && HL == 4;
    __shared__ __attribute__((aligned(16))) f32x4 s_cp[AHEAD_OK ? 3 * 64 : 1];
    __shared__ uint8_t s_cphys[AHEAD_OK ? 3 * 64 : 1];
    __shared__ uint8_t s_cand[2][AHEAD_OK ? HP : 1];
    __shared__ uint32_t s_entry[1];
    const bool ahead = AHEAD_OK && wamp != nullptr;
    const int tid = threadIdx.x, u = tid >> 2, q = tid & 3, lane = tid & 63, wave = tid >> 6;
#if defined(NAQS_HEAD_CLOCKS)
    const long long head_t0 = clock64();
#endif
    if (tid == 0) {
        for (int i = 0; i < MAXP + 2; ++i) b.U[i] = 0;
        b.U[0] = 1;
        s_ab[0][0] = 0u; s_cnt[0][0] = n_samples; s_prob[0][0] = 1.0f;
        s_entry[0] = 0u;
        if (AHEAD_OK) s_cand[0][0] = 0;
    }
    int U = 1;
#if defined(NAQS_HEAD_CLOCKS)
    if (threadIdx.x == 0) g_head_clk[7][0] += 1;
#endif
#pragma unroll 1
    for (int n = 0; n < HL; ++n) {                         // (not unrolled, one call site of the draws: see sample_multi_kernel)
        const int cur = n & 1, nxt = cur ^ 1;
        __syncthreads();                                   // previous level's LDS writes / everyone done with s_w
        HEAD_MARK(n, 0);
        if (wamp == nullptr) stage_pair_weights(d, w, n, s_w, HT);
        const bool active = u < U;
        const uint32_t ab = active ? s_ab[cur][u] : 0u;
        const int64_t cnt = active ? s_cnt[cur][u] : 0;
        const float pr = active ? s_prob[cur][u] : 0.0f;
        const int cand = (ahead && active) ? s_cand[cur][u] : 0;
        __syncthreads();
        int64_t out[4] = {0, 0, 0, 0};
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        bool phys[4] = {false, false, false, false};
        bool draws = true;                                 // (wave-uniform)
        if (ahead && wave > 0 && n < 3) {                  // (levels 0-2 have at most 16 prefixes: wave 0)
            cand_jobs(d, wamp, 0, HL, 1, 1, s_entry, s_w + (size_t)wave * 128, s_cp, s_cphys, wave, HT / WAVE, n);
            draws = false;
        } else if (ahead && n > 0) {
            if (active) {
                const f32x4 pv = s_cp[(n - 1) * 64 + cand];
                const uint32_t pm = s_cphys[(n - 1) * 64 + cand];
#pragma unroll
                for (int c = 0; c < 4; ++c) { p[c] = pv[c]; phys[c] = ((pm >> c) & 1u) != 0u; }
            }
        } else {
            expand_probs(d, s_w, n, ab, p, phys, nullptr, wamp);
        }
        HEAD_MARK(n, 1);
        if (draws) split_quad(n, ab, cnt, k0, k1, p, phys, out, nullptr);
        HEAD_MARK(n, 2);
        uint32_t mine = 0;                                 // survivors of this quad's prefix, held by its first lane
        if (active && q == 0)
#pragma unroll
            for (int c = 0; c < 4; ++c) mine += out[c] > 0 ? 1u : 0u;
        const uint32_t incl = wave_inclusive_scan(mine);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < HT / WAVE; ++i) { if (i < wave) before += s_wave[i]; total += s_wave[i]; }
        const bool to_global = n + 1 == HL;                // the last head level feeds the per-level kernels
        {   // every lane of the quad writes ITS child (all four hold the counts): one store round instead of a loop of four
            const ChildSlot cs = child_slot(out, (int)(before + (incl - mine)), q);
            if (active && cs.live) {
                const int c = q, pos = cs.pos;
                const uint32_t child = ab | ((uint32_t)(c & 1) << n) | ((uint32_t)(c >> 1) << (16 + n));
                if (to_global) {
                    b.ab[(n + 1) & 1][pos] = child; b.cnt[(n + 1) & 1][pos] = cs.cnt; b.prob[(n + 1) & 1][pos] = pr * cs.p(p);
                } else {
                    s_ab[nxt][pos] = child; s_cnt[nxt][pos] = cs.cnt; s_prob[nxt][pos] = pr * cs.p(p);
                    if (AHEAD_OK) s_cand[nxt][pos] = (uint8_t)(cand * 4 + c);
                }
            }
        }
        U = (int)total;
        if (tid == 0) b.U[n + 1] = total;
        HEAD_MARK(n, 3);
    }
}

__global__ __launch_bounds__(SB) void sample_scatter_kernel(const NetDims d, const int n, const SampleBufs b, const int cur,
                                                            const int64_t cap, const int last,
                                                            uint64_t *__restrict__ keys_out, int64_t *__restrict__ counts_out,
                                                            float *__restrict__ probs_out) {
    __shared__ int64_t s_off[SB / WAVE];
    __shared__ uint32_t s_wave[SB / WAVE];
    const int64_t U = b.U[n];
    if (b.U[MAXP + 1] != 0) return;
    const int64_t nwg = (U + SB - 1) / SB;
    if ((int64_t)blockIdx.x >= nwg) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // offset of this workgroup = survivors of all preceding workgroups
    int64_t part = 0;
    for (int64_t wg = threadIdx.x; wg < (int64_t)blockIdx.x * (SB / EXP_PARENTS); wg += SB) part += b.wg_total[wg];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_down(part, off, 64);
    if (lane == 0) s_off[wave] = part;
    const int64_t u = (int64_t)blockIdx.x * SB + threadIdx.x;
    const bool active = u < U;
    int64_t cc[4] = {0, 0, 0, 0};
    uint32_t mine = 0;
    if (active) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { cc[c] = b.child_cnt[u * 4 + c]; mine += cc[c] > 0 ? 1u : 0u; }
    }
    // exclusive scan of `mine` over the workgroup
    const uint32_t incl = wave_inclusive_scan(mine);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int64_t base = 0;
    for (int i = 0; i < SB / WAVE; ++i) base += s_off[i];
    uint32_t before = 0, total = 0;
    for (int i = 0; i < SB / WAVE; ++i) { if (i < wave) before += s_wave[i]; total += s_wave[i]; }
    int64_t pos = base + before + (incl - mine);
    if (active && mine) {
        const uint32_t ab = b.ab[cur][u];
        const int nxt = cur ^ 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (cc[c] > 0) {
                if (pos < cap) {
                    const uint32_t child = ab | ((uint32_t)(c & 1) << n) | ((uint32_t)(c >> 1) << (16 + n));
                    const float pr = b.child_prob[u * 4 + c];
                    if (last) {
                        uint64_t key = 0;
                        for (int k = 0; k < d.P; ++k) {
                            key |= (uint64_t)((child >> k) & 1u) << d.qa[k];
                            key |= (uint64_t)((child >> (16 + k)) & 1u) << d.qb[k];
                        }
                        keys_out[pos] = key;
                        counts_out[pos] = cc[c];
                        if (probs_out) probs_out[pos] = pr;
                    } else {
                        b.ab[nxt][pos] = child;
                        b.cnt[nxt][pos] = cc[c];
                        b.prob[nxt][pos] = pr;
                    }
                }
                ++pos;
            }
        }
    }
    if ((int64_t)blockIdx.x == nwg - 1 && threadIdx.x == 0) {
        const int64_t all = base + total;
        b.U[n + 1] = all < cap ? all : cap;
        if (all > cap) b.U[MAXP + 1] = 1;
    }
}

// M and the overflow flag for the host; optionally the samples' weights counts / sum(counts) (energy.py:993) — the
// integer total is exact whatever the summation order, so the weights are deterministic (naqs_net.hpp: sample_finish_body)
constexpr int FIN_THREADS = 1024;
__global__ __launch_bounds__(FIN_THREADS) void sample_finish_kernel(const naqs::SampleFinishJob j) {
    __shared__ int64_t s_part[FIN_THREADS / WAVE];
    naqs::sample_finish_body(j, s_part);
}

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

// mapped host words the sampler's launches write: [0..2] (M, overflow, call number) for a polling host (publish_info),
// [4 .. 4 + P] the level sizes of the last draw
int naqs::net_info_alloc(naqs_net *net) {
    if (net->h_info) return NAQS_OK;
    HIP_TRY(hipHostMalloc((void **)&net->h_info, (4 + U_SLOTS) * sizeof(int64_t), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(hipHostGetDevicePointer((void **)&net->d_info_alias, net->h_info, 0));
    HIP_TRY(hipMalloc((void **)&net->d_info2, 2 * sizeof(int64_t)));
    net->h_info[0] = net->h_info[1] = -1;
    net->h_info[2] = 0;
    for (int i = 0; i < U_SLOTS; ++i) net->h_info[4 + i] = -1;     // no draw yet
    return NAQS_OK;
}

static int net_sample_enqueue(naqs_net_t *net, int64_t n_samples, uint64_t seed, int64_t max_unique, uint64_t *keys_dev,
                              int64_t *counts_dev, float *probs_dev, double *weights_dev, int64_t *info_dev, void *stream,
                              int64_t *early, int64_t seq) {
    if (!net || n_samples < 0 || max_unique <= 0 || !keys_dev || !counts_dev || !info_dev) return NAQS_ERR_INVALID;
    if (!net->have_amp_weights) return NAQS_ERR_INVALID;
    if (n_samples > (1ll << 44) || max_unique >= (1ll << 31)) return NAQS_ERR_UNSUPPORTED;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    st = naqs::poll_check(net->poll);              // an earlier launch's device-side wait that gave up (naqs_poll.hpp)
    if (st != NAQS_OK) return st;
    const NetDims &d = net->dims;
    const int64_t cap = max_unique;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    st = naqs::net_info_alloc(net);
    if (st != NAQS_OK) return st;
    // carve the scratch: 2 x (ab, cnt, prob), children counts / probs, workgroup totals, level sizes.  The layout is a
    // function of the ALLOCATION's capacity, not of this call's: calls with a smaller cap (evaluation / solve_H next to
    // training) then find the look-back words (wg_state: "tag 0 = never written, never cleared") where the allocation
    // zeroed them, instead of over bytes that earlier calls used for something else
    const int64_t lay = std::max(cap, net->samp_cap);
    const int64_t nwg_cap = (lay + EXP_PARENTS - 1) / EXP_PARENTS;
    size_t off = 0;
    size_t o_ab[2], o_cnt[2], o_prob[2];
    for (int i = 0; i < 2; ++i) {
        o_ab[i] = off; off = align_up(off + (size_t)lay * sizeof(uint32_t));
        o_cnt[i] = off; off = align_up(off + (size_t)lay * sizeof(int64_t));
        o_prob[i] = off; off = align_up(off + (size_t)lay * sizeof(float));
    }
    const size_t o_cc = off; off = align_up(off + (size_t)lay * 4 * sizeof(int64_t));
    const size_t o_cp = off; off = align_up(off + (size_t)lay * 4 * sizeof(float));
    const size_t o_wg = off; off = align_up(off + (size_t)nwg_cap * sizeof(uint32_t));
    const int64_t nws_cap = lay;                            // look-back words: sample_multi_kernel<4> has a workgroup per prefix
    const size_t o_ws = off; off = align_up(off + (size_t)nws_cap * sizeof(unsigned long long));
    const size_t o_U = off; off = align_up(off + (size_t)U_SLOTS * sizeof(int64_t));
    if (cap > net->samp_cap) {
        HIP_TRY(hipDeviceSynchronize());
        if (net->d_samp) (void)hipFree(net->d_samp);
        net->d_samp = nullptr; net->samp_cap = 0;
        HIP_TRY(hipMalloc(&net->d_samp, off));
        // tag 0 = never written; on the call's own stream (a null-stream memset is not ordered against a non-blocking one)
        HIP_TRY(hipMemsetAsync(static_cast<char *>(net->d_samp) + o_ws, 0, (size_t)nws_cap * sizeof(unsigned long long), s));
        net->samp_cap = cap;
        net->samp_seq = 0;
    }
    char *base = static_cast<char *>(net->d_samp);
    SampleBufs b;
    for (int i = 0; i < 2; ++i) {
        b.ab[i] = reinterpret_cast<uint32_t *>(base + o_ab[i]);
        b.cnt[i] = reinterpret_cast<int64_t *>(base + o_cnt[i]);
        b.prob[i] = reinterpret_cast<float *>(base + o_prob[i]);
    }
    b.child_cnt = reinterpret_cast<int64_t *>(base + o_cc);
    b.child_prob = reinterpret_cast<float *>(base + o_cp);
    b.wg_total = reinterpret_cast<uint32_t *>(base + o_wg);
    b.wg_state = reinterpret_cast<unsigned long long *>(base + o_ws);
    b.U = reinterpret_cast<int64_t *>(base + o_U);
    b.ctl = net->ctl;

    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    int64_t bound = 1;                                     // worst-case prefixes entering level n: min(4^n, cap)
    int n_first = 0;
    // the blocks' MLPs on the matrix cores when the fragment form of the current weights exists (naqs_net_set_weights; after
    // naqs_net_set_amp_weights only the VALU rows are current); NAQS_SAMPLE_MFMA=0: always the VALU form.  The two forms
    // round differently (f32 FMA chains vs the six-term bf16 split), so a draw may differ between them — never between
    // the ways of cutting the tree into launches, which all take the same form
    const naqs::ushort_t *wamp = (net->d_wamp != nullptr && net->wamp_fresh && naqs::env_int("NAQS_SAMPLE_MFMA", 1) != 0) ? net->d_wamp : nullptr;
    const size_t mf_wave_bytes = 128 * sizeof(float);                 // per wave: the raw outputs [16 prefixes][8] of its item
    // 0: per-level launches only, 1 (default): 4 levels / 256 threads, 2: 5 levels / 1024 threads — measured slower
    // (94 us against 38 + 18 for the fifth level on its own: sixteen latency-bound waves on one CU)
    const int head = naqs::env_int("NAQS_SAMPLE_HEAD", 1);
    const bool use_head = head >= 1 && ((head >= 2 && d.P > 5 && cap >= 1024) || (d.P > 4 && cap >= 256));
    if (!(use_head && !(head >= 2 && d.P > 5 && cap >= 1024))) {
        // no launch of this call can host a pending amplitude re-pack (naqs_vmc_step leaves it to the four-level head launch
        // below): the amplitude jobs in order first — every level reads them
        st = naqs::net_flush_amp_pack(net, s);
        if (st != NAQS_OK) return st;
    }
    if (use_head) {
        const bool big = head >= 2 && d.P > 5 && cap >= 1024;
        const int hl = big ? 5 : 4;
        const int nin = 2 * (hl - 1);
        const size_t lds = wamp ? (size_t)(big ? 16 : 4) * mf_wave_bytes : ((size_t)d.Ha * ((nin + 1 + 5 + 3) & ~3) + 8) * sizeof(float);
        if (big) {
            if (lds > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sample_head_kernel<1024, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            NAQS_KLAUNCH((sample_head_kernel<1024, 5>), dim3(1), dim3(1024), lds, s, d, net->d_w, b, n_samples, k0, k1, wamp, naqs::PackPhaseArgs{});
        } else {
            // the last update's re-pack, if a training step left it pending, rides in this launch: the phase layers' share and
            // (matrix-core form of the block MLPs, and the fragments of the hl pairs workgroup 0 reads packed by the update's own
            // launch) the amplitude blocks'
            naqs::PackPhaseArgs pk;
            st = naqs::net_take_pending_pack(net, s, &pk, wamp != nullptr ? hl : 0);
            if (st != NAQS_OK) return st;
            NAQS_KLAUNCH((sample_head_kernel<256, 4>), dim3(1 + (unsigned)pk.n_wgs), dim3(256), lds, s, d, net->d_w, b, n_samples, k0, k1, wamp, pk);
        }
        HIP_TRY(hipGetLastError());
        n_first = hl;
        for (int n = 0; n < hl; ++n) bound *= 4;
    } else {
        NAQS_KLAUNCH(sample_init_kernel, dim3(1), dim3(64), 0, s, b, n_samples);
        HIP_TRY(hipGetLastError());
    }
    // one launch per level (expand + compaction with a look-back scan across workgroups) unless NAQS_SAMPLE_FUSED=0 or
    // the level could need more workgroups than are resident at once (the look-back waits on lower-indexed workgroups)
    const bool fused_levels = naqs::env_int("NAQS_SAMPLE_FUSED", 1) == 1;
    const int64_t resident_wg = (int64_t)net->cu_count * 8;
    if (net->samp_seq >= 0x00FFFFFFu) {                    // the 24-bit call tag is about to repeat: forget every old word
        HIP_TRY(hipMemsetAsync(b.wg_state, 0, (size_t)nws_cap * sizeof(unsigned long long), s));
        net->samp_seq = 0;
    }
    ++net->samp_seq;
    long long *clk_dev = nullptr;
    if (naqs::env_int("NAQS_DEBUG_SAMPLE_CLOCKS", 0) == 1) {          // developer aid: stamps of workgroup 0 per level
        HIP_TRY(hipMalloc((void **)&clk_dev, MAXP * 12 * sizeof(long long)));
        HIP_TRY(hipMemset(clk_dev, 0, MAXP * 12 * sizeof(long long)));
    }
    // Levels per launch beyond the head: one (sample_level_kernel; or an expand and a scatter launch) unless the PREVIOUS
    // draw's size of the level a launch would start at says that few enough workgroups will be active for
    // sample_multi_kernel (three levels while <= 2048 prefixes enter — 512 workgroups of 4 —, two while <= 8192; measured: 4096 / 16384 is 25 us slower per N2 step).  The
    // sizes are a hint read from mapped memory without synchronising (stale, or from another cap, at worst a slower cut);
    // the samples do not depend on the cut.  NAQS_SAMPLE_MULTI=1: always one level per launch.
    const int multi = fused_levels ? std::min(4, std::max(1, naqs::env_int("NAQS_SAMPLE_MULTI", 4))) : 1;
    // two runs per GPU: the launches below wait across workgroups without a bound on the waiters — the device's turn (below;
    // the head launch above needs none: one busy workgroup, and hosted waiters that only wait for jobs which wait for nobody)
    if (n_first < d.P) naqs::lookback_turn_begin(net);
    int half = n_first & 1;                                // which half of the ping-pong arrays holds the level a launch starts at
    const int64_t multi3_max = naqs::env_int("NAQS_SAMPLE_MULTI3_MAX", 2048);
    volatile const int64_t *hint = net->h_info + 4;
    for (int n = n_first; n < d.P;) {
        const int left = d.P - n;
        int nl = 1;
        if (multi > 1 && left >= 2) {
            const int64_t h = hint[n];
            // (four levels: one entry prefix per workgroup, i.e. at most `multi3_max / 4` of them enter)
            // and only while the worst-case grid is small: the launch must cover min(4^n, cap) prefixes whatever the hint says)
            if (h > 0 && 4 * h <= multi3_max && left >= 4 && left != 5 && multi >= 4 && std::min(bound, cap) <= 4096) nl = 4;   // 6 -> 4 + 2, 4 -> 4
            else {
                const int want = left == 4 ? 2 : std::min(left, 3);              // 6 -> 3 + 3, 5 -> 3 + 2, 4 -> 2 + 2
                if (h > 0 && h <= multi3_max && want == 3 && multi >= 3) nl = 3;
                else if (h > 0 && h <= 4 * multi3_max) nl = 2;
            }
        }
        const int n_end = n + nl - 1;                         // last level of this launch
        const int nin = n_end == 0 ? 1 : 2 * n_end;
        const size_t lds = wamp ? (size_t)(SB / WAVE) * mf_wave_bytes : ((size_t)d.Ha * ((nin + 1 + 5 + 3) & ~3) + 8) * sizeof(float);
        const int last = n_end == d.P - 1 ? 1 : 0;
        if (nl > 1) {
            const int E = 64 >> (2 * (nl - 1));
            const unsigned grid_m = (unsigned)((std::min(bound, cap) + E - 1) / E);
            const uint32_t tag = (net->samp_seq << 8) | (uint32_t)(n + 1);
            if (nl == 4)
                NAQS_KLAUNCH((sample_multi_kernel<4>), dim3(grid_m), dim3(SB), lds, s, d, net->d_w, n, b, half, k0, k1, tag, cap, last,
                                   keys_dev, counts_dev, probs_dev, wamp, early, seq);
            else if (nl == 3)
                NAQS_KLAUNCH((sample_multi_kernel<3>), dim3(grid_m), dim3(SB), lds, s, d, net->d_w, n, b, half, k0, k1, tag, cap, last,
                                   keys_dev, counts_dev, probs_dev, wamp, early, seq);
            else
                NAQS_KLAUNCH((sample_multi_kernel<2>), dim3(grid_m), dim3(SB), lds, s, d, net->d_w, n, b, half, k0, k1, tag, cap, last,
                                   keys_dev, counts_dev, probs_dev, wamp, early, seq);
            HIP_TRY(hipGetLastError());
            for (int i = 0; i < nl; ++i) bound = bound > cap ? bound : bound * 4;
            n += nl;
            half ^= 1;
            continue;
        }
        const unsigned grid = (unsigned)((std::min(bound, cap) + SB - 1) / SB);
        const unsigned grid_e = (unsigned)((std::min(bound, cap) + EXP_PARENTS - 1) / EXP_PARENTS);
        if (fused_levels && (int64_t)grid_e <= resident_wg) {
            const uint32_t tag = (net->samp_seq << 8) | (uint32_t)(n + 1);
            NAQS_KLAUNCH(sample_level_kernel, dim3(grid_e), dim3(SB), lds, s, d, net->d_w, n, b, half, k0, k1, tag, cap, last,
                               keys_dev, counts_dev, probs_dev, clk_dev, wamp, early, seq);
            HIP_TRY(hipGetLastError());
        } else {
            NAQS_KLAUNCH(sample_expand_kernel, dim3(grid_e), dim3(SB), lds, s, d, net->d_w, n, b, half, k0, k1, wamp);
            HIP_TRY(hipGetLastError());
            NAQS_KLAUNCH(sample_scatter_kernel, dim3(grid), dim3(SB), 0, s, d, n, b, half, cap, last, keys_dev, counts_dev,
                               probs_dev);
            HIP_TRY(hipGetLastError());
        }
        bound = bound > cap ? bound : bound * 4;
        ++n;
        half ^= 1;
    }
    naqs::SampleFinishJob fin;
    fin.U = b.U; fin.P = d.P; fin.info = info_dev; fin.counts = counts_dev; fin.weights = weights_dev; fin.early = early; fin.seq = seq;
    fin.levels_out = net->d_info_alias + 4;
    net->fin_job = fin;
    net->fin_pending = true;
    if (!net->hold_finish || clk_dev) {                    // (hold_finish: the caller's next launch hosts it, or the caller flushes)
        st = naqs::net_sample_finish_flush(net, s);
        if (st != NAQS_OK) return st;
    }
    if (clk_dev) {
        long long h[MAXP * 12];
        HIP_TRY(hipMemcpy(h, clk_dev, sizeof(h), hipMemcpyDeviceToHost));
        (void)hipFree(clk_dev);
        int64_t hu[U_SLOTS];
        HIP_TRY(hipMemcpy(hu, b.U, sizeof(hu), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[naqs sample clocks] level sizes:");
        for (int n = 0; n <= d.P; ++n) std::fprintf(stderr, " %lld", (long long)hu[n]);
        std::fprintf(stderr, "\n");
        for (int n = n_first; n < d.P; ++n) {
            std::fprintf(stderr, "[naqs sample clocks] level %d: start(rel. prev end) %lld |", n,
                         n > n_first && h[(n - 1) * 12 + 5] ? h[n * 12] - h[(n - 1) * 12 + 5] : 0ll);
            for (int k = 1; k < 9; ++k) std::fprintf(stderr, " %lld", h[n * 12 + k] ? h[n * 12 + k] - h[n * 12] : 0ll);
            std::fprintf(stderr, "\n");
        }
    }
    return NAQS_OK;
}

// Two runs on one GPU (the farm's `--per-gpu 2`; NAQS_SHARED_GPU=1 when the handle is created): their samplers' look-back launches
// can take TURNS on the device.  A look-back workgroup waits for every workgroup before it, which the dispatcher is certain to
// have started only while ONE such launch is in flight (naqs_poll.hpp; DESIGN.md 4.13: dispatch is in index order per XCD, so
// two launches of ~29 one-per-CU waiters per XCD can fill each other's XCDs and stand until their budgets expire).  A turn is
// held on the HOST: from before the first look-back launch of a call is queued (behind its head launch, which goes ahead and
// runs while the host waits for the turn) until the call's last level's size is known — naqs_vmc_step / naqs_vmc_run wait for
// exactly that word anyway (sample_and_wait), and by then every word a look-back of the call can wait for has been published —
// or, for the plain sampler calls, until the stream has drained.  A spin lock per device (a turn is tens of microseconds; a
// sleeping lock's wake-up is longer than that), no event between the two queues: those hand-overs were measured first and cost
// the farm 11 %.  The other launches of a step that wait inside (the column-split forward: at most 16
// consumers per XCD, all behind their producers; the hosted re-pack: HOSTED_WAITING_WGS, behind jobs that wait for nobody) end
// whatever runs beside them — a chain of depth two cannot close a cycle — and need no turn.
namespace {
constexpr int GATE_DEVICES = 64;
std::atomic<const naqs_net *> g_lookback_holder[GATE_DEVICES];
}  // namespace

void naqs::lookback_turn_begin(naqs_net *net) {
    if (!net->shared_gpu || net->turn_held || net->device < 0 || net->device >= GATE_DEVICES) return;
    std::atomic<const naqs_net *> &h = g_lookback_holder[net->device];
    bool waited = false;
    for (const naqs_net *none = nullptr; !h.compare_exchange_weak(none, net, std::memory_order_acquire, std::memory_order_relaxed); none = nullptr) {
        waited = true;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    if (waited) ++net->lookback_turns;
    net->turn_held = true;
}
void naqs::lookback_turn_end(naqs_net *net) {
    if (!net->turn_held) return;
    net->turn_held = false;
    g_lookback_holder[net->device].store(nullptr, std::memory_order_release);
}

static int net_sample_impl(naqs_net_t *net, int64_t n_samples, uint64_t seed, int64_t max_unique, uint64_t *keys_dev,
                           int64_t *counts_dev, float *probs_dev, double *weights_dev, int64_t *info_dev, void *stream,
                           int64_t *early = nullptr, int64_t seq = 0) {
    int st = net_sample_enqueue(net, n_samples, seed, max_unique, keys_dev, counts_dev, probs_dev, weights_dev, info_dev, stream, early, seq);
    if (net != nullptr && net->turn_held && !net->turn_caller_ends) {          // nobody above waits for this call: wait here
        if (hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)) != hipSuccess && st == NAQS_OK) st = NAQS_ERR_HIP;
        naqs::lookback_turn_end(net);
    }
    return st;
}

NAQS_API int naqs_net_share_device(naqs_net_t *net, int on, int64_t *turns) {
    if (!net || on < -1 || on > 1) return NAQS_ERR_INVALID;
    if (on >= 0) net->shared_gpu = on == 1;
    if (turns) *turns = net->lookback_turns;
    return NAQS_OK;
}

int naqs::net_sample_finish_flush(naqs_net *net, hipStream_t s) {
    if (!net->fin_pending) return NAQS_OK;
    net->fin_pending = false;
    NAQS_KLAUNCH(sample_finish_kernel, dim3(1), dim3(net->fin_job.weights ? FIN_THREADS : 64), 0, s, net->fin_job);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

NAQS_API int naqs_net_sample(naqs_net_t *net, int64_t n_samples, uint64_t seed, int64_t max_unique, uint64_t *keys_dev,
                             int64_t *counts_dev, float *probs_dev, int64_t *info_dev, void *stream) {
    return net_sample_impl(net, n_samples, seed, max_unique, keys_dev, counts_dev, probs_dev, nullptr, info_dev, stream);
}

NAQS_API int naqs_net_sample_weighted(naqs_net_t *net, int64_t n_samples, uint64_t seed, int64_t max_unique, uint64_t *keys_dev,
                                      int64_t *counts_dev, float *probs_dev, double *weights_dev, int64_t *info_dev,
                                      void *stream) {
    if (!weights_dev) return NAQS_ERR_INVALID;
    return net_sample_impl(net, n_samples, seed, max_unique, keys_dev, counts_dev, probs_dev, weights_dev, info_dev, stream);
}

// naqs_phase_grad.hip (naqs_vmc_step, naqs_vmc_sample_forward_eloc): the weighted draw whose (M, overflow, seq) also go to
// the mapped host words `early` as early as they are known
int naqs::net_sample_early(naqs_net *net, int64_t n_samples, uint64_t seed, int64_t max_unique, uint64_t *keys_dev, int64_t *counts_dev,
                           float *probs_dev, double *weights_dev, int64_t *info_dev, void *stream, int64_t *early, int64_t seq) {
    if (!weights_dev || !early) return NAQS_ERR_INVALID;
    return net_sample_impl(net, n_samples, seed, max_unique, keys_dev, counts_dev, probs_dev, weights_dev, info_dev, stream, early, seq);
}

NAQS_API int naqs_rng_binomial_host(int64_t n, double p, uint64_t seed, int64_t reps, int64_t *out) {
    if (reps < 0 || (reps > 0 && !out) || n < 0 || n > (1ll << 44)) return NAQS_ERR_INVALID;
    for (int64_t i = 0; i < reps; ++i) {
        naqs::RngStream g{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)i, (uint32_t)(i >> 32), 0u, 0u};
        out[i] = naqs::binomial(n, p, g);
    }
    return NAQS_OK;
}

// The group draws of the tree sampler on the device, for statistical tests of exactly the code the sampler runs: draw i is
// binomial_group<G> of (n_i, p_i) — arrays of `cases` entries, draw i taking entry i % cases, so that one wave holds draws of
// different regimes side by side like a tree level does — keyed by (seed, i).
namespace {
template <int G>
__global__ __launch_bounds__(256) void binomial_group_test_kernel(const int64_t *__restrict__ n, const double *__restrict__ p,
                                                                  const int cases, const uint32_t k0, const uint32_t k1,
                                                                  const int64_t reps, int64_t *__restrict__ out) {
    const int64_t lane_id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = lane_id / G;                          // the draw of this group of G lanes
    const bool need = i < reps;
    const int c = (int)(i % cases);
    const int64_t v = naqs::binomial_group<G>(need, need ? n[c] : 0, need ? p[c] : 0.0, k0, k1, (uint32_t)i, (uint32_t)(i >> 32));
    if (need && (lane_id % G) == 0) out[i] = v;
}
}  // namespace

NAQS_API int naqs_rng_binomial_device(int group, int cases, const int64_t *n_dev, const double *p_dev, uint64_t seed, int64_t reps,
                                      int64_t *out_dev, void *stream) {
    if ((group != 2 && group != 4) || cases <= 0 || !n_dev || !p_dev || reps < 0 || (reps > 0 && !out_dev)) return NAQS_ERR_INVALID;
    if (reps == 0) return NAQS_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((reps * group + 255) / 256);
    if (group == 4) NAQS_KLAUNCH(binomial_group_test_kernel<4>, dim3(grid), dim3(256), 0, s, n_dev, p_dev, cases, (uint32_t)seed, (uint32_t)(seed >> 32), reps, out_dev);
    else NAQS_KLAUNCH(binomial_group_test_kernel<2>, dim3(grid), dim3(256), 0, s, n_dev, p_dev, cases, (uint32_t)seed, (uint32_t)(seed >> 32), reps, out_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

NAQS_API int naqs_rng_philox_host(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
    if (!counter || !key || !out) return NAQS_ERR_INVALID;
    const uint32_t c[4] = {counter[0], counter[1], counter[2], counter[3]};
    uint32_t r[4];
    naqs::philox4x32_10(c, key[0], key[1], r);
    for (int i = 0; i < 4; ++i) out[i] = r[i];
    return NAQS_OK;
}

NAQS_API int naqs_rng_math_host(int fn, int64_t n, const double *x, double *y) {
    if (fn < 0 || fn > 3 || n < 0 || (n > 0 && (!x || !y))) return NAQS_ERR_INVALID;
    for (int64_t i = 0; i < n; ++i) {
        switch (fn) {
            case 0: y[i] = naqs::log_fast(x[i]); break;
            case 1: y[i] = naqs::log1m_fast(x[i]); break;
            case 2: y[i] = naqs::exp_fast(x[i]); break;
            default: {
                uint64_t b;
                __builtin_memcpy(&b, &x[i], sizeof(b));
                y[i] = naqs::u01((uint32_t)(b >> 32), (uint32_t)b);
            }
        }
    }
    return NAQS_OK;
}
