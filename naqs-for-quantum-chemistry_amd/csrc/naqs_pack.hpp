// naqs_pack.hpp — the single-phase network's re-pack in the f16x2 format (the phase layers' share; since round 6 the amplitude
// blocks' too), as device functions two kernels use: pack_net_kernel (naqs_logpsi.hip: naqs_net_set_weights, everything in one launch behind net_bounds_kernel) and
// the sampler's first launch of a training step (naqs_sample.hip: sample_head_kernel<256, 4> is ONE workgroup for ~30 us and
// reads nothing but the amplitude blocks — the workgroups behind it re-pack the phase layers of the last update meanwhile,
// instead of 13 us of launches between the update and the sampler).  Inside that launch the weight maxima are the first jobs
// and reach the packing jobs as tagged words (naqs_net.hpp: PhaseRaw): consumers have higher block indices than producers.
// All functions are written for 256-thread workgroups and take the workgroup's index within its job (bx of nbx).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "naqs_amp_mfma.hpp"
#include "naqs_net.hpp"
#include "naqs_poll.hpp"

#ifndef NAQS_PH_CBT
#define NAQS_PH_CBT 4
#endif

namespace naqs {

// Output column of MFMA tile cb, tile column nn.  When a layer's width is a multiple of 64 the four tiles of a wave are
// interleaved (column = 64 (cb / 4) + 4 nn + cb % 4): a lane then owns four ADJACENT columns of every row, and the
// write-back packs them into one 8-byte LDS store per plane with no cross-lane traffic.
__host__ __device__ __forceinline__ int tile_col(int cb, int nn, int N_pad) {
    return (NAQS_PH_CBT == 4 && (N_pad & 63) == 0) ? ((cb >> 2) << 6) + 4 * nn + (cb & 3) : cb * 16 + nn;
}

struct PhasePackJobs { int64_t src_off[MAXL]; int32_t K[MAXL], N[MAXL]; };

// the weight maxima the f16x2 scales are derived from: per phase layer max |W|, max_j sum_k |W[j][k]|, max |b| — one wave per
// row (a 512 x 512 layer: 512 waves, one load round trip each); workgroups 0 .. BOUNDS_WG - 1 of job l publish their partials
constexpr int BOUNDS_WAVES = 4;
__device__ __forceinline__ void net_bounds_body(const float *__restrict__ flat, const PhasePackJobs &jobs, PhaseRaw *__restrict__ raw,
                                                const int l, const uint32_t tag, const int bx, const PollCtl *ctl) {
    __shared__ float s_red[3][BOUNDS_WAVES];
    if (bx >= BOUNDS_WG) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = jobs.K[l], N = jobs.N[l];
    const float *W = flat + jobs.src_off[l], *b = W + (size_t)N * K;
    float mw = 0.0f, mr = 0.0f, mb = 0.0f;
    for (int j = bx * BOUNDS_WAVES + wave; j < N; j += BOUNDS_WG * BOUNDS_WAVES) {
        float sum = 0.0f;
#pragma unroll 8
        for (int k = lane; k < K; k += 64) { const float v = fabsf(W[(size_t)j * K + k]); sum += v; mw = fmaxf(mw, v); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        mr = fmaxf(mr, sum);
        mb = fmaxf(mb, fabsf(b[j]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mw = fmaxf(mw, __shfl_xor(mw, o, 64));
    if (lane == 0) { s_red[0][wave] = mw; s_red[1][wave] = mr; s_red[2][wave] = mb; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float m = 0.0f;
        for (int w = 0; w < BOUNDS_WAVES; ++w) m = fmaxf(m, s_red[threadIdx.x][w]);
        unsigned long long *dst = threadIdx.x == 0 ? raw->max_w[l] : (threadIdx.x == 1 ? raw->max_rowsum[l] : raw->max_b[l]);
        if (!(poll_drop(ctl, POLL_PACK_BOUNDS) && l == 0 && bx == 0))
            __hip_atomic_store(&dst[bx], ((unsigned long long)tag << 32) | __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// maximum of the BOUNDS_WG (= 64) per-workgroup partials: one polled load per lane + a wave reduction, every lane gets the
// result.  (A serial loop over the 64 entries — by one thread for the scale chain, by every workgroup for its weight scale —
// made the packing launch 36 us instead of 8.)  Must be called by whole waves.
static_assert(BOUNDS_WG == 64, "one partial per lane");
// (the wait is bounded — naqs_poll.hpp: `ok` turns false, for every lane of the wave, when a word did not arrive in time; the
// callers then leave without writing anything)
__device__ __forceinline__ float bounds_max(const unsigned long long (&part)[BOUNDS_WG], const uint32_t tag, const PollCtl *ctl, bool &ok) {
    unsigned long long word;
    const bool mine = poll_tagged<2>(&part[threadIdx.x & 63], tag, word, ctl, POLL_PACK_BOUNDS, threadIdx.x & 63);
    if (!__all(mine)) ok = false;
    float m = __uint_as_float((uint32_t)word);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    return m;
}

// weight scale of layer l: max |W| sw in [2^13, 2^14)
__device__ __forceinline__ float phase_weight_scale(const PhaseRaw &raw, int l, const uint32_t tag, const PollCtl *ctl, bool &ok) {
    return pow2_clamped(13 - exp_of(bounds_max(raw.max_w[l], tag, ctl, ok)));
}
// all scales of the network (called by one whole wave; lane 0 writes): activation bound chain
// |h_l| <= rowsum_l bound_{l-1} + max|b_l|, inputs +-1
__device__ __forceinline__ void phase_scales_fill(const PhaseRaw &raw, int n_lin, PhaseScales *out, const uint32_t tag, const PollCtl *ctl) {
    float bound = 1.0f, s_in = 1.0f;
    PhaseScales v;
    bool ok = true;
    for (int l = 0; l < n_lin; ++l) {
        bound = bounds_max(raw.max_rowsum[l], tag, ctl, ok) * bound + bounds_max(raw.max_b[l], tag, ctl, ok);
        const float sw = phase_weight_scale(raw, l, tag, ctl, ok);
        const float sn = l + 1 < n_lin ? pow2_clamped(14 - exp_of(bound)) : 1.0f;          // bound sn < 2^15
        v.sw[l] = sw;
        v.sn[l] = sn;
        v.isn[l] = 1.0f / sn;
        v.c[l] = (sn / s_in) / sw;                   // powers of two: exact
        s_in = sn;
    }
    if (ok && (threadIdx.x & 63) == 0)
        for (int l = 0; l < n_lin; ++l) { out->sw[l] = v.sw[l]; out->sn[l] = v.sn[l]; out->isn[l] = v.isn[l]; out->c[l] = v.c[l]; }
}

// f32 [N][K] -> two f16 planes of sw * W, zero-padded and tiled [plane][N_pad/16][Kh_pad/32][64 lanes][8]
__device__ __forceinline__ void pack_phase_f16(const float *__restrict__ src, int K, int N, int Kh_pad, int N_pad,
                                               ushort_t *__restrict__ Wd, const float sw, const int bx, const int nbx) {
    const int total = N_pad * Kh_pad;
    for (int e = bx * 256 + threadIdx.x; e < total; e += nbx * 256) {
        // e = ((cb * KC + kc) * 64 + kg * 16 + nn) * 8 + j   <-   W[cb*16 + nn][kc*32 + kg*8 + j]
        const int j = e & 7, nn = (e >> 3) & 15, kg = (e >> 7) & 3, blk = e >> 9;
        const int KC = Kh_pad >> 5, cb = blk / KC, kc = blk - cb * KC;
        const int n = tile_col(cb, nn, N_pad), k = kc * 32 + kg * 8 + j;
        const float x = (n < N && k < K) ? src[n * K + k] * sw : 0.0f;
        ushort_t h1, h2;
        split2(x, h1, h2);
        Wd[e] = h1; Wd[(size_t)total + e] = h2;
    }
}

// one phase layer in the f16x2 format: the (padded) bias, the output layer also as plain row-major f32 [N_pad][K_pad]
// (phase_kernel_ws multiplies by it on the VALU), the two scaled planes; workgroup 0 of layer 0 also writes the scales
__device__ __forceinline__ void pack_phase_job_f16x2(const float *__restrict__ flat, const NetDims &d, const PhasePackJobs &jobs,
                                                     float *__restrict__ w, ushort_t *__restrict__ wh, const int l,
                                                     const PhaseRaw *__restrict__ raw, PhaseScales *__restrict__ scales,
                                                     const uint32_t tag, const int bx, const int nbx, const PollCtl *ctl) {
    if (l == 0 && bx == 0 && threadIdx.x < 64) phase_scales_fill(*raw, d.n_lin, scales, tag, ctl);
    bool ok = true;
    const float sw = phase_weight_scale(*raw, l, tag, ctl, ok);        // (every wave of the workgroup: whole waves poll)
    if (__syncthreads_or(!ok)) return;                                  // a wait ran out: nothing of this job is written
    const float *src = flat + jobs.src_off[l];
    for (int n = bx * 256 + threadIdx.x; n < d.N_pad[l]; n += nbx * 256)
        w[d.b_off[l] + n] = n < jobs.N[l] ? src[jobs.N[l] * jobs.K[l] + n] : 0.0f;
    if (l == d.n_lin - 1) {
        const int Kp = d.K_pad[l], total = d.N_pad[l] * Kp;
        for (int e = bx * 256 + threadIdx.x; e < total; e += nbx * 256) {
            const int n = e / Kp, k = e - n * Kp;
            w[d.w_off[l] + e] = (n < jobs.N[l] && k < jobs.K[l]) ? src[n * jobs.K[l] + k] : 0.0f;
        }
    }
    pack_phase_f16(src, jobs.K[l], jobs.N[l], d.Kh_pad[l], d.N_pad[l], wh + d.wh_off[l], sw, bx, nbx);
}

// ---- the amplitude blocks' share of a re-pack (moved here from naqs_logpsi.hip in round 6: the sampler's first launch hosts it too) ----
// flat state_dict-order parameters of an amplitude block: src = [W1 [Ha][nin] | b1 [Ha] | W2 [nout][Ha] | b2 [nout]]
struct AmpSrcOff { int64_t off[MAXP]; };

// the pair's three f16x2 scales from its weight maxima: |W1| and |b1| (one scale: b1 rides in the W1 fragments), the row bound
// max_j (|b1[j]| + sum_k |W1[j][k]|) (inputs +-1) and |W2|.  Maxima are exact whatever the order they are taken in, so every
// way of splitting the rows over threads gives the same scales.
struct AmpScales { float sw1, sh, sw2; };
__device__ __forceinline__ void amp_row_maxima(const float *W1, const float *b1, const float *W2,
                                               const int Ha, const int nin, const int nout, const int n, const int j0, const int dj,
                                               float &mw1, float &mrow, float &mw2) {
    for (int j = j0; j < Ha; j += dj) {
        float sum = fabsf(b1[j]);
        mw1 = fmaxf(mw1, sum);
        if (n > 0)
            for (int k = 0; k < nin; ++k) { const float v = fabsf(W1[j * nin + k]); sum += v; mw1 = fmaxf(mw1, v); }
        mrow = fmaxf(mrow, sum);
        for (int c = 0; c < nout; ++c) mw2 = fmaxf(mw2, fabsf(W2[c * Ha + j]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mw1 = fmaxf(mw1, __shfl_xor(mw1, o, 64)); mrow = fmaxf(mrow, __shfl_xor(mrow, o, 64)); mw2 = fmaxf(mw2, __shfl_xor(mw2, o, 64));
    }
}
__device__ __forceinline__ AmpScales amp_scales_of(const float mw1, const float mrow, const float mw2) {
    return AmpScales{pow2_clamped(13 - exp_of(mw1)), pow2_clamped(14 - exp_of(mrow)), pow2_clamped(13 - exp_of(mw2))};
}
// elements [e0, e1) step de of a pair's fragments and (first == true) its 16 trailing floats
__device__ __forceinline__ void amp_fragments_write(const float *src, const int Ha, const int nin, const int nout, const int n,
                                                    const AmpScales sc, ushort_t *__restrict__ dst, const int e0, const int de, const bool first,
                                                    const int t16) {
    const int CT = Ha >> 4, KC = Ha >> 5;
    const float *W1 = src, *b1 = src + Ha * nin, *W2 = src + Ha * nin + Ha;
    if (first && t16 < 16) {
        float v = 0.0f;
        if (t16 < nout) v = src[Ha * nin + Ha + nout * Ha + t16];
        else if (t16 == 8) v = sc.sh / sc.sw1;
        else if (t16 == 9) v = (1.0f / sc.sh) / sc.sw2;
        reinterpret_cast<float *>(dst + (size_t)2 * 512 * (CT + KC))[t16] = v;
    }
    const int frag1 = CT * 512, frag2 = KC * 512;
    for (int e = e0; e < frag1 + frag2; e += de) {
        float x;
        size_t o0, plane;
        if (e < frag1) {            // e = (ct * 64 + l) * 8 + j  <-  W1[16 ct + (l & 15)][8 (l >> 4) + j]
            const int j = e & 7, l = (e >> 3) & 63, ct = e >> 9;
            const int h = 16 * ct + (l & 15), k = 8 * (l >> 4) + j;
            x = ((n > 0 && k < nin) ? W1[h * nin + k] : (k == 31 ? b1[h] : 0.0f)) * sc.sw1;        // input 31 == 1 carries b1
            o0 = (size_t)e; plane = (size_t)frag1;
        } else {                    // e' = (kc * 64 + l) * 8 + j  <-  W2[l & 15][16 (2 kc + (j >> 2)) + 4 (l >> 4) + (j & 3)]
            const int e2 = e - frag1;
            const int j = e2 & 7, l = (e2 >> 3) & 63, kc = e2 >> 9;
            const int c = l & 15, k = 16 * (2 * kc + (j >> 2)) + 4 * (l >> 4) + (j & 3);
            x = c < nout ? W2[c * Ha + k] * sc.sw2 : 0.0f;
            o0 = (size_t)2 * frag1 + e2; plane = (size_t)frag2;
        }
        ushort_t h1, h2;
        split2(x, h1, h2);
        dst[o0] = h1; dst[o0 + plane] = h2;
    }
}

// amplitude blocks as MFMA operand fragments of the transposed f16x2 item (naqs_amp_mfma.hpp): per pair W1 planes
// [2][Ha/16][64][8] (A operand: lane (m, kg) = hidden unit 16 ct + m, inputs 8 kg..8 kg + 7, input 31 = b1), W2 planes
// [2][Ha/32][64][8] (A operand: lane (m, kg) = output m, slot j = hidden unit 16 (2 kc + (j >> 2)) + 4 kg + (j & 3)), then
// 16 floats {b2[8], c1, c2}.  Every block of a pair derives the pair's scales itself (<= 1.6 k parameters).
// (256-thread workgroups; bx of nbx: the workgroup's index within pair n's job)
// (pack_amp_mfma_src: from the pair's parameter block wherever it is — grad_finish_kernel's first workgroups hand over the LDS
// copy of what they have just updated)
__device__ __forceinline__ void pack_amp_mfma_src(const float *src, const int Ha, const int nout, ushort_t *__restrict__ wamp, const int n,
                                                  const int bx, const int nbx) {
    __shared__ float s_red[3][4];
    const int nin = n == 0 ? 1 : 2 * n;
    const int CT = Ha >> 4, KC = Ha >> 5;
    // (the launch is sized for the biggest job: a surplus workgroup must leave before it derives the pair's scales — with
    // 256 workgroups per job doing that for nothing this launch took 38 us instead of 10)
    if (bx * 256 >= (CT + KC) * 512) return;
    float mw1 = 0.0f, mrow = 0.0f, mw2 = 0.0f;
    amp_row_maxima(src, src + Ha * nin, src + Ha * nin + Ha, Ha, nin, nout, n, (int)threadIdx.x, 256, mw1, mrow, mw2);
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = mw1; s_red[1][threadIdx.x >> 6] = mrow; s_red[2][threadIdx.x >> 6] = mw2; }
    __syncthreads();
    mw1 = fmaxf(fmaxf(s_red[0][0], s_red[0][1]), fmaxf(s_red[0][2], s_red[0][3]));
    mrow = fmaxf(fmaxf(s_red[1][0], s_red[1][1]), fmaxf(s_red[1][2], s_red[1][3]));
    mw2 = fmaxf(fmaxf(s_red[2][0], s_red[2][1]), fmaxf(s_red[2][2], s_red[2][3]));
    amp_fragments_write(src, Ha, nin, nout, n, amp_scales_of(mw1, mrow, mw2), wamp + (size_t)n * amp_mfma_pair_elems(Ha), bx * 256 + (int)threadIdx.x,
                        nbx * 256, bx == 0, (int)threadIdx.x);
}
__device__ __forceinline__ void pack_amp_mfma_body(const float *__restrict__ flat, const NetDims &d, const AmpSrcOff &so,
                                                   ushort_t *__restrict__ wamp, const int n, const int bx, const int nbx) {
    pack_amp_mfma_src(flat + so.off[n], d.Ha, d.n_out_amp, wamp, n, bx, nbx);
}
// parameters of pair n's block: W1 [Ha][max(1, 2n)] | b1 [Ha] | W2 [nout][Ha] | b2 [nout]
__host__ __device__ __forceinline__ int amp_raw_floats(const int Ha, const int nout, const int n) { return Ha * (n == 0 ? 1 : 2 * n) + Ha + nout * Ha + nout; }

// every amplitude block's VALU rows: Ha rows [W1[j][:] | b1[j] | W2[0..5)[j] | 0-pad to a multiple of 4 floats], then b2 padded to 8
__device__ __forceinline__ void pack_amp_body(const float *__restrict__ flat, const NetDims &d, const AmpSrcOff &so,
                                              float *__restrict__ w, const int n, const int bx, const int nbx) {
    const int Ha = d.Ha, nout = d.n_out_amp, nin = n == 0 ? 1 : 2 * n;
    const float *src = flat + so.off[n];
    float *dst = w + d.amp_off[n];
    const int S = (nin + 1 + 5 + 3) & ~3, total = Ha * S + 8;
    for (int e = bx * 256 + threadIdx.x; e < total; e += nbx * 256) {
        float v = 0.0f;
        if (e < Ha * S) {
            const int j = e / S, c = e - j * S;
            if (c < nin) v = src[j * nin + c];
            else if (c == nin) v = src[Ha * nin + j];
            else if (c - nin - 1 < nout) v = src[Ha * nin + Ha + (c - nin - 1) * Ha + j];
        } else if (e - Ha * S < nout) {
            v = src[Ha * nin + Ha + nout * Ha + (e - Ha * S)];
        }
        dst[e] = v;
    }
}

// row-major zero-padded copy y of the phase weights (what the backward GEMMs read)
__device__ __forceinline__ void pack_wb_job(const float *__restrict__ flat, const WbPackJobs &wb, const int y, const int bx, const int nbx) {
    const int total = wb.Np[y] * wb.Kp[y];
    const float *src = flat + wb.src_off[y];
    for (int e = bx * 256 + threadIdx.x; e < total; e += nbx * 256) {
        const int n = e / wb.Kp[y], k = e - n * wb.Kp[y];
        wb.dst[y][e] = (n < wb.N[y] && k < wb.K[y]) ? src[n * wb.K[y] + k] : 0.0f;
    }
}

// the phase share of a re-pack as a flat range of n_wgs workgroups: [maxima: n_lin x BOUNDS_WG][copies: wb.n x gx][layers: n_lin x gxl]
//
// HOSTED_WAITING_WGS (round 6).  The layer jobs WAIT (for the maxima's words) while they hold a workgroup slot, and the hosting
// kernel — sample_head_kernel, ~300 registers per lane — has ONE slot per CU, 32 per XCD.  A launch's workgroups are dealt to the
// eight XCDs and dispatched in index order per XCD only; with two launches in flight (the farm's two runs per GPU) and
// 3 x 256 layer workgroups each, an XCD could fill with one launch's waiting workgroups while that launch's maxima jobs sat
// queued on another XCD that the OTHER launch's waiting workgroups had filled — a circular wait that only the 2 s budget of
// naqs_poll.hpp ends, with an error for both runs.  So a hosted launch gets at most HOSTED_WAITING_WGS waiting workgroups:
// 2 launches x 120 / 8 XCDs = 30 < 32 slots, i.e. some slot of every XCD is always held by (or free for) a workgroup that waits
// for nobody, and every maxima job runs.  (The column-split log-psi kernel is inside the same bound by its own condition,
// 2 x tiles <= CUs: its consumers are at most 16 per XCD and launch.  The sampler's look-back launches are NOT: they, too, have
// one slot per CU, every one of their workgroups waits for all before it, and an N2 launch has ~29 of them per XCD — two in
// flight can in principle take a whole XCD each before the other's first workgroup there.  That needs the two launches to
// overtake each other XCD by XCD within microseconds; it is what is left of the exposure of --per-gpu 2, and why the farm
// starts a run again when a wait of it gave up: experiments/_base.py)
constexpr int HOSTED_WAITING_WGS = 120;
struct PackPhaseArgs {
    const float *flat = nullptr;       // nullptr: nothing to do
    PhasePackJobs jobs;
    WbPackJobs wb;
    float *w = nullptr;
    ushort_t *wh = nullptr;
    PhaseRaw *raw = nullptr;
    PhaseScales *scales = nullptr;
    uint32_t tag = 0;
    int gx = 0, gxl = 0, n_wgs = 0;    // workgroups per copy job / per layer job (the waiting kind: few)
    const PollCtl *ctl = nullptr;
    // round 6 — the amplitude blocks' share as well (amp != 0): [rows: P x gxa][fragments of the pairs from head_pairs on:
    // (P - head_pairs) x gxf] in FRONT of the phase jobs; the fragments of pairs 0 .. head_pairs - 1 — what the hosting launch's
    // own first workgroup reads — exist already: the update's launch packed them (grad_finish_kernel's first workgroups)
    int amp = 0, gxa = 0, gxf = 0, n_amp_wgs = 0, head_pairs = 0;
    AmpSrcOff so;
    ushort_t *wamp = nullptr;
};
__device__ __forceinline__ void pack_phase_dispatch(const NetDims &d, const PackPhaseArgs &a, int bid) {
    if (a.amp) {
        if (bid < a.n_amp_wgs) {
            const int rows = d.P * a.gxa;
            if (bid < rows) pack_amp_body(a.flat, d, a.so, a.w, bid / a.gxa, bid % a.gxa, a.gxa);
            else { bid -= rows; pack_amp_mfma_body(a.flat, d, a.so, a.wamp, a.head_pairs + bid / a.gxf, bid % a.gxf, a.gxf); }
            return;
        }
        bid -= a.n_amp_wgs;
    }
    // [maxima][row-major copies: wait for nobody][layers: WAIT for the maxima] — and few of the last kind (gxl per layer), see
    // HOSTED_WAITING_WGS
    const int nb = d.n_lin * BOUNDS_WG;
    if (bid < nb) { net_bounds_body(a.flat, a.jobs, a.raw, bid / BOUNDS_WG, a.tag, bid % BOUNDS_WG, a.ctl); return; }
    bid -= nb;
    const int n_copy = a.wb.n * a.gx;
    if (bid < n_copy) { pack_wb_job(a.flat, a.wb, bid / a.gx, bid % a.gx, a.gx); return; }
    bid -= n_copy;
    const int y = bid / a.gxl, x = bid - y * a.gxl;
    if (y < d.n_lin) pack_phase_job_f16x2(a.flat, d, a.jobs, a.w, a.wh, y, a.raw, a.scales, a.tag, x, a.gxl, a.ctl);
}

// naqs_logpsi.hip: the pending phase share of the last re-pack (naqs_vmc_step), for a launch that can host it (`out` filled,
// nothing pending afterwards) ...
// (host_levels: the leading pairs the host launch's own first workgroup reads — a pending amplitude share can only ride along
// when the fragments of those pairs were packed by the update itself, net->amp_head_packed; otherwise, or with 0, it is launched
// in order first)
int net_take_pending_pack(naqs_net *net, hipStream_t s, PackPhaseArgs *out, int host_levels = 0);
// ... or as launches of its own, in order on `s` (any other reader of the phase layers)
int net_flush_pack(naqs_net *net, hipStream_t s);

}  // namespace naqs
