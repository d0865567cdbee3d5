// naqs_phase_grad.hip — training forward/backward of the whole orbital NADE in one library call each, gfx950.
//
//   naqs_net_train_forward   keys -> (log|psi|, phase) like naqs_net_logpsi, and the phase MLP's inputs and hidden
//                            activations are left in HBM for the backward pass
//   naqs_net_train_backward  g = d loss / d (log|psi|, phase) per sample -> d loss / d theta for every parameter, flat
//                            in state_dict order
//
// This is the graph torch.autograd builds for the reference's loss.backward() (src/optimizer/energy.py:329-343 through
// src/naqs/network/nade.py:738-770).  The amplitude blocks are naqs_grad.hip; the phase block (Linear/ReLU stack, e.g.
// 18 -> 512 -> 512 -> 4) is two GEMM-shaped kernels per layer on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact
// f32, the precision autograd would use):
//
//   grad_w_kernel    dW_l[n][k] = sum_i delta_l[i][n] * in_l[i][k]  (+ db_l[n] = sum_i delta_l[i][n]): 64 x 64 output
//                    block per workgroup, the sample axis split into slices (one per blockIdx.z) whose partial
//                    blocks are added in fixed order by grad_finish_kernel -> deterministic, no float atomics.
//   grad_in_kernel   delta_{l-1}[i][k] = [in_l[i][k] > 0] * sum_n delta_l[i][n] * W_l[n][k]: 64 samples x 64 columns per
//                    workgroup, ReLU mask fused into the write-back.
//
// Both stage 32-deep operand chunks in LDS (row strides chosen so that the 64 MFMA operand reads of a wave hit
// 32 distinct banks per half-wave) and give each of the 4 waves a 32 x 32 sub-block (2 x 2 MFMA tiles).  All
// matrices live in zero-padded buffers whose leading dimensions are multiples of 64, so there is no edge code.

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <thread>

#include "naqs_common.hpp"
#include "naqs_hash.hpp"
#include "naqs_net.hpp"
#include "naqs_amp_backward.hpp"
#include "naqs_reduce.hpp"
#include "naqs_pack.hpp"

namespace {

using naqs::MAXL;
using naqs::NetDims;
using naqs::DeviceGuard;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TB = 64;       // output block edge
constexpr int CH = 32;       // reduction chunk staged in LDS
constexpr int LDT = 80;      // row stride of a [CH][64] tile: 80 = 16 mod 32 -> lanes (k, k+1) x 16 columns cover 32 banks
constexpr int LDD = 34;      // row stride of a [64][CH] tile: bank = 2 row + k -> conflict-free for 16 rows x 2 k
constexpr int W0_FUSE_MAX_ROWS = 4096;   // tables up to this size: the first layer's weight gradient is formed by the grad_in blocks
constexpr int SUMS_FUSE_MAX_ROWS = 4096; // ... and the weighted sums of E_loc by the seed kernel's first workgroup (naqs_vmc_step)

inline int pad64(int x) { return (x + 63) & ~63; }

// delta of the output layer: only the realised outcome of the last pair carries the phase gradient
__global__ __launch_bounds__(256) void top_delta_kernel(const NetDims d, const int64_t M, const uint64_t *__restrict__ keys,
                                                        const float2 *__restrict__ g, float *__restrict__ delta, const int ld) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * ld) return;
    const int64_t i = e / ld;
    const int c = (int)(e - i * ld);
    const uint64_t key = keys[i];
    const int occ = naqs::phase_out_row(d, (int)((key >> d.qa[d.P - 1]) & 1ull) + 2 * (int)((key >> d.qb[d.P - 1]) & 1ull));
    delta[e] = c == occ ? g[i].y : 0.0f;
}

// vmc_grad + split_g + top_delta of the single-phase backward as ONE launch: element e = (sample i, output column c) of the
// output layer's delta; column 0 also writes g_i, its amplitude component and (i == 0) the pair (<E>, Var).  Same float32
// arithmetic as vmc_grad_kernel (naqs_grad.hip): g = (2 w Re(E_loc - <E>), -2 w Im(E_loc - <E>)), energy.py:328-329.
__global__ __launch_bounds__(256) void vmc_seed_kernel(const NetDims d, const int64_t M, const uint64_t *__restrict__ keys,
                                                       const double2 *__restrict__ eloc, const double *__restrict__ w,
                                                       const double *__restrict__ sums, float2 *__restrict__ g,
                                                       float *__restrict__ g_amp, float *__restrict__ delta, const int ld,
                                                       double *__restrict__ ev) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e == 0) {
        const double e_mean = sums[0] / sums[3];
        ev[0] = e_mean;
        ev[1] = sums[2] / sums[3] - e_mean * e_mean;
    }
    if (e >= M * ld) return;
    const int64_t i = e / ld;
    const int c = (int)(e - i * ld);
    const float m_re = (float)sums[0], m_im = (float)sums[1];
    const double2 el = eloc[i];
    const float two_w = 2.0f * (float)w[i];
    const float gx = ((float)el.x - m_re) * two_w, gy = -(((float)el.y - m_im) * two_w);
    const uint64_t key = keys[i];
    const int occ = naqs::phase_out_row(d, (int)((key >> d.qa[d.P - 1]) & 1ull) + 2 * (int)((key >> d.qb[d.P - 1]) & 1ull));
    delta[e] = c == occ ? gy : 0.0f;
    if (c == 0) { g[i] = make_float2(gx, gy); g_amp[i] = gx; }
}

// delta of the LAST hidden layer without a GEMM: the output layer's delta has one non-zero column per sample (the realised
// outcome's phase), so  delta[i][k] = [act[i][k] > 0] * g_i.y * W_top[occ_i][k]  — the single surviving term of the product
// grad_in_kernel would form over a 64-wide zero-padded delta (bit-identical: the other terms add +0.0).  Four columns per thread.
__global__ __launch_bounds__(256) void delta_below_top_kernel(const NetDims d, const int64_t M, const uint64_t *__restrict__ keys,
                                                              const float2 *__restrict__ g, const float *__restrict__ Wtop,
                                                              const float *__restrict__ act, const int Kp, float *__restrict__ dout) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k4 = Kp >> 2;
    if (e >= M * k4) return;
    const int64_t i = e / k4;
    const int k = (int)(e - i * k4) << 2;
    const uint64_t key = keys[i];
    const int occ = naqs::phase_out_row(d, (int)((key >> d.qa[d.P - 1]) & 1ull) + 2 * (int)((key >> d.qb[d.P - 1]) & 1ull));
    const float gy = g[i].y;
    const f32x4 w = *reinterpret_cast<const f32x4 *>(Wtop + (int64_t)occ * Kp + k);
    const f32x4 a = *reinterpret_cast<const f32x4 *>(act + i * Kp + k);
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = a[c] > 0.0f ? gy * w[c] : 0.0f;
    *reinterpret_cast<f32x4 *>(dout + i * Kp + k) = o;
}

// vmc_seed_kernel and delta_below_top_kernel as ONE launch (the training step's usual case: a hidden layer below the output
// layer): thread (sample i, column quad q) forms g_i itself from (E_loc_i, w_i, sums) — the same float32 expressions, so
// every value is the one the two launches produce — writes its four columns of the last hidden layer's delta, and the
// first threads of a sample's row also write the output layer's delta, g_i, its amplitude component and (<E>, Var).
// SUMS (naqs_vmc_step at small tables): the weighted sums the seeds need are formed HERE, by workgroup 0 — reduce_kernel's
// own arithmetic in reduce_kernel's own order (naqs_reduce.hpp: bit-identical) — instead of by a launch in front of this
// one; it stores them to sums_out and hands them to the other workgroups as eight tagged words (call tag << 32 | half a
// double; relaxed agent-scope store / polled load: value and flag are one word — §4.5's hand-over).  Every other workgroup
// has a higher index than the one it waits for.
template <bool SUMS>
__global__ __launch_bounds__(256) void vmc_seed_delta_kernel(const NetDims d, const int64_t M, const uint64_t *__restrict__ keys,
                                                             const double2 *__restrict__ eloc, const double *__restrict__ w,
                                                             const double *__restrict__ sums_in, float2 *__restrict__ g,
                                                             float *__restrict__ g_amp, float *__restrict__ delta, const int ld,
                                                             double *__restrict__ ev, const float *__restrict__ Wtop,
                                                             const float *__restrict__ act, const int Kp, float *__restrict__ dout,
                                                             double *__restrict__ sums_out, unsigned long long *__restrict__ words,
                                                             const uint32_t tag, const naqs::PollCtl *ctl) {
    __shared__ double s_red[4][naqs::RED_BLOCK / 64];
    __shared__ double s_sums[4];
    __shared__ uint32_t s_half[8];
    const double *sums = sums_in;
    // this thread's element (sample i, column quad q): everything it reads that does not depend on the sums is requested BEFORE
    // the wait for them (round 6: the loads used to start when the sums arrived — one more round trip on the step's chain)
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k4 = Kp >> 2;
    const bool mine = e < M * k4;
    const int64_t i = mine ? e / k4 : 0;
    const int q = (int)(e - i * k4), k = mine ? q << 2 : 0;
    const double2 el = eloc[i];
    const double wi = w[i];
    const uint64_t key = keys[i];
    const int occ = naqs::phase_out_row(d, (int)((key >> d.qa[d.P - 1]) & 1ull) + 2 * (int)((key >> d.qb[d.P - 1]) & 1ull));
    const f32x4 wt = *reinterpret_cast<const f32x4 *>(Wtop + (int64_t)occ * Kp + k);
    const f32x4 a = *reinterpret_cast<const f32x4 *>(act + i * Kp + k);
    if (SUMS) {
        if (blockIdx.x == 0) {
            naqs::weighted_sums_block<256>(M, w, eloc, s_red, s_sums);
            __syncthreads();
            if (threadIdx.x < 8) {
                const double v = s_sums[threadIdx.x >> 1];
                const uint32_t half = (threadIdx.x & 1) ? (uint32_t)__double2loint(v) : (uint32_t)__double2hiint(v);
                if (!(naqs::poll_drop(ctl, naqs::POLL_SEED_SUMS) && threadIdx.x == 0))
                    __hip_atomic_store(&words[threadIdx.x], ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (threadIdx.x < 4) sums_out[threadIdx.x] = s_sums[threadIdx.x];
            }
        } else {
            bool ok = true;
            if (threadIdx.x < 8) {
                unsigned long long word;
                ok = naqs::poll_tagged<2>(&words[threadIdx.x], tag, word, ctl, naqs::POLL_SEED_SUMS, threadIdx.x);
                s_half[threadIdx.x] = (uint32_t)word;
            }
            if (__syncthreads_or(!ok)) return;             // the wait ran out (naqs_poll.hpp): nothing of this workgroup is written
            if (threadIdx.x < 4) s_sums[threadIdx.x] = __hiloint2double((int)s_half[2 * threadIdx.x], (int)s_half[2 * threadIdx.x + 1]);
            __syncthreads();
        }
        sums = s_sums;
    }
    if (e == 0) {
        const double e_mean = sums[0] / sums[3];
        ev[0] = e_mean;
        ev[1] = sums[2] / sums[3] - e_mean * e_mean;
    }
    if (!mine) return;
    const float m_re = (float)sums[0], m_im = (float)sums[1];
    const float two_w = 2.0f * (float)wi;
    const float gx = ((float)el.x - m_re) * two_w, gy = -(((float)el.y - m_im) * two_w);
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = a[c] > 0.0f ? gy * wt[c] : 0.0f;
    *reinterpret_cast<f32x4 *>(dout + i * Kp + k) = o;
    for (int c = q; c < ld; c += k4) delta[i * ld + c] = c == occ ? gy : 0.0f;
    if (q == 0) { g[i] = make_float2(gx, gy); g_amp[i] = gx; }
}

// both columns of g [M][2] as contiguous vectors (aggregate-phase backward: one per set of blocks)
__global__ __launch_bounds__(256) void split_g2_kernel(const int64_t M, const float2 *__restrict__ g, float *__restrict__ g_amp,
                                                       float *__restrict__ g_ph) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < M) { const float2 v = g[i]; g_amp[i] = v.x; g_ph[i] = v.y; }
}

__global__ __launch_bounds__(256) void split_g_kernel(const int64_t M, const float2 *__restrict__ g, float *__restrict__ g_amp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < M) g_amp[i] = g[i].x;
}

// Weight gradients of ALL phase layers in one launch (they only need the deltas, which the grad_in chain has produced by
// then): job l:  Cpart_l[z][n][k] = sum_{i in slice z} P_l[i][n] Q_l[i][k];  Bpart_l[z][n] = sum_{i in slice z} P_l[i][n].
// (Three launches of 8-64 workgroups each, plus three reduce launches, were 8 + 13 + 9 + 3 x 5 us at M ~ 1 200.)
struct GradWJobs {
    int n;
    const float *P[MAXL], *Q[MAXL];
    int ldp[MAXL], ldq[MAXL], Np[MAXL], Kp[MAXL], N[MAXL], K[MAXL], slices[MAXL];
    int64_t rows_per_slice[MAXL];
    int block_end[MAXL];            // running total of (Np / TB) (Kp / TB) slices
    int64_t cpart_off[MAXL], bpart_off[MAXL];
    int64_t out_off[MAXL];          // dW_l in the flat gradient (db_l follows it)
    int64_t elem_end[MAXL];         // running total of N K + N (reduce kernel)
};

// (bid: the block's index among ALL jobs' blocks; Ps, Qs: [CH * LDT] floats of LDS each)
__device__ __forceinline__ void grad_w_body(const GradWJobs &J, const int64_t M, float *__restrict__ cpart_base,
                                            float *__restrict__ bpart_base, int bid, float *Ps, float *Qs) {
    int job = 0;
    while (job + 1 < J.n && bid >= J.block_end[job]) ++job;
    if (job > 0) bid -= J.block_end[job - 1];
    const int Np = J.Np[job], Kp = J.Kp[job], nbn = Np / TB, nbk = Kp / TB;
    const int z = bid / (nbn * nbk), rem = bid - z * (nbn * nbk);
    const int n0 = (rem / nbk) * TB, k0 = (rem % nbk) * TB;
    const float *__restrict__ P = J.P[job], *__restrict__ Q = J.Q[job];
    const int ldp = J.ldp[job], ldq = J.ldq[job];
    float *__restrict__ Cpart = cpart_base + J.cpart_off[job], *__restrict__ Bpart = bpart_base + J.bpart_off[job];
    const int64_t rows_per_slice = J.rows_per_slice[job];
    const int64_t i_beg = (int64_t)z * rows_per_slice, i_end = min(M, i_beg + rows_per_slice);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1, lm = lane & 15, lq = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
    // the next chunk's operands are fetched into registers while this chunk's MFMAs run (the kernel is a chain of
    // load -> barrier -> 32 MFMAs -> barrier per chunk: latency, not throughput)
    f32x4 vp[2], vq[2];
    auto fetch = [&](int64_t i0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u, r = e >> 4, c4 = e & 15;
            const int64_t i = i0 + r;
            vp[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; vq[u] = vp[u];
            if (i < i_end) {
                vp[u] = *reinterpret_cast<const f32x4 *>(P + i * ldp + n0 + 4 * c4);
                vq[u] = *reinterpret_cast<const f32x4 *>(Q + i * ldq + k0 + 4 * c4);
            }
        }
    };
    fetch(i_beg);
    for (int64_t i0 = i_beg; i0 < i_end; i0 += CH) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u, r = e >> 4, c4 = e & 15;
            *reinterpret_cast<f32x4 *>(Ps + r * LDT + 4 * c4) = vp[u];
            *reinterpret_cast<f32x4 *>(Qs + r * LDT + 4 * c4) = vq[u];
        }
        __syncthreads();
        if (i0 + CH < i_end) fetch(i0 + CH);
#pragma unroll
        for (int ks = 0; ks < CH / 4; ++ks) {
            const int kk = ks * 4 + lq;
            const float a0 = Ps[kk * LDT + wn * 32 + lm], a1 = Ps[kk * LDT + wn * 32 + 16 + lm];
            const float b0 = Qs[kk * LDT + wk * 32 + lm], b1 = Qs[kk * LDT + wk * 32 + 16 + lm];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
            bsum[0] += a0;
            bsum[1] += a1;
        }
        __syncthreads();
    }
    // C/D layout: column = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 32 + tn * 16 + 4 * lq + r, k = k0 + wk * 32 + tk * 16 + lm;
                Cpart[((int64_t)z * Np + n) * Kp + k] = acc[tn][tk][r];
            }
    if (k0 == 0 && wk == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v = bsum[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) Bpart[(int64_t)z * Np + n0 + wn * 32 + t * 16 + lane] = v;
        }
    }
}

// bid0: index of the launch's first block among all jobs' blocks (the launch may cover a sub-range of the jobs)
__global__ __launch_bounds__(256) void grad_w_kernel(const GradWJobs J, const int64_t M, float *__restrict__ cpart_base,
                                                     float *__restrict__ bpart_base, const int bid0) {
    __shared__ __attribute__((aligned(16))) float Ps[CH * LDT];
    __shared__ __attribute__((aligned(16))) float Qs[CH * LDT];
    grad_w_body(J, M, cpart_base, bpart_base, bid0 + (int)blockIdx.x, Ps, Qs);
}

// sum_{b < n} p[b * stride], added in the order b = 0, 1, ... (the result every reduction of this file has always had), the
// loads issued sixteen at a time so that the chain is not one memory round trip per term (eight until the first layer's
// weight gradient came in one slice per row tile: 40 slices at a training step's size; 32 at a time measured slower: 10.5 us against 9.2)
__device__ __forceinline__ float ordered_sum(const float *__restrict__ p, const int64_t stride, const int n) {
    constexpr int U = 16;
    float s = 0.0f;
    for (int b0 = 0; b0 < n; b0 += U) {
        float v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = b0 + j < n ? p[(int64_t)(b0 + j) * stride] : 0.0f;
#pragma unroll
        for (int j = 0; j < U; ++j)
            if (b0 + j < n) s += v[j];
    }
    return s;
}

// The whole gradient's fixed-order reductions in ONE launch — the per-pair block sets' workgroup partials (amp_reduce_kernel's
// sums: amplitude blocks, and the phase blocks of an aggregate-phase network) and the phase MLP's slices
// (grad_finish_kernel's sums), element by element in the flat state_dict layout — and, when asked, Adam's update of that
// element straight from the register (naqs_vmc_step): three launches of ~5 us each at training sizes become one.
struct GradFinish {
    int n_sets = 0;
    naqs::BlockReduceJob set[2];
    int64_t set_end[2] = {0, 0};       // running totals of the sets' elements
    int64_t set_out[2] = {0, 0};       // where each set starts in the flat gradient
    int64_t total = 0;                 // sets + MLP elements
};
// Round 6 — the launch's first workgroups, ONE per leading amplitude pair (naqs_vmc_step: the pairs the next sampler call's
// single busy workgroup reads).  Such a workgroup finishes ALL of its pair's gradient elements (up to HEAD_EPT per thread, the
// loads of all of them in flight together; each element's sum in ordered_sum's order: the same bits), applies Adam's update,
// keeps the updated parameters in LDS and packs the pair's matrix-core fragments from them (naqs_pack.hpp: the arithmetic of
// pack_amp_mfma_body on the same values) — so that the amplitude blocks' whole re-pack can wait for the sampler's first launch
// to host it (PACK_DEFER) instead of being a launch between the update and the sampler.  The pair's scales need every one of
// its parameters, which is why a pair is one workgroup's; 773 parameters for pair 3 of 64-unit blocks.
constexpr int HEAD_EPT = 4, HEAD_MAX_PAIRS = 4;
struct HeadPairs {
    int pairs = 0;                         // 0: no such workgroups
    int Ha = 0, nout = 0;
    int64_t off[HEAD_MAX_PAIRS] = {0, 0, 0, 0};      // pair p's first element within set 0
    int64_t skip = 0;                      // the elements of set 0 these workgroups cover: [0, skip)
    unsigned short *wamp = nullptr;
};
__device__ __forceinline__ void head_pair_finish(const GradFinish &F, float *__restrict__ grad, const naqs::AdamArgs &A, const HeadPairs &H,
                                                 const int p) {
    __shared__ float s_new[HEAD_EPT * 256];
    const int tid = threadIdx.x;
    const int nin = p == 0 ? 1 : 2 * p;
    const int cnt = H.Ha * nin + H.Ha + H.nout * H.Ha + H.nout;            // <= HEAD_EPT * 256 (the host checked)
    const float *__restrict__ part = F.set[0].partial + H.off[p];
    const int64_t stride = F.set[0].stride;
    const int np = F.set[0].n_partials;
    constexpr int U = 8;                                   // (sixteen: 73 registers, six waves per SIMD for every workgroup of the launch — 12.1 us against 11.0)
    float s[HEAD_EPT], p0[HEAD_EPT], m0[HEAD_EPT], v0[HEAD_EPT];
#pragma unroll
    for (int k = 0; k < HEAD_EPT; ++k) {                   // (the update's own operands: requested first, used last)
        s[k] = 0.0f;
        const int64_t o = F.set_out[0] + H.off[p] + min(tid + 256 * k, cnt - 1);
        p0[k] = A.p[o]; m0[k] = A.m[o]; v0[k] = A.v[o];
    }
    for (int b0 = 0; b0 < np; b0 += U) {
        float v[HEAD_EPT][U];
#pragma unroll
        for (int k = 0; k < HEAD_EPT; ++k)
#pragma unroll
            for (int j = 0; j < U; ++j) v[k][j] = (tid + 256 * k < cnt && b0 + j < np) ? part[(int64_t)(b0 + j) * stride + tid + 256 * k] : 0.0f;
#pragma unroll
        for (int k = 0; k < HEAD_EPT; ++k)
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (b0 + j < np) s[k] += v[k][j];
    }
#pragma unroll
    for (int k = 0; k < HEAD_EPT; ++k) {
        const int e = tid + 256 * k;
        if (e < cnt) {
            const int64_t o = F.set_out[0] + H.off[p] + e;
            grad[o] = s[k];
            s_new[e] = naqs::adam_update_loaded(A, o, s[k], p0[k], m0[k], v0[k]);
        }
    }
    __syncthreads();
    naqs::pack_amp_mfma_src(s_new, H.Ha, H.nout, H.wamp, p, 0, 1);
}

// (e0, e1: the elements this launch covers — all of them, or the block sets' and the MLP's as two launches on two streams)
__global__ __launch_bounds__(256) void grad_finish_kernel(const GradFinish F, const GradWJobs J, const float *__restrict__ cpart_base,
                                                          const float *__restrict__ bpart_base, float *__restrict__ grad,
                                                          const naqs::AdamArgs A, const int64_t e0, const int64_t e1, const HeadPairs H) {
    if ((int)blockIdx.x < H.pairs) {                       // (workgroup-uniform)
        head_pair_finish(F, grad, A, H, (int)blockIdx.x);
        return;
    }
    int64_t e = e0 + H.skip + (int64_t)((int)blockIdx.x - H.pairs) * 256 + threadIdx.x;
    if (e >= e1) return;
    const int64_t sets_total = F.n_sets > 0 ? F.set_end[F.n_sets - 1] : 0;
    float s = 0.0f;
    int64_t o;
    if (e < sets_total) {
        const int k = (F.n_sets > 1 && e >= F.set_end[0]) ? 1 : 0;
        if (k) e -= F.set_end[0];
        s = ordered_sum(F.set[k].partial + e, F.set[k].stride, F.set[k].n_partials);
        o = F.set_out[k] + e;
    } else {
        e -= sets_total;
        int job = 0;
        while (job + 1 < J.n && e >= J.elem_end[job]) ++job;
        if (job > 0) e -= J.elem_end[job - 1];
        const int N = J.N[job], K = J.K[job], Np = J.Np[job], Kp = J.Kp[job], slices = J.slices[job];
        const float *Cpart = cpart_base + J.cpart_off[job], *Bpart = bpart_base + J.bpart_off[job];
        if (e < (int64_t)N * K) {
            const int n = (int)(e / K), k = (int)(e - (int64_t)n * K);
            s = ordered_sum(Cpart + (int64_t)n * Kp + k, (int64_t)Np * Kp, slices);
        } else {
            const int n = (int)(e - (int64_t)N * K);
            s = ordered_sum(Bpart + n, Np, slices);
        }
        o = J.out_off[job] + e;          // db_l follows dW_l in the flat gradient: element N K + n
    }
    grad[o] = s;
    // (the update's operands requested ahead of the partial sums, as the launch's first workgroups do: measured neutral here)
    if (A.p != nullptr) naqs::adam_update(A, o, s);
}

// Dout[i][k] = [In[i][k] > 0] * sum_n D[i][n] W[n][k]
// (bx, by: sample tile and column tile; Ds: [TBM * LDD], Ws: [CH * LDT] floats of LDS)
// TBM = 64 or 32 samples per tile (a wave owns TBM / 2 of them x 32 columns).  The kernel is a chain of Np / CH chunks of
// load -> barrier -> MFMAs -> barrier; at a training step's ~1 200 rows the 64-row form is 160 workgroups of 16 chunks x 32
// exact-f32 MFMAs each (20.8 us as a launch of its own), the 32-row form twice the workgroups of half the MFMAs per chunk.
// Every output element accumulates over n in the same order in both: identical results.
// w0 (backward_mega_kernel, small tables): the tile's delta never leaves the workgroup — it is the P operand of the FIRST
// layer's weight gradient dW0[n][j] = sum_i delta[i][n] x[i][j], whose slice `bx` (= this tile's rows) the workgroup forms
// right here, chunk by chunk of 32 rows with grad_w_body's own loop (same operands, same MFMA order: the numbers of
// grad_w_kernel on job 0 with rows_per_slice = TBM), instead of a launch of its own behind this one.
struct W0Fuse { const float *x; int x_ld; float *cpart, *bpart; int Kp0; };

template <int TBM>
__device__ __forceinline__ void grad_in_body(const float *__restrict__ D, const float *__restrict__ W,
                                             const float *__restrict__ In, const int64_t M, const int Np, const int Kp,
                                             float *__restrict__ Dout, const int bx, const int by, float *Ds, float *Ws,
                                             const W0Fuse *w0 = nullptr) {
    static_assert(TBM == 64 || TBM == 32, "two or one 16-row MFMA tiles per wave");
    constexpr int RA = TBM / 32;                        // 16-row tiles of a wave
    constexpr int UD = TBM / 32;                        // float4 of the delta tile per thread and chunk
    const int64_t i0 = (int64_t)bx * TBM;
    const int k0 = by * TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wk = wave & 1, lm = lane & 15, lq = lane >> 4;
    f32x4 acc[RA][2];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // next chunk's tiles in registers while this chunk's MFMAs run (see grad_w_kernel)
    f32x4 vd[UD], vw[2];
    auto fetch = [&](int nn0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u;
            if (u < UD) {
                const int r = e >> 3, c4 = e & 7;
                const int64_t i = i0 + r;
                vd[u < UD ? u : 0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (i < M) vd[u < UD ? u : 0] = *reinterpret_cast<const f32x4 *>(D + i * Np + nn0 + 4 * c4);
            }
            const int rw = e >> 4, cw = e & 15;
            vw[u] = *reinterpret_cast<const f32x4 *>(W + (int64_t)(nn0 + rw) * Kp + k0 + 4 * cw);
        }
    };
    fetch(0);
    for (int nn0 = 0; nn0 < Np; nn0 += CH) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u;
            if (u < UD) {   // delta tile: TBM samples x 32 n
                const int r = e >> 3, c4 = e & 7;
                float2 *dst = reinterpret_cast<float2 *>(Ds + r * LDD + 4 * c4);     // LDD even: 8-byte aligned
                dst[0] = make_float2(vd[u < UD ? u : 0][0], vd[u < UD ? u : 0][1]);
                dst[1] = make_float2(vd[u < UD ? u : 0][2], vd[u < UD ? u : 0][3]);
            }
            {   // weight tile: 32 n x 64 k
                const int r = e >> 4, c4 = e & 15;
                *reinterpret_cast<f32x4 *>(Ws + r * LDT + 4 * c4) = vw[u];
            }
        }
        __syncthreads();
        if (nn0 + CH < Np) fetch(nn0 + CH);
#pragma unroll
        for (int ks = 0; ks < CH / 4; ++ks) {
            const int kk = ks * 4 + lq;
            const float b0 = Ws[kk * LDT + wk * 32 + lm], b1 = Ws[kk * LDT + wk * 32 + 16 + lm];
#pragma unroll
            for (int a = 0; a < RA; ++a) {
                const float av = Ds[(wi * (TBM / 2) + a * 16 + lm) * LDD + kk];
                acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[a][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (w0 == nullptr) {
#pragma unroll
        for (int ti = 0; ti < RA; ++ti)
#pragma unroll
            for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t i = i0 + wi * (TBM / 2) + ti * 16 + 4 * lq + r;
                    const int k = k0 + wk * 32 + tk * 16 + lm;
                    if (i < M) Dout[i * Kp + k] = In[i * Kp + k] > 0.0f ? acc[ti][tk][r] : 0.0f;
                }
        return;
    }
    // (the main loop ended with a barrier: the LDS tiles are free)
    float *Ps = Ds, *Qs = Ds + CH * LDT;                  // [CH][LDT] each: masked delta rows x 64 columns, input rows x Kp0
    const int wn = wave >> 1;
    f32x4 acc2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < TBM / CH; ++c) {                  // chunk c: the tile's rows 32 c .. 32 c + 31
        if (c > 0) __syncthreads();
#pragma unroll
        for (int ti = 0; ti < RA; ++ti)
#pragma unroll
            for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wi * (TBM / 2) + ti * 16 + 4 * lq + r;              // row of the tile this value belongs to
                    if (row / CH == c) {
                        const int64_t i = i0 + row;
                        const int k = k0 + wk * 32 + tk * 16 + lm;
                        Ps[(row - c * CH) * LDT + wk * 32 + tk * 16 + lm] = (i < M && In[i * Kp + k] > 0.0f) ? acc[ti][tk][r] : 0.0f;
                    }
                }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u, r = e >> 4, c4 = e & 15;
            const int64_t i = i0 + c * CH + r;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (i < M && 4 * c4 < w0->Kp0) v = *reinterpret_cast<const f32x4 *>(w0->x + i * w0->x_ld + 4 * c4);
            *reinterpret_cast<f32x4 *>(Qs + r * LDT + 4 * c4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < CH / 4; ++ks) {
            const int kk = ks * 4 + lq;
            const float a0 = Ps[kk * LDT + wn * 32 + lm], a1 = Ps[kk * LDT + wn * 32 + 16 + lm];
            const float b0 = Qs[kk * LDT + wk * 32 + lm], b1 = Qs[kk * LDT + wk * 32 + 16 + lm];
            acc2[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc2[0][0], 0, 0, 0);
            acc2[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc2[0][1], 0, 0, 0);
            acc2[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc2[1][0], 0, 0, 0);
            acc2[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc2[1][1], 0, 0, 0);
            bsum[0] += a0;
            bsum[1] += a1;
        }
    }
    // slice bx of job 0's partials: rows n = this tile's delta columns, columns k = the Kp0 (padded) inputs
    const int Kp0 = w0->Kp0;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = k0 + wn * 32 + tn * 16 + 4 * lq + r, k = wk * 32 + tk * 16 + lm;
                if (k < Kp0) w0->cpart[((int64_t)bx * Kp + n) * Kp0 + k] = acc2[tn][tk][r];
            }
    if (wk == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v = bsum[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) w0->bpart[(int64_t)bx * Kp + k0 + wn * 32 + t * 16 + lane] = v;
        }
    }
}

__global__ __launch_bounds__(256) void grad_in_kernel(const float *__restrict__ D, const float *__restrict__ W,
                                                      const float *__restrict__ In, const int64_t M, const int Np, const int Kp,
                                                      float *__restrict__ Dout) {
    __shared__ __attribute__((aligned(16))) float Ds[TB * LDD];
    __shared__ __attribute__((aligned(16))) float Ws[CH * LDT];
    grad_in_body<TB>(D, W, In, M, Np, Kp, Dout, (int)blockIdx.x, (int)blockIdx.y, Ds, Ws);
}

// The three independent pieces of the backward pass below the seeds, as ONE launch (the training step's usual network: two
// hidden phase layers, 64-unit amplitude blocks): blocks [0, n_gin) are grad_in_kernel's (the first hidden layer's delta),
// the next n_amp amp_backward_kernel's (workgroup x pair) and the rest grad_w_kernel's for the layers whose deltas exist
// already (all but the first: its delta is what the grad_in blocks are producing; that job gets its own small launch
// afterwards).  Each piece alone leaves most of the chip idle (240 / 300 / 630 workgroups of a 15-18 us latency chain);
// together they take about as long as the longest.  Same device functions, same operands: same numbers.
// (Measured and rejected: the first layer's blocks inside this launch too, waiting on a device-scope count of the finished
// grad_in blocks — the fences and the polling cost 12-18 us per step more than the 7 us launch they replace.)
struct MegaArgs {
    int n_gin, gin_tiles_i, gin_tbm, n_amp, amp_wgs;      // gin_tbm: 64 or 32 samples per grad_in tile
    int fuse_w0;                                          // the grad_in blocks also form the first layer's weight gradient
    W0Fuse w0;
    const float *D, *W, *In;
    float *Dout;
    int Np, Kp;
    int64_t M;
    float *cpart, *bpart;
    int gw_bid0;
    const float *amp_w, *g_amp;
    const uint64_t *keys;
    float *amp_partial;
    int64_t amp_stride;
};
__global__ __launch_bounds__(256) void backward_mega_kernel(const MegaArgs A, const NetDims d, const GradWJobs J,
                                                            const naqs::ampbw::AmpSrc src) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    if (bid < A.n_gin) {
        const W0Fuse *w0 = A.fuse_w0 ? &A.w0 : nullptr;
        if (A.gin_tbm == 32) grad_in_body<32>(A.D, A.W, A.In, A.M, A.Np, A.Kp, A.Dout, bid % A.gin_tiles_i, bid / A.gin_tiles_i, smem, smem + TB * LDD, w0);
        else grad_in_body<TB>(A.D, A.W, A.In, A.M, A.Np, A.Kp, A.Dout, bid % A.gin_tiles_i, bid / A.gin_tiles_i, smem, smem + TB * LDD, w0);
        return;
    }
    bid -= A.n_gin;
    if (bid < A.n_amp) {
        const int wg = bid % A.amp_wgs, n = bid / A.amp_wgs;
        naqs::ampbw::pair_dispatch(n, d, A.amp_w, A.M, A.keys, A.g_amp, A.amp_partial + (int64_t)wg * A.amp_stride + src.off[n], smem, 0,
                                   wg, A.amp_wgs);
        return;
    }
    bid -= A.n_amp;
    grad_w_body(J, A.M, A.cpart, A.bpart, A.gw_bid0 + bid, smem, smem + CH * LDT);
}

struct TrainLayout {            // carve-up of net->d_train for `cap` rows
    size_t x, act[MAXL], delta[MAXL], top, g_amp, cpart, bpart, total;
    int x_ld, act_ld[MAXL], top_ld, max_ld;
    int64_t cpart_floats;
};

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

TrainLayout train_layout(const naqs_net *net, int64_t cap) {
    const NetDims &d = net->dims;
    TrainLayout L{};
    size_t off = 0;
    const int H = d.n_lin - 1;
    L.x_ld = pad64(net->phase_K[0]);
    L.x = off; off = up256(off + (size_t)cap * L.x_ld * sizeof(float));
    L.max_ld = L.x_ld;
    for (int l = 0; l < H; ++l) {
        L.act_ld[l] = pad64(net->phase_N[(size_t)l]);
        L.max_ld = std::max(L.max_ld, L.act_ld[l]);
        L.act[l] = off; off = up256(off + (size_t)cap * L.act_ld[l] * sizeof(float));
    }
    for (int b = 0; b < H; ++b) { L.delta[b] = off; off = up256(off + (size_t)cap * L.max_ld * sizeof(float)); }   // delta of hidden layer b
    L.top_ld = pad64(net->phase_N[(size_t)H]);
    L.top = off; off = up256(off + (size_t)cap * L.top_ld * sizeof(float));
    L.g_amp = off; off = up256(off + (size_t)cap * sizeof(float));
    // GEMM partials: at most ~2 x CU-count workgroups per launch -> slices * blocks <= 512 (+ one slice minimum)
    int64_t worst = 0, worst_b = 0;                 // all layers' partials live at once (one launch computes them all)
    for (int l = 0; l <= H; ++l) {
        const int64_t Np = pad64(net->phase_N[(size_t)l]), Kp = pad64(net->phase_K[(size_t)l]);
        const int64_t blocks = (Np / TB) * (Kp / TB);
        int64_t slices = std::max<int64_t>(1, 512 / blocks);
        if (l == 0) slices = std::max<int64_t>(slices, W0_FUSE_MAX_ROWS / 32);      // one slice per 32-row tile of the fused form
        worst += slices * Np * Kp;
        worst_b += slices * Np;
    }
    L.cpart_floats = worst;
    L.cpart = off; off = up256(off + (size_t)worst * sizeof(float));
    L.bpart = off; off = up256(off + (size_t)worst_b * sizeof(float));
    L.total = off;
    return L;
}

int ensure_train_scratch(naqs_net *net, int64_t M) {
    if (M <= net->train_cap && net->d_train) return NAQS_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (net->d_train) (void)hipFree(net->d_train);
    net->d_train = nullptr; net->train_cap = 0;
    const int64_t cap = std::max<int64_t>(1024, M + M / 4);
    const TrainLayout L = train_layout(net, cap);
    HIP_TRY(hipMalloc(&net->d_train, L.total));
    HIP_TRY(hipMemset(net->d_train, 0, L.total));          // padding columns of x stay zero for good
    HIP_TRY(hipDeviceSynchronize());                       // (null-stream fill: finished before the caller's non-blocking stream writes there)
    net->train_cap = cap;
    return NAQS_OK;
}

size_t wb_offset(const naqs_net *net, int l) {
    size_t off = 0;
    for (int q = 0; q < l; ++q) off += (size_t)pad64(net->phase_N[(size_t)q]) * pad64(net->phase_K[(size_t)q]);
    return off;
}

}  // namespace

// called by naqs_net_set_weights (naqs_logpsi.hip): row-major padded copies of the phase weights for grad_in_kernel
// phase weights, row-major and zero-padded to multiples of 64 in both dimensions (Wb_l [Np][Kp]): what the packing
// launch of naqs_net_set_weights has to copy (layer 0 has no delta to propagate)
int naqs::net_backward_pack_jobs(naqs_net *net, naqs::WbPackJobs *jobs) {
    const NetDims &d = net->dims;
    if (!net->d_wb) {
        HIP_TRY(hipMalloc((void **)&net->d_wb, wb_offset(net, d.n_lin) * sizeof(float)));
    }
    jobs->n = 0;
    for (int l = 1; l < d.n_lin; ++l) {
        const int i = jobs->n++;
        jobs->src_off[i] = net->phase_src_off[(size_t)l];
        jobs->N[i] = net->phase_N[(size_t)l]; jobs->K[i] = net->phase_K[(size_t)l];
        jobs->Np[i] = pad64(jobs->N[i]); jobs->Kp[i] = pad64(jobs->K[i]);
        jobs->dst[i] = net->d_wb + wb_offset(net, l);
    }
    return NAQS_OK;
}

NAQS_API int naqs_net_train_forward(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, float *logpsi_dev, void *stream) {
    if (!net || M < 0 || (M > 0 && (!keys_dev || !logpsi_dev))) return NAQS_ERR_INVALID;
    if (!net->have_weights) return NAQS_ERR_INVALID;
    if (M == 0) return NAQS_OK;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    if (net->aggregate) {                                   // the per-pair blocks are recomputed by the backward pass: nothing to keep
        naqs::ElocFeed none{};
        return naqs::net_logpsi_impl(net, M, keys_dev, logpsi_dev, stream, none, naqs::PhaseSave{});
    }
    st = ensure_train_scratch(net, M);
    if (st != NAQS_OK) return st;
    const TrainLayout L = train_layout(net, net->train_cap);
    char *base = static_cast<char *>(net->d_train);
    naqs::PhaseSave save;
    save.x = reinterpret_cast<float *>(base + L.x);
    save.x_ld = L.x_ld;
    for (int l = 0; l + 1 < net->dims.n_lin; ++l) {
        save.act[l] = reinterpret_cast<float *>(base + L.act[l]);
        save.act_ld[l] = L.act_ld[l];
    }
    naqs::ElocFeed none{};
    return naqs::net_logpsi_impl(net, M, keys_dev, logpsi_dev, stream, none, save);
}

NAQS_API int naqs_net_train_forward_eloc(naqs_net_t *net, naqs_ham_t *ham, int64_t M, const uint64_t *keys_dev,
                                         const double *w_dev, float *logpsi_dev, double *eloc_dev, double *out4_dev,
                                         void *stream) {
    if (!net || !ham || M < 0 || (w_dev == nullptr) != (out4_dev == nullptr)) return NAQS_ERR_INVALID;
    if (M > 0 && (!keys_dev || !logpsi_dev || !eloc_dev)) return NAQS_ERR_INVALID;
    if (!net->have_weights) return NAQS_ERR_INVALID;
    if (naqs::ham_device(ham) != net->device) return NAQS_ERR_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    if (M == 0) {
        if (out4_dev) HIP_TRY(hipMemsetAsync(out4_dev, 0, 4 * sizeof(double), s));
        return NAQS_OK;
    }
    naqs::PhaseSave save;
    if (!net->aggregate) {
        st = ensure_train_scratch(net, M);
        if (st != NAQS_OK) return st;
        const TrainLayout L = train_layout(net, net->train_cap);
        char *base = static_cast<char *>(net->d_train);
        save.x = reinterpret_cast<float *>(base + L.x);
        save.x_ld = L.x_ld;
        for (int l = 0; l + 1 < net->dims.n_lin; ++l) {
            save.act[l] = reinterpret_cast<float *>(base + L.act[l]);
            save.act_ld[l] = L.act_ld[l];
        }
    }
    naqs::ElocFeed feed{};
    st = naqs::eloc_begin(ham, M, s, &feed);
    if (st != NAQS_OK) return st;
    st = naqs::net_logpsi_impl(net, M, keys_dev, logpsi_dev, stream, feed, save);     // also fills the E_loc tables
    if (st != NAQS_OK) return st;
    return naqs::eloc_main(ham, M, feed, eloc_dev, w_dev, out4_dev, s);
}

// The training step's one host synchronisation: (M, overflow) of the sampler, published to mapped host memory together with
// the call's sequence number as soon as the last level's size is known (naqs_sample.hip: publish_info).  The host polls
// those words instead of waiting for the stream (hipStreamSynchronize goes to sleep on the queue's completion signal after a
// short spin; waking up costs tens of microseconds during which the GPU has nothing to do): it sees them about a
// microsecond after the store and queues the forward pass while the sampler's last launches are still running.
// NAQS_SPIN_WAIT=0: wait for the stream instead.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#else
    std::this_thread::yield();
#endif
}
struct AfterSampler {          // what the caller queues behind the sampler's launches before the host starts waiting for M
    virtual int operator()() = 0;  // (the sampler's finish job is still pending — net->fin_job — and may ride in what is queued here)
    virtual ~AfterSampler() = default;
};
static int sample_and_wait(naqs_net_t *net, int64_t n_samples, uint64_t seed, int64_t max_unique, uint64_t *keys_dev,
                           int64_t *counts_dev, float *probs_dev, double *weights_dev, hipStream_t s, int64_t out[2],
                           int64_t *info_dev = nullptr, AfterSampler *after = nullptr) {
    // info_dev: where the sampler leaves its plain (M, overflow) words on the device (default: the handle's own two words)
    static const bool spin = [] { const char *e = getenv("NAQS_SPIN_WAIT"); return !e || atoi(e) != 0; }();
    int st0 = naqs::net_info_alloc(net);
    if (st0 != NAQS_OK) return st0;
    const int64_t seq = ++net->info_seq;
    const bool prof = net->prof_samp.armed();              // (bench.py's train_step.sampler_us: every stride-th step)
    if (prof) { int stp = net->prof_samp.begin(s); if (stp != NAQS_OK) return stp; }
    // two runs per GPU: the sampler takes the device's look-back turn before its level launches; it is given back here, when M is
    // known (or this call fails) — naqs_sample.hip
    struct Turn {
        naqs_net_t *n;
        explicit Turn(naqs_net_t *net_) : n(net_) { n->turn_caller_ends = true; }
        ~Turn() { n->turn_caller_ends = false; naqs::lookback_turn_end(n); }
    } turn(net);
    net->hold_finish = after != nullptr;
    int st = naqs::net_sample_early(net, n_samples, seed, max_unique, keys_dev, counts_dev, probs_dev, weights_dev,
                                    info_dev ? info_dev : net->d_info2, s,
                                    net->d_info_alias, seq);
    net->hold_finish = false;
    if (st != NAQS_OK) return st;
    if (prof) { int stp = net->prof_samp.end(s); if (stp != NAQS_OK) return stp; }
    if (after) {
        st = (*after)();
        const int st2 = naqs::net_sample_finish_flush(net, s);     // (nobody hosted the finish job: a launch of its own)
        if (st != NAQS_OK) return st;
        if (st2 != NAQS_OK) return st2;
    }
    volatile int64_t *h = net->h_info;
    if (spin) {
        // bounded: after ~2 s of polling (a sampler call is < 1 ms) the wait falls back to the stream's own completion signal
        const auto t0 = std::chrono::steady_clock::now();
        for (uint64_t it = 1; h[2] != seq; ++it) {
            if ((it & 0x3FFF) == 0) {                      // every ~0.2 ms: is the stream still alive?
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) return NAQS_ERR_HIP;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { HIP_TRY(hipStreamSynchronize(s)); break; }
            }
            cpu_relax();
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    } else {
        HIP_TRY(hipStreamSynchronize(s));
    }
    { const int pc = naqs::poll_check(net->poll); if (pc != NAQS_OK) return pc; }      // a look-back wait that gave up (naqs_poll.hpp)
    if (h[2] != seq) return NAQS_ERR_HIP;                   // the stream drained and nothing was published
    out[0] = h[0];
    out[1] = h[1];
    return NAQS_OK;
}

NAQS_API int naqs_vmc_sample_forward_eloc(naqs_net_t *net, naqs_ham_t *ham, int64_t n_samples, uint64_t seed, int64_t max_unique,
                                          uint64_t *keys_dev, int64_t *counts_dev, float *probs_dev, double *weights_dev,
                                          float *logpsi_dev, double *eloc_dev, double *out4_dev, int64_t *info_dev,
                                          int64_t info_host[2], void *stream) {
    if (!net || !ham || !weights_dev || !logpsi_dev || !eloc_dev || !out4_dev || !info_dev || !info_host) return NAQS_ERR_INVALID;
    if (!net->have_weights) return NAQS_ERR_INVALID;              // the forward needs the phase layers packed too
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // (M, overflow) land in host memory the sampler writes to directly (mapped, coherent): no copy launch between the sampler
    // and the synchronisation, nothing but the synchronisation between the sampler and the forward pass
    // (info_dev receives the same (M, overflow) on the device, in stream order: the sampler writes it)
    st = sample_and_wait(net, n_samples, seed, max_unique, keys_dev, counts_dev, probs_dev, weights_dev, s, info_host, info_dev);
    if (st != NAQS_OK) return st;
    if (info_host[1] != 0 || info_host[0] <= 0) {
        // nothing was evaluated (overflow, or an empty draw): the sums read as an empty table, not as whatever was there
        HIP_TRY(hipMemsetAsync(out4_dev, 0, 4 * sizeof(double), s));
        return NAQS_OK;
    }
    return naqs_net_train_forward_eloc(net, ham, info_host[0], keys_dev, weights_dev, logpsi_dev, eloc_dev, out4_dev, stream);
}

// seeds: nullptr = g_dev holds the loss gradient (naqs_net_train_backward); else the loss gradient is formed here from
// (E_loc, w, sums) together with its amplitude column and the output delta (naqs_net_train_backward_vmc, single-phase only)
struct VmcSeeds { const double *eloc, *w, *sums; float *g_out; double *ev; bool form_sums = false; };   // form_sums: `sums` is an OUTPUT of the seed kernel
static int launch_grad_finish(const GradFinish &F, const GradWJobs &J, const float *cpart, const float *bpart, float *grad_dev,
                              const naqs::AdamArgs *adam, hipStream_t s, int64_t e0 = 0, int64_t e1 = -1, const HeadPairs *hp = nullptr) {
    if (e1 < 0) e1 = F.total;
    if (e1 <= e0) return NAQS_OK;
    const HeadPairs H = hp ? *hp : HeadPairs{};
    if (H.pairs > 0 && (e0 != 0 || adam == nullptr || H.skip > e1)) return NAQS_ERR_INVALID;
    NAQS_KLAUNCH(grad_finish_kernel, dim3((unsigned)(H.pairs + (e1 - e0 - H.skip + 255) / 256)), dim3(256), 0, s, F, J, cpart, bpart, grad_dev,
                       adam ? *adam : naqs::AdamArgs{}, e0, e1, H);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

// adam: nullptr = the gradient only; else the launch that finishes the gradient also applies Adam's update (naqs_vmc_step)
static int train_backward_impl(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, const float *g_dev, float *grad_dev, void *stream,
                               const VmcSeeds *seeds, const naqs::AdamArgs *adam = nullptr) {
    if (!net || M < 0 || !grad_dev || (M > 0 && (!keys_dev || !g_dev))) return NAQS_ERR_INVALID;
    {   // (the row-major weight copies this pass reads belong to the phase share of the re-pack)
        const int stj = naqs::net_flush_pack(net, reinterpret_cast<hipStream_t>(stream));
        if (stj != NAQS_OK) return stj;
    }
    if (!net->have_weights || !net->have_wb) return NAQS_ERR_INVALID;
    if (!net->aggregate && (M > net->train_cap || !net->d_train)) return NAQS_ERR_INVALID;   // naqs_net_train_forward of the same batch comes first
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (M == 0) {
        if (adam) return NAQS_ERR_INVALID;
        HIP_TRY(hipMemsetAsync(grad_dev, 0, (size_t)net->n_params * sizeof(float), s));
        return NAQS_OK;
    }
    GradFinish F;
    if (net->aggregate) {
        // both sets of per-pair blocks through the same backward kernel: amplitude blocks on g[:, 0], phase blocks (raw
        // outputs, no conditional) on g[:, 1]; the forward scratch ([2 P][cap] floats, free by now) holds the two columns
        if (M > net->cap_M || !net->d_scratch) return NAQS_ERR_INVALID;
        if (net->dims.Ha == net->dph.Ha && net->dims.P == net->dph.P && (naqs::env_int("NAQS_AGG_MERGE", 7) & 2)) {
            // one launch for both sets, the two columns of g read where they are
            st = naqs::net_blocks_backward2(net, M, keys_dev, g_dev, g_dev + 1, 2, F.set, s);
            if (st != NAQS_OK) return st;
        } else {
            float *g_amp = net->d_scratch, *g_ph = net->d_scratch + M;
            NAQS_KLAUNCH(split_g2_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, M, reinterpret_cast<const float2 *>(g_dev),
                               g_amp, g_ph);
            HIP_TRY(hipGetLastError());
            st = naqs::net_blocks_backward(net, net->dims, net->d_w, net->amp_src_off, net->amp_params, M, keys_dev, g_amp, grad_dev, 0, s,
                                           &F.set[0], 0);
            if (st != NAQS_OK) return st;
            st = naqs::net_blocks_backward(net, net->dph, net->d_wph, net->ph_src_off, net->ph_params, M, keys_dev, g_ph,
                                           grad_dev + net->amp_params, 1, s, &F.set[1], 1);
            if (st != NAQS_OK) return st;
        }
        F.n_sets = 2;
        F.set_end[0] = net->amp_params; F.set_end[1] = net->amp_params + net->ph_params;
        F.set_out[0] = 0; F.set_out[1] = net->amp_params;
        F.total = F.set_end[1];
        return launch_grad_finish(F, GradWJobs{}, nullptr, nullptr, grad_dev, adam, s);
    }
    const NetDims &d = net->dims;
    const TrainLayout L = train_layout(net, net->train_cap);
    char *base = static_cast<char *>(net->d_train);
    float *x = reinterpret_cast<float *>(base + L.x);
    float *top = reinterpret_cast<float *>(base + L.top);
    float *g_amp = reinterpret_cast<float *>(base + L.g_amp);
    float *cpart = reinterpret_cast<float *>(base + L.cpart), *bpart = reinterpret_cast<float *>(base + L.bpart);
    const float2 *g2 = reinterpret_cast<const float2 *>(g_dev);

    // amplitude blocks.  NAQS_TRAIN_SIDE_STREAM=1 runs them on a second stream beside the phase MLP's backward (independent:
    // disjoint parts of the gradient, own scratch; 37 us and 59 us of latency-bound launches at M ~ 1 200).  Measured, three
    // interleaved rounds on one box: N2 0.466 / 0.462 / 0.493 ms per step with it, 0.440 / 0.469 / 0.450 without; H2O 0.395 vs
    // 0.387-0.399 — the fork / join events cost what the overlap wins, as they did for the forward pass in round 1.  Off.
    const bool side = naqs::env_int("NAQS_TRAIN_SIDE_STREAM", 0) == 1;
    hipStream_t sa = s;
    if (side) {
        if (!net->side_stream) HIP_TRY(hipStreamCreateWithFlags(&net->side_stream, hipStreamNonBlocking));
        if (!net->ev_fork) HIP_TRY(hipEventCreateWithFlags(&net->ev_fork, hipEventDisableTiming));
        if (!net->ev_join) HIP_TRY(hipEventCreateWithFlags(&net->ev_join, hipEventDisableTiming));
        sa = net->side_stream;
        HIP_TRY(hipEventRecord(net->ev_fork, s));                 // g (and the keys) are ready on the caller's stream
        HIP_TRY(hipStreamWaitEvent(sa, net->ev_fork, 0));
    }
    const int H = d.n_lin - 1;
    const bool seed_delta = seeds != nullptr && H >= 1;   // the seeds and the last hidden layer's delta from one launch
    if (seed_delta) {
        const int Kp = pad64(net->phase_K[(size_t)H]);
        if (seeds->form_sums) {
            if (++net->sums_seq == 0u) {                   // the 32-bit tag is about to repeat: forget the old words
                HIP_TRY(hipMemsetAsync(net->d_sum_words, 0, 8 * sizeof(unsigned long long), s));
                net->sums_seq = 1u;
            }
            NAQS_KLAUNCH(vmc_seed_delta_kernel<true>, dim3((unsigned)((M * (Kp >> 2) + 255) / 256)), dim3(256), 0, s, d, M, keys_dev,
                               reinterpret_cast<const double2 *>(seeds->eloc), seeds->w, nullptr, reinterpret_cast<float2 *>(seeds->g_out),
                               g_amp, top, L.top_ld, seeds->ev, net->d_wb + wb_offset(net, H),
                               reinterpret_cast<const float *>(base + L.act[H - 1]), Kp, reinterpret_cast<float *>(base + L.delta[H - 1]),
                               const_cast<double *>(seeds->sums), net->d_sum_words, net->sums_seq, net->ctl);
        } else {
            NAQS_KLAUNCH(vmc_seed_delta_kernel<false>, dim3((unsigned)((M * (Kp >> 2) + 255) / 256)), dim3(256), 0, s, d, M, keys_dev,
                               reinterpret_cast<const double2 *>(seeds->eloc), seeds->w, seeds->sums, reinterpret_cast<float2 *>(seeds->g_out),
                               g_amp, top, L.top_ld, seeds->ev, net->d_wb + wb_offset(net, H),
                               reinterpret_cast<const float *>(base + L.act[H - 1]), Kp, reinterpret_cast<float *>(base + L.delta[H - 1]),
                               nullptr, nullptr, 0u, nullptr);
        }
        HIP_TRY(hipGetLastError());
        if (side) { HIP_TRY(hipEventRecord(net->ev_fork, s)); HIP_TRY(hipStreamWaitEvent(sa, net->ev_fork, 0)); }
    } else if (seeds != nullptr) {
        // (on the caller's stream, before the fork: both halves of the backward pass read what it writes)
        NAQS_KLAUNCH(vmc_seed_kernel, dim3((unsigned)((M * L.top_ld + 255) / 256)), dim3(256), 0, s, d, M, keys_dev,
                           reinterpret_cast<const double2 *>(seeds->eloc), seeds->w, seeds->sums, reinterpret_cast<float2 *>(seeds->g_out),
                           g_amp, top, L.top_ld, seeds->ev);
        HIP_TRY(hipGetLastError());
        if (side) { HIP_TRY(hipEventRecord(net->ev_fork, s)); HIP_TRY(hipStreamWaitEvent(sa, net->ev_fork, 0)); }
    } else {
        NAQS_KLAUNCH(split_g_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, sa, M, g2, g_amp);
        HIP_TRY(hipGetLastError());
    }
    // NAQS_TRAIN_MEGA=0: every piece its own launch
    const bool mega = H == 2 && d.Ha == 64 && !side && naqs::env_int("NAQS_TRAIN_MEGA", 1) == 1;
    // naqs_vmc_run's steps: the phase MLP's half of what follows goes to the side stream (see the launch below)
    const bool defer = net->defer_phase && mega && adam != nullptr && seed_delta;
    hipStream_t sp = s;
    if (defer) {
        if (!net->side_stream) HIP_TRY(hipStreamCreateWithFlags(&net->side_stream, hipStreamNonBlocking));
        if (!net->ev_fork) HIP_TRY(hipEventCreateWithFlags(&net->ev_fork, hipEventDisableTiming));
        if (!net->ev_phase_done) HIP_TRY(hipEventCreateWithFlags(&net->ev_phase_done, hipEventDisableTiming));
        sp = net->side_stream;
    }
    naqs::ampbw::AmpSrc amp_src{};
    if (mega) st = naqs::net_blocks_backward_plan(net, net->dims, net->amp_src_off, net->amp_params, M, 0, &F.set[0], &amp_src);
    else st = naqs::net_blocks_backward(net, net->dims, net->d_w, net->amp_src_off, net->amp_params, M, keys_dev, g_amp, grad_dev, 0, sa,
                                        &F.set[0], 0);
    if (st != NAQS_OK) return st;
    F.n_sets = 1;
    F.set_end[0] = net->amp_params;
    if (side) HIP_TRY(hipEventRecord(net->ev_join, sa));

    // phase block.  First the chain of deltas, output layer down (the critical path: each needs the one above) ...
    if (seeds == nullptr) {
        NAQS_KLAUNCH(top_delta_kernel, dim3((unsigned)((M * L.top_ld + 255) / 256)), dim3(256), 0, s, d, M, keys_dev, g2, top, L.top_ld);
        HIP_TRY(hipGetLastError());
    }
    const float *dl[MAXL];                            // delta of linear layer l's output
    dl[H] = top;
    MegaArgs A{};
    // tile height of the first hidden layer's delta (grad_in): 32 rows while even those leave CUs to spare beside the launch's
    // other pieces (NAQS_GRAD_IN_TBM: 32 / 64 forces).  Small tables: the same workgroups form the first layer's weight
    // gradient (W0Fuse; NAQS_FUSE_W0=0: a launch of its own) — job 0 is then sliced by those tiles in EITHER form of the
    // backward pass, so that the one launch and its pieces stay bit-identical
    const int tbm_env = naqs::env_int("NAQS_GRAD_IN_TBM", 0);
    const int gin_tbm = tbm_env == 32 || tbm_env == 64 ? tbm_env
                                                       : ((M + 31) / 32 * (pad64(net->phase_K[(size_t)(H >= 1 ? 1 : 0)]) / TB) <= 4ll * net->cu_count ? 32 : 64);
    const bool w0_tiles = H == 2 && d.Ha == 64 && M <= W0_FUSE_MAX_ROWS && pad64(net->phase_K[0]) == TB;
    for (int l = H; l > 0; --l) {
        const int Np = pad64(net->phase_N[(size_t)l]), Kp = pad64(net->phase_K[(size_t)l]);
        const float *in = reinterpret_cast<const float *>(base + L.act[l - 1]);
        float *dnext = reinterpret_cast<float *>(base + L.delta[l - 1]);
        if (l == H && seed_delta) { dl[l - 1] = dnext; continue; }
        if (l == H) {                                     // below the output layer: one term per element, no GEMM
            const float2 *gsrc = seeds != nullptr ? reinterpret_cast<const float2 *>(seeds->g_out) : g2;
            NAQS_KLAUNCH(delta_below_top_kernel, dim3((unsigned)((M * (Kp >> 2) + 255) / 256)), dim3(256), 0, s, d, M, keys_dev, gsrc,
                               net->d_wb + wb_offset(net, l), in, Kp, dnext);
            HIP_TRY(hipGetLastError());
            dl[l - 1] = dnext;
            continue;
        }
        if (mega) {                                       // (l == 1 here: part of the one launch below)
            A.gin_tbm = gin_tbm;
            A.gin_tiles_i = (int)((M + A.gin_tbm - 1) / A.gin_tbm);
            A.n_gin = A.gin_tiles_i * (Kp / TB);
            A.D = dl[l]; A.W = net->d_wb + wb_offset(net, l); A.In = in; A.Dout = dnext; A.Np = Np; A.Kp = Kp;
            dl[l - 1] = dnext;
            continue;
        }
        NAQS_KLAUNCH(grad_in_kernel, dim3((unsigned)((M + TB - 1) / TB), Kp / TB), dim3(256), 0, s, dl[l],
                           net->d_wb + wb_offset(net, l), in, M, Np, Kp, dnext);
        HIP_TRY(hipGetLastError());
        dl[l - 1] = dnext;
    }
    // ... then every layer's weight gradient in one launch and one fixed-order reduction
    GradWJobs J{};
    J.n = H + 1;
    int blocks_total = 0;
    int64_t c_off = 0, b_off = 0, elems = 0;
    for (int l = 0; l <= H; ++l) {
        const int N = net->phase_N[(size_t)l], K = net->phase_K[(size_t)l], Np = pad64(N), Kp = pad64(K);
        const int blocks = (Np / TB) * (Kp / TB);
        int slices = (int)std::min<int64_t>(std::max(1, 512 / blocks), (M + 127) / 128);
        slices = std::max(1, slices);
        int64_t rows = ((M + slices - 1) / slices + CH - 1) / CH * CH;
        if (l == 0 && w0_tiles) rows = gin_tbm;                                      // one slice per grad_in row tile
        slices = (int)((M + rows - 1) / rows);
        J.P[l] = dl[l]; J.ldp[l] = Np;
        J.Q[l] = l == 0 ? x : reinterpret_cast<const float *>(base + L.act[l - 1]);
        J.ldq[l] = l == 0 ? L.x_ld : L.act_ld[l - 1];                  // == Kp
        J.Np[l] = Np; J.Kp[l] = Kp; J.N[l] = N; J.K[l] = K; J.slices[l] = slices; J.rows_per_slice[l] = rows;
        blocks_total += blocks * slices;
        J.block_end[l] = blocks_total;
        J.cpart_off[l] = c_off; c_off += (int64_t)slices * Np * Kp;
        J.bpart_off[l] = b_off; b_off += (int64_t)slices * Np;
        J.out_off[l] = net->phase_src_off[(size_t)l];
        elems += (int64_t)N * K + N;
        J.elem_end[l] = elems;
    }
    if (mega) {
        A.M = M; A.cpart = cpart; A.bpart = bpart; A.gw_bid0 = J.block_end[0];
        A.amp_wgs = F.set[0].n_partials; A.n_amp = A.amp_wgs * d.P;
        A.amp_w = net->d_w; A.g_amp = g_amp; A.keys = keys_dev;
        A.amp_partial = const_cast<float *>(F.set[0].partial); A.amp_stride = F.set[0].stride;
        const size_t lds = std::max(naqs::ampbw::smem_floats(d), (size_t)std::max(TB * LDD + CH * LDT, 2 * CH * LDT)) * sizeof(float);
        if (lds > 64 * 1024) return NAQS_ERR_UNSUPPORTED;
        A.fuse_w0 = w0_tiles && naqs::env_int("NAQS_FUSE_W0", 1) != 0 ? 1 : 0;
        A.w0 = W0Fuse{x, L.x_ld, cpart + J.cpart_off[0], bpart + J.bpart_off[0], J.Kp[0]};
        if (defer) {
            // naqs_vmc_run: the amplitude blocks' pieces alone on the caller's stream — the next sampler call needs nothing
            // else — and the phase MLP's pieces (first hidden layer's delta, every layer's weight gradient) on the side
            // stream BEHIND them, where they run beside that sampler call's almost empty launches instead of in front of them.
            // Same device functions on the same operands: the same numbers as the one launch.
            MegaArgs Aa = A;
            Aa.n_gin = 0;
            NAQS_KLAUNCH(backward_mega_kernel, dim3((unsigned)A.n_amp), dim3(256), lds, s, Aa, d, J, amp_src);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(net->ev_fork, s));
            HIP_TRY(hipStreamWaitEvent(sp, net->ev_fork, 0));
            MegaArgs Ap = A;
            Ap.n_amp = 0;
            NAQS_KLAUNCH(backward_mega_kernel, dim3((unsigned)(A.n_gin + blocks_total - J.block_end[0])), dim3(256), lds, sp, Ap, d, J, amp_src);
            HIP_TRY(hipGetLastError());
        } else {
            NAQS_KLAUNCH(backward_mega_kernel, dim3((unsigned)(A.n_gin + A.n_amp + blocks_total - J.block_end[0])), dim3(256), lds, s,
                               A, d, J, amp_src);
            HIP_TRY(hipGetLastError());
        }
        if (!A.fuse_w0) {
            NAQS_KLAUNCH(grad_w_kernel, dim3((unsigned)J.block_end[0]), dim3(256), 0, sp, J, M, cpart, bpart, 0);
            HIP_TRY(hipGetLastError());
        }
    } else {
        NAQS_KLAUNCH(grad_w_kernel, dim3((unsigned)blocks_total), dim3(256), 0, s, J, M, cpart, bpart, 0);
        HIP_TRY(hipGetLastError());
    }
    if (side) HIP_TRY(hipStreamWaitEvent(s, net->ev_join, 0));   // the amplitude blocks' partial sums are complete
    F.total = F.set_end[0] + elems;
    if (defer) {                                             // each half's reductions + Adam update behind its own pieces
        st = launch_grad_finish(F, J, cpart, bpart, grad_dev, adam, s, 0, F.set_end[0]);
        if (st != NAQS_OK) return st;
        st = launch_grad_finish(F, J, cpart, bpart, grad_dev, adam, sp, F.set_end[0], F.total);
        if (st != NAQS_OK) return st;
        net->phase_pending = true;                           // (the re-pack and ev_phase_done follow: naqs_vmc_step)
        return NAQS_OK;
    }
    // naqs_vmc_step (NAQS_PACK_OVERLAP=2, the default): the launch's first workgroups each finish, update and PACK one of the
    // pairs the next sampler call's own first workgroup reads, so that the rest of the amplitude re-pack can ride in that call's
    // first launch (naqs_pack.hpp).  Conditions: what that launch needs to be the four-level head with the block MLPs on the matrix
    // cores (more than four pairs, fragments allocated, f16x2 format) and a pair's parameters within a workgroup's reach.
    HeadPairs hp;
    net->amp_head_packed = 0;
    if (adam != nullptr && naqs::env_int("NAQS_PACK_OVERLAP", 2) >= 2 && net->d_wamp != nullptr && net->packed_fmt == 2 && d.P > HEAD_MAX_PAIRS &&
        naqs::amp_raw_floats(d.Ha, d.n_out_amp, HEAD_MAX_PAIRS - 1) <= HEAD_EPT * 256 && F.set_out[0] == 0) {
        hp.pairs = HEAD_MAX_PAIRS; hp.Ha = d.Ha; hp.nout = d.n_out_amp; hp.wamp = net->d_wamp;
        bool contiguous = net->amp_src_off[0] == 0;
        for (int p = 0; p < HEAD_MAX_PAIRS; ++p) {
            hp.off[p] = net->amp_src_off[p] - net->amp_src_off[0];
            contiguous = contiguous && net->amp_src_off[p + 1] - net->amp_src_off[p] == naqs::amp_raw_floats(d.Ha, d.n_out_amp, p);
        }
        hp.skip = net->amp_src_off[HEAD_MAX_PAIRS] - net->amp_src_off[0];
        if (!contiguous) hp = HeadPairs{};
    }
    st = launch_grad_finish(F, J, cpart, bpart, grad_dev, adam, s, 0, -1, hp.pairs > 0 ? &hp : nullptr);
    if (st == NAQS_OK) net->amp_head_packed = hp.pairs;
    return st;
}

static int train_backward_vmc_impl(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, const double *eloc_dev,
                                   const double *w_dev, const double *sums_dev, float *g_dev, double *ev_dev, float *grad_dev,
                                   void *stream, const naqs::AdamArgs *adam, bool form_sums = false);

NAQS_API int naqs_vmc_step(naqs_net_t *net, naqs_ham_t *ham, int64_t n_samples, uint64_t seed, int64_t max_unique, int64_t m_lo,
                           int64_t m_hi, uint64_t *keys_dev, int64_t *counts_dev, float *probs_dev, double *weights_dev,
                           float *logpsi_dev, double *eloc_dev, double *sums_dev, float *g_dev, double *ev_dev, float *grad_dev,
                           float *param_dev, float *exp_avg_dev, float *exp_avg_sq_dev, double lr, double beta1, double beta2,
                           double eps, double weight_decay, int64_t adam_step, int64_t info_host[3], void *stream) {
    if (!net || !info_host || !g_dev || !ev_dev || !grad_dev) return NAQS_ERR_INVALID;
    if (adam_step >= 1 && (!param_dev || !exp_avg_dev || !exp_avg_sq_dev)) return NAQS_ERR_INVALID;
    info_host[2] = 0;
    int64_t info2[2] = {0, 0};
    // sampler, the step's one host synchronisation, forward pass + E_loc of the sampled table
    // (an abandoned step must not have evaluated anything: sample first, look at M, then go on)
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!net->have_weights) return NAQS_ERR_INVALID;
    // The training forward goes out BEHIND THE SAMPLER'S LAUNCHES, before the host knows M (NAQS_SPEC_FORWARD=0: after): between
    // the sampler's last launch and a forward pass launched once M is known the GPU idled ~7 us per step (the store reaching
    // the host, the launch reaching the GPU).  The kernel reads M on the device; the launch covers the last accepted draw's M
    // plus an eighth (at least 64 rows) in the kernel form that M gets.  Once the host has M it checks that the real M gets
    // the same form and fits the launch — otherwise (and after an abandoned draw, whose forward pass was for nothing) the
    // ordinary launch follows and overwrites everything: the same kernel on the same rows either way, bit for bit.
    struct SpecForward : AfterSampler {
        naqs_net_t *net; naqs_ham_t *ham; hipStream_t s; const uint64_t *keys; float *logpsi; int64_t max_unique;
        bool launched = false; int64_t rows = 0; naqs::ElocFeed feed{}; naqs::PhaseForm form;
        int operator()() override {
            const int64_t hint = net->spec_hint;
            if (hint <= 0 || net->aggregate || !ham || !logpsi || naqs::env_int("NAQS_SPEC_FORWARD", 1) == 0) return NAQS_OK;
            if (naqs::ham_device(ham) != net->device) return NAQS_OK;
            int64_t cover = std::min<int64_t>(max_unique, hint + std::max<int64_t>(64, hint / 8));
            if (naqs::env_int("NAQS_DEBUG_SPEC_SHRINK", 0) != 0) cover = std::max<int64_t>(16, hint / 2);      // (tests: a launch that does not fit)
            form = naqs::net_logpsi_form(net, hint, /*training=*/true);
            if (form.kind != 1) return NAQS_OK;
            int st = ensure_train_scratch(net, cover);
            if (st != NAQS_OK) return st;
            const TrainLayout L = train_layout(net, net->train_cap);
            char *base = static_cast<char *>(net->d_train);
            naqs::PhaseSave save;
            save.x = reinterpret_cast<float *>(base + L.x);
            save.x_ld = L.x_ld;
            for (int l = 0; l + 1 < net->dims.n_lin; ++l) {
                save.act[l] = reinterpret_cast<float *>(base + L.act[l]);
                save.act_ld[l] = L.act_ld[l];
            }
            st = naqs::eloc_begin(ham, cover, s, &feed);
            if (st != NAQS_OK) return st;
            naqs::SpecRows spec;
            spec.U = net->fin_job.U; spec.P = net->fin_job.P; spec.m_var = hint;
            spec.host_finish = naqs::env_int("NAQS_SPEC_HOST_FINISH", 1) != 0;
            st = naqs::net_logpsi_impl(net, cover, keys, logpsi, s, feed, save, &spec);
            if (st == NAQS_ERR_UNSUPPORTED) return NAQS_OK;            // (nothing was launched: the ordinary forward follows)
            if (st != NAQS_OK) return st;
            launched = true; rows = cover;
            ++net->spec_launched;
            return NAQS_OK;
        }
    } specf;
    specf.net = net; specf.ham = ham; specf.s = s; specf.keys = keys_dev; specf.logpsi = logpsi_dev; specf.max_unique = max_unique;
    st = sample_and_wait(net, n_samples, seed, max_unique, keys_dev, counts_dev, probs_dev, weights_dev, s, info2, nullptr, &specf);
    if (st != NAQS_OK) return st;
    info_host[0] = info2[0]; info_host[1] = info2[1];
    const int64_t M = info2[0];
    if (info2[1] != 0 || M <= 0 || M < m_lo || M > m_hi) return NAQS_OK;        // abandoned: the caller adapts n_samples
    net->spec_hint = M;
    const bool spec_hit = specf.launched && M <= specf.rows && naqs::net_logpsi_form(net, M, true) == specf.form;
    if (spec_hit) ++net->spec_hits;
    // small tables: the weighted sums of E_loc are formed by the first workgroup of the backward pass's seed kernel (same
    // arithmetic, same order: naqs_reduce.hpp) instead of by a launch between E_loc and the seeds (NAQS_FUSE_SUMS=0: the launch)
    const bool form_sums = !net->aggregate && net->dims.n_lin >= 2 && M <= SUMS_FUSE_MAX_ROWS && sums_dev != nullptr &&
                           net->d_sum_words != nullptr && naqs::env_int("NAQS_FUSE_SUMS", 1) != 0;
    if (spec_hit) {
        if (!eloc_dev || (!form_sums && (!weights_dev || !sums_dev))) return NAQS_ERR_INVALID;
        st = naqs::eloc_main(ham, M, specf.feed, eloc_dev, form_sums ? nullptr : weights_dev, form_sums ? nullptr : sums_dev, s);
    } else {
        st = naqs_net_train_forward_eloc(net, ham, M, keys_dev, form_sums ? nullptr : weights_dev, logpsi_dev, eloc_dev,
                                         form_sums ? nullptr : sums_dev, stream);
    }
    if (st != NAQS_OK) return st;
    naqs::AdamArgs adam;
    if (adam_step >= 1) adam = naqs::adam_args(param_dev, exp_avg_dev, exp_avg_sq_dev, lr, beta1, beta2, eps, weight_decay, adam_step);
    st = train_backward_vmc_impl(net, M, keys_dev, eloc_dev, weights_dev, sums_dev, g_dev, ev_dev, grad_dev, stream,
                                 adam_step >= 1 ? &adam : nullptr, form_sums);
    if (st != NAQS_OK) return st;
    if (adam_step >= 1) {
        // the whole re-pack rides in the next sampling call's first launch, in the workgroups behind its one busy workgroup — the
        // fragments of the four pairs THAT workgroup reads were packed by the update's own launch, from the values it had just
        // written (grad_finish_kernel's first workgroups); naqs_pack.hpp.  NAQS_PACK_OVERLAP=1: round 4's form — the amplitude
        // jobs as a launch here, the phase layers' share hosted; 0: everything here, in order
        const bool deferred = net->phase_pending;            // (train_backward_impl put the phase MLP's half on the side stream)
        net->phase_pending = false;                          // (naqs_net_set_weights must not wait for what it is part of)
        const int overlap = naqs::env_int("NAQS_PACK_OVERLAP", 2);
        // (2 needs the update to have packed the leading pairs' fragments: train_backward_impl, net->amp_head_packed)
        net->overlap_next_pack = deferred ? 1 : std::min(net->amp_head_packed > 0 ? 2 : 1, std::max(0, overlap));
        st = naqs_net_set_weights(net, param_dev, net->n_params, stream);
        net->overlap_next_pack = 0;
        if (st != NAQS_OK) return st;
        if (deferred) {
            // the phase layers' re-pack behind their update, on the side stream; whoever reads them, the gradient or the
            // parameters next waits for ev_phase_done (net_flush_pack) — the next step's forward pass, ~100 us from here
            net->pack_stream = net->side_stream;             // (nothing to order: the update ran on this very stream)
            st = naqs::net_flush_pack(net, net->side_stream);
            if (st != NAQS_OK) return st;
            HIP_TRY(hipEventRecord(net->ev_phase_done, net->side_stream));
            net->phase_pending = true;
        }
    }
    info_host[2] = 1;
    return NAQS_OK;
}

// The loop of PartialSamplingOptimizer.run (energy.py:975-1008) with get_samples' adaptive sample count (energy.py:936-971)
// around naqs_vmc_step — the rules are naqs_amd.optimizer._onecall_step's, statement for statement, so that a run through this
// call and a run through the per-step calls are the same sequence of launches with the same seeds.
static inline uint64_t sample_seed(const uint64_t base, const int64_t call) {        // naqs_amd.wavefunction._next_sample_seed
    uint64_t x = base + 0x9E3779B97F4A7C15ull * (uint64_t)call;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

NAQS_API int naqs_net_spec_counts(const naqs_net_t *net, int64_t counts[2]) {
    if (!net || !counts) return NAQS_ERR_INVALID;
    counts[0] = net->spec_launched; counts[1] = net->spec_hits;
    return NAQS_OK;
}

NAQS_API int naqs_vmc_run(naqs_net_t *net, naqs_ham_t *ham, int64_t n_steps, naqs_vmc_run_args_t *a, void *stream) {
    if (!net || !ham || !a || n_steps < 0) return NAQS_ERR_INVALID;
    if (!a->param_dev || !a->exp_avg_dev || !a->exp_avg_sq_dev || !a->grad_dev || !a->keys_dev || !a->counts_dev || !a->probs_dev ||
        !a->weights_dev || !a->logpsi_dev || !a->eloc_dev || !a->g_dev || !a->ev_log_dev || !a->sums_log_dev || !a->m_log_host ||
        !a->ns_log_host || !a->t_log_host || (a->events_cap > 0 && !a->events))
        return NAQS_ERR_INVALID;
    if (a->n_unq_samples_max <= 0 || a->n_samples <= 0 || a->adam_step < 0 || a->ring_elems < 0 || a->ring_off < 0) return NAQS_ERR_INVALID;
    if (a->ring_elems > 0 && a->ring_elems < a->n_unq_samples_max) return NAQS_ERR_INVALID;
    a->n_events = a->steps_done = 0;
    a->stop_reason = 0;
    a->last_keys_off = a->ring_elems > 0 ? a->ring_off : 0;
    const int64_t cap = a->n_unq_samples_max;
    const auto t0 = std::chrono::steady_clock::now();
    // NAQS_DEFER_PHASE=1: the phase MLP's half of every step's backward pass / update / re-pack beside the next step's sampler
    // (naqs_net.hpp); joined below, before the caller sees anything.  OFF by default — measured slower: the two event
    // hand-overs between the streams cost 7-30 us of queue time per step on this pool where the split saves 2 (H2O) to 12 us
    // (N2) of kernel time on the caller's stream: N2 0.188-0.190 ms per step against 0.185-0.187, H2O 0.142 against 0.130
    // (profiles/r05_defer_phase_timeline.txt).  Same numbers either way (tests/test_optimizer_gpu.py).
    struct DeferScope {
        naqs_net_t *net; hipStream_t s;
        ~DeferScope() {
            net->defer_phase = false;
            (void)naqs::net_finish_pending(net, s);
        }
    } scope{net, reinterpret_cast<hipStream_t>(stream)};
    net->defer_phase = naqs::env_int("NAQS_DEFER_PHASE", 0) != 0;
    for (int64_t i = 0; i < n_steps; ++i) {
        if (a->ring_elems > 0 && a->ring_off + cap > a->ring_elems) { a->stop_reason = 1; break; }
        uint64_t *keys = a->keys_dev + (a->ring_elems > 0 ? a->ring_off : 0);
        int last_action = 0;
        int64_t info[3] = {0, 0, 0};
        for (;;) {
            const bool free_ = a->n_samples != a->n_unq_samples_min && a->n_samples != a->n_samples_max;
            const int64_t m_lo = (free_ && last_action >= 0) ? a->n_unq_samples_min : 0;
            if (a->n_events >= a->events_cap) { a->stop_reason = 2; break; }       // (room for the event this draw may cause)
            const uint64_t seed = sample_seed(a->seed_base, ++a->sample_calls);
            const int st = naqs_vmc_step(net, ham, a->n_samples, seed, cap, m_lo, cap, keys, a->counts_dev, a->probs_dev, a->weights_dev,
                                         a->logpsi_dev, a->eloc_dev, a->sums_log_dev + 4 * i, a->g_dev, a->ev_log_dev + 2 * i, a->grad_dev,
                                         a->param_dev, a->exp_avg_dev, a->exp_avg_sq_dev, a->lr, a->beta1, a->beta2, a->eps,
                                         a->weight_decay, a->adam_step + 1, info, stream);
            if (st != NAQS_OK) return st;
            if (info[2]) break;                            // taken
            int64_t n_unq = info[0];
            int action = 0;
            const bool overflow = info[1] != 0;
            if (overflow) { n_unq = cap + 1; action = -1; }
            if (free_ || overflow) {
                if (n_unq < a->n_unq_samples_min && last_action >= 0) {
                    action = 1;
                    a->n_samples = std::min<int64_t>(a->n_samples > a->n_samples_max / 10 ? a->n_samples_max : a->n_samples * 10, a->n_samples_max);
                } else if (n_unq > cap && last_action <= 0) {
                    action = -1;
                    a->n_samples = std::max<int64_t>(a->n_samples / 10, a->n_unq_samples_min);
                }
            }
            naqs_vmc_event_t &e = a->events[a->n_events++];
            e.step = i; e.n_unique = n_unq; e.overflow = overflow ? 1 : 0; e.n_samples = a->n_samples;
            // (overflow right after an increase: get_samples' rules leave n_samples alone and draw again)
            e.action = (overflow && !(n_unq > cap && last_action <= 0)) ? 0 : action;
            if (action == 0) { a->stop_reason = 3; break; }
            last_action = action;
        }
        if (a->stop_reason != 0) break;
        a->adam_step += 1;
        a->m_log_host[i] = info[0];
        a->ns_log_host[i] = a->n_samples;
        a->t_log_host[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        a->last_keys_off = a->ring_elems > 0 ? a->ring_off : 0;
        if (a->ring_elems > 0) a->ring_off += info[0];
        a->steps_done = i + 1;
    }
    return NAQS_OK;
}

// the sharded step's first and last library calls (include/naqs_hip.h: the four-call sequence with its three collectives)
NAQS_API int naqs_vmc_shard_sample_forward(naqs_net_t *net, int64_t n_samples, uint64_t seed, int64_t max_unique, int64_t m_lo,
                                           int64_t m_hi, int rank, int world, uint64_t *keys_dev, int64_t *counts_dev,
                                           float *probs_dev, double *weights_dev, float *logpsi_shard_dev, int64_t info_host[3],
                                           void *stream) {
    if (!net || !info_host || !logpsi_shard_dev || world < 1 || rank < 0 || rank >= world) return NAQS_ERR_INVALID;
    info_host[2] = 0;
    int64_t info2[2] = {0, 0};
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    if (!net->have_weights) return NAQS_ERR_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    st = sample_and_wait(net, n_samples, seed, max_unique, keys_dev, counts_dev, probs_dev, weights_dev, s, info2);
    if (st != NAQS_OK) return st;
    info_host[0] = info2[0]; info_host[1] = info2[1];
    const int64_t M = info2[0];
    if (info2[1] != 0 || M <= 0 || M < m_lo || M > m_hi) return NAQS_OK;        // abandoned: the caller adapts n_samples
    const int64_t S = (M + world - 1) / world, b = std::min<int64_t>(M, (int64_t)rank * S), e = std::min<int64_t>(M, b + S);
    if (e > b) {
        st = naqs_net_train_forward(net, e - b, keys_dev + b, logpsi_shard_dev, stream);
        if (st != NAQS_OK) return st;
    }
    info_host[2] = 1;
    return NAQS_OK;
}

NAQS_API int naqs_vmc_shard_update(naqs_net_t *net, const float *grad_dev, float *param_dev, float *exp_avg_dev, float *exp_avg_sq_dev,
                                   double lr, double beta1, double beta2, double eps, double weight_decay, int64_t adam_step,
                                   void *stream) {
    if (!net || !grad_dev || !param_dev || !exp_avg_dev || !exp_avg_sq_dev || adam_step < 1) return NAQS_ERR_INVALID;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    st = naqs_adam_step(net->n_params, param_dev, grad_dev, exp_avg_dev, exp_avg_sq_dev, lr, beta1, beta2, eps, weight_decay, adam_step,
                        stream);
    if (st != NAQS_OK) return st;
    return naqs_net_set_weights(net, param_dev, net->n_params, stream);
}

NAQS_API int naqs_net_train_backward(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, const float *g_dev,
                                     float *grad_dev, void *stream) {
    return train_backward_impl(net, M, keys_dev, g_dev, grad_dev, stream, nullptr);
}

static int train_backward_vmc_impl(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, const double *eloc_dev,
                                   const double *w_dev, const double *sums_dev, float *g_dev, double *ev_dev, float *grad_dev,
                                   void *stream, const naqs::AdamArgs *adam, bool form_sums) {
    if (!net || !sums_dev || !ev_dev || !g_dev || (M > 0 && (!eloc_dev || !w_dev))) return NAQS_ERR_INVALID;
    if (net->aggregate || M == 0) {                   // per-pair phase blocks (or nothing to do): the two separate calls
        int st = naqs_vmc_loss_grad_ev(M, eloc_dev, w_dev, sums_dev, g_dev, ev_dev, stream);
        if (st != NAQS_OK) return st;
        return train_backward_impl(net, M, keys_dev, g_dev, grad_dev, stream, nullptr, adam);
    }
    if (form_sums && net->dims.n_lin < 2) return NAQS_ERR_INVALID;              // (the caller's condition: the seed + delta kernel runs)
    const VmcSeeds seeds{eloc_dev, w_dev, sums_dev, g_dev, ev_dev, form_sums};
    return train_backward_impl(net, M, keys_dev, g_dev, grad_dev, stream, &seeds, adam);
}

NAQS_API int naqs_net_train_backward_vmc(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, const double *eloc_dev,
                                         const double *w_dev, const double *sums_dev, float *g_dev, double *ev_dev, float *grad_dev,
                                         void *stream) {
    return train_backward_vmc_impl(net, M, keys_dev, eloc_dev, w_dev, sums_dev, g_dev, ev_dev, grad_dev, stream, nullptr);
}
