// naqs_net.hpp — the network handle and the device pieces of the amplitude conditionals shared by the
// translation units that evaluate (naqs_logpsi.hip), sample (naqs_sample.hip) and differentiate
// (naqs_grad.hip) the orbital NADE.
#pragma once
#include <cstdint>
#include <cmath>
#include <vector>

#include "naqs_common.hpp"
#include "naqs_poll.hpp"

namespace naqs {

constexpr int MAXP = NAQS_NET_MAX_PAIRS;
constexpr int MAXL = NAQS_NET_MAX_PHASE_LAYERS + 1;   // linear layers of the phase block

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct NetDims {
    int32_t P;                         // orbital pairs
    int32_t n_alpha, n_beta;           // < 0: unrestricted
    int32_t n_alpha_down, n_beta_down;
    int32_t min_n_set;
    int32_t masking;                   // 0 NONE, 1 PARTIAL, 2 FULL
    int32_t sym;                       // amplitude spin symmetry
    int32_t phase_sym;                 // phase spin symmetry (-phase_sym: spin-ordered inputs, 3 outputs for |00>, |01>=|10>, |11>, sign shift)
    int32_t Ha;                        // amplitude hidden width
    int32_t n_out_amp;                 // 5 with symmetry, 4 without
    uint8_t qa[MAXP], qb[MAXP];        // qubit (bit) index of the alpha / beta orbital of model pair n
    int32_t amp_off[MAXP];             // offset (floats) of pair n's packed parameters: Ha rows of
                                       //   [W1[j][0..nin) | b1[j] | W2[0..5)[j] | pad] (16-byte multiples), then b2 [8]
    // phase MLP, zero-padded and tiled: layer l: W [N_pad/16][K_pad/16][64 lanes][4], bias [N_pad]
    int32_t n_lin;
    int32_t K_pad[MAXL], N_pad[MAXL], w_off[MAXL], b_off[MAXL];
    int32_t ld;                        // LDS row stride (floats)
    // the same MLP split in 3 bf16 planes for the bf16 matrix cores (phase_kernel_bf16x3):
    // layer l: planes [3][N_pad/16][Kh_pad/32][64 lanes][8 bf16] at wh_off (in bf16 units), bias (f32) at b_off
    int32_t Kh_pad[MAXL], wh_off[MAXL];
    int32_t ldh;                       // LDS row stride of one activation plane (bf16 units)
};

// f16x2 form of the phase MLP (phase_kernel_h<.., FMT = 2>): every f32 value v is carried as two f16 planes of s*v (s a power
// of two per tensor: hi = f16(s v), lo = f16(s v - hi)), so that the values sit high in the f16 range and lo stays normal.
// Per linear layer l: sw = scale of the weights, sn = scale of the layer's OUTPUT activations (1 for the last layer),
// c = sn / (s_in sw) takes an accumulator to the scaled output, isn = 1 / sn.  Written by pack_net_kernel from the
// weight-derived bounds of net_bounds_kernel (|h_l| <= rowsum_l * bound_{l-1} + max|b_l| < 2^15 / sn: no overflow).
struct PhaseScales { float sw[MAXL], c[MAXL], sn[MAXL], isn[MAXL]; };
// per-workgroup partial maxima of the weights (naqs_pack.hpp: net_bounds_body): every consumer reduces the BOUNDS_WG entries
// itself (no atomic maxima — ~10^3 of them on nine addresses were 13 of the kernel's 14 us — and nothing to zero between
// calls).  Each entry is a 64-bit word (pack tag << 32 | float bits) written with a relaxed agent-scope store and polled by its
// readers, which may be workgroups of the SAME launch with higher block indices: value and flag are one word — no fence
constexpr int BOUNDS_WG = 64;
struct PhaseRaw { unsigned long long max_w[MAXL][BOUNDS_WG], max_rowsum[MAXL][BOUNDS_WG], max_b[MAXL][BOUNDS_WG]; };

// ---- the phase block under use_phase_spin_sym (nade.py:281, 507-533, 590-610) ----
// row of the phase block's output layer that the realised outcome `occ` (alpha + 2 beta) of the last pair selects: the outcome
// itself, or — 3 outputs for |00>, |01> = |10>, |11> (nade.py:593-595: phase_i[:, [0, 1, 1, 2]]) — the shared row
__host__ __device__ inline int phase_out_row(const NetDims &d, const int occ) { return d.phase_sym ? ((occ + 1) >> 1) : occ; }
#if defined(__HIPCC__)
// spin-ordered inputs (nade.py:519-530): the alpha and beta strings of the first `np` pairs change places when idx(alpha) > idx(beta)
__device__ __forceinline__ void phase_order_inputs(const NetDims &d, const int np, uint32_t &a, uint32_t &b) {
    if (d.phase_sym) {
        const uint32_t mask = (1u << np) - 1u, ap = a & mask, bp = b & mask;
        if (ap > bp) { a = (a & ~mask) | bp; b = (b & ~mask) | ap; }
    }
}
// the sign by which spin-exchanged partners differ (nade.py:597-610): + pi (N_01 mod 2) where idx(alpha) < idx(beta) over all P
// pairs, N_01 = pairs with the alpha orbital empty and the beta orbital occupied; float32 pi like the reference's tensor
__device__ __forceinline__ float phase_sym_shift(const NetDims &d, const uint32_t a, const uint32_t b) {
    if (!d.phase_sym || !(a < b)) return 0.0f;
    return (__popc(~a & b & ((1u << d.P) - 1u)) & 1) ? 3.14159274101257324f : 0.0f;
}
__device__ __forceinline__ void key_strings(const NetDims &d, const uint64_t key, uint32_t &a, uint32_t &b) {
    a = b = 0u;
    for (int k = 0; k < d.P; ++k) { a |= (uint32_t)((key >> d.qa[k]) & 1ull) << k; b |= (uint32_t)((key >> d.qb[k]) & 1ull) << k; }
}
#endif

// one packed row [W1[j][0..NIN) | b1[j] | W2[0..5)[j] | pad] from LDS as 16-byte reads (rows are 16-byte multiples);
// element-wise `row[k]` reads compile to one ds_read_b32 each and those, not the FMAs, were the time of this loop
template <int S>
__device__ __forceinline__ void load_row(const float *__restrict__ row, float (&v)[S]) {
    const f32x4 *r4 = reinterpret_cast<const f32x4 *>(row);
#pragma unroll
    for (int q = 0; q < S / 4; ++q) {
        const f32x4 t = r4[q];
        v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
}

// partial output sums of pair NB over hidden units [j0, j1): o[c] += W2[c][j] * relu(W1[j].x + b1[j])
template <int NB>
__device__ __forceinline__ void amp_partial(const NetDims &d, const float *__restrict__ w, uint32_t first,
                                            uint32_t second, int j0, int j1, float (&o)[5]) {
    constexpr int NIN = NB == 0 ? 1 : 2 * NB;
    float x[NIN];
    if (NB == 0) {
        x[0] = 0.0f;
    } else {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            x[k] = ((first >> k) & 1u) ? 1.0f : -1.0f;
            x[NB + k] = ((second >> k) & 1u) ? 1.0f : -1.0f;
        }
    }
    const int nout = d.n_out_amp;
    constexpr int S = (NIN + 1 + 5 + 3) & ~3;          // packed row: W1[j][:], b1[j], W2[:][j], padded to 16 bytes
    const float *rows = w;                             // this pair's rows, staged in LDS by the workgroup
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        float row[S];                                  // same address in every lane -> LDS broadcast reads, 16 bytes each
        load_row<S>(rows + j * S, row);
        // two interleaved accumulation chains keep the FMA pipe busier than one 2n-long dependent chain
        float h0 = row[NIN], h1 = 0.0f;
#pragma unroll
        for (int k = 0; k + 1 < NIN; k += 2) { h0 = fmaf(row[k], x[k], h0); h1 = fmaf(row[k + 1], x[k + 1], h1); }
        if (NIN & 1) h0 = fmaf(row[NIN - 1], x[NIN - 1], h0);
        const float h = fmaxf(h0 + h1, 0.0f);
#pragma unroll
        for (int c = 0; c < 5; ++c)
            if (c < nout) o[c] = fmaf(row[NIN + 1 + c], h, o[c]);
    }
}

// electron-budget mask of pair NB given the prefix (nade.py:417-474): outcome (a, b) allowed iff the alpha / beta
// strings can still take that value; all true when unrestricted or NB < max(min_n_set, 1), like the reference
__device__ __forceinline__ void amp_budget_mask(const NetDims &d, int NB, uint32_t abits, uint32_t bbits, bool (&ok)[4]) {
    ok[0] = ok[1] = ok[2] = ok[3] = true;
    if (d.n_alpha >= 0 && NB >= max(d.min_n_set, 1)) {
        const int ua = __popc(abits), ub = __popc(bbits);
        const bool a_up = ua < d.n_alpha, a_dn = (NB - ua) < d.n_alpha_down;
        const bool b_up = ub < d.n_beta, b_dn = (NB - ub) < d.n_beta_down;
        ok[0] = a_dn && b_dn; ok[1] = a_up && b_dn; ok[2] = a_dn && b_up; ok[3] = a_up && b_up;
    }
}

// symmetrise the 5 raw outputs to the 4 outcome logits (nade.py:585-586): (o[0,1,1,2] + o[idx2sort[x_order]]) / 2
// (BITSEL: the two selects as bit masks on the values.  The compiler may turn `cond ? o[3] : o[4]` on an array it keeps in memory
// into ONE load from a selected address — in amp_backward_pair that made `o` a scratch array, five stores and two dependent loads
// of private memory per tile and a kernel that needs scratch set up at dispatch; round 6.  Same values bit for bit.)
template <bool BITSEL = false>
__device__ __forceinline__ void amp_symmetrise(const NetDims &d, const float (&o)[5], uint32_t abits, uint32_t bbits,
                                               float (&a4)[4]) {
    if (d.sym) {
        const int x_order = abits > bbits ? 0 : (abits == bbits ? 1 : 2);
        float s1, s2;
        if (BITSEL) {
            const uint32_t u1 = __float_as_uint(o[1]), u3 = __float_as_uint(o[3]), u4 = __float_as_uint(o[4]);
            const uint32_t m0 = x_order == 0 ? 0xFFFFFFFFu : 0u, m1 = x_order == 1 ? 0xFFFFFFFFu : 0u, m2 = x_order == 2 ? 0xFFFFFFFFu : 0u;
            s1 = __uint_as_float((m0 & u3) | (m1 & u1) | (m2 & u4));
            s2 = __uint_as_float((m0 & u4) | (m1 & u1) | (m2 & u3));
        } else {
            s1 = x_order == 0 ? o[3] : (x_order == 1 ? o[1] : o[4]);
            s2 = x_order == 0 ? o[4] : (x_order == 1 ? o[1] : o[3]);
        }
        a4[0] = (o[0] + o[0]) * 0.5f;
        a4[1] = (o[1] + s1) * 0.5f;
        a4[2] = (o[1] + s2) * 0.5f;
        a4[3] = (o[2] + o[2]) * 0.5f;
    } else {
        a4[0] = o[0]; a4[1] = o[1]; a4[2] = o[2]; a4[3] = o[3];
    }
}

// the conditional of pair NB: la[c] = 0.5 * log_softmax(2 a)[c] over the allowed outcomes (activations.py:40-46),
// -inf where masked; the mask is skipped on the last pair under PARTIAL masking (nade.py:615-617)
template <bool BITSEL = false>
__device__ __forceinline__ void amp_conditional(const NetDims &d, int NB, const float (&o)[5], uint32_t abits,
                                                uint32_t bbits, float (&la)[4], bool (&ok)[4]) {
    float a4[4];
    amp_symmetrise<BITSEL>(d, o, abits, bbits, a4);
    const bool mask_active = !(d.masking == 0 || (d.masking == 1 && NB == d.P - 1));
    if (mask_active) amp_budget_mask(d, NB, abits, bbits, ok);
    else ok[0] = ok[1] = ok[2] = ok[3] = true;
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 4; ++c) { a4[c] *= 2.0f; if (ok[c]) m = fmaxf(m, a4[c]); }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) if (ok[c]) s += expf(a4[c] - m);
    const float ls = logf(s);
#pragma unroll
    for (int c = 0; c < 4; ++c) la[c] = ok[c] ? 0.5f * ((a4[c] - m) - ls) : -INFINITY;
}

// log-amplitude of the realised outcome
__device__ __forceinline__ float amp_finish(const NetDims &d, int NB, const float (&o)[5], uint32_t abits,
                                            uint32_t bbits, int occ) {
    float la[4];
    bool ok[4];
    amp_conditional(d, NB, o, abits, bbits, la, ok);
    return occ == 0 ? la[0] : (occ == 1 ? la[1] : (occ == 2 ? la[2] : la[3]));
}

// The sampler's last launch — (M, overflow) for the caller and the polling host, the level sizes as hints for the next call,
// the weights counts / sum(counts) (energy.py:993) — described as a job: naqs_sample.hip launches it as sample_finish_kernel,
// or leaves it pending for a launch of the caller's that can host it (the training forward launched ahead of M).
struct SampleFinishJob {
    const int64_t *U = nullptr;             // level sizes U[0..P], overflow flag U[MAXP + 1]
    int P = 0;
    int64_t *info = nullptr;                // (M, overflow) on the device
    const int64_t *counts = nullptr;
    double *weights = nullptr;              // nullptr: no weights
    int64_t *early = nullptr;               // mapped host words (publish_info); nullptr: none
    int64_t seq = 0;
    int64_t *levels_out = nullptr;
};
#if defined(__HIPCC__)
// (M, overflow) for a host that polls mapped memory (naqs_vmc_step): system-scope stores, the call's sequence number last —
// the host waits for ITS number, so words left by an earlier call are never mistaken for this one's.
__device__ __forceinline__ void publish_info(int64_t *early, const int64_t M, const int64_t overflow, const int64_t seq) {
    __hip_atomic_store(&early[0], M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&early[1], overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&early[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// one workgroup of blockDim.x threads (a multiple of 64, <= 1024); s_part: 16 int64 of LDS.  The integer total is exact
// whatever the summation order, so the weights do not depend on the workgroup's size.
__device__ __forceinline__ void sample_finish_body(const SampleFinishJob &j, int64_t *s_part) {
    const int nt = (int)blockDim.x;
    const int64_t overflow = j.U[MAXP + 1];
    const int64_t M = overflow ? 0 : j.U[j.P];
    if (threadIdx.x == 0) {
        j.info[0] = M;
        j.info[1] = overflow;
        if (j.early != nullptr) publish_info(j.early, M, overflow, j.seq);
    }
    // the level sizes of this draw, for the host's choice of launches in the NEXT call (a hint: mapped memory it reads
    // without synchronising)
    if (j.levels_out != nullptr && (int)threadIdx.x <= j.P) j.levels_out[threadIdx.x] = j.U[threadIdx.x];
    if (j.weights == nullptr) return;
    int64_t part = 0;
    for (int64_t i = threadIdx.x; i < M; i += nt) part += j.counts[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_down(part, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
    int64_t total = 0;
    for (int i = 0; i < nt / WAVE; ++i) total += s_part[i];
    const double tot = (double)total;
    for (int64_t i = threadIdx.x; i < M; i += nt) j.weights[i] = (double)j.counts[i] / tot;
}
#endif

}  // namespace naqs

struct naqs_net {
    int device = 0;
    naqs_net_config_t cfg{};
    naqs::NetDims dims{};
    int64_t n_params = 0;
    int64_t amp_params = 0;                 // floats of all amplitude blocks in the flat source
    int64_t amp_src_off[naqs::MAXP] = {};         // per pair: offset in the flat source
    std::vector<int64_t> phase_src_off;     // per phase linear layer: offset in the flat source
    std::vector<int> phase_K, phase_N;
    // aggregate_phase: one phase block per pair, described as a second "amplitude-shaped" network (4 raw outputs, no
    // symmetry, the realised outcome's output is the pair's phase): the amplitude kernels run on it in raw mode
    bool aggregate = false;
    naqs::NetDims dph{};
    int64_t ph_src_off[naqs::MAXP] = {};          // per pair: offset of its phase block in the flat source
    int64_t ph_params = 0;
    float *d_wph = nullptr;                 // packed phase blocks (layout of the amplitude rows)
    float *d_w = nullptr;                   // [amp params | packed phase layers]
    unsigned short *d_wh = nullptr;         // phase layers as 3 bf16 planes (phase_kernel_bf16x3)
    int64_t wh_elems = 0;
    unsigned short *d_wamp = nullptr;       // amplitude blocks as bf16x3 MFMA fragments (phase kernel prologue); null: amp_kernel
    int64_t w_floats = 0;
    float *d_scratch = nullptr;             // [P][cap_M] log-amplitude contributions
    int64_t cap_M = 0;
    void *d_samp = nullptr;                 // tree-sampler scratch (naqs_sample.hip), sized for samp_cap unique prefixes
    int64_t samp_cap = 0;
    uint32_t samp_seq = 0;                  // sampling calls so far: tags the per-workgroup scan words of the fused level kernel
    int cu_count = 256;
    bool wamp_fresh = false;                // d_wamp was packed from the current parameters
    bool have_weights = false;              // amplitude AND phase layers packed from the current parameters
    bool packed_f32 = false;                // the f32-MFMA weight tiles are current (only packed when phase_kernel will run)
    int packed_fmt = 0;                     // split format of d_wh: 1 = three bf16 planes, 2 = two scaled f16 planes
    naqs::PhaseRaw *d_raw = nullptr;        // partial weight maxima of the phase layers (net_bounds_kernel -> pack_net_kernel)
    naqs::PhaseScales *d_scales = nullptr;  // the f16x2 scales of the current weights
    unsigned short *d_wt = nullptr;            // the big layer's planes in phase_kernel_wt's order (W1's contraction index permuted per chunk)
    bool have_wt = false;                      // ... packed from the current parameters (naqs_net_set_weights packs them; a training step's re-pack does not)
    unsigned long long *d_ws_xchg = nullptr;
    size_t ws_xchg_words = 0;   // phase_kernel_ws<.., SPLIT>: the producers' partial rows (tag << 32 | float), [cu_count / 2][64]
    uint32_t ws_seq = 0;                       // call tag of the last split launch (0 = no word written yet)
    unsigned long long *d_sum_words = nullptr; // vmc_seed_delta_kernel<true>: the four weighted sums as eight tagged words (tag << 32 | half a double)
    uint32_t sums_seq = 0;
    uint32_t pack_seq = 0;                     // tag of the last re-pack's PhaseRaw words
    const float *pack_pending = nullptr;       // parameters whose phase share of the re-pack has not been started (naqs_vmc_step; naqs_pack.hpp)
    bool pack_pending_amp = false;             // ... and the amplitude blocks' share has not been started either (round 6)
    int amp_head_packed = 0;                   // leading pairs whose fragments the update's own launch packed from the parameters it had just
                                               // written (grad_finish_kernel's first workgroups; 0 after any other update / re-pack)
    int overlap_next_pack = 0;                 // naqs_vmc_step -> naqs_net_set_weights: 1 = amplitude jobs now, phase jobs pending; 2 = everything
                                               // pending (the next sampler call's first launch hosts the whole re-pack, its first workgroup
                                               // packing the fragments of its own levels' pairs itself)
    bool have_amp_weights = false;          // amplitude layers packed (naqs_net_set_amp_weights leaves the phase stale)
    float *d_gpart = nullptr;               // per-workgroup partial gradients (naqs_grad.hip)
    void *d_train = nullptr;                // phase activations / deltas / GEMM partials (naqs_phase_grad.hip)
    int64_t train_cap = 0;                  // rows the training scratch holds
    float *d_wb = nullptr;                  // phase weights row-major [N_pad64][K_pad64] per layer (backward GEMMs)
    bool have_wb = false;
    bool grad_attr_set = false, grad2_attr_set = false;
    naqs::EventRing prof;                   // HIP-event brackets around the log-psi kernel (naqs_net_prof_*)
    naqs::EventRing prof_samp;              // ... around the sampler's launches of a training step (naqs_net_prof_select(net, 1))
    int prof_which = 0;
    char last_kernel[96] = {0};             // naqs_net_last_kernel
    int64_t *h_info = nullptr;              // mapped host words the sampler publishes (M, overflow, call sequence number) to
    int64_t *d_info_alias = nullptr;        // their device address
    int64_t *d_info2 = nullptr;             // device words for the sampler's plain (M, overflow) output of those calls
    int64_t info_seq = 0;                   // sampling calls that published there
    naqs::PollHandle poll;                  // this handle's bounded-wait control block and error word (naqs_poll.hpp) ...
    const naqs::PollCtl *ctl = nullptr;     // ... = poll.dev: every kernel that polls gets it
    hipStream_t side_stream = nullptr;      // NAQS_TRAIN_SIDE_STREAM=1: the amplitude blocks' backward beside the phase MLP's; naqs_vmc_run: the
                                            // phase MLP's share of a step's backward pass, update and re-pack, beside the NEXT step's sampler
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // naqs_vmc_run (round 5): the next sampler call reads the amplitude blocks only, and its three launches leave the chip
    // almost empty for ~90 us — so the phase MLP's half of the backward pass, its reductions + Adam update and its re-pack
    // run on side_stream behind the amplitude blocks' backward, and the caller's stream goes straight on to the sampler.
    // Whoever needs the phase layers, the gradient or the parameters next waits for ev_phase_done first (net_flush_pack /
    // naqs_net_finish_pending); naqs_vmc_run itself does before it returns, so its callers never see the pending state.
    bool defer_phase = false;               // set by naqs_vmc_run around its steps (NAQS_DEFER_PHASE=0: never)
    bool phase_pending = false;             // side_stream work the caller's stream has not been ordered behind yet
    hipEvent_t ev_phase_done = nullptr;
    hipStream_t pack_stream = nullptr;      // the stream whose work (the update) a pending re-pack must follow
    // naqs_vmc_step: the training forward launched ahead of the host's look at M (naqs::SpecRows)
    int64_t spec_hint = 0;                  // unique samples of the last accepted draw (0: none yet)
    int last_form_kind = 0, last_form_rb = 0, last_form_split = 0;      // what net_logpsi_impl launched last
    int64_t spec_launched = 0, spec_hits = 0;                           // naqs_net_spec_counts
    // two runs per GPU (NAQS_SHARED_GPU=1 when the handle is created — the farm's `--per-gpu 2` sets it): this handle's sampler
    // calls take turns with the other handles' of the device (naqs_sample.hip: lookback_turn_begin)
    bool shared_gpu = false;
    bool turn_held = false;                 // this handle holds the device's look-back turn
    bool turn_caller_ends = false;          // ... and the sampler's caller ends it (sample_and_wait), not the sampler's entry point
    int64_t lookback_turns = 0;             // sampler calls of this handle that had to wait for another handle's (naqs_net_share_device reports it)
    bool hold_finish = false;               // in: the sampler leaves its finish job pending (fin_job) instead of launching it
    bool fin_pending = false;               // a finish job nobody has launched or hosted yet (naqs::net_sample_finish_flush)
    naqs::SampleFinishJob fin_job{};
};

namespace naqs {
struct ElocFeed;
// where the training forward leaves what the backward pass needs (all nullptr: inference)
struct PhaseSave {
    float *x = nullptr;                     // [M][x_ld] +-1 inputs of the phase block (columns >= 2(P-1) stay zero)
    int x_ld = 0;
    float *act[MAXL] = {};                  // post-ReLU activations of hidden layer l: [M][act_ld[l]]
    int act_ld[MAXL] = {};
    long long *clk = nullptr;               // NAQS_DEBUG_CLOCKS=1: [8 waves][16 marks] cycle counter of workgroup 0
};
// A forward pass launched BEFORE the host knows the table's size (naqs_vmc_step: behind the sampler's launches, while the host
// still polls for M): the kernel reads the row count from the sampler's level sizes `U` (U[P], or 0 when the overflow word
// U[MAXP + 1] is set; with host_finish the launch's first workgroup is the sampler's finish job, net->fin_job), the launch covers M rows (the
// caller's upper estimate) and the kernel form is the one `m_var` rows would get — the caller checks afterwards that the real M
// gets the same form and fits the launch, and launches again the ordinary way if not.  Only the wave-specialised form takes it
// (net_logpsi_impl returns NAQS_ERR_UNSUPPORTED otherwise, before launching anything).
struct SpecRows { const int64_t *U = nullptr; int P = 0; int64_t m_var = 0; bool host_finish = false; };
// the log-psi kernel form net_logpsi_impl chose last / would choose for M rows (kind: 0 other, 1 phase_kernel_ws)
struct PhaseForm { int kind = 0, rb = 0, split = 0; };
inline bool operator==(const PhaseForm &a, const PhaseForm &b) { return a.kind == b.kind && a.rb == b.rb && a.split == b.split; }
// naqs_logpsi.hip: amp_kernel + phase kernel -> (log|psi|, phase)
int net_logpsi_impl(naqs_net *net, int64_t M, const uint64_t *keys_dev, float *logpsi_dev, void *stream,
                    const ElocFeed &feed, const PhaseSave &save, const SpecRows *spec = nullptr);
PhaseForm net_logpsi_form(const naqs_net *net, int64_t M, bool training);
// naqs_phase_grad.hip: row-major padded copies of the phase weights for the backward GEMMs — described as jobs for the
// one packing launch of naqs_net_set_weights (allocates the destination on first use)
struct WbPackJobs {
    int n = 0;
    int64_t src_off[MAXL] = {};
    int32_t N[MAXL] = {}, K[MAXL] = {}, Np[MAXL] = {}, Kp[MAXL] = {};
    float *dst[MAXL] = {};
};
int net_backward_pack_jobs(naqs_net *net, WbPackJobs *jobs);
// naqs_logpsi.hip: order `s` behind whatever this handle still has in flight elsewhere (the deferred phase chain of
// naqs_vmc_run), and start a pending re-pack of the phase layers on it
int net_finish_pending(naqs_net *net, hipStream_t s);
// naqs_logpsi.hip: the amplitude blocks' share of a training step's re-pack, if it is still waiting for a launch to host it
// (naqs_pack.hpp) — for every reader of the amplitude rows / fragments that is not that launch
int net_flush_amp_pack(naqs_net *net, hipStream_t s);
// naqs_grad.hip: d/d theta sum_i g_i f(key_i) for one set of per-pair blocks (amplitude blocks, or the phase blocks of an
// aggregate-phase network with raw = 1); grad_dev receives n_block_params floats in state_dict order
// With `defer` the fixed-order reduction of the workgroups' partial sums is not launched but described there, for the
// caller's one launch that finishes the whole gradient (naqs_phase_grad.hip: grad_finish_kernel); `slot` picks the half of
// the partial-sum scratch (two block sets may be pending at once).
struct BlockReduceJob {
    int64_t count = 0, stride = 0;
    int n_partials = 0;
    const float *partial = nullptr;
};
int net_blocks_backward(naqs_net *net, const NetDims &d, const float *w, const int64_t *src_off, int64_t n_block_params,
                        int64_t M, const uint64_t *keys_dev, const float *g_dev, float *grad_dev, int raw, hipStream_t s,
                        BlockReduceJob *defer = nullptr, int slot = 0);
namespace ampbw { struct AmpSrc; }
int net_blocks_backward_plan(naqs_net *net, const NetDims &d, const int64_t *src_off, int64_t n_block_params, int64_t M, int slot,
                             BlockReduceJob *job, ampbw::AmpSrc *src);
int net_blocks_backward2(naqs_net *net, int64_t M, const uint64_t *keys_dev, const float *g_amp, const float *g_ph, int g_stride,
                         BlockReduceJob jobs[2], hipStream_t s);
// Adam (Kingma & Ba) on element i of a flat parameter vector, torch.optim.Adam's update rule (no amsgrad):
// m <- m + (1 - b1)(g - m); v <- b2 v + (1 - b2) g^2; p <- p - step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
struct AdamArgs {
    float *p = nullptr, *m = nullptr, *v = nullptr;      // p == nullptr: no update
    float step_size = 0, beta1 = 0, beta2 = 0, bc2_sqrt = 1, eps = 0, weight_decay = 0;
};
inline AdamArgs adam_args(float *p, float *m, float *v, double lr, double beta1, double beta2, double eps, double weight_decay,
                          int64_t step) {
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    AdamArgs a;
    a.p = p; a.m = m; a.v = v;
    a.step_size = (float)(lr / bc1); a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.bc2_sqrt = (float)std::sqrt(bc2);
    a.eps = (float)eps; a.weight_decay = (float)weight_decay;
    return a;
}
#if defined(__HIPCC__)
// (pi, m0, v0: the element's parameter and moments as loaded by the caller — so that it can have them in flight under other loads)
__device__ __forceinline__ float adam_update_loaded(const AdamArgs &a, const int64_t i, float gi, const float pi, const float m0, const float v0) {
    if (a.weight_decay != 0.0f) gi = fmaf(a.weight_decay, pi, gi);
    const float mi = m0 + (gi - m0) * (1.0f - a.beta1);
    const float vi = a.beta2 * v0 + (1.0f - a.beta2) * gi * gi;
    a.m[i] = mi;
    a.v[i] = vi;
    const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
    const float pn = pi - a.step_size * (mi / denom);
    a.p[i] = pn;
    return pn;
}
__device__ __forceinline__ float adam_update(const AdamArgs &a, const int64_t i, float gi) {      // -> the updated parameter
    return adam_update_loaded(a, i, gi, a.p[i], a.m[i], a.v[i]);
}
#endif
// naqs_sample.hip
// NAQS_SHARED_GPU / naqs_net_share_device: the device's turn for look-back launches (naqs_sample.hip).  The sampler takes it before
// its first look-back launch (net->turn_held); whoever waits for the draw ends it: sample_and_wait once the draw's size is known
// (it sets net->turn_caller_ends around the call), the plain sampler entry points once their stream has drained
void lookback_turn_begin(naqs_net *net);
void lookback_turn_end(naqs_net *net);
int net_sample_finish_flush(naqs_net *net, hipStream_t s);      // launch a pending finish job as a kernel of its own (no-op if none)
int net_info_alloc(naqs_net *net);
int net_sample_early(naqs_net *net, int64_t n_samples, uint64_t seed, int64_t max_unique, uint64_t *keys_dev, int64_t *counts_dev,
                     float *probs_dev, double *weights_dev, int64_t *info_dev, void *stream, int64_t *early, int64_t seq);
// naqs_logpsi.hip: grow the [P][M] scratch and launch amp_kernel (feed == nullptr: no E_loc hand-over)
int net_amp_forward(naqs_net *net, int64_t M, const uint64_t *keys_dev, hipStream_t s, const ElocFeed *feed = nullptr,
                    bool launch = true);      // launch = false: only grow the scratch
}  // namespace naqs
