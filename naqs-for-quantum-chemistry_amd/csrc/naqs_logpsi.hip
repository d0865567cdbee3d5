// naqs_logpsi.hip — fused teacher-forced evaluation  keys -> (log|psi|, phase)  of the orbital NADE
// on gfx950 (MI355X).  Semantics: src/naqs/network/nade.py:738-770 (+ helpers :417-630) and
// src/naqs/wavefunction.py:167-183, :397-414 of the reference, float32 like the reference.
//
// Two kernels per call (the PyTorch-eager formulation of the same thing is ~200 launches):
//
//   amp_kernel    one workgroup = 4 tiles of 64 samples x one orbital pair n; 4 waves per tile split the hidden units.
//                 The 2n prefix occupations are built from key bits, spin-ordered (nade.py:519-530)
//                 and kept in registers; the pair's MLP weights are staged once per workgroup in LDS and
//                 read back as broadcast 16-byte reads (scalar loads were measured 5x slower here: ten
//                 pairs' weight sets thrash the small scalar cache).  Symmetrisation (:585-586), the
//                 electron-budget mask (:426-474) and 0.5*log_softmax(2x) (activations.py:40-46)
//                 are fused; the log-amplitude of the realised outcome goes to scratch[n][i].
//   phase_kernel  the phase MLP (e.g. 18 -> 512 -> 512 -> 4) for a tile of BM samples per workgroup,
//                 every layer on the f32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32, the
//                 reference's precision).  Activations stay in LDS between layers (one buffer: the
//                 MFMA accumulators hold a layer's output until every wave has finished reading its
//                 input), weights are pre-tiled in the MFMA operand layout and stream from L2 as contiguous
//                 1 KiB wave loads, bias +
//                 ReLU are fused into the write-back, and the epilogue adds the N/2 log-amplitudes in
//                 fixed order and stores (log|psi|, phase).
//
// The 512x512 layer is GEMM-shaped (5.2 GFLOP for 10 000 samples) -> MFMA roofline; everything else
// is negligible next to it.

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "naqs_common.hpp"
#include "naqs_hash.hpp"
#include "naqs_net.hpp"
#include "naqs_amp_mfma.hpp"
#include "naqs_pack.hpp"

namespace {

using naqs::WAVE;
using naqs::DeviceGuard;
using naqs::ElocFeed;

using naqs::MAXP;
using naqs::MAXL;
using naqs::f32x4;
using naqs::NetDims;
using naqs::amp_partial;
using naqs::amp_finish;

// ------------------------------------------------------------------------------------------------
// amplitude conditionals
// ------------------------------------------------------------------------------------------------
#ifndef NAQS_AMP_TILES
#define NAQS_AMP_TILES 4
#endif
#ifndef NAQS_AMP_SPLIT
#define NAQS_AMP_SPLIT 4
#endif
constexpr int AMP_TILES = NAQS_AMP_TILES;   // 64-sample tiles per workgroup sharing one staged weight set
constexpr int AMP_SPLIT = NAQS_AMP_SPLIT;   // waves splitting the hidden units of one tile

// workgroup = AMP_TILES x AMP_SPLIT waves for one orbital pair n (blockIdx.y): the pair's packed weights are
// staged once, wave (t, q) runs hidden-unit slice q for the 64 samples of tile t, the AMP_SPLIT partial
// outputs of a tile meet in LDS in fixed order.  (The kernel is latency-bound: splitting the hidden units
// over more waves was measured 2x faster than giving each wave all of them.)
__device__ __forceinline__ void amp_body(const NetDims &d, const float *__restrict__ w, int64_t M,
                                         const uint64_t *__restrict__ keys, float *__restrict__ scratch, const ElocFeed &feed,
                                         const int raw, float (*s_part)[AMP_SPLIT][5][WAVE], float *s_w) {
    // raw: the blocks are phase blocks (aggregate_phase; d describes them: 4 outputs, no symmetry): the output of the
    // realised outcome goes to scratch as it is (nade.py:556-569) instead of through the conditional
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = wave / AMP_SPLIT, q = wave % AMP_SPLIT;
    const int n = blockIdx.y;            // workgroup-uniform
    const int nin = n == 0 ? 1 : 2 * n;
    const int S = (nin + 1 + 5 + 3) & ~3;
    {
        const int total = d.Ha * S + 8;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(w + d.amp_off[n]);      // offsets are multiples of 4 floats
        f32x4 *dst = reinterpret_cast<f32x4 *>(s_w);
        for (int e = threadIdx.x; e < total / 4; e += AMP_TILES * AMP_SPLIT * WAVE) dst[e] = src[e];
    }
    const int64_t i = ((int64_t)blockIdx.x * AMP_TILES + tile) * WAVE + lane;
    const uint64_t key = i < M ? keys[i] : 0ull;
    // fused log-psi + E_loc call: the (light) n == 0 workgroups also narrow the keys and build the hash table
    // of the E_loc stage, which then needs no prep kernel
    if (feed.tab != nullptr && n == 0 && q == 0 && i < M) {
        if (feed.key_bits == 32) naqs::feed_key<uint32_t>(feed, i, key);
        else naqs::feed_key<uint64_t>(feed, i, key);
    }
    // prefix occupations of pairs 0..n-1 and the realised outcome of pair n
    uint32_t abits = 0, bbits = 0;
    for (int k = 0; k < n; ++k) {
        abits |= (uint32_t)((key >> d.qa[k]) & 1ull) << k;
        bbits |= (uint32_t)((key >> d.qb[k]) & 1ull) << k;
    }
    const int occ = (int)((key >> d.qa[n]) & 1ull) + 2 * (int)((key >> d.qb[n]) & 1ull);
    // spin ordering: the string with the smaller index goes first (nade.py:399-405, :519-530); the phase blocks (raw) order
    // their inputs under use_phase_spin_sym
    const bool swap = (raw ? d.phase_sym != 0 : d.sym != 0) && abits > bbits;
    const uint32_t first = swap ? bbits : abits, second = swap ? abits : bbits;
    const int per = (d.Ha + AMP_SPLIT - 1) / AMP_SPLIT;
    const int j0 = min(d.Ha, q * per), j1 = min(d.Ha, j0 + per);
    float o[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    __syncthreads();                     // weights staged
    switch (n) {
#define CASE(NB) case NB: amp_partial<NB>(d, s_w, first, second, j0, j1, o); break;
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7)
        CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
#undef CASE
        default: break;
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) s_part[tile][q][c][lane] = o[c];
    __syncthreads();
    if (q == 0 && i < M) {
        const float *b2 = s_w + d.Ha * S;
        float t[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            float v = c < d.n_out_amp ? b2[c] : 0.0f;
#pragma unroll
            for (int u = 0; u < AMP_SPLIT; ++u) v += s_part[tile][u][c][lane];      // fixed order
            t[c] = v;
        }
        if (raw) {
            const int row = naqs::phase_out_row(d, occ);       // (-phase_sym: |01> and |10> share the middle output, nade.py:593-595)
            float ph = row == 0 ? t[0] : (row == 1 ? t[1] : (row == 2 ? t[2] : t[3]));
            // ... and the last block's phase carries the sign of the spin-exchanged partner (nade.py:597-610, :758-759)
            if (n == d.P - 1) ph += naqs::phase_sym_shift(d, abits | ((uint32_t)(occ & 1) << n), bbits | ((uint32_t)(occ >> 1) << n));
            scratch[(int64_t)n * M + i] = ph;
        } else {
            scratch[(int64_t)n * M + i] = amp_finish(d, n, t, abits, bbits, occ);
        }
    }
}

__global__ __launch_bounds__(AMP_TILES * AMP_SPLIT * WAVE) void amp_kernel(const NetDims d, const float *__restrict__ w,
                                                                           int64_t M, const uint64_t *__restrict__ keys,
                                                                           float *__restrict__ scratch, const ElocFeed feed,
                                                                           const int raw) {
    __shared__ float s_part[AMP_TILES][AMP_SPLIT][5][WAVE];
    extern __shared__ __attribute__((aligned(16))) float s_w[];   // this pair's packed rows + b2
    amp_body(d, w, M, keys, scratch, feed, raw, s_part, s_w);
}

// aggregate_phase: the amplitude blocks (blockIdx.z == 0) and the per-pair phase blocks (1, raw) of the same batch in ONE
// launch — two latency-bound launches of P workgroup columns each otherwise
__global__ __launch_bounds__(AMP_TILES * AMP_SPLIT * WAVE) void amp2_kernel(const NetDims d0, const float *__restrict__ w0,
                                                                            float *__restrict__ scratch0, const ElocFeed feed,
                                                                            const NetDims d1, const float *__restrict__ w1,
                                                                            float *__restrict__ scratch1, int64_t M,
                                                                            const uint64_t *__restrict__ keys) {
    __shared__ float s_part[AMP_TILES][AMP_SPLIT][5][WAVE];
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    if (blockIdx.z == 0) amp_body(d0, w0, M, keys, scratch0, feed, 0, s_part, s_w);
    else { const ElocFeed none{}; amp_body(d1, w1, M, keys, scratch1, none, 1, s_part, s_w); }
}

// aggregate_phase epilogue: (log|psi|, phase) = (sum_n conditional log-amplitudes, sum_n phases), pair 0 first (the
// order of the fused kernel's amplitude sum); on the fused log-psi + E_loc entry also psi in float64
__global__ __launch_bounds__(256) void agg_finish_kernel(const int P, const int64_t M, const float *__restrict__ s_amp,
                                                         const float *__restrict__ s_ph, float2 *__restrict__ out,
                                                         const ElocFeed feed) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float la = 0.0f, ph = 0.0f;
    for (int n = 0; n < P; ++n) { la += s_amp[(int64_t)n * M + i]; ph += s_ph[(int64_t)n * M + i]; }
    out[i] = make_float2(la, ph);
    if (feed.psi != nullptr) naqs::feed_psi(feed, i, la, ph);
}

// ------------------------------------------------------------------------------------------------
// phase MLP on the f32 matrix cores
// ------------------------------------------------------------------------------------------------
#ifndef NAQS_PH_THREADS
#define NAQS_PH_THREADS 512
#endif
#ifndef NAQS_PH_CBT
#define NAQS_PH_CBT 4
#endif
constexpr int PH_THREADS = NAQS_PH_THREADS;      // 8 waves = 2 per SIMD: one computes while the other waits on loads
constexpr int PH_WAVES = PH_THREADS / WAVE;
constexpr int CBT = NAQS_PH_CBT;                 // 16-column blocks per wave per pass
#ifndef NAQS_PH_DBUF
#define NAQS_PH_DBUF 1                           // f16x2: weight fragments of the next K chunk in flight under the MFMAs
#endif

// K loop of one linear layer for one wave: RB row blocks x NC column blocks of 16x16 outputs.
// Operand layout of v_mfma_f32_16x16x4_f32: A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15].
// One 16-byte load per lane covers 4 MFMAs: lane (m, kq) holds k = k0 + 4*kq + j for j = 0..3 on both
// operands, so MFMA j contracts {k0+j, k0+4+j, k0+8+j, k0+12+j} — over a 16-wide chunk every k is
// visited exactly once.
template <int RB, int NC>
__device__ __forceinline__ void mlp_load(const float *__restrict__ a_ptr, int ld, const float *__restrict__ w_ptr,
                                         int K_pad, int k0, f32x4 (&a)[RB], f32x4 (&b)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) b[c] = *reinterpret_cast<const f32x4 *>(w_ptr + ((size_t)c * K_pad + k0) * 16);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const f32x4 *>(a_ptr + rb * 16 * ld + k0);
}

template <int RB, int NC>
__device__ __forceinline__ void mlp_mfma(const f32x4 (&a)[RB], const f32x4 (&b)[NC], f32x4 (&acc)[RB][CBT]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][j], b[c][j], acc[rb][c], 0, 0, 0);
}

template <int RB, int NC>
__device__ __forceinline__ void mlp_accumulate(const float *__restrict__ a_ptr, int ld,
                                               const float *__restrict__ w_ptr, int K_pad, f32x4 (&acc)[RB][CBT]) {
    // register double buffering: the loads of chunk k+1 (L2 -> VGPR weights, LDS activations) are in
    // flight while the 4*RB*NC MFMAs of chunk k issue; K_pad is a multiple of 16, chunks are 16 wide
    f32x4 a0[RB], b0[NC], a1[RB], b1[NC];
    mlp_load<RB, NC>(a_ptr, ld, w_ptr, K_pad, 0, a0, b0);
    int k0 = 0;
    // sched_barrier(0): hipcc otherwise sinks the prefetch loads to the end of the MFMA block (seen in
    // the ISA), which puts their full L2 latency back on the critical path
    for (; k0 + 32 <= K_pad; k0 += 32) {
        mlp_load<RB, NC>(a_ptr, ld, w_ptr, K_pad, k0 + 16, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mlp_mfma<RB, NC>(a0, b0, acc);
        __builtin_amdgcn_sched_barrier(0);
        // unconditional (clamped to the last chunk): a branch here makes the compiler's s_waitcnt merge
        // conservative (vmcnt(0) on the just-issued loads)
        mlp_load<RB, NC>(a_ptr, ld, w_ptr, K_pad, min(k0 + 32, K_pad - 16), a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mlp_mfma<RB, NC>(a1, b1, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (k0 < K_pad) mlp_mfma<RB, NC>(a0, b0, acc);       // odd number of chunks: the last one is already loaded
}

// one linear layer for RB*16 rows held in LDS: buf[rows][ld] (K_pad valid floats per row) ->
// buf[rows][0..N_pad) = act(buf * W^T + b).  All waves of the workgroup call it together.  The outputs
// are written only after every wave has finished reading the input (the MFMA accumulators hold them
// meanwhile), so a single LDS buffer suffices; the host guarantees N_pad <= PH_WAVES*CBT*16.
template <int RB>
__device__ __forceinline__ void mlp_layer(float *__restrict__ buf, int ld, int K_pad, int N_pad,
                                          const float *__restrict__ W, const float *__restrict__ bias, bool relu,
                                          int wave, int lane) {
    const int m = lane & 15, kq = lane >> 4;
    const int ncb = N_pad >> 4;
    f32x4 acc[RB][CBT];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int c = 0; c < CBT; ++c) acc[rb][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ncb == 1) {
        // a layer with <= 16 outputs (the last one: 512 -> 4) has a single column block: instead of leaving it
        // to one wave (a serial tail of K/4 * RB MFMAs while 7 waves idle), the waves split K and the partial
        // tiles are summed through LDS in fixed order
        constexpr int BM = RB * 16;
        const int chunks = K_pad >> 4, cpw = (chunks + PH_WAVES - 1) / PH_WAVES;
        const int kbeg = min(chunks, wave * cpw) << 4, klen = (min(chunks, (wave + 1) * cpw) << 4) - kbeg;
        if (klen > 0)
            mlp_accumulate<RB, 1>(buf + m * ld + 4 * kq + kbeg, ld, W + (size_t)kbeg * 16 + lane * 4, klen, acc);
        __syncthreads();                                  // everyone is done reading the input
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[(wave * BM + rb * 16 + kq * 4 + r) * 16 + m] = acc[rb][0][r];
        __syncthreads();
        constexpr int PER = (BM * 16 + PH_THREADS - 1) / PH_THREADS;      // outputs per thread (BM*16 may exceed the block)
        float v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = u * PH_THREADS + wave * WAVE + lane;
            v[u] = 0.0f;
            if (t < BM * 16) {
                v[u] = bias[t & 15];
#pragma unroll
                for (int q = 0; q < PH_WAVES; ++q) v[u] += buf[q * BM * 16 + t];
                if (relu) v[u] = fmaxf(v[u], 0.0f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = u * PH_THREADS + wave * WAVE + lane;
            if (t < BM * 16) buf[(t >> 4) * ld + (t & 15)] = v[u];
        }
        __syncthreads();
        return;
    }
    const int cb0 = wave * CBT;
    const int my_cb = min(CBT, max(0, ncb - cb0));       // wave-uniform
    const float *a_ptr = buf + m * ld + 4 * kq;
    // weights are pre-tiled in the MFMA operand order: [column block][16-wide k chunk][lane][4 floats],
    // so one wave-wide 16-byte load is one contiguous 1 KiB block (a row-major [N][K] weight would make the
    // 16 rows of a load 2 KiB apart: two L2 channels take all the traffic of every CU at once)
    const float *w_ptr = W + (size_t)cb0 * K_pad * 16 + lane * 4;
    if (my_cb == CBT) mlp_accumulate<RB, CBT>(a_ptr, ld, w_ptr, K_pad, acc);
    else if (my_cb >= 4) mlp_accumulate<RB, (CBT > 4 ? 4 : 1)>(a_ptr, ld, w_ptr, K_pad, acc);   // partial passes: rare
    else if (my_cb == 3) mlp_accumulate<RB, (CBT > 3 ? 3 : 1)>(a_ptr, ld, w_ptr, K_pad, acc);
    else if (my_cb == 2) mlp_accumulate<RB, (CBT > 2 ? 2 : 1)>(a_ptr, ld, w_ptr, K_pad, acc);
    else if (my_cb == 1) mlp_accumulate<RB, 1>(a_ptr, ld, w_ptr, K_pad, acc);
    __syncthreads();                                      // everyone is done reading the input
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int c = 0; c < CBT; ++c) {
        if (c < my_cb) {
            const int col = (cb0 + c) * 16 + m;
            const float bv = bias[col];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[rb][c][r] + bv;
                    if (relu) v = fmaxf(v, 0.0f);
                    buf[(rb * 16 + kq * 4 + r) * ld + col] = v;
                }
        }
    }
    __syncthreads();
}

template <int RB>
__global__ __launch_bounds__(PH_THREADS) void phase_kernel(const NetDims d, const float *__restrict__ w, int64_t M,
                                                           const uint64_t *__restrict__ keys,
                                                           const float *__restrict__ scratch,
                                                           float2 *__restrict__ out, const ElocFeed feed) {
    extern __shared__ __attribute__((aligned(16))) float buf[];
    constexpr int BM = RB * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int P = d.P, ld = d.ld;

    // log-amplitude sum of this thread's row: issued now, consumed in the epilogue (latency hidden under the MLP)
    float la = 0.0f;
    if (tid < BM && row0 + tid < M)
        for (int n = 0; n < P; ++n) la += scratch[(int64_t)n * M + row0 + tid];   // fixed order: block 0..P-1

    // layer-0 input: [alpha occupations of pairs 0..P-2 | beta ...] as +-1 (no spin ordering for the
    // phase block, nade.py:531-537), zero-padded to K_pad[0]; rows past M are zero
    const int K0 = d.K_pad[0];
    for (int e = tid; e < BM * K0; e += PH_THREADS) {
        const int r = e / K0, k = e - r * K0;
        const int64_t i = row0 + r;
        float v = 0.0f;
        if (i < M && k < 2 * (P - 1)) {
            const uint64_t key = keys[i];
            if (d.phase_sym) {                          // spin-ordered inputs (nade.py:519-530)
                uint32_t a_, b_;
                naqs::key_strings(d, key, a_, b_);
                naqs::phase_order_inputs(d, P - 1, a_, b_);
                v = (k < P - 1 ? ((a_ >> k) & 1u) : ((b_ >> (k - (P - 1))) & 1u)) ? 1.0f : -1.0f;
            } else {
                const int q = k < P - 1 ? d.qa[k] : d.qb[k - (P - 1)];
                v = ((key >> q) & 1ull) ? 1.0f : -1.0f;
            }
        }
        buf[r * ld + k] = v;
    }
    __syncthreads();

    for (int l = 0; l < d.n_lin; ++l)
        mlp_layer<RB>(buf, ld, d.K_pad[l], d.N_pad[l], w + d.w_off[l], w + d.b_off[l], l + 1 < d.n_lin, wave, lane);

    // epilogue: phase of the realised outcome of the last pair + sum of the log-amplitudes
    if (tid < BM) {
        const int64_t i = row0 + tid;
        if (i < M) {
            const uint64_t key = keys[i];
            const int occ = (int)((key >> d.qa[P - 1]) & 1ull) + 2 * (int)((key >> d.qb[P - 1]) & 1ull);
            float ph = buf[tid * ld + naqs::phase_out_row(d, occ)];
            if (d.phase_sym) { uint32_t a_, b_; naqs::key_strings(d, key, a_, b_); ph += naqs::phase_sym_shift(d, a_, b_); }
            out[i] = make_float2(la, ph);
            if (feed.psi != nullptr) naqs::feed_psi(feed, i, la, ph);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// phase MLP on the bf16 matrix cores with f32-equivalent accuracy ("bf16x3"): every f32 value is split
// into three bf16 planes x = x1 + x2 + x3 (8 + 8 + 8 mantissa bits: exact), and a product is formed from
// the six cross terms a1b1 + a1b2 + a2b1 + a1b3 + a3b1 + a2b2 (each exact in f32; the three dropped
// terms are <= 2^-24 |ab|, below f32 rounding), accumulated in f32 by v_mfma_f32_16x16x32_bf16.
// bf16 MFMA is 16x the f32 MFMA rate, so 6 of them per product are still ~2.7x faster.
// ------------------------------------------------------------------------------------------------
using naqs::bf16x8;
using naqs::ushort_t;
using naqs::split3t_pair;
using naqs::relu1;
using naqs::amp_mfma_pair_elems;
using naqs::AmpFrag;
using naqs::amp_mfma_load;
using naqs::amp_mfma_item;

__device__ __forceinline__ ushort_t f32_to_bf16_rne(float x) {
    const uint32_t u = __float_as_uint(x);
    return (ushort_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_to_f32(ushort_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ void split3(float x, ushort_t &h1, ushort_t &h2, ushort_t &h3) {
    h1 = f32_to_bf16_rne(x);
    const float r1 = x - bf16_to_f32(h1);
    h2 = f32_to_bf16_rne(r1);
    const float r2 = r1 - bf16_to_f32(h2);
    h3 = f32_to_bf16_rne(r2);
}

// exact 3-way split by truncation (activations, on the critical path of every layer's write-back): each part keeps the
// top 8 significant bits of what is left, so h1 + h2 + h3 == x exactly like the rounding split, in a third of the
// instructions
__device__ __forceinline__ void split3t(float x, ushort_t &h1, ushort_t &h2, ushort_t &h3) {
    uint32_t u = __float_as_uint(x);
    h1 = (ushort_t)(u >> 16);
    const float r1 = x - __uint_as_float(u & 0xFFFF0000u);
    u = __float_as_uint(r1);
    h2 = (ushort_t)(u >> 16);
    const float r2 = r1 - __uint_as_float(u & 0xFFFF0000u);
    h3 = (ushort_t)(__float_as_uint(r2) >> 16);
}

using naqs::tile_col;              // naqs_pack.hpp

// The split formats of the phase MLP.  FMT 1 ("bf16x3"): three bf16 planes, six cross terms per product.  FMT 2 ("f16x2"):
// two f16 planes of the power-of-two-scaled value (naqs::PhaseScales) — hi = f16(s v) carries 11 significant bits, lo =
// f16(s v - hi) the next 11 (round-to-nearest at both levels: |s v - hi - lo| <= 2^-24 |s v|, an f32 rounding), and a product
// is hi hi + hi lo + lo hi: THREE MFMAs instead of six; the dropped lo lo is <= 2^-24 |ab|.  The scales keep every tensor
// high in the f16 range (largest entries at 2^13..2^15), so lo is a normal f16 for entries down to 2^-17 of the largest and
// degrades gracefully (absolute error 2^-25 on the scaled value) below that.
using naqs::f16x8;
using naqs::f16x2;
using naqs::split2_pair;
using naqs::split2;
using naqs::pow2_clamped;
using naqs::exp_of;
template <int FMT> struct fmt_planes { static constexpr int value = FMT == 2 ? 2 : 3; };

template <int FMT>
__device__ __forceinline__ f32x4 mfma_h(const bf16x8 &a, const bf16x8 &b, const f32x4 &c) {
    if constexpr (FMT == 2)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// A1: the activations have a single non-zero plane (layer 0: the +-1 / 0 inputs are exact in bf16 and f16) -> one term per
// weight plane
template <int RB, int NC, bool A1 = false, int FMT = 1>
__device__ __forceinline__ void mlp_accumulate_h(const ushort_t *__restrict__ a_ptr, int ldh, int plane_stride,
                                                 const ushort_t *__restrict__ w_ptr, int Kh_pad, size_t wplane,
                                                 f32x4 (&acc)[RB][CBT]) {
    // a_ptr: this lane's row/k-group inside plane 0 of the activation tile; w_ptr: this lane's 8 weights of
    // column block 0, chunk 0, plane 0.  Chunks are 32 wide (one MFMA K).  (Register double buffering of the weight
    // fragments was tried: 96 VGPRs of fragments on top of the accumulators spill — 89 VGPRs to scratch, +20 us; the
    // second wave of the SIMD is what hides the L2 latency here.  Also tried, round 2: fetching the fragments as two
    // half-chunks so that each load has half a chunk of MFMAs to land in, with the next chunk's activation fragments
    // prefetched from LDS — no faster (layer 1: 51.5k vs 49.7k cycles) and 58 VGPRs spilled around the loop: the loop
    // already runs at ~88 % of what v_mfma_f32_16x16x32_bf16 sustains (2 waves x 72 MFMAs x ~19 cycles per chunk).)
    constexpr int NP = fmt_planes<FMT>::value;
    constexpr int AP = A1 ? 1 : NP;
    auto load_b = [&](int k0, bf16x8 (&b)[NP][NC]) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                b[p][c] = *reinterpret_cast<const bf16x8 *>(w_ptr + p * wplane + ((size_t)c * Kh_pad + k0) * 16);
    };
    auto load_a = [&](int k0, bf16x8 (&a)[AP][RB]) {
#pragma unroll
        for (int p = 0; p < AP; ++p)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                a[p][rb] = *reinterpret_cast<const bf16x8 *>(a_ptr + p * plane_stride + rb * 16 * ldh + k0);
    };
    auto mma = [&](const bf16x8 (&a)[AP][RB], const bf16x8 (&b)[NP][NC]) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x4 v = acc[rb][c];
                // smallest terms first
                if constexpr (FMT == 2) {
                    if constexpr (!A1) v = mfma_h<2>(a[1 % AP][rb], b[0][c], v);
                    v = mfma_h<2>(a[0][rb], b[1][c], v);
                    v = mfma_h<2>(a[0][rb], b[0][c], v);
                } else if constexpr (A1) {
                    v = mfma_h<1>(a[0][rb], b[2 % NP][c], v);
                    v = mfma_h<1>(a[0][rb], b[1][c], v);
                    v = mfma_h<1>(a[0][rb], b[0][c], v);
                } else {
                    v = mfma_h<1>(a[1 % AP][rb], b[1][c], v);
                    v = mfma_h<1>(a[2 % AP][rb], b[0][c], v);
                    v = mfma_h<1>(a[0][rb], b[2 % NP][c], v);
                    v = mfma_h<1>(a[1 % AP][rb], b[0][c], v);
                    v = mfma_h<1>(a[0][rb], b[1][c], v);
                    v = mfma_h<1>(a[0][rb], b[0][c], v);
                }
                acc[rb][c] = v;
            }
    };
    if constexpr (FMT == 2 && NAQS_PH_DBUF) {
        // f16x2: two planes of weight fragments are 32 VGPRs per chunk — room to keep the NEXT chunk's in flight under this
        // chunk's MFMAs (with the three bf16 planes the same double buffer spilled, see above)
#if NAQS_PH_DBUF == 2
        bf16x8 b0[NP][NC], b1[NP][NC], a0[AP][RB], a1[AP][RB];
        load_b(0, b0);
        load_a(0, a0);
        int k0 = 0;
        for (; k0 + 64 <= Kh_pad; k0 += 64) {
            load_b(k0 + 32, b1);
            load_a(k0 + 32, a1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            load_b(min(k0 + 64, Kh_pad - 32), b0);
            load_a(min(k0 + 64, Kh_pad - 32), a0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (k0 < Kh_pad) mma(a0, b0);
#else
        bf16x8 b0[NP][NC], b1[NP][NC], a[AP][RB];
        load_b(0, b0);
        int k0 = 0;
        for (; k0 + 64 <= Kh_pad; k0 += 64) {
            load_b(k0 + 32, b1);
            load_a(k0, a);
            __builtin_amdgcn_sched_barrier(0);
            mma(a, b0);
            __builtin_amdgcn_sched_barrier(0);
            load_b(min(k0 + 64, Kh_pad - 32), b0);         // unconditional (clamped): a branch makes the s_waitcnt merge conservative
            load_a(k0 + 32, a);
            __builtin_amdgcn_sched_barrier(0);
            mma(a, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (k0 < Kh_pad) { load_a(k0, a); mma(a, b0); }     // odd number of chunks: the last one is already loaded
#endif
    } else {
        for (int k0 = 0; k0 < Kh_pad; k0 += 32) {
            bf16x8 a[AP][RB], b[NP][NC];
            load_b(k0, b);
            load_a(k0, a);
            mma(a, b);
        }
    }
}

// scales of one layer in the f16x2 format (all 1 for bf16x3, which carries unscaled values)
struct LayerScale { float c = 1.0f, sn = 1.0f, isn = 1.0f; };

// write-back of a wide hidden layer: the accumulators (+ bias, ReLU) go back to the LDS tile as the format's planes.
// FMT 2: the planes hold sn * h, computed as max(acc * c + sn * b, 0) (c, sn powers of two: the same rounding as the unscaled
// sum); bvs arrive pre-scaled.
template <int RB, bool SAVE, int FMT>
__device__ __forceinline__ void mlp_writeback_h(ushort_t *__restrict__ planes, int ldh, int N_pad, f32x4 (&acc)[RB][CBT],
                                                const float (&bvs)[CBT], int cb0, int my_cb, bool inter, int wave, int lane,
                                                float *__restrict__ save, int save_ld, int64_t row0, int64_t M,
                                                long long *clk, const LayerScale sc) {
    constexpr int BM = RB * 16;
    const int m = lane & 15, kg = lane >> 4;
    const int plane_stride = BM * ldh;
    if (clk != nullptr && blockIdx.x == 0 && lane == 0) clk[wave * 16 + 9] = clock64();
    __syncthreads();                                      // everyone is done reading the input
    if (clk != nullptr && blockIdx.x == 0 && lane == 0) clk[wave * 16 + 10] = clock64();
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg; bias + ReLU, then split into the planes.
    // Neighbouring lanes hold neighbouring columns: the even lane takes rows 0,1 of both columns and the odd lane rows
    // 2,3 (two DPP exchanges), so every LDS store is a full dword (two 16-bit parts) instead of a 2-byte store — half the
    // store instructions of the write-back and no sub-dword merging.
    if (inter && my_cb == CBT) {
        const int col0 = (cb0 >> 2) * 64 + 4 * m;         // this lane's four adjacent columns: tile c <-> col0 + c
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float h[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if constexpr (FMT == 2) h[c] = fmaxf(fmaf(acc[rb][c % CBT][r], sc.c, bvs[c % CBT]), 0.0f);
                    else h[c] = fmaxf(acc[rb][c % CBT][r] + bvs[c % CBT], 0.0f);
                }
                const int row = rb * 16 + kg * 4 + r;
                if (SAVE && save != nullptr && row0 + row < M) {
                    if constexpr (FMT == 2)
                        *reinterpret_cast<f32x4 *>(save + (row0 + row) * save_ld + col0) =
                            (f32x4){h[0] * sc.isn, h[1] * sc.isn, h[2] * sc.isn, h[3] * sc.isn};
                    else
                        *reinterpret_cast<f32x4 *>(save + (row0 + row) * save_ld + col0) = (f32x4){h[0], h[1], h[2], h[3]};
                }
                ushort_t *dst = planes + row * ldh + col0;                              // 8-byte aligned
                if constexpr (FMT == 2) {
                    uint32_t a1, a2, b1, b2;
                    split2_pair(h[0], h[1], a1, a2);
                    split2_pair(h[2], h[3], b1, b2);
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(a1, b1);
                    *reinterpret_cast<uint2 *>(dst + plane_stride) = make_uint2(a2, b2);
                } else {
                    uint32_t a1, a2, a3, b1, b2, b3;
                    split3t_pair(h[0], h[1], a1, a2, a3);
                    split3t_pair(h[2], h[3], b1, b2, b3);
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(a1, b1);
                    *reinterpret_cast<uint2 *>(dst + plane_stride) = make_uint2(a2, b2);
                    *reinterpret_cast<uint2 *>(dst + 2 * plane_stride) = make_uint2(a3, b3);
                }
            }
        __syncthreads();
        return;
    }
    uint32_t *planes32 = reinterpret_cast<uint32_t *>(planes);
    const int ldw = ldh >> 1, plane_w = plane_stride >> 1;
    const bool odd = m & 1;
#pragma unroll
    for (int c = 0; c < CBT; ++c) {
        if (c < my_cb) {
            const int col = (cb0 + c) * 16 + m;
            const float bv = bvs[c];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float h[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (FMT == 2) h[r] = fmaxf(fmaf(acc[rb][c][r], sc.c, bv), 0.0f);
                    else h[r] = fmaxf(acc[rb][c][r] + bv, 0.0f);
                    const int row = rb * 16 + kg * 4 + r;
                    if (SAVE && save != nullptr && row0 + row < M) save[(row0 + row) * save_ld + col] = FMT == 2 ? h[r] * sc.isn : h[r];
                }
                const float x0 = __shfl_xor(odd ? h[0] : h[2], 1, 64), x1 = __shfl_xor(odd ? h[1] : h[3], 1, 64);
                const float lo[2] = {odd ? x0 : h[0], odd ? x1 : h[1]}, hi[2] = {odd ? h[2] : x0, odd ? h[3] : x1};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int o = (rb * 16 + kg * 4 + (odd ? 2 : 0) + j) * ldw + (col >> 1);
                    if constexpr (FMT == 2) {
                        uint32_t w1, w2;
                        split2_pair(lo[j], hi[j], w1, w2);
                        planes32[o] = w1;
                        planes32[plane_w + o] = w2;
                    } else {
                        uint32_t w1, w2, w3;
                        split3t_pair(lo[j], hi[j], w1, w2, w3);
                        planes32[o] = w1;
                        planes32[plane_w + o] = w2;
                        planes32[2 * plane_w + o] = w3;
                    }
                }
            }
        }
    }
    __syncthreads();
}


// one layer: planes[NP][BM][ldh] (16-bit parts) -> planes (hidden layers) or f32 out[BM][16] in the same LDS (last layer)
template <int RB, bool SAVE, int FMT>
__device__ __forceinline__ void mlp_layer_h(ushort_t *__restrict__ planes, int ldh, int Kh_pad, int N_pad,
                                            const ushort_t *__restrict__ W, const float *__restrict__ bias, bool last,
                                            int wave, int lane, float *__restrict__ save, int save_ld,
                                            int64_t row0, int64_t M, long long *clk,
                                            bool in_plane0_only, const LayerScale sc) {
    // save != nullptr (training forward): the post-ReLU activations of this hidden layer also go to HBM
    // ([M][save_ld] float32) for the backward pass
    constexpr int BM = RB * 16;
    const int m = lane & 15, kg = lane >> 4;
    const int ncb = N_pad >> 4;
    const int plane_stride = BM * ldh;
    const size_t wplane = (size_t)N_pad * Kh_pad;
    f32x4 acc[RB][CBT];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int c = 0; c < CBT; ++c) acc[rb][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const ushort_t *a_ptr = planes + m * ldh + 8 * kg;
    if (ncb == 1) {
        // <= 16 outputs: the waves split K, partial tiles are summed through LDS in fixed order
        const int chunks = Kh_pad >> 5, cpw = (chunks + PH_WAVES - 1) / PH_WAVES;
        const int kbeg = min(chunks, wave * cpw) << 5, klen = (min(chunks, (wave + 1) * cpw) << 5) - kbeg;
        if (klen > 0)
            mlp_accumulate_h<RB, 1, false, FMT>(a_ptr + kbeg, ldh, plane_stride, W + (size_t)kbeg * 16 + lane * 8, klen, wplane, acc);
        // NB: inside mlp_accumulate_h the chunk offset is ((c*Kh_pad + k0)*16) with the *layer's* Kh_pad only for
        // c > 0; with NC == 1 the sub-range length may be passed as Kh_pad.
        __syncthreads();
        float *part = reinterpret_cast<float *>(planes);          // inputs are dead: reuse as f32 scratch
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(wave * BM + rb * 16 + kg * 4 + r) * 16 + m] = acc[rb][0][r];
        __syncthreads();
        constexpr int PER = (BM * 16 + PH_THREADS - 1) / PH_THREADS;
        float v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = u * PH_THREADS + wave * WAVE + lane;
            v[u] = 0.0f;
            if (t < BM * 16) {
                if constexpr (FMT == 2) {
#pragma unroll
                    for (int q = 0; q < PH_WAVES; ++q) v[u] += part[q * BM * 16 + t];
                    v[u] = fmaf(v[u], sc.c, bias[t & 15] * sc.sn);         // last layer: sn == 1 -> the unscaled output
                } else {
                    v[u] = bias[t & 15];
#pragma unroll
                    for (int q = 0; q < PH_WAVES; ++q) v[u] += part[q * BM * 16 + t];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = u * PH_THREADS + wave * WAVE + lane;
            if (t < BM * 16) {
                if (last) part[t] = v[u];                          // f32 [BM][16]
                else {
                    const float hv = fmaxf(v[u], 0.0f);
                    const int o = (t >> 4) * ldh + (t & 15);
                    if constexpr (FMT == 2) {
                        ushort_t h1, h2;
                        split2(hv, h1, h2);
                        planes[o] = h1; planes[plane_stride + o] = h2;
                    } else {
                        ushort_t h1, h2, h3;
                        split3t(hv, h1, h2, h3);
                        planes[o] = h1; planes[plane_stride + o] = h2; planes[2 * plane_stride + o] = h3;
                    }
                    if (SAVE && save != nullptr && row0 + (t >> 4) < M)
                        save[(row0 + (t >> 4)) * save_ld + (t & 15)] = FMT == 2 ? hv * sc.isn : hv;
                }
            }
        }
        __syncthreads();
        return;
    }
    const int cb0 = wave * CBT;
    const int my_cb = min(CBT, max(0, ncb - cb0));       // wave-uniform
    const bool inter = CBT == 4 && (N_pad & 63) == 0;     // interleaved column map (tile_col): my_cb == CBT for every wave
    float bvs[CBT];                                       // fetched under the MFMAs, not after the barrier
#pragma unroll
    for (int c = 0; c < CBT; ++c) {
        bvs[c] = c < my_cb ? bias[tile_col(cb0 + c, m, N_pad)] : 0.0f;
        if constexpr (FMT == 2) bvs[c] *= sc.sn;
    }
    const ushort_t *w_ptr = W + (size_t)cb0 * Kh_pad * 16 + lane * 8;
    if (my_cb == CBT) {
        // (layer 0's fragments were also requested ahead, from the prologue, so that their first-touch latency would be
        // off the critical path: the 48 registers they hold across the conditionals spill — layer 0 got 1.8k cycles
        // shorter and the kernel 3.9k longer, A/B on one box)
        if (in_plane0_only) mlp_accumulate_h<RB, CBT, true, FMT>(a_ptr, ldh, plane_stride, w_ptr, Kh_pad, wplane, acc);
        else mlp_accumulate_h<RB, CBT, false, FMT>(a_ptr, ldh, plane_stride, w_ptr, Kh_pad, wplane, acc);
    }
    else if (my_cb >= 4) mlp_accumulate_h<RB, (CBT > 4 ? 4 : 1), false, FMT>(a_ptr, ldh, plane_stride, w_ptr, Kh_pad, wplane, acc);
    else if (my_cb == 3) mlp_accumulate_h<RB, (CBT > 3 ? 3 : 1), false, FMT>(a_ptr, ldh, plane_stride, w_ptr, Kh_pad, wplane, acc);
    else if (my_cb == 2) mlp_accumulate_h<RB, (CBT > 2 ? 2 : 1), false, FMT>(a_ptr, ldh, plane_stride, w_ptr, Kh_pad, wplane, acc);
    else if (my_cb == 1) mlp_accumulate_h<RB, 1, false, FMT>(a_ptr, ldh, plane_stride, w_ptr, Kh_pad, wplane, acc);
    mlp_writeback_h<RB, SAVE, FMT>(planes, ldh, N_pad, acc, bvs, cb0, my_cb, inter, wave, lane, save, save_ld, row0, M, clk, sc);
}

// layer 0 of the published shape (one K chunk, all eight waves own CBT column tiles): its weight fragments are requested
// by the caller BEFORE the input tile is built, so their L2 round trip overlaps the build instead of following it
template <int RB, int NP>
__device__ __forceinline__ void mlp_layer0_fetch(const NetDims &d, const ushort_t *__restrict__ W, int wave, int lane,
                                                 bf16x8 (&pre)[NP][CBT]) {
    const int Kh_pad = d.Kh_pad[0];
    const size_t wplane = (size_t)d.N_pad[0] * Kh_pad;
    const ushort_t *w_ptr = W + (size_t)(wave * CBT) * Kh_pad * 16 + lane * 8;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int c = 0; c < CBT; ++c) pre[p][c] = *reinterpret_cast<const bf16x8 *>(w_ptr + p * wplane + (size_t)c * Kh_pad * 16);
}
template <int RB, bool SAVE, int FMT>
__device__ __forceinline__ void mlp_layer0_pre(ushort_t *__restrict__ planes, int ldh, int N_pad,
                                               const bf16x8 (&pre)[fmt_planes<FMT>::value][CBT],
                                               const float *__restrict__ bias, int wave, int lane, float *__restrict__ save,
                                               int save_ld, int64_t row0, int64_t M, long long *clk, const LayerScale sc) {
    constexpr int NP = fmt_planes<FMT>::value;
    const int m = lane & 15, kg = lane >> 4;
    const int cb0 = wave * CBT;
    float bvs[CBT];
#pragma unroll
    for (int c = 0; c < CBT; ++c) {
        bvs[c] = bias[tile_col(cb0 + c, m, N_pad)];
        if constexpr (FMT == 2) bvs[c] *= sc.sn;
    }
    const ushort_t *a_ptr = planes + m * ldh + 8 * kg;
    f32x4 acc[RB][CBT];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(a_ptr + rb * 16 * ldh);
#pragma unroll
        for (int c = 0; c < CBT; ++c) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = NP - 1; p >= 0; --p) v = mfma_h<FMT>(a, pre[p][c], v);       // smallest terms first
            acc[rb][c] = v;
        }
    }
    mlp_writeback_h<RB, SAVE, FMT>(planes, ldh, N_pad, acc, bvs, cb0, CBT, CBT == 4 && (N_pad & 63) == 0, wave, lane, save, save_ld,
                                   row0, M, clk, sc);
}

// all (tile, pair) items of the workgroup, dealt round-robin to its waves, the loads of item k+1 in flight under item k;
// then the conditionals (symmetrise, mask, log-softmax, gather) for all (pair, sample) at once -> s_lan
template <int CT, int RB>
__device__ __forceinline__ void amp_mfma_prologue(const NetDims &d, const ushort_t *__restrict__ wamp, int64_t M, int64_t row0,
                                                  const uint64_t *__restrict__ keys, const ElocFeed &feed, uint32_t *s_ab,
                                                  float (*s_lan)[RB * 16], ushort_t *__restrict__ planes, int tid,
                                                  long long *clk) {
    constexpr int BM = RB * 16;
    const int lane = tid & 63, wave = tid >> 6;
    float *s_o = reinterpret_cast<float *>(planes);                        // [P][BM][8] raw outputs
    const size_t pair_elems = amp_mfma_pair_elems(CT * 16);
    const int P = d.P, items = RB * P;
    // items in pair-major order (q = n * RB + t), a contiguous range per wave: consecutive items share the pair, so its
    // 12 KB of fragments are fetched once per wave (the CU's 64 B/clk vector-memory path is what bounds this stage) and
    // the next pair's are in flight while the current one is used
    const int q0 = items * wave / PH_WAVES, q1 = items * (wave + 1) / PH_WAVES;
    AmpFrag<CT> f0, f1;
    int na = q0 < q1 ? q0 / RB : -1, nb = -1;
    const int n_last = q0 < q1 ? (q1 - 1) / RB : -1;
    // the key load goes out BEFORE the 18 fragment loads of the first pair: vector loads return in order, so behind them
    // the first barrier would wait for the whole 147 KB the workgroup's waves request (~2.3 k cycles of the CU's 64 B/clk)
    uint64_t key = 0ull;
    if (tid < BM && row0 + tid < M) key = keys[row0 + tid];
    __builtin_amdgcn_sched_barrier(0);
    // (unconditional — a wave without items fetches pair 0 for nothing: inside a branch the compiler cannot count the
    // loads that follow the key's and waits for all of them)
    amp_mfma_load<CT>(wamp + (size_t)max(na, 0) * pair_elems, lane, f0);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < BM) {                      // model-order occupation strings of the tile's samples; E_loc hand-over of the key
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {                  // unrolled: the qa/qb look-ups are independent scalar loads
            if (k < P) {
                a |= (uint32_t)((key >> d.qa[k]) & 1ull) << k;
                b |= (uint32_t)((key >> d.qb[k]) & 1ull) << k;
            }
        }
        s_ab[tid] = a | (b << 16);
    }
    __syncthreads();
    if (clk != nullptr && blockIdx.x == 0 && lane == 0) clk[wave * 16 + 1] = clock64();
    for (int q = q0; q < q1; ++q) {
        const int n = q / RB, t = q - n * RB;
        const uint32_t ab = s_ab[t * 16 + (lane & 15)];
        float *outs = s_o + ((size_t)n * BM + t * 16) * 8;
        if (n == na) {
            if (nb < n && n_last > n) { nb = n + 1; amp_mfma_load<CT>(wamp + (size_t)nb * pair_elems, lane, f1); }
            __builtin_amdgcn_sched_barrier(0);
            amp_mfma_item<CT>(d, f0, n, ab, lane, outs);
        } else {
            if (na < n && n_last > n) { na = n + 1; amp_mfma_load<CT>(wamp + (size_t)na * pair_elems, lane, f0); }
            __builtin_amdgcn_sched_barrier(0);
            amp_mfma_item<CT>(d, f1, n, ab, lane, outs);
        }
        if (clk != nullptr && blockIdx.x == 0 && lane == 0 && q - q0 < 4) clk[wave * 16 + 11 + (q - q0)] = clock64();
    }
    if (clk != nullptr && blockIdx.x == 0 && lane == 0) clk[wave * 16 + 8] = clock64();
    // fused log-psi + E_loc call: narrow the key and insert it into the E_loc hash table.  Done here, by the threads of
    // the waves that got the fewest items (tid < BM: waves 0..), and not next to the key gather above: the insert is an
    // atomicCAS round trip to L2 that nothing in this kernel waits for, and in front of the first barrier it delayed
    // every wave's first item
    if (tid < BM && feed.tab != nullptr && row0 + tid < M) {
        const uint64_t key = keys[row0 + tid];
        if (feed.key_bits == 32) naqs::feed_key<uint32_t>(feed, row0 + tid, key);
        else naqs::feed_key<uint64_t>(feed, row0 + tid, key);
    }
    __syncthreads();
    for (int e = tid; e < P * BM; e += PH_THREADS) {
        const int n = e / BM, r = e - n * BM;
        float o[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) o[c] = s_o[(size_t)e * 8 + c];
        const uint32_t ab = s_ab[r], mask = (1u << n) - 1u;
        const int occ = (int)((ab >> n) & 1u) + 2 * (int)((ab >> (16 + n)) & 1u);
        s_lan[n][r] = naqs::amp_finish(d, n, o, ab & mask, (ab >> 16) & mask, occ);
    }
    if (clk != nullptr && blockIdx.x == 0 && lane == 0) clk[wave * 16 + 15] = clock64();
    __syncthreads();
}

// the same (tile, pair) items as a kernel of its own (NAQS_AMP_MODE=2): inside the phase kernel the prologue runs at
// 2 waves per SIMD on the CUs that own a tile and pays its own latencies (key gather, item round trips, conditionals)
// in full; here a wave owns one pair and AMPK_TG tiles of 16 samples, there are ceil(M/32) * P independent waves and
// ~4 of them share a SIMD, so those latencies overlap.  No workgroup barrier: everything is private to the wave.
// scratch[n][i] = conditional log-amplitude of sample i's outcome at pair n (summed by the phase kernel, block 0..P-1)
constexpr int AMPK_TG = 2, AMPK_WAVES = 4;
template <int CT>
__global__ __launch_bounds__(AMPK_WAVES * 64) __attribute__((amdgpu_waves_per_eu(CT == 8 ? 2 : 4))) void amp_mfma_kernel(const NetDims d, const ushort_t *__restrict__ wamp, int64_t M,
                                                                   const uint64_t *__restrict__ keys,
                                                                   float *__restrict__ scratch, const ElocFeed feed) {
    __shared__ __attribute__((aligned(16))) float s_outs[AMPK_WAVES][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = d.P;
    const int64_t g = (int64_t)blockIdx.x * AMPK_WAVES + wave;
    const int64_t grp = g / P;
    const int n = (int)(g - grp * P);
    const int64_t row0 = grp * (AMPK_TG * 16);
    if (row0 >= M) return;                                                       // wave-uniform
    float *outs = s_outs[wave];                                                  // [16][8] raw outputs
    AmpFrag<CT> f;
    amp_mfma_load<CT>(wamp + (size_t)n * amp_mfma_pair_elems(CT * 16), lane, f);
    const int64_t i = row0 + (lane & (AMPK_TG * 16 - 1));
    const uint64_t key = i < M ? keys[i] : 0ull;
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        if (k < P) {
            a |= (uint32_t)((key >> d.qa[k]) & 1ull) << k;
            b |= (uint32_t)((key >> d.qb[k]) & 1ull) << k;
        }
    }
    const uint32_t ab_all = a | (b << 16);
#pragma unroll
    for (int t = 0; t < AMPK_TG; ++t) {
        if (row0 + t * 16 < M) {                                                 // wave-uniform
            const uint32_t ab = (uint32_t)__shfl((int)ab_all, t * 16 + (lane & 15), 64);
            amp_mfma_item<CT>(d, f, n, ab, lane, outs);
            // outs is private to this wave: LDS operations of a wave complete in order, so a wave-level fence suffices
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int64_t i2 = row0 + t * 16 + lane;
            if (lane < 16 && i2 < M) {
                float o[5];
#pragma unroll
                for (int c = 0; c < 5; ++c) o[c] = outs[lane * 8 + c];
                const uint32_t mask = (1u << n) - 1u;
                const int occ = (int)((ab >> n) & 1u) + 2 * (int)((ab >> (16 + n)) & 1u);
                scratch[(int64_t)n * M + i2] = naqs::amp_finish(d, n, o, ab & mask, (ab >> 16) & mask, occ);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                                     // outs is rewritten by the next tile's item
        }
    }
    // fused log-psi + E_loc call: the waves of pair 0 narrow the keys and insert them into the E_loc hash table
    if (n == 0 && lane < AMPK_TG * 16 && feed.tab != nullptr && i < M) {
        if (feed.key_bits == 32) naqs::feed_key<uint32_t>(feed, i, key);
        else naqs::feed_key<uint64_t>(feed, i, key);
    }
}

// SAVE: training forward (inputs and hidden activations also go to HBM); a template parameter because the stores'
// address arithmetic and bounds branches are ~40 % of the write-back's instructions even when they are skipped.
// FMT: the split format of the phase MLP's operands (1 = bf16x3, 2 = f16x2, see mfma_h above); the amplitude prologue is
// bf16x3 in both.
template <int RB, bool SAVE, int FMT>
__global__ __launch_bounds__(PH_THREADS) void phase_kernel_h(const NetDims d, const float *__restrict__ w,
                                                             const ushort_t *__restrict__ wh, int64_t M,
                                                             const uint64_t *__restrict__ keys,
                                                             const float *__restrict__ scratch,
                                                             float2 *__restrict__ out, const ElocFeed feed,
                                                             const naqs::PhaseSave save,
                                                             const ushort_t *__restrict__ wamp,
                                                             const naqs::PhaseScales *__restrict__ scales) {
    extern __shared__ __attribute__((aligned(16))) ushort_t planes[];
    __shared__ uint32_t s_ab[RB * 16];                 // model-order occupation strings of the tile's samples
    __shared__ float s_lan[MAXP][RB * 16];             // conditional log-amplitudes, pair-major
    constexpr int BM = RB * 16;
    constexpr int NP = fmt_planes<FMT>::value;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: SGPR
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int P = d.P, ldh = d.ldh;

#define NAQS_MARK(idx) do { if (save.clk != nullptr && blockIdx.x == 0 && lane == 0) save.clk[wave * 16 + (idx)] = clock64(); } while (0)
    NAQS_MARK(0);
    float la = 0.0f;
    if (wamp == nullptr) {
        if (tid < BM && row0 + tid < M)
            for (int n = 0; n < P; ++n) la += scratch[(int64_t)n * M + row0 + tid];   // fixed order: block 0..P-1
    } else {
        // touch layer 0's weight planes now (one dword per 128-byte line): their first use is ~20 k cycles away and at
        // launch start every XCD's L2 misses them; the value only has to stay live until the prologue is over
        uint32_t warm0 = 0, warm1 = 0;
        {
            const uint32_t *w0 = reinterpret_cast<const uint32_t *>(wh + d.wh_off[0]);
            const int n_dw = NP * d.N_pad[0] * d.Kh_pad[0] / 2;
            if (tid * 32 < n_dw) warm0 = w0[tid * 32];
            if ((tid + PH_THREADS) * 32 < n_dw) warm1 = w0[(tid + PH_THREADS) * 32];
        }
        // amplitude conditionals of this tile on the matrix cores; the activation planes are still free -> scratch
        if (d.Ha == 64) amp_mfma_prologue<4, RB>(d, wamp, M, row0, keys, feed, s_ab, s_lan, planes, tid, save.clk);
        else amp_mfma_prologue<2, RB>(d, wamp, M, row0, keys, feed, s_ab, s_lan, planes, tid, save.clk);
        NAQS_MARK(2);
        asm volatile("" ::"v"(warm0), "v"(warm1));
        if (tid < BM)
            for (int n = 0; n < P; ++n) la += s_lan[n][tid];                           // fixed order: block 0..P-1
        __syncthreads();                                                                // scratch becomes the activation planes
    }
    NAQS_MARK(3);

    // layer 0's weight fragments go out now; they land while the input tile is built
    const bool pre0 = d.n_lin > 1 && d.Kh_pad[0] == 32 && d.N_pad[0] == PH_WAVES * CBT * 16;
    bf16x8 pre[NP][CBT];
    if (pre0) mlp_layer0_fetch<RB, NP>(d, wh + d.wh_off[0], wave, lane, pre);

    // layer-0 input (+-1 / 0: exact in bf16 and f16, the other planes are zero); one thread builds 8 consecutive inputs of a
    // row and stores them as one 16-byte LDS write per plane (ldh and Kh_pad are multiples of 8)
    constexpr uint32_t ONE_P = FMT == 2 ? 0x3C00u : 0x3F80u, ONE_M = FMT == 2 ? 0xBC00u : 0xBF80u;
    const int K0 = d.Kh_pad[0], G0 = K0 >> 3;
    for (int e = tid; e < BM * G0; e += PH_THREADS) {
        const int r = e / G0, k8 = (e - r * G0) << 3;
        const int64_t i = row0 + r;
        uint32_t pk[4] = {0u, 0u, 0u, 0u};
        if (i < M && k8 < 2 * (P - 1)) {
            uint32_t ab;
            if (wamp != nullptr) ab = s_ab[r];                                  // occupation strings already gathered in LDS
            else {
                const uint64_t key = keys[i];
                uint32_t a_ = 0, b_ = 0;
                for (int k = 0; k < P; ++k) { a_ |= (uint32_t)((key >> d.qa[k]) & 1ull) << k; b_ |= (uint32_t)((key >> d.qb[k]) & 1ull) << k; }
                ab = a_ | (b_ << 16);
            }
            if (d.phase_sym) {                                                  // spin-ordered inputs (nade.py:519-530)
                uint32_t a_ = ab & 0xffffu, b_ = ab >> 16;
                naqs::phase_order_inputs(d, P - 1, a_, b_);
                ab = a_ | (b_ << 16);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k8 + j;
                if (k < 2 * (P - 1)) {
                    const bool set = k < P - 1 ? ((ab >> k) & 1u) : ((ab >> (16 + k - (P - 1))) & 1u);
                    pk[j >> 1] |= (set ? ONE_P : ONE_M) << (16 * (j & 1));      // +1.0 / -1.0
                    if (SAVE && save.x != nullptr) save.x[i * save.x_ld + k] = set ? 1.0f : -1.0f;
                }
            }
        }
        ushort_t *dst = planes + r * ldh + k8;
        *reinterpret_cast<uint4 *>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
#pragma unroll
        for (int p = 1; p < NP; ++p) *reinterpret_cast<uint4 *>(dst + p * BM * ldh) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();

    NAQS_MARK(4);
    LayerScale sc;
    if constexpr (FMT == 2) { sc.c = scales->c[0]; sc.sn = scales->sn[0]; sc.isn = scales->isn[0]; }
    if (pre0) {
        mlp_layer0_pre<RB, SAVE, FMT>(planes, ldh, d.N_pad[0], pre, w + d.b_off[0], wave, lane, save.act[0], save.act_ld[0], row0, M,
                                      save.clk, sc);
        NAQS_MARK(5);
    }
    for (int l = pre0 ? 1 : 0; l < d.n_lin; ++l) {
        if constexpr (FMT == 2) { sc.c = scales->c[l]; sc.sn = scales->sn[l]; sc.isn = scales->isn[l]; }
        mlp_layer_h<RB, SAVE, FMT>(planes, ldh, d.Kh_pad[l], d.N_pad[l], wh + d.wh_off[l], w + d.b_off[l], l + 1 == d.n_lin, wave, lane,
                        l + 1 < d.n_lin ? save.act[l] : nullptr, save.act_ld[l], row0, M, l == 0 ? save.clk : nullptr,
                        /*in_plane0_only=*/l == 0, sc);
        NAQS_MARK(5 + l);
    }

    if (tid < BM) {
        const int64_t i = row0 + tid;
        if (i < M) {
            const uint64_t key = keys[i];
            const int occ = (int)((key >> d.qa[P - 1]) & 1ull) + 2 * (int)((key >> d.qb[P - 1]) & 1ull);
            float ph = reinterpret_cast<const float *>(planes)[tid * 16 + naqs::phase_out_row(d, occ)];
            if (d.phase_sym) { uint32_t a_, b_; naqs::key_strings(d, key, a_, b_); ph += naqs::phase_sym_shift(d, a_, b_); }
            out[i] = make_float2(la, ph);
            if (feed.psi != nullptr) naqs::feed_psi(feed, i, la, ph);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// phase_kernel_ws — the same tile of log psi with the workgroup's waves SPECIALISED during the big hidden layer (round 4).
// phase_kernel_h runs prologue -> layer 0 -> layer 1 -> layer 2 with all eight waves in step: the matrix cores idle through
// the amplitude prologue (14 k of the tile's 51 k cycles at N2 / 48 rows) and the VALU idles through layer 1.  The two only
// share the occupation strings, so here layer 1 belongs to waves 0-3 — one per SIMD, eight column tiles each: 32 RB
// accumulator registers x 3, every weight fragment re-requested right after its last use, i.e. 7/8 of a chunk (~1 k cycles)
// ahead of its next one, no second buffer — while waves 4-7, their SIMD partners, run the (tile, pair) amplitude items, the
// E_loc key hand-over and the conditionals of their own items.  The f16x2 format is what makes it fit: 96 accumulators + 64
// fragment registers + two sets of activation fragments stay under the 256 registers two waves per SIMD can have, and two
// planes of 48 rows leave LDS for the items' raw outputs beside the activation tile.  Same arithmetic in the same order as
// phase_kernel_h (chunks ascending, smallest term first; the items are the shared amp_mfma_item): bit-identical results.
// Shape: hidden layers of 512 units, first layer one K chunk (2 (P - 1) <= 32), f16x2 format, amplitude width <= 64.
// ------------------------------------------------------------------------------------------------
constexpr int WS_MW = 4;                         // matrix waves (0 .. WS_MW - 1); the others are the amplitude waves
constexpr int WS_NCT = 8;                        // column tiles of a matrix wave in the big layer: 4 x 8 x 16 = 512 columns

template <int RB, int DBG = 0, int NC = WS_NCT>
__device__ __forceinline__ void ws_accumulate(const ushort_t *__restrict__ a_ptr, int ldh, int plane_stride,
                                              const ushort_t *__restrict__ w_ptr, int Kh_pad, size_t wplane,
                                              f32x4 (&acc)[RB][NC]) {
    bf16x8 b[2][NC], a0[2][RB], a1[2][RB];
    auto load_b2 = [&](int k0, int c0) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c = c0; c < c0 + 2; ++c)
                b[p][c] = *reinterpret_cast<const bf16x8 *>(w_ptr + p * wplane + ((size_t)c * Kh_pad + (DBG == 2 ? 0 : k0)) * 16);
    };
    auto load_a = [&](int k0, bf16x8 (&a)[2][RB]) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                a[p][rb] = *reinterpret_cast<const bf16x8 *>(a_ptr + p * plane_stride + rb * 16 * ldh + k0);
    };
    auto mma2 = [&](const bf16x8 (&a)[2][RB], int c0) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int c = c0; c < c0 + 2; ++c) {
                f32x4 v = acc[rb][c];
                if constexpr (DBG == 1) {                       // (timing aid: the stream without the matrix work)
                    v[0] += __builtin_bit_cast(f32x4, b[0][c])[0] + __builtin_bit_cast(f32x4, b[1][c])[1] + __builtin_bit_cast(f32x4, a[0][rb])[0] + __builtin_bit_cast(f32x4, a[1][rb])[0];
                    acc[rb][c] = v;
                    continue;
                }
                v = mfma_h<2>(a[1][rb], b[0][c], v);            // smallest terms first (the order of mlp_accumulate_h)
                v = mfma_h<2>(a[0][rb], b[1][c], v);
                v = mfma_h<2>(a[0][rb], b[0][c], v);
                acc[rb][c] = v;
            }
    };
#pragma unroll
    for (int c0 = 0; c0 < NC; c0 += 2) load_b2(0, c0);
    load_a(0, a0);
    int k0 = 0;
    for (; k0 + 64 <= Kh_pad; k0 += 64) {
        load_a(k0 + 32, a1);
#pragma unroll
        for (int c0 = 0; c0 < NC; c0 += 2) {
            __builtin_amdgcn_sched_barrier(0);
            mma2(a0, c0);
            __builtin_amdgcn_sched_barrier(0);
            load_b2(k0 + 32, c0);
        }
        const int kn = min(k0 + 64, Kh_pad - 32);          // unconditional (clamped): a branch makes the s_waitcnt merge conservative
        load_a(kn, a0);
#pragma unroll
        for (int c0 = 0; c0 < NC; c0 += 2) {
            __builtin_amdgcn_sched_barrier(0);
            mma2(a1, c0);
            __builtin_amdgcn_sched_barrier(0);
            load_b2(kn, c0);
        }
    }
    if (k0 < Kh_pad) {                                      // odd number of chunks: the last one is already loaded
#pragma unroll
        for (int c0 = 0; c0 < NC; c0 += 2) mma2(a0, c0);
    }
}

// the training forward's saved activations: written once, read by a later launch
#ifndef NAQS_SAVE_NT
#define NAQS_SAVE_NT 1
#endif
__device__ __forceinline__ void save_store(f32x4 *ptr, const f32x4 val) {
#if NAQS_SAVE_NT
    __builtin_nontemporal_store(val, ptr);
#else
    *ptr = val;
#endif
}
// write-back of four interleaved column tiles (64 adjacent columns: a lane owns four adjacent ones, tile_col) of a matrix wave
template <int RB, bool SAVE, int NCT, int C0>
__device__ __forceinline__ void ws_writeback4(ushort_t *__restrict__ planes, int ldh, f32x4 (&acc)[RB][NCT],
                                              const float (&bvs)[NCT], int col0, int lane, float *__restrict__ save, int save_ld,
                                              int64_t row0, int64_t M, const LayerScale sc) {
    constexpr int BM = RB * 16;
    const int kg = lane >> 4;
    const int plane_stride = BM * ldh;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float h[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) h[c] = fmaxf(fmaf(acc[rb][C0 + c][r], sc.c, bvs[C0 + c]), 0.0f);
            const int row = rb * 16 + kg * 4 + r;
            if (SAVE && save != nullptr && row0 + row < M)
                save_store(reinterpret_cast<f32x4 *>(save + (row0 + row) * save_ld + col0),
                           (f32x4){h[0] * sc.isn, h[1] * sc.isn, h[2] * sc.isn, h[3] * sc.isn});
            ushort_t *dst = planes + row * ldh + col0;                              // 8-byte aligned
            uint32_t a1, a2, b1, b2;
            split2_pair(h[0], h[1], a1, a2);
            split2_pair(h[2], h[3], b1, b2);
            *reinterpret_cast<uint2 *>(dst) = make_uint2(a1, b1);
            *reinterpret_cast<uint2 *>(dst + plane_stride) = make_uint2(a2, b2);
        }
}

// the amplitude waves' share of a tile: items q0 .. q1 (pair-major, q = n RB + t), the E_loc key hand-over, then the
// conditionals of exactly those items — nothing here needs another wave, so there is no workgroup barrier inside
template <int CT, int RB>
__device__ __forceinline__ void ws_amp_work(const NetDims &d, const ushort_t *__restrict__ wamp, int64_t M, int64_t row0,
                                            const uint64_t *__restrict__ keys, const ElocFeed &feed, const uint32_t *s_ab,
                                            float (*s_lan)[RB * 16], float *__restrict__ s_o, AmpFrag<CT> &f0, int aw, int lane,
                                            long long *clk, int wave, int qlo, int qhi, bool do_feed) {
    // [qlo, qhi): the workgroup's items (all of them, or its half of a tile shared by two workgroups)
    constexpr int BM = RB * 16;
    constexpr int AW = PH_WAVES - WS_MW;
    const size_t pair_elems = amp_mfma_pair_elems(CT * 16);
    const int q0 = qlo + (qhi - qlo) * aw / AW, q1 = qlo + (qhi - qlo) * (aw + 1) / AW;
    AmpFrag<CT> f1;
    int na = q0 < q1 ? q0 / RB : -1, nb = -1;                   // f0 holds pair na (requested by the caller before the first barrier)
    const int n_last = q0 < q1 ? (q1 - 1) / RB : -1;
    for (int q = q0; q < q1; ++q) {
        const int n = q / RB, t = q - n * RB;
        const uint32_t ab = s_ab[t * 16 + (lane & 15)];
        float *outs = s_o + ((size_t)n * BM + t * 16) * 8;
        if (n == na) {
            if (nb < n && n_last > n) { nb = n + 1; amp_mfma_load<CT>(wamp + (size_t)nb * pair_elems, lane, f1); }
            __builtin_amdgcn_sched_barrier(0);
            amp_mfma_item<CT>(d, f0, n, ab, lane, outs);
        } else {
            if (na < n && n_last > n) { na = n + 1; amp_mfma_load<CT>(wamp + (size_t)na * pair_elems, lane, f0); }
            __builtin_amdgcn_sched_barrier(0);
            amp_mfma_item<CT>(d, f1, n, ab, lane, outs);
        }
    }
    if (clk != nullptr && blockIdx.x == 0 && lane == 0) clk[wave * 16 + 8] = clock64();
    // fused log-psi + E_loc call: narrow the key and insert it into the E_loc hash table (an atomicCAS round trip to L2
    // that nothing in this kernel waits for)
    {
        const int r = aw * WAVE + lane;
        if (do_feed && r < BM && feed.tab != nullptr && row0 + r < M) {
            const uint64_t key = keys[row0 + r];
            if (feed.key_bits == 32) naqs::feed_key<uint32_t>(feed, row0 + r, key);
            else naqs::feed_key<uint64_t>(feed, row0 + r, key);
        }
    }
    // the raw outputs of this wave's items were written by this wave: LDS operations of a wave complete in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = q0 * 16 + lane; e < q1 * 16; e += WAVE) {
        const int q = e >> 4, s = e & 15;
        const int n = q / RB, r = (q - n * RB) * 16 + s;
        float o[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) o[c] = s_o[((size_t)n * BM + r) * 8 + c];
        const uint32_t ab = s_ab[r], mask = (1u << n) - 1u;
        const int occ = (int)((ab >> n) & 1u) + 2 * (int)((ab >> (16 + n)) & 1u);
        s_lan[n][r] = naqs::amp_finish(d, n, o, ab & mask, (ab >> 16) & mask, occ);
    }
    if (clk != nullptr && blockIdx.x == 0 && lane == 0) clk[wave * 16 + 15] = clock64();
}

// SPLIT (small tables: twice the tiles still fit the chip, one workgroup per CU): the big layer of a tile is shared by TWO
// workgroups — it is bound by the 1 MB weight stream through ONE CU's vector-memory path whatever the tile's height, and a
// training step's ~1 200 rows are 75 tiles on 256 CUs.  Workgroup `tile` (the producer: lower index, dispatched first) takes
// columns 256 .. 511, workgroup `n_tiles + tile` (the consumer) columns 0 .. 255, the amplitude items and the epilogue; each
// matrix wave owns one group of 64 columns.  The amplitude items are shared too (lower pairs: consumer, upper: producer).  The
// producer's output-layer partial rows (BM x 4 floats) and its conditionals travel as 64-bit words (call tag << 32 | float
// bits; relaxed agent-scope store / polled load: value and flag are one word, so no fence and no L2 write-back — the
// sampler's look-back words, naqs_sample.hip); the consumer adds the partial rows last:
// ((g0 + g1) + (g2 + g3)) + ((g4 + g5) + (g6 + g7)), a fixed order, but not the unsplit kernel's (last-bit differences of the
// phase between the two forms; every other value — log|psi|, saved activations — is the same).
struct WsSplit {
    unsigned long long *xchg; uint32_t tag; const naqs::PollCtl *ctl;
    // naqs::SpecRows (launched before the host knew the row count): M = U[MAXP + 1] ? 0 : U[P]; nullptr: M is the argument.
    // fin.U != nullptr: workgroup 0 is the sampler's finish job and the tiles start at workgroup 1.
    const int64_t *spec_U; int spec_P;
    naqs::SampleFinishJob fin;
};

template <int RB, bool SAVE, bool SPLIT = false>
__global__ __launch_bounds__(PH_THREADS) void phase_kernel_ws(const NetDims d, const float *__restrict__ w,
                                                              const ushort_t *__restrict__ wh, int64_t M,
                                                              const uint64_t *__restrict__ keys,
                                                              const float *__restrict__ scratch, float2 *__restrict__ out,
                                                              const ElocFeed feed, const naqs::PhaseSave save,
                                                              const ushort_t *__restrict__ wamp,
                                                              const naqs::PhaseScales *__restrict__ scales, const int flags,
                                                              const WsSplit split) {
    // wamp == nullptr: the amplitude conditionals were computed by a kernel of their own (scratch[n][i]); waves 4-7 then
    // only wait at the barrier
    extern __shared__ __attribute__((aligned(16))) ushort_t planes[];
    __shared__ uint32_t s_ab[RB * 16];                 // model-order occupation strings of the tile's samples
    __shared__ float s_lan[MAXP][RB * 16];             // conditional log-amplitudes, pair-major
    __shared__ __attribute__((aligned(16))) float s_part[WS_MW][RB * 16][4];   // the output layer's partial rows, per matrix wave
    __shared__ float s_recv[SPLIT ? RB * 16 : 1][4];                           // SPLIT: the producer's partial rows
    constexpr int BM = RB * 16;
    constexpr int FMT = 2, NP = 2;
    constexpr int AW = PH_WAVES - WS_MW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: SGPR
    const int hosted = split.fin.U != nullptr ? 1 : 0;    // (launch-uniform)
    if (hosted && blockIdx.x == 0) {                      // the sampler's finish job rides in this launch (naqs_net.hpp)
        __shared__ int64_t s_fin[16];
        naqs::sample_finish_body(split.fin, s_fin);
        return;
    }
    const int bx = (int)blockIdx.x - hosted;
    const int n_tiles = SPLIT ? (int)((gridDim.x - hosted) >> 1) : (int)gridDim.x - hosted;
    const bool producer = SPLIT && bx < n_tiles;                                // (workgroup-uniform)
    const int tile = SPLIT && !producer ? bx - n_tiles : bx;
    const int64_t row0 = (int64_t)tile * BM;
    if (split.spec_U != nullptr) {                        // launched ahead of the host's look at the sampler's M (workgroup-uniform)
        // never more rows than the launch was sized for (the argument: the host's cover) — the scratch behind `save`, `feed`
        // and `out` is only guaranteed for that many; a table that outgrew the cover is a miss the host relaunches anyway
        const int64_t m_dev = split.spec_U[MAXP + 1] != 0 ? 0 : split.spec_U[split.spec_P];
        M = m_dev < M ? m_dev : M;
        if (row0 >= M) return;                            // (both halves of a shared tile leave)
    }
    const int P = d.P, ldh = d.ldh;
    float *s_o = reinterpret_cast<float *>(planes + (size_t)NP * BM * ldh);     // [P][BM][8] raw outputs of the items
    const bool ha64 = d.Ha == 64;

    NAQS_MARK(0);
    // touch layer 0's weight planes (one dword per 128-byte line): at launch start every XCD's L2 misses them
    uint32_t warm0 = 0, warm1 = 0;
    {
        const uint32_t *w0 = reinterpret_cast<const uint32_t *>(wh + d.wh_off[0]);
        const int n_dw = NP * d.N_pad[0] * d.Kh_pad[0] / 2;
        if (tid * 32 < n_dw) warm0 = w0[tid * 32];
        if ((tid + PH_THREADS) * 32 < n_dw) warm1 = w0[(tid + PH_THREADS) * 32];
    }
    // the tile's keys are gathered by wave 0 (BM <= 64); only the amplitude waves request the fragments of their first pair —
    // in a branch of their own, so that nothing wave 0 waits for sits behind them (vector loads return in order, and every
    // wave fetching 12 KB of fragments it may never use delayed layer 0's own weight fetch)
    static_assert(BM <= WAVE, "the tile's rows are gathered by wave 0");
    uint64_t key = 0ull;
    if (tid < BM && row0 + tid < M) key = keys[row0 + tid];
    const int aw = wave >= WS_MW ? wave - WS_MW : 0;
    const int items = RB * P;
    // a shared tile's items: the lower pairs with the consumer (which also has the hand-over and the epilogue), the upper with
    // the producer, whose conditionals travel like its partial rows
    const int q_split = SPLIT ? (P / 2) * RB : items;
    const int qlo = producer ? q_split : 0, qhi = producer ? items : q_split;
    const int first_pair = (qlo + (qhi - qlo) * aw / AW) / RB;
    AmpFrag<4> f4;
    AmpFrag<2> f2;
    if (wamp != nullptr && wave >= WS_MW) {                // (wave-uniform)
        if (ha64) amp_mfma_load<4>(wamp + (size_t)first_pair * amp_mfma_pair_elems(64), lane, f4);
        else amp_mfma_load<2>(wamp + (size_t)first_pair * amp_mfma_pair_elems(32), lane, f2);
    }
    // layer 0's weight fragments go out now too (every wave owns four of its column tiles): they land behind the barrier's wait
    bf16x8 pre[NP][CBT];
    mlp_layer0_fetch<RB, NP>(d, wh + d.wh_off[0], wave, lane, pre);
    if (tid < BM) {                      // model-order occupation strings of the tile's samples
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            if (k < P) {
                a |= (uint32_t)((key >> d.qa[k]) & 1ull) << k;
                b |= (uint32_t)((key >> d.qb[k]) & 1ull) << k;
            }
        }
        s_ab[tid] = a | (b << 16);
    }
    __syncthreads();
    NAQS_MARK(1);

    // layer 0 (one K chunk): the +-1 inputs are built in registers as the A operand (lane (m, kg): sample m of the row block,
    // inputs 8 kg .. 8 kg + 7 as four f16 pairs; 2 (P - 1) is even: a pair is valid or not as a whole) — no input tile in LDS,
    // no barrier before the layer.  Same MFMAs in the same order as mlp_layer0_pre.
    const int m = lane & 15, kg = lane >> 4;
    LayerScale sc;
    sc.c = scales->c[0]; sc.sn = scales->sn[0]; sc.isn = scales->isn[0];
    {
        const int N0 = d.N_pad[0], cb0 = wave * CBT;
        float bvs0[CBT];
#pragma unroll
        for (int c = 0; c < CBT; ++c) bvs0[c] = (w + d.b_off[0])[tile_col(cb0 + c, m, N0)] * sc.sn;
        const uint32_t pmask = (1u << (P - 1)) - 1u;
        const int nv = min(max(2 * (P - 1) - 8 * kg, 0), 8) >> 1;
        f32x4 acc0[RB][CBT];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const uint32_t ab = s_ab[rb * 16 + m];
            const uint32_t tb = ((ab & pmask) | (((ab >> 16) & pmask) << (P - 1))) >> (8 * kg);
            const bool live = row0 + rb * 16 + m < M;                      // rows past M: zero inputs
            uint32_t aw4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t wv = 0xBC00BC00u ^ ((tb << (15 - 2 * j)) & 0x8000u) ^ ((tb << (30 - 2 * j)) & 0x80000000u);
                aw4[j] = (live && j < nv) ? wv : 0u;
            }
            bf16x8 a;
            __builtin_memcpy(&a, aw4, sizeof(a));
#pragma unroll
            for (int c = 0; c < CBT; ++c) {
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int p = NP - 1; p >= 0; --p) v = mfma_h<FMT>(a, pre[p][c], v);       // smallest terms first
                acc0[rb][c] = v;
            }
        }
        if (SAVE && save.x != nullptr && !producer) {                      // training forward: the inputs also go to HBM
            const int nin = 2 * (P - 1);
            for (int e = tid; e < BM * nin; e += PH_THREADS) {
                const int r = e / nin, k = e - r * nin;
                if (row0 + r < M) {
                    const uint32_t ab = s_ab[r];
                    const bool set = k < P - 1 ? ((ab >> k) & 1u) : ((ab >> (16 + k - (P - 1))) & 1u);
                    save.x[(row0 + r) * save.x_ld + k] = set ? 1.0f : -1.0f;
                }
            }
        }
        ws_writeback4<RB, SAVE, CBT, 0>(planes, ldh, acc0, bvs0, (cb0 >> 2) * 64 + 4 * m, lane, producer ? nullptr : save.act[0], save.act_ld[0],
                                        row0, M, sc);
    }
    asm volatile("" ::"v"(warm0), "v"(warm1));
    __syncthreads();
    NAQS_MARK(5);

    // the big layer: matrix waves | amplitude waves
    sc.c = scales->c[1]; sc.sn = scales->sn[1]; sc.isn = scales->isn[1];
    const int N1 = d.N_pad[1], Kh1 = d.Kh_pad[1];
    if (wave < WS_MW) {
        constexpr int NCT = SPLIT ? WS_NCT / 2 : WS_NCT;                   // column tiles of this wave: one or two groups of 64 columns
        constexpr int NG = NCT / 4;
        f32x4 acc[RB][NCT];
        float bvs[NCT];
        if (flags & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[rb][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int cb0 = (producer ? WS_MW * NCT : 0) + wave * NCT;
        const float *bias = w + d.b_off[1];
#pragma unroll
        for (int c = 0; c < NCT; ++c) bvs[c] = bias[tile_col(cb0 + c, m, N1)] * sc.sn;
        // (developer aid: bits 4 / 8 skip the amplitude / the matrix work, 16 / 32 the MFMAs / the stream — timing only, wrong results)
        const ushort_t *a_ptr = planes + m * ldh + 8 * kg, *w_ptr = wh + d.wh_off[1] + (size_t)cb0 * Kh1 * 16 + lane * 8;
        if (flags & 8) {}
#ifdef NAQS_WS_DEBUG
        else if (flags & 16) ws_accumulate<RB, 1, NCT>(a_ptr, ldh, BM * ldh, w_ptr, Kh1, (size_t)N1 * Kh1, acc);
        else if (flags & 32) ws_accumulate<RB, 2, NCT>(a_ptr, ldh, BM * ldh, w_ptr, Kh1, (size_t)N1 * Kh1, acc);
#endif
        else ws_accumulate<RB, 0, NCT>(a_ptr, ldh, BM * ldh, w_ptr, Kh1, (size_t)N1 * Kh1, acc);
        if (flags & 1) __builtin_amdgcn_s_setprio(0);
        NAQS_MARK(9);
        // the output layer (512 -> 4) straight from the accumulators, in f32: h = max(acc c + sn b, 0) is sn x the hidden
        // activation; this lane holds it for 12 RB / 3 rows x 8 columns (two groups of four adjacent ones) and adds its
        // columns' products with the four output rows of W (plain row-major f32, packed by pack_net_kernel); the 16 lanes of
        // a row then add up (four DPP exchanges), and the four matrix waves' partial rows meet in LDS — fixed order
        // throughout.  No write-back of the 512-wide activations, no second pass over them, no barrier in between.
        const float *W2 = w + d.w_off[2];
        const int K2 = d.K_pad[2];
        const int g0 = cb0 >> 2;
        f32x4 wv[NG][4];                                                   // [group][output]: four adjacent columns each
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int o = 0; o < 4; ++o) wv[g][o] = *reinterpret_cast<const f32x4 *>(W2 + (size_t)o * K2 + (g0 + g) * 64 + 4 * m);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rb * 16 + kg * 4 + r;
                float po[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float h[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) h[c] = fmaxf(fmaf(acc[rb][4 * g + c][r], sc.c, bvs[4 * g + c]), 0.0f);
                    if (SAVE && save.act[1] != nullptr && row0 + row < M)
                        save_store(reinterpret_cast<f32x4 *>(save.act[1] + (row0 + row) * save.act_ld[1] + (g0 + g) * 64 + 4 * m),
                                   (f32x4){h[0] * sc.isn, h[1] * sc.isn, h[2] * sc.isn, h[3] * sc.isn});
#pragma unroll
                    for (int o = 0; o < 4; ++o)
#pragma unroll
                        for (int c = 0; c < 4; ++c) po[o] = fmaf(h[c], wv[g][o][c], po[o]);
                }
                // (all four outputs are reduced although only the realised outcome's is used: selecting it first needs the row's
                // occupation string from LDS per row — measured 5.8 k cycles for this stage against 3.9 k)
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float v = po[o];
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
                    po[o] = v;
                }
                if (m == 0) *reinterpret_cast<f32x4 *>(&s_part[wave][row][0]) = (f32x4){po[0], po[1], po[2], po[3]};
            }
    } else {
        if (flags & 2) __builtin_amdgcn_s_setprio(1);
        if ((flags & 4) || wamp == nullptr) {}
        else if (ha64) ws_amp_work<4, RB>(d, wamp, M, row0, keys, feed, s_ab, s_lan, s_o, f4, aw, lane, save.clk, wave, qlo, qhi, !producer);
        else ws_amp_work<2, RB>(d, wamp, M, row0, keys, feed, s_ab, s_lan, s_o, f2, aw, lane, save.clk, wave, qlo, qhi, !producer);
        if (flags & 2) __builtin_amdgcn_s_setprio(0);
    }
    NAQS_MARK(10);
    __syncthreads();                                      // the output partials and s_lan are complete
    NAQS_MARK(6);

    if constexpr (SPLIT) {
        // words of a tile: [BM][4] partial rows of the output layer, then [pairs of the producer][BM] conditional log-amplitudes
        constexpr int TILE_WORDS = BM * 4 + MAXP * BM;
        unsigned long long *xw = split.xchg + (size_t)tile * TILE_WORDS;
        const int n_split = q_split / RB, n_amp = wamp != nullptr ? (P - n_split) * BM : 0;
        const unsigned long long tagw = (unsigned long long)split.tag << 32;
        if (producer) {                                   // hand the upper columns' partial rows and the upper pairs to the consumer, and done
            for (int x = tid; x < BM * 4 + n_amp; x += PH_THREADS) {
                float v;
                if (x < BM * 4) {
                    const int r = x >> 2, o = x & 3;
                    v = (s_part[0][r][o] + s_part[1][r][o]) + (s_part[2][r][o] + s_part[3][r][o]);
                } else {
                    const int e = x - BM * 4;
                    v = s_lan[n_split + e / BM][e % BM];
                }
                if (!(naqs::poll_drop(split.ctl, naqs::POLL_LOGPSI_SPLIT) && tile == 0 && x == 0))      // (debug knob: naqs_poll.hpp)
                    __hip_atomic_store(&xw[x], tagw | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        // the consumer: a word per thread (and round), all requests in flight together.  The producer has a lower workgroup
        // index: it was dispatched before this workgroup, so it is running or has finished; the wait is bounded all the same
        // (naqs_poll.hpp): when it runs out the tile's results are not written and the host gets NAQS_ERR_HIP
        bool ok = true;
        for (int x = tid; x < BM * 4 + n_amp; x += PH_THREADS) {
            unsigned long long word;
            ok = naqs::poll_tagged<2>(&xw[x], split.tag, word, split.ctl, naqs::POLL_LOGPSI_SPLIT, (uint32_t)x) && ok;
            const float v = __uint_as_float((uint32_t)word);
            if (x < BM * 4) s_recv[x >> 2][x & 3] = v;
            else { const int e = x - BM * 4; s_lan[n_split + e / BM][e % BM] = v; }
        }
        if (__syncthreads_or(!ok)) return;
    }
    if (tid < BM) {
        const int64_t i = row0 + tid;
        if (i < M) {
            float la = 0.0f;
            if (wamp != nullptr) for (int n = 0; n < P; ++n) la += s_lan[n][tid];      // fixed order: block 0..P-1
            else for (int n = 0; n < P; ++n) la += scratch[(int64_t)n * M + i];
            const uint32_t ab = s_ab[tid];
            const int occ = (int)((ab >> (P - 1)) & 1u) + 2 * (int)((ab >> (16 + P - 1)) & 1u);
            float psum = (s_part[0][tid][occ] + s_part[1][tid][occ]) + (s_part[2][tid][occ] + s_part[3][tid][occ]);
            if constexpr (SPLIT) psum += s_recv[tid][occ];
            const float ph = fmaf(psum, sc.isn, (w + d.b_off[2])[occ]);
            out[i] = make_float2(la, ph);
            if (feed.psi != nullptr) naqs::feed_psi(feed, i, la, ph);
        }
    }
    NAQS_MARK(7);
}

// ------------------------------------------------------------------------------------------------
// phase_kernel_wt (round 5; NAQS_PHASE_WT=1 — NOT the default: measured no faster) — the big layer TRANSPOSED,
// H1^T = W1 . H0^T, with layer 0 produced just in time.
// phase_kernel_ws prices the big layer by the bytes of weight planes a CU streams (1 MB per 48-row workgroup whatever its rows:
// 27.9 k of the tile's 40 k cycles, the matrix pipe at 35 %); the only way to stream less per CU is the column split
// (two workgroups per tile, 0.5 MB each), and at 10^4 rows that needs tiles of 80 rows to stay within one round of 256
// workgroups — whose 160 KB activation tile does not exist.  Here it does not have to: the weights are the A operand (the packed
// fragments are symmetric in A / B: lane (index, k-group)), the activations the B operand, and a chunk of H0^T — 32 hidden
// units x 16 samples — comes out of layer 0's accumulators already in B-operand order if W1's contraction index is packed in
// the order those accumulators hold it (slot 8 g + 4 t + r of chunk c <- unit tile_col(2 c + t, 4 g + r): naqs_amp_mfma.hpp's
// second stage does the same).  The amplitude waves produce those chunks between their items — 20 MFMAs + bias / ReLU / split
// per chunk of 80 samples — into a ring of eight 10 KB chunks in LDS (flags in LDS, no fences: a wave's LDS operations execute
// in order), the matrix waves read their B fragments from it: nothing of the 512-wide activations is ever written as a tile,
// and a workgroup is 80 rows x 256 output units: 125 tiles x 2 halves = 250 workgroups for 10^4 rows.
// The output layer, the hand-over between the two halves and the epilogue are phase_kernel_ws<SPLIT>'s.
// Arithmetic: the same three-term f16x2 products in the same order of terms and chunks; the 32 products inside one MFMA meet in
// a different slot order and the output layer adds its columns in a different order: the phase differs from phase_kernel_ws in
// the last bits (4.5e-7 at most on the bench's table; <= 2e-5 of the PyTorch modules like every other form), log|psi| not at all.
// MEASURED (N2, 10^4 rows, cycle stamps of workgroup 0, tools/wt_check.py; phase_kernel_ws on the same box: 23.1 us):
//   * first form — every matrix wave recomputes layer 0 itself, fragments in registers: big layer 38.6 k cycles, 29.5 us;
//   * this form — big layer 24.3 k cycles (27.9 k in phase_kernel_ws), but the amplitude waves, which share their SIMDs'
//     issue with matrix waves that no longer stall on the weight stream, finish their items at 33-38 k instead of 27 k and end
//     the workgroup at 41 k cycles: 24.5 us.  The stream was what the MATRIX waves waited for, not what the workgroup waits
//     for: 1 128 MFMAs + the items' and the output stage's VALU work per SIMD are the same as before.
//   Stop rule of the review (big layer <= 20 k cycles) not met: kept selectable, off.
constexpr int WT_NT = 5;                          // sample tiles (16 rows each) of a workgroup
__host__ __device__ __forceinline__ int wt_in_unit(int c, int s) {       // slot s of chunk c of W1's contraction index -> hidden unit of layer 0
    const int g = s >> 3, t = (s >> 2) & 1, r = s & 3, cb = 2 * c + t, nn = 4 * g + r;
    return ((cb >> 2) << 6) + 4 * nn + (cb & 3);                          // = tile_col(cb, nn, 512)
}
// W1 [512 out][512 in] f32 -> two scaled f16 planes [plane][out tile 32][chunk 16][64 lanes][8]: lane (m, kg), slot 8 kg + j
__global__ __launch_bounds__(256) void pack_wt_kernel(const float *__restrict__ W1, const naqs::PhaseScales *__restrict__ scales,
                                                      ushort_t *__restrict__ wt) {
    constexpr int total = 512 * 512;
    const float sw = scales->sw[1];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int j = e & 7, lane = (e >> 3) & 63, c = (e >> 9) & 15, mt = e >> 13;
        const int out = mt * 16 + (lane & 15), in = wt_in_unit(c, 8 * (lane >> 4) + j);
        ushort_t h1, h2;
        split2(W1[out * 512 + in] * sw, h1, h2);
        wt[e] = h1; wt[total + e] = h2;
    }
}

// the amplitude waves' share of a phase_kernel_wt tile: ws_amp_work's items, hand-over and conditionals, with the wave's chunks
// of H0^T produced in between (`produce(k)`: the k-th of its WT_PROD chunks; the first before any item, then one every
// `every` items, so that a chunk is in the ring before the matrix waves get to it)
constexpr int WT_RING = 8;                        // chunks of H0^T the LDS ring holds (10 KB each at 80 rows)
constexpr int WT_PROD = 16 / (PH_WAVES - WS_MW);  // chunks an amplitude wave produces
template <int CT, int RB, typename Produce>
__device__ __forceinline__ void wt_amp_work(const NetDims &d, const ushort_t *__restrict__ wamp, int64_t M, int64_t row0,
                                            const uint64_t *__restrict__ keys, const ElocFeed &feed, const uint32_t *s_ab,
                                            float (*s_lan)[RB * 16], float *__restrict__ s_o, AmpFrag<CT> &f0, int aw, int lane,
                                            long long *clk, int wave, int qlo, int qhi, bool do_feed, Produce &&produce) {
    constexpr int BM = RB * 16;
    constexpr int AW = PH_WAVES - WS_MW;
    const size_t pair_elems = amp_mfma_pair_elems(CT * 16);
    const int q0 = qlo + (qhi - qlo) * aw / AW, q1 = qlo + (qhi - qlo) * (aw + 1) / AW;
    AmpFrag<CT> f1;
    int na = q0 < q1 ? q0 / RB : -1, nb = -1;
    const int n_last = q0 < q1 ? (q1 - 1) / RB : -1;
    // the ring holds eight chunks: a wave's first two go in at once, the others one item apart (each waits for the matrix
    // waves to have read the chunk whose slot it takes)
    int made = 0;
    produce(made++);
    produce(made++);
    for (int q = q0; q < q1; ++q) {
        if (made < WT_PROD && (q - q0) >= made - 1) produce(made++);
        const int n = q / RB, t = q - n * RB;
        const uint32_t ab = s_ab[t * 16 + (lane & 15)];
        float *outs = s_o + ((size_t)n * BM + t * 16) * 8;
        if (n == na) {
            if (nb < n && n_last > n) { nb = n + 1; amp_mfma_load<CT>(wamp + (size_t)nb * pair_elems, lane, f1); }
            __builtin_amdgcn_sched_barrier(0);
            amp_mfma_item<CT>(d, f0, n, ab, lane, outs);
        } else {
            if (na < n && n_last > n) { na = n + 1; amp_mfma_load<CT>(wamp + (size_t)na * pair_elems, lane, f0); }
            __builtin_amdgcn_sched_barrier(0);
            amp_mfma_item<CT>(d, f1, n, ab, lane, outs);
        }
    }
    while (made < WT_PROD) produce(made++);
    if (clk != nullptr && blockIdx.x == 0 && lane == 0) clk[wave * 16 + 8] = clock64();
    {
        const int r = aw * WAVE + lane;
        if (do_feed && r < BM && feed.tab != nullptr && row0 + r < M) {
            const uint64_t key = keys[row0 + r];
            if (feed.key_bits == 32) naqs::feed_key<uint32_t>(feed, row0 + r, key);
            else naqs::feed_key<uint64_t>(feed, row0 + r, key);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = q0 * 16 + lane; e < q1 * 16; e += WAVE) {
        const int q = e >> 4, sidx = e & 15;
        const int n = q / RB, r = (q - n * RB) * 16 + sidx;
        float o[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) o[c] = s_o[((size_t)n * BM + r) * 8 + c];
        const uint32_t ab = s_ab[r], mask = (1u << n) - 1u;
        const int occ = (int)((ab >> n) & 1u) + 2 * (int)((ab >> (16 + n)) & 1u);
        s_lan[n][r] = naqs::amp_finish(d, n, o, ab & mask, (ab >> 16) & mask, occ);
    }
    if (clk != nullptr && blockIdx.x == 0 && lane == 0) clk[wave * 16 + 15] = clock64();
}

template <int DBG = 0>
__global__ __launch_bounds__(PH_THREADS) void phase_kernel_wt(const NetDims d, const float *__restrict__ w, const ushort_t *__restrict__ wh,
                                                              const ushort_t *__restrict__ wt, int64_t M, const uint64_t *__restrict__ keys,
                                                              float2 *__restrict__ out, const ElocFeed feed, const ushort_t *__restrict__ wamp,
                                                              const naqs::PhaseScales *__restrict__ scales, const WsSplit split,
                                                              const naqs::PhaseSave save) {
    constexpr int NT = WT_NT, BM = NT * 16;
    constexpr int CHUNK_U4 = 2 * NT * 64;                                  // 16-byte fragments of one chunk of H0^T: [plane][sample tile][lane]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wt[];
    uint4 *ring = reinterpret_cast<uint4 *>(smem_wt);                       // [WT_RING][2][NT][64]
    float *s_o = reinterpret_cast<float *>(smem_wt + (size_t)WT_RING * CHUNK_U4 * sizeof(uint4));   // [P][BM][8] raw outputs of the items
    __shared__ uint32_t s_ab[BM];
    __shared__ float s_lan[MAXP][BM];
    __shared__ __attribute__((aligned(16))) float s_part[WS_MW][4][BM][4];  // the output layer's partial rows: [matrix wave][lane group]
    __shared__ float s_recv[BM][4];
    __shared__ __attribute__((aligned(16))) float s_b0[16][4][8];          // layer 0's scaled bias in the big layer's slot order
    __shared__ uint32_t s_ready[16], s_done[16];                           // chunk c of H0^T is in the ring / matrix waves that have read it
    constexpr int AW = PH_WAVES - WS_MW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = (int)(gridDim.x >> 1);
    const bool producer = (int)blockIdx.x < n_tiles;                       // upper 256 output units; the consumer has the lower ones + the epilogue
    const int tile = producer ? (int)blockIdx.x : (int)blockIdx.x - n_tiles;
    const int64_t row0 = (int64_t)tile * BM;
    const int P = d.P;
    NAQS_MARK(0);
    uint64_t key = 0ull;
    if (tid < BM && row0 + tid < M) key = keys[row0 + tid];
    const int aw = wave >= WS_MW ? wave - WS_MW : 0;
    const int items = NT * P;
    const int q_split = (P / 2) * NT;
    const int qlo = producer ? q_split : 0, qhi = producer ? items : q_split;
    const int first_pair = (qlo + (qhi - qlo) * aw / AW) / NT;
    AmpFrag<4> f4;
    AmpFrag<2> f2;
    const bool ha64 = d.Ha == 64;
    if (wave >= WS_MW) {
        if (ha64) amp_mfma_load<4>(wamp + (size_t)first_pair * amp_mfma_pair_elems(64), lane, f4);
        else amp_mfma_load<2>(wamp + (size_t)first_pair * amp_mfma_pair_elems(32), lane, f2);
    }
    const float sn0 = scales->sn[0], c0 = scales->c[0];
    for (int e = tid; e < 512; e += PH_THREADS) s_b0[e >> 5][(e >> 3) & 3][e & 7] = (w + d.b_off[0])[wt_in_unit(e >> 5, e & 31)] * sn0;
    if (tid < 16) { s_ready[tid] = 0u; s_done[tid] = 0u; }
    if (tid < BM) {
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            if (k < P) {
                a |= (uint32_t)((key >> d.qa[k]) & 1ull) << k;
                b |= (uint32_t)((key >> d.qb[k]) & 1ull) << k;
            }
        }
        s_ab[tid] = a | (b << 16);
    }
    __syncthreads();
    NAQS_MARK(1);

    LayerScale sc;
    sc.c = scales->c[1]; sc.sn = scales->sn[1]; sc.isn = scales->isn[1];
    const int n = lane & 15, g = lane >> 4;
    if (wave < WS_MW) {
        // ---- matrix waves: H1^T tiles mt0 .. mt0 + 3 (64 output units) x 80 samples; A = W1 fragments from L2, B = H0^T chunks from the ring
        const int mt0 = (producer ? 16 : 0) + wave * 4;
        const size_t WTPL = (size_t)512 * 512;
        const ushort_t *wtp = wt + ((size_t)mt0 * 16 * 64 + lane) * 8;     // tile mt0 + i, chunk c: + (i * 16 + c) * 512
        bf16x8 wa[2][4], hb0[2][NT], hb1[2][NT];
        f32x4 acc[4][NT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[i][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto load_wa2 = [&](int c, int i0) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int i = i0; i < i0 + 2; ++i) wa[p][i] = *reinterpret_cast<const bf16x8 *>(wtp + p * WTPL + (size_t)(i * 16 + c) * 512);
        };
        auto take = [&](int c, bf16x8 (&hb)[2][NT]) {                      // wait for chunk c, read this wave's copy of its fragments
            // (LDS operations of a wave execute in order and every producer's fragments precede its flag in its own LDS
            // stream: no fence — a workgroup-scope fence would also wait for the weight fragments in flight from L2 — only the
            // compiler is kept from moving anything across)
            while (__hip_atomic_load(&s_ready[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            const uint4 *src = ring + (size_t)(c % WT_RING) * CHUNK_U4 + lane;
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) hb[p][nt] = __builtin_bit_cast(bf16x8, src[(p * NT + nt) * 64]);
        };
        auto release = [&](int c) {                                        // (after the reads have returned: the slot may be overwritten)
            asm volatile("" ::: "memory");                                 // (the add is behind the reads in this wave's LDS stream)
            if (lane == 0) __hip_atomic_fetch_add(&s_done[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // term-major: consecutive MFMAs go to different accumulators (ten of them), so that none waits for the one before —
        // written accumulator-major the three dependent products of an accumulator sat back to back (29 cycles per MFMA)
        auto mma2 = [&](int i0, const bf16x8 (&hb)[2][NT]) {
            if constexpr (DBG == 1) return;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = i0; i < i0 + 2; ++i) acc[i][nt] = mfma_h<2>(wa[0][i], hb[1][nt], acc[i][nt]);      // w hi x h lo: smallest terms first
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = i0; i < i0 + 2; ++i) acc[i][nt] = mfma_h<2>(wa[1][i], hb[0][nt], acc[i][nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = i0; i < i0 + 2; ++i) acc[i][nt] = mfma_h<2>(wa[0][i], hb[0][nt], acc[i][nt]);
        };
        load_wa2(0, 0);
        load_wa2(0, 2);
        take(0, hb0);
        release(0);
        NAQS_MARK(5);
#pragma unroll 1
        for (int c = 0; c < 16; c += 2) {
            take(c + 1, hb1);
            __builtin_amdgcn_sched_barrier(0);
            mma2(0, hb0);
            __builtin_amdgcn_sched_barrier(0);
            load_wa2(c + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma2(2, hb0);
            __builtin_amdgcn_sched_barrier(0);
            load_wa2(c + 1, 2);
            release(c + 1);
            const int cn = min(c + 2, 15);
            if (c + 2 < 16) take(c + 2, hb0);
            __builtin_amdgcn_sched_barrier(0);
            mma2(0, hb1);
            __builtin_amdgcn_sched_barrier(0);
            load_wa2(cn, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma2(2, hb1);
            __builtin_amdgcn_sched_barrier(0);
            load_wa2(cn, 2);
            if (c + 2 < 16) release(c + 2);
        }
        NAQS_MARK(9);
        // the output layer (512 -> 4) from the accumulators: this lane holds, for sample n of every tile, units 16 (mt0 + i) + 4 g + r
        const float *W2 = w + d.w_off[2], *bias1 = w + d.b_off[1];
        const int K2 = d.K_pad[2];
        float po[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int o = 0; o < 4; ++o) po[nt][o] = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u0 = (mt0 + i) * 16 + 4 * g;
            const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bias1 + u0) * sc.sn;
            f32x4 w2[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) w2[o] = *reinterpret_cast<const f32x4 *>(W2 + (size_t)o * K2 + u0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = fmaxf(fmaf(acc[i][nt][r], sc.c, b1[r]), 0.0f);
#pragma unroll
                    for (int o = 0; o < 4; ++o) po[nt][o] = fmaf(h, w2[o][r], po[nt][o]);
                }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            *reinterpret_cast<f32x4 *>(&s_part[wave][g][nt * 16 + n][0]) = (f32x4){po[nt][0], po[nt][1], po[nt][2], po[nt][3]};
    } else {
        // ---- amplitude waves: chunks aw, aw + 4, ... of H0^T (layer 0 for all 80 samples of 32 hidden units, straight from its
        // accumulators into B-operand fragments) between their (tile, pair) items
        bf16x8 xb[NT];
        {
            const uint32_t pmask = (1u << (P - 1)) - 1u;
            const int nv = min(max(2 * (P - 1) - 8 * g, 0), 8) >> 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const uint32_t ab = s_ab[nt * 16 + n];
                const uint32_t tb = ((ab & pmask) | (((ab >> 16) & pmask) << (P - 1))) >> (8 * g);
                const bool live = row0 + nt * 16 + n < M;
                uint32_t aw4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t wv = 0xBC00BC00u ^ ((tb << (15 - 2 * j)) & 0x8000u) ^ ((tb << (30 - 2 * j)) & 0x80000000u);
                    aw4[j] = (live && j < nv) ? wv : 0u;
                }
                __builtin_memcpy(&xb[nt], aw4, sizeof(bf16x8));
            }
        }
        const size_t W0PL = (size_t)d.N_pad[0] * d.Kh_pad[0];
        const ushort_t *w0p = wh + d.wh_off[0] + lane * 8;                 // layer 0's tile cb, its one K chunk: + cb * 512
        bf16x8 w0[2][2];
        auto load_w0 = [&](int c) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) w0[p][t] = *reinterpret_cast<const bf16x8 *>(w0p + p * W0PL + (size_t)(2 * c + t) * 512);
        };
        load_w0(aw);
        auto produce = [&](int k) {
            const int c = aw + AW * k;
            if (c >= WT_RING)                                              // the slot's previous chunk has been read by all four matrix waves
                while (__hip_atomic_load(&s_done[c - WT_RING], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)WS_MW) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            uint4 *dst = ring + (size_t)(c % WT_RING) * CHUNK_U4 + lane;
            const f32x4 ba = *reinterpret_cast<const f32x4 *>(&s_b0[c][g][0]), bb = *reinterpret_cast<const f32x4 *>(&s_b0[c][g][4]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 d0 = (f32x4){0.f, 0.f, 0.f, 0.f}, d1 = d0;
                d0 = mfma_h<2>(w0[1][0], xb[nt], d0);                      // smallest term first (the inputs are exact: two terms)
                d1 = mfma_h<2>(w0[1][1], xb[nt], d1);
                d0 = mfma_h<2>(w0[0][0], xb[nt], d0);
                d1 = mfma_h<2>(w0[0][1], xb[nt], d1);
                float h[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h[r] = fmaxf(fmaf(d0[r], c0, ba[r]), 0.0f);
                    h[4 + r] = fmaxf(fmaf(d1[r], c0, bb[r]), 0.0f);
                }
                uint32_t hh[4], hl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) split2_pair(h[2 * q], h[2 * q + 1], hh[q], hl[q]);
                dst[nt * 64] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
                dst[(NT + nt) * 64] = make_uint4(hl[0], hl[1], hl[2], hl[3]);
            }
            if (k + 1 < WT_PROD) load_w0(c + AW);
            asm volatile("" ::: "memory");                                 // (the flag is behind the fragments in this wave's LDS stream)
            if (lane == 0) __hip_atomic_store(&s_ready[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        if (ha64) wt_amp_work<4, NT>(d, wamp, M, row0, keys, feed, s_ab, s_lan, s_o, f4, aw, lane, save.clk, wave, qlo, qhi, !producer, produce);
        else wt_amp_work<2, NT>(d, wamp, M, row0, keys, feed, s_ab, s_lan, s_o, f2, aw, lane, save.clk, wave, qlo, qhi, !producer, produce);
    }
    NAQS_MARK(10);
    __syncthreads();
    NAQS_MARK(6);
    auto part_sum = [&](int r, int o) {                    // fixed order: matrix waves, then their lane groups
        float v = 0.0f;
#pragma unroll
        for (int wv = 0; wv < WS_MW; ++wv) v += (s_part[wv][0][r][o] + s_part[wv][1][r][o]) + (s_part[wv][2][r][o] + s_part[wv][3][r][o]);
        return v;
    };
    {
        constexpr int TILE_WORDS = BM * 4 + MAXP * BM;
        unsigned long long *xw = split.xchg + (size_t)tile * TILE_WORDS;
        const int n_split = q_split / NT, n_amp = (P - n_split) * BM;
        const unsigned long long tagw = (unsigned long long)split.tag << 32;
        if (producer) {
            for (int x = tid; x < BM * 4 + n_amp; x += PH_THREADS) {
                float v;
                if (x < BM * 4) v = part_sum(x >> 2, x & 3);
                else { const int e = x - BM * 4; v = s_lan[n_split + e / BM][e % BM]; }
                if (!(naqs::poll_drop(split.ctl, naqs::POLL_LOGPSI_SPLIT) && tile == 0 && x == 0))
                    __hip_atomic_store(&xw[x], tagw | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        bool ok = true;
        for (int x = tid; x < BM * 4 + n_amp; x += PH_THREADS) {
            unsigned long long word;
            ok = naqs::poll_tagged<2>(&xw[x], split.tag, word, split.ctl, naqs::POLL_LOGPSI_SPLIT, (uint32_t)x) && ok;
            const float v = __uint_as_float((uint32_t)word);
            if (x < BM * 4) s_recv[x >> 2][x & 3] = v;
            else { const int e = x - BM * 4; s_lan[n_split + e / BM][e % BM] = v; }
        }
        if (__syncthreads_or(!ok)) return;
    }
    if (tid < BM) {
        const int64_t i = row0 + tid;
        if (i < M) {
            float la = 0.0f;
            for (int nn = 0; nn < P; ++nn) la += s_lan[nn][tid];
            const uint32_t ab = s_ab[tid];
            const int occ = (int)((ab >> (P - 1)) & 1u) + 2 * (int)((ab >> (16 + P - 1)) & 1u);
            const float psum = part_sum(tid, occ) + s_recv[tid][occ];
            const float ph = fmaf(psum, sc.isn, (w + d.b_off[2])[occ]);
            out[i] = make_float2(la, ph);
            if (feed.psi != nullptr) naqs::feed_psi(feed, i, la, ph);
        }
    }
    NAQS_MARK(7);
}

// the weight maxima, their reduction and the scales derived from them: naqs_pack.hpp
using naqs::PhasePackJobs;
using naqs::phase_weight_scale;
__global__ __launch_bounds__(256) void net_bounds_kernel(const float *__restrict__ flat, const PhasePackJobs jobs,
                                                         naqs::PhaseRaw *__restrict__ raw, const uint32_t tag,
                                                         const naqs::PollCtl *ctl) {
    naqs::net_bounds_body(flat, jobs, raw, blockIdx.y, tag, blockIdx.x, ctl);
}

// f32 [N][K] -> three bf16 planes, zero-padded and tiled [plane][N_pad/16][Kh_pad/32][64 lanes][8]
__device__ __forceinline__ void pack_phase_bf16(const float *__restrict__ src, int K, int N, int Kh_pad,
                                                int N_pad, ushort_t *__restrict__ Wd) {
    const int total = N_pad * Kh_pad;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        // e = ((cb * KC + kc) * 64 + kg * 16 + nn) * 8 + j   <-   W[cb*16 + nn][kc*32 + kg*8 + j]
        const int j = e & 7, nn = (e >> 3) & 15, kg = (e >> 7) & 3, blk = e >> 9;
        const int KC = Kh_pad >> 5, cb = blk / KC, kc = blk - cb * KC;
        const int n = tile_col(cb, nn, N_pad), k = kc * 32 + kg * 8 + j;
        const float x = (n < N && k < K) ? src[n * K + k] : 0.0f;
        ushort_t h1, h2, h3;
        split3(x, h1, h2, h3);
        Wd[e] = h1; Wd[(size_t)total + e] = h2; Wd[2 * (size_t)total + e] = h3;
    }
}
// (AmpSrcOff, pack_amp_body and pack_amp_mfma_body — the amplitude blocks' re-pack — live in naqs_pack.hpp: the sampler's first
// launch hosts them too)
using naqs::AmpSrcOff;
using naqs::pack_amp_body;
using naqs::pack_amp_mfma_body;

__global__ __launch_bounds__(256) void pack_amp_mfma_kernel(const float *__restrict__ flat, const NetDims d, const AmpSrcOff so,
                                                            ushort_t *__restrict__ wamp) {
    pack_amp_mfma_body(flat, d, so, wamp, blockIdx.y, (int)blockIdx.x, (int)gridDim.x);
}

__global__ __launch_bounds__(256) void pack_amp_kernel(const float *__restrict__ flat, const NetDims d, const AmpSrcOff so,
                                                       float *__restrict__ w) {
    pack_amp_body(flat, d, so, w, blockIdx.y, (int)blockIdx.x, (int)gridDim.x);
}
// the amplitude blocks alone (naqs_net_set_amp_weights): the VALU rows (blockIdx.z = 0) and the matrix-core fragments (1) in
// one launch, so that the sampler takes the same form of the block MLPs whichever call packed the weights last
__global__ __launch_bounds__(256) void pack_amp_both_kernel(const float *__restrict__ flat, const NetDims d, const AmpSrcOff so,
                                                            float *__restrict__ w, ushort_t *__restrict__ wamp) {
    if (blockIdx.z == 0) pack_amp_body(flat, d, so, w, blockIdx.y, (int)blockIdx.x, (int)gridDim.x);
    else pack_amp_mfma_body(flat, d, so, wamp, blockIdx.y, (int)blockIdx.x, (int)gridDim.x);
}
// aggregate_phase: both sets of per-pair blocks (blockIdx.z = 0, 1) and, when they exist, the amplitude blocks' matrix-core
// fragments (2) in one launch
__global__ __launch_bounds__(256) void pack_amp2_kernel(const float *__restrict__ flat, const NetDims d0, const AmpSrcOff so0,
                                                        float *__restrict__ w0, const NetDims d1, const AmpSrcOff so1,
                                                        float *__restrict__ w1, ushort_t *__restrict__ wamp) {
    if (blockIdx.z == 0) pack_amp_body(flat, d0, so0, w0, blockIdx.y, (int)blockIdx.x, (int)gridDim.x);
    else if (blockIdx.z == 1) pack_amp_body(flat, d1, so1, w1, blockIdx.y, (int)blockIdx.x, (int)gridDim.x);
    else pack_amp_mfma_body(flat, d0, so0, wamp, blockIdx.y, (int)blockIdx.x, (int)gridDim.x);
}

__device__ __forceinline__ void pack_phase_f32(const float *__restrict__ src, int K, int N, int K_pad, int N_pad,
                                               float *__restrict__ Wd, float *__restrict__ bd) {
    const int total = N_pad * K_pad;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total + N_pad; e += gridDim.x * 256) {
        if (e < total) {
            // e = ((cb * KC + kc) * 64 + kq * 16 + nn) * 4 + j   <-   W[cb*16 + nn][kc*16 + kq*4 + j]
            const int j = e & 3, nn = (e >> 2) & 15, kq = (e >> 6) & 3, blk = e >> 8;
            const int KC = K_pad >> 4, cb = blk / KC, kc = blk - cb * KC;
            const int n = cb * 16 + nn, k = kc * 16 + kq * 4 + j;
            Wd[e] = (n < N && k < K) ? src[n * K + k] : 0.0f;
        } else {
            const int n = e - total;
            bd[n] = n < N ? src[N * K + n] : 0.0f;
        }
    }
}

// one phase layer: f32 MFMA tiles + bias, and the split planes (fmt 1: three bf16, fmt 2: two scaled f16)
__device__ __forceinline__ void pack_phase_body(const float *__restrict__ flat, const NetDims &d, const PhasePackJobs &jobs,
                                                float *__restrict__ w, ushort_t *__restrict__ wh, const int with_f32, const int l,
                                                const int fmt, const naqs::PhaseRaw *__restrict__ raw,
                                                naqs::PhaseScales *__restrict__ scales, const uint32_t tag,
                                                const naqs::PollCtl *ctl) {
    if (fmt == 2 && !with_f32) {                        // (the training step's format: shared with the sampler's first launch)
        naqs::pack_phase_job_f16x2(flat, d, jobs, w, wh, l, raw, scales, tag, blockIdx.x, gridDim.x, ctl);
        return;
    }
    float sw = 1.0f;
    if (fmt == 2) {                                     // (whole waves poll; a wait that ran out: nothing of this job is written)
        bool ok = true;
        sw = phase_weight_scale(*raw, l, tag, ctl, ok);
        if (__syncthreads_or(!ok)) return;
    }
    const float *src = flat + jobs.src_off[l];
    if (with_f32) pack_phase_f32(src, jobs.K[l], jobs.N[l], d.K_pad[l], d.N_pad[l], w + d.w_off[l], w + d.b_off[l]);
    else {                                              // the split kernels only need the (padded) bias from this buffer ...
        for (int n = blockIdx.x * 256 + threadIdx.x; n < d.N_pad[l]; n += gridDim.x * 256)
            w[d.b_off[l] + n] = n < jobs.N[l] ? src[jobs.N[l] * jobs.K[l] + n] : 0.0f;
        // ... and phase_kernel_ws the output layer as plain row-major f32 [N_pad][K_pad] (it multiplies by it on the VALU)
        if (l == d.n_lin - 1) {
            const int Kp = d.K_pad[l], total = d.N_pad[l] * Kp;
            for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
                const int n = e / Kp, k = e - n * Kp;
                w[d.w_off[l] + e] = (n < jobs.N[l] && k < jobs.K[l]) ? src[n * jobs.K[l] + k] : 0.0f;
            }
        }
    }
    if (fmt == 2) naqs::pack_phase_f16(src, jobs.K[l], jobs.N[l], d.Kh_pad[l], d.N_pad[l], wh + d.wh_off[l], sw, blockIdx.x, gridDim.x);
    else pack_phase_bf16(src, jobs.K[l], jobs.N[l], d.Kh_pad[l], d.N_pad[l], wh + d.wh_off[l]);
}
// naqs_net_set_weights of the single-phase network in ONE launch (it runs once per training step, and every launch of a
// few thousand elements costs its 4-5 us): blockIdx.y walks the amplitude rows (P jobs), the amplitude fragments (P), the
// phase layers (n_lin) and the row-major copies the backward GEMMs read (naqs::WbPackJobs, from naqs_phase_grad.hip).
// fmt 2: raw holds the partial weight maxima of net_bounds_kernel (launched just before; tagged words); the first phase job
// also writes the scales the phase kernel reads.  y_base: the launch covers the jobs from there on (all of them, or — the
// phase share a training step left pending and nobody hosted — the phase jobs alone).
__global__ __launch_bounds__(256) void pack_net_kernel(const float *__restrict__ flat, const NetDims d, const AmpSrcOff so,
                                                       const PhasePackJobs jobs, const naqs::WbPackJobs wb, float *__restrict__ w,
                                                       ushort_t *__restrict__ wh, ushort_t *__restrict__ wamp, const int with_f32,
                                                       const int fmt, const naqs::PhaseRaw *__restrict__ raw,
                                                       naqs::PhaseScales *__restrict__ scales, const int y_base, const uint32_t tag,
                                                       const naqs::PollCtl *ctl) {
    int y = blockIdx.y + y_base;
    if (y < d.P) { pack_amp_body(flat, d, so, w, y, (int)blockIdx.x, (int)gridDim.x); return; }
    y -= d.P;
    if (wamp != nullptr) {
        if (y < d.P) { pack_amp_mfma_body(flat, d, so, wamp, y, (int)blockIdx.x, (int)gridDim.x); return; }
        y -= d.P;
    }
    if (y < d.n_lin) { pack_phase_body(flat, d, jobs, w, wh, with_f32, y, fmt, raw, scales, tag, ctl); return; }
    y -= d.n_lin;
    if (y < wb.n) naqs::pack_wb_job(flat, wb, y, blockIdx.x, gridDim.x);
}

}  // namespace

NAQS_API int naqs_net_create(const naqs_net_config_t *cfg, int device, naqs_net_t **out) {
    if (!cfg || !out) return NAQS_ERR_INVALID;
    *out = nullptr;
    const int N = cfg->n_qubits;
    if (N <= 0 || (N & 1)) return NAQS_ERR_INVALID;
    const int P = N / 2;
    if (P > MAXP) return NAQS_ERR_UNSUPPORTED;
    if (P < 2) return NAQS_ERR_UNSUPPORTED;
    if (cfg->amp_hidden <= 0 || cfg->n_phase_hidden < 1 || cfg->n_phase_hidden > NAQS_NET_MAX_PHASE_LAYERS)
        return NAQS_ERR_INVALID;
    const bool aggregate = cfg->aggregate_phase != 0;
    if (aggregate && (cfg->n_phase_hidden != 1 || cfg->phase_hidden[0] <= 0)) return NAQS_ERR_UNSUPPORTED;
    if (cfg->masking < 0 || cfg->masking > 2) return NAQS_ERR_INVALID;
    if ((cfg->n_alpha < 0) != (cfg->n_beta < 0)) return NAQS_ERR_INVALID;
    std::vector<bool> seen((size_t)N, false);
    for (int i = 0; i < N; ++i) {
        const int q = cfg->qubit2model[i];
        if (q < 0 || q >= N || seen[(size_t)q]) return NAQS_ERR_INVALID;
        seen[(size_t)q] = true;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NAQS_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return NAQS_ERR_INVALID;

    naqs_net *net = new (std::nothrow) naqs_net();
    if (!net) return NAQS_ERR_NOMEM;
    net->device = device;
    (void)naqs::poll_handle_create(device, &net->poll);
    net->ctl = net->poll.dev;
    net->shared_gpu = naqs::env_int("NAQS_SHARED_GPU", 0) == 1;    // (naqs_net_share_device: the same switch per handle)
    net->cfg = *cfg;
    NetDims &d = net->dims;
    d.P = P;
    d.n_alpha = cfg->n_alpha; d.n_beta = cfg->n_beta;
    d.n_alpha_down = (N + 1) / 2 - cfg->n_alpha; d.n_beta_down = N / 2 - cfg->n_beta;
    d.min_n_set = cfg->n_alpha < 0 ? 0 : std::min(std::min(d.n_alpha, d.n_beta), std::min(d.n_alpha_down, d.n_beta_down));
    d.masking = cfg->masking;
    d.sym = cfg->use_amp_spin_sym ? 1 : 0;
    d.phase_sym = cfg->use_phase_spin_sym ? 1 : 0;
    const int n_out_phase = d.phase_sym ? 3 : 4;               // nade.py:281
    d.Ha = cfg->amp_hidden;
    d.n_out_amp = d.sym ? 5 : 4;
    for (int n = 0; n < P; ++n) { d.qa[n] = (uint8_t)cfg->qubit2model[2 * n]; d.qb[n] = (uint8_t)cfg->qubit2model[2 * n + 1]; }
    int64_t off = 0, poff = 0;
    for (int n = 0; n < P; ++n) {
        net->amp_src_off[n] = off;
        d.amp_off[n] = (int32_t)poff;
        const int nin = n == 0 ? 1 : 2 * n;
        off += (int64_t)d.Ha * nin + d.Ha + (int64_t)d.n_out_amp * d.Ha + d.n_out_amp;
        poff += (int64_t)d.Ha * ((nin + 1 + 5 + 3) & ~3) + 8;
    }
    net->amp_params = off;
    net->aggregate = aggregate;
    if (aggregate) {
        // phase blocks as a second amplitude-shaped network: same inputs, Hp hidden units, 4 raw outputs, no symmetry
        NetDims &q = net->dph;
        q = d;
        q.Ha = cfg->phase_hidden[0];
        q.sym = 0;
        q.n_out_amp = n_out_phase;             // (-phase_sym: 3 raw outputs, the middle one for |01> and |10>; q.phase_sym orders the inputs)
        int64_t poff2 = 0;
        for (int n = 0; n < P; ++n) {
            net->ph_src_off[n] = off;
            q.amp_off[n] = (int32_t)poff2;
            const int nin = n == 0 ? 1 : 2 * n;
            off += (int64_t)q.Ha * nin + q.Ha + (int64_t)n_out_phase * q.Ha + n_out_phase;
            poff2 += (int64_t)q.Ha * ((nin + 1 + 5 + 3) & ~3) + 8;
        }
        net->ph_params = off - net->amp_params;
        net->n_params = off;
        net->w_floats = poff;
        d.n_lin = 0;
        d.ld = d.ldh = 0;
        DeviceGuard guard0;
        int st0 = guard0.init(device);
        if (st0 == NAQS_OK) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, device) == hipSuccess) net->cu_count = prop.multiProcessorCount;
            if (hipMalloc((void **)&net->d_w, (size_t)poff * sizeof(float)) != hipSuccess) st0 = NAQS_ERR_NOMEM;
            if (st0 == NAQS_OK && hipMalloc((void **)&net->d_wph, (size_t)poff2 * sizeof(float)) != hipSuccess) st0 = NAQS_ERR_NOMEM;
            // the amplitude blocks as matrix-core fragments too: the sampler's block MLP (the forward pass of this family keeps
            // the one merged VALU launch for both sets of blocks)
            if (st0 == NAQS_OK && (d.Ha == 32 || d.Ha == 64 || d.Ha == 128) && 2 * (P - 1) <= 30) {
                const size_t elems = (size_t)P * amp_mfma_pair_elems(d.Ha);
                if (hipMalloc((void **)&net->d_wamp, elems * sizeof(unsigned short)) != hipSuccess) st0 = NAQS_ERR_NOMEM;
            }
        }
        if (st0 != NAQS_OK) { naqs_net_destroy(net); return st0; }
        *out = net;
        return NAQS_OK;
    }
    // phase block: 2(P-1) -> hidden... -> 4
    int K = std::max(1, 2 * (P - 1));
    d.n_lin = cfg->n_phase_hidden + 1;
    int64_t src = off, dst = poff;
    int max_k = 0;
    for (int l = 0; l < d.n_lin; ++l) {
        const int Nout = l < cfg->n_phase_hidden ? cfg->phase_hidden[l] : n_out_phase;
        if (Nout <= 0) { delete net; return NAQS_ERR_INVALID; }
        d.K_pad[l] = (K + 15) & ~15;
        d.N_pad[l] = (Nout + 15) & ~15;
        if (d.N_pad[l] > PH_WAVES * CBT * 16) { delete net; return NAQS_ERR_UNSUPPORTED; }   // <= 512 outputs per layer
        dst = (dst + 3) & ~3ll;                                   // 16-byte aligned rows for the float4 weight loads
        d.w_off[l] = (int32_t)dst; dst += (int64_t)d.N_pad[l] * d.K_pad[l];
        d.b_off[l] = (int32_t)dst; dst += d.N_pad[l];
        net->phase_src_off.push_back(src);
        net->phase_K.push_back(K); net->phase_N.push_back(Nout);
        src += (int64_t)Nout * K + Nout;
        max_k = std::max(max_k, std::max(d.K_pad[l], d.N_pad[l]));
        K = Nout;
    }
    d.ld = max_k + 4;                       // +4 floats: consecutive rows start one 16-byte LDS slot apart
    {   // bf16x3 layout: K padded to the 32-wide bf16 MFMA chunk
        int64_t hoff = 0;
        int Kh = std::max(1, 2 * (P - 1)), max_kh = 0;
        for (int l = 0; l < d.n_lin; ++l) {
            const int Nout = l < cfg->n_phase_hidden ? cfg->phase_hidden[l] : n_out_phase;
            d.Kh_pad[l] = (Kh + 31) & ~31;
            d.wh_off[l] = (int32_t)hoff;
            hoff += 3ll * d.N_pad[l] * d.Kh_pad[l];
            max_kh = std::max(max_kh, std::max(d.Kh_pad[l], d.N_pad[l]));
            Kh = Nout;
        }
        d.ldh = max_kh + 8;                 // +16 bytes: consecutive rows start one LDS slot apart
        net->wh_elems = hoff;
    }
    net->n_params = src;
    net->w_floats = dst;

    DeviceGuard guard;
    int st = guard.init(device);
    if (st == NAQS_OK) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) net->cu_count = prop.multiProcessorCount;
        if (hipMalloc((void **)&net->d_w, (size_t)net->w_floats * sizeof(float)) != hipSuccess) st = NAQS_ERR_NOMEM;
        if (st == NAQS_OK && hipMalloc((void **)&net->d_wh, (size_t)net->wh_elems * sizeof(unsigned short)) != hipSuccess) st = NAQS_ERR_NOMEM;
        if (st == NAQS_OK && (d.Ha == 32 || d.Ha == 64 || d.Ha == 128)) {   // amplitude blocks as MFMA fragments (phase kernel prologue);
            const size_t elems = (size_t)P * amp_mfma_pair_elems(d.Ha);   // input slot 31 must be free for the bias: 2 (P - 1) <= 30
            if (hipMalloc((void **)&net->d_wamp, elems * sizeof(unsigned short)) != hipSuccess) st = NAQS_ERR_NOMEM;
        }
        // the activation tile of 48/64 rows x 516 floats exceeds the 64 KiB default of dynamic LDS
        const int lds_max = 4 * 16 * d.ld * (int)sizeof(float);
        if (lds_max > 160 * 1024) st = NAQS_ERR_UNSUPPORTED;
        if (st == NAQS_OK) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max / 4);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max / 2);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max / 4 * 3);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
            // (the amplitude prologue's scratch can exceed one 16-row slab: allow the maximum for every variant)
            const int lds_all = 160 * 1024 - 5 * 1024;         // the static part (s_ab, s_lan) is < 5 KiB for every RB
#define NAQS_PH_ATTR(RB, FMT)                                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel_h<RB, false, FMT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_all); \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel_h<RB, true, FMT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_all);
            NAQS_PH_ATTR(1, 1) NAQS_PH_ATTR(2, 1) NAQS_PH_ATTR(3, 1)
            NAQS_PH_ATTR(1, 2) NAQS_PH_ATTR(2, 2) NAQS_PH_ATTR(3, 2) NAQS_PH_ATTR(4, 2)
#undef NAQS_PH_ATTR
#define NAQS_WS_ATTR(RB)                                                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel_ws<RB, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_all); \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel_ws<RB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_all); \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel_ws<RB, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_all); \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel_ws<RB, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_all);
            NAQS_WS_ATTR(1) NAQS_WS_ATTR(2) NAQS_WS_ATTR(3)
#undef NAQS_WS_ATTR
            {   // phase_kernel_wt: the ring of H0^T chunks + the items' raw outputs
                const int lds_wt = (int)((size_t)WT_RING * 2 * WT_NT * 64 * 16 + (size_t)MAXP * 16 * WT_NT * 8 * sizeof(float));
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel_wt<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_wt);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel_wt<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_wt);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&phase_kernel_wt<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_wt);
            }
            (void)hipGetLastError();            // a refused attribute must not stay behind as the runtime's "last error"
        }
        if (st == NAQS_OK && hipMalloc((void **)&net->d_raw, sizeof(naqs::PhaseRaw)) != hipSuccess) st = NAQS_ERR_NOMEM;
        if (st == NAQS_OK && hipMalloc((void **)&net->d_scales, sizeof(naqs::PhaseScales)) != hipSuccess) st = NAQS_ERR_NOMEM;
        if (st == NAQS_OK && hipMemset(net->d_raw, 0, sizeof(naqs::PhaseRaw)) != hipSuccess) st = NAQS_ERR_HIP;
        if (st == NAQS_OK && hipMalloc((void **)&net->d_sum_words, 8 * sizeof(unsigned long long)) != hipSuccess) st = NAQS_ERR_NOMEM;
        if (st == NAQS_OK && hipMemset(net->d_sum_words, 0, 8 * sizeof(unsigned long long)) != hipSuccess) st = NAQS_ERR_HIP;
        if (st == NAQS_OK && hipDeviceSynchronize() != hipSuccess) st = NAQS_ERR_HIP;     // null-stream fill: done before any non-blocking stream writes there
    }
    if (st != NAQS_OK) { naqs_net_destroy(net); return st; }
    *out = net;
    return NAQS_OK;
}

NAQS_API int naqs_net_destroy(naqs_net_t *net) {
    if (!net) return NAQS_OK;
    DeviceGuard guard;
    (void)guard.init(net->device);
    (void)net->prof.enable(0);
    (void)net->prof_samp.enable(0);
    if (net->d_w) (void)hipFree(net->d_w);
    if (net->d_wph) (void)hipFree(net->d_wph);
    if (net->d_wh) (void)hipFree(net->d_wh);
    if (net->d_wamp) (void)hipFree(net->d_wamp);
    if (net->d_wt) (void)hipFree(net->d_wt);
    if (net->d_scratch) (void)hipFree(net->d_scratch);
    if (net->d_samp) (void)hipFree(net->d_samp);
    if (net->d_gpart) (void)hipFree(net->d_gpart);
    if (net->d_train) (void)hipFree(net->d_train);
    if (net->d_wb) (void)hipFree(net->d_wb);
    if (net->h_info) (void)hipHostFree(net->h_info);
    if (net->d_info2) (void)hipFree(net->d_info2);
    if (net->ev_fork) (void)hipEventDestroy(net->ev_fork);
    if (net->ev_join) (void)hipEventDestroy(net->ev_join);
    if (net->ev_phase_done) (void)hipEventDestroy(net->ev_phase_done);
    if (net->side_stream) (void)hipStreamDestroy(net->side_stream);
    if (net->d_raw) (void)hipFree(net->d_raw);
    if (net->d_scales) (void)hipFree(net->d_scales);
    if (net->d_ws_xchg) (void)hipFree(net->d_ws_xchg);
    if (net->d_sum_words) (void)hipFree(net->d_sum_words);
    naqs::poll_handle_destroy(&net->poll);
    delete net;
    return NAQS_OK;
}

NAQS_API int naqs_net_param_count(const naqs_net_t *net, int64_t *count) {
    if (!net || !count) return NAQS_ERR_INVALID;
    *count = net->n_params;
    return NAQS_OK;
}

// which kernel evaluates the phase MLP.  NAQS_PHASE_MODE: 2 (default) f16x2 split on the f16 matrix cores, 1 bf16x3 split on
// the bf16 matrix cores, 0 exact-f32 MFMA.  A split format needs its 16-row slab of activation planes to fit the LDS
// three times over (the tiling the amplitude prologue's scratch was sized for); else the f32 kernel runs.
static int phase_format(const NetDims &d) {
    const int mode = naqs::env_int("NAQS_PHASE_MODE", 2);
    const size_t slab3 = 3 * 16 * (size_t)d.ldh * sizeof(unsigned short);
    if (mode < 1 || 3 * slab3 > 160 * 1024) return 0;
    return mode == 1 ? 1 : 2;
}
static size_t phase_slab_bytes(const NetDims &d, int fmt) { return (size_t)(fmt == 2 ? 2 : 3) * 16 * d.ldh * sizeof(unsigned short); }
static int phase_rb_max(const NetDims &d, int fmt) {
    if (fmt == 0) return 4;
    return (int)std::min<size_t>(fmt == 2 ? 4 : 3, (size_t)(155 * 1024) / phase_slab_bytes(d, fmt));
}

static int pack_blocks(const NetDims &d, const int64_t *src_off, float *dst, const float *flat_dev, hipStream_t s) {
    AmpSrcOff so;
    for (int n = 0; n < MAXP; ++n) so.off[n] = src_off[n];
    const int total_max = d.Ha * ((2 * (d.P - 1) + 1 + 5 + 3) & ~3) + 8;
    NAQS_KLAUNCH(pack_amp_kernel, dim3((total_max + 255) / 256, d.P), dim3(256), 0, s, flat_dev, d, so, dst);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}
static int pack_amp_blocks(naqs_net_t *net, const float *flat_dev, hipStream_t s) {
    net->wamp_fresh = false;
    return pack_blocks(net->dims, net->amp_src_off, net->d_w, flat_dev, s);
}
// rows + fragments of the amplitude blocks in one launch (naqs_net_set_amp_weights)
static int pack_amp_both(naqs_net_t *net, const float *flat_dev, hipStream_t s) {
    if (!net->d_wamp) return pack_amp_blocks(net, flat_dev, s);
    const NetDims &d = net->dims;
    AmpSrcOff so;
    for (int n = 0; n < MAXP; ++n) so.off[n] = net->amp_src_off[n];
    const int frag = ((d.Ha >> 4) + (d.Ha >> 5)) * 512;
    const int total_max = std::max(d.Ha * ((2 * (d.P - 1) + 1 + 5 + 3) & ~3) + 8, frag);
    net->wamp_fresh = false;
    NAQS_KLAUNCH(pack_amp_both_kernel, dim3((total_max + 255) / 256, d.P, 2), dim3(256), 0, s, flat_dev, d, so, net->d_w, net->d_wamp);
    HIP_TRY(hipGetLastError());
    net->wamp_fresh = true;
    return NAQS_OK;
}
// the amplitude blocks as MFMA fragments alone (the aggregate-phase family's unmerged path)
static int pack_amp_fragments(naqs_net_t *net, const float *flat_dev, hipStream_t s) {
    if (!net->d_wamp) return NAQS_OK;
    const NetDims &d = net->dims;
    AmpSrcOff so;
    for (int n = 0; n < MAXP; ++n) so.off[n] = net->amp_src_off[n];
    const int frag = ((d.Ha >> 4) + (d.Ha >> 5)) * 512;          // elements of one plane
    NAQS_KLAUNCH(pack_amp_mfma_kernel, dim3((frag + 255) / 256, d.P), dim3(256), 0, s, flat_dev, d, so, net->d_wamp);
    HIP_TRY(hipGetLastError());
    net->wamp_fresh = true;
    return NAQS_OK;
}

NAQS_API int naqs_net_set_amp_weights(naqs_net_t *net, const float *flat_dev, int64_t count, void *stream) {
    if (!net || !flat_dev || (count != net->n_params && count != net->amp_params)) return NAQS_ERR_INVALID;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    net->have_weights = net->have_wb = false;   // the packed phase layers no longer belong to these parameters
    net->have_amp_weights = false;
    net->amp_head_packed = 0;
    net->pack_pending_amp = false;              // (a training step's pending amplitude share is superseded by this re-pack; a hosted
                                                //  copy of it would write the blocks from the OTHER parameter vector over these)
    // rows AND fragments: the sampler picks the matrix-core form of the block MLPs whenever the fragments are current, and the
    // two forms round differently — the same (parameters, seed) must not draw differently depending on which call packed last
    st = pack_amp_both(net, flat_dev, reinterpret_cast<hipStream_t>(stream));
    if (st != NAQS_OK) return st;
    net->have_amp_weights = true;
    return NAQS_OK;
}

// The single-phase network's re-pack.  PACK_ALL: everything on `s` (weight maxima, then one launch for all jobs).  PACK_AMP
// (naqs_vmc_step, f16x2 format): the amplitude jobs only — what the next sampler call reads — and the phase share stays
// PENDING: the sampler's first launch hosts it (PACK_TAKE fills the arguments: naqs_pack.hpp), or whoever reads the phase
// layers first starts it in order (PACK_PHASE).
// PACK_DEFER (round 6; naqs_vmc_step with NAQS_PACK_OVERLAP=2, the default): NOTHING is launched — the amplitude share is pending too
// (net->pack_pending_amp) and the next sampler call's first launch hosts all of it (PACK_TAKE with `head_pairs` > 0: the fragments
// of pairs 0 .. head_pairs - 1, the ones that launch's own first workgroup reads, were packed by the update's launch —
// grad_finish_kernel's first workgroups, net->amp_head_packed); any other reader of the amplitude blocks starts the amplitude
// jobs in order first (net_flush_pack).
enum PackMode { PACK_ALL = 0, PACK_AMP = 1, PACK_PHASE = 2, PACK_TAKE = 3, PACK_DEFER = 4 };
static int pack_single_phase(naqs_net *net, const float *flat_dev, hipStream_t s, PackMode mode, naqs::PackPhaseArgs *take, const int head_pairs = 0) {
    const NetDims &d = net->dims;
    PhasePackJobs jobs{};
    int biggest = d.Ha * ((2 * (d.P - 1) + 1 + 5 + 3) & ~3) + 8;                 // an amplitude block's packed rows
    const int amp_biggest = std::max(biggest, (d.Ha / 16 + d.Ha / 32) * 512);
    for (int l = 0; l < d.n_lin; ++l) {
        jobs.src_off[l] = net->phase_src_off[(size_t)l];
        jobs.K[l] = net->phase_K[(size_t)l];
        jobs.N[l] = net->phase_N[(size_t)l];
        biggest = std::max(biggest, d.N_pad[l] * std::max(d.K_pad[l], d.Kh_pad[l]));
    }
    // the f32-MFMA weight tiles are only read by phase_kernel (NAQS_PHASE_MODE=0)
    const int fmt = phase_format(d);
    const int with_f32 = fmt == 0 ? 1 : 0;
    naqs::WbPackJobs wb{};
    int st = naqs::net_backward_pack_jobs(net, &wb);
    if (st != NAQS_OK) return st;
    for (int i = 0; i < wb.n; ++i) biggest = std::max(biggest, wb.Np[i] * wb.Kp[i]);
    AmpSrcOff so;
    for (int n = 0; n < MAXP; ++n) so.off[n] = net->amp_src_off[n];
    const int gx = std::min(256, (biggest + 255) / 256);
    const int gy_amp = d.P + (net->d_wamp ? d.P : 0), gy_phase = d.n_lin + wb.n;
    naqs::PhaseRaw *raw = net->d_raw;
    if (mode == PACK_DEFER && (fmt != 2 || net->d_wamp == nullptr || net->amp_head_packed <= 0)) mode = PACK_AMP;      // (nothing to host the amplitude share with)
    if (mode == PACK_ALL || mode == PACK_AMP || mode == PACK_DEFER) {
        net->packed_f32 = with_f32 != 0;
        net->packed_fmt = fmt;
        net->wamp_fresh = false;
    }
    if (mode != PACK_TAKE) net->have_wt = false;          // (a re-pack of any kind: phase_kernel_wt's copy is only refreshed by PACK_ALL, below)
    const bool split = mode != PACK_ALL && fmt == 2;      // (PACK_AMP on another format: everything now, nothing pending)
    if (fmt == 2 && (!split || mode == PACK_PHASE || mode == PACK_TAKE)) {
        if (++net->pack_seq == 0u) {                       // the 32-bit tag is about to repeat: forget every old word
            HIP_TRY(hipMemsetAsync(raw, 0, sizeof(naqs::PhaseRaw), s));
            net->pack_seq = 1u;
        }
    }
    if (!split) {
        if (fmt == 2) {
            // weight maxima -> scales (device side; no host round trip)
            NAQS_KLAUNCH(net_bounds_kernel, dim3(naqs::BOUNDS_WG, d.n_lin), dim3(256), 0, s, flat_dev, jobs, raw, net->pack_seq, net->ctl);
            HIP_TRY(hipGetLastError());
        }
        NAQS_KLAUNCH(pack_net_kernel, dim3(gx, gy_amp + gy_phase), dim3(256), 0, s, flat_dev, d, so, jobs, wb, net->d_w, net->d_wh, net->d_wamp,
                           with_f32, fmt, raw, net->d_scales, 0, net->pack_seq, net->ctl);
        HIP_TRY(hipGetLastError());
        net->pack_pending = nullptr;
        net->pack_pending_amp = false;
        // the big layer once more in phase_kernel_wt's order (inference on tables of a few thousand rows and up; the training
        // step's re-packs leave it stale and its forward passes never read it)
        if (fmt == 2 && d.n_lin == 3 && jobs.N[0] == 512 && jobs.K[1] == 512 && jobs.N[1] == 512 && naqs::env_int("NAQS_PHASE_WT", 0) != 0) {
            if (!net->d_wt) HIP_TRY(hipMalloc((void **)&net->d_wt, (size_t)2 * 512 * 512 * sizeof(unsigned short)));
            NAQS_KLAUNCH(pack_wt_kernel, dim3(256), dim3(256), 0, s, flat_dev + jobs.src_off[1], net->d_scales, net->d_wt);
            HIP_TRY(hipGetLastError());
            net->have_wt = true;
        }
    } else if (mode == PACK_AMP) {
        NAQS_KLAUNCH(pack_net_kernel, dim3(std::min(256, (amp_biggest + 255) / 256), gy_amp), dim3(256), 0, s, flat_dev, d, so, jobs, wb, net->d_w,
                           net->d_wh, net->d_wamp, with_f32, fmt, raw, net->d_scales, 0, 0u, net->ctl);
        HIP_TRY(hipGetLastError());
        net->pack_pending = flat_dev;
        net->pack_pending_amp = false;
        net->pack_stream = s;
    } else if (mode == PACK_DEFER) {
        net->pack_pending = flat_dev;
        net->pack_pending_amp = true;
        net->pack_stream = s;
    } else if (mode == PACK_PHASE) {
        NAQS_KLAUNCH(net_bounds_kernel, dim3(naqs::BOUNDS_WG, d.n_lin), dim3(256), 0, s, flat_dev, jobs, raw, net->pack_seq, net->ctl);
        HIP_TRY(hipGetLastError());
        NAQS_KLAUNCH(pack_net_kernel, dim3(gx, gy_phase), dim3(256), 0, s, flat_dev, d, so, jobs, wb, net->d_w, net->d_wh, net->d_wamp, with_f32,
                           fmt, raw, net->d_scales, gy_amp, net->pack_seq, net->ctl);
        HIP_TRY(hipGetLastError());
        net->pack_pending = nullptr;
    } else {                                               // PACK_TAKE: the caller's launch hosts the jobs
        take->flat = flat_dev; take->jobs = jobs; take->wb = wb; take->w = net->d_w; take->wh = net->d_wh; take->raw = raw;
        take->scales = net->d_scales; take->tag = net->pack_seq; take->gx = gx; take->ctl = net->ctl;
        take->gxl = std::max(1, std::min(gx, naqs::HOSTED_WAITING_WGS / std::max(1, d.n_lin)));      // (naqs_pack.hpp: HOSTED_WAITING_WGS)
        take->n_wgs = d.n_lin * naqs::BOUNDS_WG + wb.n * gx + d.n_lin * take->gxl;
        if (net->pack_pending_amp) {                       // the amplitude share rides along, in front (naqs_pack.hpp)
            take->amp = 1; take->so = so; take->wamp = net->d_wamp; take->head_pairs = std::min(head_pairs, d.P);
            take->gxa = std::max(1, ((d.Ha * ((2 * (d.P - 1) + 1 + 5 + 3) & ~3) + 8) + 255) / 256);
            take->gxf = std::max(1, ((d.Ha / 16 + d.Ha / 32) * 512 + 255) / 256);
            take->n_amp_wgs = d.P * take->gxa + (d.P - take->head_pairs) * take->gxf;
            take->n_wgs += take->n_amp_wgs;
        }
        net->pack_pending = nullptr;
        net->pack_pending_amp = false;
    }
    if (mode == PACK_ALL || mode == PACK_AMP || mode == PACK_DEFER) {
        net->wamp_fresh = net->d_wamp != nullptr;
        net->have_wb = true;
    }
    return NAQS_OK;
}

// a pending re-pack reads the parameters the update wrote on `pack_stream`: a different stream is ordered behind it first
static int pack_follow_update(naqs_net *net, hipStream_t s) {
    if (net->pack_stream != nullptr && net->pack_stream != s) {
        if (!net->ev_fork) HIP_TRY(hipEventCreateWithFlags(&net->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(net->ev_fork, net->pack_stream));
        HIP_TRY(hipStreamWaitEvent(s, net->ev_fork, 0));
    }
    return NAQS_OK;
}
int naqs::net_take_pending_pack(naqs_net *net, hipStream_t s, naqs::PackPhaseArgs *out, const int head_pairs) {
    *out = naqs::PackPhaseArgs{};
    if (net->pack_pending == nullptr) return NAQS_OK;
    int st = pack_follow_update(net, s);
    if (st != NAQS_OK) return st;
    // the amplitude share rides along only if what the host's own workgroup reads — the fragments of its `head_pairs` leading
    // pairs — was packed by the update itself; else the amplitude jobs as a launch first
    if (net->pack_pending_amp && (head_pairs <= 0 || net->amp_head_packed < head_pairs)) {
        st = pack_single_phase(net, net->pack_pending, s, PACK_AMP, nullptr);
        if (st != NAQS_OK || net->pack_pending == nullptr) return st;
    }
    return pack_single_phase(net, net->pack_pending, s, PACK_TAKE, out, net->amp_head_packed);
}
// the amplitude blocks' share alone, if a training step left it pending (every reader of d_w's amplitude rows / d_wamp that is
// not the hosting sampler launch); the phase share stays pending
int naqs::net_flush_amp_pack(naqs_net *net, hipStream_t s) {
    if (net->pack_pending == nullptr || !net->pack_pending_amp) return NAQS_OK;
    const int st = pack_follow_update(net, s);
    if (st != NAQS_OK) return st;
    return pack_single_phase(net, net->pack_pending, s, PACK_AMP, nullptr);
}
int naqs::net_flush_pack(naqs_net *net, hipStream_t s) {
    if (net->phase_pending) {                              // naqs_vmc_run's deferred phase chain: `s` goes behind it
        HIP_TRY(hipStreamWaitEvent(s, net->ev_phase_done, 0));
        net->phase_pending = false;
    }
    if (net->pack_pending == nullptr) return NAQS_OK;
    int st = pack_follow_update(net, s);
    if (st != NAQS_OK) return st;
    if (net->pack_pending_amp) {
        st = pack_single_phase(net, net->pack_pending, s, PACK_AMP, nullptr);
        if (st != NAQS_OK || net->pack_pending == nullptr) return st;      // (another number format by now: that launch packed everything)
    }
    return pack_single_phase(net, net->pack_pending, s, PACK_PHASE, nullptr);
}
int naqs::net_finish_pending(naqs_net *net, hipStream_t s) { return naqs::net_flush_pack(net, s); }

NAQS_API int naqs_net_check(naqs_net_t *net) {
    if (!net) return NAQS_ERR_INVALID;
    return naqs::poll_check(net->poll);
}

NAQS_API int naqs_net_finish_pending(naqs_net_t *net, void *stream) {
    if (!net) return NAQS_ERR_INVALID;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    return naqs::net_flush_pack(net, reinterpret_cast<hipStream_t>(stream));
}

NAQS_API int naqs_net_set_weights(naqs_net_t *net, const float *flat_dev, int64_t count, void *stream) {
    if (!net || !flat_dev || count != net->n_params) return NAQS_ERR_INVALID;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const NetDims &d = net->dims;
    if (net->phase_pending) {                             // (a deferred phase chain still writes what this re-pack reads and overwrites)
        HIP_TRY(hipStreamWaitEvent(s, net->ev_phase_done, 0));
        net->phase_pending = false;
    }
    net->have_weights = net->have_amp_weights = net->have_wb = false;
    net->pack_pending = nullptr;                          // (whatever was pending is superseded by this re-pack)
    net->pack_pending_amp = false;
    if (net->overlap_next_pack != 2) net->amp_head_packed = 0;      // (only naqs_vmc_step's update packs leading pairs itself)
    if (net->aggregate) {                                   // the phase blocks in the amplitude rows' layout; nothing else to pack
        if (net->dims.P == net->dph.P && (naqs::env_int("NAQS_AGG_MERGE", 7) & 4)) {
            AmpSrcOff so0, so1;
            for (int n = 0; n < MAXP; ++n) { so0.off[n] = net->amp_src_off[n]; so1.off[n] = net->ph_src_off[n]; }
            const int frag = ((d.Ha >> 4) + (d.Ha >> 5)) * 512;
            const int total_max = std::max(std::max(d.Ha, net->dph.Ha) * ((2 * (d.P - 1) + 1 + 5 + 3) & ~3) + 8, net->d_wamp ? frag : 0);
            net->wamp_fresh = false;
            NAQS_KLAUNCH(pack_amp2_kernel, dim3((total_max + 255) / 256, d.P, net->d_wamp ? 3 : 2), dim3(256), 0, s, flat_dev, d, so0, net->d_w,
                               net->dph, so1, net->d_wph, net->d_wamp);
            HIP_TRY(hipGetLastError());
            net->wamp_fresh = net->d_wamp != nullptr;
            net->have_weights = net->have_amp_weights = net->have_wb = true;
            return NAQS_OK;
        } else {
            st = pack_amp_blocks(net, flat_dev, s);
            if (st != NAQS_OK) return st;
            st = pack_blocks(net->dph, net->ph_src_off, net->d_wph, flat_dev, s);
            if (st != NAQS_OK) return st;
        }
        st = pack_amp_fragments(net, flat_dev, s);
        if (st != NAQS_OK) return st;
        net->have_weights = net->have_amp_weights = net->have_wb = true;
        return NAQS_OK;
    }
    st = pack_single_phase(net, flat_dev, s, net->overlap_next_pack == 2 ? PACK_DEFER : (net->overlap_next_pack ? PACK_AMP : PACK_ALL), nullptr);
    if (st != NAQS_OK) return st;
    net->have_weights = net->have_amp_weights = true;
    return NAQS_OK;
}

static int launch_amp_kernel(const NetDims &d, const float *w, int64_t M, const uint64_t *keys_dev, float *scratch,
                             const ElocFeed &feed, int raw, hipStream_t s) {
    const size_t amp_lds = ((size_t)d.Ha * ((2 * (d.P - 1) + 1 + 5 + 3) & ~3) + 8) * sizeof(float);
    if (amp_lds > 64 * 1024) return NAQS_ERR_UNSUPPORTED;
    NAQS_KLAUNCH(amp_kernel, dim3((unsigned)((M + AMP_TILES * WAVE - 1) / (AMP_TILES * WAVE)), (unsigned)d.P), dim3(AMP_TILES * AMP_SPLIT * WAVE), amp_lds, s, d, w, M,
                       keys_dev, scratch, feed, raw);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

int naqs::net_amp_forward(naqs_net *net, int64_t M, const uint64_t *keys_dev, hipStream_t s, const ElocFeed *feed, bool launch) {
    const NetDims &d = net->dims;
    if (M >= (1ll << 31)) return NAQS_ERR_UNSUPPORTED;
    if (M > net->cap_M) {
        HIP_TRY(hipDeviceSynchronize());
        if (net->d_scratch) (void)hipFree(net->d_scratch);
        net->d_scratch = nullptr; net->cap_M = 0;
        const int64_t cap = std::max<int64_t>(1024, M + M / 4);
        // [P][cap] conditional log-amplitudes (+ [P][cap] phases of the per-pair phase blocks)
        HIP_TRY(hipMalloc((void **)&net->d_scratch, (size_t)cap * d.P * (net->aggregate ? 2 : 1) * sizeof(float)));
        net->cap_M = cap;
    }
    if (!launch) return NAQS_OK;
    const ElocFeed none{};
    if (net->d_wamp != nullptr && net->wamp_fresh && naqs::env_int("NAQS_AMP_MODE", 1) != 0) {      // matrix-core form (0: the VALU amp_kernel)
        const int64_t waves = (M + AMPK_TG * 16 - 1) / (AMPK_TG * 16) * d.P;
        const unsigned grid = (unsigned)((waves + AMPK_WAVES - 1) / AMPK_WAVES);
        const size_t lds = 0;
        if (d.Ha == 128) NAQS_KLAUNCH(amp_mfma_kernel<8>, dim3(grid), dim3(AMPK_WAVES * 64), lds, s, d, net->d_wamp, M, keys_dev, net->d_scratch, feed ? *feed : none);
        else if (d.Ha == 64) NAQS_KLAUNCH(amp_mfma_kernel<4>, dim3(grid), dim3(AMPK_WAVES * 64), lds, s, d, net->d_wamp, M, keys_dev, net->d_scratch, feed ? *feed : none);
        else NAQS_KLAUNCH(amp_mfma_kernel<2>, dim3(grid), dim3(AMPK_WAVES * 64), lds, s, d, net->d_wamp, M, keys_dev, net->d_scratch, feed ? *feed : none);
        HIP_TRY(hipGetLastError());
        return NAQS_OK;
    }
    return launch_amp_kernel(d, net->d_w, M, keys_dev, net->d_scratch, feed ? *feed : none, 0, s);
}

// aggregate_phase: amplitude blocks, phase blocks (raw), then the sums
static int agg_logpsi(naqs_net *net, int64_t M, const uint64_t *keys_dev, float *logpsi_dev, hipStream_t s, const ElocFeed &feed) {
    // (NAQS_AMP_MODE=2: the amplitude blocks as matrix-core items + a second launch for the phase blocks; default: one merged VALU launch)
    const bool mfma_amp = net->d_wamp != nullptr && net->wamp_fresh && naqs::env_int("NAQS_AMP_MODE", 1) == 2;
    const bool prof = net->prof.armed();
    int st;
    float *s_ph;
    if (!mfma_amp && !prof && net->dims.P == net->dph.P && (naqs::env_int("NAQS_AGG_MERGE", 7) & 1)) {
        st = naqs::net_amp_forward(net, M, keys_dev, s, nullptr, /*launch=*/false);          // (the scratch only)
        if (st != NAQS_OK) return st;
        s_ph = net->d_scratch + (size_t)net->dims.P * net->cap_M;
        const NetDims &d0 = net->dims, &d1 = net->dph;
        const size_t lds = ((size_t)std::max(d0.Ha, d1.Ha) * ((2 * (d0.P - 1) + 1 + 5 + 3) & ~3) + 8) * sizeof(float);
        if (lds > 64 * 1024) return NAQS_ERR_UNSUPPORTED;
        NAQS_KLAUNCH(amp2_kernel, dim3((unsigned)((M + AMP_TILES * WAVE - 1) / (AMP_TILES * WAVE)), (unsigned)d0.P, 2),
                           dim3(AMP_TILES * AMP_SPLIT * WAVE), lds, s, d0, net->d_w, net->d_scratch, feed, d1, net->d_wph, s_ph, M, keys_dev);
        HIP_TRY(hipGetLastError());
    } else {
        if (mfma_amp) st = naqs::net_amp_forward(net, M, keys_dev, s, &feed);
        else {                                             // the VALU form, like the merged launch (same numbers)
            st = naqs::net_amp_forward(net, M, keys_dev, s, nullptr, /*launch=*/false);
            if (st == NAQS_OK) st = launch_amp_kernel(net->dims, net->d_w, M, keys_dev, net->d_scratch, feed, 0, s);
        }
        if (st != NAQS_OK) return st;
        s_ph = net->d_scratch + (size_t)net->dims.P * net->cap_M;
        const ElocFeed none{};
        if (prof) { st = net->prof.begin(s); if (st != NAQS_OK) return st; }
        st = launch_amp_kernel(net->dph, net->d_wph, M, keys_dev, s_ph, none, 1, s);
        if (st != NAQS_OK) return st;
        if (prof) { st = net->prof.end(s); if (st != NAQS_OK) return st; }
    }
    NAQS_KLAUNCH(agg_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, net->dims.P, M, net->d_scratch, s_ph,
                       reinterpret_cast<float2 *>(logpsi_dev), feed);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

// which form of the log-psi kernel M rows get (net_logpsi_impl; naqs::net_logpsi_form for callers that must know in advance)
struct FormSel { bool ws = false, ws_split = false; int rb = 1; };
static FormSel select_form(const naqs_net *net, const int64_t M, const int fmt, const size_t lds_h16, const int rb_max) {
    const NetDims &d = net->dims;
    FormSel f;
    // wave-specialised form (phase_kernel_ws): the published shape in the f16x2 format, tiles of up to 48 rows
    const bool ws_shape = fmt == 2 && d.n_lin == 3 && d.N_pad[0] == PH_WAVES * CBT * 16 && d.N_pad[1] == WS_MW * WS_NCT * 16 &&
                          d.Kh_pad[0] == 32 && d.N_pad[2] == 16 && (d.Ha == 64 || d.Ha == 32) && CBT == 4 && PH_WAVES == 8;
    const int ws_mode = naqs::env_int("NAQS_PHASE_WS", 1);
    int rb = naqs::env_int("NAQS_PHASE_RB", 0);
    if (rb < 1 || rb > rb_max) rb = (int)std::min<int64_t>(rb_max, std::max<int64_t>(1, (M + 16ll * net->cu_count - 1) / (16ll * net->cu_count)));
    bool ws = ws_shape && ws_mode != 0 && !d.phase_sym;        // (the spin-ordered inputs and the sign shift are phase_kernel_h's)
    if (ws) {
        const int rb_ws = std::min(rb, 3);
        if (rb_ws * lds_h16 + (size_t)d.P * rb_ws * 16 * 8 * sizeof(float) > 155 * 1024) ws = false;
        else rb = rb_ws;
    }
    // small tables: two workgroups per tile share the big layer (phase_kernel_ws<RB, SAVE, true>) while both halves of every
    // tile are resident together, one workgroup per CU; NAQS_WS_SPLIT=0: never; NAQS_WS_SPLIT_RB: the tile height of that form
    bool ws_split = false;
    if (ws && naqs::env_int("NAQS_WS_SPLIT", 1) != 0) {
        int rbs = naqs::env_int("NAQS_WS_SPLIT_RB", 0);
        if (rbs < 1 || rbs > 3) rbs = 1;
        const int64_t tiles = (M + 16 * rbs - 1) / (16 * rbs);
        if (2 * tiles <= net->cu_count && rbs * lds_h16 + (size_t)d.P * rbs * 16 * 8 * sizeof(float) <= 155 * 1024) { ws_split = true; rb = rbs; }
    }
    f.ws = ws; f.ws_split = ws_split; f.rb = rb;
    return f;
}

naqs::PhaseForm naqs::net_logpsi_form(const naqs_net *net, const int64_t M, const bool training) {
    PhaseForm f;
    const NetDims &d = net->dims;
    if (net->aggregate) return f;
    const int fmt = phase_format(d);
    if (fmt == 0) return f;
    const size_t lds_h16 = phase_slab_bytes(d, fmt);
    const int rb_max = phase_rb_max(d, fmt);
    const size_t amp_scratch = (size_t)d.P * rb_max * 16 * 8 * sizeof(float);
    const bool amp_in_phase = net->d_wamp != nullptr && net->wamp_fresh && naqs::env_int("NAQS_AMP_MODE", 1) == 1 &&
                              amp_scratch <= 150 * 1024 && d.Ha <= 64;
    const FormSel sel = select_form(net, M, fmt, lds_h16, rb_max);
    const bool wt = sel.ws && !sel.ws_split && !training && naqs::env_int("NAQS_PHASE_WT", 0) != 0;
    f.kind = (sel.ws && amp_in_phase && !wt) ? 1 : 0; f.rb = sel.rb; f.split = sel.ws_split ? 1 : 0;
    return f;
}

int naqs::net_logpsi_impl(naqs_net *net, int64_t M, const uint64_t *keys_dev, float *logpsi_dev, void *stream,
                          const ElocFeed &feed, const PhaseSave &save, const SpecRows *spec) {
    if (!net || M < 0 || (M > 0 && (!keys_dev || !logpsi_dev))) return NAQS_ERR_INVALID;
    if (!net->have_weights) return NAQS_ERR_INVALID;
    if (M == 0) return NAQS_OK;
    if (M >= (1ll << 31)) return NAQS_ERR_UNSUPPORTED;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    st = naqs::poll_check(net->poll);                      // an earlier launch's device-side wait that gave up (naqs_poll.hpp)
    if (st != NAQS_OK) return st;
    const NetDims &d = net->dims;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    st = naqs::net_flush_pack(net, s);                     // (the phase share of the last step's re-pack, if no launch hosted it)
    if (st != NAQS_OK) return st;
    if (net->aggregate) return agg_logpsi(net, M, keys_dev, logpsi_dev, s, feed);
    const int fmt = phase_format(d);
    if (fmt != net->packed_fmt) return NAQS_ERR_INVALID;                       // NAQS_PHASE_MODE changed since naqs_net_set_weights
    const bool use_h = fmt != 0;
    const size_t lds_h16 = use_h ? phase_slab_bytes(d, fmt) : 0;
    const int rb_max = phase_rb_max(d, fmt);
    // amplitude conditionals inside the phase kernel (matrix cores) unless NAQS_AMP_MODE=0 or the width does not tile
    const size_t amp_scratch = (size_t)d.P * rb_max * 16 * 8 * sizeof(float);          // [P][BM][8] raw outputs of the items
    const bool amp_in_phase = use_h && net->d_wamp != nullptr && net->wamp_fresh && naqs::env_int("NAQS_AMP_MODE", 1) == 1 &&
                              amp_scratch <= 150 * 1024 && d.Ha <= 64;      // (128-unit blocks: two fragment sets do not fit the prologue's registers)
    if (spec && !(amp_in_phase && save.x != nullptr && select_form(net, spec->m_var, fmt, lds_h16, rb_max).ws)) return NAQS_ERR_UNSUPPORTED;
    if (!amp_in_phase) {
        st = naqs::net_amp_forward(net, M, keys_dev, s, &feed);
        if (st != NAQS_OK) return st;
    }
    const unsigned short *wamp = amp_in_phase ? net->d_wamp : nullptr;
    naqs::PhaseSave save_dbg = save;
    long long *clk_dev = nullptr;
    if (naqs::env_int("NAQS_DEBUG_CLOCKS", 0) == 1) {            // developer aid: phase boundaries of workgroup 0, in cycles
        HIP_TRY(hipMalloc((void **)&clk_dev, 128 * sizeof(long long)));
        HIP_TRY(hipMemset(clk_dev, 0, 128 * sizeof(long long)));
        save_dbg.clk = clk_dev;
    }

    // rows per workgroup: fill the CUs once if possible (16-row granularity of the MFMA tile)
    if (save.x != nullptr && !use_h) return NAQS_ERR_UNSUPPORTED;             // activations are saved by the split kernels only
    if (!use_h && !net->packed_f32) return NAQS_ERR_INVALID;
    // (a speculative launch, naqs::SpecRows: the form of m_var rows on a grid that covers M)
    const FormSel sel = select_form(net, spec ? spec->m_var : M, fmt, lds_h16, rb_max);
    const bool ws = sel.ws, ws_split = sel.ws_split;
    const int rb = sel.rb;
    // NAQS_PHASE_WT=1 (off by default: measured no faster — see the kernel's header): tables between the two take the transposed
    // big layer (phase_kernel_wt: 80-row tiles, two workgroups per tile, layer 0 just in time) while one round of workgroups covers
    // the table; 2: whatever the size.  The variable must be set when the weights are packed (naqs_net_set_weights).
    const int wt_mode = naqs::env_int("NAQS_PHASE_WT", 0);
    const int64_t wt_tiles = (M + 16 * WT_NT - 1) / (16 * WT_NT);
    const bool wt = ws && !ws_split && wt_mode != 0 && save.x == nullptr && amp_in_phase && net->have_wt && net->d_wt != nullptr &&
                    (wt_mode == 2 || 2 * wt_tiles <= net->cu_count);
    const int bm = rb * 16;
    const unsigned grid = (unsigned)((M + bm - 1) / bm);
    float2 *out = reinterpret_cast<float2 *>(logpsi_dev);
    WsSplit split{nullptr, 0u, net->ctl, spec ? spec->U : nullptr, spec ? spec->P : 0, naqs::SampleFinishJob{}};
    const bool host_fin = spec && spec->host_finish && net->fin_pending && !wt;
    if (host_fin) split.fin = net->fin_job;
    const unsigned hosted = host_fin ? 1u : 0u;
    if (ws_split || wt) {
        // per tile (of up to 80 rows): partial rows + the producer's conditionals  (a speculative launch of the split form may
        // cover more tiles than the form is chosen for: its words exist all the same)
        const size_t tiles_cap = std::max<size_t>(std::max<size_t>((size_t)(net->cu_count / 2), wt ? (size_t)wt_tiles : 0), ws_split ? (size_t)grid : 0);
        const size_t words = tiles_cap * (16 * WT_NT * 4 + MAXP * 16 * WT_NT);
        if (net->d_ws_xchg && net->ws_xchg_words < words) {
            HIP_TRY(hipDeviceSynchronize());
            (void)hipFree(net->d_ws_xchg);
            net->d_ws_xchg = nullptr;
        }
        net->ws_xchg_words = std::max(net->ws_xchg_words, words);
        if (!net->d_ws_xchg) {
            HIP_TRY(hipMalloc((void **)&net->d_ws_xchg, net->ws_xchg_words * sizeof(unsigned long long)));
            HIP_TRY(hipMemset(net->d_ws_xchg, 0, net->ws_xchg_words * sizeof(unsigned long long)));      // tag 0 = never written
            HIP_TRY(hipDeviceSynchronize());               // (a null-stream fill is not ordered against the callers' streams)
            net->ws_seq = 0;
        }
        if (++net->ws_seq == 0u) {                         // the 32-bit call tag is about to repeat: forget every old word
            HIP_TRY(hipMemsetAsync(net->d_ws_xchg, 0, net->ws_xchg_words * sizeof(unsigned long long), s));
            net->ws_seq = 1u;
        }
        split.xchg = net->d_ws_xchg;
        split.tag = net->ws_seq;
    }
    net->last_form_kind = (ws && amp_in_phase && !wt) ? 1 : 0; net->last_form_rb = rb; net->last_form_split = ws_split ? 1 : 0;
    const bool prof = net->prof.armed();
    if (prof) { st = net->prof.begin(s); if (st != NAQS_OK) return st; }
    if (wt) {
        std::snprintf(net->last_kernel, sizeof(net->last_kernel), "phase_kernel_wt<80 rows x 256 units> (f16x2, transposed big layer, amplitude waves beside)");
        const size_t lds = (size_t)WT_RING * 2 * WT_NT * 64 * 16 + (size_t)d.P * 16 * WT_NT * 8 * sizeof(float);
        const int dbg = naqs::env_int("NAQS_WT_DEBUG", 0);           // developer aid (timing only, wrong results): 1 = no big-layer MFMAs, 2 = no layer-0 MFMAs
        if (dbg == 1) NAQS_KLAUNCH((phase_kernel_wt<1>), dim3((unsigned)(2 * wt_tiles)), dim3(PH_THREADS), lds, s, d, net->d_w, net->d_wh, net->d_wt, M, keys_dev, out, feed, wamp, net->d_scales, split, save_dbg);
        else if (dbg == 2) NAQS_KLAUNCH((phase_kernel_wt<2>), dim3((unsigned)(2 * wt_tiles)), dim3(PH_THREADS), lds, s, d, net->d_w, net->d_wh, net->d_wt, M, keys_dev, out, feed, wamp, net->d_scales, split, save_dbg);
        else NAQS_KLAUNCH((phase_kernel_wt<0>), dim3((unsigned)(2 * wt_tiles)), dim3(PH_THREADS), lds, s, d, net->d_w, net->d_wh, net->d_wt, M, keys_dev, out, feed, wamp, net->d_scales, split, save_dbg);
    } else if (ws) std::snprintf(net->last_kernel, sizeof(net->last_kernel), "phase_kernel_ws<RB=%d, SAVE=%d%s> (f16x2%s)", rb,
                          save.x != nullptr ? 1 : 0, ws_split ? ", SPLIT=1" : "", amp_in_phase ? ", amplitude waves beside the matrix waves" : "");
    else if (use_h) std::snprintf(net->last_kernel, sizeof(net->last_kernel), "phase_kernel_h<RB=%d, SAVE=%d, FMT=%d (%s)>%s", rb, save.x != nullptr ? 1 : 0,
                             fmt, fmt == 2 ? "f16x2" : "bf16x3", amp_in_phase ? " incl. amplitude prologue" : "");
    else std::snprintf(net->last_kernel, sizeof(net->last_kernel), "phase_kernel<RB=%d> (f32 MFMA)", rb);
    if (wt) {
    } else if (ws) {
        const size_t lds = rb * lds_h16 + (size_t)d.P * bm * 8 * sizeof(float);
        const int flags = naqs::env_int("NAQS_WS_FLAGS", 0);
#define NAQS_WS_LAUNCH(RB)                                                                                                              \
        do {                                                                                                                            \
            if (save.x != nullptr) NAQS_KLAUNCH((phase_kernel_ws<RB, true>), dim3(grid + hosted), dim3(PH_THREADS), lds, s, d, net->d_w, net->d_wh, M, keys_dev, net->d_scratch, out, feed, save_dbg, wamp, net->d_scales, flags, split); \
            else NAQS_KLAUNCH((phase_kernel_ws<RB, false>), dim3(grid + hosted), dim3(PH_THREADS), lds, s, d, net->d_w, net->d_wh, M, keys_dev, net->d_scratch, out, feed, save_dbg, wamp, net->d_scales, flags, split); \
        } while (0)
#define NAQS_WS_LAUNCH_SPLIT(RB)                                                                                                        \
        do {                                                                                                                            \
            if (save.x != nullptr) NAQS_KLAUNCH((phase_kernel_ws<RB, true, true>), dim3(2 * grid + hosted), dim3(PH_THREADS), lds, s, d, net->d_w, net->d_wh, M, keys_dev, net->d_scratch, out, feed, save_dbg, wamp, net->d_scales, flags, split); \
            else NAQS_KLAUNCH((phase_kernel_ws<RB, false, true>), dim3(2 * grid + hosted), dim3(PH_THREADS), lds, s, d, net->d_w, net->d_wh, M, keys_dev, net->d_scratch, out, feed, save_dbg, wamp, net->d_scales, flags, split); \
        } while (0)
        if (ws_split) {
            switch (rb) {
                case 1: NAQS_WS_LAUNCH_SPLIT(1); break;
                case 2: NAQS_WS_LAUNCH_SPLIT(2); break;
                default: NAQS_WS_LAUNCH_SPLIT(3); break;
            }
        } else {
            switch (rb) {
                case 1: NAQS_WS_LAUNCH(1); break;
                case 2: NAQS_WS_LAUNCH(2); break;
                default: NAQS_WS_LAUNCH(3); break;
            }
        }
#undef NAQS_WS_LAUNCH_SPLIT
#undef NAQS_WS_LAUNCH
        if (hosted) net->fin_pending = false;             // (workgroup 0 of that launch is the sampler's finish job)
    } else if (use_h) {
        const size_t lds = std::max(rb * lds_h16, amp_in_phase ? amp_scratch : (size_t)0);
#define NAQS_PH_LAUNCH(RB, FMT)                                                                                                         \
        do {                                                                                                                            \
            if (save.x != nullptr) NAQS_KLAUNCH((phase_kernel_h<RB, true, FMT>), dim3(grid), dim3(PH_THREADS), lds, s, d, net->d_w, net->d_wh, M, keys_dev, net->d_scratch, out, feed, save_dbg, wamp, net->d_scales); \
            else NAQS_KLAUNCH((phase_kernel_h<RB, false, FMT>), dim3(grid), dim3(PH_THREADS), lds, s, d, net->d_w, net->d_wh, M, keys_dev, net->d_scratch, out, feed, save_dbg, wamp, net->d_scales); \
        } while (0)
        if (fmt == 2) {
            switch (rb) {
                case 1: NAQS_PH_LAUNCH(1, 2); break;
                case 2: NAQS_PH_LAUNCH(2, 2); break;
                case 3: NAQS_PH_LAUNCH(3, 2); break;
                default: NAQS_PH_LAUNCH(4, 2); break;
            }
        } else {
            switch (rb) {
                case 1: NAQS_PH_LAUNCH(1, 1); break;
                case 2: NAQS_PH_LAUNCH(2, 1); break;
                default: NAQS_PH_LAUNCH(3, 1); break;
            }
        }
#undef NAQS_PH_LAUNCH
    } else {
        const size_t lds = (size_t)bm * d.ld * sizeof(float);
        switch (rb) {
            case 1: NAQS_KLAUNCH(phase_kernel<1>, dim3(grid), dim3(PH_THREADS), lds, s, d, net->d_w, M, keys_dev, net->d_scratch, out, feed); break;
            case 2: NAQS_KLAUNCH(phase_kernel<2>, dim3(grid), dim3(PH_THREADS), lds, s, d, net->d_w, M, keys_dev, net->d_scratch, out, feed); break;
            case 3: NAQS_KLAUNCH(phase_kernel<3>, dim3(grid), dim3(PH_THREADS), lds, s, d, net->d_w, M, keys_dev, net->d_scratch, out, feed); break;
            default: NAQS_KLAUNCH(phase_kernel<4>, dim3(grid), dim3(PH_THREADS), lds, s, d, net->d_w, M, keys_dev, net->d_scratch, out, feed); break;
        }
    }
    HIP_TRY(hipGetLastError());
    if (prof) { st = net->prof.end(s); if (st != NAQS_OK) return st; }
    if (clk_dev) {
        long long h[128];
        HIP_TRY(hipMemcpy(h, clk_dev, sizeof(h), hipMemcpyDeviceToHost));
        (void)hipFree(clk_dev);
        for (int wv = 0; wv < 8; ++wv) {
            std::fprintf(stderr, "[naqs clocks] wave %d:", wv);
            for (int k = 1; k < 16; ++k) std::fprintf(stderr, " %lld", h[wv * 16 + k] ? h[wv * 16 + k] - h[wv * 16] : 0ll);
            std::fprintf(stderr, "\n");
        }
    }
    return NAQS_OK;
}

NAQS_API int naqs_net_logpsi(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, float *logpsi_dev, void *stream) {
    ElocFeed none{};
    return naqs::net_logpsi_impl(net, M, keys_dev, logpsi_dev, stream, none, naqs::PhaseSave{});
}

NAQS_API int naqs_logpsi_eloc(naqs_net_t *net, naqs_ham_t *ham, int64_t M, const uint64_t *keys_dev,
                              const double *w_dev, float *logpsi_dev, double *eloc_dev, double *out4_dev,
                              void *stream) {
    if (!net || !ham || M < 0 || (w_dev == nullptr) != (out4_dev == nullptr)) return NAQS_ERR_INVALID;
    if (M > 0 && (!keys_dev || !logpsi_dev || !eloc_dev)) return NAQS_ERR_INVALID;
    if (naqs::ham_device(ham) != net->device) return NAQS_ERR_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    if (M == 0) {
        if (out4_dev) HIP_TRY(hipMemsetAsync(out4_dev, 0, 4 * sizeof(double), s));
        return NAQS_OK;
    }
    ElocFeed feed{};
    st = naqs::eloc_begin(ham, M, s, &feed);
    if (st != NAQS_OK) return st;
    st = naqs::net_logpsi_impl(net, M, keys_dev, logpsi_dev, stream, feed, naqs::PhaseSave{});      // amp kernel feeds keys, phase kernel feeds psi
    if (st != NAQS_OK) return st;
    return naqs::eloc_main(ham, M, feed, eloc_dev, w_dev, out4_dev, s);
}

NAQS_API int naqs_net_prof_enable(naqs_net_t *net, int max_records) {
    if (!net || max_records < 0) return NAQS_ERR_INVALID;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    return (net->prof_which == 1 ? net->prof_samp : net->prof).enable(max_records);
}

NAQS_API int naqs_net_prof_select(naqs_net_t *net, int which) {
    if (!net || which < 0 || which > 1) return NAQS_ERR_INVALID;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    (void)net->prof.enable(0);
    (void)net->prof_samp.enable(0);
    net->prof_which = which;
    return NAQS_OK;
}

NAQS_API int naqs_net_prof_read(naqs_net_t *net, double *total_ms, int64_t *launches) {
    if (!net || !total_ms || !launches) return NAQS_ERR_INVALID;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    return (net->prof_which == 1 ? net->prof_samp : net->prof).read(total_ms, launches);
}

NAQS_API int naqs_net_last_kernel(const naqs_net_t *net, char *buf, int buf_len) {
    if (!net || !buf || buf_len <= 0) return NAQS_ERR_INVALID;
    std::snprintf(buf, (size_t)buf_len, "%s", net->aggregate ? "amp_mfma_kernel + amp_kernel(raw) + agg_finish_kernel" : net->last_kernel);
    return NAQS_OK;
}

NAQS_API int naqs_net_prof_stride(naqs_net_t *net, int stride) {
    if (!net || stride < 1) return NAQS_ERR_INVALID;
    naqs::EventRing &ring = net->prof_which == 1 ? net->prof_samp : net->prof;
    ring.stride = stride;
    ring.tick = 0;
    return NAQS_OK;
}
