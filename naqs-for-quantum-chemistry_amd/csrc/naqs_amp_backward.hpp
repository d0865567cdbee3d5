// naqs_amp_backward.hpp — the amplitude blocks' backward pass for one orbital pair, as a device function shared by
// amp_backward_kernel (naqs_grad.hip) and the training step's fused backward launch (naqs_phase_grad.hip:
// backward_mega_kernel).  See naqs_grad.hip for the formulation.
#pragma once
#include <cstdint>

#include "naqs_common.hpp"
#include "naqs_net.hpp"

namespace naqs {
namespace ampbw {

using naqs::MAXP;
using naqs::NetDims;
using naqs::WAVE;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_TILE_WGS = 64;   // workgroups per pair (workgroup wg walks tiles wg, wg + n_wgs, ...)

struct AmpSrc { int64_t off[MAXP]; };     // flat (state_dict) offset of pair n's parameters

// one orbital pair NB: all tiles of this workgroup.  Round 3: a tile is 64 samples and the workgroup has one wave per 16
// hidden units (Ha / 16 waves: 4 for the published 64-unit blocks, 8 for the reference's default 128) — wave q owns hidden
// units 16 q .. 16 q + 15 in BOTH stages, lane = sample:
//   (1) forward of its 16 units from the key bits (weights staged in LDS), partial outputs -> LDS; barrier; every wave adds
//       the partials in fixed order, forms d log-amp / d outputs (softmax residual through the symmetrisation,
//       nade.py:585-586) scaled by g_i, and the d pre-activations of its own units; h and d-pre go to two LDS tiles
//       [unit][sample];
//   (2) the sums over the tile's samples are GEMMs with the sample axis as K — dW1^T[k][j] = sum_s x[s][k] dpre[s][j] (x =
//       +-1 from the input bits, bias = an input that is always 1), dW2[c][j] = sum_s dout[s][c] h[s][j] — on the f32 matrix
//       cores (v_mfma_f32_16x16x4_f32), accumulators living across all tiles of the workgroup.
// (Rounds 1-2: 256-sample tiles with thread = sample walking ALL hidden units — a serial chain of Ha LDS-fed iterations per
// thread, twice: 37 us for the one tile a workgroup gets at M ~ 1 200, whatever the tile size, and 2 x 67 us for 128-unit
// blocks.  Splitting the units over the waves cuts the chain by Ha / 16 and gives four times as many workgroups.)
// smem: weights | d-pre tile [Ha][65] | h tile [Ha][65] | d-out [5][64] | input bits [64] | partial outputs [Ha/16][5][64]
// raw: phase blocks of an aggregate-phase network (d describes them: 4 outputs, no symmetry) — the differentiated
// quantity is the raw output of the realised outcome, d out[c] = g_i [c == occ], no conditional in between.
constexpr int GT = 64;                                    // samples per tile = lanes of a wave
template <int NB>
__device__ __forceinline__ void amp_backward_pair(const NetDims &d, const float *__restrict__ w, const int64_t M,
                                                  const uint64_t *__restrict__ keys, const float *__restrict__ g,
                                                  float *__restrict__ out, float *smem, const int raw, const int wg,
                                                  const int n_wgs, const int g_stride) {
    constexpr int NIN = NB == 0 ? 1 : 2 * NB;
    constexpr int S = (NIN + 1 + 5 + 3) & ~3;
    constexpr int RT = (NIN + 1 + 15) / 16;               // 16-row tiles of the input axis (inputs + the bias input)
    constexpr int LD = GT + 1;
    const int Ha = d.Ha, nout = d.n_out_amp, NW = Ha >> 4, NT = NW * WAVE;
    const int w_floats = (Ha * S + 8 + 3) & ~3;
    float *s_w = smem;
    float *s_dpre = s_w + w_floats;
    float *s_h = s_dpre + Ha * LD;
    float *s_do = s_h + Ha * LD;                          // [5][GT]
    uint32_t *s_x = reinterpret_cast<uint32_t *>(s_do + 5 * GT);
    float *s_part = reinterpret_cast<float *>(s_x + GT);  // [NW][5][GT]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, kq = lane >> 4;
    {
        const f32x4 *from = reinterpret_cast<const f32x4 *>(w + d.amp_off[NB]);
        f32x4 *to = reinterpret_cast<f32x4 *>(s_w);
        for (int e = tid; e < (Ha * S + 8) / 4; e += NT) to[e] = from[e];
    }
    const float *b2 = s_w + Ha * S;
    const int j0 = wave * 16;                              // this wave's hidden units
    f32x4 acc1[RT], acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc1[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float accb2[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int64_t t0 = (int64_t)wg * GT; t0 < M; t0 += (int64_t)n_wgs * GT) {
        const int64_t i = t0 + lane;
        const bool valid = i < M;
        const uint64_t key = valid ? keys[i] : 0ull;
        uint32_t abits = 0, bbits = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            abits |= (uint32_t)((key >> d.qa[k]) & 1ull) << k;
            bbits |= (uint32_t)((key >> d.qb[k]) & 1ull) << k;
        }
        const int occ = (int)((key >> d.qa[NB]) & 1ull) + 2 * (int)((key >> d.qb[NB]) & 1ull);
        const bool swap = (raw ? d.phase_sym != 0 : d.sym != 0) && abits > bbits;      // (raw: the phase blocks' spin-ordered inputs)
        const uint32_t first = swap ? bbits : abits, second = swap ? abits : bbits;
        float x[NIN];
        if (NB == 0) {
            x[0] = 0.0f;                                   // pair 0 sees a constant-zero input (nade.py:509-511)
        } else {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                x[k] = ((first >> k) & 1u) ? 1.0f : -1.0f;
                x[NB + k] = ((second >> k) & 1u) ? 1.0f : -1.0f;
            }
        }
        const float gi = valid ? g[i * g_stride] : 0.0f;      // (g_stride 2: a column of the loss gradient [M][2])
        // forward of this wave's 16 hidden units; the activations go to their tile, the partial outputs to LDS
        float o[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int jj = j0; jj < j0 + 16; ++jj) {
            float rv[S];
            naqs::load_row<S>(s_w + jj * S, rv);
            float h0 = rv[NIN], h1 = 0.0f;
#pragma unroll
            for (int k = 0; k + 1 < NIN; k += 2) { h0 = fmaf(rv[k], x[k], h0); h1 = fmaf(rv[k + 1], x[k + 1], h1); }
            if (NIN & 1) h0 = fmaf(rv[NIN - 1], x[NIN - 1], h0);
            const float h = fmaxf(h0 + h1, 0.0f);
            s_h[jj * LD + lane] = h;
#pragma unroll
            for (int c = 0; c < 5; ++c)
                if (c < nout) o[c] = fmaf(rv[NIN + 1 + c], h, o[c]);
        }
#pragma unroll
        for (int c = 0; c < 5; ++c) s_part[(wave * 5 + c) * GT + lane] = o[c];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            float v = c < nout ? b2[c] : 0.0f;
            for (int q = 0; q < NW; ++q) v += s_part[(q * 5 + c) * GT + lane];            // fixed order: wave 0 first
            o[c] = v;
        }
        float da4[4];
        if (raw) {
#pragma unroll
            for (int c = 0; c < 4; ++c) da4[c] = valid && c == naqs::phase_out_row(d, occ) ? gi : 0.0f;
        } else {
            float la[4];
            bool ok[4];
            naqs::amp_conditional<true>(d, NB, o, abits, bbits, la, ok);
            // d la[occ] / d a4[c] = [c == occ] - softmax(2 a4)[c] on the allowed outcomes
            const bool live = valid && (occ == 0 ? ok[0] : (occ == 1 ? ok[1] : (occ == 2 ? ok[2] : ok[3])));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float sm = ok[c] ? expf(2.0f * la[c]) : 0.0f;
                da4[c] = live && ok[c] ? gi * ((c == occ ? 1.0f : 0.0f) - sm) : 0.0f;
            }
        }
        float dout[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (d.sym) {                                       // transpose of amp_symmetrise
            const int x_order = abits > bbits ? 0 : (abits == bbits ? 1 : 2);
            dout[0] = da4[0];
            dout[2] = da4[3];
            dout[1] = 0.5f * (da4[1] + da4[2]);
            if (x_order == 1) dout[1] += 0.5f * (da4[1] + da4[2]);
            else if (x_order == 0) { dout[3] = 0.5f * da4[1]; dout[4] = 0.5f * da4[2]; }
            else { dout[4] = 0.5f * da4[1]; dout[3] = 0.5f * da4[2]; }
        } else {
            dout[0] = da4[0]; dout[1] = da4[1]; dout[2] = da4[2]; dout[3] = da4[3];
        }
        if (wave == 0) {                                   // (every wave computed the same d-out; one copy for the GEMMs)
#pragma unroll
            for (int c = 0; c < 5; ++c) { s_do[c * GT + lane] = dout[c]; accb2[c] += dout[c]; }
            s_x[lane] = (NB == 0 ? 0u : (first | (second << NB))) | (1u << NIN);   // bit NIN: the bias input
        }
        // d pre-activations of this wave's units -> tile (h > 0 <=> pre > 0); W2[:][jj] sits at floats NIN+1.. of the packed row
#pragma unroll 4
        for (int jj = j0; jj < j0 + 16; ++jj) {
            const float *row = s_w + jj * S + NIN + 1;
            float dh = 0.0f;
#pragma unroll
            for (int c = 0; c < 5; ++c)
                if (c < nout) dh = fmaf(row[c], dout[c], dh);
            s_dpre[jj * LD + lane] = s_h[jj * LD + lane] > 0.0f ? dh : 0.0f;
        }
        __syncthreads();
        // sums over the tile's samples on the matrix cores: hidden tile ct = wave
        {
            const float *bd = s_dpre + (j0 + m) * LD + kq, *bh = s_h + (j0 + m) * LD + kq;
            const float *ad = s_do + m * GT + kq;
#pragma unroll 4
            for (int s0 = 0; s0 < GT; s0 += 4) {
                const uint32_t xb = s_x[s0 + kq];
                const float vd = bd[s0], vh = bh[s0];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int k = rt * 16 + m;
                    const float a = k <= NIN ? (((xb >> k) & 1u) ? 1.0f : -1.0f) : 0.0f;
                    acc1[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vd, acc1[rt], 0, 0, 0);
                }
                const float a2 = m < 5 ? ad[s0] : 0.0f;
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, vh, acc2, 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // partial sums of this workgroup in state_dict order: W1 [Ha][NIN], b1 [Ha], W2 [nout][Ha], b2 [nout]
    {
        const int j = j0 + m;                               // D layout: col = lane & 15 (hidden unit), row = 4 (lane >> 4) + r
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = rt * 16 + 4 * kq + r;
                if (k < NIN) out[j * NIN + k] = NB == 0 ? 0.0f : acc1[rt][r];
                else if (k == NIN) out[Ha * NIN + j] = acc1[rt][r];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * kq + r;
            if (c < nout) out[Ha * NIN + Ha + c * Ha + j] = acc2[r];
        }
    }
    // db2[c] = sum over this workgroup's samples of d-out[c] (wave 0 holds them)
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            float v = accb2[c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && c < nout) out[Ha * NIN + Ha + nout * Ha + c] = v;
        }
    }
}


// floats of LDS one workgroup needs (any pair of the network)
inline size_t smem_floats(const NetDims &d) {
    const int nin_max = 2 * (d.P - 1);
    const int S_max = (nin_max + 1 + 5 + 3) & ~3;
    const int NW = d.Ha >> 4;
    return (size_t)((d.Ha * S_max + 8 + 3) & ~3) + 2 * (size_t)d.Ha * (GT + 1) + 5 * GT + GT + (size_t)NW * 5 * GT;
}

// all pairs: the switch over the compile-time pair index
__device__ __forceinline__ void pair_dispatch(const int n, const NetDims &d, const float *__restrict__ w, const int64_t M,
                                              const uint64_t *__restrict__ keys, const float *__restrict__ g, float *__restrict__ out,
                                              float *smem, const int raw, const int wg, const int n_wgs, const int g_stride = 1) {
    switch (n) {
#define NAQS_CASE(NB) case NB: amp_backward_pair<NB>(d, w, M, keys, g, out, smem, raw, wg, n_wgs, g_stride); break;
        NAQS_CASE(0) NAQS_CASE(1) NAQS_CASE(2) NAQS_CASE(3) NAQS_CASE(4) NAQS_CASE(5) NAQS_CASE(6) NAQS_CASE(7)
        NAQS_CASE(8) NAQS_CASE(9) NAQS_CASE(10) NAQS_CASE(11) NAQS_CASE(12) NAQS_CASE(13) NAQS_CASE(14) NAQS_CASE(15)
#undef NAQS_CASE
        default: break;
    }
}

}  // namespace ampbw
}  // namespace naqs
