// naqs_grad.hip — training-time evaluation of the amplitude half of the orbital NADE on gfx950 (MI355X):
//   naqs_net_logamp        log|psi|(key_i) = sum_n [conditional log-amplitude of the realised outcome of pair n]
//   naqs_net_amp_backward  d/d theta  sum_i g_i log|psi|(key_i)  for every amplitude-block parameter
// i.e. the forward and backward of the amplitude part of _forward_predict (src/naqs/network/nade.py:738-770 with the
// helpers :417-630, activations.py:40-46) that the reference leaves to PyTorch autograd (energy.py:329-343).  The eager
// formulation is ~10 orbital pairs x (2 Linear + mask + softmax + gathers), forward and backward; here it is two
// kernels each way, and the gradient is deterministic (fixed-order reductions, no float atomics).
//
// amp_backward_kernel: workgroup = one orbital pair n (blockIdx.y) x a strided set of 256-sample tiles.  Per tile every
// thread owns a sample: it recomputes the block's forward from the key bits (weights staged in LDS), forms
// d log-amp / d outputs (softmax residual through the symmetrisation, nade.py:585-586) scaled by g_i and
// back-propagates to the hidden pre-activations.  The per-sample factors go through one LDS tile [hidden][sample];
// the threads then re-partition over the PARAMETERS (thread = hidden unit j x a subset of input columns) and walk
// the tile's samples, accumulating dW1, db1 (from d pre) and dW2, db2 (from h) in registers across all tiles of the
// workgroup.  Partial sums per workgroup go to scratch in state_dict order; amp_reduce_kernel adds them in order.

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "naqs_common.hpp"
#include "naqs_net.hpp"

namespace {

using naqs::MAXP;
using naqs::NetDims;
using naqs::WAVE;
using naqs::DeviceGuard;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_TILE_WGS = 64;   // workgroups per pair (each walks tiles blockIdx.x, +gridDim.x, ...)

struct AmpSrc { int64_t off[MAXP]; };     // flat (state_dict) offset of pair n's parameters

// out[i] = sum_n scratch[n][i] in the order of the fused log-psi epilogue
__global__ __launch_bounds__(256) void logamp_sum_kernel(int P, int64_t M, const float *__restrict__ scratch,
                                                         float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float s = 0.0f;
    for (int n = 0; n < P; ++n) s += scratch[(int64_t)n * M + i];
    out[i] = s;
}

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
                                                  float *__restrict__ out, float *smem, const int raw) {
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

    for (int64_t t0 = (int64_t)blockIdx.x * GT; t0 < M; t0 += (int64_t)gridDim.x * GT) {
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
        const bool swap = d.sym && abits > bbits;
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
        const float gi = valid ? g[i] : 0.0f;
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
            for (int c = 0; c < 4; ++c) da4[c] = valid && c == occ ? gi : 0.0f;
        } else {
            float la[4];
            bool ok[4];
            naqs::amp_conditional(d, NB, o, abits, bbits, la, ok);
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

__global__ __launch_bounds__(512) void amp_backward_kernel(const NetDims d, const float *__restrict__ w, const int64_t M,
                                                           const uint64_t *__restrict__ keys, const float *__restrict__ g,
                                                           float *__restrict__ partial, const int64_t partial_stride,
                                                           const AmpSrc src, const int raw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = blockIdx.y;
    float *out = partial + (int64_t)blockIdx.x * partial_stride + src.off[n];
    switch (n) {
#define CASE(NB) case NB: amp_backward_pair<NB>(d, w, M, keys, g, out, smem, raw); break;
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7)
        CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
#undef CASE
        default: break;
    }
}

__global__ __launch_bounds__(256) void amp_reduce_kernel(int64_t count, int n_partials, int64_t partial_stride,
                                                         const float *__restrict__ partial, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    float s = 0.0f;
    for (int b = 0; b < n_partials; ++b) s += partial[(int64_t)b * partial_stride + e];
    out[e] = s;
}

// inputs of the phase block from key bits: x[i][c] = +-1 occupation of (alpha of model pairs 0..P-2 | beta of the same),
// occ[i] = realised outcome of the last pair (selects the phase output, nade.py:563-569)
__global__ __launch_bounds__(256) void phase_inputs_kernel(const NetDims d, const int64_t M, const uint64_t *__restrict__ keys,
                                                           float *__restrict__ x, int64_t *__restrict__ occ) {
    const int W = 2 * (d.P - 1);
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * W) return;
    const int64_t i = e / W;
    const int c = (int)(e - i * W);
    const uint64_t key = keys[i];
    const int bit = c < d.P - 1 ? d.qa[c] : d.qb[c - (d.P - 1)];
    x[e] = ((key >> bit) & 1ull) ? 1.0f : -1.0f;
    if (c == 0) occ[i] = (int64_t)((key >> d.qa[d.P - 1]) & 1ull) + 2 * (int64_t)((key >> d.qb[d.P - 1]) & 1ull);
}

// d loss / d (log|psi|, phase) of the VMC loss 2 Re sum_i w_i log psi_i (E_loc_i - <E>)^* in the reference's float32
// arithmetic (energy.py:328-329 with complex.py:49-58): g = (2 w Re(E_loc - <E>), -2 w Im(E_loc - <E>))
__global__ __launch_bounds__(256) void vmc_grad_kernel(const int64_t M, const double2 *__restrict__ eloc,
                                                       const double *__restrict__ w, const double *__restrict__ sums,
                                                       float2 *__restrict__ g, double *__restrict__ ev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ev != nullptr && i == 0) {                     // <E> and Var of energy.py:372-375 from the same accumulators
        const double e_mean = sums[0] / sums[3];
        ev[0] = e_mean;
        ev[1] = sums[2] / sums[3] - e_mean * e_mean;
    }
    if (i >= M) return;
    const float m_re = (float)sums[0], m_im = (float)sums[1];
    const double2 e = eloc[i];
    const float two_w = 2.0f * (float)w[i];
    g[i] = make_float2(((float)e.x - m_re) * two_w, -(((float)e.y - m_im) * two_w));
}

// Adam on one flat parameter vector (naqs::adam_update, naqs_net.hpp)
__global__ __launch_bounds__(256) void adam_kernel(const int64_t n, const float *__restrict__ g, const naqs::AdamArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    naqs::adam_update(a, i, g[i]);
}

// the amp kernel of naqs_logpsi.hip is reached through naqs::net_amp_forward
}  // namespace

NAQS_API int naqs_adam_step(int64_t n, float *param_dev, const float *grad_dev, float *exp_avg_dev, float *exp_avg_sq_dev,
                            double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                            void *stream) {
    if (n < 0 || step < 1 || (n > 0 && (!param_dev || !grad_dev || !exp_avg_dev || !exp_avg_sq_dev))) return NAQS_ERR_INVALID;
    if (n == 0) return NAQS_OK;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, grad_dev,
                       naqs::adam_args(param_dev, exp_avg_dev, exp_avg_sq_dev, lr, beta1, beta2, eps, weight_decay, step));
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

NAQS_API int naqs_net_phase_inputs(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, float *x_dev, int64_t *occ_dev,
                                   void *stream) {
    if (!net || M < 0 || (M > 0 && (!keys_dev || !x_dev || !occ_dev))) return NAQS_ERR_INVALID;
    if (M == 0) return NAQS_OK;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    const int64_t total = M * 2 * (net->dims.P - 1);
    hipLaunchKernelGGL(phase_inputs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       net->dims, M, keys_dev, x_dev, occ_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

static int vmc_loss_grad_impl(int64_t M, const double *eloc_dev, const double *w_dev, const double *sums_dev, float *g_dev,
                              double *ev_dev, void *stream) {
    if (M < 0 || (M > 0 && (!eloc_dev || !w_dev || !sums_dev || !g_dev))) return NAQS_ERR_INVALID;
    if (M == 0 && ev_dev == nullptr) return NAQS_OK;
    if (!sums_dev) return NAQS_ERR_INVALID;
    hipLaunchKernelGGL(vmc_grad_kernel, dim3((unsigned)std::max<int64_t>(1, (M + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), M, reinterpret_cast<const double2 *>(eloc_dev), w_dev, sums_dev,
                       reinterpret_cast<float2 *>(g_dev), ev_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

NAQS_API int naqs_vmc_loss_grad(int64_t M, const double *eloc_dev, const double *w_dev, const double *sums_dev, float *g_dev,
                                void *stream) {
    return vmc_loss_grad_impl(M, eloc_dev, w_dev, sums_dev, g_dev, nullptr, stream);
}

NAQS_API int naqs_vmc_loss_grad_ev(int64_t M, const double *eloc_dev, const double *w_dev, const double *sums_dev, float *g_dev,
                                   double *ev_dev, void *stream) {
    if (!ev_dev) return NAQS_ERR_INVALID;
    return vmc_loss_grad_impl(M, eloc_dev, w_dev, sums_dev, g_dev, ev_dev, stream);
}

namespace {
__global__ __launch_bounds__(1024) void shard_proof_kernel(const int64_t M, const uint64_t *__restrict__ keys,
                                                           const double *__restrict__ sums, double *__restrict__ ext) {
    __shared__ unsigned long long s_part[16];
    unsigned long long acc = 0ull;
    for (int64_t i = threadIdx.x; i < M; i += 1024) acc += keys[i];          // modulo 2^64: order-independent
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0ull;
        for (int w = 0; w < 16; ++w) t += s_part[w];
        const double c = (double)(t & 0xFFFFFull), m = (double)M;
        ext[0] = sums[0]; ext[1] = sums[1]; ext[2] = sums[2]; ext[3] = sums[3];
        ext[4] = m; ext[5] = m * m; ext[6] = c; ext[7] = c * c;
    }
}
}  // namespace

NAQS_API int naqs_shard_proof(int64_t M, const uint64_t *keys_dev, const double *sums_dev, double *ext_dev, void *stream) {
    if (M < 0 || (M > 0 && !keys_dev) || !sums_dev || !ext_dev) return NAQS_ERR_INVALID;
    hipLaunchKernelGGL(shard_proof_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), M, keys_dev, sums_dev, ext_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

NAQS_API int naqs_net_amp_param_count(const naqs_net_t *net, int64_t *count) {
    if (!net || !count) return NAQS_ERR_INVALID;
    *count = net->amp_params;
    return NAQS_OK;
}

NAQS_API int naqs_net_logamp(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, float *logamp_dev, void *stream) {
    if (!net || M < 0 || (M > 0 && (!keys_dev || !logamp_dev))) return NAQS_ERR_INVALID;
    if (!net->have_amp_weights) return NAQS_ERR_INVALID;
    if (M == 0) return NAQS_OK;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    st = naqs::net_amp_forward(net, M, keys_dev, s);
    if (st != NAQS_OK) return st;
    hipLaunchKernelGGL(logamp_sum_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, net->dims.P, M, net->d_scratch,
                       logamp_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

// gradient of sum_i g_i f(key_i) for one set of per-pair blocks (amplitude blocks: f = log|psi|; raw: the phase blocks
// of an aggregate-phase network, f = phase): partial sums per workgroup, then a fixed-order reduction
int naqs::net_blocks_backward(naqs_net *net, const NetDims &d, const float *w, const int64_t *src_off, int64_t n_block_params,
                              int64_t M, const uint64_t *keys_dev, const float *g_dev, float *grad_dev, int raw, hipStream_t s,
                              BlockReduceJob *defer, int slot) {
    if (d.Ha > 128 || (d.Ha & 15)) return NAQS_ERR_UNSUPPORTED;                 // a wave owns 16-unit hidden tiles
    if (M == 0) {
        if (defer) return NAQS_ERR_INVALID;
        HIP_TRY(hipMemsetAsync(grad_dev, 0, (size_t)n_block_params * sizeof(float), s));
        return NAQS_OK;
    }
    const int n_wg = (int)std::min<int64_t>(MAX_TILE_WGS, (M + GT - 1) / GT);
    const int64_t stride = (std::max(net->amp_params, net->ph_params) + 3) & ~3ll;
    if (!net->d_gpart) {
        HIP_TRY(hipMalloc((void **)&net->d_gpart, 2 * (size_t)MAX_TILE_WGS * stride * sizeof(float)));
    }
    float *gpart = net->d_gpart + (slot ? (size_t)MAX_TILE_WGS * stride : 0);
    const int nin_max = 2 * (d.P - 1);
    const int S_max = (nin_max + 1 + 5 + 3) & ~3;
    const int NW = d.Ha >> 4;
    const size_t lds = ((size_t)((d.Ha * S_max + 8 + 3) & ~3) + 2 * (size_t)d.Ha * (GT + 1) + 5 * GT + GT + (size_t)NW * 5 * GT) * sizeof(float);
    if (lds > 156 * 1024) return NAQS_ERR_UNSUPPORTED;
    if (!net->grad_attr_set) {
        const int lds_max = 156 * 1024;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&amp_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        net->grad_attr_set = true;
    }
    AmpSrc src;
    for (int n = 0; n < MAXP; ++n) src.off[n] = src_off[n] - src_off[0];         // relative to this set's first parameter
    hipLaunchKernelGGL(amp_backward_kernel, dim3((unsigned)n_wg, (unsigned)d.P), dim3((unsigned)(NW * WAVE)), lds, s, d, w, M, keys_dev,
                       g_dev, gpart, stride, src, raw);
    HIP_TRY(hipGetLastError());
    if (defer) {
        defer->count = n_block_params; defer->stride = stride; defer->n_partials = n_wg; defer->partial = gpart;
        return NAQS_OK;
    }
    hipLaunchKernelGGL(amp_reduce_kernel, dim3((unsigned)((n_block_params + 255) / 256)), dim3(256), 0, s, n_block_params, n_wg,
                       stride, gpart, grad_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

NAQS_API int naqs_net_amp_backward(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, const float *g_dev,
                                   float *grad_dev, void *stream) {
    if (!net || M < 0 || !grad_dev || (M > 0 && (!keys_dev || !g_dev))) return NAQS_ERR_INVALID;
    if (!net->have_amp_weights) return NAQS_ERR_INVALID;
    DeviceGuard guard;
    int st = guard.init(net->device);
    if (st != NAQS_OK) return st;
    return naqs::net_blocks_backward(net, net->dims, net->d_w, net->amp_src_off, net->amp_params, M, keys_dev, g_dev, grad_dev, 0,
                                     reinterpret_cast<hipStream_t>(stream));
}
