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
#include "naqs_amp_backward.hpp"

namespace {

using naqs::MAXP;
using naqs::NetDims;
using naqs::WAVE;
using naqs::DeviceGuard;
typedef float f32x4 __attribute__((ext_vector_type(4)));

using naqs::ampbw::MAX_TILE_WGS;
using naqs::ampbw::AmpSrc;
using naqs::ampbw::GT;

// out[i] = sum_n scratch[n][i] in the order of the fused log-psi epilogue
__global__ __launch_bounds__(256) void logamp_sum_kernel(int P, int64_t M, const float *__restrict__ scratch,
                                                         float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float s = 0.0f;
    for (int n = 0; n < P; ++n) s += scratch[(int64_t)n * M + i];
    out[i] = s;
}

__global__ __launch_bounds__(512) void amp_backward_kernel(const NetDims d, const float *__restrict__ w, const int64_t M,
                                                           const uint64_t *__restrict__ keys, const float *__restrict__ g,
                                                           float *__restrict__ partial, const int64_t partial_stride,
                                                           const AmpSrc src, const int raw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = blockIdx.y;
    float *out = partial + (int64_t)blockIdx.x * partial_stride + src.off[n];
    naqs::ampbw::pair_dispatch(n, d, w, M, keys, g, out, smem, raw, (int)blockIdx.x, (int)gridDim.x);
}

// aggregate_phase: the amplitude blocks (blockIdx.z == 0, on g[:, 0]) and the per-pair phase blocks (1, raw, on g[:, 1]) in ONE
// launch; the loss gradient's two columns are read in place (stride 2), no split launch in front
// (the two sets' descriptions are separate kernel arguments: nested in a struct their dynamically indexed tables end up in
// scratch memory — measured 2x slower than two launches)
__global__ __launch_bounds__(512) void amp_backward2_kernel(const NetDims d0, const float *__restrict__ w0, float *__restrict__ partial0,
                                                            const AmpSrc src0, const NetDims d1, const float *__restrict__ w1,
                                                            float *__restrict__ partial1, const AmpSrc src1, const int64_t M,
                                                            const uint64_t *__restrict__ keys, const float *__restrict__ g_amp,
                                                            const float *__restrict__ g_ph, const int64_t partial_stride,
                                                            const int g_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = blockIdx.y;
    if (blockIdx.z == 0)
        naqs::ampbw::pair_dispatch(n, d0, w0, M, keys, g_amp, partial0 + (int64_t)blockIdx.x * partial_stride + src0.off[n], smem, 0,
                                   (int)blockIdx.x, (int)gridDim.x, g_stride);
    else
        naqs::ampbw::pair_dispatch(n, d1, w1, M, keys, g_ph, partial1 + (int64_t)blockIdx.x * partial_stride + src1.off[n], smem, 1,
                                   (int)blockIdx.x, (int)gridDim.x, g_stride);
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
    if (d.phase_sym) {                                  // spin-ordered inputs; occ = the row of the 3-output layer (naqs_net.hpp)
        uint32_t a_, b_;
        naqs::key_strings(d, key, a_, b_);
        naqs::phase_order_inputs(d, d.P - 1, a_, b_);
        x[e] = (c < d.P - 1 ? ((a_ >> c) & 1u) : ((b_ >> (c - (d.P - 1))) & 1u)) ? 1.0f : -1.0f;
    } else {
        const int bit = c < d.P - 1 ? d.qa[c] : d.qb[c - (d.P - 1)];
        x[e] = ((key >> bit) & 1ull) ? 1.0f : -1.0f;
    }
    if (c == 0) occ[i] = naqs::phase_out_row(d, (int)((key >> d.qa[d.P - 1]) & 1ull) + 2 * (int)((key >> d.qb[d.P - 1]) & 1ull));
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
    NAQS_KLAUNCH(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, grad_dev,
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
    NAQS_KLAUNCH(phase_inputs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       net->dims, M, keys_dev, x_dev, occ_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

static int vmc_loss_grad_impl(int64_t M, const double *eloc_dev, const double *w_dev, const double *sums_dev, float *g_dev,
                              double *ev_dev, void *stream) {
    if (M < 0 || (M > 0 && (!eloc_dev || !w_dev || !sums_dev || !g_dev))) return NAQS_ERR_INVALID;
    if (M == 0 && ev_dev == nullptr) return NAQS_OK;
    if (!sums_dev) return NAQS_ERR_INVALID;
    NAQS_KLAUNCH(vmc_grad_kernel, dim3((unsigned)std::max<int64_t>(1, (M + 255) / 256)), dim3(256), 0,
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
    NAQS_KLAUNCH(shard_proof_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), M, keys_dev, sums_dev, ext_dev);
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
    st = naqs::net_flush_amp_pack(net, s);                 // (a training step's re-pack still waiting for a launch to host it)
    if (st != NAQS_OK) return st;
    st = naqs::net_amp_forward(net, M, keys_dev, s);
    if (st != NAQS_OK) return st;
    NAQS_KLAUNCH(logamp_sum_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, net->dims.P, M, net->d_scratch,
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
    const int NW = d.Ha >> 4;
    const size_t lds = naqs::ampbw::smem_floats(d) * sizeof(float);
    if (lds > 156 * 1024) return NAQS_ERR_UNSUPPORTED;
    if (!net->grad_attr_set) {
        const int lds_max = 156 * 1024;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&amp_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        net->grad_attr_set = true;
    }
    AmpSrc src;
    for (int n = 0; n < MAXP; ++n) src.off[n] = src_off[n] - src_off[0];         // relative to this set's first parameter
    NAQS_KLAUNCH(amp_backward_kernel, dim3((unsigned)n_wg, (unsigned)d.P), dim3((unsigned)(NW * WAVE)), lds, s, d, w, M, keys_dev,
                       g_dev, gpart, stride, src, raw);
    HIP_TRY(hipGetLastError());
    if (defer) {
        defer->count = n_block_params; defer->stride = stride; defer->n_partials = n_wg; defer->partial = gpart;
        return NAQS_OK;
    }
    NAQS_KLAUNCH(amp_reduce_kernel, dim3((unsigned)((n_block_params + 255) / 256)), dim3(256), 0, s, n_block_params, n_wg,
                       stride, gpart, grad_dev);
    HIP_TRY(hipGetLastError());
    return NAQS_OK;
}

// what net_blocks_backward would launch, for a caller that runs the workgroups inside a launch of its own
// (naqs_phase_grad.hip: backward_mega_kernel): the partial-sum scratch (allocated on first use), the reduction it needs
// afterwards and the pairs' offsets
int naqs::net_blocks_backward_plan(naqs_net *net, const NetDims &d, const int64_t *src_off, int64_t n_block_params, int64_t M, int slot,
                                   BlockReduceJob *job, naqs::ampbw::AmpSrc *src) {
    if (d.Ha > 128 || (d.Ha & 15) || M <= 0 || !job || !src) return NAQS_ERR_INVALID;
    const int n_wg = (int)std::min<int64_t>(MAX_TILE_WGS, (M + GT - 1) / GT);
    const int64_t stride = (std::max(net->amp_params, net->ph_params) + 3) & ~3ll;
    if (!net->d_gpart) {
        HIP_TRY(hipMalloc((void **)&net->d_gpart, 2 * (size_t)MAX_TILE_WGS * stride * sizeof(float)));
    }
    job->count = n_block_params; job->stride = stride; job->n_partials = n_wg;
    job->partial = net->d_gpart + (slot ? (size_t)MAX_TILE_WGS * stride : 0);
    for (int n = 0; n < MAXP; ++n) src->off[n] = src_off[n] - src_off[0];
    return NAQS_OK;
}

// both block sets of an aggregate-phase network in one launch (same hidden width and pair count); the reductions are left
// to the caller like net_blocks_backward's with `defer`
int naqs::net_blocks_backward2(naqs_net *net, int64_t M, const uint64_t *keys_dev, const float *g_amp, const float *g_ph, int g_stride,
                               BlockReduceJob jobs[2], hipStream_t s) {
    const NetDims &d0 = net->dims, &d1 = net->dph;
    if (d0.Ha != d1.Ha || d0.P != d1.P || M <= 0) return NAQS_ERR_INVALID;
    AmpSrc src0, src1;
    int st = net_blocks_backward_plan(net, d0, net->amp_src_off, net->amp_params, M, 0, &jobs[0], &src0);
    if (st != NAQS_OK) return st;
    st = net_blocks_backward_plan(net, d1, net->ph_src_off, net->ph_params, M, 1, &jobs[1], &src1);
    if (st != NAQS_OK) return st;
    const size_t lds = std::max(naqs::ampbw::smem_floats(d0), naqs::ampbw::smem_floats(d1)) * sizeof(float);
    if (lds > 156 * 1024) return NAQS_ERR_UNSUPPORTED;
    if (!net->grad2_attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&amp_backward2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        net->grad2_attr_set = true;
    }
    NAQS_KLAUNCH(amp_backward2_kernel, dim3((unsigned)jobs[0].n_partials, (unsigned)d0.P, 2), dim3((unsigned)((d0.Ha >> 4) * WAVE)), lds, s,
                       d0, net->d_w, const_cast<float *>(jobs[0].partial), src0, d1, net->d_wph, const_cast<float *>(jobs[1].partial), src1, M,
                       keys_dev, g_amp, g_ph, jobs[0].stride, g_stride);
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
    st = naqs::net_flush_amp_pack(net, reinterpret_cast<hipStream_t>(stream));
    if (st != NAQS_OK) return st;
    return naqs::net_blocks_backward(net, net->dims, net->d_w, net->amp_src_off, net->amp_params, M, keys_dev, g_dev, grad_dev, 0,
                                     reinterpret_cast<hipStream_t>(stream));
}
