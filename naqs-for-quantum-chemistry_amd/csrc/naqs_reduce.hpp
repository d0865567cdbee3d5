// naqs_reduce.hpp — the weighted sums (sum w Re E, sum w Im E, sum w Re(E)^2, sum w) of a table, in ONE fixed order, for every
// kernel that forms them: reduce_kernel (naqs_hip.hip, 1024 threads) and the training step's seed kernel (naqs_phase_grad.hip,
// whose first workgroup of 256 threads forms them instead of a launch of its own).  Same per-lane strides, same wave
// reduction, same final adds: bit-identical results whatever the block size.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace naqs {

// sum over the 64 lanes, every lane gets the result: four DPP exchanges inside the rows of 16 lanes (xor 1, xor 2, mirror in
// 8, mirror in 16) and three scalar adds of the row sums.  Fixed order, no LDS traffic (__shfl_xor on a double is two
// ds_bpermute per step: 24 LDS round trips for a (re, im) pair).
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

constexpr int RED_BLOCK = 1024;      // the order is that of 1024 threads striding the table
// called by all NT threads of a workgroup (NT = 1024, 512 or 256: a thread plays RED_BLOCK / NT of the 1024, wave w the
// waves w, w + NT / 64, ...); the four sums are in s[.][0 .. 15] -> out[0 .. 3] by threads 0 .. 3 after the barrier
template <int NT>
__device__ __forceinline__ void weighted_sums_block(const int64_t n, const double *__restrict__ w, const double2 *__restrict__ e,
                                                    double (*s)[RED_BLOCK / 64], double *out) {
    constexpr int Q = RED_BLOCK / NT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        double a = 0, b = 0, c = 0, d = 0;
        for (int64_t i = threadIdx.x + q * NT; i < n; i += RED_BLOCK) {
            const double wi = w[i];
            const double2 ei = e[i];
            a += wi * ei.x; b += wi * ei.y; c += wi * ei.x * ei.x; d += wi;
        }
        a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); d = wave_sum(d);
        if (lane == 0) { s[0][wv + q * (NT / 64)] = a; s[1][wv + q * (NT / 64)] = b; s[2][wv + q * (NT / 64)] = c; s[3][wv + q * (NT / 64)] = d; }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0;
        for (int k = 0; k < RED_BLOCK / 64; ++k) t += s[threadIdx.x][k];
        out[threadIdx.x] = t;
    }
}

}  // namespace naqs
