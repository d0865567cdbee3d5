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
// waves w, w + NT / 64, ...); the four sums are in s[.][0 .. 15] -> out[0 .. 3] by threads 0 .. 3 after the barrier.
// Each of the 1024 (virtual) threads adds its elements i, i + 1024, ... in ascending order — but the LOADS of eight
// consecutive elements (over the Q virtual threads a real one plays) are issued together before any is used: the plain
// grid-stride loop compiled to one dependent load round trip per element (ten of them for 10^4 rows: 4 of reduce_kernel's
// 5.4 us, and 8 round trips at the head of the training step's backward pass, whose other workgroups wait for these sums).
template <int NT>
__device__ __forceinline__ void weighted_sums_block(const int64_t n, const double *__restrict__ w, const double2 *__restrict__ e,
                                                    double (*s)[RED_BLOCK / 64], double *out) {
    constexpr int Q = RED_BLOCK / NT;
    constexpr int U = 8 / Q;             // elements per virtual thread and batch: Q * U = 8 (w, e) pairs in flight
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double a[Q], b[Q], c[Q], d[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) a[q] = b[q] = c[q] = d[q] = 0.0;
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += (int64_t)RED_BLOCK * U) {
        double wi[Q][U];
        double2 ei[Q][U];
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + q * NT + (int64_t)u * RED_BLOCK;
                if (i < n) { wi[q][u] = w[i]; ei[q][u] = e[i]; }
            }
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + q * NT + (int64_t)u * RED_BLOCK;
                if (i < n) {
                    a[q] += wi[q][u] * ei[q][u].x; b[q] += wi[q][u] * ei[q][u].y;
                    c[q] += wi[q][u] * ei[q][u].x * ei[q][u].x; d[q] += wi[q][u];
                }
            }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const double ra = wave_sum(a[q]), rb = wave_sum(b[q]), rc = wave_sum(c[q]), rd = wave_sum(d[q]);
        if (lane == 0) { s[0][wv + q * (NT / 64)] = ra; s[1][wv + q * (NT / 64)] = rb; s[2][wv + q * (NT / 64)] = rc; s[3][wv + q * (NT / 64)] = rd; }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0;
        for (int k = 0; k < RED_BLOCK / 64; ++k) t += s[threadIdx.x][k];
        out[threadIdx.x] = t;
    }
}

}  // namespace naqs
