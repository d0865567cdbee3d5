// Amplitude conditionals on the bf16 matrix cores: one wave evaluates orbital pair n for 16 samples (shared by the
// log-psi kernel's prologue, the standalone amp_mfma_kernel and the tree sampler).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "naqs_net.hpp"

namespace naqs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short ushort_t;

// the same split for two values at once, each plane already packed as (hi << 16) | lo: v_perm_b32 picks the two upper
// halves in one instruction (the write-back is VALU-bound: two waves per SIMD, 48 values per lane)
__device__ __forceinline__ void split3t_pair(float lo, float hi, uint32_t &w1, uint32_t &w2, uint32_t &w3) {
    constexpr uint32_t SEL = 0x07060302u;                  // bytes {hi.3, hi.2, lo.3, lo.2}
    uint32_t a = __float_as_uint(lo), b = __float_as_uint(hi);
    w1 = __builtin_amdgcn_perm(b, a, SEL);
    const float ra = lo - __uint_as_float(a & 0xFFFF0000u), rb = hi - __uint_as_float(b & 0xFFFF0000u);
    a = __float_as_uint(ra); b = __float_as_uint(rb);
    w2 = __builtin_amdgcn_perm(b, a, SEL);
    const float sa = ra - __uint_as_float(a & 0xFFFF0000u), sb = rb - __uint_as_float(b & 0xFFFF0000u);
    w3 = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), SEL);
}


// ------------------------------------------------------------------------------------------------
// amplitude conditionals on the bf16 matrix cores, inside the phase kernel's workgroups (no amp_kernel launch):
// one wave evaluates orbital pair n for a tile of 16 samples.  Layer 1: the +-1 inputs are exact in bf16, so
// x . (W1_hi + W1_mid + W1_lo) is three 16x16x32 MFMAs per 16 hidden units (2n <= 30 inputs fit one K chunk);
// the ReLU output is split into three planes through the wave's LDS scratch and layer 2 (Ha -> 5) is the usual
// six-term bf16x3 product.  Weight fragments are pre-tiled per pair (pack_amp_mfma_kernel):
//   W1 planes [3][Ha/16][64 lanes][8], W2 planes [3][Ha/32][64 lanes][8]   (bf16)
// ------------------------------------------------------------------------------------------------
// per pair: the fragments, then b2 as 16 floats
__device__ __host__ __forceinline__ size_t amp_mfma_pair_elems(int Ha) { return (size_t)3 * 512 * ((Ha >> 4) + (Ha >> 5)) + 32; }

// registers of one (tile, pair) work item: every global load is issued up front, one item ahead of its use
template <int CT>
struct AmpFrag {
    bf16x8 w1[3][CT];
    bf16x8 w2[3][CT / 2];
    float b2;
};

template <int CT>
__device__ __forceinline__ void amp_mfma_load(const ushort_t *__restrict__ wp, int lane, AmpFrag<CT> &f) {
    constexpr int KC = CT / 2;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            f.w1[p][ct] = *reinterpret_cast<const bf16x8 *>(wp + ((size_t)(p * CT + ct) * 64 + lane) * 8);
    const ushort_t *w2 = wp + (size_t)3 * CT * 512;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
            f.w2[p][kc] = *reinterpret_cast<const bf16x8 *>(w2 + ((size_t)(p * KC + kc) * 64 + lane) * 8);
    f.b2 = reinterpret_cast<const float *>(w2 + (size_t)3 * KC * 512)[lane & 15];
}

// max(x, 0) in one instruction: fmaxf() canonicalises an operand that is not known to be quiet (a matrix-core result)
// with a second v_max first (and folds a median-of-three the same way).  As integers: a float >= +0 is a non-negative
// int32 with the same bits, anything with the sign bit set (negative, -0) is a negative int32 -> signed max with 0.
// (Not inline asm: the compiler does not see an asm's operands when it inserts the wait states between an MFMA and a
// VALU read of its result, and the read then returns the previous contents of the accumulator — measured.)
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// one wave, one (tile of 16 samples, pair n): the block's 5 raw outputs -> outs[sample][8]
template <int CT>
__device__ __forceinline__ void amp_mfma_item(const NetDims &d, const AmpFrag<CT> &f, int n, uint32_t ab,
                                              int lane, ushort_t *__restrict__ hs, float *__restrict__ outs) {
    constexpr int KC = CT / 2, HLD = CT * 16 + 8;
    const int m = lane & 15, kg = lane >> 4;
    const uint32_t mask = (1u << n) - 1u;
    {
        const uint32_t abits = ab & mask, bbits = (ab >> 16) & mask;
        const bool swap = d.sym && abits > bbits;                               // nade.py:519-530
        const uint32_t xbits = (swap ? bbits : abits) | ((swap ? abits : bbits) << n);
        // this lane's 8 inputs k = 8 kg .. 8 kg + 7 as four bf16 pairs: +-1 for k < 2n (2n is even: a pair is valid or not
        // as a whole), 0 beyond, and input 31 the constant 1 that carries b1 (2n <= 30)
        const uint32_t tb = xbits >> (8 * kg);
        const int nv = min(max(2 * n - 8 * kg, 0), 8) >> 1;
        uint32_t aw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = 0xBF80BF80u ^ ((tb << (15 - 2 * j)) & 0x8000u) ^ ((tb << (30 - 2 * j)) & 0x80000000u);
            aw[j] = j < nv ? w : 0u;
        }
        if (kg == 3) aw[3] |= 0x3F800000u;
        bf16x8 ax;
        __builtin_memcpy(&ax, aw, sizeof(ax));
        f32x4 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 2; p >= 0; --p)                                            // smallest plane first; CT independent chains
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, f.w1[p][ct], acc[ct], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {                       // D: row = 4 kg + r (sample); tile ct of lane m = hidden unit CT m + ct
            ushort_t *dst = hs + (4 * kg + r) * HLD + CT * m;
            uint32_t a1, a2, a3;
            split3t_pair(relu1(acc[0][r]), relu1(acc[1][r]), a1, a2, a3);
            if (CT == 4) {
                uint32_t b1, b2, b3;
                split3t_pair(relu1(acc[2 % CT][r]), relu1(acc[3 % CT][r]), b1, b2, b3);
                *reinterpret_cast<uint2 *>(dst) = make_uint2(a1, b1);
                *reinterpret_cast<uint2 *>(dst + 16 * HLD) = make_uint2(a2, b2);
                *reinterpret_cast<uint2 *>(dst + 32 * HLD) = make_uint2(a3, b3);
            } else {
                *reinterpret_cast<uint32_t *>(dst) = a1;
                *reinterpret_cast<uint32_t *>(dst + 16 * HLD) = a2;
                *reinterpret_cast<uint32_t *>(dst + 32 * HLD) = a3;
            }
        }
    }
    // the scratch is private to this wave: LDS operations of a wave complete in order, so a wave-level fence suffices
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        f32x4 v[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {                                       // one accumulation chain per K chunk
            bf16x8 a[3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
                a[p] = *reinterpret_cast<const bf16x8 *>(hs + p * 16 * HLD + m * HLD + kc * 32 + 8 * kg);
            f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], f.w2[1][kc], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], f.w2[0][kc], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], f.w2[2][kc], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], f.w2[0][kc], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], f.w2[1][kc], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], f.w2[0][kc], c, 0, 0, 0);
            v[kc] = c;
        }
        if (m < 8) {                                                            // D: row = 4 kg + r (sample), col = m (output)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o = v[0][r];
#pragma unroll
                for (int kc = 1; kc < KC; ++kc) o += v[kc][r];
                outs[(4 * kg + r) * 8 + m] = o + f.b2;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                            // hs is rewritten by the wave's next item
}

}  // namespace naqs
