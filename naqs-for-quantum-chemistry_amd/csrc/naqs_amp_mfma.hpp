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


// two adjacent values -> two f16 planes (hi = f16(x), lo = f16(x - hi)), each packed as (x1 << 16) | x0
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_pair(float x0, float x1, uint32_t &w1, uint32_t &w2) {
    // hi: one packed conversion.  The residuals x - hi as v_fma_mix_f32 (hi read as the f16 half it is, times -1, plus x:
    // exact): one instruction per value instead of a conversion back to f32 and a subtraction — the write-back of a
    // 512-wide layer is VALU-bound (6 -> 4 instructions per value).  hipcc folds the equivalent C++ (fmaf(float(h), -1, x))
    // back into cvt + sub, hence the asm; its operands are plain VALU results (no MFMA / transcendental hazard applies).
    const f16x2 h = {(_Float16)x0, (_Float16)x1};
    w1 = __builtin_bit_cast(uint32_t, h);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(w1), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(w1), "v"(x1));
    const f16x2 l = {(_Float16)r0, (_Float16)r1};
    w2 = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ void split2(float x, ushort_t &h1, ushort_t &h2) {
    const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
    h1 = __builtin_bit_cast(ushort_t, h);
    h2 = __builtin_bit_cast(ushort_t, l);
}
// 2^e with e clamped to what keeps every product of two scales finite
__device__ __forceinline__ float pow2_clamped(int e) { return __uint_as_float((uint32_t)(127 + min(max(e, -40), 40)) << 23); }
// floor(log2 x) of a non-negative float from its exponent field (zero / subnormal: -127; inf / nan: 128)
__device__ __forceinline__ int exp_of(float x) { return (int)((__float_as_uint(x) >> 23) & 0xFFu) - 127; }

// ------------------------------------------------------------------------------------------------
// amplitude conditionals on the f16 matrix cores, inside the phase kernel's workgroups, the standalone amp_mfma_kernel
// and the tree sampler: one wave evaluates orbital pair n for a tile of 16 samples, in the f16x2 split (two f16 planes
// of the power-of-two-scaled value: three MFMAs per f32 product, see naqs_logpsi.hip) and TRANSPOSED, so that the hidden
// activations never leave the registers:
//   layer 1   H^T [Ha x 16] = W1 [Ha x 32] . X^T [32 x 16]: the weights are the A operand, the +-1 inputs (exact in f16;
//             input 31 == 1 carries b1; 2n <= 30) the B operand -> D[row = hidden unit, col = sample]: a lane holds, for
//             its sample, hidden units 16 ct + 4 kg + r of tile ct.
//   layer 2   O^T [16 x 16] = W2 [16 x Ha] . H^T: exactly what a B operand wants — lane (sample, kg) supplies 8 values of
//             the contraction index per 32-wide chunk; chunk kc takes tiles 2 kc and 2 kc + 1, i.e. slot j of lane-group
//             kg is hidden unit 16 (2 kc + (j >> 2)) + 4 kg + (j & 3), and the W2 fragments are packed in that order.
//   (Round 2 ran layer 1 with the samples as rows: the ReLU output then went through a per-wave LDS scratch — three
//   bf16 planes written, a wave barrier, six 16-byte reads — to become layer 2's A operand.)
// Fragments per pair (pack_amp_mfma_body): W1 planes [2][Ha/16][64 lanes][8], W2 planes [2][Ha/32][64 lanes][8] (f16), then
// 16 floats: b2[0..8), c1 = s_h / s_w1 (layer-1 accumulator -> scaled activation), c2 = 1 / (s_h s_w2).
// ------------------------------------------------------------------------------------------------
__device__ __host__ __forceinline__ size_t amp_mfma_pair_elems(int Ha) { return (size_t)2 * 512 * ((Ha >> 4) + (Ha >> 5)) + 32; }

// registers of one (tile, pair) work item: every global load is issued up front, one item ahead of its use
template <int CT>
struct AmpFrag {
    bf16x8 w1[2][CT];
    bf16x8 w2[2][CT / 2];
    f32x4 b2;                // b2[4 (kg & 1) + r]: the outputs this lane's accumulator rows belong to
    float c1, c2;
};

template <int CT>
__device__ __forceinline__ void amp_mfma_load(const ushort_t *__restrict__ wp, int lane, AmpFrag<CT> &f) {
    constexpr int KC = CT / 2;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            f.w1[p][ct] = *reinterpret_cast<const bf16x8 *>(wp + ((size_t)(p * CT + ct) * 64 + lane) * 8);
    const ushort_t *w2 = wp + (size_t)2 * CT * 512;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
            f.w2[p][kc] = *reinterpret_cast<const bf16x8 *>(w2 + ((size_t)(p * KC + kc) * 64 + lane) * 8);
    const float *cst = reinterpret_cast<const float *>(w2 + (size_t)2 * KC * 512);
    f.b2 = *reinterpret_cast<const f32x4 *>(cst + 4 * ((lane >> 4) & 1));
    f.c1 = cst[8];
    f.c2 = cst[9];
}

// max(x, 0) in one instruction: fmaxf() canonicalises an operand that is not known to be quiet (a matrix-core result)
// with a second v_max first (and folds a median-of-three the same way).  As integers: a float >= +0 is a non-negative
// int32 with the same bits, anything with the sign bit set (negative, -0) is a negative int32 -> signed max with 0.
// (Not inline asm: the compiler does not see an asm's operands when it inserts the wait states between an MFMA and a
// VALU read of its result, and the read then returns the previous contents of the accumulator — measured.)
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

__device__ __forceinline__ f32x4 mfma_f16(const bf16x8 &a, const bf16x8 &b, const f32x4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// one wave, one (tile of 16 samples, pair n): the block's 5 raw outputs -> outs[sample][8].  ab: the occupation strings
// (alpha | beta << 16) of sample lane & 15.  No LDS scratch, no barrier: the caller orders its reads of outs.
template <int CT>
__device__ __forceinline__ void amp_mfma_item(const NetDims &d, const AmpFrag<CT> &f, int n, uint32_t ab,
                                              int lane, float *__restrict__ outs) {
    constexpr int KC = CT / 2;
    const int s = lane & 15, kg = lane >> 4;
    const uint32_t mask = (1u << n) - 1u;
    const uint32_t abits = ab & mask, bbits = (ab >> 16) & mask;
    const bool swap = d.sym && abits > bbits;                                   // nade.py:519-530
    const uint32_t xbits = (swap ? bbits : abits) | ((swap ? abits : bbits) << n);
    // this lane's 8 inputs k = 8 kg .. 8 kg + 7 as four f16 pairs: +-1 for k < 2n (2n is even: a pair is valid or not as a
    // whole), 0 beyond, and input 31 the constant 1 that carries b1 (2n <= 30)
    const uint32_t tb = xbits >> (8 * kg);
    const int nv = min(max(2 * n - 8 * kg, 0), 8) >> 1;
    uint32_t aw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t w = 0xBC00BC00u ^ ((tb << (15 - 2 * j)) & 0x8000u) ^ ((tb << (30 - 2 * j)) & 0x80000000u);
        aw[j] = j < nv ? w : 0u;
    }
    if (kg == 3) aw[3] |= 0x3C000000u;
    bf16x8 x;
    __builtin_memcpy(&x, aw, sizeof(x));
    f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 1; p >= 0; --p)                                                // smallest plane first; CT independent chains
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = mfma_f16(f.w1[p][ct], x, acc[ct]);
    f32x4 v[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {                                           // one accumulation chain per K chunk
        uint32_t hh[4], hl[4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 &a = acc[2 * kc + t];
            split2_pair(relu1(a[0]) * f.c1, relu1(a[1]) * f.c1, hh[2 * t], hl[2 * t]);
            split2_pair(relu1(a[2]) * f.c1, relu1(a[3]) * f.c1, hh[2 * t + 1], hl[2 * t + 1]);
        }
        bf16x8 bh, bl;
        __builtin_memcpy(&bh, hh, sizeof(bh));
        __builtin_memcpy(&bl, hl, sizeof(bl));
        f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
        c = mfma_f16(f.w2[1][kc], bh, c);
        c = mfma_f16(f.w2[0][kc], bl, c);
        c = mfma_f16(f.w2[0][kc], bh, c);
        v[kc] = c;
    }
    if (kg < 2) {                                                               // D: row = 4 kg + r (output), col = s (sample)
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = v[0][r];
#pragma unroll
            for (int kc = 1; kc < KC; ++kc) t += v[kc][r];
            o[r] = fmaf(t, f.c2, f.b2[r]);
        }
        *reinterpret_cast<f32x4 *>(outs + s * 8 + 4 * kg) = o;
    }
}

}  // namespace naqs
