// naqs_rng.hpp — counter-based random numbers and an exact binomial generator for the tree sampler.
//
// The reference draws the children counts of a prefix with numpy's binomial on the host
// (src/naqs/network/nade.py:20-37, conditional-binomial chain).  Here every draw is a pure function of
// (seed, prefix bits, block, outcome, attempt): Philox4x32-10 (Salmon et al., SC'11) keyed by the seed, so a
// sample is reproducible whatever the launch geometry, and the same code runs on the host for the statistical
// tests (naqs_rng_*_host in the C ABI).  Binomial(n, p), n up to 2^44: sequential inversion when n*min(p,q) < 10,
// otherwise Hörmann's transformed rejection with squeeze (BTRS, "The generation of binomial random variates",
// J. Stat. Comput. Simul. 46 (1993)) — exact, ~1.15 uniform pairs per variate.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define NAQS_HD __host__ __device__ __forceinline__
#else
#define NAQS_HD inline
#endif

namespace naqs {

NAQS_HD void mulhilo32(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
    const uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}

NAQS_HD void philox4x32_10(const uint32_t (&ctr)[4], uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo32(0xD2511F53u, c0, hi0, lo0);
        mulhilo32(0xCD9E8D57u, c2, hi1, lo1);
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform on the open interval (0, 1) with 53 random bits
NAQS_HD double u01(uint32_t hi, uint32_t lo) {
    const uint64_t bits = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);      // 27 + 26 bits
    return ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
}

// one stream of uniform pairs: counter = (c0, c1, c2, attempt)
struct RngStream {
    uint32_t k0, k1, c0, c1, c2, attempt;
    NAQS_HD void pair(double &u, double &v) {
        const uint32_t ctr[4] = {c0, c1, c2, attempt++};
        uint32_t r[4];
        philox4x32_10(ctr, k0, k1, r);
        u = u01(r[0], r[1]);
        v = u01(r[2], r[3]);
    }
};

// 1 / x for the generator's inner loops.  On the device a float64 division is a ~100-cycle dependent chain
// (div_scale, rcp, five fmas, div_fmas, div_fixup) and these loops are latency-bound — one wave's draws are the critical
// path of a whole tree level — so the reciprocal comes from v_rcp_f64 refined by two Newton steps (full double
// accuracy for the well-scaled arguments that occur here: k + 1 >= 1, 1 - p >= 1/2, b >= 1.15, ...).
// The host build (naqs_rng_binomial_host: known-answer and chi-square tests of the generator's DISTRIBUTION) divides
// exactly, so host and device run the same algorithm but not bit-identical arithmetic: a last-ulp difference can flip a
// floor() or an accept test, i.e. individual variates may differ between the two.  Bit-exact comparisons of draws are
// therefore device-vs-device only (tests/test_sampler_gpu.py: launch fusions, head kernel); host/device agreement is
// statistical.
NAQS_HD double rcp_fast(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

// log and exp for the generator (round 5).  The library functions are ~510 (log), ~900 (log1p) and ~800 (exp) cycles of one wave —
// full-range, correctly rounded to an ulp, with their special cases — and they sit on the chain that a tree level waits for
// (tools/binomial_parts_probe.cpp: the exact acceptance test was 1.9 k of a 4.0 k-cycle BTRS call, exp(n log1p(-p)) 1.75 k of a
// 3.5 k-cycle inversion call).  The arguments here are finite, normal and positive (log) / in [-745, 0] (exp), which leaves
// one range reduction and one polynomial each; errors are a few ulp, i.e. ~1e-15 relative in an acceptance bound that the
// float64 generator meets with a margin of 2^-9 at the largest supported n (2^44).  Host and device run the same arithmetic
// except for the reciprocal (rcp_fast).
// log(x), x > 0 finite and normal: x = 2^e m, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716
// (No guard for x <= 0, on purpose — it would sit on every draw's critical chain: the smallest argument this generator ever forms
// is the exact test's v alpha us^2 / (a + b us^2) with v, us >= 2^-54 (u01 below never returns 0: its smallest value is
// (0 + 1/2) 2^-53), alpha = O(1) and a + b us^2 <= a + b / 4 = O(sqrt(n)) <= 2^23, i.e. >= 2^-190: normal.  The host test
// test_generator_log_has_no_zero_argument holds u01's bounds.)
NAQS_HD double log_fast(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double m = __builtin_amdgcn_frexp_mant(x);                  // [1/2, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
#else
    int e;
    double m = std::frexp(x, &e);
#endif
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    const double f = m - 1.0;                                   // exact
    const double s = f * rcp_fast(2.0 + f);
    const double z = s * s;                                     // <= 0.02944: z^11 / 23 < 1e-18
    double q = 1.0 / 23.0;
    q = fma(q, z, 1.0 / 21.0); q = fma(q, z, 1.0 / 19.0); q = fma(q, z, 1.0 / 17.0); q = fma(q, z, 1.0 / 15.0);
    q = fma(q, z, 1.0 / 13.0); q = fma(q, z, 1.0 / 11.0); q = fma(q, z, 1.0 / 9.0); q = fma(q, z, 1.0 / 7.0);
    q = fma(q, z, 1.0 / 5.0); q = fma(q, z, 1.0 / 3.0);
    const double s2 = 2.0 * s;
    const double l = fma(s2 * z, q, s2);                        // 2 s (1 + z / 3 + z^2 / 5 + ...)
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, l));      // e ln2 (hi: 32 trailing zero bits, lo)
}
// log(1 - p), 0 < p <= 1/2, from p itself (1 - p alone has lost p's low bits): Kahan's log(w) (-p) / (w - 1), w = fl(1 - p)
NAQS_HD double log1m_fast(double p) {
    const double w = 1.0 - p;
    return w == 1.0 ? -p : log_fast(w) * (-p) * rcp_fast(w - 1.0);
}
// exp(x), -745 < x <= 0: x = k ln2 + r, |r| <= 0.3466, Taylor to r^13 (remainder < 5e-18)
NAQS_HD double exp_fast(double x) {
    const double kd = rint(x * 1.44269504088896338700);
    const double r = fma(-kd, 1.90821492927058770002e-10, fma(-kd, 6.93147180369123816490e-01, x));
    double q = 1.0 / 6227020800.0;
    q = fma(q, r, 1.0 / 479001600.0); q = fma(q, r, 1.0 / 39916800.0); q = fma(q, r, 1.0 / 3628800.0); q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0); q = fma(q, r, 1.0 / 5040.0); q = fma(q, r, 1.0 / 720.0); q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0); q = fma(q, r, 1.0 / 6.0); q = fma(q, r, 0.5); q = fma(q, r, 1.0); q = fma(q, r, 1.0);
    return ldexp(q, (int)kd);
}

// log(k!) - [ (k + 1/2) log(k + 1) - (k + 1) + log(2 pi)/2 ]  (Stirling series remainder)
NAQS_HD double stirling_tail(double k) {
    if (k < 10.0) {
        switch ((int)k) {
            case 0: return 0.08106146679532726;
            case 1: return 0.04134069595540929;
            case 2: return 0.02767792568499834;
            case 3: return 0.02079067210376509;
            case 4: return 0.01664469118982119;
            case 5: return 0.01387612882307075;
            case 6: return 0.01189670994589177;
            case 7: return 0.01041126526197209;
            case 8: return 0.009255462182712733;
            default: return 0.008330563433362871;
        }
    }
    const double r = rcp_fast(k + 1.0), r2 = r * r;      // (1/12 - (1/360 - 1/1260 / s) / s) / (k + 1), s = (k + 1)^2
    return (1.0 / 12.0 - (1.0 / 360.0 - (1.0 / 1260.0) * r2) * r2) * r;
}

// Binomial(n, p), 0 < p <= 1/2, n p < 10: invert the CDF upwards from 0
NAQS_HD double binomial_inversion(double n, double p, RngStream &g) {
    const double s = p * rcp_fast(1.0 - p);
    double u, v;
    g.pair(u, v);
    double f = exp_fast(n * log1m_fast(p)); // P(0) = q^n >= e^-15 here
    double k = 0.0;
    for (int it = 0; it < 400 && u > f && k < n; ++it) {
        u -= f;
        k += 1.0;
        f *= s * (n - k + 1.0) * rcp_fast(k);
    }
    return k;
}

// Binomial(n, p), 0 < p <= 1/2, n p >= 10 (BTRS), split into set-up and ONE attempt so that the sequential host/
// single-lane form below and the sampler's quad-parallel form (several attempts of one draw evaluated by neighbouring
// lanes, the first accepted one in attempt order wins: naqs_sample.hip) run the same arithmetic.
struct Btrs {
    double n, p, spq, b, a, c, rb, vr, m;
    // constants of the exact acceptance test: only needed when a proposal falls outside the squeeze (~1 in 7), so they
    // are formed on first use — a log, two Stirling tails and a division that most draws never pay for
    bool have_slow;
    double alpha, r, h_m;
};

NAQS_HD void btrs_setup(Btrs &t, double n, double p) {
    const double q = 1.0 - p;
    t.n = n; t.p = p;
    t.spq = sqrt(n * p * q);
    t.b = 1.15 + 2.53 * t.spq;
    t.a = -0.0873 + 0.0248 * t.b + 0.01 * p;
    t.c = n * p + 0.5;
    t.rb = rcp_fast(t.b);
    t.vr = 0.92 - 4.2 * t.rb;
    t.m = floor((n + 1.0) * p);
    t.have_slow = false;
    t.alpha = t.r = t.h_m = 0.0;
}

// one proposal from the uniform pair (u, v): true and the variate in k when it is accepted
NAQS_HD bool btrs_attempt(Btrs &t, double u, double v, double &k) {
    u -= 0.5;
    const double us = 0.5 - fabs(u);
    k = floor((2.0 * t.a * rcp_fast(us) + t.b) * u + t.c);
    if (us >= 0.07 && v <= t.vr) return true;                     // inside the squeeze: accept immediately
    if (k < 0.0 || k > t.n) return false;
    if (!t.have_slow) {
        t.have_slow = true;
        t.alpha = (2.83 + 5.1 * t.rb) * t.spq;
        t.r = t.p * rcp_fast(1.0 - t.p);
        t.h_m = (t.m + 0.5) * log_fast((t.m + 1.0) * rcp_fast(t.r * (t.n - t.m + 1.0))) + stirling_tail(t.m) + stirling_tail(t.n - t.m);
    }
    // (log1p as log(1 + x) x / ((1 + x) - 1), the form the group draw below evaluates)
    const double x1 = (k - t.m) * rcp_fast(t.n - k + 1.0), a1 = 1.0 + x1;
    const double l1p = a1 == 1.0 ? x1 : log_fast(a1) * x1 * rcp_fast(a1 - 1.0);
    const double lv = log_fast(v * t.alpha * rcp_fast(t.a * rcp_fast(us * us) + t.b));
    const double ub = t.h_m + (t.n + 1.0) * l1p +
                      (k + 0.5) * log_fast(t.r * (t.n - k + 1.0) * rcp_fast(k + 1.0)) - stirling_tail(k) - stirling_tail(t.n - k);
    return lv <= ub;
}

constexpr int BTRS_MAX_ATTEMPTS = 1000;                           // unreachable in practice (acceptance ~0.87 per pair)

NAQS_HD double binomial_btrs(double n, double p, RngStream &g) {
    Btrs t;
    btrs_setup(t, n, p);
    for (int it = 0; it < BTRS_MAX_ATTEMPTS; ++it) {
        double u, v, k;
        g.pair(u, v);
        if (btrs_attempt(t, u, v, k)) return k;
    }
    return t.m;
}

NAQS_HD int64_t binomial(int64_t n, double p, RngStream &g) {
    if (n <= 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    const bool flip = p > 0.5;
    const double pp = flip ? 1.0 - p : p;
    const double nd = (double)n;
    const double k = nd * pp < 10.0 ? binomial_inversion(nd, pp, g) : binomial_btrs(nd, pp, g);
    int64_t ki = (int64_t)k;
    ki = ki < 0 ? 0 : (ki > n ? n : ki);
    return flip ? n - ki : ki;
}

#if defined(__HIPCC__)
// Binomial(n, p) drawn by a group of G consecutive lanes (G = 4: a quad, G = 2: half of one).  The lanes of a group pass
// the same arguments and all return the same variate — the one binomial() returns for the stream (k0, k1, c0, c1).
// What the group buys is latency: a tree level lasts as long as its slowest wave, and a wave's BTRS rejection loop as
// long as its unluckiest lane (~3 rounds of 16-32 concurrent draws, each round through the log-heavy exact test).
// Here lane j of the group evaluates attempt round * G + j of the SAME draw and the first accepted attempt in attempt
// order wins, so a draw needs a second round with probability 0.13^G instead of 0.13.  Must be called by all 64 lanes
// (wave-uniform control flow: it votes and shuffles); lanes with nothing to draw pass need = false.
// Round 3: the exact acceptance test is shared out too.  A level's time is its slowest wave's two dependent binomial calls,
// and every call ran the exact test (three float64 logs + the mode's, one after the other, on whichever lanes fell outside
// the squeeze: practically always somebody among a wave's 16-32 draws).  Now every lane only CLASSIFIES its attempt (squeeze
// accept / out of range / needs the exact test), the group resolves its attempts in order, and an attempt that needs the
// exact test gets it from the whole group: its four logarithms — log v-term, log1p (as log(1 + x) x / ((1 + x) - 1): one
// uniform log call for every lane), the k-term and the mode's — are evaluated by different lanes at once (a quad: one log
// call deep; a pair: two) and exchanged, and the test is assembled with the arithmetic of btrs_attempt.  Attempts
// behind an accepted one are never tested.
// Lane jj of every group of G consecutive lanes (G = 4: quads, G = 2: the two pairs of a quad), broadcast to the group: a
// DPP quad_perm move — one VALU instruction per 32 bits, where __shfl goes through the LDS crossbar (ds_bpermute, ~100
// cycles of latency on a chain with nothing else to overlap it; the generator has ~30 of them per draw).
template <int CTRL>
__device__ __forceinline__ int dpp_quad(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int G>
__device__ __forceinline__ int group_bcast(int v, int jj) {
    if (G == 4) {
        switch (jj & 3) {                                 // quad_perm [jj, jj, jj, jj]
            case 0: return dpp_quad<0x00>(v);
            case 1: return dpp_quad<0x55>(v);
            case 2: return dpp_quad<0xAA>(v);
            default: return dpp_quad<0xFF>(v);
        }
    }
    return (jj & 1) ? dpp_quad<0xF5>(v) : dpp_quad<0xA0>(v);   // quad_perm [1, 1, 3, 3] / [0, 0, 2, 2]
}
template <int G>
__device__ __forceinline__ double group_bcast(double v, int jj) {
    return __hiloint2double(group_bcast<G>(__double2hiint(v), jj), group_bcast<G>(__double2loint(v), jj));
}
template <int G>
__device__ __forceinline__ int64_t group_bcast(int64_t v, int jj) {
    const int lo = group_bcast<G>((int)(uint32_t)v, jj), hi = group_bcast<G>((int)(uint32_t)((uint64_t)v >> 32), jj);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

#if defined(NAQS_SAMPLE_STATS)
// developer build (tools/sample_regime_stats.py): per binomial_group call of a wave, which regimes its lanes were in (0 none,
// 1 inversion only, 2 BTRS only, 3 both — the two loops then run one after the other), how many BTRS rounds and exact
// acceptance tests it went through and how many steps its longest inversion search took.  Counts only: cycle stamps around
// these regions measure the counters' own atomics (whichever region first waits for outstanding memory operations pays for
// them) — cycles per regime come from tools/binomial_probe.cpp.
extern __device__ unsigned long long g_sample_stats[2][4][4];      // [G == 4 ? 0 : 1][class][calls, rounds, exact tests, steps of the longest inversion]
#endif

template <int G>
__device__ __forceinline__ int64_t binomial_group(bool need, const int64_t n, const double p, const uint32_t k0,
                                                  const uint32_t k1, const uint32_t c0, const uint32_t c1) {
    static_assert(G == 2 || G == 4, "a pair or a quad");
    const int lane = threadIdx.x & 63, j = lane & (G - 1);
    int64_t fixed = 0;
    if (need && (n <= 0 || !(p > 0.0))) need = false;
    if (need && p >= 1.0) { fixed = n; need = false; }
    const bool flip = p > 0.5;
    const double pp = flip ? 1.0 - p : p, nd = (double)n;
    const bool inv = need && nd * pp < 10.0, bt = need && !inv;
#if defined(NAQS_SAMPLE_STATS)
    const int stat_cls = (__ballot(inv) != 0ull ? 1 : 0) | (__ballot(bt) != 0ull ? 2 : 0);
    unsigned long long stat_rounds = 0, stat_exact = 0;
#endif
    double k = 0.0;
    if (inv) {                                                  // short sequential search: every lane of the group runs it
        RngStream g{k0, k1, c0, c1, 0u, 0u};
        k = binomial_inversion(nd, pp, g);
    }
#if defined(NAQS_SAMPLE_STATS)
    int stat_steps = inv ? (int)k : 0;
    for (int o = 32; o > 0; o >>= 1) stat_steps = max(stat_steps, __shfl_xor(stat_steps, o, 64));
#endif
    Btrs t;
    if (bt) btrs_setup(t, nd, pp);
    bool pending = bt;
    for (int round = 0; round < BTRS_MAX_ATTEMPTS / G && __ballot(pending) != 0ull; ++round) {
#if defined(NAQS_SAMPLE_STATS)
        ++stat_rounds;
#endif
        // my attempt, classified: 0 accepted by the squeeze, 1 needs the exact test, 2 rejected (out of range) / nothing
        int cls = 2;
        double us = 0.5, v = 0.0, kk = 0.0;
        if (pending) {
            RngStream g{k0, k1, c0, c1, 0u, (uint32_t)(round * G + j)};
            double u;
            g.pair(u, v);
            u -= 0.5;
            us = 0.5 - fabs(u);
            kk = floor((2.0 * t.a * rcp_fast(us) + t.b) * u + t.c);
            cls = (us >= 0.07 && v <= t.vr) ? 0 : ((kk < 0.0 || kk > t.n) ? 2 : 1);
        }
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {                        // the group's attempts, in attempt order
            int cls_j = group_bcast<G>(cls, jj);
            const bool exact = pending && cls_j == 1;           // uniform within a group
            if (__ballot(exact) != 0ull) {
#if defined(NAQS_SAMPLE_STATS)
                ++stat_exact;
#endif
                if (exact) {
                    const double us_j = group_bcast<G>(us, jj), v_j = group_bcast<G>(v, jj), k_j = group_bcast<G>(kk, jj);
                    // The test's four logarithms and four Stirling remainders, shared out over the group's lanes.  Every lane
                    // forms ITS term as one quotient — one reciprocal, where forming all four arguments on every lane took six:
                    //   0: log( v alpha us^2 / (a + b us^2) )            (= log(v alpha / (a / us^2 + b)))
                    //   1: log1p(x), x = (k - m) / (n - k + 1), as log(1 + x) x / ((1 + x) - 1)
                    //   2: log( p (n - k + 1) / ((1 - p)(k + 1)) )       (= log(r (n - k + 1) / (k + 1)), r = p / (1 - p))
                    //   3: log( (m + 1)(1 - p) / (p (n - m + 1)) )       (= log((m + 1) / (r (n - m + 1))))
                    const double q1 = 1.0 - t.p, us2 = us_j * us_j;
                    const double num[4] = {v_j * ((2.83 + 5.1 * t.rb) * t.spq) * us2, k_j - t.m, t.p * (t.n - k_j + 1.0), (t.m + 1.0) * q1};
                    const double den[4] = {fma(t.b, us2, t.a), t.n - k_j + 1.0, q1 * (k_j + 1.0), t.p * (t.n - t.m + 1.0)};
                    const double b[4] = {k_j, t.n - k_j, t.m, t.n - t.m};                  // arguments of the Stirling remainders
                    double T[4], S[4];
                    constexpr int R = 4 / G;                                                // terms per lane
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int term = G == 4 ? j : 2 * r + j;                            // (a pair: lane j takes terms j and 2 + j)
                        const double nu = term == 0 ? num[0] : (term == 1 ? num[1] : (term == 2 ? num[2] : num[3]));
                        const double de = term == 0 ? den[0] : (term == 1 ? den[1] : (term == 2 ? den[2] : den[3]));
                        const double bb = term == 0 ? b[0] : (term == 1 ? b[1] : (term == 2 ? b[2] : b[3]));
                        const double qt = nu * rcp_fast(de);
                        const bool is_l1p = term == 1;
                        const double arg = is_l1p ? 1.0 + qt : qt;
                        const double lg = log_fast(arg);
                        const double am1 = arg - 1.0;                                       // (exact near 1; only used by term 1)
                        const double kahan = lg * qt * rcp_fast(is_l1p && am1 != 0.0 ? am1 : 1.0);
                        const double mine = is_l1p ? (am1 == 0.0 ? qt : kahan) : lg;
                        const double tail = stirling_tail(bb);
                        if (G == 4) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { T[q] = group_bcast<4>(mine, q); S[q] = group_bcast<4>(tail, q); }
                        } else {
                            T[2 * r] = group_bcast<2>(mine, 0); T[2 * r + 1] = group_bcast<2>(mine, 1);
                            S[2 * r] = group_bcast<2>(tail, 0); S[2 * r + 1] = group_bcast<2>(tail, 1);
                        }
                    }
                    const double h_m = (t.m + 0.5) * T[3] + S[2] + S[3];
                    const double ub = h_m + (t.n + 1.0) * T[1] + (k_j + 0.5) * T[2] - S[0] - S[1];
                    cls_j = T[0] <= ub ? 0 : 2;
                }
            }
            if (pending && cls_j == 0) {
                k = group_bcast<G>(kk, jj);
                pending = false;
            }
        }
    }
    if (pending) k = t.m;                                       // unreachable in practice
#if defined(NAQS_SAMPLE_STATS)
    if (lane == 0) {
        unsigned long long *st = g_sample_stats[G == 4 ? 0 : 1][stat_cls];
        atomicAdd(&st[0], 1ull);
        atomicAdd(&st[1], stat_rounds);
        atomicAdd(&st[2], stat_exact);
        atomicAdd(&st[3], (unsigned long long)stat_steps);
    }
#endif
    if (!need) return fixed;
    int64_t ki = (int64_t)k;
    ki = ki < 0 ? 0 : (ki > n ? n : ki);
    return flip ? n - ki : ki;
}
#endif  // __HIPCC__

}  // namespace naqs
