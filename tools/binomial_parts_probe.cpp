// Cycle cost of the pieces of naqs::binomial_group<4> on one wave: developer aid.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build/probe/parts tools/binomial_parts_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../naqs-for-quantum-chemistry_amd/csrc/naqs_rng.hpp"

// The stamps only mean something if the work cannot move across them: every timed piece takes its inputs through KEEP (an
// empty volatile asm that "modifies" the value) right after the first stamp and hands its outputs to KEEP right before the
// second — volatile asms and the clock reads keep their order.
#define KEEP(x) asm volatile("" : "+v"(x))
#define TIME(idx, ...)                                   \
    do {                                                 \
        const long long t0 = clock64();                  \
        __VA_ARGS__;                                     \
        const long long t1 = clock64();                  \
        if (threadIdx.x == 0) cyc[idx] += t1 - t0;       \
    } while (0)

__global__ void probe(const double *nn, const double *pp, long long *cyc, double *sink, int reps) {
    const int lane = threadIdx.x, j = lane & 3;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        double n = nn[lane >> 2], p = pp[lane >> 2];
        double u = 0, v = 0;
        uint32_t c0 = (uint32_t)(lane * 977 + r);
        TIME(0, { KEEP(c0); uint32_t ctr[4] = {c0, (uint32_t)r, 0u, (uint32_t)j}; uint32_t o[4]; naqs::philox4x32_10(ctr, 11u, 22u, o);
                  KEEP(o[0]); KEEP(o[1]); KEEP(o[2]); KEEP(o[3]); u = naqs::u01(o[0], o[1]); v = naqs::u01(o[2], o[3]); });
        uint32_t w0 = c0 ^ 5u, w1 = c0 ^ 7u, w2 = c0 ^ 9u, w3 = c0 ^ 11u;
        TIME(1, { KEEP(w0); KEEP(w1); KEEP(w2); KEEP(w3); u = naqs::u01(w0, w1); v = naqs::u01(w2, w3); KEEP(u); KEEP(v); });
        naqs::Btrs t;
        TIME(2, { KEEP(n); KEEP(p); naqs::btrs_setup(t, n, p); KEEP(t.spq); KEEP(t.b); KEEP(t.a); KEEP(t.c); KEEP(t.rb); KEEP(t.vr); KEEP(t.m); });
        double us = 0, kk = 0; int cls = 0;
        TIME(3, {
            KEEP(u); KEEP(v);
            double uu = u - 0.5;
            us = 0.5 - fabs(uu);
            kk = floor((2.0 * t.a * naqs::rcp_fast(us) + t.b) * uu + t.c);
            cls = (us >= 0.07 && v <= t.vr) ? 0 : ((kk < 0.0 || kk > t.n) ? 2 : 1);
            KEEP(cls); KEEP(kk); KEEP(us);
        });
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, x1 = 0;
        TIME(4, {
            KEEP(us); KEEP(v); KEEP(kk);
            const double us_j = naqs::group_bcast<4>(us, 0), v_j = naqs::group_bcast<4>(v, 0), k_j = naqs::group_bcast<4>(kk, 0);
            const double alpha = (2.83 + 5.1 * t.rb) * t.spq;
            const double rr = t.p * naqs::rcp_fast(1.0 - t.p);
            x1 = (k_j - t.m) * naqs::rcp_fast(t.n - k_j + 1.0);
            a0 = v_j * alpha * naqs::rcp_fast(t.a * naqs::rcp_fast(us_j * us_j) + t.b);
            a1 = 1.0 + x1;
            a2 = rr * (t.n - k_j + 1.0) * naqs::rcp_fast(k_j + 1.0);
            a3 = (t.m + 1.0) * naqs::rcp_fast(rr * (t.n - t.m + 1.0));
            kk = k_j;
            KEEP(a0); KEEP(a1); KEEP(a2); KEEP(a3); KEEP(x1); KEEP(kk);
        });
        double mine = 0;
        TIME(5, { KEEP(a0); KEEP(a1); KEEP(a2); KEEP(a3); mine = log(j == 0 ? a0 : (j == 1 ? a1 : (j == 2 ? a2 : a3))); KEEP(mine); });
        double T[4];
        TIME(6, { KEEP(mine); for (int q = 0; q < 4; ++q) { T[q] = naqs::group_bcast<4>(mine, q); KEEP(T[q]); } });
        double st4 = 0;
        TIME(7, { KEEP(kk); KEEP(t.m); KEEP(t.n);
                  st4 = naqs::stirling_tail(t.m) + naqs::stirling_tail(t.n - t.m) - naqs::stirling_tail(kk) - naqs::stirling_tail(t.n - kk); KEEP(st4); });
        double dec = 0;
        TIME(8, {
            KEEP(T[0]); KEEP(T[1]); KEEP(T[2]); KEEP(T[3]); KEEP(st4); KEEP(x1); KEEP(a1);
            const double l1p = a1 == 1.0 ? x1 : T[1] * x1 * naqs::rcp_fast(a1 - 1.0);
            const double ub = (t.m + 0.5) * T[3] + st4 + (t.n + 1.0) * l1p + (kk + 0.5) * T[2];
            dec = T[0] <= ub ? 1.0 : 0.0;
            KEEP(dec);
        });
        acc += dec + cls;
        double whole = 0;
        int64_t ni = (int64_t)n;
        TIME(9, { KEEP(p); whole = (double)naqs::binomial_group<4>(true, ni, p, 11u, 22u, (uint32_t)((lane >> 2) * 977 + r), (uint32_t)r); KEEP(whole); });
        acc += whole;
        double inv = 0;
        TIME(10, { KEEP(p); naqs::RngStream g{11u, 22u, (uint32_t)(lane * 977 + r), (uint32_t)r, 0u, 0u}; inv = naqs::binomial_inversion(40.0, p * 0.5, g); KEEP(inv); });
        acc += inv;
        double f0 = 0;
        TIME(11, { KEEP(n); KEEP(p); f0 = exp(n * 1e-5 * log1p(-p)); KEEP(f0); });
        acc += f0;
        float lf = (float)a0, lg = 0;
        TIME(12, { KEEP(lf); lg = __logf(lf); KEEP(lg); });
        acc += lg;
        double sq = 0;
        TIME(13, { KEEP(n); sq = sqrt(n); KEEP(sq); });
        acc += sq;
        double rc = 0;
        TIME(14, { KEEP(n); rc = naqs::rcp_fast(n); KEEP(rc); });
        acc += rc;
        int64_t ki = 0;
        TIME(15, { KEEP(whole); ki = (int64_t)whole; ki = ki < 0 ? 0 : (ki > ni ? ni : ki); double kd = (double)ki; KEEP(kd); acc += kd; });
    }
    sink[lane] = acc;
}

int main() {
    double hn[16], hp[16];
    for (int i = 0; i < 16; ++i) { hn[i] = 1e6 + 1000.0 * i; hp[i] = 0.05 + 0.025 * i; }
    double *dn, *dp, *ds; long long *dc;
    hipMalloc(&dn, sizeof hn); hipMalloc(&dp, sizeof hp); hipMalloc(&ds, 64 * 8); hipMalloc(&dc, 32 * 8);
    hipMemcpy(dn, hn, sizeof hn, hipMemcpyHostToDevice); hipMemcpy(dp, hp, sizeof hp, hipMemcpyHostToDevice);
    hipMemset(dc, 0, 32 * 8);
    const int reps = 200;
    probe<<<1, 64>>>(dn, dp, dc, ds, reps);
    long long c[32]; hipMemcpy(c, dc, sizeof c, hipMemcpyDeviceToHost);
    const char *names[] = {"philox4x32-10 + two u01", "two u01 alone", "btrs_setup", "attempt + classify", "exact: broadcasts + arguments (6 rcp)",
                           "exact: log (float64)", "exact: exchange logs (8 DPP)", "exact: 4 stirling tails", "exact: assemble", "binomial_group<4> whole",
                           "binomial_inversion n=40", "exp(n log1p(-p))", "__logf (float32)", "sqrt (float64)", "rcp_fast", "to int64 + clamp + back"};
    for (int i = 0; i < 16; ++i) std::printf("%-40s %6lld cycles\n", names[i], c[i] / reps);
    return 0;
}
