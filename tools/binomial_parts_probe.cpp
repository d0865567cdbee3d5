// Cycle cost of the pieces of naqs::binomial_group<4> on one wave: developer aid.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build/probe/parts tools/binomial_parts_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../naqs-for-quantum-chemistry_amd/csrc/naqs_rng.hpp"

#define TIME(idx, ...)                                   \
    do {                                                 \
        __builtin_amdgcn_s_barrier();                    \
        const long long t0 = clock64();                  \
        __VA_ARGS__;                                     \
        const long long t1 = clock64();                  \
        if (threadIdx.x == 0) cyc[idx] += t1 - t0;       \
    } while (0)

__global__ void probe(const double *nn, const double *pp, long long *cyc, double *sink, int reps) {
    const int lane = threadIdx.x, base = lane & ~3, j = lane & 3;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        const double n = nn[lane >> 2], p = pp[lane >> 2];
        double u = 0, v = 0;
        TIME(0, { naqs::RngStream g{11u, 22u, (uint32_t)(lane * 977 + r), (uint32_t)r, 0u, (uint32_t)j}; g.pair(u, v); });
        naqs::Btrs t;
        TIME(1, { naqs::btrs_setup(t, n, p); acc += t.vr; });
        double us = 0, kk = 0; int cls = 0;
        TIME(2, {
            double uu = u - 0.5;
            us = 0.5 - fabs(uu);
            kk = floor((2.0 * t.a * naqs::rcp_fast(us) + t.b) * uu + t.c);
            cls = (us >= 0.07 && v <= t.vr) ? 0 : ((kk < 0.0 || kk > t.n) ? 2 : 1);
            acc += cls;
        });
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, x1 = 0;
        TIME(3, {
            const double us_j = __shfl(us, base, 64), v_j = __shfl(v, base, 64), k_j = __shfl(kk, base, 64);
            const double alpha = (2.83 + 5.1 * t.rb) * t.spq;
            const double rr = t.p * naqs::rcp_fast(1.0 - t.p);
            x1 = (k_j - t.m) * naqs::rcp_fast(t.n - k_j + 1.0);
            a0 = v_j * alpha * naqs::rcp_fast(t.a * naqs::rcp_fast(us_j * us_j) + t.b);
            a1 = 1.0 + x1;
            a2 = rr * (t.n - k_j + 1.0) * naqs::rcp_fast(k_j + 1.0);
            a3 = (t.m + 1.0) * naqs::rcp_fast(rr * (t.n - t.m + 1.0));
            kk = k_j;
        });
        double mine = 0;
        TIME(4, { mine = log(j == 0 ? a0 : (j == 1 ? a1 : (j == 2 ? a2 : a3))); });
        double T[4];
        TIME(5, { for (int q = 0; q < 4; ++q) T[q] = __shfl(mine, base + q, 64); });
        TIME(6, {
            const double l1p = a1 == 1.0 ? x1 : T[1] * x1 * naqs::rcp_fast(a1 - 1.0);
            const double h_m = (t.m + 0.5) * T[3] + naqs::stirling_tail(t.m) + naqs::stirling_tail(t.n - t.m);
            const double ub = h_m + (t.n + 1.0) * l1p + (kk + 0.5) * T[2] - naqs::stirling_tail(kk) - naqs::stirling_tail(t.n - kk);
            acc += T[0] <= ub ? 1.0 : 0.0;
        });
        TIME(7, { acc += (double)naqs::binomial_group<4>(true, (int64_t)n, p, 11u, 22u, (uint32_t)((lane >> 2) * 977 + r), (uint32_t)r); });
        TIME(8, { naqs::RngStream g{11u, 22u, (uint32_t)(lane * 977 + r), (uint32_t)r, 0u, 0u}; acc += naqs::binomial_inversion(40.0, 0.05, g); });
        TIME(9, { acc += exp(n * 1e-12 * log1p(-p)); });
    }
    sink[lane] = acc;
}

int main() {
    double hn[16], hp[16];
    for (int i = 0; i < 16; ++i) { hn[i] = 1e6 + 1000.0 * i; hp[i] = 0.05 + 0.025 * i; }
    double *dn, *dp, *ds; long long *dc;
    hipMalloc(&dn, sizeof hn); hipMalloc(&dp, sizeof hp); hipMalloc(&ds, 64 * 8); hipMalloc(&dc, 16 * 8);
    hipMemcpy(dn, hn, sizeof hn, hipMemcpyHostToDevice); hipMemcpy(dp, hp, sizeof hp, hipMemcpyHostToDevice);
    hipMemset(dc, 0, 16 * 8);
    const int reps = 200;
    probe<<<1, 64>>>(dn, dp, dc, ds, reps);
    long long c[16]; hipMemcpy(c, dc, sizeof c, hipMemcpyDeviceToHost);
    const char *names[] = {"philox pair", "btrs_setup", "attempt + classify", "exact: shuffles + arguments", "exact: log", "exact: exchange logs",
                           "exact: assemble (4 stirling tails)", "binomial_group<4> whole", "binomial_inversion n=40 p=.05", "exp(n log1p(-p))"};
    for (int i = 0; i < 10; ++i) std::printf("%-36s %6lld cycles\n", names[i], c[i] / reps);
    return 0;
}
