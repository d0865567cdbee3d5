// Cycle cost of naqs::binomial_group (one wave) by regime: developer aid.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../naqs-for-quantum-chemistry_amd/csrc/naqs_rng.hpp"

template <int G>
__global__ void probe(const int64_t *n, const double *p, long long *cyc, int64_t *out, int reps) {
    const int lane = threadIdx.x;
    const int g = lane / G;
    long long tot = 0;
    int64_t acc = 0;
    for (int r = 0; r < reps; ++r) {
        __builtin_amdgcn_s_barrier();
        const long long t0 = clock64();
        const int64_t k = naqs::binomial_group<G>(true, n[g], p[g], 11u, 22u, (uint32_t)(g * 977 + r), (uint32_t)r);
        const long long t1 = clock64();
        tot += t1 - t0;
        acc += k;
    }
    if (lane == 0) cyc[0] = tot / reps;
    out[lane] = acc;
}

template <int G>
static void run(const char *name, std::vector<int64_t> n, std::vector<double> p) {
    int64_t *dn, *dout; double *dp; long long *dc;
    hipMalloc(&dn, 64 * 8); hipMalloc(&dp, 64 * 8); hipMalloc(&dc, 8); hipMalloc(&dout, 64 * 8);
    n.resize(64 / G, n.back()); p.resize(64 / G, p.back());
    hipMemcpy(dn, n.data(), n.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dp, p.data(), p.size() * 8, hipMemcpyHostToDevice);
    probe<G><<<1, 64>>>(dn, dp, dc, dout, 200);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    std::printf("G=%d %-34s %6lld cycles per call\n", G, name, c);
}

int main() {
    std::vector<int64_t> big(16, 1000000000000ll), mid(16, 5000), small(16, 40), mixed(16);
    std::vector<double> half(16, 0.37), tiny(16, 0.05);
    for (int i = 0; i < 16; ++i) mixed[i] = (i & 1) ? 1000000000ll : 60;
    run<4>("BTRS, n = 1e12, p = 0.37", big, half);
    run<4>("BTRS, n = 5000, p = 0.37", mid, half);
    run<4>("inversion, n = 40, p = 0.05", small, tiny);
    std::vector<int64_t> near10(16, 190), mixed10(16);
    for (int i = 0; i < 16; ++i) mixed10[i] = (i & 1) ? 1000000000ll : 190;
    run<4>("inversion, n = 190, p = 0.05 (np 9.5)", near10, tiny);
    run<4>("mixed BTRS / inversion np 9.5", mixed10, tiny);
    run<2>("inversion, n = 190, p = 0.05 (np 9.5)", near10, tiny);
    std::vector<int64_t> above10(16, 240), above30(16, 600);
    run<4>("BTRS, n = 240, p = 0.05 (np 12)", above10, tiny);
    run<4>("BTRS, n = 600, p = 0.05 (np 30)", above30, tiny);
    run<2>("BTRS, n = 240, p = 0.05 (np 12)", above10, tiny);
    {   // every quad its own (n, p): inversion draws with n p = 0.6 .. 9.6 beside BTRS draws with n p = 10.5 .. 300
        std::vector<int64_t> nn(16); std::vector<double> pp(16);
        for (int i = 0; i < 16; ++i) {
            if (i & 1) { nn[i] = 350ll * (1ll << (i / 3)); pp[i] = 0.03 + 0.02 * (i % 5); }
            else { nn[i] = 20 * (i + 1); pp[i] = 0.03; }
        }
        run<4>("diverse mixed (np 0.6..9.6 | 10..300)", nn, pp);
        run<2>("diverse mixed (np 0.6..9.6 | 10..300)", nn, pp);
        for (int i = 0; i < 16; i += 2) { nn[i] = nn[i + 1]; pp[i] = pp[i + 1]; }
        run<4>("diverse BTRS only (np 10..300)", nn, pp);
        run<2>("diverse BTRS only (np 10..300)", nn, pp);
    }
    run<4>("mixed BTRS / inversion", mixed, tiny);
    run<2>("BTRS, n = 1e12, p = 0.37", big, half);
    run<2>("inversion, n = 40, p = 0.05", small, tiny);
    run<2>("mixed BTRS / inversion", mixed, tiny);
    return 0;
}
