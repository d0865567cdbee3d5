// Probe of v_mfma_f32_16x16x32_f16 on gfx950 for the f16x2 split (developer aid; build: hipcc --offload-arch=gfx950 -O2).
//  1. are f16-subnormal A/B inputs honoured (not flushed)?
//  2. how exactly are the 32 products of one instruction summed (vs an exact sum rounded once)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const _Float16 *A, const _Float16 *B, const float *C, float *D) {
    // A [16][32] row-major, B [32][16] (k-major): lane (m = lane & 15, kg = lane >> 4) holds k = 8 kg .. 8 kg + 7
    const int lane = threadIdx.x, m = lane & 15, kg = lane >> 4;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[m * 32 + 8 * kg + j]; b[j] = B[(8 * kg + j) * 16 + m]; }
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(kg * 4 + r) * 16 + m];
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(kg * 4 + r) * 16 + m] = c[r];
}

int main() {
    std::vector<_Float16> A(16 * 32), B(32 * 16);
    std::vector<float> C(256), D(256);
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    auto run = [&]() {
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    };
    // 1. subnormal inputs
    for (auto &x : A) x = 0; for (auto &x : B) x = 0; for (auto &x : C) x = 0;
    A[0] = (_Float16)std::ldexp(1.0f, -20);            // subnormal f16
    B[0] = (_Float16)1024.0f;
    A[1 * 32 + 3] = (_Float16)3.0f; B[3 * 16 + 1] = (_Float16)std::ldexp(1.0f, -24);   // smallest subnormal as B
    run();
    std::printf("subnormal A: D[0][0] = %g (want %g)   subnormal B: D[1][1] = %g (want %g)\n", D[0], std::ldexp(1.0f, -10),
                D[1 * 16 + 1], 3.0 * std::ldexp(1.0, -24));
    // 2. summation: random products, many trials; compare with the exactly rounded sum and with sequential f32 adds
    std::srand(1);
    double worst = 0, mean = 0, worst_seq = 0; int cnt = 0;
    for (int trial = 0; trial < 200; ++trial) {
        for (auto &x : A) x = (_Float16)((std::rand() / (float)RAND_MAX - 0.5f) * 2000.0f);
        for (auto &x : B) x = (_Float16)((std::rand() / (float)RAND_MAX - 0.5f) * 2000.0f);
        for (auto &x : C) x = (std::rand() / (float)RAND_MAX - 0.5f) * 1e7f;
        run();
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double ex = C[i * 16 + j]; float seq = C[i * 16 + j]; double mag = std::fabs((double)C[i * 16 + j]);
            for (int k = 0; k < 32; ++k) { const double p = (double)(float)A[i * 32 + k] * (double)(float)B[k * 16 + j]; ex += p; seq += (float)p; mag += std::fabs(p); }
            const double ulp = std::ldexp(1.0, -24) * mag;
            const double e = std::fabs((double)D[i * 16 + j] - ex) / ulp, es = std::fabs((double)seq - ex) / ulp;
            worst = std::fmax(worst, e); worst_seq = std::fmax(worst_seq, es); mean += e; ++cnt;
        }
    }
    std::printf("one MFMA vs exact sum, in units of 2^-24 * sum|terms|: max %.3f mean %.4f   (32 sequential f32 adds: max %.3f)\n", worst, mean / cnt, worst_seq);
    return 0;
}
