// abi_smoke.cpp — drives libnaqs_hip.so through the C ABI only (no Python, no torch):
//   ./abi_smoke <dump.bin>      (dump written by tools/dump_case.py)
// Checks E_loc against the expected values stored in the dump.  Exit code 0 on success.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "naqs_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
#define NQ(x) do { int st__ = (x); if (st__ != NAQS_OK) { std::fprintf(stderr, "%s: %s (%s)\n", #x, naqs_strerror(st__), naqs_last_hip_error_string()); return 3; } } while (0)

template <typename T> static bool rd(FILE *f, std::vector<T> &v, size_t n) { v.resize(n); return std::fread(v.data(), sizeof(T), n, f) == n; }

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s dump.bin\n", argv[0]); return 1; }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 1; }
    int64_t hdr[5];  // n_qubits, n_alpha, n_beta, K, M
    if (std::fread(hdr, sizeof(int64_t), 5, f) != 5) return 1;
    const int64_t K = hdr[3], M = hdr[4];
    std::vector<uint64_t> xy, yz, keys; std::vector<double> c, psi, want;
    if (!rd(f, xy, K) || !rd(f, yz, K) || !rd(f, c, K) || !rd(f, keys, M) || !rd(f, psi, 2 * M) || !rd(f, want, 2 * M)) return 1;
    std::fclose(f);
    std::printf("case: N=%lld K=%lld M=%lld, devices=%d\n", (long long)hdr[0], (long long)K, (long long)M, naqs_device_count());

    naqs_ham_t *h = nullptr;
    NQ(naqs_ham_create((int)hdr[0], (int)hdr[1], (int)hdr[2], K, xy.data(), yz.data(), c.data(), 0, &h));
    int64_t info[8];
    NQ(naqs_ham_info(h, info));
    std::printf("ham: K=%lld Kxy=%lld key_bits=%lld diag_terms=%lld\n", (long long)info[0], (long long)info[1], (long long)info[5], (long long)info[6]);

    uint64_t *d_keys; double *d_psi, *d_e;
    CK(hipMalloc(&d_keys, M * 8)); CK(hipMalloc(&d_psi, M * 16)); CK(hipMalloc(&d_e, M * 16));
    CK(hipMemcpy(d_keys, keys.data(), M * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_psi, psi.data(), M * 16, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    NQ(naqs_eloc(h, M, d_keys, d_psi, NAQS_PSI_F64, 0, M, d_e, s));
    std::printf("launched\n"); std::fflush(stdout);
    CK(hipStreamSynchronize(s));
    std::vector<double> got(2 * M);
    CK(hipMemcpy(got.data(), d_e, M * 16, hipMemcpyDeviceToHost));
    double err = 0;
    for (int64_t i = 0; i < 2 * M; ++i) err = std::fmax(err, std::fabs(got[i] - want[i]) / std::fmax(1.0, std::fabs(want[i])));
    char kname[128] = {0};
    NQ(naqs_ham_last_kernel(h, kname, (int)sizeof(kname)));
    std::printf("max rel err vs expected = %.3e  (%s, ABI %d, sources %s) -> %s\n", err, kname, naqs_abi_version(), naqs_source_hash(),
                err < 1e-10 ? "OK" : "MISMATCH");
    NQ(naqs_ham_destroy(h));
    return err < 1e-10 ? 0 : 4;
}
