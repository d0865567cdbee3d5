/*
 * naqs_oracle.c — CPU restatement of the reference's local-energy path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library, and only as the checker /
 * the timed CPU baseline.  The product path (libnaqs_hip.so) never links or calls it
 * and fails loudly when the HIP library is missing.
 *
 * Parity is PINNED: every function here is checked in tests/test_oracle.py against
 * golden vectors dumped from the reference itself (tests/golden/make_golden.py imports
 * tomdbar/naqs-for-quantum-chemistry in the build container): popcount_parity and
 * get_Hij_cy bit-exactly, sparse_dense_mv / calculate_local_energy to <= 1e-12.
 *
 * Each function cites the reference lines it restates (paths relative to the
 * reference repository root).  Plain C99 + OpenMP, f64 arithmetic, like the reachable
 * reference bodies (`__inner_int64_double`, `__sparse_dense_par_mv_64bitElem_32bitIdx`).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

ORACLE_API int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORACLE_API void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * popcount_parity — src_cpp/hamiltonian_math.pyx:295-484.
 *   out[i,j] = 1 - 2*(popcount(arr[i,j]) % 2)  as int8; rows in parallel (prange).
 * The reference dispatches on dtype (8 typed bodies); a narrow signed int is promoted to C int
 * before __builtin_popcount, so a negative value gains an even number of sign bits and the
 * parity is unchanged.  The three signed widths the idx-dtype policy can produce
 * (src/utils/hilbert.py:405-410) are restated with the same promotion.
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_popcount_parity_i64(const int64_t *arr, int8_t *out, int64_t rows, int64_t cols) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; ++i)
        for (int64_t j = 0; j < cols; ++j)
            out[i * cols + j] = (int8_t)(1 - 2 * (__builtin_popcountll((unsigned long long)arr[i * cols + j]) % 2));
}

ORACLE_API void oracle_popcount_parity_i32(const int32_t *arr, int8_t *out, int64_t rows, int64_t cols) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; ++i)
        for (int64_t j = 0; j < cols; ++j)
            out[i * cols + j] = (int8_t)(1 - 2 * (__builtin_popcount((unsigned int)arr[i * cols + j]) % 2));
}

ORACLE_API void oracle_popcount_parity_i16(const int16_t *arr, int8_t *out, int64_t rows, int64_t cols) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; ++i)
        for (int64_t j = 0; j < cols; ++j)
            out[i * cols + j] = (int8_t)(1 - 2 * (__builtin_popcount((unsigned int)(int)arr[i * cols + j]) % 2));
}

/* ------------------------------------------------------------------------------------------
 * get_Hij_cy — src_cpp/hamiltonian_math.pyx:198-288, reachable body __inner_int64_double :85-100.
 *   H_ij[i*Kxy + g(k)] += P[i, y(k)] * c[k]   for i < M (OpenMP static), k < K (serial, ascending)
 * P is the int8 parity table (the reference up-casts it to int64 first; the values are +-1).
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_get_hij(int64_t M, int64_t Kxy, int64_t K, int64_t Kyz,
                               const int64_t *unique2all_xy, const int8_t *P,
                               const int64_t *unique2all_yz, const double *coeff, double *H_ij) {
    memset(H_ij, 0, (size_t)(M * Kxy) * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        const int64_t base = i * Kxy;
        for (int64_t k = 0; k < K; ++k)
            H_ij[base + unique2all_xy[k]] += (double)P[i * Kyz + unique2all_yz[k]] * coeff[k];
    }
}

/* ------------------------------------------------------------------------------------------
 * sparse_dense_mv — src_cpp/sparse_math.pyx:47-83, body :85-100 (f64 data, int32 indices,
 * complex128 vector).  out[r] = sum_p data[p] * v[indices[p]], rows in parallel.
 * v / out are interleaved (re, im).
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_csr_mv(int64_t rows, const double *data, const int32_t *indices,
                              const int32_t *indptr, const double *v, double *out) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < rows; ++r) {
        double re = 0.0, im = 0.0;
        for (int32_t p = indptr[r]; p < indptr[r + 1]; ++p) {
            re += data[p] * v[2 * indices[p]];
            im += data[p] * v[2 * indices[p] + 1];
        }
        out[2 * r] = re;
        out[2 * r + 1] = im;
    }
}

/* position of `key` in the ascending array keys[M], or -1 */
static inline int64_t find_sorted(const uint64_t *keys, int64_t M, uint64_t key) {
    int64_t lo = 0, hi = M;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < M && keys[lo] == key) ? lo : -1;
}

/* complex division a / b, the way numpy does it for complex128 (Smith's algorithm) */
static inline void cdiv(double ar, double ai, double br, double bi, double *qr, double *qi) {
    if (fabs(br) >= fabs(bi)) {
        const double rat = bi / br, scl = 1.0 / (br + bi * rat);
        *qr = (ar + ai * rat) * scl;
        *qi = (ai - ar * rat) * scl;
    } else {
        const double rat = br / bi, scl = 1.0 / (bi + br * rat);
        *qr = (ar * rat + ai) * scl;
        *qi = (ai * rat - ar) * scl;
    }
}

/* ------------------------------------------------------------------------------------------
 * calculate_local_energy, staged like the reference (the CPU baseline that bench.py times):
 *   src/optimizer/energy.py:219-263 -> src/optimizer/hamiltonian.py:272-370 (update_H, cold cache)
 *   -> hamiltonian.py:93-111 (get_H slice) -> sparse_math.pyx:85-100 (SpMV) -> / psi -> conj.
 *
 *   stage 1  P_bits = key[:,None] & uYZ[None,:]; P = popcount_parity(P_bits)      (:301-305)
 *   stage 2  j_full = key[:,None] ^ uXY[None,:]                                   (:313)
 *            physical mask; the reference reads a 2^N look-up table (:321-328), here the
 *            equivalent particle-number test popc(j & alpha) == n_alpha && popc(j & beta) == n_beta
 *   stage 3  H_ij = get_Hij_cy(...)                                               (:335)
 *   stage 4  CSR restricted to the sampled columns: the reference builds the CSR over the whole
 *            restricted space (:350) and slices H[r[:,None], r] (:93-94); un-sampled columns are
 *            dropped either way (energy.py:247-248), so we index columns by position in `keys`
 *   stage 5  E_loc = conj( (H_sub psi) / psi )                                    (energy.py:248)
 *
 * keys[M] ascending & unique (what the sampler hands over, wavefunction.py:488-521 with
 * qubit_ordering=-1); psi[M][2] (re, im) f64 — the float32 psi of the reference up-cast, exactly
 * as sparse_math.pyx:33-37 does; eloc[M][2].
 * Returns 0, or -1 on allocation failure.
 * ---------------------------------------------------------------------------------------- */
ORACLE_API int oracle_eloc_staged(int n_qubits, int n_alpha, int n_beta,
                                  int64_t K, int64_t Kxy, int64_t Kyz,
                                  const uint64_t *unique_xy, const int64_t *unique2all_xy,
                                  const uint64_t *unique_yz, const int64_t *unique2all_yz,
                                  const double *coeff,
                                  int64_t M, const uint64_t *keys, const double *psi, double *eloc) {
    uint64_t amask = 0, bmask = 0;
    for (int q = 0; q < n_qubits; ++q) { if (q & 1) bmask |= 1ull << q; else amask |= 1ull << q; }

    int8_t *P = (int8_t *)malloc((size_t)(M * Kyz));
    double *H_ij = (double *)malloc((size_t)(M * Kxy) * sizeof(double));
    int32_t *col = (int32_t *)malloc((size_t)(M * Kxy) * sizeof(int32_t));
    int32_t *indptr = (int32_t *)malloc((size_t)(M + 1) * sizeof(int32_t));
    if (!P || !H_ij || !col || !indptr) { free(P); free(H_ij); free(col); free(indptr); return -1; }

    /* stage 1 */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i)
        for (int64_t y = 0; y < Kyz; ++y)
            P[i * Kyz + y] = (int8_t)(1 - 2 * (__builtin_popcountll(keys[i] & unique_yz[y]) % 2));

    /* stage 2: connected states, physicality, column index among the samples */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i)
        for (int64_t g = 0; g < Kxy; ++g) {
            const uint64_t j = keys[i] ^ unique_xy[g];
            int32_t c = -1;
            if (__builtin_popcountll(j & amask) == n_alpha && __builtin_popcountll(j & bmask) == n_beta)
                c = (int32_t)find_sorted(keys, M, j);
            col[i * Kxy + g] = c;
        }

    /* stage 3 */
    oracle_get_hij(M, Kxy, K, Kyz, unique2all_xy, P, unique2all_yz, coeff, H_ij);

    /* stage 4: compact to CSR (explicit zeros kept, like the reference) */
    indptr[0] = 0;
    for (int64_t i = 0; i < M; ++i) {
        int32_t n = 0;
        for (int64_t g = 0; g < Kxy; ++g) n += (col[i * Kxy + g] >= 0);
        indptr[i + 1] = indptr[i] + n;
    }
    const int64_t nnz = indptr[M];
    double *data = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
    int32_t *indices = (int32_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t));
    double *hv = (double *)malloc((size_t)(2 * M > 0 ? 2 * M : 1) * sizeof(double));
    if (!data || !indices || !hv) { free(P); free(H_ij); free(col); free(indptr); free(data); free(indices); free(hv); return -1; }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        int32_t p = indptr[i];
        for (int64_t g = 0; g < Kxy; ++g)
            if (col[i * Kxy + g] >= 0) { data[p] = H_ij[i * Kxy + g]; indices[p] = col[i * Kxy + g]; ++p; }
    }

    /* stage 5 */
    oracle_csr_mv(M, data, indices, indptr, psi, hv);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        double qr, qi;
        cdiv(hv[2 * i], hv[2 * i + 1], psi[2 * i], psi[2 * i + 1], &qr, &qi);
        eloc[2 * i] = qr;
        eloc[2 * i + 1] = -qi;
    }
    free(P); free(H_ij); free(col); free(indptr); free(data); free(indices); free(hv);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * The same quantity as one formula (verified against the reference):
 *   E_loc[i] = conj( sum_g [key_i ^ xy_g in samples] ( sum_{k in g} c_k (-1)^{popc(key_i & yz_k)} )
 *                    * psi[key_i ^ xy_g] / psi[i] )
 * with the terms given group-packed (CSR by unique xy): xy_g[Kxy], row_ptr[Kxy+1], yz_t[K], c_t[K].
 * Evaluates rows [row_begin, row_begin + n_rows) against the whole (keys, psi) table — the
 * multi-GPU shard shape.  Serial, f64; used by the tests as an independent second opinion.
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_eloc_matrix_free(int64_t Kxy, const uint64_t *xy_g, const int32_t *row_ptr,
                                        const uint64_t *yz_t, const double *c_t,
                                        int64_t M, const uint64_t *keys, const double *psi,
                                        int64_t row_begin, int64_t n_rows, double *eloc) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t i = row_begin + r;
        double sr = 0.0, si = 0.0;
        for (int64_t g = 0; g < Kxy; ++g) {
            const int64_t j = find_sorted(keys, M, keys[i] ^ xy_g[g]);
            if (j < 0) continue;
            double h = 0.0;
            for (int32_t t = row_ptr[g]; t < row_ptr[g + 1]; ++t)
                h += (__builtin_popcountll(keys[i] & yz_t[t]) & 1) ? -c_t[t] : c_t[t];
            sr += h * psi[2 * j];
            si += h * psi[2 * j + 1];
        }
        double qr, qi;
        cdiv(sr, si, psi[2 * i], psi[2 * i + 1], &qr, &qi);
        eloc[2 * r] = qr;
        eloc[2 * r + 1] = -qi;
    }
}

/* ------------------------------------------------------------------------------------------
 * Energy statistics of _SGD_step — src/optimizer/energy.py:367-377:
 *   w /= sum w ; E = sum w Re(E_loc) ; Var = sum w (Re(E_loc) - E)^2
 * out = { sum w*Re, sum w*Im, sum w*Re^2, sum w }  (raw sums; the caller normalises)
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_eloc_reduce(int64_t M, const double *w, const double *eloc, double out[4]) {
    double a = 0, b = 0, c = 0, d = 0;
    for (int64_t i = 0; i < M; ++i) {
        a += w[i] * eloc[2 * i];
        b += w[i] * eloc[2 * i + 1];
        c += w[i] * eloc[2 * i] * eloc[2 * i];
        d += w[i];
    }
    out[0] = a; out[1] = b; out[2] = c; out[3] = d;
}
