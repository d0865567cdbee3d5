/*
 * naqs_hip.h — C ABI of libnaqs_hip.so: the MI355X (gfx950) local-energy path of a
 * neural-autoregressive-quantum-state VMC step.
 *
 * The reference (tomdbar/naqs-for-quantum-chemistry) has no C ABI of its own; its native seam is
 * three Cython extension modules that its Python imports.  Every entry
 * point below names the reference interface it replaces (file:line relative to the reference
 * repository).  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross the boundary;
 *   - every function returns NAQS_OK (0) or a negative naqs_status; nothing throws;
 *   - `*_dev` pointers are device memory on the handle's device and stay owned by the caller;
 *   - all device work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null
 *     stream) and is asynchronous: results are valid after the stream is synchronised;
 *   - a handle is bound to one device, owns its packed term tables and scratch buffers, and must
 *     not be used from two streams/threads at the same time;
 *   - bit-string convention: qubit q <-> bit q of the key; even bits = alpha spin-orbitals, odd
 *     bits = beta (src/utils/hilbert.py:446-449, :573-581).
 */
#ifndef NAQS_HIP_H
#define NAQS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAQS_ABI_VERSION 9

typedef struct naqs_ham naqs_ham_t;

enum naqs_status {
    NAQS_OK = 0,
    NAQS_ERR_INVALID = -1,      /* bad argument (NULL pointer, negative size, range outside table) */
    NAQS_ERR_HIP = -2,          /* a HIP runtime call failed; see naqs_last_hip_error() */
    NAQS_ERR_NOMEM = -3,        /* host or device allocation failed */
    NAQS_ERR_UNSUPPORTED = -4,  /* e.g. n_qubits > 64, unsupported element width */
    NAQS_ERR_NO_DEVICE = -5     /* no usable HIP device / wrong architecture */
};

/* How the wave function of the sampled states is handed over (second index: 2 components). */
enum naqs_psi_kind {
    NAQS_PSI_F32 = 0,     /* float  [M][2] = (Re psi, Im psi): what the reference passes, energy.py:241-243 */
    NAQS_PSI_F64 = 1,     /* double [M][2] = (Re psi, Im psi) */
    NAQS_LOGPSI_F32 = 2,  /* float  [M][2] = (log|psi|, phase): network output, wavefunction.py:167-183 */
    NAQS_LOGPSI_F64 = 3   /* double [M][2] = (log|psi|, phase) */
};

int naqs_abi_version(void);
/* 16 hex digits: SHA-256 prefix of the kernel sources (the .hip and .hpp files under csrc) this library was built from.  No counterpart in
 * the reference; it lets measurement tooling (bench.py, tools/collect_pmc.py) tie replayed hardware-counter files to the
 * exact kernels they were collected on. */
const char *naqs_source_hash(void);
const char *naqs_strerror(int status);
/* hipError_t of the most recent failing HIP call on this thread (0 if none) and its text.  When the failure was a
 * device-side wait that ran out of its budget (naqs_device_check below), the text names the kernel site, the waiting
 * workgroup and the index it waited for. */
int naqs_last_hip_error(void);
const char *naqs_last_hip_error_string(void);
/* Kernels of this library that wait for a word another workgroup publishes (sampler look-back, column-split log-psi
 * tiles, fused sums, re-pack scale chain) give up after NAQS_POLL_BUDGET_MS (default 2000) instead of hanging: the wave
 * records the site in the error word of the network handle that launched the kernel (ABI 8; one word per device before: two
 * runs sharing a GPU could take each other's failures) and leaves without writing results.  Every entry point of a handle
 * looks at the handle's word when it is called; this call looks at the words of ALL live handles of the device on demand
 * (e.g. after a stream synchronisation) and reports the first it finds: NAQS_OK, or NAQS_ERR_HIP with
 * naqs_last_hip_error_string() describing the wait.  The reference has no counterpart (its only failure path in the loop is
 * MaxBatchSizeExceededError, src/naqs/network/nade.py:39-40, 710-712 -> src/optimizer/energy.py:939-946); a hang has to
 * become an error somewhere. */
int naqs_device_check(int device);

/* Number of HIP devices visible to the library (0 when there is none); never fails. */
int naqs_device_count(void);

/*
 * Host-only: group K Pauli terms by their XY (bit-flip) mask — the dedupe of
 * src/optimizer/hamiltonian.py:248-252 (np.unique(XY, return_inverse)) turned into a CSR:
 *   xy_g[Kxy] ascending unique masks, row_ptr[Kxy+1], and per-term yz_t[K], c_t[K], order[K]
 *   (order[t] = original term index; terms keep ascending original order inside a group, so a
 *   sequential sum over a group reproduces the reference's summation order, hamiltonian_math.pyx:95-98).
 * Output arrays are caller-allocated with room for K (row_ptr: K+1) entries.
 */
int naqs_terms_group(int64_t K, const uint64_t *xy, const uint64_t *yz, const double *coeff,
                     int64_t *Kxy_out, uint64_t *xy_g, int32_t *row_ptr,
                     uint64_t *yz_t, double *c_t, int64_t *order);

/*
 * Create a device-resident Pauli Hamiltonian from K pre-processed terms
 *   xy[k]   bit q set iff the Pauli on qubit q is X or Y      (hamiltonian.py:389-390)
 *   yz[k]   bit q set iff the Pauli on qubit q is Y or Z      (hamiltonian.py:391-393, :402-403)
 *   coeff[k] = Re(i^{nY}) * coefficient, real                 (hamiltonian.py:416, :424)
 * host arrays, copied.  Replaces _PauliHamiltonianDynamic.__init__ (hamiltonian.py:241-262); the
 * lazily cached scipy CSR matrix of the reference (hamiltonian.py:86, :350-363) has no counterpart:
 * matrix elements are regenerated on the fly.
 * n_alpha / n_beta: electrons per spin (particle-number filter, replaces the 2^N look-up table of
 * src/utils/hilbert.py:429-434); pass -1/-1 for an unrestricted space.
 */
int naqs_ham_create(int n_qubits, int n_alpha, int n_beta, int64_t K,
                    const uint64_t *xy, const uint64_t *yz, const double *coeff,
                    int device, naqs_ham_t **out);
int naqs_ham_destroy(naqs_ham_t *h);

/* info[0..7] = { K, Kxy, n_qubits, n_alpha, n_beta, key_bits (32|64), diag_terms, device } */
int naqs_ham_info(const naqs_ham_t *h, int64_t info[8]);

/* Pre-size the scratch buffers for up to M sampled states (otherwise grown on demand, which
 * synchronises the device). */
int naqs_ham_reserve(naqs_ham_t *h, int64_t M);

/*
 * Local energies.  Replaces OptimizerBase.calculate_local_energy (src/optimizer/energy.py:219-263)
 * = update_H (hamiltonian.py:272-370: get_Hij_cy + popcount_parity) + get_H (hamiltonian.py:93-111)
 * + sparse_dense_mv (sparse_math.pyx:47-100), fused and matrix-free:
 *
 *   E_loc[i] = conj( sum_j H[i,j] psi[j] / psi[i] ),  j restricted to the M sampled states
 *   H[i, key_i ^ xy_g] = sum_{k in g} coeff_k (-1)^{popcount(key_i & yz_k)}
 *
 * keys_dev[M]   unique sampled bit-strings (any order), all physical
 * psi_dev       per psi_kind, [M][2]
 * rows          E_loc is produced for table rows [row_begin, row_begin + n_rows) only — the shard of
 *               one rank; couplings are still looked up in the whole table
 * eloc_dev      double [n_rows][2] = (Re, Im)
 */
int naqs_eloc(naqs_ham_t *h, int64_t M, const uint64_t *keys_dev, const void *psi_dev, int psi_kind,
              int64_t row_begin, int64_t n_rows, double *eloc_dev, void *stream);

/*
 * naqs_eloc followed by naqs_eloc_reduce over the produced rows, enqueued by one call:
 * w_dev[n_rows] are the weights of the produced rows, out4_dev as below.
 */
int naqs_eloc_reduced(naqs_ham_t *h, int64_t M, const uint64_t *keys_dev, const void *psi_dev, int psi_kind,
                      int64_t row_begin, int64_t n_rows, const double *w_dev, double *eloc_dev,
                      double *out4_dev, void *stream);

/*
 * Weighted sums over n local energies, deterministic order:
 *   out4_dev = { sum w Re(E), sum w Im(E), sum w Re(E)^2, sum w }
 * from which <E> and Var follow as in _SGD_step (src/optimizer/energy.py:367-377).
 */
int naqs_eloc_reduce(naqs_ham_t *h, int64_t n, const double *w_dev, const double *eloc_dev,
                     double *out4_dev, void *stream);

/*
 * Matrix-free product with the Hamiltonian restricted to the sampled states: out_i = sum_j H_ij v_j over the M keys
 * (complex float64 [M][2] in and out; H is real symmetric).  The mat-vec of get_H(idxs) (src/optimizer/hamiltonian.py:93-111)
 * without materialising the sub-matrix — what the sampled-subspace diagonalisation of solve_H (src/optimizer/energy.py:762-786)
 * needs once the sample set is too large for a dense M x Kxy table.  Same kernel and row sharding as naqs_eloc.
 */
int naqs_hmatvec(naqs_ham_t *h, int64_t M, const uint64_t *keys_dev, const double *v_dev, int64_t row_begin,
                 int64_t n_rows, double *out_dev, void *stream);

/* ---- inner ring: device versions of the three Cython entry points the reference imports ---- */

/* src.utils.hamiltonian_math.popcount_parity (hamiltonian_math.pyx:455-484):
 * out[i] = 1 - 2*(popcount(arr[i]) & 1), arr of signed ints with elem_bytes in {1,2,4,8} (the sign extension of a
 * narrow negative value adds an even number of bits, so unsigned arrays of the same width give the same parities). */
int naqs_popcount_parity(const void *arr_dev, int elem_bytes, int64_t n, int8_t *out_dev, void *stream);

/* src.utils.hamiltonian_math.get_Hij_cy (hamiltonian_math.pyx:198-288): dense matrix elements
 * hij_dev[i*Kxy + g] = H[i, key_i ^ xy_g] for M states, row-major, bit-identical to the reference
 * (same summation order). */
int naqs_get_hij(naqs_ham_t *h, int64_t M, const uint64_t *keys_dev, double *hij_dev, void *stream);

/* get_Hij_cy with the reference's own argument list (hamiltonian_math.pyx:198-288, body :85-100): the parity table
 * parity_dev int8 [M][Kyz] (= popcount_parity(key & unique_YZ)), and the K terms GROUPED BY OUTPUT COLUMN
 * (group_ptr_dev int32 [Kxy+1]; inside a column ascending original term index — a stable sort of unique2all_XY):
 * term_yz_dev int32 [K] = unique2all_YZ of the term, term_coeff_dev [K] float64 (coeff_bytes 8) or float32 (4).
 * hij_dev [M*Kxy] of the coupling type, hij[i*Kxy + g] = sum_{k in g} parity[i][yz(k)] * coeff[k], accumulated in
 * ascending k in the coupling type: bit-identical to the reference's loop.  naqs_amd/compat/hamiltonian_math.py. */
int naqs_hij_from_parity(int64_t M, int64_t Kxy, int64_t Kyz, int64_t K, const int8_t *parity_dev,
                         const int32_t *group_ptr_dev, const int32_t *term_yz_dev, const void *term_coeff_dev,
                         int coeff_bytes, void *hij_dev, void *stream);

/* src.utils.sparse_math.sparse_dense_mv (sparse_math.pyx:47-100): CSR (f64 data, int32 indices)
 * times complex128 vector; v_dev/out_dev are [.][2] = (Re, Im). */
int naqs_csr_mv(int64_t rows, const double *data_dev, const int32_t *indices_dev,
                const int32_t *indptr_dev, const double *v_dev, double *out_dev, void *stream);

/* ---- measurement ---- */

/* Record a hipEvent pair around every launch of the main E_loc kernel (up to max_records launches;
 * 0 disables and frees the events). */
int naqs_prof_enable(naqs_ham_t *h, int max_records);
/* Synchronises the recorded events: total milliseconds and number of launches since enable. */
int naqs_prof_read(naqs_ham_t *h, double *total_ms, int64_t *launches);
/* Name (with template arguments) of the local-energy kernel the handle's most recent call launched, e.g.
 * "eloc_kernel2<uint32_t, STAGE=2, NT=1024, BLOOM=0>"; empty before the first call.  Measurement aid (bench.py). */
int naqs_ham_last_kernel(const naqs_ham_t *h, char *buf, int buf_len);
/* Record only every stride-th launch (default 1): an event pair costs a few microseconds of queue time,
 * so a sparse sample keeps the timed region representative. */
int naqs_prof_stride(naqs_ham_t *h, int stride);


/* ================================================================================================
 * Fused log-psi evaluation of the orbital NADE (inference; gradients stay with PyTorch autograd).
 * Replaces wavefunction.log_psi(states) (src/naqs/wavefunction.py:167-183) ->
 * _forward_predict (src/naqs/network/nade.py:738-770) for the published architecture family:
 * one amplitude MLP per orbital pair (one hidden layer), and either a single phase MLP on the last pair
 * (aggregate_phase = False: -single_phase, the published runs) or one single-hidden-layer phase block per pair whose
 * outputs are summed (aggregate_phase = True: the reference's default); SoftmaxLogProbAmps amplitudes, no phase symmetry.
 * ============================================================================================== */
typedef struct naqs_net naqs_net_t;

#define NAQS_NET_MAX_PAIRS 16
#define NAQS_NET_MAX_PHASE_LAYERS 8

typedef struct naqs_net_config {
    int32_t n_qubits;                 /* even, <= 2 * NAQS_NET_MAX_PAIRS */
    int32_t n_alpha, n_beta;          /* electron budget of the masks (nade.py:417-474); -1/-1: unrestricted */
    int32_t masking;                  /* NadeMasking: 0 NONE, 1 PARTIAL, 2 FULL (network/base.py:20-23) */
    int32_t use_amp_spin_sym;         /* 1: 5 amplitude outputs + symmetrisation (nade.py:576-594) */
    int32_t amp_hidden;               /* width of the single hidden layer of every amplitude block */
    int32_t n_phase_hidden;           /* hidden layers of the phase block (>= 1) */
    int32_t phase_hidden[NAQS_NET_MAX_PHASE_LAYERS];   /* their widths */
    int32_t qubit2model[2 * NAQS_NET_MAX_PAIRS];       /* model position -> qubit (wavefunction.py:56-83, :369-383) */
    int32_t aggregate_phase;          /* 0: one phase MLP on the last pair (the published runs, -single_phase);
                                       * 1: one phase block per orbital pair, phases summed — the reference's default
                                       *    (experiments/run.py:31, nade.py:556-569): every block is
                                       *    Linear(max(1, 2n), phase_hidden[0]) + ReLU + Linear(phase_hidden[0], 4); n_phase_hidden must be 1 */
    int32_t use_phase_spin_sym;       /* 1 (-phase_sym, nade.py:281, 507-533, 590-610; ABI 7): the phase block reads spin-ordered inputs
                                       *    (alpha and beta strings of the first P-1 pairs exchanged when idx(alpha) > idx(beta)), has
                                       *    3 outputs for |00>, |01> = |10>, |11>, and the phase gets + pi (N_01 mod 2) where
                                       *    idx(alpha) < idx(beta).  With aggregate_phase every per-pair block orders ITS prefix and
                                       *    the last block's phase carries the shift (nade.py:758-759). */
} naqs_net_config_t;

int naqs_net_create(const naqs_net_config_t *cfg, int device, naqs_net_t **out);
int naqs_net_destroy(naqs_net_t *net);
/* Number of float parameters expected by naqs_net_set_weights: the reference's state_dict order,
 * flattened (amp_layers.0.layers.0.0.weight, .bias, amp_layers.0.layers.1.0.weight, .bias, ...,
 * phase_layers.0.layers.<l>.0.weight, .bias). */
int naqs_net_param_count(const naqs_net_t *net, int64_t *count);
/* Copy/re-pack the parameters (device pointer, float32, state_dict order) into the kernels' layout. */
int naqs_net_set_weights(naqs_net_t *net, const float *flat_dev, int64_t count, void *stream);
/* logpsi_dev: float [M][2] = (log|psi|, phase) for M keys. */
int naqs_net_logpsi(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, float *logpsi_dev, void *stream);
/*
 * The whole hot path of one VMC evaluation in one call: log psi of the M keys (naqs_net_logpsi) and their
 * local energies against the same table (naqs_eloc with NAQS_LOGPSI_F32 over all rows, + naqs_eloc_reduce
 * when w_dev/out4_dev are given).  The amplitude kernel inserts the keys into the E_loc hash table and the
 * phase kernel writes psi in float64, so the E_loc stage needs no preparation kernel of its own.
 * net and ham must live on the same device.  Results are identical to the two separate calls.
 */
int naqs_logpsi_eloc(naqs_net_t *net, naqs_ham_t *ham, int64_t M, const uint64_t *keys_dev,
                     const double *w_dev, float *logpsi_dev, double *eloc_dev, double *out4_dev, void *stream);

/* HIP-event timing of the phase-MLP kernel, like naqs_prof_enable / naqs_prof_read. */
int naqs_net_prof_enable(naqs_net_t *net, int max_records);
int naqs_net_prof_read(naqs_net_t *net, double *total_ms, int64_t *launches);
int naqs_net_prof_stride(naqs_net_t *net, int stride);
/* What the three calls above bracket: 0 (default) the log-psi kernel, 1 the sampler's launches of a training step (first
 * sampler launch .. the launch that writes the weights; naqs_vmc_step / naqs_vmc_run) — bench.py's `train_step.sampler_us`.
 * Selecting disarms the ring (enable it again afterwards). */
int naqs_net_prof_select(naqs_net_t *net, int which);
/* Kernel launches this library has issued in this process so far (all handles, all streams): bench.py's `launches per step`
 * is the difference over a timed region divided by its steps.  No counterpart in the reference. */
int64_t naqs_launch_count(void);
/* naqs_vmc_step / naqs_vmc_run launch the training forward behind the sampler's launches, BEFORE the host knows the number of
 * unique samples M (the kernel reads M on the device; the launch covers the last accepted draw's M plus an eighth in the kernel
 * form that M gets), and launch it again the ordinary way when the real M does not fit that launch or gets another form — the
 * same kernel on the same rows either way.  counts[0] = forwards launched ahead, counts[1] = of those, the ones that stood.
 * NAQS_SPEC_FORWARD=0: never ahead.  No counterpart in the reference (its loop is synchronous: energy.py:975-998). */
int naqs_net_spec_counts(const naqs_net_t *net, int64_t counts[2]);
/* Two (or more) handles driven from different threads / streams of ONE GPU — the farm's `experiments.run --farm --per-gpu 2`, no
 * counterpart in the reference, whose runs are one process each (experiments/bash/naqs/batch_train.sh:11-15).  on = 1: this
 * handle's sampler calls take turns with those of the device's other sharing handles: the sampler's look-back workgroups wait
 * for every workgroup before them, which is only certain to end while one such launch is in flight.  The turn
 * is held on the host (a spin lock per device, no event between the streams): naqs_vmc_step / naqs_vmc_run / 
 * naqs_vmc_sample_forward_eloc keep it from the sampler's first launch until they have read the draw's size, which they wait for
 * anyway; naqs_net_sample / naqs_net_sample_weighted wait for `stream` to drain before they return (in this mode only).
 * on = 0: off; on = -1: leave as is.  Default: the value of NAQS_SHARED_GPU (0) when the handle was created.
 * turns (may be NULL) <- sampler calls of this handle so far that had to wait for another handle's turn to end.
 * The samples do not depend on the switch. */
int naqs_net_share_device(naqs_net_t *net, int on, int64_t *turns);
/* Name of the log-psi kernel the most recent naqs_net_logpsi / naqs_logpsi_eloc / training forward launched. */
int naqs_net_last_kernel(const naqs_net_t *net, char *buf, int buf_len);


/* ================================================================================================
 * Autoregressive tree sampler on the device.
 * Replaces wavefunction.sample(n) (src/naqs/wavefunction.py:488-521) -> _forward_sample
 * (src/naqs/network/nade.py:632-736) with multinomial_arr (:20-37): n_samples draws from |psi|^2 as the unique
 * bit-strings with their multiplicities.  Children that violate the electron budget are dropped after the draw
 * exactly like the reference (:695), so sum(counts) <= n_samples.  Outputs are in (prefix, outcome) order, which is
 * ascending key order for qubit_ordering = -1 (the reference's order).
 *   keys_dev [max_unique] uint64 (qubit order), counts_dev [max_unique] int64, probs_dev [max_unique] float32 or
 *   NULL (product of the float32 conditional probabilities, = the reference's `probs`),
 *   info_dev [2] int64: {number of unique samples M, overflow flag}.  overflow = 1 (and M = 0) when more than
 *   max_unique prefixes were alive at some level — the reference's MaxBatchSizeExceededError (:710-712).
 * Draws are a pure function of (weights, n_samples, seed): Philox4x32-10 keyed by the seed and indexed by the
 * prefix, exact binomials (inversion / BTRS).  Parity with the reference is statistical (numpy's generator is not
 * reproduced).  n_samples <= 2^44 (the float64 acceptance test of the binomial generator loses its margin beyond; the
 * reference caps n_samples at 1e12).
 * ============================================================================================== */
int naqs_net_sample(naqs_net_t *net, int64_t n_samples, uint64_t seed, int64_t max_unique, uint64_t *keys_dev,
                    int64_t *counts_dev, float *probs_dev, int64_t *info_dev, void *stream);

/* naqs_net_sample that also writes the samples' weights counts / sum(counts) (float64 [max_unique]; the `weights`
 * of PartialSamplingOptimizer.run, src/optimizer/energy.py:993) from the same final launch — the integer total is
 * exact, so the weights do not depend on a summation order. */
int naqs_net_sample_weighted(naqs_net_t *net, int64_t n_samples, uint64_t seed, int64_t max_unique, uint64_t *keys_dev,
                             int64_t *counts_dev, float *probs_dev, double *weights_dev, int64_t *info_dev, void *stream);

/* ================================================================================================
 * Training-time amplitude network: forward and backward of log|psi| in two launches each.
 * The reference back-propagates 2 Re sum_i w_i log psi_i (E_loc_i - <E>)^* through PyTorch autograd
 * (src/optimizer/energy.py:329-343); the amplitude half of that graph — ~10 orbital pairs x (2 Linear + mask +
 * log-softmax + gathers), src/naqs/network/nade.py:738-770 — is what these replace (the phase MLP is three plain
 * Linear layers and stays with the BLAS library).  Gradients are deterministic (fixed-order reductions).
 * ============================================================================================== */
/* Number of amplitude-block parameters = the leading part of the flat state_dict order of naqs_net_param_count. */
int naqs_net_amp_param_count(const naqs_net_t *net, int64_t *count);
/* Re-pack only the amplitude blocks (flat_dev: the whole flat parameter vector or just its amplitude part).
 * The packed phase layers become stale: naqs_net_logpsi / naqs_logpsi_eloc return NAQS_ERR_INVALID until the next
 * naqs_net_set_weights.  Enough for naqs_net_sample, naqs_net_logamp and naqs_net_amp_backward. */
int naqs_net_set_amp_weights(naqs_net_t *net, const float *flat_dev, int64_t count, void *stream);
/* logamp_dev [M] float32: log|psi(key_i)| (identical to column 0 of naqs_net_logpsi). */
int naqs_net_logamp(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, float *logamp_dev, void *stream);
/* grad_dev [amp_param_count] float32 (state_dict order) = d/d theta  sum_i g_dev[i] * log|psi(key_i)|. */
int naqs_net_amp_backward(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, const float *g_dev, float *grad_dev,
                          void *stream);

/* The whole network, forward and backward, for one batch of keys (no autograd graph):
 *   naqs_net_train_forward   = naqs_net_logpsi, and the phase MLP's inputs / hidden activations stay in the handle;
 *   naqs_net_train_backward  g_dev [M][2] float32 = d loss / d (log|psi_i|, phase_i)  ->  grad_dev [naqs_net_param_count]
 *                            float32 = d loss / d theta in state_dict order (amplitude blocks via naqs_net_amp_backward,
 *                            the phase Linear/ReLU stack as f32-MFMA GEMMs).  Must follow naqs_net_train_forward of the
 *                            SAME keys with no naqs_net_set_weights in between.  Deterministic. */
int naqs_net_train_forward(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, float *logpsi_dev, void *stream);
int naqs_net_train_backward(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, const float *g_dev, float *grad_dev,
                            void *stream);
/* naqs_vmc_loss_grad_ev + naqs_net_train_backward in ONE call (single-phase networks): the loss gradient g, its amplitude
 * column and the output layer's delta are written by one launch instead of three (vmc_grad, split, top-delta), then the backward
 * pass proper.  g_dev [M][2] and ev_dev [2] are outputs like naqs_vmc_loss_grad_ev's; grad_dev like naqs_net_train_backward's.
 * = loss.backward() of _SGD_step (src/optimizer/energy.py:328-343) given E_loc, the weights and the accumulators. */
int naqs_net_train_backward_vmc(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, const double *eloc_dev, const double *w_dev,
                                const double *sums_dev, float *g_dev, double *ev_dev, float *grad_dev, void *stream);
/* naqs_net_train_forward and the local energies of the same table in one call (= naqs_logpsi_eloc that also keeps the
 * activations for naqs_net_train_backward): the single-GPU training step's forward half, three launches. */
int naqs_net_train_forward_eloc(naqs_net_t *net, naqs_ham_t *ham, int64_t M, const uint64_t *keys_dev, const double *w_dev,
                                float *logpsi_dev, double *eloc_dev, double *out4_dev, void *stream);

/* One VMC step's first half in ONE call: naqs_net_sample_weighted, then — after the only host synchronisation of the step,
 * inside the library (the host has to learn M to size the launches) — naqs_net_train_forward_eloc of the sampled table.
 * Replaces wavefunction.sample + hilbert.state2idx + calculate_local_energy of PartialSamplingOptimizer.run
 * (src/optimizer/energy.py:975-998) as a sequence.  All `*_dev` outputs hold max_unique rows; rows >= M are unspecified.
 * info_host[0] = M, info_host[1] = overflow flag (then nothing was evaluated: MaxBatchSizeExceededError).  Between the
 * sampler's last kernel and the forward kernel the GPU waits for one stream synchronisation instead of for a return to
 * the caller's interpreter and a second library call. */
int naqs_vmc_sample_forward_eloc(naqs_net_t *net, naqs_ham_t *ham, int64_t n_samples, uint64_t seed, int64_t max_unique,
                                 uint64_t *keys_dev, int64_t *counts_dev, float *probs_dev, double *weights_dev,
                                 float *logpsi_dev, double *eloc_dev, double *out4_dev, int64_t *info_dev,
                                 int64_t info_host[2], void *stream);
/* ONE VMC training step in ONE call (single GPU, single-phase or aggregate-phase fused network):
 *   naqs_net_sample_weighted  ->  host learns (M, overflow)  ->  [accept M?]  ->  naqs_net_train_forward_eloc  ->
 *   naqs_net_train_backward_vmc  ->  naqs_adam_step on the flat parameter vector  ->  naqs_net_set_weights (re-pack).
 * = wavefunction.sample + calculate_local_energy + loss.backward() + optimizer.step() of PartialSamplingOptimizer.run /
 * _SGD_step (src/optimizer/energy.py:975-1008, :273-377).  The reference's adaptive sample count (energy.py:936-971) stays
 * with the caller: the step is abandoned after sampling — nothing evaluated, nothing updated — when the tree overflowed or
 * M is outside [m_lo, m_hi]; info_host[0] = M, [1] = overflow, [2] = 1 iff the step was taken.  adam_step < 1: stop after the
 * backward pass (the caller owns the optimiser).  The point of one call: between the sampler's last kernel and the re-pack
 * the GPU never waits for the caller's interpreter (measured: ~0.1 ms per step between the forward and the backward pass).
 * The one host/device rendezvous, M, is read from mapped host memory that the sampler writes as soon as the last level's size
 * is known (polled; NAQS_SPIN_WAIT=0: wait for the stream instead).  With adam_step >= 1 the reductions of the backward pass
 * apply Adam's update in the same launch (torch.optim.Adam's rule, as naqs_adam_step).
 * Re-pack: the kernels' packed copies of the updated parameters are made LATER, in stream order, from param_dev — inside the
 * next sampler call's first launch (whose own busy workgroup reads the four leading pairs' fragments: those are packed by the
 * update's launch itself), or at the start of whichever call of this handle first reads the amplitude blocks / the phase
 * layers (it starts the share it reads); a naqs_net_set_weights supersedes it.  param_dev must therefore stay allocated, and
 * unchanged unless naqs_net_set_weights follows, until one of those calls has been made (it is the optimiser's own parameter
 * vector in every caller here).  NAQS_PACK_OVERLAP=1: the amplitude blocks' share before the call returns, only the phase
 * layers' pending (ABI 6/7); 0: the whole re-pack before the call returns, as in ABI 5.  Same numbers in every form. */
int naqs_vmc_step(naqs_net_t *net, naqs_ham_t *ham, int64_t n_samples, uint64_t seed, int64_t max_unique, int64_t m_lo,
                  int64_t m_hi, uint64_t *keys_dev, int64_t *counts_dev, float *probs_dev, double *weights_dev,
                  float *logpsi_dev, double *eloc_dev, double *sums_dev, float *g_dev, double *ev_dev, float *grad_dev,
                  float *param_dev, float *exp_avg_dev, float *exp_avg_sq_dev, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int64_t adam_step, int64_t info_host[3], void *stream);
/* The LOOP of PartialSamplingOptimizer.run (src/optimizer/energy.py:975-1008) around naqs_vmc_step, with get_samples'
 * adaptive sample count (energy.py:936-971) in C: n_steps training steps in ONE call, nothing returns to the caller's
 * interpreter in between.  Per step: draw with the current n_samples (seed of the k-th sampling call of the run =
 * splitmix64(seed_base + 0x9E3779B97F4A7C15 k), the rule of naqs_amd.wavefunction._next_sample_seed); while the library's
 * naqs_vmc_step abandons the draw — tree overflow (MaxBatchSizeExceededError, nade.py:710-712), or fewer than
 * n_unq_samples_min unique samples while the count may still grow — adapt n_samples x10 / /10 exactly as get_samples does and
 * draw again, recording an event per adaptation (the caller prints the reference's messages from them); then forward + E_loc,
 * backward, Adam, re-pack.  Everything a step leaves behind is what naqs_vmc_step leaves: the `*_dev` row buffers (max_unique =
 * n_unq_samples_max rows) hold the LAST step's table on return.  Per-step records: (<E>, Var) -> ev_log_dev[i][2] and the four
 * weighted sums -> sums_log_dev[i][4] (device, written in stream order), M / n_samples / host seconds since the call began ->
 * the three host arrays.  Sampled-state tracking (energy.py:300): with ring_elems > 0 step i's keys are written at
 * keys_dev + ring_off and ring_off advances by M; the run stops early (stop_reason 1) when the next step's max_unique keys would
 * not fit, so that the caller can fold the buffer.  stop_reason: 0 all n_steps taken, 1 tracking buffer full, 2 event buffer
 * full, 3 a draw was abandoned without a rule to adapt by (an error in the reference too).  steps_done steps were taken in any
 * case; Adam's step count, the sampling-call counter, n_samples and ring_off are updated in place.
 * Results are bit-identical to calling naqs_vmc_step step by step with the same seeds (tests/test_optimizer_gpu.py).
 * (continued below the type definitions) */
typedef struct naqs_vmc_event {
    int64_t step;            /* 0-based index (within this call) of the step whose draw was being adapted */
    int64_t n_unique;        /* unique samples of the abandoned draw (n_unq_samples_max + 1 for an overflow) */
    int32_t overflow;        /* 1: the tree overflowed (the reference prints "MaxBatchSizeExceededError") */
    int32_t action;          /* +1 n_samples x10, -1 n_samples /10, 0 unchanged (overflow right after an increase) */
    int64_t n_samples;       /* n_samples after the adaptation */
} naqs_vmc_event_t;
typedef struct naqs_vmc_run_args {
    /* sampling policy (in/out: n_samples, sample_calls) */
    int64_t n_samples, n_samples_max, n_unq_samples_min, n_unq_samples_max;
    uint64_t seed_base;
    int64_t sample_calls;
    /* Adam on the flat parameter vector (in/out: adam_step = updates applied so far) */
    float *param_dev, *exp_avg_dev, *exp_avg_sq_dev, *grad_dev;
    double lr, beta1, beta2, eps, weight_decay;
    int64_t adam_step;
    /* row buffers, n_unq_samples_max rows each (keys_dev: ring_elems elements when tracking) */
    uint64_t *keys_dev;
    int64_t ring_elems, ring_off;
    int64_t *counts_dev;
    float *probs_dev;
    double *weights_dev;
    float *logpsi_dev;
    double *eloc_dev;
    float *g_dev;
    /* per-step records, n_steps entries each */
    double *ev_log_dev, *sums_log_dev;
    int64_t *m_log_host, *ns_log_host;
    double *t_log_host;
    naqs_vmc_event_t *events;
    int64_t events_cap;
    /* out */
    int64_t n_events, steps_done, last_keys_off;
    int32_t stop_reason, pad;
} naqs_vmc_run_args_t;
/* NAQS_DEFER_PHASE=1 (off by default: measured slower on this pool, the cross-stream hand-overs cost more than the overlap
 * wins): inside the run the phase MLP's half of a step — its share of the backward pass, its reductions + Adam update, its
 * re-pack — is issued on a second stream of the handle behind the amplitude blocks' half, so that it runs beside the NEXT
 * step's sampler (launches that read the amplitude blocks only and leave the chip almost empty) instead of in front of it;
 * the next forward pass waits for it, and so does this call before it returns: the caller never sees a half-updated
 * parameter vector.  Same kernels on the same operands either way. */
int naqs_vmc_run(naqs_net_t *net, naqs_ham_t *ham, int64_t n_steps, naqs_vmc_run_args_t *args, void *stream);
/* naqs_device_check for ONE network handle: its own wait-failure word only (what a caller that shares the device with other
 * runs asks; ABI 8). */
int naqs_net_check(naqs_net_t *net);
/* Order `stream` behind whatever work of this handle is still in flight on its own streams (the deferred phase chain of a
 * naqs_vmc_run that ended early with an error; a re-pack of the phase layers that no launch has hosted yet is started on
 * `stream`).  Every entry point that reads the phase layers does this itself; callers that read the flat parameter or
 * gradient buffers directly after such an error call it first.  No counterpart in the reference. */
int naqs_net_finish_pending(naqs_net_t *net, void *stream);
/* One Adam step on a flat float32 parameter vector (device pointers): torch.optim.Adam's rule without amsgrad —
 * the reference's optimiser, experiments/_base.py:228 (betas (0.9, 0.99), eps 1e-15).  `step` is the 1-based count
 * after this update (bias corrections 1 - beta^step are formed on the host in float64). */
int naqs_adam_step(int64_t n, float *param_dev, const float *grad_dev, float *exp_avg_dev, float *exp_avg_sq_dev,
                   double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, void *stream);
/* Inputs of the phase block from key bits: x_dev [M][2 (n_qubits/2 - 1)] float32 (+-1 occupations: alpha strings of
 * model pairs 0..P-2, then beta), occ_dev [M] int64 = realised outcome (alpha + 2 beta) of the last pair, which selects
 * the phase output (nade.py:563-569). */
int naqs_net_phase_inputs(naqs_net_t *net, int64_t M, const uint64_t *keys_dev, float *x_dev, int64_t *occ_dev,
                          void *stream);
/* g_dev [M][2] float32 = d loss / d (log|psi_i|, phase_i) of the VMC loss 2 Re sum_i w_i log psi_i (E_loc_i - <E>)^*
 * (energy.py:328-329), <E> = (sums_dev[0], sums_dev[1]) as produced by naqs_eloc_reduce, evaluated in float32 like
 * the reference: (2 w Re(E_loc - <E>), -2 w Im(E_loc - <E>)).  eloc_dev [M][2] float64, w_dev [M] float64. */
int naqs_vmc_loss_grad(int64_t M, const double *eloc_dev, const double *w_dev, const double *sums_dev, float *g_dev,
                       void *stream);

/* naqs_vmc_loss_grad that also writes ev_dev[2] = { <E> = sums[0]/sums[3], Var = sums[2]/sums[3] - <E>^2 } (float64), the
 * two numbers _SGD_step returns (energy.py:372-375), from the same launch. */
int naqs_vmc_loss_grad_ev(int64_t M, const double *eloc_dev, const double *w_dev, const double *sums_dev, float *g_dev,
                          double *ev_dev, void *stream);

/* Multi-GPU step: the payload of the accumulator all-reduce, assembled in one launch — ext_dev[8] (float64) =
 * { sums_dev[0..4), M, M^2, c, c^2 } with c = (sum of the M keys) mod 2^20.  After a SUM over W ranks,
 * W * sum(x^2) == (sum x)^2 for x = M and x = c holds iff every rank contributed the same table (all values are exact
 * integers in float64): the per-step proof that the ranks row-sharded ONE table (the reference is single-process; the
 * counterpart is the `weights` / `sampled_idxs` bookkeeping of src/optimizer/energy.py:300, :993). */
int naqs_shard_proof(int64_t M, const uint64_t *keys_dev, const double *sums_dev, double *ext_dev, void *stream);

/* ---- the row-sharded training step (world > 1) as FOUR library calls with the three collectives in between (round 4) ----
 * One process per GPU; every rank draws the same table and owns the rows [b, e) = [min(M, rank S), min(M, (rank + 1) S)),
 * S = ceil(M / world) (src/optimizer/energy.py:273-377 split over ranks as SURVEY 8e describes).  The caller's sequence:
 *   naqs_vmc_shard_sample_forward          sampler -> host learns (M, overflow) -> [accept M?] -> training forward of MY rows,
 *                                          (log|psi|, phase) of them written to the front of `logpsi_shard_dev` (>= S rows)
 *   all-gather of the shards               -> table [world][S_pad][2] float32 (S_pad >= S: equal padded contributions)
 *   naqs_eloc_gathered                     prep (hash table, psi) straight from the gathered layout, E_loc of my rows against
 *                                          the whole table, weighted sums -> ext8[0..4), same-table proof -> ext8[4..8)
 *   all-reduce of ext8
 *   naqs_net_train_backward_vmc            loss gradient + backward of my rows (sums = the reduced ext8)
 *   all-reduce of the flat gradient
 *   naqs_vmc_shard_update                  Adam on the flat parameter vector + re-pack of the kernels' weight layouts
 * Same kernels and the same arithmetic as the call-by-call path of round 3 (energies and parameters agree bit for bit);
 * what goes is the interpreter time between the pieces — the GPU idled for it.  info_host[0] = M, [1] = overflow,
 * [2] = 1 iff the step was taken (M inside [m_lo, m_hi], no overflow; otherwise nothing was evaluated). */
int naqs_vmc_shard_sample_forward(naqs_net_t *net, int64_t n_samples, uint64_t seed, int64_t max_unique, int64_t m_lo, int64_t m_hi,
                                  int rank, int world, uint64_t *keys_dev, int64_t *counts_dev, float *probs_dev,
                                  double *weights_dev, float *logpsi_shard_dev, int64_t info_host[3], void *stream);
/* table_dev: [world][S_pad][2] float32 (log|psi|, phase), shard r = rows r S .. of the M-row table; w_dev, eloc_dev: MY rows
 * (n_rows entries, row_begin = my first row); ext8_dev: {sum w Re E, sum w Im E, sum w Re^2 E, sum w, M, M^2, c, c^2}
 * (naqs_eloc_reduced + naqs_shard_proof). */
int naqs_eloc_gathered(naqs_ham_t *h, int64_t M, const uint64_t *keys_dev, const float *table_dev, int64_t S, int64_t S_pad,
                       int64_t row_begin, int64_t n_rows, const double *w_dev, double *eloc_dev, double *ext8_dev, void *stream);
/* naqs_adam_step on the network's flat parameter vector followed by naqs_net_set_weights of the updated parameters. */
int naqs_vmc_shard_update(naqs_net_t *net, const float *grad_dev, float *param_dev, float *exp_avg_dev, float *exp_avg_sq_dev,
                          double lr, double beta1, double beta2, double eps, double weight_decay, int64_t adam_step,
                          void *stream);

/* Host evaluation of the sampler's generators, for known-answer and statistical tests (no device needed):
 * out[i] = Binomial(n, p) drawn from stream (seed, i);  Philox4x32-10 block function. */
int naqs_rng_binomial_host(int64_t n, double p, uint64_t seed, int64_t reps, int64_t *out);
int naqs_rng_philox_host(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);
/* The generator's own elementary functions on the host (the code the device runs, except for its refined reciprocal), for
 * accuracy tests: y[i] = f(x[i]) with fn 0: log (x > 0, normal), 1: log(1 - x) (0 < x <= 1/2), 2: exp (-745 < x <= 0),
 * 3: the uniform on (0, 1) made from the two 32-bit words packed in x[i]'s bit pattern (hi << 32 | lo). */
int naqs_rng_math_host(int fn, int64_t n, const double *x, double *y);
/* The sampler's GROUP draws on the device (the code a tree level runs: `group` = 4 lanes per draw as in the first split of a
 * prefix's count, 2 as in the second), for statistical tests: out_dev[i] = Binomial(n_dev[i % cases], p_dev[i % cases]) from
 * stream (seed, i), i < reps — neighbouring draws of a wave take different cases, so waves hold the inversion and the BTRS
 * regime side by side.  n <= 2^44.  No counterpart in the reference (numpy's generator, nade.py:20-37). */
int naqs_rng_binomial_device(int group, int cases, const int64_t *n_dev, const double *p_dev, uint64_t seed, int64_t reps,
                             int64_t *out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NAQS_HIP_H */
