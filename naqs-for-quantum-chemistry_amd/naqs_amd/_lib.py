"""ctypes binding of ``include/naqs_hip.h``.

The product path has no CPU fallback: if ``libnaqs_hip.so`` is missing or does not export
the ABI, importing a compute entry point raises ``NaqsError`` loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_DIR = os.path.join(os.path.dirname(_HERE), "lib")

NAQS_OK = 0
ABI_VERSION = 9          # NAQS_ABI_VERSION of include/naqs_hip.h
PSI_F32, PSI_F64, LOGPSI_F32, LOGPSI_F64 = 0, 1, 2, 3

c_i64, c_u64p, c_f64p, c_vp = ctypes.c_int64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double), ctypes.c_void_p

# symbol -> (restype, argtypes); mirrors include/naqs_hip.h one-to-one (tests/test_abi.py checks it)
SIGNATURES = {
    "naqs_abi_version": (ctypes.c_int, []),
    "naqs_source_hash": (ctypes.c_char_p, []),
    "naqs_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "naqs_last_hip_error": (ctypes.c_int, []),
    "naqs_last_hip_error_string": (ctypes.c_char_p, []),
    "naqs_device_count": (ctypes.c_int, []),
    "naqs_device_check": (ctypes.c_int, [ctypes.c_int]),
    "naqs_net_check": (ctypes.c_int, [c_vp]),
    "naqs_terms_group": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp, ctypes.POINTER(c_i64), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "naqs_ham_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i64, c_vp, c_vp, c_vp,
                                       ctypes.c_int, ctypes.POINTER(c_vp)]),
    "naqs_ham_destroy": (ctypes.c_int, [c_vp]),
    "naqs_ham_info": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i64 * 8)]),
    "naqs_ham_reserve": (ctypes.c_int, [c_vp, c_i64]),
    "naqs_eloc": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, ctypes.c_int, c_i64, c_i64, c_vp, c_vp]),
    "naqs_eloc_reduced": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, ctypes.c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "naqs_eloc_reduce": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "naqs_hmatvec": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "naqs_popcount_parity": (ctypes.c_int, [c_vp, ctypes.c_int, c_i64, c_vp, c_vp]),
    "naqs_get_hij": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "naqs_hij_from_parity": (ctypes.c_int, [c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_vp, c_vp]),
    "naqs_csr_mv": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "naqs_ham_last_kernel": (ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_int]),
    "naqs_net_last_kernel": (ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_int]),
    "naqs_prof_enable": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "naqs_prof_read": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
    "naqs_prof_stride": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "naqs_net_prof_stride": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "naqs_net_create": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "naqs_net_destroy": (ctypes.c_int, [c_vp]),
    "naqs_net_param_count": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i64)]),
    "naqs_net_set_weights": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "naqs_net_logpsi": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "naqs_logpsi_eloc": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "naqs_net_prof_enable": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "naqs_net_prof_read": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
    "naqs_net_sample": (ctypes.c_int, [c_vp, c_i64, ctypes.c_uint64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "naqs_net_sample_weighted": (ctypes.c_int, [c_vp, c_i64, ctypes.c_uint64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "naqs_net_amp_param_count": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i64)]),
    "naqs_net_set_amp_weights": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "naqs_net_logamp": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "naqs_net_amp_backward": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "naqs_net_train_forward": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "naqs_net_train_forward_eloc": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "naqs_net_train_backward_vmc": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "naqs_vmc_step": (ctypes.c_int, [c_vp, c_vp, c_i64, ctypes.c_uint64, c_i64, c_i64, c_i64] + [c_vp] * 13 +
                      [ctypes.c_double] * 5 + [c_i64, ctypes.POINTER(c_i64), c_vp]),
    "naqs_vmc_run": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "naqs_net_finish_pending": (ctypes.c_int, [c_vp, c_vp]),
    "naqs_net_prof_select": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "naqs_launch_count": (c_i64, []),
    "naqs_net_spec_counts": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i64)]),
    "naqs_net_share_device": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.POINTER(c_i64)]),
    "naqs_shard_proof": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp, c_vp]),
    "naqs_vmc_shard_sample_forward": (ctypes.c_int, [c_vp, c_i64, ctypes.c_uint64, c_i64, c_i64, c_i64, ctypes.c_int, ctypes.c_int,
                                                     c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_i64), c_vp]),
    "naqs_eloc_gathered": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "naqs_vmc_shard_update": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp] + [ctypes.c_double] * 5 + [c_i64, c_vp]),
    "naqs_vmc_sample_forward_eloc": (ctypes.c_int, [c_vp, c_vp, c_i64, ctypes.c_uint64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                    c_vp, ctypes.POINTER(c_i64), c_vp]),
    "naqs_net_train_backward": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "naqs_adam_step": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_double, c_i64, c_vp]),
    "naqs_net_phase_inputs": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "naqs_vmc_loss_grad": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "naqs_vmc_loss_grad_ev": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "naqs_rng_binomial_host": (ctypes.c_int, [c_i64, ctypes.c_double, ctypes.c_uint64, c_i64, c_vp]),
    "naqs_rng_binomial_device": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_vp, c_vp, ctypes.c_uint64, c_i64, c_vp, c_vp]),
    "naqs_rng_philox_host": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "naqs_rng_math_host": (ctypes.c_int, [ctypes.c_int, c_i64, c_vp, c_vp]),
}

NET_MAX_PAIRS, NET_MAX_PHASE_LAYERS = 16, 8


class NetConfig(ctypes.Structure):
    """naqs_net_config_t"""
    _fields_ = [("n_qubits", ctypes.c_int32), ("n_alpha", ctypes.c_int32), ("n_beta", ctypes.c_int32),
                ("masking", ctypes.c_int32), ("use_amp_spin_sym", ctypes.c_int32), ("amp_hidden", ctypes.c_int32),
                ("n_phase_hidden", ctypes.c_int32), ("phase_hidden", ctypes.c_int32 * NET_MAX_PHASE_LAYERS),
                ("qubit2model", ctypes.c_int32 * (2 * NET_MAX_PAIRS)), ("aggregate_phase", ctypes.c_int32),
                ("use_phase_spin_sym", ctypes.c_int32)]


class VmcEvent(ctypes.Structure):
    """naqs_vmc_event_t"""
    _fields_ = [("step", c_i64), ("n_unique", c_i64), ("overflow", ctypes.c_int32), ("action", ctypes.c_int32), ("n_samples", c_i64)]


class VmcRunArgs(ctypes.Structure):
    """naqs_vmc_run_args_t"""
    _fields_ = [("n_samples", c_i64), ("n_samples_max", c_i64), ("n_unq_samples_min", c_i64), ("n_unq_samples_max", c_i64),
                ("seed_base", ctypes.c_uint64), ("sample_calls", c_i64),
                ("param_dev", c_vp), ("exp_avg_dev", c_vp), ("exp_avg_sq_dev", c_vp), ("grad_dev", c_vp),
                ("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double),
                ("weight_decay", ctypes.c_double), ("adam_step", c_i64),
                ("keys_dev", c_vp), ("ring_elems", c_i64), ("ring_off", c_i64), ("counts_dev", c_vp), ("probs_dev", c_vp),
                ("weights_dev", c_vp), ("logpsi_dev", c_vp), ("eloc_dev", c_vp), ("g_dev", c_vp),
                ("ev_log_dev", c_vp), ("sums_log_dev", c_vp), ("m_log_host", ctypes.POINTER(c_i64)),
                ("ns_log_host", ctypes.POINTER(c_i64)), ("t_log_host", ctypes.POINTER(ctypes.c_double)),
                ("events", ctypes.POINTER(VmcEvent)), ("events_cap", c_i64),
                ("n_events", c_i64), ("steps_done", c_i64), ("last_keys_off", c_i64), ("stop_reason", ctypes.c_int32),
                ("pad", ctypes.c_int32)]


class NaqsError(RuntimeError):
    pass


_lib = None


def lib_path():
    return os.environ.get("NAQS_HIP_LIB", os.path.join(_LIB_DIR, "libnaqs_hip.so"))


def load_library():
    """Load libnaqs_hip.so and type its entry points.  Raises NaqsError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) and
    # resolves it by file name; it must be in the process BEFORE libnaqs_hip.so so that the dynamic
    # loader binds our NEEDED libamdhip64.so.7 to that same copy — two HIP runtimes in one process
    # cannot share device pointers or streams (and the second one finds no device).
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise NaqsError(f"{path} not found: build it with `make -C naqs-for-quantum-chemistry_amd/csrc` "
                        "(or __graft_entry__.build()); there is no CPU fallback for the product path")
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise NaqsError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            if os.environ.get("NAQS_LOADER_LAX") == "1":      # A/B runs against older builds (tools/ only)
                continue
            raise NaqsError(f"{path} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.naqs_abi_version() != ABI_VERSION:
        raise NaqsError(f"{path}: ABI version {lib.naqs_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def check(status, what):
    if status != NAQS_OK:
        lib = load_library()
        msg = lib.naqs_strerror(status).decode()
        if status == -2:
            msg += ": " + lib.naqs_last_hip_error_string().decode()
        raise NaqsError(f"{what} failed: {msg} ({status})")
