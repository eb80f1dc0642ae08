"""ctypes binding of include/he_amd.h (no compute happens in Python)."""
import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# HEAMD_LIBRARY points at another build of the same ABI (a deployment's install path, or an A/B kernel experiment:
# bench_tools/ab_variants.py)
_LIB_PATH = os.environ.get("HEAMD_LIBRARY") or os.path.join(_PKG, "lib", "libhe_amd.so")

U64P = ctypes.POINTER(ctypes.c_uint64)
vp = ctypes.c_void_p
HOST_CALLBACK = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
c_u64 = ctypes.c_uint64
c_u32 = ctypes.c_uint32
c_size = ctypes.c_size_t

STATUS_NAMES = {
    0: "ok", 1: "invalidDegree", 2: "invalidModulus", 3: "coprimeModuli", 4: "emptyModulus",
    5: "invalidNttModulus", 6: "invalidPolyContext", 7: "polyContextMismatch", 8: "invalidCiphertext",
    9: "incompatibleCiphertexts", 10: "incompatibleCiphertextAndPlaintext", 11: "missingRelinearizationKey",
    12: "unequalContexts", 13: "notEnoughPrimes", 14: "notInvertible", 15: "invalidEncryptionParameters",
    16: "invalidArgument", 17: "deviceError", 18: "unsupportedHeOperation", 19: "missingGaloisKey",
    20: "serializedBufferSizeMismatch", 21: "invalidCoefficientPacking",
}


class HeError(RuntimeError):
    """Mirror of the reference's thrown HeError (HomomorphicEncryption/Error.swift:19-54)."""

    def __init__(self, code, detail=""):
        self.code = code
        self.name = STATUS_NAMES.get(code, "unknown")
        super().__init__(f"HeError.{self.name} ({code}) {detail}".strip())


# Every symbol include/he_amd.h declares: (name, restype, argtypes)
SIGNATURES = [
    ("he_status_string", ctypes.c_char_p, [ctypes.c_int]),
    ("he_last_error_message", ctypes.c_char_p, []),
    ("he_version", ctypes.c_char_p, []),
    ("he_device_count", ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    ("he_words_copy_device", ctypes.c_int, [vp, vp, c_size, ctypes.c_int, vp]),
    ("he_get_device", ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    ("he_set_device", ctypes.c_int, [ctypes.c_int]),
    ("he_set_scratch_cache", ctypes.c_int, [c_u64]),
    ("he_device_trim_scratch", ctypes.c_int, [c_u64]),
    ("he_scratch_cached_bytes", ctypes.c_int, [ctypes.POINTER(c_u64)]),
    ("he_shard_bounds", ctypes.c_int, [c_size, c_u32, c_u32, ctypes.POINTER(c_size), ctypes.POINTER(c_size)]),
    ("he_device_group_create", ctypes.c_int, [ctypes.POINTER(ctypes.c_int), c_u32, c_u32, c_u32, c_u64,
                                              ctypes.POINTER(c_u64), c_u32, ctypes.POINTER(vp)]),
    ("he_device_group_destroy", None, [vp]),
    ("he_device_group_size", c_u32, [vp]),
    ("he_device_group_device", ctypes.c_int, [vp, c_u32, ctypes.POINTER(ctypes.c_int)]),
    ("he_device_group_context", vp, [vp, c_u32]),
    ("he_device_group_stream", vp, [vp, c_u32]),
    ("he_device_group_synchronize", ctypes.c_int, [vp]),
    ("he_ntt_forward_group", ctypes.c_int, [vp, c_u32, ctypes.POINTER(vp), c_size]),
    ("he_ntt_inverse_group", ctypes.c_int, [vp, c_u32, ctypes.POINTER(vp), c_size]),
    ("he_pir_dim0_columns_group", ctypes.c_int, [vp, vp, c_size, ctypes.POINTER(vp), ctypes.POINTER(vp), c_size, vp, vp]),
    ("he_pir_compute_response_chunk_group", ctypes.c_int, [vp, ctypes.POINTER(c_u32), c_u32, vp, vp, c_size,
                                                           ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp, vp]),
    ("he_pir_compute_response_group", ctypes.c_int, [vp, ctypes.POINTER(c_u32), c_u32, vp, vp, c_size, ctypes.POINTER(vp),
                                                     ctypes.POINTER(vp), c_size, vp, vp, vp]),
    ("he_pir_remaining_dimensions_chunks_device", ctypes.c_int, [vp, ctypes.POINTER(c_u32), c_u32, c_size, vp, vp, c_size, vp,
                                                                 vp, vp]),
    ("he_device_malloc", ctypes.c_int, [ctypes.POINTER(vp), c_size]),
    ("he_device_free", ctypes.c_int, [vp]),
    ("he_host_malloc", ctypes.c_int, [ctypes.POINTER(vp), c_size]),
    ("he_host_free", ctypes.c_int, [vp]),
    ("he_memcpy_h2d", ctypes.c_int, [vp, vp, c_size, vp]),
    ("he_memcpy_d2h", ctypes.c_int, [vp, vp, c_size, vp]),
    ("he_stream_synchronize", ctypes.c_int, [vp]),
    ("he_stream_create", ctypes.c_int, [ctypes.POINTER(vp)]),
    ("he_stream_destroy", ctypes.c_int, [vp]),
    ("he_event_create", ctypes.c_int, [ctypes.POINTER(vp)]),
    ("he_event_destroy", ctypes.c_int, [vp]),
    ("he_event_record", ctypes.c_int, [vp, vp]),
    ("he_event_synchronize", ctypes.c_int, [vp]),
    ("he_event_query", ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int)]),
    ("he_stream_wait_event", ctypes.c_int, [vp, vp]),
    ("he_stream_add_callback", ctypes.c_int, [vp, HOST_CALLBACK, vp]),
    ("he_poly_context_create", ctypes.c_int, [c_u32, U64P, c_u32, ctypes.POINTER(vp)]),
    ("he_poly_context_destroy", None, [vp]),
    ("he_poly_context_degree", c_u32, [vp]),
    ("he_poly_context_moduli_count", c_u32, [vp]),
    ("he_poly_context_moduli", ctypes.c_int, [vp, U64P]),
    ("he_poly_context_max_lazy_product_accumulation_count", c_u64, [vp]),
    ("he_poly_context_q_remainder", ctypes.c_int, [vp, c_u64, U64P]),
    ("he_generate_primes", ctypes.c_int, [ctypes.POINTER(ctypes.c_int32), c_u32, ctypes.c_int, c_u32, U64P]),
    ("he_ntt_forward", ctypes.c_int, [vp, U64P, c_size]),
    ("he_ntt_inverse", ctypes.c_int, [vp, U64P, c_size]),
    ("he_ntt_forward_device", ctypes.c_int, [vp, vp, c_size, vp]),
    ("he_ntt_inverse_device", ctypes.c_int, [vp, vp, c_size, vp]),
    ("he_ntt_forward_rows_device", ctypes.c_int, [vp, c_u64, vp, c_size, vp]),
    ("he_ntt_inverse_rows_device", ctypes.c_int, [vp, c_u64, vp, c_size, vp]),
    ("he_poly_add_device", ctypes.c_int, [vp, vp, vp, c_size, vp]),
    ("he_poly_sub_device", ctypes.c_int, [vp, vp, vp, c_size, vp]),
    ("he_poly_neg_device", ctypes.c_int, [vp, vp, c_size, vp]),
    ("he_poly_mul_device", ctypes.c_int, [vp, vp, vp, c_size, vp]),
    ("he_poly_mul_scalar_device", ctypes.c_int, [vp, vp, U64P, c_size, vp]),
    ("he_poly_divide_and_round_q_last_device", ctypes.c_int, [vp, vp, vp, c_size, vp]),
    ("he_poly_divide_and_round_q_last", ctypes.c_int, [vp, U64P, U64P, c_size]),
    ("he_poly_adding_lazy_product_device", ctypes.c_int, [vp, vp, vp, vp, vp]),
    ("he_poly_reduce_accumulator_device", ctypes.c_int, [vp, vp, vp, vp]),
    ("he_poly_apply_galois_device", ctypes.c_int, [vp, vp, vp, c_size, c_u64, ctypes.c_int, vp]),
    ("he_poly_random_from_seeds_device", ctypes.c_int, [vp, vp, c_size, vp, vp]),
    ("he_ntt_forward_device_u32", ctypes.c_int, [vp, vp, c_size, vp]),
    ("he_ntt_inverse_device_u32", ctypes.c_int, [vp, vp, c_size, vp]),
    ("he_poly_add_device_u32", ctypes.c_int, [vp, vp, vp, c_size, vp]),
    ("he_poly_sub_device_u32", ctypes.c_int, [vp, vp, vp, c_size, vp]),
    ("he_poly_neg_device_u32", ctypes.c_int, [vp, vp, c_size, vp]),
    ("he_poly_mul_device_u32", ctypes.c_int, [vp, vp, vp, c_size, vp]),
    ("he_poly_mul_scalar_device_u32", ctypes.c_int, [vp, vp, ctypes.POINTER(ctypes.c_uint32), c_size, vp]),
    ("he_poly_divide_and_round_q_last_device_u32", ctypes.c_int, [vp, vp, vp, c_size, vp]),
    ("he_words_widen_u32_device", ctypes.c_int, [vp, vp, c_size, vp]),
    ("he_words_narrow_u64_device", ctypes.c_int, [vp, vp, c_size, vp]),
    ("he_poly_serialization_byte_count", c_size, [vp, ctypes.c_int]),
    ("he_poly_serialize_device", ctypes.c_int, [vp, vp, c_size, ctypes.c_int, vp, vp]),
    ("he_poly_deserialize_device", ctypes.c_int, [vp, vp, c_size, c_size, ctypes.c_int, vp, vp]),
    ("he_poly_multiply_power_of_x_device", ctypes.c_int, [vp, vp, vp, c_size, ctypes.c_int64, vp]),
    ("he_bfv_context_create", ctypes.c_int, [c_u32, c_u64, U64P, c_u32, ctypes.POINTER(vp)]),
    ("he_bfv_context_create_u32", ctypes.c_int, [c_u32, c_u64, U64P, c_u32, ctypes.POINTER(vp)]),
    ("he_bfv_context_destroy", None, [vp]),
    ("he_bfv_ciphertext_moduli_count", c_u32, [vp]),
    ("he_bfv_ciphertext_context", vp, [vp, c_u32]),
    ("he_bfv_key_switching_context", vp, [vp, c_u32]),
    ("he_bfv_qbsk_context", vp, [vp, c_u32]),
    ("he_rns_lift_q_to_qbsk_device", ctypes.c_int, [vp, c_u32, vp, vp, c_size, vp]),
    ("he_rns_floor_qbsk_to_q_device", ctypes.c_int, [vp, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_mul_workspace_bytes", c_size, [vp, c_u32, c_size]),
    ("he_bfv_relinearize_workspace_bytes", c_size, [vp, c_u32, c_size]),
    ("he_bfv_inner_product_workspace_bytes", c_size, [vp, c_u32, c_size]),
    ("he_bfv_mul_device", ctypes.c_int, [vp, c_u32, vp, vp, vp, c_size, vp, c_size, vp]),
    ("he_bfv_relinearize_device", ctypes.c_int, [vp, c_u32, vp, vp, vp, c_size, vp, c_size, vp]),
    ("he_bfv_mod_switch_down_device", ctypes.c_int, [vp, c_u32, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_mod_switch_down_to_single_device", ctypes.c_int, [vp, c_u32, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_mul_plain_device", ctypes.c_int, [vp, c_u32, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_add_plain_device", ctypes.c_int, [vp, c_u32, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_sub_plain_device", ctypes.c_int, [vp, c_u32, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_inner_product_plain_device", ctypes.c_int,
     [vp, c_u32, c_u32, vp, vp, ctypes.POINTER(ctypes.c_uint8), c_size, c_size, vp, vp]),
    ("he_bfv_inner_product_plain_resident_device", ctypes.c_int,
     [vp, c_u32, c_u32, vp, vp, vp, c_size, c_size, vp, vp]),
    ("he_bfv_inner_product_device", ctypes.c_int, [vp, c_u32, vp, vp, c_size, vp, vp, c_size, vp]),
    ("he_bfv_inner_product_shared_device", ctypes.c_int, [vp, c_u32, vp, vp, c_size, c_size, vp, vp]),
    ("he_bfv_packed_plaintext_words", c_size, [vp, c_u32]),
    ("he_bfv_pack_plaintexts_device", ctypes.c_int, [vp, c_u32, vp, c_size, vp, vp]),
    ("he_bfv_inner_product_plain_packed_device", ctypes.c_int, [vp, c_u32, c_u32, vp, vp, vp, c_size, c_size, vp, vp]),
    ("he_pir_dim0_columns_packed_device", ctypes.c_int, [vp, vp, c_size, vp, vp, c_size, vp, vp]),
    ("he_pir_compute_response_to_query_device", ctypes.c_int,
     [vp, ctypes.POINTER(c_u32), c_u32, vp, c_size, c_size, U64P, ctypes.POINTER(vp), c_size, vp, ctypes.POINTER(vp),
      ctypes.POINTER(vp), c_size, c_size, vp, vp]),
    ("he_pir_compute_response_packed_device", ctypes.c_int,
     [vp, ctypes.POINTER(c_u32), c_u32, vp, vp, c_size, vp, vp, c_size, vp, vp, vp]),
    # Bfv<UInt32> on packed 4-byte slabs
    ("he_rns_lift_q_to_qbsk_device_u32", ctypes.c_int, [vp, c_u32, vp, vp, c_size, vp]),
    ("he_rns_floor_qbsk_to_q_device_u32", ctypes.c_int, [vp, c_u32, vp, vp, c_size, vp]),
    ("he_rns_scale_and_round_device_u32", ctypes.c_int, [vp, c_u32, vp, c_u64, vp, c_size, vp]),
    ("he_bfv_mul_device_u32", ctypes.c_int, [vp, c_u32, vp, vp, vp, c_size, vp, c_size, vp]),
    ("he_bfv_relinearize_device_u32", ctypes.c_int, [vp, c_u32, vp, vp, vp, c_size, vp, c_size, vp]),
    ("he_bfv_apply_galois_device_u32", ctypes.c_int, [vp, c_u32, vp, c_u64, vp, vp, c_size, vp, c_size, vp]),
    ("he_bfv_mod_switch_down_device_u32", ctypes.c_int, [vp, c_u32, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_mul_plain_device_u32", ctypes.c_int, [vp, c_u32, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_add_plain_device_u32", ctypes.c_int, [vp, c_u32, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_sub_plain_device_u32", ctypes.c_int, [vp, c_u32, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_inner_product_plain_resident_device_u32", ctypes.c_int,
     [vp, c_u32, c_u32, vp, vp, vp, c_size, c_size, vp, vp]),
    ("he_bfv_inner_product_device_u32", ctypes.c_int, [vp, c_u32, vp, vp, c_size, vp, vp, c_size, vp]),
    ("he_bfv_inner_product_shared_device_u32", ctypes.c_int, [vp, c_u32, vp, vp, c_size, c_size, vp, vp]),
    ("he_pir_compute_response_device_u32", ctypes.c_int,
     [vp, ctypes.POINTER(c_u32), c_u32, vp, vp, c_size, vp, vp, c_size, vp, vp, vp]),
    ("he_pir_compute_response_queries_device_u32", ctypes.c_int,
     [vp, ctypes.POINTER(c_u32), c_u32, c_size, vp, vp, c_size, vp, vp, c_size, ctypes.POINTER(vp), vp, vp]),
    ("he_pir_compute_response_to_query_device_u32", ctypes.c_int,
     [vp, ctypes.POINTER(c_u32), c_u32, vp, c_size, c_size, U64P, ctypes.POINTER(vp), c_size, vp, ctypes.POINTER(vp),
      ctypes.POINTER(vp), c_size, c_size, vp, vp]),
    ("he_bfv_plaintext_to_eval_device_u32", ctypes.c_int, [vp, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_plaintext_to_coeff_device_u32", ctypes.c_int, [vp, c_u32, vp, vp, c_size, vp]),
    ("he_galois_element_swapping_rows", ctypes.c_int, [c_u64, U64P]),
    ("he_galois_element_rotating_columns", ctypes.c_int, [ctypes.c_int64, c_u64, U64P]),
    ("he_bfv_apply_galois_workspace_bytes", c_size, [vp, c_u32, c_size]),
    ("he_bfv_apply_galois_device", ctypes.c_int, [vp, c_u32, vp, c_u64, vp, vp, c_size, vp, c_size, vp]),
    ("he_rns_scale_and_round_device", ctypes.c_int, [vp, c_u32, vp, c_u64, vp, c_size, vp]),
    ("he_bfv_plaintext_to_eval_device", ctypes.c_int, [vp, c_u32, vp, vp, c_size, vp]),
    ("he_bfv_plaintext_to_coeff_device", ctypes.c_int, [vp, c_u32, vp, vp, c_size, vp]),
    ("he_pir_compute_response_chunk_device", ctypes.c_int,
     [vp, ctypes.POINTER(c_u32), c_u32, vp, vp, c_size, vp, ctypes.POINTER(ctypes.c_uint8), vp, vp, vp]),
    ("he_pir_dim0_columns_device", ctypes.c_int, [vp, vp, c_size, vp, vp, c_size, vp, vp]),
    ("he_pir_remaining_dimensions_device", ctypes.c_int,
     [vp, ctypes.POINTER(c_u32), c_u32, vp, vp, c_size, vp, vp, vp]),
    ("he_pir_compute_response_device", ctypes.c_int,
     [vp, ctypes.POINTER(c_u32), c_u32, vp, vp, c_size, vp, vp, c_size, vp, vp, vp]),
    ("he_pir_compute_response_queries_device", ctypes.c_int,
     [vp, ctypes.POINTER(c_u32), c_u32, c_size, vp, vp, c_size, vp, vp, c_size, ctypes.POINTER(vp), vp, vp]),
    ("he_pir_expand_batch_device", ctypes.c_int,
     [vp, vp, c_size, c_size, c_size, U64P, ctypes.POINTER(vp), c_size, vp, vp]),
    ("he_bfv_apply_galois_grouped_device", ctypes.c_int,
     [vp, c_u32, vp, c_u64, ctypes.POINTER(vp), c_size, c_size, vp, vp, c_size, vp]),
    ("he_pir_expand_device", ctypes.c_int,
     [vp, vp, c_size, c_size, U64P, ctypes.POINTER(vp), c_size, vp, vp]),
    # diagnostics / test hooks
    ("he_poly_context_create_host_only", ctypes.c_int, [c_u32, U64P, c_u32, ctypes.POINTER(vp)]),
    ("he_poly_context_copy_ntt_tables", ctypes.c_int, [vp, c_u32, U64P, U64P, U64P, U64P, U64P, U64P]),
    ("he_ntt_device_variant", ctypes.c_int, [vp, vp, c_size, ctypes.c_int, ctypes.c_int, vp]),
    ("he_bfv_context_create_host_only", ctypes.c_int, [c_u32, c_u64, U64P, c_u32, ctypes.POINTER(vp)]),
    ("he_bfv_copy_bsk_moduli", ctypes.c_int, [vp, U64P]),
]

_lib = None


def library_path():
    return _LIB_PATH


def load_library():
    """Loads libhe_amd.so.  Fails loudly when it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} is missing: build the HIP extension first "
                "(python swift-homomorphic-encryption_amd/build.py, or __graft_entry__.build()).")
        try:
            # PyTorch ships its own libamdhip64.so.7 / libhsa-runtime64.so.1.  Import it first so that libhe_amd.so
            # binds to the SAME HIP runtime instance as the tensors it is handed (two runtimes in one process do not
            # see each other's devices).  Without PyTorch the library uses the system ROCm runtime.
            import torch  # noqa: F401
        except ImportError:  # pragma: no cover
            pass
        lib = ctypes.CDLL(_LIB_PATH)
        for name, restype, argtypes in SIGNATURES:
            fn = getattr(lib, name)  # AttributeError = header and library disagree
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def _check(code):
    if code != 0:
        detail = load_library().he_last_error_message().decode(errors="replace")
        raise HeError(code, detail)


def version():
    return load_library().he_version().decode()


def device_count():
    n = ctypes.c_int(0)
    _check(load_library().he_device_count(ctypes.byref(n)))
    return n.value


def current_device():
    n = ctypes.c_int(0)
    _check(load_library().he_get_device(ctypes.byref(n)))
    return n.value


def set_device(device):
    _check(load_library().he_set_device(device))


def set_scratch_cache(nbytes=2**64 - 1):
    """Let the library's scratch pool of the current device keep up to `nbytes` of freed scratch (default: all)."""
    _check(load_library().he_set_scratch_cache(nbytes))


def trim_scratch(keep_bytes=0):
    _check(load_library().he_device_trim_scratch(keep_bytes))


def scratch_cached_bytes():
    """Bytes of released scratch the library's cache holds on the current device (he_scratch_cached_bytes)."""
    out = c_u64(0)
    _check(load_library().he_scratch_cached_bytes(ctypes.byref(out)))
    return out.value


def shard_bounds(total, members, member):
    """he_shard_bounds: member `member` of `members` owns units [begin, end) of `total`."""
    begin, end = c_size(0), c_size(0)
    _check(load_library().he_shard_bounds(total, members, member, ctypes.byref(begin), ctypes.byref(end)))
    return begin.value, end.value


GROUP_STAGE_ALL = 1


class DeviceGroup:
    """he_device_group: one Context<Bfv<UInt64>> and one stream per member device of ONE process; the units of a call split
    over the members by shard_bounds, the finished shards gathered on member 0's device (include/he_amd.h "Device groups")."""

    def __init__(self, devices, degree, plaintext_modulus, coefficient_moduli, stage_all=False):
        ids = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        q = (c_u64 * len(coefficient_moduli))(*[int(m) for m in coefficient_moduli])
        handle = vp()
        _check(load_library().he_device_group_create(ids, len(devices), GROUP_STAGE_ALL if stage_all else 0, degree,
                                                     int(plaintext_modulus), q, len(coefficient_moduli), ctypes.byref(handle)))
        self.h = handle
        self.degree = degree
        self.devices = [int(d) for d in devices]
        self.L = len(coefficient_moduli) - 1 if len(coefficient_moduli) > 1 else 1

    def __del__(self):
        if getattr(self, "h", None):
            load_library().he_device_group_destroy(self.h)
            self.h = None

    def __len__(self):
        return int(load_library().he_device_group_size(self.h))

    def bounds(self, total, member):
        return shard_bounds(total, len(self), member)

    def stream(self, member):
        """The member's stream as a torch stream (owned by the group)."""
        import torch

        return torch.cuda.ExternalStream(int(load_library().he_device_group_stream(self.h, member)),
                                         device=self.devices[member])

    def synchronize(self):
        _check(load_library().he_device_group_synchronize(self.h))

    def _shards(self, tensors):
        return (vp * len(tensors))(*[vp(0) if t is None else vp(t.data_ptr()) for t in tensors])

    def forward_ntt_(self, slab_shards, batch, moduli_count=None):
        _check(load_library().he_ntt_forward_group(self.h, moduli_count or self.L, self._shards(slab_shards), batch))
        return slab_shards

    def inverse_ntt_(self, slab_shards, batch, moduli_count=None):
        _check(load_library().he_ntt_inverse_group(self.h, moduli_count or self.L, self._shards(slab_shards), batch))
        return slab_shards

    def pir_dim0_columns(self, dim0_query_eval, database_shards, columns, present_shards=None, stream=None):
        """he_pir_dim0_columns_group: database_shards[m] = member m's columns [share][d0][L][N] on its device; the result
        [columns][2][L][N] on member 0's device (where dim0_query_eval lives)."""
        import torch

        d0 = dim0_query_eval.numel() // (2 * self.L * self.degree)
        out = torch.empty((columns, 2, self.L, self.degree), dtype=dim0_query_eval.dtype, device=dim0_query_eval.device)
        masks = None if present_shards is None else self._shards(present_shards)
        _check(load_library().he_pir_dim0_columns_group(self.h, _ptr(dim0_query_eval), d0, self._shards(database_shards),
                                                        masks, columns, _ptr(out), _stream(stream)))
        return out

    def pir_compute_response(self, dimensions, dim0_query_eval, remaining_query, database_shards, chunk_count,
                             present_shards=None, relinearization_key=None, stream=None):
        """he_pir_compute_response_group: the chunk loop over the group; database_shards[m] = member m's share of the
        chunk_count x columns columns of all chunks -> [chunks][2][1][N] on member 0's device."""
        import torch

        dims = (c_u32 * len(dimensions))(*[int(d) for d in dimensions])
        out = torch.empty((chunk_count, 2, 1, self.degree), dtype=dim0_query_eval.dtype, device=dim0_query_eval.device)
        rest = vp() if remaining_query is None else _ptr(remaining_query)
        rest_count = 0 if remaining_query is None else remaining_query.numel() // (2 * self.L * self.degree)
        key = vp() if relinearization_key is None else _ptr(relinearization_key)
        masks = None if present_shards is None else self._shards(present_shards)
        _check(load_library().he_pir_compute_response_group(self.h, dims, len(dims), _ptr(dim0_query_eval), rest, rest_count,
                                                            self._shards(database_shards), masks, chunk_count, key,
                                                            _ptr(out), _stream(stream)))
        return out

    def pir_compute_response_chunk(self, dimensions, dim0_query_eval, remaining_query, database_shards,
                                   present_shards=None, relinearization_key=None, stream=None):
        """he_pir_compute_response_chunk_group -> [2][1][N] on member 0's device."""
        import torch

        dims = (c_u32 * len(dimensions))(*[int(d) for d in dimensions])
        out = torch.empty((2, 1, self.degree), dtype=dim0_query_eval.dtype, device=dim0_query_eval.device)
        rest = vp() if remaining_query is None else _ptr(remaining_query)
        rest_count = 0 if remaining_query is None else remaining_query.numel() // (2 * self.L * self.degree)
        key = vp() if relinearization_key is None else _ptr(relinearization_key)
        masks = None if present_shards is None else self._shards(present_shards)
        _check(load_library().he_pir_compute_response_chunk_group(self.h, dims, len(dims), _ptr(dim0_query_eval), rest,
                                                                  rest_count, self._shards(database_shards), masks, key,
                                                                  _ptr(out), _stream(stream)))
        return out


def stream_copy(src, dst, non_temporal=False, stream=None):
    """dst <- src with the library's streaming copy (8 bytes per lane): the roofline's reference rate."""
    _check(load_library().he_words_copy_device(vp(src.data_ptr()), vp(dst.data_ptr()), src.numel(),
                                               1 if non_temporal else 0, _stream(stream)))


def widen_u32(slab32, stream=None):
    """[UInt32] words (int32 tensor) -> zero-extended 8-byte words (int64 tensor of the same shape), on the device."""
    import torch

    out = torch.empty(slab32.shape, dtype=torch.int64, device=slab32.device)
    _check(load_library().he_words_widen_u32_device(vp(slab32.data_ptr()), vp(out.data_ptr()), slab32.numel(),
                                                    _stream(stream)))
    return out


def narrow_u64(slab64, stream=None):
    """8-byte words holding UInt32 values -> packed int32 tensor of the same shape."""
    import torch

    out = torch.empty(slab64.shape, dtype=torch.int32, device=slab64.device)
    _check(load_library().he_words_narrow_u64_device(vp(slab64.data_ptr()), vp(out.data_ptr()), slab64.numel(),
                                                     _stream(stream)))
    return out


def galois_element_swapping_rows(degree):
    out = ctypes.c_uint64(0)
    _check(load_library().he_galois_element_swapping_rows(degree, ctypes.byref(out)))
    return out.value


def galois_element_rotating_columns(step, degree):
    out = ctypes.c_uint64(0)
    _check(load_library().he_galois_element_rotating_columns(step, degree, ctypes.byref(out)))
    return out.value


def generate_primes(bit_counts, preferring_small, ntt_degree=1):
    bits = (ctypes.c_int32 * len(bit_counts))(*bit_counts)
    out = np.zeros(len(bit_counts), dtype=np.uint64)
    _check(load_library().he_generate_primes(bits, len(bit_counts), int(preferring_small), ntt_degree,
                                            out.ctypes.data_as(U64P)))
    return [int(v) for v in out]


def _u64(values):
    return np.ascontiguousarray(values, dtype=np.uint64)


def to_device(array, device="cuda"):
    """numpy uint64 array -> torch int64 CUDA tensor holding the same words."""
    import torch

    a = np.ascontiguousarray(array, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_host(tensor):
    """torch int64 tensor -> numpy uint64 array (same words)."""
    return tensor.detach().cpu().contiguous().numpy().view(np.uint64)


def to_device32(array, device="cuda"):
    """numpy array of values < 2^32 -> torch int32 CUDA tensor holding them as packed UInt32 words."""
    import torch

    a = np.ascontiguousarray(array, dtype=np.uint64).astype(np.uint32)
    return torch.from_numpy(a.view(np.int32)).to(device)


def to_host32(tensor):
    """torch int32 tensor of packed UInt32 words -> numpy uint64 array of the values."""
    return tensor.detach().cpu().contiguous().numpy().view(np.uint32).astype(np.uint64)


def _ptr32(tensor):
    if not tensor.is_cuda or not tensor.is_contiguous() or tensor.element_size() != 4:
        raise ValueError("expected a contiguous CUDA tensor of 32-bit words (int32 storage)")
    return vp(tensor.data_ptr())


def _ptr(tensor):
    if not tensor.is_cuda:
        raise ValueError("expected a CUDA (HIP) tensor")
    if not tensor.is_contiguous():
        raise ValueError("expected a contiguous tensor")
    if tensor.element_size() != 8:
        raise ValueError("expected 64-bit words (int64 storage)")
    return vp(tensor.data_ptr())


def _stream(stream):
    import torch

    if stream is None:
        stream = torch.cuda.current_stream()
    return vp(stream.cuda_stream)


class PolyContext:
    """PolyContext<UInt64> (reference PolyRq/PolyContext.swift:19-123) resident on the current GPU."""

    def __init__(self, degree, moduli, host_only=False, _borrowed=None, _keepalive=None):
        lib = load_library()
        self._owned = _borrowed is None
        self._keepalive = _keepalive
        if _borrowed is not None:
            self.h = vp(_borrowed)
        else:
            arr = _u64(list(moduli))
            h = vp()
            create = lib.he_poly_context_create_host_only if host_only else lib.he_poly_context_create
            _check(create(degree, arr.ctypes.data_as(U64P), len(arr), ctypes.byref(h)))
            self.h = h
        self.degree = int(lib.he_poly_context_degree(self.h))
        count = int(lib.he_poly_context_moduli_count(self.h))
        out = np.zeros(count, dtype=np.uint64)
        _check(lib.he_poly_context_moduli(self.h, out.ctypes.data_as(U64P)))
        self.moduli = [int(v) for v in out]

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "h", None) and _lib is not None:
            _lib.he_poly_context_destroy(self.h)
            self.h = None

    # ---- host-side queries
    def max_lazy_product_accumulation_count(self):
        return int(load_library().he_poly_context_max_lazy_product_accumulation_count(self.h))

    def q_remainder(self, modulus):
        out = c_u64(0)
        _check(load_library().he_poly_context_q_remainder(self.h, modulus, ctypes.byref(out)))
        return int(out.value)

    def ntt_tables(self, rns_index):
        n = self.degree
        arrs = [np.zeros(n, dtype=np.uint64) for _ in range(4)]
        inv_n, inv_n_root = c_u64(0), c_u64(0)
        _check(load_library().he_poly_context_copy_ntt_tables(
            self.h, rns_index, *[a.ctypes.data_as(U64P) for a in arrs], ctypes.byref(inv_n), ctypes.byref(inv_n_root)))
        return dict(root_powers=arrs[0], root_factors=arrs[1], inv_root_powers=arrs[2], inv_root_factors=arrs[3],
                    inverse_degree=int(inv_n.value), inverse_degree_root=int(inv_n_root.value))

    def _batch(self, tensor, rows=None):
        per = (len(self.moduli) if rows is None else rows) * self.degree
        if tensor.numel() % per:
            raise ValueError(f"slab of {tensor.numel()} words is not [batch][{per // self.degree}][{self.degree}]")
        return tensor.numel() // per

    # ---- PolyRq.forwardNtt / inverseNtt (in place, device tensors)
    def forward_ntt_(self, slab, stream=None):
        _check(load_library().he_ntt_forward_device(self.h, _ptr(slab), self._batch(slab), _stream(stream)))
        return slab

    def inverse_ntt_(self, slab, stream=None):
        _check(load_library().he_ntt_inverse_device(self.h, _ptr(slab), self._batch(slab), _stream(stream)))
        return slab

    def ntt_variant_(self, slab, inverse, variant, stream=None):
        _check(load_library().he_ntt_device_variant(self.h, _ptr(slab), self._batch(slab), int(inverse), variant,
                                                    _stream(stream)))
        return slab

    def forward_ntt_rows_(self, modulus, rows, stream=None):
        _check(load_library().he_ntt_forward_rows_device(self.h, modulus, _ptr(rows), rows.numel() // self.degree,
                                                         _stream(stream)))
        return rows

    def inverse_ntt_rows_(self, modulus, rows, stream=None):
        _check(load_library().he_ntt_inverse_rows_device(self.h, modulus, _ptr(rows), rows.numel() // self.degree,
                                                         _stream(stream)))
        return rows

    # host-pointer forms (numpy in, numpy out): the reference's borrowed-pointer seam
    def forward_ntt_host(self, array):
        out = _u64(array).copy()
        _check(load_library().he_ntt_forward(self.h, out.ctypes.data_as(U64P), self._batch_np(out)))
        return out

    def forward_ntt_host_(self, array):
        """In place on a host array (C-contiguous uint64): the call PolyRq.forwardNtt() on a host polynomial maps to."""
        assert array.dtype == np.uint64 and array.flags["C_CONTIGUOUS"] and array.flags["WRITEABLE"]
        _check(load_library().he_ntt_forward(self.h, array.ctypes.data_as(U64P), self._batch_np(array)))
        return array

    def inverse_ntt_host_(self, array):
        assert array.dtype == np.uint64 and array.flags["C_CONTIGUOUS"] and array.flags["WRITEABLE"]
        _check(load_library().he_ntt_inverse(self.h, array.ctypes.data_as(U64P), self._batch_np(array)))
        return array

    def inverse_ntt_host(self, array):
        out = _u64(array).copy()
        _check(load_library().he_ntt_inverse(self.h, out.ctypes.data_as(U64P), self._batch_np(out)))
        return out

    def _batch_np(self, array):
        per = len(self.moduli) * self.degree
        if array.size % per:
            raise ValueError("slab is not [batch][L][N]")
        return array.size // per

    # ---- element-wise (in place on lhs)
    def add_(self, lhs, rhs, stream=None):
        _check(load_library().he_poly_add_device(self.h, _ptr(lhs), _ptr(rhs), self._batch(lhs), _stream(stream)))
        return lhs

    def sub_(self, lhs, rhs, stream=None):
        _check(load_library().he_poly_sub_device(self.h, _ptr(lhs), _ptr(rhs), self._batch(lhs), _stream(stream)))
        return lhs

    def neg_(self, data, stream=None):
        _check(load_library().he_poly_neg_device(self.h, _ptr(data), self._batch(data), _stream(stream)))
        return data

    def mul_(self, lhs, rhs, stream=None):
        _check(load_library().he_poly_mul_device(self.h, _ptr(lhs), _ptr(rhs), self._batch(lhs), _stream(stream)))
        return lhs

    def mul_scalar_(self, data, scalar_residues, stream=None):
        s = _u64(list(scalar_residues))
        _check(load_library().he_poly_mul_scalar_device(self.h, _ptr(data), s.ctypes.data_as(U64P), self._batch(data),
                                                        _stream(stream)))
        return data

    def divide_and_round_q_last(self, slab, stream=None):
        import torch

        batch = self._batch(slab)
        out = torch.empty((batch, max(len(self.moduli) - 1, 0), self.degree), dtype=torch.int64, device=slab.device)
        _check(load_library().he_poly_divide_and_round_q_last_device(self.h, _ptr(slab), vp(out.data_ptr()), batch,
                                                                     _stream(stream)))
        return out

    def divide_and_round_q_last_host(self, array):
        a = _u64(array)
        batch = self._batch_np(a)
        out = np.zeros((batch, len(self.moduli) - 1, self.degree), dtype=np.uint64)
        _check(load_library().he_poly_divide_and_round_q_last(self.h, a.ctypes.data_as(U64P),
                                                              out.ctypes.data_as(U64P), batch))
        return out

    # ---- PolyRq<UInt32>: int32 tensors [batch][L][N] holding UInt32 words ----
    def _batch32(self, slab):
        return slab.numel() // (len(self.moduli) * self.degree)

    def forward_ntt_u32_(self, slab, stream=None):
        _check(load_library().he_ntt_forward_device_u32(self.h, vp(slab.data_ptr()), self._batch32(slab), _stream(stream)))
        return slab

    def inverse_ntt_u32_(self, slab, stream=None):
        _check(load_library().he_ntt_inverse_device_u32(self.h, vp(slab.data_ptr()), self._batch32(slab), _stream(stream)))
        return slab

    def elementwise_u32_(self, op, lhs, rhs=None, stream=None):
        lib = load_library()
        batch = self._batch32(lhs)
        if op == "neg":
            _check(lib.he_poly_neg_device_u32(self.h, vp(lhs.data_ptr()), batch, _stream(stream)))
        else:
            fn = {"add": lib.he_poly_add_device_u32, "sub": lib.he_poly_sub_device_u32,
                  "mul": lib.he_poly_mul_device_u32}[op]
            _check(fn(self.h, vp(lhs.data_ptr()), vp(rhs.data_ptr()), batch, _stream(stream)))
        return lhs

    def mul_scalar_u32_(self, data, scalar_residues, stream=None):
        arr = (ctypes.c_uint32 * len(self.moduli))(*[int(v) for v in scalar_residues])
        _check(load_library().he_poly_mul_scalar_device_u32(self.h, vp(data.data_ptr()), arr, self._batch32(data),
                                                            _stream(stream)))
        return data

    def divide_and_round_q_last_u32(self, slab, stream=None):
        import torch

        batch = self._batch32(slab)
        out = torch.empty((batch, len(self.moduli) - 1, self.degree), dtype=torch.int32, device=slab.device)
        _check(load_library().he_poly_divide_and_round_q_last_device_u32(self.h, vp(slab.data_ptr()),
                                                                         vp(out.data_ptr()), batch, _stream(stream)))
        return out

    def random_from_seeds(self, seeds, stream=None):
        """PolyRq.random(context:using: NistAes128Ctr(seed:)) per seed: uint8 tensor [batch][32] -> [batch][L][N]."""
        import torch

        batch = seeds.numel() // 32
        out = torch.empty((batch, len(self.moduli), self.degree), dtype=torch.int64, device=seeds.device)
        _check(load_library().he_poly_random_from_seeds_device(self.h, vp(seeds.data_ptr()), batch,
                                                               vp(out.data_ptr()), _stream(stream)))
        return out

    def serialization_byte_count(self, skip_lsbs=0):
        return int(load_library().he_poly_serialization_byte_count(self.h, skip_lsbs))

    def serialize(self, slab, skip_lsbs=0, stream=None):
        """PolyRq.serialize per polynomial: [batch][L][N] -> uint8 tensor [batch][byte count]."""
        import torch

        batch = self._batch(slab)
        out = torch.empty((batch, self.serialization_byte_count(skip_lsbs)), dtype=torch.uint8, device=slab.device)
        _check(load_library().he_poly_serialize_device(self.h, _ptr(slab), batch, skip_lsbs, vp(out.data_ptr()),
                                                       _stream(stream)))
        return out

    def deserialize(self, data, skip_lsbs=0, stream=None):
        """PolyRq(deserialize:context:skipLSBs:) per record: uint8 tensor [batch][bytes] -> [batch][L][N]."""
        import torch

        batch, per = data.shape[0], data.shape[1]
        out = torch.empty((batch, len(self.moduli), self.degree), dtype=torch.int64, device=data.device)
        _check(load_library().he_poly_deserialize_device(self.h, vp(data.data_ptr()), per, batch, skip_lsbs,
                                                         vp(out.data_ptr()), _stream(stream)))
        return out

    def apply_galois(self, slab, element, eval_format=False, stream=None):
        """PolyRq.applyGalois(element:) on [batch][L][N]; returns a new slab (the permutation is out of place)."""
        import torch

        out = torch.empty_like(slab)
        _check(load_library().he_poly_apply_galois_device(self.h, _ptr(slab), vp(out.data_ptr()), self._batch(slab),
                                                          int(element), int(bool(eval_format)), _stream(stream)))
        return out

    def multiply_power_of_x(self, slab, power, stream=None):
        """PolyRq<Coeff>.multiplyPowerOfX(power); returns a new slab."""
        import torch

        out = torch.empty_like(slab)
        _check(load_library().he_poly_multiply_power_of_x_device(self.h, _ptr(slab), vp(out.data_ptr()),
                                                                 self._batch(slab), int(power), _stream(stream)))
        return out

    def adding_lazy_product_(self, lhs, rhs, acc, stream=None):
        _check(load_library().he_poly_adding_lazy_product_device(self.h, _ptr(lhs), _ptr(rhs), _ptr(acc),
                                                                 _stream(stream)))
        return acc

    def reduce_accumulator(self, acc, stream=None):
        import torch

        out = torch.empty((len(self.moduli), self.degree), dtype=torch.int64, device=acc.device)
        _check(load_library().he_poly_reduce_accumulator_device(self.h, _ptr(acc), vp(out.data_ptr()), _stream(stream)))
        return out


class BfvContext:
    """Context<Bfv<UInt64>> (reference Context.swift:94-159) plus the Bfv operations on the hot path."""

    def __init__(self, degree, plaintext_modulus, coefficient_moduli, host_only=False, word_bits=64):
        """word_bits=32: Context<Bfv<UInt32>> constants on 8-byte words (he_bfv_context_create_u32)."""
        lib = load_library()
        arr = _u64(list(coefficient_moduli))
        h = vp()
        create = lib.he_bfv_context_create_host_only if host_only else lib.he_bfv_context_create
        if word_bits == 32:
            create = lib.he_bfv_context_create_u32
        _check(create(degree, plaintext_modulus, arr.ctypes.data_as(U64P), len(arr), ctypes.byref(h)))
        self.h = h
        self.degree = degree
        self.t = plaintext_modulus
        self.coefficient_moduli = [int(v) for v in arr]
        self.L = int(lib.he_bfv_ciphertext_moduli_count(self.h))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.he_bfv_context_destroy(self.h)
            self.h = None

    def _L(self, moduli_count):
        return self.L if moduli_count is None else moduli_count

    def _sub(self, getter, moduli_count):
        handle = getter(self.h, self._L(moduli_count))
        if not handle:
            raise HeError(16, "moduli_count out of range")
        return PolyContext(None, None, _borrowed=handle, _keepalive=self)

    def ciphertext_context(self, moduli_count=None):
        return self._sub(load_library().he_bfv_ciphertext_context, moduli_count)

    def key_switching_context(self, moduli_count=None):
        return self._sub(load_library().he_bfv_key_switching_context, moduli_count)

    def qbsk_context(self, moduli_count=None):
        return self._sub(load_library().he_bfv_qbsk_context, moduli_count)

    def bsk_moduli(self):
        out = np.zeros(self.L + 1, dtype=np.uint64)
        _check(load_library().he_bfv_copy_bsk_moduli(self.h, out.ctypes.data_as(U64P)))
        return [int(v) for v in out]

    def _empty(self, shape, like):
        import torch

        return torch.empty(shape, dtype=torch.int64, device=like.device)

    def lift_q_to_qbsk(self, polys, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = polys.numel() // (L * self.degree)
        out = self._empty((batch, 2 * L + 1, self.degree), polys)
        _check(load_library().he_rns_lift_q_to_qbsk_device(self.h, L, _ptr(polys), _ptr(out), batch, _stream(stream)))
        return out

    def floor_qbsk_to_q(self, polys, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = polys.numel() // ((2 * L + 1) * self.degree)
        out = self._empty((batch, L, self.degree), polys)
        _check(load_library().he_rns_floor_qbsk_to_q_device(self.h, L, _ptr(polys), _ptr(out), batch, _stream(stream)))
        return out

    def mul(self, lhs, rhs, moduli_count=None, stream=None, workspace=None):
        """Bfv.mulAssign(ct, ct): [batch][2][L][N] x [batch][2][L][N] -> [batch][3][L][N] (Coeff)."""
        L = self._L(moduli_count)
        batch = lhs.numel() // (2 * L * self.degree)
        out = self._empty((batch, 3, L, self.degree), lhs)
        ws_ptr, ws_bytes = (vp(workspace.data_ptr()), workspace.numel() * workspace.element_size()) if workspace is not None else (vp(), 0)
        _check(load_library().he_bfv_mul_device(self.h, L, _ptr(lhs), _ptr(rhs), _ptr(out), batch, ws_ptr, ws_bytes,
                                                _stream(stream)))
        return out

    def mul_workspace_bytes(self, batch, moduli_count=None):
        return int(load_library().he_bfv_mul_workspace_bytes(self.h, self._L(moduli_count), batch))

    def relinearize_workspace_bytes(self, batch, moduli_count=None):
        return int(load_library().he_bfv_relinearize_workspace_bytes(self.h, self._L(moduli_count), batch))

    def relinearize(self, ct3, key, moduli_count=None, stream=None, workspace=None):
        """Bfv.relinearize: [batch][3][L][N] + key [L_top][2][L_top+1][N] -> [batch][2][L][N]."""
        L = self._L(moduli_count)
        batch = ct3.numel() // (3 * L * self.degree)
        out = self._empty((batch, 2, L, self.degree), ct3)
        key_ptr = vp() if key is None else _ptr(key)
        ws_ptr, ws_bytes = (vp(workspace.data_ptr()), workspace.numel() * workspace.element_size()) if workspace is not None else (vp(), 0)
        _check(load_library().he_bfv_relinearize_device(self.h, L, _ptr(ct3), key_ptr, _ptr(out), batch, ws_ptr,
                                                        ws_bytes, _stream(stream)))
        return out

    def apply_galois_workspace_bytes(self, batch, moduli_count=None):
        return int(load_library().he_bfv_apply_galois_workspace_bytes(self.h, self._L(moduli_count), batch))

    def apply_galois(self, ct, element, key, moduli_count=None, stream=None, workspace=None):
        """Bfv.applyGalois: [batch][2][L][N] Coeff + the element's Galois key -> [batch][2][L][N]."""
        L = self._L(moduli_count)
        batch = ct.numel() // (2 * L * self.degree)
        out = self._empty((batch, 2, L, self.degree), ct)
        key_ptr = vp() if key is None else _ptr(key)
        ws_ptr, ws_bytes = (vp(workspace.data_ptr()), workspace.numel() * workspace.element_size()) if workspace is not None else (vp(), 0)
        _check(load_library().he_bfv_apply_galois_device(self.h, L, _ptr(ct), int(element), key_ptr, _ptr(out), batch,
                                                         ws_ptr, ws_bytes, _stream(stream)))
        return out

    def scale_and_round(self, poly, scaling_factor=1, moduli_count=None, stream=None):
        """_RnsTool.scaleAndRound: [batch][L][N] Coeff -> [batch][N] mod t."""
        L = self._L(moduli_count)
        batch = poly.numel() // (L * self.degree)
        out = self._empty((batch, self.degree), poly)
        _check(load_library().he_rns_scale_and_round_device(self.h, L, _ptr(poly), int(scaling_factor), _ptr(out),
                                                            batch, _stream(stream)))
        return out

    def plaintext_to_eval(self, plaintext, moduli_count=None, stream=None):
        """Plaintext.convertToEvalFormat: [batch][N] (values < t) -> [batch][L][N] Eval."""
        L = self._L(moduli_count)
        batch = plaintext.numel() // self.degree
        out = self._empty((batch, L, self.degree), plaintext)
        _check(load_library().he_bfv_plaintext_to_eval_device(self.h, L, _ptr(plaintext), _ptr(out), batch,
                                                              _stream(stream)))
        return out

    def plaintext_to_coeff(self, plaintext_eval, moduli_count=None, stream=None):
        """Plaintext.convertToCoeffFormat: [batch][L][N] Eval -> [batch][N] (values < t)."""
        L = self._L(moduli_count)
        batch = plaintext_eval.numel() // (L * self.degree)
        out = self._empty((batch, self.degree), plaintext_eval)
        _check(load_library().he_bfv_plaintext_to_coeff_device(self.h, L, _ptr(plaintext_eval), _ptr(out), batch,
                                                               _stream(stream)))
        return out

    def pir_expand(self, ciphertexts, output_count, galois_keys, stream=None):
        """PirUtil.expand: [count][2][L][N] Coeff + {element: key tensor} -> [output_count][2][L][N]."""
        count = ciphertexts.numel() // (2 * self.L * self.degree)
        out = self._empty((output_count, 2, self.L, self.degree), ciphertexts)
        elements = sorted(galois_keys)
        element_array = _u64(elements)
        key_array = (vp * max(len(elements), 1))(*[vp(galois_keys[e].data_ptr()) for e in elements])
        _check(load_library().he_pir_expand_device(self.h, _ptr(ciphertexts), count, output_count,
                                                   element_array.ctypes.data_as(U64P), key_array, len(elements),
                                                   _ptr(out), _stream(stream)))
        return out

    def pir_expand_batch(self, ciphertexts, output_count, galois_keys_per_query, stream=None):
        """`queries` expansions of one shape: ciphertexts [queries][count][2][L][N]; galois_keys_per_query: one
        {element: key tensor} dict per query (the same elements in each) -> [queries][output_count][2][L][N]."""
        queries = len(galois_keys_per_query)
        count = ciphertexts.numel() // (queries * 2 * self.L * self.degree)
        out = self._empty((queries, output_count, 2, self.L, self.degree), ciphertexts)
        elements = sorted(galois_keys_per_query[0])
        element_array = _u64(elements)
        pointers = [vp(keys[e].data_ptr()) for keys in galois_keys_per_query for e in elements]
        key_array = (vp * max(len(pointers), 1))(*pointers)
        _check(load_library().he_pir_expand_batch_device(self.h, _ptr(ciphertexts), queries, count, output_count,
                                                         element_array.ctypes.data_as(U64P), key_array, len(elements),
                                                         _ptr(out), _stream(stream)))
        return out

    def pir_compute_response_chunk(self, dimensions, dim0_query_eval, remaining_query, database, present=None,
                                   relinearization_key=None, stream=None):
        """PirUtilProtocol.computeResponseForOneChunk -> response ciphertext [2][1][N] (Coeff, one modulus)."""
        dims = (c_u32 * len(dimensions))(*[int(d) for d in dimensions])
        out = self._empty((2, 1, self.degree), dim0_query_eval)
        pres = None
        if present is not None:
            pres_arr = np.ascontiguousarray(present, dtype=np.uint8)
            pres = pres_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        rest = vp() if remaining_query is None else _ptr(remaining_query)
        rest_count = 0 if remaining_query is None else remaining_query.numel() // (2 * self.L * self.degree)
        key = vp() if relinearization_key is None else _ptr(relinearization_key)
        _check(load_library().he_pir_compute_response_chunk_device(self.h, dims, len(dimensions),
                                                                   _ptr(dim0_query_eval), rest, rest_count,
                                                                   _ptr(database), pres, key, _ptr(out),
                                                                   _stream(stream)))
        return out

    def pir_dim0_columns(self, dim0_query_eval, database, present_device=None, stream=None):
        """PirUtil.swift:428-446 for a column shard: database [columns][d0][L][N] Eval -> [columns][2][L][N] Coeff.
        present_device: uint8 device tensor [columns][d0] or None.  Enqueue-only."""
        d0 = dim0_query_eval.numel() // (2 * self.L * self.degree)
        columns = database.numel() // (d0 * self.L * self.degree)
        out = self._empty((columns, 2, self.L, self.degree), dim0_query_eval)
        mask = vp() if present_device is None else vp(present_device.data_ptr())
        _check(load_library().he_pir_dim0_columns_device(self.h, _ptr(dim0_query_eval), d0, _ptr(database), mask,
                                                         columns, _ptr(out), _stream(stream)))
        return out

    def pir_remaining_dimensions(self, dimensions, intermediate, remaining_query, relinearization_key=None, stream=None):
        """PirUtil.swift:448-485 on all columns' dim-0 results ([columns][2][L][N] Coeff, consumed) -> [2][1][N]."""
        dims = (c_u32 * len(dimensions))(*[int(d) for d in dimensions])
        out = self._empty((2, 1, self.degree), intermediate)
        rest = vp() if remaining_query is None else _ptr(remaining_query)
        rest_count = 0 if remaining_query is None else remaining_query.numel() // (2 * self.L * self.degree)
        key = vp() if relinearization_key is None else _ptr(relinearization_key)
        _check(load_library().he_pir_remaining_dimensions_device(self.h, dims, len(dimensions), _ptr(intermediate), rest,
                                                                 rest_count, key, _ptr(out), _stream(stream)))
        return out

    def pir_compute_response(self, dimensions, dim0_query_eval, remaining_query, database, chunk_count,
                             present_device=None, relinearization_key=None, stream=None):
        """PirUtil.computeResponse's chunk loop for one query: database [chunks][prod(dims)][L][N] -> [chunks][2][1][N]."""
        dims = (c_u32 * len(dimensions))(*[int(d) for d in dimensions])
        out = self._empty((chunk_count, 2, 1, self.degree), dim0_query_eval)
        rest = vp() if remaining_query is None else _ptr(remaining_query)
        rest_count = 0 if remaining_query is None else remaining_query.numel() // (2 * self.L * self.degree)
        key = vp() if relinearization_key is None else _ptr(relinearization_key)
        mask = vp() if present_device is None else vp(present_device.data_ptr())
        _check(load_library().he_pir_compute_response_device(self.h, dims, len(dimensions), _ptr(dim0_query_eval), rest,
                                                             rest_count, _ptr(database), mask, chunk_count, key,
                                                             _ptr(out), _stream(stream)))
        return out

    def pir_compute_response_to_query(self, dimensions, query_ciphertexts, indices_count, galois_keys, relinearization_key,
                                      databases, chunk_count, present_devices=None, stream=None):
        """PirUtil.computeResponse: Query.ciphertexts [count][2][L][N] Coeff + the evaluation key ({element: Galois key
        tensor}, relinearization key tensor or None) + one database tensor (shared by all indices) or a list of one per
        index (present_devices likewise: None, one mask or a list) -> [indices][chunks][2][1][N]."""
        dims = (c_u32 * len(dimensions))(*[int(d) for d in dimensions])
        count = query_ciphertexts.numel() // (2 * self.L * self.degree)
        out = self._empty((indices_count, chunk_count, 2, 1, self.degree), query_ciphertexts)
        elements = sorted(galois_keys)
        element_array = _u64(elements)
        key_array = (vp * max(len(elements), 1))(*[vp(galois_keys[e].data_ptr()) for e in elements])
        relin = vp() if relinearization_key is None else _ptr(relinearization_key)
        database_list = list(databases) if isinstance(databases, (list, tuple)) else [databases]
        database_array = (vp * len(database_list))(*[vp(d.data_ptr()) for d in database_list])
        mask_array = None
        if present_devices is not None:
            mask_list = list(present_devices) if isinstance(present_devices, (list, tuple)) else [present_devices]
            mask_array = (vp * len(mask_list))(*[vp() if m is None else vp(m.data_ptr()) for m in mask_list])
        _check(load_library().he_pir_compute_response_to_query_device(
            self.h, dims, len(dimensions), _ptr(query_ciphertexts), count, indices_count,
            element_array.ctypes.data_as(U64P), key_array, len(elements), relin, database_array, mask_array,
            len(database_list), chunk_count, _ptr(out), _stream(stream)))
        return out

    def pir_compute_response_queries(self, dimensions, dim0_queries_eval, remaining_queries, database, chunk_count,
                                     relinearization_keys, present_device=None, stream=None):
        """`queries` queries over one database in one call: dim0_queries_eval [d0][queries][2][L][N] Eval,
        remaining_queries [queries][rest][2][L][N] (or None), relinearization_keys: one tensor per query (or None)
        -> [queries][chunks][2][1][N]."""
        dims = (c_u32 * len(dimensions))(*[int(d) for d in dimensions])
        queries = dim0_queries_eval.numel() // (int(dimensions[0]) * 2 * self.L * self.degree)
        out = self._empty((queries, chunk_count, 2, 1, self.degree), dim0_queries_eval)
        rest = vp() if remaining_queries is None else _ptr(remaining_queries)
        rest_count = 0 if remaining_queries is None else remaining_queries.numel() // (queries * 2 * self.L * self.degree)
        keys = None
        if relinearization_keys is not None:
            keys = (vp * queries)(*[vp(k.data_ptr()) for k in relinearization_keys])
        mask = vp() if present_device is None else vp(present_device.data_ptr())
        _check(load_library().he_pir_compute_response_queries_device(
            self.h, dims, len(dimensions), queries, _ptr(dim0_queries_eval), rest, rest_count, _ptr(database), mask,
            chunk_count, keys, _ptr(out), _stream(stream)))
        return out

    def mod_switch_down(self, ct, poly_count, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = ct.numel() // (poly_count * L * self.degree)
        out = self._empty((batch, poly_count, L - 1, self.degree), ct)
        _check(load_library().he_bfv_mod_switch_down_device(self.h, L, poly_count, _ptr(ct), _ptr(out), batch,
                                                            _stream(stream)))
        return out

    def mod_switch_down_to_single(self, ct, poly_count, moduli_count=None, stream=None):
        """Ciphertext.modSwitchDownToSingle: [batch][polys][L][N] -> [batch][polys][1][N]."""
        L = self._L(moduli_count)
        batch = ct.numel() // (poly_count * L * self.degree)
        out = self._empty((batch, poly_count, 1, self.degree), ct)
        _check(load_library().he_bfv_mod_switch_down_to_single_device(self.h, L, poly_count, _ptr(ct), _ptr(out), batch,
                                                                      _stream(stream)))
        return out

    def mul_plain_(self, ct, pt, poly_count, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = pt.numel() // (L * self.degree)
        _check(load_library().he_bfv_mul_plain_device(self.h, L, poly_count, _ptr(ct), _ptr(pt), batch,
                                                      _stream(stream)))
        return ct

    def add_plain_(self, ct, plaintexts, poly_count=2, subtract=False, moduli_count=None, stream=None):
        """Bfv.addAssignCoeff / subAssignCoeff(ciphertext, plaintext): ct [batch][polys][L][N] Coeff in place,
        plaintexts [batch][N] mod t."""
        L = self._L(moduli_count)
        batch = plaintexts.numel() // self.degree
        fn = load_library().he_bfv_sub_plain_device if subtract else load_library().he_bfv_add_plain_device
        _check(fn(self.h, L, poly_count, _ptr(ct), _ptr(plaintexts), batch, _stream(stream)))
        return ct

    def inner_product_plain(self, cts, pts, present=None, poly_count=2, columns=1, moduli_count=None, stream=None):
        """cts [count][polys][L][N]; pts [columns][count][L][N]; present: host bytes [columns][count] or None."""
        L = self._L(moduli_count)
        count = cts.numel() // (poly_count * L * self.degree)
        out = self._empty((columns, poly_count, L, self.degree), cts)
        pres = None
        if present is not None:
            pres_arr = np.ascontiguousarray(present, dtype=np.uint8)
            pres = pres_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        _check(load_library().he_bfv_inner_product_plain_device(self.h, L, poly_count, _ptr(cts), _ptr(pts), pres,
                                                                count, columns, _ptr(out), _stream(stream)))
        return out

    def inner_product_plain_resident(self, cts, pts, present_device=None, poly_count=2, columns=1, moduli_count=None,
                                     stream=None):
        """inner_product_plain with the nil-plaintext mask as a uint8 DEVICE tensor [columns][count]: enqueue-only."""
        L = self._L(moduli_count)
        count = cts.numel() // (poly_count * L * self.degree)
        out = self._empty((columns, poly_count, L, self.degree), cts)
        mask = vp() if present_device is None else vp(present_device.data_ptr())
        _check(load_library().he_bfv_inner_product_plain_resident_device(self.h, L, poly_count, _ptr(cts), _ptr(pts),
                                                                         mask, count, columns, _ptr(out),
                                                                         _stream(stream)))
        return out

    def packed_plaintext_words(self, moduli_count=None):
        return int(load_library().he_bfv_packed_plaintext_words(self.h, self._L(moduli_count)))

    def pack_plaintexts(self, plaintexts_eval, moduli_count=None, stream=None):
        """[...][L][N] Eval plaintexts -> [count * words + 1] packed words (one word of padding: the kernels read 8 bytes
        past the last field)."""
        import torch

        L = self._L(moduli_count)
        count = plaintexts_eval.numel() // (L * self.degree)
        words = self.packed_plaintext_words(L)
        out = torch.zeros(count * words + 1, dtype=torch.int64, device=plaintexts_eval.device)
        _check(load_library().he_bfv_pack_plaintexts_device(self.h, L, _ptr(plaintexts_eval), count, _ptr(out),
                                                            _stream(stream)))
        return out

    def inner_product_plain_packed(self, cts, packed_pts, present_device=None, poly_count=2, columns=1, moduli_count=None,
                                   stream=None):
        L = self._L(moduli_count)
        count = cts.numel() // (poly_count * L * self.degree)
        out = self._empty((columns, poly_count, L, self.degree), cts)
        mask = vp() if present_device is None else vp(present_device.data_ptr())
        _check(load_library().he_bfv_inner_product_plain_packed_device(self.h, L, poly_count, _ptr(cts), _ptr(packed_pts),
                                                                       mask, count, columns, _ptr(out), _stream(stream)))
        return out

    def pir_compute_response_packed(self, dimensions, dim0_query_eval, remaining_query, packed_database, chunk_count,
                                    present_device=None, relinearization_key=None, stream=None):
        dims = (c_u32 * len(dimensions))(*[int(d) for d in dimensions])
        out = self._empty((chunk_count, 2, 1, self.degree), dim0_query_eval)
        rest = vp() if remaining_query is None else _ptr(remaining_query)
        rest_count = 0 if remaining_query is None else remaining_query.numel() // (2 * self.L * self.degree)
        key = vp() if relinearization_key is None else _ptr(relinearization_key)
        mask = vp() if present_device is None else vp(present_device.data_ptr())
        _check(load_library().he_pir_compute_response_packed_device(self.h, dims, len(dimensions), _ptr(dim0_query_eval),
                                                                    rest, rest_count, _ptr(packed_database), mask,
                                                                    chunk_count, key, _ptr(out), _stream(stream)))
        return out

    def inner_product(self, lhs, rhs, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        count = lhs.numel() // (2 * L * self.degree)
        out = self._empty((3, L, self.degree), lhs)
        _check(load_library().he_bfv_inner_product_device(self.h, L, _ptr(lhs), _ptr(rhs), count, _ptr(out), vp(), 0,
                                                          _stream(stream)))
        return out

    def inner_product_shared(self, lhs, rhs, moduli_count=None, stream=None):
        """rhs [items][count][2][L][N]: `items` inner products with the same left vector -> [items][3][L][N]."""
        L = self._L(moduli_count)
        count = lhs.numel() // (2 * L * self.degree)
        items = rhs.numel() // (count * 2 * L * self.degree)
        out = self._empty((items, 3, L, self.degree), lhs)
        _check(load_library().he_bfv_inner_product_shared_device(self.h, L, _ptr(lhs), _ptr(rhs), count, items, _ptr(out),
                                                                 _stream(stream)))
        return out


class BfvContext32(BfvContext):
    """Context<Bfv<UInt32>> on PACKED [UInt32] slabs (int32 tensors): the `_u32` entry points of include/he_amd.h.
    Same shapes as BfvContext's methods; nothing is widened in memory."""

    def __init__(self, degree, plaintext_modulus, coefficient_moduli, host_only=False):
        super().__init__(degree, plaintext_modulus, coefficient_moduli, host_only=host_only, word_bits=32)

    def _empty32(self, shape, like):
        import torch

        return torch.empty(shape, dtype=torch.int32, device=like.device)

    @staticmethod
    def _ws(workspace):
        return (vp(workspace.data_ptr()), workspace.numel() * workspace.element_size()) if workspace is not None else (vp(), 0)

    def lift_q_to_qbsk(self, polys, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = polys.numel() // (L * self.degree)
        out = self._empty32((batch, 2 * L + 1, self.degree), polys)
        _check(load_library().he_rns_lift_q_to_qbsk_device_u32(self.h, L, _ptr32(polys), _ptr32(out), batch, _stream(stream)))
        return out

    def floor_qbsk_to_q(self, polys, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = polys.numel() // ((2 * L + 1) * self.degree)
        out = self._empty32((batch, L, self.degree), polys)
        _check(load_library().he_rns_floor_qbsk_to_q_device_u32(self.h, L, _ptr32(polys), _ptr32(out), batch, _stream(stream)))
        return out

    def scale_and_round(self, poly, scaling_factor=1, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = poly.numel() // (L * self.degree)
        out = self._empty32((batch, self.degree), poly)
        _check(load_library().he_rns_scale_and_round_device_u32(self.h, L, _ptr32(poly), int(scaling_factor), _ptr32(out),
                                                                batch, _stream(stream)))
        return out

    def mul(self, lhs, rhs, moduli_count=None, stream=None, workspace=None):
        L = self._L(moduli_count)
        batch = lhs.numel() // (2 * L * self.degree)
        out = self._empty32((batch, 3, L, self.degree), lhs)
        ws_ptr, ws_bytes = self._ws(workspace)
        _check(load_library().he_bfv_mul_device_u32(self.h, L, _ptr32(lhs), _ptr32(rhs), _ptr32(out), batch, ws_ptr,
                                                    ws_bytes, _stream(stream)))
        return out

    def relinearize(self, ct3, key, moduli_count=None, stream=None, workspace=None):
        L = self._L(moduli_count)
        batch = ct3.numel() // (3 * L * self.degree)
        out = self._empty32((batch, 2, L, self.degree), ct3)
        ws_ptr, ws_bytes = self._ws(workspace)
        _check(load_library().he_bfv_relinearize_device_u32(self.h, L, _ptr32(ct3), _ptr32(key), _ptr32(out), batch,
                                                            ws_ptr, ws_bytes, _stream(stream)))
        return out

    def apply_galois(self, ct, element, key, moduli_count=None, stream=None, workspace=None):
        L = self._L(moduli_count)
        batch = ct.numel() // (2 * L * self.degree)
        out = self._empty32((batch, 2, L, self.degree), ct)
        ws_ptr, ws_bytes = self._ws(workspace)
        _check(load_library().he_bfv_apply_galois_device_u32(self.h, L, _ptr32(ct), int(element), _ptr32(key), _ptr32(out),
                                                             batch, ws_ptr, ws_bytes, _stream(stream)))
        return out

    def mod_switch_down(self, ct, poly_count, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = ct.numel() // (poly_count * L * self.degree)
        out = self._empty32((batch, poly_count, L - 1, self.degree), ct)
        _check(load_library().he_bfv_mod_switch_down_device_u32(self.h, L, poly_count, _ptr32(ct), _ptr32(out), batch,
                                                                _stream(stream)))
        return out

    def mul_plain_(self, ct, pt, poly_count, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = pt.numel() // (L * self.degree)
        _check(load_library().he_bfv_mul_plain_device_u32(self.h, L, poly_count, _ptr32(ct), _ptr32(pt), batch,
                                                          _stream(stream)))
        return ct

    def add_plain_(self, ct, plaintexts, poly_count=2, subtract=False, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = plaintexts.numel() // self.degree
        fn = load_library().he_bfv_sub_plain_device_u32 if subtract else load_library().he_bfv_add_plain_device_u32
        _check(fn(self.h, L, poly_count, _ptr32(ct), _ptr32(plaintexts), batch, _stream(stream)))
        return ct

    def inner_product_plain_resident(self, cts, pts, present_device=None, poly_count=2, columns=1, moduli_count=None,
                                     stream=None):
        L = self._L(moduli_count)
        count = cts.numel() // (poly_count * L * self.degree)
        out = self._empty32((columns, poly_count, L, self.degree), cts)
        mask = vp() if present_device is None else vp(present_device.data_ptr())
        _check(load_library().he_bfv_inner_product_plain_resident_device_u32(self.h, L, poly_count, _ptr32(cts),
                                                                             _ptr32(pts), mask, count, columns,
                                                                             _ptr32(out), _stream(stream)))
        return out

    def inner_product(self, lhs, rhs, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        count = lhs.numel() // (2 * L * self.degree)
        out = self._empty32((3, L, self.degree), lhs)
        _check(load_library().he_bfv_inner_product_device_u32(self.h, L, _ptr32(lhs), _ptr32(rhs), count, _ptr32(out), vp(),
                                                              0, _stream(stream)))
        return out

    def pir_compute_response(self, dimensions, dim0_query_eval, remaining_query, database, chunk_count,
                             present_device=None, relinearization_key=None, stream=None):
        """he_pir_compute_response_device_u32: everything in packed UInt32 words -> [chunks][2][1][N] (int32)."""
        dims = (c_u32 * len(dimensions))(*[int(d) for d in dimensions])
        out = self._empty32((chunk_count, 2, 1, self.degree), dim0_query_eval)
        rest = vp() if remaining_query is None else _ptr32(remaining_query)
        rest_count = 0 if remaining_query is None else remaining_query.numel() // (2 * self.L * self.degree)
        key = vp() if relinearization_key is None else _ptr32(relinearization_key)
        mask = vp() if present_device is None else vp(present_device.data_ptr())
        _check(load_library().he_pir_compute_response_device_u32(self.h, dims, len(dimensions), _ptr32(dim0_query_eval),
                                                                 rest, rest_count, _ptr32(database), mask, chunk_count,
                                                                 key, _ptr32(out), _stream(stream)))
        return out

    def pir_compute_response_queries(self, dimensions, dim0_queries_eval, remaining_queries, database, chunk_count,
                                     relinearization_keys, present_device=None, stream=None):
        """he_pir_compute_response_queries_device_u32: dim0_queries_eval [d0][queries][2][L][N] Eval, remaining_queries
        [queries][rest][2][L][N] (or None), one relinearization key per query (or None) -> [queries][chunks][2][1][N]."""
        dims = (c_u32 * len(dimensions))(*[int(d) for d in dimensions])
        queries = dim0_queries_eval.numel() // (int(dimensions[0]) * 2 * self.L * self.degree)
        out = self._empty32((queries, chunk_count, 2, 1, self.degree), dim0_queries_eval)
        rest = vp() if remaining_queries is None else _ptr32(remaining_queries)
        rest_count = 0 if remaining_queries is None else remaining_queries.numel() // (queries * 2 * self.L * self.degree)
        keys = None
        if relinearization_keys is not None:
            keys = (vp * queries)(*[vp(k.data_ptr()) for k in relinearization_keys])
        mask = vp() if present_device is None else vp(present_device.data_ptr())
        _check(load_library().he_pir_compute_response_queries_device_u32(
            self.h, dims, len(dimensions), queries, _ptr32(dim0_queries_eval), rest, rest_count, _ptr32(database), mask,
            chunk_count, keys, _ptr32(out), _stream(stream)))
        return out

    def pir_compute_response_to_query(self, dimensions, query_ciphertexts, indices_count, galois_keys_wide,
                                      relinearization_key, databases, chunk_count, present_devices=None, stream=None):
        """he_pir_compute_response_to_query_device_u32: int32 query / relinearization key / databases, Galois keys as
        int64 (widened) tensors -> [indices][chunks][2][1][N] (int32)."""
        dims = (c_u32 * len(dimensions))(*[int(d) for d in dimensions])
        count = query_ciphertexts.numel() // (2 * self.L * self.degree)
        out = self._empty32((indices_count, chunk_count, 2, 1, self.degree), query_ciphertexts)
        elements = sorted(galois_keys_wide)
        element_array = _u64(elements)
        key_array = (vp * max(len(elements), 1))(*[vp(galois_keys_wide[e].data_ptr()) for e in elements])
        relin = vp() if relinearization_key is None else _ptr32(relinearization_key)
        database_list = list(databases) if isinstance(databases, (list, tuple)) else [databases]
        database_array = (vp * len(database_list))(*[vp(d.data_ptr()) for d in database_list])
        mask_array = None
        if present_devices is not None:
            mask_list = list(present_devices) if isinstance(present_devices, (list, tuple)) else [present_devices]
            mask_array = (vp * len(mask_list))(*[vp() if m is None else vp(m.data_ptr()) for m in mask_list])
        _check(load_library().he_pir_compute_response_to_query_device_u32(
            self.h, dims, len(dimensions), _ptr32(query_ciphertexts), count, indices_count,
            element_array.ctypes.data_as(U64P), key_array, len(elements), relin, database_array, mask_array,
            len(database_list), chunk_count, _ptr32(out), _stream(stream)))
        return out

    def plaintext_to_eval(self, plaintext, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = plaintext.numel() // self.degree
        out = self._empty32((batch, L, self.degree), plaintext)
        _check(load_library().he_bfv_plaintext_to_eval_device_u32(self.h, L, _ptr32(plaintext), _ptr32(out), batch,
                                                                  _stream(stream)))
        return out

    def plaintext_to_coeff(self, plaintext_eval, moduli_count=None, stream=None):
        L = self._L(moduli_count)
        batch = plaintext_eval.numel() // (L * self.degree)
        out = self._empty32((batch, self.degree), plaintext_eval)
        _check(load_library().he_bfv_plaintext_to_coeff_device_u32(self.h, L, _ptr32(plaintext_eval), _ptr32(out), batch,
                                                                   _stream(stream)))
        return out
