"""ctypes binding for the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See oracle/he_oracle.h."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

U64P = ctypes.POINTER(ctypes.c_uint64)
c_u64 = ctypes.c_uint64
c_size = ctypes.c_size_t

ERROR_NAMES = {
    0: "ok", 1: "invalidDegree", 2: "invalidModulus", 3: "coprimeModuli", 4: "emptyModulus",
    5: "invalidNttModulus", 6: "invalidPolyContext", 7: "polyContextMismatch", 8: "invalidCiphertext",
    9: "incompatibleCiphertexts", 10: "incompatibleCiphertextAndPlaintext", 11: "missingRelinearizationKey",
    12: "unequalContexts", 13: "notEnoughPrimes", 14: "notInvertible", 15: "invalidEncryptionParameters",
    16: "invalidArgument",
}


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error {code} ({ERROR_NAMES.get(code, '?')})")
        self.code = code
        self.name = ERROR_NAMES.get(code, "?")


def build(force=False):
    """Compile oracle/liboracle.so with gcc (test infrastructure build, see oracle/Makefile)."""
    src = os.path.join(_HERE, "he_oracle.c")
    hdr = os.path.join(_HERE, "he_oracle.h")
    if not force and os.path.exists(_LIB_PATH):
        newest = max(os.path.getmtime(src), os.path.getmtime(hdr))
        if os.path.getmtime(_LIB_PATH) >= newest:
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    vp = ctypes.c_void_p
    L.orc_pow_mod.restype = c_u64
    L.orc_pow_mod.argtypes = [c_u64, c_u64, c_u64]
    L.orc_is_prime.argtypes = [c_u64]
    L.orc_generate_primes.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, c_u64,
                                      ctypes.c_int, U64P]
    L.orc_inverse_mod.argtypes = [c_u64, c_u64, U64P]
    L.orc_reverse_bits.restype = ctypes.c_uint32
    L.orc_reverse_bits.argtypes = [ctypes.c_uint32, ctypes.c_int]
    L.orc_is_primitive_root_of_unity.argtypes = [c_u64, c_u64, c_u64]
    L.orc_min_primitive_root_of_unity.restype = c_u64
    L.orc_min_primitive_root_of_unity.argtypes = [c_u64, c_u64]
    for name, n in (("orc_barrett_reduce_u64", 2), ("orc_barrett_reduce_u128", 3), ("orc_barrett_reduce_product", 3),
                    ("orc_shoup_factor", 2), ("orc_shoup_multiply_mod_lazy", 3), ("orc_shoup_multiply_mod", 3)):
        getattr(L, name).restype = c_u64
        getattr(L, name).argtypes = [c_u64] * n
    L.orc_poly_context_create.argtypes = [c_u64, U64P, c_size, ctypes.POINTER(vp)]
    L.orc_poly_context_destroy.argtypes = [vp]
    L.orc_poly_context_destroy.restype = None
    L.orc_poly_context_degree.restype = c_u64
    L.orc_poly_context_degree.argtypes = [vp]
    L.orc_poly_context_moduli_count.restype = c_size
    L.orc_poly_context_moduli_count.argtypes = [vp]
    L.orc_poly_context_moduli.argtypes = [vp, U64P]
    L.orc_poly_context_moduli.restype = None
    L.orc_poly_context_max_lazy_product_accumulation_count.restype = c_u64
    L.orc_poly_context_max_lazy_product_accumulation_count.argtypes = [vp, ctypes.c_int]
    L.orc_poly_context_q_remainder.restype = c_u64
    L.orc_poly_context_q_remainder.argtypes = [vp, c_u64]
    L.orc_poly_context_ntt_tables.argtypes = [vp, c_size, U64P, U64P, U64P, U64P, U64P, U64P]
    for name in ("orc_forward_ntt", "orc_inverse_ntt", "orc_poly_neg"):
        getattr(L, name).argtypes = [vp, U64P, c_size]
    for name in ("orc_forward_ntt_mt", "orc_inverse_ntt_mt"):
        getattr(L, name).argtypes = [vp, U64P, c_size, ctypes.c_int]
    for name in ("orc_poly_add", "orc_poly_sub", "orc_poly_mul", "orc_poly_mul_scalar"):
        getattr(L, name).argtypes = [vp, U64P, U64P, c_size]
    L.orc_poly_divide_and_round_q_last.argtypes = [vp, U64P, U64P, c_size]
    L.orc_poly_divide_and_round_q_last_mt.argtypes = [vp, U64P, U64P, c_size, ctypes.c_int]
    L.orc_poly_adding_lazy_product.argtypes = [vp, U64P, U64P, U64P]
    L.orc_poly_reduce_accumulator.argtypes = [vp, U64P, U64P]
    L.orc_rns_tool_create.argtypes = [vp, c_u64, ctypes.POINTER(vp)]
    L.orc_rns_tool_create_word.argtypes = [vp, c_u64, ctypes.c_int, ctypes.POINTER(vp)]
    L.orc_rns_tool_destroy.argtypes = [vp]
    L.orc_rns_tool_destroy.restype = None
    L.orc_rns_tool_bsk_count.restype = c_size
    L.orc_rns_tool_bsk_count.argtypes = [vp]
    L.orc_rns_tool_bsk_moduli.argtypes = [vp, U64P]
    L.orc_rns_tool_bsk_moduli.restype = None
    for name in ("orc_rns_convert_approximate_bsk_mtilde", "orc_rns_lift_q_to_qbsk", "orc_rns_approximate_floor",
                 "orc_rns_convert_approximate_bsk_to_q", "orc_rns_floor_qbsk_to_q"):
        getattr(L, name).argtypes = [vp, U64P, U64P]
    L.orc_rns_small_montgomery_reduce.argtypes = [vp, U64P]
    L.orc_rns_convert_approximate.argtypes = [vp, vp, U64P, U64P]
    L.orc_rns_scale_and_round.argtypes = [vp, U64P, c_u64, U64P]
    L.orc_bfv_context_create.argtypes = [c_u64, c_u64, U64P, c_size, ctypes.POINTER(vp)]
    L.orc_bfv_context_create_word.argtypes = [c_u64, c_u64, U64P, c_size, ctypes.c_int, ctypes.POINTER(vp)]
    L.orc_bfv_context_destroy.argtypes = [vp]
    L.orc_bfv_context_destroy.restype = None
    L.orc_bfv_ciphertext_moduli_count.restype = c_size
    L.orc_bfv_ciphertext_moduli_count.argtypes = [vp]
    for name in ("orc_bfv_ciphertext_context", "orc_bfv_key_switching_context", "orc_bfv_qbsk_context",
                 "orc_bfv_rns_tool"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp, c_size]
    L.orc_bfv_mul.argtypes = [vp, c_size, U64P, U64P, U64P, c_size]
    L.orc_bfv_mul_mt.argtypes = [vp, c_size, U64P, U64P, U64P, c_size, ctypes.c_int]
    L.orc_bfv_relinearize.argtypes = [vp, c_size, U64P, U64P, U64P, c_size]
    L.orc_bfv_relinearize_mt.argtypes = [vp, c_size, U64P, U64P, U64P, c_size, ctypes.c_int]
    L.orc_bfv_key_switching_update.argtypes = [vp, c_size, U64P, U64P, U64P]
    L.orc_bfv_mod_switch_down.argtypes = [vp, c_size, c_size, U64P, U64P, c_size]
    L.orc_bfv_mul_plain.argtypes = [vp, c_size, c_size, U64P, U64P, c_size]
    L.orc_bfv_plaintext_translate.argtypes = [vp, c_size, c_size, U64P, U64P, ctypes.c_int, c_size]
    L.orc_bfv_inner_product_plain.argtypes = [vp, c_size, c_size, U64P, U64P, ctypes.POINTER(ctypes.c_uint8), c_size,
                                              U64P]
    L.orc_bfv_inner_product.argtypes = [vp, c_size, U64P, U64P, c_size, U64P]
    U8P = ctypes.POINTER(ctypes.c_uint8)
    L.orc_coefficients_to_bytes_byte_count.restype = c_size
    L.orc_coefficients_to_bytes_byte_count.argtypes = [c_size, ctypes.c_int, ctypes.c_int]
    L.orc_bytes_to_coefficients_coeff_count.restype = c_size
    L.orc_bytes_to_coefficients_coeff_count.argtypes = [c_size, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.orc_coefficients_to_bytes.argtypes = [U64P, c_size, ctypes.c_int, ctypes.c_int, U8P, c_size]
    L.orc_bytes_to_coefficients.argtypes = [U8P, c_size, ctypes.c_int, ctypes.c_int, U64P, c_size]
    L.orc_poly_serialization_byte_count.restype = c_size
    L.orc_poly_serialization_byte_count.argtypes = [vp, ctypes.c_int]
    L.orc_poly_serialize.argtypes = [vp, U64P, ctypes.c_int, U8P]
    L.orc_poly_deserialize.argtypes = [vp, U8P, c_size, ctypes.c_int, U64P]
    L.orc_ctr_drbg_create.argtypes = [U8P, ctypes.POINTER(vp)]
    L.orc_ctr_drbg_destroy.argtypes = [vp]
    L.orc_ctr_drbg_destroy.restype = None
    L.orc_ctr_drbg_state.argtypes = [vp, U8P, U8P]
    L.orc_ctr_drbg_state.restype = None
    L.orc_ctr_drbg_generate.argtypes = [vp, U8P, c_size]
    L.orc_poly_random_from_seed.argtypes = [vp, U8P, U64P]
    L.orc_is_valid_galois_element.argtypes = [c_u64, c_u64]
    L.orc_poly_apply_galois_coeff.argtypes = [vp, U64P, U64P, c_u64, c_size]
    L.orc_poly_apply_galois_eval.argtypes = [vp, U64P, U64P, c_u64, c_size]
    L.orc_poly_multiply_power_of_x.argtypes = [vp, U64P, ctypes.c_int64, c_size]
    L.orc_bfv_apply_galois.argtypes = [vp, c_size, U64P, c_u64, U64P, U64P, c_size]
    L.orc_bfv_plaintext_to_eval.argtypes = [vp, c_size, U64P, U64P, c_size]
    L.orc_bfv_plaintext_to_coeff.argtypes = [vp, c_size, U64P, U64P, c_size]


def _check(code):
    if code != 0:
        raise OracleError(code)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def _p(a):
    return a.ctypes.data_as(U64P)


# ---------------------------------------------------------------- wire format
def _u8p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def coefficients_to_bytes(coeffs, bits_per_coeff, skip_lsbs=0):
    """CoefficientPacking.coefficientsToBytes (CoefficientPacking.swift:155-166)."""
    coeffs = _u64(list(coeffs))
    count = int(lib().orc_coefficients_to_bytes_byte_count(len(coeffs), bits_per_coeff, skip_lsbs))
    out = np.zeros(count, dtype=np.uint8)
    _check(lib().orc_coefficients_to_bytes(_p(coeffs), len(coeffs), bits_per_coeff, skip_lsbs, _u8p(out), count))
    return out


def bytes_to_coefficients(data, bits_per_coeff, decode, skip_lsbs=0):
    """CoefficientPacking.bytesToCoefficients (CoefficientPacking.swift:59-72)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    count = int(lib().orc_bytes_to_coefficients_coeff_count(len(data), bits_per_coeff, int(decode), skip_lsbs))
    out = np.zeros(count, dtype=np.uint64)
    _check(lib().orc_bytes_to_coefficients(_u8p(data), len(data), bits_per_coeff, skip_lsbs, _p(out), count))
    return out


class CtrDrbg:
    """NistCtrDrbg (Random/NistCtrDrbg.swift)."""

    def __init__(self, entropy):
        e = np.frombuffer(bytes(entropy), dtype=np.uint8).copy()
        assert len(e) == 32
        h = ctypes.c_void_p()
        _check(lib().orc_ctr_drbg_create(_u8p(e), ctypes.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ctr_drbg_destroy(self.h)
            self.h = None

    def state(self):
        key, nonce = np.zeros(16, dtype=np.uint8), np.zeros(16, dtype=np.uint8)
        lib().orc_ctr_drbg_state(self.h, _u8p(key), _u8p(nonce))
        return bytes(key), bytes(nonce)

    def generate(self, count):
        out = np.zeros(count, dtype=np.uint8)
        _check(lib().orc_ctr_drbg_generate(self.h, _u8p(out), count))
        return bytes(out)


# ---------------------------------------------------------------- scalar layer
def pow_mod(b, e, p):
    return int(lib().orc_pow_mod(b, e, p))


def is_prime(n):
    return bool(lib().orc_is_prime(n))


def generate_primes(bit_counts, preferring_small, ntt_degree=1, word_bits=64):
    bits = (ctypes.c_int * len(bit_counts))(*bit_counts)
    out = np.zeros(len(bit_counts), dtype=np.uint64)
    _check(lib().orc_generate_primes(bits, len(bit_counts), int(preferring_small), ntt_degree, word_bits, _p(out)))
    return [int(x) for x in out]


def inverse_mod(x, p):
    out = c_u64(0)
    _check(lib().orc_inverse_mod(x, p, ctypes.byref(out)))
    return int(out.value)


def is_primitive_root_of_unity(root, degree, modulus):
    return bool(lib().orc_is_primitive_root_of_unity(root, degree, modulus))


def min_primitive_root_of_unity(modulus, degree):
    return int(lib().orc_min_primitive_root_of_unity(modulus, degree))


def reverse_bits(x, bit_count):
    return int(lib().orc_reverse_bits(x, bit_count))


def barrett_reduce_u64(p, x):
    return int(lib().orc_barrett_reduce_u64(p, x))


def barrett_reduce_u128(p, x):
    return int(lib().orc_barrett_reduce_u128(p, x >> 64, x & ((1 << 64) - 1)))


def barrett_reduce_product(p, x, y):
    return int(lib().orc_barrett_reduce_product(p, x, y))


def shoup_factor(c, p):
    return int(lib().orc_shoup_factor(c, p))


def shoup_multiply_mod_lazy(c, p, x):
    return int(lib().orc_shoup_multiply_mod_lazy(c, p, x))


def shoup_multiply_mod(c, p, x):
    return int(lib().orc_shoup_multiply_mod(c, p, x))


# ---------------------------------------------------------------- PolyContext
class PolyContext:
    """Mirror of PolyContext<UInt64> (HomomorphicEncryption/PolyRq/PolyContext.swift)."""

    def __init__(self, degree, moduli, _borrowed=None):
        self._owned = _borrowed is None
        if _borrowed is not None:
            self.h = ctypes.c_void_p(_borrowed)
        else:
            arr = _u64(list(moduli))
            h = ctypes.c_void_p()
            _check(lib().orc_poly_context_create(degree, _p(arr), len(arr), ctypes.byref(h)))
            self.h = h
        self.degree = int(lib().orc_poly_context_degree(self.h))
        count = int(lib().orc_poly_context_moduli_count(self.h))
        out = np.zeros(count, dtype=np.uint64)
        lib().orc_poly_context_moduli(self.h, _p(out))
        self.moduli = [int(x) for x in out]

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "h", None):
            lib().orc_poly_context_destroy(self.h)
            self.h = None

    @property
    def shape(self):
        return (len(self.moduli), self.degree)

    def max_lazy_product_accumulation_count(self, word_bits=64):
        return int(lib().orc_poly_context_max_lazy_product_accumulation_count(self.h, word_bits))

    def q_remainder(self, modulus):
        return int(lib().orc_poly_context_q_remainder(self.h, modulus))

    def ntt_tables(self, rns_index):
        n = self.degree
        arrs = [np.zeros(n, dtype=np.uint64) for _ in range(4)]
        inv_n, inv_n_root = c_u64(0), c_u64(0)
        _check(lib().orc_poly_context_ntt_tables(self.h, rns_index, *[_p(a) for a in arrs], ctypes.byref(inv_n),
                                                 ctypes.byref(inv_n_root)))
        return dict(root_powers=arrs[0], root_factors=arrs[1], inv_root_powers=arrs[2], inv_root_factors=arrs[3],
                    inverse_degree=int(inv_n.value), inverse_degree_root=int(inv_n_root.value))

    def _batch(self, data):
        L, n = self.shape
        assert data.size % (L * n) == 0, "slab is not [batch][L][N]"
        return data.size // (L * n)

    def forward_ntt(self, data, threads=1):
        out = _u64(data).copy()
        _check(lib().orc_forward_ntt_mt(self.h, _p(out), self._batch(out), threads))
        return out

    def inverse_ntt(self, data, threads=1):
        out = _u64(data).copy()
        _check(lib().orc_inverse_ntt_mt(self.h, _p(out), self._batch(out), threads))
        return out

    def forward_ntt_inplace(self, data, threads=1):
        _check(lib().orc_forward_ntt_mt(self.h, _p(data), self._batch(data), threads))

    def inverse_ntt_inplace(self, data, threads=1):
        _check(lib().orc_inverse_ntt_mt(self.h, _p(data), self._batch(data), threads))

    def _binary(self, fn, lhs, rhs):
        out = _u64(lhs).copy()
        rhs = _u64(rhs)
        assert out.shape == rhs.shape
        _check(fn(self.h, _p(out), _p(rhs), self._batch(out)))
        return out

    def add(self, lhs, rhs):
        return self._binary(lib().orc_poly_add, lhs, rhs)

    def sub(self, lhs, rhs):
        return self._binary(lib().orc_poly_sub, lhs, rhs)

    def mul(self, lhs, rhs):
        return self._binary(lib().orc_poly_mul, lhs, rhs)

    def neg(self, data):
        out = _u64(data).copy()
        _check(lib().orc_poly_neg(self.h, _p(out), self._batch(out)))
        return out

    def mul_scalar(self, data, scalar_residues):
        out = _u64(data).copy()
        s = _u64(list(scalar_residues))
        assert len(s) == len(self.moduli)
        _check(lib().orc_poly_mul_scalar(self.h, _p(out), _p(s), self._batch(out)))
        return out

    def random_from_seeds(self, seeds):
        """PolyRq.random(context:using: NistAes128Ctr(seed:)) per 32-byte seed: uint8 [batch][32] -> [batch][L][N]."""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint8).reshape(-1, 32)
        L, n = self.shape
        out = np.zeros((len(seeds), L, n), dtype=np.uint64)
        for b, seed in enumerate(seeds):
            tmp = np.zeros(L * n, dtype=np.uint64)
            _check(lib().orc_poly_random_from_seed(self.h, _u8p(np.ascontiguousarray(seed)), _p(tmp)))
            out[b] = tmp.reshape(L, n)
        return out

    def serialization_byte_count(self, skip_lsbs=0):
        return int(lib().orc_poly_serialization_byte_count(self.h, skip_lsbs))

    def serialize(self, data, skip_lsbs=0):
        """PolyRq.serialize (PolyRq+Serialize.swift:69-87) per polynomial: [batch][L][N] -> uint8 [batch][bytes]."""
        data = _u64(data)
        batch, per = self._batch(data), self.serialization_byte_count(skip_lsbs)
        L, n = self.shape
        out = np.zeros((batch, per), dtype=np.uint8)
        flat = data.reshape(batch, L * n)
        for b in range(batch):
            _check(lib().orc_poly_serialize(self.h, _p(np.ascontiguousarray(flat[b])), skip_lsbs, _u8p(out[b])))
        return out

    def deserialize(self, data, skip_lsbs=0):
        """PolyRq(deserialize:context:skipLSBs:) per polynomial: uint8 [batch][bytes] -> [batch][L][N]."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        data = data.reshape(-1, data.shape[-1])
        L, n = self.shape
        out = np.zeros((data.shape[0], L, n), dtype=np.uint64)
        for b in range(data.shape[0]):
            row = np.ascontiguousarray(data[b])
            tmp = np.zeros(L * n, dtype=np.uint64)
            _check(lib().orc_poly_deserialize(self.h, _u8p(row), len(row), skip_lsbs, _p(tmp)))
            out[b] = tmp.reshape(L, n)
        return out

    def apply_galois(self, data, element, eval_format=False):
        """PolyRq.applyGalois (PolyRq/Galois.swift:115-168): f(x) -> f(x^element) in Coeff or Eval format."""
        data = _u64(data)
        out = np.zeros_like(data)
        fn = lib().orc_poly_apply_galois_eval if eval_format else lib().orc_poly_apply_galois_coeff
        _check(fn(self.h, _p(data), _p(out), element, self._batch(data)))
        return out

    def multiply_power_of_x(self, data, power):
        """PolyRq<Coeff>.multiplyPowerOfX (PolyRq/PolyRq.swift:398-422)."""
        out = _u64(data).copy()
        _check(lib().orc_poly_multiply_power_of_x(self.h, _p(out), power, self._batch(out)))
        return out

    def divide_and_round_q_last(self, data, threads=1):
        L, n = self.shape
        data = _u64(data)
        batch = self._batch(data)
        out = np.zeros((batch, max(L - 1, 0), n), dtype=np.uint64)
        _check(lib().orc_poly_divide_and_round_q_last_mt(self.h, _p(data), _p(out), batch, threads))
        return out

    def adding_lazy_product(self, lhs, rhs, acc):
        """acc: uint64 [L][N][2] (lo, hi); updated in place."""
        _check(lib().orc_poly_adding_lazy_product(self.h, _p(_u64(lhs)), _p(_u64(rhs)), _p(acc)))

    def reduce_accumulator(self, acc):
        L, n = self.shape
        out = np.zeros((L, n), dtype=np.uint64)
        _check(lib().orc_poly_reduce_accumulator(self.h, _p(acc), _p(out)))
        return out


def convert_approximate(input_ctx, output_ctx, data):
    data = _u64(data)
    out = np.zeros((len(output_ctx.moduli), output_ctx.degree), dtype=np.uint64)
    _check(lib().orc_rns_convert_approximate(input_ctx.h, output_ctx.h, _p(data), _p(out)))
    return out


# ---------------------------------------------------------------- RnsTool
class RnsTool:
    """Mirror of _RnsTool<UInt64> (HomomorphicEncryption/RnsTool.swift)."""

    def __init__(self, input_ctx, t, _borrowed=None, word_bits=64):
        self._owned = _borrowed is None
        self.input_ctx = input_ctx
        self.t = t
        if _borrowed is not None:
            self.h = ctypes.c_void_p(_borrowed)
        else:
            h = ctypes.c_void_p()
            _check(lib().orc_rns_tool_create_word(input_ctx.h, t, word_bits, ctypes.byref(h)))
            self.h = h
        count = int(lib().orc_rns_tool_bsk_count(self.h))
        out = np.zeros(count, dtype=np.uint64)
        lib().orc_rns_tool_bsk_moduli(self.h, _p(out))
        self.bsk = [int(x) for x in out]
        self.L = len(input_ctx.moduli)
        self.n = input_ctx.degree

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "h", None):
            lib().orc_rns_tool_destroy(self.h)
            self.h = None

    def _call(self, fn, data, out_rows):
        data = _u64(data)
        out = np.zeros((out_rows, self.n), dtype=np.uint64)
        _check(fn(self.h, _p(data), _p(out)))
        return out

    def convert_approximate_bsk_mtilde(self, data):
        return self._call(lib().orc_rns_convert_approximate_bsk_mtilde, data, self.L + 2)

    def small_montgomery_reduce(self, data):
        buf = _u64(data).copy()
        _check(lib().orc_rns_small_montgomery_reduce(self.h, _p(buf)))
        return buf.reshape(self.L + 2, self.n)[: self.L + 1].copy()

    def lift_q_to_qbsk(self, data):
        return self._call(lib().orc_rns_lift_q_to_qbsk, data, 2 * self.L + 1)

    def approximate_floor(self, data):
        return self._call(lib().orc_rns_approximate_floor, data, self.L + 1)

    def convert_approximate_bsk_to_q(self, data):
        return self._call(lib().orc_rns_convert_approximate_bsk_to_q, data, self.L)

    def floor_qbsk_to_q(self, data):
        return self._call(lib().orc_rns_floor_qbsk_to_q, data, self.L)

    def scale_and_round(self, data, scaling_factor=1):
        data = _u64(data)
        out = np.zeros(self.n, dtype=np.uint64)
        _check(lib().orc_rns_scale_and_round(self.h, _p(data), scaling_factor, _p(out)))
        return out


# ---------------------------------------------------------------- Context<Bfv<UInt64>>
class BfvContext:
    """Mirror of Context<Bfv<UInt64>> (HomomorphicEncryption/Context.swift) + the Bfv ops on the hot path."""

    def __init__(self, degree, plaintext_modulus, coefficient_moduli, word_bits=64):
        """word_bits = 32 builds Context<Bfv<UInt32>>: moduli <= 2^30 - 1, gamma = 2^30 - 20405, mTilde = 2^16 and
        29-bit Bsk primes (ModularArithmetic/Scalar.swift:498-511, RnsTool.swift:30-33)."""
        arr = _u64(list(coefficient_moduli))
        h = ctypes.c_void_p()
        _check(lib().orc_bfv_context_create_word(degree, plaintext_modulus, _p(arr), len(arr), word_bits,
                                                 ctypes.byref(h)))
        self.h = h
        self.word_bits = word_bits
        self.degree = degree
        self.t = plaintext_modulus
        self.coefficient_moduli = [int(x) for x in arr]
        self.L = int(lib().orc_bfv_ciphertext_moduli_count(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_bfv_context_destroy(self.h)
            self.h = None

    def ciphertext_context(self, moduli_count=None):
        k = self.L if moduli_count is None else moduli_count
        ctx = PolyContext(None, None, _borrowed=lib().orc_bfv_ciphertext_context(self.h, k))
        ctx._keepalive = self
        return ctx

    def key_switching_context(self, moduli_count=None):
        k = self.L if moduli_count is None else moduli_count
        ctx = PolyContext(None, None, _borrowed=lib().orc_bfv_key_switching_context(self.h, k))
        ctx._keepalive = self
        return ctx

    def qbsk_context(self, moduli_count=None):
        k = self.L if moduli_count is None else moduli_count
        ctx = PolyContext(None, None, _borrowed=lib().orc_bfv_qbsk_context(self.h, k))
        ctx._keepalive = self
        return ctx

    def rns_tool(self, moduli_count=None):
        k = self.L if moduli_count is None else moduli_count
        tool = RnsTool(self.ciphertext_context(k), self.t, _borrowed=lib().orc_bfv_rns_tool(self.h, k))
        tool._keepalive = self
        return tool

    def _L(self, moduli_count):
        return self.L if moduli_count is None else moduli_count

    def mul(self, lhs, rhs, moduli_count=None, threads=1):
        L = self._L(moduli_count)
        lhs, rhs = _u64(lhs), _u64(rhs)
        per = 2 * L * self.degree
        batch = lhs.size // per
        out = np.zeros((batch, 3, L, self.degree), dtype=np.uint64)
        _check(lib().orc_bfv_mul_mt(self.h, L, _p(lhs), _p(rhs), _p(out), batch, threads))
        return out

    def relinearize(self, ct3, key, moduli_count=None, threads=1):
        L = self._L(moduli_count)
        ct3, key = _u64(ct3), _u64(key)
        batch = ct3.size // (3 * L * self.degree)
        out = np.zeros((batch, 2, L, self.degree), dtype=np.uint64)
        _check(lib().orc_bfv_relinearize_mt(self.h, L, _p(ct3), _p(key), _p(out), batch, threads))
        return out

    def key_switching_update(self, target, key, moduli_count=None):
        L = self._L(moduli_count)
        target, key = _u64(target), _u64(key)
        out = np.zeros((2, L, self.degree), dtype=np.uint64)
        _check(lib().orc_bfv_key_switching_update(self.h, L, _p(target), _p(key), _p(out)))
        return out

    def apply_galois(self, ct, element, key, moduli_count=None):
        """Bfv.applyGalois (Bfv/Bfv.swift:174-198) on [batch][2][L][N] Coeff ciphertexts."""
        L = self._L(moduli_count)
        ct = _u64(ct)
        batch = ct.size // (2 * L * self.degree)
        out = np.zeros((batch, 2, L, self.degree), dtype=np.uint64)
        _check(lib().orc_bfv_apply_galois(self.h, L, _p(ct), element, _p(_u64(key)), _p(out), batch))
        return out

    def plaintext_to_eval(self, plaintext, moduli_count=None):
        """Plaintext.convertToEvalFormat (Plaintext.swift:149-170): [batch][N] mod t -> [batch][L][N]."""
        L = self._L(moduli_count)
        pt = _u64(plaintext)
        batch = pt.size // self.degree
        out = np.zeros((batch, L, self.degree), dtype=np.uint64)
        _check(lib().orc_bfv_plaintext_to_eval(self.h, L, _p(pt), _p(out), batch))
        return out

    def plaintext_to_coeff(self, plaintext_eval, moduli_count=None):
        """Plaintext.convertToCoeffFormat (Plaintext.swift:176-191): [batch][L][N] -> [batch][N] mod t."""
        L = self._L(moduli_count)
        pt = _u64(plaintext_eval)
        batch = pt.size // (L * self.degree)
        out = np.zeros((batch, self.degree), dtype=np.uint64)
        _check(lib().orc_bfv_plaintext_to_coeff(self.h, L, _p(pt), _p(out), batch))
        return out

    def mod_switch_down(self, ct, poly_count, moduli_count=None):
        L = self._L(moduli_count)
        ct = _u64(ct)
        batch = ct.size // (poly_count * L * self.degree)
        out = np.zeros((batch, poly_count, L - 1, self.degree), dtype=np.uint64)
        _check(lib().orc_bfv_mod_switch_down(self.h, L, poly_count, _p(ct), _p(out), batch))
        return out

    def mul_plain(self, ct, pt, poly_count, moduli_count=None):
        L = self._L(moduli_count)
        out = _u64(ct).copy()
        pt = _u64(pt)
        batch = pt.size // (L * self.degree)
        _check(lib().orc_bfv_mul_plain(self.h, L, poly_count, _p(out), _p(pt), batch))
        return out

    def plaintext_translate(self, ct, plaintexts, poly_count=2, subtract=False, moduli_count=None):
        """Bfv.addAssignCoeff / subAssignCoeff(ciphertext, plaintext) (Bfv.swift:110-117, Bfv+Encrypt.swift:75-140):
        ct [batch][polys][L][N] Coeff, plaintexts [batch][N] mod t."""
        L = self._L(moduli_count)
        out = _u64(ct).copy()
        pt = _u64(plaintexts)
        batch = pt.size // self.degree
        _check(lib().orc_bfv_plaintext_translate(self.h, L, poly_count, _p(out), _p(pt), 1 if subtract else 0, batch))
        return out

    def inner_product_plain(self, cts, pts, present=None, poly_count=2, moduli_count=None):
        L = self._L(moduli_count)
        cts, pts = _u64(cts), _u64(pts)
        count = pts.size // (L * self.degree)
        out = np.zeros((poly_count, L, self.degree), dtype=np.uint64)
        pres = None
        if present is not None:
            pres_arr = np.ascontiguousarray(present, dtype=np.uint8)
            pres = pres_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        _check(lib().orc_bfv_inner_product_plain(self.h, L, poly_count, _p(cts), _p(pts), pres, count, _p(out)))
        return out

    def inner_product(self, lhs, rhs, moduli_count=None):
        L = self._L(moduli_count)
        lhs, rhs = _u64(lhs), _u64(rhs)
        count = lhs.size // (2 * L * self.degree)
        out = np.zeros((3, L, self.degree), dtype=np.uint64)
        _check(lib().orc_bfv_inner_product(self.h, L, _p(lhs), _p(rhs), count, _p(out)))
        return out
