/*
 * oracle/he_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the BFV PolyRq/NTT hot path of apple/swift-homomorphic-encryption
 * (SURVEY.md section 8a).  It exists only so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check / time-beside the HIP path.  Nothing under
 * swift-homomorphic-encryption_amd/ may include, link or call it.
 *
 * Parity status: PINNED in-tree.  The restatement is checked (tests/test_oracle_*.py) against every
 * known-answer vector the reference's own tests hold for this path (transcribed into tests/golden/):
 *   Tests/HomomorphicEncryptionTests/NttTests.swift:38-191, ScalarTests.swift:172-210,
 *   PolyRqTests/PolyRqTests.swift:45-176, PolyRqTests/PolyContextTests.swift:106-115,175-191,
 *   RnsToolTests.swift:118-166, plus the deterministic big-int properties of RnsToolTests.swift:168-305.
 * The reference itself (Swift 6.2) cannot be compiled in this image, so there is no oracle/_ref.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/Sources).
 */
#ifndef HE_ORACLE_H
#define HE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes: one per HeError case reachable on the path (HomomorphicEncryption/Error.swift:19-54). */
enum {
    ORC_OK = 0,
    ORC_ERR_INVALID_DEGREE = 1,
    ORC_ERR_INVALID_MODULUS = 2,
    ORC_ERR_COPRIME_MODULI = 3,
    ORC_ERR_EMPTY_MODULUS = 4,
    ORC_ERR_INVALID_NTT_MODULUS = 5,
    ORC_ERR_INVALID_POLY_CONTEXT = 6,
    ORC_ERR_POLY_CONTEXT_MISMATCH = 7,
    ORC_ERR_INVALID_CIPHERTEXT = 8,
    ORC_ERR_INCOMPATIBLE_CIPHERTEXTS = 9,
    ORC_ERR_INCOMPATIBLE_CIPHERTEXT_AND_PLAINTEXT = 10,
    ORC_ERR_MISSING_RELINEARIZATION_KEY = 11,
    ORC_ERR_UNEQUAL_CONTEXTS = 12,
    ORC_ERR_NOT_ENOUGH_PRIMES = 13,
    ORC_ERR_NOT_INVERTIBLE = 14,
    ORC_ERR_INVALID_ENCRYPTION_PARAMETERS = 15,
    ORC_ERR_INVALID_ARGUMENT = 16
};

typedef unsigned __int128 orc_u128;

/* ---------- scalar layer (ModularArithmetic/, HomomorphicEncryption/Scalar.swift) ---------- */
uint64_t orc_pow_mod(uint64_t base, uint64_t exponent, uint64_t modulus);
int orc_is_prime(uint64_t value);
int orc_generate_primes(const int* significant_bit_counts, int count, int preferring_small, uint64_t ntt_degree,
                        int word_bits, uint64_t* out);
int orc_inverse_mod(uint64_t value, uint64_t modulus, uint64_t* out);
uint32_t orc_reverse_bits(uint32_t x, int bit_count);
int orc_is_primitive_root_of_unity(uint64_t root, uint64_t degree, uint64_t modulus);
uint64_t orc_min_primitive_root_of_unity(uint64_t modulus, uint64_t degree); /* 0 = none */

/* Barrett / Shoup forms exactly as the reference computes them (for property tests vs % and /). */
uint64_t orc_barrett_reduce_u64(uint64_t modulus, uint64_t x);
uint64_t orc_barrett_reduce_u128(uint64_t modulus, uint64_t x_hi, uint64_t x_lo);
uint64_t orc_barrett_reduce_product(uint64_t modulus, uint64_t x, uint64_t y);
uint64_t orc_shoup_factor(uint64_t multiplicand, uint64_t modulus);
uint64_t orc_shoup_multiply_mod_lazy(uint64_t multiplicand, uint64_t modulus, uint64_t x);
uint64_t orc_shoup_multiply_mod(uint64_t multiplicand, uint64_t modulus, uint64_t x);

/* ---------- PolyContext (HomomorphicEncryption/PolyRq/PolyContext.swift) ---------- */
typedef struct orc_poly_context orc_poly_context;
int orc_poly_context_create(uint64_t degree, const uint64_t* moduli, size_t moduli_count, orc_poly_context** out);
void orc_poly_context_destroy(orc_poly_context* ctx);
uint64_t orc_poly_context_degree(const orc_poly_context* ctx);
size_t orc_poly_context_moduli_count(const orc_poly_context* ctx);
void orc_poly_context_moduli(const orc_poly_context* ctx, uint64_t* out);
/* word_bits = 32 or 64 selects T.DoubleWidth for the reference's UInt32/UInt64 KATs. */
uint64_t orc_poly_context_max_lazy_product_accumulation_count(const orc_poly_context* ctx, int word_bits);
uint64_t orc_poly_context_q_remainder(const orc_poly_context* ctx, uint64_t modulus);
/* NTT tables of modulus row `rns_index`: each out array has `degree` words (may be NULL). */
int orc_poly_context_ntt_tables(const orc_poly_context* ctx, size_t rns_index, uint64_t* root_powers,
                                uint64_t* root_factors, uint64_t* inv_root_powers, uint64_t* inv_root_factors,
                                uint64_t* inverse_degree, uint64_t* inverse_degree_root);

/* ---------- PolyRq ops on row-major [batch][L][N] slabs (PolyRq.swift, PolyRq+Ntt.swift) ---------- */
int orc_forward_ntt(const orc_poly_context* ctx, uint64_t* data, size_t batch);
int orc_inverse_ntt(const orc_poly_context* ctx, uint64_t* data, size_t batch);
/* Same, but `threads` host threads each take whole polynomials (how the reference parallelises). */
int orc_forward_ntt_mt(const orc_poly_context* ctx, uint64_t* data, size_t batch, int threads);
int orc_inverse_ntt_mt(const orc_poly_context* ctx, uint64_t* data, size_t batch, int threads);
int orc_poly_add(const orc_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch);
int orc_poly_sub(const orc_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch);
int orc_poly_neg(const orc_poly_context* ctx, uint64_t* data, size_t batch);
int orc_poly_mul(const orc_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch);
int orc_poly_mul_scalar(const orc_poly_context* ctx, uint64_t* data, const uint64_t* scalar_residues, size_t batch);
/* in: [batch][L][N] -> out: [batch][L-1][N] */
int orc_poly_divide_and_round_q_last(const orc_poly_context* ctx, const uint64_t* in, uint64_t* out, size_t batch);
int orc_poly_divide_and_round_q_last_mt(const orc_poly_context* ctx, const uint64_t* in, uint64_t* out,
                                        size_t batch, int threads);
/* acc (UInt128 as lo,hi pairs, [L][N][2]) += lhs*rhs, wrapping; then reduce. */
int orc_poly_adding_lazy_product(const orc_poly_context* ctx, const uint64_t* lhs, const uint64_t* rhs,
                                 uint64_t* acc_lo_hi);
int orc_poly_reduce_accumulator(const orc_poly_context* ctx, const uint64_t* acc_lo_hi, uint64_t* out);

/* ---------- RnsTool (RnsTool.swift, RnsBaseConverter.swift, CrtComposer.swift) ---------- */
typedef struct orc_rns_tool orc_rns_tool;
/* Standalone tool, as `_RnsTool(from:to:)` (RnsTool.swift:254-261): Bsk generated for this input context. */
int orc_rns_tool_create(const orc_poly_context* input, uint64_t t, orc_rns_tool** out);
/* the same for T = UInt32 (word_bits 32: gamma 2^30-20405, mTilde 2^16, 29-bit Bsk) or UInt64 (word_bits 64) */
int orc_rns_tool_create_word(const orc_poly_context* input, uint64_t t, int word_bits, orc_rns_tool** out);
void orc_rns_tool_destroy(orc_rns_tool* tool);
size_t orc_rns_tool_bsk_count(const orc_rns_tool* tool);
void orc_rns_tool_bsk_moduli(const orc_rns_tool* tool, uint64_t* out);
/* in [L][N] -> out [L+2][N] over [Bsk, mTilde] */
int orc_rns_convert_approximate_bsk_mtilde(const orc_rns_tool* tool, const uint64_t* in, uint64_t* out);
/* inout [L+2][N] over [Bsk, mTilde]; result in first L+1 rows */
int orc_rns_small_montgomery_reduce(const orc_rns_tool* tool, uint64_t* inout);
/* in [L][N] -> out [2L+1][N] over [Q, Bsk] */
int orc_rns_lift_q_to_qbsk(const orc_rns_tool* tool, const uint64_t* in, uint64_t* out);
/* in [2L+1][N] -> out [L+1][N] over Bsk */
int orc_rns_approximate_floor(const orc_rns_tool* tool, const uint64_t* in, uint64_t* out);
/* in [L+1][N] over Bsk -> out [L][N] over Q */
int orc_rns_convert_approximate_bsk_to_q(const orc_rns_tool* tool, const uint64_t* in, uint64_t* out);
/* in [2L+1][N] -> out [L][N] */
int orc_rns_floor_qbsk_to_q(const orc_rns_tool* tool, const uint64_t* in, uint64_t* out);
/* generic Q->t_j fast base conversion for the set-valued property test */
int orc_rns_convert_approximate(const orc_poly_context* input, const orc_poly_context* output, const uint64_t* in,
                                uint64_t* out);
/* decrypt-side scaleAndRound (RnsTool.swift:272-302): in [L][N] over Q -> out [N] mod t */
int orc_rns_scale_and_round(const orc_rns_tool* tool, const uint64_t* in, uint64_t scaling_factor, uint64_t* out);

/* ---------- BFV context + scheme ops (Context.swift and the files under Bfv/) ---------- */
typedef struct orc_bfv_context orc_bfv_context;
int orc_bfv_context_create(uint64_t degree, uint64_t plaintext_modulus, const uint64_t* coefficient_moduli,
                           size_t moduli_count, orc_bfv_context** out);
int orc_bfv_context_create_word(uint64_t degree, uint64_t plaintext_modulus, const uint64_t* coefficient_moduli,
                                size_t moduli_count, int word_bits, orc_bfv_context** out);
void orc_bfv_context_destroy(orc_bfv_context* ctx);
size_t orc_bfv_ciphertext_moduli_count(const orc_bfv_context* ctx); /* L at top level */
const orc_poly_context* orc_bfv_ciphertext_context(const orc_bfv_context* ctx, size_t moduli_count);
const orc_poly_context* orc_bfv_key_switching_context(const orc_bfv_context* ctx, size_t ciphertext_moduli_count);
const orc_poly_context* orc_bfv_qbsk_context(const orc_bfv_context* ctx, size_t moduli_count);
const orc_rns_tool* orc_bfv_rns_tool(const orc_bfv_context* ctx, size_t moduli_count);

/* Bfv.mulAssign(ct,ct) (Bfv+Multiply.swift:18-85): lhs,rhs [batch][2][L][N] Coeff -> out [batch][3][L][N] Coeff */
int orc_bfv_mul(const orc_bfv_context* ctx, size_t moduli_count, const uint64_t* lhs, const uint64_t* rhs,
                uint64_t* out, size_t batch);
int orc_bfv_mul_mt(const orc_bfv_context* ctx, size_t moduli_count, const uint64_t* lhs, const uint64_t* rhs,
                   uint64_t* out, size_t batch, int threads);
/* Bfv.relinearize (Bfv.swift:201-219): ct3 [batch][3][L][N] Coeff, key [L_top][2][L_top+1][N] Eval
 * -> out [batch][2][L][N] Coeff */
int orc_bfv_relinearize(const orc_bfv_context* ctx, size_t moduli_count, const uint64_t* ct3, const uint64_t* key,
                        uint64_t* out, size_t batch);
int orc_bfv_relinearize_mt(const orc_bfv_context* ctx, size_t moduli_count, const uint64_t* ct3,
                           const uint64_t* key, uint64_t* out, size_t batch, int threads);
/* _computeKeySwitchingUpdate (Bfv+Keys.swift:123-208): target [L][N] Coeff -> update [2][L][N] Coeff */
int orc_bfv_key_switching_update(const orc_bfv_context* ctx, size_t moduli_count, const uint64_t* target,
                                 const uint64_t* key, uint64_t* update);
/* Bfv.modSwitchDown (Bfv.swift:163-171): ct [batch][polys][L][N] -> [batch][polys][L-1][N] */
int orc_bfv_mod_switch_down(const orc_bfv_context* ctx, size_t moduli_count, size_t poly_count, const uint64_t* in,
                            uint64_t* out, size_t batch);
/* Bfv.mulAssign(EvalCiphertext, EvalPlaintext) (Bfv.swift:120-129): ct [batch][polys][L][N] *= pt [batch][L][N] */
int orc_bfv_mul_plain(const orc_bfv_context* ctx, size_t moduli_count, size_t poly_count, uint64_t* ct,
                      const uint64_t* pt, size_t batch);
/* Bfv+Encrypt.swift:75-140 plaintextTranslate: ct [batch][poly_count][L][N] (Coeff) +-= plaintexts [batch][N] (< t) */
int orc_bfv_plaintext_translate(const orc_bfv_context* ctx, size_t moduli_count, size_t poly_count, uint64_t* ct,
                                const uint64_t* plaintexts, int subtract, size_t batch);
/* Bfv.innerProduct(ciphertexts:plaintexts:) (Bfv.swift:476-505): cts [count][polys][L][N], pts [count][L][N],
 * present[count] (0 = nil plaintext) -> out [polys][L][N] */
int orc_bfv_inner_product_plain(const orc_bfv_context* ctx, size_t moduli_count, size_t poly_count,
                                const uint64_t* cts, const uint64_t* pts, const uint8_t* present, size_t count,
                                uint64_t* out);
/* Bfv.innerProduct(ct,ct) (Bfv.swift:315-361): lhs,rhs [count][2][L][N] Coeff -> out [3][L][N] Coeff */
int orc_bfv_inner_product(const orc_bfv_context* ctx, size_t moduli_count, const uint64_t* lhs, const uint64_t* rhs,
                          size_t count, uint64_t* out);


/* ---- "next" rows ---- */
/* isValidGaloisElement (PolyRq/Galois.swift:100-105) */
int orc_is_valid_galois_element(uint64_t element, uint64_t degree);
/* PolyRq<Coeff>.applyGalois (PolyRq/Galois.swift:115-143): f(x) -> f(x^element); in/out [batch][L][N] */
int orc_poly_apply_galois_coeff(const orc_poly_context* ctx, const uint64_t* in, uint64_t* out, uint64_t element,
                                size_t batch);
/* PolyRq<Eval>.applyGalois (PolyRq/Galois.swift:153-168) */
int orc_poly_apply_galois_eval(const orc_poly_context* ctx, const uint64_t* in, uint64_t* out, uint64_t element,
                               size_t batch);
/* PolyRq<Coeff>.multiplyPowerOfX (PolyRq/PolyRq.swift:398-422), in place */
int orc_poly_multiply_power_of_x(const orc_poly_context* ctx, uint64_t* data, int64_t power, size_t batch);
/* Bfv.applyGalois (Bfv/Bfv.swift:174-198): ct [batch][2][L][N] Coeff, key = the Galois key of `element`
 * ([L][2][L_top+1][N] Eval, same layout as the relinearization key) -> out [batch][2][L][N] */
int orc_bfv_apply_galois(const orc_bfv_context* ctx, size_t moduli_count, const uint64_t* ct, uint64_t element,
                         const uint64_t* key, uint64_t* out, size_t batch);
/* Plaintext.convertToEvalFormat (Plaintext.swift:149-170): [batch][N] mod t -> [batch][L][N] Eval */
int orc_bfv_plaintext_to_eval(const orc_bfv_context* ctx, size_t moduli_count, const uint64_t* plaintext,
                              uint64_t* out, size_t batch);
/* Plaintext.convertToCoeffFormat (Plaintext.swift:176-191): [batch][L][N] Eval -> [batch][N] mod t */
int orc_bfv_plaintext_to_coeff(const orc_bfv_context* ctx, size_t moduli_count, const uint64_t* plaintext_eval,
                               uint64_t* out, size_t batch);


/* ---- wire format: CoefficientPacking.swift, PolyRq/PolyRq+Serialize.swift ---- */
size_t orc_coefficients_to_bytes_byte_count(size_t coeff_count, int bits_per_coeff, int skip_lsbs);
size_t orc_bytes_to_coefficients_coeff_count(size_t byte_count, int bits_per_coeff, int decode, int skip_lsbs);
int orc_coefficients_to_bytes(const uint64_t* coeffs, size_t coeff_count, int bits_per_coeff, int skip_lsbs,
                              uint8_t* bytes, size_t bytes_count);
int orc_bytes_to_coefficients(const uint8_t* bytes, size_t byte_count, int bits_per_coeff, int skip_lsbs,
                              uint64_t* coeffs, size_t coeff_count);
size_t orc_poly_serialization_byte_count(const orc_poly_context* ctx, int skip_lsbs);
/* data [L][N] -> bytes[orc_poly_serialization_byte_count] */
int orc_poly_serialize(const orc_poly_context* ctx, const uint64_t* data, int skip_lsbs, uint8_t* bytes);
int orc_poly_deserialize(const orc_poly_context* ctx, const uint8_t* bytes, size_t byte_count, int skip_lsbs,
                         uint64_t* data);


/* ---- seeded polynomials: Random/NistCtrDrbg.swift, PolyRq/PolyRq+Randomize.swift:56-75 ---- */
typedef struct orc_ctr_drbg orc_ctr_drbg;
int orc_ctr_drbg_create(const uint8_t entropy[32], orc_ctr_drbg** out);
void orc_ctr_drbg_destroy(orc_ctr_drbg* drbg);
void orc_ctr_drbg_state(const orc_ctr_drbg* drbg, uint8_t key[16], uint8_t nonce_big_endian[16]);
int orc_ctr_drbg_generate(orc_ctr_drbg* drbg, uint8_t* out, size_t count);
/* PolyRq.random(context:using: NistAes128Ctr(seed: seed)) -> out [L][N] */
int orc_poly_random_from_seed(const orc_poly_context* ctx, const uint8_t seed[32], uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* HE_ORACLE_H */
