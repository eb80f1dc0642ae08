/*
 * he_amd.h -- C ABI of the MI355X-native BFV ciphertext-arithmetic engine (libhe_amd.so).
 *
 * This is the drop-in boundary for the PolyRq/NTT hot path of apple/swift-homomorphic-encryption.  It is meant to
 * be wrapped as a SwiftPM C target next to `CUtil` (reference Package.swift:100-105) and called from
 * `Bfv<UInt64>` / `PolyContext<UInt64>`; INTEGRATION.md shows the Swift binding.  Only plain pointers and sizes
 * cross the boundary.  All citations below are file:line relative to /root/reference/Sources/.
 *
 * Conventions:
 *  - Data layout: a polynomial is the reference's Array2d row-major (moduli x N) slab of UInt64
 *    (HomomorphicEncryption/Array2d.swift:117-119); batches are [batch][L][N]; ciphertexts [batch][polys][L][N].
 *  - Every word handed across the boundary is canonical (< its row's modulus), as PolyRq asserts
 *    (PolyRq/PolyRq.swift:36,85-95); every word handed back is canonical and bit-identical to the reference.
 *  - Errors: functions return an int status, 0 = HE_OK, otherwise one code per reachable HeError case
 *    (HomomorphicEncryption/Error.swift:19-54); the Swift wrapper throws.  Nothing aborts.
 *  - Threading: contexts are immutable after creation and every entry point is re-entrant on a shared context
 *    (the reference calls the path concurrently from many tasks: Bfv/Bfv.swift:270-287).  `*_device` functions
 *    enqueue on the caller's HIP stream and return without synchronising; host-pointer functions block.
 *  - Ownership: host pointers are borrowed for the duration of the call only (the reference's seam is
 *    `withUnsafeMutableBufferPointer`, PolyRq/PolyRq+Ntt.swift:215-218).  Device buffers are owned by the caller
 *    (he_device_malloc / any HIP allocation, e.g. a torch tensor's data_ptr()).
 *  - A context lives on the HIP device that was current when it was created; calls must be made with that device
 *    current (one process per GPU).
 */
#ifndef HE_AMD_H
#define HE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: HeError cases reachable on the path (HomomorphicEncryption/Error.swift:19-54) ---- */
enum he_status {
    HE_OK = 0,
    HE_ERR_INVALID_DEGREE = 1,                          /* invalidDegree            PolyRq/PolyContext.swift:46 */
    HE_ERR_INVALID_MODULUS = 2,                         /* invalidModulus           PolyRq/PolyContext.swift:55,58 */
    HE_ERR_COPRIME_MODULI = 3,                          /* coprimeModuli            PolyRq/PolyContext.swift:65,68 */
    HE_ERR_EMPTY_MODULUS = 4,                           /* emptyModulus             PolyRq/PolyContext.swift:71 */
    HE_ERR_INVALID_NTT_MODULUS = 5,                     /* invalidNttModulus        PolyRq/PolyContext.swift:175-181 */
    HE_ERR_INVALID_POLY_CONTEXT = 6,                    /* invalidPolyContext       PolyRq/PolyRq.swift:366-368 */
    HE_ERR_POLY_CONTEXT_MISMATCH = 7,                   /* polyContextMismatch      PolyRq/PolyRq.swift:332-336 */
    HE_ERR_INVALID_CIPHERTEXT = 8,                      /* invalidCiphertext        Bfv/Bfv+Multiply.swift:32-34,67-72 */
    HE_ERR_INCOMPATIBLE_CIPHERTEXTS = 9,                /* incompatibleCiphertexts  Bfv/Bfv+Multiply.swift:73-75 */
    HE_ERR_INCOMPATIBLE_CIPHERTEXT_AND_PLAINTEXT = 10,  /*                          Bfv/Bfv.swift:122-124 */
    HE_ERR_MISSING_RELINEARIZATION_KEY = 11,            /* missingRelinearizationKey Bfv/Bfv.swift:208-210 */
    HE_ERR_UNEQUAL_CONTEXTS = 12,                       /* unequalContexts          HeScheme.swift:1451-1455 */
    HE_ERR_NOT_ENOUGH_PRIMES = 13,                      /* notEnoughPrimes          Scalar.swift:147-152 */
    HE_ERR_NOT_INVERTIBLE = 14,                         /* notInvertible            Scalar.swift:79-85 */
    HE_ERR_INVALID_ENCRYPTION_PARAMETERS = 15,          /* invalidEncryptionParameters EncryptionParameters.swift:136-166 */
    HE_ERR_INVALID_ARGUMENT = 16,  /* what the reference traps on with `precondition` (null pointer, shape) */
    HE_ERR_DEVICE = 17,            /* HIP runtime failure (no GPU, out of memory, launch error) */
    HE_ERR_UNSUPPORTED = 18,       /* unsupportedHeOperation */
    HE_ERR_MISSING_GALOIS_KEY = 19, /* missingGaloisKey / missingGaloisElement  Bfv/Bfv.swift:184-189 */
    HE_ERR_SERIALIZED_BUFFER_SIZE_MISMATCH = 20, /* serializedBufferSizeMismatch  PolyRq/PolyRq+Serialize.swift:41-51 */
    HE_ERR_INVALID_COEFFICIENT_PACKING = 21      /* invalidCoefficientPacking     CoefficientPacking.swift:27-31 */
};

typedef struct he_poly_context he_poly_context; /* PolyContext<UInt64>   PolyRq/PolyContext.swift:19-35 */
typedef struct he_bfv_context he_bfv_context;   /* Context<Bfv<UInt64>>  Context.swift:19 */
typedef void* he_stream;                        /* hipStream_t; NULL = the default stream */

const char* he_status_string(int status);
/* Thread-local detail of the last failing call on this thread (HIP error text etc.); never NULL. */
const char* he_last_error_message(void);
/* Library version / build target, e.g. "he_amd 0.1 gfx950". */
const char* he_version(void);

/* ---- device plumbing (so a Swift host never has to link HIP) ---- */
int he_device_count(int* out_count);
int he_get_device(int* out_device); /* the calling thread's current HIP device: contexts and buffers belong to the device
                                     * that was current when they were created (HE_ERR_DEVICE on any other) */
int he_set_device(int device);
int he_device_malloc(void** out_ptr, size_t bytes);
int he_device_free(void* ptr);
/* (every destroy / free entry point of this header -- he_device_free, he_host_free, he_stream_destroy, he_event_destroy,
 * he_*_context_destroy, he_device_group_destroy -- may be called at any time from any thread, also while a stream of the
 * process is being captured into a graph: a deinit or finaliser cannot choose its moment, so these calls relax the calling
 * thread's capture mode for their duration instead of invalidating the capture) */
/* Page-locked host memory: a copy between it and the device is truly asynchronous (from pageable memory the runtime
 * stages the bytes before he_memcpy_h2d returns, and a borrowed pageable pointer has to be waited for).  What the Swift
 * host stages ciphertexts, keys and databases in: one copy and no wait per object instead of one per polynomial. */
int he_host_malloc(void** out_ptr, size_t bytes);
int he_host_free(void* ptr);
int he_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes, he_stream stream);
int he_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes, he_stream stream);
int he_stream_synchronize(he_stream stream);
int he_stream_create(he_stream* out);   /* a non-blocking HIP stream */
int he_stream_destroy(he_stream stream);
/* Scratch of the `_device` calls that take no workspace is stream-ordered and the library's own (never the process's
 * default pool).  By default nothing is retained: it comes from a HIP memory pool with release threshold 0 and returns to the
 * driver at the next synchronisation -- and, a property of hipFreeAsync in ROCm 7.2, a call that took scratch returns only
 * once the work enqueued before the previous release of that scratch has finished.  he_set_scratch_cache(bytes) switches the
 * current device to a block cache inside the library that keeps up to `bytes` of released scratch (UINT64_MAX: everything):
 * a stream gets the blocks it released back without any driver call, another stream after an event wait, and the `_device`
 * calls are enqueue-only.  What a server sets once (mapping tens of gigabytes per expansion costs seconds otherwise);
 * he_device_trim_scratch(keep_bytes) hands everything above `keep_bytes` back; he_scratch_cached_bytes reads what is held.
 * While a stream is being captured into a graph its scratch always comes from the HIP pool (allocation nodes). */
int he_set_scratch_cache(uint64_t bytes);
int he_device_trim_scratch(uint64_t keep_bytes);
int he_scratch_cached_bytes(uint64_t* out_bytes);

/* ---- completion primitives: what the reference's `...Async` twins (HomomorphicEncryption/HeSchemeAsync.swift:16-141)
 * await.  A Swift `async` wrapper enqueues the `_device` call and then either suspends in a continuation resumed by
 * he_stream_add_callback, or records an event and polls / waits on it -- it never parks a thread of the cooperative
 * pool in he_stream_synchronize.  INTEGRATION.md section 7 and swift/Sources/HeAmd/HeAmdStream.swift show the wrapper. */
typedef void* he_event; /* hipEvent_t (timing disabled) */
typedef void (*he_host_callback)(void* user_data);
int he_event_create(he_event* out);
int he_event_destroy(he_event event);
int he_event_record(he_event event, he_stream stream);       /* marks everything enqueued on `stream` so far */
int he_event_synchronize(he_event event);                    /* blocks the calling thread */
int he_event_query(he_event event, int* out_done);           /* *out_done = 1 when the recorded work has finished */
int he_stream_wait_event(he_stream stream, he_event event);  /* later work on `stream` waits for the event (fork/join) */
/* Runs callback(user_data) on a runtime thread once everything enqueued on `stream` before this call has finished.
 * The callback must not call into this library's device entry points. */
int he_stream_add_callback(he_stream stream, he_host_callback callback, void* user_data);

/* =====================================================================================================
 * B1/B2: PolyContext and PolyRq operations
 * =================================================================================================== */

/* PolyContext.init(degree:moduli:) incl. its validation and error order (PolyRq/PolyContext.swift:45-141).
 * Builds, on the current device, the per-modulus NTT tables (_NttContext.init, PolyRq/PolyRq+Ntt.swift:118-169),
 * Barrett constants (Modulus.swift:24-45) and inverseQLast (PolyRq/PolyContext.swift:108-111). */
int he_poly_context_create(uint32_t degree, const uint64_t* moduli, uint32_t moduli_count, he_poly_context** out);
void he_poly_context_destroy(he_poly_context* ctx);
uint32_t he_poly_context_degree(const he_poly_context* ctx);
uint32_t he_poly_context_moduli_count(const he_poly_context* ctx);
int he_poly_context_moduli(const he_poly_context* ctx, uint64_t* out_moduli);
/* PolyContext.maxLazyProductAccumulationCount() (PolyRq/PolyContext.swift:246-253) */
uint64_t he_poly_context_max_lazy_product_accumulation_count(const he_poly_context* ctx);
/* PolyContext.qRemainder(dividingBy:) (PolyRq/PolyContext.swift:184-191) */
int he_poly_context_q_remainder(const he_poly_context* ctx, uint64_t modulus, uint64_t* out);
/* ScalarType.generatePrimes (Scalar.swift:113-154), for callers that build parameters host-side. */
int he_generate_primes(const int32_t* significant_bit_counts, uint32_t count, int preferring_small,
                       uint32_t ntt_degree, uint64_t* out_primes);

/* PolyRq<UInt64,Coeff>.forwardNtt() / PolyRq<UInt64,Eval>.inverseNtt() over a batch of polynomials
 * (PolyRq/PolyRq+Ntt.swift:209-232,524-543).  In place.  Throws invalidNttModulus like validateNttModuli. */
int he_ntt_forward(const he_poly_context* ctx, uint64_t* host_slab, size_t batch);
int he_ntt_inverse(const he_poly_context* ctx, uint64_t* host_slab, size_t batch);
int he_ntt_forward_device(const he_poly_context* ctx, uint64_t* device_slab, size_t batch, he_stream stream);
int he_ntt_inverse_device(const he_poly_context* ctx, uint64_t* device_slab, size_t batch, he_stream stream);
/* PolyContext.forwardNtt(dataPtr:modulus:) -- the reference's existing raw-pointer seam
 * (PolyRq/PolyRq+Ntt.swift:329-347): `rows` contiguous length-N rows, all transformed mod `modulus`, which must be
 * one of the context's moduli (else invalidPolyContext). */
int he_ntt_forward_rows_device(const he_poly_context* ctx, uint64_t modulus, uint64_t* device_rows, size_t rows,
                               he_stream stream);
int he_ntt_inverse_rows_device(const he_poly_context* ctx, uint64_t modulus, uint64_t* device_rows, size_t rows,
                               he_stream stream);

/* PolyRq += / -= / prefix - / *= (Eval) / *= [T]  (PolyRq/PolyRq.swift:147-174,299-309,184-204,232-245).
 * lhs/data are updated in place; slabs are [batch][L][N]. */
int he_poly_add_device(const he_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch, he_stream s);
int he_poly_sub_device(const he_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch, he_stream s);
int he_poly_neg_device(const he_poly_context* ctx, uint64_t* data, size_t batch, he_stream s);
int he_poly_mul_device(const he_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch, he_stream s);
/* scalar_residues: HOST array of L residues (y mod q_i) */
int he_poly_mul_scalar_device(const he_poly_context* ctx, uint64_t* data, const uint64_t* scalar_residues,
                              size_t batch, he_stream s);
/* PolyRq<_,Coeff>.divideAndRoundQLast() (PolyRq/PolyRq.swift:365-393): in [batch][L][N] -> out [batch][L-1][N].
 * invalidPolyContext when the context has no next (L == 1). */
int he_poly_divide_and_round_q_last_device(const he_poly_context* ctx, const uint64_t* in, uint64_t* out,
                                           size_t batch, he_stream s);
int he_poly_divide_and_round_q_last(const he_poly_context* ctx, const uint64_t* host_in, uint64_t* host_out,
                                    size_t batch);
/* PolyRq.addingLazyProduct (PolyRq/PolyRq.swift:210-225): acc[k] &+= lhs[k]*rhs[k] in wrapping UInt128.
 * acc is [L][N] UInt128 stored little-endian as (lo, hi) word pairs.  he_poly_reduce_accumulator_device is
 * Bfv.reduceToCiphertext's per-polynomial step (Bfv/Bfv.swift:380-394). */
int he_poly_adding_lazy_product_device(const he_poly_context* ctx, const uint64_t* lhs, const uint64_t* rhs,
                                       uint64_t* acc_lo_hi, he_stream s);
int he_poly_reduce_accumulator_device(const he_poly_context* ctx, const uint64_t* acc_lo_hi, uint64_t* out,
                                      he_stream s);

/* ---- "next" rows of the scope table: index permutations; in and out must not alias ---- */
/* PolyRq.applyGalois(element:) f(x) -> f(x^element) on [batch][L][N] slabs; eval_format = 0 for Coeff
 * (PolyRq/Galois.swift:115-143), 1 for Eval (PolyRq/Galois.swift:153-168).  An invalid element (even, <= 1,
 * >= 2N; Galois.swift:100-105) is the reference's precondition -> HE_ERR_INVALID_ARGUMENT. */
int he_poly_apply_galois_device(const he_poly_context* ctx, const uint64_t* in, uint64_t* out, size_t batch,
                                uint64_t element, int eval_format, he_stream s);
/* PolyRq<Coeff>.multiplyPowerOfX(power) (PolyRq/PolyRq.swift:398-422): f(x) x^power mod (x^N + 1), any sign. */
int he_poly_multiply_power_of_x_device(const he_poly_context* ctx, const uint64_t* in, uint64_t* out, size_t batch,
                                       int64_t power, he_stream s);

/* ---- wire format: PolyRq.serialize / PolyRq(deserialize:) on device buffers ----
 * Each residue row is a big-endian bit stream of N coefficients, ceilLog2(q_r) - skip_lsbs bits each, zero-padded
 * to a byte; rows follow one another (CoefficientPacking.swift:169-213, PolyRq/PolyRq+Serialize.swift:69-99).
 * he_poly_serialization_byte_count = PolyContext.serializationByteCount(skipLSBs:) (0 on invalid arguments).
 * device_bytes is [batch][byte count] (serialize) / [batch][bytes_per_poly] (deserialize; the leading byte count
 * of each record is read, a shorter record is serializedBufferSizeMismatch). */
size_t he_poly_serialization_byte_count(const he_poly_context* ctx, int skip_lsbs);
int he_poly_serialize_device(const he_poly_context* ctx, const uint64_t* device_slab, size_t batch, int skip_lsbs,
                             uint8_t* device_bytes, he_stream s);
int he_poly_deserialize_device(const he_poly_context* ctx, const uint8_t* device_bytes, size_t bytes_per_poly,
                               size_t batch, int skip_lsbs, uint64_t* device_slab, he_stream s);

/* The second polynomial of a seeded ciphertext (SerializedCiphertext.swift:53-58, Bfv/Bfv+Encrypt.swift:155-156):
 * device_slab[b] = PolyRq<UInt64, Eval>.random(context:using: NistAes128Ctr(seed: device_seeds[b])), seeds are
 * [batch][32] bytes (NistCtrDrbg.SeedCount).  NIST SP 800-90A CTR_DRBG/AES-128 as Random/NistCtrDrbg.swift has it,
 * 4096-byte refills (Random/NistAes128Ctr.swift), 128 stream bits per coefficient reduced mod q_i
 * (PolyRq/PolyRq+Randomize.swift:56-75).  A Coeff ciphertext then takes he_ntt_inverse_device of the result. */
int he_poly_random_from_seeds_device(const he_poly_context* ctx, const uint8_t* device_seeds, size_t batch,
                                     uint64_t* device_slab, he_stream s);

/* ---- PolyRq<UInt32> (polynomial layer): the same operations on slabs of 4-byte words ----
 * The context is an ordinary he_poly_context whose moduli all fit UInt32 (<= 2^30 - 1, ModularArithmetic/
 * Scalar.swift:498-511; e.g. the n_4096_logq_27_28_28 parameter sets, EncryptionParameters.swift:313-378); a larger
 * modulus returns HE_ERR_INVALID_MODULUS.  Layout [batch][L][N] of uint32_t; semantics as the UInt64 entry points
 * (the reference's generic code is the same for both word types).  Degrees above 32768: HE_ERR_UNSUPPORTED. */
int he_ntt_forward_device_u32(const he_poly_context* ctx, uint32_t* device_slab, size_t batch, he_stream s);
int he_ntt_inverse_device_u32(const he_poly_context* ctx, uint32_t* device_slab, size_t batch, he_stream s);
int he_poly_add_device_u32(const he_poly_context* ctx, uint32_t* lhs, const uint32_t* rhs, size_t batch, he_stream s);
int he_poly_sub_device_u32(const he_poly_context* ctx, uint32_t* lhs, const uint32_t* rhs, size_t batch, he_stream s);
int he_poly_neg_device_u32(const he_poly_context* ctx, uint32_t* data, size_t batch, he_stream s);
int he_poly_mul_device_u32(const he_poly_context* ctx, uint32_t* lhs, const uint32_t* rhs, size_t batch, he_stream s);
int he_poly_mul_scalar_device_u32(const he_poly_context* ctx, uint32_t* data, const uint32_t* scalar_residues,
                                  size_t batch, he_stream s);
int he_poly_divide_and_round_q_last_device_u32(const he_poly_context* ctx, const uint32_t* device_in,
                                               uint32_t* device_out, size_t batch, he_stream s);

/* =====================================================================================================
 * B3: Context<Bfv<UInt64>> and the HeScheme operations on the hot path
 * =================================================================================================== */

/* Context.init(encryptionParameters:) with EncryptionParameters' own checks at securityLevel .unchecked
 * (Context.swift:94-143, EncryptionParameters.swift:136-166).  The last coefficient modulus is the key-switching
 * modulus when more than one is given.  Builds the ciphertext / key-switching PolyContexts and one _RnsTool per
 * level (RnsTool.swift:132-251) incl. the Bsk primes (RnsTool.swift:28-45). */
int he_bfv_context_create(uint32_t degree, uint64_t plaintext_modulus, const uint64_t* coefficient_moduli,
                          uint32_t moduli_count, he_bfv_context** out);
void he_bfv_context_destroy(he_bfv_context* ctx);
uint32_t he_bfv_ciphertext_moduli_count(const he_bfv_context* ctx); /* L of a fresh ciphertext */
/* Borrowed sub-contexts (valid while ctx lives); moduli_count in 1..L. NULL when out of range. */
const he_poly_context* he_bfv_ciphertext_context(const he_bfv_context* ctx, uint32_t moduli_count);
const he_poly_context* he_bfv_key_switching_context(const he_bfv_context* ctx, uint32_t moduli_count);
const he_poly_context* he_bfv_qbsk_context(const he_bfv_context* ctx, uint32_t moduli_count);

/* _RnsTool pieces, exposed so each can be parity-tested on its own (RnsTool.swift):
 *   liftQToQBsk  :324-331  in [batch][L][N]    -> out [batch][2L+1][N]
 *   floorQBskToQ :453-456  in [batch][2L+1][N] -> out [batch][L][N]          */
int he_rns_lift_q_to_qbsk_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* in, uint64_t* out,
                                 size_t batch, he_stream s);
int he_rns_floor_qbsk_to_q_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* in, uint64_t* out,
                                  size_t batch, he_stream s);

/* Workspace: multi-kernel operations need device scratch.  Pass workspace == NULL to let the library allocate
 * stream-ordered scratch itself (hipMallocAsync on `s`); otherwise pass at least *_workspace_bytes(). */
size_t he_bfv_mul_workspace_bytes(const he_bfv_context* ctx, uint32_t moduli_count, size_t batch);
size_t he_bfv_relinearize_workspace_bytes(const he_bfv_context* ctx, uint32_t moduli_count, size_t batch);

/* Bfv.mulAssign(_: inout CanonicalCiphertext, _: CanonicalCiphertext) = multiplyWithoutScaling + dropExtendedBase
 * (Bfv/Bfv+Multiply.swift:18-85).  lhs, rhs: [batch][2][L][N] Coeff; out: [batch][3][L][N] Coeff, overlapping neither
 * operand (HE_ERR_INVALID_ARGUMENT: parts of the batch are stored while others are still read).  Enqueue-only on `s` as seen
 * from the caller: large batches also run part of the pipeline on a second stream the context owns, forked off `s` and joined
 * back into `s` by events before the call returns (not while `s` is being captured into a graph).  The context keeps up to 8
 * such streams and gives each caller stream the one it used last, so calls on different streams do not queue behind one
 * another's backlog; a call that finds all of them taken runs entirely on `s`. */
int he_bfv_mul_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* lhs, const uint64_t* rhs,
                      uint64_t* out, size_t batch, void* workspace, size_t workspace_bytes, he_stream s);
/* Bfv.relinearize (Bfv/Bfv.swift:201-219) via _computeKeySwitchingUpdate (Bfv/Bfv+Keys.swift:123-208).
 * ct3: [batch][3][L][N] Coeff; key: the relinearization key's L_top ciphertexts, [L_top][2][L_top+1][N] Eval over
 * the top key-switching context (Keys.swift:66-99); out: [batch][2][L][N] Coeff, not overlapping ct3 (the last kernel adds
 * the update to (c0, c1) as it reads them, item by item in no particular order): an overlap is HE_ERR_INVALID_ARGUMENT,
 * except a single ciphertext (batch == 1) relinearized onto its own first two polynomials (out == ct3).  key == NULL ->
 * missingRelinearizationKey. */
int he_bfv_relinearize_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* ct3,
                              const uint64_t* key, uint64_t* out, size_t batch, void* workspace,
                              size_t workspace_bytes, he_stream s);
/* Bfv.modSwitchDown (Bfv/Bfv.swift:163-171): [batch][polys][L][N] -> [batch][polys][L-1][N]. */
int he_bfv_mod_switch_down_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                  const uint64_t* in, uint64_t* out, size_t batch, he_stream s);
/* Ciphertext.modSwitchDownToSingle (Bfv/Bfv.swift:163-171): in [batch][polys][moduli_count][N] -> out [batch][polys][1][N],
 * the moduli_count - 1 divideAndRoundQLast steps in one kernel (word for word the chain of he_bfv_mod_switch_down_device). */
int he_bfv_mod_switch_down_to_single_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                            const uint64_t* in, uint64_t* out, size_t batch, he_stream s);
/* Bfv.mulAssign(_: inout EvalCiphertext, _: EvalPlaintext) (Bfv/Bfv.swift:120-129):
 * ct [batch][polys][L][N] *= pt [batch][L][N]. */
int he_bfv_mul_plain_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint64_t* ct,
                            const uint64_t* pt, size_t batch, he_stream s);
/* Bfv.addAssignCoeff / subAssignCoeff(_: inout CoeffCiphertext, _: CoeffPlaintext) (Bfv/Bfv.swift:110-117), i.e.
 * plaintextTranslate (Bfv/Bfv+Encrypt.swift:75-140): c0 +-= floor(Q/t) m + floor(((Q mod t) m + (t + 1)/2) / t) per
 * residue row.  ct [batch][polys][L][N] Coeff in place (only c0 is touched), plaintexts [batch][N] Coeff plaintexts
 * with values < t (batch b's plaintext goes to batch b's ciphertext).  `plaintext - ciphertext`
 * (HeScheme.swift:1540-1548) is he_poly_neg_device over the ciphertext's polynomials, then this add. */
int he_bfv_add_plain_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint64_t* ct,
                            const uint64_t* plaintexts, size_t batch, he_stream s);
int he_bfv_sub_plain_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint64_t* ct,
                            const uint64_t* plaintexts, size_t batch, he_stream s);
/* Bfv.innerProduct(ciphertexts:plaintexts:) (Bfv/Bfv.swift:476-505), batched over `columns` independent outputs
 * that share the ciphertext vector (the PIR dim-0 shape, PrivateInformationRetrieval/IndexPir/PirUtil.swift:427-445):
 *   cts      [count][polys][L][N]           Eval
 *   pts      [columns][count][L][N]         Eval plaintexts over the ciphertext context
 *   present  HOST [columns][count] bytes, 0 = nil plaintext (skipped, Bfv.swift:494); NULL = all present
 *   out      [columns][polys][L][N]         Eval
 * Every word must be a canonical residue (< q_i), as the reference's polynomials are: the accumulators count on it.
 * polys = 1, 2 or 3 polynomials per ciphertext -- or 4, 6, 8: the ciphertext vectors of 2, 3 or 4 QUERIES laid side by
 * side ([count][query][2][L][N]), which then share every plaintext word the kernel streams (out [columns][query][2]
 * [L][N]): a server that answers several queries over one database reads the database once for all of them. */
int he_bfv_inner_product_plain_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                      const uint64_t* cts, const uint64_t* pts, const uint8_t* present, size_t count,
                                      size_t columns, uint64_t* out, he_stream s);
/* The same with the mask resident on the DEVICE (NULL = all present): enqueue-only, never synchronises, may be
 * captured into a HIP graph.  (The host-mask form above waits once for the upload of the borrowed mask.) */
int he_bfv_inner_product_plain_resident_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                               const uint64_t* cts, const uint64_t* pts, const uint8_t* present_device,
                                               size_t count, size_t columns, uint64_t* out, he_stream s);
/* The plaintexts without the zero top bits of their words -- a layout private to this library for databases that stay
 * resident in HBM (the ct x pt inner product is bound by the bytes of plaintext it streams): row r of a plaintext is a
 * little-endian bit stream of N fields of bits(q_r) bits, rows in whole 8-byte words; a 55-bit modulus takes 6.875 bytes
 * per word instead of 8.  Degree >= 256, at most 8 moduli.
 *   he_bfv_packed_plaintext_words     8-byte words per packed plaintext (0: unsupported parameters)
 *   he_bfv_pack_plaintexts_device     plaintexts_eval [count][L][N] -> packed [count][words]
 *   he_bfv_inner_product_plain_packed_device   he_bfv_inner_product_plain_resident_device with pts packed
 *       ([columns][count][words]; polys 1..3); the same words out. */
size_t he_bfv_packed_plaintext_words(const he_bfv_context* ctx, uint32_t moduli_count);
int he_bfv_pack_plaintexts_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* plaintexts_eval,
                                  size_t count, uint64_t* packed, he_stream s);
int he_bfv_inner_product_plain_packed_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                             const uint64_t* cts, const uint64_t* packed_pts, const uint8_t* present_device,
                                             size_t count, size_t columns, uint64_t* out, he_stream s);
/* Bfv.innerProduct(_: [CanonicalCiphertext], _: [CanonicalCiphertext]) (Bfv/Bfv.swift:315-361):
 * lhs, rhs [count][2][L][N] Coeff -> out [3][L][N] Coeff. */
int he_bfv_inner_product_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* lhs,
                                const uint64_t* rhs, size_t count, uint64_t* out, void* workspace,
                                size_t workspace_bytes, he_stream s);
size_t he_bfv_inner_product_workspace_bytes(const he_bfv_context* ctx, uint32_t moduli_count, size_t count);
/* `items` such inner products that share their left vector -- the remaining-dimension step of a PIR response over all
 * result groups of all chunks (PirUtil.swift:448-479): lhs [count][2][L][N], rhs [items][count][2][L][N] Coeff ->
 * out [items][3][L][N] Coeff.  Item i equals he_bfv_inner_product_device(lhs, rhs + i * count ciphertexts); every stage is
 * one launch over all items and the left vector is lifted and transformed once.  Stream-ordered scratch. */
int he_bfv_inner_product_shared_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* lhs,
                                       const uint64_t* rhs, size_t count, size_t items, uint64_t* out, he_stream s);

/* Context<Bfv<UInt32>> (scheme layer): the word type fixes the largest modulus (2^30 - 1), gamma =
 * 2^30 - 20405, mTilde = 2^16 and the 29-bit Bsk primes (ModularArithmetic/Scalar.swift:498-511,
 * RnsTool.swift:30-33).  Results equal the reference's Bfv<UInt32> words exactly. */
int he_bfv_context_create_u32(uint32_t degree, uint64_t plaintext_modulus, const uint64_t* coefficient_moduli,
                              uint32_t moduli_count, he_bfv_context** out);
/* The scheme operations on PACKED [UInt32] slabs -- what Swift's Bfv<UInt32> holds (its PolyRq arrays are [UInt32]).
 * Same layouts, arguments, errors and citations as the entry points of the same name without the suffix, every slab
 * (ciphertexts, keys, plaintexts, workspaces) in 4-byte words; the context must come from he_bfv_context_create_u32.
 * The BEHZ / key-switching / inner-product kernels are the 8-byte ones instantiated on 4-byte slabs (a word is widened
 * in its register, never in memory); the transforms are the 4-byte NTT kernels (he_ntt_*_device_u32).  Workspaces: half
 * the bytes the he_bfv_*_workspace_bytes functions report, or NULL for stream-ordered scratch. */
int he_rns_lift_q_to_qbsk_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* in, uint32_t* out,
                                     size_t batch, he_stream s);
int he_rns_floor_qbsk_to_q_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* in,
                                      uint32_t* out, size_t batch, he_stream s);
int he_rns_scale_and_round_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* in,
                                      uint64_t scaling_factor, uint32_t* out, size_t batch, he_stream s);
int he_bfv_mul_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* lhs, const uint32_t* rhs,
                          uint32_t* out, size_t batch, void* workspace, size_t workspace_bytes, he_stream s);
int he_bfv_relinearize_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* ct3,
                                  const uint32_t* key, uint32_t* out, size_t batch, void* workspace,
                                  size_t workspace_bytes, he_stream s);
int he_bfv_apply_galois_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* ct, uint64_t element,
                                   const uint32_t* galois_key, uint32_t* out, size_t batch, void* workspace,
                                   size_t workspace_bytes, he_stream s);
int he_bfv_mod_switch_down_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                      const uint32_t* in, uint32_t* out, size_t batch, he_stream s);
int he_bfv_mul_plain_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint32_t* ct,
                                const uint32_t* pt, size_t batch, he_stream s);
int he_bfv_add_plain_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint32_t* ct,
                                const uint32_t* plaintexts, size_t batch, he_stream s);
int he_bfv_sub_plain_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint32_t* ct,
                                const uint32_t* plaintexts, size_t batch, he_stream s);
int he_bfv_inner_product_plain_resident_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                                   const uint32_t* cts, const uint32_t* pts,
                                                   const uint8_t* present_device, size_t count, size_t columns,
                                                   uint32_t* out, he_stream s);
int he_bfv_inner_product_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* lhs,
                                    const uint32_t* rhs, size_t count, uint32_t* out, void* workspace,
                                    size_t workspace_bytes, he_stream s);
int he_bfv_inner_product_shared_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* lhs,
                                           const uint32_t* rhs, size_t count, size_t items, uint32_t* out, he_stream s);
/* he_pir_compute_response_device on packed 4-byte slabs (query, database, key and responses in UInt32 words): the PIR
 * parameter sets with 27/28-bit moduli (EncryptionParameters.swift:313-345) stream half the database bytes of the
 * 8-byte route.  Same shapes, same words as the reference's Bfv<UInt32>.  Enqueue-only. */
int he_pir_compute_response_device_u32(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                                       const uint32_t* dim0_query_eval, const uint32_t* remaining_query,
                                       size_t remaining_query_count, const uint32_t* database,
                                       const uint8_t* present_device, size_t chunk_count,
                                       const uint32_t* relinearization_key, uint32_t* out, he_stream s);
/* he_pir_compute_response_queries_device on packed 4-byte slabs (a Bfv<UInt32> context): 1..4 queries share one pass
 * over the 4-byte database; layouts as there, 4-byte words. */
int he_pir_compute_response_queries_device_u32(const he_bfv_context* ctx, const uint32_t* dimensions,
                                               uint32_t dimension_count, size_t queries, const uint32_t* dim0_queries_eval,
                                               const uint32_t* remaining_queries, size_t remaining_query_count,
                                               const uint32_t* database, const uint8_t* present_device, size_t chunk_count,
                                               const uint32_t* const* relinearization_keys, uint32_t* out, he_stream s);
/* he_pir_compute_response_to_query_device for Bfv<UInt32> on packed 4-byte slabs (query, relinearization key, databases,
 * responses in UInt32 words).  The expansion, which is bound by its key switches and not by bytes, runs on widened words:
 * galois_keys_wide are the Galois keys as 8-byte slabs (he_words_widen_u32_device), as he_pir_expand_device takes them for
 * a UInt32 context.  Indices that share the database share its pass four at a time. */
int he_pir_compute_response_to_query_device_u32(const he_bfv_context* ctx, const uint32_t* dimensions,
                                                uint32_t dimension_count, const uint32_t* query_ciphertexts,
                                                size_t query_ciphertext_count, size_t indices_count,
                                                const uint64_t* galois_elements, const uint64_t* const* galois_keys_wide,
                                                size_t galois_key_count, const uint32_t* relinearization_key,
                                                const uint32_t* const* databases, const uint8_t* const* present_masks,
                                                size_t database_count, size_t chunk_count, uint32_t* out, he_stream s);
int he_bfv_plaintext_to_eval_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* plaintext,
                                        uint32_t* out, size_t batch, he_stream s);
int he_bfv_plaintext_to_coeff_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* plaintext_eval,
                                         uint32_t* out, size_t batch, he_stream s);
/* A Bfv<UInt32> context also works with every 8-byte entry point (he_bfv_*, he_rns_*, he_pir_*) on slabs of
 * zero-extended words; the PIR hooks (he_pir_*) exist in that form only.  Word-size bridge: widen a packed slab to
 * zero-extended 8-byte words and narrow results back (values of a UInt32 context are < 2^30, so the low half is the
 * word).  `words` counts coefficients; slabs 16-byte aligned; in != out. */
int he_words_widen_u32_device(const uint32_t* device_in, uint64_t* device_out, size_t words, he_stream s);
int he_words_narrow_u64_device(const uint64_t* device_in, uint32_t* device_out, size_t words, he_stream s);

/* ---- "next" rows of the scope table ---- */
/* Bfv.applyGalois(ciphertext:element:using:) (Bfv/Bfv.swift:174-198): ct [batch][2][L][N] Coeff,
 * galois_key = EvaluationKey.galoisKey.keys[element] in the relinearization key's layout
 * ([L_top][2][L_top+1][N] Eval; Bfv+Keys.swift:38-45,69-103) -> out [batch][2][L][N] Coeff.
 * NULL key -> HE_ERR_MISSING_GALOIS_KEY. */
int he_bfv_apply_galois_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* ct,
                               uint64_t element, const uint64_t* galois_key, uint64_t* out, size_t batch,
                               void* workspace, size_t workspace_bytes, he_stream s);
/* GaloisElement.swappingRows(degree:) and GaloisElement.rotatingColumns(by:degree:) (PolyRq/Galois.swift:174-212): the
 * elements Bfv.swapRows / Bfv.rotateColumns hand to applyGalois (HeScheme.swift:1047-1094).  step > 0 rotates right,
 * < 0 left, |step| in [1, N/2 - 1] (else HE_ERR_INVALID_ARGUMENT = invalidRotationStep); degree a power of two. */
int he_galois_element_swapping_rows(uint64_t degree, uint64_t* out_element);
int he_galois_element_rotating_columns(int64_t step, uint64_t degree, uint64_t* out_element);
/* Bfv.applyGalois on a batch whose items belong to different evaluation keys: `groups` runs of `group_size` consecutive
 * ciphertexts, group g under galois_keys[g] (host array of device pointers, he_bfv_apply_galois_device's key layout).
 * Workspace as he_bfv_apply_galois_workspace_bytes(ctx, moduli_count, groups * group_size). */
int he_bfv_apply_galois_grouped_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* ct,
                                       uint64_t element, const uint64_t* const* galois_keys, size_t groups,
                                       size_t group_size, uint64_t* out, void* workspace, size_t workspace_bytes,
                                       he_stream s);
size_t he_bfv_apply_galois_workspace_bytes(const he_bfv_context* ctx, uint32_t moduli_count, size_t batch);
/* _RnsTool.scaleAndRound(poly:scalingFactor:) (RnsTool.swift:272-302), the RNS step of Bfv.decrypt
 * (Bfv/Bfv+Decrypt.swift): in [batch][L][N] Coeff (c0 + c1 s [+ c2 s^2]) -> out [batch][N], values < t.
 * scaling_factor < t (the ciphertext's correction factor). */
int he_rns_scale_and_round_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* in,
                                  uint64_t scaling_factor, uint64_t* out, size_t batch, he_stream s);
/* Plaintext<Coeff>.convertToEvalFormat(moduliCount:) (Plaintext.swift:149-170): plaintext [batch][N] (values < t)
 * -> out [batch][L][N] Eval.  This is the database preprocessing step of PIR (MulPir.swift:507-556). */
int he_bfv_plaintext_to_eval_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* plaintext,
                                    uint64_t* out, size_t batch, he_stream s);
/* Plaintext<Eval>.convertToCoeffFormat() (Plaintext.swift:176-191): [batch][L][N] Eval -> [batch][N] (values < t). */
int he_bfv_plaintext_to_coeff_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* plaintext_eval,
                                     uint64_t* out, size_t batch, he_stream s);

/* =====================================================================================================
 * B4: application hook -- the PIR server's per-chunk response, device-resident
 * =================================================================================================== */

/* PirUtilProtocol.computeResponseForOneChunk (PrivateInformationRetrieval/IndexPir/PirUtil.swift:408-486,
 * MulPirServer.computeResponseForOneChunk IndexPir/MulPir.swift:369-410), all device pointers except `present`:
 *   dimensions          host, [dimension_count]             IndexPirParameter.dimensions
 *   dim0_query_eval     [d0][2][L][N] Eval                   expandedDim0Query
 *   remaining_query     [remaining_query_count][2][L][N] Coeff   expandedRemainingQuery (NULL / 0 for one dimension)
 *   database            [prod(dimensions)][L][N] Eval        dataChunk; plaintext k of column c at index c*d0 + k
 *                                                            (MulPir.swift:547-555)
 *   present             host, [prod(dimensions)] or NULL     0 = nil plaintext
 *   relinearization_key [L][2][L+1][N] Eval                  needed when dimension_count > 1
 *   out                 [2][1][N] Coeff over q_0             the response ciphertext after modSwitchDownToSingle
 * L = he_bfv_ciphertext_moduli_count(ctx).  The reference's precondition (columns == 1 or == remaining query count)
 * and a final result count != 1 return HE_ERR_INVALID_ARGUMENT. */
int he_pir_compute_response_chunk_device(const he_bfv_context* ctx, const uint32_t* dimensions,
                                         uint32_t dimension_count, const uint64_t* dim0_query_eval,
                                         const uint64_t* remaining_query, size_t remaining_query_count,
                                         const uint64_t* database, const uint8_t* present,
                                         const uint64_t* relinearization_key, uint64_t* out, he_stream s);

/* The two halves of computeResponseForOneChunk, for callers that shard the database by column (the reference groups
 * columns over tasks, PirUtil.swift:427-445; here: one column shard per GPU, the results gathered with RCCL):
 *   he_pir_dim0_columns_device          PirUtil.swift:428-446 for `columns` columns: inner product of the dim-0 query with
 *                                       each column's d0 plaintexts, then canonical (Coeff) format
 *                                       database [columns][d0][L][N] Eval, present_device DEVICE [columns][d0] or NULL
 *                                       out [columns][2][L][N] Coeff
 *   he_pir_remaining_dimensions_device  PirUtil.swift:448-485 on all columns' intermediate results:
 *                                       intermediate [columns][2][L][N] Coeff is consumed (overwritten)
 * Both are enqueue-only.  chunk = dim0_columns over all columns, then remaining_dimensions. */
int he_pir_dim0_columns_device(const he_bfv_context* ctx, const uint64_t* dim0_query_eval, size_t d0,
                               const uint64_t* database, const uint8_t* present_device, size_t columns, uint64_t* out,
                               he_stream s);
int he_pir_remaining_dimensions_device(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                                       uint64_t* intermediate, const uint64_t* remaining_query,
                                       size_t remaining_query_count, const uint64_t* relinearization_key, uint64_t* out,
                                       he_stream s);
/* he_pir_remaining_dimensions_device for `chunk_count` chunks at once: intermediate [chunk][columns][2][L][N] Coeff (consumed)
 * -> out [chunk][2][1][N]; every stage one batch over the result groups of all chunks (what the chunk loop below runs after its
 * dim-0 launch, for callers that produced the dim-0 results themselves: he_pir_compute_response_group). */
int he_pir_remaining_dimensions_chunks_device(const he_bfv_context* ctx, const uint32_t* dimensions,
                                              uint32_t dimension_count, size_t chunk_count, uint64_t* intermediate,
                                              const uint64_t* remaining_query, size_t remaining_query_count,
                                              const uint64_t* relinearization_key, uint64_t* out, he_stream s);
/* PirUtilProtocol.computeResponse's chunk loop for one query (PirUtil.swift:533-563): the database holds `chunk_count`
 * chunks of prod(dimensions) plaintexts each ([chunk][prod(dimensions)][L][N] Eval), the same expanded query serves
 * every chunk, out [chunk_count][2][1][N].  The chunks are independent (the reference maps them over tasks) and are
 * answered together: one dim-0 launch over the columns of all chunks, then every stage of every remaining dimension as
 * one batch over the result groups of all chunks (he_bfv_inner_product_shared_device), in groups of chunks that keep the
 * intermediate ciphertexts under about 1 GiB.  present_device: DEVICE [chunk][prod(dimensions)] or NULL.  Enqueue-only. */
int he_pir_compute_response_device(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                                   const uint64_t* dim0_query_eval, const uint64_t* remaining_query,
                                   size_t remaining_query_count, const uint64_t* database, const uint8_t* present_device,
                                   size_t chunk_count, const uint64_t* relinearization_key, uint64_t* out, he_stream s);
/* The same for `queries` (1..4) queries over the same database in one call: their dim-0 inner products share one pass
 * over the database (the plaintexts are read from HBM once for all queries; 1.7-2.1 x the single-query rate per query
 * at 2-4 queries), the remaining dimensions run query by query.
 *   dim0_queries_eval   [dimensions[0]][queries][2][L][N] Eval  (the queries' dim-0 ciphertexts side by side)
 *   remaining_queries   [queries][remaining_query_count][2][L][N] Coeff
 *   relinearization_keys HOST array of `queries` device pointers (one key per query; NULL for one-dimensional databases)
 *   out                 [queries][chunk_count][2][1][N]
 * Each query's responses equal he_pir_compute_response_device's for that query alone.  Enqueue-only. */
int he_pir_compute_response_queries_device(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                                           size_t queries, const uint64_t* dim0_queries_eval,
                                           const uint64_t* remaining_queries, size_t remaining_query_count,
                                           const uint64_t* database, const uint8_t* present_device, size_t chunk_count,
                                           const uint64_t* const* relinearization_keys, uint64_t* out, he_stream s);
/* The same two with the database packed (he_bfv_pack_plaintexts_device at the top level: [chunk][prod(dimensions)]
 * [he_bfv_packed_plaintext_words] words): the dim-0 pass, which is bound by the
 * database bytes it streams, reads bits(q) / 64 of them.  Same responses, word for word. */
int he_pir_dim0_columns_packed_device(const he_bfv_context* ctx, const uint64_t* dim0_query_eval, size_t d0,
                                      const uint64_t* packed_database, const uint8_t* present_device, size_t columns,
                                      uint64_t* out, he_stream s);
int he_pir_compute_response_packed_device(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                                          const uint64_t* dim0_query_eval, const uint64_t* remaining_query,
                                          size_t remaining_query_count, const uint64_t* packed_database,
                                          const uint8_t* present_device, size_t chunk_count,
                                          const uint64_t* relinearization_key, uint64_t* out, he_stream s);

/* PirUtilProtocol.computeResponse(to:using:databases:parameter:context:callOptions:) (PirUtil.swift:490-568) -- the whole
 * server side of a Query in one call: expand its ciphertexts into sum(dimensions) selection ciphertexts per queried index
 * (he_pir_expand_device), take each index's first dimensions[0] of them to Eval, answer every chunk.
 *   query_ciphertexts   [query_ciphertext_count][2][L][N] Coeff (Query.ciphertexts)
 *   indices_count       Query.indicesCount
 *   galois_elements / galois_keys / galois_key_count   the evaluation key's Galois keys, as for he_pir_expand_device
 *   relinearization_key device key (NULL for one-dimensional databases)
 *   databases           HOST array of database_count device pointers, each [chunk_count][prod(dimensions)][L][N] Eval;
 *                       database_count == 1: every index queries that database (four indices at a time then share one
 *                       pass over it); database_count >= indices_count: index i queries databases[i].  Anything else is
 *                       the reference's PirError.invalidBatchSize: HE_ERR_INVALID_ARGUMENT.
 *   present_masks       HOST array of as many DEVICE masks [chunk_count][prod(dimensions)] (entries or the array NULL:
 *                       all present)
 *   out                 [indices_count][chunk_count][2][1][N]  (Response.ciphertexts)
 * Enqueue-only once the context has seen the query's shape (he_pir_expand_device). */
int he_pir_compute_response_to_query_device(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                                            const uint64_t* query_ciphertexts, size_t query_ciphertext_count,
                                            size_t indices_count, const uint64_t* galois_elements,
                                            const uint64_t* const* galois_keys, size_t galois_key_count,
                                            const uint64_t* relinearization_key, const uint64_t* const* databases,
                                            const uint8_t* const* present_masks, size_t database_count, size_t chunk_count,
                                            uint64_t* out, he_stream s);

/* PirUtil.expand(ciphertexts:outputCount:using:) (PrivateInformationRetrieval/IndexPir/PirUtil.swift:196-355):
 * oblivious expansion of `ciphertext_count` query ciphertexts [..][2][L][N] (Coeff, top level) into `output_count`
 * ciphertexts, in the reference's output order.  The evaluation key is given as parallel host arrays:
 * galois_elements[k] and the device pointer galois_keys[k] of that element's key (layout as in
 * he_bfv_apply_galois_device).  Per tree level the largest element <= the target element is applied
 * 2^(log2(target-1) - log2(element-1)) times (PirUtil.swift:217-231); none available -> HE_ERR_MISSING_GALOIS_KEY.
 * The count preconditions of PirUtil.swift:325-326 return HE_ERR_INVALID_ARGUMENT.
 * The recursion is planned on the host once per (ciphertext_count, output_count) and context: the first call of a shape
 * uploads the plan (a blocking copy of a few KB, kept by the context); every later one is enqueue-only. */
int he_pir_expand_device(const he_bfv_context* ctx, const uint64_t* ciphertexts, size_t ciphertext_count,
                         size_t output_count, const uint64_t* galois_elements, const uint64_t* const* galois_keys,
                         size_t galois_key_count, uint64_t* out, he_stream s);
/* `queries` independent expansions of one shape in one call (several clients' queries answered together): every tree
 * level is one batch over all queries, so the first levels -- 1, 2, 4, ... ciphertexts per query -- fill the chip too.
 *   ciphertexts  [queries][ciphertext_count][2][L][N]      out  [queries][output_count][2][L][N]
 *   galois_keys  host array [queries][galois_key_count] of device pointers: query q's key of galois_elements[k] at
 *                [q * galois_key_count + k] (queries of one client repeat the pointer; only the inner product with the
 *                key is launched per run of equal keys)
 * he_pir_expand_device is this call with queries = 1. */
int he_pir_expand_batch_device(const he_bfv_context* ctx, const uint64_t* ciphertexts, size_t queries,
                               size_t ciphertext_count, size_t output_count, const uint64_t* galois_elements,
                               const uint64_t* const* galois_keys, size_t galois_key_count, uint64_t* out, he_stream s);

/* =====================================================================================================
 * Device groups: the path's multi-GPU split in one process
 * =====================================================================================================
 * Every unit of the path -- a polynomial of a batch (Bfv.swift:266-287), a database column of a PIR chunk
 * (PirUtil.swift:424-445) -- is independent; the reference spreads them over the tasks of one process.  A device group spreads
 * them over GPUs: one he_bfv_context and one stream per member device (contexts replicated, a few MiB of tables each), `total`
 * units split by he_shard_bounds -- member m owns [begin, end), the first total % members members one unit more -- and nothing
 * on the data path crossing devices: the query is copied to the members on the way in, the finished shards to the home
 * device (member 0's) on the way out, as peer copies on the members' streams joined into the caller's stream by events.
 * A device may be listed more than once (two shards on one GPU).  HE_GROUP_STAGE_ALL makes every member but the first go
 * through the copies of a remote device even when it is the same GPU -- the whole exchange on a one-GPU box (tests).
 * Group calls are enqueue-only; `home_stream` is a stream of member 0's device (NULL: its default stream), pointers
 * marked "home" live there, shard m lives on member m's device. */
typedef struct he_device_group he_device_group;
#define HE_GROUP_STAGE_ALL 1u
int he_shard_bounds(size_t total, uint32_t members, uint32_t member, size_t* out_begin, size_t* out_end);
int he_device_group_create(const int* devices, uint32_t device_count, uint32_t flags, uint32_t degree,
                           uint64_t plaintext_modulus, const uint64_t* coefficient_moduli, uint32_t moduli_count,
                           he_device_group** out);
void he_device_group_destroy(he_device_group* group);
uint32_t he_device_group_size(const he_device_group* group);
int he_device_group_device(const he_device_group* group, uint32_t member, int* out_device);
const he_bfv_context* he_device_group_context(const he_device_group* group, uint32_t member); /* owned by the group */
he_stream he_device_group_stream(const he_device_group* group, uint32_t member);              /* owned by the group */
int he_device_group_synchronize(he_device_group* group); /* blocks until every member's stream has drained */
/* PolyContext.forwardNtt / inverseNtt over a batch that lives sharded: member m transforms
 * slab_shards[m] = [its share of batch][moduli_count][N] in place on its own stream; nothing moves between devices. */
int he_ntt_forward_group(he_device_group* group, uint32_t moduli_count, uint64_t* const* slab_shards, size_t batch);
int he_ntt_inverse_group(he_device_group* group, uint32_t moduli_count, uint64_t* const* slab_shards, size_t batch);
/* he_pir_dim0_columns_device over a database sharded by column (BASELINE configs[4]): database_shards[m] =
 * [its share of columns][d0][L][N] Eval (present_shards[m] its mask or NULL; present_shards itself may be NULL);
 * dim0_query_eval (home) is replicated to the members; out (home) [columns][2][L][N] Coeff receives every member's columns
 * at their place.  Whole columns stay on one device, so there is no reduction, only this gather. */
int he_pir_dim0_columns_group(he_device_group* group, const uint64_t* dim0_query_eval, size_t d0,
                              const uint64_t* const* database_shards, const uint8_t* const* present_shards, size_t columns,
                              uint64_t* out, he_stream home_stream);
/* computeResponseForOneChunk over the group (chunk_count = 1): he_pir_dim0_columns_group, then the remaining dimensions on the
 * home device (they need every column).  remaining_query, relinearization_key, out: home. */
/* The chunk loop of PirUtil.computeResponse (PirUtil.swift:533-563) over the group: the columns of all `chunk_count` chunks are
 * one column range, sharded as above (database_shards[m] = member m's share of chunk_count x columns columns); one dim-0 pass
 * per member, the gather, then the remaining dimensions of all chunks on the home device.  out (home) [chunk_count][2][1][N].
 * Where he_shard_bounds falls on chunk boundaries for every member (chunk_count a multiple of the group's size) each member
 * answers its chunks from start to finish instead -- he_pir_compute_response_device on its own stream, the queries and the key
 * replicated to it -- and only the finished responses come back: no part of the call is left to one device alone. */
int he_pir_compute_response_group(he_device_group* group, const uint32_t* dimensions, uint32_t dimension_count,
                                  const uint64_t* dim0_query_eval, const uint64_t* remaining_query,
                                  size_t remaining_query_count, const uint64_t* const* database_shards,
                                  const uint8_t* const* present_shards, size_t chunk_count,
                                  const uint64_t* relinearization_key, uint64_t* out, he_stream home_stream);
int he_pir_compute_response_chunk_group(he_device_group* group, const uint32_t* dimensions, uint32_t dimension_count,
                                        const uint64_t* dim0_query_eval, const uint64_t* remaining_query,
                                        size_t remaining_query_count, const uint64_t* const* database_shards,
                                        const uint8_t* const* present_shards, const uint64_t* relinearization_key,
                                        uint64_t* out, he_stream home_stream);

/* =====================================================================================================
 * Diagnostics and test hooks (not part of the reference's surface)
 * =================================================================================================== */

/* Builds the host-side precomputation only (no device upload): lets the setup math (roots, twiddle order, Barrett /
 * Shoup constants) be checked where no GPU exists.  Every compute entry point on such a context returns
 * HE_ERR_DEVICE. */
int he_poly_context_create_host_only(uint32_t degree, const uint64_t* moduli, uint32_t moduli_count,
                                     he_poly_context** out);
/* Copies the _NttContext tables of modulus row `rns_index` (PolyRq/PolyRq+Ntt.swift:109-115); any out may be NULL. */
int he_poly_context_copy_ntt_tables(const he_poly_context* ctx, uint32_t rns_index, uint64_t* root_powers,
                                    uint64_t* root_factors, uint64_t* inverse_root_powers,
                                    uint64_t* inverse_root_factors, uint64_t* inverse_degree,
                                    uint64_t* inverse_degree_root);
/* A streaming copy of `words` 8-byte words, 8 bytes per lane -- the access width of the transforms' row loads and stores;
 * non_temporal != 0 also takes their cache policy.  What bench.py reports as the attainable rate next to the nominal
 * 8 TB/s (`roofline.copy_rate`: the faster of the two policies). */
int he_words_copy_device(const uint64_t* device_in, uint64_t* device_out, size_t words, int non_temporal, he_stream s);
/* NTT with a named kernel schedule -- every accepted variant computes the same canonical transform (parity tests
 * pin each schedule against the oracle): 0 = auto (production), 1 = exact-quotient butterflies, 2 = generic radix-2
 * kernel, 3 = 16 words per lane, 10 = [0, 8p) butterflies.  Anything else:
 * HE_ERR_INVALID_ARGUMENT. */
int he_ntt_device_variant(const he_poly_context* ctx, uint64_t* device_slab, size_t batch, int inverse, int variant,
                          he_stream stream);
/* Same host-only construction for the BFV context; he_bfv_copy_bsk_moduli returns the L+1 Bsk primes
 * (RnsTool.swift:28-33). */
int he_bfv_context_create_host_only(uint32_t degree, uint64_t plaintext_modulus, const uint64_t* coefficient_moduli,
                                    uint32_t moduli_count, he_bfv_context** out);
int he_bfv_copy_bsk_moduli(const he_bfv_context* ctx, uint64_t* out_bsk);

#ifdef __cplusplus
}
#endif
#endif /* HE_AMD_H */
