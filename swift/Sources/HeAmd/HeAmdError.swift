// Status codes of he_amd.h -> HeError (reference Sources/HomomorphicEncryption/Error.swift:18-56): one status per
// reachable case, raised in the reference's order because the C side validates in the reference's order.
import CHeAmd
import HomomorphicEncryption

/// Throws the `HeError` a status of the C ABI stands for; returns normally on `HE_OK`.
/// Payloads that are Swift values which never crossed the boundary (the offending modulus, the two contexts) are
/// replaced by the library's thread-local detail string where the case carries a description, by zero otherwise.
@inline(__always)
public func heAmdCheck(_ status: Int32) throws {
    guard status != Int32(HE_OK.rawValue) else { return }
    let detail = String(cString: he_last_error_message())
    switch he_status(UInt32(status)) {
    case HE_ERR_INVALID_DEGREE: throw HeError.invalidDegree(0)
    case HE_ERR_INVALID_MODULUS: throw HeError.invalidModulus(0)
    case HE_ERR_COPRIME_MODULI: throw HeError.coprimeModuli(moduli: [])
    case HE_ERR_EMPTY_MODULUS: throw HeError.emptyModulus
    case HE_ERR_INVALID_NTT_MODULUS: throw HeError.invalidNttModulus(modulus: 0, degree: 0)
    case HE_ERR_INVALID_POLY_CONTEXT: throw HeError.invalidPolyContext(detail)
    case HE_ERR_POLY_CONTEXT_MISMATCH: throw HeError.polyContextMismatch(detail)
    case HE_ERR_INVALID_CIPHERTEXT: throw HeError.invalidCiphertext(detail)
    case HE_ERR_INCOMPATIBLE_CIPHERTEXTS: throw HeError.incompatibleCiphertexts(detail)
    case HE_ERR_INCOMPATIBLE_CIPHERTEXT_AND_PLAINTEXT: throw HeError.incompatibleCiphertextAndPlaintext(detail)
    case HE_ERR_MISSING_RELINEARIZATION_KEY: throw HeError.missingRelinearizationKey
    case HE_ERR_MISSING_GALOIS_KEY: throw HeError.missingGaloisKey
    case HE_ERR_UNEQUAL_CONTEXTS: throw HeError.unequalContexts(detail)
    case HE_ERR_NOT_ENOUGH_PRIMES:
        throw HeError.notEnoughPrimes(significantBitCounts: [], preferringSmall: false, nttDegree: 0)
    case HE_ERR_NOT_INVERTIBLE: throw HeError.notInvertible(modulus: 0)
    case HE_ERR_INVALID_ENCRYPTION_PARAMETERS: throw HeError.invalidEncryptionParameters(detail)
    case HE_ERR_SERIALIZED_BUFFER_SIZE_MISMATCH:
        throw HeError.serializedBufferSizeMismatch(polyContext: detail, actual: 0, expected: 0)
    case HE_ERR_INVALID_COEFFICIENT_PACKING: throw HeError.invalidCoefficientPacking(bitsPerCoeff: 0, skipLSBs: 0)
    default: // HE_ERR_INVALID_ARGUMENT (a `precondition` of the reference), HE_ERR_DEVICE, HE_ERR_UNSUPPORTED
        throw HeError.unsupportedHeOperation(
            description: "he_amd: \(String(cString: he_status_string(status))): \(detail)")
    }
}
