// Device twins of the reference's immutable contexts.  PolyContext and Context are final classes of another module, so
// the handles cannot be stored on them; they are cached here, keyed by what defines the context (degree, moduli,
// plaintext modulus), created once and kept for the life of the process -- contexts are immutable and Sendable
// (PolyRq/PolyContext.swift:19, Context.swift:19), and every entry point of the C ABI is re-entrant on a shared handle.
import CHeAmd
import Foundation
import HomomorphicEncryption

struct ContextKey: Hashable {
    let degree: Int
    let moduli: [UInt64]
    let plaintextModulus: UInt64
}

final class GpuContextCache: @unchecked Sendable {
    static let shared = GpuContextCache()
    private let lock = NSLock()
    private var polyContexts: [ContextKey: OpaquePointer] = [:]
    private var bfvContexts: [ContextKey: OpaquePointer] = [:]

    /// `he_poly_context` of a PolyContext<UInt64> (PolyContext.init's validation already passed on the Swift side; the
    /// C side repeats it in the same order and builds the device tables).
    func handle(for context: PolyContext<UInt64>) throws -> OpaquePointer {
        let key = ContextKey(degree: context.degree, moduli: context.moduli, plaintextModulus: 0)
        lock.lock()
        defer { lock.unlock() }
        if let cached = polyContexts[key] { return cached }
        var out: OpaquePointer?
        try context.moduli.withUnsafeBufferPointer { moduli in
            try heAmdCheck(he_poly_context_create(UInt32(context.degree), moduli.baseAddress, UInt32(moduli.count), &out))
        }
        guard let out else { throw HeError.unsupportedHeOperation(description: "he_poly_context_create returned nil") }
        polyContexts[key] = out
        return out
    }

    /// `he_bfv_context` of a Context<Bfv<UInt64>> (Context.swift:94-143): all coefficient moduli, the last one being
    /// the key-switching modulus when there are several.
    func handle(for context: Context<Bfv<UInt64>>) throws -> OpaquePointer {
        let key = ContextKey(degree: context.degree, moduli: context.coefficientModuli,
                             plaintextModulus: context.plaintextModulus)
        lock.lock()
        defer { lock.unlock() }
        if let cached = bfvContexts[key] { return cached }
        var out: OpaquePointer?
        try context.coefficientModuli.withUnsafeBufferPointer { moduli in
            try heAmdCheck(he_bfv_context_create(UInt32(context.degree), context.plaintextModulus, moduli.baseAddress,
                                                 UInt32(moduli.count), &out))
        }
        guard let out else { throw HeError.unsupportedHeOperation(description: "he_bfv_context_create returned nil") }
        bfvContexts[key] = out
        return out
    }
}

extension PolyContext where T == UInt64 {
    /// The device twin of this context (created on first use, on the HIP device current at that time).
    public var gpu: OpaquePointer {
        get throws { try GpuContextCache.shared.handle(for: self) }
    }
}

extension Context where Scheme == Bfv<UInt64> {
    /// The device twin of this context.
    public var gpu: OpaquePointer {
        get throws { try GpuContextCache.shared.handle(for: self) }
    }
}
