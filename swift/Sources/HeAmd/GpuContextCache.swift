// Device twins of the reference's immutable contexts.  PolyContext and Context are final classes of another module, so
// the handles cannot be stored on them; they are cached here, keyed by what defines the context (degree, moduli,
// plaintext modulus) AND by the HIP device that was current when the twin was built -- a handle belongs to its device
// (every C entry point answers HE_ERR_DEVICE on another one), so a process that drives several GPUs, one thread or task
// per device after `he_set_device`, gets one twin per device.  Twins live for the life of the process: contexts are
// immutable and Sendable (PolyRq/PolyContext.swift:19, Context.swift:19) and every entry point of the C ABI is
// re-entrant on a shared handle.
import CHeAmd
import Foundation
import HomomorphicEncryption

struct ContextKey: Hashable {
    let device: Int32
    let degree: Int
    let moduli: [UInt64]
    let plaintextModulus: UInt64
}

final class GpuContextCache: @unchecked Sendable {
    static let shared = GpuContextCache()
    private let lock = NSLock()
    private var polyContexts: [ContextKey: OpaquePointer] = [:]
    private var bfvContexts: [ContextKey: OpaquePointer] = [:]

    /// The calling thread's current HIP device.
    static func currentDevice() throws -> Int32 {
        var device: Int32 = 0
        try heAmdCheck(he_get_device(&device))
        return device
    }

    /// `he_poly_context` of a PolyContext<UInt64> (PolyContext.init's validation already passed on the Swift side; the
    /// C side repeats it in the same order and builds the device tables).
    func handle(for context: PolyContext<UInt64>) throws -> OpaquePointer {
        let device = try Self.currentDevice()
        let key = ContextKey(device: device, degree: context.degree, moduli: context.moduli, plaintextModulus: 0)
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

    /// `he_bfv_context` of any scheme's context over UInt64 (Context.swift:94-143): all coefficient moduli, the last
    /// one being the key-switching modulus when there are several.
    func handle(degree: Int, coefficientModuli: [UInt64], plaintextModulus: UInt64) throws -> OpaquePointer {
        let device = try Self.currentDevice()
        let key = ContextKey(device: device, degree: degree, moduli: coefficientModuli, plaintextModulus: plaintextModulus)
        lock.lock()
        defer { lock.unlock() }
        if let cached = bfvContexts[key] { return cached }
        var out: OpaquePointer?
        try coefficientModuli.withUnsafeBufferPointer { moduli in
            try heAmdCheck(he_bfv_context_create(UInt32(degree), plaintextModulus, moduli.baseAddress,
                                                 UInt32(moduli.count), &out))
        }
        guard let out else { throw HeError.unsupportedHeOperation(description: "he_bfv_context_create returned nil") }
        bfvContexts[key] = out
        return out
    }
}

/// The library's scratch on the current HIP device (include/he_amd.h `he_set_scratch_cache`).  By default nothing is retained
/// (a HIP memory pool with release threshold 0) -- and a call that took scratch returns one call late, because `hipFreeAsync`
/// waits for the work before the block's previous release.  A server opts in ONCE with a bound it chooses: the library then
/// keeps released scratch in its own stream-ordered block cache, its calls are enqueue-only, and a batched expansion's tens of
/// gigabytes are not mapped anew per call.  `trimScratch` hands the memory back when the server goes idle.  The package never
/// opts in on the host's behalf.
public enum HeAmdScratch {
    /// Lets the library keep up to `bytes` of released scratch on the current device (`UInt64.max`: everything).
    public static func setScratchCache(bytes: UInt64) throws {
        try heAmdCheck(he_set_scratch_cache(bytes))
    }

    /// Returns everything above `keepBytes` to the driver.
    public static func trimScratch(keepBytes: UInt64 = 0) throws {
        try heAmdCheck(he_device_trim_scratch(keepBytes))
    }
}

extension PolyContext where T == UInt64 {
    /// The device twin of this context on the current HIP device (created on first use).
    public var gpu: OpaquePointer {
        get throws { try GpuContextCache.shared.handle(for: self) }
    }
}

extension HeContext where Scalar == UInt64 {
    /// The device twin of this context on the current HIP device -- for `Context<Bfv<UInt64>>` and `Context<GpuBfv>`
    /// alike (HeScheme.swift:88-101: what defines a context is its encryption parameters).
    public var gpu: OpaquePointer {
        get throws {
            try GpuContextCache.shared.handle(degree: degree, coefficientModuli: coefficientModuli,
                                              plaintextModulus: plaintextModulus)
        }
    }
}
