// GpuBfv <-> Bfv<UInt64>: the same polynomials under two scheme tags.
//
// `Ciphertext<GpuBfv, F>` and `Ciphertext<Bfv<UInt64>, F>` are distinct types with distinct `Context<Scheme>` objects
// (SURVEY.md 8b), but what they hold -- `PolyRq<UInt64, F>` over `PolyContext<UInt64>` -- does not mention the scheme, and
// the reference exposes the initialisers an out-of-module scheme needs (underscore-public, "not subject to semantic
// versioning"): Ciphertext.init(_context:_polys:_correctionFactor:_auxiliaryData:_seed:) Ciphertext.swift:64-84,
// Plaintext.init(_context:_poly:) Plaintext.swift:36, SecretKey.init(_poly:) Keys.swift:34,
// _KeySwitchKey.init(_context:_ciphertexts:) Keys.swift:90, _RelinearizationKey.init(_keySwitchKey:) Keys.swift:123,
// _GaloisKey.init(_keys:) Keys.swift:155, EvaluationKey.init(_galoisKey:_relinearizationKey:) Keys.swift:210.
// Re-wrapping copies array REFERENCES (copy-on-write storage), never words.  Everything GpuBfv does not accelerate goes
// through these to the reference's own `Bfv<UInt64>`.
import Foundation
import HomomorphicEncryption

/// The CPU scheme GpuBfv forwards to and must agree with word for word.
public typealias CpuBfv = Bfv<UInt64>

final class TwinContexts: @unchecked Sendable {
    static let shared = TwinContexts()
    private let lock = NSLock()
    private var cpu: [EncryptionParameters<UInt64>: Context<CpuBfv>] = [:]
    private var gpu: [EncryptionParameters<UInt64>: Context<GpuBfv>] = [:]

    func cpuContext(_ parameters: EncryptionParameters<UInt64>) throws -> Context<CpuBfv> {
        lock.lock()
        defer { lock.unlock() }
        if let cached = cpu[parameters] { return cached }
        let context = try Context<CpuBfv>(encryptionParameters: parameters) // Context.swift:94
        cpu[parameters] = context
        return context
    }

    func gpuContext(_ parameters: EncryptionParameters<UInt64>) throws -> Context<GpuBfv> {
        lock.lock()
        defer { lock.unlock() }
        if let cached = gpu[parameters] { return cached }
        let context = try Context<GpuBfv>(encryptionParameters: parameters)
        gpu[parameters] = context
        return context
    }
}

extension Context where Scheme == GpuBfv {
    /// The reference's own context over the same encryption parameters (built once per parameter set).
    public var cpu: Context<CpuBfv> {
        get throws { try TwinContexts.shared.cpuContext(encryptionParameters) }
    }
}

extension Context where Scheme == CpuBfv {
    /// The GpuBfv context over the same encryption parameters.
    public var accelerated: Context<GpuBfv> {
        get throws { try TwinContexts.shared.gpuContext(encryptionParameters) }
    }
}

extension Plaintext where Scheme == GpuBfv {
    public func cpu() throws -> Plaintext<CpuBfv, Format> {
        try Plaintext<CpuBfv, Format>(_context: context.cpu, _poly: _poly)
    }

    public init(cpu plaintext: Plaintext<CpuBfv, Format>, context: Context<GpuBfv>) throws {
        try self.init(_context: context, _poly: plaintext._poly)
    }
}

extension Ciphertext where Scheme == GpuBfv {
    public func cpu() throws -> Ciphertext<CpuBfv, Format> {
        try Ciphertext<CpuBfv, Format>(_context: context.cpu, _polys: polys, _correctionFactor: correctionFactor,
                                       _auxiliaryData: nil, _seed: seed)
    }

    public init(cpu ciphertext: Ciphertext<CpuBfv, Format>, context: Context<GpuBfv>) throws {
        try self.init(_context: context, _polys: ciphertext.polys, _correctionFactor: ciphertext.correctionFactor,
                      _auxiliaryData: nil, _seed: ciphertext.seed)
    }
}

extension Ciphertext where Scheme == CpuBfv {
    /// The same ciphertext under the GpuBfv tag: `ciphertext.accelerated()` is all a caller of generic code
    /// (`MulPirServer<PirUtil<GpuBfv>>`, PNNS) needs to move an existing `Bfv<UInt64>` value over.
    public func accelerated() throws -> Ciphertext<GpuBfv, Format> {
        try Ciphertext<GpuBfv, Format>(cpu: self, context: context.accelerated)
    }
}

extension SecretKey where Scheme == GpuBfv {
    /// (a copy of the key polynomial; the copy zeroizes itself like the original, Keys.swift:46-48)
    public func cpu() -> SecretKey<CpuBfv> {
        SecretKey<CpuBfv>(_poly: _poly)
    }
}

extension _KeySwitchKey where Scheme == GpuBfv {
    func cpu() throws -> _KeySwitchKey<CpuBfv> {
        try _KeySwitchKey<CpuBfv>(_context: _context.cpu, _ciphertexts: _ciphertexts.map { try $0.cpu() })
    }

    init(cpu key: _KeySwitchKey<CpuBfv>, context: Context<GpuBfv>) throws {
        try self.init(_context: context,
                      _ciphertexts: key._ciphertexts.map { try Ciphertext<GpuBfv, Eval>(cpu: $0, context: context) })
    }
}

extension EvaluationKey where Scheme == GpuBfv {
    public func cpu() throws -> EvaluationKey<CpuBfv> {
        let galois: _GaloisKey<CpuBfv>? = try _galoisKey.map { key in
            try _GaloisKey<CpuBfv>(_keys: key._keys.mapValues { try $0.cpu() })
        }
        let relinearization: _RelinearizationKey<CpuBfv>? = try _relinearizationKey.map { key in
            try _RelinearizationKey<CpuBfv>(_keySwitchKey: key._keySwitchKey.cpu())
        }
        return EvaluationKey<CpuBfv>(_galoisKey: galois, _relinearizationKey: relinearization)
    }

    public init(cpu key: EvaluationKey<CpuBfv>, context: Context<GpuBfv>) throws {
        let galois: _GaloisKey<GpuBfv>? = try key._galoisKey.map { galoisKey in
            try _GaloisKey<GpuBfv>(_keys: galoisKey._keys.mapValues { try _KeySwitchKey<GpuBfv>(cpu: $0, context: context) })
        }
        let relinearization: _RelinearizationKey<GpuBfv>? = try key._relinearizationKey.map { relinearizationKey in
            try _RelinearizationKey<GpuBfv>(
                _keySwitchKey: _KeySwitchKey<GpuBfv>(cpu: relinearizationKey._keySwitchKey, context: context))
        }
        self.init(_galoisKey: galois, _relinearizationKey: relinearization)
    }
}
