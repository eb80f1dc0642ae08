// What a PIR server keeps in HBM between queries, and the caches that find it again from the reference's VALUE types:
//   * GpuResidentDatabase   a ProcessedDatabase (IndexPirProtocol.swift:249-290): every Eval plaintext of every chunk
//                           plus the nil mask, uploaded once;
//   * GpuEvaluationKey      an EvaluationKey (Keys.swift:186-219): its Galois keys by element and its relinearization
//                           key in the layout the C ABI takes;
//   * GpuResidentCache      ProcessedDatabase and EvaluationKey are structs (no object identity), so residency is keyed by
//                           the identity of their copy-on-write STORAGE -- the address of the first polynomial's words,
//                           the element count and a fingerprint of a few words -- and by the HIP device.  A server that
//                           replaces a database drops the old entry with `evict` (or `removeAll`).
// GpuPirUtil (GpuPirUtil.swift) answers PirUtilProtocol's requirements from these.
import CHeAmd
import Foundation
import HomomorphicEncryption
import PrivateInformationRetrieval

public final class GpuResidentDatabase<Scheme: HeScheme>: @unchecked Sendable where Scheme.Scalar == UInt64 {
    /// [chunk][prod(dimensions)][L][N] Eval words, `nil` plaintexts left as zeros.
    public let plaintexts: DeviceBuffer
    /// One byte per plaintext, 0 = nil (Bfv.swift:486-489), on the device.
    public let present: DeviceBuffer
    public let plaintextCount: Int

    /// Uploads `database` once (MulPir.swift:547-555 fixes the order: plaintext k of column c of a chunk at c * d0 + k).
    public init(_ database: ProcessedDatabase<Scheme>, polyContext: PolyContext<UInt64>) throws {
        let polyWords = polyContext.moduli.count * polyContext.degree
        let stream = try HeAmdStream()
        plaintextCount = database.plaintexts.count
        plaintexts = try DeviceBuffer(count: plaintextCount * polyWords)
        present = try DeviceBuffer(count: (plaintextCount + 7) / 8) // the mask travels as bytes; buffers count 8-byte words
        var mask = [UInt8](repeating: 0, count: plaintextCount)
        for (index, plaintext) in database.plaintexts.enumerated() {
            guard let plaintext else { continue }
            mask[index] = 1
            try plaintexts.upload(plaintext._poly, at: index * polyWords, on: stream) // Plaintext.swift:28
        }
        try mask.withUnsafeBufferPointer { bytes in
            try present.upload(bytes: bytes, atByte: 0, on: stream)
        }
    }

    var maskPointer: UnsafePointer<UInt8> {
        UnsafePointer(UnsafeRawPointer(present.pointer).assumingMemoryBound(to: UInt8.self))
    }
}

public final class GpuEvaluationKey: @unchecked Sendable {
    /// Galois elements in ascending order and their keys, parallel arrays as he_pir_expand_device takes them.
    public let galoisElements: [UInt64]
    public let galoisKeys: [DeviceKeySwitchKey]
    public let relinearizationKey: DeviceKeySwitchKey?

    public init<Scheme: HeScheme>(_ evaluationKey: EvaluationKey<Scheme>) throws
        where Scheme.Scalar == UInt64, Scheme.KeySwitchKey == _KeySwitchKey<Scheme>
    {
        let stream = try HeAmdStream()
        let galois = (evaluationKey._galoisKey?._keys ?? [:]).sorted { $0.key < $1.key } // Keys.swift:143-149
        galoisElements = galois.map { UInt64($0.key) }
        galoisKeys = try galois.map { try DeviceKeySwitchKey($0.value, on: stream) }
        relinearizationKey = try evaluationKey._relinearizationKey.map { // Keys.swift:108-117
            try DeviceKeySwitchKey($0._keySwitchKey, on: stream)
        }
    }

    var galoisPointers: [UnsafePointer<UInt64>?] {
        galoisKeys.map { UnsafePointer($0.buffer.pointer) }
    }
}

public final class GpuResidentCache: @unchecked Sendable {
    public static let shared = GpuResidentCache()

    struct StorageKey: Hashable {
        let device: Int32
        let storage: UInt
        let count: Int
        let fingerprint: UInt64
    }

    private let lock = NSLock()
    private var databases: [StorageKey: AnyObject] = [:]
    private var keys: [StorageKey: GpuEvaluationKey] = [:]

    /// Identity of a polynomial's storage: where its words live and what the first and last of them are.
    private static func identity<F: PolyFormat>(of poly: PolyRq<UInt64, F>) -> (UInt, UInt64) {
        poly.data.withDataSpan { span in
            span.withUnsafeBufferPointer { words in
                (UInt(bitPattern: words.baseAddress), (words.first ?? 0) &* 0x9E37_79B9_7F4A_7C15 ^ (words.last ?? 0))
            }
        }
    }

    private static func key<Scheme: HeScheme>(for database: ProcessedDatabase<Scheme>) throws -> StorageKey
        where Scheme.Scalar == UInt64
    {
        let device = try GpuContextCache.currentDevice()
        guard let first = database.plaintexts.lazy.compactMap(\.self).first else {
            return StorageKey(device: device, storage: 0, count: database.plaintexts.count, fingerprint: 0)
        }
        let (storage, fingerprint) = identity(of: first._poly)
        return StorageKey(device: device, storage: storage, count: database.plaintexts.count, fingerprint: fingerprint)
    }

    /// The resident copy of `database` on the current device, uploaded on first sight.
    public func resident<Scheme: HeScheme>(_ database: ProcessedDatabase<Scheme>,
                                           polyContext: PolyContext<UInt64>) throws -> GpuResidentDatabase<Scheme>
        where Scheme.Scalar == UInt64
    {
        let key = try Self.key(for: database)
        lock.lock()
        if let cached = databases[key] as? GpuResidentDatabase<Scheme> {
            lock.unlock()
            return cached
        }
        lock.unlock()
        let uploaded = try GpuResidentDatabase(database, polyContext: polyContext) // (outside the lock: seconds of PCIe)
        lock.lock()
        defer { lock.unlock() }
        if let raced = databases[key] as? GpuResidentDatabase<Scheme> { return raced }
        databases[key] = uploaded
        return uploaded
    }

    /// The resident copy of `evaluationKey` on the current device.
    public func resident<Scheme: HeScheme>(_ evaluationKey: EvaluationKey<Scheme>) throws -> GpuEvaluationKey
        where Scheme.Scalar == UInt64, Scheme.KeySwitchKey == _KeySwitchKey<Scheme>
    {
        let device = try GpuContextCache.currentDevice()
        let galois = evaluationKey._galoisKey?._keys ?? [:]
        let anyKey = evaluationKey._relinearizationKey?._keySwitchKey ?? galois.min { $0.key < $1.key }?.value
        guard let sample = anyKey?._ciphertexts.first?.polys.first else {
            return try GpuEvaluationKey(evaluationKey) // nothing to key on: an empty evaluation key
        }
        let (storage, fingerprint) = Self.identity(of: sample)
        let key = StorageKey(device: device, storage: storage, count: galois.count, fingerprint: fingerprint)
        lock.lock()
        if let cached = keys[key] {
            lock.unlock()
            return cached
        }
        lock.unlock()
        let uploaded = try GpuEvaluationKey(evaluationKey)
        lock.lock()
        defer { lock.unlock() }
        if let raced = keys[key] { return raced }
        keys[key] = uploaded
        return uploaded
    }

    /// Frees the resident copy of `database` (its HBM returns when the last response in flight is done).
    public func evict<Scheme: HeScheme>(_ database: ProcessedDatabase<Scheme>) throws where Scheme.Scalar == UInt64 {
        let key = try Self.key(for: database)
        lock.lock()
        defer { lock.unlock() }
        databases[key] = nil
    }

    public func removeAll() {
        lock.lock()
        defer { lock.unlock() }
        databases.removeAll()
        keys.removeAll()
        _ = he_device_trim_scratch(0)
    }
}
