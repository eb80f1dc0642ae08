// What a PIR server keeps in HBM between queries, and the caches that find it again from the reference's VALUE types:
//   * GpuResidentDatabase   a ProcessedDatabase (IndexPirProtocol.swift:249-290): every Eval plaintext of every chunk
//                           plus the nil mask, uploaded once;
//   * GpuEvaluationKey      an EvaluationKey (Keys.swift:186-219): its Galois keys by element and its relinearization
//                           key in the layout the C ABI takes;
//   * GpuResidentCache      ProcessedDatabase and EvaluationKey are structs (no object identity), so residency is keyed by
//                           a fingerprint of their copy-on-write STORAGE -- addresses and words of polynomials sampled
//                           over the whole value, the first and the last always among them -- and by the HIP device, or
//                           by a token the server owns.  Evaluation keys are kept under a byte / entry budget (least
//                           recently used first); see GpuResidentCache.StorageKey for what a fingerprint cannot see.
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

    /// Uploads `database` once (MulPir.swift:547-555 fixes the order: plaintext k of column c of a chunk at c * d0 + k),
    /// in pinned blocks of 256 MiB: one copy and one wait per block.
    public init(_ database: ProcessedDatabase<Scheme>, polyContext: PolyContext<UInt64>) throws {
        let polyWords = polyContext.moduli.count * polyContext.degree
        let stream = try HeAmdStream()
        plaintextCount = database.plaintexts.count
        plaintexts = try DeviceBuffer(count: plaintextCount * polyWords)
        present = try DeviceBuffer(count: (plaintextCount + 7) / 8) // the mask travels as bytes; buffers count 8-byte words
        let mask = try plaintexts.upload(plaintexts: database.plaintexts, polyWords: polyWords, on: stream)
        try mask.withUnsafeBufferPointer { bytes in
            try present.upload(bytes: bytes, atByte: 0, on: stream)
        }
    }

    /// Bytes of HBM the database occupies.
    public var byteCount: Int {
        (plaintexts.count + present.count) * MemoryLayout<UInt64>.stride
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
        // The uploads were only enqueued, on a stream that dies with this initialiser, and the kernels that read the keys
        // run on OTHER non-blocking streams with no ordering against it: wait here, so that a resident key is a landed
        // key, then hand back the page-locked copies (up to the cache's budget of unswappable host memory otherwise).
        try stream.synchronize()
        for key in galoisKeys { key.uploadCompleted() }
        relinearizationKey?.uploadCompleted()
    }

    var galoisPointers: [UnsafePointer<UInt64>?] {
        galoisKeys.map { UnsafePointer($0.buffer.pointer) }
    }

    /// Bytes of HBM the keys occupy.
    public var byteCount: Int {
        galoisKeys.reduce(relinearizationKey?.byteCount ?? 0) { $0 + $1.byteCount }
    }
}

public final class GpuResidentCache: @unchecked Sendable {
    public static let shared = GpuResidentCache()

    /// What residency is keyed on.  ProcessedDatabase and EvaluationKey are VALUE types, so there is no object identity to
    /// key on; the key is a fingerprint of the value's storage: for up to `sampleCount` polynomials spread over the whole
    /// value -- always including the first and the LAST one -- the address of the words and their first, middle and last
    /// word, plus the element count and the HIP device.
    ///
    /// **Staleness.**  A fingerprint is not the content.  A database that is freed and replaced by one of the same size
    /// whose storage lands on the same addresses AND agrees in every sampled word would be answered from the old copy in
    /// HBM.  Updating a database in place (copy-on-write gives the new value new storage for every plaintext it touched)
    /// or loading an unrelated one changes some sampled address or word with overwhelming probability, but a server that
    /// replaces databases should say so: `replace(_:with:polyContext:)` / `evict(_:)` drop the old entry explicitly, and
    /// `residentDatabase(forToken:)` / `register(_:token:polyContext:)` key residency on a token the server owns instead.
    struct StorageKey: Hashable {
        let device: Int32
        let count: Int
        let fingerprint: [UInt64]
    }

    /// How many polynomials of a value are sampled into its key.
    public static let sampleCount = 16

    /// Upper bound on the evaluation keys kept resident (least recently used ones go first) in bytes and in entries: a PIR
    /// server sees a new key per client, tens of megabytes each.  Both can be changed at any time.
    public var evaluationKeyBudgetBytes = 8 << 30
    public var evaluationKeyBudgetCount = 256

    private let lock = NSLock()
    private var databases: [StorageKey: AnyObject] = [:]
    private var tokenDatabases: [String: AnyObject] = [:]
    private var keys: [StorageKey: GpuEvaluationKey] = [:]
    private var keyUse: [StorageKey: UInt64] = [:] // last use, on a logical clock
    private var clock: UInt64 = 0

    /// Where the words of `poly` live, and its first, middle and last word.
    private static func sample<F: PolyFormat>(of poly: PolyRq<UInt64, F>, into fingerprint: inout [UInt64]) {
        poly.data.withDataSpan { span in
            span.withUnsafeBufferPointer { words in
                fingerprint.append(UInt64(UInt(bitPattern: words.baseAddress)))
                fingerprint.append(words.first ?? 0)
                fingerprint.append(words.isEmpty ? 0 : words[words.count / 2])
                fingerprint.append(words.last ?? 0)
            }
        }
    }

    /// Indices of up to `sampleCount` of `count` elements, evenly spread, the first and the last always among them.
    static func sampleIndices(count: Int) -> [Int] {
        guard count > sampleCount else { return Array(0..<count) }
        return (0..<sampleCount).map { $0 * (count - 1) / (sampleCount - 1) }
    }

    private static func key<Scheme: HeScheme>(for database: ProcessedDatabase<Scheme>) throws -> StorageKey
        where Scheme.Scalar == UInt64
    {
        let device = try GpuContextCache.currentDevice()
        // a lookup happens on every query: sixteen positions spread over the list, each moved on to the next plaintext that
        // is not nil (the last one back to the previous) -- never a pass over the whole database
        let plaintexts = database.plaintexts
        var fingerprint: [UInt64] = []
        for position in sampleIndices(count: plaintexts.count) {
            var index = position
            while index < plaintexts.count, plaintexts[index] == nil { index += 1 }
            if index == plaintexts.count {
                index = position
                while index > 0, plaintexts[index] == nil { index -= 1 }
            }
            guard let plaintext = plaintexts[index] else { continue }
            fingerprint.append(UInt64(index))
            sample(of: plaintext._poly, into: &fingerprint)
        }
        return StorageKey(device: device, count: plaintexts.count, fingerprint: fingerprint)
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

    /// Residency under a name the server owns (no fingerprint involved): uploads `database` and files it under `token`,
    /// replacing -- and thereby freeing -- whatever the token held before.
    @discardableResult
    public func register<Scheme: HeScheme>(_ database: ProcessedDatabase<Scheme>, token: String,
                                           polyContext: PolyContext<UInt64>) throws -> GpuResidentDatabase<Scheme>
        where Scheme.Scalar == UInt64
    {
        let device = try GpuContextCache.currentDevice()
        let uploaded = try GpuResidentDatabase(database, polyContext: polyContext)
        lock.lock()
        defer { lock.unlock() }
        tokenDatabases["\(device):\(token)"] = uploaded
        return uploaded
    }

    /// The database filed under `token` on the current device, if any.
    public func residentDatabase<Scheme: HeScheme>(forToken token: String,
                                                   scheme _: Scheme.Type) throws -> GpuResidentDatabase<Scheme>?
        where Scheme.Scalar == UInt64
    {
        let device = try GpuContextCache.currentDevice()
        lock.lock()
        defer { lock.unlock() }
        return tokenDatabases["\(device):\(token)"] as? GpuResidentDatabase<Scheme>
    }

    /// Drops the database filed under `token` on the current device.
    public func evict(token: String) throws {
        let device = try GpuContextCache.currentDevice()
        lock.lock()
        defer { lock.unlock() }
        tokenDatabases["\(device):\(token)"] = nil
    }

    /// A server's database update in one step: the old value's resident copy is dropped, the new one uploaded.
    @discardableResult
    public func replace<Scheme: HeScheme>(_ old: ProcessedDatabase<Scheme>, with new: ProcessedDatabase<Scheme>,
                                          polyContext: PolyContext<UInt64>) throws -> GpuResidentDatabase<Scheme>
        where Scheme.Scalar == UInt64
    {
        try evict(old)
        return try resident(new, polyContext: polyContext)
    }

    private static func key<Scheme: HeScheme>(for evaluationKey: EvaluationKey<Scheme>) throws -> StorageKey?
        where Scheme.Scalar == UInt64, Scheme.KeySwitchKey == _KeySwitchKey<Scheme>
    {
        let device = try GpuContextCache.currentDevice()
        let galois = (evaluationKey._galoisKey?._keys ?? [:]).sorted { $0.key < $1.key }
        var switchKeys = galois.map(\.value)
        if let relinearization = evaluationKey._relinearizationKey?._keySwitchKey { switchKeys.append(relinearization) }
        // every key-switching key contributes its first and its last polynomial
        var fingerprint: [UInt64] = galois.map { UInt64($0.key) }
        for switchKey in switchKeys {
            if let first = switchKey._ciphertexts.first?.polys.first { sample(of: first, into: &fingerprint) }
            if let last = switchKey._ciphertexts.last?.polys.last { sample(of: last, into: &fingerprint) }
        }
        guard !switchKeys.isEmpty else { return nil } // nothing to key on: an empty evaluation key
        return StorageKey(device: device, count: switchKeys.count, fingerprint: fingerprint)
    }

    /// The resident copy of `evaluationKey` on the current device; least recently used keys beyond the budget are dropped
    /// (their HBM returns when the last call that holds them is done).
    public func resident<Scheme: HeScheme>(_ evaluationKey: EvaluationKey<Scheme>) throws -> GpuEvaluationKey
        where Scheme.Scalar == UInt64, Scheme.KeySwitchKey == _KeySwitchKey<Scheme>
    {
        guard let key = try Self.key(for: evaluationKey) else {
            return try GpuEvaluationKey(evaluationKey)
        }
        lock.lock()
        clock += 1
        if let cached = keys[key] {
            keyUse[key] = clock
            lock.unlock()
            return cached
        }
        lock.unlock()
        let uploaded = try GpuEvaluationKey(evaluationKey)
        lock.lock()
        defer { lock.unlock() }
        if let raced = keys[key] { return raced }
        keys[key] = uploaded
        keyUse[key] = clock
        trimKeys(keeping: key)
        return uploaded
    }

    /// Drops least recently used evaluation keys until both budgets hold (the lock is held by the caller).
    private func trimKeys(keeping newest: StorageKey) {
        var bytes = keys.values.reduce(0) { $0 + $1.byteCount }
        while keys.count > 1, keys.count > evaluationKeyBudgetCount || bytes > evaluationKeyBudgetBytes {
            guard let oldest = keyUse.filter({ $0.key != newest }).min(by: { $0.value < $1.value })?.key else { break }
            bytes -= keys[oldest]?.byteCount ?? 0
            keys[oldest] = nil
            keyUse[oldest] = nil
        }
    }

    /// Frees the resident copy of `evaluationKey` (a session that has ended).
    public func evict<Scheme: HeScheme>(_ evaluationKey: EvaluationKey<Scheme>) throws
        where Scheme.Scalar == UInt64, Scheme.KeySwitchKey == _KeySwitchKey<Scheme>
    {
        guard let key = try Self.key(for: evaluationKey) else { return }
        lock.lock()
        defer { lock.unlock() }
        keys[key] = nil
        keyUse[key] = nil
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
        tokenDatabases.removeAll()
        keys.removeAll()
        keyUse.removeAll()
        _ = he_device_trim_scratch(0)
    }
}
