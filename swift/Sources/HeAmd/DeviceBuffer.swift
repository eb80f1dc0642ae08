// Device-resident slabs in the reference's own layout: a polynomial is its Array2d row-major (moduli x N) words
// (Sources/HomomorphicEncryption/Array2d.swift:117-119), a ciphertext its polynomials back to back, a batch its
// ciphertexts back to back -- exactly what the C ABI documents, so upload / download are plain copies.  Everything here
// is generic over the scheme (any HeScheme whose Scalar is UInt64): `Bfv<UInt64>` and `GpuBfv` share it.
import CHeAmd
import HomomorphicEncryption

/// `capacity` UInt64 words of page-locked host memory (`he_host_malloc`): where the polynomials of a ciphertext, a key or a
/// block of a database are laid back to back before they go up in ONE asynchronous copy.  A copy out of pinned memory
/// needs no wait before the next call is enqueued behind it on the same stream; the staging only has to outlive the copy,
/// so the `DeviceBuffer` it was uploaded to keeps it (see `DeviceBuffer.upload(staged:at:on:)`).
public final class HostStaging: @unchecked Sendable {
    public let pointer: UnsafeMutablePointer<UInt64>
    public let capacity: Int
    private var used = 0
    /// Words appended so far.
    public var count: Int { used }

    public init(capacity: Int) throws {
        var raw: UnsafeMutableRawPointer?
        try heAmdCheck(he_host_malloc(&raw, max(capacity, 1) * MemoryLayout<UInt64>.stride))
        guard let raw else { throw HeError.unsupportedHeOperation(description: "he_host_malloc returned nil") }
        pointer = raw.bindMemory(to: UInt64.self, capacity: max(capacity, 1))
        self.capacity = capacity
    }

    deinit {
        _ = he_host_free(pointer)
    }

    /// Appends `words`; returns the word offset they were put at.
    @discardableResult
    public func append(words: UnsafeBufferPointer<UInt64>) -> Int {
        precondition(used + words.count <= capacity)
        let offset = used
        if let source = words.baseAddress {
            (pointer + offset).update(from: source, count: words.count)
        }
        used += words.count
        return offset
    }

    /// Appends the row-major words of `poly` (Array2d.swift:95, :117-119).
    @discardableResult
    public func append<F: PolyFormat>(_ poly: PolyRq<UInt64, F>) -> Int {
        poly.data.withDataSpan { span in
            span.withUnsafeBufferPointer { words in append(words: words) }
        }
    }

    /// Appends all polynomials of `ciphertext`, back to back (Ciphertext.swift:23).
    @discardableResult
    public func append<S: HeScheme, F: PolyFormat>(_ ciphertext: Ciphertext<S, F>) -> Int where S.Scalar == UInt64 {
        let offset = count
        for poly in ciphertext.polys { append(poly) }
        return offset
    }

    /// Forgets the appended words (after the copy that read them has been waited for).
    public func reset() {
        used = 0
    }
}

/// `count` UInt64 words of HBM (`he_device_malloc`), freed on deinit.
public final class DeviceBuffer: @unchecked Sendable {
    public let pointer: UnsafeMutablePointer<UInt64>
    public let count: Int
    /// Pinned sources of copies that may still be in flight; released with the buffer or by `releaseStaging()`.
    private var staged: [HostStaging] = []

    public init(count: Int) throws {
        var raw: UnsafeMutableRawPointer?
        try heAmdCheck(he_device_malloc(&raw, max(count, 1) * MemoryLayout<UInt64>.stride))
        guard let raw else { throw HeError.unsupportedHeOperation(description: "he_device_malloc returned nil") }
        pointer = raw.bindMemory(to: UInt64.self, capacity: max(count, 1))
        self.count = count
    }

    deinit {
        _ = he_device_free(pointer)
    }

    /// ONE asynchronous copy of everything appended to `staging` to word offset `offset`; no wait.  The buffer keeps the
    /// staging alive (a `DeviceWork` holds its buffers until the result has been read back, i.e. past the copy).
    public func upload(staged staging: HostStaging, at offset: Int, on stream: HeAmdStream) throws {
        precondition(offset + staging.count <= count)
        try heAmdCheck(he_memcpy_h2d(pointer + offset, staging.pointer, staging.count * MemoryLayout<UInt64>.stride,
                                     stream.raw))
        staged.append(staging)
    }

    /// Drops the pinned sources once the stream they were copied on has been waited for.
    public func releaseStaging() {
        staged.removeAll()
    }

    /// Copies `words` to word offset `offset` from a pointer borrowed for the duration of a closure: `he_memcpy_h2d` is an
    /// asynchronous copy from pageable memory here, so the copy is waited for before the pointer goes out of scope (the C
    /// side does the same for its borrowed host masks, csrc/pir_api.cpp).  For single small objects; anything made of
    /// several polynomials goes through a `HostStaging`.
    public func upload(words: UnsafeBufferPointer<UInt64>, at offset: Int, on stream: HeAmdStream) throws {
        precondition(offset + words.count <= count)
        try heAmdCheck(he_memcpy_h2d(pointer + offset, words.baseAddress, words.count * MemoryLayout<UInt64>.stride,
                                     stream.raw))
        try heAmdCheck(he_stream_synchronize(stream.raw))
    }

    /// Copies `bytes` (a mask, a serialized record) to byte offset `byteOffset`; waits like `upload(words:)`.
    public func upload(bytes: UnsafeBufferPointer<UInt8>, atByte byteOffset: Int, on stream: HeAmdStream) throws {
        precondition(byteOffset + bytes.count <= count * MemoryLayout<UInt64>.stride)
        try heAmdCheck(he_memcpy_h2d(UnsafeMutableRawPointer(pointer) + byteOffset, bytes.baseAddress, bytes.count,
                                     stream.raw))
        try heAmdCheck(he_stream_synchronize(stream.raw))
    }

    /// Copies the words of `poly` to word offset `offset`.  A single polynomial goes through the borrowed-pointer path
    /// (`upload(words:)`: copy, then wait): page-locking a block per small object costs more than the wait it saves --
    /// `hipHostMalloc` is slow and `hipHostFree` synchronizes the device.
    public func upload<F: PolyFormat>(_ poly: PolyRq<UInt64, F>, at offset: Int, on stream: HeAmdStream) throws {
        try poly.data.withDataSpan { span in
            try span.withUnsafeBufferPointer { words in try upload(words: words, at: offset, on: stream) }
        }
    }

    /// All polynomials of `ciphertext`, back to back, starting at word offset `offset` (a single ciphertext: the
    /// borrowed-pointer path, polynomial by polynomial).
    public func upload<S: HeScheme, F: PolyFormat>(_ ciphertext: Ciphertext<S, F>, at offset: Int,
                                                   on stream: HeAmdStream) throws where S.Scalar == UInt64
    {
        var at = offset
        for poly in ciphertext.polys {
            try upload(poly, at: at, on: stream)
            at += poly.data.count
        }
    }

    /// `ciphertexts` back to back from word offset `offset` (a query, the operands of an inner product, a key): ONE staged
    /// copy, no wait.  Any collection of ciphertexts (the PIR protocol hands over generic `Collection`s).  The staging
    /// stays with the buffer until `releaseStaging()` -- call it once the stream has been waited for.
    public func upload<S: HeScheme, F: PolyFormat>(contentsOf ciphertexts: some Collection<Ciphertext<S, F>>, at offset: Int,
                                                   on stream: HeAmdStream) throws where S.Scalar == UInt64
    {
        let words = ciphertexts.reduce(0) { sum, ciphertext in sum + ciphertext.polys.reduce(0) { $0 + $1.data.count } }
        let staging = try HostStaging(capacity: words)
        for ciphertext in ciphertexts { staging.append(ciphertext) }
        try upload(staged: staging, at: offset, on: stream)
    }

    /// A run of optional Eval plaintexts (a database, a chunk of one: IndexPirProtocol.swift:249-290) to word offset 0,
    /// plaintext k at k * polyWords, through ONE reusable pinned block of at most `blockBytes`: one copy and one wait per
    /// block instead of one per plaintext.  Returns the presence mask (0 = nil, Bfv.swift:486-489); the words of a nil
    /// plaintext are never used by the kernels (its mask byte travels with them) and are left as they are.
    public func upload<S: HeScheme>(plaintexts: some Collection<Plaintext<S, Eval>?>, polyWords: Int, on stream: HeAmdStream,
                                    blockBytes: Int = 256 << 20) throws -> [UInt8] where S.Scalar == UInt64
    {
        precondition(plaintexts.count * polyWords <= count)
        let perBlock = max(1, blockBytes / (polyWords * MemoryLayout<UInt64>.stride))
        let staging = try HostStaging(capacity: min(perBlock, max(plaintexts.count, 1)) * polyWords)
        var present = [UInt8](repeating: 0, count: plaintexts.count)
        var blockStart = 0, inBlock = 0
        func flush() throws {
            guard inBlock > 0 else { return }
            try heAmdCheck(he_memcpy_h2d(pointer + blockStart * polyWords, staging.pointer,
                                         inBlock * polyWords * MemoryLayout<UInt64>.stride, stream.raw))
            try heAmdCheck(he_stream_synchronize(stream.raw)) // the block is reused
            blockStart += inBlock
            inBlock = 0
        }
        for (index, plaintext) in plaintexts.enumerated() {
            if let plaintext {
                present[index] = 1
                plaintext._poly.data.withDataSpan { span in // Plaintext.swift:28, Array2d.swift:95
                    span.withUnsafeBufferPointer { words in
                        (staging.pointer + inBlock * polyWords).update(from: words.baseAddress!, count: polyWords)
                    }
                }
            }
            inBlock += 1
            if inBlock == perBlock { try flush() }
        }
        try flush()
        return present
    }

    /// Reads `rowCount * degree` words back as one polynomial over `context`.  Waits for the copy before the destination
    /// pointer goes out of scope.
    public func downloadPoly<F: PolyFormat>(context: PolyContext<UInt64>, at offset: Int,
                                            on stream: HeAmdStream) throws -> PolyRq<UInt64, F>
    {
        let wordCount = context.moduli.count * context.degree
        var words = [UInt64](repeating: 0, count: wordCount)
        try words.withUnsafeMutableBufferPointer { destination in
            try heAmdCheck(he_memcpy_d2h(destination.baseAddress, pointer + offset,
                                         wordCount * MemoryLayout<UInt64>.stride, stream.raw))
            try heAmdCheck(he_stream_synchronize(stream.raw))
        }
        let data = Array2d(data: words, rowCount: context.moduli.count, columnCount: context.degree)
        return PolyRq(context: context, data: data) // PolyRq.swift:31
    }

    /// Reads `polyCount` polynomials back as a ciphertext with no seed, as every evaluation result of the reference
    /// carries (Ciphertext.swift:64-84, `clearSeed()` at the end of each operation): ONE copy and one wait for all of them.
    public func downloadCiphertext<S: HeScheme, F: PolyFormat>(
        context: S.Context, polyContext: PolyContext<UInt64>, polyCount: Int, at offset: Int,
        correctionFactor: UInt64 = 1, on stream: HeAmdStream) throws -> Ciphertext<S, F> where S.Scalar == UInt64
    {
        let stride = polyContext.moduli.count * polyContext.degree
        var words = [UInt64](repeating: 0, count: polyCount * stride)
        try words.withUnsafeMutableBufferPointer { destination in
            try heAmdCheck(he_memcpy_d2h(destination.baseAddress, pointer + offset,
                                         polyCount * stride * MemoryLayout<UInt64>.stride, stream.raw))
            try heAmdCheck(he_stream_synchronize(stream.raw))
        }
        releaseStaging() // the stream has been waited for: every copy into this buffer is done
        let polys: [PolyRq<UInt64, F>] = (0..<polyCount).map { index in
            let data = Array2d(data: Array(words[index * stride..<(index + 1) * stride]),
                               rowCount: polyContext.moduli.count, columnCount: polyContext.degree)
            return PolyRq(context: polyContext, data: data) // PolyRq.swift:31
        }
        return try Ciphertext(_context: context, _polys: polys, _correctionFactor: correctionFactor,
                              _auxiliaryData: nil)
    }
}

/// An evaluation key's key-switching key resident on the device in the layout he_bfv_relinearize_device /
/// he_bfv_apply_galois_device take: [L][2][L+1][N] Eval -- the ciphertexts of `_KeySwitchKey` (Keys.swift:66-99) back
/// to back, each two polynomials over the key-switching context.
public final class DeviceKeySwitchKey: @unchecked Sendable {
    public let buffer: DeviceBuffer

    public init<S: HeScheme>(_ key: _KeySwitchKey<S>, on stream: HeAmdStream) throws where S.Scalar == UInt64 {
        let ciphertexts = key._ciphertexts
        let words = ciphertexts.reduce(0) { sum, ct in sum + ct.polys.reduce(0) { $0 + $1.data.count } }
        buffer = try DeviceBuffer(count: words)
        try buffer.upload(contentsOf: ciphertexts, at: 0, on: stream) // the whole key: one staged copy
    }

    /// The key is resident: call once the upload stream has been waited for.  Drops the page-locked copy of the key (a
    /// resident key would otherwise pin its own size in host memory for as long as it stays in the cache).
    public func uploadCompleted() {
        buffer.releaseStaging()
    }

    /// Bytes of HBM the key occupies.
    public var byteCount: Int {
        buffer.count * MemoryLayout<UInt64>.stride
    }
}
