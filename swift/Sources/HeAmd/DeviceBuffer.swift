// Device-resident slabs in the reference's own layout: a polynomial is its Array2d row-major (moduli x N) words
// (Sources/HomomorphicEncryption/Array2d.swift:117-119), a ciphertext its polynomials back to back, a batch its
// ciphertexts back to back -- exactly what the C ABI documents, so upload / download are plain copies.  Everything here
// is generic over the scheme (any HeScheme whose Scalar is UInt64): `Bfv<UInt64>` and `GpuBfv` share it.
import CHeAmd
import HomomorphicEncryption

/// `count` UInt64 words of HBM (`he_device_malloc`), freed on deinit.
public final class DeviceBuffer: @unchecked Sendable {
    public let pointer: UnsafeMutablePointer<UInt64>
    public let count: Int

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

    /// Copies `words` to word offset `offset`.  The source is a pointer borrowed for the duration of a closure and
    /// `he_memcpy_h2d` is an asynchronous copy from pageable memory, so the copy is waited for before the pointer goes
    /// out of scope (the C side does the same for its borrowed host masks, csrc/pir_api.cpp).
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

    /// Copies the words of `poly` to word offset `offset`.
    public func upload<F: PolyFormat>(_ poly: PolyRq<UInt64, F>, at offset: Int, on stream: HeAmdStream) throws {
        try poly.data.withDataSpan { span in // Array2d.swift:95: the storage, row-major
            try span.withUnsafeBufferPointer { words in
                try upload(words: words, at: offset, on: stream)
            }
        }
    }

    /// All polynomials of `ciphertext`, back to back, starting at word offset `offset`.
    public func upload<S: HeScheme, F: PolyFormat>(_ ciphertext: Ciphertext<S, F>, at offset: Int,
                                                   on stream: HeAmdStream) throws where S.Scalar == UInt64
    {
        var cursor = offset
        for poly in ciphertext.polys { // Ciphertext.swift:23
            try upload(poly, at: cursor, on: stream)
            cursor += poly.data.count
        }
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
    /// carries (Ciphertext.swift:64-84, `clearSeed()` at the end of each operation).
    public func downloadCiphertext<S: HeScheme, F: PolyFormat>(
        context: S.Context, polyContext: PolyContext<UInt64>, polyCount: Int, at offset: Int,
        correctionFactor: UInt64 = 1, on stream: HeAmdStream) throws -> Ciphertext<S, F> where S.Scalar == UInt64
    {
        let stride = polyContext.moduli.count * polyContext.degree
        let polys: [PolyRq<UInt64, F>] = try (0..<polyCount).map { index in
            try downloadPoly(context: polyContext, at: offset + index * stride, on: stream)
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
        var cursor = 0
        for ciphertext in ciphertexts {
            try buffer.upload(ciphertext, at: cursor, on: stream)
            cursor += ciphertext.polys.reduce(0) { $0 + $1.data.count }
        }
    }
}
