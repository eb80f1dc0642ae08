// Device-resident slabs in the reference's own layout: a polynomial is its Array2d row-major (moduli x N) words
// (Sources/HomomorphicEncryption/Array2d.swift:117-119), a ciphertext its polynomials back to back, a batch its
// ciphertexts back to back -- exactly what the C ABI documents, so upload / download are plain copies.
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

    /// Copies the words of `poly` to word offset `offset`; enqueued on `stream`.
    public func upload<F: PolyFormat>(_ poly: PolyRq<UInt64, F>, at offset: Int, on stream: HeAmdStream) throws {
        try poly.data.withDataSpan { span in // Array2d.swift:95: the storage, row-major
            try span.withUnsafeBufferPointer { words in
                precondition(offset + words.count <= count)
                try heAmdCheck(he_memcpy_h2d(pointer + offset, words.baseAddress,
                                             words.count * MemoryLayout<UInt64>.stride, stream.raw))
            }
        }
    }

    /// All polynomials of `ciphertext`, back to back, starting at word offset `offset`.
    public func upload<F: PolyFormat>(_ ciphertext: Ciphertext<Bfv<UInt64>, F>, at offset: Int,
                                      on stream: HeAmdStream) throws
    {
        var cursor = offset
        for poly in ciphertext.polys { // Ciphertext.swift:23
            try upload(poly, at: cursor, on: stream)
            cursor += poly.data.count
        }
    }

    /// Reads `rowCount * degree` words back as one polynomial over `context`.  Waits for the copy.
    public func downloadPoly<F: PolyFormat>(context: PolyContext<UInt64>, at offset: Int,
                                            on stream: HeAmdStream) throws -> PolyRq<UInt64, F>
    {
        let wordCount = context.moduli.count * context.degree
        var words = [UInt64](repeating: 0, count: wordCount)
        try words.withUnsafeMutableBufferPointer { destination in
            try heAmdCheck(he_memcpy_d2h(destination.baseAddress, pointer + offset,
                                         wordCount * MemoryLayout<UInt64>.stride, stream.raw))
        }
        try stream.synchronize()
        let data = Array2d(data: words, rowCount: context.moduli.count, columnCount: context.degree)
        return PolyRq(context: context, data: data) // PolyRq.swift:31
    }

    /// Reads `polyCount` polynomials back as a ciphertext with correction factor 1 and no seed, as every
    /// evaluation result of the reference carries (Ciphertext.swift:64-84).
    public func downloadCiphertext<F: PolyFormat>(context: Context<Bfv<UInt64>>, polyContext: PolyContext<UInt64>,
                                                  polyCount: Int, at offset: Int,
                                                  on stream: HeAmdStream) throws -> Ciphertext<Bfv<UInt64>, F>
    {
        let stride = polyContext.moduli.count * polyContext.degree
        let polys: [PolyRq<UInt64, F>] = try (0..<polyCount).map { index in
            try downloadPoly(context: polyContext, at: offset + index * stride, on: stream)
        }
        return try Ciphertext(_context: context, _polys: polys, _correctionFactor: 1, _auxiliaryData: nil)
    }
}
