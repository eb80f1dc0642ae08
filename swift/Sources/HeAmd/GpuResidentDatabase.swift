// A processed PIR database kept in HBM, and PirUtilProtocol.computeResponse(to:using:databases:...) for it in one C call
// (Sources/PrivateInformationRetrieval/IndexPir/PirUtil.swift:490-568): the expansion of the query, the conversion of the
// dim-0 ciphertexts to Eval and every chunk stay on the device; the indices of a Query share the pass over the database
// four at a time.  GpuPirUtil.computeResponseForOneChunk uploads its chunk per call to keep the protocol's signature; a
// server that answers many queries builds one of these per ProcessedDatabase instead.
import CHeAmd
import HomomorphicEncryption
import PrivateInformationRetrieval

public final class GpuResidentDatabase: @unchecked Sendable {
    public typealias Scheme = Bfv<UInt64>

    /// [chunk][prod(dimensions)][L][N] Eval words, `nil` plaintexts left as zeros.
    public let plaintexts: DeviceBuffer
    /// One byte per plaintext, 0 = nil (Bfv.swift:486-489), on the device.
    public let present: DeviceBuffer
    public let parameter: IndexPirParameter
    public let context: Context<Scheme>
    public let chunkCount: Int

    /// Uploads `database` once.  `chunkCount` as MulPirServer computes it (MulPir.swift:361-364).
    public init(_ database: ProcessedDatabase<Scheme>, parameter: IndexPirParameter, context: Context<Scheme>,
                chunkCount: Int) throws
    {
        self.parameter = parameter
        self.context = context
        self.chunkCount = chunkCount
        let polyContext = context.ciphertextContext
        let polyWords = polyContext.moduli.count * polyContext.degree
        let stream = try HeAmdStream()
        plaintexts = try DeviceBuffer(count: database.plaintexts.count * polyWords)
        // the mask travels as bytes; DeviceBuffer counts 8-byte words
        present = try DeviceBuffer(count: (database.plaintexts.count + 7) / 8)
        var mask = [UInt8](repeating: 0, count: database.plaintexts.count)
        for (index, plaintext) in database.plaintexts.enumerated() {
            guard let plaintext else { continue }
            mask[index] = 1
            try plaintexts.upload(plaintext._poly, at: index * polyWords, on: stream) // Plaintext.swift:28
        }
        try mask.withUnsafeBufferPointer { bytes in
            try heAmdCheck(he_memcpy_h2d(present.pointer, bytes.baseAddress, bytes.count, stream.raw))
        }
        try stream.synchronize()
    }

    /// PirUtilProtocol.computeResponse(to:using:databases:parameter:context:callOptions:) with this one database.
    public func computeResponse(to query: Query<Scheme>,
                                using evaluationKey: EvaluationKey<Scheme>) async throws -> Response<Scheme>
    {
        guard let first = query.ciphertexts.first else {
            throw HeError.incompatibleCiphertextCount("empty query")
        }
        let polyContext = first.polys[0].context
        let degree = polyContext.degree, polyWords = polyContext.moduli.count * degree
        let stream = try HeAmdStream()

        let ciphertexts = try DeviceBuffer(count: query.ciphertexts.count * 2 * polyWords)
        for (index, ciphertext) in query.ciphertexts.enumerated() {
            try ciphertexts.upload(ciphertext, at: index * 2 * polyWords, on: stream)
        }
        // the evaluation key: Galois keys by element (Keys.swift:143-149), the relinearization key when the database
        // has more than one dimension (PirUtil.swift:448-479)
        let galois = (evaluationKey._galoisKey?._keys ?? [:]).sorted { $0.key < $1.key }
        let elements = galois.map { UInt64($0.key) }
        let galoisKeys = try galois.map { try DeviceKeySwitchKey($0.value, on: stream) }
        let galoisPointers: [UnsafePointer<UInt64>?] = galoisKeys.map { UnsafePointer($0.buffer.pointer) }
        var relinearization: DeviceKeySwitchKey?
        if parameter.dimensions.count > 1 {
            guard let key = evaluationKey._relinearizationKey else { throw HeError.missingRelinearizationKey }
            relinearization = try DeviceKeySwitchKey(key._keySwitchKey, on: stream)
        }

        let responses = try DeviceBuffer(count: query.indicesCount * chunkCount * 2 * degree) // [index][chunk][2][1][N]
        let dimensions = parameter.dimensions.map { UInt32($0) }
        let databases: [UnsafePointer<UInt64>?] = [UnsafePointer(plaintexts.pointer)]
        let masks: [UnsafePointer<UInt8>?] = [UnsafeRawPointer(present.pointer).assumingMemoryBound(to: UInt8.self)]
        try dimensions.withUnsafeBufferPointer { dims in
            try elements.withUnsafeBufferPointer { elementPointer in
                try galoisPointers.withUnsafeBufferPointer { keyPointers in
                    try databases.withUnsafeBufferPointer { databasePointers in
                        try masks.withUnsafeBufferPointer { maskPointers in
                            try heAmdCheck(he_pir_compute_response_to_query_device(
                                context.gpu, dims.baseAddress, UInt32(dims.count), ciphertexts.pointer,
                                query.ciphertexts.count, query.indicesCount, elementPointer.baseAddress,
                                keyPointers.baseAddress, galois.count, relinearization?.buffer.pointer,
                                databasePointers.baseAddress, maskPointers.baseAddress, 1, chunkCount,
                                responses.pointer, stream.raw))
                        }
                    }
                }
            }
        }
        try await stream.completion()
        // the single-modulus context modSwitchDownToSingle ends on (Bfv.swift:163-171)
        var single = polyContext
        while single.moduli.count > 1, let next = single.next { single = next }
        let responseWords = 2 * degree
        let perIndex: [[Ciphertext<Scheme, Coeff>]] = try (0..<query.indicesCount).map { index in
            try (0..<chunkCount).map { chunk in
                try responses.downloadCiphertext(context: context, polyContext: single, polyCount: 2,
                                                 at: (index * chunkCount + chunk) * responseWords, on: stream)
            }
        }
        return Response(ciphertexts: perIndex)
    }
}
