// Level B4: a PirUtilProtocol conformer (Sources/PrivateInformationRetrieval/IndexPir/PirUtil.swift:22-147) whose server
// side runs on the device.  Servers are generic over the conformer (IndexPir/MulPir.swift:292), so
// `MulPirServer<GpuPirUtil<Bfv<UInt64>>>` -- or `<GpuPirUtil<GpuBfv>>` -- is the drop-in.  Three requirements are
// overridden, everything else keeps the protocol's default implementation:
//   computeResponse(to:using:databases:parameter:context:callOptions:)   PirUtil.swift:490-568 -- the whole server side of a
//       Query in ONE C call over databases and evaluation keys that stay resident in HBM (GpuResidentCache): expansion,
//       dim-0 to Eval, every chunk; only the query goes up and the response comes down;
//   expand(ciphertexts:outputCount:using:callOptions:)                    PirUtil.swift:313-355 -- the oblivious expansion;
//   computeResponseForOneChunk(...)                                       PirUtil.swift:408-486 -- for callers that drive
//       chunks themselves: the chunk is a slice of a database the cache already holds, or is uploaded for this call.
import CHeAmd
import HomomorphicEncryption
import PrivateInformationRetrieval

public enum GpuPirUtil<Scheme: HeScheme>: PirUtilProtocol
    where Scheme.Scalar == UInt64, Scheme.CanonicalCiphertextFormat == Coeff, Scheme.KeySwitchKey == _KeySwitchKey<Scheme>
{
    public typealias Scalar = UInt64

    /// The single-modulus context `modSwitchDownToSingle` ends on (HeScheme.swift:1481-1485).
    static func singleModulusContext(of polyContext: PolyContext<UInt64>) -> PolyContext<UInt64> {
        var single = polyContext
        while single.moduli.count > 1, let next = single.next { single = next }
        return single
    }

    // MARK: computeResponse(to:using:databases:...) -- resident databases, one C call

    // swiftlint:disable:next function_parameter_count
    public static func computeResponse(
        to query: Query<Scheme>,
        using evaluationKey: EvaluationKey<Scheme>,
        databases: [ProcessedDatabase<Scheme>],
        parameter: IndexPirParameter,
        context: Scheme.Context,
        callOptions _: CallOptions) async throws -> Response<Scheme>
    {
        guard databases.count == 1 || databases.count >= query.indicesCount else { // PirUtil.swift:498-500
            throw PirError.invalidBatchSize(queryCount: query.indicesCount, databaseCount: databases.count)
        }
        guard let first = query.ciphertexts.first else {
            throw HeError.incompatibleCiphertextCount("empty query")
        }
        let polyContext = first.polys[0].context
        let degree = polyContext.degree, polyWords = polyContext.moduli.count * degree
        // chunks per database: every chunk holds prod(dimensions) plaintexts (PirUtil.swift:507, :536-537)
        let perChunk = parameter.dimensions.reduce(1, *)
        let chunkCount = databases[0].count / perChunk
        precondition(databases.allSatisfy { $0.count == chunkCount * perChunk })

        let resident = try databases.map { try GpuResidentCache.shared.resident($0, polyContext: polyContext) }
        let keys = try GpuResidentCache.shared.resident(evaluationKey)
        if parameter.dimensions.count > 1, keys.relinearizationKey == nil {
            throw HeError.missingRelinearizationKey // PirUtil.swift:448-479 relinearizes after every further dimension
        }

        let stream = try HeAmdStream()
        let ciphertexts = try DeviceBuffer(count: query.ciphertexts.count * 2 * polyWords)
        try ciphertexts.upload(contentsOf: query.ciphertexts, at: 0, on: stream) // the whole query: one staged copy
        let responses = try DeviceBuffer(count: query.indicesCount * chunkCount * 2 * degree) // [index][chunk][2][1][N]
        let dimensions = parameter.dimensions.map { UInt32($0) }
        let databasePointers: [UnsafePointer<UInt64>?] = resident.map { UnsafePointer($0.plaintexts.pointer) }
        let maskPointers: [UnsafePointer<UInt8>?] = resident.map(\.maskPointer)
        let galoisPointers = keys.galoisPointers
        let handle = try context.gpu
        try dimensions.withUnsafeBufferPointer { dims in
            try keys.galoisElements.withUnsafeBufferPointer { elements in
                try galoisPointers.withUnsafeBufferPointer { galois in
                    try databasePointers.withUnsafeBufferPointer { slabs in
                        try maskPointers.withUnsafeBufferPointer { masks in
                            try heAmdCheck(he_pir_compute_response_to_query_device(
                                handle, dims.baseAddress, UInt32(dims.count), ciphertexts.pointer,
                                query.ciphertexts.count, query.indicesCount, elements.baseAddress, galois.baseAddress,
                                galois.count, keys.relinearizationKey?.buffer.pointer, slabs.baseAddress,
                                masks.baseAddress, slabs.count, chunkCount, responses.pointer, stream.raw))
                        }
                    }
                }
            }
        }
        try await stream.completion()
        let single = singleModulusContext(of: polyContext)
        let responseWords = 2 * degree
        // (inputs, databases and keys stay alive until the results are back: the call above only enqueued the kernels)
        let perIndex: [[Scheme.CoeffCiphertext]] = try withExtendedLifetime((resident, keys, ciphertexts)) {
            try (0..<query.indicesCount).map { index in
                try (0..<chunkCount).map { chunk in
                    try responses.downloadCiphertext(context: context, polyContext: single, polyCount: 2,
                                                     at: (index * chunkCount + chunk) * responseWords, on: stream)
                }
            }
        }
        return Response(ciphertexts: perIndex)
    }

    // MARK: expand(ciphertexts:outputCount:using:callOptions:) -- the recursion tree, level by level, on the device

    public static func expand(
        ciphertexts: consuming [CanonicalCiphertext],
        outputCount: Int,
        using evaluationKey: EvaluationKey<Scheme>,
        callOptions _: CallOptions) async throws -> [CanonicalCiphertext]
    {
        precondition((ciphertexts.count - 1) * ciphertexts[0].degree < outputCount) // PirUtil.swift:325-326
        precondition(ciphertexts.count * ciphertexts[0].degree >= outputCount)
        let context = ciphertexts[0].context
        let polyContext = ciphertexts[0].polys[0].context
        let polyWords = polyContext.moduli.count * polyContext.degree
        let keys = try GpuResidentCache.shared.resident(evaluationKey)
        let stream = try HeAmdStream()
        let input = try DeviceBuffer(count: ciphertexts.count * 2 * polyWords)
        try input.upload(contentsOf: ciphertexts, at: 0, on: stream)
        let output = try DeviceBuffer(count: outputCount * 2 * polyWords)
        let galoisPointers = keys.galoisPointers
        let handle = try context.gpu
        let inputCount = ciphertexts.count
        try keys.galoisElements.withUnsafeBufferPointer { elements in
            try galoisPointers.withUnsafeBufferPointer { galois in
                try heAmdCheck(he_pir_expand_device(handle, input.pointer, inputCount, outputCount, elements.baseAddress,
                                                    galois.baseAddress, galois.count, output.pointer, stream.raw))
            }
        }
        try await stream.completion()
        return try withExtendedLifetime((keys, input)) {
            try (0..<outputCount).map { index in
                try output.downloadCiphertext(context: context, polyContext: polyContext, polyCount: 2,
                                              at: index * 2 * polyWords, on: stream)
            }
        }
    }

    // MARK: computeResponseForOneChunk -- one chunk, uploaded for the call

    // swiftlint:disable:next function_parameter_count
    public static func computeResponseForOneChunk<
        ExpandedQueries: Sendable & Collection<CanonicalCiphertext>,
        DataChunk: Sendable & Collection<Plaintext<Scheme, Eval>?>,
    >(
        expandedDim0Query: [Ciphertext<Scheme, Eval>],
        expandedRemainingQuery: ExpandedQueries,
        dataChunk: DataChunk,
        using evaluationKey: EvaluationKey<Scheme>,
        parameter: IndexPirParameter,
        callOptions _: CallOptions) async throws -> Ciphertext<Scheme, Coeff>
        where ExpandedQueries.Index == Int, DataChunk.Index == Int
    {
        guard let first = expandedDim0Query.first else {
            throw HeError.incompatibleCiphertextCount("empty dim-0 query")
        }
        let context = first.context
        let polyContext = first.polys[0].context
        let moduliCount = polyContext.moduli.count, degree = polyContext.degree
        let polyWords = moduliCount * degree
        let perChunk = parameter.dimensions.reduce(1, *)
        let columns = perChunk / parameter.dimensions[0]
        precondition(columns == 1 || columns == expandedRemainingQuery.count) // PirUtil.swift:422
        let stream = try HeAmdStream()

        let dim0 = try DeviceBuffer(count: expandedDim0Query.count * 2 * polyWords)
        try dim0.upload(contentsOf: expandedDim0Query, at: 0, on: stream)
        let rest = try DeviceBuffer(count: expandedRemainingQuery.count * 2 * polyWords)
        try rest.upload(contentsOf: expandedRemainingQuery, at: 0, on: stream)
        // The chunk: plaintext k of column c at index c * d0 + k (MulPir.swift:547-555); nil plaintexts are masked out.
        // This requirement hands over a slice by value, so the slice is uploaded; computeResponse(to:...) above is the
        // path that never moves the database.
        let database = try DeviceBuffer(count: perChunk * polyWords)
        let present = try database.upload(plaintexts: dataChunk.prefix(perChunk), polyWords: polyWords, on: stream)
        let keys = try GpuResidentCache.shared.resident(evaluationKey) // resident after the first chunk
        if parameter.dimensions.count > 1, keys.relinearizationKey == nil {
            throw HeError.missingRelinearizationKey
        }
        let response = try DeviceBuffer(count: 2 * degree) // [2][1][N] after modSwitchDownToSingle
        let dimensions = parameter.dimensions.map { UInt32($0) }
        let handle = try context.gpu
        let remainingCount = expandedRemainingQuery.count
        try dimensions.withUnsafeBufferPointer { dims in
            try present.withUnsafeBufferPointer { mask in
                try heAmdCheck(he_pir_compute_response_chunk_device(
                    handle, dims.baseAddress, UInt32(dims.count), dim0.pointer, rest.pointer, remainingCount,
                    database.pointer, mask.baseAddress, keys.relinearizationKey?.buffer.pointer, response.pointer,
                    stream.raw))
            }
        }
        try await stream.completion()
        let single = singleModulusContext(of: polyContext)
        return try withExtendedLifetime((keys, dim0, rest, database)) {
            try response.downloadCiphertext(context: context, polyContext: single, polyCount: 2, at: 0, on: stream)
        }
    }
}
