// Level B4: a PirUtilProtocol conformer (Sources/PrivateInformationRetrieval/IndexPir/PirUtil.swift:22-160) whose
// per-chunk response runs on the device.  Servers are generic over the conformer (IndexPir/MulPir.swift:292), so
// `MulPirServer<GpuPirUtil>` is the drop-in; every other requirement keeps the protocol's default implementation.
import CHeAmd
import HomomorphicEncryption
import PrivateInformationRetrieval

public enum GpuPirUtil: PirUtilProtocol {
    public typealias Scheme = Bfv<UInt64>
    public typealias Scalar = UInt64

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
        for (index, ciphertext) in expandedDim0Query.enumerated() {
            try dim0.upload(ciphertext, at: index * 2 * polyWords, on: stream)
        }
        let rest = try DeviceBuffer(count: expandedRemainingQuery.count * 2 * polyWords)
        for (index, ciphertext) in expandedRemainingQuery.enumerated() {
            try rest.upload(ciphertext, at: index * 2 * polyWords, on: stream)
        }
        // the chunk: plaintext k of column c at index c * d0 + k (MulPir.swift:547-555); nil plaintexts are masked out.
        // A server that answers many queries keeps this slab resident and uploads it once (he_pir_compute_response_device
        // takes all chunks of a database at once, he_pir_compute_response_queries_device up to four queries that then
        // share one pass over it, he_pir_compute_response_to_query_device the whole computeResponse(to:...) of a
        // Query); it is uploaded per call here to keep the protocol's signature.
        let database = try DeviceBuffer(count: perChunk * polyWords)
        var present = [UInt8](repeating: 0, count: perChunk)
        for (index, plaintext) in dataChunk.enumerated() where index < perChunk {
            guard let plaintext else { continue }
            present[index] = 1
            try database.upload(plaintext._poly, at: index * polyWords, on: stream) // Plaintext.swift:28
        }
        var key: DeviceKeySwitchKey?
        if parameter.dimensions.count > 1 {
            guard let relinearizationKey = evaluationKey._relinearizationKey else {
                throw HeError.missingRelinearizationKey
            }
            key = try DeviceKeySwitchKey(relinearizationKey._keySwitchKey, on: stream)
        }
        let response = try DeviceBuffer(count: 2 * degree) // [2][1][N] after modSwitchDownToSingle
        let dimensions = parameter.dimensions.map { UInt32($0) }
        try dimensions.withUnsafeBufferPointer { dims in
            try present.withUnsafeBufferPointer { mask in
                try heAmdCheck(he_pir_compute_response_chunk_device(
                    context.gpu, dims.baseAddress, UInt32(dims.count), dim0.pointer, rest.pointer,
                    expandedRemainingQuery.count, database.pointer, mask.baseAddress, key?.buffer.pointer,
                    response.pointer, stream.raw))
            }
        }
        try await stream.completion()
        // the single-modulus context the reference's modSwitchDownToSingle ends on (Ciphertext.swift, Bfv.swift:163-171)
        var single = polyContext
        while single.moduli.count > 1, let next = single.next { single = next }
        return try response.downloadCiphertext(context: context, polyContext: single, polyCount: 2, at: 0, on: stream)
    }
}
