// GpuBfv -- the HeScheme conformer (level B3 of include/he_amd.h, SURVEY.md 8b): BFV over UInt64 whose polynomial
// arithmetic runs on the MI355X.  Everything above the scheme in the reference is generic over `Scheme: HeScheme`
// (Ciphertext / Plaintext operators, PirUtilProtocol's defaults, MulPirServer, PNNS), so the drop-in is the type
// parameter: `Context<GpuBfv>`, `MulPirServer<PirUtil<GpuBfv>>`, ... instead of `Bfv<UInt64>`.
//
// What a conformer must supply itself is what NoOpScheme supplies (Sources/HomomorphicEncryption/NoOpScheme.swift:35-368);
// everything else has a default in HeScheme.swift / HeSchemeAsync.swift.  Members fall in two groups:
//   * forwarded: key generation, encoding, encryption, decryption, noise budget, transparency -- client-side, randomised
//     or metadata-only (SURVEY.md 2 OUT OF SCOPE).  They re-wrap their arguments (GpuBfvTwin.swift: array references,
//     not words) and call `Bfv<UInt64>`, so their results are the reference's by construction;
//   * accelerated: the PolyRq / NTT hot path -- ct x ct, relinearize, applyGalois, modSwitchDown(ToSingle), ct x pt,
//     the six innerProduct overloads, ciphertext +- plaintext, forwardNtt / inverseNtt -- each with its `...Async` twin
//     (HeScheme.swift:516-930 declares both; the defaults of HeSchemeAsync.swift:16-141 would call the blocking form).
//     The sync forms wait on the stream, the async forms suspend on its completion callback (HeAmdStream.completion()).
// Each accelerated member makes the reference's own metadata checks before it touches the device, in the reference's
// order, and returns the reference's words (tests/test_gpu_bfv.py holds the C entry points to the oracle).
//
// A per-ciphertext call pays the PCIe round trip of its operands; the batched forms (the innerProduct overloads here,
// Bfv+Gpu.swift, GpuPirUtil) are where the device earns its keep.
import CHeAmd
import HomomorphicEncryption

public enum GpuBfv: HeScheme {
    public typealias CiphertextAuxiliaryData = EmptyAuxiliary<Self>
    public typealias PlaintextAuxiliaryData = EmptyAuxiliary<Self>

    public typealias Context = HomomorphicEncryption.Context<Self>
    public typealias KeySwitchKey = HomomorphicEncryption._KeySwitchKey<Self>
    public typealias GaloisKey = HomomorphicEncryption._GaloisKey<Self>

    public typealias Scalar = UInt64
    public typealias SignedScalar = Int64
    public typealias CanonicalCiphertextFormat = Coeff

    public static var cryptosystem: HeCryptoSystem {
        .bfv
    }

    public static var freshCiphertextPolyCount: Int {
        CpuBfv.freshCiphertextPolyCount
    }

    public static var minNoiseBudget: Double {
        CpuBfv.minNoiseBudget
    }

    // MARK: forwarded to Bfv<UInt64> -- keys, encoding, encryption, decryption (client side, randomised)

    public static func generateSecretKey(context: Context) throws -> SecretKey<GpuBfv> {
        try SecretKey<GpuBfv>(_poly: CpuBfv.generateSecretKey(context: context.cpu)._poly)
    }

    public static func generateEvaluationKey(
        context: Context,
        config: EvaluationKeyConfig,
        using secretKey: SecretKey<GpuBfv>) throws -> EvaluationKey<GpuBfv>
    {
        let key = try CpuBfv.generateEvaluationKey(context: context.cpu, config: config, using: secretKey.cpu())
        return try EvaluationKey<GpuBfv>(cpu: key, context: context)
    }

    public static func simdDimensions(for encryptionParameter: EncryptionParameters<UInt64>) -> SimdEncodingDimensions? {
        CpuBfv.simdDimensions(for: encryptionParameter)
    }

    public static func encodeSimdDimensions(for parameters: EncryptionParameters<UInt64>) -> SimdEncodingDimensions? {
        CpuBfv.encodeSimdDimensions(for: parameters)
    }

    public static func encode(context: Context, values: some Collection<Scalar>,
                              format: EncodeFormat) throws -> CoeffPlaintext
    {
        try CoeffPlaintext(cpu: CpuBfv.encode(context: context.cpu, values: values, format: format), context: context)
    }

    public static func encode(context: Context, signedValues: some Collection<SignedScalar>,
                              format: EncodeFormat) throws -> CoeffPlaintext
    {
        try CoeffPlaintext(cpu: CpuBfv.encode(context: context.cpu, signedValues: signedValues, format: format),
                           context: context)
    }

    public static func encode(context: Context, values: some Collection<Scalar>, format: EncodeFormat,
                              moduliCount: Int?) throws -> EvalPlaintext
    {
        try EvalPlaintext(
            cpu: CpuBfv.encode(context: context.cpu, values: values, format: format, moduliCount: moduliCount),
            context: context)
    }

    public static func encode(
        context: Context,
        signedValues: some Collection<SignedScalar>,
        format: EncodeFormat,
        moduliCount: Int?) throws -> EvalPlaintext
    {
        try EvalPlaintext(
            cpu: CpuBfv.encode(context: context.cpu, signedValues: signedValues, format: format,
                               moduliCount: moduliCount),
            context: context)
    }

    public static func decodeCoeff(plaintext: CoeffPlaintext, format: EncodeFormat) throws -> [Scalar] {
        try CpuBfv.decodeCoeff(plaintext: plaintext.cpu(), format: format)
    }

    public static func decodeCoeff(plaintext: CoeffPlaintext, format: EncodeFormat) throws -> [SignedScalar] {
        try CpuBfv.decodeCoeff(plaintext: plaintext.cpu(), format: format)
    }

    public static func decodeEval(plaintext: EvalPlaintext, format: EncodeFormat) throws -> [Scalar] {
        try CpuBfv.decodeEval(plaintext: plaintext.cpu(), format: format)
    }

    public static func decodeEval(plaintext: EvalPlaintext, format: EncodeFormat) throws -> [SignedScalar] {
        try CpuBfv.decodeEval(plaintext: plaintext.cpu(), format: format)
    }

    public static func skipLSBsForDecryption(for parameter: EncryptionParameters<UInt64>) -> [Int] {
        CpuBfv.skipLSBsForDecryption(for: parameter)
    }

    public static func encrypt(_ plaintext: CoeffPlaintext,
                               using secretKey: SecretKey<GpuBfv>) throws -> CanonicalCiphertext
    {
        try CanonicalCiphertext(cpu: CpuBfv.encrypt(plaintext.cpu(), using: secretKey.cpu()),
                                context: plaintext.context)
    }

    public static func zeroCiphertextCoeff(context: Context, moduliCount: Int?) throws -> CoeffCiphertext {
        try CoeffCiphertext(cpu: CpuBfv.zeroCiphertextCoeff(context: context.cpu, moduliCount: moduliCount),
                            context: context)
    }

    public static func zeroCiphertextEval(context: Context, moduliCount: Int?) throws -> EvalCiphertext {
        try EvalCiphertext(cpu: CpuBfv.zeroCiphertextEval(context: context.cpu, moduliCount: moduliCount),
                           context: context)
    }

    public static func isTransparentCoeff(ciphertext: CoeffCiphertext) -> Bool {
        // Bfv+Encrypt.swift:48-56: every polynomial after the first is zero -- a property of the words alone
        (try? CpuBfv.isTransparentCoeff(ciphertext: ciphertext.cpu())) ?? false
    }

    public static func isTransparentEval(ciphertext: EvalCiphertext) -> Bool {
        (try? CpuBfv.isTransparentEval(ciphertext: ciphertext.cpu())) ?? false
    }

    public static func decryptCoeff(_ ciphertext: CoeffCiphertext,
                                    using secretKey: SecretKey<GpuBfv>) throws -> CoeffPlaintext
    {
        try CoeffPlaintext(cpu: CpuBfv.decryptCoeff(ciphertext.cpu(), using: secretKey.cpu()),
                           context: ciphertext.context)
    }

    public static func decryptEval(_ ciphertext: EvalCiphertext,
                                   using secretKey: SecretKey<GpuBfv>) throws -> CoeffPlaintext
    {
        try CoeffPlaintext(cpu: CpuBfv.decryptEval(ciphertext.cpu(), using: secretKey.cpu()),
                           context: ciphertext.context)
    }

    public static func skipLSBsForDecryption(for ciphertext: CoeffCiphertext) -> [Int] {
        (try? CpuBfv.skipLSBsForDecryption(for: ciphertext.cpu())) ?? skipLSBsForDecryption(
            for: ciphertext.context.encryptionParameters)
    }

    /// - Warning: The noise budget value **must not** be forwarded to any other party (HeScheme.swift:1030-1045).
    public static func noiseBudgetCoeff(of ciphertext: CoeffCiphertext, using secretKey: SecretKey<GpuBfv>,
                                        variableTime: Bool) throws -> Double
    {
        try CpuBfv.noiseBudgetCoeff(of: ciphertext.cpu(), using: secretKey.cpu(), variableTime: variableTime)
    }

    /// - Warning: The noise budget value **must not** be forwarded to any other party.
    public static func noiseBudgetEval(of ciphertext: EvalCiphertext, using secretKey: SecretKey<GpuBfv>,
                                       variableTime: Bool) throws -> Double
    {
        try CpuBfv.noiseBudgetEval(of: ciphertext.cpu(), using: secretKey.cpu(), variableTime: variableTime)
    }

    public static func rotateColumns(
        of ciphertext: inout CanonicalCiphertext,
        by step: Int,
        using evaluationKey: EvaluationKey<GpuBfv>) throws
    {
        // HeScheme.swift:1463-1490: the element of the rotation, then applyGalois -- which is the accelerated member
        let element = try GaloisElement.rotatingColumns(by: step, degree: ciphertext.context.degree)
        try applyGalois(ciphertext: &ciphertext, element: element, using: evaluationKey)
    }

    public static func rotateColumnsAsync(
        of ciphertext: inout CanonicalCiphertext,
        by step: Int,
        using evaluationKey: EvaluationKey<GpuBfv>) async throws
    {
        let element = try GaloisElement.rotatingColumns(by: step, degree: ciphertext.context.degree)
        try await applyGaloisAsync(ciphertext: &ciphertext, element: element, using: evaluationKey)
    }

    public static func swapRows(of ciphertext: inout CanonicalCiphertext,
                                using evaluationKey: EvaluationKey<GpuBfv>) throws
    {
        let element = GaloisElement.swappingRows(degree: ciphertext.context.degree)
        try applyGalois(ciphertext: &ciphertext, element: element, using: evaluationKey)
    }

    public static func swapRowsAsync(of ciphertext: inout CanonicalCiphertext,
                                     using evaluationKey: EvaluationKey<GpuBfv>) async throws
    {
        let element = GaloisElement.swappingRows(degree: ciphertext.context.degree)
        try await applyGaloisAsync(ciphertext: &ciphertext, element: element, using: evaluationKey)
    }

    // MARK: element-wise plaintext / ciphertext arithmetic -- on the host
    //
    // One polynomial in, one out: 2 x 256 KiB over a 63 GB/s link to save 10 us of host adds is a loss; these keep the
    // reference's PolyRq operators (PolyRq.swift:147-245).  Their device forms exist for RESIDENT batches
    // (PolyContext+Gpu.swift gpuAdd / gpuSubtract / gpuNegate).

    public static func addAssign(_ lhs: inout CoeffPlaintext, _ rhs: CoeffPlaintext) throws {
        try validateEquality(of: lhs.context, and: rhs.context)
        lhs = try CoeffPlaintext(_context: lhs.context, _poly: lhs._poly + rhs._poly)
    }

    public static func addAssign(_ lhs: inout EvalPlaintext, _ rhs: EvalPlaintext) throws {
        try validateEquality(of: lhs.context, and: rhs.context)
        lhs = try EvalPlaintext(_context: lhs.context, _poly: lhs._poly + rhs._poly)
    }

    public static func addAssignCoeff(_ lhs: inout CoeffCiphertext, _ rhs: CoeffCiphertext) throws {
        try addAssignSameType(&lhs, rhs)
    }

    public static func addAssignEval(_ lhs: inout EvalCiphertext, _ rhs: EvalCiphertext) throws {
        try addAssignSameType(&lhs, rhs)
    }

    public static func subAssignCoeff(_ lhs: inout CoeffCiphertext, _ rhs: CoeffCiphertext) throws {
        try subAssignSameType(&lhs, rhs)
    }

    public static func subAssignEval(_ lhs: inout EvalCiphertext, _ rhs: EvalCiphertext) throws {
        try subAssignSameType(&lhs, rhs)
    }

    static func addAssignSameType<F: PolyFormat>(_ lhs: inout Ciphertext<GpuBfv, F>,
                                                 _ rhs: Ciphertext<GpuBfv, F>) throws
    {
        // Bfv.swift:70-79
        try validateEquality(of: lhs.context, and: rhs.context)
        for (polyIndex, rhsPoly) in zip(lhs.polys.indices, rhs.polys) {
            lhs.polys[polyIndex] += rhsPoly
        }
        lhs.seed = []
    }

    static func subAssignSameType<F: PolyFormat>(_ lhs: inout Ciphertext<GpuBfv, F>,
                                                 _ rhs: Ciphertext<GpuBfv, F>) throws
    {
        // Bfv.swift:98-107
        try validateEquality(of: lhs.context, and: rhs.context)
        for (polyIndex, rhsPoly) in zip(lhs.polys.indices, rhs.polys) {
            lhs.polys[polyIndex] -= rhsPoly
        }
        lhs.seed = []
    }

    public static func negAssignCoeff(_ ciphertext: inout CoeffCiphertext) {
        for polyIndex in ciphertext.polys.indices { // Bfv.swift:132-138
            ciphertext.polys[polyIndex] = -ciphertext.polys[polyIndex]
        }
        ciphertext.seed = []
    }

    public static func negAssignEval(_ ciphertext: inout EvalCiphertext) {
        for polyIndex in ciphertext.polys.indices { // Bfv.swift:140-146
            ciphertext.polys[polyIndex] = -ciphertext.polys[polyIndex]
        }
        ciphertext.seed = []
    }

    public static func addAssignEval(_: inout EvalCiphertext, _: EvalPlaintext) throws {
        // Bfv.swift:153-156: BFV keeps NTT conversions explicit, no Eval-format plaintext addition
        throw HeError.unsupportedHeOperation(description: "addAssignEval(ciphertext, plaintext)")
    }

    public static func subAssignEval(_: inout EvalCiphertext, _: EvalPlaintext) throws {
        throw HeError.unsupportedHeOperation(description: "subAssignEval(ciphertext, plaintext)") // Bfv.swift:158-161
    }

    // MARK: accelerated -- ciphertext +- plaintext (plaintextTranslate, Bfv+Encrypt.swift:75-140)

    public static func addAssignCoeff(_ ciphertext: inout CoeffCiphertext, _ plaintext: CoeffPlaintext) throws {
        let work = try translate(ciphertext, plaintext, subtract: false)
        try work.stream.synchronize()
        ciphertext = try work.finish()
    }

    public static func addAssignCoeffAsync(_ ciphertext: inout CoeffCiphertext, _ plaintext: CoeffPlaintext) async throws {
        let work = try translate(ciphertext, plaintext, subtract: false)
        try await work.stream.completion()
        ciphertext = try work.finish()
    }

    public static func subAssignCoeff(_ ciphertext: inout CoeffCiphertext, _ plaintext: CoeffPlaintext) throws {
        let work = try translate(ciphertext, plaintext, subtract: true)
        try work.stream.synchronize()
        ciphertext = try work.finish()
    }

    public static func subAssignCoeffAsync(_ ciphertext: inout CoeffCiphertext, _ plaintext: CoeffPlaintext) async throws {
        let work = try translate(ciphertext, plaintext, subtract: true)
        try await work.stream.completion()
        ciphertext = try work.finish()
    }

    // MARK: accelerated -- ciphertext * plaintext (Bfv.swift:120-129)

    public static func mulAssign(_ ciphertext: inout EvalCiphertext, _ plaintext: EvalPlaintext) throws {
        let work = try multiplyPlain(ciphertext, plaintext)
        try work.stream.synchronize()
        ciphertext = try work.finish()
    }

    public static func mulAssignAsync(_ ciphertext: inout EvalCiphertext, _ plaintext: EvalPlaintext) async throws {
        let work = try multiplyPlain(ciphertext, plaintext)
        try await work.stream.completion()
        ciphertext = try work.finish()
    }

    // MARK: accelerated -- ciphertext * ciphertext (Bfv+Multiply.swift:18-85)

    public static func mulAssign(_ lhs: inout CanonicalCiphertext, _ rhs: CanonicalCiphertext) throws {
        let work = try multiply(lhs, rhs)
        try work.stream.synchronize()
        lhs = try work.finish()
    }

    public static func mulAssignAsync(_ lhs: inout CanonicalCiphertext, _ rhs: CanonicalCiphertext) async throws {
        let work = try multiply(lhs, rhs)
        try await work.stream.completion()
        lhs = try work.finish()
    }

    // MARK: accelerated -- relinearize, applyGalois (Bfv.swift:174-219, Bfv+Keys.swift:123-208)

    public static func relinearize(_ ciphertext: inout CanonicalCiphertext, using key: EvaluationKey<GpuBfv>) throws {
        let work = try relinearization(ciphertext, key)
        try work.stream.synchronize()
        ciphertext = try work.finish()
    }

    public static func relinearizeAsync(_ ciphertext: inout CanonicalCiphertext,
                                        using key: EvaluationKey<GpuBfv>) async throws
    {
        let work = try relinearization(ciphertext, key)
        try await work.stream.completion()
        ciphertext = try work.finish()
    }

    public static func applyGalois(
        ciphertext: inout CanonicalCiphertext,
        element: Int,
        using evaluationKey: EvaluationKey<GpuBfv>) throws
    {
        let work = try galois(ciphertext, element, evaluationKey)
        try work.stream.synchronize()
        ciphertext = try work.finish()
    }

    public static func applyGaloisAsync(
        ciphertext: inout CanonicalCiphertext,
        element: Int,
        using evaluationKey: EvaluationKey<GpuBfv>) async throws
    {
        let work = try galois(ciphertext, element, evaluationKey)
        try await work.stream.completion()
        ciphertext = try work.finish()
    }

    // MARK: accelerated -- modulus switching (Bfv.swift:163-171; HeScheme.swift modSwitchDownToSingle)

    public static func modSwitchDown(_ ciphertext: inout CanonicalCiphertext) throws {
        let work = try modulusSwitch(ciphertext, toSingle: false)
        try work.stream.synchronize()
        ciphertext = try work.finish()
    }

    public static func modSwitchDownAsync(_ ciphertext: inout CanonicalCiphertext) async throws {
        let work = try modulusSwitch(ciphertext, toSingle: false)
        try await work.stream.completion()
        ciphertext = try work.finish()
    }

    public static func modSwitchDownToSingle(_ ciphertext: inout CanonicalCiphertext) throws {
        guard ciphertext.polys[0].context.moduli.count > 1 else { return }
        let work = try modulusSwitch(ciphertext, toSingle: true)
        try work.stream.synchronize()
        ciphertext = try work.finish()
    }

    public static func modSwitchDownToSingleAsync(_ ciphertext: inout CanonicalCiphertext) async throws {
        guard ciphertext.polys[0].context.moduli.count > 1 else { return }
        let work = try modulusSwitch(ciphertext, toSingle: true)
        try await work.stream.completion()
        ciphertext = try work.finish()
    }

    // MARK: accelerated -- ciphertext NTT (Bfv.swift:654-669)

    public static func forwardNtt(_ ciphertext: inout CoeffCiphertext) throws -> EvalCiphertext {
        let work: DeviceWork<Ciphertext<GpuBfv, Eval>> = try transform(ciphertext, inverse: false)
        try work.stream.synchronize()
        return try work.finish()
    }

    public static func forwardNttAsync(_ ciphertext: inout CoeffCiphertext) async throws -> EvalCiphertext {
        let work: DeviceWork<Ciphertext<GpuBfv, Eval>> = try transform(ciphertext, inverse: false)
        try await work.stream.completion()
        return try work.finish()
    }

    public static func inverseNtt(_ ciphertext: inout EvalCiphertext) throws -> CoeffCiphertext {
        let work: DeviceWork<Ciphertext<GpuBfv, Coeff>> = try transform(ciphertext, inverse: true)
        try work.stream.synchronize()
        return try work.finish()
    }

    public static func inverseNttAsync(_ ciphertext: inout EvalCiphertext) async throws -> CoeffCiphertext {
        let work: DeviceWork<Ciphertext<GpuBfv, Coeff>> = try transform(ciphertext, inverse: true)
        try await work.stream.completion()
        return try work.finish()
    }

    // MARK: accelerated -- inner products (Bfv.swift:224-651; HeScheme.swift:740-830): the six overloads

    public static func innerProduct(
        _ lhs: some Collection<CanonicalCiphertext>,
        _ rhs: some Collection<CanonicalCiphertext>) throws -> CanonicalCiphertext
    {
        let work = try ciphertextInnerProduct(Array(lhs), Array(rhs))
        try work.stream.synchronize()
        return try work.finish()
    }

    public static func innerProductAsync(
        _ lhs: some Collection<CanonicalCiphertext>,
        _ rhs: some Collection<CanonicalCiphertext>) async throws -> CanonicalCiphertext
    {
        let work = try ciphertextInnerProduct(Array(lhs), Array(rhs))
        try await work.stream.completion()
        return try work.finish()
    }

    /// (`maxConcurrentTasks` bounds the reference's task group; one device launch has nothing to divide)
    public static func innerProduct(
        _ lhs: some Collection<CanonicalCiphertext>,
        _ rhs: some Collection<CanonicalCiphertext>,
        maxConcurrentTasks _: Int) async throws -> CanonicalCiphertext
    {
        try await innerProductAsync(lhs, rhs)
    }

    public static func innerProduct(ciphertexts: some Collection<EvalCiphertext>,
                                    plaintexts: some Collection<EvalPlaintext>) throws -> EvalCiphertext
    {
        let work = try plaintextInnerProduct(Array(ciphertexts), plaintexts.map { Optional($0) })
        try work.stream.synchronize()
        return try work.finish()
    }

    public static func innerProductAsync(ciphertexts: some Collection<EvalCiphertext>,
                                         plaintexts: some Collection<EvalPlaintext>) async throws -> EvalCiphertext
    {
        let work = try plaintextInnerProduct(Array(ciphertexts), plaintexts.map { Optional($0) })
        try await work.stream.completion()
        return try work.finish()
    }

    public static func innerProduct(ciphertexts: some Collection<EvalCiphertext>,
                                    plaintexts: some Collection<EvalPlaintext?>) throws -> EvalCiphertext
    {
        let work = try plaintextInnerProduct(Array(ciphertexts), Array(plaintexts))
        try work.stream.synchronize()
        return try work.finish()
    }

    public static func innerProductAsync(ciphertexts: some Collection<EvalCiphertext>,
                                         plaintexts: some Collection<EvalPlaintext?>) async throws -> EvalCiphertext
    {
        let work = try plaintextInnerProduct(Array(ciphertexts), Array(plaintexts))
        try await work.stream.completion()
        return try work.finish()
    }

    public static func innerProduct(ciphertexts: some Collection<EvalCiphertext>,
                                    plaintexts: some Collection<EvalPlaintext>,
                                    maxConcurrentTasks _: Int) async throws -> EvalCiphertext
    {
        try await innerProductAsync(ciphertexts: ciphertexts, plaintexts: plaintexts)
    }

    public static func innerProduct(ciphertexts: some Collection<EvalCiphertext>,
                                    plaintexts: some Collection<EvalPlaintext?>,
                                    maxConcurrentTasks _: Int) async throws -> EvalCiphertext
    {
        try await innerProductAsync(ciphertexts: ciphertexts, plaintexts: plaintexts)
    }
}
