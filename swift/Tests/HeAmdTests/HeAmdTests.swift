// Parity of the GPU paths with the reference's CPU paths on identical inputs (run on a box with an MI355X and
// libhe_amd.so installed): the same comparisons tests/test_gpu_*.py make through ctypes against the C oracle, through
// the Swift surface.  `GpuBfv` is held to `Bfv<UInt64>` word for word: a ciphertext re-tagged with `accelerated()` goes
// through GpuBfv's member, its twin through Bfv's, and the polynomials must be equal.
import HeAmd
import HomomorphicEncryption
import PrivateInformationRetrieval
import Testing

@Suite
struct HeAmdTests {
    @Test
    func forwardAndInverseNttMatchTheReference() throws {
        let degree = 8192
        let moduli = try UInt64.generatePrimes(significantBitCounts: [55, 55, 55, 55], preferringSmall: false,
                                               nttDegree: degree)
        let context = try PolyContext<UInt64>(degree: degree, moduli: moduli)
        let poly = PolyRq<UInt64, Coeff>.random(context: context)
        let reference = try poly.forwardNtt() // PolyRq+Ntt.swift:209-222
        let gpu = try context.gpuForwardNtt(poly)
        #expect(gpu == reference)
        #expect(try context.gpuInverseNtt(gpu) == poly)
    }

    @Test
    func multiplyRelinearizeMatchesTheReference() async throws {
        let parameters = try EncryptionParameters<UInt64>(from: .n_8192_logq_3x55_logt_42)
        let context = try Context<Bfv<UInt64>>(encryptionParameters: parameters)
        let secretKey = try context.generateSecretKey()
        let evaluationKey = try context.generateEvaluationKey(
            config: EvaluationKeyConfig(hasRelinearizationKey: true), using: secretKey)
        let values: [UInt64] = (0..<UInt64(context.degree)).map { $0 % context.plaintextModulus }
        let plaintext: Plaintext<Bfv<UInt64>, Coeff> = try context.encode(values: values, format: .coefficient)
        var gpu = try [plaintext.encrypt(using: secretKey), plaintext.encrypt(using: secretKey)]
        var cpu = gpu
        for index in cpu.indices {
            try cpu[index] *= cpu[index]
            try cpu[index].relinearize(using: evaluationKey)
        }
        try await Bfv<UInt64>.gpuMultiplyRelinearize(&gpu, gpu, using: evaluationKey)
        #expect(gpu == cpu)
    }

    @Test
    func gpuBfvAgreesWithBfv() async throws {
        let parameters = try EncryptionParameters<UInt64>(from: .n_8192_logq_3x55_logt_42)
        let cpuContext = try Context<Bfv<UInt64>>(encryptionParameters: parameters)
        let gpuContext = try cpuContext.accelerated
        let secretKey = try cpuContext.generateSecretKey()
        let element = try GaloisElement.rotatingColumns(by: 1, degree: cpuContext.degree)
        let cpuKey = try cpuContext.generateEvaluationKey(
            config: EvaluationKeyConfig(galoisElements: [element], hasRelinearizationKey: true), using: secretKey)
        let gpuKey = try EvaluationKey<GpuBfv>(cpu: cpuKey, context: gpuContext)
        let values: [UInt64] = (0..<UInt64(cpuContext.degree)).map { ($0 * 7 + 3) % cpuContext.plaintextModulus }
        let plaintext: Plaintext<Bfv<UInt64>, Coeff> = try cpuContext.encode(values: values, format: .simd)
        var cpu = try plaintext.encrypt(using: secretKey)
        var gpu = try cpu.accelerated()
        let other = try plaintext.encrypt(using: secretKey)

        try cpu *= other // Bfv.mulAssign, Bfv+Multiply.swift:18-21
        try await GpuBfv.mulAssignAsync(&gpu, other.accelerated())
        #expect(gpu.polys == cpu.polys)
        try cpu.relinearize(using: cpuKey)
        try GpuBfv.relinearize(&gpu, using: gpuKey)
        #expect(gpu.polys == cpu.polys)
        try cpu.applyGalois(element: element, using: cpuKey)
        try await GpuBfv.applyGaloisAsync(ciphertext: &gpu, element: element, using: gpuKey)
        #expect(gpu.polys == cpu.polys)
        try cpu.modSwitchDown()
        try GpuBfv.modSwitchDown(&gpu)
        #expect(gpu.polys == cpu.polys)
        let evalCpu = try cpu.forwardNtt(), evalGpu = try GpuBfv.forwardNtt(&gpu)
        #expect(evalGpu.polys == evalCpu.polys)
        // and it still decrypts through the forwarded members
        let decrypted = try GpuBfv.decryptCoeff(gpu, using: SecretKey<GpuBfv>(_poly: secretKey._poly))
        #expect(try decrypted.cpu() == cpu.decrypt(using: secretKey))
    }

    @Test
    func pirServerOnTheDeviceAnswersLikeTheReference() async throws {
        let parameters = try EncryptionParameters<UInt64>(from: .n_4096_logq_27_28_28_logt_5)
        let context = try Context<Bfv<UInt64>>(encryptionParameters: parameters)
        let database: [[UInt8]] = (0..<1000).map { index in (0..<16).map { UInt8(truncatingIfNeeded: index &* 31 &+ $0) } }
        let config = try IndexPirConfig(entryCount: database.count, entrySizeInBytes: 16, dimensionCount: 2, batchSize: 2,
                                        unevenDimensions: false, keyCompression: .noCompression)
        let parameter = MulPirServer<PirUtil<Bfv<UInt64>>>.generateParameter(config: config, with: context)
        let processed = try MulPirServer<PirUtil<Bfv<UInt64>>>.process(database: database, with: context,
                                                                      using: parameter)
        let cpuServer = try MulPirServer<PirUtil<Bfv<UInt64>>>(parameter: parameter, context: context,
                                                              database: processed)
        let gpuServer = try MulPirServer<GpuPirUtil<Bfv<UInt64>>>(parameter: parameter, context: context,
                                                                 database: processed)
        let client = MulPirClient<PirUtil<Bfv<UInt64>>>(parameter: parameter, context: context)
        let secretKey = try context.generateSecretKey()
        let evaluationKey = try client.generateEvaluationKey(using: secretKey)
        let query = try client.generateQuery(at: [17, 923], using: secretKey)
        let expected = try await cpuServer.computeResponse(to: query, using: evaluationKey)
        let response = try await gpuServer.computeResponse(to: query, using: evaluationKey)
        #expect(response.ciphertexts == expected.ciphertexts)
        #expect(try client.decrypt(response: response, at: 17, using: secretKey) == database[17])
    }
}
