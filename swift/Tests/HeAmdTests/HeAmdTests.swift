// Parity of the GPU paths with the reference's CPU paths on identical inputs (run on a box with an MI355X and
// libhe_amd.so installed): the same comparisons tests/test_gpu_*.py make through ctypes against the C oracle.
import HeAmd
import HomomorphicEncryption
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
}
