// The reference's OWN generic test suites, instantiated with the drop-in types -- the proof that `GpuBfv` is a `HeScheme`
// everything above the scheme accepts, and that `GpuPirUtil` is a `PirUtilProtocol` the index-PIR servers accept.
//
// The reference ships its scheme tests as a library generic over `Scheme: HeScheme` (product `_TestUtilities`,
// Package.swift:74,157; Sources/_TestUtilities/HeApiTestUtils.swift:195-1285) and drives it once per scheme in
// Tests/HomomorphicEncryptionTests/HeAPITests.swift:144-221 (`runBfvTests`); its index-PIR tests are generic over the
// server and client types (Sources/_TestUtilities/PirUtilities/IndexPirTests.swift:23-147).  This file makes the same
// calls with `GpuBfv.self` / `MulPirServer<GpuPirUtil<...>>.self`.  Run on a box with an MI355X, libhe_amd.so and a
// Swift 6.2 toolchain (INTEGRATION.md section 9):
//
//     SWIFT_HE_PATH=/path/to/swift-homomorphic-encryption HE_AMD_LIB_DIR=/opt/he_amd/lib swift test --filter ReferenceSuites
import HeAmd
import HomomorphicEncryption
import PrivateInformationRetrieval
import Testing
@testable import _TestUtilities // indexPirTest(server:client:) is `@inlinable static` (internal) there; `swift test` builds with testability

@Suite
struct ReferenceSuites {
    /// The contexts `runBfvTests(UInt64.self)` runs over (HeAPITests.swift:145-175): every predefined parameter set of
    /// degree <= 512 that fits UInt64, and the custom set of the test utilities.  Its third entry -- 32 coefficient moduli
    /// of 60 bits -- is beyond what the device tables are specialised for (1-16 ciphertext moduli: 16 x 55 bits is the
    /// N = 32768 security cap, EncryptionParameters.swift:204-205); `deviceRefusesMoreThanSixteenCiphertextModuli` below
    /// holds the device to refusing it loudly.
    static func contexts() throws -> [Context<GpuBfv>] {
        let predefined: [EncryptionParameters<UInt64>] = try PredefinedRlweParameters.allCases
            .filter { rlweParams in rlweParams.supportsScalar(UInt64.self) }
            .filter { rlweParams in rlweParams.polyDegree <= 512 }
            .map { rlweParams in try EncryptionParameters<UInt64>(from: rlweParams) }
        let custom = try EncryptionParameters<UInt64>(
            polyDegree: TestUtils.testPolyDegree,
            plaintextModulus: UInt64.generatePrimes(
                significantBitCounts: [12],
                preferringSmall: true,
                nttDegree: TestUtils.testPolyDegree)[0],
            coefficientModuli: TestUtils.testCoefficientModuli(),
            errorStdDev: ErrorStdDev.stdDev32,
            securityLevel: SecurityLevel.unchecked)
        return try (predefined + [custom]).map { try Context<GpuBfv>(encryptionParameters: $0) }
    }

    /// HeAPITests.swift:176-221 with `scheme: GpuBfv.self`, call for call.
    @Test
    func heApiSuiteOverGpuBfv() async throws {
        for context in try Self.contexts() {
            // Sync tests
            try HeAPITestHelpers.schemeEncodeDecodeTest(context: context, scheme: GpuBfv.self)
            try HeAPITestHelpers.schemeEncryptDecryptTest(context: context, scheme: GpuBfv.self)
            try HeAPITestHelpers.schemeEncryptZeroDecryptTest(context: context, scheme: GpuBfv.self)
            try HeAPITestHelpers.schemeEvaluationKeyTest(context: context)
            try HeAPITestHelpers.noiseBudgetTest(context: context, scheme: GpuBfv.self)

            // Async tests
            try await HeAPITestHelpers.schemeEncryptZeroAddDecryptTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeEncryptZeroMultiplyDecryptTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextAdditionTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextSubtractionTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextPlaintextAdditionTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextPlaintextSubtractionTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextPlaintextMultiplicationTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextMultiplySubtractPlainTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextPlaintextMultiplyAddPlainTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextPlaintextMultiplySubtractPlainTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextMultiplyAddPlainTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextCiphertextMultiplicationTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextPlaintextInnerProductTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeThreePolyCiphertextPlaintextInnerProductTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextCiphertextInnerProductTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextMultiplyAddTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextMultiplySubTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeCiphertextNegateTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeApplyGaloisTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeRotationTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeSwapRowsTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.repeatedAdditionTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeSumTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.multiplyPowerOfXTest(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeTestNtt(context: context, scheme: GpuBfv.self)
            try await HeAPITestHelpers.schemeTestFormats(context: context, scheme: GpuBfv.self)
        }
    }

    /// The third context of `runBfvTests` (HeAPITests.swift:163-174): the device has no tables for 32 ciphertext moduli and
    /// says so -- `unsupportedHeOperation` from the first accelerated member, never different words.
    @Test
    func deviceRefusesMoreThanSixteenCiphertextModuli() throws {
        let manyModuli = try EncryptionParameters<UInt64>(
            polyDegree: TestUtils.testPolyDegree,
            plaintextModulus: UInt64.generatePrimes(
                significantBitCounts: [12],
                preferringSmall: true,
                nttDegree: TestUtils.testPolyDegree)[0],
            coefficientModuli: UInt64.generatePrimes(
                significantBitCounts: Array(repeating: UInt64.bitWidth - 4, count: 32),
                preferringSmall: false,
                nttDegree: TestUtils.testPolyDegree),
            errorStdDev: ErrorStdDev.stdDev32,
            securityLevel: SecurityLevel.unchecked)
        let context = try Context<GpuBfv>(encryptionParameters: manyModuli)
        #expect(throws: HeError.self) { _ = try context.gpu }
    }

    /// IndexPirTests.swift:65-137 (`indexPirTest(server:client:)`: fourteen configurations, one and two dimensions, every
    /// key compression, ten random query batches each) with the device-resident server, over both scheme types.
    @Test
    func indexPirSuiteOverGpuPirUtil() async throws {
        try await PirTestUtils.IndexPirTests.indexPirTest(
            server: MulPirServer<GpuPirUtil<Bfv<UInt64>>>.self,
            client: MulPirClient<GpuPirUtil<Bfv<UInt64>>>.self)
        try await PirTestUtils.IndexPirTests.indexPirTest(
            server: MulPirServer<GpuPirUtil<GpuBfv>>.self,
            client: MulPirClient<GpuPirUtil<GpuBfv>>.self)
    }

    /// IndexPirTests.swift:140-147 (`indexPir(scheme:)`: the reference's own `PirUtil`, every scheme operation of the PIR
    /// flow through `GpuBfv`) -- the public entry point, as Tests/PrivateInformationRetrievalTests/IndexPirTests.swift:146-148
    /// calls it for `Bfv<UInt64>`.
    @Test
    func indexPirSuiteOverGpuBfv() async throws {
        try await PirTestUtils.IndexPirTests.indexPir(scheme: GpuBfv.self)
    }
}
