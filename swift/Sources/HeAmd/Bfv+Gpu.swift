// Level B3 of include/he_amd.h next to Bfv<UInt64>: the scheme operations on the hot path as BATCHED, device-resident
// pipelines for callers that hold batches (a server loop).  The per-ciphertext HeScheme surface is GpuBfv.swift; these
// are the forms that amortise the PCIe round trip over a batch or skip it altogether (resident buffers).  Each function
// is the batched form of one HeScheme requirement and performs the reference's own metadata checks on the Swift side
// (they never cross the boundary):
//   mulAssign(ct, ct)        Sources/HomomorphicEncryption/Bfv/Bfv+Multiply.swift:18-85
//   relinearize              Bfv/Bfv.swift:201-219, Bfv/Bfv+Keys.swift:123-208
//   modSwitchDown            Bfv/Bfv.swift:163-171
//   innerProduct(cts, pts)   Bfv/Bfv.swift:476-505
//   addAssignCoeff / subAssignCoeff(ct, pt)   Bfv/Bfv.swift:110-117, Bfv/Bfv+Encrypt.swift:75-140
import CHeAmd
import HomomorphicEncryption

extension Bfv where T == UInt64 {
    /// `lhs[i] *= rhs[i]` followed by `relinearize(using:)` for a whole batch: one upload, ten kernel launches, one
    /// download.  The async form suspends on the stream's completion callback instead of blocking.
    public static func gpuMultiplyRelinearize(_ lhs: inout [CanonicalCiphertext], _ rhs: [CanonicalCiphertext],
                                              using evaluationKey: EvaluationKey<Bfv<UInt64>>) async throws
    {
        guard let first = lhs.first else { return }
        guard lhs.count == rhs.count else {
            throw HeError.incompatibleCiphertextCount("lhs \(lhs.count) != rhs \(rhs.count)")
        }
        // the checks of Bfv+Multiply.swift:62-76
        for (left, right) in zip(lhs, rhs) {
            guard left.polys.count == 2, right.polys.count == 2, left.correctionFactor == 1,
                  right.correctionFactor == 1
            else {
                throw HeError.invalidCiphertext("ct x ct wants two polynomials and correction factor 1")
            }
            guard left.context == right.context,
                  left.polys[0].context.moduli.count == right.polys[0].context.moduli.count
            else {
                throw HeError.incompatibleCiphertexts("contexts or levels differ")
            }
        }
        guard let relinearizationKey = evaluationKey._relinearizationKey else {
            throw HeError.missingRelinearizationKey // Bfv.swift:208-210
        }
        let context = first.context
        let polyContext = first.polys[0].context
        let handle = try context.gpu
        let level = UInt32(polyContext.moduli.count)
        let polyWords = polyContext.moduli.count * polyContext.degree
        let batch = lhs.count
        let stream = try HeAmdStream()
        let left = try DeviceBuffer(count: batch * 2 * polyWords), right = try DeviceBuffer(count: batch * 2 * polyWords)
        let product = try DeviceBuffer(count: batch * 3 * polyWords), out = try DeviceBuffer(count: batch * 2 * polyWords)
        try left.upload(contentsOf: lhs, at: 0, on: stream) // each operand batch: one staged copy, no wait
        try right.upload(contentsOf: rhs, at: 0, on: stream)
        let key = try DeviceKeySwitchKey(relinearizationKey._keySwitchKey, on: stream)
        try heAmdCheck(he_bfv_mul_device(handle, level, left.pointer, right.pointer, product.pointer, batch, nil, 0,
                                         stream.raw))
        try heAmdCheck(he_bfv_relinearize_device(handle, level, product.pointer, key.buffer.pointer, out.pointer, batch,
                                                 nil, 0, stream.raw))
        try await stream.completion()
        for index in 0..<batch {
            lhs[index] = try out.downloadCiphertext(context: context, polyContext: polyContext, polyCount: 2,
                                                    at: index * 2 * polyWords, on: stream)
        }
    }

    /// `Bfv.modSwitchDown` (Bfv.swift:163-171) on resident ciphertexts: [batch][polyCount][L][N] -> [..][L-1][N].
    public static func gpuModSwitchDown(context: Context<Bfv<UInt64>>, moduliCount: Int, polyCount: Int,
                                        input: DeviceBuffer, output: DeviceBuffer, batch: Int,
                                        on stream: HeAmdStream) throws
    {
        try heAmdCheck(he_bfv_mod_switch_down_device(context.gpu, UInt32(moduliCount), UInt32(polyCount), input.pointer,
                                                     output.pointer, batch, stream.raw))
    }

    /// `Bfv.addAssignCoeff` / `subAssignCoeff(_: inout CoeffCiphertext, _: CoeffPlaintext)` (Bfv.swift:110-117) on
    /// resident ciphertexts: ciphertexts [batch][polyCount][L][N] Coeff in place, plaintexts [batch][N] (values < t).
    /// The caller checks `correctionFactor == 1` and the contexts (Bfv+Encrypt.swift:80-83) before the call.
    public static func gpuTranslate(context: Context<Bfv<UInt64>>, moduliCount: Int, polyCount: Int,
                                    ciphertexts: DeviceBuffer, plaintexts: DeviceBuffer, subtract: Bool, batch: Int,
                                    on stream: HeAmdStream) throws
    {
        let translate = subtract ? he_bfv_sub_plain_device : he_bfv_add_plain_device
        try heAmdCheck(translate(context.gpu, UInt32(moduliCount), UInt32(polyCount), ciphertexts.pointer,
                                 plaintexts.pointer, batch, stream.raw))
    }

    /// `Bfv.innerProduct(ciphertexts:plaintexts:)` (Bfv.swift:476-505) for `columns` outputs that share the ciphertext
    /// vector; `presentMask` is the nil-plaintext mask resident on the device ([columns][count] bytes) or nil.
    public static func gpuInnerProduct(context: Context<Bfv<UInt64>>, moduliCount: Int, ciphertexts: DeviceBuffer,
                                       plaintexts: DeviceBuffer, presentMask: UnsafePointer<UInt8>?, count: Int,
                                       columns: Int, output: DeviceBuffer, on stream: HeAmdStream) throws
    {
        try heAmdCheck(he_bfv_inner_product_plain_resident_device(context.gpu, UInt32(moduliCount), 2,
                                                                  ciphertexts.pointer, plaintexts.pointer, presentMask,
                                                                  count, columns, output.pointer, stream.raw))
    }
}
