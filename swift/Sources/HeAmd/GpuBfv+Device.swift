// The device side of GpuBfv's accelerated members: each function makes the reference's metadata checks, uploads the
// operands, enqueues ONE C entry point on a fresh stream and returns a `DeviceWork` whose `finish()` reads the result
// back.  The caller either blocks (`stream.synchronize()`, the sync requirement) or suspends (`await stream.completion()`,
// the `...Async` twin) in between -- the enqueue is the same.
import CHeAmd
import HomomorphicEncryption

/// Work enqueued on `stream`.  Holds the device buffers until the result has been read back.
final class DeviceWork<Result>: @unchecked Sendable {
    let stream: HeAmdStream
    private let buffers: [AnyObject]
    private let read: () throws -> Result

    init(stream: HeAmdStream, buffers: [AnyObject], read: @escaping () throws -> Result) {
        self.stream = stream
        self.buffers = buffers
        self.read = read
    }

    /// Call once the stream has reached this work (after `synchronize()` / `completion()`).
    func finish() throws -> Result {
        try withExtendedLifetime(buffers) { try read() }
    }
}

extension GpuBfv {
    /// The ciphertext's level as the C ABI counts it, its polynomial context and words per polynomial.
    static func shape<F: PolyFormat>(_ ciphertext: Ciphertext<GpuBfv, F>)
        -> (level: UInt32, polyContext: PolyContext<UInt64>, polyWords: Int)
    {
        let polyContext = ciphertext.polys[0].context
        return (UInt32(polyContext.moduli.count), polyContext, polyContext.moduli.count * polyContext.degree)
    }

    /// All polynomials of `ciphertext` in a fresh device buffer.
    static func resident<F: PolyFormat>(_ ciphertext: Ciphertext<GpuBfv, F>, on stream: HeAmdStream) throws -> DeviceBuffer {
        let words = ciphertext.polys.reduce(0) { $0 + $1.data.count }
        let buffer = try DeviceBuffer(count: words)
        try buffer.upload(ciphertext, at: 0, on: stream)
        return buffer
    }

    // ---- ciphertext +- plaintext: plaintextTranslate (Bfv+Encrypt.swift:75-140) -> he_bfv_add_plain_device / _sub_
    static func translate(_ ciphertext: CoeffCiphertext, _ plaintext: CoeffPlaintext,
                          subtract: Bool) throws -> DeviceWork<CoeffCiphertext>
    {
        guard ciphertext.correctionFactor == 1 else { // Bfv+Encrypt.swift:80-82
            throw HeError.invalidCorrectionFactor("\(ciphertext.correctionFactor)")
        }
        try validateEquality(of: ciphertext.context, and: plaintext.context)
        let context = ciphertext.context
        let (level, polyContext, _) = shape(ciphertext)
        let polyCount = ciphertext.polys.count
        let stream = try HeAmdStream()
        let slab = try resident(ciphertext, on: stream)
        let message = try DeviceBuffer(count: plaintext._poly.data.count) // [N] values < t
        try message.upload(plaintext._poly, at: 0, on: stream)
        if subtract {
            try heAmdCheck(he_bfv_sub_plain_device(context.gpu, level, UInt32(polyCount), slab.pointer, message.pointer, 1,
                                                   stream.raw))
        } else {
            try heAmdCheck(he_bfv_add_plain_device(context.gpu, level, UInt32(polyCount), slab.pointer, message.pointer, 1,
                                                   stream.raw))
        }
        return DeviceWork(stream: stream, buffers: [slab, message]) {
            try slab.downloadCiphertext(context: context, polyContext: polyContext, polyCount: polyCount, at: 0,
                                        on: stream)
        }
    }

    // ---- ciphertext * plaintext (Bfv.swift:120-129) -> he_bfv_mul_plain_device
    static func multiplyPlain(_ ciphertext: EvalCiphertext, _ plaintext: EvalPlaintext) throws -> DeviceWork<EvalCiphertext> {
        try validateEquality(of: ciphertext.context, and: plaintext.context)
        guard ciphertext.polys[0].context.moduli.count == plaintext._poly.context.moduli.count else {
            throw HeError.incompatibleCiphertextAndPlaintext("ciphertext and plaintext levels differ") // Bfv.swift:122-124
        }
        let context = ciphertext.context
        let (level, polyContext, _) = shape(ciphertext)
        let polyCount = ciphertext.polys.count
        let correctionFactor = ciphertext.correctionFactor
        let stream = try HeAmdStream()
        let slab = try resident(ciphertext, on: stream)
        let factor = try DeviceBuffer(count: plaintext._poly.data.count)
        try factor.upload(plaintext._poly, at: 0, on: stream)
        try heAmdCheck(he_bfv_mul_plain_device(context.gpu, level, UInt32(polyCount), slab.pointer, factor.pointer, 1,
                                               stream.raw))
        return DeviceWork(stream: stream, buffers: [slab, factor]) {
            try slab.downloadCiphertext(context: context, polyContext: polyContext, polyCount: polyCount, at: 0,
                                        correctionFactor: correctionFactor, on: stream)
        }
    }

    // ---- ciphertext * ciphertext (Bfv+Multiply.swift:18-85) -> he_bfv_mul_device
    static func validateProductInput(_ lhs: CanonicalCiphertext, _ rhs: CanonicalCiphertext) throws {
        // multiplyWithoutScaling's checks, in its order (Bfv+Multiply.swift:66-75)
        try validateEquality(of: lhs.context, and: rhs.context)
        guard lhs.polys.count == freshCiphertextPolyCount, lhs.correctionFactor == 1 else {
            throw HeError.invalidCiphertext("lhs: \(lhs.polys.count) polynomials, correction factor \(lhs.correctionFactor)")
        }
        guard rhs.polys.count == freshCiphertextPolyCount, rhs.correctionFactor == 1 else {
            throw HeError.invalidCiphertext("rhs: \(rhs.polys.count) polynomials, correction factor \(rhs.correctionFactor)")
        }
        guard lhs.polys[0].context == rhs.polys[0].context else {
            throw HeError.incompatibleCiphertexts("the ciphertexts are at different levels")
        }
    }

    static func multiply(_ lhs: CanonicalCiphertext, _ rhs: CanonicalCiphertext) throws -> DeviceWork<CanonicalCiphertext> {
        try validateProductInput(lhs, rhs)
        let context = lhs.context
        let (level, polyContext, polyWords) = shape(lhs)
        let stream = try HeAmdStream()
        let left = try resident(lhs, on: stream), right = try resident(rhs, on: stream)
        let product = try DeviceBuffer(count: 3 * polyWords)
        try heAmdCheck(he_bfv_mul_device(context.gpu, level, left.pointer, right.pointer, product.pointer, 1, nil, 0,
                                         stream.raw))
        return DeviceWork(stream: stream, buffers: [left, right, product]) {
            try product.downloadCiphertext(context: context, polyContext: polyContext, polyCount: 3, at: 0, on: stream)
        }
    }

    // ---- relinearize (Bfv.swift:201-219) -> he_bfv_relinearize_device
    static func relinearization(_ ciphertext: CanonicalCiphertext,
                                _ key: EvaluationKey<GpuBfv>) throws -> DeviceWork<CanonicalCiphertext>
    {
        precondition(ciphertext.correctionFactor == 1,
                     "BFV relinearization not implemented for correction factor not equal to 1")
        precondition(ciphertext.polys.count == 3, "ciphertext must have three polys when relinearizing")
        guard let relinearizationKey = key._relinearizationKey else {
            throw HeError.missingRelinearizationKey // Bfv.swift:208-210
        }
        let context = ciphertext.context
        let (level, polyContext, polyWords) = shape(ciphertext)
        let stream = try HeAmdStream()
        let slab = try resident(ciphertext, on: stream)
        let deviceKey = try DeviceKeySwitchKey(relinearizationKey._keySwitchKey, on: stream)
        let out = try DeviceBuffer(count: 2 * polyWords)
        try heAmdCheck(he_bfv_relinearize_device(context.gpu, level, slab.pointer, deviceKey.buffer.pointer, out.pointer, 1,
                                                 nil, 0, stream.raw))
        return DeviceWork(stream: stream, buffers: [slab, deviceKey, out]) {
            try out.downloadCiphertext(context: context, polyContext: polyContext, polyCount: 2, at: 0, on: stream)
        }
    }

    // ---- applyGalois (Bfv.swift:174-198) -> he_bfv_apply_galois_device
    static func galois(_ ciphertext: CanonicalCiphertext, _ element: Int,
                       _ evaluationKey: EvaluationKey<GpuBfv>) throws -> DeviceWork<CanonicalCiphertext>
    {
        precondition(ciphertext.polys.count == 2, "ciphertext must have two polys when applying galois")
        precondition(ciphertext.correctionFactor == 1,
                     "BFV Galois automorphisms not implemented for correction factor not equal to 1")
        guard let galoisKey = evaluationKey._galoisKey else {
            throw HeError.missingGaloisKey
        }
        guard let keySwitchingKey = galoisKey._keys[element] else {
            throw HeError.missingGaloisElement(element: element)
        }
        let context = ciphertext.context
        let (level, polyContext, polyWords) = shape(ciphertext)
        let stream = try HeAmdStream()
        let slab = try resident(ciphertext, on: stream)
        let deviceKey = try DeviceKeySwitchKey(keySwitchingKey, on: stream)
        let out = try DeviceBuffer(count: 2 * polyWords)
        try heAmdCheck(he_bfv_apply_galois_device(context.gpu, level, slab.pointer, UInt64(element),
                                                  deviceKey.buffer.pointer, out.pointer, 1, nil, 0, stream.raw))
        return DeviceWork(stream: stream, buffers: [slab, deviceKey, out]) {
            try out.downloadCiphertext(context: context, polyContext: polyContext, polyCount: 2, at: 0, on: stream)
        }
    }

    // ---- modSwitchDown / modSwitchDownToSingle (Bfv.swift:163-171, HeScheme.swift:1481-1485)
    //      -> he_bfv_mod_switch_down_device / he_bfv_mod_switch_down_to_single_device
    static func modulusSwitch(_ ciphertext: CanonicalCiphertext, toSingle: Bool) throws -> DeviceWork<CanonicalCiphertext> {
        precondition(ciphertext.correctionFactor == 1,
                     "BFV modulus switching not implemented for correction factor not equal to 1")
        let context = ciphertext.context
        let (level, polyContext, _) = shape(ciphertext)
        // divideAndRoundQLast drops the last modulus: `next` (PolyContext.swift:24, :366-368 throws without one)
        guard var lower = polyContext.next else {
            throw HeError.invalidPolyContext("modulus switching needs at least two moduli")
        }
        if toSingle {
            while lower.moduli.count > 1, let next = lower.next { lower = next }
        }
        let polyCount = ciphertext.polys.count
        let stream = try HeAmdStream()
        let slab = try resident(ciphertext, on: stream)
        let out = try DeviceBuffer(count: polyCount * lower.moduli.count * lower.degree)
        if toSingle {
            try heAmdCheck(he_bfv_mod_switch_down_to_single_device(context.gpu, level, UInt32(polyCount), slab.pointer,
                                                                   out.pointer, 1, stream.raw))
        } else {
            try heAmdCheck(he_bfv_mod_switch_down_device(context.gpu, level, UInt32(polyCount), slab.pointer, out.pointer,
                                                         1, stream.raw))
        }
        let result = lower
        return DeviceWork(stream: stream, buffers: [slab, out]) {
            try out.downloadCiphertext(context: context, polyContext: result, polyCount: polyCount, at: 0, on: stream)
        }
    }

    // ---- forwardNtt / inverseNtt of a ciphertext (Bfv.swift:654-669): correction factor and seed are preserved
    //      -> he_ntt_forward_device / he_ntt_inverse_device over [polys][L][N]
    static func transform<From: PolyFormat, To: PolyFormat>(_ ciphertext: Ciphertext<GpuBfv, From>,
                                                            inverse: Bool) throws -> DeviceWork<Ciphertext<GpuBfv, To>>
    {
        let context = ciphertext.context
        let (_, polyContext, polyWords) = shape(ciphertext)
        let polyCount = ciphertext.polys.count
        let correctionFactor = ciphertext.correctionFactor, seed = ciphertext.seed
        let stream = try HeAmdStream()
        let slab = try resident(ciphertext, on: stream)
        if inverse {
            try heAmdCheck(he_ntt_inverse_device(polyContext.gpu, slab.pointer, polyCount, stream.raw))
        } else {
            try heAmdCheck(he_ntt_forward_device(polyContext.gpu, slab.pointer, polyCount, stream.raw))
        }
        return DeviceWork(stream: stream, buffers: [slab]) {
            let polys: [PolyRq<UInt64, To>] = try (0..<polyCount).map { index in
                try slab.downloadPoly(context: polyContext, at: index * polyWords, on: stream)
            }
            return try Ciphertext<GpuBfv, To>(_context: context, _polys: polys, _correctionFactor: correctionFactor,
                                              _auxiliaryData: nil, _seed: seed)
        }
    }

    // ---- innerProduct(_: [ct], _: [ct]) (Bfv.swift:315-361): one dropExtendedBase for the whole sum
    //      -> he_bfv_inner_product_device
    static func ciphertextInnerProduct(_ lhs: [CanonicalCiphertext],
                                       _ rhs: [CanonicalCiphertext]) throws -> DeviceWork<CanonicalCiphertext>
    {
        precondition(lhs.count == rhs.count)
        guard let first = lhs.first else {
            preconditionFailure("Empty ciphertexts") // Bfv.swift:334-336
        }
        for (left, right) in zip(lhs, rhs) {
            try validateEquality(of: left.context, and: right.context) // validateInnerProductInput, Bfv.swift:224-235
            guard left.polys.count == freshCiphertextPolyCount, left.correctionFactor == 1 else {
                throw HeError.invalidCiphertext("lhs: \(left.polys.count) polynomials, correction factor \(left.correctionFactor)")
            }
            guard right.polys.count == freshCiphertextPolyCount, right.correctionFactor == 1 else {
                throw HeError.invalidCiphertext("rhs: \(right.polys.count) polynomials, correction factor \(right.correctionFactor)")
            }
            // the operands are laid out in `first`'s shape: one level for all (Bfv+Multiply.swift:73-75)
            guard left.polys[0].context == first.polys[0].context, right.polys[0].context == first.polys[0].context else {
                throw HeError.incompatibleCiphertexts("the ciphertexts are at different levels")
            }
        }
        let context = first.context
        let (level, polyContext, polyWords) = shape(first)
        let count = lhs.count
        let stream = try HeAmdStream()
        let left = try DeviceBuffer(count: count * 2 * polyWords), right = try DeviceBuffer(count: count * 2 * polyWords)
        try left.upload(contentsOf: lhs, at: 0, on: stream) // each operand vector: one staged copy
        try right.upload(contentsOf: rhs, at: 0, on: stream)
        let out = try DeviceBuffer(count: 3 * polyWords)
        try heAmdCheck(he_bfv_inner_product_device(context.gpu, level, left.pointer, right.pointer, count, out.pointer, nil,
                                                   0, stream.raw))
        return DeviceWork(stream: stream, buffers: [left, right, out]) {
            try out.downloadCiphertext(context: context, polyContext: polyContext, polyCount: 3, at: 0, on: stream)
        }
    }

    // ---- innerProduct(ciphertexts:plaintexts:) (Bfv.swift:476-505), nil plaintexts skipped (:494)
    //      -> he_bfv_inner_product_plain_device with one column
    static func plaintextInnerProduct(_ ciphertexts: [EvalCiphertext],
                                      _ plaintexts: [EvalPlaintext?]) throws -> DeviceWork<EvalCiphertext>
    {
        precondition(plaintexts.count == ciphertexts.count)
        guard let first = ciphertexts.first else {
            preconditionFailure("Empty ciphertexts")
        }
        precondition(ciphertexts.allSatisfy { $0.polys.count == first.polys.count },
                     "All ciphertexts must have the same polynomial count")
        let context = first.context
        let (level, polyContext, polyWords) = shape(first)
        // every operand is written at index * polyWords in `first`'s shape: all of them must be at its level.  The
        // reference checks each term as it multiplies it (lazyMultiply, Bfv.swift:365-394: validateEquality, then
        // ciphertext.moduli.count == plaintext.moduli.count -> incompatibleCiphertextAndPlaintext); ciphertexts of
        // different levels cannot be added (Bfv.swift:80-93 -> incompatibleCiphertexts).
        for (ciphertext, plaintext) in zip(ciphertexts, plaintexts) {
            try validateEquality(of: context, and: ciphertext.context)
            guard ciphertext.polys[0].context == polyContext else {
                throw HeError.incompatibleCiphertexts("the ciphertexts are at different levels")
            }
            guard let plaintext else { continue }
            try validateEquality(of: context, and: plaintext.context)
            guard plaintext._poly.context.moduli.count == polyContext.moduli.count else {
                throw HeError.incompatibleCiphertextAndPlaintext("ciphertext and plaintext levels differ")
            }
        }
        let polyCount = first.polys.count, count = ciphertexts.count
        let correctionFactor = first.correctionFactor
        let stream = try HeAmdStream()
        let vector = try DeviceBuffer(count: count * polyCount * polyWords)
        let factors = try DeviceBuffer(count: count * polyWords)
        try vector.upload(contentsOf: ciphertexts, at: 0, on: stream) // the ciphertext vector: one staged copy
        let present = try factors.upload(plaintexts: plaintexts, polyWords: polyWords, on: stream)
        let out = try DeviceBuffer(count: polyCount * polyWords)
        try present.withUnsafeBufferPointer { mask in // (the host-mask form waits for the mask's upload itself)
            try heAmdCheck(he_bfv_inner_product_plain_device(context.gpu, level, UInt32(polyCount), vector.pointer,
                                                             factors.pointer, mask.baseAddress, count, 1, out.pointer,
                                                             stream.raw))
        }
        return DeviceWork(stream: stream, buffers: [vector, factors, out]) {
            try out.downloadCiphertext(context: context, polyContext: polyContext, polyCount: polyCount, at: 0,
                                       correctionFactor: correctionFactor, on: stream)
        }
    }
}
