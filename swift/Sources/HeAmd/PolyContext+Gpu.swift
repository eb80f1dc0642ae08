// Level B1/B2 of include/he_amd.h behind PolyContext<UInt64> / PolyRq<UInt64, _>: the transforms of
// Sources/HomomorphicEncryption/PolyRq/PolyRq+Ntt.swift:209-232, 524-543 and the element-wise operators of
// PolyRq/PolyRq.swift:147-245, 299-309, on batches that stay resident on the device between calls.
import CHeAmd
import HomomorphicEncryption

extension PolyContext where T == UInt64 {
    /// `PolyRq<UInt64, Coeff>.forwardNtt()` on one polynomial through the raw-pointer seam the reference already has
    /// (PolyRq+Ntt.swift:215-218): host pointer in, blocking.  PCIe-bound -- it exists for parity, not throughput.
    public func gpuForwardNtt(_ poly: consuming PolyRq<UInt64, Coeff>) throws -> PolyRq<UInt64, Eval> {
        var data = poly.data
        let handle = try gpu
        try data.withMutableDataSpan { span in // Array2d.swift:107
            try span.withUnsafeMutableBufferPointer { words in
                try heAmdCheck(he_ntt_forward(handle, words.baseAddress, 1))
            }
        }
        return PolyRq(context: self, data: data)
    }

    /// `PolyRq<UInt64, Eval>.inverseNtt()`, same seam.
    public func gpuInverseNtt(_ poly: consuming PolyRq<UInt64, Eval>) throws -> PolyRq<UInt64, Coeff> {
        var data = poly.data
        let handle = try gpu
        try data.withMutableDataSpan { span in
            try span.withUnsafeMutableBufferPointer { words in
                try heAmdCheck(he_ntt_inverse(handle, words.baseAddress, 1))
            }
        }
        return PolyRq(context: self, data: data)
    }

    /// Forward NTT of `batch` polynomials resident at `slab` ([batch][L][N]), in place; enqueue-only.
    public func gpuForwardNtt(resident slab: DeviceBuffer, batch: Int, on stream: HeAmdStream) throws {
        try heAmdCheck(he_ntt_forward_device(gpu, slab.pointer, batch, stream.raw))
    }

    /// Inverse NTT of `batch` resident polynomials, in place; enqueue-only.
    public func gpuInverseNtt(resident slab: DeviceBuffer, batch: Int, on stream: HeAmdStream) throws {
        try heAmdCheck(he_ntt_inverse_device(gpu, slab.pointer, batch, stream.raw))
    }

    /// `lhs += rhs`, `-=`, `*=` (Eval) and negation over `batch` resident polynomials (PolyRq.swift:147-245).
    public func gpuAdd(_ lhs: DeviceBuffer, _ rhs: DeviceBuffer, batch: Int, on stream: HeAmdStream) throws {
        try heAmdCheck(he_poly_add_device(gpu, lhs.pointer, rhs.pointer, batch, stream.raw))
    }

    public func gpuSubtract(_ lhs: DeviceBuffer, _ rhs: DeviceBuffer, batch: Int, on stream: HeAmdStream) throws {
        try heAmdCheck(he_poly_sub_device(gpu, lhs.pointer, rhs.pointer, batch, stream.raw))
    }

    public func gpuMultiply(_ lhs: DeviceBuffer, _ rhs: DeviceBuffer, batch: Int, on stream: HeAmdStream) throws {
        try heAmdCheck(he_poly_mul_device(gpu, lhs.pointer, rhs.pointer, batch, stream.raw))
    }

    public func gpuNegate(_ data: DeviceBuffer, batch: Int, on stream: HeAmdStream) throws {
        try heAmdCheck(he_poly_neg_device(gpu, data.pointer, batch, stream.raw))
    }

    /// `PolyRq.divideAndRoundQLast()` (PolyRq.swift:365-393) over `batch` resident polynomials:
    /// [batch][L][N] -> [batch][L-1][N].
    public func gpuDivideAndRoundQLast(_ input: DeviceBuffer, into output: DeviceBuffer, batch: Int,
                                       on stream: HeAmdStream) throws
    {
        try heAmdCheck(he_poly_divide_and_round_q_last_device(gpu, input.pointer, output.pointer, batch, stream.raw))
    }
}
