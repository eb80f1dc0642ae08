// The path's multi-GPU split for a Swift server (include/he_amd.h "Device groups", SURVEY.md 8e): ONE process, one context and
// one stream per GPU, a chunk's database sharded by COLUMN over the GPUs where it is uploaded, the query replicated on the
// way in and the finished columns gathered on the way out -- the partition the reference makes over tasks
// (Sources/PrivateInformationRetrieval/IndexPir/PirUtil.swift:424-445) made over devices.  Whole columns stay on one GPU, so
// nothing is reduced across GPUs; the remaining dimensions (PirUtil.swift:448-485) run on the home device (member 0).
//   GpuDeviceGroup            he_device_group: the members, he_shard_bounds, a member's device made current
//   GpuShardedChunk           the plaintexts of one or more chunks, member m holding its share of their columns (+ the nil mask), uploaded once
//   GpuPirUtil.computeResponseForChunks(group:...)     PirUtil.swift:408-486 for every chunk over the group, one C call
import CHeAmd
import HomomorphicEncryption
import PrivateInformationRetrieval

public final class GpuDeviceGroup: @unchecked Sendable {
    let raw: OpaquePointer
    /// The HIP device of every member; a device may be listed more than once (two shards on one GPU).
    public let devices: [Int32]

    /// One `Context<Bfv<UInt64>>` per member, built from the same parameters on each device (Context.swift:94-143).
    /// `stageAll`: every member but the first goes through a remote device's copies even on the same GPU (tests on one GPU).
    public init(devices: [Int32], degree: Int, plaintextModulus: UInt64, coefficientModuli: [UInt64],
                stageAll: Bool = false) throws
    {
        var handle: OpaquePointer?
        try devices.withUnsafeBufferPointer { ids in
            try coefficientModuli.withUnsafeBufferPointer { moduli in
                try heAmdCheck(he_device_group_create(ids.baseAddress, UInt32(ids.count),
                                                      stageAll ? UInt32(HE_GROUP_STAGE_ALL) : 0, UInt32(degree),
                                                      plaintextModulus, moduli.baseAddress, UInt32(moduli.count),
                                                      &handle))
            }
        }
        guard let handle else { throw HeError.unsupportedHeOperation(description: "he_device_group_create returned nil") }
        raw = handle
        self.devices = devices
    }

    deinit {
        he_device_group_destroy(raw)
    }

    public var count: Int { Int(he_device_group_size(raw)) }

    /// The units of `total` member `member` owns: the first `total % count` members one more than the others.
    public func bounds(of total: Int, member: Int) throws -> Range<Int> {
        var begin = 0, end = 0
        try heAmdCheck(he_shard_bounds(total, UInt32(count), UInt32(member), &begin, &end))
        return begin..<end
    }

    /// Makes member `member`'s device the calling thread's current one and returns the device that was current: buffers and
    /// streams belong to the device that was current when they were made (he_amd.h `he_get_device`).  Pair with `restore`.
    func makeCurrent(member: Int) throws -> Int32 {
        var previous: Int32 = 0
        try heAmdCheck(he_get_device(&previous))
        try heAmdCheck(he_set_device(devices[member]))
        return previous
    }

    func restore(device previous: Int32) {
        _ = he_set_device(previous)
    }

    /// Blocks until every member's stream has drained.
    public func synchronize() throws {
        try heAmdCheck(he_device_group_synchronize(raw))
    }
}

/// `chunkCount` chunks of a database -- each `prod(dimensions)` optional Eval plaintexts, plaintext k of column c at index
/// c * d0 + k (MulPir.swift:547-555), the chunks back to back, so that their columns form ONE column range -- with member m of
/// `group` holding its share of that range in its own HBM.
public final class GpuShardedChunk<Scheme: HeScheme>: @unchecked Sendable where Scheme.Scalar == UInt64 {
    public let group: GpuDeviceGroup
    public let dimensions: [Int]
    public let chunkCount: Int
    /// Per member: its columns' plaintexts `[share][d0][L][N]` and their nil mask (one byte per plaintext); nil = no column.
    let shards: [DeviceBuffer?]
    let masks: [DeviceBuffer?]

    public init(_ dataChunk: some Collection<Plaintext<Scheme, Eval>?>, dimensions: [Int], chunkCount: Int = 1,
                polyContext: PolyContext<UInt64>, group: GpuDeviceGroup) throws
    {
        let polyWords = polyContext.moduli.count * polyContext.degree
        let perChunk = dimensions.reduce(1, *)
        let d0 = dimensions[0], columns = perChunk / d0 * chunkCount
        precondition(dataChunk.count >= perChunk * chunkCount)
        let plaintexts = Array(dataChunk.prefix(perChunk * chunkCount))
        var shards: [DeviceBuffer?] = [], masks: [DeviceBuffer?] = []
        for member in 0..<group.count {
            let mine = try group.bounds(of: columns, member: member)
            if mine.isEmpty {
                shards.append(nil)
                masks.append(nil)
                continue
            }
            let previous = try group.makeCurrent(member: member)
            defer { group.restore(device: previous) }
            let stream = try HeAmdStream()
            let count = mine.count * d0
            let shard = try DeviceBuffer(count: count * polyWords)
            let mask = try DeviceBuffer(count: (count + 7) / 8) // the mask travels as bytes; buffers count 8-byte words
            let present = try shard.upload(plaintexts: plaintexts[mine.lowerBound * d0..<mine.upperBound * d0],
                                           polyWords: polyWords, on: stream)
            try present.withUnsafeBufferPointer { bytes in
                try mask.upload(bytes: bytes, atByte: 0, on: stream)
            }
            try stream.synchronize()
            shards.append(shard)
            masks.append(mask)
        }
        self.group = group
        self.dimensions = dimensions
        self.chunkCount = chunkCount
        self.shards = shards
        self.masks = masks
    }

    var shardPointers: [UnsafePointer<UInt64>?] {
        shards.map { $0.map { UnsafePointer($0.pointer) } }
    }

    var maskPointers: [UnsafePointer<UInt8>?] {
        masks.map { $0.map { UnsafePointer(UnsafeRawPointer($0.pointer).assumingMemoryBound(to: UInt8.self)) } }
    }
}

public extension GpuPirUtil {
    /// `computeResponseForOneChunk` (PirUtil.swift:408-486) for every chunk of `chunk` over a device group -- the chunk loop of
    /// `computeResponse` (PirUtil.swift:533-563): the dim-0 inner products of every column on the GPU that holds the column
    /// (PirUtil.swift:428-446), the columns gathered on the home device, the remaining dimensions and
    /// `modSwitchDownToSingle` there (PirUtil.swift:448-485).  The queries and the key go to the home device.  One response
    /// ciphertext per chunk.
    static func computeResponseForChunks(
        group: GpuDeviceGroup,
        expandedDim0Query: [Ciphertext<Scheme, Eval>],
        expandedRemainingQuery: [CanonicalCiphertext],
        chunk: GpuShardedChunk<Scheme>,
        using evaluationKey: EvaluationKey<Scheme>,
        callOptions _: CallOptions) async throws -> [Ciphertext<Scheme, Coeff>]
    {
        guard let first = expandedDim0Query.first else {
            throw HeError.incompatibleCiphertextCount("empty dim-0 query")
        }
        let context = first.context
        let polyContext = first.polys[0].context
        let degree = polyContext.degree, polyWords = polyContext.moduli.count * degree
        let chunkCount = chunk.chunkCount
        let columns = chunk.dimensions.reduce(1, *) / chunk.dimensions[0]
        precondition(columns == 1 || columns == expandedRemainingQuery.count) // PirUtil.swift:422
        let dimensions = chunk.dimensions.map { UInt32($0) }
        let remainingCount = expandedRemainingQuery.count
        let shardPointers = chunk.shardPointers, maskPointers = chunk.maskPointers
        // everything the caller hands over lives on the home device (member 0's); the thread's device is put back before the
        // task suspends (it may resume on another thread)
        let stream: HeAmdStream, dim0: DeviceBuffer, rest: DeviceBuffer, response: DeviceBuffer, keys: GpuEvaluationKey
        do {
            let previous = try group.makeCurrent(member: 0)
            defer { group.restore(device: previous) }
            stream = try HeAmdStream()
            dim0 = try DeviceBuffer(count: expandedDim0Query.count * 2 * polyWords)
            try dim0.upload(contentsOf: expandedDim0Query, at: 0, on: stream)
            rest = try DeviceBuffer(count: max(remainingCount, 1) * 2 * polyWords)
            try rest.upload(contentsOf: expandedRemainingQuery, at: 0, on: stream)
            keys = try GpuResidentCache.shared.resident(evaluationKey)
            if chunk.dimensions.count > 1, keys.relinearizationKey == nil {
                throw HeError.missingRelinearizationKey
            }
            response = try DeviceBuffer(count: chunkCount * 2 * degree) // [chunk][2][1][N] after modSwitchDownToSingle
            try dimensions.withUnsafeBufferPointer { dims in
                try shardPointers.withUnsafeBufferPointer { slabs in
                    try maskPointers.withUnsafeBufferPointer { masks in
                        try heAmdCheck(he_pir_compute_response_group(
                            group.raw, dims.baseAddress, UInt32(dims.count), dim0.pointer, rest.pointer, remainingCount,
                            slabs.baseAddress, masks.baseAddress, chunkCount, keys.relinearizationKey?.buffer.pointer,
                            response.pointer, stream.raw))
                    }
                }
            }
        }
        try await stream.completion()
        let single = singleModulusContext(of: polyContext)
        return try withExtendedLifetime((dim0, rest, keys, chunk)) {
            let resumed = try group.makeCurrent(member: 0)
            defer { group.restore(device: resumed) }
            return try (0..<chunkCount).map { index in
                try response.downloadCiphertext(context: context, polyContext: single, polyCount: 2, at: index * 2 * degree,
                                                on: stream)
            }
        }
    }
}
