// A HIP stream owned by Swift, and the completion primitive the reference's `...Async` twins await
// (Sources/HomomorphicEncryption/HeSchemeAsync.swift:16-141 default every async requirement to its synchronous twin;
// a GPU-backed twin enqueues the `_device` call and suspends until the stream reaches that point -- no thread of the
// cooperative pool sits in he_stream_synchronize).
import CHeAmd

/// `he_stream` (a non-blocking hipStream_t).  Every `*_device` entry point only enqueues on it.
public final class HeAmdStream: @unchecked Sendable {
    public let raw: he_stream?

    public init() throws {
        var stream: he_stream?
        try heAmdCheck(he_stream_create(&stream))
        raw = stream
    }

    deinit {
        _ = he_stream_destroy(raw)
    }

    /// Blocks the calling thread until the stream is idle (tests and synchronous twins only).
    public func synchronize() throws {
        try heAmdCheck(he_stream_synchronize(raw))
    }

    /// Suspends the calling task until everything enqueued so far has finished; the thread is released meanwhile.
    public func completion() async throws {
        try await withCheckedThrowingContinuation { (continuation: CheckedContinuation<Void, any Error>) in
            let box = Unmanaged.passRetained(ContinuationBox(continuation))
            let status = he_stream_add_callback(raw, { userData in
                guard let userData else { return }
                Unmanaged<ContinuationBox>.fromOpaque(userData).takeRetainedValue().continuation.resume()
            }, box.toOpaque())
            if status != Int32(HE_OK.rawValue) {
                let unused = box.takeRetainedValue()
                do { try heAmdCheck(status) } catch { unused.continuation.resume(throwing: error) }
            }
        }
    }
}

private final class ContinuationBox {
    let continuation: CheckedContinuation<Void, any Error>
    init(_ continuation: CheckedContinuation<Void, any Error>) { self.continuation = continuation }
}

/// `he_event`: fork / join between streams (`he_stream_wait_event`) and a pollable completion flag.
public final class HeAmdEvent: @unchecked Sendable {
    public let raw: he_event?

    public init() throws {
        var event: he_event?
        try heAmdCheck(he_event_create(&event))
        raw = event
    }

    deinit {
        _ = he_event_destroy(raw)
    }

    public func record(on stream: HeAmdStream) throws {
        try heAmdCheck(he_event_record(raw, stream.raw))
    }

    public var isComplete: Bool {
        get throws {
            var done: Int32 = 0
            try heAmdCheck(he_event_query(raw, &done))
            return done != 0
        }
    }
}
