"""Threading and stream contract of the C ABI (SURVEY.md 8b "Threading": every entry point is re-entrant on a shared
context and works on the caller's stream): concurrent host threads on their own HIP streams, and capture of the
multi-kernel BFV pipelines into a HIP graph."""
import threading

import numpy as np
import pytest

import heamd

pytestmark = pytest.mark.gpu


def _slab(rng, batch, moduli, degree):
    return np.stack([rng.integers(0, q, size=(batch, degree), dtype=np.uint64) for q in moduli], axis=1).copy()


def test_threads_share_one_context(oracle):
    """The reference calls the path from many tasks on one shared context (Bfv.swift:270-287)."""
    import torch

    degree = 4096
    moduli = oracle.generate_primes([55, 55, 50], False, degree)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(90)
    inputs = [_slab(rng, 3, moduli, degree) for _ in range(4)]
    expected = [ref.forward_ntt(x) for x in inputs]
    results, errors = [None] * 4, []

    def worker(index):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                slab = heamd.to_device(inputs[index])
                for _ in range(5):  # forward / inverse round trips, ending on a forward transform
                    ours.forward_ntt_(slab, stream=stream)
                    ours.inverse_ntt_(slab, stream=stream)
                ours.forward_ntt_(slab, stream=stream)
                stream.synchronize()
                results[index] = heamd.to_host(slab)
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for got, want in zip(results, expected):
        assert np.array_equal(got, want)


def test_row_fused_mul_shares_the_side_lane_and_is_capturable(oracle):
    """ct x ct on batches that take the row-fused kernels with the Q band on a lane of the context (a second stream forked off
    the caller's and joined back inside the call, csrc/side_lane.hpp): four host threads on their own streams share one
    context -- each stream gets a lane of its own from the context's pool -- and every product equals the oracle's; the same
    call captured into a HIP graph (where no lane is used) replays on new operands."""
    import torch
    from conftest import host_threads

    degree, batch = 4096, 512
    q = oracle.generate_primes([55, 55, 55], False, degree)
    t = oracle.generate_primes([17], True, degree)[0]
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    moduli = q[:-1]
    assert batch * ours.L >= 1024
    rng = np.random.default_rng(92)

    def uniform(prefix):
        rows = [rng.integers(0, m, size=tuple(prefix) + (degree,), dtype=np.uint64) for m in moduli]
        return np.ascontiguousarray(np.stack(rows, axis=len(prefix)))

    operands = [(uniform((batch, 2)), uniform((batch, 2))) for _ in range(4)]
    expected = [ref.mul(a, b, threads=host_threads()) for a, b in operands]
    results, errors = [None] * 4, []

    def worker(index):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                lhs, rhs = heamd.to_device(operands[index][0]), heamd.to_device(operands[index][1])
                out = None
                for _ in range(3):
                    out = ours.mul(lhs, rhs, stream=stream)
                stream.synchronize()
                results[index] = heamd.to_host(out)
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for got, want in zip(results, expected):
        assert np.array_equal(got, want)
    # captured: one stream, replayed on new operands
    lhs_static, rhs_static = heamd.to_device(operands[0][0]), heamd.to_device(operands[0][1])
    workspace = torch.empty(ours.mul_workspace_bytes(batch) // 8, dtype=torch.int64, device="cuda")
    ours.mul(lhs_static, rhs_static, workspace=workspace)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_static = ours.mul(lhs_static, rhs_static, workspace=workspace)
    for trial in (1, 2):
        lhs_static.copy_(heamd.to_device(operands[trial][0]))
        rhs_static.copy_(heamd.to_device(operands[trial][1]))
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(heamd.to_host(out_static), expected[trial]), trial


def test_mul_relinearize_pipeline_in_a_hip_graph(oracle):
    """ct x ct + relinearize is ten kernel launches; with caller-provided workspaces nothing in it allocates or
    synchronises, so it can be captured once and replayed (hipGraph) on new inputs."""
    import torch

    degree = 1024
    q = oracle.generate_primes([40, 40, 41], False, degree)
    t = oracle.generate_primes([17], True, degree)[0]
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    rng = np.random.default_rng(91)
    moduli = q[:-1]
    batch = 3

    def uniform(prefix, mods):
        rows = [rng.integers(0, m, size=tuple(prefix) + (degree,), dtype=np.uint64) for m in mods]
        return np.ascontiguousarray(np.stack(rows, axis=len(prefix)))

    key = uniform((ours.L, 2), q)
    key_dev = heamd.to_device(key)
    lhs_static = heamd.to_device(uniform((batch, 2), moduli))
    rhs_static = heamd.to_device(uniform((batch, 2), moduli))
    ws_mul = torch.empty(ours.mul_workspace_bytes(batch) // 8, dtype=torch.int64, device="cuda")
    ws_relin = torch.empty(ours.relinearize_workspace_bytes(batch) // 8, dtype=torch.int64, device="cuda")
    # warm-up outside the capture (first-use attribute setup), then capture
    ours.relinearize(ours.mul(lhs_static, rhs_static, workspace=ws_mul), key_dev, workspace=ws_relin)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_static = ours.relinearize(ours.mul(lhs_static, rhs_static, workspace=ws_mul), key_dev, workspace=ws_relin)
    for trial in range(2):
        lhs, rhs = uniform((batch, 2), moduli), uniform((batch, 2), moduli)
        lhs_static.copy_(heamd.to_device(lhs))
        rhs_static.copy_(heamd.to_device(rhs))
        graph.replay()
        torch.cuda.synchronize()
        expected = ref.relinearize(ref.mul(lhs, rhs), key)
        assert np.array_equal(heamd.to_host(out_static), expected), trial


def test_masked_inner_product_is_enqueue_only_and_graph_capturable(oracle):
    """With the nil-plaintext mask resident on the device, Bfv.innerProduct(ciphertexts:plaintexts:) never synchronises:
    it can be captured into a HIP graph and replayed on new operands and a new mask; the host-mask form gives the same
    words."""
    import torch

    degree = 1024
    q = oracle.generate_primes([40, 40, 41], False, degree)
    t = oracle.generate_primes([17], True, degree)[0]
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    rng = np.random.default_rng(92)
    moduli = q[:-1]
    count, columns = 6, 3

    def uniform(prefix):
        rows = [rng.integers(0, m, size=tuple(prefix) + (degree,), dtype=np.uint64) for m in moduli]
        return np.ascontiguousarray(np.stack(rows, axis=len(prefix)))

    cts_static = heamd.to_device(uniform((count, 2)))
    pts_static = heamd.to_device(uniform((columns, count)))
    mask_static = torch.ones((columns, count), dtype=torch.uint8, device="cuda")
    ours.inner_product_plain_resident(cts_static, pts_static, mask_static, 2, columns)  # warm-up outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_static = ours.inner_product_plain_resident(cts_static, pts_static, mask_static, 2, columns)
    for trial in range(2):
        cts, pts = uniform((count, 2)), uniform((columns, count))
        mask = (rng.integers(0, 4, size=(columns, count)) != 0).astype(np.uint8)
        cts_static.copy_(heamd.to_device(cts))
        pts_static.copy_(heamd.to_device(pts))
        mask_static.copy_(torch.from_numpy(mask).cuda())
        graph.replay()
        torch.cuda.synchronize()
        via_host_mask = heamd.to_host(ours.inner_product_plain(heamd.to_device(cts), heamd.to_device(pts), mask, 2, columns))
        assert np.array_equal(heamd.to_host(out_static), via_host_mask), trial
        for c in range(columns):
            expected = ref.inner_product_plain(cts, pts[c], mask[c])
            assert np.array_equal(via_host_mask[c], expected), (trial, c)


def test_events_and_stream_callbacks(oracle):
    """The completion primitives the `...Async` twins await (HeSchemeAsync.swift:16-141): an event recorded after an
    enqueued transform reports completion, a second stream can wait on it, and a host callback fires after the work."""
    import ctypes
    import threading as th

    import torch

    from heamd import binding

    lib = binding.load_library()
    degree = 4096
    moduli = oracle.generate_primes([55, 55], False, degree)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(93)
    x = _slab(rng, 4, moduli, degree)
    slab = heamd.to_device(x)
    torch.cuda.synchronize()
    producer, consumer, event = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.he_stream_create(ctypes.byref(producer)) == 0 and lib.he_stream_create(ctypes.byref(consumer)) == 0
    assert lib.he_event_create(ctypes.byref(event)) == 0
    fired = th.Event()
    seen = []

    def on_done(user_data):
        seen.append(user_data)
        fired.set()

    callback = binding.HOST_CALLBACK(on_done)
    assert lib.he_ntt_forward_device(ours.h, ctypes.c_void_p(slab.data_ptr()), 4, producer) == 0
    assert lib.he_event_record(event, producer) == 0
    assert lib.he_stream_wait_event(consumer, event) == 0
    assert lib.he_ntt_inverse_device(ours.h, ctypes.c_void_p(slab.data_ptr()), 4, consumer) == 0  # after the forward
    assert lib.he_stream_add_callback(consumer, callback, ctypes.c_void_p(0x5EED)) == 0
    assert fired.wait(timeout=30), "the stream callback never fired"
    assert seen == [0x5EED]
    done = ctypes.c_int(0)
    assert lib.he_event_query(event, ctypes.byref(done)) == 0 and done.value == 1
    assert lib.he_event_synchronize(event) == 0
    assert np.array_equal(heamd.to_host(slab), x)  # forward then inverse, ordered across the two streams by the event
    assert np.array_equal(heamd.to_host(ours.forward_ntt_(slab)), ref.forward_ntt(x))
    assert lib.he_event_destroy(event) == 0
    assert lib.he_stream_destroy(producer) == 0 and lib.he_stream_destroy(consumer) == 0


def test_pinned_host_staging_round_trip(oracle):
    """he_host_malloc / he_host_free: a ciphertext's polynomials staged back to back in page-locked memory go up with ONE
    asynchronous copy and no wait before the transform is enqueued behind it on the same stream -- what the Swift host's
    DeviceBuffer does for a ciphertext, a key or a database block -- and the result read back into the same staging
    equals the oracle's transform."""
    import ctypes

    from heamd import binding

    lib = binding.load_library()
    degree = 4096
    moduli = oracle.generate_primes([55, 55], False, degree)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    x = _slab(np.random.default_rng(94), 3, moduli, degree)
    nbytes = x.nbytes
    staging, device, stream = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.he_host_malloc(ctypes.byref(staging), nbytes) == 0 and staging.value
    assert lib.he_device_malloc(ctypes.byref(device), nbytes) == 0
    assert lib.he_stream_create(ctypes.byref(stream)) == 0
    ctypes.memmove(staging, x.ctypes.data, nbytes)
    assert lib.he_memcpy_h2d(device, staging, nbytes, stream) == 0
    assert lib.he_ntt_forward_device(ours.h, device, 3, stream) == 0
    assert lib.he_memcpy_d2h(staging, device, nbytes, stream) == 0
    assert lib.he_stream_synchronize(stream) == 0
    got = np.frombuffer((ctypes.c_uint8 * nbytes).from_address(staging.value), dtype=np.uint64).reshape(x.shape).copy()
    assert np.array_equal(got, ref.forward_ntt(x))
    assert lib.he_host_free(staging) == 0 and lib.he_host_free(None) == 0
    assert lib.he_device_free(device) == 0 and lib.he_stream_destroy(stream) == 0
    assert lib.he_host_malloc(None, 8) == 16  # invalidArgument


def test_scratch_cache_is_enqueue_only_and_shared_by_streams(oracle):
    """he_set_scratch_cache: with the library's block cache on, calls that take scratch (relinearize without a workspace)
    return to the host while their work is still queued -- the HIP pool's hipFreeAsync holds every such call until the
    previous one's work has finished (profiles/r06b_pool_probe.txt) -- the cache stops growing once the shapes have been
    seen, two streams taking turns on the same blocks (an event wait, no driver allocation) both give the oracle's words,
    the same call captured into a graph still works (scratch from the HIP pool there), and a trim hands everything back."""
    import time

    import torch

    degree, batch = 4096, 512
    q = oracle.generate_primes([55, 55, 55], False, degree)
    t = oracle.generate_primes([17], True, degree)[0]
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    moduli = q[:-1]
    rng = np.random.default_rng(93)

    def uniform(prefix, mods):
        rows = [rng.integers(0, m, size=tuple(prefix) + (degree,), dtype=np.uint64) for m in mods]
        return np.ascontiguousarray(np.stack(rows, axis=len(prefix)))

    ct3, key = uniform((batch, 3), moduli), uniform((ours.L, 2), q)
    expected = ref.relinearize(ct3, key)
    ct3_device, key_device = heamd.to_device(ct3), heamd.to_device(key)
    heamd.trim_scratch(0)
    heamd.set_scratch_cache()
    try:
        assert heamd.scratch_cached_bytes() == 0
        first = ours.relinearize(ct3_device, key_device)
        torch.cuda.synchronize()
        held = heamd.scratch_cached_bytes()
        assert held >= ours.relinearize_workspace_bytes(batch)
        # enqueue-only: many calls queue up behind one another without the host waiting for any of them
        calls = 40
        begin = time.perf_counter()
        outs = [ours.relinearize(ct3_device, key_device) for _ in range(calls)]
        host_s = time.perf_counter() - begin
        torch.cuda.synchronize()
        device_s = time.perf_counter() - begin
        assert heamd.scratch_cached_bytes() == held, "the cache grew on a shape it had seen"
        assert host_s < 0.6 * device_s, (host_s, device_s)
        for out in (first, outs[0], outs[-1]):
            assert np.array_equal(heamd.to_host(out), expected)
        # another stream takes the cached block over (the event orders it behind the last release), then the first again
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            other = ours.relinearize(ct3_device, key_device, stream=side)
        back = ours.relinearize(ct3_device, key_device)
        side.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(heamd.to_host(other), expected) and np.array_equal(heamd.to_host(back), expected)
        assert heamd.scratch_cached_bytes() <= 2 * held
        # captured: allocation nodes of the HIP pool, the cache is left alone
        before = heamd.scratch_cached_bytes()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = ours.relinearize(ct3_device, key_device)
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(heamd.to_host(captured), expected)
        assert heamd.scratch_cached_bytes() == before
        heamd.trim_scratch(0)
        assert heamd.scratch_cached_bytes() == 0
    finally:
        heamd.set_scratch_cache(0)


def test_scratch_cache_with_a_small_bound_evicts_and_stays_correct(oracle):
    """he_set_scratch_cache(bytes) with a bound smaller than one call's scratch: released blocks beyond the bound go back to the
    driver once their release has completed (blocks still in flight stay), two threads on their own streams keep calling through
    it, every result is the oracle's, and what stays cached after a synchronisation and another release is within the bound."""
    import torch

    degree, batch = 4096, 96
    q = oracle.generate_primes([55, 55, 55], False, degree)
    t = oracle.generate_primes([17], True, degree)[0]
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    moduli = q[:-1]
    rng = np.random.default_rng(94)

    def uniform(prefix, mods):
        rows = [rng.integers(0, m, size=tuple(prefix) + (degree,), dtype=np.uint64) for m in mods]
        return np.ascontiguousarray(np.stack(rows, axis=len(prefix)))

    ct3, key = uniform((batch, 3), moduli), uniform((ours.L, 2), q)
    expected = ref.relinearize(ct3, key)
    ct3_device, key_device = heamd.to_device(ct3), heamd.to_device(key)
    bound = 1 << 20
    assert ours.relinearize_workspace_bytes(batch) > 4 * bound
    heamd.trim_scratch(0)
    heamd.set_scratch_cache(bound)
    results, errors = {}, []

    def worker(index):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                outs = [ours.relinearize(ct3_device, key_device, stream=stream) for _ in range(6)]
                stream.synchronize()
                results[index] = [heamd.to_host(o) for o in (outs[0], outs[-1])]
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    try:
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors
        for outs in results.values():
            for out in outs:
                assert np.array_equal(out, expected)
        torch.cuda.synchronize()
        last = ours.relinearize(ct3_device, key_device)  # its release evicts what has completed
        torch.cuda.synchronize()
        assert np.array_equal(heamd.to_host(last), expected)
        once_more = ours.relinearize(ct3_device, key_device)
        torch.cuda.synchronize()
        assert np.array_equal(heamd.to_host(once_more), expected)
        assert heamd.scratch_cached_bytes() <= ours.relinearize_workspace_bytes(batch) + (2 << 20)
        heamd.trim_scratch(0)
        assert heamd.scratch_cached_bytes() == 0
    finally:
        heamd.set_scratch_cache(0)
