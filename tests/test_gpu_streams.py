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
