"""The N > 1 path on CPU: two processes over gloo exercise the product's sharding / gather / clock-reduction helpers
(heamd/sharding.py, used by bench.py with RCCL on GPUs).  The per-shard compute here is the CPU oracle -- this test
is about who owns which polynomials and that the gathered job equals the unsharded transform, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for extra in (ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd")):
    if extra not in sys.path:
        sys.path.insert(0, extra)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _global_slab(total, moduli, degree):
    rng = np.random.default_rng(20240924)
    return np.stack([rng.integers(0, q, size=(total, degree), dtype=np.uint64) for q in moduli], axis=1).copy()


def _worker(rank, world, port, total, degree, bits, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist

    import oracle
    from heamd import sharding

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        assert sharding.rank_and_world() == (rank, rank, world)
        moduli = oracle.generate_primes(bits, False, degree)
        ctx = oracle.PolyContext(degree, moduli)
        full = _global_slab(total, moduli, degree)
        begin, end = sharding.shard_bounds(total, world, rank)
        shard = full[begin:end].copy()
        if end > begin:
            ctx.forward_ntt_inplace(shard, threads=1)
        local = torch.from_numpy(shard.view(np.int64))
        gathered = sharding.gather_shards(local, total)
        got = gathered.numpy().view(np.uint64)
        expected = ctx.forward_ntt(full) if total else full
        slowest = sharding.max_over_ranks(1.0 + rank)
        results[rank] = (bool(np.array_equal(got, expected)), tuple(got.shape), slowest, (begin, end))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7, 1])
def test_two_rank_sharded_ntt_matches_unsharded(total):
    import torch.multiprocessing as mp

    world, degree, bits = 2, 256, [40, 41]
    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_worker, args=(world, _free_port(), total, degree, bits, results), nprocs=world, join=True)
    assert sorted(results.keys()) == [0, 1]
    covered = []
    for rank in range(world):
        equal, shape, slowest, bounds = results[rank]
        assert equal, f"rank {rank}: gathered result differs from the unsharded transform"
        assert shape == (total, len(bits), degree)
        assert slowest == 2.0  # MAX over ranks of (1 + rank)
        covered.append(bounds)
    assert covered[0][0] == 0 and covered[0][1] == covered[1][0] and covered[1][1] == total


def _pir_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist

    import oracle
    from heamd import sharding

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        degree, dims = 32, [3, 5]
        t = oracle.generate_primes([17], True, degree)[0]
        q = oracle.generate_primes([40, 40, 41], False, degree)
        bfv = oracle.BfvContext(degree, t, q)
        moduli = bfv.ciphertext_context().moduli
        rng = np.random.default_rng(77)  # every rank draws the same job; it keeps only its column shard of the database

        def uniform(prefix, mods):
            return np.ascontiguousarray(np.stack(
                [rng.integers(0, m, size=tuple(prefix) + (degree,), dtype=np.uint64) for m in mods], axis=len(prefix)))

        dim0, rest = uniform((dims[0], 2), moduli), uniform((dims[1], 2), moduli)
        database = uniform((dims[1], dims[0]), moduli)  # [column][row]
        key = uniform((bfv.L, 2), q)
        begin, end = sharding.shard_bounds(dims[1], world, rank)  # this rank's columns (bench.py --workload c5)
        mine = oracle.pir.dim0_columns(bfv, dim0, database[begin:end])
        gathered = sharding.gather_shards(torch.from_numpy(mine.view(np.int64)), dims[1]).numpy().view(np.uint64)
        response = oracle.pir.remaining_dimensions(bfv, dims, gathered, rest, key)
        whole = oracle.pir.compute_response_for_one_chunk(bfv, dims, dim0, rest, database.reshape(-1, bfv.L, degree),
                                                          None, key)
        results[rank] = (bool(np.array_equal(response, whole)), (begin, end))
    finally:
        dist.destroy_process_group()


def test_two_rank_column_sharded_pir_response_matches_unsharded():
    """BASELINE configs[4]'s partitioning on CPU: the database is sharded by column over two gloo ranks, each computes
    its columns' dim-0 inner products (PirUtil.swift:427-445), the shards are all-gathered (ragged: 3 + 2 columns) and
    the remaining dimensions give the unsharded chunk response word for word."""
    import torch.multiprocessing as mp

    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_pir_worker, args=(2, _free_port(), results), nprocs=2, join=True)
    assert results[0] == (True, (0, 3)) and results[1] == (True, (3, 5))


def _bench_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import argparse
    import importlib.util

    import torch
    import torch.distributed as dist

    import oracle
    from heamd import sharding

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        degree, total_polys = 64, 7
        begin, end = sharding.shard_bounds(total_polys, world, rank)
        per_rank = end - begin  # ragged shards: 4 and 3 polynomials
        moduli = oracle.generate_primes([40, 41], False, degree)
        ctx = oracle.PolyContext(degree, moduli)

        class CpuJob:  # the workload interface of bench.py on CPU tensors (the oracle stands in for the kernels)
            def __init__(self):
                self.total = total_polys
                rng = np.random.default_rng(5 + rank)
                self.slab = np.stack([rng.integers(0, q, size=(per_rank, degree), dtype=np.uint64) for q in moduli], axis=1)
                self.units = 2 * per_rank
                self.steps_run = 0

            def step(self):
                ctx.forward_ntt_inplace(self.slab, threads=1)
                ctx.inverse_ntt_inplace(self.slab, threads=1)
                self.steps_run += 1

            def result(self):
                return torch.from_numpy(self.slab.view(np.int64)), self.total

            def describe(self, w):
                return {"metric": "m", "unit": "u", "dtype": "u64", "config": {"workload": "cpu dry run", "world": w}}

            def roofline(self, steps, r):
                return {"bound": "hbm", "frac": 0.0}, {"note": "dry run"}

        jobs = []

        def make_job():
            jobs.append(CpuJob())
            return jobs[-1]

        args = argparse.Namespace(steps=3, warmup=2, skip_gather=False)
        line = bench.run_benchmark(args, make_job, rank, world, device="cpu", dist=dist)
        results[rank] = (line, jobs[0].steps_run, jobs[0].units)
    finally:
        dist.destroy_process_group()


def test_bench_contract_with_two_ranks():
    """bench.py's timed contract (warm-up, barrier, exactly K steps, max over ranks, units of all ranks, gather after
    the steps) driven by two gloo ranks with ragged shards: rank 0 alone reports, `value` counts every rank's units."""
    import torch.multiprocessing as mp

    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_bench_worker, args=(2, _free_port(), results), nprocs=2, join=True)
    line0, steps0, units0 = results[0]
    line1, steps1, units1 = results[1]
    assert line1 is None and line0 is not None
    assert (units0, units1) == (8, 6)
    # (warm-up + timed steps) twice -- straight after set-up and again after the pre-roll (none on CPU) -- then the steps
    # repeated with the gather
    assert steps0 == steps1 == 2 * (2 + 3) + 3
    assert line0["value_without_pre_roll"] > 0 and line0["ms_per_step_without_pre_roll"] > 0
    assert line0["n_gpus"] == 2 and line0["steps"] == 3 and line0["warmup"] == 2 and line0["scaling"] == "weak"
    assert abs(line0["value"] * line0["ms_per_step"] * 1e-3 - (units0 + units1)) < 1e-6  # (8 + 6) units per step
    assert line0["extras"]["all_gather_ms"] > 0 and line0["extras"]["value_with_all_gather"] > 0
    assert line0["higher_is_better"] is True and line0["config"]["world"] == 2


def _c5_flow_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import argparse
    import importlib.util

    import torch
    import torch.distributed as dist

    import oracle
    import oracle.pir
    from heamd import sharding

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        degree, dims = 64, [4, 5]
        t = oracle.generate_primes([17], True, degree)[0]
        q = oracle.generate_primes([40, 40, 41], False, degree)
        bfv = oracle.BfvContext(degree, t, q)
        moduli = bfv.ciphertext_context().moduli
        rng = np.random.default_rng(77)  # the same stream on both ranks: query, key and database are replicated inputs

        def uniform(prefix, row_moduli):
            return np.ascontiguousarray(np.stack(
                [rng.integers(0, m, size=tuple(prefix) + (degree,), dtype=np.uint64) for m in row_moduli], axis=len(prefix)))

        dim0 = uniform((dims[0], 2), moduli)
        rest = uniform((dims[1], 2), moduli)
        database = uniform((dims[1], dims[0]), moduli)
        key = uniform((bfv.L, 2), q)
        begin, end = sharding.shard_bounds(dims[1], world, rank)

        class PirJob:  # PirDim0Workload's interface on CPU tensors, the oracle standing in for the kernels
            def __init__(self):
                self.units = (end - begin) * dims[0]
                self.response = None
                self.consumed = 0

            def step(self):
                self.out = oracle.pir.dim0_columns(bfv, dim0, database[begin:end])

            def result(self):
                return torch.from_numpy(self.out.view(np.int64)), dims[1]

            def consume(self, gathered):
                self.consumed += 1
                self.response = oracle.pir.remaining_dimensions(bfv, dims, gathered.numpy().view(np.uint64), rest, key)

            def describe(self, w):
                return {"metric": "m", "unit": "ct-pt-mac/s", "dtype": "u64", "config": {"workload": "c5 dry run"}}

            def roofline(self, steps, r):
                return {"bound": "hbm", "frac": 0.0}, {}

        jobs = []

        def make_job():
            jobs.append(PirJob())
            return jobs[-1]

        args = argparse.Namespace(steps=2, warmup=1, skip_gather=False)
        line = bench.run_benchmark(args, make_job, rank, world, device="cpu", dist=dist)
        whole = oracle.pir.compute_response_for_one_chunk(bfv, dims, dim0, rest, database.reshape(-1, bfv.L, degree), None, key)
        results[rank] = (line, jobs[0].consumed, bool(np.array_equal(jobs[0].response, whole)), jobs[0].units)
    finally:
        dist.destroy_process_group()


def test_bench_c5_flow_with_two_ranks():
    """The whole of BASELINE configs[4] under bench.py's contract with two gloo ranks: each step computes the rank's
    columns (dim 0), the gather phase all-gathers them (ragged 3 + 2) and hands the set to the job's `consume`, which runs
    the remaining dimension -- and the response equals the unsharded chunk response word for word on both ranks.  The JSON
    line records what each GPU contributes to the all-gather."""
    import torch.multiprocessing as mp

    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_c5_flow_worker, args=(2, _free_port(), results), nprocs=2, join=True)
    line0, consumed0, equal0, units0 = results[0]
    line1, consumed1, equal1, units1 = results[1]
    assert line1 is None and line0 is not None
    assert consumed0 == consumed1 == 2 and equal0 and equal1
    assert (units0, units1) == (12, 8)
    assert line0["extras"]["all_gather_bytes_per_gpu"] == 3 * 2 * 2 * 64 * 8  # rank 0: 3 columns x [2][L=2][N=64] words
    assert line0["extras"]["value_with_all_gather"] > 0


def test_shard_bounds_partition():
    from heamd import sharding

    for total in (0, 1, 5, 4096, 4097):
        for world in (1, 2, 3, 8):
            edges = [sharding.shard_bounds(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_sizes(total, world)
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, 2)


def test_c_abi_shard_bounds_is_the_same_partition():
    """he_shard_bounds -- what a device group splits a call's units by (include/he_amd.h "Device groups") -- is the
    partition the one-process-per-GPU harness uses; its argument errors are HE_ERR_INVALID_ARGUMENT.  No GPU involved."""
    import heamd
    from heamd import sharding

    for total in (0, 1, 5, 128, 4097):
        for members in (1, 2, 3, 8):
            for member in range(members):
                assert heamd.shard_bounds(total, members, member) == sharding.shard_bounds(total, members, member)
    for bad in ((4, 0, 0), (4, 2, 2)):
        with pytest.raises(heamd.HeError) as err:
            heamd.shard_bounds(*bad)
        assert err.value.name == "invalidArgument"
