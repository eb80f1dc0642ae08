"""Device-resident timings of every BASELINE.json config on one GPU (HIP events on the launch stream).

    python bench_tools/path_bench.py [--quick]

Prints one JSON object; bench.py embeds the same numbers under "extras".  Algorithmic bytes per unit follow
SURVEY.md section 8(d).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))


class Timing(float):
    """Seconds per call (the mean over the back-to-back timed calls) that also carries the spread of the single calls."""

    spread = None  # {"calls", "min_ms", "median_ms", "max_ms"}: device time between the events either side of each call
    per_call_ms = None
    host_enqueue_ms = None  # host time each call took to return (a blocking allocation shows here)


def _timed(torch, fn, reps, warmup=2, settle_s=0.03):
    """Seconds per call.  The first calls after another kernel mix run at whatever clocks that mix left behind (N=4096
    NTTs measured 18 % slow over 12 launches): warm up for at least `settle_s` of GPU time, and time at least as long.
    The calls are enqueued back to back with one event between each pair: the result is the mean, its `.spread` the
    minimum / median / maximum of the single calls -- a one-off stall inside the timed region shows as max >> median
    instead of hiding in the mean."""
    import time

    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    begin, extra = time.perf_counter(), 0
    while time.perf_counter() - begin < settle_s:
        fn()
        torch.cuda.synchronize()
        extra += 1
    per_call = (time.perf_counter() - begin) / max(extra, 1)
    reps = max(reps, int(2 * settle_s / max(per_call, 1e-6)))
    events = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    host = []
    torch.cuda.synchronize()
    events[0].record()
    for i in range(reps):
        t0 = time.perf_counter()
        fn()
        host.append((time.perf_counter() - t0) * 1e3)
        events[i + 1].record()
    events[-1].synchronize()
    calls = [events[i].elapsed_time(events[i + 1]) for i in range(reps)]
    ordered = sorted(calls)
    result = Timing(events[0].elapsed_time(events[-1]) * 1e-3 / reps)
    result.per_call_ms, result.host_enqueue_ms = calls, host
    result.spread = {"calls": reps, "min_ms": ordered[0], "median_ms": ordered[reps // 2], "max_ms": ordered[-1]}
    return result


def _uniform(torch, moduli, prefix, degree, seed):
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(*([1] * len(prefix)), len(moduli), 1)
    x = torch.randint(0, 1 << 62, tuple(prefix) + (len(moduli), degree), dtype=torch.int64, device="cuda", generator=gen)
    return x % bound


def _profiled(key):
    """HBM bytes per unit from the committed rocprofv3 counter passes (bench.py TRAFFIC_PROFILE), or None -- bench.py
    profiled_traffic's rules: not when the entry claims fewer bytes than the algorithm must move, nor when it was counted
    on kernels the current library no longer contains."""
    import bench

    entry = bench.profiled_traffic(key)
    if entry and entry.get("hbm_bytes_per_unit", 0) < entry.get("algorithmic_bytes_per_unit", 0):
        return None
    return entry


def config1_ntt(torch, heamd, batch=8192, reps=10, degree=4096, moduli_count=2):
    """Forward / inverse NTT at BASELINE configs[0]'s shape on the GPU: N=4096, 2 moduli (55-bit), `batch` polynomials
    (and, with other arguments, at the other ring sizes)."""
    moduli = heamd.generate_primes([55] * moduli_count, False, degree)
    ctx = heamd.PolyContext(degree, moduli)
    x = _uniform(torch, moduli, (batch,), degree, 11)
    forward = _timed(torch, lambda: ctx.forward_ntt_(x), reps)
    inverse = _timed(torch, lambda: ctx.inverse_ntt_(x), reps)
    bytes_per_poly = 2 * moduli_count * degree * 8
    return {"batch": batch, "spread_ms": {"forward": forward.spread, "inverse": inverse.spread},
            "forward_poly_ntt_per_s": batch / forward, "inverse_poly_ntt_per_s": batch / inverse,
            "forward_GBps": bytes_per_poly * batch / forward / 1e9, "inverse_GBps": bytes_per_poly * batch / inverse / 1e9,
            "forward_frac_of_8TBps": bytes_per_poly * batch / forward / 8e12,
            "inverse_frac_of_8TBps": bytes_per_poly * batch / inverse / 8e12}


def config3_ct_mul(torch, heamd, batch=1024, reps=5):
    """ct x ct + relinearize, N=8192, 4 ciphertext moduli + 1 key-switching modulus (BASELINE configs[2])."""
    degree = 8192
    q = heamd.generate_primes([55] * 5, False, degree)
    ctx = heamd.BfvContext(degree, 557057, q)
    moduli = q[:-1]
    lhs = _uniform(torch, moduli, (batch, 2), degree, 1)
    rhs = _uniform(torch, moduli, (batch, 2), degree, 2)
    key = _uniform(torch, q, (ctx.L, 2), degree, 3)
    ws_mul = torch.empty(ctx.mul_workspace_bytes(batch) // 8, dtype=torch.int64, device="cuda")
    ws_relin = torch.empty(ctx.relinearize_workspace_bytes(batch) // 8, dtype=torch.int64, device="cuda")
    state = {}

    def mul():
        state["ct3"] = ctx.mul(lhs, rhs, workspace=ws_mul)

    def relin():
        state["ct2"] = ctx.relinearize(state["ct3"], key, workspace=ws_relin)

    def both():
        mul()
        relin()

    t_mul = _timed(torch, mul, reps)
    t_relin = _timed(torch, relin, reps)
    t_both = _timed(torch, both, reps)
    compulsory = 1_572_864  # read 2 cts x 2 polys + write 2 polys
    import bench

    valu = bench.valu_roofline(batch / t_both) or {}
    return {
        "batch": batch,
        "spread_ms": {"ct_mul": t_mul.spread, "relinearize": t_relin.spread, "ct_mul_relinearize": t_both.spread},
        "ct_mul_per_s": batch / t_mul,
        "relinearize_per_s": batch / t_relin,
        "ct_mul_relinearize_per_s": batch / t_both,
        "compulsory_GBps": compulsory * batch / t_both / 1e9,
        "frac_of_8TBps_at_compulsory_bytes": compulsory * batch / t_both / 8e12,
        # the binding roofline (bench.py valu_roofline): the pipeline's 64-bit integer instruction stream against the measured
        # v_mad_u64_u32 rate, and its whole VALU stream against the time the part needs to issue it
        "valu_frac": valu.get("frac"),
        "valu_issue_frac": valu.get("issue_frac"),
        "valu": valu or None,
        # counter-measured HBM bytes per product (rocprofv3 --pmc passes, profiles/) at this run's rate
        "measured_bytes_per_product": (_profiled("c3_ct_mul_relinearize") or {}).get("hbm_bytes_per_unit"),
        "traffic_GBps": ((_profiled("c3_ct_mul_relinearize") or {}).get("hbm_bytes_per_unit", 0) * batch / t_both / 1e9
                         or None),
    }


def config3_small_batches(torch, heamd, batches=(1, 8, 64), reps=20):
    """The same pipeline on a few ciphertexts -- what a caller's single mulAssign / relinearize is: chains of small launches
    whose cost is latency (microseconds per call, device time)."""
    degree = 8192
    q = heamd.generate_primes([55] * 5, False, degree)
    ctx = heamd.BfvContext(degree, 557057, q)
    moduli = q[:-1]
    key = _uniform(torch, q, (ctx.L, 2), degree, 3)
    out = {}
    for batch in batches:
        lhs, rhs = _uniform(torch, moduli, (batch, 2), degree, 1), _uniform(torch, moduli, (batch, 2), degree, 2)
        state = {}

        def mul():
            state["ct3"] = ctx.mul(lhs, rhs)

        def relin():
            state["ct2"] = ctx.relinearize(state["ct3"], key)

        t_mul, t_relin = _timed(torch, mul, reps), _timed(torch, relin, reps)
        out["batch_%d" % batch] = {"ct_mul_us": t_mul.spread["median_ms"] * 1e3, "relinearize_us": t_relin.spread["median_ms"] * 1e3}
    return out


def config4_mod_switch(torch, heamd, batch=8192, reps=5):
    """divideAndRoundQLast, N=16384, 6 -> 5 moduli (BASELINE configs[3])."""
    degree = 16384
    moduli = heamd.generate_primes([55] * 6, False, degree)
    ctx = heamd.PolyContext(degree, moduli)
    x = _uniform(torch, moduli, (batch,), degree, 4)
    t = _timed(torch, lambda: ctx.divide_and_round_q_last(x), reps)
    bytes_per_poly = (6 + 5) * degree * 8
    measured = (_profiled("c4_mod_switch") or {}).get("hbm_bytes_per_unit")
    return {"batch": batch, "spread_ms": t.spread, "poly_per_s": batch / t, "GBps": bytes_per_poly * batch / t / 1e9,
            "frac_of_8TBps": bytes_per_poly * batch / t / 8e12,
            "traffic_GBps": measured * batch / t / 1e9 if measured else None}


def config5_inner_product(torch, heamd, count=256, columns=64, reps=3, queries=1, masked=False, packed=False):
    """PIR dim-0 shape on one GPU: `columns` outputs, each sum of `count` ct x pt products, N=8192, L=4.  queries > 1:
    that many queries' ciphertext vectors side by side share the database stream (poly_count = 2 queries)."""
    degree = 8192
    q = heamd.generate_primes([55] * 5, False, degree)
    ctx = heamd.BfvContext(degree, 557057, q)
    moduli = q[:-1]
    cts = _uniform(torch, moduli, (count, 2 * queries), degree, 5)
    pts = _uniform(torch, moduli, (columns, count), degree, 6)
    if packed:  # the database without the zero top bits of its words (55 of 64 bits)
        database = ctx.pack_plaintexts(pts)
        del pts
        t = _timed(torch, lambda: ctx.inner_product_plain_packed(cts, database, None, 2, columns), reps)
    elif masked:  # a device-resident nil-plaintext mask (the last rows of a real database are padding): 1 in 64 nil
        present = (torch.arange(columns * count, device="cuda") % 64 != 63).to(torch.uint8)
        t = _timed(torch, lambda: ctx.inner_product_plain_resident(cts, pts, present, 2 * queries, columns), reps)
    else:
        t = _timed(torch, lambda: ctx.inner_product_plain(cts, pts, None, 2 * queries, columns), reps)
    macs = count * columns
    db_bytes = macs * 4 * degree * 8
    measured = (_profiled("c5_inner_product_plain") or {}).get("hbm_bytes_per_unit") if queries == 1 else None
    return {"count": count, "columns": columns, "queries": queries, "masked": masked, "packed": packed,
            "spread_ms": t.spread,
            "ct_pt_mac_per_s": queries * macs / t,
            "database_GBps": db_bytes / t / 1e9, "frac_of_8TBps": db_bytes / t / 8e12,
            "traffic_GBps": measured * macs / t / 1e9 if measured else None}


def config5_pir_chunk(torch, heamd, d0=256, d1=64, reps=3):
    """One PIR chunk response end to end on the device (PirUtil.computeResponseForOneChunk): d0 x d1 database of Eval
    plaintexts, dim-0 ct x pt inner products, one ct x ct inner product + relinearize, mod-switch to one modulus."""
    degree = 8192
    q = heamd.generate_primes([55] * 5, False, degree)
    ctx = heamd.BfvContext(degree, 557057, q)
    moduli = q[:-1]
    dim0 = _uniform(torch, moduli, (d0, 2), degree, 7)
    rest = _uniform(torch, moduli, (d1, 2), degree, 8)
    database = _uniform(torch, moduli, (d0 * d1,), degree, 9)
    key = _uniform(torch, q, (ctx.L, 2), degree, 10)
    t = _timed(torch, lambda: ctx.pir_compute_response_chunk([d0, d1], dim0, rest, database, None, key), reps)
    db_bytes = d0 * d1 * 4 * degree * 8
    return {"dimensions": [d0, d1], "database_GB": db_bytes / 1e9, "chunk_response_ms": t * 1e3, "spread_ms": t.spread,
            "chunk_responses_per_s": 1 / t, "database_GBps": db_bytes / t / 1e9, "frac_of_8TBps": db_bytes / t / 8e12}


def config5_pir_chunk_loop(torch, heamd, d0=256, d1=64, chunks=8, reps=3):
    """PirUtil.computeResponse's chunk loop for one query (he_pir_compute_response_device): `chunks` chunks of a d0 x d1
    database answered in one call: dim-0 of all chunks in one launch, the remaining dimension batched over chunks."""
    degree = 8192
    q = heamd.generate_primes([55] * 5, False, degree)
    ctx = heamd.BfvContext(degree, 557057, q)
    moduli = q[:-1]
    dim0 = _uniform(torch, moduli, (d0, 2), degree, 7)
    rest = _uniform(torch, moduli, (d1, 2), degree, 8)
    database = _uniform(torch, moduli, (chunks, d0 * d1), degree, 9)
    key = _uniform(torch, q, (ctx.L, 2), degree, 10)
    t = _timed(torch, lambda: ctx.pir_compute_response([d0, d1], dim0, rest, database, chunks, relinearization_key=key), reps)
    db_bytes = chunks * d0 * d1 * 4 * degree * 8
    return {"dimensions": [d0, d1], "chunks": chunks, "database_GB": db_bytes / 1e9, "ms_per_chunk": t / chunks * 1e3,
            "spread_ms": t.spread,
            "chunk_responses_per_s": chunks / t, "database_GBps": db_bytes / t / 1e9, "frac_of_8TBps": db_bytes / t / 8e12}


def config5_pir_queries(torch, heamd, d0=256, d1=64, chunks=8, queries=4, reps=3):
    """`queries` queries over the same `chunks` x (d0 x d1) database in one call (he_pir_compute_response_queries_device):
    chunk responses per second over all queries, and the database rate one query's share of the call amounts to."""
    degree = 8192
    q = heamd.generate_primes([55] * 5, False, degree)
    ctx = heamd.BfvContext(degree, 557057, q)
    moduli = q[:-1]
    dim0 = _uniform(torch, moduli, (d0, queries, 2), degree, 7)
    rest = _uniform(torch, moduli, (queries, d1, 2), degree, 8)
    database = _uniform(torch, moduli, (chunks, d0 * d1), degree, 9)
    keys = [_uniform(torch, q, (ctx.L, 2), degree, 10 + i) for i in range(queries)]
    t = _timed(torch, lambda: ctx.pir_compute_response_queries([d0, d1], dim0, rest, database, chunks, keys), reps)
    db_bytes = chunks * d0 * d1 * 4 * degree * 8
    return {"dimensions": [d0, d1], "chunks": chunks, "queries": queries, "spread_ms": t.spread,
            "ms_per_chunk_per_query": t / chunks / queries * 1e3,
            "chunk_responses_per_s": chunks * queries / t, "database_GBps_per_query_share": db_bytes * queries / t / 1e9}


def config5_pir_whole_query(torch, heamd, d0=256, d1=64, chunks=8, indices=1, reps=3):
    """The whole server side of one Query (he_pir_compute_response_to_query_device = PirUtil.computeResponse with one
    database): expansion of one query ciphertext into (d0 + d1) x indices selection ciphertexts, dim-0 to Eval, every
    chunk answered.  Uniform words and keys (the arithmetic does not depend on the values)."""
    degree = 8192
    q = heamd.generate_primes([55] * 5, False, degree)
    ctx = heamd.BfvContext(degree, 557057, q)
    moduli = q[:-1]
    total = (d0 + d1) * indices
    query = _uniform(torch, moduli, (1, 2), degree, 7)
    elements = sorted({(degree >> level) + 1 for level in range(max((total - 1).bit_length(), 1))})
    galois = {e: _uniform(torch, q, (ctx.L, 2), degree, 20 + i) for i, e in enumerate(elements)}
    relin = _uniform(torch, q, (ctx.L, 2), degree, 10)
    database = _uniform(torch, moduli, (chunks, d0 * d1), degree, 9)
    t = _timed(torch, lambda: ctx.pir_compute_response_to_query([d0, d1], query, indices, galois, relin, database, chunks),
               max(reps, 30))
    return {"dimensions": [d0, d1], "chunks": chunks, "indices": indices, "ms_per_query": t * 1e3,
            "ms_per_query_min": t.spread["min_ms"], "ms_per_query_median": t.spread["median_ms"],
            "ms_per_query_max": t.spread["max_ms"], "calls": t.spread["calls"],
            "per_call_ms": [round(x, 3) for x in t.per_call_ms], "host_enqueue_ms": [round(x, 3) for x in t.host_enqueue_ms],
            "ms_per_index": t / indices * 1e3, "chunk_responses_per_s": chunks * indices / t}


def run_all(quick=False, only=None):
    """Every leg, in the order bench.py reports them; `only`: the names of the legs to run."""
    import torch

    import heamd

    heamd.set_scratch_cache()  # a server's setting: the library keeps its released scratch (he_set_scratch_cache)

    legs = [
        ("config1_ntt_n4096_l2", lambda: config1_ntt(torch, heamd, batch=1024 if quick else 8192)),
        ("ntt_n16384_l4", lambda: config1_ntt(torch, heamd, batch=256 if quick else 1024, degree=16384, moduli_count=4)),
        ("config3_ct_mul", lambda: config3_ct_mul(torch, heamd, batch=256 if quick else 1024)),
        ("config3_small_batches", lambda: config3_small_batches(torch, heamd)),
        ("config4_mod_switch", lambda: config4_mod_switch(torch, heamd, batch=1024 if quick else 8192)),
        # the per-GPU shard of BASELINE configs[4]: d0 = 1024 rows x d1 / 8 = 128 columns (34 GB of plaintexts)
        ("config5_inner_product_1gpu", lambda: config5_inner_product(torch, heamd, count=64 if quick else 1024,
                                                                     columns=16 if quick else 128)),
        ("config5_pir_chunk_response_1gpu", lambda: config5_pir_chunk(torch, heamd, d0=64 if quick else 256,
                                                                      d1=16 if quick else 64)),
        ("config5_pir_chunk_loop_1gpu", lambda: config5_pir_chunk_loop(torch, heamd, d0=64 if quick else 256,
                                                                       d1=16 if quick else 64, chunks=2 if quick else 8)),
        # four queries that share one pass over the same database (not a BASELINE line: the reference answers one at a time)
        ("config5_pir_4_queries_1gpu", lambda: config5_pir_queries(torch, heamd, d0=64 if quick else 256,
                                                                   d1=16 if quick else 64, chunks=2 if quick else 8, queries=4)),
        ("config5_pir_whole_query_1gpu", lambda: config5_pir_whole_query(torch, heamd, d0=64 if quick else 256,
                                                                         d1=16 if quick else 64, chunks=2 if quick else 8,
                                                                         indices=1)),
    ]
    return {name: leg() for name, leg in legs if only is None or name in only}


if __name__ == "__main__":
    chosen = [a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--only=")]
    print(json.dumps(run_all("--quick" in sys.argv, only=chosen[0] if chosen else None), indent=1))
