"""CPU oracle (TEST INFRASTRUCTURE): the PIR server's per-chunk response, composed from the pinned C oracle.

Restates PirUtilProtocol.computeResponseForOneChunk
(/root/reference Sources/PrivateInformationRetrieval/IndexPir/PirUtil.swift:408-486; the MulPirServer twin at
IndexPir/MulPir.swift:369-410 is the same algorithm without the task fan-out):

  1. per database column c: Bfv.innerProduct(ciphertexts: dim-0 query, plaintexts: column c) in Eval form, then
     convertToCanonicalFormat (Coeff for BFV)                                          PirUtil.swift:428-446
  2. per remaining dimension of size d: results <- [relinearize(innerProduct(query[cur..<cur+d],
     results[j d ..< (j+1) d])) for j]                                                PirUtil.swift:448-479
  3. modSwitchDownToSingle, Coeff format                                               PirUtil.swift:481-485
"""
import numpy as np


def compute_response_for_one_chunk(bfv, dimensions, dim0_query_eval, remaining_query, database, present=None,
                                   relinearization_key=None):
    """bfv: oracle.BfvContext.  dim0_query_eval [d0][2][L][N] Eval; remaining_query [d1+d2+..][2][L][N] Coeff;
    database [prod(dimensions)][L][N] Eval, plaintext k of column c at index c*d0 + k; present: [prod] 0/1 or None.
    Returns the response ciphertext [2][1][N] (Coeff, single modulus)."""
    dimensions = [int(d) for d in dimensions]
    d0 = dimensions[0]
    per_chunk = int(np.prod(dimensions))
    columns = per_chunk // d0
    L, n = bfv.L, bfv.degree
    database = np.asarray(database, dtype=np.uint64).reshape(per_chunk, L, n)
    dim0_query_eval = np.asarray(dim0_query_eval, dtype=np.uint64).reshape(d0, 2, L, n)
    remaining_count = 0 if remaining_query is None else np.asarray(remaining_query).size // (2 * L * n)
    assert columns == 1 or columns == remaining_count  # precondition, PirUtil.swift:422
    qctx = bfv.ciphertext_context()
    results = []
    for c in range(columns):
        mask = None if present is None else np.asarray(present, dtype=np.uint8)[c * d0:(c + 1) * d0]
        product = bfv.inner_product_plain(dim0_query_eval, database[c * d0:(c + 1) * d0], present=mask, poly_count=2)
        results.append(qctx.inverse_ntt(product[None])[0])
    cursor = 0
    for d in dimensions[1:]:
        query = np.asarray(remaining_query, dtype=np.uint64).reshape(-1, 2, L, n)[cursor:cursor + d]
        next_results = []
        for start in range(0, len(results), d):
            product = bfv.inner_product(query, np.stack(results[start:start + d]))
            next_results.append(bfv.relinearize(product[None], relinearization_key)[0])
        results = next_results
        cursor += d
    assert len(results) == 1
    ct = results[0][None]
    for level in range(L, 1, -1):  # Bfv.modSwitchDownToSingle: modSwitchDown until one modulus is left
        ct = bfv.mod_switch_down(ct, poly_count=2, moduli_count=level)
    return ct[0]


# ---------------------------------------------------------------------------------------------------------------------
# Query expansion (PirUtil.swift:196-355), restated with the same recursion and output order.


def dim0_columns(bfv, dim0_query_eval, database_columns, present=None):
    """PirUtil.swift:428-446 for a range of columns: database_columns [columns][d0][L][N] Eval -> [columns][2][L][N]
    Coeff.  What one member of a column-sharded deployment computes (he_pir_dim0_columns_device)."""
    qctx = bfv.ciphertext_context()
    out = []
    for c, column in enumerate(np.asarray(database_columns, dtype=np.uint64)):
        mask = None if present is None else np.asarray(present, dtype=np.uint8)[c]
        product = bfv.inner_product_plain(dim0_query_eval, column, present=mask, poly_count=2)
        out.append(qctx.inverse_ntt(product[None])[0])
    return np.stack(out)


def remaining_dimensions(bfv, dimensions, intermediate, remaining_query, relinearization_key=None):
    """PirUtil.swift:448-485 on all columns' dim-0 results (he_pir_remaining_dimensions_device)."""
    L, n = bfv.L, bfv.degree
    results = list(np.asarray(intermediate, dtype=np.uint64))
    cursor = 0
    for d in [int(x) for x in dimensions[1:]]:
        query = np.asarray(remaining_query, dtype=np.uint64).reshape(-1, 2, L, n)[cursor:cursor + d]
        results = [bfv.relinearize(bfv.inner_product(query, np.stack(results[start:start + d]))[None],
                                   relinearization_key)[0] for start in range(0, len(results), d)]
        cursor += d
    assert len(results) == 1
    ct = results[0][None]
    for level in range(L, 1, -1):
        ct = bfv.mod_switch_down(ct, poly_count=2, moduli_count=level)
    return ct[0]


def _log2(x):
    return x.bit_length() - 1


def _ceil_log2(x):
    return _log2(x) + (0 if x & (x - 1) == 0 else 1)


def expand_ciphertext_for_one_step(bfv, ct, log_step, galois_keys):
    """PirUtil.expandCiphertextForOneStep (PirUtil.swift:204-236). ct [2][L][N] Coeff; galois_keys {element: key}."""
    degree = bfv.degree
    qctx = bfv.ciphertext_context()
    shifting_power = 1 << (log_step - 1)
    target = (1 << (_log2(degree) - log_step + 1)) + 1
    usable = [e for e in galois_keys if e <= target]
    if not usable:
        raise KeyError("missingGaloisKey")
    element = max(usable)
    count = 1 << (_log2(target - 1) - _log2(element - 1))
    c1, current = ct, 1
    for _ in range(count):
        c1 = bfv.apply_galois(c1[None], element, galois_keys[element])[0]
        current = (current * element) % (2 * degree)
    assert current == target
    difference = qctx.sub(ct, c1)
    difference = qctx.multiply_power_of_x(difference, -shifting_power)
    return qctx.add(c1, ct), difference


def expand_ciphertext(bfv, ct, output_count, log_step, expected_height, galois_keys):
    """PirUtil.expandCiphertext (PirUtil.swift:249-300)."""
    qctx = bfv.ciphertext_context()
    if output_count == 1:
        return [ct] if log_step > expected_height else [qctx.add(ct, ct)]
    second = output_count >> 1
    first = output_count - second
    p0, p1 = expand_ciphertext_for_one_step(bfv, ct, log_step, galois_keys)
    first_half = expand_ciphertext(bfv, p0, first, log_step + 1, expected_height, galois_keys)
    second_half = expand_ciphertext(bfv, p1, second, log_step + 1, expected_height, galois_keys)
    out = []
    for a, b in zip(first_half[:second], second_half):
        out += [a, b]
    return out + first_half[len(first_half) - (first - second):] if first > second else out


def expand(bfv, ciphertexts, output_count, galois_keys):
    """PirUtil.expand (PirUtil.swift:313-355): ciphertexts [count][2][L][N] Coeff -> [output_count][2][L][N]."""
    ciphertexts = np.asarray(ciphertexts, dtype=np.uint64)
    degree = bfv.degree
    assert (len(ciphertexts) - 1) * degree < output_count <= len(ciphertexts) * degree
    remaining, out = output_count, []
    for ct in ciphertexts:
        to_generate = min(remaining, degree)
        remaining -= to_generate
        out += expand_ciphertext(bfv, ct, to_generate, 1, _ceil_log2(to_generate), galois_keys)
    return np.stack(out)
