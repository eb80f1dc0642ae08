/* A plain-C caller of the C ABI (what a SwiftPM C target sees): context creation with the reference's validation,
 * device buffer management, forward + inverse NTT round trip, element-wise add, error reporting.
 * Built and run by tests/test_abi_c.py:  gcc abi_roundtrip.c -I include -L lib -lhe_amd */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "he_amd.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int status_ = (call);                                                                         \
        if (status_ != HE_OK) {                                                                       \
            fprintf(stderr, "%s -> %s (%s)\n", #call, he_status_string(status_), he_last_error_message()); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

int main(void) {
    const uint32_t degree = 4096;
    int bits[2] = {55, 55};
    uint64_t moduli[2];
    CHECK(he_generate_primes(bits, 2, 0, degree, moduli));
    printf("moduli %llu %llu\n", (unsigned long long)moduli[0], (unsigned long long)moduli[1]);

    /* validation mirrors PolyContext.init: a repeated modulus is coprimeModuli */
    he_poly_context* bad = NULL;
    uint64_t repeated[2] = {moduli[0], moduli[0]};
    if (he_poly_context_create(degree, repeated, 2, &bad) != HE_ERR_COPRIME_MODULI) {
        fprintf(stderr, "expected coprimeModuli\n");
        return 1;
    }

    he_poly_context* ctx = NULL;
    CHECK(he_poly_context_create(degree, moduli, 2, &ctx));
    const size_t batch = 3, words = batch * 2 * degree;
    uint64_t* host = (uint64_t*)malloc(words * sizeof(uint64_t));
    uint64_t* back = (uint64_t*)malloc(words * sizeof(uint64_t));
    uint64_t state = 88172645463325252ull;
    for (size_t i = 0; i < words; ++i) { /* xorshift, reduced into the row's modulus */
        state ^= state << 13;
        state ^= state >> 7;
        state ^= state << 17;
        host[i] = state % moduli[(i / degree) % 2];
    }
    void* device = NULL;
    CHECK(he_device_malloc(&device, words * sizeof(uint64_t)));
    CHECK(he_memcpy_h2d(device, host, words * sizeof(uint64_t), NULL));
    CHECK(he_ntt_forward_device(ctx, (uint64_t*)device, batch, NULL));
    CHECK(he_memcpy_d2h(back, device, words * sizeof(uint64_t), NULL));
    CHECK(he_stream_synchronize(NULL));
    if (memcmp(back, host, words * sizeof(uint64_t)) == 0) {
        fprintf(stderr, "forward transform left the data unchanged\n");
        return 1;
    }
    for (size_t i = 0; i < words; ++i)
        if (back[i] >= moduli[(i / degree) % 2]) {
            fprintf(stderr, "non-canonical word at %zu\n", i);
            return 1;
        }
    CHECK(he_ntt_inverse_device(ctx, (uint64_t*)device, batch, NULL));
    CHECK(he_memcpy_d2h(back, device, words * sizeof(uint64_t), NULL));
    CHECK(he_stream_synchronize(NULL));
    if (memcmp(back, host, words * sizeof(uint64_t)) != 0) {
        fprintf(stderr, "round trip mismatch\n");
        return 1;
    }
    /* x + x through the element-wise entry point, checked on the host */
    CHECK(he_poly_add_device(ctx, (uint64_t*)device, (const uint64_t*)device, batch, NULL));
    CHECK(he_memcpy_d2h(back, device, words * sizeof(uint64_t), NULL));
    CHECK(he_stream_synchronize(NULL));
    for (size_t i = 0; i < words; ++i) {
        const uint64_t q = moduli[(i / degree) % 2];
        if (back[i] != (host[i] + host[i]) % q) {
            fprintf(stderr, "add mismatch at %zu\n", i);
            return 1;
        }
    }
    CHECK(he_device_free(device));
    he_poly_context_destroy(ctx);
    free(host);
    free(back);
    printf("abi round trip ok (%s)\n", he_version());
    return 0;
}
