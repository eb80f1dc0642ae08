/* A plain-C caller of the scheme level of the C ABI (B3 / B4): what ctypes silently coerces -- struct layouts,
 * size_t / uint32_t / uint64_t argument widths, NULL workspaces -- is fixed here by a C compiler.
 *
 *   abi_scheme <fixture>
 *
 * The fixture (written by tests/test_abi_c.py from seeded inputs and the CPU oracle's outputs) is a stream of uint64
 * words:  degree, t, moduli_count, moduli[], batch, d0,
 *         lhs [batch][2][L][N], rhs [batch][2][L][N], key [L][2][L+1][N],
 *         expected ct x ct [batch][3][L][N], expected relinearized [batch][2][L][N],
 *         expected modSwitchDownToSingle of that [batch][2][1][N],
 *         cts [d0][2][L][N], pts [d0][L][N], mask words [d0] (0 = nil plaintext), expected inner product [2][L][N].
 * Every comparison is word for word. */
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

static uint64_t* words;
static size_t cursor;
static const uint64_t* take(size_t count) {
    const uint64_t* at = words + cursor;
    cursor += count;
    return at;
}
static int upload(uint64_t** device, const uint64_t* host, size_t count) {
    void* raw = NULL;
    int status = he_device_malloc(&raw, count * sizeof(uint64_t));
    if (status != HE_OK) return status;
    *device = (uint64_t*)raw;
    return host == NULL ? HE_OK : he_memcpy_h2d(raw, host, count * sizeof(uint64_t), NULL);
}
static int same(const char* what, const uint64_t* device, const uint64_t* expected, size_t count) {
    uint64_t* back = (uint64_t*)malloc(count * sizeof(uint64_t));
    if (he_memcpy_d2h(back, device, count * sizeof(uint64_t), NULL) != HE_OK || he_stream_synchronize(NULL) != HE_OK) {
        fprintf(stderr, "%s: copy failed (%s)\n", what, he_last_error_message());
        return 0;
    }
    for (size_t i = 0; i < count; ++i)
        if (back[i] != expected[i]) {
            fprintf(stderr, "%s: word %zu is %llu, the oracle says %llu\n", what, i, (unsigned long long)back[i],
                    (unsigned long long)expected[i]);
            free(back);
            return 0;
        }
    free(back);
    return 1;
}

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    FILE* file = fopen(argv[1], "rb");
    if (file == NULL) return 2;
    fseek(file, 0, SEEK_END);
    const long bytes = ftell(file);
    fseek(file, 0, SEEK_SET);
    words = (uint64_t*)malloc((size_t)bytes);
    if (fread(words, 1, (size_t)bytes, file) != (size_t)bytes) return 2;
    fclose(file);

    const uint32_t degree = (uint32_t)*take(1);
    const uint64_t t = *take(1);
    const uint32_t moduli_count = (uint32_t)*take(1);
    const uint64_t* moduli = take(moduli_count);
    const size_t batch = (size_t)*take(1), d0 = (size_t)*take(1);
    he_bfv_context* ctx = NULL;
    CHECK(he_bfv_context_create(degree, t, moduli, moduli_count, &ctx));
    const uint32_t L = he_bfv_ciphertext_moduli_count(ctx);
    if (L + 1 != moduli_count) {
        fprintf(stderr, "L = %u for %u moduli\n", L, moduli_count);
        return 1;
    }
    const size_t poly = (size_t)L * degree;

    /* ct x ct, relinearize (library scratch: NULL workspace), modSwitchDownToSingle */
    uint64_t *lhs, *rhs, *key, *product, *relinearized, *single;
    CHECK(upload(&lhs, take(batch * 2 * poly), batch * 2 * poly));
    CHECK(upload(&rhs, take(batch * 2 * poly), batch * 2 * poly));
    CHECK(upload(&key, take((size_t)L * 2 * (L + 1) * degree), (size_t)L * 2 * (L + 1) * degree));
    CHECK(upload(&product, NULL, batch * 3 * poly));
    CHECK(upload(&relinearized, NULL, batch * 2 * poly));
    CHECK(upload(&single, NULL, batch * 2 * degree));
    CHECK(he_bfv_mul_device(ctx, L, lhs, rhs, product, batch, NULL, 0, NULL));
    if (!same("ct x ct", product, take(batch * 3 * poly), batch * 3 * poly)) return 1;
    CHECK(he_bfv_relinearize_device(ctx, L, product, key, relinearized, batch, NULL, 0, NULL));
    if (!same("relinearize", relinearized, take(batch * 2 * poly), batch * 2 * poly)) return 1;
    if (he_bfv_relinearize_device(ctx, L, product, NULL, relinearized, batch, NULL, 0, NULL) !=
        HE_ERR_MISSING_RELINEARIZATION_KEY) {
        fprintf(stderr, "a NULL key must be missingRelinearizationKey\n");
        return 1;
    }
    CHECK(he_bfv_mod_switch_down_to_single_device(ctx, L, 2, relinearized, single, batch, NULL));
    if (!same("modSwitchDownToSingle", single, take(batch * 2 * degree), batch * 2 * degree)) return 1;
    /* the same products with a caller workspace of exactly the advertised size */
    const size_t workspace_bytes = he_bfv_mul_workspace_bytes(ctx, L, batch);
    void* workspace = NULL;
    CHECK(he_device_malloc(&workspace, workspace_bytes));
    uint64_t* again;
    CHECK(upload(&again, NULL, batch * 3 * poly));
    CHECK(he_bfv_mul_device(ctx, L, lhs, rhs, again, batch, workspace, workspace_bytes, NULL));
    if (!same("ct x ct with a workspace", again, words + (cursor - batch * 2 * degree - batch * 2 * poly - batch * 3 * poly),
              batch * 3 * poly))
        return 1;

    /* ct x pt inner product with a host mask (uint8 per plaintext) */
    uint64_t *cts, *pts, *sum;
    CHECK(upload(&cts, take(d0 * 2 * poly), d0 * 2 * poly));
    CHECK(upload(&pts, take(d0 * poly), d0 * poly));
    const uint64_t* mask_words = take(d0);
    uint8_t* present = (uint8_t*)malloc(d0);
    for (size_t i = 0; i < d0; ++i) present[i] = (uint8_t)(mask_words[i] != 0);
    CHECK(upload(&sum, NULL, 2 * poly));
    CHECK(he_bfv_inner_product_plain_device(ctx, L, 2, cts, pts, present, d0, 1, sum, NULL));
    if (!same("inner product with plaintexts", sum, take(2 * poly), 2 * poly)) return 1;

    CHECK(he_stream_synchronize(NULL));
    he_bfv_context_destroy(ctx);
    printf("abi scheme ok (%zu products, %zu-term inner product)\n", batch, d0);
    return 0;
}
