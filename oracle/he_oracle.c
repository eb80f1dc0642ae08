/*
 * oracle/he_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See he_oracle.h.
 *
 * Restates, in plain C with unsigned __int128, the algorithms of the reference's BFV PolyRq/NTT hot path.
 * Citations are file:line relative to /root/reference/Sources/ (MA = ModularArithmetic, HE = HomomorphicEncryption).
 * The transforms keep the reference's lazy-reduction schedule (Harvey butterflies) so that this file is also a
 * fair single-thread CPU baseline ("port") for bench.py.
 */
#include "he_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef orc_u128 u128;

#define ORC_MAX_MODULUS ((((uint64_t)1) << 62) - 1) /* MA/Modulus.swift:177-180 */
#define ORC_MTILDE (((uint64_t)1) << 32)             /* MA/Scalar.swift:523-525 */
#define ORC_GAMMA ((((uint64_t)1) << 62) - 40797)    /* MA/Scalar.swift:517-519 rnsCorrectionFactor */
#define ORC_MTILDE32 (((uint64_t)1) << 16)           /* MA/Scalar.swift:508-510 (UInt32) */
#define ORC_GAMMA32 ((((uint64_t)1) << 30) - 20405)  /* MA/Scalar.swift:502-506 (UInt32) */

/* ------------------------------------------------------------------------------------------------
 * Scalar helpers
 * ---------------------------------------------------------------------------------------------- */

static inline uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) >> 64); }

/* MA/Scalar.swift:162-167 */
static inline uint64_t subtract_if_exceeds(uint64_t x, uint64_t p) {
    uint64_t difference = x - p;
    uint64_t mask = (uint64_t)0 - (difference >> 63);
    return difference + (p & mask);
}
/* MA/Scalar.swift:146-152 */
static inline uint64_t add_mod(uint64_t x, uint64_t y, uint64_t p) { return subtract_if_exceeds(x + y, p); }
/* MA/Scalar.swift:188-193 */
static inline uint64_t sub_mod(uint64_t x, uint64_t y, uint64_t p) { return subtract_if_exceeds(x + p - y, p); }
/* MA/Scalar.swift:175-178 */
static inline uint64_t neg_mod(uint64_t x, uint64_t p) { return subtract_if_exceeds(p - x, p); }

static inline int is_power_of_two(uint64_t x) { return x != 0 && (x & (x - 1)) == 0; }
static inline int log2_floor(uint64_t x) { return 63 - __builtin_clzll(x); }
static inline int significant_bits(uint64_t x) { return x == 0 ? 0 : 64 - __builtin_clzll(x); }

static inline uint64_t mul_mod_slow(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)(((u128)a * b) % p); }

/* high 128 bits of a 128x128 product */
static u128 mulhi128(u128 a, u128 b) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
    return p11 + (p01 >> 64) + (p10 >> 64) + (mid >> 64);
}

/* MA/Scalar.swift:207-230 powMod (square-and-multiply; result is the mathematical value). */
uint64_t orc_pow_mod(uint64_t base, uint64_t exponent, uint64_t modulus) {
    if (modulus == 1) return 0;
    uint64_t result = 1 % modulus;
    base %= modulus;
    while (exponent) {
        if (exponent & 1) result = mul_mod_slow(result, base, modulus);
        base = mul_mod_slow(base, base, modulus);
        exponent >>= 1;
    }
    return result;
}

/* HE/Scalar.swift:162-202 isPrime: trial division by, then Miller-Rabin with, bases 2..37. */
int orc_is_prime(uint64_t n) {
    static const uint64_t bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n <= 1) return 0;
    for (int i = 0; i < 12; ++i) {
        if (n == bases[i]) return 1;
        if (n % bases[i] == 0) return 0;
    }
    int r = 63;
    while (r > 0 && ((n - 1) % (((uint64_t)1) << r)) != 0) --r;
    uint64_t d = (n - 1) >> r;
    for (int i = 0; i < 12; ++i) {
        uint64_t x = orc_pow_mod(bases[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int witness_passed = 0;
        for (int k = 0; k < r; ++k) {
            x = mul_mod_slow(x, x, n);
            if (x == n - 1) {
                witness_passed = 1;
                break;
            }
        }
        if (!witness_passed) return 0;
    }
    return 1;
}

/* HE/PolyRq/PolyRq+Ntt.swift:24-27 */
static int is_ntt_modulus(uint64_t p, uint64_t degree) {
    return is_power_of_two(degree) && (p % (2 * degree)) == 1 && p != 1;
}

/* HE/Scalar.swift:113-154 generatePrimes */
int orc_generate_primes(const int* significant_bit_counts, int count, int preferring_small, uint64_t ntt_degree,
                        int word_bits, uint64_t* out) {
    if (!is_power_of_two(ntt_degree)) return ORC_ERR_INVALID_ARGUMENT;
    int found = 0;
    for (int k = 0; k < count; ++k) {
        int bits = significant_bit_counts[k];
        if (bits > word_bits || bits < 1) return ORC_ERR_INVALID_ARGUMENT;
        u128 upper = (bits == word_bits) ? ((((u128)1) << word_bits) - 1) : (((u128)1) << bits);
        u128 lower = ((u128)1) << (bits - 1);
        u128 step = (u128)2 * ntt_degree;
        /* Swift traps on unsigned underflow of `range.upperBound - step`; treat as "no prime". */
        if (!preferring_small && upper < step) continue;
        u128 candidate = preferring_small ? lower + 1 : (upper - step) + 1;
        while (candidate >= lower && candidate < upper) {
            uint64_t c = (uint64_t)candidate;
            int duplicate = 0;
            for (int j = 0; j < found; ++j) duplicate |= (out[j] == c);
            if (!duplicate && orc_is_prime(c) && is_ntt_modulus(c, ntt_degree)) {
                out[found++] = c;
                break;
            }
            if (preferring_small) {
                candidate += step;
            } else {
                if (candidate < step) break;
                candidate -= step;
            }
        }
    }
    return found == count ? ORC_OK : ORC_ERR_NOT_ENOUGH_PRIMES;
}

/* HE/Scalar.swift:76-96 inverseMod: extended Euclid in Int64. */
int orc_inverse_mod(uint64_t value, uint64_t modulus, uint64_t* out) {
    if (value == 0 || modulus == 0) return ORC_ERR_NOT_INVERTIBLE;
    int64_t a = (int64_t)value, m = (int64_t)modulus, x0 = 0, inverse = 1;
    while (a > 1) {
        if (m == 0) return ORC_ERR_NOT_INVERTIBLE;
        inverse -= (a / m) * x0;
        a %= m;
        int64_t tmp = a;
        a = m;
        m = tmp;
        tmp = x0;
        x0 = inverse;
        inverse = tmp;
    }
    if (inverse < 0) inverse += (int64_t)modulus;
    *out = (uint64_t)inverse;
    return ORC_OK;
}

/* MA/Scalar.swift:238-254 */
uint32_t orc_reverse_bits(uint32_t x, int bit_count) {
    x = ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
    x = ((x & 0xCCCCCCCCu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xF0F0F0F0u) >> 4) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x & 0xFF00FF00u) >> 8) | ((x & 0x00FF00FFu) << 8);
    x = (x >> 16) | (x << 16);
    return bit_count >= 32 ? x : (bit_count <= 0 ? 0 : x >> (32 - bit_count));
}

/* HE/PolyRq/PolyRq+Ntt.swift:30-37 */
int orc_is_primitive_root_of_unity(uint64_t root, uint64_t degree, uint64_t modulus) {
    return orc_pow_mod(root, degree / 2, modulus) == modulus - 1;
}

/* HE/PolyRq/PolyRq+Ntt.swift:45-105.  The reference draws random candidates for *a* primitive root and then
 * takes the minimum over its odd powers; that set is exactly all primitive degree'th roots, so the result is the
 * smallest primitive root whatever candidate is found.  We search candidates 2,3,... deterministically. */
uint64_t orc_min_primitive_root_of_unity(uint64_t modulus, uint64_t degree) {
    if (!is_power_of_two(degree) || degree < 2) return 0;
    uint64_t lambda = modulus - 1;
    if (lambda % degree != 0) return 0;
    uint64_t generator = 0;
    for (uint64_t candidate = 2; candidate < modulus && candidate < 4096; ++candidate) {
        uint64_t root = orc_pow_mod(candidate, lambda / degree, modulus);
        if (orc_is_primitive_root_of_unity(root, degree, modulus)) {
            generator = root;
            break;
        }
    }
    if (!generator) return 0;
    uint64_t smallest = generator, current = generator;
    uint64_t squared = mul_mod_slow(generator, generator, modulus);
    for (uint64_t i = 0; i < degree / 2; ++i) {
        if (current < smallest) smallest = current;
        current = mul_mod_slow(current, squared, modulus);
    }
    return smallest;
}

/* ------------------------------------------------------------------------------------------------
 * Modulus<T> / ReduceModulus<T> / MultiplyConstantModulus<T>
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    uint64_t p;
    uint64_t single_factor;  /* floor(2^64 / p)                 MA/Modulus.swift:206-209 */
    u128 double_factor;      /* floor(2^128 / p)                MA/Modulus.swift:224-232 */
    uint64_t product_factor; /* floor(2^(bits(p)+62) / p)       MA/Modulus.swift:235-240 */
    int product_shift;       /* bits(p) - 2                     MA/Modulus.swift:351-354 */
} orc_modulus;

/* HE/Modulus.swift:24-45 */
static orc_modulus modulus_init(uint64_t p) {
    orc_modulus m;
    m.p = p;
    m.single_factor = (uint64_t)((((u128)1) << 64) / p); /* p == 1 truncates to 0 like `.low` */
    if (is_power_of_two(p)) {
        int lg = log2_floor(p);
        m.double_factor = lg == 0 ? 0 : (((u128)1) << (128 - lg));
    } else {
        m.double_factor = (~(u128)0) / p;
    }
    int n = significant_bits(p);
    m.product_factor = (uint64_t)((((u128)1) << (n + 62)) / p);
    m.product_shift = n - 2;
    return m;
}

/* MA/Modulus.swift:258-263 */
static inline uint64_t reduce_u64(const orc_modulus* m, uint64_t x) {
    uint64_t q_hat = mulhi64(x, m->single_factor);
    uint64_t z = x - q_hat * m->p;
    return subtract_if_exceeds(z, m->p);
}
/* MA/Modulus.swift:319-325 */
static inline uint64_t reduce_u128(const orc_modulus* m, u128 x) {
    u128 q_hat_high = mulhi128(x, m->double_factor);
    u128 q_p = q_hat_high * (u128)m->p;
    u128 z = x - q_p;
    return subtract_if_exceeds((uint64_t)z, m->p);
}
/* MA/Modulus.swift:349-360 */
static inline uint64_t reduce_product(const orc_modulus* m, u128 x) {
    u128 x_shift = m->product_shift >= 0 ? (x >> m->product_shift) : (x << (-m->product_shift));
    uint64_t q_hat = mulhi64((uint64_t)x_shift, m->product_factor);
    uint64_t z = (uint64_t)x - q_hat * m->p;
    return subtract_if_exceeds(z, m->p);
}
/* MA/Modulus.swift:89-94 */
static inline uint64_t multiply_mod(const orc_modulus* m, uint64_t x, uint64_t y) {
    return reduce_product(m, (u128)x * y);
}

typedef struct {
    uint64_t multiplicand;
    uint64_t factor; /* floor(multiplicand * 2^64 / p)  HE/Modulus.swift:92-103 (both the variable-time
                        `dividingFullWidth` and the constant-time `dividingFloor` forms compute this floor) */
    uint64_t p;
} orc_shoup;

static orc_shoup shoup_init(uint64_t multiplicand, uint64_t p) {
    orc_shoup s;
    s.multiplicand = multiplicand;
    s.p = p;
    s.factor = (uint64_t)((((u128)multiplicand) << 64) / p);
    return s;
}
/* MA/Modulus.swift:401-410 */
static inline uint64_t shoup_mul_lazy(const orc_shoup* s, uint64_t x) {
    uint64_t q = mulhi64(x, s->factor);
    return x * s->multiplicand - q * s->p;
}
/* MA/Modulus.swift:413-415 */
static inline uint64_t shoup_mul(const orc_shoup* s, uint64_t x) {
    return subtract_if_exceeds(shoup_mul_lazy(s, x), s->p);
}

uint64_t orc_barrett_reduce_u64(uint64_t modulus, uint64_t x) {
    orc_modulus m = modulus_init(modulus);
    return reduce_u64(&m, x);
}
uint64_t orc_barrett_reduce_u128(uint64_t modulus, uint64_t x_hi, uint64_t x_lo) {
    orc_modulus m = modulus_init(modulus);
    return reduce_u128(&m, (((u128)x_hi) << 64) | x_lo);
}
uint64_t orc_barrett_reduce_product(uint64_t modulus, uint64_t x, uint64_t y) {
    orc_modulus m = modulus_init(modulus);
    return multiply_mod(&m, x, y);
}
uint64_t orc_shoup_factor(uint64_t multiplicand, uint64_t modulus) { return shoup_init(multiplicand, modulus).factor; }
uint64_t orc_shoup_multiply_mod_lazy(uint64_t multiplicand, uint64_t modulus, uint64_t x) {
    orc_shoup s = shoup_init(multiplicand, modulus);
    return shoup_mul_lazy(&s, x);
}
uint64_t orc_shoup_multiply_mod(uint64_t multiplicand, uint64_t modulus, uint64_t x) {
    orc_shoup s = shoup_init(multiplicand, modulus);
    return shoup_mul(&s, x);
}

/* ------------------------------------------------------------------------------------------------
 * _NttContext
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    int valid;
    uint64_t degree;
    uint64_t p;
    uint64_t* root_powers;      /* [N] bit-reversed order       HE/PolyRq/PolyRq+Ntt.swift:125-143 */
    uint64_t* root_factors;     /* [N] Shoup factors */
    uint64_t* inv_root_powers;  /* [N] stage-major re-ordered   HE/PolyRq/PolyRq+Ntt.swift:146-157 */
    uint64_t* inv_root_factors; /* [N] */
    orc_shoup inverse_degree;            /* N^-1                HE/PolyRq/PolyRq+Ntt.swift:159-160 */
    orc_shoup inverse_degree_root;       /* N^-1 psi^(-N/2)     HE/PolyRq/PolyRq+Ntt.swift:162-168 */
} orc_ntt_context;

static void ntt_context_free(orc_ntt_context* c) {
    free(c->root_powers);
    free(c->root_factors);
    free(c->inv_root_powers);
    free(c->inv_root_factors);
    memset(c, 0, sizeof(*c));
}

/* HE/PolyRq/PolyRq+Ntt.swift:118-169 */
static int ntt_context_init(orc_ntt_context* c, uint64_t degree, uint64_t p) {
    memset(c, 0, sizeof(*c));
    uint64_t root = orc_min_primitive_root_of_unity(p, 2 * degree);
    if (!root) return ORC_ERR_INVALID_NTT_MODULUS;
    uint64_t inverse_root;
    int status = orc_inverse_mod(root, p, &inverse_root);
    if (status) return status;
    size_t n = (size_t)degree;
    int log2n = log2_floor(degree);
    c->degree = degree;
    c->p = p;
    c->root_powers = (uint64_t*)malloc(n * sizeof(uint64_t));
    c->root_factors = (uint64_t*)malloc(n * sizeof(uint64_t));
    c->inv_root_powers = (uint64_t*)malloc(n * sizeof(uint64_t));
    c->inv_root_factors = (uint64_t*)malloc(n * sizeof(uint64_t));
    uint64_t* inverse_powers = (uint64_t*)malloc(n * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) c->root_powers[i] = inverse_powers[i] = 1;
    size_t previous = 0;
    for (uint32_t idx = 1; idx < (uint32_t)n; ++idx) {
        size_t rev = orc_reverse_bits(idx, log2n);
        c->root_powers[rev] = mul_mod_slow(root, c->root_powers[previous], p);
        inverse_powers[rev] = mul_mod_slow(inverse_root, inverse_powers[previous], p);
        previous = rev;
    }
    size_t inverse_idx = 1;
    for (size_t i = 0; i < n; ++i) c->inv_root_powers[i] = 1;
    for (int lg = log2n - 1; lg >= 0; --lg) {
        size_t m = ((size_t)1) << lg;
        for (size_t i = 0; i < m; ++i) c->inv_root_powers[inverse_idx++] = inverse_powers[m + i];
    }
    free(inverse_powers);
    for (size_t i = 0; i < n; ++i) {
        c->root_factors[i] = shoup_init(c->root_powers[i], p).factor;
        c->inv_root_factors[i] = shoup_init(c->inv_root_powers[i], p).factor;
    }
    uint64_t inverse_degree;
    status = orc_inverse_mod(degree % p, p, &inverse_degree);
    if (status) {
        ntt_context_free(c);
        return status;
    }
    c->inverse_degree = shoup_init(inverse_degree, p);
    c->inverse_degree_root = shoup_init(mul_mod_slow(inverse_degree, c->inv_root_powers[n - 1], p), p);
    c->valid = 1;
    return ORC_OK;
}

/* HE/PolyRq/PolyRq+Ntt.swift:182-201 forwardButterfly */
#define FWD_BUTTERFLY(X, Y, W, F)                         \
    do {                                                  \
        uint64_t q__ = mulhi64((Y), (F));                 \
        uint64_t t__ = (Y) * (W)-q__ * p;                 \
        (Y) = (X) + two_p - t__;                          \
        (X) = (X) + t__;                                  \
    } while (0)

/* HE/PolyRq/PolyRq+Ntt.swift:237-319 forwardNtt(dataPtr:...) on one residue row, in place. */
static void forward_ntt_row(const orc_ntt_context* c, const orc_modulus* mod, uint64_t* data) {
    const uint64_t p = c->p, two_p = p << 1;
    const size_t n = (size_t)c->degree;
    const int log2n = log2_floor(c->degree);
    const uint64_t* W = c->root_powers;
    const uint64_t* F = c->root_factors;
    int64_t lazy = -1;
    const uint64_t max_lazy = UINT64_MAX / (2 * p) - 1;
    for (int log2m = 0; log2m < log2n; ++log2m) {
        size_t m = ((size_t)1) << log2m;
        size_t t = n >> (log2m + 1);
        lazy += 2;
        int time_to_reduce = (uint64_t)lazy > max_lazy;
        if (time_to_reduce) {
            if (t == 1) {
                lazy = (lazy - 2 > 2) ? lazy - 2 : 2;
            } else {
                lazy = 1;
            }
        }
        if (t == 1) {
            for (size_t i = 0; i < m; ++i) {
                uint64_t x = data[2 * i], y = data[2 * i + 1];
                if (time_to_reduce) x = subtract_if_exceeds(x, two_p);
                FWD_BUTTERFLY(x, y, W[m + i], F[m + i]);
                data[2 * i] = reduce_u64(mod, x);
                data[2 * i + 1] = reduce_u64(mod, y);
            }
        } else {
            for (size_t i = 0; i < m; ++i) {
                const uint64_t w = W[m + i], f = F[m + i];
                uint64_t* xs = data + 2 * i * t;
                uint64_t* ys = xs + t;
                if (time_to_reduce) {
                    for (size_t j = 0; j < t; ++j) {
                        uint64_t x = reduce_u64(mod, xs[j]), y = ys[j];
                        FWD_BUTTERFLY(x, y, w, f);
                        xs[j] = x;
                        ys[j] = y;
                    }
                } else {
                    for (size_t j = 0; j < t; ++j) {
                        uint64_t x = xs[j], y = ys[j];
                        FWD_BUTTERFLY(x, y, w, f);
                        xs[j] = x;
                        ys[j] = y;
                    }
                }
            }
        }
    }
}

/* HE/PolyRq/PolyRq+Ntt.swift:359-375 inverseButterfly */
#define INV_BUTTERFLY(X, Y, W, F)                         \
    do {                                                  \
        uint64_t t__ = (X) + k_p - (Y);                   \
        (X) = (X) + (Y);                                  \
        uint64_t q__ = mulhi64(t__, (F));                 \
        (Y) = t__ * (W)-q__ * p;                          \
    } while (0)

/* HE/PolyRq/PolyRq+Ntt.swift:379-483 inverseNtt(dataPtr:...) on one residue row, in place. */
static void inverse_ntt_row(const orc_ntt_context* c, const orc_modulus* mod, uint64_t* data) {
    const uint64_t p = c->p;
    const size_t n = (size_t)c->degree;
    const int log2n = log2_floor(c->degree);
    const uint64_t* W = c->inv_root_powers;
    const uint64_t* F = c->inv_root_factors;
    const int leading_zeros = __builtin_clzll(p);
    const int modulus_multiples_count = (log2n + 1 < leading_zeros) ? log2n + 1 : leading_zeros;
    size_t root_idx = 1;
    int lazy = -1;
    const size_t n_div2 = n >> 1;
    for (int log2m = log2n - 1; log2m >= 0; --log2m) {
        size_t m = ((size_t)1) << log2m;
        size_t t = n >> (log2m + 1);
        lazy += 1;
        int time_to_reduce = lazy == modulus_multiples_count;
        if (time_to_reduce) {
            if (m == 1) {
                lazy -= 1;
            } else {
                lazy = 0;
            }
        }
        const uint64_t k_p = p << lazy;
        if (m == 1) {
            for (size_t xi = 0; xi < n_div2; ++xi) {
                size_t yi = xi + n_div2;
                uint64_t x = data[xi], y = data[yi];
                if (time_to_reduce) {
                    x = subtract_if_exceeds(x, k_p);
                    y = subtract_if_exceeds(y, k_p);
                }
                uint64_t tx = x + y;
                uint64_t ty = x + k_p - y;
                data[xi] = shoup_mul(&c->inverse_degree, tx);
                data[yi] = shoup_mul(&c->inverse_degree_root, ty);
            }
        } else {
            for (size_t i = 0; i < m; ++i) {
                const uint64_t w = W[root_idx + i], f = F[root_idx + i];
                uint64_t* xs = data + 2 * i * t;
                uint64_t* ys = xs + t;
                for (size_t j = 0; j < t; ++j) {
                    uint64_t x = xs[j], y = ys[j];
                    if (time_to_reduce) {
                        x = reduce_u64(mod, x);
                        y = reduce_u64(mod, y);
                    }
                    INV_BUTTERFLY(x, y, w, f);
                    xs[j] = x;
                    ys[j] = y;
                }
            }
        }
        root_idx += m;
    }
}

/* ------------------------------------------------------------------------------------------------
 * PolyContext
 * ---------------------------------------------------------------------------------------------- */

struct orc_poly_context {
    uint64_t degree;
    size_t count;            /* L */
    uint64_t* moduli;        /* [L] */
    orc_modulus* reduce;     /* [L]  reduceModuli, HE/PolyRq/PolyContext.swift:96-99 */
    orc_ntt_context* ntt;    /* [L]  nttContext of the chain element whose last modulus is q_i, :112-122 */
    /* inverse_q_last[k][i] = q_{k-1}^-1 mod q_i, i < k-1: the `inverseQLast` of the chain element with k moduli,
     * HE/PolyRq/PolyContext.swift:108-111 */
    orc_shoup** inverse_q_last; /* [L+1] */
};

static int all_unique(const uint64_t* moduli, size_t count) {
    for (size_t i = 0; i < count; ++i)
        for (size_t j = i + 1; j < count; ++j)
            if (moduli[i] == moduli[j]) return 0;
    return 1;
}

/* HE/PolyRq/PolyContext.swift:49-62 validate(modulus:) */
static int validate_modulus(uint64_t modulus) {
    if (!(orc_is_prime(modulus) || is_power_of_two(modulus))) return ORC_ERR_INVALID_MODULUS;
    if (!(modulus >= 1 && modulus <= ORC_MAX_MODULUS)) return ORC_ERR_INVALID_MODULUS;
    return ORC_OK;
}

void orc_poly_context_destroy(orc_poly_context* ctx) {
    if (!ctx) return;
    if (ctx->ntt)
        for (size_t i = 0; i < ctx->count; ++i) ntt_context_free(&ctx->ntt[i]);
    if (ctx->inverse_q_last)
        for (size_t k = 0; k <= ctx->count; ++k) free(ctx->inverse_q_last[k]);
    free(ctx->inverse_q_last);
    free(ctx->ntt);
    free(ctx->reduce);
    free(ctx->moduli);
    free(ctx);
}

/* The designated initialiser (HE/PolyRq/PolyContext.swift:45-123) run on the prefix with `k` moduli; `has_next`
 * says whether a next context exists (then only the last modulus is validated, :74-76). */
static int poly_context_check_prefix(uint64_t degree, const uint64_t* moduli, size_t k, int has_next) {
    if (!is_power_of_two(degree)) return ORC_ERR_INVALID_DEGREE;
    size_t power_of_two_count = 0;
    for (size_t i = 0; i < k; ++i) power_of_two_count += is_power_of_two(moduli[i]);
    if (power_of_two_count > 1) return ORC_ERR_COPRIME_MODULI;
    if (!all_unique(moduli, k)) return ORC_ERR_COPRIME_MODULI;
    if (k == 0) return ORC_ERR_EMPTY_MODULUS;
    if (has_next) return validate_modulus(moduli[k - 1]);
    for (size_t i = 0; i < k; ++i) {
        int status = validate_modulus(moduli[i]);
        if (status) return status;
    }
    return ORC_OK;
}

/* HE/PolyRq/PolyContext.swift:131-141 init(degree:moduli:) -- builds the chain prefix by prefix. */
int orc_poly_context_create(uint64_t degree, const uint64_t* moduli, size_t moduli_count, orc_poly_context** out) {
    *out = NULL;
    if (moduli_count <= 1) {
        int status = poly_context_check_prefix(degree, moduli, moduli_count, 0);
        if (status) return status;
    } else {
        for (size_t k = 1; k <= moduli_count; ++k) {
            int status = poly_context_check_prefix(degree, moduli, k, k > 1);
            if (status) return status;
        }
    }
    orc_poly_context* ctx = (orc_poly_context*)calloc(1, sizeof(*ctx));
    ctx->degree = degree;
    ctx->count = moduli_count;
    ctx->moduli = (uint64_t*)malloc(moduli_count * sizeof(uint64_t));
    memcpy(ctx->moduli, moduli, moduli_count * sizeof(uint64_t));
    ctx->reduce = (orc_modulus*)calloc(moduli_count, sizeof(orc_modulus));
    ctx->ntt = (orc_ntt_context*)calloc(moduli_count, sizeof(orc_ntt_context));
    ctx->inverse_q_last = (orc_shoup**)calloc(moduli_count + 1, sizeof(orc_shoup*));
    for (size_t i = 0; i < moduli_count; ++i) ctx->reduce[i] = modulus_init(moduli[i]);
    for (size_t k = 1; k <= moduli_count; ++k) {
        uint64_t q_last = moduli[k - 1];
        ctx->inverse_q_last[k] = (orc_shoup*)calloc(k, sizeof(orc_shoup));
        for (size_t i = 0; i + 1 < k; ++i) {
            uint64_t inverse;
            int status = orc_inverse_mod(q_last % moduli[i], moduli[i], &inverse);
            if (status) {
                orc_poly_context_destroy(ctx);
                return status;
            }
            ctx->inverse_q_last[k][i] = shoup_init(inverse, moduli[i]);
        }
        if (!is_power_of_two(q_last) && is_ntt_modulus(q_last, degree)) {
            int status = ntt_context_init(&ctx->ntt[k - 1], degree, q_last);
            if (status) {
                orc_poly_context_destroy(ctx);
                return status;
            }
        }
    }
    *out = ctx;
    return ORC_OK;
}

uint64_t orc_poly_context_degree(const orc_poly_context* ctx) { return ctx->degree; }
size_t orc_poly_context_moduli_count(const orc_poly_context* ctx) { return ctx->count; }
void orc_poly_context_moduli(const orc_poly_context* ctx, uint64_t* out) {
    memcpy(out, ctx->moduli, ctx->count * sizeof(uint64_t));
}

/* HE/PolyRq/PolyContext.swift:246-253 */
uint64_t orc_poly_context_max_lazy_product_accumulation_count(const orc_poly_context* ctx, int word_bits) {
    uint64_t q_max = 0;
    for (size_t i = 0; i < ctx->count; ++i)
        if (ctx->moduli[i] > q_max) q_max = ctx->moduli[i];
    u128 max_product = (u128)(q_max - 1) * (q_max - 1);
    u128 double_width_max = word_bits == 32 ? (u128)UINT64_MAX : ~(u128)0;
    if (max_product == 0) return (uint64_t)INT64_MAX;
    u128 count = (double_width_max - q_max) / max_product;
    return count > (u128)INT64_MAX ? (uint64_t)INT64_MAX : (uint64_t)count;
}

/* HE/PolyRq/PolyContext.swift:184-191 qRemainder(dividingBy:) */
static uint64_t q_remainder_n(const uint64_t* moduli, size_t count, const orc_modulus* mod) {
    uint64_t prod = 1;
    for (size_t i = 0; i < count; ++i) prod = reduce_u128(mod, (u128)prod * moduli[i]);
    return prod;
}
uint64_t orc_poly_context_q_remainder(const orc_poly_context* ctx, uint64_t modulus) {
    orc_modulus m = modulus_init(modulus);
    return q_remainder_n(ctx->moduli, ctx->count, &m);
}

int orc_poly_context_ntt_tables(const orc_poly_context* ctx, size_t rns_index, uint64_t* root_powers,
                                uint64_t* root_factors, uint64_t* inv_root_powers, uint64_t* inv_root_factors,
                                uint64_t* inverse_degree, uint64_t* inverse_degree_root) {
    if (rns_index >= ctx->count || !ctx->ntt[rns_index].valid) return ORC_ERR_INVALID_NTT_MODULUS;
    const orc_ntt_context* c = &ctx->ntt[rns_index];
    size_t bytes = (size_t)ctx->degree * sizeof(uint64_t);
    if (root_powers) memcpy(root_powers, c->root_powers, bytes);
    if (root_factors) memcpy(root_factors, c->root_factors, bytes);
    if (inv_root_powers) memcpy(inv_root_powers, c->inv_root_powers, bytes);
    if (inv_root_factors) memcpy(inv_root_factors, c->inv_root_factors, bytes);
    if (inverse_degree) *inverse_degree = c->inverse_degree.multiplicand;
    if (inverse_degree_root) *inverse_degree_root = c->inverse_degree_root.multiplicand;
    return ORC_OK;
}

/* HE/PolyRq/PolyContext.swift:175-181 validateNttModuli */
static int validate_ntt_moduli(const orc_poly_context* ctx) {
    for (size_t i = 0; i < ctx->count; ++i)
        if (!is_ntt_modulus(ctx->moduli[i], ctx->degree) || !ctx->ntt[i].valid) return ORC_ERR_INVALID_NTT_MODULUS;
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * PolyRq operations on [batch][L][N] slabs
 * ---------------------------------------------------------------------------------------------- */

static void forward_ntt_poly(const orc_poly_context* ctx, uint64_t* poly) {
    size_t n = (size_t)ctx->degree;
    /* HE/PolyRq/PolyRq+Ntt.swift:209-222: walk the chain from the last modulus down. */
    for (size_t row = ctx->count; row-- > 0;) forward_ntt_row(&ctx->ntt[row], &ctx->reduce[row], poly + row * n);
}
static void inverse_ntt_poly(const orc_poly_context* ctx, uint64_t* poly) {
    size_t n = (size_t)ctx->degree;
    /* HE/PolyRq/PolyRq+Ntt.swift:524-533 */
    for (size_t row = ctx->count; row-- > 0;) inverse_ntt_row(&ctx->ntt[row], &ctx->reduce[row], poly + row * n);
}

int orc_forward_ntt(const orc_poly_context* ctx, uint64_t* data, size_t batch) {
    int status = validate_ntt_moduli(ctx);
    if (status) return status;
    size_t stride = ctx->count * (size_t)ctx->degree;
    for (size_t b = 0; b < batch; ++b) forward_ntt_poly(ctx, data + b * stride);
    return ORC_OK;
}
int orc_inverse_ntt(const orc_poly_context* ctx, uint64_t* data, size_t batch) {
    int status = validate_ntt_moduli(ctx);
    if (status) return status;
    size_t stride = ctx->count * (size_t)ctx->degree;
    for (size_t b = 0; b < batch; ++b) inverse_ntt_poly(ctx, data + b * stride);
    return ORC_OK;
}

/* ---- tiny thread pool helper: each worker takes a contiguous range of batch items (the reference fans whole
 * polynomials / ciphertexts out to tasks: HE/Bfv/Bfv.swift:266-287, HE/Util.swift:139-173). ---- */
typedef void (*range_fn)(void* arg, size_t begin, size_t end);
typedef struct {
    range_fn fn;
    void* arg;
    size_t begin, end;
} range_job;
static void* range_trampoline(void* p) {
    range_job* job = (range_job*)p;
    job->fn(job->arg, job->begin, job->end);
    return NULL;
}
static void parallel_ranges(size_t total, int threads, range_fn fn, void* arg) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > total) threads = (int)(total ? total : 1);
    if (threads == 1) {
        fn(arg, 0, total);
        return;
    }
    pthread_t* ids = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    range_job* jobs = (range_job*)malloc(sizeof(range_job) * threads);
    for (int t = 0; t < threads; ++t) {
        jobs[t].fn = fn;
        jobs[t].arg = arg;
        jobs[t].begin = total * t / threads;
        jobs[t].end = total * (t + 1) / threads;
        pthread_create(&ids[t], NULL, range_trampoline, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(ids[t], NULL);
    free(ids);
    free(jobs);
}

typedef struct {
    const orc_poly_context* ctx;
    uint64_t* data;
    int inverse;
} ntt_mt_arg;
static void ntt_mt_range(void* p, size_t begin, size_t end) {
    ntt_mt_arg* a = (ntt_mt_arg*)p;
    size_t stride = a->ctx->count * (size_t)a->ctx->degree;
    for (size_t b = begin; b < end; ++b) {
        if (a->inverse) {
            inverse_ntt_poly(a->ctx, a->data + b * stride);
        } else {
            forward_ntt_poly(a->ctx, a->data + b * stride);
        }
    }
}
int orc_forward_ntt_mt(const orc_poly_context* ctx, uint64_t* data, size_t batch, int threads) {
    int status = validate_ntt_moduli(ctx);
    if (status) return status;
    ntt_mt_arg arg = {ctx, data, 0};
    parallel_ranges(batch, threads, ntt_mt_range, &arg);
    return ORC_OK;
}
int orc_inverse_ntt_mt(const orc_poly_context* ctx, uint64_t* data, size_t batch, int threads) {
    int status = validate_ntt_moduli(ctx);
    if (status) return status;
    ntt_mt_arg arg = {ctx, data, 1};
    parallel_ranges(batch, threads, ntt_mt_range, &arg);
    return ORC_OK;
}

/* HE/PolyRq/PolyRq.swift:147-157 */
int orc_poly_add(const orc_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch) {
    size_t n = (size_t)ctx->degree, L = ctx->count;
    for (size_t b = 0; b < batch; ++b)
        for (size_t i = 0; i < L; ++i) {
            size_t off = (b * L + i) * n;
            for (size_t k = 0; k < n; ++k) lhs[off + k] = add_mod(lhs[off + k], rhs[off + k], ctx->moduli[i]);
        }
    return ORC_OK;
}
/* HE/PolyRq/PolyRq.swift:164-174 */
int orc_poly_sub(const orc_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch) {
    size_t n = (size_t)ctx->degree, L = ctx->count;
    for (size_t b = 0; b < batch; ++b)
        for (size_t i = 0; i < L; ++i) {
            size_t off = (b * L + i) * n;
            for (size_t k = 0; k < n; ++k) lhs[off + k] = sub_mod(lhs[off + k], rhs[off + k], ctx->moduli[i]);
        }
    return ORC_OK;
}
/* HE/PolyRq/PolyRq.swift:299-309 */
int orc_poly_neg(const orc_poly_context* ctx, uint64_t* data, size_t batch) {
    size_t n = (size_t)ctx->degree, L = ctx->count;
    for (size_t b = 0; b < batch; ++b)
        for (size_t i = 0; i < L; ++i) {
            size_t off = (b * L + i) * n;
            for (size_t k = 0; k < n; ++k) data[off + k] = neg_mod(data[off + k], ctx->moduli[i]);
        }
    return ORC_OK;
}
/* HE/PolyRq/PolyRq.swift:184-204 (Eval *= Eval, Barrett reduceProduct) */
int orc_poly_mul(const orc_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch) {
    size_t n = (size_t)ctx->degree, L = ctx->count;
    for (size_t b = 0; b < batch; ++b)
        for (size_t i = 0; i < L; ++i) {
            size_t off = (b * L + i) * n;
            for (size_t k = 0; k < n; ++k) lhs[off + k] = multiply_mod(&ctx->reduce[i], lhs[off + k], rhs[off + k]);
        }
    return ORC_OK;
}
/* HE/PolyRq/PolyRq.swift:232-245 (poly *= scalarResidues, Shoup) */
int orc_poly_mul_scalar(const orc_poly_context* ctx, uint64_t* data, const uint64_t* scalar_residues, size_t batch) {
    size_t n = (size_t)ctx->degree, L = ctx->count;
    for (size_t b = 0; b < batch; ++b)
        for (size_t i = 0; i < L; ++i) {
            orc_shoup s = shoup_init(scalar_residues[i], ctx->moduli[i]);
            size_t off = (b * L + i) * n;
            for (size_t k = 0; k < n; ++k) data[off + k] = shoup_mul(&s, data[off + k]);
        }
    return ORC_OK;
}

/* HE/PolyRq/PolyRq.swift:365-393 divideAndRoundQLast on one polynomial with `L` moduli. */
static void divide_and_round_q_last_poly(const orc_poly_context* ctx, size_t L, const uint64_t* in, uint64_t* out,
                                         uint64_t* scratch_last) {
    size_t n = (size_t)ctx->degree;
    uint64_t q_last = ctx->moduli[L - 1];
    uint64_t q_last_div2 = q_last >> 1;
    const uint64_t* last = in + (L - 1) * n;
    for (size_t k = 0; k < n; ++k) scratch_last[k] = add_mod(last[k], q_last_div2, q_last);
    for (size_t i = 0; i + 1 < L; ++i) {
        const orc_modulus* qi = &ctx->reduce[i];
        const orc_shoup* inverse_q_last = &ctx->inverse_q_last[L][i];
        uint64_t q_last_div2_mod_qi = reduce_u64(qi, q_last_div2);
        for (size_t k = 0; k < n; ++k) {
            uint64_t tmp = reduce_u64(qi, scratch_last[k]);
            uint64_t coeff = sub_mod(add_mod(in[i * n + k], q_last_div2_mod_qi, qi->p), tmp, qi->p);
            out[i * n + k] = shoup_mul(inverse_q_last, coeff);
        }
    }
}

typedef struct {
    const orc_poly_context* ctx;
    const uint64_t* in;
    uint64_t* out;
} divround_arg;
static void divround_range(void* p, size_t begin, size_t end) {
    divround_arg* a = (divround_arg*)p;
    size_t n = (size_t)a->ctx->degree, L = a->ctx->count;
    uint64_t* scratch = (uint64_t*)malloc(n * sizeof(uint64_t));
    for (size_t b = begin; b < end; ++b)
        divide_and_round_q_last_poly(a->ctx, L, a->in + b * L * n, a->out + b * (L - 1) * n, scratch);
    free(scratch);
}
int orc_poly_divide_and_round_q_last_mt(const orc_poly_context* ctx, const uint64_t* in, uint64_t* out,
                                        size_t batch, int threads) {
    if (ctx->count < 2) return ORC_ERR_INVALID_POLY_CONTEXT; /* no next context, PolyRq.swift:366-368 */
    divround_arg arg = {ctx, in, out};
    parallel_ranges(batch, threads, divround_range, &arg);
    return ORC_OK;
}
int orc_poly_divide_and_round_q_last(const orc_poly_context* ctx, const uint64_t* in, uint64_t* out, size_t batch) {
    return orc_poly_divide_and_round_q_last_mt(ctx, in, out, batch, 1);
}

/* HE/PolyRq/PolyRq.swift:210-225 addingLazyProduct (wrapping UInt128 accumulate) */
int orc_poly_adding_lazy_product(const orc_poly_context* ctx, const uint64_t* lhs, const uint64_t* rhs,
                                 uint64_t* acc_lo_hi) {
    size_t total = ctx->count * (size_t)ctx->degree;
    for (size_t k = 0; k < total; ++k) {
        u128 acc = (((u128)acc_lo_hi[2 * k + 1]) << 64) | acc_lo_hi[2 * k];
        acc += (u128)lhs[k] * rhs[k];
        acc_lo_hi[2 * k] = (uint64_t)acc;
        acc_lo_hi[2 * k + 1] = (uint64_t)(acc >> 64);
    }
    return ORC_OK;
}
/* HE/Bfv/Bfv.swift:380-394 reduceToCiphertext (double-word Barrett per word) */
int orc_poly_reduce_accumulator(const orc_poly_context* ctx, const uint64_t* acc_lo_hi, uint64_t* out) {
    size_t n = (size_t)ctx->degree;
    for (size_t i = 0; i < ctx->count; ++i)
        for (size_t k = 0; k < n; ++k) {
            size_t idx = i * n + k;
            u128 acc = (((u128)acc_lo_hi[2 * idx + 1]) << 64) | acc_lo_hi[2 * idx];
            out[idx] = reduce_u128(&ctx->reduce[i], acc);
        }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * _RnsBaseConverter  (HE/RnsBaseConverter.swift, HE/CrtComposer.swift)
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    size_t in_count, out_count;
    uint64_t* in_moduli;
    orc_modulus* out_reduce;
    uint64_t* punctured;           /* [out][in]  (q/q_i) mod t_j       RnsBaseConverter.swift:41-54 */
    orc_shoup* inverse_punctured;  /* [in]       (q/q_i)^-1 mod q_i    CrtComposer.swift:32-50 */
} orc_base_converter;

static void base_converter_free(orc_base_converter* c) {
    free(c->in_moduli);
    free(c->out_reduce);
    free(c->punctured);
    free(c->inverse_punctured);
    memset(c, 0, sizeof(*c));
}

static int base_converter_init(orc_base_converter* c, const uint64_t* in_moduli, size_t in_count,
                               const uint64_t* out_moduli, size_t out_count) {
    memset(c, 0, sizeof(*c));
    c->in_count = in_count;
    c->out_count = out_count;
    c->in_moduli = (uint64_t*)malloc(in_count * sizeof(uint64_t));
    memcpy(c->in_moduli, in_moduli, in_count * sizeof(uint64_t));
    c->out_reduce = (orc_modulus*)malloc(out_count * sizeof(orc_modulus));
    c->punctured = (uint64_t*)malloc(in_count * out_count * sizeof(uint64_t));
    c->inverse_punctured = (orc_shoup*)malloc(in_count * sizeof(orc_shoup));
    for (size_t j = 0; j < out_count; ++j) {
        c->out_reduce[j] = modulus_init(out_moduli[j]);
        for (size_t i = 0; i < in_count; ++i) {
            uint64_t prod = 1;
            for (size_t k = 0; k < in_count; ++k)
                if (in_moduli[k] != in_moduli[i]) prod = reduce_u128(&c->out_reduce[j], (u128)prod * in_moduli[k]);
            c->punctured[j * in_count + i] = prod;
        }
    }
    for (size_t i = 0; i < in_count; ++i) {
        orc_modulus qi = modulus_init(in_moduli[i]);
        uint64_t prod = 1;
        for (size_t k = 0; k < in_count; ++k)
            if (in_moduli[k] != in_moduli[i]) prod = reduce_u128(&qi, (u128)prod * in_moduli[k]);
        uint64_t inverse;
        int status = orc_inverse_mod(prod, in_moduli[i], &inverse);
        if (status) {
            base_converter_free(c);
            return status;
        }
        c->inverse_punctured[i] = shoup_init(inverse, in_moduli[i]);
    }
    return ORC_OK;
}

/* RnsBaseConverter.swift:97-106 convertApproximateProducts: y_i = x_i (q/q_i)^-1 mod q_i; in/out [in][N] */
static void base_converter_products(const orc_base_converter* c, const uint64_t* in, uint64_t* products, size_t n) {
    for (size_t i = 0; i < c->in_count; ++i)
        for (size_t k = 0; k < n; ++k) products[i * n + k] = shoup_mul(&c->inverse_punctured[i], in[i * n + k]);
}
/* RnsBaseConverter.swift:117-143 convertApproximate(using:): exact wrapping-128 sum, then Barrett-128. */
static void base_converter_convert_products(const orc_base_converter* c, const uint64_t* products, uint64_t* out,
                                            size_t n) {
    for (size_t j = 0; j < c->out_count; ++j)
        for (size_t k = 0; k < n; ++k) {
            u128 sum = 0;
            for (size_t i = 0; i < c->in_count; ++i) sum += (u128)products[i * n + k] * c->punctured[j * c->in_count + i];
            out[j * n + k] = reduce_u128(&c->out_reduce[j], sum);
        }
}

int orc_rns_convert_approximate(const orc_poly_context* input, const orc_poly_context* output, const uint64_t* in,
                                uint64_t* out) {
    if (input->degree != output->degree) return ORC_ERR_INVALID_ARGUMENT;
    orc_base_converter c;
    int status = base_converter_init(&c, input->moduli, input->count, output->moduli, output->count);
    if (status) return status;
    size_t n = (size_t)input->degree;
    uint64_t* products = (uint64_t*)malloc(input->count * n * sizeof(uint64_t));
    base_converter_products(&c, in, products, n);
    base_converter_convert_products(&c, products, out, n);
    free(products);
    base_converter_free(&c);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * _RnsTool  (HE/RnsTool.swift)
 * ---------------------------------------------------------------------------------------------- */

struct orc_rns_tool {
    uint64_t degree;
    size_t L;                 /* input moduli count */
    uint64_t* q;              /* [L] */
    uint64_t t;
    uint64_t gamma, mtilde;   /* T.rnsCorrectionFactor, T.mTilde (MA/Scalar.swift:498-525): 2^62-40797 / 2^32 for
                               * UInt64, 2^30-20405 / 2^16 for UInt32 */
    orc_modulus t_reduce;
    size_t ext_count;         /* L+2: the prefix of [Bsk..., mTilde] this tool sees (RnsTool.swift:185-186) */
    uint64_t* ext_moduli;     /* [L+2]; rows 0..L are "Bsk", row L+1 is "mTilde" (see SURVEY 4.3 for lower levels) */
    orc_poly_context* qbsk;   /* [Q, Bsk]  RnsTool.swift:235-239 */
    orc_shoup neg_inverse_q_mod_mtilde;   /* RnsTool.swift:163-169 */
    orc_shoup inverse_b_mod_msk;          /* RnsTool.swift:246-250 */
    orc_shoup inverse_gamma_mod_t;        /* RnsTool.swift:150-153 */
    uint64_t* m_tilde_mod_q;              /* [L]    RnsTool.swift:233 */
    uint64_t* prod_gamma_t_mod_q;         /* [L]    RnsTool.swift:149 */
    uint64_t neg_inverse_q_mod_t_gamma[2];/*        RnsTool.swift:157-160 */
    orc_shoup* q_mod_bsk;                 /* [L+1]  RnsTool.swift:224-227 */
    orc_shoup* inverse_mtilde_mod_bsk;    /* [L+1]  RnsTool.swift:228-231 */
    orc_shoup* inverse_q_mod_bsk;         /* [L+1]  RnsTool.swift:241-245 */
    orc_shoup* b_mod_q;                   /* [L]    RnsTool.swift:211-216 */
    orc_shoup* neg_b_mod_q;               /* [L]    RnsTool.swift:217-223 */
    orc_base_converter q_to_bsk, q_to_bsk_mtilde, b_to_msk, b_to_q, q_to_t_gamma;
};

void orc_rns_tool_destroy(orc_rns_tool* tool) {
    if (!tool) return;
    free(tool->q);
    free(tool->ext_moduli);
    orc_poly_context_destroy(tool->qbsk);
    free(tool->m_tilde_mod_q);
    free(tool->prod_gamma_t_mod_q);
    free(tool->q_mod_bsk);
    free(tool->inverse_mtilde_mod_bsk);
    free(tool->inverse_q_mod_bsk);
    free(tool->b_mod_q);
    free(tool->neg_b_mod_q);
    base_converter_free(&tool->q_to_bsk);
    base_converter_free(&tool->q_to_bsk_mtilde);
    base_converter_free(&tool->b_to_msk);
    base_converter_free(&tool->b_to_q);
    base_converter_free(&tool->q_to_t_gamma);
    free(tool);
}

/* RnsTool.swift:132-251.  `bsk_mtilde` is the full [Bsk..., mTilde] list of the shared RnsToolContext
 * (RnsTool.swift:28-45); the tool takes its first L+2 entries (:185-186). */
static int rns_tool_create_shared(const orc_poly_context* input, uint64_t t, const uint64_t* bsk_mtilde,
                                  size_t bsk_mtilde_count, int word_bits, orc_rns_tool** out) {
    *out = NULL;
    size_t L = input->count;
    if (L + 2 > bsk_mtilde_count) return ORC_ERR_INVALID_POLY_CONTEXT;
    int status = validate_modulus(t);
    if (status) return status;
    orc_rns_tool* tool = (orc_rns_tool*)calloc(1, sizeof(*tool));
    tool->degree = input->degree;
    tool->L = L;
    tool->t = t;
    tool->t_reduce = modulus_init(t);
    tool->q = (uint64_t*)malloc(L * sizeof(uint64_t));
    memcpy(tool->q, input->moduli, L * sizeof(uint64_t));
    tool->ext_count = L + 2;
    tool->ext_moduli = (uint64_t*)malloc((L + 2) * sizeof(uint64_t));
    memcpy(tool->ext_moduli, bsk_mtilde, (L + 2) * sizeof(uint64_t));
    const uint64_t* bsk = tool->ext_moduli; /* L+1 entries */
    const uint64_t m_sk = bsk[L];
    const uint64_t gamma = word_bits == 32 ? ORC_GAMMA32 : ORC_GAMMA;
    const uint64_t mtilde = word_bits == 32 ? ORC_MTILDE32 : ORC_MTILDE;
    tool->gamma = gamma;
    tool->mtilde = mtilde;

#define TOOL_FAIL(code)           \
    do {                          \
        orc_rns_tool_destroy(tool); \
        return (code);            \
    } while (0)

    tool->prod_gamma_t_mod_q = (uint64_t*)malloc(L * sizeof(uint64_t));
    for (size_t i = 0; i < L; ++i) tool->prod_gamma_t_mod_q[i] = reduce_u128(&input->reduce[i], (u128)gamma * t);
    {
        uint64_t inverse;
        status = orc_inverse_mod(gamma, t, &inverse); /* Swift passes gamma unreduced to Euclid */
        if (status) TOOL_FAIL(status);
        tool->inverse_gamma_mod_t = shoup_init(inverse % t, t);
    }
    {
        /* tGammaContext = [t, gamma] (RnsTool.swift:62-64): validation as PolyContext would do it. */
        uint64_t t_gamma[2] = {t, gamma};
        orc_poly_context* t_gamma_ctx;
        status = orc_poly_context_create(input->degree, t_gamma, 2, &t_gamma_ctx);
        if (status) TOOL_FAIL(status);
        orc_poly_context_destroy(t_gamma_ctx);
        status = base_converter_init(&tool->q_to_t_gamma, input->moduli, L, t_gamma, 2);
        if (status) TOOL_FAIL(status);
        for (int j = 0; j < 2; ++j) {
            orc_modulus m = modulus_init(t_gamma[j]);
            uint64_t q_mod = q_remainder_n(input->moduli, L, &m);
            uint64_t inverse;
            status = orc_inverse_mod(q_mod, t_gamma[j], &inverse);
            if (status) TOOL_FAIL(status);
            tool->neg_inverse_q_mod_t_gamma[j] = neg_mod(inverse, t_gamma[j]);
        }
    }
    {
        orc_modulus m_tilde = modulus_init(mtilde);
        uint64_t q_mod = q_remainder_n(input->moduli, L, &m_tilde);
        uint64_t inverse;
        status = orc_inverse_mod(q_mod, mtilde, &inverse);
        if (status) TOOL_FAIL(status);
        tool->neg_inverse_q_mod_mtilde = shoup_init(neg_mod(inverse, mtilde), mtilde);
    }
    tool->b_mod_q = (orc_shoup*)malloc(L * sizeof(orc_shoup));
    tool->neg_b_mod_q = (orc_shoup*)malloc(L * sizeof(orc_shoup));
    for (size_t i = 0; i < L; ++i) {
        uint64_t b_mod_qi = q_remainder_n(bsk, L, &input->reduce[i]);
        tool->b_mod_q[i] = shoup_init(b_mod_qi, input->moduli[i]);
        tool->neg_b_mod_q[i] = shoup_init(neg_mod(b_mod_qi, input->moduli[i]), input->moduli[i]);
    }
    tool->q_mod_bsk = (orc_shoup*)malloc((L + 1) * sizeof(orc_shoup));
    tool->inverse_mtilde_mod_bsk = (orc_shoup*)malloc((L + 1) * sizeof(orc_shoup));
    tool->inverse_q_mod_bsk = (orc_shoup*)malloc((L + 1) * sizeof(orc_shoup));
    for (size_t j = 0; j <= L; ++j) {
        orc_modulus m = modulus_init(bsk[j]);
        uint64_t q_mod = q_remainder_n(input->moduli, L, &m);
        tool->q_mod_bsk[j] = shoup_init(q_mod, bsk[j]);
        uint64_t inverse;
        status = orc_inverse_mod(mtilde % bsk[j], bsk[j], &inverse);
        if (status) TOOL_FAIL(status);
        tool->inverse_mtilde_mod_bsk[j] = shoup_init(inverse, bsk[j]);
        status = orc_inverse_mod(q_mod, bsk[j], &inverse);
        if (status) TOOL_FAIL(status);
        tool->inverse_q_mod_bsk[j] = shoup_init(inverse, bsk[j]);
    }
    tool->m_tilde_mod_q = (uint64_t*)malloc(L * sizeof(uint64_t));
    for (size_t i = 0; i < L; ++i) tool->m_tilde_mod_q[i] = reduce_u64(&input->reduce[i], mtilde);
    {
        uint64_t* qbsk_moduli = (uint64_t*)malloc((2 * L + 1) * sizeof(uint64_t));
        memcpy(qbsk_moduli, input->moduli, L * sizeof(uint64_t));
        memcpy(qbsk_moduli + L, bsk, (L + 1) * sizeof(uint64_t));
        /* PolyContext(degree:moduli:child:) (PolyContext.swift:151-172) validates only the appended moduli
         * one prefix at a time; uniqueness is checked on every prefix. */
        status = ORC_OK;
        for (size_t k = L + 1; k <= 2 * L + 1 && !status; ++k)
            status = poly_context_check_prefix(input->degree, qbsk_moduli, k, 1);
        if (!status) status = orc_poly_context_create(input->degree, qbsk_moduli, 2 * L + 1, &tool->qbsk);
        free(qbsk_moduli);
        if (status) TOOL_FAIL(status);
    }
    /* RnsTool.swift:240-250 with the shared RnsToolContext (:44-62): mSkContext holds the TOP level's m_sk for every
     * level, so rnsConvertBtoMSk converts B to the top m_sk and bModMSk is B mod the top m_sk, while the multiplicand's
     * inverse and the MultiplyConstantModulus are taken mod this level's m_sk (the same prime only at the top level).
     * inverseMod (HE/Scalar.swift:76-96) is given a value that may exceed its modulus; orc_inverse_mod restates its
     * loop statement by statement, so it returns what the reference computes. */
    const uint64_t top_m_sk = bsk_mtilde[bsk_mtilde_count - 2];
    {
        orc_modulus m = modulus_init(top_m_sk);
        uint64_t b_mod_msk = q_remainder_n(bsk, L, &m);
        uint64_t inverse;
        status = orc_inverse_mod(b_mod_msk, m_sk, &inverse);
        if (status) TOOL_FAIL(status);
        tool->inverse_b_mod_msk = shoup_init(inverse, m_sk);
    }
    status = base_converter_init(&tool->q_to_bsk, input->moduli, L, bsk, L + 1);
    if (status) TOOL_FAIL(status);
    status = base_converter_init(&tool->q_to_bsk_mtilde, input->moduli, L, tool->ext_moduli, L + 2);
    if (status) TOOL_FAIL(status);
    status = base_converter_init(&tool->b_to_msk, bsk, L, &top_m_sk, 1);
    if (status) TOOL_FAIL(status);
    status = base_converter_init(&tool->b_to_q, bsk, L, input->moduli, L);
    if (status) TOOL_FAIL(status);
#undef TOOL_FAIL
    *out = tool;
    return ORC_OK;
}

/* RnsToolContext.init (RnsTool.swift:28-45): Bsk = L+1 NTT-friendly primes of bitWidth-3 bits (61 for UInt64, 29 for
 * UInt32), ascending. */
static int generate_bsk_mtilde(uint64_t degree, size_t L, int word_bits, uint64_t** out, size_t* out_count) {
    int* bits = (int*)malloc((L + 1) * sizeof(int));
    for (size_t i = 0; i <= L; ++i) bits[i] = word_bits - 3;
    uint64_t* list = (uint64_t*)malloc((L + 2) * sizeof(uint64_t));
    int status = orc_generate_primes(bits, (int)(L + 1), 1, degree, word_bits, list);
    free(bits);
    if (status) {
        free(list);
        return status;
    }
    list[L + 1] = word_bits == 32 ? ORC_MTILDE32 : ORC_MTILDE;
    /* bSkMTildeContext = PolyContext(degree, moduli: Bsk + [mTilde]) (RnsTool.swift:36-37) */
    orc_poly_context* check;
    status = orc_poly_context_create(degree, list, L + 2, &check);
    if (status) {
        free(list);
        return status;
    }
    orc_poly_context_destroy(check);
    *out = list;
    *out_count = L + 2;
    return ORC_OK;
}

int orc_rns_tool_create_word(const orc_poly_context* input, uint64_t t, int word_bits, orc_rns_tool** out) {
    if (word_bits != 32 && word_bits != 64) return ORC_ERR_INVALID_ARGUMENT;
    uint64_t* list;
    size_t count;
    int status = generate_bsk_mtilde(input->degree, input->count, word_bits, &list, &count);
    if (status) return status;
    status = rns_tool_create_shared(input, t, list, count, word_bits, out);
    free(list);
    return status;
}
int orc_rns_tool_create(const orc_poly_context* input, uint64_t t, orc_rns_tool** out) {
    return orc_rns_tool_create_word(input, t, 64, out);
}

size_t orc_rns_tool_bsk_count(const orc_rns_tool* tool) { return tool->L + 1; }
void orc_rns_tool_bsk_moduli(const orc_rns_tool* tool, uint64_t* out) {
    memcpy(out, tool->ext_moduli, (tool->L + 1) * sizeof(uint64_t));
}

/* RnsTool.swift:313-316 convertApproximateBskMTilde */
int orc_rns_convert_approximate_bsk_mtilde(const orc_rns_tool* tool, const uint64_t* in, uint64_t* out) {
    size_t n = (size_t)tool->degree, L = tool->L;
    uint64_t* scaled = (uint64_t*)malloc(L * n * sizeof(uint64_t));
    for (size_t i = 0; i < L; ++i) {
        orc_shoup s = shoup_init(tool->m_tilde_mod_q[i], tool->q[i]);
        for (size_t k = 0; k < n; ++k) scaled[i * n + k] = shoup_mul(&s, in[i * n + k]);
    }
    base_converter_products(&tool->q_to_bsk_mtilde, scaled, scaled, n);
    base_converter_convert_products(&tool->q_to_bsk_mtilde, scaled, out, n);
    free(scaled);
    return ORC_OK;
}

/* RnsTool.swift:339-368 smallMontgomeryReduce */
int orc_rns_small_montgomery_reduce(const orc_rns_tool* tool, uint64_t* inout) {
    size_t n = (size_t)tool->degree, L = tool->L;
    const uint64_t m_tilde_div_threshold = tool->mtilde >> 1;
    uint64_t* m_tilde_row = inout + (L + 1) * n;
    for (size_t k = 0; k < n; ++k) m_tilde_row[k] = shoup_mul(&tool->neg_inverse_q_mod_mtilde, m_tilde_row[k]);
    for (size_t j = 0; j <= L; ++j) {
        uint64_t bsk = tool->ext_moduli[j];
        for (size_t k = 0; k < n; ++k) {
            uint64_t r = m_tilde_row[k];
            r = (r < m_tilde_div_threshold) ? r : r + bsk - tool->mtilde;
            uint64_t x = inout[j * n + k];
            x += shoup_mul_lazy(&tool->q_mod_bsk[j], r);
            inout[j * n + k] = shoup_mul(&tool->inverse_mtilde_mod_bsk[j], x);
        }
    }
    return ORC_OK;
}

/* RnsTool.swift:324-331 liftQToQBsk */
int orc_rns_lift_q_to_qbsk(const orc_rns_tool* tool, const uint64_t* in, uint64_t* out) {
    size_t n = (size_t)tool->degree, L = tool->L;
    uint64_t* ext = (uint64_t*)malloc((L + 2) * n * sizeof(uint64_t));
    orc_rns_convert_approximate_bsk_mtilde(tool, in, ext);
    orc_rns_small_montgomery_reduce(tool, ext);
    memcpy(out, in, L * n * sizeof(uint64_t));
    memcpy(out + L * n, ext, (L + 1) * n * sizeof(uint64_t));
    free(ext);
    return ORC_OK;
}

/* RnsTool.swift:378-398 approximateFloor */
int orc_rns_approximate_floor(const orc_rns_tool* tool, const uint64_t* in, uint64_t* out) {
    size_t n = (size_t)tool->degree, L = tool->L;
    uint64_t* products = (uint64_t*)calloc(L * n, sizeof(uint64_t));
    base_converter_products(&tool->q_to_bsk, in, products, n);
    base_converter_convert_products(&tool->q_to_bsk, products, out, n);
    free(products);
    const uint64_t* poly_mod_bsk = in + L * n;
    for (size_t j = 0; j <= L; ++j) {
        uint64_t bsk = tool->ext_moduli[j];
        for (size_t k = 0; k < n; ++k) {
            uint64_t input_coeff = poly_mod_bsk[j * n + k];
            uint64_t output_coeff = out[j * n + k];
            out[j * n + k] = shoup_mul(&tool->inverse_q_mod_bsk[j], input_coeff + bsk - output_coeff);
        }
    }
    return ORC_OK;
}

/* RnsTool.swift:402-450 convertApproximateBskToQ */
int orc_rns_convert_approximate_bsk_to_q(const orc_rns_tool* tool, const uint64_t* in, uint64_t* out) {
    size_t n = (size_t)tool->degree, L = tool->L;
    const uint64_t m_sk = tool->ext_moduli[L];
    const uint64_t* poly_mod_msk = in + L * n;
    uint64_t* products = (uint64_t*)calloc(L * n, sizeof(uint64_t));
    uint64_t* alpha = (uint64_t*)malloc(n * sizeof(uint64_t));
    uint8_t* exceeds = (uint8_t*)malloc(n);
    base_converter_products(&tool->b_to_msk, in, products, n);
    base_converter_convert_products(&tool->b_to_msk, products, alpha, n);
    const uint64_t threshold = m_sk >> 1;
    for (size_t k = 0; k < n; ++k) {
        uint64_t a = shoup_mul(&tool->inverse_b_mod_msk, alpha[k] + m_sk - poly_mod_msk[k]);
        alpha[k] = a;
        exceeds[k] = a > threshold;
    }
    base_converter_convert_products(&tool->b_to_q, products, out, n);
    for (size_t i = 0; i < L; ++i) {
        uint64_t qi = tool->q[i];
        for (size_t k = 0; k < n; ++k) {
            uint64_t adjust = exceeds[k] ? shoup_mul(&tool->b_mod_q[i], m_sk - alpha[k])
                                         : shoup_mul(&tool->neg_b_mod_q[i], alpha[k]);
            out[i * n + k] = add_mod(out[i * n + k], adjust, qi);
        }
    }
    free(products);
    free(alpha);
    free(exceeds);
    return ORC_OK;
}

/* RnsTool.swift:453-456 floorQBskToQ */
int orc_rns_floor_qbsk_to_q(const orc_rns_tool* tool, const uint64_t* in, uint64_t* out) {
    size_t n = (size_t)tool->degree, L = tool->L;
    uint64_t* floored = (uint64_t*)malloc((L + 1) * n * sizeof(uint64_t));
    orc_rns_approximate_floor(tool, in, floored);
    orc_rns_convert_approximate_bsk_to_q(tool, floored, out);
    free(floored);
    return ORC_OK;
}

/* RnsTool.swift:272-302 scaleAndRound */
int orc_rns_scale_and_round(const orc_rns_tool* tool, const uint64_t* in, uint64_t scaling_factor, uint64_t* out) {
    size_t n = (size_t)tool->degree, L = tool->L;
    const uint64_t t = tool->t, gamma = tool->gamma;
    uint64_t* scaled = (uint64_t*)malloc(L * n * sizeof(uint64_t));
    uint64_t* t_gamma = (uint64_t*)malloc(2 * n * sizeof(uint64_t));
    for (size_t i = 0; i < L; ++i) {
        orc_shoup s = shoup_init(tool->prod_gamma_t_mod_q[i], tool->q[i]);
        for (size_t k = 0; k < n; ++k) scaled[i * n + k] = shoup_mul(&s, in[i * n + k]);
    }
    base_converter_products(&tool->q_to_t_gamma, scaled, scaled, n);
    base_converter_convert_products(&tool->q_to_t_gamma, scaled, t_gamma, n);
    orc_shoup neg_inv_t = shoup_init(tool->neg_inverse_q_mod_t_gamma[0], t);
    orc_shoup neg_inv_gamma = shoup_init(tool->neg_inverse_q_mod_t_gamma[1], gamma);
    const uint64_t corrected_gamma = gamma / 2;
    uint64_t scaled_inverse_gamma_mod_t = shoup_mul(&tool->inverse_gamma_mod_t, scaling_factor);
    orc_shoup final_mul = shoup_init(scaled_inverse_gamma_mod_t, t);
    for (size_t k = 0; k < n; ++k) {
        uint64_t poly_mod_t = shoup_mul(&neg_inv_t, t_gamma[k]);
        uint64_t poly_mod_gamma = shoup_mul(&neg_inv_gamma, t_gamma[n + k]);
        uint64_t s_gamma_greater = neg_mod(reduce_u64(&tool->t_reduce, gamma - poly_mod_gamma), t);
        uint64_t s_gamma_less = reduce_u64(&tool->t_reduce, poly_mod_gamma);
        uint64_t s_gamma = poly_mod_gamma > corrected_gamma ? s_gamma_greater : s_gamma_less;
        out[k] = shoup_mul(&final_mul, sub_mod(poly_mod_t, s_gamma, t));
    }
    free(scaled);
    free(t_gamma);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Context<Bfv<UInt64>>  (HE/Context.swift:94-159, HE/EncryptionParameters.swift:123-167)
 * ---------------------------------------------------------------------------------------------- */

struct orc_bfv_context {
    uint64_t degree, t;
    size_t coefficient_count;       /* all moduli incl. the key-switching one */
    uint64_t* coefficient_moduli;
    size_t L;                       /* ciphertext moduli at top level */
    orc_poly_context** ciphertext;  /* [L+1]: ciphertext[k] has k moduli (k >= 1) */
    orc_poly_context** key_switching; /* [L+1]: key_switching[k] = (q_0..q_{k-1}, q_ks); NULL if no ks modulus */
    orc_rns_tool** tools;           /* [L+1]: tools[k] for k ciphertext moduli */
};

void orc_bfv_context_destroy(orc_bfv_context* ctx) {
    if (!ctx) return;
    for (size_t k = 0; k <= ctx->L; ++k) {
        if (ctx->ciphertext) orc_poly_context_destroy(ctx->ciphertext[k]);
        if (ctx->key_switching) orc_poly_context_destroy(ctx->key_switching[k]);
        if (ctx->tools) orc_rns_tool_destroy(ctx->tools[k]);
    }
    free(ctx->ciphertext);
    free(ctx->key_switching);
    free(ctx->tools);
    free(ctx->coefficient_moduli);
    free(ctx);
}

int orc_bfv_context_create(uint64_t degree, uint64_t t, const uint64_t* q, size_t count, orc_bfv_context** out) {
    return orc_bfv_context_create_word(degree, t, q, count, 64, out);
}

/* Context<Bfv<T>>.init for T = UInt64 (word_bits 64) or UInt32 (word_bits 32): the word type fixes the largest
 * modulus (2^62-1 / 2^30-1), gamma, mTilde and the Bsk prime size (MA/Scalar.swift:498-525, RnsTool.swift:30-33). */
int orc_bfv_context_create_word(uint64_t degree, uint64_t t, const uint64_t* q, size_t count, int word_bits,
                                orc_bfv_context** out) {
    *out = NULL;
    if (word_bits != 32 && word_bits != 64) return ORC_ERR_INVALID_ARGUMENT;
    const uint64_t max_modulus = word_bits == 32 ? ((((uint64_t)1) << 30) - 1) : ORC_MAX_MODULUS;
    const uint64_t gamma = word_bits == 32 ? ORC_GAMMA32 : ORC_GAMMA;
    const uint64_t mtilde = word_bits == 32 ? ORC_MTILDE32 : ORC_MTILDE;
    /* EncryptionParameters.init checks (securityLevel: .unchecked), HE/EncryptionParameters.swift:136-166 */
    if (!is_power_of_two(degree)) return ORC_ERR_INVALID_ENCRYPTION_PARAMETERS;
    if (count == 0 || count > 32) return ORC_ERR_INVALID_ENCRYPTION_PARAMETERS;
    for (size_t i = 0; i < count; ++i)
        if (!(q[i] > t) || !is_ntt_modulus(q[i], degree)) return ORC_ERR_INVALID_ENCRYPTION_PARAMETERS;
    for (size_t i = 0; i <= count; ++i) {
        uint64_t m = i < count ? q[i] : t;
        if (!orc_is_prime(m) || m < 1 || m > max_modulus || m == gamma || m == mtilde)
            return ORC_ERR_INVALID_ENCRYPTION_PARAMETERS;
    }
    int status;
    {
        orc_poly_context* secret_key_context; /* HE/Context.swift:98-100 */
        status = orc_poly_context_create(degree, q, count, &secret_key_context);
        if (status) return status;
        orc_poly_context_destroy(secret_key_context);
    }
    orc_bfv_context* ctx = (orc_bfv_context*)calloc(1, sizeof(*ctx));
    ctx->degree = degree;
    ctx->t = t;
    ctx->coefficient_count = count;
    ctx->coefficient_moduli = (uint64_t*)malloc(count * sizeof(uint64_t));
    memcpy(ctx->coefficient_moduli, q, count * sizeof(uint64_t));
    size_t L = count > 1 ? count - 1 : count; /* HE/Context.swift:102-107 */
    int has_ks = count > 1;
    ctx->L = L;
    ctx->ciphertext = (orc_poly_context**)calloc(L + 1, sizeof(void*));
    ctx->key_switching = (orc_poly_context**)calloc(L + 1, sizeof(void*));
    ctx->tools = (orc_rns_tool**)calloc(L + 1, sizeof(void*));
    for (size_t k = 1; k <= L && !status; ++k) status = orc_poly_context_create(degree, q, k, &ctx->ciphertext[k]);
    if (has_ks) {
        uint64_t* moduli = (uint64_t*)malloc((L + 1) * sizeof(uint64_t));
        for (size_t k = 1; k <= L && !status; ++k) {
            memcpy(moduli, q, k * sizeof(uint64_t));
            moduli[k] = q[count - 1];
            status = orc_poly_context_create(degree, moduli, k + 1, &ctx->key_switching[k]);
            /* HE/Context.swift:122-124 */
            if (!status &&
                !((uint64_t)(k + 1) <
                  orc_poly_context_max_lazy_product_accumulation_count(ctx->key_switching[k], word_bits)))
                status = ORC_ERR_INVALID_ENCRYPTION_PARAMETERS;
        }
        free(moduli);
    }
    if (!status) {
        orc_poly_context* plaintext_context; /* HE/Context.swift:128-130 */
        status = orc_poly_context_create(degree, &t, 1, &plaintext_context);
        if (!status) orc_poly_context_destroy(plaintext_context);
    }
    if (!status) {
        uint64_t* bsk_mtilde;
        size_t bsk_mtilde_count;
        status = generate_bsk_mtilde(degree, L, word_bits, &bsk_mtilde, &bsk_mtilde_count); /* HE/Context.swift:133-135 */
        for (size_t k = L; k >= 1 && !status; --k) /* HE/Context.swift:136-141 */
            status = rns_tool_create_shared(ctx->ciphertext[k], t, bsk_mtilde, bsk_mtilde_count, word_bits,
                                            &ctx->tools[k]);
        if (bsk_mtilde_count) free(bsk_mtilde);
    }
    if (status) {
        orc_bfv_context_destroy(ctx);
        return status;
    }
    *out = ctx;
    return ORC_OK;
}

size_t orc_bfv_ciphertext_moduli_count(const orc_bfv_context* ctx) { return ctx->L; }
const orc_poly_context* orc_bfv_ciphertext_context(const orc_bfv_context* ctx, size_t k) {
    return (k >= 1 && k <= ctx->L) ? ctx->ciphertext[k] : NULL;
}
const orc_poly_context* orc_bfv_key_switching_context(const orc_bfv_context* ctx, size_t k) {
    return (k >= 1 && k <= ctx->L) ? ctx->key_switching[k] : NULL;
}
const orc_poly_context* orc_bfv_qbsk_context(const orc_bfv_context* ctx, size_t k) {
    return (k >= 1 && k <= ctx->L) ? ctx->tools[k]->qbsk : NULL;
}
const orc_rns_tool* orc_bfv_rns_tool(const orc_bfv_context* ctx, size_t k) {
    return (k >= 1 && k <= ctx->L) ? ctx->tools[k] : NULL;
}

/* ------------------------------------------------------------------------------------------------
 * Bfv scheme operations
 * ---------------------------------------------------------------------------------------------- */

/* Bfv+Multiply.swift:51-57 computeBehzPolys for one polynomial: lift to [Q,Bsk] then forward NTT. */
static void behz_poly(const orc_rns_tool* tool, const uint64_t* poly, uint64_t* out) {
    orc_rns_lift_q_to_qbsk(tool, poly, out);
    forward_ntt_poly(tool->qbsk, out);
}

/* Bfv+Multiply.swift:31-48 dropExtendedBase for one polynomial (Eval over [Q,Bsk] -> Coeff over Q). */
static void drop_extended_base_poly(const orc_rns_tool* tool, uint64_t* poly_qbsk, uint64_t* out) {
    const orc_poly_context* qbsk = tool->qbsk;
    size_t n = (size_t)qbsk->degree;
    for (size_t i = 0; i < qbsk->count; ++i) {
        orc_shoup s = shoup_init(tool->t, qbsk->moduli[i]);
        for (size_t k = 0; k < n; ++k) poly_qbsk[i * n + k] = shoup_mul(&s, poly_qbsk[i * n + k]);
    }
    inverse_ntt_poly(qbsk, poly_qbsk);
    orc_rns_floor_qbsk_to_q(tool, poly_qbsk, out);
}

/* Bfv+Multiply.swift:18-85 for one pair. lhs,rhs: [2][L][N]; out: [3][L][N]. */
static void bfv_mul_one(const orc_bfv_context* ctx, size_t L, const uint64_t* lhs, const uint64_t* rhs,
                        uint64_t* out) {
    const orc_rns_tool* tool = ctx->tools[L];
    const orc_poly_context* qbsk = tool->qbsk;
    size_t n = (size_t)ctx->degree, ext = qbsk->count * n, poly = L * n;
    uint64_t* buf = (uint64_t*)malloc(7 * ext * sizeof(uint64_t));
    uint64_t *a0 = buf, *a1 = buf + ext, *b0 = buf + 2 * ext, *b1 = buf + 3 * ext;
    uint64_t *d0 = buf + 4 * ext, *d1 = buf + 5 * ext, *d2 = buf + 6 * ext;
    behz_poly(tool, lhs, a0);
    behz_poly(tool, lhs + poly, a1);
    behz_poly(tool, rhs, b0);
    behz_poly(tool, rhs + poly, b1);
    for (size_t i = 0; i < qbsk->count; ++i) {
        const orc_modulus* m = &qbsk->reduce[i];
        for (size_t k = 0; k < n; ++k) {
            size_t idx = i * n + k;
            d0[idx] = multiply_mod(m, a0[idx], b0[idx]);
            d1[idx] = add_mod(multiply_mod(m, a0[idx], b1[idx]), multiply_mod(m, a1[idx], b0[idx]), m->p);
            d2[idx] = multiply_mod(m, a1[idx], b1[idx]);
        }
    }
    drop_extended_base_poly(tool, d0, out);
    drop_extended_base_poly(tool, d1, out + poly);
    drop_extended_base_poly(tool, d2, out + 2 * poly);
    free(buf);
}

typedef struct {
    const orc_bfv_context* ctx;
    size_t L;
    const uint64_t *lhs, *rhs;
    uint64_t* out;
} mul_arg;
static void mul_range(void* p, size_t begin, size_t end) {
    mul_arg* a = (mul_arg*)p;
    size_t poly = a->L * (size_t)a->ctx->degree;
    for (size_t b = begin; b < end; ++b)
        bfv_mul_one(a->ctx, a->L, a->lhs + b * 2 * poly, a->rhs + b * 2 * poly, a->out + b * 3 * poly);
}
int orc_bfv_mul_mt(const orc_bfv_context* ctx, size_t L, const uint64_t* lhs, const uint64_t* rhs, uint64_t* out,
                   size_t batch, int threads) {
    if (L < 1 || L > ctx->L) return ORC_ERR_INVALID_ARGUMENT;
    mul_arg arg = {ctx, L, lhs, rhs, out};
    parallel_ranges(batch, threads, mul_range, &arg);
    return ORC_OK;
}
int orc_bfv_mul(const orc_bfv_context* ctx, size_t L, const uint64_t* lhs, const uint64_t* rhs, uint64_t* out,
                size_t batch) {
    return orc_bfv_mul_mt(ctx, L, lhs, rhs, out, batch, 1);
}

/* Bfv+Keys.swift:123-208 _computeKeySwitchingUpdate.
 * key layout: [L_top][2][L_top+1][N] -- ciphertext j, component c, row r of the top key-switching context. */
int orc_bfv_key_switching_update(const orc_bfv_context* ctx, size_t L, const uint64_t* target, const uint64_t* key,
                                 uint64_t* update) {
    if (L < 1 || L > ctx->L) return ORC_ERR_INVALID_ARGUMENT;
    if (!ctx->key_switching[L]) return ORC_ERR_INVALID_ARGUMENT;
    const orc_poly_context* ks = ctx->key_switching[L];
    const size_t n = (size_t)ctx->degree;
    const size_t rns_count = L + 1;
    const size_t top_rows = ctx->L + 1;
    uint64_t* prod = (uint64_t*)malloc(2 * rns_count * n * sizeof(uint64_t)); /* [2][L+1][N] Eval */
    uint64_t* buffer = (uint64_t*)malloc(n * sizeof(uint64_t));
    u128* acc = (u128*)malloc(2 * n * sizeof(u128));
    for (size_t rns = 0; rns < rns_count; ++rns) {
        size_t key_index = (rns == rns_count - 1) ? top_rows - 1 : rns;
        const orc_modulus* key_modulus = &ks->reduce[rns];
        memset(acc, 0, 2 * n * sizeof(u128));
        for (size_t j = 0; j < L; ++j) {
            memcpy(buffer, target + j * n, n * sizeof(uint64_t));
            if (ks->moduli[j] > key_modulus->p)
                for (size_t k = 0; k < n; ++k) buffer[k] = reduce_u64(key_modulus, buffer[k]);
            forward_ntt_row(&ks->ntt[rns], key_modulus, buffer);
            for (size_t c = 0; c < 2; ++c) {
                const uint64_t* key_row = key + ((j * 2 + c) * top_rows + key_index) * n;
                for (size_t k = 0; k < n; ++k) acc[c * n + k] += (u128)buffer[k] * key_row[k];
            }
        }
        for (size_t c = 0; c < 2; ++c)
            for (size_t k = 0; k < n; ++k) prod[(c * rns_count + rns) * n + k] = reduce_u128(key_modulus, acc[c * n + k]);
    }
    uint64_t* scratch = (uint64_t*)malloc(n * sizeof(uint64_t));
    for (size_t c = 0; c < 2; ++c) {
        uint64_t* poly = prod + c * rns_count * n;
        inverse_ntt_poly(ks, poly);                                           /* :204 */
        divide_and_round_q_last_poly(ks, rns_count, poly, update + c * L * n, scratch); /* :206 */
    }
    free(scratch);
    free(acc);
    free(buffer);
    free(prod);
    return ORC_OK;
}

typedef struct {
    const orc_bfv_context* ctx;
    size_t L;
    const uint64_t *ct3, *key;
    uint64_t* out;
} relin_arg;
static void relin_range(void* p, size_t begin, size_t end) {
    relin_arg* a = (relin_arg*)p;
    size_t n = (size_t)a->ctx->degree, poly = a->L * n;
    const orc_poly_context* qctx = a->ctx->ciphertext[a->L];
    uint64_t* update = (uint64_t*)malloc(2 * poly * sizeof(uint64_t));
    for (size_t b = begin; b < end; ++b) {
        const uint64_t* ct = a->ct3 + b * 3 * poly;
        uint64_t* out = a->out + b * 2 * poly;
        orc_bfv_key_switching_update(a->ctx, a->L, ct + 2 * poly, a->key, update);
        memcpy(out, ct, 2 * poly * sizeof(uint64_t));
        for (size_t c = 0; c < 2; ++c) /* Bfv.swift:216-217 */
            for (size_t i = 0; i < a->L; ++i)
                for (size_t k = 0; k < n; ++k) {
                    size_t idx = c * poly + i * n + k;
                    out[idx] = add_mod(out[idx], update[idx], qctx->moduli[i]);
                }
    }
    free(update);
}
/* Bfv.swift:201-219 relinearize */
int orc_bfv_relinearize_mt(const orc_bfv_context* ctx, size_t L, const uint64_t* ct3, const uint64_t* key,
                           uint64_t* out, size_t batch, int threads) {
    if (L < 1 || L > ctx->L) return ORC_ERR_INVALID_ARGUMENT;
    if (!key || !ctx->key_switching[L]) return ORC_ERR_MISSING_RELINEARIZATION_KEY;
    relin_arg arg = {ctx, L, ct3, key, out};
    parallel_ranges(batch, threads, relin_range, &arg);
    return ORC_OK;
}
int orc_bfv_relinearize(const orc_bfv_context* ctx, size_t L, const uint64_t* ct3, const uint64_t* key,
                        uint64_t* out, size_t batch) {
    return orc_bfv_relinearize_mt(ctx, L, ct3, key, out, batch, 1);
}

/* Bfv.swift:163-171 modSwitchDown */
int orc_bfv_mod_switch_down(const orc_bfv_context* ctx, size_t L, size_t poly_count, const uint64_t* in,
                            uint64_t* out, size_t batch) {
    if (L < 1 || L > ctx->L) return ORC_ERR_INVALID_ARGUMENT;
    if (L < 2) return ORC_ERR_INVALID_POLY_CONTEXT;
    return orc_poly_divide_and_round_q_last(ctx->ciphertext[L], in, out, batch * poly_count);
}

/* Bfv.swift:120-129 mulAssign(EvalCiphertext, EvalPlaintext) */
int orc_bfv_mul_plain(const orc_bfv_context* ctx, size_t L, size_t poly_count, uint64_t* ct, const uint64_t* pt,
                      size_t batch) {
    if (L < 1 || L > ctx->L) return ORC_ERR_INVALID_ARGUMENT;
    const orc_poly_context* qctx = ctx->ciphertext[L];
    size_t poly = L * (size_t)ctx->degree;
    for (size_t b = 0; b < batch; ++b)
        for (size_t c = 0; c < poly_count; ++c) orc_poly_mul(qctx, ct + (b * poly_count + c) * poly, pt + b * poly, 1);
    return ORC_OK;
}

/* floor(Q / t) mod p for Q = q_0 ... q_{L-1}: the composed Q in 64-bit limbs, long division by the single word t,
 * the quotient's residue by Horner (RnsTool.swift:170-182 composes Q and t with Width32 integers). */
static uint64_t q_div_t_mod(const uint64_t* q, size_t L, uint64_t t, uint64_t p) {
    uint64_t limbs[16] = {1};
    size_t used = 1;
    for (size_t i = 0; i < L; ++i) {
        uint64_t carry = 0;
        for (size_t w = 0; w < used; ++w) {
            u128 v = (u128)limbs[w] * q[i] + carry;
            limbs[w] = (uint64_t)v;
            carry = (uint64_t)(v >> 64);
        }
        if (carry) limbs[used++] = carry;
    }
    uint64_t remainder = 0, residue = 0;
    for (size_t w = used; w-- > 0;) {
        u128 v = ((u128)remainder << 64) | limbs[w];
        uint64_t digit = (uint64_t)(v / t);
        remainder = (uint64_t)(v % t);
        residue = (uint64_t)((((u128)residue << 64) | digit) % p);
    }
    return residue;
}

/* Bfv+Encrypt.swift:75-140 plaintextTranslate (Bfv.addAssignCoeff / subAssignCoeff of a Coeff ciphertext and a Coeff
 * plaintext, Bfv.swift:110-117): c0 +- (floor(Q/t) m + floor((Q mod t) m + (t + 1)/2) / t)) per residue row.
 * ct [batch][poly_count][L][N] in place, plaintexts [batch][N] with values < t. */
int orc_bfv_plaintext_translate(const orc_bfv_context* ctx, size_t L, size_t poly_count, uint64_t* ct,
                                const uint64_t* plaintexts, int subtract, size_t batch) {
    if (L < 1 || L > ctx->L || poly_count < 1) return ORC_ERR_INVALID_ARGUMENT;
    const orc_poly_context* qctx = ctx->ciphertext[L];
    const size_t n = (size_t)ctx->degree, poly = L * n;
    const uint64_t t = ctx->t, t_threshold = (t + 1) / 2; /* RnsTool.swift:123-125 */
    uint64_t q_mod_t = 1 % t;                             /* RnsTool.swift:167 */
    for (size_t i = 0; i < L; ++i) q_mod_t = mul_mod_slow(q_mod_t, qctx->moduli[i] % t, t);
    for (size_t i = 0; i < L; ++i) {
        const uint64_t p = qctx->moduli[i];
        const orc_shoup delta = shoup_init(q_div_t_mod(qctx->moduli, L, t, p), p); /* qDivT, RnsTool.swift:176-182 */
        for (size_t b = 0; b < batch; ++b) {
            uint64_t* c0 = ct + b * poly_count * poly + i * n;
            for (size_t k = 0; k < n; ++k) {
                const uint64_t m = plaintexts[b * n + k];
                const uint64_t adjust = (uint64_t)(((u128)q_mod_t * m + t_threshold) / t); /* :92-107 */
                const uint64_t round_q_times_mt = add_mod(shoup_mul(&delta, m), adjust, p);
                c0[k] = subtract ? sub_mod(c0[k], round_q_times_mt, p) : add_mod(c0[k], round_q_times_mt, p);
            }
        }
    }
    return ORC_OK;
}

/* Bfv.swift:476-505 innerProduct(ciphertexts:plaintexts:) with the lazy accumulator and the
 * maxLazyProductAccumulationCount reduce cadence (Bfv.swift:365-376,496-500). */
int orc_bfv_inner_product_plain(const orc_bfv_context* ctx, size_t L, size_t poly_count, const uint64_t* cts,
                                const uint64_t* pts, const uint8_t* present, size_t count, uint64_t* out) {
    if (L < 1 || L > ctx->L || count == 0) return ORC_ERR_INVALID_ARGUMENT;
    const orc_poly_context* qctx = ctx->ciphertext[L];
    size_t n = (size_t)ctx->degree, poly = L * n;
    uint64_t max_product_count = orc_poly_context_max_lazy_product_accumulation_count(qctx, 64);
    u128* acc = (u128*)calloc(poly_count * poly, sizeof(u128));
    uint64_t reduce_count = 0;
    for (size_t item = 0; item < count; ++item) {
        if (present && !present[item]) continue;
        const uint64_t* pt = pts + item * poly;
        for (size_t c = 0; c < poly_count; ++c) {
            const uint64_t* ct = cts + (item * poly_count + c) * poly;
            for (size_t k = 0; k < poly; ++k) acc[c * poly + k] += (u128)ct[k] * pt[k];
        }
        reduce_count += 1;
        if (reduce_count >= max_product_count) {
            reduce_count = 0;
            for (size_t c = 0; c < poly_count; ++c)
                for (size_t i = 0; i < L; ++i)
                    for (size_t k = 0; k < n; ++k) {
                        size_t idx = c * poly + i * n + k;
                        acc[idx] = reduce_u128(&qctx->reduce[i], acc[idx]);
                    }
        }
    }
    for (size_t c = 0; c < poly_count; ++c)
        for (size_t i = 0; i < L; ++i)
            for (size_t k = 0; k < n; ++k) {
                size_t idx = c * poly + i * n + k;
                out[idx] = reduce_u128(&qctx->reduce[i], acc[idx]);
            }
    free(acc);
    return ORC_OK;
}

/* Bfv.swift:315-361 innerProduct(ct,ct): lazy tensor accumulation in [Q,Bsk], one dropExtendedBase. */
int orc_bfv_inner_product(const orc_bfv_context* ctx, size_t L, const uint64_t* lhs, const uint64_t* rhs,
                          size_t count, uint64_t* out) {
    if (L < 1 || L > ctx->L || count == 0) return ORC_ERR_INVALID_ARGUMENT;
    const orc_rns_tool* tool = ctx->tools[L];
    const orc_poly_context* qbsk = tool->qbsk;
    size_t n = (size_t)ctx->degree, ext = qbsk->count * n, poly = L * n;
    uint64_t max_product_count = orc_poly_context_max_lazy_product_accumulation_count(qbsk, 64) / 2;
    u128* acc = (u128*)calloc(3 * ext, sizeof(u128));
    uint64_t* buf = (uint64_t*)malloc(4 * ext * sizeof(uint64_t));
    uint64_t *a0 = buf, *a1 = buf + ext, *b0 = buf + 2 * ext, *b1 = buf + 3 * ext;
    uint64_t reduce_count = 0;
    for (size_t item = 0; item < count; ++item) {
        behz_poly(tool, lhs + item * 2 * poly, a0);
        behz_poly(tool, lhs + item * 2 * poly + poly, a1);
        behz_poly(tool, rhs + item * 2 * poly, b0);
        behz_poly(tool, rhs + item * 2 * poly + poly, b1);
        for (size_t k = 0; k < ext; ++k) {
            acc[k] += (u128)a0[k] * b0[k];
            acc[ext + k] += (u128)a0[k] * b1[k];
            acc[ext + k] += (u128)a1[k] * b0[k];
            acc[2 * ext + k] += (u128)a1[k] * b1[k];
        }
        reduce_count += 1;
        if (reduce_count >= max_product_count) {
            reduce_count = 0;
            for (size_t c = 0; c < 3; ++c)
                for (size_t i = 0; i < qbsk->count; ++i)
                    for (size_t k = 0; k < n; ++k) {
                        size_t idx = c * ext + i * n + k;
                        acc[idx] = reduce_u128(&qbsk->reduce[i], acc[idx]);
                    }
        }
    }
    uint64_t* sum = (uint64_t*)malloc(ext * sizeof(uint64_t));
    for (size_t c = 0; c < 3; ++c) {
        for (size_t i = 0; i < qbsk->count; ++i)
            for (size_t k = 0; k < n; ++k) sum[i * n + k] = reduce_u128(&qbsk->reduce[i], acc[c * ext + i * n + k]);
        drop_extended_base_poly(tool, sum, out + c * poly);
    }
    free(sum);
    free(buf);
    free(acc);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------------------------
 * "Next" rows of the scope table: Galois automorphisms, x^k multiplication, plaintext <-> Eval.
 * ------------------------------------------------------------------------------------------------------------------ */

/* Galois.swift:100-105 isValidGaloisElement */
int orc_is_valid_galois_element(uint64_t element, uint64_t degree) {
    return degree != 0 && (degree & (degree - 1)) == 0 && (element & 1) == 1 && element < (degree << 1) && element > 1;
}

/* Galois.swift:115-143 PolyRq<Coeff>.applyGalois with GaloisCoeffIterator (:34-48):
 * coefficient i goes to (i * element) mod N, negated when floor(i * element / N) is odd. */
int orc_poly_apply_galois_coeff(const orc_poly_context* ctx, const uint64_t* in, uint64_t* out, uint64_t element,
                                size_t batch) {
    const uint64_t n = ctx->degree;
    if (!orc_is_valid_galois_element(element, n)) return ORC_ERR_INVALID_ARGUMENT;
    int log2n = 0;
    while (((uint64_t)1 << log2n) < n) ++log2n;
    for (size_t b = 0; b < batch; ++b)
        for (size_t r = 0; r < ctx->count; ++r) {
            const uint64_t* src = in + (b * ctx->count + r) * n;
            uint64_t* dst = out + (b * ctx->count + r) * n;
            uint64_t raw_out_index = 0, out_index = 0;
            for (uint64_t i = 0; i < n; ++i) {
                int negate = ((raw_out_index >> log2n) & 1) != 0;
                dst[out_index] = negate ? neg_mod(src[i], ctx->moduli[r]) : src[i];
                raw_out_index += element;
                out_index = raw_out_index & (n - 1);
            }
        }
    return ORC_OK;
}

/* Galois.swift:153-168 PolyRq<Eval>.applyGalois with GaloisEvalIterator (:81-92):
 * out[i] = in[bitrev_logN(((element * bitrev_{logN+1}(i + N)) >> 1) mod N)]. */
int orc_poly_apply_galois_eval(const orc_poly_context* ctx, const uint64_t* in, uint64_t* out, uint64_t element,
                               size_t batch) {
    const uint64_t n = ctx->degree;
    if (!orc_is_valid_galois_element(element, n)) return ORC_ERR_INVALID_ARGUMENT;
    int log2n = 0;
    while (((uint64_t)1 << log2n) < n) ++log2n;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t reversed = orc_reverse_bits((uint32_t)(i + n), log2n + 1);
        uint64_t index_raw = ((element * reversed) >> 1) & (n - 1);
        uint64_t in_index = orc_reverse_bits((uint32_t)index_raw, log2n);
        for (size_t b = 0; b < batch; ++b)
            for (size_t r = 0; r < ctx->count; ++r) out[(b * ctx->count + r) * n + i] = in[(b * ctx->count + r) * n + in_index];
    }
    return ORC_OK;
}

/* PolyRq.swift:398-422 multiplyPowerOfX, literally: rotate the columns (Array2d.swift:193-211), then negate a range. */
int orc_poly_multiply_power_of_x(const orc_poly_context* ctx, uint64_t* data, int64_t power, size_t batch) {
    const int64_t n = (int64_t)ctx->degree, twice = n << 1;
    const int64_t abs_power = power < 0 ? -power : power;
    const int64_t abs_step = abs_power % twice;
    if (abs_step == 0) return ORC_OK;
    const int64_t rotation_step = power < 0 ? -abs_step : abs_step;
    int64_t effective = (rotation_step % n) % n; /* Swift %: sign of the dividend */
    if (effective < 0) effective += n;           /* toRemainder */
    int64_t neg_begin, neg_end;
    if (power < 0 && abs_step < n) { neg_begin = n - abs_step; neg_end = n; }
    else if (power < 0) { neg_begin = 0; neg_end = twice - abs_step; }
    else if (abs_step < n) { neg_begin = 0; neg_end = abs_step; }
    else { neg_begin = abs_step - n; neg_end = n; }
    uint64_t* row = (uint64_t*)malloc((size_t)n * sizeof(uint64_t));
    for (size_t b = 0; b < batch; ++b)
        for (size_t r = 0; r < ctx->count; ++r) {
            uint64_t* d = data + (b * ctx->count + r) * (size_t)n;
            if (effective != 0) { /* new row = d[n-e..n) + d[0..n-e) */
                memcpy(row, d + (n - effective), (size_t)effective * sizeof(uint64_t));
                memcpy(row + effective, d, (size_t)(n - effective) * sizeof(uint64_t));
                memcpy(d, row, (size_t)n * sizeof(uint64_t));
            }
            for (int64_t k = neg_begin; k < neg_end; ++k) d[k] = neg_mod(d[k], ctx->moduli[r]);
        }
    free(row);
    return ORC_OK;
}

/* Bfv.swift:174-198 applyGalois(ciphertext:element:using:): 2-poly Coeff ciphertexts,
 * c0' = galois(c0) + update0, c1' = update1, update = keySwitchingUpdate(galois(c1), galoisKey[element]). */
int orc_bfv_apply_galois(const orc_bfv_context* ctx, size_t L, const uint64_t* ct, uint64_t element,
                         const uint64_t* key, uint64_t* out, size_t batch) {
    if (L < 1 || L > ctx->L) return ORC_ERR_INVALID_ARGUMENT;
    if (!key || !ctx->key_switching[L]) return ORC_ERR_MISSING_RELINEARIZATION_KEY;
    const orc_poly_context* qctx = ctx->ciphertext[L];
    const size_t n = (size_t)ctx->degree, poly = L * n;
    if (!orc_is_valid_galois_element(element, ctx->degree)) return ORC_ERR_INVALID_ARGUMENT;
    uint64_t* rotated = (uint64_t*)malloc(2 * poly * sizeof(uint64_t));
    uint64_t* update = (uint64_t*)malloc(2 * poly * sizeof(uint64_t));
    for (size_t b = 0; b < batch; ++b) {
        orc_poly_apply_galois_coeff(qctx, ct + b * 2 * poly, rotated, element, 2);
        orc_bfv_key_switching_update(ctx, L, rotated + poly, key, update);
        uint64_t* o = out + b * 2 * poly;
        for (size_t i = 0; i < L; ++i)
            for (size_t k = 0; k < n; ++k) {
                o[i * n + k] = add_mod(rotated[i * n + k], update[i * n + k], qctx->moduli[i]);
                o[poly + i * n + k] = update[poly + i * n + k];
            }
    }
    free(update);
    free(rotated);
    return ORC_OK;
}

/* Plaintext.swift:149-170 convertToEvalFormat: centered lift of coefficients mod t into each q_i
 * (x < (t+1)/2 ? x : x + (q_i - t); RnsTool.swift:123-125,168), then forward NTT. */
int orc_bfv_plaintext_to_eval(const orc_bfv_context* ctx, size_t L, const uint64_t* plaintext, uint64_t* out,
                              size_t batch) {
    if (L < 1 || L > ctx->L) return ORC_ERR_INVALID_ARGUMENT;
    const orc_poly_context* qctx = ctx->ciphertext[L];
    const size_t n = (size_t)ctx->degree;
    const uint64_t t_threshold = (ctx->t + 1) / 2;
    for (size_t b = 0; b < batch; ++b)
        for (size_t i = 0; i < L; ++i) {
            const uint64_t t_increment = qctx->moduli[i] - ctx->t;
            for (size_t k = 0; k < n; ++k) {
                uint64_t x = plaintext[b * n + k];
                out[(b * L + i) * n + k] = x < t_threshold ? x : x + t_increment;
            }
        }
    return orc_forward_ntt(qctx, out, batch);
}

/* Plaintext.swift:176-191 convertToCoeffFormat: inverse NTT, undo the lift on row 0, keep row 0. */
int orc_bfv_plaintext_to_coeff(const orc_bfv_context* ctx, size_t L, const uint64_t* plaintext_eval, uint64_t* out,
                               size_t batch) {
    if (L < 1 || L > ctx->L) return ORC_ERR_INVALID_ARGUMENT;
    const orc_poly_context* qctx = ctx->ciphertext[L];
    const size_t n = (size_t)ctx->degree;
    const uint64_t t_threshold = (ctx->t + 1) / 2, t_increment = qctx->moduli[0] - ctx->t;
    uint64_t* tmp = (uint64_t*)malloc(L * n * sizeof(uint64_t));
    for (size_t b = 0; b < batch; ++b) {
        memcpy(tmp, plaintext_eval + b * L * n, L * n * sizeof(uint64_t));
        orc_inverse_ntt(qctx, tmp, 1);
        for (size_t k = 0; k < n; ++k) out[b * n + k] = tmp[k] >= t_threshold ? tmp[k] - t_increment : tmp[k];
    }
    free(tmp);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Wire format: CoefficientPacking + PolyRq.serialize / load
 * ------------------------------------------------------------------------------------------------------------------ */

/* CoefficientPacking.coefficientsToBytesByteCount (CoefficientPacking.swift:141-144) */
size_t orc_coefficients_to_bytes_byte_count(size_t coeff_count, int bits_per_coeff, int skip_lsbs) {
    size_t bits = coeff_count * (size_t)(bits_per_coeff - skip_lsbs);
    return (bits + 7) / 8;
}
/* CoefficientPacking.bytesToCoefficientsCoeffCount (CoefficientPacking.swift:34-44) */
size_t orc_bytes_to_coefficients_coeff_count(size_t byte_count, int bits_per_coeff, int decode, int skip_lsbs) {
    size_t serialized = (size_t)(bits_per_coeff - skip_lsbs);
    if (decode) return 8 * byte_count / serialized;
    return (8 * byte_count + serialized - 1) / serialized;
}
static int packing_valid(int bits_per_coeff, int skip_lsbs) { /* :27-31 */
    return bits_per_coeff > 0 && bits_per_coeff > skip_lsbs && skip_lsbs >= 0;
}

/* CoefficientPacking.coefficientsToBytesInplace (CoefficientPacking.swift:169-213), the byte-by-byte loop as written */
int orc_coefficients_to_bytes(const uint64_t* coeffs, size_t coeff_count, int bits_per_coeff, int skip_lsbs,
                              uint8_t* bytes, size_t bytes_count) {
    if (!packing_valid(bits_per_coeff, skip_lsbs)) return ORC_ERR_INVALID_ARGUMENT;
    if (bytes_count == 0) return ORC_OK;
    size_t byte_index = 0;
    const int serialized_bit_count = bits_per_coeff - skip_lsbs;
    uint8_t byte = 0;
    int remaining_bits = 8;
    for (size_t k = 0; k < coeff_count; ++k) {
        uint64_t coeff = coeffs[k] >> skip_lsbs;
        int remaining_coeff_bits = serialized_bit_count;
        do {
            if (remaining_bits == 0) {
                remaining_bits = 8;
                bytes[byte_index] = byte;
                byte_index += 1;
                if (byte_index == bytes_count) return ORC_OK;
                byte = 0;
            }
            int shift = remaining_bits < remaining_coeff_bits ? remaining_bits : remaining_coeff_bits;
            uint8_t byte_value = (uint8_t)((coeff >> (remaining_coeff_bits - shift)) & 0xff);
            byte = (uint8_t)(((unsigned)byte << shift) | byte_value);
            remaining_coeff_bits -= shift;
            remaining_bits -= shift;
        } while (remaining_coeff_bits > 0);
    }
    if (byte_index < bytes_count) {
        byte = (uint8_t)((unsigned)byte << remaining_bits);
        bytes[byte_index] = byte;
        byte_index += 1;
    }
    return byte_index == bytes_count ? ORC_OK : ORC_ERR_INVALID_ARGUMENT;
}

/* CoefficientPacking.bytesToCoefficientsInplace (CoefficientPacking.swift:75-138), the 64-bit buffer walk as written */
int orc_bytes_to_coefficients(const uint8_t* bytes, size_t byte_count, int bits_per_coeff, int skip_lsbs,
                              uint64_t* coeffs, size_t coeff_count) {
    if (!packing_valid(bits_per_coeff, skip_lsbs)) return ORC_ERR_INVALID_ARGUMENT;
    const int serialized_bit_count = bits_per_coeff - skip_lsbs;
    size_t coeff_index = 0;
    int unused_bit_count = 0;
    uint64_t unused_bits = 0;
    for (size_t chunk = 0; chunk < byte_count; chunk += 8) {
        size_t end = chunk + 8 < byte_count ? chunk + 8 : byte_count;
        uint64_t buffer = 0; /* BufferType(bigEndianBytes:), left-aligned when short (Util.swift) */
        for (size_t b = chunk; b < end; ++b) buffer |= (uint64_t)bytes[b] << (8 * (7 - (b - chunk)));
        int new_bits_count = 0;
        if (unused_bit_count != 0) {
            new_bits_count = serialized_bit_count - unused_bit_count;
            uint64_t coeff = buffer >> (64 - new_bits_count);
            coeff |= unused_bits << new_bits_count;
            if (coeff_index < coeff_count) coeffs[coeff_index] = coeff << skip_lsbs;
            coeff_index += 1;
        }
        size_t remaining = coeff_count > coeff_index ? coeff_count - coeff_index : 0;
        size_t per_buffer = (size_t)((64 - new_bits_count) / serialized_bit_count);
        size_t take = remaining < per_buffer ? remaining : per_buffer;
        for (size_t i = 0; i < take; ++i) {
            int msbs_to_clear = new_bits_count + (int)i * serialized_bit_count;
            uint64_t coeff = buffer << msbs_to_clear;
            coeff >>= 64 - serialized_bit_count;
            coeffs[coeff_index] = coeff << skip_lsbs;
            coeff_index += 1;
        }
        unused_bit_count += 64 % serialized_bit_count;
        if (unused_bit_count >= serialized_bit_count) unused_bit_count -= serialized_bit_count;
        unused_bits = unused_bit_count == 0 ? 0 : buffer & (((uint64_t)1 << unused_bit_count) - 1);
    }
    if (coeff_index < coeff_count) {
        coeffs[coeff_index] = unused_bits << (serialized_bit_count - unused_bit_count + skip_lsbs);
        coeff_index += 1;
    }
    return coeff_index == coeff_count ? ORC_OK : ORC_ERR_INVALID_ARGUMENT;
}

static int ceil_log2_u64(uint64_t x) { /* ModularArithmetic/Scalar.swift:266-269 */
    int log2 = 63 - __builtin_clzll(x);
    return log2 + ((x & (x - 1)) == 0 ? 0 : 1);
}

/* PolyContext.serializationByteCount (PolyRq+Serialize.swift:90-99) */
size_t orc_poly_serialization_byte_count(const orc_poly_context* ctx, int skip_lsbs) {
    size_t total = 0;
    for (size_t r = 0; r < ctx->count; ++r)
        total += orc_coefficients_to_bytes_byte_count((size_t)ctx->degree, ceil_log2_u64(ctx->moduli[r]), skip_lsbs);
    return total;
}
/* PolyRq.serialize (PolyRq+Serialize.swift:69-87): rows packed one after the other, each padded to a byte */
int orc_poly_serialize(const orc_poly_context* ctx, const uint64_t* data, int skip_lsbs, uint8_t* bytes) {
    size_t offset = 0;
    for (size_t r = 0; r < ctx->count; ++r) {
        int bits = ceil_log2_u64(ctx->moduli[r]);
        size_t count = orc_coefficients_to_bytes_byte_count((size_t)ctx->degree, bits, skip_lsbs);
        int status = orc_coefficients_to_bytes(data + r * ctx->degree, (size_t)ctx->degree, bits, skip_lsbs,
                                               bytes + offset, count);
        if (status != ORC_OK) return status;
        offset += count;
    }
    return ORC_OK;
}
/* PolyRq.load(from:skipLSBs:) (PolyRq+Serialize.swift:31-62); a short buffer is serializedBufferSizeMismatch */
int orc_poly_deserialize(const orc_poly_context* ctx, const uint8_t* bytes, size_t byte_count, int skip_lsbs,
                         uint64_t* data) {
    size_t offset = 0;
    for (size_t r = 0; r < ctx->count; ++r) {
        int bits = ceil_log2_u64(ctx->moduli[r]);
        size_t count = orc_coefficients_to_bytes_byte_count((size_t)ctx->degree, bits, skip_lsbs);
        if (offset + count > byte_count) return ORC_ERR_INVALID_ARGUMENT;
        int status = orc_bytes_to_coefficients(bytes + offset, count, bits, skip_lsbs, data + r * ctx->degree,
                                               (size_t)ctx->degree);
        if (status != ORC_OK) return status;
        offset += count;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Seeded polynomials: NistAes128Ctr = BufferedRng<NistCtrDrbg> (Random/NistCtrDrbg.swift,
 * Random/NistAes128Ctr.swift, Random/BufferedRng.swift) and PolyRq.randomizeUniform(using:)
 * (PolyRq/PolyRq+Randomize.swift:56-75).  AES-128 itself is FIPS-197 (the reference calls swift-crypto's
 * AES._CTR, a pinned third-party dependency: swift-crypto 3.15.1); restated byte-wise here.
 * ------------------------------------------------------------------------------------------------------------------ */

static uint8_t aes_sbox[256];
static int aes_sbox_ready = 0;
static uint8_t gf_mul(uint8_t a, uint8_t b) {
    uint8_t r = 0;
    while (b) {
        if (b & 1) r ^= a;
        a = (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0));
        b >>= 1;
    }
    return r;
}
static void aes_init_sbox(void) { /* FIPS-197 5.1.1: multiplicative inverse then the affine map */
    if (aes_sbox_ready) return;
    for (int x = 0; x < 256; ++x) {
        uint8_t inv = 0;
        if (x) for (int y = 1; y < 256; ++y) if (gf_mul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        uint8_t s = inv;
        for (int k = 1; k <= 4; ++k) s ^= (uint8_t)((inv << k) | (inv >> (8 - k)));
        aes_sbox[x] = (uint8_t)(s ^ 0x63);
    }
    aes_sbox_ready = 1;
}
typedef struct { uint8_t round_keys[11][16]; } aes128_key;
static void aes128_expand(const uint8_t key[16], aes128_key* out) { /* FIPS-197 5.2 */
    aes_init_sbox();
    memcpy(out->round_keys[0], key, 16);
    uint8_t rcon = 1;
    for (int r = 1; r <= 10; ++r) {
        const uint8_t* prev = out->round_keys[r - 1];
        uint8_t* cur = out->round_keys[r];
        uint8_t t[4] = {aes_sbox[prev[13]], aes_sbox[prev[14]], aes_sbox[prev[15]], aes_sbox[prev[12]]};
        t[0] ^= rcon;
        rcon = (uint8_t)((rcon << 1) ^ ((rcon & 0x80) ? 0x1b : 0));
        for (int i = 0; i < 4; ++i) cur[i] = prev[i] ^ t[i];
        for (int i = 4; i < 16; ++i) cur[i] = prev[i] ^ cur[i - 4];
    }
}
static void aes128_encrypt_block(const aes128_key* k, const uint8_t in[16], uint8_t out[16]) { /* FIPS-197 5.1 */
    uint8_t s[16];
    for (int i = 0; i < 16; ++i) s[i] = in[i] ^ k->round_keys[0][i];
    for (int r = 1; r <= 10; ++r) {
        uint8_t t[16];
        for (int c = 0; c < 4; ++c)          /* SubBytes + ShiftRows: state is column-major, s[4c + row] */
            for (int row = 0; row < 4; ++row) t[4 * c + row] = aes_sbox[s[4 * ((c + row) & 3) + row]];
        if (r < 10) {
            for (int c = 0; c < 4; ++c) {    /* MixColumns */
                const uint8_t* a = t + 4 * c;
                s[4 * c + 0] = (uint8_t)(gf_mul(a[0], 2) ^ gf_mul(a[1], 3) ^ a[2] ^ a[3]);
                s[4 * c + 1] = (uint8_t)(a[0] ^ gf_mul(a[1], 2) ^ gf_mul(a[2], 3) ^ a[3]);
                s[4 * c + 2] = (uint8_t)(a[0] ^ a[1] ^ gf_mul(a[2], 2) ^ gf_mul(a[3], 3));
                s[4 * c + 3] = (uint8_t)(gf_mul(a[0], 3) ^ a[1] ^ a[2] ^ gf_mul(a[3], 2));
            }
        } else {
            memcpy(s, t, 16);
        }
        for (int i = 0; i < 16; ++i) s[i] ^= k->round_keys[r][i];
    }
    memcpy(out, s, 16);
}

/* NistCtrDrbg (Random/NistCtrDrbg.swift:23-96): key, 128-bit big-endian counter V ("nonce"), reseed counter */
struct orc_ctr_drbg {
    uint8_t key[16];
    uint8_t v[16]; /* big-endian */
    int64_t reseed_counter;
};
static void be128_add(uint8_t v[16], uint64_t amount) {
    for (int i = 15; i >= 0 && amount; --i) {
        uint64_t sum = (uint64_t)v[i] + (amount & 0xff);
        v[i] = (uint8_t)sum;
        amount = (amount >> 8) + (sum >> 8);
    }
}
/* AES._CTR.encrypt(data, key, nonce: V + 1): keystream blocks E(V+1), E(V+2), ... xor data */
static void drbg_ctr_encrypt(const orc_ctr_drbg* d, const uint8_t* data, size_t count, uint8_t* out) {
    aes128_key k;
    aes128_expand(d->key, &k);
    uint8_t counter[16], block[16];
    memcpy(counter, d->v, 16);
    for (size_t off = 0; off < count; off += 16) {
        be128_add(counter, 1);
        aes128_encrypt_block(&k, counter, block);
        for (size_t i = 0; i < 16 && off + i < count; ++i) out[off + i] = (uint8_t)(block[i] ^ (data ? data[off + i] : 0));
    }
}
static void drbg_update(orc_ctr_drbg* d, const uint8_t provided[32]) { /* ctrDrbgUpdate, :58-66 */
    uint8_t x[32];
    drbg_ctr_encrypt(d, provided, 32, x);
    memcpy(d->key, x, 16);
    memcpy(d->v, x + 16, 16);
}
int orc_ctr_drbg_create(const uint8_t entropy[32], orc_ctr_drbg** out) { /* init(entropy:), :47-55 */
    orc_ctr_drbg* d = (orc_ctr_drbg*)calloc(1, sizeof(orc_ctr_drbg));
    d->reseed_counter = 1;
    drbg_update(d, entropy);
    *out = d;
    return ORC_OK;
}
void orc_ctr_drbg_destroy(orc_ctr_drbg* d) { free(d); }
void orc_ctr_drbg_state(const orc_ctr_drbg* d, uint8_t key[16], uint8_t nonce_be[16]) {
    memcpy(key, d->key, 16);
    memcpy(nonce_be, d->v, 16);
}
int orc_ctr_drbg_generate(orc_ctr_drbg* d, uint8_t* out, size_t count) { /* ctrDrbgGenerate, :68-83 */
    if (count > ((size_t)1 << 16)) return ORC_ERR_INVALID_ARGUMENT;
    drbg_ctr_encrypt(d, NULL, count, out);
    be128_add(d->v, (count + 15) / 16);
    uint8_t zeros[32] = {0};
    drbg_update(d, zeros);
    d->reseed_counter += 1;
    return ORC_OK;
}

/* PolyRq.random(context:using: NistAes128Ctr(seed:)) -- the `a` polynomial of a seeded ciphertext
 * (SerializedCiphertext.swift:53-58, Bfv+Encrypt.swift:155-156): 128 stream bits per coefficient, little-endian,
 * reduced mod q_i; the stream is the concatenation of generate(4096) calls (BufferedRng, bufferCount 4096). */
int orc_poly_random_from_seed(const orc_poly_context* ctx, const uint8_t seed[32], uint64_t* out) {
    orc_ctr_drbg* d = NULL;
    orc_ctr_drbg_create(seed, &d);
    const size_t n = (size_t)ctx->degree;
    uint8_t buffer[4096];
    size_t offset = sizeof(buffer);
    for (size_t r = 0; r < ctx->count; ++r)
        for (size_t k = 0; k < n; ++k) {
            if (offset == sizeof(buffer)) {
                orc_ctr_drbg_generate(d, buffer, sizeof(buffer));
                offset = 0;
            }
            u128 value = 0;
            for (int b = 15; b >= 0; --b) value = (value << 8) | buffer[offset + b];
            offset += 16;
            out[r * n + k] = reduce_u128(&ctx->reduce[r], value);
        }
    orc_ctr_drbg_destroy(d);
    return ORC_OK;
}
