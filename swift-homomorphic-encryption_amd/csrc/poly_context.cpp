// poly_context.cpp -- builds the immutable per-(N, moduli) precomputation on the host and uploads it once.
#include "poly_context.hpp"

#include <cstring>

#include "../../include/he_amd.h"

namespace heamd {

namespace {
thread_local std::string g_last_error;
}

void set_last_error(const std::string& message) { g_last_error = message; }
const char* last_error() { return g_last_error.c_str(); }

int device_failure(hipError_t error, const char* where) {
    set_last_error(std::string("HIP error in ") + where + ": " + hipGetErrorString(error));
    return HE_ERR_DEVICE;
}

// PolyContext.swift:49-62 validate(modulus:)
static int validate_modulus(u64 modulus) {
    if (!(is_prime(modulus) || is_power_of_two(modulus))) return HE_ERR_INVALID_MODULUS;
    if (modulus < 1 || modulus > kMaxModulus) return HE_ERR_INVALID_MODULUS;
    return HE_OK;
}

// PolyContext.swift:45-93: degree, power-of-two count, uniqueness, emptiness, then modulus validation (only the last
// modulus when a next context exists).
int validate_poly_context_prefix(uint32_t degree, const uint64_t* moduli, uint32_t count, bool has_next) {
    if (!is_power_of_two(degree)) return HE_ERR_INVALID_DEGREE;
    uint32_t powers_of_two = 0;
    for (uint32_t i = 0; i < count; ++i) powers_of_two += is_power_of_two(moduli[i]) ? 1 : 0;
    if (powers_of_two > 1) return HE_ERR_COPRIME_MODULI;
    for (uint32_t i = 0; i < count; ++i)
        for (uint32_t j = i + 1; j < count; ++j)
            if (moduli[i] == moduli[j]) return HE_ERR_COPRIME_MODULI;
    if (count == 0) return HE_ERR_EMPTY_MODULUS;
    if (has_next) return validate_modulus(moduli[count - 1]);
    for (uint32_t i = 0; i < count; ++i) {
        const int status = validate_modulus(moduli[i]);
        if (status != HE_OK) return status;
    }
    return HE_OK;
}

static DeviceModulus make_constants(u64 p) {
    DeviceModulus m{};
    m.p = p;
    m.barrett64 = static_cast<u64>((static_cast<u128>(1) << 64) / p);  // Modulus.swift (MA):206-209
    u128 f128;
    if (is_power_of_two(p)) {
        const int lg = floor_log2(p);
        f128 = lg == 0 ? 0 : (static_cast<u128>(1) << (128 - lg));  // Modulus.swift (MA):227-228
    } else {
        f128 = ~static_cast<u128>(0) / p;  // Modulus.swift (MA):230-231
    }
    m.barrett128_lo = static_cast<u64>(f128);
    m.barrett128_hi = static_cast<u64>(f128 >> 64);
    const int bits = bit_length(p);
    m.product_factor = static_cast<u64>((static_cast<u128>(1) << (bits + 62)) / p);  // Modulus.swift (MA):235-240
    m.product_shift = static_cast<uint32_t>(bits >= 2 ? bits - 2 : 0);
    m.two64_mod_p = static_cast<u64>((static_cast<u128>(1) << 64) % p);
    m.two64_mod_p_shoup = shoup_factor(m.two64_mod_p, p);
    if (bits >= 34 && bits <= 61 && !is_power_of_two(p)) {
        m.wide_shift = static_cast<uint32_t>(bits - 1);
        m.wide_factor = static_cast<u64>((static_cast<u128>(1) << (64 + bits - 1)) / p);
    }
    // p = 2^bits - delta with delta < 2^(bits - 32): the shift-folded product (device_math.hpp fold_mul) of any 64-bit word by a
    // constant below p is r = (V mod 2^(bits+2)) + (V >> (bits+2)) 4 delta with V < 2^(bits+33), so
    // r <= 2^(bits+2) - 1 + (2^31 - 1) 4 delta < 6p  <=>  delta (2^33 + 2) < 2^(bits+1), which delta < 2^(bits-32) gives for
    // every bits < 64 (tests/test_fold_product_bounds.py holds the corners).  Rounds 3-5 asked for delta < 2^(bits-33), the
    // bound of the shifted quotient factors this flag was introduced for; the folded product never needed it, and at
    // N = 16384 only two of the four standard 55-bit primes met it (NTT-friendly primes are 2N apart).
    if (bits >= 41 && bits <= 55 && ((static_cast<u64>(1) << bits) - p) < (static_cast<u64>(1) << (bits - 32)))
        m.split_shift = static_cast<uint32_t>(bits - 31);
    return m;
}

void set_inverse_degree_constants(DeviceModulus& m, u64 inverse_degree, u64 inverse_degree_root) {
    m.inv_degree = inverse_degree;
    m.inv_degree_shoup = shoup_factor(inverse_degree, m.p);
    m.inv_degree_root = inverse_degree_root;
    m.inv_degree_root_shoup = shoup_factor(inverse_degree_root, m.p);
    m.inv_degree_split = split_shifted(inverse_degree, m.p);
    m.inv_degree_factors = split_factors(inverse_degree, m.p);
    m.inv_degree_root_split = split_shifted(inverse_degree_root, m.p);
    m.inv_degree_root_factors = split_factors(inverse_degree_root, m.p);
}

int PolyContext::create(uint32_t degree, const uint64_t* moduli, uint32_t count, std::unique_ptr<PolyContext>& out,
                        bool host_only) {
    out.reset();
    if (count > 0 && moduli == nullptr) return HE_ERR_INVALID_ARGUMENT;
    if (count <= 1) {
        const int status = validate_poly_context_prefix(degree, moduli, count, false);
        if (status != HE_OK) return status;
    } else {
        for (uint32_t k = 1; k <= count; ++k) {
            const int status = validate_poly_context_prefix(degree, moduli, k, k > 1);
            if (status != HE_OK) return status;
        }
    }
    std::unique_ptr<PolyContext> ctx(new PolyContext());
    ctx->degree_ = degree;
    ctx->log_degree_ = static_cast<uint32_t>(floor_log2(degree));
    ctx->moduli_.assign(moduli, moduli + count);
    const size_t n = degree;
    ctx->host_moduli_.resize(count);
    ctx->host_forward_.assign(static_cast<size_t>(count) * n, U64x2{0, 0});
    ctx->host_inverse_.assign(static_cast<size_t>(count) * n, U64x2{0, 0});
    ctx->host_inverse_q_last_.assign(static_cast<size_t>(count) * count, U64x2{0, 0});

    for (uint32_t i = 0; i < count; ++i) {
        const u64 p = moduli[i];
        DeviceModulus m = make_constants(p);
        // inverseQLast of the chain element whose last modulus is q_i (PolyContext.swift:108-111)
        for (uint32_t j = 0; j < i; ++j) {
            u64 inverse = 0;
            if (!inverse_mod(p % moduli[j], moduli[j], inverse)) return HE_ERR_NOT_INVERTIBLE;
            ctx->host_inverse_q_last_[static_cast<size_t>(i) * count + j] = U64x2{inverse, shoup_factor(inverse, moduli[j])};
        }
        // _NttContext.init (PolyRq+Ntt.swift:118-169) when q_i is an NTT modulus (PolyContext.swift:117-121)
        if (!is_power_of_two(p) && is_ntt_modulus(p, degree) && degree >= 1) {
            const u64 psi = min_primitive_root_of_unity(p, 2 * static_cast<u64>(degree));
            if (psi == 0) return HE_ERR_INVALID_NTT_MODULUS;
            u64 inverse_psi = 0;
            if (!inverse_mod(psi, p, inverse_psi)) return HE_ERR_NOT_INVERTIBLE;
            U64x2* forward = ctx->host_forward_.data() + static_cast<size_t>(i) * n;
            U64x2* inverse = ctx->host_inverse_.data() + static_cast<size_t>(i) * n;
            std::vector<u64> inverse_powers(n, 1);
            // rootOfUnityPowers[bitrev(k)] = psi^k  (PolyRq+Ntt.swift:125-137)
            u64 power = 1, inverse_power = 1;
            forward[0] = U64x2{1, shoup_factor(1 % p, p)};
            for (uint32_t k = 1; k < degree; ++k) {
                power = mul_mod(power, psi, p);
                inverse_power = mul_mod(inverse_power, inverse_psi, p);
                const uint32_t rev = reverse_bits(k, static_cast<int>(ctx->log_degree_));
                forward[rev] = U64x2{power, shoup_factor(power, p)};
                inverse_powers[rev] = inverse_power;
            }
            // stage-major re-ordering of the inverse powers (PolyRq+Ntt.swift:146-157)
            size_t slot = 1;
            inverse[0] = U64x2{1, shoup_factor(1 % p, p)};
            for (int lg = static_cast<int>(ctx->log_degree_) - 1; lg >= 0; --lg) {
                const size_t group = static_cast<size_t>(1) << lg;
                for (size_t k = 0; k < group; ++k, ++slot) {
                    const u64 w = inverse_powers[group + k];
                    inverse[slot] = U64x2{w, shoup_factor(w, p)};
                }
            }
            u64 inverse_degree = 0;
            if (!inverse_mod(degree % p, p, inverse_degree)) return HE_ERR_NOT_INVERTIBLE;
            const u64 inverse_degree_root = mul_mod(inverse_degree, inverse[n - 1].x, p);  // PolyRq+Ntt.swift:162-168
            m.has_ntt = kNttPlainInverseDegree;
            set_inverse_degree_constants(m, inverse_degree, inverse_degree_root);
        }
        ctx->host_moduli_[i] = m;
    }
    ctx->dev_.degree = degree;
    ctx->dev_.log_degree = ctx->log_degree_;
    ctx->dev_.moduli_count = count;
    ctx->dev_.moduli_stride = count;
    if (!host_only) {
        const int status = ctx->upload();
        if (status != HE_OK) return status;
    }
    out = std::move(ctx);
    return HE_OK;
}

int PolyContext::upload() {
    HEAMD_HIP_TRY(hipGetDevice(&device_));
    const size_t count = moduli_.size(), n = degree_;
    auto round_up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    const size_t bytes_moduli = round_up(count * sizeof(DeviceModulus));
    const size_t bytes_twiddles = round_up(count * n * sizeof(U64x2));
    const size_t bytes_inverse_q_last = round_up(count * count * sizeof(U64x2));
    // limb-wise Shoup form of both twiddle tables: pairs (16 B) and quotient factors (8 B) per entry
    const size_t bytes_factors = round_up(count * n * sizeof(u64));
    // lane-major stage blocks for the tiled kernels with 8 words per lane (ntt_kernels.hip kLaneMajorTwiddles): N = 4096, 8192;
    // N = 16384 / 32768: for the 8192-point transforms of the interleaved sub-rows (ntt_forward_interleaved /
    // ntt_inverse_interleaved), which index the first 8192 entries of the forward table and the last 8192 of the inverse one
    // exactly as an N = 8192 transform indexes its own
    const bool sub_rows = log_degree_ == 14 || log_degree_ == 15;
    const bool lane_major = log_degree_ == 12 || log_degree_ == 13 || sub_rows;
    const size_t total = bytes_moduli + (lane_major ? 9 : 5) * bytes_twiddles + bytes_inverse_q_last + (lane_major ? 5 : 2) * bytes_factors;
    HEAMD_HIP_TRY(hipMalloc(&device_block_, total));
    char* base = static_cast<char*>(device_block_);
    HEAMD_HIP_TRY(hipMemcpy(base, host_moduli_.data(), count * sizeof(DeviceModulus), hipMemcpyHostToDevice));
    char* forward = base + bytes_moduli;
    char* inverse = forward + bytes_twiddles;
    char* inverse_q_last = inverse + bytes_twiddles;
    HEAMD_HIP_TRY(hipMemcpy(forward, host_forward_.data(), count * n * sizeof(U64x2), hipMemcpyHostToDevice));
    HEAMD_HIP_TRY(hipMemcpy(inverse, host_inverse_.data(), count * n * sizeof(U64x2), hipMemcpyHostToDevice));
    HEAMD_HIP_TRY(hipMemcpy(inverse_q_last, host_inverse_q_last_.data(), count * count * sizeof(U64x2),
                            hipMemcpyHostToDevice));
    char* forward_pairs = inverse_q_last + bytes_inverse_q_last;
    char* inverse_pairs = forward_pairs + bytes_twiddles;
    char* forward_factors = inverse_pairs + bytes_twiddles;
    char* inverse_factors = forward_factors + bytes_factors;
    char* inverse_pairs_signed = inverse_factors + bytes_factors;
    // lane-major copies: pairs forward | inverse | inverse signed | inverse, top partition; factors forward | inverse | inverse, top
    char* lanes_pairs = inverse_pairs_signed + bytes_twiddles;
    char* lanes_factors = lanes_pairs + 4 * bytes_twiddles;
    {
        // Every stage's block lane-major (ntt_common.hpp Twiddles::lanes): the stage on element bit b lies in a pass whose highest
        // bit is top(b); its lanes hold 2^g twiddles each, g = top(b) - b, and entry o of its block moves to
        // (o mod 2^g) (m / 2^g) + (o >> g).  Forward block of bit b: [2^s, 2^(s+1)), s = logN - 1 - b, m = 2^s; inverse block:
        // [N - 2m + 1, N - m], m = N >> (b + 1).  Partitions (8 words per lane = passes of three bits): partial pass on the low
        // bits (every forward kernel, the fused inverse ones; N = 8192: 0 | 3-1 | 6-4 | 9-7 | 12-10, N = 4096: four full passes) or
        // on the top bit (the plain-slab inverse at N = 8192: 2-0 | 5-3 | 8-6 | 11-9 | 12).
        // (interleaved sub-rows: the blocks of a 13-bit transform, found `region` entries into the table -- 0 for the forward
        // table, N - 8192 for the inverse one; the entries outside them, which the cross stages read, stay where they are)
        const int logn = sub_rows ? 13 : static_cast<int>(log_degree_);
        const size_t sub_n = size_t(1) << logn;
        const int partial = logn % 3;  // width of the partial pass (0: none)
        auto top_low = [&](int b) { return b < partial ? partial - 1 : partial + ((b - partial) / 3) * 3 + 2; };
        auto top_top = [&](int b) { const int t = (b / 3) * 3 + 2; return t < logn ? t : logn - 1; };
        auto lane_major_copy = [&](bool inverse_direction, bool top_partition, const auto& source, auto& out) {
            out = source;
            for (int b = 0; b < logn; ++b) {
                const int g = (top_partition ? top_top(b) : top_low(b)) - b;
                if (g == 0) continue;
                const size_t m = inverse_direction ? (sub_n >> (b + 1)) : (size_t(1) << (logn - 1 - b));
                const size_t region = inverse_direction ? n - sub_n : 0;
                const size_t start = region + (inverse_direction ? sub_n - 2 * m + 1 : m);
                for (size_t i = 0; i < count; ++i)
                    for (size_t o = 0; o < m; ++o)
                        out[i * n + start + (o & ((size_t(1) << g) - 1)) * (m >> g) + (o >> g)] = source[i * n + start + o];
            }
        };
        std::vector<U64x2> pairs(count * n), moved_pairs;
        std::vector<u64> factors(count * n), moved_factors;
        for (int direction = 0; direction < 2; ++direction) {
            const std::vector<U64x2>& table = direction == 0 ? host_forward_ : host_inverse_;
            for (size_t i = 0; i < count; ++i) {
                const u64 p = moduli_[i];
                for (size_t k = 0; k < n; ++k) {
                    const u64 w = table[i * n + k].x;
                    pairs[i * n + k] = U64x2{w, split_shifted(w, p)};
                    factors[i * n + k] = split_factors(w, p);
                }
            }
            HEAMD_HIP_TRY(hipMemcpy(direction == 0 ? forward_pairs : inverse_pairs, pairs.data(),
                                    count * n * sizeof(U64x2), hipMemcpyHostToDevice));
            HEAMD_HIP_TRY(hipMemcpy(direction == 0 ? forward_factors : inverse_factors, factors.data(),
                                    count * n * sizeof(u64), hipMemcpyHostToDevice));
            if (lane_major) {
                lane_major_copy(direction == 1, false, pairs, moved_pairs);
                lane_major_copy(direction == 1, false, factors, moved_factors);
                HEAMD_HIP_TRY(hipMemcpy(lanes_pairs + direction * bytes_twiddles, moved_pairs.data(), count * n * sizeof(U64x2),
                                        hipMemcpyHostToDevice));
                HEAMD_HIP_TRY(hipMemcpy(lanes_factors + direction * bytes_factors, moved_factors.data(), count * n * sizeof(u64),
                                        hipMemcpyHostToDevice));
                if (direction == 1 && !sub_rows) {
                    lane_major_copy(true, true, pairs, moved_pairs);
                    lane_major_copy(true, true, factors, moved_factors);
                    HEAMD_HIP_TRY(hipMemcpy(lanes_pairs + 3 * bytes_twiddles, moved_pairs.data(), count * n * sizeof(U64x2),
                                            hipMemcpyHostToDevice));
                    HEAMD_HIP_TRY(hipMemcpy(lanes_factors + 2 * bytes_factors, moved_factors.data(), count * n * sizeof(u64),
                                            hipMemcpyHostToDevice));
                }
            }
        }
        // the inverse table once more with w 2^32 mod p in signed limbs, t0s + t1' 2^32 with t1' = t1 + (t0 >> 31): the
        // butterflies of kModeSplitSigned multiply a signed difference (device_math.hpp split_mul_signed); `pairs` still
        // holds the inverse direction
        for (U64x2& pair : pairs) pair.y += (pair.y & 0x80000000ull) << 1;
        HEAMD_HIP_TRY(hipMemcpy(inverse_pairs_signed, pairs.data(), count * n * sizeof(U64x2), hipMemcpyHostToDevice));
        if (lane_major) {
            lane_major_copy(true, false, pairs, moved_pairs);
            HEAMD_HIP_TRY(hipMemcpy(lanes_pairs + 2 * bytes_twiddles, moved_pairs.data(), count * n * sizeof(U64x2),
                                    hipMemcpyHostToDevice));
        }
    }
    dev_.forward_split_pairs = reinterpret_cast<const U64x2*>(forward_pairs);
    dev_.inverse_split_pairs = reinterpret_cast<const U64x2*>(inverse_pairs);
    dev_.inverse_split_pairs_signed = reinterpret_cast<const U64x2*>(inverse_pairs_signed);
    auto pairs_at = [&](int k) { return lane_major ? reinterpret_cast<const U64x2*>(lanes_pairs + k * bytes_twiddles) : nullptr; };
    auto factors_at = [&](int k) { return lane_major ? reinterpret_cast<const u64*>(lanes_factors + k * bytes_factors) : nullptr; };
    dev_.forward_split_pairs_lanes = pairs_at(0);
    dev_.inverse_split_pairs_lanes = pairs_at(1);
    dev_.inverse_split_pairs_signed_lanes = pairs_at(2);
    dev_.inverse_split_pairs_lanes_top = pairs_at(3);
    dev_.forward_split_factors_lanes = factors_at(0);
    dev_.inverse_split_factors_lanes = factors_at(1);
    dev_.inverse_split_factors_lanes_top = factors_at(2);
    dev_.forward_split_factors = reinterpret_cast<const u64*>(forward_factors);
    dev_.inverse_split_factors = reinterpret_cast<const u64*>(inverse_factors);
    dev_.moduli = reinterpret_cast<const DeviceModulus*>(base);
    dev_.forward_twiddles = reinterpret_cast<const U64x2*>(forward);
    dev_.inverse_twiddles = reinterpret_cast<const U64x2*>(inverse);
    dev_.inverse_q_last = reinterpret_cast<const U64x2*>(inverse_q_last);
    dev_.degree = degree_;
    dev_.log_degree = log_degree_;
    dev_.moduli_count = static_cast<uint32_t>(count);
    dev_.moduli_stride = static_cast<uint32_t>(count);
    return HE_OK;
}

PolyContext::~PolyContext() {
    if (device_block_ != nullptr) (void)hipFree(device_block_);
    if (device_block32_ != nullptr) (void)hipFree(device_block32_);
}

int PolyContext::device_context32(uint32_t count, DeviceContext32& out) const {
    for (uint32_t i = 0; i < count; ++i)
        if (moduli_[i] > ((static_cast<u64>(1) << 30) - 1)) {
            set_last_error("modulus " + std::to_string(moduli_[i]) + " does not fit UInt32 (max 2^30 - 1)");
            return HE_ERR_INVALID_MODULUS;
        }
    const int status = check_device();
    if (status != HE_OK) return status;
    std::lock_guard<std::mutex> guard(word32_lock_);
    if (device_block32_ == nullptr) {
        const size_t total = moduli_.size() * degree_;
        std::vector<U32x2> forward(total), inverse(total);
        for (size_t k = 0; k < total; ++k) {
            // floor(floor(w 2^64 / p) / 2^32) = floor(w 2^32 / p)
            forward[k] = U32x2{static_cast<uint32_t>(host_forward_[k].x), static_cast<uint32_t>(host_forward_[k].y >> 32)};
            inverse[k] = U32x2{static_cast<uint32_t>(host_inverse_[k].x), static_cast<uint32_t>(host_inverse_[k].y >> 32)};
        }
        void* block = nullptr;
        HEAMD_HIP_TRY(hipMalloc(&block, 2 * total * sizeof(U32x2) + 16));
        HEAMD_HIP_TRY(hipMemcpy(block, forward.data(), total * sizeof(U32x2), hipMemcpyHostToDevice));
        HEAMD_HIP_TRY(hipMemcpy(static_cast<char*>(block) + total * sizeof(U32x2), inverse.data(), total * sizeof(U32x2),
                                hipMemcpyHostToDevice));
        device_block32_ = block;
        dev32_.moduli = dev_.moduli;
        dev32_.forward_twiddles = static_cast<const U32x2*>(block);
        dev32_.inverse_twiddles = static_cast<const U32x2*>(block) + total;
        dev32_.inverse_q_last = dev_.inverse_q_last;
        dev32_.degree = degree_;
        dev32_.log_degree = log_degree_;
        dev32_.moduli_stride = static_cast<uint32_t>(moduli_.size());
    }
    out = dev32_;
    out.moduli_count = count;
    return HE_OK;
}

bool PolyContext::all_ntt(uint32_t count) const {
    for (uint32_t i = 0; i < count && i < host_moduli_.size(); ++i)
        if (!host_moduli_[i].has_ntt) return false;
    return true;
}

int PolyContext::modulus_index(u64 modulus) const {
    for (size_t i = 0; i < moduli_.size(); ++i)
        if (moduli_[i] == modulus) return static_cast<int>(i);
    return -1;
}

u64 PolyContext::max_lazy_product_accumulation_count(uint32_t count) const {
    u64 q_max = 0;
    for (uint32_t i = 0; i < count; ++i) q_max = moduli_[i] > q_max ? moduli_[i] : q_max;
    const u128 max_product = static_cast<u128>(q_max - 1) * (q_max - 1);
    if (max_product == 0) return static_cast<u64>(INT64_MAX);
    const u128 result = (~static_cast<u128>(0) - q_max) / max_product;
    return result > static_cast<u128>(INT64_MAX) ? static_cast<u64>(INT64_MAX) : static_cast<u64>(result);
}

DeviceContext PolyContext::device_context(uint32_t count) const {
    DeviceContext d = dev_;
    d.moduli_count = count;
    uint32_t approx = 1, headroom = 1, prefix = 0, shift = 1, shift_prefix = 0;
    for (uint32_t i = 0; i < count; ++i) {
        if (moduli_[i] >= (static_cast<u64>(1) << 61)) approx = 0;
        if (moduli_[i] >= (static_cast<u64>(1) << 55) || moduli_[i] < (static_cast<u64>(1) << 40)) headroom = 0;
        if (headroom != 0) prefix = i + 1;
        if (headroom == 0 || host_moduli_[i].split_shift == 0) shift = 0;
        if (shift != 0) shift_prefix = i + 1;
    }
    d.approx_ok = approx;
    d.headroom_ok = headroom;
    d.headroom_prefix = prefix;
    d.shift_prefix = shift_prefix;
    d.fold_minus_mask = d.fold_plus_mask = 0;
    for (uint32_t i = 0; i < count && i < 64; ++i) {
        const u64 p = moduli_[i];
        const int bits = 64 - __builtin_clzll(p);
        if (bits >= 56 && bits <= 60 && (static_cast<u64>(1) << bits) - p < (static_cast<u64>(1) << (bits - 33)))
            d.fold_minus_mask |= static_cast<u64>(1) << i;
        if (bits == 61 && p - (static_cast<u64>(1) << 60) < (static_cast<u64>(1) << 24))
            d.fold_plus_mask |= static_cast<u64>(1) << i;
    }
    return d;
}

int PolyContext::check_device() const {
    if (device_block_ == nullptr) {
        set_last_error("context was created host-only (no device tables); compute entry points need a GPU context");
        return HE_ERR_DEVICE;
    }
    int current = -1;
    HEAMD_HIP_TRY(hipGetDevice(&current));
    if (current != device_) {
        set_last_error("context lives on HIP device " + std::to_string(device_) + " but device " +
                       std::to_string(current) + " is current");
        return HE_ERR_DEVICE;
    }
    return HE_OK;
}

}  // namespace heamd
