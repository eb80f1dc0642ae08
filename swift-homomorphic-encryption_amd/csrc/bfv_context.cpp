// bfv_context.cpp -- builds Context<Bfv<UInt64>>: parameter checks, the ciphertext / key-switching PolyContexts and
// one _RnsTool per level, all host-side, then uploads the constant tables the BEHZ kernels read.
#include "bfv_context.hpp"

#include <cstring>

#include "../../include/he_amd.h"

namespace heamd {

namespace {

U64x2 shoup_pair(u64 multiplicand, u64 p) { return U64x2{multiplicand, shoup_factor(multiplicand, p)}; }

DeviceModulus barrett_constants(u64 p) {
    DeviceModulus m{};
    m.p = p;
    m.barrett64 = static_cast<u64>((static_cast<u128>(1) << 64) / p);
    u128 f128;
    if (is_power_of_two(p)) {
        const int lg = floor_log2(p);
        f128 = lg == 0 ? 0 : (static_cast<u128>(1) << (128 - lg));
    } else {
        f128 = ~static_cast<u128>(0) / p;
    }
    m.barrett128_lo = static_cast<u64>(f128);
    m.barrett128_hi = static_cast<u64>(f128 >> 64);
    const int bits = bit_length(p);
    m.product_factor = static_cast<u64>((static_cast<u128>(1) << (bits + 62)) / p);
    m.product_shift = static_cast<uint32_t>(bits >= 2 ? bits - 2 : 0);
    m.two64_mod_p = static_cast<u64>((static_cast<u128>(1) << 64) % p);
    m.two64_mod_p_shoup = shoup_factor(m.two64_mod_p, p);
    if (bits >= 34 && bits <= 61 && !is_power_of_two(p)) {
        m.wide_shift = static_cast<uint32_t>(bits - 1);
        m.wide_factor = static_cast<u64>((static_cast<u128>(1) << (64 + bits - 1)) / p);
    }
    return m;
}

// (prod of `moduli` except index `skip`) mod m      -- RnsBaseConverter.swift:41-54, CrtComposer.swift:35-40
u64 punctured_product(const u64* moduli, size_t count, size_t skip, u64 m) {
    u64 prod = 1 % m;
    for (size_t k = 0; k < count; ++k)
        if (moduli[k] != moduli[skip]) prod = mul_mod(prod, moduli[k] % m, m);
    return prod;
}

// Bump allocator over a host staging buffer mirrored 1:1 on the device.
class Arena {
  public:
    template <typename T>
    size_t reserve(size_t count) {
        offset_ = (offset_ + 15) & ~static_cast<size_t>(15);
        const size_t at = offset_;
        offset_ += count * sizeof(T);
        bytes_.resize(offset_);
        return at;
    }
    template <typename T>
    T* at(size_t offset) {
        return reinterpret_cast<T*>(bytes_.data() + offset);
    }
    size_t size() const { return bytes_.size(); }
    const char* data() const { return bytes_.data(); }

  private:
    std::vector<char> bytes_;
    size_t offset_ = 0;
};

}  // namespace

BfvContext::~BfvContext() {
    for (auto& level : tools_)
        if (level.device_block != nullptr) (void)hipFree(level.device_block);
}

int BfvContext::create(uint32_t degree, u64 t, const u64* q, uint32_t count, std::unique_ptr<BfvContext>& out,
                       bool host_only, int word_bits) {
    out.reset();
    if (count > 0 && q == nullptr) return HE_ERR_INVALID_ARGUMENT;
    if (word_bits != 32 && word_bits != 64) return HE_ERR_INVALID_ARGUMENT;
    const u64 max_modulus = word_bits == 32 ? ((static_cast<u64>(1) << 30) - 1) : kMaxModulus;
    const u64 gamma = word_bits == 32 ? ((static_cast<u64>(1) << 30) - 20405) : kGamma;   // MA/Scalar.swift:502-519
    const u64 mtilde = word_bits == 32 ? (static_cast<u64>(1) << 16) : kMTilde;           // MA/Scalar.swift:508-525
    // EncryptionParameters.init at securityLevel .unchecked (EncryptionParameters.swift:136-166)
    if (!is_power_of_two(degree)) return HE_ERR_INVALID_ENCRYPTION_PARAMETERS;
    if (count == 0 || count > 32) return HE_ERR_INVALID_ENCRYPTION_PARAMETERS;
    for (uint32_t i = 0; i < count; ++i)
        if (!(q[i] > t) || !is_ntt_modulus(q[i], degree)) return HE_ERR_INVALID_ENCRYPTION_PARAMETERS;
    for (uint32_t i = 0; i <= count; ++i) {
        const u64 m = i < count ? q[i] : t;
        if (!is_prime(m) || m < 1 || m > max_modulus || m == gamma || m == mtilde)
            return HE_ERR_INVALID_ENCRYPTION_PARAMETERS;
    }
    // Context.init (Context.swift:94-143)
    {   // secretKeyContext: validates the full chain (uniqueness etc.)
        std::unique_ptr<PolyContext> secret_key_context;
        const int status = PolyContext::create(degree, q, count, secret_key_context, true);
        if (status != HE_OK) return status;
    }
    std::unique_ptr<BfvContext> ctx(new BfvContext());
    ctx->degree_ = degree;
    ctx->t_ = t;
    ctx->gamma_ = gamma;
    ctx->mtilde_ = mtilde;
    ctx->word_bits_ = word_bits;
    ctx->host_only_ = host_only;
    ctx->coefficient_moduli_.assign(q, q + count);
    ctx->has_ks_ = count > 1;
    const uint32_t L = count > 1 ? count - 1 : count;  // Context.swift:102-107
    ctx->L_ = L;
    ctx->ciphertext_.resize(L + 1);
    ctx->key_switching_.resize(L + 1);
    ctx->tools_.resize(L + 1);
    for (uint32_t k = 1; k <= L; ++k) {
        const int status = PolyContext::create(degree, q, k, ctx->ciphertext_[k], host_only);
        if (status != HE_OK) return status;
    }
    if (ctx->has_ks_) {
        std::vector<u64> moduli(L + 1);
        for (uint32_t k = 1; k <= L; ++k) {  // Context.swift:114-127
            std::memcpy(moduli.data(), q, k * sizeof(u64));
            moduli[k] = q[count - 1];
            const int status = PolyContext::create(degree, moduli.data(), k + 1, ctx->key_switching_[k], host_only);
            if (status != HE_OK) return status;
            u64 max_lazy = ctx->key_switching_[k]->max_lazy_product_accumulation_count(k + 1);
            if (word_bits == 32) {  // T.DoubleWidth = UInt64 (PolyContext.swift:246-253)
                u64 q_max = 0;
                for (uint32_t i = 0; i <= k; ++i) q_max = moduli[i] > q_max ? moduli[i] : q_max;
                const u64 square = (q_max - 1) * (q_max - 1);
                max_lazy = square == 0 ? static_cast<u64>(INT32_MAX) : (~static_cast<u64>(0) - q_max) / square;
            }
            if (!(static_cast<u64>(k + 1) < max_lazy)) return HE_ERR_INVALID_ENCRYPTION_PARAMETERS;  // Context.swift:122-124
        }
    }
    {   // plaintextContext (Context.swift:128-130)
        std::unique_ptr<PolyContext> plaintext_context;
        const int status = PolyContext::create(degree, &t, 1, plaintext_context, true);
        if (status != HE_OK) return status;
    }
    {   // RnsToolContext.init (RnsTool.swift:28-45): L+1 ascending NTT-friendly 61-bit primes, then mTilde
        std::vector<int> bits(L + 1, word_bits - 3);  // T.bitWidth - 3 (RnsTool.swift:30-33)
        if (!generate_primes(bits, true, degree, ctx->bsk_mtilde_)) return HE_ERR_NOT_ENOUGH_PRIMES;
        ctx->bsk_mtilde_.push_back(mtilde);
        std::unique_ptr<PolyContext> check;
        const int status = PolyContext::create(degree, ctx->bsk_mtilde_.data(),
                                               static_cast<uint32_t>(ctx->bsk_mtilde_.size()), check, true);
        if (status != HE_OK) return status;
        // tGammaContext = [t, gamma] (RnsTool.swift:62-64) only needs to be constructible here
        const u64 t_gamma[2] = {t, gamma};
        const int tg = PolyContext::create(degree, t_gamma, 2, check, true);
        if (tg != HE_OK) return tg;
    }
    for (uint32_t k = L; k >= 1; --k) {  // Context.swift:136-141
        const int status = ctx->build_tool(k);
        if (status != HE_OK) return status;
    }
    out = std::move(ctx);
    return HE_OK;
}

// _RnsTool.init(from:to:rnsToolContext:) (RnsTool.swift:132-251) for the level with k ciphertext moduli.
int BfvContext::build_tool(uint32_t k) {
    RnsToolLevel& level = tools_[k];
    const u64* q = coefficient_moduli_.data();
    const size_t L = k;
    if (L + 2 > bsk_mtilde_.size()) return HE_ERR_INVALID_POLY_CONTEXT;
    level.ext_moduli.assign(bsk_mtilde_.begin(), bsk_mtilde_.begin() + L + 2);  // getContext(moduliCount: L+2), :185-186
    const u64* ext = level.ext_moduli.data();
    const u64* bsk = ext;  // L+1 entries; B = bsk[0..L), m_sk = bsk[L]
    const u64 m_sk = bsk[L];
    // The shared RnsToolContext holds ONE mSkContext, built from the top level's m_sk (RnsTool.swift:44-62): at every
    // level rnsConvertBtoMSk converts B to that prime and inverseBModMSk is (B mod top m_sk)^-1 taken mod this level's
    // m_sk (RnsTool.swift:240-250).  The two primes coincide only at the top level; below it this restates what the
    // reference computes word for word (as the oracle does), not the textbook conversion.
    const u64 top_m_sk = bsk_mtilde_[bsk_mtilde_.size() - 2];

    // qBskContext (RnsTool.swift:234-239): validates the appended moduli and uniqueness
    std::vector<u64> qbsk(q, q + L);
    qbsk.insert(qbsk.end(), bsk, bsk + L + 1);
    for (uint32_t prefix = static_cast<uint32_t>(L) + 1; prefix <= 2 * L + 1; ++prefix) {
        const int status = validate_poly_context_prefix(degree_, qbsk.data(), prefix, true);
        if (status != HE_OK) return status;
    }
    {
        const int status = PolyContext::create(degree_, qbsk.data(), static_cast<uint32_t>(qbsk.size()), level.qbsk,
                                               host_only_);
        if (status != HE_OK) return status;
    }

    Arena arena;
    const size_t o_q_moduli = arena.reserve<DeviceModulus>(L);
    const size_t o_ext_moduli = arena.reserve<DeviceModulus>(L + 2);
    const size_t o_lift_scale = arena.reserve<U64x2>(L);
    const size_t o_inv_punct_q = arena.reserve<U64x2>(L);
    const size_t o_q_to_ext = arena.reserve<u64>((L + 2) * L);
    const size_t o_q_mod_bsk = arena.reserve<U64x2>(L + 1);
    const size_t o_inv_mtilde = arena.reserve<U64x2>(L + 1);
    const size_t o_q_to_bsk_scaled = arena.reserve<u64>((L + 1) * L);
    const size_t o_q_mod_bsk_scaled = arena.reserve<U64x2>(L + 1);
    const size_t o_inv_q_bsk = arena.reserve<U64x2>(L + 1);
    const size_t o_inv_punct_b = arena.reserve<U64x2>(L);
    const size_t o_floor_scale_b = arena.reserve<U64x2>(L);
    const size_t o_b_to_msk = arena.reserve<u64>(L);
    const size_t o_b_to_q = arena.reserve<u64>(L * L);
    const size_t o_b_mod_q = arena.reserve<U64x2>(L);
    const size_t o_neg_b_mod_q = arena.reserve<U64x2>(L);
    const size_t o_scaled = arena.reserve<DeviceModulus>(2 * L + 1);
    const size_t o_sr_scale = arena.reserve<U64x2>(L);
    const size_t o_q_to_tg = arena.reserve<u64>(2 * L);
    const size_t o_tg_moduli = arena.reserve<DeviceModulus>(2);
    const size_t o_neg_inv_q_tg = arena.reserve<U64x2>(2);
    const size_t o_alpha_modulus = arena.reserve<DeviceModulus>(1);
    const size_t o_q_div_t = arena.reserve<U64x2>(L);

    for (size_t i = 0; i < L; ++i) arena.at<DeviceModulus>(o_q_moduli)[i] = barrett_constants(q[i]);
    for (size_t j = 0; j < L + 2; ++j) arena.at<DeviceModulus>(o_ext_moduli)[j] = barrett_constants(ext[j]);
    arena.at<DeviceModulus>(o_alpha_modulus)[0] = barrett_constants(top_m_sk);
    // the lift / floor kernels keep sums of unfolded residues mod a Bsk prime: 6 Bsk_j must stay below 2^63.  The
    // reference's Bsk primes sit just above 2^60 (2^28 for UInt32 contexts), RnsTool.swift:28-66, so this always holds.
    for (size_t j = 0; j <= L; ++j) {
        if (bsk[j] >= (static_cast<u64>(1) << 63) / 6) {
            set_last_error("Bsk prime too large for the unfolded BEHZ sums");
            return HE_ERR_UNSUPPORTED;
        }
    }
    for (size_t i = 0; i < L; ++i) {
        u64 inverse = 0;
        if (!inverse_mod(punctured_product(q, L, i, q[i]), q[i], inverse)) return HE_ERR_NOT_INVERTIBLE;
        arena.at<U64x2>(o_inv_punct_q)[i] = shoup_pair(inverse, q[i]);
        // poly * mTildeModQ then convertApproximateProducts (RnsTool.swift:313-316, RnsBaseConverter.swift:97-106):
        // two exact multiplications mod q_i == one by the product of the constants
        arena.at<U64x2>(o_lift_scale)[i] = shoup_pair(mul_mod(mtilde_ % q[i], inverse, q[i]), q[i]);
        for (size_t j = 0; j < L + 2; ++j)
            arena.at<u64>(o_q_to_ext)[j * L + i] = punctured_product(q, L, i, ext[j]);
    }
    for (size_t j = 0; j <= L; ++j) {
        const u64 q_mod = product_mod(q, L, bsk[j]);
        arena.at<U64x2>(o_q_mod_bsk)[j] = shoup_pair(q_mod, bsk[j]);
        u64 inverse = 0;
        if (!inverse_mod(mtilde_ % bsk[j], bsk[j], inverse)) return HE_ERR_NOT_INVERTIBLE;
        arena.at<U64x2>(o_inv_mtilde)[j] = shoup_pair(inverse, bsk[j]);
        arena.at<U64x2>(o_q_mod_bsk_scaled)[j] = shoup_pair(mul_mod(q_mod, inverse, bsk[j]), bsk[j]);
        for (size_t i = 0; i < L; ++i)
            arena.at<u64>(o_q_to_bsk_scaled)[j * L + i] = mul_mod(punctured_product(q, L, i, bsk[j]), inverse, bsk[j]);
        if (!inverse_mod(q_mod, bsk[j], inverse)) return HE_ERR_NOT_INVERTIBLE;
        arena.at<U64x2>(o_inv_q_bsk)[j] = shoup_pair(inverse, bsk[j]);
    }
    for (size_t i = 0; i < L; ++i) {
        u64 inverse = 0;
        if (!inverse_mod(punctured_product(bsk, L, i, bsk[i]), bsk[i], inverse)) return HE_ERR_NOT_INVERTIBLE;
        arena.at<U64x2>(o_inv_punct_b)[i] = shoup_pair(inverse, bsk[i]);
        arena.at<U64x2>(o_floor_scale_b)[i] =
            shoup_pair(mul_mod(arena.at<U64x2>(o_inv_q_bsk)[i].x, inverse, bsk[i]), bsk[i]);
        arena.at<u64>(o_b_to_msk)[i] = punctured_product(bsk, L, i, top_m_sk);
        for (size_t row = 0; row < L; ++row)
            arena.at<u64>(o_b_to_q)[row * L + i] = punctured_product(bsk, L, i, q[row]);
        const u64 b_mod_qi = product_mod(bsk, L, q[i]);
        arena.at<U64x2>(o_b_mod_q)[i] = shoup_pair(b_mod_qi, q[i]);
        arena.at<U64x2>(o_neg_b_mod_q)[i] = shoup_pair(neg_mod(b_mod_qi, q[i]), q[i]);
    }
    U64x2 neg_inv_q_mod_mtilde{}, inv_b_mod_msk{};
    {
        u64 inverse = 0;
        if (!inverse_mod(product_mod(q, L, mtilde_), mtilde_, inverse)) return HE_ERR_NOT_INVERTIBLE;
        neg_inv_q_mod_mtilde = shoup_pair(neg_mod(inverse, mtilde_), mtilde_);
        if (!inverse_mod(product_mod(bsk, L, top_m_sk), m_sk, inverse)) return HE_ERR_NOT_INVERTIBLE;
        inv_b_mod_msk = shoup_pair(inverse, m_sk);
    }
    // dropExtendedBase multiplies by t before the inverse NTT (Bfv+Multiply.swift:41-44); both are exact maps mod
    // each modulus, so t is folded into the inverse transform's final N^-1 constants instead.
    for (size_t r = 0; r < 2 * L + 1; ++r) {
        DeviceModulus m = level.qbsk->host_constants()[r];
        const u64 p = m.p;
        set_inverse_degree_constants(m, mul_mod(m.inv_degree, t_ % p, p), mul_mod(m.inv_degree_root, t_ % p, p));
        if (m.has_ntt) m.has_ntt = kNttScaledInverseDegree;  // the last inverse stage multiplies by t N^-1, not by N^-1
        arena.at<DeviceModulus>(o_scaled)[r] = m;
    }

    // scaleAndRound tables (RnsTool.swift:145-169): [t, gamma] is the output base of rnsConvertQToTGamma
    const u64 t_gamma[2] = {t_, gamma_};
    u64 inv_gamma_mod_t = 0;
    if (!inverse_mod(gamma_ % t_, t_, inv_gamma_mod_t)) return HE_ERR_NOT_INVERTIBLE;
    for (size_t j = 0; j < 2; ++j) {
        arena.at<DeviceModulus>(o_tg_moduli)[j] = barrett_constants(t_gamma[j]);
        u64 inverse = 0;
        if (!inverse_mod(product_mod(q, L, t_gamma[j]), t_gamma[j], inverse)) return HE_ERR_NOT_INVERTIBLE;
        arena.at<U64x2>(o_neg_inv_q_tg)[j] = shoup_pair(neg_mod(inverse, t_gamma[j]), t_gamma[j]);
        for (size_t i = 0; i < L; ++i) arena.at<u64>(o_q_to_tg)[j * L + i] = punctured_product(q, L, i, t_gamma[j]);
    }
    for (size_t i = 0; i < L; ++i) {
        const u64 gamma_t = mul_mod(gamma_ % q[i], t_ % q[i], q[i]);
        // poly *= prodGammaTModQ, then the converter's (Q/q_i)^-1: two exact products mod q_i = one
        arena.at<U64x2>(o_sr_scale)[i] = shoup_pair(mul_mod(gamma_t, arena.at<U64x2>(o_inv_punct_q)[i].x, q[i]), q[i]);
    }

    // plaintextTranslate (RnsTool.swift:167-182): Q = t floor(Q/t) + (Q mod t) and Q = 0 mod q_i, hence
    // floor(Q/t) = -(Q mod t) t^-1 (mod q_i) -- no composed Q needed (the oracle divides the composed integer instead)
    const u64 q_mod_t = product_mod(q, L, t_);
    for (size_t i = 0; i < L; ++i) {
        u64 inverse = 0;
        if (!inverse_mod(t_ % q[i], q[i], inverse)) return HE_ERR_NOT_INVERTIBLE;
        arena.at<U64x2>(o_q_div_t)[i] = shoup_pair(neg_mod(mul_mod(q_mod_t % q[i], inverse, q[i]), q[i]), q[i]);
    }

    RnsToolDevice& d = level.device;
    d.L = static_cast<uint32_t>(L);
    d.q_mod_t = q_mod_t;
    d.inv_gamma_mod_t = inv_gamma_mod_t;
    d.mtilde = mtilde_;
    {
        // the merged sum of a Q row is sum_i z_i (B/Bsk_i mod q) + alpha' (+-B mod q) with z_i < Bsk_i and
        // alpha' < m_sk: at most (L + 1) products of a Bsk prime by a residue mod q
        u64 bsk_max = 0, q_max = 0;
        for (size_t j = 0; j <= L; ++j) bsk_max = bsk[j] > bsk_max ? bsk[j] : bsk_max;
        for (size_t i = 0; i < L; ++i) q_max = q[i] > q_max ? q[i] : q_max;
        const unsigned __int128 worst = static_cast<unsigned __int128>(bsk_max - 1) * (q_max - 1);
        d.floor_merge_ok = (worst == 0 || (static_cast<unsigned __int128>(1) << 127) / worst > L + 1) ? 1u : 0u;
    }
    {
        // worst exact sums of the two kernels against the bound 2^(64 + wide_shift) of the modulus they are reduced by:
        // lift and approximateFloor: sum_i (q_i - 1) (ext_j - 1); the alpha sum: sum_j (Bsk_j - 1) (alpha modulus - 1);
        // the Q rows: sum_j (Bsk_j - 1) (q - 1) + (m_sk - 1) (q - 1)
        auto fits = [](unsigned __int128 worst, const DeviceModulus& m) {
            return m.wide_shift != 0 && (worst >> (64 + m.wide_shift)) == 0;
        };
        bool ok = true;
        unsigned __int128 q_sum = 0, bsk_sum = 0;
        for (size_t i = 0; i < L; ++i) q_sum += q[i] - 1;
        for (size_t j = 0; j < L; ++j) bsk_sum += bsk[j] - 1;
        // (the lift adds the mTilde correction's product, two residues mod Bsk_j, to its sums)
        for (size_t j = 0; j <= L; ++j)
            ok = ok && fits((q_sum + (bsk[j] - 1)) * (bsk[j] - 1), arena.at<DeviceModulus>(o_ext_moduli)[j]);
        ok = ok && fits(bsk_sum * (top_m_sk - 1), arena.at<DeviceModulus>(o_alpha_modulus)[0]);
        ok = ok && fits(bsk_sum * (m_sk - 1), arena.at<DeviceModulus>(o_ext_moduli)[L]);
        for (size_t i = 0; i < L; ++i)
            ok = ok && fits((bsk_sum + (m_sk - 1)) * (q[i] - 1), arena.at<DeviceModulus>(o_q_moduli)[i]);
        // ... and their cross columns cannot wrap (device_math.hpp product_sum_add_uniform_short): terms x (high word of
        // the largest residue + high word of the largest constant + 2) <= 2^32; the Q rows take one more term (alpha)
        // through the general multiply-add, which counts its carries
        auto high = [](u64 v) { return static_cast<unsigned __int128>(v >> 32) + 1; };
        u64 q_max = 0, ext_max = top_m_sk;
        for (size_t i = 0; i < L; ++i) q_max = q[i] > q_max ? q[i] : q_max;
        for (size_t j = 0; j <= L; ++j) ext_max = bsk[j] > ext_max ? bsk[j] : ext_max;
        ok = ok && (L + 1) * (high(q_max) + high(ext_max)) <= (static_cast<unsigned __int128>(1) << 32);
        d.wide_reduce_ok = ok ? 1u : 0u;
    }
    d.log_degree = static_cast<uint32_t>(floor_log2(degree_));
    d.neg_inv_q_mod_mtilde = neg_inv_q_mod_mtilde;
    d.inv_b_mod_msk = inv_b_mod_msk;
    d.alpha_modulus_is_msk = top_m_sk == m_sk ? 1u : 0u;
    if (host_only_) return HE_OK;

    HEAMD_HIP_TRY(hipMalloc(&level.device_block, arena.size()));
    HEAMD_HIP_TRY(hipMemcpy(level.device_block, arena.data(), arena.size(), hipMemcpyHostToDevice));
    const char* base = static_cast<const char*>(level.device_block);
    d.q_moduli = reinterpret_cast<const DeviceModulus*>(base + o_q_moduli);
    d.ext_moduli = reinterpret_cast<const DeviceModulus*>(base + o_ext_moduli);
    d.lift_scale = reinterpret_cast<const U64x2*>(base + o_lift_scale);
    d.inv_punctured_q = reinterpret_cast<const U64x2*>(base + o_inv_punct_q);
    d.q_to_ext = reinterpret_cast<const uint64_t*>(base + o_q_to_ext);
    d.q_mod_bsk = reinterpret_cast<const U64x2*>(base + o_q_mod_bsk);
    d.inv_mtilde_mod_bsk = reinterpret_cast<const U64x2*>(base + o_inv_mtilde);
    d.q_to_bsk_scaled = reinterpret_cast<const uint64_t*>(base + o_q_to_bsk_scaled);
    d.q_mod_bsk_scaled = reinterpret_cast<const U64x2*>(base + o_q_mod_bsk_scaled);
    d.inv_q_mod_bsk = reinterpret_cast<const U64x2*>(base + o_inv_q_bsk);
    d.inv_punctured_b = reinterpret_cast<const U64x2*>(base + o_inv_punct_b);
    d.floor_scale_b = reinterpret_cast<const U64x2*>(base + o_floor_scale_b);
    d.b_to_msk = reinterpret_cast<const uint64_t*>(base + o_b_to_msk);
    d.b_to_q = reinterpret_cast<const uint64_t*>(base + o_b_to_q);
    d.b_mod_q = reinterpret_cast<const U64x2*>(base + o_b_mod_q);
    d.neg_b_mod_q = reinterpret_cast<const U64x2*>(base + o_neg_b_mod_q);
    level.qbsk_moduli_scaled_by_t = reinterpret_cast<const DeviceModulus*>(base + o_scaled);
    d.scale_round_scale = reinterpret_cast<const U64x2*>(base + o_sr_scale);
    d.q_to_t_gamma = reinterpret_cast<const uint64_t*>(base + o_q_to_tg);
    d.t_gamma = reinterpret_cast<const DeviceModulus*>(base + o_tg_moduli);
    d.neg_inv_q_mod_t_gamma = reinterpret_cast<const U64x2*>(base + o_neg_inv_q_tg);
    d.alpha_modulus = reinterpret_cast<const DeviceModulus*>(base + o_alpha_modulus);
    d.q_div_t = reinterpret_cast<const U64x2*>(base + o_q_div_t);
    return HE_OK;
}

}  // namespace heamd
