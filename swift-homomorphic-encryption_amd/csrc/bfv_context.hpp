// bfv_context.hpp -- host-side mirror of Context<Bfv<UInt64>> (reference Sources/HomomorphicEncryption/
// Context.swift:94-159) with one _RnsTool per ciphertext level (RnsTool.swift:74-251), and their device images.
#pragma once

#include <memory>
#include <vector>

#include "device_context.hpp"
#include "host_math.hpp"
#include "poly_context.hpp"

namespace heamd {

// Device image of one _RnsTool (input base Q = q_0..q_{L-1}, extended base [Bsk_0..Bsk_L, mTilde]).
// All pointers index one device block owned by the BfvContext.  (w, w') pairs are Shoup constants.
struct RnsToolDevice {
    uint32_t L;                        // input moduli
    uint32_t log_degree;
    const DeviceModulus* q_moduli;     // [L]     Barrett constants of q_i
    const DeviceModulus* ext_moduli;   // [L+2]   Barrett constants of Bsk_0..Bsk_L and the (L+2)'th extended modulus
    const U64x2* lift_scale;           // [L]     (mTilde * (Q/q_i)^-1) mod q_i           RnsTool.swift:313-316
    const U64x2* inv_punctured_q;      // [L]     (Q/q_i)^-1 mod q_i                      CrtComposer.swift:32-50
    const uint64_t* q_to_ext;          // [L+2][L] (Q/q_i) mod ext_j                      RnsBaseConverter.swift:41-54
    const U64x2* q_mod_bsk;            // [L+1]   Q mod Bsk_j                             RnsTool.swift:224-227
    const U64x2* inv_mtilde_mod_bsk;   // [L+1]   mTilde^-1 mod Bsk_j                     RnsTool.swift:228-231
    // liftQToQBsk ends with "* mTilde^-1 mod Bsk_j" (RnsTool.swift:364); that exact product is folded into the two
    // constants it distributes over: out_j = sum_i y_i ((Q/q_i) mTilde^-1) + r_centred (Q mTilde^-1)   (mod Bsk_j)
    const uint64_t* q_to_bsk_scaled;   // [L+1][L] (Q/q_i) mTilde^-1 mod Bsk_j
    const U64x2* q_mod_bsk_scaled;     // [L+1]   Q mTilde^-1 mod Bsk_j
    const U64x2* inv_q_mod_bsk;        // [L+1]   Q^-1 mod Bsk_j                          RnsTool.swift:241-245
    const U64x2* inv_punctured_b;      // [L]     (B/Bsk_i)^-1 mod Bsk_i
    const U64x2* floor_scale_b;        // [L]     Q^-1 (B/Bsk_i)^-1 mod Bsk_i: approximateFloor's Q^-1 and the Bsk -> Q
                                       //         converter's first product are consecutive exact products mod Bsk_i
    const uint64_t* b_to_msk;          // [L]     (B/Bsk_i) mod the TOP level's m_sk (the shared mSkContext, RnsTool.swift:44-62)
    const DeviceModulus* alpha_modulus;//         Barrett constants of that prime
    uint32_t alpha_modulus_is_msk;     //         1 at the top level, where it is this level's m_sk = ext_moduli[L]
    const uint64_t* b_to_q;            // [L][L]  (B/Bsk_k) mod q_i  (row i, column k)
    const U64x2* b_mod_q;              // [L]     B mod q_i                               RnsTool.swift:211-216
    const U64x2* neg_b_mod_q;          // [L]     -B mod q_i                              RnsTool.swift:217-223
    // scaleAndRound (RnsTool.swift:272-302): base conversion Q -> [t, gamma]
    const U64x2* scale_round_scale;    // [L]     (gamma t mod q_i) (Q/q_i)^-1 mod q_i            RnsTool.swift:149, :97-106
    const uint64_t* q_to_t_gamma;      // [2][L]  (Q/q_i) mod t, (Q/q_i) mod gamma
    const DeviceModulus* t_gamma;      // [2]     Barrett constants of t and gamma
    const U64x2* neg_inv_q_mod_t_gamma;// [2]     -(Q^-1) mod t, mod gamma                          RnsTool.swift:157-160
    uint64_t inv_gamma_mod_t;          //         gamma^-1 mod t                                    RnsTool.swift:150-153
    // plaintextTranslate (Bfv+Encrypt.swift:75-140)
    const U64x2* q_div_t;              // [L]     floor(Q / t) mod q_i                              RnsTool.swift:170-182
    uint64_t q_mod_t;                  //         Q mod t                                           RnsTool.swift:167
    uint64_t mtilde;                   //         T.mTilde: 2^32 (UInt64) or 2^16 (UInt32)          MA/Scalar.swift:508-525
    uint32_t wide_reduce_ok;           //         every dot product of lift / floor stays below 2^(64 + wide_shift) of its
                                       //         modulus (DeviceModulus::wide_shift != 0 for all of them): the kernels take
                                       //         the one-word-quotient Barrett (device_math.hpp reduce_product_sum_bounded)
    uint32_t floor_merge_ok;           //         (L + 1) (Bsk_max - 1) (q_max - 1) < 2^127: the alpha correction of the Bsk -> Q
                                       //         conversion may join that row's product sum (one reduction for both)
    U64x2 neg_inv_q_mod_mtilde;        //         -(Q^-1) mod mTilde                      RnsTool.swift:163-169
    U64x2 inv_b_mod_msk;               //         (B mod top m_sk)^-1 mod this level's m_sk  RnsTool.swift:246-250
};

struct RnsToolLevel {
    std::vector<u64> ext_moduli;              // L+2 entries
    std::unique_ptr<PolyContext> qbsk;        // [q_0..q_{L-1}, Bsk_0..Bsk_L]  (RnsTool.swift:235-239)
    RnsToolDevice device{};
    const DeviceModulus* qbsk_moduli_scaled_by_t = nullptr;  // qbsk constants with N^-1 replaced by t * N^-1
    void* device_block = nullptr;
};

class BfvContext {
  public:
    // word_bits = 64: Context<Bfv<UInt64>>; 32: Context<Bfv<UInt32>> (moduli <= 2^30 - 1, gamma = 2^30 - 20405,
    // mTilde = 2^16, 29-bit Bsk primes; MA/Scalar.swift:498-511, RnsTool.swift:30-33) on the same 8-byte words
    static int create(uint32_t degree, u64 plaintext_modulus, const u64* coefficient_moduli, uint32_t count,
                      std::unique_ptr<BfvContext>& out, bool host_only, int word_bits = 64);
    ~BfvContext();

    uint32_t degree() const { return degree_; }
    u64 plaintext_modulus() const { return t_; }
    uint32_t top_level() const { return L_; }
    bool has_key_switching() const { return has_ks_; }
    const std::vector<u64>& bsk_mtilde() const { return bsk_mtilde_; }
    // k = number of ciphertext moduli, 1..L
    const PolyContext* ciphertext(uint32_t k) const { return valid(k) ? ciphertext_[k].get() : nullptr; }
    const PolyContext* key_switching(uint32_t k) const { return valid(k) ? key_switching_[k].get() : nullptr; }
    const RnsToolLevel* tool(uint32_t k) const { return valid(k) ? &tools_[k] : nullptr; }
    bool valid(uint32_t k) const { return k >= 1 && k <= L_; }
    bool host_only() const { return host_only_; }
    int word_bits() const { return word_bits_; }  // 64: Context<Bfv<UInt64>>, 32: Context<Bfv<UInt32>>

  private:
    BfvContext() = default;
    int build_tool(uint32_t k);

    uint32_t degree_ = 0, L_ = 0;
    u64 t_ = 0;
    u64 gamma_ = kGamma, mtilde_ = kMTilde;
    int word_bits_ = 64;
    bool has_ks_ = false, host_only_ = false;
    std::vector<u64> coefficient_moduli_;
    std::vector<u64> bsk_mtilde_;  // Bsk_0..Bsk_L, mTilde  (RnsTool.swift:28-37)
    std::vector<std::unique_ptr<PolyContext>> ciphertext_, key_switching_;
    std::vector<RnsToolLevel> tools_;
};

}  // namespace heamd
