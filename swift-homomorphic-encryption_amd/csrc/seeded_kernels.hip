// seeded_kernels.hip -- regenerates the uniform polynomial `a` of a seeded ciphertext on the device.
//
// Reference: Ciphertext(deserialize: .seeded(poly0:seed:)) (SerializedCiphertext.swift:53-58) and the encryptor
// (Bfv/Bfv+Encrypt.swift:155-156) draw `a` with PolyRq.random(context:using: NistAes128Ctr(seed:)):
//   NistAes128Ctr = BufferedRng<NistCtrDrbg>(bufferCount: 4096)      Random/NistAes128Ctr.swift, Random/BufferedRng.swift
//   NistCtrDrbg   = NIST SP 800-90A CTR_DRBG, AES-128, no derivation function, 128-bit counter
//                                                                    Random/NistCtrDrbg.swift:23-96
//   randomizeUniform: 128 stream bits per coefficient, little-endian, reduced mod q_i
//                                                                    PolyRq/PolyRq+Randomize.swift:56-75
// The stream is a chain of 4096-byte generate() calls, each followed by a re-key (update with zero input), so one
// seed is sequential at 4 KiB granularity: one wavefront owns one seed, its 64 lanes encrypt the 256 counter blocks
// of a chunk (4 each = 4 coefficients each) and every lane recomputes the two re-key blocks and the key schedule.
// AES (FIPS-197) uses one 1 KiB T-table in LDS (the other three are byte rotations of it).
#include <hip/hip_runtime.h>


#include "device_context.hpp"
#include "device_math.hpp"
#include "kernels.hpp"

namespace heamd {

namespace {

// Te0[x] = (2 S[x], S[x], S[x], 3 S[x]) as a big-endian word, FIPS-197 5.1.1 / 5.1.3: built at compile time, so the
// library has no table upload (nothing to synchronise inside a stream capture) and no mutable device state.
struct AesTable {
    uint32_t te0[256];
};
constexpr uint32_t gf256_mul(uint32_t a, uint32_t b) {
    uint32_t r = 0;
    while (b) {
        if (b & 1) r ^= a;
        a = ((a << 1) ^ ((a & 0x80) ? 0x1b : 0)) & 0xff;
        b >>= 1;
    }
    return r;
}
constexpr AesTable make_aes_table() {
    AesTable t{};
    for (uint32_t x = 0; x < 256; ++x) {
        // the inverse in GF(2^8) is x^254 (0 -> 0)
        uint32_t inverse = 1, power = x;
        for (int bit = 1; bit < 8; ++bit) {
            power = gf256_mul(power, power);  // x^(2^bit)
            inverse = gf256_mul(inverse, power);
        }
        uint32_t s = inverse;
        for (int k = 1; k <= 4; ++k) s ^= ((inverse << k) | (inverse >> (8 - k))) & 0xff;
        s ^= 0x63;
        t.te0[x] = (gf256_mul(s, 2) << 24) | (s << 16) | (s << 8) | gf256_mul(s, 3);
    }
    return t;
}
__device__ constexpr AesTable g_aes_table = make_aes_table();
static_assert(g_aes_table.te0[0] == 0xc66363a5u && g_aes_table.te0[1] == 0xf87c7c84u, "FIPS-197 S-box: S[0] = 0x63, S[1] = 0x7c");

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int k) { return (x >> k) | (x << (32 - k)); }
__device__ __forceinline__ uint32_t sbox(const uint32_t* te0, uint32_t x) { return (te0[x] >> 8) & 0xffu; }
__device__ __forceinline__ uint32_t sub_word(const uint32_t* te0, uint32_t w) {
    return (sbox(te0, w >> 24) << 24) | (sbox(te0, (w >> 16) & 0xff) << 16) | (sbox(te0, (w >> 8) & 0xff) << 8) |
           sbox(te0, w & 0xff);
}

struct Block {
    uint32_t w[4];  // big-endian column words
};

__device__ __forceinline__ void expand_key(const uint32_t* te0, const Block& key, uint32_t (&rk)[44]) {
    // FIPS-197 5.2
#pragma unroll
    for (int i = 0; i < 4; ++i) rk[i] = key.w[i];
    uint32_t rcon = 0x01000000u;
#pragma unroll
    for (int i = 4; i < 44; ++i) {
        uint32_t t = rk[i - 1];
        if (i % 4 == 0) {
            t = sub_word(te0, (t << 8) | (t >> 24)) ^ rcon;
            rcon = (rcon << 1) ^ ((rcon & 0x80000000u) ? 0x1b000000u : 0u);
        }
        rk[i] = rk[i - 4] ^ t;
    }
}

__device__ __forceinline__ Block encrypt(const uint32_t* te0, const uint32_t (&rk)[44], Block in) {
    uint32_t s0 = in.w[0] ^ rk[0], s1 = in.w[1] ^ rk[1], s2 = in.w[2] ^ rk[2], s3 = in.w[3] ^ rk[3];
#pragma unroll
    for (int r = 1; r < 10; ++r) {
        const uint32_t t0 = te0[s0 >> 24] ^ rotr32(te0[(s1 >> 16) & 0xff], 8) ^ rotr32(te0[(s2 >> 8) & 0xff], 16) ^
                            rotr32(te0[s3 & 0xff], 24) ^ rk[4 * r];
        const uint32_t t1 = te0[s1 >> 24] ^ rotr32(te0[(s2 >> 16) & 0xff], 8) ^ rotr32(te0[(s3 >> 8) & 0xff], 16) ^
                            rotr32(te0[s0 & 0xff], 24) ^ rk[4 * r + 1];
        const uint32_t t2 = te0[s2 >> 24] ^ rotr32(te0[(s3 >> 16) & 0xff], 8) ^ rotr32(te0[(s0 >> 8) & 0xff], 16) ^
                            rotr32(te0[s1 & 0xff], 24) ^ rk[4 * r + 2];
        const uint32_t t3 = te0[s3 >> 24] ^ rotr32(te0[(s0 >> 16) & 0xff], 8) ^ rotr32(te0[(s1 >> 8) & 0xff], 16) ^
                            rotr32(te0[s2 & 0xff], 24) ^ rk[4 * r + 3];
        s0 = t0, s1 = t1, s2 = t2, s3 = t3;
    }
    Block out;
    out.w[0] = ((sbox(te0, s0 >> 24) << 24) | (sbox(te0, (s1 >> 16) & 0xff) << 16) | (sbox(te0, (s2 >> 8) & 0xff) << 8) |
                sbox(te0, s3 & 0xff)) ^ rk[40];
    out.w[1] = ((sbox(te0, s1 >> 24) << 24) | (sbox(te0, (s2 >> 16) & 0xff) << 16) | (sbox(te0, (s3 >> 8) & 0xff) << 8) |
                sbox(te0, s0 & 0xff)) ^ rk[41];
    out.w[2] = ((sbox(te0, s2 >> 24) << 24) | (sbox(te0, (s3 >> 16) & 0xff) << 16) | (sbox(te0, (s0 >> 8) & 0xff) << 8) |
                sbox(te0, s1 & 0xff)) ^ rk[42];
    out.w[3] = ((sbox(te0, s3 >> 24) << 24) | (sbox(te0, (s0 >> 16) & 0xff) << 16) | (sbox(te0, (s1 >> 8) & 0xff) << 8) |
                sbox(te0, s2 & 0xff)) ^ rk[43];
    return out;
}

// 128-bit big-endian counter + small amount
__device__ __forceinline__ Block counter_add(Block v, uint32_t amount) {
    uint64_t sum = static_cast<uint64_t>(v.w[3]) + amount;
    v.w[3] = static_cast<uint32_t>(sum);
    sum = static_cast<uint64_t>(v.w[2]) + (sum >> 32);
    v.w[2] = static_cast<uint32_t>(sum);
    sum = static_cast<uint64_t>(v.w[1]) + (sum >> 32);
    v.w[1] = static_cast<uint32_t>(sum);
    v.w[0] += static_cast<uint32_t>(sum >> 32);
    return v;
}

__device__ __forceinline__ uint32_t load_be32(const uint8_t* p) {
    return (static_cast<uint32_t>(p[0]) << 24) | (static_cast<uint32_t>(p[1]) << 16) | (static_cast<uint32_t>(p[2]) << 8) |
           p[3];
}

constexpr int kSeedsPerBlock = 4;  // one wavefront per seed

__global__ void __launch_bounds__(64 * kSeedsPerBlock)
    seeded_uniform_kernel(const uint8_t* __restrict__ seeds, uint64_t* __restrict__ out, const DeviceContext ctx,
                          size_t batch) {
    __shared__ uint32_t te0[256];
    te0[threadIdx.x] = g_aes_table.te0[threadIdx.x];
    __syncthreads();
    const size_t seed_index = static_cast<size_t>(blockIdx.x) * kSeedsPerBlock + (threadIdx.x >> 6);
    if (seed_index >= batch) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint8_t* seed = seeds + seed_index * 32;
    const size_t words = static_cast<size_t>(ctx.moduli_count) << ctx.log_degree;
    uint64_t* poly = out + seed_index * words;

    // NistCtrDrbg.init(entropy:): key = 0, V = 0, then update(entropy)          NistCtrDrbg.swift:47-66
    Block key{{0, 0, 0, 0}}, v{{0, 0, 0, 0}};
    uint32_t rk[44];
    auto update = [&](const Block& provided_key, const Block& provided_v) {
        expand_key(te0, key, rk);
        const Block a = encrypt(te0, rk, counter_add(v, 1)), b = encrypt(te0, rk, counter_add(v, 2));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            key.w[i] = a.w[i] ^ provided_key.w[i];
            v.w[i] = b.w[i] ^ provided_v.w[i];
        }
    };
    {
        Block e0, e1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            e0.w[i] = load_be32(seed + 4 * i);
            e1.w[i] = load_be32(seed + 16 + 4 * i);
        }
        update(e0, e1);
    }
    const Block zero{{0, 0, 0, 0}};
    // BufferedRng refills of 4096 bytes = 256 counter blocks, each followed by the re-key of ctrDrbgGenerate
    for (size_t chunk_first = 0; chunk_first < words; chunk_first += 256) {
        expand_key(te0, key, rk);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t word = chunk_first + 4 * lane + j;
            const Block stream = encrypt(te0, rk, counter_add(v, 1 + 4 * lane + j));
            if (word < words) {
                // UInt128(littleEndianBytes:) of the 16 stream bytes                      PolyRq+Randomize.swift:66-70
                U128 value;
                value.lo = static_cast<uint64_t>(__builtin_bswap32(stream.w[0])) |
                           (static_cast<uint64_t>(__builtin_bswap32(stream.w[1])) << 32);
                value.hi = static_cast<uint64_t>(__builtin_bswap32(stream.w[2])) |
                           (static_cast<uint64_t>(__builtin_bswap32(stream.w[3])) << 32);
                const DeviceModulus m = ctx.moduli[word >> ctx.log_degree];
                poly[word] = barrett_reduce128(value, m.p, m.barrett128_lo, m.barrett128_hi);
            }
        }
        v = counter_add(v, 256);  // nonce += ceil(4096 / 16)                             NistCtrDrbg.swift:76
        update(zero, zero);       // ctrDrbgUpdate(providedData: zeros)                   NistCtrDrbg.swift:78-79
    }
}

}  // namespace

hipError_t launch_seeded_uniform(const uint8_t* seeds, uint64_t* out, const DeviceContext& ctx, size_t batch,
                                 hipStream_t stream) {
    if (batch == 0) return hipSuccess;
    const size_t blocks = (batch + kSeedsPerBlock - 1) / kSeedsPerBlock;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(seeded_uniform_kernel, dim3(static_cast<unsigned>(blocks)), dim3(64 * kSeedsPerBlock), 0, stream,
                       seeds, out, ctx, batch);
    return hipGetLastError();
}

}  // namespace heamd
