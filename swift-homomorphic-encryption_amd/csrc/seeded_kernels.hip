// seeded_kernels.hip -- regenerates the uniform polynomial `a` of a seeded ciphertext on the device.
//
// Reference: Ciphertext(deserialize: .seeded(poly0:seed:)) (SerializedCiphertext.swift:53-58) and the encryptor
// (Bfv/Bfv+Encrypt.swift:155-156) draw `a` with PolyRq.random(context:using: NistAes128Ctr(seed:)):
//   NistAes128Ctr = BufferedRng<NistCtrDrbg>(bufferCount: 4096)      Random/NistAes128Ctr.swift, Random/BufferedRng.swift
//   NistCtrDrbg   = NIST SP 800-90A CTR_DRBG, AES-128, no derivation function, 128-bit counter
//                                                                    Random/NistCtrDrbg.swift:23-96
//   randomizeUniform: 128 stream bits per coefficient, little-endian, reduced mod q_i
//                                                                    PolyRq/PolyRq+Randomize.swift:56-75
// The stream is a chain of 4096-byte generate() calls, each followed by a re-key (update with zero input): chunk c of
// a seed (256 counter blocks = 256 coefficients) is AES-CTR under (key_c, V_c), and (key_c+1, V_c+1) are two more blocks
// under key_c.  Two kernels:
//   seeded_chain_kernel   8 lanes per seed walk the re-key chain (the only sequential part: one key expansion and two
//                         blocks per chunk) and record every chunk's round keys and counter, 192 bytes per chunk;
//   seeded_stream_kernel  one wavefront per (seed, chunk): the round keys arrive as wave-uniform scalar loads, every lane
//                         encrypts 4 counter blocks and reduces its 4 coefficients -- no lane repeats another's work.
// AES (FIPS-197) uses one T-table (the other three are byte rotations of it), held in LDS `kTableCopies` times
// interleaved so that lane l reads copy l mod kTableCopies: the data-dependent lookups of a wavefront then spread over
// the banks instead of colliding on the 256 words of a single copy.
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
// a lane's view of the interleaved LDS table: entry x of its copy
template <int COPIES>
struct TableView {
    const uint32_t* base;  // &table[lane % COPIES]
    __device__ __forceinline__ uint32_t operator[](uint32_t x) const { return base[x * COPIES]; }
};

template <typename Table>
__device__ __forceinline__ uint32_t sbox(const Table& te0, uint32_t x) { return (te0[x] >> 8) & 0xffu; }
template <typename Table>
__device__ __forceinline__ uint32_t sub_word(const Table& te0, uint32_t w) {
    return (sbox(te0, w >> 24) << 24) | (sbox(te0, (w >> 16) & 0xff) << 16) | (sbox(te0, (w >> 8) & 0xff) << 8) |
           sbox(te0, w & 0xff);
}

struct Block {
    uint32_t w[4];  // big-endian column words
};

template <typename Table>
__device__ __forceinline__ void expand_key(const Table& te0, const Block& key, uint32_t (&rk)[44]) {
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

template <typename Table, typename RoundKeys>
__device__ __forceinline__ Block encrypt(const Table& te0, const RoundKeys& rk, Block in) {
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

constexpr uint32_t kChunkBlocks = 256;  // BufferedRng's 4096-byte refill = 256 counter blocks = 256 coefficients
constexpr uint32_t kRecordWords = 48;   // per chunk: 44 round-key words + the 4 counter words

template <int COPIES, int THREADS>
__device__ __forceinline__ void fill_table(uint32_t* table) {
    for (uint32_t i = threadIdx.x; i < 256u * COPIES; i += THREADS) table[i] = g_aes_table.te0[i / COPIES];
    __syncthreads();
}

// ---- the re-key chain: 8 lanes per seed ----------------------------------------------------------------------------
// A chain step is one key expansion and two blocks under it, all sequential in time; what can be shared out is the
// width.  Lane (b, c) of a seed's 8 owns column c of block b (b = 0: the next key, b = 1: the next counter) and column c
// of the key schedule; a round fetches the other three columns of its quad with DPP quad permutes, and the expansion's
// running XOR across the four columns is four quad broadcasts.  A lane then issues about a fifth of the instructions
// the one-lane-per-seed form needs, and the chain is issue-latency bound.
template <int CTRL>
__device__ __forceinline__ uint32_t quad_permute(uint32_t x) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, 0xf, 0xf, false));
}
constexpr int quad_ctrl(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }

struct ColumnLane {
    uint32_t column;           // c
    uint32_t m1, m2, m3;       // all-ones when c >= 1, 2, 3
    __device__ __forceinline__ uint32_t pick(const Block& b) const {
        return column == 0 ? b.w[0] : column == 1 ? b.w[1] : column == 2 ? b.w[2] : b.w[3];
    }
};

// FIPS-197 5.2 with word 4 r + c of the schedule in lane c of the quad
template <typename Table>
__device__ __forceinline__ void expand_key_columns(const Table& te0, const ColumnLane& me, uint32_t key_column,
                                                   uint32_t (&rkc)[11]) {
    rkc[0] = key_column;
    uint32_t rcon = 0x01000000u;
#pragma unroll
    for (int r = 1; r <= 10; ++r) {
        const uint32_t w = rkc[r - 1];
        const uint32_t w0 = quad_permute<quad_ctrl(0, 0, 0, 0)>(w), w1 = quad_permute<quad_ctrl(1, 1, 1, 1)>(w),
                       w2 = quad_permute<quad_ctrl(2, 2, 2, 2)>(w), w3 = quad_permute<quad_ctrl(3, 3, 3, 3)>(w);
        const uint32_t t = sub_word(te0, (w3 << 8) | (w3 >> 24)) ^ rcon;
        rcon = (rcon << 1) ^ ((rcon & 0x80000000u) ? 0x1b000000u : 0u);
        rkc[r] = t ^ w0 ^ (w1 & me.m1) ^ (w2 & me.m2) ^ (w3 & me.m3);  // w'[c] = t ^ w[0] ^ ... ^ w[c]
    }
}

// one column of FIPS-197 5.1 per lane; `s` is column c of the input block
template <typename Table>
__device__ __forceinline__ uint32_t encrypt_column(const Table& te0, const uint32_t (&rkc)[11], uint32_t s) {
    s ^= rkc[0];
#pragma unroll
    for (int r = 1; r < 10; ++r) {
        const uint32_t s1 = quad_permute<quad_ctrl(1, 2, 3, 0)>(s), s2 = quad_permute<quad_ctrl(2, 3, 0, 1)>(s),
                       s3 = quad_permute<quad_ctrl(3, 0, 1, 2)>(s);
        s = te0[s >> 24] ^ rotr32(te0[(s1 >> 16) & 0xff], 8) ^ rotr32(te0[(s2 >> 8) & 0xff], 16) ^
            rotr32(te0[s3 & 0xff], 24) ^ rkc[r];
    }
    const uint32_t s1 = quad_permute<quad_ctrl(1, 2, 3, 0)>(s), s2 = quad_permute<quad_ctrl(2, 3, 0, 1)>(s),
                   s3 = quad_permute<quad_ctrl(3, 0, 1, 2)>(s);
    return ((sbox(te0, s >> 24) << 24) | (sbox(te0, (s1 >> 16) & 0xff) << 16) | (sbox(te0, (s2 >> 8) & 0xff) << 8) |
            sbox(te0, s3 & 0xff)) ^ rkc[10];
}

constexpr uint32_t kChainLanes = 8, kChainSeedsPerWave = 64 / kChainLanes;

// NistCtrDrbg.init(entropy:) and the chain of re-keys (NistCtrDrbg.swift:47-96)
__global__ void __launch_bounds__(64)
    seeded_chain_kernel(const uint8_t* __restrict__ seeds, uint32_t* __restrict__ chain, size_t batch, uint32_t chunks) {
    __shared__ uint32_t table[256];
    fill_table<1, 64>(table);
    const TableView<1> te0{table};
    const uint32_t lane = threadIdx.x, first_lane = lane & ~(kChainLanes - 1), block = (lane >> 2) & 1;
    ColumnLane me;
    me.column = lane & 3;
    me.m1 = me.column >= 1 ? ~0u : 0u;
    me.m2 = me.column >= 2 ? ~0u : 0u;
    me.m3 = me.column >= 3 ? ~0u : 0u;
    size_t seed_index = static_cast<size_t>(blockIdx.x) * kChainSeedsPerWave + (lane >> 3);
    const bool live = seed_index < batch;
    if (!live) seed_index = batch - 1;  // idle groups shadow the last seed: the lanes of a wavefront stay in step
    const uint8_t* seed = seeds + seed_index * 32;
    uint32_t rkc[11];
    // the two blocks E(V + 1), E(V + 2) under the current schedule; afterwards every lane holds the new key column and V
    uint32_t key_column = 0;
    Block v{{0, 0, 0, 0}};
    auto rekey = [&](uint32_t provided) {
        const uint32_t out = encrypt_column(te0, rkc, me.pick(counter_add(v, 1 + block))) ^ provided;
        key_column = __shfl(out, static_cast<int>(first_lane + me.column), 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) v.w[i] = __shfl(out, static_cast<int>(first_lane + 4 + i), 64);
    };
    // key = 0, V = 0, then update(entropy)                                               NistCtrDrbg.swift:47-66
    expand_key_columns(te0, me, key_column, rkc);
    rekey(load_be32(seed + 16 * block + 4 * me.column));
    uint32_t* record = chain + seed_index * chunks * kRecordWords;
    for (uint32_t chunk = 0; chunk < chunks; ++chunk, record += kRecordWords) {
        expand_key_columns(te0, me, key_column, rkc);
        if (live) {
            if (block == 0) {
#pragma unroll
                for (int r = 0; r <= 10; ++r) record[4 * r + me.column] = rkc[r];
            } else {
                record[44 + me.column] = me.pick(v);
            }
        }
        v = counter_add(v, kChunkBlocks);  // nonce += ceil(4096 / 16)                    NistCtrDrbg.swift:76
        rekey(0);                          // ctrDrbgUpdate(providedData: zeros)          NistCtrDrbg.swift:78-79
    }
}

constexpr int kTableCopies = 16;  // interleaved copies of the AES T-table in LDS (profiles/r02n_wire_format_table_copies.txt)
constexpr int kStreamWaves = 4;

// the 256 counter blocks of one (seed, chunk) per wavefront, 4 per lane
__global__ void __launch_bounds__(64 * kStreamWaves)
    seeded_stream_kernel(const uint32_t* __restrict__ chain, uint64_t* __restrict__ out, const DeviceContext ctx,
                         uint32_t chunks, size_t total_chunks) {
    __shared__ uint32_t table[256 * kTableCopies];
    fill_table<kTableCopies, 64 * kStreamWaves>(table);
    const uint32_t lane = threadIdx.x & 63;
    const TableView<kTableCopies> te0{table + (lane & (kTableCopies - 1))};
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t words = static_cast<size_t>(ctx.moduli_count) << ctx.log_degree;
    for (size_t item = static_cast<size_t>(blockIdx.x) * kStreamWaves + wave; item < total_chunks;
         item += static_cast<size_t>(gridDim.x) * kStreamWaves) {
        const size_t seed_index = item / chunks;
        const uint32_t chunk = static_cast<uint32_t>(item - seed_index * chunks);
        const uint32_t* record = chain + item * kRecordWords;  // wave-uniform: scalar loads
        uint32_t rk[44];
#pragma unroll
        for (int i = 0; i < 44; ++i) rk[i] = record[i];
        const Block v{{record[44], record[45], record[46], record[47]}};
        uint64_t* poly = out + seed_index * words;
        const size_t first = static_cast<size_t>(chunk) * kChunkBlocks + 4 * lane;
        Block stream[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) stream[j] = encrypt(te0, rk, counter_add(v, 1 + 4 * lane + j));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t word = first + j;
            if (word < words) {
                // UInt128(littleEndianBytes:) of the 16 stream bytes                      PolyRq+Randomize.swift:66-70
                U128 value;
                value.lo = static_cast<uint64_t>(__builtin_bswap32(stream[j].w[0])) |
                           (static_cast<uint64_t>(__builtin_bswap32(stream[j].w[1])) << 32);
                value.hi = static_cast<uint64_t>(__builtin_bswap32(stream[j].w[2])) |
                           (static_cast<uint64_t>(__builtin_bswap32(stream[j].w[3])) << 32);
                const DeviceModulus m = ctx.moduli[word >> ctx.log_degree];
                poly[word] = barrett_reduce128(value, m.p, m.barrett128_lo, m.barrett128_hi);
            }
        }
    }
}

}  // namespace

size_t seeded_uniform_scratch_bytes(const DeviceContext& ctx, size_t batch) {
    const size_t words = static_cast<size_t>(ctx.moduli_count) << ctx.log_degree;
    const size_t chunks = (words + kChunkBlocks - 1) / kChunkBlocks;
    return batch * chunks * kRecordWords * sizeof(uint32_t);
}

hipError_t launch_seeded_uniform(const uint8_t* seeds, uint64_t* out, const DeviceContext& ctx, size_t batch,
                                 void* scratch, hipStream_t stream) {
    if (batch == 0) return hipSuccess;
    const size_t words = static_cast<size_t>(ctx.moduli_count) << ctx.log_degree;
    const size_t chunks = (words + kChunkBlocks - 1) / kChunkBlocks;
    const size_t chain_blocks = (batch + kChainSeedsPerWave - 1) / kChainSeedsPerWave, total_chunks = batch * chunks;
    if (chain_blocks > 0x7fffffffull || chunks > 0xffffffffull) return hipErrorInvalidValue;
    uint32_t* chain = static_cast<uint32_t*>(scratch);
    hipLaunchKernelGGL(seeded_chain_kernel, dim3(static_cast<unsigned>(chain_blocks)), dim3(64), 0, stream, seeds, chain,
                       batch, static_cast<uint32_t>(chunks));
    hipError_t status = hipGetLastError();
    if (status != hipSuccess) return status;
    const size_t stream_blocks = (total_chunks + kStreamWaves - 1) / kStreamWaves, cap = 256 * 16;
    hipLaunchKernelGGL(seeded_stream_kernel, dim3(static_cast<unsigned>(stream_blocks < cap ? stream_blocks : cap)),
                       dim3(64 * kStreamWaves), 0, stream, chain, out, ctx, static_cast<uint32_t>(chunks), total_chunks);
    return hipGetLastError();
}

}  // namespace heamd
