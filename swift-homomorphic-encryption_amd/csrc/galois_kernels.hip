// galois_kernels.hip -- index-permuting PolyRq kernels and the plaintext <-> ciphertext-ring lifts for gfx950.
//
// Reference semantics (Sources/HomomorphicEncryption/):
//   PolyRq<Coeff>.applyGalois   f(x) -> f(x^g): coefficient i goes to (i g) mod N, negated when floor(i g / N) is odd
//                               PolyRq/Galois.swift:115-143 (GaloisCoeffIterator :34-48)
//   PolyRq<Eval>.applyGalois    out[i] = in[bitrev(((g bitrev_{logN+1}(i + N)) >> 1) mod N)]
//                               PolyRq/Galois.swift:153-168 (GaloisEvalIterator :81-92)
//   PolyRq<Coeff>.multiplyPowerOfX   f(x) x^k mod (x^N + 1)          PolyRq/PolyRq.swift:398-422
//   Plaintext.convertToEvalFormat / convertToCoeffFormat             Plaintext.swift:149-191
//
// All are pure data movement (one read, one write per word, no multiplies): writes are coalesced -- each lane owns
// consecutive OUTPUT words and gathers its inputs -- because a permutation's scattered side costs less as reads,
// which L2 absorbs (a residue row is 64 KiB).  in and out must not alias.
#include <hip/hip_runtime.h>

#include "device_context.hpp"
#include "device_math.hpp"
#include "kernels.hpp"

namespace heamd {

namespace {

constexpr unsigned kThreads = 256;

constexpr size_t kGridCap = (size_t(1) << 31) - 1;
// One workgroup per kThreads work items, up to the grid limit: the kernels keep their grid-stride loops for what lies beyond it,
// but a lane that walks many items serialises its loads -- divideAndRoundQLast at N = 16384, L = 6 ran at 0.66 of 8 TB/s on
// 256 x 8 workgroups and at 0.79 with one item per lane (profiles/r06y_exact_grids.txt)
inline unsigned grid_for(size_t work_items) {
    const size_t blocks = (work_items + kThreads - 1) / kThreads;
    const size_t cap = kGridCap;
    return static_cast<unsigned>(blocks < cap ? (blocks ? blocks : 1) : cap);
}
// ... except the gathers that measured faster on a few workgroups per CU walking their items (the Coeff-form automorphism:
// 0.67 against 0.57 of 8 TB/s; the plaintext unlift)
inline unsigned grid_capped(size_t work_items) {
    const unsigned blocks = grid_for(work_items);
    return blocks < 256u * 8u ? blocks : 256u * 8u;
}

// Output word j of a row takes input word src (sign flipped when `negate`): shared by the automorphism and by the
// multiplication by a power of x, which differ only in how the doubled index i in [0, 2N) is derived from j.
//   automorphism:  i = j * g^-1 mod 2N     (then i g = j or j + N mod 2N)
//   x^s:           i = (j - s) mod 2N
template <typename W>
__device__ __forceinline__ uint64_t signed_gather(const W* __restrict__ row, uint32_t doubled_index, uint32_t n,
                                                  uint64_t p) {
    const bool negate = doubled_index >= n;
    const uint64_t v = row[negate ? doubled_index - n : doubled_index];
    return negate ? neg_mod(v, p) : v;
}

enum class CoeffMap { Galois, PowerOfX };

template <CoeffMap MAP, typename W>
__global__ void __launch_bounds__(kThreads)
    coeff_permute_kernel(const W* __restrict__ in, W* __restrict__ out, const DeviceContext ctx, uint32_t parameter,
                         size_t words) {
    const uint32_t logn = ctx.log_degree, n = ctx.degree, mask2n = 2 * n - 1;
    for (size_t idx = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; idx < words;
         idx += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t row = idx >> logn;
        const uint32_t j = static_cast<uint32_t>(idx) & (n - 1);
        const uint64_t p = ctx.moduli[row % ctx.moduli_count].p;
        const uint32_t i = (MAP == CoeffMap::Galois ? j * parameter : j + 2 * n - parameter) & mask2n;
        out[idx] = static_cast<W>(signed_gather(in + (row << logn), i, n, p));
    }
}

__global__ void __launch_bounds__(kThreads)
    galois_eval_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, const DeviceContext ctx,
                       uint32_t element, size_t words) {
    const uint32_t logn = ctx.log_degree, n = ctx.degree;
    for (size_t idx = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; idx < words;
         idx += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t row = idx >> logn;
        const uint32_t i = static_cast<uint32_t>(idx) & (n - 1);
        const uint32_t reversed = __brev(i + n) >> (32 - (logn + 1));
        const uint32_t raw = ((element * reversed) >> 1) & (n - 1);
        const uint32_t source = logn == 0 ? 0 : __brev(raw) >> (32 - logn);
        out[idx] = in[(row << logn) + source];
    }
}

// plaintext [batch][N] (values < t) -> out [batch][L][N]: x < (t+1)/2 ? x : x + (q_i - t)      Plaintext.swift:157-167
template <typename W>
__global__ void __launch_bounds__(kThreads)
    plaintext_lift_kernel(const W* __restrict__ plaintext, W* __restrict__ out, const DeviceContext ctx, uint64_t t,
                          size_t words) {
    const uint32_t logn = ctx.log_degree, n = ctx.degree;
    const uint64_t threshold = (t + 1) >> 1;
    for (size_t idx = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; idx < words;
         idx += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t row = idx >> logn;
        const size_t poly = row / ctx.moduli_count;
        const uint32_t mi = static_cast<uint32_t>(row - poly * ctx.moduli_count);
        const uint64_t x = plaintext[(poly << logn) + (idx & (n - 1))];
        out[idx] = static_cast<W>(x < threshold ? x : x + (ctx.moduli[mi].p - t));
    }
}

// first residue rows, Coeff form, [batch][N] in place: x >= (t+1)/2 ? x - (q_0 - t) : x          Plaintext.swift:181-186
template <typename W>
__global__ void __launch_bounds__(kThreads)
    plaintext_unlift_kernel(W* __restrict__ rows, uint64_t q0, uint64_t t, size_t words) {
    const uint64_t threshold = (t + 1) >> 1, increment = q0 - t;
    for (size_t idx = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; idx < words;
         idx += static_cast<size_t>(gridDim.x) * kThreads) {
        const uint64_t x = rows[idx];
        rows[idx] = static_cast<W>(x >= threshold ? x - increment : x);
    }
}

// copies residue row 0 of every polynomial: in [batch][L][N] -> out [batch][N]
template <typename W>
__global__ void __launch_bounds__(kThreads)
    first_row_kernel(const W* __restrict__ in, W* __restrict__ out, uint32_t logn, uint32_t rows_per_poly, size_t words) {
    for (size_t idx = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; idx < words;
         idx += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t poly = idx >> logn;
        out[idx] = in[((poly * rows_per_poly) << logn) + (idx & ((size_t(1) << logn) - 1))];
    }
}

// ---- PirUtil.expand, one tree level (PirUtil.swift:204-236, 262-299) ---------------------------------------------
// parents, c1: [batch][2][L][N] Coeff (c1 = applyGalois(parent)); next: [2 batch][2][L][N] with the children interleaved:
//   next[2k]     = c1_k + parent_k                                     (p0)
//   next[2k + 1] = (parent_k - c1_k) * x^(-2^(logStep-1))               (p1), shift = that power mod 2N
__global__ void __launch_bounds__(kThreads)
    expand_step_kernel(const uint64_t* __restrict__ parents, const uint64_t* __restrict__ c1,
                       uint64_t* __restrict__ next, const DeviceContext ctx, uint32_t shift, size_t words) {
    const uint32_t logn = ctx.log_degree, n = ctx.degree, mask2n = 2 * n - 1;
    const size_t ct_words = (size_t(2) * ctx.moduli_count) << logn;
    for (size_t idx = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; idx < words;
         idx += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t ct = idx / ct_words, w = idx - ct * ct_words;
        const size_t row = w >> logn;
        const uint32_t j = static_cast<uint32_t>(w) & (n - 1);
        const uint64_t p = ctx.moduli[row % ctx.moduli_count].p;
        const size_t row_base = ct * ct_words + (row << logn);
        next[2 * ct * ct_words + w] = add_mod(c1[row_base + j], parents[row_base + j], p);
        const uint32_t i = (j + 2 * n - shift) & mask2n;
        const bool negate = i >= n;
        const uint32_t source = negate ? i - n : i;
        const uint64_t difference = sub_mod(parents[row_base + source], c1[row_base + source], p);
        next[(2 * ct + 1) * ct_words + w] = negate ? neg_mod(difference, p) : difference;
    }
}

// table entry k = (source ciphertext index, destination index << 1 | doubled): dst[destination] = src[source], times
// two when `doubled` (a leaf above its tree's height is emitted as ciphertext + ciphertext, PirUtil.swift:262-268)
// The same table serves `queries` independent expansions of one shape: query q reads src + q * src_stride ciphertexts
// and writes dst + q * dst_stride ciphertexts (strides in ciphertexts).
__global__ void __launch_bounds__(kThreads)
    expand_move_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, const uint32_t* __restrict__ table,
                       const DeviceContext ctx, size_t count, size_t src_stride, size_t dst_stride, size_t words) {
    const uint32_t logn = ctx.log_degree;
    const size_t ct_words = (size_t(2) * ctx.moduli_count) << logn;
    for (size_t idx = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; idx < words;
         idx += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t item = idx / ct_words, w = idx - item * ct_words;
        const size_t query = item / count, k = item - query * count;
        const uint32_t source = table[2 * k], packed = table[2 * k + 1];
        const uint64_t x = src[(query * src_stride + source) * ct_words + w];
        const uint64_t p = ctx.moduli[(w >> logn) % ctx.moduli_count].p;
        dst[(query * dst_stride + static_cast<size_t>(packed >> 1)) * ct_words + w] = (packed & 1u) ? add_mod(x, x, p) : x;
    }
}

// The same with a workgroup inside ONE ciphertext (2 L N words a multiple of 512): the table entry and the row's modulus are
// wave-uniform, no lane divides, 16 bytes per lane -- the two moves of an expansion to 320 ciphertexts (192 leaves out, 64 parents
// gathered: 54 + 17 us) 24 us shorter, the expansion 0.903 -> 0.879 ms (profiles/r06w_small_launches.txt)
__global__ void __launch_bounds__(kThreads)
    expand_move_pairs_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, const uint32_t* __restrict__ table,
                             const DeviceContext ctx, uint32_t count, size_t src_stride, size_t dst_stride,
                             uint32_t blocks_per_ct) {
    const uint32_t logn = ctx.log_degree;
    const size_t ct_words = (size_t(2) * ctx.moduli_count) << logn;
    const uint32_t item = blockIdx.x / blocks_per_ct, block = blockIdx.x - item * blocks_per_ct;  // uniform
    const uint32_t query = item / count, k = item - query * count;
    const uint32_t source = table[2 * k], packed = table[2 * k + 1];
    const size_t first_word = size_t(block) * (2 * kThreads);                                    // of this workgroup
    const uint64_t p = ctx.moduli[(first_word >> logn) % ctx.moduli_count].p;                    // (rows are >= 512 words or the
    const size_t w = first_word + 2 * threadIdx.x;                                               //  launcher took the other kernel)
    U64x2 x = *reinterpret_cast<const U64x2*>(src + (size_t(query) * src_stride + source) * ct_words + w);
    if ((packed & 1u) != 0) {
        x.x = add_mod(x.x, x.x, p);
        x.y = add_mod(x.y, x.y, p);
    }
    *reinterpret_cast<U64x2*>(dst + (size_t(query) * dst_stride + static_cast<size_t>(packed >> 1)) * ct_words + w) = x;
}

}  // namespace

hipError_t launch_expand_step(const uint64_t* parents, const uint64_t* c1, uint64_t* next, const DeviceContext& ctx,
                              uint32_t shift, size_t batch, hipStream_t stream) {
    const size_t words = (batch * 2 * ctx.moduli_count) << ctx.log_degree;
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL(expand_step_kernel, dim3(grid_for(words)), dim3(kThreads), 0, stream, parents, c1, next, ctx,
                       shift, words);
    return hipGetLastError();
}

constexpr bool kExpandMovePairs = true;
hipError_t launch_expand_move(const uint64_t* src, uint64_t* dst, const uint32_t* table, const DeviceContext& ctx,
                              size_t count, size_t queries, size_t src_stride, size_t dst_stride, hipStream_t stream) {
    const size_t words = (queries * count * 2 * ctx.moduli_count) << ctx.log_degree;
    if (words == 0) return hipSuccess;
    const size_t ct_words = (size_t(2) * ctx.moduli_count) << ctx.log_degree, blocks = words / (2 * kThreads);
    if (kExpandMovePairs && ctx.degree >= 2 * kThreads && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0 &&
        blocks < (size_t(1) << 31) && queries * count < (size_t(1) << 31)) {
        hipLaunchKernelGGL(expand_move_pairs_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, src, dst, table, ctx,
                           static_cast<uint32_t>(count), src_stride, dst_stride, static_cast<uint32_t>(ct_words / (2 * kThreads)));
        return hipGetLastError();
    }
    hipLaunchKernelGGL(expand_move_kernel, dim3(grid_for(words)), dim3(kThreads), 0, stream, src, dst, table, ctx, count,
                       src_stride, dst_stride, words);
    return hipGetLastError();
}

template <typename W>
hipError_t launch_galois_coeff(const W* in, W* out, const DeviceContext& ctx, uint32_t inverse_element, size_t rows,
                               hipStream_t stream) {
    const size_t words = rows << ctx.log_degree;
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL((coeff_permute_kernel<CoeffMap::Galois, W>), dim3(grid_capped(words)), dim3(kThreads), 0, stream, in,
                       out, ctx, inverse_element, words);
    return hipGetLastError();
}
template hipError_t launch_galois_coeff<uint64_t>(const uint64_t*, uint64_t*, const DeviceContext&, uint32_t, size_t,
                                                  hipStream_t);
template hipError_t launch_galois_coeff<uint32_t>(const uint32_t*, uint32_t*, const DeviceContext&, uint32_t, size_t,
                                                  hipStream_t);

hipError_t launch_multiply_power_of_x(const uint64_t* in, uint64_t* out, const DeviceContext& ctx, uint32_t shift,
                                      size_t rows, hipStream_t stream) {
    const size_t words = rows << ctx.log_degree;
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL((coeff_permute_kernel<CoeffMap::PowerOfX, uint64_t>), dim3(grid_for(words)), dim3(kThreads), 0,
                       stream, in, out, ctx, shift, words);
    return hipGetLastError();
}

hipError_t launch_galois_eval(const uint64_t* in, uint64_t* out, const DeviceContext& ctx, uint32_t element, size_t rows,
                              hipStream_t stream) {
    const size_t words = rows << ctx.log_degree;
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL(galois_eval_kernel, dim3(grid_for(words)), dim3(kThreads), 0, stream, in, out, ctx, element,
                       words);
    return hipGetLastError();
}

template <typename W>
hipError_t launch_plaintext_lift(const W* plaintext, W* out, const DeviceContext& ctx, uint64_t t, size_t batch,
                                 hipStream_t stream) {
    const size_t words = (batch * ctx.moduli_count) << ctx.log_degree;
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL(plaintext_lift_kernel<W>, dim3(grid_for(words)), dim3(kThreads), 0, stream, plaintext, out, ctx, t,
                       words);
    return hipGetLastError();
}

template <typename W>
hipError_t launch_plaintext_unlift(W* rows, uint64_t q0, uint64_t t, size_t words, hipStream_t stream) {
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL(plaintext_unlift_kernel<W>, dim3(grid_capped(words)), dim3(kThreads), 0, stream, rows, q0, t, words);
    return hipGetLastError();
}

template <typename W>
hipError_t launch_first_rows(const W* in, W* out, const DeviceContext& ctx, size_t batch, hipStream_t stream) {
    const size_t words = batch << ctx.log_degree;
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL(first_row_kernel<W>, dim3(grid_for(words)), dim3(kThreads), 0, stream, in, out, ctx.log_degree,
                       ctx.moduli_count, words);
    return hipGetLastError();
}
#define HEAMD_INSTANTIATE_PLAINTEXT(W)                                                                              \
    template hipError_t launch_plaintext_lift<W>(const W*, W*, const DeviceContext&, uint64_t, size_t, hipStream_t); \
    template hipError_t launch_plaintext_unlift<W>(W*, uint64_t, uint64_t, size_t, hipStream_t);                     \
    template hipError_t launch_first_rows<W>(const W*, W*, const DeviceContext&, size_t, hipStream_t);
HEAMD_INSTANTIATE_PLAINTEXT(uint64_t)
HEAMD_INSTANTIATE_PLAINTEXT(uint32_t)
#undef HEAMD_INSTANTIATE_PLAINTEXT

}  // namespace heamd

// ---- wire format: CoefficientPacking / PolyRq.serialize / load ----------------------------------
// Each residue row is a big-endian bit stream of N coefficients, `width[r]` = ceilLog2(q_r) - skipLSBs bits each (the
// value shifted right by skipLSBs), padded with zero bits to a whole byte; rows follow one another
// (CoefficientPacking.swift:169-213, PolyRq/PolyRq+Serialize.swift:69-87).
namespace heamd {

namespace {

__device__ __forceinline__ uint32_t row_of_byte(const SerializeLayout& layout, uint64_t byte_in_poly) {
    uint32_t r = 0;
    while (r + 1 < layout.rows && byte_in_poly >= layout.byte_offset[r + 1]) ++r;
    return r;
}

// one lane = one output byte
__global__ void __launch_bounds__(256)
    serialize_kernel(const uint64_t* __restrict__ slab, uint8_t* __restrict__ bytes, const SerializeLayout layout,
                     uint32_t logn, uint32_t skip, size_t total_bytes) {
    const uint64_t per_poly = layout.byte_offset[layout.rows];
    const uint32_t n = 1u << logn;
    for (size_t idx = blockIdx.x * size_t(256) + threadIdx.x; idx < total_bytes; idx += size_t(gridDim.x) * 256) {
        const size_t poly = idx / per_poly;
        const uint64_t in_poly = idx - poly * per_poly;
        const uint32_t r = row_of_byte(layout, in_poly);
        const uint32_t w = layout.width[r];
        const uint64_t bit = (in_poly - layout.byte_offset[r]) * 8;
        uint32_t k = static_cast<uint32_t>(bit / w), offset = static_cast<uint32_t>(bit - uint64_t(k) * w);
        const uint64_t* row = slab + ((poly * layout.rows + r) << logn);
        uint32_t byte = 0, needed = 8;
        while (needed > 0 && k < n) {
            const uint32_t available = w - offset;
            const uint32_t take = available < needed ? available : needed;
            const uint64_t value = row[k] >> skip;
            byte = (byte << take) | static_cast<uint32_t>((value >> (available - take)) & ((1u << take) - 1u));
            needed -= take;
            offset += take;
            if (offset == w) {
                offset = 0;
                ++k;
            }
        }
        bytes[idx] = static_cast<uint8_t>(byte << needed);  // zero padding after the last coefficient
    }
}

// one lane = one coefficient; bytes past the row's end read as zero (CoefficientPacking.swift:128-133)
__global__ void __launch_bounds__(256)
    deserialize_kernel(const uint8_t* __restrict__ bytes, uint64_t* __restrict__ slab, const SerializeLayout layout,
                       uint32_t logn, uint32_t skip, size_t bytes_per_poly, size_t total_words) {
    const uint32_t n = 1u << logn;
    for (size_t idx = blockIdx.x * size_t(256) + threadIdx.x; idx < total_words; idx += size_t(gridDim.x) * 256) {
        const size_t row_index = idx >> logn;
        const uint32_t k = static_cast<uint32_t>(idx) & (n - 1);
        const size_t poly = row_index / layout.rows;
        const uint32_t r = static_cast<uint32_t>(row_index - poly * layout.rows);
        const uint32_t w = layout.width[r];
        const uint64_t row_bytes = layout.byte_offset[r + 1] - layout.byte_offset[r];
        const uint8_t* row = bytes + poly * bytes_per_poly + layout.byte_offset[r];
        const uint64_t bit = uint64_t(k) * w;
        const uint64_t first = bit >> 3;
        const uint32_t offset = static_cast<uint32_t>(bit & 7);
        uint64_t window = 0;  // 8 bytes big-endian starting at `first`
#pragma unroll
        for (int b = 0; b < 8; ++b) window = (window << 8) | (first + b < row_bytes ? row[first + b] : 0);
        const uint64_t ninth = first + 8 < row_bytes ? row[first + 8] : 0;
        const uint64_t aligned = offset == 0 ? window : ((window << offset) | (ninth >> (8 - offset)));
        slab[idx] = (aligned >> (64 - w)) << skip;
    }
}

// ---- 8 bytes per lane (rows and records 8-byte aligned: degree a multiple of 64, aligned buffers) ------------------
// The packed row is a big-endian bit stream of `width`-bit fields; byte j of the stream is bits [8j, 8j + 8).
__device__ __forceinline__ uint64_t byte_swap64(uint64_t v) {
    return (static_cast<uint64_t>(__builtin_bswap32(static_cast<uint32_t>(v))) << 32) |
           __builtin_bswap32(static_cast<uint32_t>(v >> 32));
}

// one lane = 8 output bytes = stream bits [64 c, 64 c + 64) of one row
__global__ void __launch_bounds__(256)
    serialize_words_kernel(const uint64_t* __restrict__ slab, uint64_t* __restrict__ words, const SerializeLayout layout,
                           uint32_t logn, uint32_t skip, size_t total_words) {
    const uint64_t words_per_poly = layout.byte_offset[layout.rows] >> 3;
    const uint32_t n = 1u << logn;
    for (size_t idx = blockIdx.x * size_t(256) + threadIdx.x; idx < total_words; idx += size_t(gridDim.x) * 256) {
        const size_t poly = idx / words_per_poly;
        const uint64_t word_in_poly = idx - poly * words_per_poly;
        const uint32_t r = row_of_byte(layout, word_in_poly << 3);
        const uint32_t w = layout.width[r];
        const uint64_t bit = ((word_in_poly << 3) - layout.byte_offset[r]) << 3;
        uint32_t k = static_cast<uint32_t>(bit / w), offset = static_cast<uint32_t>(bit - uint64_t(k) * w);
        const uint64_t* row = slab + ((poly * layout.rows + r) << logn);
        const uint64_t field_mask = w == 64 ? ~uint64_t(0) : ((uint64_t(1) << w) - 1);
        uint64_t out = 0;
        uint32_t needed = 64;
        while (needed > 0 && k < n) {
            const uint32_t available = w - offset;
            const uint32_t take = available < needed ? available : needed;
            const uint64_t value = (row[k] >> skip) & field_mask;
            const uint64_t piece = (value >> (available - take)) & (take == 64 ? ~uint64_t(0) : ((uint64_t(1) << take) - 1));
            out = (take == 64 ? 0 : (out << take)) | piece;
            needed -= take;
            offset += take;
            if (offset == w) {
                offset = 0;
                ++k;
            }
        }
        out = needed == 64 ? 0 : (out << needed);  // zero padding after the last coefficient
        words[idx] = byte_swap64(out);
    }
}

// one lane = one coefficient, read from the two aligned 8-byte words that hold its field
__global__ void __launch_bounds__(256)
    deserialize_words_kernel(const uint64_t* __restrict__ words, uint64_t* __restrict__ slab, const SerializeLayout layout,
                             uint32_t logn, uint32_t skip, size_t words_per_poly, size_t total_coefficients) {
    const uint32_t n = 1u << logn;
    for (size_t idx = blockIdx.x * size_t(256) + threadIdx.x; idx < total_coefficients;
         idx += size_t(gridDim.x) * 256) {
        const size_t row_index = idx >> logn;
        const uint32_t k = static_cast<uint32_t>(idx) & (n - 1);
        const size_t poly = row_index / layout.rows;
        const uint32_t r = static_cast<uint32_t>(row_index - poly * layout.rows);
        const uint32_t w = layout.width[r];
        const uint64_t row_words = (layout.byte_offset[r + 1] - layout.byte_offset[r]) >> 3;
        const uint64_t* row = words + poly * words_per_poly + (layout.byte_offset[r] >> 3);
        const uint64_t bit = uint64_t(k) * w;
        const uint64_t first = bit >> 6;
        const uint32_t offset = static_cast<uint32_t>(bit & 63);
        const uint64_t high = byte_swap64(row[first]);
        const uint64_t low = first + 1 < row_words ? byte_swap64(row[first + 1]) : 0;  // past the row: zero bits
        const uint64_t aligned = offset == 0 ? high : ((high << offset) | (low >> (64 - offset)));
        slab[idx] = (aligned >> (64 - w)) << skip;
    }
}

// ---- one wavefront per 128 coefficients (degree a multiple of 128, 16-byte aligned buffers) -------------------------
// 128 fields of `w` bits are exactly 2 w stream words, so a row tiles into (128 coefficients <-> w 16-byte pairs) and a
// tile never shares a word with its neighbours.  The wavefront reads its side with one 16-byte access per lane, passes
// the words through a wave-private LDS tile and writes the other side 16 bytes per lane: both directions are coalesced
// in full lines, and the only division is a 32-bit one by the row's width.
constexpr uint32_t kTileCoefficients = 128;
constexpr uint32_t kTileWaves = 4;

__device__ __forceinline__ void wave_private_tile_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 64 stream bits starting inside field k at bit `offset` (from the field's top) of a tile of w-bit fields
__device__ __forceinline__ uint64_t gather_stream_word(const uint64_t* fields, uint32_t w, uint32_t& k, uint32_t& offset) {
    uint64_t out = 0;
    uint32_t needed = 64;
    while (needed > 0) {
        const uint32_t available = w - offset;
        const uint32_t take = available < needed ? available : needed;
        const uint64_t piece = (fields[k] >> (available - take)) & (take == 64 ? ~uint64_t(0) : ((uint64_t(1) << take) - 1));
        out = (take == 64 ? 0 : (out << take)) | piece;
        needed -= take;
        offset += take;
        if (offset == w) {
            offset = 0;
            ++k;
        }
    }
    return out;
}

// one workgroup per residue row: the width, the row's place in the record and each lane's place in a tile are fixed
__global__ void __launch_bounds__(kTileWaves * 64)
    serialize_tiles_kernel(const uint64_t* __restrict__ slab, uint8_t* __restrict__ bytes, const SerializeLayout layout,
                           uint32_t logn, uint32_t skip) {
    __shared__ uint64_t tiles[kTileWaves][kTileCoefficients];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint64_t* fields = tiles[wave];
    const size_t row_index = blockIdx.x, poly = row_index / layout.rows;
    const uint32_t r = static_cast<uint32_t>(row_index - poly * layout.rows);
    const uint32_t w = layout.width[r], tiles_per_row = 1u << (logn - 7);
    const uint64_t mask = w == 64 ? ~uint64_t(0) : ((uint64_t(1) << w) - 1);
    const U64x2* in = reinterpret_cast<const U64x2*>(slab + (row_index << logn)) + lane;
    U64x2* out = reinterpret_cast<U64x2*>(bytes + poly * layout.byte_offset[layout.rows] + layout.byte_offset[r]) + lane;
    // stream pair `lane` of a tile is bits [128 lane, 128 lane + 128): it starts in field k0 at bit offset0
    const uint32_t k0 = (128 * lane) / w, offset0 = 128 * lane - k0 * w;
    for (uint32_t t = wave; t < tiles_per_row; t += kTileWaves) {
        const U64x2 pair = stream_load(in + size_t(t) * (kTileCoefficients / 2));
        fields[2 * lane] = (pair.x >> skip) & mask;
        fields[2 * lane + 1] = (pair.y >> skip) & mask;
        wave_private_tile_fence();
        if (lane < w) {
            uint32_t k = k0, offset = offset0;
            U64x2 packed;
            packed.x = byte_swap64(gather_stream_word(fields, w, k, offset));
            packed.y = byte_swap64(gather_stream_word(fields, w, k, offset));
            stream_store(out + size_t(t) * w, packed);
        }
        wave_private_tile_fence();
    }
}

__global__ void __launch_bounds__(kTileWaves * 64)
    deserialize_tiles_kernel(const uint8_t* __restrict__ bytes, uint64_t* __restrict__ slab, const SerializeLayout layout,
                             uint32_t logn, uint32_t skip, size_t bytes_per_poly) {
    __shared__ uint64_t tiles[kTileWaves][kTileCoefficients + 2];  // the word after the last one may be read, never used
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint64_t* words = tiles[wave];
    const size_t row_index = blockIdx.x, poly = row_index / layout.rows;
    const uint32_t r = static_cast<uint32_t>(row_index - poly * layout.rows);
    const uint32_t w = layout.width[r], tiles_per_row = 1u << (logn - 7);
    const U64x2* in = reinterpret_cast<const U64x2*>(bytes + poly * bytes_per_poly + layout.byte_offset[r]) + lane;
    U64x2* out = reinterpret_cast<U64x2*>(slab + (row_index << logn)) + lane;
    // coefficients 2 lane and 2 lane + 1 of a tile: the stream word each starts in and its bit offset there
    const uint32_t bit_x = 2 * lane * w, bit_y = bit_x + w;
    const uint32_t first_x = bit_x >> 6, offset_x = bit_x & 63, first_y = bit_y >> 6, offset_y = bit_y & 63;
    for (uint32_t t = wave; t < tiles_per_row; t += kTileWaves) {
        if (lane < w) {
            const U64x2 pair = stream_load(in + size_t(t) * w);
            words[2 * lane] = byte_swap64(pair.x);
            words[2 * lane + 1] = byte_swap64(pair.y);
        }
        wave_private_tile_fence();
        U64x2 value;
        {
            const uint64_t high = words[first_x], low = words[first_x + 1];
            const uint64_t aligned = offset_x == 0 ? high : ((high << offset_x) | (low >> (64 - offset_x)));
            value.x = (aligned >> (64 - w)) << skip;
        }
        {
            const uint64_t high = words[first_y], low = words[first_y + 1];
            const uint64_t aligned = offset_y == 0 ? high : ((high << offset_y) | (low >> (64 - offset_y)));
            value.y = (aligned >> (64 - w)) << skip;
        }
        stream_store(out + size_t(t) * (kTileCoefficients / 2), value);
        wave_private_tile_fence();
    }
}

// ---- packed residue rows (kernels.hpp PackedLayout): one wavefront per 64 coefficients = `width` stream words ---------
__global__ void __launch_bounds__(kTileWaves * 64)
    pack_rows_kernel(const uint64_t* __restrict__ slab, uint64_t* __restrict__ packed, const PackedLayout layout,
                     uint32_t logn) {
    __shared__ uint64_t tiles[kTileWaves][64 + 2];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint64_t* fields = tiles[wave];
    const size_t row_index = blockIdx.x, poly = row_index / layout.rows;
    const uint32_t r = static_cast<uint32_t>(row_index - poly * layout.rows);
    const uint32_t w = layout.width[r], tiles_per_row = 1u << (logn - 6);
    const uint64_t* in = slab + (row_index << logn) + lane;
    uint64_t* out = packed + poly * layout.word_offset[layout.rows] + layout.word_offset[r] + lane;
    if (lane < 2) fields[64 + lane] = 0;  // read (and shifted out of range) by the last words of a tile
    // stream word `lane` of a tile is bits [64 lane, 64 lane + 64): it starts inside field k0, `consumed0` bits in
    const uint32_t k0 = (64 * lane) / w, consumed0 = 64 * lane - k0 * w;
    for (uint32_t t = wave; t < tiles_per_row; t += kTileWaves) {
        fields[lane] = stream_load(in + size_t(t) * 64);
        wave_private_tile_fence();
        if (lane < w) {
            uint32_t k = k0;
            uint64_t word = fields[k] >> consumed0;
            for (uint32_t filled = w - consumed0; filled < 64; filled += w) word |= fields[++k] << filled;
            stream_store(out + size_t(t) * w, word);
        }
        wave_private_tile_fence();
    }
}

}  // namespace

namespace {
// every row and the record itself start on an 8-byte boundary of an 8-byte aligned buffer
bool word_aligned(const SerializeLayout& layout, const void* bytes) {
    if ((reinterpret_cast<uintptr_t>(bytes) & 7) != 0) return false;
    for (uint32_t r = 0; r <= layout.rows; ++r)
        if ((layout.byte_offset[r] & 7) != 0) return false;
    return true;
}
// whole 128-coefficient tiles, every row, the record and both buffers on 16-byte boundaries
bool tile_aligned(const SerializeLayout& layout, const void* bytes, const void* slab, uint32_t log_degree) {
    if (log_degree < 7) return false;
    if (((reinterpret_cast<uintptr_t>(bytes) | reinterpret_cast<uintptr_t>(slab)) & 15) != 0) return false;
    for (uint32_t r = 0; r <= layout.rows; ++r)
        if ((layout.byte_offset[r] & 15) != 0) return false;
    for (uint32_t r = 0; r < layout.rows; ++r)
        if (layout.width[r] == 0 || layout.width[r] > 64) return false;
    return true;
}
}  // namespace

hipError_t launch_pack_rows(const uint64_t* slab, uint64_t* packed, const PackedLayout& layout, uint32_t log_degree,
                            size_t polys, hipStream_t stream) {
    const size_t rows = polys * layout.rows;
    if (rows == 0) return hipSuccess;
    if (log_degree < 6 || rows > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(static_cast<unsigned>(rows)), dim3(kTileWaves * 64), 0, stream, slab, packed,
                       layout, log_degree);
    return hipGetLastError();
}

hipError_t launch_serialize(const uint64_t* slab, uint8_t* bytes, const SerializeLayout& layout, uint32_t log_degree,
                            uint32_t skip_lsbs, size_t batch, hipStream_t stream) {
    const size_t total = batch * layout.byte_offset[layout.rows];
    if (total == 0) return hipSuccess;
    if (tile_aligned(layout, bytes, slab, log_degree) && batch * layout.rows <= 0x7fffffffull) {
        hipLaunchKernelGGL(serialize_tiles_kernel, dim3(static_cast<unsigned>(batch * layout.rows)), dim3(kTileWaves * 64), 0,
                           stream, slab, bytes, layout, log_degree, skip_lsbs);
        return hipGetLastError();
    }
    if (word_aligned(layout, bytes)) {
        hipLaunchKernelGGL(serialize_words_kernel, dim3(grid_for(total >> 3)), dim3(256), 0, stream, slab,
                           reinterpret_cast<uint64_t*>(bytes), layout, log_degree, skip_lsbs, total >> 3);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(serialize_kernel, dim3(grid_for(total)), dim3(256), 0, stream, slab, bytes, layout, log_degree,
                       skip_lsbs, total);
    return hipGetLastError();
}

hipError_t launch_deserialize(const uint8_t* bytes, uint64_t* slab, const SerializeLayout& layout, uint32_t log_degree,
                              uint32_t skip_lsbs, size_t bytes_per_poly, size_t batch, hipStream_t stream) {
    const size_t total = (batch * layout.rows) << log_degree;
    if (total == 0) return hipSuccess;
    if (tile_aligned(layout, bytes, slab, log_degree) && (bytes_per_poly & 15) == 0 && batch * layout.rows <= 0x7fffffffull) {
        hipLaunchKernelGGL(deserialize_tiles_kernel, dim3(static_cast<unsigned>(batch * layout.rows)), dim3(kTileWaves * 64),
                           0, stream, bytes, slab, layout, log_degree, skip_lsbs, bytes_per_poly);
        return hipGetLastError();
    }
    if (word_aligned(layout, bytes) && (bytes_per_poly & 7) == 0) {
        hipLaunchKernelGGL(deserialize_words_kernel, dim3(grid_for(total)), dim3(256), 0, stream,
                           reinterpret_cast<const uint64_t*>(bytes), slab, layout, log_degree, skip_lsbs,
                           bytes_per_poly >> 3, total);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(deserialize_kernel, dim3(grid_for(total)), dim3(256), 0, stream, bytes, slab, layout, log_degree,
                       skip_lsbs, bytes_per_poly, total);
    return hipGetLastError();
}

}  // namespace heamd
