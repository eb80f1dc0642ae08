COMPILE = ["ntt_kernels.hip", "behz_kernels.hip"]  # the edited text lives in a header both include
DESCRIPTION = ("the plain-slab inverse at N = 8192 (unsigned butterflies) with the lane index re-derived per step instead of carried in a "
               "register: no scratch with the lane-major twiddle blocks (12 B with the register)")
EDITS = [
    ("ntt_rows.hpp", """template <int MODE>
__device__ __forceinline__ uint32_t step_lane(uint32_t lane) {
    if constexpr (!kLateLaneAddresses) return lane;
    else if constexpr (MODE == kModeSplitSigned) return late_lane(lane);""",
     """template <int MODE, bool INVERSE = true>
__device__ __forceinline__ uint32_t step_lane(uint32_t lane) {
    if constexpr (!kLateLaneAddresses) return lane;
    else if constexpr (INVERSE && (MODE == kModeSplitSigned || MODE == kModeSplit)) return late_lane(lane);"""),
    ("ntt_rows.hpp", "    const uint32_t tid = step_lane<MODE>(lane);\n    const TwiddleWords first = forward_first_twiddle",
     "    const uint32_t tid = step_lane<MODE, false>(lane);\n    const TwiddleWords first = forward_first_twiddle"),
]
