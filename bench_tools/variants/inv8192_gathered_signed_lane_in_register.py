COMPILE = ["ntt_kernels.hip", "behz_kernels.hip"]
DESCRIPTION = ("inv8192_gathered_signed with the lane index in a register")
EDITS = [
    ("ntt_kernels.hip", "constexpr bool kSignedInverse = !(LOGN == 13 && (SOURCE == kInverseFromSlab || SOURCE == kInverseFromSlabScaled));",
     "constexpr bool kSignedInverse = true;"),
    ("ntt_common.hpp", """        second = uniform ? split_mul_signed<true>(x - y, w.w, w.second, w.factors, neg_p, bias)""",
     """        second = uniform ? split_mul_add<true, false>(0, x + bound - y, w.w, w.second - (uint64_t(lo32(w.second) >> 31) << 32), w.factors, neg_p)"""),
    ("ntt_common.hpp", "            if (uniform && !last_stage) bias = vector_copy(bias);", "            (void)last_stage;"),
    ("ntt_rows.hpp", "    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(lane))) + within;", "    (void)within;\n    return opaque32(lane);"),
]
