COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = ("the plain-slab inverse at N = 8192 signed in the stages whose twiddles are gathered only (scalar bias, no vector copy), "
               "unsigned beside wave-uniform twiddles; lane index re-derived")
EDITS = [
    ("ntt_kernels.hip", "constexpr bool kSignedInverse = !(LOGN == 13 && (SOURCE == kInverseFromSlab || SOURCE == kInverseFromSlabScaled));",
     "constexpr bool kSignedInverse = true;"),
    ("ntt_common.hpp", """        second = uniform ? split_mul_signed<true>(x - y, w.w, w.second, w.factors, neg_p, bias)""",
     """        second = uniform ? split_mul_add<true, false>(0, x + bound - y, w.w, w.second - (uint64_t(lo32(w.second) >> 31) << 32), w.factors, neg_p)"""),
    ("ntt_common.hpp", "            if (uniform && !last_stage) bias = vector_copy(bias);", "            (void)last_stage;"),
]
