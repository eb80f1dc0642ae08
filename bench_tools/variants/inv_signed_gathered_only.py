COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = ("the signed difference only in the stages whose twiddles are gathered per lane; the stages with wave-uniform "
               "twiddles (constants in SGPRs, no scalar operand left for the bias) turn the table's word back and keep x + bound - y")
EDITS = [
    ("ntt_common.hpp", """        second = uniform ? split_mul_signed<true>(x - y, w.w, w.second, w.factors, neg_p, bias)""",
     """        second = uniform ? split_mul_add<true, false>(0, x + bound - y, w.w, w.second - (uint64_t(lo32(w.second) >> 31) << 32), w.factors, neg_p)"""),
]
