COMPILE = ["ntt_kernels.hip", "poly_context.cpp"]
DESCRIPTION = ("the inverse limb-wise butterfly multiplies x + bound - y as an unsigned word (rounds 2-4) instead of the signed "
               "difference x - y: one more 64-bit addition per butterfly, plain inverse tables")
EDITS = [
    ("poly_context.cpp", "                    if (direction == 1 && p >= (static_cast<u64>(1) << 40) && p < (static_cast<u64>(1) << 55))",
     "                    if (false && direction == 1 && p >= (static_cast<u64>(1) << 40) && p < (static_cast<u64>(1) << 55))"),
    ("ntt_common.hpp", """        second = uniform ? split_mul_signed<true>(x - y, w.w, w.second, w.factors, neg_p, bias)
                         : split_mul_signed<false>(x - y, w.w, w.second, w.factors, neg_p, bias);""",
     """        (void)bias;
        second = uniform ? split_mul_add<true, false>(0, x + bound - y, w.w, w.second, w.factors, neg_p)
                         : split_mul_add<false, false>(0, x + bound - y, w.w, w.second, w.factors, neg_p);"""),
]
