COMPILE = ["ntt_kernels.hip", "behz_kernels.hip"]
DESCRIPTION = "plain-slab inverse at N = 8192 with the partial pass on the LOW bit first (0 | 1-3 | 4-6 | 7-9 | 10-12, the fused kernels' order) instead of on the top bit last"
EDITS = [("ntt_rows.hpp", "constexpr bool kTopPartialOrder = LOGN == 13 && LOGE == 3 && INVERSE && FROM_SLAB;", "constexpr bool kTopPartialOrder = false;"),
         ("ntt_rows.hpp", "    : (INVERSE && PLAIN && LOGN == 13)                                                              ? 2", "    : false                                                                                         ? 2")]
