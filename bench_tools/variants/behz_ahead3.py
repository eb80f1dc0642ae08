COMPILE = ["behz_kernels.hip"]
DESCRIPTION = "row-fused BEHZ kernel: three twiddles in flight ahead of the butterflies"
EDITS = [("ntt_common.hpp", "constexpr int kWideGroupTwiddlesAhead = 1;", "constexpr int kWideGroupTwiddlesAhead = 3;")]
