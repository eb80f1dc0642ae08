COMPILE = ["behz_kernels.hip"]  # only the row groups of three and four read the knob
DESCRIPTION = "row-fused BEHZ kernel: two twiddles in flight ahead of the butterflies (128 registers per lane: room for them)"
EDITS = [("ntt_common.hpp", "constexpr int kWideGroupTwiddlesAhead = 1;", "constexpr int kWideGroupTwiddlesAhead = 2;")]
