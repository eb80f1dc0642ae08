COMPILE = ["behz_kernels.hip"]
DESCRIPTION = "row-fused BEHZ kernel: two twiddles ahead, the inverse passes' first ones requested before the exchange"
EDITS = [("ntt_common.hpp", "constexpr int kWideGroupTwiddlesAhead = 1;", "constexpr int kWideGroupTwiddlesAhead = 2;"),
         ("ntt_rows.hpp", "constexpr bool kWideGroupFirstTwiddleEarly = false;", "constexpr bool kWideGroupFirstTwiddleEarly = true;")]
