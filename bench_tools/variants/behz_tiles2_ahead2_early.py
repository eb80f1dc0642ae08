COMPILE = ["behz_kernels.hip"]
DESCRIPTION = "row-fused BEHZ kernel: two LDS tiles, two twiddles ahead, early first twiddles"
EDITS = [("ntt_rows.hpp", "constexpr int kWideGroupTiles = 1;", "constexpr int kWideGroupTiles = 2;"),
         ("ntt_common.hpp", "constexpr int kWideGroupTwiddlesAhead = 1;", "constexpr int kWideGroupTwiddlesAhead = 2;"),
         ("ntt_rows.hpp", "constexpr bool kWideGroupFirstTwiddleEarly = false;", "constexpr bool kWideGroupFirstTwiddleEarly = true;")]
