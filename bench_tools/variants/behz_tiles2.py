COMPILE = ["behz_kernels.hip"]
DESCRIPTION = "row-fused BEHZ kernel: two LDS tiles side by side (152 KB), two rows per store / fence / load round"
EDITS = [("ntt_rows.hpp", "constexpr int kWideGroupTiles = 1;", "constexpr int kWideGroupTiles = 2;")]
