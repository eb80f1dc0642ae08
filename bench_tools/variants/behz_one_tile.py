COMPILE = ["behz_kernels.hip"]
DESCRIPTION = "row-fused BEHZ kernel: the four / three rows of a workgroup through ONE LDS tile in turn (as the row pairs of the plain transforms)"
EDITS = [("ntt_rows.hpp", "constexpr int kWideGroupTiles = 2;", "constexpr int kWideGroupTiles = 1;")]
