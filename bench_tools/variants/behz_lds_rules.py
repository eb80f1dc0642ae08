COMPILE = ["behz_kernels.hip"]
DESCRIPTION = "row-fused BEHZ kernel: per-transpose LDS padding rules in the limb-wise inverse"
EDITS = [("ntt_rows.hpp", "constexpr bool kWideGroupPerTransposeLds = false;", "constexpr bool kWideGroupPerTransposeLds = true;")]
