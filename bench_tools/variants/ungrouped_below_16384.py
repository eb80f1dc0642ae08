COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "launches of at most 16384 rows go one row per workgroup: the headline batch without its row pairs (production: 4096)"
EDITS = [("ntt_kernels.hip", "constexpr size_t kUngroupedBelowRows = 4096;", "constexpr size_t kUngroupedBelowRows = 16384;")]
