COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "launches of at most 512 rows go one row per workgroup -- one row per workgroup slot of the chip (production: 4096)"
EDITS = [("ntt_kernels.hip", "constexpr size_t kUngroupedBelowRows = 4096;", "constexpr size_t kUngroupedBelowRows = 512;")]
