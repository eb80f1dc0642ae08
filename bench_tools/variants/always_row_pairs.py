COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "row pairs per workgroup at every launch size (production: launches of at most 4096 rows go one row per workgroup)"
EDITS = [("ntt_kernels.hip", "constexpr size_t kUngroupedBelowRows = 4096;", "constexpr size_t kUngroupedBelowRows = 0;")]
