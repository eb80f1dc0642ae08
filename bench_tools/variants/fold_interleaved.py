DESCRIPTION = "the interleaved sub-row transforms (N = 16384 / 32768) on the shift-folded products too (measured the same as the limb-wise ones)"
EDITS = [("ntt_kernels.hip", "constexpr bool kFoldLazyInterleaved = false;", "constexpr bool kFoldLazyInterleaved = true;")]
