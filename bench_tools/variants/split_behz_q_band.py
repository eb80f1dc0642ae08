DESCRIPTION = "the row-fused ct x ct kernel's Q band on the limb-wise Shoup butterflies instead of the fold-at-2^(b+2) ones"
EDITS = [("behz_kernels.hip", "constexpr bool kBehzFoldLazy = true;", "constexpr bool kBehzFoldLazy = false;")]
