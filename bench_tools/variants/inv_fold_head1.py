DESCRIPTION = "plain-slab inverse transform on the shift-folded products: one twiddle of the first pass requested behind the row loads (production: three, before them)"
EDITS = [("ntt_kernels.hip", "constexpr int kInverseHeadTwiddles = (MODE == kModeFoldLazy && SOURCE == 0) ? 3 : 1;", "constexpr int kInverseHeadTwiddles = 1;")]
