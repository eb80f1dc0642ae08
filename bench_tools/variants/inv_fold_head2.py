DESCRIPTION = "plain-slab inverse transform on the shift-folded products: 2 twiddles of the first pass requested before the row loads (production: three)"
EDITS = [("ntt_kernels.hip", "constexpr int kInverseHeadTwiddles = (MODE == kModeFoldLazy && SOURCE == 0) ? 3 : 1;", "constexpr int kInverseHeadTwiddles = (MODE == kModeFoldLazy && SOURCE == 0) ? 2 : 1;")]
