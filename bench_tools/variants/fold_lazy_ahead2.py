DESCRIPTION = "the shift-folded butterflies with two twiddles in flight (production: three)"
EDITS = [("ntt_common.hpp", "template <int MODE>\nconstexpr int kTwiddlesAhead = MODE == kModeFoldLazy ? 3 : 1;", "template <int MODE>\nconstexpr int kTwiddlesAhead = MODE == kModeFoldLazy ? 2 : 1;")]
COMPILE = ["ntt_kernels.hip", "behz_kernels.hip"]
