DESCRIPTION = "the fold-at-2^(b+2) butterflies with one twiddle in flight like the limb-wise form (production: three -- a twiddle is 4 registers there, not 6)"
EDITS = [("ntt_common.hpp", "template <int MODE>\nconstexpr int kTwiddlesAhead = MODE == kModeFoldLazy ? 3 : 1;", "template <int MODE>\nconstexpr int kTwiddlesAhead = 1;")]
COMPILE = ["ntt_kernels.hip", "behz_kernels.hip"]
