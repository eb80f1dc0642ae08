COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "N = 16384 / 32768: the BEHZ primes' rows on the [0, 8p) butterflies, as until round 6 (production: the fold butterflies of the plus form)"
EDITS = [("ntt_kernels.hip", "constexpr bool kFoldPlusInterleaved = true;", "constexpr bool kFoldPlusInterleaved = false;")]
