COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = ("N = 16384: the q_ks row's key inner product fused into the interleaved inverse transform, the other rows on the "
               "16-words-per-lane tile with the key switch's end in its store (the first form of round 5)")
EDITS = [("ntt_kernels.hip", "constexpr bool kFusedKeyMacAt16384 = false;", "constexpr bool kFusedKeyMacAt16384 = true;")]
