COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = ("N = 16384: the key MAC as interleaved sub-rows + the separate finish kernel instead of the 16-words-per-lane tile with "
               "the key switch's end in its store")
EDITS = [("ntt_kernels.hip", "constexpr bool kInterleavedKeyMacAt16384 = false;", "constexpr bool kInterleavedKeyMacAt16384 = true;"),
         ("ntt_kernels.hip", "constexpr bool kFusedKeyMacAt16384 = false;", "constexpr bool kFusedKeyMacAt16384 = true;")]
