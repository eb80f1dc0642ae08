COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = ("N = 16384: the fused-load transforms (decomposition, lift, Q band; tensor product, key MAC) on the 16-words-per-lane "
               "tile, one workgroup per CU, with the key switch's end in the key-MAC transform's store (rounds 2-4)")
EDITS = [("ntt_kernels.hip", "constexpr bool kInterleavedFusedLoads = true;", "constexpr bool kInterleavedFusedLoads = false;")]
