COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "N = 32768: the key inner product fused into the interleaved inverse transform's load (four sub-rows, one workgroup per CU)"
EDITS = [("ntt_kernels.hip", "constexpr bool kFusedSpreadAt32768 = true, kFusedKeyMacAt32768 = false;", "constexpr bool kFusedSpreadAt32768 = true, kFusedKeyMacAt32768 = true;")]
