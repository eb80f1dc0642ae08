COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "N = 32768: the key-switching decomposition as its own kernel + plain forward transforms"
EDITS = [("ntt_kernels.hip", "constexpr bool kFusedSpreadAt32768 = true, kFusedKeyMacAt32768 = false;", "constexpr bool kFusedSpreadAt32768 = false, kFusedKeyMacAt32768 = false;")]
