COMPILE = ["ntt_kernels.hip"]
DESCRIPTION = "N = 16384 / 32768 on the shift-folded products: the inverse's cross stages with three twiddles in flight as the forward transform has (production: two; three keep 20 B of scratch)"
EDITS = [("ntt_kernels.hip", "constexpr int kCrossAheadInverse = MODE == kModeFoldLazy ? 2 :", "constexpr int kCrossAheadInverse = MODE == kModeFoldLazy ? 3 :")]
