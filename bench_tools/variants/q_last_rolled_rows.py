COMPILE = ["poly_kernels.hip"]
DESCRIPTION = "divideAndRoundQLast as the rolled loop over the rows, one load in flight per lane (production: the rows at compile time, all loads issued first)"
EDITS = [("poly_kernels.hip", "constexpr bool kDivideAndRoundRowsAtCompileTime = true;", "constexpr bool kDivideAndRoundRowsAtCompileTime = false;")]
